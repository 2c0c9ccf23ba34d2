"""Run one of the REFERENCE's own driver scripts, unchanged, against this package (CPU; test infrastructure).

    python tests/run_reference_driver.py /root/reference/bnn_classify.py

``np_bnn`` is made to resolve to ``npbnn_amd`` before the script is executed from where it lies (nothing of it is copied); the
device is absent here, so the package's device seams are served by the float64 oracle: the sampler's backend
(npbnn_amd.sampler._make_backend), the posterior predictor (npbnn_amd.posterior._SamplePredictor) and the two reductions behind
the stand-alone accuracy helpers (npbnn_amd.device_ops._confusion / _sse) and the stand-alone forward pass on host arrays
(npbnn_amd.layers._forward: RunPredict called directly, as bnn_regress.py does).  Everything else the
script touches - get_data, npBNN, ActFun, MCMC, postLogger, run_mcmc, predictBNN, feature_importance, npBNN(pickle_file=...) -
is the product's host code."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import npbnn_amd  # noqa: E402
import oracle as orc  # noqa: E402
from npbnn_amd import posterior, sampler  # noqa: E402
from npbnn_amd.layers import output_kind  # noqa: E402
from oracle_backend import OracleChainBackend  # noqa: E402


class OraclePredictor:
    """Stand-in for posterior._SamplePredictor: every stored sample's predictions by the oracle's forward pass."""

    def __init__(self, n_features, post_samples, actFun, output_act_fun):
        self.samples, self.act, self.kind, self.out_fn = post_samples, actFun, output_kind(output_act_fun), output_act_fun

    def predict(self, features):
        out_fn = {0: orc.out_softmax, 1: orc.out_identity, 2: orc.out_regress_error}.get(self.kind, self.out_fn)
        ys = []
        for s in self.samples:
            act = orc.Act(self.act._function, prm=np.asarray(s["alphas"], dtype=float), trainable=self.act._trainable)
            ys.append(orc.forward(np.asarray(features, dtype=float), s["weights"], act, out_fn))
        return np.array(ys)

    def close(self):
        pass


def _confusion(y, lab):
    """Stand-in for device_ops._confusion (npbnn_op_confusion): [true, predicted] counts and predicted-class counts."""
    y = np.asarray(y, dtype=float)
    pred = np.argmax(y, axis=1)
    counts = np.bincount(pred, minlength=y.shape[1]).astype(np.int64)
    if lab is None:
        return None, counts
    return orc.confusion_counts(y, np.asarray(lab, dtype=np.int64), n_classes=y.shape[1]), counts


def _sse(y, lab, link, first_col_only):
    """Stand-in for device_ops._sse (npbnn_op_sse): per-column sums of squared residuals under the identity / exp / 10^ link."""
    y, t = np.asarray(y, dtype=float), np.asarray(lab, dtype=float)
    t = t.reshape(-1, 1) if t.ndim == 1 else t
    if first_col_only:
        t = t[:, :1]
    pred = y[:, :t.shape[1]]
    pred = np.exp(pred) if link == 1 else (10.0 ** pred if link == 2 else pred)
    return np.sum((pred - t) ** 2, axis=0), y.shape[0]


def _forward(data, weights, actFun, out_fn, final_activation=False, layer_offset=0):
    """Stand-in for layers._forward (the stand-alone RunPredict / RunHiddenLayer / MatrixMultiplication on host arrays, one temporary
    device context per call): the oracle's layer loop."""
    z = np.asarray(data, dtype=float)
    n = len(weights)
    act = None
    if actFun:
        act = orc.Act(actFun._function, prm=np.asarray(actFun._prm, dtype=float) if np.ndim(actFun._prm) else None, trainable=actFun._trainable)
    for i, w in enumerate(weights):
        z = orc.dense(z, np.asarray(w, dtype=float))
        if act is not None and (i + 1 < n or final_activation):
            z = orc.activate(z, act, i + layer_offset)
    return z if out_fn is None else out_fn(z)


def _backend(bnn, likelihood_f):
    kind = output_kind(bnn._output_act_fun)
    return OracleChainBackend(bnn, 0 if kind is None else kind)


if __name__ == "__main__":
    from npbnn_amd import device_ops, layers
    layers._forward = _forward
    sampler._make_backend = _backend
    posterior._SamplePredictor = OraclePredictor
    device_ops._confusion, device_ops._sse = _confusion, _sse
    sys.modules["np_bnn"] = npbnn_amd
    runpy.run_path(sys.argv[1], run_name="__main__")
