"""TEST-ONLY evaluation backend: serves the product's sampler from the CPU oracle so that the host
logic (proposal stream, prior, accept/reject, adaptation, MC3 swaps) can be exercised without a GPU.
Lives under tests/ on purpose - the product ships the HIP backend only."""
import numpy as np

import oracle as orc


class OracleBackend:
    fused_likelihood = True

    def __init__(self, bnn, out_kind=0):
        self.bnn = bnn
        self.out_kind = out_kind
        self.n_eval = 0

    def _act(self, slopes):
        fun = self.bnn._act_fun._function
        trainable = self.bnn._act_fun._trainable
        return orc.Act(fun, prm=slopes if slopes is not None else np.zeros(8), trainable=trainable)

    def _data(self, which):
        return (self.bnn._data, self.bnn._labels) if which == 0 else (self.bnn._test_data, self.bnn._test_labels)

    def _out_fn(self):
        return {0: orc.out_softmax, 1: orc.out_identity, 2: orc.out_regress_error}[self.out_kind]

    def predict(self, weights, slopes=None, col_override=None, which=0, apply_out_fn=True):
        x, _ = self._data(which)
        co = None
        if col_override is not None:
            co = ((~np.isnan(col_override)).astype(int) * 0 + np.isnan(col_override).astype(int), np.nan_to_num(col_override))
        z = orc.forward_logits(x, weights, self._act(slopes), col_override=co)
        return self._out_fn()(z) if apply_out_fn else z

    def evaluate(self, weights, slopes=None, col_override=None, lik_temp=1.0, sigma=None, which=0, want_confusion=False):
        self.n_eval += 1
        x, lab = self._data(which)
        y = self.predict(weights, slopes, col_override, which)
        out = dict(confusion=None, sigma=None, sum_r=None, sum_r2=None, n_rows=len(x))
        if self.bnn._estimation_mode == "classification":
            iw = self.bnn._instance_weights if which == 0 else None
            cw = self.bnn._class_w if which == 0 else []
            with np.errstate(divide="ignore"):
                out["loglik"] = orc.lik_categorical(y, lab, np.arange(len(x)), class_weight=cw, instance_weight=iw,
                                                    lik_temp=lik_temp)
            if want_confusion:
                out["confusion"] = orc.confusion_counts(y, lab)
        else:
            r = lab - y[:, :lab.shape[1]]
            sig = np.std(y - lab, axis=0) if sigma is None else np.asarray(sigma, dtype=float)
            out["loglik"] = orc.lik_gaussian(y, lab, None, lik_temp=lik_temp, sig2=sig)
            out["sigma"] = sig
            out["sum_r"] = r.sum(axis=0)
            out["sum_r2"] = (r * r).sum(axis=0)
        return out
