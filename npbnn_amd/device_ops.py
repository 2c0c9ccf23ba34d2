"""Stand-alone device operators on host arrays (npbnn_op_* in the C ABI).

These back the reference's small helper callables when user code calls them directly — ``bn.tanh_f(z, 0)``,
``bn.SoftMax(z)``, ``bn.calc_likelihood(y, labels, ids)``, ``bn.CalcAccuracy(y, labels)`` … — so that even the slow,
matrix-in / matrix-out path runs on the GPU.  The sampler itself never takes this route for the built-in functions
(it fuses them into the evaluation kernel)."""
import ctypes as C

import numpy as np

from . import _capi as capi
from .backend import default_device

_I64P = C.POINTER(C.c_int64)


def _lib():
    lib = capi.load_library()
    n = C.c_int(0)
    if lib.npbnn_device_count(C.byref(n)) != 0 or n.value < 1:
        raise capi.BackendUnavailable("no HIP device visible: npbnn_amd operators need an MI355X (there is no CPU fallback)")
    return lib, default_device() % n.value


def _chk(lib, rc):
    capi.check(lib, None, rc)


def activation(z, kind, prm=0):
    """relu / leaky relu mutate their argument like the reference (BNN_lib.py:50-56); swish / tanh return a new array."""
    lib, dev = _lib()
    z = np.asarray(z)
    buf = np.ascontiguousarray(z, dtype=np.float64).copy()
    _chk(lib, lib.npbnn_op_activation(dev, int(kind), float(prm), capi.dptr(buf), buf.size))
    if kind in (capi.ACT_RELU, capi.ACT_LEAKY) and isinstance(z, np.ndarray) and z.dtype == np.float64:
        z[...] = buf.reshape(z.shape)
        return z
    return buf.reshape(z.shape)


def softplus(z):
    lib, dev = _lib()
    buf = np.ascontiguousarray(z, dtype=np.float64).copy()
    _chk(lib, lib.npbnn_op_activation(dev, 4, 0.0, capi.dptr(buf), buf.size))
    return buf.reshape(np.shape(z))


def output_fn(z, kind, ind=None):
    """SoftMax returns a new matrix; RegressTransform returns its argument; RegressTransformError rewrites the
    columns >= ind in place (BNN_lib.py:166-182)."""
    if kind == capi.OUT_IDENTITY:
        return z
    lib, dev = _lib()
    z = np.asarray(z)
    buf = np.ascontiguousarray(z, dtype=np.float64).copy()
    rows, cols = buf.shape
    _chk(lib, lib.npbnn_op_output(dev, int(kind), capi.dptr(buf), rows, cols, -1 if ind is None else int(ind)))
    if kind == capi.OUT_SOFTPLUS_HALF and z.dtype == np.float64:
        z[...] = buf
        return z
    return buf


def likelihood(kind, prediction, labels, class_weight=None, instance_weight=None, lik_temp=1, sig2=None):
    lib, dev = _lib()
    pred = capi.as_f64(prediction)
    rows, cols = pred.shape
    out = C.c_double(0)
    lab = tg = None
    k = 0
    cw = None if class_weight is None or len(class_weight) == 0 else capi.as_f64(class_weight)
    iw = None if instance_weight is None else capi.as_f64(instance_weight)
    sg = None
    if kind == capi.LIK_CATEGORICAL:
        if cw is not None and iw is not None:
            # upstream sums a vector over axis 1 here and fails (BNN_lib.py:105); keep the failure
            raise np.exceptions.AxisError("axis 1 is out of bounds for array of dimension 1")
        lab = np.ascontiguousarray(labels, dtype=np.int64)
    else:
        tg = capi.as_f64(labels)
        if tg.ndim == 1:
            tg = tg.reshape(-1, 1)
        k = tg.shape[1]
        if kind == capi.LIK_GAUSS:
            sg = capi.as_f64(np.broadcast_to(1 if sig2 is None else sig2, (k,)))
    _chk(lib, lib.npbnn_op_likelihood(dev, int(kind), capi.dptr(pred), rows, cols,
                                      None if lab is None else lab.ctypes.data_as(_I64P), capi.dptr(tg), k, capi.dptr(iw),
                                      capi.dptr(cw), 0 if cw is None else cw.shape[0], float(lik_temp), capi.dptr(sg),
                                      C.byref(out)))
    return out.value


def _confusion(y, lab):
    lib, dev = _lib()
    pred = capi.as_f64(y)
    rows, cols = pred.shape
    counts = np.zeros(cols, dtype=np.int64)
    conf = None
    labp = None
    if lab is not None:
        lab = np.ascontiguousarray(lab, dtype=np.int64)
        labp = lab.ctypes.data_as(_I64P)
        conf = np.zeros((cols, cols), dtype=np.int64)
    _chk(lib, lib.npbnn_op_confusion(dev, capi.dptr(pred), rows, cols, labp,
                                     None if conf is None else conf.ctypes.data_as(_I64P), counts.ctypes.data_as(_I64P)))
    return conf, counts


def _sse(y, lab, link, first_col_only):
    lib, dev = _lib()
    pred = capi.as_f64(y)
    tg = capi.as_f64(lab)
    if tg.ndim == 1:
        tg = tg.reshape(-1, 1)
    if first_col_only:
        tg = np.ascontiguousarray(tg[:, :1])
    k = tg.shape[1]
    out = np.zeros(k)
    _chk(lib, lib.npbnn_op_sse(dev, capi.dptr(pred), capi.dptr(tg), pred.shape[0], pred.shape[1], k, link, capi.dptr(out)))
    return out, pred.shape[0]


def statistic(kind, y, lab):
    """The accuracy helpers of the reference on an explicit prediction matrix (BNN_lib.py:195-233, BNN_lik.py:81-99)."""
    if kind == "acc":
        y = np.asarray(y)
        if y.ndim == 3:           # one accuracy per posterior sample (BNN_lib.py:204-205)
            return np.array([statistic("acc", yi, lab) for yi in y])
        conf, _ = _confusion(y, lab)
        return np.trace(conf) / len(lab)
    if kind == "label_acc":
        conf, _ = _confusion(y, lab)
        present = np.unique(np.asarray(lab, dtype=np.int64))
        return np.array([conf[c, c] / conf[c].sum() for c in present])
    if kind == "label_freq":
        _, counts = _confusion(y, None)
        return counts / np.shape(y)[0]
    if kind in ("mse", "label_mse"):
        sse, n = _sse(y, lab, 0, False)
        return float(np.sum(sse) / (n * len(sse))) if kind == "mse" else sse / n
    if kind == "mse_exp":
        sse, n = _sse(y, lab, 1, False)
        return float(np.sum(sse) / (n * len(sse)))
    if kind in ("mse_exp_col0", "mse_pow10_col0"):
        sse, n = _sse(y, lab, 1 if kind == "mse_exp_col0" else 2, True)
        return float(sse[0] / n)
    raise ValueError(kind)
