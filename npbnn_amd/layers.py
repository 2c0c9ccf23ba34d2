"""Network description objects and the forward-pass call surface.

Mirrors the reference's operator interface for the hot path (np_bnn/BNN_lib.py:16-272):
``ActFun``, the output functions (``SoftMax``, ``RegressTransform``, ``RegressTransformError``,
``SoftPlus``), ``RunHiddenLayer``, ``MatrixMultiplication[D]``, ``RunPredict``, ``RunPredictInd``
and ``create_mask`` keep their names, argument meaning and results, but every numeric call goes to
the HIP kernels through the C ABI: the objects below are *descriptions* (kind tags) that the device
path dispatches on, plus thin host wrappers that move host arrays through a temporary device context.
"""
import numpy as np

from . import _capi as capi

_ACT_KIND = {"relu": capi.ACT_RELU, "leaky": capi.ACT_LEAKY, "swish": capi.ACT_SWISH, "tanh": capi.ACT_TANH}


# ---------------------------------------------------------------------------------------------
# activation functions
# ---------------------------------------------------------------------------------------------
class _Activation:
    """Callable tag for one activation kind; calling it on a host array runs the device kernel."""

    def __init__(self, name):
        self.name = name
        self.kind = _ACT_KIND[name]

    def __call__(self, z, prm=0):
        from . import device_ops
        return device_ops.activation(z, self.kind, prm)

    def __repr__(self):
        return "<activation %s>" % self.name


relu_f = _Activation("relu")            # reference: BNN_lib.py:50-52
leaky_relu_f = _Activation("leaky")     # reference: BNN_lib.py:54-56
swish_f = _Activation("swish")          # reference: BNN_lib.py:58-61
tanh_f = _Activation("tanh")            # reference: BNN_lib.py:63-66  (1 - 2 / (exp(2z) + 1))


class ActFun:
    """Activation of the hidden layers (reference: BNN_lib.py:68-94).

    Selection follows the reference's chain of independent tests: "ReLU" -> relu; "genReLU" or
    ``trainable`` -> leaky relu; "swish"; "tanh" (a later match overrides an earlier one).  The
    per-layer slope ``_prm[layer_n]`` is used only when ``fun == "genReLU"``; otherwise the slope is 0.
    """

    def __init__(self, fun='ReLU', prm=np.zeros(1), trainable=False):
        self._prm = prm
        self._acc_prm = prm
        self._trainable = trainable
        self._function = fun
        if fun == "ReLU":
            self.activate = relu_f
        if fun == "genReLU" or trainable is True:
            self.activate = leaky_relu_f
        if fun == "swish":
            self.activate = swish_f
        if fun == "tanh":
            self.activate = tanh_f

    # -- description consumed by the device path --
    def device_kind(self):
        return self.activate.kind

    def device_slopes(self, n_hidden, accepted=False):
        """Slopes of the hidden layers as the device takes them (None: the activation has none).  ``accepted``: the slopes of
        the chain's accepted state rather than the last proposed ones - they differ while slopes are trainable, because a
        proposal is installed in ``_prm`` whether it is accepted or not (np_bnn/BNN_env.py:421 against :502-503)."""
        if self._function != "genReLU":
            return None
        prm = np.asarray(self._acc_prm if (accepted and self._trainable) else self._prm, dtype=float)
        return np.array([prm[i] for i in range(n_hidden)], dtype=float)

    # -- reference API --
    def eval(self, z, layer_n):
        if self._function == "genReLU":
            return self.activate(z, self._prm[layer_n])
        return self.activate(z, 0)

    def reset_prm(self, prm):
        self._prm = prm

    def reset_accepted_prm(self):
        self._acc_prm = self._prm + 0


# ---------------------------------------------------------------------------------------------
# output functions
# ---------------------------------------------------------------------------------------------
class _OutputFn:
    def __init__(self, name, kind):
        self.__name__ = name
        self.kind = kind

    def __call__(self, z, ind=None):
        from . import device_ops
        return device_ops.output_fn(z, self.kind, ind)

    def __repr__(self):
        return "<output function %s>" % self.__name__

    def __reduce__(self):       # pickles as a reference to the module-level singleton
        return self.__name__


SoftMax = _OutputFn("SoftMax", capi.OUT_SOFTMAX)                                   # BNN_lib.py:166-168
RegressTransform = _OutputFn("RegressTransform", capi.OUT_IDENTITY)                # BNN_lib.py:174-175
RegressTransformError = _OutputFn("RegressTransformError", capi.OUT_SOFTPLUS_HALF)  # BNN_lib.py:177-182


def SoftPlus(z):
    """log(1 + exp(z)) without overflow (reference: BNN_lib.py:170-172)."""
    from . import device_ops
    return device_ops.softplus(z)


def output_kind(fn):
    """Device kind of a built-in output function, or None for a user callable (the device then
    returns the last layer's values and the callable runs on the host)."""
    return getattr(fn, "kind", None) if isinstance(fn, _OutputFn) else None


# ---------------------------------------------------------------------------------------------
# forward pass on host arrays (one temporary device context per call)
# ---------------------------------------------------------------------------------------------
def _apply_transform(data, data_transform):
    return data if data_transform is None else data_transform.transform(data)


def _forward(data, weights, actFun, out_fn, final_activation=False, layer_offset=0):
    from .backend import HipContext
    data = np.asarray(data)
    kind = output_kind(out_fn)
    ctx = HipContext()
    try:
        ctx.set_data(data)
        act_kind = actFun.device_kind() if actFun else capi.ACT_RELU
        ctx.set_arch_from_weights(weights, data.shape[1], act_kind,
                                  capi.OUT_IDENTITY if kind is None else kind, capi.LIK_NONE,
                                  final_activation=final_activation)
        slopes = None
        if actFun:
            n_act = len(weights) - (0 if final_activation else 1)
            slopes = actFun.device_slopes(n_act + layer_offset)
            if slopes is not None:
                slopes = slopes[layer_offset:]
        y = ctx.predict(weights, act_prm=slopes, apply_out_fn=kind is not None)
    finally:
        ctx.close()
    if kind is None and out_fn is not None:
        y = out_fn(y)
    return y


def MatrixMultiplicationD(x1, x2):
    """x1 (N x in) times x2 (out x in[+1]); when x2 has in+1 columns, column 0 is the bias
    (reference: BNN_lib.py:154-162)."""
    return _forward(x1, [np.asarray(x2)], None, RegressTransform)


def MatrixMultiplication(x1, x2):
    """Same result as MatrixMultiplicationD (reference: BNN_lib.py:146-152, an einsum variant)."""
    return MatrixMultiplicationD(x1, x2)


def RunHiddenLayer(z0, w01, actFun, layer_n, data_transform=None):
    """One layer: optional column transform, GEMM (+ bias), activation unless ``actFun`` is False
    (reference: BNN_lib.py:184-193)."""
    z0 = _apply_transform(z0, data_transform)
    if actFun:
        return _forward(z0, [np.asarray(w01)], actFun, RegressTransform, final_activation=True,
                        layer_offset=layer_n)
    return _forward(z0, [np.asarray(w01)], None, RegressTransform)


def RunPredict(data, weights, actFun, output_act_fun, data_transform=None):
    """Full forward pass returning the predictions (reference: BNN_lib.py:245-256)."""
    return _forward(_apply_transform(data, data_transform), list(weights), actFun, output_act_fun)


def RunPredictInd(data, weights, ind, actFun, output_act_fun, data_transform=None):
    """Forward pass with the first layer's weights multiplied by the indicators
    (reference: BNN_lib.py:258-272)."""
    w = [np.asarray(weights[0]) * ind] + [np.asarray(wi) for wi in weights[1:]]
    return _forward(_apply_transform(data, data_transform), w, actFun, output_act_fun)


# ---------------------------------------------------------------------------------------------
# block-sparse masks
# ---------------------------------------------------------------------------------------------
def create_mask(w_layers, indx_input_list, nodes_per_feature_list):
    """0/1 masks wiring groups of input columns to blocks of nodes (reference: BNN_lib.py:16-47).

    For layer l, ``indx_input_list[l]`` gives a group id per input column (consecutive equal ids form
    a group; empty = fully connected) and ``nodes_per_feature_list[l][g]`` the number of nodes of
    group g.  Groups own consecutive blocks of rows.
    """
    masks = []
    for layer, w in enumerate(w_layers):
        group_of_col = indx_input_list[layer]
        if len(group_of_col) == 0:
            masks.append(np.ones(w.shape))
            continue
        sizes = nodes_per_feature_list[layer]
        m = np.zeros(w.shape)
        first_row, used_rows, g = 0, 0, 0
        for col, gid in enumerate(group_of_col):
            if col > 0 and gid != group_of_col[col - 1]:
                g += 1
                first_row = used_rows
            m[first_row:first_row + sizes[g], col] = 1
            used_rows = first_row + sizes[g]
        masks.append(m)
    return masks
