"""``npBNN`` - the model container: weights, prior settings and handles to the data.

This is the host-side object the sampler, the loggers and user scripts talk to.  Its constructor signature, method names
and attribute names are the reference's call surface (np_bnn/BNN_env.py:19-270; SURVEY.md section 8b lists the attributes
other code reads), so code written against ``np_bnn`` finds what it expects.  How the object is put together is this
package's own: the per-mode output layout comes from a table, the construction is split into small steps, and nothing
numeric on the sampling path lives here - the sampler hands ``_w_layers`` to the HIP backend, which keeps its own
resident copy of ``_data``.
"""
import copy
from collections import namedtuple

import numpy as np

from .layers import ActFun, RegressTransform, SoftMax
from .proposals import (GibbsSampleNormStdGamma2D, GibbsSampleNormStdGammaONE,
                        GibbsSampleNormStdGammaVector, init_weight_prm)

_HALF_LOG_2PI = 0.5 * np.log(2 * np.pi)
_DEVICE_HANDLE = "_npbnn_backend"          # attribute holding the resident device context; never copied or pickled

# prior kinds (the constructor's prior_f): densities the device chain and calc_prior know in closed form
UNIFORM, NORMAL, CAUCHY, LAPLACE = 0, 1, 2, 3

# Output layout per estimation mode (np_bnn/BNN_env.py:51-74):
#   outputs(k, size_output)   nodes of the last layer, k = target columns
#   params(k, size_output)    likelihood parameters predicted per row
#   integer_labels            labels are class indices
#   unit_error                the model carries one error parameter per target column, initially 1
_Mode = namedtuple("_Mode", "outputs params integer_labels unit_error")
_MODES = {
    "classification": _Mode(lambda k, s: s, lambda k, s: s, True, False),
    "regression": _Mode(lambda k, s: k, lambda k, s: k, False, True),
    "regression-error": _Mode(lambda k, s: 2 * k, lambda k, s: k, False, False),
    "custom": _Mode(lambda k, s: s, lambda k, s: s, False, False),
}


class data_transform_obj():
    """Feature-indicator transform (np_bnn/BNN_env.py:9-17): a column whose indicator is 0 reads as its mean.  The device
    never materialises the transformed matrix - ``column_override`` is what the layer-0 bias fold takes."""

    def __init__(self, feature_indicators, feature_means):
        self.feature_indicators = feature_indicators
        self.feature_means = feature_means

    def _switched_off(self):
        return np.asarray(self.feature_indicators) == 0

    def transform(self, data):
        return np.where(self._switched_off()[None, :], np.asarray(self.feature_means)[None, :], data)

    def column_override(self):
        """Per column: NaN = keep the data, a number = the constant that replaces the column."""
        return np.where(self._switched_off(), self.feature_means, np.nan).astype(float)


def _log_density(kind, w, scale):
    """log of the zero-centred prior density at ``w`` (closed forms of the scipy.stats logpdf calls the reference makes,
    np_bnn/BNN_env.py:135-150)."""
    z = w / scale
    if kind == CAUCHY:
        return -np.log(np.pi * scale * (1 + z * z))
    if kind == LAPLACE:
        return -np.log(2 * scale) - np.abs(z)
    return -0.5 * z * z - np.log(scale) - _HALF_LOG_2PI


def _balanced_class_weights(labels):
    """Inverse class frequencies, normalised to mean 1 (np_bnn/BNN_env.py:98-102)."""
    counts = np.unique(labels, return_counts=True)[1].astype(float)
    inv = counts.max() / counts
    return inv / inv.mean()


def _count_parameters(weights, act_fun, n_layers, nonzero_only=False):
    n = sum(int(np.count_nonzero(w)) if nonzero_only else int(np.size(w)) for w in weights)
    return n + (n_layers if act_fun._trainable else 0)


class npBNN():
    def __init__(self, dat, n_nodes=[50, 5],
                 use_bias_node=1, init_std=0.1, p_scale=1, prior_ind1=0.5,
                 prior_f=1, hyper_p=0, freq_indicator=0, w_bound=np.inf,
                 pickle_file="", seed=1234, use_class_weights=0, actFun=None, init_weights=None,
                 estimation_mode="classification",
                 instance_weights=None,
                 empirical_error=False,
                 size_output=None,
                 output_act_fun=None,
                 feature_indicators=None,
                 ):
        """``dat``: dict with ``data``, ``labels``, ``test_data``, ``test_labels``.  ``prior_f``: 0 uniform on
        [-p_scale, p_scale], 1 normal, 2 Cauchy, 3 Laplace (scale ``p_scale``).  ``hyper_p``: 0 fixed scales, 1 one per
        layer, 2 one per input node, 3 one per weight.  ``freq_indicator``: share of iterations that propose a change of
        the layer-0 weight indicators.  ``estimation_mode``: classification, regression, regression-error or custom."""
        if estimation_mode not in _MODES:
            raise ValueError("estimation_mode %r; expected one of %s" % (estimation_mode, sorted(_MODES)))
        self._estimation_mode = estimation_mode
        self._seed = seed
        self._attach_data(dat)
        self._describe_outputs(size_output, output_act_fun)
        self._architecture(n_nodes, use_bias_node, init_std)
        self._prior_settings(prior_f, p_scale, prior_ind1, hyper_p, freq_indicator, w_bound)
        self._class_w = _balanced_class_weights(self._labels) if use_class_weights else []
        self._instance_weights = instance_weights
        self._empirical_error = empirical_error
        self._act_fun = ActFun() if actFun is None else actFun
        self._starting_point(init_weights, pickle_file)
        self._mask = None
        self._feature_switches(feature_indicators)
        self._n_params = _count_parameters(self._w_layers, self._act_fun, self._n_layers)
        self._announce()

    # ---- construction steps ------------------------------------------------------------------
    def _attach_data(self, dat):
        as_classes = _MODES[self._estimation_mode].integer_labels
        self._data = dat['data']
        self._test_data = dat['test_data']
        self._labels = dat['labels'].astype(int) if as_classes else dat['labels']
        held_out = dat['test_labels']
        if len(held_out) == 0:
            self._test_labels = []
        else:
            self._test_labels = held_out.astype(int) if as_classes else held_out
        self._n_samples, self._n_features = self._data.shape
        self._sample_id = np.arange(self._n_samples)

    def _describe_outputs(self, size_output, output_act_fun):
        mode = _MODES[self._estimation_mode]
        if mode.integer_labels:
            k, declared = None, len(np.unique(self._labels))
            self._output_act_fun = SoftMax
        else:
            k = self._labels.shape[1] if np.ndim(self._labels) > 1 else None
            declared = size_output
            self._output_act_fun = RegressTransform if output_act_fun is None else output_act_fun
        self._size_output = mode.outputs(k, declared)
        self._n_output_prm = mode.params(k, declared)
        self._error_prm = np.ones(self._size_output) if mode.unit_error else []

    def _architecture(self, n_nodes, use_bias_node, init_std):
        self._n_nodes = list(n_nodes) if np.iterable(n_nodes) else [n_nodes]
        self._n_layers = len(self._n_nodes) + 1
        self._use_bias_node = use_bias_node
        self._init_std = init_std

    def _prior_settings(self, prior_f, p_scale, prior_ind1, hyper_p, freq_indicator, w_bound):
        self._prior = prior_f
        self._p_scale = p_scale
        self._prior_ind1 = prior_ind1
        self._hyper_p = hyper_p
        self._freq_indicator = freq_indicator
        # a uniform prior is a pair of reflecting walls at +-p_scale
        self._w_bound = p_scale if prior_f == UNIFORM else w_bound
        self._prior_scale = np.full(self._n_layers, p_scale, dtype=float)

    def _starting_point(self, init_weights, pickle_file):
        """Weights of the first state: given, read from a checkpoint's last posterior sample, or random."""
        sample = None
        if init_weights is not None:
            weights = init_weights
        elif pickle_file:
            from .files import load_obj
            sample = load_obj(pickle_file)[2]._post_weight_samples[-1]
            weights = sample['weights']
        else:
            # (upstream draws with a standard deviation of 0.1 whatever init_std says, np_bnn/BNN_env.py:111-115; so do we)
            weights = init_weight_prm(self._n_nodes, self._n_features, self._size_output, init_std=0.1,
                                      bias_node=self._use_bias_node)
        self._w_layers = weights
        self._indicators = np.ones(weights[0].shape)
        if sample is not None and self._act_fun._trainable:
            self._act_fun.reset_prm(sample['alphas'])

    def _feature_switches(self, wanted):
        if wanted:
            self._feature_indicators = np.ones(self._n_features, dtype=int)
            self._feature_means = self.get_feature_mean()
        else:
            self._feature_indicators = None
            self._feature_means = None

    def _announce(self):
        held_out = self._test_data.shape[0] if len(self._test_data) > 0 else 0
        print("npBNN: %d training rows (%d held out), %d features, %d parameters, layers %s"
              % (self._n_samples, held_out, self._n_features, self._n_params,
                 " ".join("%dx%d" % w.shape for w in self._w_layers)))
        if len(self._class_w):
            print("class weights:", self._class_w)

    # ---- prior (np_bnn/BNN_env.py:180-194) ---------------------------------------------------
    def _prior_kind(self):
        """Density family of the weight prior; anything that is not uniform / Cauchy / Laplace is the normal."""
        return self._prior if self._prior in (NORMAL, CAUCHY, LAPLACE) else NORMAL

    def calc_prior(self, w=0, ind=[]):
        weights = self._w_layers if (isinstance(w, int) and w == 0) else w
        indicators = self._indicators if len(ind) == 0 else ind
        total = 0
        if self._prior != UNIFORM:
            kind = self._prior_kind()
            total = sum(np.sum(_log_density(kind, layer, scale)) for layer, scale in zip(weights, self._prior_scale))
        if self._freq_indicator:
            n_on = np.sum(indicators)
            total = total + (n_on * np.log(self._prior_ind1) + (self._indicators.size - n_on) * np.log(1 - self._prior_ind1))
        return total

    _GIBBS = {1: lambda w: GibbsSampleNormStdGammaVector(w.flatten()), 2: GibbsSampleNormStdGamma2D, 3: GibbsSampleNormStdGammaONE}

    def sample_prior_scale(self):
        """Gibbs draw of the prior scales given the weights (np_bnn/BNN_env.py:196-221): one per layer, per input node or
        per weight, as ``hyper_p`` says.  Only defined for the normal prior."""
        if self._prior != NORMAL:
            print("Hyper-priors available only for Normal priors.")
            quit()
        draw = self._GIBBS.get(self._hyper_p)
        if draw is not None:
            self._prior_scale = [draw(layer) for layer in self._w_layers]

    def sample_from_prior(self, reset_weights=True):
        """Weights drawn from the prior (np_bnn/BNN_env.py:223-240).  Upstream returns after the FIRST layer for the Cauchy,
        Laplace and unknown priors - callers see one array there, and so they do here."""
        drawn = []
        for layer, scale in zip(self._w_layers, self._prior_scale):
            shape = layer.shape
            if self._prior == UNIFORM:
                drawn.append(np.random.uniform(-self._w_bound, self._w_bound, shape))
            elif self._prior == NORMAL:
                drawn.append(np.random.normal(0, scale, shape))
            elif self._prior == CAUCHY:
                return np.random.standard_cauchy(shape) * scale
            elif self._prior == LAPLACE:
                return np.random.laplace(0, scale=scale, size=shape)
            else:
                return np.random.standard_normal(shape)
        if not reset_weights:
            return drawn
        self.reset_weights(drawn)

    # ---- state setters the sampler uses (np_bnn/BNN_env.py:244-270) --------------------------
    def reset_weights(self, w):
        self._w_layers = w

    def reset_indicators(self, ind):
        self._indicators = ind

    def reset_error_prm(self, p):
        self._error_prm = p

    def reset_seed(self, seed):
        self._seed = seed

    def update_data(self, data_dict):
        for attr, key in (("_data", 'data'), ("_labels", 'labels'), ("_test_data", 'test_data'), ("_test_labels", 'test_labels')):
            setattr(self, attr, data_dict[key])
        self.__dict__.pop(_DEVICE_HANDLE, None)      # the resident device copy no longer matches

    def apply_mask(self, m=None):
        """Multiply every layer by its 0/1 mask (given now, or the one stored earlier) and report what is left."""
        if m is not None:
            self._mask = m
        self._w_layers = [w * keep for w, keep in zip(self._w_layers, self._mask)]
        print("npBNN: %d parameters after masking, layers %s"
              % (_count_parameters(self._w_layers, self._act_fun, self._n_layers, nonzero_only=True),
                 " ".join("%dx%d" % w.shape for w in self._w_layers)))

    def get_feature_mean(self):
        return np.mean(self._data, axis=0)

    # ---- the device handle stays with the object it was made for -----------------------------
    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k != _DEVICE_HANDLE}

    def __deepcopy__(self, memo):
        twin = self.__class__.__new__(self.__class__)
        memo[id(self)] = twin
        twin.__dict__.update((k, copy.deepcopy(v, memo)) for k, v in self.__dict__.items() if k != _DEVICE_HANDLE)
        return twin
