"""``npBNN`` — the model container (weights, priors, data handles).

Host-side mirror of the reference's model object (np_bnn/BNN_env.py:9-270): same constructor,
methods and public attributes, so sampler code, loggers and pickles written against the reference
keep working.  The numerics of the hot path do not live here: the sampler hands this object's
weights to the HIP backend.
"""
import numpy as np

from .layers import ActFun, RegressTransform, SoftMax
from .proposals import (GibbsSampleNormStdGamma2D, GibbsSampleNormStdGammaONE,
                        GibbsSampleNormStdGammaVector, init_weight_prm)

_LOG_SQRT_2PI = 0.5 * np.log(2 * np.pi)


class data_transform_obj():
    """Replace the feature columns whose indicator is 0 by the column mean
    (reference: BNN_env.py:9-17).  On the device this is folded into the layer-0 bias."""

    def __init__(self, feature_indicators, feature_means):
        self.feature_indicators = feature_indicators
        self.feature_means = feature_means

    def transform(self, data):
        out = data + 0
        off = self.feature_indicators == 0
        out[:, off] = self.feature_means[off]
        return out

    def column_override(self):
        """NaN where a column is kept, the replacement constant where it is overridden."""
        return np.where(self.feature_indicators == 0, self.feature_means, np.nan).astype(float)


def _log_density(kind, w, scale):
    """Elementwise log prior density, closed forms of scipy.stats.{norm,cauchy,laplace}.logpdf(w, 0, scale)."""
    if kind == 2:
        return -np.log(np.pi * scale * (1 + (w / scale) ** 2))
    if kind == 3:
        return -np.log(2 * scale) - np.abs(w) / scale
    return -0.5 * (w / scale) ** 2 - np.log(scale) - _LOG_SQRT_2PI


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
        """Arguments as in the reference (BNN_env.py:20-34): ``dat`` is a dict with data, labels,
        test_data, test_labels; prior_f 0 uniform / 1 normal / 2 Cauchy / 3 Laplace; hyper_p 0-3;
        estimation_mode classification / regression / regression-error / custom."""
        if actFun is None:
            actFun = ActFun()
        self._seed = seed
        self._data = dat['data']
        classification = estimation_mode == "classification"
        self._labels = dat['labels'].astype(int) if classification else dat['labels']
        self._test_data = dat['test_data']
        test_labels = dat['test_labels']
        if len(test_labels) > 0:
            self._test_labels = test_labels.astype(int) if classification else test_labels
        else:
            self._test_labels = []

        self._error_prm = []
        if classification:
            self._size_output = len(np.unique(self._labels))
            self._n_output_prm = self._size_output
            self._output_act_fun = SoftMax
        else:
            self._output_act_fun = RegressTransform if output_act_fun is None else output_act_fun
            if estimation_mode == "regression":
                self._size_output = self._labels.shape[1]
                self._n_output_prm = self._labels.shape[1]
                self._error_prm = np.ones(self._size_output)
            elif estimation_mode == "regression-error":
                self._size_output = self._labels.shape[1] * 2
                self._n_output_prm = self._labels.shape[1]
            elif estimation_mode == "custom":
                self._size_output = size_output
                self._n_output_prm = size_output

        self._empirical_error = empirical_error
        self._init_std = init_std
        try:
            n_nodes = list(n_nodes)
        except TypeError:
            n_nodes = [n_nodes]
        self._n_layers = len(n_nodes) + 1
        self._n_nodes = n_nodes
        self._use_bias_node = use_bias_node
        self._n_samples, self._n_features = self._data.shape[0], self._data.shape[1]
        self._w_bound = w_bound
        self._freq_indicator = freq_indicator
        self._hyper_p = hyper_p
        self._sample_id = np.arange(self._n_samples)
        self._prior = prior_f
        self._p_scale = p_scale
        self._prior_ind1 = prior_ind1
        self._estimation_mode = estimation_mode
        self._mask = None
        self._feature_indicators = feature_indicators
        self._feature_means = None

        if use_class_weights:
            counts = np.unique(self._labels, return_counts=True)[1]
            cw = 1 / (counts / np.max(counts))
            self._class_w = cw / np.mean(cw)
            print("Using class weights:", self._class_w)
        else:
            self._class_w = []
        self._instance_weights = instance_weights

        post_samples = None
        if init_weights is not None:
            w_layers = init_weights
        elif pickle_file == "":
            # the reference passes a fixed 0.1 here, whatever init_std says (BNN_env.py:111-115)
            w_layers = init_weight_prm(self._n_nodes, self._n_features, self._size_output,
                                       init_std=0.1, bias_node=use_bias_node)
        else:
            from .files import load_obj
            _, _, logger_obj = load_obj(pickle_file)
            post_samples = logger_obj._post_weight_samples
            w_layers = post_samples[-1]['weights']
        self._w_layers = w_layers
        self._indicators = np.ones(self._w_layers[0].shape)

        self._act_fun = actFun
        if post_samples is not None and actFun._trainable:
            self._act_fun.reset_prm(post_samples[-1]['alphas'])

        if self._prior == 0:
            self._w_bound = self._p_scale       # uniform prior: p_scale is the boundary
        elif self._prior not in (1, 2, 3):
            print('Using default prior N(0,s)')
        self._prior_scale = np.ones(self._n_layers) * self._p_scale

        if len(self._test_data) > 0:
            print("\nTraining set:", self._n_samples, "test set:", self._test_data.shape[0])
        else:
            print("\nTraining set:", self._n_samples, "test set:", None)
        print("Number of features:", self._n_features)
        n_params = np.sum(np.array([np.size(i) for i in self._w_layers]))
        if self._act_fun._trainable:
            n_params += self._n_layers
        print("N. of parameters:", n_params)
        for w in self._w_layers:
            print(w.shape)
        self._n_params = n_params

        if self._feature_indicators:
            self._feature_indicators = np.ones(self._data.shape[1]).astype(int)
            self._feature_means = np.mean(self._data, axis=0)
        else:
            self._feature_indicators = None

    # ---- prior (reference: BNN_env.py:180-194) ---------------------------------------------
    def _prior_kind(self):
        return self._prior if self._prior in (1, 2, 3) else 1

    def calc_prior(self, w=0, ind=[]):
        if isinstance(w, int) and w == 0:
            w = self._w_layers
        if len(ind) == 0:
            ind = self._indicators
        logPrior = 0
        if self._prior != 0:
            kind = self._prior_kind()
            for i in range(self._n_layers):
                logPrior += np.sum(_log_density(kind, w[i], self._prior_scale[i]))
        if self._freq_indicator:
            on = np.sum(ind)
            logPrior += on * np.log(self._prior_ind1) + (self._indicators.size - on) * np.log(1 - self._prior_ind1)
        return logPrior

    def sample_prior_scale(self):
        """Gibbs update of the prior scales (reference: BNN_env.py:196-221)."""
        if self._prior != 1:
            print("Hyper-priors available only for Normal priors.")
            quit()
        if self._hyper_p == 1:
            self._prior_scale = [GibbsSampleNormStdGammaVector(x.flatten()) for x in self._w_layers]
        elif self._hyper_p == 2:
            self._prior_scale = [GibbsSampleNormStdGamma2D(x) for x in self._w_layers]
        elif self._hyper_p == 3:
            self._prior_scale = [GibbsSampleNormStdGammaONE(x) for x in self._w_layers]

    def sample_from_prior(self, reset_weights=True):
        """Draw weights from the prior (reference: BNN_env.py:223-240, including its early
        return of a single layer for the Cauchy / Laplace priors)."""
        w = []
        for n, s in zip(self._w_layers, self._prior_scale):
            if self._prior == 0:
                w.append(np.random.uniform(-self._w_bound, self._w_bound, n.shape))
            elif self._prior == 1:
                w.append(np.random.normal(0, s, n.shape))
            elif self._prior == 2:
                return np.random.standard_cauchy(n.shape) * s
            elif self._prior == 3:
                return np.random.laplace(0, scale=s, size=n.shape)
            else:
                return np.random.standard_normal(n.shape)
        if reset_weights:
            self.reset_weights(w)
        else:
            return w

    # ---- setters (reference: BNN_env.py:244-270) --------------------------------------------
    def reset_weights(self, w):
        self._w_layers = w

    def reset_indicators(self, ind):
        self._indicators = ind

    def reset_error_prm(self, p):
        self._error_prm = p

    def update_data(self, data_dict):
        self._data = data_dict['data']
        self._labels = data_dict['labels']
        self._test_data = data_dict['test_data']
        self._test_labels = data_dict['test_labels']
        self.__dict__.pop("_npbnn_backend", None)      # resident device copy is stale

    def apply_mask(self, m=None):
        if m is not None:
            self._mask = m
        self._w_layers = [self._w_layers[i] * self._mask[i] for i in range(self._n_layers)]
        n_params = np.sum(np.array([np.size(i[i != 0]) for i in self._w_layers]))
        if self._act_fun._trainable:
            n_params += self._n_layers
        print("N. of parameters:", n_params)
        for w in self._w_layers:
            print(w.shape)

    def reset_seed(self, seed):
        self._seed = seed

    def get_feature_mean(self):
        return np.mean(self._data, axis=0)

    # ---- device handle is never pickled / deep-copied ----------------------------------------
    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_npbnn_backend", None)
        return state

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_npbnn_backend":
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new
