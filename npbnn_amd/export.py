"""Checkpoints in upstream's own format: write the ``[bnn, mcmc, logger]`` triple so that ``np_bnn.load_obj`` opens it as
np_bnn objects - and read such a file (written here or by np_bnn itself) as objects of this package.

Upstream's checkpoint is a pickle of its three live objects (np_bnn/BNN_env.py:655-658 through ``SaveObject``,
np_bnn/BNN_lib.py:241-243); its tools read from it ``logger._post_weight_samples``, ``bnn._act_fun``, ``bnn._output_act_fun``,
``bnn._data / _test_data / _test_labels / _feature_indicators / _feature_means / _estimation_mode / _size_output``
(``predictBNN`` np_bnn/BNN_lib.py:404-501, ``feature_importance`` :504-597, ``get_posterior_threshold`` :640-660,
``get_posterior_est`` :715-748, ``pdp`` np_bnn/BNN_pdp.py:87-107) and ``npBNN(pickle_file=...)`` restarts from its last posterior
sample (np_bnn/BNN_env.py:117-120,129-131).  A pickle names classes and functions by module path, so a file whose names are
upstream's *is* upstream's format.  Neither direction imports np_bnn:

  * writing: a pickler that emits ``np_bnn.<module> <name>`` for this package's counterparts (classes become instances
    rebuilt from their attribute dictionaries, tagged callables become references to upstream's functions);
  * reading: an unpickler that resolves ``np_bnn.*`` names to this package's objects.

Attributes that exist only here (device handles, lazily computed statistics, pre-draw state) are dropped on the way out and
upstream's are restored (``_rs``, ``_y``, ``_prior_f`` ...), so the exported sampler can also continue a run under np_bnn.
"""
import copyreg
import io
import pickle
import types

_UPSTREAM = "np_bnn"

# where upstream defines what this package mirrors (module of every public name a checkpoint can reference)
_CLASS_HOME = {"npBNN": "BNN_env", "MCMC": "BNN_env", "postLogger": "BNN_env", "data_transform_obj": "BNN_env", "ActFun": "BNN_lib"}
_LIB_NAMES = ("SoftMax", "RegressTransform", "RegressTransformError", "SoftPlus", "relu_f", "leaky_relu_f", "swish_f", "tanh_f",
              "calc_likelihood", "calc_likelihood_regression", "calc_likelihood_regression_error", "CalcAccuracy", "CalcLabelAccuracy",
              "CalcAccuracyRegression", "CalcLabelAccuracyRegression", "CalcLabelFreq", "SkipAccuracy", "SkipAccuracyVec")
_LIK_NAMES = ("poi_likelihood", "negbin_likelihood", "negbin_likelihood2d", "negbin_likelihood_base10", "gamma_likelihood",
              "negbin_acc", "negbin_acc_base10", "negbin2d_acc", "poi_acc", "gamma_acc")
_MCMC_NAMES = ("UpdateNormal", "UpdateNormal1D", "UpdateFixedNormal", "UpdateNormalNormalized", "UpdateUniform", "UpdateBinomial")
_FUNCTION_HOME = dict([(n, "BNN_lib") for n in _LIB_NAMES] + [(n, "BNN_lik") for n in _LIK_NAMES] + [(n, "BNN_mcmc") for n in _MCMC_NAMES])
_ACTIVATION_NAME = {"relu": "relu_f", "leaky": "leaky_relu_f", "swish": "swish_f", "tanh": "tanh_f"}


class _Named:
    """Stands in the object graph for something upstream defines: pickled as a reference to ``np_bnn.<module>.<name>``."""

    def __init__(self, module, name):
        self.module, self.name = "%s.%s" % (_UPSTREAM, module), name


class _Rebuilt:
    """An instance of an upstream class given by its attribute dictionary."""

    def __init__(self, cls_name, state):
        self.cls = _Named(_CLASS_HOME[cls_name], cls_name)
        self.state = state

    def __reduce_ex__(self, protocol):
        # object.__new__(cls) on load, then the attribute dictionary: what pickle does for a plain instance
        return copyreg._reconstructor, (self.cls, object, None), self.state


_STOCK_SAVERS = {type(None): "save_none", bool: "save_bool", int: "save_long", float: "save_float", bytes: "save_bytes",
                 bytearray: "save_bytearray", str: "save_str", tuple: "save_tuple", list: "save_list", dict: "save_dict",
                 set: "save_set", frozenset: "save_frozenset", types.FunctionType: "save_global", type: "save_type"}


class _UpstreamPickler(pickle._Pickler):
    """The pure-Python pickler with one more type: a ``_Named`` is written as a GLOBAL record of upstream's path, without the
    import-and-compare check the stock pickler makes for classes and functions (np_bnn need not be installed to write)."""
    # (the stock table, rebuilt by name rather than copied: dill, once imported anywhere in the process, replaces entries of
    # pickle._Pickler.dispatch by savers that write references to dill itself - and this file must load where only numpy, scipy
    # and np_bnn are installed)
    dispatch = {kind: getattr(pickle._Pickler, saver) for kind, saver in _STOCK_SAVERS.items() if hasattr(pickle._Pickler, saver)}

    def _save_named(self, obj):
        self.write(pickle.GLOBAL + obj.module.encode("ascii") + b"\n" + obj.name.encode("ascii") + b"\n")
        self.memoize(obj)

    dispatch[_Named] = _save_named


def _callable_ref(fn):
    """Upstream's counterpart of one of this package's tagged callables (likelihoods, statistics, output functions, proposal
    functions); a user's own callable goes out as it is (it must be importable where the file is read, as under upstream)."""
    name = getattr(fn, "__name__", None)
    home = _FUNCTION_HOME.get(name)
    module = getattr(fn, "__module__", "") or ""
    if home is not None and module.startswith("npbnn_amd"):
        return _Named(home, name)
    return fn


def _act_state(act):
    kind = getattr(act.activate, "name", None)
    return dict(_prm=act._prm, _acc_prm=act._acc_prm, _trainable=act._trainable, _function=act._function,
                activate=_Named("BNN_lib", _ACTIVATION_NAME[kind]) if kind in _ACTIVATION_NAME else act.activate)


_BNN_ATTRS = ("_seed", "_data", "_labels", "_test_data", "_test_labels", "_error_prm", "_size_output", "_n_output_prm",
              "_empirical_error", "_init_std", "_n_layers", "_n_nodes", "_use_bias_node", "_n_samples", "_n_features", "_w_bound",
              "_freq_indicator", "_hyper_p", "_sample_id", "_prior", "_p_scale", "_prior_ind1", "_estimation_mode", "_mask",
              "_feature_indicators", "_feature_means", "_class_w", "_instance_weights", "_w_layers", "_indicators", "_prior_scale",
              "_n_params")


def _prior_density(kind):
    """The scipy density upstream's ``calc_prior`` calls (np_bnn/BNN_env.py:135-150); a uniform prior has none."""
    import scipy.stats
    return {1: scipy.stats.norm.logpdf, 2: scipy.stats.cauchy.logpdf, 3: scipy.stats.laplace.logpdf}.get(kind, scipy.stats.norm.logpdf)


def _bnn_state(bnn):
    from .files import DetachedMatrix
    state = {}
    for name in _BNN_ATTRS:
        value = getattr(bnn, name)
        if isinstance(value, DetachedMatrix):
            raise ValueError("the model holds no %s (a light checkpoint was loaded without its side file): attach the data first" % name)
        state[name] = value
    state["_output_act_fun"] = _callable_ref(bnn._output_act_fun)
    state["_act_fun"] = _Rebuilt("ActFun", _act_state(bnn._act_fun))
    if bnn._prior != 0:
        state["_prior_f"] = _prior_density(bnn._prior)
    return state


_MCMC_ATTRS = ("_runID", "_update_f", "_update_ws", "_update_n", "_temperature", "_n_iterations", "_sampling_f", "_print_f",
               "_current_iteration", "_logLik", "_logPrior", "_logPost", "_sample_from_prior", "_last_accepted", "_last_accepted_mem",
               "_acceptance_rate", "_lik_temp", "_mcmc_id", "_randomize_seed", "_counter", "_n_post_samples", "_freq_layer_update",
               "_adapt_f", "_adapt_fM", "_adapt_verbose", "_adapt_stop", "_adapt_freq", "_max_n", "_estimate_error")
_MCMC_STATS = ("_y", "_accuracy", "_label_acc", "_y_test", "_test_accuracy", "_label_freq")


def _mcmc_state(mcmc):
    state = {name: getattr(mcmc, name) for name in _MCMC_ATTRS}
    for name in _MCMC_STATS:                      # computed now if nobody has read them since the last accepted state
        state[name] = getattr(mcmc, name)
    state["_rs"] = mcmc._rs
    for name in ("_likelihood_f", "_accuracy_f", "_accuracy_lab_f", "update_function"):
        state[name] = _callable_ref(getattr(mcmc, name))
    return state


_LOGGER_ATTRS = ("_logfile", "_w_file", "_pklfile", "_log_all_weights", "_post_weight_samples", "_estimation_mode")


def upstream_form(obj):
    """``obj`` as it goes into an upstream-format checkpoint: this package's model, sampler, logger, activation and
    data-transform objects become upstream instances, anything else stays what it is."""
    from .layers import ActFun
    from .logger import postLogger
    from .model import data_transform_obj, npBNN
    from .sampler import MCMC
    if isinstance(obj, npBNN):
        return _Rebuilt("npBNN", _bnn_state(obj))
    if isinstance(obj, MCMC):
        return _Rebuilt("MCMC", _mcmc_state(obj))
    if isinstance(obj, postLogger):
        return _Rebuilt("postLogger", {name: getattr(obj, name) for name in _LOGGER_ATTRS})
    if isinstance(obj, ActFun):
        return _Rebuilt("ActFun", _act_state(obj))
    if isinstance(obj, data_transform_obj):
        return _Rebuilt("data_transform_obj", dict(feature_indicators=obj.feature_indicators, feature_means=obj.feature_means))
    return obj


def dumps_upstream(objs):
    """Bytes of the upstream-format pickle of ``objs`` (one object, or the usual ``[bnn, mcmc, logger]`` list)."""
    out = io.BytesIO()
    payload = [upstream_form(o) for o in objs] if isinstance(objs, (list, tuple)) else upstream_form(objs)
    _UpstreamPickler(out, protocol=4).dump(payload)
    return out.getvalue()


def save_upstream(objs, file_name):
    """Write ``objs`` where ``np_bnn.load_obj(file_name)`` finds np_bnn objects (see the module docstring)."""
    data = dumps_upstream(objs)
    with open(file_name, "wb") as f:
        f.write(data)


# ---- reading -------------------------------------------------------------------------------------------------------------
class _ToThisPackage(pickle.Unpickler):
    """Resolves ``np_bnn.*`` names to this package's objects (``npbnn_amd`` re-exports upstream's flat namespace)."""

    def find_class(self, module, name):
        if module == _UPSTREAM or module.startswith(_UPSTREAM + "."):
            import npbnn_amd
            if hasattr(npbnn_amd, name):
                return getattr(npbnn_amd, name)
            raise pickle.UnpicklingError("%s.%s has no counterpart in npbnn_amd" % (module, name))
        return super().find_class(module, name)


def names_upstream(data):
    """True when the pickle ``data`` (bytes) refers to np_bnn's modules - an upstream-format checkpoint."""
    return (b"c" + _UPSTREAM.encode() + b".") in data or (_UPSTREAM.encode() + b".BNN_") in data


def loads_as_this_package(data):
    """The objects of an upstream-format pickle as objects of this package.  A model loaded this way holds upstream's attribute
    dictionary, which is this package's too; its ``_prior_f`` (a scipy function upstream stores) is dropped - the prior here is
    computed from ``_prior`` - and an activation's ``activate`` becomes this package's tag for the same function."""
    from .model import npBNN
    from .sampler import MCMC
    obj = _ToThisPackage(io.BytesIO(data)).load()
    group = obj if isinstance(obj, (list, tuple)) else [obj]
    models = [o for o in group if isinstance(o, npBNN)]
    for o in models:
        o.__dict__.pop("_prior_f", None)
        for name in ("_feature_indicators", "_feature_means", "_mask"):
            o.__dict__.setdefault(name, None)
    for o in group:
        if isinstance(o, MCMC) and o.__dict__.get("_bnn") is None and len(models) == 1:
            o._bnn = models[0]            # (upstream's sampler holds no model; here its on-demand statistics are the model's)
        if isinstance(o, MCMC) and len(models) == 1 and models[0]._act_fun._trainable and "_slope_term_in_prior" not in o.__dict__:
            # upstream's _logPrior carries the prior of trainable slopes from the first accepted proposal on, and not before
            # (np_bnn/BNN_env.py:320 against :419); the device chain is told which - read it off the number itself
            bare = models[0].calc_prior()
            o._slope_term_in_prior = bool(abs(o._logPrior - bare) > 1e-9 * max(1.0, abs(bare)))
    return obj
