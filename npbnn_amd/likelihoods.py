"""Likelihood and accuracy functions of the hot path, as tagged callables.

The sampler recognises these objects and runs them *fused* with the forward pass on the device
(no prediction matrix is ever materialised).  Called directly on a host prediction matrix — the
reference's operator signature ``f(prediction, labels, sample_id, class_weight=, instance_weight=,
lik_temp=, sig2=)`` (np_bnn/BNN_env.py:313-319) — they run the stand-alone device kernel in
``device_ops``.  User callables with the same signature are honoured by the sampler through the
slow path (device forward pass -> host callable).
"""
import sys

import numpy as np

from . import _capi as capi


class _Likelihood:
    def __init__(self, name, kind, doc):
        self.__name__ = name
        self.kind = kind
        self.__doc__ = doc

    def __call__(self, prediction, labels, sample_id=None, class_weight=None, instance_weight=None,
                 lik_temp=1, sig2=None):
        from . import device_ops
        if self.kind in (capi.LIK_GAUSS, capi.LIK_GAUSS_PRED_SIGMA) and instance_weight is not None:
            sys.exit("instance_weight not implemented for regression")     # reference: BNN_lib.py:129-130,140-141
        return device_ops.likelihood(self.kind, prediction, labels, class_weight=class_weight,
                                     instance_weight=instance_weight, lik_temp=lik_temp, sig2=sig2)

    def __repr__(self):
        return "<likelihood %s>" % self.__name__

    def __reduce__(self):
        return self.__name__


calc_likelihood = _Likelihood(
    "calc_likelihood", capi.LIK_CATEGORICAL,
    "Categorical log-likelihood lik_temp * sum_n w_n log p[n, y_n] (reference: BNN_lib.py:100-121).")
calc_likelihood_regression = _Likelihood(
    "calc_likelihood_regression", capi.LIK_GAUSS,
    "Gaussian log-likelihood with a standard deviation per target column, passed as `sig2` "
    "(reference: BNN_lib.py:123-131).")
calc_likelihood_regression_error = _Likelihood(
    "calc_likelihood_regression_error", capi.LIK_GAUSS_PRED_SIGMA,
    "Gaussian log-likelihood with the standard deviations predicted by the second half of the outputs "
    "(reference: BNN_lib.py:134-143).")
poi_likelihood = _Likelihood(
    "poi_likelihood", capi.LIK_POISSON,
    "Poisson log-likelihood with rate exp(prediction[:, 0]) (reference: BNN_lik.py:5-14).")
negbin_likelihood = _Likelihood(
    "negbin_likelihood", capi.LIK_NEGBIN,
    "Negative-binomial log-likelihood, mean exp(eta0), p = logistic(eta1) (reference: BNN_lik.py:16-30).")
negbin_likelihood2d = _Likelihood(
    "negbin_likelihood2d", capi.LIK_NEGBIN2D,
    "Negative-binomial log-likelihood over k target columns (reference: BNN_lik.py:33-49).")
negbin_likelihood_base10 = _Likelihood(
    "negbin_likelihood_base10", capi.LIK_NEGBIN_BASE10,
    "Negative-binomial log-likelihood with base-10 links (reference: BNN_lik.py:55-66).")


def gamma_likelihood(prediction, true_values, sample_id=None, class_weight=None, instance_weight=None, lik_temp=1, sig2=0):
    """Gamma log-likelihood exactly as upstream evaluates it (np_bnn/BNN_lik.py:68-78): shape a = exp(prediction[:, 0]) and
    b = exp(prediction[:, 1]) handed to ``scipy.stats.gamma.logpdf(true_values, a, b)``, where the third positional argument
    is the LOCATION, and an N x 1 ``true_values`` broadcasts against the N shapes to an N x N table that is summed whole.
    That all-pairs sum does not decompose over the rows of the feature matrix, so it is not fused into the streaming kernel:
    the sampler gives this function the device-computed prediction matrix (the path every user-supplied likelihood takes)
    and the closed form below runs on the host:  sum (a-1) log(x-b) - (x-b) - lgamma(a)  over every pair with x > b."""
    from scipy.special import gammaln
    eta = np.asarray(prediction, dtype=float)
    a, b = np.exp(eta[:, 0]), np.exp(eta[:, 1])
    x = np.asarray(true_values, dtype=float)
    z = x - b                                          # (N x 1 against N: the all-pairs table; N against N: row by row)
    with np.errstate(divide="ignore", invalid="ignore"):
        table = np.where(z > 0, (a - 1) * np.log(z) - z - gammaln(a), -np.inf)
    return np.sum(table)


def gamma_acc(y, lab):
    """Mean squared error of exp(y[:, 0]) against the targets (np_bnn/BNN_lik.py:98-99 computes it and forgets to return it;
    here it is returned)."""
    return np.mean((np.exp(np.asarray(y)[:, 0]) - np.asarray(lab).flatten()) ** 2)


def likelihood_kind(fn):
    return fn.kind if isinstance(fn, _Likelihood) else None


# ---- accuracy statistics (reference: BNN_lib.py:195-239, BNN_lik.py:81-99) ---------------------
class _Stat:
    """kind: 'acc' | 'label_acc' | 'mse' | 'label_mse' | 'skip' | 'skip_vec' | exp-link MSE variants."""

    def __init__(self, name, kind):
        self.__name__ = name
        self.kind = kind

    def __call__(self, y, lab):
        from . import device_ops
        return device_ops.statistic(self.kind, y, lab)

    def __repr__(self):
        return "<statistic %s>" % self.__name__

    def __reduce__(self):
        return self.__name__


CalcAccuracy = _Stat("CalcAccuracy", "acc")                                  # BNN_lib.py:203-209
CalcLabelAccuracy = _Stat("CalcLabelAccuracy", "label_acc")                  # BNN_lib.py:211-219
CalcAccuracyRegression = _Stat("CalcAccuracyRegression", "mse")              # BNN_lib.py:195-197
CalcLabelAccuracyRegression = _Stat("CalcLabelAccuracyRegression", "label_mse")  # BNN_lib.py:199-201
negbin_acc = _Stat("negbin_acc", "mse_exp_col0")                             # BNN_lik.py:81-83
negbin_acc_base10 = _Stat("negbin_acc_base10", "mse_pow10_col0")             # BNN_lik.py:85-87
negbin2d_acc = _Stat("negbin2d_acc", "mse_exp")                              # BNN_lik.py:89-91
poi_acc = _Stat("poi_acc", "mse_exp_col0")                                   # BNN_lik.py:94-96


def SkipAccuracy(_, __):
    return 1.0                                                               # BNN_lib.py:235-236


def SkipAccuracyVec(_, __):
    return np.ones(1)                                                        # BNN_lib.py:238-239


def CalcLabelFreq(y):
    """Frequency of each class among the argmax predictions (reference: BNN_lib.py:228-233)."""
    from . import device_ops
    return device_ops.statistic("label_freq", y, None)


def stat_kind(fn):
    return fn.kind if isinstance(fn, _Stat) else None


def stats_from_confusion(conf, labels_present=None):
    """accuracy, per-class accuracy (classes present in the labels, ascending) and predicted-class
    frequencies from a C x C [true, predicted] count matrix."""
    conf = np.asarray(conf, dtype=np.int64)
    n = conf.sum()
    per_true = conf.sum(axis=1)
    present = per_true > 0 if labels_present is None else labels_present
    acc = np.trace(conf) / n
    label_acc = np.diag(conf)[present] / per_true[present]
    label_freq = conf.sum(axis=0) / n
    return acc, label_acc, label_freq
