"""npbnn_amd — MI355X-native backend for npBNN's MCMC hot path.

Flat namespace like the reference package (``import npbnn_amd as bn``): ``bn.npBNN``, ``bn.MCMC``,
``bn.MC3``, ``bn.ActFun``, ``bn.run_mcmc``, ``bn.RunPredict`` ...  The per-proposal forward pass +
likelihood of the Metropolis-Hastings loop runs in hand-written HIP kernels (gfx950) behind the C ABI
declared in ``include/npbnn_hip.h``; host code is Python + ctypes.  There is no CPU fallback: without
the built library and an MI355X the device-backed calls raise ``BackendUnavailable``.
"""
__version__ = "0.1.0"

from ._capi import BackendUnavailable, NpbnnError  # noqa: F401
from .backend import HipContext, pack_weights  # noqa: F401
from .proposals import *  # noqa: F401,F403
from .proposals import init_weight_prm  # noqa: F401
from .layers import (ActFun, MatrixMultiplication, MatrixMultiplicationD, RegressTransform,  # noqa: F401
                     RegressTransformError, RunHiddenLayer, RunPredict, RunPredictInd, SoftMax, SoftPlus,
                     create_mask, leaky_relu_f, relu_f, swish_f, tanh_f)
from .likelihoods import (CalcAccuracy, CalcAccuracyRegression, CalcLabelAccuracy,  # noqa: F401
                          CalcLabelAccuracyRegression, CalcLabelFreq, SkipAccuracy, SkipAccuracyVec,
                          calc_likelihood, calc_likelihood_regression, calc_likelihood_regression_error, gamma_acc,
                          gamma_likelihood,
                          negbin2d_acc, negbin_acc, negbin_acc_base10, negbin_likelihood, negbin_likelihood2d,
                          negbin_likelihood_base10, poi_acc, poi_likelihood)
from .model import data_transform_obj, npBNN  # noqa: F401
from .sampler import MCMC, predict  # noqa: F401
from .driver import run_mcmc  # noqa: F401
from .files import (DetachedMatrix, SaveObject, attach_data, get_data, load_obj, randomize_data,  # noqa: F401
                    turn_labels_to_numeric)
from .logger import init_output_files, postLogger  # noqa: F401
from .mc3 import MC3  # noqa: F401
from .posterior import (feature_importance, get_posterior_cat_prob, get_posterior_est, predictBNN,  # noqa: F401
                        sample_from_categorical)
from . import comm  # noqa: F401

BNN = npBNN                       # BASELINE.json's wording
