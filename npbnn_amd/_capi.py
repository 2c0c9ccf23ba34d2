"""ctypes binding of the C ABI in ``include/npbnn_hip.h`` (libnpbnn_hip.so).

There is no CPU fallback: if the shared library is missing or cannot be
loaded the import of any device-backed functionality raises
:class:`BackendUnavailable`.
"""
import ctypes as C
import os

import numpy as np

MAX_LAYERS = 8
MAX_WIDTH = 4096
MAX_TARGETS = 16
XSTATE_DOUBLES = 4 + MAX_TARGETS      # doubles per exchange in ChainJob.out_state: logLik, logPrior, temperature, iterations, sigma[...]

ACT_RELU, ACT_LEAKY, ACT_SWISH, ACT_TANH = 0, 1, 2, 3
OUT_SOFTMAX, OUT_IDENTITY, OUT_SOFTPLUS_HALF = 0, 1, 2
(LIK_CATEGORICAL, LIK_GAUSS, LIK_GAUSS_PRED_SIGMA, LIK_POISSON, LIK_NEGBIN, LIK_NEGBIN2D,
 LIK_NEGBIN_BASE10, LIK_NONE) = range(8)
PRIOR_UNIFORM, PRIOR_NORMAL, PRIOR_CAUCHY, PRIOR_LAPLACE = 0, 1, 2, 3
TRAIN, TEST = 0, 1
OPT_L0_PRECISION = 1
OPT_FAST_TAILS = 2
OPT_PERSISTENT = 3
OPT_TRAINABLE_SLOPES = 4
OPT_WIDE = 5
L0_AUTO, L0_F32, L0_F16 = 0, 1, 2
INFO_L0_F16, INFO_WAVES_PER_BLOCK, INFO_N_CU, INFO_FAST_TAILS, INFO_TURN_NS_OVERLAPPED, INFO_TURN_NS_BETWEEN, INFO_MAX_CANDIDATES = 1, 2, 3, 4, 5, 6, 7
INFO_IT_NS_OVERLAPPED, INFO_IT_NS_BETWEEN = 8, 9
INFO_WIDE = 10
INFO_F16_MOVED_COLUMNS, INFO_F16_MAX_MOVE = 11, 12
E_RANGE = -6
E_SYNC = -7

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
LIB_PATH = os.path.join(LIB_DIR, "libnpbnn_hip.so")


class BackendUnavailable(RuntimeError):
    """The HIP shared library is missing or no MI355X is usable."""


class NpbnnError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("npbnn_hip error %d: %s" % (code, message))
        self.code = code


class Arch(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("in_dim", C.c_int32),
                ("out_dim", C.c_int32 * MAX_LAYERS), ("has_bias", C.c_int32 * MAX_LAYERS),
                ("act_kind", C.c_int32), ("out_kind", C.c_int32), ("lik_kind", C.c_int32),
                ("n_targets", C.c_int32), ("final_act", C.c_int32)]


class EvalOut(C.Structure):
    _fields_ = [("loglik", C.c_double), ("sigma", C.c_double * MAX_TARGETS),
                ("sum_r", C.c_double * MAX_TARGETS), ("sum_r2", C.c_double * MAX_TARGETS),
                ("n_rows", C.c_int64)]


class ChainCfg(C.Structure):
    _fields_ = [("prior_kind", C.c_int32), ("prior_scale", C.c_double * MAX_LAYERS), ("w_bound", C.c_double),
                ("temperature", C.c_double), ("lik_temp", C.c_double), ("sigma_given", C.c_int32),
                ("sigma", C.c_double * MAX_TARGETS), ("cur_loglik", C.c_double), ("cur_logprior", C.c_double),
                ("cur_sigma", C.c_double * MAX_TARGETS), ("force_f32", C.c_int32), ("n_candidates", C.c_int32),
                ("schedule", C.c_int32), ("reserved_", C.c_int32), ("sigma_mult", C.POINTER(C.c_double)),
                ("hastings", C.POINTER(C.c_double)), ("prior_scale_w", C.POINTER(C.c_double)),
                ("slope_idx", C.POINTER(C.c_int32)), ("slope_delta", C.POINTER(C.c_double)),
                ("cur_slopes", C.c_double * MAX_LAYERS), ("n_slopes", C.c_int32), ("slope_term_in_prior", C.c_int32)]


class ChainResult(C.Structure):
    _fields_ = [("loglik", C.c_double), ("logprior", C.c_double), ("sigma", C.c_double * MAX_TARGETS),
                ("n_accepted", C.c_int64), ("n_passes", C.c_int32), ("n_candidates", C.c_int32),
                ("n_void_passes", C.c_int32), ("schedule", C.c_int32), ("temperature", C.c_double),
                ("iterations_done", C.c_int32), ("overflow", C.c_int32), ("slopes", C.c_double * MAX_LAYERS)]


class GeneralCfg(C.Structure):
    """npbnn_general_cfg: the pre-drawn proposals of the general device chain (npbnn_chain_run_general)."""
    _fields_ = [("proposal_kind", C.c_int32), ("M", C.c_int32), ("idx", C.POINTER(C.c_int32)), ("val", C.POINTER(C.c_double)),
                ("cnt", C.POINTER(C.c_int32)), ("h_idx", C.POINTER(C.c_int32)), ("h_val", C.POINTER(C.c_double)),
                ("h_fac", C.POINTER(C.c_double)), ("h_cnt", C.POINTER(C.c_int32)), ("layer_mask", C.POINTER(C.c_int32)),
                ("ind_inout", C.POINTER(C.c_double)), ("ind_ptr", C.POINTER(C.c_int32)), ("ind_pos", C.POINTER(C.c_int32)),
                ("prior_ind1", C.c_double), ("has_indicator_prior", C.c_int32), ("reserved_", C.c_int32),
                ("find_inout", C.POINTER(C.c_double)), ("feature_means", C.POINTER(C.c_double)),
                ("find_ptr", C.POINTER(C.c_int32)), ("find_pos", C.POINTER(C.c_int32)), ("find_use", C.POINTER(C.c_int32))]


PROP_NORMAL, PROP_UNIFORM, PROP_FIXED_NORMAL, PROP_NORMAL_NORMALIZED = 0, 1, 2, 3


class ChainJob(C.Structure):
    """One chain's share of npbnn_chains_run_exchange (npbnn_chain_job)."""
    _fields_ = [("ctx", C.c_void_p), ("cfg", C.POINTER(ChainCfg)), ("W_inout", C.POINTER(C.c_double)),
                ("mask_packed", C.POINTER(C.c_double)), ("M", C.c_int32), ("chain_id", C.c_int32),
                ("idx", C.POINTER(C.c_int32)), ("delta", C.POINTER(C.c_double)), ("cnt", C.POINTER(C.c_int32)),
                ("log_u", C.POINTER(C.c_double)), ("out_accepted", C.POINTER(C.c_uint8)),
                ("out_loglik_prop", C.POINTER(C.c_double)), ("out_logprior_prop", C.POINTER(C.c_double)),
                ("out_state", C.POINTER(C.c_double)), ("out_cold_w", C.POINTER(C.c_double)),
                ("result", C.POINTER(ChainResult))]


REC_DOUBLES = 4      # record of a chain at an exchange: logPost, temperature, finished-the-segment flag, iterations done

SCHED_AUTO, SCHED_SERIAL, SCHED_OVERLAP, SCHED_OVERLAP2, SCHED_PERSIST, SCHED_PERSIST_SERIAL = 0, 1, 2, 3, 4, 5


_P = C.c_void_p
_DP = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/npbnn_hip.h declares
SIGNATURES = {
    "npbnn_abi_version": (C.c_int, []),
    "npbnn_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "npbnn_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "npbnn_destroy": (None, [_P]),
    "npbnn_last_error": (C.c_char_p, [_P]),
    "npbnn_set_data_f64": (C.c_int, [_P, _DP, C.c_int64, C.c_int32, C.c_int]),
    "npbnn_set_data_f32": (C.c_int, [_P, C.POINTER(C.c_float), C.c_int64, C.c_int32, C.c_int]),
    "npbnn_set_labels_i64": (C.c_int, [_P, C.POINTER(C.c_int64), C.c_int64, C.c_int]),
    "npbnn_set_targets_f64": (C.c_int, [_P, _DP, C.c_int64, C.c_int32, C.c_int]),
    "npbnn_set_row_weights": (C.c_int, [_P, _DP, C.c_int64, _DP, C.c_int32]),
    "npbnn_set_arch": (C.c_int, [_P, C.POINTER(Arch)]),
    "npbnn_set_layer_mask": (C.c_int, [_P, _DP]),
    "npbnn_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "npbnn_get_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "npbnn_eval": (C.c_int, [_P, _DP, _DP, _DP, C.c_double, _DP, C.c_int, C.POINTER(EvalOut),
                             C.POINTER(C.c_int64)]),
    "npbnn_predict": (C.c_int, [_P, _DP, _DP, _DP, C.c_int, C.c_int, _DP]),
    "npbnn_predict_sets": (C.c_int, [_P, _DP, _DP, C.c_int32, C.c_int, C.c_int, _DP]),
    "npbnn_time_eval": (C.c_int, [_P, _DP, C.c_int, _DP, _DP]),
    "npbnn_time_pass": (C.c_int, [_P, _DP, C.c_int, C.c_int, _DP, C.POINTER(C.c_int)]),
    "npbnn_time_wide": (C.c_int, [_P, _DP, C.c_int, _DP, _DP, C.POINTER(C.c_int)]),
    "npbnn_share_data": (C.c_int, [_P, _P]),
    "npbnn_pinned_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "npbnn_pinned_free": (None, [C.c_void_p]),
    "npbnn_chain_run": (C.c_int, [_P, C.POINTER(ChainCfg), _DP, _DP, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _DP,
                                  C.POINTER(C.c_int32), _DP, C.POINTER(C.c_uint8), _DP, _DP, C.POINTER(ChainResult)]),
    "npbnn_chains_run_exchange": (C.c_int, [_P, C.POINTER(ChainJob), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32), _DP, C.c_double, _DP,
                                            C.POINTER(C.c_int32)]),
    "npbnn_chains_run_batched": (C.c_int, [C.POINTER(ChainJob), C.c_int32, C.c_int32]),
    "npbnn_chain_run_general": (C.c_int, [_P, C.POINTER(ChainCfg), C.POINTER(GeneralCfg), _DP, _DP, C.c_int32, _DP, C.POINTER(C.c_uint8),
                                          _DP, _DP, C.POINTER(ChainResult)]),
    "npbnn_op_activation": (C.c_int, [C.c_int, C.c_int, C.c_double, _DP, C.c_int64]),
    "npbnn_op_output": (C.c_int, [C.c_int, C.c_int, _DP, C.c_int64, C.c_int32, C.c_int32]),
    "npbnn_op_likelihood": (C.c_int, [C.c_int, C.c_int, _DP, C.c_int64, C.c_int32, C.POINTER(C.c_int64), _DP, C.c_int32, _DP, _DP,
                                      C.c_int32, C.c_double, _DP, _DP]),
    "npbnn_op_confusion": (C.c_int, [C.c_int, _DP, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64)]),
    "npbnn_op_sse": (C.c_int, [C.c_int, _DP, _DP, C.c_int64, C.c_int32, C.c_int32, C.c_int, _DP]),
    "npbnn_comm_unique_id": (C.c_int, [C.c_char * 128]),
    "npbnn_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char * 128, C.POINTER(_P)]),
    "npbnn_comm_allgather_f64": (C.c_int, [_P, _DP, C.c_int, _DP]),
    "npbnn_comm_bcast_i64": (C.c_int, [_P, C.POINTER(C.c_int64), C.c_int, C.c_int]),
    "npbnn_comm_destroy": (None, [_P]),
    "npbnn_comm_runtime": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "npbnn_device_synchronize": (C.c_int, [C.c_int]),
    "npbnn_set_row_shard": (C.c_int, [_P, _P, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64]),
}

# npbnn_gather_fn: int (*)(void* user, const double* send, double* recv, int32_t count)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32)

_lib = None


def load_library(path=None):
    """Load libnpbnn_hip.so and bind every declared symbol.  Raises
    BackendUnavailable (never falls back) when that is impossible."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("NPBNN_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise BackendUnavailable(
            "HIP backend library not found at %s; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C npbnn_amd/csrc` (there is no CPU fallback)" % p)
    try:
        lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise BackendUnavailable("cannot load %s: %s" % (p, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise BackendUnavailable("%s does not export %s" % (p, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.npbnn_abi_version() != 1:
        raise BackendUnavailable("ABI version mismatch in %s" % p)
    if path is None:
        _lib = lib
    return lib


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def chain_run_by_address(lib):
    """npbnn_chain_run with its array arguments declared as plain addresses (``array.ctypes.data``): the call every dispatch of a
    device chain makes, without building a typed ctypes pointer per array."""
    fn = getattr(lib, "_npbnn_chain_run_by_address", None)
    if fn is None:
        vp = C.c_void_p
        proto = C.CFUNCTYPE(C.c_int, _P, C.POINTER(ChainCfg), vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, C.POINTER(ChainResult))
        fn = lib._npbnn_chain_run_by_address = proto(("npbnn_chain_run", lib))
    return fn


def dptr(a):
    return None if a is None else a.ctypes.data_as(_DP)


def check(lib, ctx, rc):
    if rc != 0:
        msg = lib.npbnn_last_error(ctx)
        raise NpbnnError(rc, msg.decode() if msg else "?")
