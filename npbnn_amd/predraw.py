"""Binding of libnpbnn_host.so: pre-draw of K Metropolis-Hastings iterations' random numbers with
numpy-identical streams (see csrc/npbnn_host.c)."""
import ctypes as C
import os

import numpy as np

from ._capi import LIB_DIR, BackendUnavailable

HOST_LIB_PATH = os.path.join(LIB_DIR, "libnpbnn_host.so")
MAX_LAYERS = 8
_M64 = (1 << 64) - 1
_NO_STATE_HANDOVER = bool(os.environ.get("NPBNN_NO_STATE_HANDOVER"))     # A/B: draw through numpy's live generator as before


class ProposalSpec(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("rows", C.c_int32 * MAX_LAYERS), ("cols", C.c_int32 * MAX_LAYERS),
                ("w_off", C.c_int32 * MAX_LAYERS), ("update_n", C.c_int32 * MAX_LAYERS),
                ("update_ws", C.POINTER(C.c_double) * MAX_LAYERS), ("freq_layer_update", C.c_double * MAX_LAYERS),
                ("ws_uniform", C.c_int32 * MAX_LAYERS)]


_lib = None


def load_host_library():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise BackendUnavailable("host pre-draw library not found at %s; run `make -C npbnn_amd/csrc`" % HOST_LIB_PATH)
        lib = C.CDLL(HOST_LIB_PATH)
        lib.npbnn_host_predraw.restype = C.c_int
        lib.npbnn_host_predraw.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.POINTER(ProposalSpec), C.c_int,
                                           C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int]
        lib.npbnn_host_predraw2.restype = C.c_int
        lib.npbnn_host_predraw2.argtypes = lib.npbnn_host_predraw.argtypes + [C.c_int, C.c_double, C.POINTER(C.c_double),
                                                                              C.POINTER(C.c_double)]
        lib.npbnn_host_predraw3.restype = C.c_int
        lib.npbnn_host_predraw3.argtypes = lib.npbnn_host_predraw2.argtypes + [C.c_int, C.c_double, C.POINTER(C.c_int32),
                                                                               C.POINTER(C.c_double)]
        vp = C.c_void_p             # (the same entry with its arrays declared as plain addresses: no typed pointer objects per call)
        proto = C.CFUNCTYPE(C.c_int, vp, C.c_int, C.c_int64, C.c_int64, C.c_int, C.POINTER(ProposalSpec), C.c_int, vp, vp, vp, vp, vp,
                            C.c_int, C.c_int, C.c_double, vp, vp, C.c_int, C.c_double, vp, vp)
        lib.npbnn_host_predraw3_by_address = proto(("npbnn_host_predraw3", lib))
        # the generator handed over by value (six integers of numpy's PCG64 state dictionary) and taken back advanced
        proto_state = C.CFUNCTYPE(C.c_int, vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(ProposalSpec), C.c_int, vp, vp, vp, vp, vp,
                                  C.c_int, C.c_int, C.c_double, vp, vp, C.c_int, C.c_double, vp, vp)
        lib.npbnn_host_predraw_state_by_address = proto_state(("npbnn_host_predraw_state", lib))
        lib.npbnn_host_fast_predraw.restype = C.c_int
        lib.npbnn_host_selftest_doubles.restype = C.c_int
        lib.npbnn_host_selftest_doubles.argtypes = [C.c_uint64, C.c_int, C.POINTER(C.c_double)]
        _lib = lib
    return _lib


class PredrawPlan:
    """Everything about a pre-draw that only changes with the sampler's proposal settings - the proposal description the C
    routine reads (layer shapes, entries per layer, step-size matrices, layer frequencies) - built once and kept; a dispatch then
    only allocates its outputs and makes the call.  ``update_ws`` are used as they are (float64, layer shaped): the caller keeps
    them unchanged for the life of the plan."""

    def __init__(self, weights, update_n, update_ws, freq_layer_update):
        spec = ProposalSpec()
        n_layers = len(weights)
        spec.n_layers = n_layers
        self._keep = []
        off = 0
        for i, w in enumerate(weights):
            spec.rows[i], spec.cols[i], spec.w_off[i] = w.shape[0], w.shape[1], off
            off += w.size
            spec.update_n[i] = int(update_n[i])
            ws = np.ascontiguousarray(np.broadcast_to(update_ws[i], w.shape), dtype=np.float64)
            self._keep.append(ws)
            spec.update_ws[i] = ws.ctypes.data_as(C.POINTER(C.c_double))
            spec.ws_uniform[i] = 1 if ws.size and bool(np.all(ws == ws.flat[0])) else 0      # (kept unchanged for the life of the plan)
            spec.freq_layer_update[i] = float(freq_layer_update[i])
        self.spec = spec
        self.spec_ref = C.byref(spec)
        self.n_weights = off
        self.M = int(sum(int(n) for n in update_n))

    def run(self, rs, randomize_seed, first_iteration, mcmc_id, K, empty=None, sigma_k=0, sigma_f=0.5, n_slopes=0, slope_d=0.05,
            empty_group=None, state=None):
        """``state``: ``rs.bit_generator.state`` if the caller has just read it (saves reading it again)."""
        if empty is None:
            empty = np.empty
        lib = load_host_library()
        M = self.M
        if empty_group is not None:       # indices and deviates side by side: they travel to the device in one copy
            idx, delta, cnt, u = empty_group([((K, M), np.int32), ((K, M), np.float64), ((K,), np.int32), ((K,), np.float64)])
        else:
            idx, delta, cnt, u = empty((K, M), np.int32), empty((K, M), np.float64), empty((K,), np.int32), empty((K,), np.float64)
        # (unused entries of a row are set to -1 / 0 by the C routine, row by row, while the row is in cache)
        lmask = np.empty(K, dtype=np.int32)
        bitgen = None
        if not randomize_seed:
            bitgen = rs.bit_generator.ctypes.bit_generator
        sigma_k = int(sigma_k)
        n_slopes = int(n_slopes)
        chosen = u_sigma = slope_idx = slope_delta = None
        if sigma_k:
            chosen = np.empty((K, sigma_k), dtype=np.float64)
            u_sigma = np.empty((K, sigma_k), dtype=np.float64)
        if n_slopes:
            slope_idx = np.zeros(K, dtype=np.int32)
            slope_delta = np.zeros(K, dtype=np.float64)
        with rs.bit_generator.lock:
            by_value = None
            if not randomize_seed and not _NO_STATE_HANDOVER:
                # a PCG64 generator travels by value: its state as six integers in, the advanced state out - the C side then draws
                # with its own inlined generator and distributions (npbnn_host.c, "fast path": 2-3x numpy's routines called through
                # the generator's function pointers) and nothing depends on the layout of numpy's structs
                st = rs.bit_generator.state if state is None else state
                if st.get("bit_generator") == "PCG64":
                    s128, i128 = st["state"]["state"], st["state"]["inc"]
                    by_value = (C.c_uint64 * 6)(s128 >> 64, s128 & _M64, i128 >> 64, i128 & _M64, st["has_uint32"], st["uinteger"])
            if by_value is not None:
                rc = lib.npbnn_host_predraw_state_by_address(
                    C.addressof(by_value), int(first_iteration), int(mcmc_id), K, self.spec_ref, M, idx.ctypes.data, delta.ctypes.data,
                    cnt.ctypes.data, u.ctypes.data, lmask.ctypes.data, self.n_weights, sigma_k, float(sigma_f),
                    chosen.ctypes.data if sigma_k else None, u_sigma.ctypes.data if sigma_k else None, n_slopes, float(slope_d),
                    slope_idx.ctypes.data if n_slopes else None, slope_delta.ctypes.data if n_slopes else None)
                if rc == 0:
                    rs.bit_generator.state = {"bit_generator": "PCG64",
                                              "state": {"state": (by_value[0] << 64) | by_value[1], "inc": (by_value[2] << 64) | by_value[3]},
                                              "has_uint32": int(by_value[4]), "uinteger": int(by_value[5])}
            else:
                rc = lib.npbnn_host_predraw3_by_address(
                    bitgen, 1 if randomize_seed else 0, int(first_iteration), int(mcmc_id), K, self.spec_ref, M, idx.ctypes.data,
                    delta.ctypes.data, cnt.ctypes.data, u.ctypes.data, lmask.ctypes.data, self.n_weights, sigma_k, float(sigma_f),
                    chosen.ctypes.data if sigma_k else None, u_sigma.ctypes.data if sigma_k else None, n_slopes, float(slope_d),
                    slope_idx.ctypes.data if n_slopes else None, slope_delta.ctypes.data if n_slopes else None)
        if rc != 0:
            raise RuntimeError("npbnn_host_predraw failed with code %d" % rc)
        out = (idx, delta, cnt, u, lmask)
        if sigma_k > 0:
            out += (chosen, u_sigma)
        if n_slopes > 0:
            out += (slope_idx, slope_delta)
        return out


def predraw(rs, randomize_seed, first_iteration, mcmc_id, K, weights, update_n, update_ws, freq_layer_update, empty=None,
            sigma_k=0, sigma_f=0.5, n_slopes=0, slope_d=0.05, empty_group=None):
    """Draw K iterations' proposals.  ``rs`` is the chain's numpy Generator (advanced in place unless
    ``randomize_seed``).  Returns (idx [K,M] int32, delta [K,M] float64, cnt [K], u [K], layer_mask [K]).
    ``empty(shape, dtype)`` allocates the four arrays that travel to the device (page-locked memory, pinned.py);
    default numpy.  ``sigma_k`` > 0: every iteration first draws the regression error-parameter proposal
    (multiplier_proposal_vector on ``sigma_k`` columns, BNN_env.py:435-442); two more arrays are returned,
    chosen [K, sigma_k] (0/1) and u_sigma [K, sigma_k].  ``n_slopes`` > 0: the very first draws of an iteration are those of the
    trainable activation slopes (UpdateNormal1D on ``n_slopes`` values, n = 1, BNN_env.py:416-421); the result then ends with
    slope_idx [K] int32 and slope_delta [K].  (One-off form of :class:`PredrawPlan`.)"""
    plan = PredrawPlan(weights, update_n, update_ws, freq_layer_update)
    return plan.run(rs, randomize_seed, first_iteration, mcmc_id, K, empty=empty, sigma_k=sigma_k, sigma_f=sigma_f, n_slopes=n_slopes,
                    slope_d=slope_d, empty_group=empty_group)
