"""Device context of one chain: thin object wrapper over the C ABI.

``HipContext`` owns one ``npbnn_ctx`` (one HIP stream on one MI355X) holding
the resident training / test matrices and the network description.  All
numerics of the hot path run in the hand-written HIP kernels behind it.
"""
import ctypes as C
import os
from operator import is_ as _is
from time import perf_counter as _now

import numpy as np

from . import _capi as capi


_PACKED_VIEWS = {}      # id(first layer) -> (the layer arrays, the packed vector they are consecutive views of); strong references: ids stay valid


def note_packed_views(layers, base):
    """``layers`` were just made as the consecutive views of the packed vector ``base`` (the state a device batch returns): the next
    pack_weights of exactly these objects is one copy of ``base``."""
    if len(_PACKED_VIEWS) >= 32:
        _PACKED_VIEWS.clear()
    _PACKED_VIEWS[id(layers[0])] = (tuple(layers), base)


def pack_weights(weights):
    """Concatenate the layer matrices (each row-major) into the packed float64 vector the C ABI takes (always a fresh array).
    Layers that are consecutive views of one packed vector - what a device batch leaves in the model - are copied in one piece;
    a set of layer objects once found to be such views is recognised by identity afterwards (a view cannot be moved)."""
    first = weights[0]
    known = _PACKED_VIEWS.get(id(first))
    if known is not None and len(known[0]) == len(weights) and all(map(_is, known[0], weights)):
        return known[1].copy()
    base = getattr(first, "base", None)
    if (isinstance(base, np.ndarray) and base.ndim == 1 and base.dtype == np.float64 and base.flags.c_contiguous
            and base.size == sum(w.size for w in weights)):
        at = base.ctypes.data
        for w in weights:
            if w.base is not base or w.dtype != np.float64 or not w.flags.c_contiguous or w.ctypes.data != at:
                break
            at += w.nbytes
        else:
            if len(_PACKED_VIEWS) >= 32:
                _PACKED_VIEWS.clear()
            _PACKED_VIEWS[id(first)] = (tuple(weights), base)
            return base.copy()
    return np.concatenate([np.ascontiguousarray(w, dtype=np.float64).ravel() for w in weights])


_NO_SIGMA = np.zeros(0)


def _addr(a):
    """Address of an array's first element (a third of the cost of ``a.ctypes.data``, which builds a helper object per use)."""
    try:
        return C.addressof(C.c_char.from_buffer(a))
    except (TypeError, ValueError):          # read-only or empty buffers
        return a.ctypes.data


def default_device():
    """Device of this process: NPBNN_DEVICE, else LOCAL_RANK (one process per
    GPU under torch.distributed.run), else 0."""
    for key in ("NPBNN_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(key)
        if v is not None and v != "":
            return int(v)
    return 0


class HipContext:
    def __init__(self, device=None):
        self._lib = capi.load_library()
        self._ctx = capi._P()
        dev = default_device() if device is None else int(device)
        n = C.c_int(0)
        rc = self._lib.npbnn_device_count(C.byref(n))
        if rc != 0 or n.value < 1:
            raise capi.BackendUnavailable("no HIP device visible: the npbnn_amd hot path needs an MI355X "
                                          "(there is no CPU fallback)")
        capi.check(self._lib, None, self._lib.npbnn_create(dev % n.value, C.byref(self._ctx)))
        self.device = dev % n.value
        opt = os.environ.get("NPBNN_L0", "").lower()
        if opt in ("f32", "f16", "auto"):
            self.set_l0_precision(opt)
        self.arch = None
        self.n_rows = {}
        self.n_out = None
        self.seconds_in_chain_run = 0.0
        self.sync_fallbacks = 0      # batches of the (opt-in) two-stream schedule that timed out and were repeated on one stream

    # -- lifecycle --------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.npbnn_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        capi.check(self._lib, self._ctx, rc)

    # -- options ----------------------------------------------------------------------
    def set_l0_precision(self, mode):
        """'auto' (default): fp16-split first layer when the data allows, else float32; 'f32'; 'f16'."""
        value = {"auto": capi.L0_AUTO, "f32": capi.L0_F32, "f16": capi.L0_F16}[mode]
        self._chk(self._lib.npbnn_set_option(self._ctx, capi.OPT_L0_PRECISION, value))

    def set_fast_tails(self, on):
        """Shape-specialised builds of the evaluation kernel for likelihood-only launches (default on; same results)."""
        self._chk(self._lib.npbnn_set_option(self._ctx, capi.OPT_FAST_TAILS, 1 if on else 0))

    def set_trainable_slopes(self, on):
        """Reserve a slot per hidden layer in the weight image for the activation slope, so that the candidates of a chain pass
        can each carry their own (ActFun(trainable=True) on the device chain); such a network runs on the general builds."""
        on = bool(on)
        if on != getattr(self, "_trainable_slopes", False):
            self._chk(self._lib.npbnn_set_option(self._ctx, capi.OPT_TRAINABLE_SLOPES, 1 if on else 0))
            self._trainable_slopes = on

    def set_row_shard(self, comm_handle, gather_cb, rank, n_ranks, n_rows_total):
        """npbnn_set_row_shard: ``comm_handle`` an npbnn_comm* (ctypes void pointer) or None, ``gather_cb`` a capi.GATHER_FN or None."""
        cb = C.cast(gather_cb, C.c_void_p) if gather_cb is not None else None
        self._chk(self._lib.npbnn_set_row_shard(self._ctx, comm_handle, cb, None, int(rank), int(n_ranks), int(n_rows_total)))

    def set_wide(self, on):
        """Run every network of this context on the weight-streamed path (layers as tiled matrix products, weights streamed from
        HBM), not only those the LDS-resident path cannot hold (layers of more than 128 nodes, weight images that crowd out the
        waves): for A/B timing and tests.  Results agree to rounding, not bit for bit."""
        self._chk(self._lib.npbnn_set_option(self._ctx, capi.OPT_WIDE, 1 if on else 0))

    def is_wide(self):
        """Does the architecture set last run on the weight-streamed path?"""
        return bool(self.info(capi.INFO_WIDE))

    def f16_moved_columns(self):
        """(columns of the training matrix whose fp16 scale was moved up because of heavy tails, largest move in powers of two)."""
        return self.info(capi.INFO_F16_MOVED_COLUMNS), self.info(capi.INFO_F16_MAX_MOVE)

    def set_persistent(self, on):
        """May the library pick the persistent form of the overlapped chain schedule by itself (default on)?"""
        self._chk(self._lib.npbnn_set_option(self._ctx, capi.OPT_PERSISTENT, 1 if on else 0))

    def info(self, what):
        out = C.c_int(0)
        self._chk(self._lib.npbnn_get_info(self._ctx, what, C.byref(out)))
        return out.value

    def l0_mode(self):
        """Layer-0 path of the most recent launch: 'f16-split' or 'f32'."""
        return "f16-split" if self.info(capi.INFO_L0_F16) else "f32"

    # -- resident data ----------------------------------------------------------------
    def set_data(self, X, which=capi.TRAIN):
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("data must be a 2-D matrix")
        if X.dtype == np.float32:
            Xc = np.ascontiguousarray(X)
            self._chk(self._lib.npbnn_set_data_f32(self._ctx, Xc.ctypes.data_as(C.POINTER(C.c_float)),
                                                   X.shape[0], X.shape[1], which))
        else:
            Xc = capi.as_f64(X)
            self._chk(self._lib.npbnn_set_data_f64(self._ctx, capi.dptr(Xc), X.shape[0], X.shape[1], which))
        self.n_rows[which] = X.shape[0]

    def share_data(self, owner):
        """Use the feature matrices (training and test) resident in ``owner`` (another HipContext on the same device)
        instead of uploading copies (npbnn_share_data); labels / targets / row weights are set per context afterwards."""
        self._chk(self._lib.npbnn_share_data(self._ctx, owner._ctx))
        self.n_rows = dict(owner.n_rows) if isinstance(owner.n_rows, dict) else list(owner.n_rows)
        self._data_owner = owner            # (the library keeps the memory alive by itself; this is for introspection)

    def set_labels(self, labels, which=capi.TRAIN):
        lab = np.ascontiguousarray(labels, dtype=np.int64)
        self._chk(self._lib.npbnn_set_labels_i64(self._ctx, lab.ctypes.data_as(C.POINTER(C.c_int64)), lab.shape[0], which))

    def set_targets(self, targets, which=capi.TRAIN):
        t = capi.as_f64(targets)
        if t.ndim == 1:
            t = t.reshape(-1, 1)
        self._chk(self._lib.npbnn_set_targets_f64(self._ctx, capi.dptr(t), t.shape[0], t.shape[1], which))

    def set_row_weights(self, instance_w=None, class_w=None):
        iw = None if instance_w is None else capi.as_f64(instance_w)
        cw = None if class_w is None or len(class_w) == 0 else capi.as_f64(class_w)
        self._chk(self._lib.npbnn_set_row_weights(self._ctx, capi.dptr(iw), 0 if iw is None else iw.shape[0],
                                                  capi.dptr(cw), 0 if cw is None else cw.shape[0]))

    def set_arch(self, in_dim, out_dims, has_bias, act_kind, out_kind, lik_kind, n_targets=0, final_activation=False):
        if len(out_dims) > capi.MAX_LAYERS:
            raise capi.NpbnnError(-1, "at most %d layers are supported" % capi.MAX_LAYERS)
        a = capi.Arch()
        a.n_layers = len(out_dims)
        a.in_dim = int(in_dim)
        for i, (o, b) in enumerate(zip(out_dims, has_bias)):
            a.out_dim[i] = int(o)
            a.has_bias[i] = int(bool(b))
        a.act_kind, a.out_kind, a.lik_kind, a.n_targets = int(act_kind), int(out_kind), int(lik_kind), int(n_targets)
        a.final_act = 1 if final_activation else 0
        self._chk(self._lib.npbnn_set_arch(self._ctx, C.byref(a)))
        self.arch = a
        self.n_out = int(out_dims[-1])

    def set_layer_mask(self, mask):
        """Declare the 0/1 mask of the network (list of per-layer matrices, a packed vector, or None for dense): blocks of the
        first layer in which it is all zero cost neither device memory nor matrix-core work (npbnn_set_layer_mask).  Weights
        passed afterwards must be zero where the mask is."""
        m = None if mask is None else (pack_weights(mask) if isinstance(mask, (list, tuple)) else capi.as_f64(mask))
        self._chk(self._lib.npbnn_set_layer_mask(self._ctx, capi.dptr(m)))

    def set_arch_from_weights(self, weights, in_dim, act_kind, out_kind, lik_kind, n_targets=0, final_activation=False):
        """Derive layer sizes and bias flags from the weight shapes: layer l has a
        bias iff its matrix has in_l + 1 columns (reference: BNN_lib.py:157-161)."""
        out_dims, has_bias = [], []
        cur = in_dim
        for w in weights:
            out_dims.append(w.shape[0])
            if w.shape[1] == cur:
                has_bias.append(0)
            elif w.shape[1] == cur + 1:
                has_bias.append(1)
            else:
                raise ValueError("weight matrix with %d columns does not match %d inputs" % (w.shape[1], cur))
            cur = w.shape[0]
        self.set_arch(in_dim, out_dims, has_bias, act_kind, out_kind, lik_kind, n_targets, final_activation)

    # -- hot path ---------------------------------------------------------------------
    def eval(self, weights, act_prm=None, col_override=None, lik_temp=1.0, sigma=None, which=capi.TRAIN,
             want_confusion=False):
        w = pack_weights(weights) if isinstance(weights, (list, tuple)) else capi.as_f64(weights)
        ap = None if act_prm is None else capi.as_f64(act_prm)
        co = None if col_override is None else capi.as_f64(col_override)
        sg = None if sigma is None else capi.as_f64(np.broadcast_to(sigma, (self.arch.n_targets,)))
        out = capi.EvalOut()
        conf = None
        cptr = None
        if want_confusion:
            conf = np.zeros((self.n_out, self.n_out), dtype=np.int64)
            cptr = conf.ctypes.data_as(C.POINTER(C.c_int64))
        self._chk(self._lib.npbnn_eval(self._ctx, capi.dptr(w), capi.dptr(ap), capi.dptr(co), float(lik_temp),
                                       capi.dptr(sg), which, C.byref(out), cptr))
        k = self.arch.n_targets
        return dict(loglik=out.loglik, sigma=np.array(out.sigma[:k]), sum_r=np.array(out.sum_r[:k]),
                    sum_r2=np.array(out.sum_r2[:k]), n_rows=out.n_rows, confusion=conf)

    def predict(self, weights, act_prm=None, col_override=None, which=capi.TRAIN, apply_out_fn=True):
        w = pack_weights(weights) if isinstance(weights, (list, tuple)) else capi.as_f64(weights)
        ap = None if act_prm is None else capi.as_f64(act_prm)
        co = None if col_override is None else capi.as_f64(col_override)
        y = np.empty((self.n_rows[which], self.n_out), dtype=np.float64)
        self._chk(self._lib.npbnn_predict(self._ctx, capi.dptr(w), capi.dptr(ap), capi.dptr(co), which,
                                          1 if apply_out_fn else 0, capi.dptr(y)))
        return y

    def predict_sets(self, weight_sets, act_prm_sets=None, which=capi.TRAIN, apply_out_fn=True):
        """Predictions of several weight sets on the resident matrix (npbnn_predict_sets): [n_sets, n_rows, n_out].
        ``weight_sets``: list of per-layer lists (or packed vectors); ``act_prm_sets``: per set the activation slopes
        of the hidden layers, or None."""
        packed = np.stack([pack_weights(w) if isinstance(w, (list, tuple)) else capi.as_f64(w).ravel() for w in weight_sets])
        packed = capi.as_f64(packed)
        n_sets = packed.shape[0]
        ap = None
        if act_prm_sets is not None and self.arch.n_layers > 1:
            ap = capi.as_f64(np.stack([np.asarray(a, dtype=np.float64).ravel()[: self.arch.n_layers - 1] for a in act_prm_sets]))
        out = np.empty((n_sets, self.n_rows[which], self.n_out), dtype=np.float64)
        self._chk(self._lib.npbnn_predict_sets(self._ctx, capi.dptr(packed), capi.dptr(ap), n_sets, which,
                                               1 if apply_out_fn else 0, capi.dptr(out)))
        return out

    def time_eval(self, weights, iters=20):
        w = pack_weights(weights) if isinstance(weights, (list, tuple)) else capi.as_f64(weights)
        a, b = C.c_double(0), C.c_double(0)
        self._chk(self._lib.npbnn_time_eval(self._ctx, capi.dptr(w), int(iters), C.byref(a), C.byref(b)))
        return a.value, b.value

    def time_pass(self, weights, n_candidates=0, iters=20):
        """Mean duration (ms) of the evaluation kernel of a speculative chain pass and the candidates it evaluates."""
        w = pack_weights(weights) if isinstance(weights, (list, tuple)) else capi.as_f64(weights)
        ms, used = C.c_double(0), C.c_int(0)
        self._chk(self._lib.npbnn_time_pass(self._ctx, capi.dptr(w), int(n_candidates), int(iters), C.byref(ms), C.byref(used)))
        return ms.value, used.value

    def time_wide(self, weights, iters=20):
        """Weight-streamed path: mean duration (ms) of the first layer's product alone and of a whole pass, and the product's
        geometry (rows, outputs of a workgroup's block, K-slices, workgroups)."""
        w = pack_weights(weights) if isinstance(weights, (list, tuple)) else capi.as_f64(weights)
        ms0, ms = C.c_double(0), C.c_double(0)
        info = (C.c_int * 4)()
        self._chk(self._lib.npbnn_time_wide(self._ctx, capi.dptr(w), int(iters), C.byref(ms0), C.byref(ms), info))
        return ms0.value, ms.value, dict(block_rows=info[0], block_outputs=info[1], k_slices=info[2], workgroups=info[3])

    def _fill_chain_cfg(self, cfg, prior_kind, prior_scale, w_bound, temperature, lik_temp, cur_loglik, cur_logprior,
                        cur_sigma=None, sigma=None, n_candidates=0, schedule=0, sigma_mult=None, hastings=None, slopes=None,
                        fixed_slopes=None):
        # the settings of a plain batch rarely change between dispatches: when they are the ones this struct already holds, only the
        # chain's state goes in
        plain = (slopes is None and sigma_mult is None and type(prior_scale) is np.ndarray and prior_scale.ndim == 1
                 and prior_scale.dtype == np.float64)
        key = None
        if plain:
            key = (prior_kind, prior_scale.tobytes(), w_bound, temperature, lik_temp, n_candidates, schedule, sigma is None, cur_sigma is None,
                   None if fixed_slopes is None else np.asarray(fixed_slopes, dtype=np.float64).tobytes())
            if cfg.__dict__.get("_plain_key") == key:
                cfg.cur_loglik, cfg.cur_logprior = cur_loglik, cur_logprior
                cfg.force_f32 = 0
                if sigma is not None or cur_sigma is not None:
                    k = self.arch.n_targets
                    if sigma is not None:
                        for j, v in enumerate(np.broadcast_to(sigma, (k,))):
                            cfg.sigma[j] = float(v)
                    if cur_sigma is not None:
                        for j, v in enumerate(np.broadcast_to(cur_sigma, (k,))):
                            cfg.cur_sigma[j] = float(v)
                return
        cfg._plain_key = key
        cfg.slope_idx = cfg.slope_delta = None
        cfg.n_slopes = cfg.slope_term_in_prior = 0
        cfg._keep_slopes = None
        if fixed_slopes is not None and slopes is None:      # ActFun("genReLU", prm=...): the hidden layers' slopes, the same for every proposal
            fixed = np.asarray(fixed_slopes, dtype=np.float64).ravel()
            cfg.n_slopes = len(fixed)
            for i, v in enumerate(fixed):
                cfg.cur_slopes[i] = float(v)
        if slopes is not None:        # (slope_idx [K] int32, slope_delta [K], accepted slopes, does cur_logprior hold their term?)
            s_idx = np.ascontiguousarray(slopes[0], dtype=np.int32)
            s_delta = capi.as_f64(slopes[1])
            cur = np.asarray(slopes[2], dtype=np.float64).ravel()
            cfg._keep_slopes = (s_idx, s_delta)
            cfg.slope_idx = s_idx.ctypes.data_as(C.POINTER(C.c_int32))
            cfg.slope_delta = capi.dptr(s_delta)
            cfg.n_slopes = len(cur)
            for i, v in enumerate(cur):
                cfg.cur_slopes[i] = float(v)
            cfg.slope_term_in_prior = 1 if slopes[3] else 0
        cfg.prior_kind = int(prior_kind)
        cfg._keep_scale = None
        cfg.prior_scale_w = None
        if any(np.ndim(s) != 0 for s in prior_scale):
            # hyper_p = 2 / 3: a scale per input node (one row, broadcast over the nodes) or per weight -> one per packed weight
            rows = list(self.arch.out_dim[:self.arch.n_layers])
            cols = [self.arch.in_dim + self.arch.has_bias[0]] + [rows[l - 1] + self.arch.has_bias[l] for l in range(1, self.arch.n_layers)]
            per_w = np.concatenate([np.broadcast_to(np.asarray(s, dtype=np.float64), (r, c)).ravel()
                                    for s, r, c in zip(prior_scale, rows, cols)])
            cfg._keep_scale = capi.as_f64(per_w)
            cfg.prior_scale_w = capi.dptr(cfg._keep_scale)
        else:
            for i, s in enumerate(prior_scale):
                cfg.prior_scale[i] = float(s)
        cfg.w_bound = float(w_bound)
        cfg.temperature = float(temperature)
        cfg.lik_temp = float(lik_temp)
        cfg.sigma_given = 0 if sigma is None else 1
        k = self.arch.n_targets
        if sigma is not None:
            for j, v in enumerate(np.broadcast_to(sigma, (k,))):
                cfg.sigma[j] = float(v)
        if cur_sigma is not None:
            for j, v in enumerate(np.broadcast_to(cur_sigma, (k,))):
                cfg.cur_sigma[j] = float(v)
        cfg.cur_loglik, cfg.cur_logprior = float(cur_loglik), float(cur_logprior)
        cfg.n_candidates = int(n_candidates)
        cfg.schedule = int(schedule)
        cfg.force_f32 = 0
        if sigma_mult is not None:         # (the arrays are kept on the struct object: it only holds pointers)
            cfg._keep = (capi.as_f64(sigma_mult), capi.as_f64(hastings))
            cfg.sigma_mult, cfg.hastings = capi.dptr(cfg._keep[0]), capi.dptr(cfg._keep[1])
        else:
            cfg._keep = None
            cfg.sigma_mult = cfg.hastings = None

    def _result_dict(self, res):
        k = self.arch.n_targets
        return dict(loglik=res.loglik, logprior=res.logprior, sigma=np.array(res.sigma[:k]) if k else _NO_SIGMA,
                    n_accepted=res.n_accepted, n_passes=res.n_passes, n_candidates=res.n_candidates,
                    n_void_passes=res.n_void_passes, schedule=res.schedule, temperature=res.temperature,
                    iterations_done=res.iterations_done, overflow=res.overflow, slopes=res.slopes)

    def chain_run(self, weights, idx, delta, cnt, log_u, prior_kind, prior_scale, w_bound, temperature, lik_temp,
                  cur_loglik, cur_logprior, cur_sigma=None, sigma=None, mask=None, n_candidates=0, schedule=0, sigma_mult=None,
                  hastings=None, slopes=None, fixed_slopes=None):
        """K device-resident Metropolis-Hastings iterations (npbnn_chain_run).  Returns
        (new packed weights, accepted flags, proposed logLik, proposed logPrior, result dict)."""
        w = pack_weights(weights) if isinstance(weights, (list, tuple)) else capi.as_f64(weights).copy()
        K, M = idx.shape
        cfg = self.__dict__.get("_chain_cfg")
        if cfg is None:
            cfg = self._chain_cfg = capi.ChainCfg()
            self._chain_res = capi.ChainResult()
        self._fill_chain_cfg(cfg, prior_kind, prior_scale, w_bound, temperature, lik_temp, cur_loglik, cur_logprior,
                             cur_sigma, sigma, n_candidates, schedule, sigma_mult, hastings, slopes, fixed_slopes)
        m = None if mask is None else (pack_weights(mask) if isinstance(mask, (list, tuple)) else capi.as_f64(mask))
        if idx.dtype != np.int32 or not idx.flags.c_contiguous:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
        if cnt.dtype != np.int32 or not cnt.flags.c_contiguous:
            cnt = np.ascontiguousarray(cnt, dtype=np.int32)
        delta = capi.as_f64(delta)
        log_u = capi.as_f64(log_u)
        acc = np.empty(K, dtype=np.uint8)
        llp, lpp = np.empty(K), np.empty(K)
        res = self._chain_res
        run = capi.chain_run_by_address(self._lib)
        addr = (_addr(w), None if m is None else _addr(m), _addr(idx), _addr(delta), _addr(cnt), _addr(log_u), _addr(acc), _addr(llp),
                _addr(lpp))
        attempt, sync_retried = 0, False
        while True:
            cfg.force_f32 = attempt
            t_in = _now()
            rc = run(self._ctx, C.byref(cfg), addr[0], addr[1], K, M, addr[2], addr[3], addr[4], addr[5], addr[6], addr[7], addr[8],
                     C.byref(res))
            self.seconds_in_chain_run += _now() - t_in          # (what a dispatch spends inside the library: tools/profile_dispatch.py)
            if rc == capi.E_RANGE and attempt == 0:
                attempt = 1
                continue        # a weight left the fp16 range: same batch again on the float32 path (state untouched)
            if rc == capi.E_SYNC and not sync_retried:
                sync_retried = True
                if cfg.schedule in (capi.SCHED_OVERLAP2, capi.SCHED_PERSIST, capi.SCHED_PERSIST_SERIAL):
                    cfg.schedule = capi.SCHED_OVERLAP
                    cfg._plain_key = None      # (the struct no longer holds what the caller asked for)
                self.sync_fallbacks += 1
                if self.sync_fallbacks == 1:
                    import warnings
                    warnings.warn("npbnn_amd: a device-side wait of the flag-ordered (two-stream / persistent) chain schedule timed out; the batch is repeated on "
                                  "one stream and this context stays on one stream from now on (HipContext.sync_fallbacks counts them)")
                continue        # (the state was left untouched)
            self._chk(rc)
            break
        return w, acc, llp, lpp, self._result_dict(res)


    def chain_run_general(self, weights, draws, log_u, mask=None, indicators=None, feature_indicators=None, feature_means=None,
                          prior_ind1=0.5, has_indicator_prior=False, **cfg_kw):
        """K iterations of the general device chain (npbnn_chain_run_general).  ``draws``: dict with ``kind`` (PROP_*), ``idx``,
        ``val`` [K, M], ``cnt`` [K], ``layer_mask`` [K] and, as the sampler's settings need them, ``h_idx`` / ``h_val`` / ``h_fac`` /
        ``h_cnt`` (every draw of the fixed-normal proposal), ``ind_ptr`` / ``ind_pos`` (weight-indicator flips), ``find_ptr`` /
        ``find_pos`` / ``find_use`` (feature-indicator flips).  Returns (weights, indicators, feature indicators, accepted flags,
        proposed logLik, proposed logPrior, result dict)."""
        i32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)       # noqa: E731
        ip = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))        # noqa: E731
        w = pack_weights(weights) if isinstance(weights, (list, tuple)) else capi.as_f64(weights).copy()
        idx, cnt = i32(draws["idx"]), i32(draws["cnt"])
        val = capi.as_f64(draws["val"])
        K, M = idx.shape
        cfg, res, g = capi.ChainCfg(), capi.ChainResult(), capi.GeneralCfg()
        self._fill_chain_cfg(cfg, **cfg_kw)
        g.proposal_kind, g.M = int(draws["kind"]), int(M)
        keep = [idx, cnt, val]
        g.idx, g.val, g.cnt = ip(idx), capi.dptr(val), ip(cnt)
        lm = i32(draws.get("layer_mask") if draws.get("layer_mask") is not None else np.zeros(K))
        g.layer_mask = ip(lm)
        keep.append(lm)
        if draws.get("h_idx") is not None:
            h_idx, h_cnt = i32(draws["h_idx"]), i32(draws["h_cnt"])
            h_val, h_fac = capi.as_f64(draws["h_val"]), capi.as_f64(draws["h_fac"])
            g.h_idx, g.h_val, g.h_fac, g.h_cnt = ip(h_idx), capi.dptr(h_val), capi.dptr(h_fac), ip(h_cnt)
            keep += [h_idx, h_cnt, h_val, h_fac]
        ind = find = None
        if indicators is not None:
            ind = capi.as_f64(indicators).copy()
            iptr, ipos = i32(draws["ind_ptr"]), i32(draws["ind_pos"] if len(draws["ind_pos"]) else np.zeros(1))
            g.ind_inout, g.ind_ptr, g.ind_pos = capi.dptr(ind), ip(iptr), ip(ipos)
            g.prior_ind1, g.has_indicator_prior = float(prior_ind1), 1 if has_indicator_prior else 0
            keep += [iptr, ipos]
        if feature_indicators is not None:
            find = capi.as_f64(feature_indicators).copy()
            means = capi.as_f64(feature_means)
            fptr, fpos, fuse = i32(draws["find_ptr"]), i32(draws["find_pos"] if len(draws["find_pos"]) else np.zeros(1)), i32(draws["find_use"])
            g.find_inout, g.feature_means, g.find_ptr, g.find_pos, g.find_use = capi.dptr(find), capi.dptr(means), ip(fptr), ip(fpos), ip(fuse)
            keep += [means, fptr, fpos, fuse]
        m = None if mask is None else (pack_weights(mask) if isinstance(mask, (list, tuple)) else capi.as_f64(mask))
        log_u = capi.as_f64(log_u)
        acc = np.empty(K, dtype=np.uint8)
        llp, lpp = np.empty(K), np.empty(K)
        self._chk(self._lib.npbnn_chain_run_general(self._ctx, C.byref(cfg), C.byref(g), capi.dptr(w), capi.dptr(m), K, capi.dptr(log_u),
                                                    acc.ctypes.data_as(C.POINTER(C.c_uint8)), capi.dptr(llp), capi.dptr(lpp), C.byref(res)))
        del keep
        return w, ind, find, acc, llp, lpp, self._result_dict(res)


class FastBatch:
    """A dispatch of the patch-list device chain whose settings are fixed: the settings struct filled once, the result struct, the
    accept flags and the packed mask kept, the C entry bound once - :meth:`run` only puts in the chain's state and the addresses of
    this batch's draws.  (What ``HipContext.chain_run`` does per call, minus everything that does not change between calls.)"""

    def __init__(self, ctx, K, M, mask, cfg_kwargs):
        self.ctx = ctx
        self.K, self.M = int(K), int(M)
        self.cfg = capi.ChainCfg()
        ctx._fill_chain_cfg(self.cfg, **cfg_kwargs)
        self.cfg_ref = C.byref(self.cfg)
        self.res = capi.ChainResult()
        self.res_ref = C.byref(self.res)
        self.acc = np.empty(self.K, dtype=np.uint8)
        self.acc_addr = _addr(self.acc)
        self.mask = None if mask is None else (pack_weights(mask) if isinstance(mask, (list, tuple)) else capi.as_f64(mask))
        self.mask_addr = None if self.mask is None else _addr(self.mask)
        self.n_targets = ctx.arch.n_targets
        self.entry = capi.chain_run_by_address(ctx._lib)

    def run(self, w, idx, delta, cnt, log_u, cur_loglik, cur_logprior, temperature, cur_sigma=None, just_before=None):
        """0 and the results in ``self.res`` / ``self.acc`` / ``w`` - or the C ABI's error code with the state untouched (a weight
        left the fp16 range, a device-side wait timed out ...: the caller repeats the batch through ``chain_run``, which knows the
        remedies).  ``just_before()`` is called as the last thing before the C entry: work for another thread is handed over there,
        so that the other thread finds the interpreter lock free (the C call releases it) instead of taking it from this one in
        the middle of its preparations (measured: copies of the weight vector that took 300 us instead of 4)."""
        cfg = self.cfg
        cfg.cur_loglik, cfg.cur_logprior = cur_loglik, cur_logprior
        cfg.temperature = temperature        # (state, not a setting: MC3 swaps it between two dispatches)
        if cur_sigma is not None:
            for j in range(self.n_targets):
                cfg.cur_sigma[j] = cur_sigma[j]
        ctx = self.ctx
        a_w, a_idx, a_delta, a_cnt, a_u = _addr(w), _addr(idx), _addr(delta), _addr(cnt), _addr(log_u)
        if just_before is not None:
            just_before()
        t_in = _now()
        rc = self.entry(ctx._ctx, self.cfg_ref, a_w, self.mask_addr, self.K, self.M, a_idx, a_delta, a_cnt, a_u, self.acc_addr, None, None,
                        self.res_ref)
        ctx.seconds_in_chain_run += _now() - t_in
        return rc


def chains_run_exchange(jobs, n_chains, seg_len, n_seg, swap_j, swap_k, swap_logu, comm=None, launch_slack=1.25,
                        want_cold_w=True):
    """Several chains advance ``n_seg`` swap intervals of ``seg_len`` iterations with the temperature swaps done on the GPU
    (npbnn_chains_run_exchange; reference loop: np_bnn/BNN_mc3.py:94-112).

    ``jobs``: one dict per chain of this process - ``ctx`` (HipContext), ``chain_id``, ``weights``, ``idx``, ``delta``,
    ``cnt``, ``log_u`` (K = n_seg * seg_len rows), ``mask`` and the keyword arguments of :meth:`HipContext.chain_run`
    under ``cfg``.  ``comm``: the native communicator handle (``RcclComm._comm``) or None for the chains of this process.

    Returns ``(outs, records, segments_done)``: per job ``dict(w, accepted, loglik_prop, logprior_prop, state, cold_w,
    result)`` valid for the first ``result['iterations_done']`` iterations; ``records[s, i] = (logPost, temperature before
    swap s, finished flag, iterations done)`` of chain i; ``segments_done < n_seg`` when some chain fell short."""
    K = int(seg_len) * int(n_seg)
    lib = jobs[0]["ctx"]._lib
    arr = (capi.ChainJob * len(jobs))()
    keep, outs = [], []
    i32p = C.POINTER(C.c_int32)
    for q, job in enumerate(jobs):
        ctx = job["ctx"]
        weights = job["weights"]
        w = pack_weights(weights) if isinstance(weights, (list, tuple)) else capi.as_f64(weights).copy()
        idx, cnt = job["idx"], job["cnt"]
        if idx.shape[0] != K or len(cnt) != K:
            raise ValueError("job %d: %d rows of draws for %d x %d iterations" % (q, idx.shape[0], n_seg, seg_len))
        if idx.dtype != np.int32 or not idx.flags.c_contiguous:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
        if cnt.dtype != np.int32 or not cnt.flags.c_contiguous:
            cnt = np.ascontiguousarray(cnt, dtype=np.int32)
        delta, log_u = capi.as_f64(job["delta"]), capi.as_f64(job["log_u"])
        mask = job.get("mask")
        m = None if mask is None else (pack_weights(mask) if isinstance(mask, (list, tuple)) else capi.as_f64(mask))
        cfg, res = capi.ChainCfg(), capi.ChainResult()
        ctx._fill_chain_cfg(cfg, **job["cfg"])
        acc = np.zeros(K, dtype=np.uint8)
        llp, lpp = np.zeros(K), np.zeros(K)
        state = np.zeros((n_seg, capi.XSTATE_DOUBLES))
        cold = np.zeros((n_seg, w.size)) if want_cold_w else None
        J = arr[q]
        J.ctx = ctx._ctx
        J.cfg = C.pointer(cfg)
        J.W_inout = capi.dptr(w)
        J.mask_packed = capi.dptr(m)
        J.M = idx.shape[1]
        J.chain_id = int(job["chain_id"])
        J.idx = idx.ctypes.data_as(i32p)
        J.delta = capi.dptr(delta)
        J.cnt = cnt.ctypes.data_as(i32p)
        J.log_u = capi.dptr(log_u)
        J.out_accepted = acc.ctypes.data_as(C.POINTER(C.c_uint8))
        J.out_loglik_prop = capi.dptr(llp)
        J.out_logprior_prop = capi.dptr(lpp)
        J.out_state = capi.dptr(state)
        J.out_cold_w = capi.dptr(cold)
        J.result = C.pointer(res)
        keep.append((w, idx, cnt, delta, log_u, m, cfg, res))
        outs.append(dict(w=w, accepted=acc, loglik_prop=llp, logprior_prop=lpp, state=state, cold_w=cold, _res=res, _ctx=ctx))
    sj = np.ascontiguousarray(swap_j, dtype=np.int32)
    sk = np.ascontiguousarray(swap_k, dtype=np.int32)
    su = capi.as_f64(swap_logu)
    records = np.zeros((n_seg, n_chains, capi.REC_DOUBLES))
    done = C.c_int32(0)
    rc = lib.npbnn_chains_run_exchange(comm, arr, len(jobs), int(n_chains), int(seg_len), int(n_seg), sj.ctypes.data_as(i32p),
                                       sk.ctypes.data_as(i32p), capi.dptr(su), float(launch_slack), capi.dptr(records),
                                       C.byref(done))
    capi.check(lib, None, rc)
    for o in outs:
        o["result"] = o.pop("_ctx")._result_dict(o.pop("_res"))
    return outs, records, int(done.value)


def chains_run_batched(jobs, K):
    """2 or 3 chains of one model on one GPU advance ``K`` iterations each, one proposal per chain per streaming read of the
    feature matrix (npbnn_chains_run_batched).  ``jobs``: dicts as for :func:`chains_run_exchange` (``chain_id`` not needed).
    Returns per job ``dict(w, accepted, loglik_prop, logprior_prop, result)``.  A weight leaving the fp16 range sends the whole
    group through once more on the float32 layer-0 path."""
    K = int(K)
    lib = jobs[0]["ctx"]._lib
    i32p = C.POINTER(C.c_int32)
    for attempt in (0, 1):
        arr = (capi.ChainJob * len(jobs))()
        keep, outs = [], []
        for q, job in enumerate(jobs):
            ctx = job["ctx"]
            weights = job["weights"]
            w = pack_weights(weights) if isinstance(weights, (list, tuple)) else capi.as_f64(weights).copy()
            idx, cnt = job["idx"], job["cnt"]
            if idx.shape[0] != K or len(cnt) != K:
                raise ValueError("job %d: %d rows of draws for %d iterations" % (q, idx.shape[0], K))
            if idx.dtype != np.int32 or not idx.flags.c_contiguous:
                idx = np.ascontiguousarray(idx, dtype=np.int32)
            if cnt.dtype != np.int32 or not cnt.flags.c_contiguous:
                cnt = np.ascontiguousarray(cnt, dtype=np.int32)
            delta, log_u = capi.as_f64(job["delta"]), capi.as_f64(job["log_u"])
            mask = job.get("mask")
            m = None if mask is None else (pack_weights(mask) if isinstance(mask, (list, tuple)) else capi.as_f64(mask))
            cfg, res = capi.ChainCfg(), capi.ChainResult()
            ctx._fill_chain_cfg(cfg, **job["cfg"])
            cfg.force_f32 = attempt
            acc = np.zeros(K, dtype=np.uint8)
            llp, lpp = np.zeros(K), np.zeros(K)
            J = arr[q]
            J.ctx = ctx._ctx
            J.cfg = C.pointer(cfg)
            J.W_inout = capi.dptr(w)
            J.mask_packed = capi.dptr(m)
            J.M = idx.shape[1]
            J.chain_id = q
            J.idx = idx.ctypes.data_as(i32p)
            J.delta = capi.dptr(delta)
            J.cnt = cnt.ctypes.data_as(i32p)
            J.log_u = capi.dptr(log_u)
            J.out_accepted = acc.ctypes.data_as(C.POINTER(C.c_uint8))
            J.out_loglik_prop = capi.dptr(llp)
            J.out_logprior_prop = capi.dptr(lpp)
            J.result = C.pointer(res)
            keep.append((w, idx, cnt, delta, log_u, m, cfg, res))
            outs.append(dict(w=w, accepted=acc, loglik_prop=llp, logprior_prop=lpp, _res=res, _ctx=ctx))
        rc = lib.npbnn_chains_run_batched(arr, len(jobs), K)
        if rc == capi.E_RANGE and attempt == 0:
            continue
        capi.check(lib, jobs[0]["ctx"]._ctx, rc)
        break
    for o in outs:
        o["result"] = o.pop("_ctx")._result_dict(o.pop("_res"))
    return outs
