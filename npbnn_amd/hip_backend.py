"""Binding between a model object (``npBNN``) and its resident device context.

``HipBackend`` is the only evaluation backend the package ships.  It uploads the training / test
matrices once, describes the network to the C ABI and then serves the sampler:

    evaluate(weights, ...) -> log-likelihood (+ sigma, residual moments, confusion counts)
    predict(weights, ...)  -> prediction matrix

The sampler accepts any object with these two methods (the CPU test-suite injects an oracle-backed
one); when none is given it builds a ``HipBackend`` and fails loudly if the HIP library or the GPU
is missing.
"""
import os
import weakref

import numpy as np

from . import _capi as capi
from .backend import HipContext
from .layers import output_kind
from .likelihoods import likelihood_kind

_FUSED = (capi.LIK_CATEGORICAL, capi.LIK_GAUSS, capi.LIK_GAUSS_PRED_SIGMA, capi.LIK_POISSON, capi.LIK_NEGBIN,
          capi.LIK_NEGBIN2D, capi.LIK_NEGBIN_BASE10)
_TARGET_KINDS = _FUSED[1:]


def bias_flags(weights, n_features):
    """Layer l has a bias iff its matrix has in_l + 1 columns (bias = column 0;
    reference: BNN_lib.py:157-161)."""
    flags, cur = [], n_features
    for w in weights:
        if w.shape[1] == cur:
            flags.append(0)
        elif w.shape[1] == cur + 1:
            flags.append(1)
        else:
            raise ValueError("weight matrix with %d columns does not match %d inputs" % (w.shape[1], cur))
        cur = w.shape[0]
    return flags


# backends by the host arrays whose device copies they hold: models that hold the very same data / test_data array objects (the
# per-chain replicas MC3 makes of one model) share one resident copy per device
_DATA_OWNERS = weakref.WeakValueDictionary()


class HipBackend:
    def __init__(self, bnn, likelihood_f=None, device=None):
        self.ctx = HipContext(device)
        self.n_features = bnn._data.shape[1]
        self.has_test = len(bnn._test_data) > 0
        self._host_arrays = (bnn._data, bnn._test_data)          # (kept: the sharing key below is their identity)
        key = (id(bnn._data), id(bnn._test_data) if self.has_test else 0, bnn._data.shape,
               np.shape(bnn._test_data) if self.has_test else (), self.ctx.device)
        owner = _DATA_OWNERS.get(key)
        if owner is not None and getattr(owner.ctx, "_ctx", None) and os.environ.get("NPBNN_NO_DATA_SHARING") is None:
            self.ctx.share_data(owner.ctx)
            self.data_shared_with = owner
        else:
            self.ctx.set_data(bnn._data, capi.TRAIN)
            if self.has_test:
                self.ctx.set_data(bnn._test_data, capi.TEST)
            self.data_shared_with = None
            _DATA_OWNERS[key] = self
        self.out_kind = output_kind(bnn._output_act_fun)          # None -> host callable
        lik = likelihood_kind(likelihood_f) if likelihood_f is not None else None
        self.lik_kind = lik if lik in _FUSED else capi.LIK_NONE
        self.fused_likelihood = self.lik_kind != capi.LIK_NONE
        self.n_targets = 0
        classification = bnn._estimation_mode == "classification"
        if classification:
            self.ctx.set_labels(bnn._labels, capi.TRAIN)
            if self.has_test and len(bnn._test_labels) > 0:
                self.ctx.set_labels(bnn._test_labels, capi.TEST)
        elif self.lik_kind in _TARGET_KINDS:
            self.n_targets = bnn._labels.shape[1]
            self.ctx.set_targets(bnn._labels, capi.TRAIN)
            if self.has_test and len(bnn._test_labels) > 0:
                self.ctx.set_targets(bnn._test_labels, capi.TEST)
        if self.lik_kind == capi.LIK_CATEGORICAL:
            iw = bnn._instance_weights
            cw = bnn._class_w if len(bnn._class_w) else None
            if iw is not None or cw is not None:
                if iw is not None and cw is not None:
                    # the reference's combined branch raises (np.sum(..., axis=1) on a vector, BNN_lib.py:105)
                    raise np.exceptions.AxisError("axis 1 is out of bounds for array of dimension 1")
                self.ctx.set_row_weights(instance_w=iw, class_w=cw)
        self._shapes = None
        self._act = bnn._act_fun
        self._model = weakref.ref(bnn)          # (the model owns this object: no cycle)
        self._mask_seen = None
        self._configure(bnn._w_layers)
        self._pinned = None

    def host_empty(self, shape, dtype):
        """Page-locked host array for data on its way to the device (pre-drawn proposals)."""
        if self._pinned is None:
            from .pinned import PinnedPool
            self._pinned = PinnedPool(self.ctx._lib)
        return self._pinned.empty(shape, dtype)

    def host_empty_group(self, specs):
        """Several page-locked arrays in one block (PinnedPool.empty_group)."""
        if self._pinned is None:
            from .pinned import PinnedPool
            self._pinned = PinnedPool(self.ctx._lib)
        return self._pinned.empty_group(specs)

    def _configure(self, weights):
        shapes = tuple(w.shape for w in weights)
        # slopes that the sampler proposes AND the forward pass uses (ActFun(fun="genReLU", trainable=True)): the device chain gives
        # every candidate its own, which needs slots in the weight image
        self.ctx.set_trainable_slopes(bool(self._act._trainable) and self._act._function == "genReLU")
        if shapes != self._shapes:
            flags = bias_flags(weights, self.n_features)
            self.ctx.set_arch(self.n_features, [s[0] for s in shapes], flags, self._act.device_kind(),
                              capi.OUT_IDENTITY if self.out_kind is None else self.out_kind, self.lik_kind,
                              self.n_targets)
            self._shapes = shapes
            self._mask_seen = None
        # block structure of the first layer (npBNN.apply_mask): the device stores and multiplies only the blocks the mask keeps
        model = self._model()
        mask = getattr(model, "_mask", None) if model is not None else None
        if mask is not self._mask_seen:
            ok = mask is not None and len(mask) == len(shapes) and all(np.shape(m) == sh for m, sh in zip(mask, shapes))
            self.ctx.set_layer_mask(mask if ok else None)
            self._mask_seen = mask

    def evaluate(self, weights, slopes=None, col_override=None, lik_temp=1.0, sigma=None, which=capi.TRAIN,
                 want_confusion=False):
        self._configure(weights)
        return self.ctx.eval(weights, act_prm=slopes, col_override=col_override, lik_temp=lik_temp, sigma=sigma,
                             which=which, want_confusion=want_confusion)

    def predict(self, weights, slopes=None, col_override=None, which=capi.TRAIN, apply_out_fn=True):
        self._configure(weights)
        return self.ctx.predict(weights, act_prm=slopes, col_override=col_override, which=which,
                                apply_out_fn=apply_out_fn and self.out_kind is not None)

    def train_rows(self):
        return int(self.ctx.n_rows[capi.TRAIN])

    def refresh_row_weights(self, bnn):
        """The model's class weights changed (a row-sharded sampler recomputes them from all ranks' label counts)."""
        if self.lik_kind == capi.LIK_CATEGORICAL and len(bnn._class_w):
            self.ctx.set_row_weights(instance_w=bnn._instance_weights, class_w=bnn._class_w)

    def set_row_shard(self, sharded):
        """This context holds one rank's share of the rows (npbnn_amd.rowshard.RowShardedBackend): npbnn_chain_run gathers the
        per-pass sums of all ranks before every step - on the stream through RCCL when the communicator offers a handle on this
        GPU, else through the host."""
        self._sharded = sharded             # (keeps the gather callback alive as long as the context may call it)
        self.ctx.set_row_shard(sharded.rccl_handle(), None if sharded.rccl_handle() else sharded.gather_callback(),
                               sharded.rank, sharded.world, sharded.n_rows_total)

    def run_chain(self, weights, **kw):
        """Device-resident Metropolis-Hastings iterations; see HipContext.chain_run."""
        self._configure(weights)
        return self.ctx.chain_run(weights, **kw)

    def run_chain_general(self, weights, **kw):
        """Iterations of the general device chain; see HipContext.chain_run_general."""
        self._configure(weights)
        return self.ctx.chain_run_general(weights, **kw)

    exchange_slack = 1.5     # launches given to a swap interval of an exchange run, relative to the expected number (adapts)
    exchange_slack_floor = 1.15

    def exchange_job(self, weights, chain_id, idx, delta, cnt, log_u, mask, cfg):
        """This chain's share of an exchange run (see :func:`npbnn_amd.backend.chains_run_exchange`)."""
        self._configure(weights)
        return dict(ctx=self.ctx, chain_id=chain_id, weights=weights, idx=idx, delta=delta, cnt=cnt, log_u=log_u, mask=mask, cfg=cfg)

    @staticmethod
    def run_exchange(jobs, n_chains, seg_len, n_seg, swap_j, swap_k, swap_logu, comm=None, launch_slack=1.25, want_cold_w=True):
        from .backend import chains_run_exchange
        return chains_run_exchange(jobs, n_chains, seg_len, n_seg, swap_j, swap_k, swap_logu, comm=comm,
                                   launch_slack=launch_slack, want_cold_w=want_cold_w)

    @property
    def group_size(self):
        """Chains a group pass evaluates together: the candidate slots one launch of the evaluation kernel has for this network."""
        return max(1, self.ctx.info(capi.INFO_MAX_CANDIDATES))

    @staticmethod
    def run_batched(jobs, K):
        from .backend import chains_run_batched
        return chains_run_batched(jobs, K)

    def close(self):
        self.ctx.close()
