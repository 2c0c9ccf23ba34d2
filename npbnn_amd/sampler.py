"""``MCMC`` — the Metropolis-Hastings state machine, with the forward pass + likelihood on the GPU.

Host-side mirror of the reference sampler (np_bnn/BNN_env.py:273-550): same constructor, the same
``mh_step(bnn_obj, additional_prob=0, return_bnn=False)`` contract, the same public attributes and
the same consumption of the numpy ``Generator`` stream (so a chain draws the very proposals the
reference draws for the same seed).  What changed is where the arithmetic happens:

  * the per-layer forward loop, output function and likelihood of a proposal (BNN_env.py:449-491)
    are ONE fused device evaluation (``backend.evaluate``);
  * accuracy statistics and prediction matrices (``_accuracy``, ``_label_acc``, ``_label_freq``,
    ``_test_accuracy``, ``_y``, ``_y_test``; BNN_env.py:507-518) are produced on demand from the
    accepted weights instead of after every accepted step: they are functions of the accepted state
    only, so the values are the same, and nothing is downloaded while nobody looks.
"""
import _thread
import functools
import operator
import os
import queue
import threading

import numpy as np

from . import _capi as capi
from .backend import note_packed_views, pack_weights
from .likelihoods import (CalcAccuracy, CalcAccuracyRegression, CalcLabelAccuracy,
                          CalcLabelAccuracyRegression, SkipAccuracy, SkipAccuracyVec, calc_likelihood,
                          calc_likelihood_regression, calc_likelihood_regression_error, likelihood_kind,
                          stat_kind, stats_from_confusion)
from .model import data_transform_obj, npBNN
from .proposals import (UpdateBinomial, UpdateFixedNormal, UpdateNormal, UpdateNormal1D, UpdateNormalNormalized,
                        multiplier_proposal_vector)

_LAZY = ("_y", "_y_test", "_accuracy", "_test_accuracy", "_label_acc", "_label_freq")
# proposal kernels the general device chain knows (npbnn_general_cfg.proposal_kind)
_GENERAL_PROPOSALS = {UpdateNormal: capi.PROP_NORMAL, UpdateFixedNormal: capi.PROP_FIXED_NORMAL,
                      UpdateNormalNormalized: capi.PROP_NORMAL_NORMALIZED}

_POOL = None


class _DrawJob:
    """A call handed to the helper thread and, later, its outcome (``result()`` waits for it).  What a dispatch needs of a
    concurrent.futures.Future, at a tenth of its cost per use."""
    __slots__ = ("fn", "done", "value", "error", "saved", "queued")

    def __init__(self, fn):
        self.fn = fn
        self.done = _thread.allocate_lock()
        self.done.acquire()
        self.value = self.error = None
        self.queued = False       # handed to the helper thread (a job nobody queued never completes: do not wait for it)
        self.saved = None         # state of the chain's generator before the draw, when the draw can be taken back (the job fills it in)

    def run(self):
        try:
            self.value = self.fn()
        except BaseException as e:      # noqa: BLE001 - (handed to whoever asks for the result)
            self.error = e
        finally:
            self.fn = None
            self.done.release()

    def result(self):
        self.done.acquire()
        self.done.release()
        if self.error is not None:
            raise self.error
        return self.value


class _DrawThread:
    """One helper thread that pre-draws the next sub-batch of random numbers while the GPU runs the current one
    (both the pre-draw and the device call release the GIL)."""

    def __init__(self):
        self._jobs = queue.SimpleQueue()
        self._thread = threading.Thread(target=self._serve, name="npbnn-predraw", daemon=True)
        self._thread.start()

    def _serve(self):
        while True:
            self._jobs.get().run()

    def submit(self, fn, *args):
        job = _DrawJob(functools.partial(fn, *args) if args else fn)
        job.queued = True
        self._jobs.put(job)
        return job

    def enqueue(self, job):
        job.queued = True
        self._jobs.put(job)


def _draw_pool():
    global _POOL
    if _POOL is None:
        _POOL = _DrawThread()
    return _POOL


_is = operator.is_


def _same_ends(copy, live):
    """Cheap test for an in-place edit of a step-size matrix since it was copied (``mcmc._update_ws[i] *= 0.5``): such edits
    rescale the whole matrix, so its first and last entries tell; a full comparison per dispatch costs more than the dispatch's
    other book-keeping together."""
    a, b = copy.flat, live.flat
    return a[0] == b[0] and a[-1] == b[-1]


def _make_backend(bnn_obj, likelihood_f):
    """Builds the device context of a model.  The one seam of the package: the CPU test-suite points this name at a stand-in
    served by the oracle (tests/oracle_backend.py) to exercise the host logic without a GPU; the product only ever builds
    the HIP backend, which fails loudly when the library or the GPU is missing."""
    from .hip_backend import HipBackend
    return HipBackend(bnn_obj, likelihood_f)


def get_backend(bnn_obj, likelihood_f):
    """The model's resident device context (created on first use, shared by every sampler built on
    this model object)."""
    be = bnn_obj.__dict__.get("_npbnn_backend")
    if be is None or getattr(be, "_lik_f", None) is not likelihood_f:
        be = _make_backend(bnn_obj, likelihood_f)
        be._lik_f = likelihood_f
        bnn_obj.__dict__["_npbnn_backend"] = be
    return be


def _enqueue_draws(jobs):
    pool = _draw_pool()
    for job in jobs:
        pool.enqueue(job)


class _FastDispatch:
    """A repeat of the dispatch before it: ``run_steps(bnn, k)`` on the patch-list device chain, called again with the same
    ``k`` while nothing the dispatch depends on has changed - the call pattern of MC3's workers and of ``run_mcmc`` between two
    samples.  Everything such a dispatch derives from the sampler's settings (which device chain, batch cuts, the pre-draw plan,
    the C ABI's settings struct, buffers) is kept here; a dispatch then costs one comparison of what the settings are now with
    what they were, the wait for the draws made ahead, the request for the next ones, the C call, and the book-keeping of its
    outcome.  Any difference - another ``k``, an attribute set from outside, new step-size or weight arrays, an adaptation
    boundary ahead, an error code from the library - sends the call down :meth:`MCMC.run_steps`'s general path, which builds a
    new one when it applies.  The general path and this one leave the sampler in the same state, draws made ahead included."""
    __slots__ = ("missed_on_boundary", "k", "objects", "scalars", "n_bytes", "freq_bytes", "scale_bytes", "ws_src", "ws_copies", "plan", "key_tail", "batch",
                 "layers", "regression", "fixed_sigma", "n_out", "adapt_possible", "empty", "empty_group", "fixed_slopes")

    @staticmethod
    def _objects(mcmc, bnn):
        return (bnn, mcmc._backend, bnn._mask, bnn._act_fun, bnn._prior_scale, mcmc._update_ws, mcmc._update_n, mcmc._freq_layer_update,
                mcmc.update_function, mcmc._likelihood_f, None if mcmc._randomize_seed else mcmc._gen)

    @staticmethod
    def _scalars(mcmc, bnn, k):
        return (k, mcmc._lik_temp, bnn._w_bound, mcmc.n_candidates, mcmc.device_schedule, mcmc._randomize_seed,
                mcmc._mcmc_id, bnn._prior, mcmc._sample_from_prior, mcmc._adapt_f, mcmc._adapt_fM, mcmc._adapt_freq, mcmc._adapt_stop,
                bnn._estimation_mode, bnn._empirical_error, bnn._freq_indicator, bnn._feature_indicators is None,
                bnn._act_fun._trainable, mcmc._estimate_error, mcmc.SUB_BATCH)

    @classmethod
    def build(cls, mcmc, bnn, k):
        """The fast dispatch for ``run_steps(bnn, k)`` as the sampler stands, or None where the general path must stay: anything
        but plain batches of the patch-list chain (trainable slopes, an estimated sigma, indicators, rows split over ranks), a
        call longer than one sub-batch, a backend without the C ABI underneath (the CPU test stand-ins)."""
        be = mcmc._backend
        ctx = getattr(be, "ctx", None)
        cache = mcmc._ws_copies
        if ctx is None or cache is None or not 1 <= k <= mcmc.SUB_BATCH or getattr(be, "row_sharded", False):
            return None
        regression = bnn._estimation_mode == "regression"
        # (a regression whose sigma is estimated: kept only while sigma is still fixed at 1 - BNN_env.py:375-379,435-444 -, i.e. while the
        # call ends at or before iteration _estimate_error; the proposals on sigma that follow bring per-batch arrays of their own)
        fixed_sigma = regression and not bnn._empirical_error
        if fixed_sigma and mcmc._current_iteration + k - 1 > mcmc._estimate_error:
            return None
        if bnn._act_fun._trainable or mcmc._device_mode(bnn, k) != "patch":
            return None
        if type(bnn._prior_scale) is not np.ndarray or bnn._prior_scale.ndim != 1:
            return None
        from .backend import FastBatch
        self = cls()
        self.k = k
        self.objects = cls._objects(mcmc, bnn)
        self.scalars = cls._scalars(mcmc, bnn, k)
        if type(mcmc._update_n) is not np.ndarray or type(mcmc._freq_layer_update) is not np.ndarray:
            return None
        self.n_bytes = mcmc._update_n.tobytes()
        self.freq_bytes = mcmc._freq_layer_update.tobytes()
        self.scale_bytes = bnn._prior_scale.tobytes()
        self.fixed_slopes = mcmc._accepted_slopes(bnn)            # (None unless the activation is genReLU)
        self.ws_src, self.ws_copies, self.plan = cache[0], cache[1], cache[2]
        if len(self.ws_src) != len(mcmc._update_ws) or not all(map(_is, self.ws_src, mcmc._update_ws)):
            return None
        self.key_tail = mcmc._draw_key(bnn, 0, k)[2:]
        self.regression = regression
        self.fixed_sigma = fixed_sigma
        self.n_out = bnn._size_output
        self.adapt_possible = not (mcmc._adapt_f <= 0 and mcmc._adapt_fM >= 1)
        self.empty = getattr(be, "host_empty", None)
        self.empty_group = getattr(be, "host_empty_group", None)
        be._configure(bnn._w_layers)
        self.batch = FastBatch(ctx, k, self.plan.M, bnn._mask, mcmc._device_chain_cfg(bnn))
        self.layers = bnn._w_layers
        return self

    def matches(self, mcmc, bnn, k):
        self.missed_on_boundary = False
        if k != self.k or bnn._w_layers is not self.layers:
            return False
        for now, then in zip(self._objects(mcmc, bnn), self.objects):
            if now is not then:
                return False
        if self._scalars(mcmc, bnn, k) != self.scalars:
            return False
        # arrays that may be edited in place
        if (mcmc._update_n.tobytes() != self.n_bytes or mcmc._freq_layer_update.tobytes() != self.freq_bytes
                or bnn._prior_scale.tobytes() != self.scale_bytes):
            return False
        if self.fixed_slopes is not None and not np.array_equal(mcmc._accepted_slopes(bnn), self.fixed_slopes):
            return False
        live = mcmc._update_ws
        for src, copy, now in zip(self.ws_src, self.ws_copies, live):
            if now is not src or not _same_ends(copy, now):
                return False
        if self.fixed_sigma and mcmc._current_iteration + k - 1 > mcmc._estimate_error:
            return False
        if self.adapt_possible:
            # (what MCMC._adapt itself tests: nothing adapts from adapt_stop on, whatever the iteration count divides by)
            boundary = mcmc._next_adapt_boundary()
            at_one = mcmc._current_iteration % mcmc._adapt_freq == 0 and mcmc._current_iteration < mcmc._adapt_stop
            if at_one or (boundary is not None and boundary < mcmc._current_iteration + k):
                self.missed_on_boundary = True      # this dispatch goes the general way; the kept one stays for the next
                return False
        return True

    def _draw_ahead(self, mcmc, first_it):
        """Start the pre-draw of iterations first_it .. first_it + k - 1 (what MCMC._submit_draw does, from the kept plan)."""
        rs, randomize, mcmc_id, k, plan = mcmc._gen, mcmc._randomize_seed, mcmc._mcmc_id, self.k, self.plan
        empty, empty_group = self.empty, self.empty_group
        keep_state = not randomize

        def draw():
            if keep_state:
                job.saved = rs.bit_generator.state
            out = plan.run(rs, randomize, first_it, mcmc_id, k, empty=empty, empty_group=empty_group, state=job.saved)
            np.log(out[3], out=out[3])
            return out[0], out[1], out[2], out[3], None, None

        job = _DrawJob(draw)
        return (first_it, k) + self.key_tail, job, keep_state, self.ws_src, self

    def run(self, mcmc, bnn):
        """One dispatch.  False: nothing was done (the library returned an error code with the chain untouched) - the caller goes
        the general way."""
        k = self.k
        it = mcmc._current_iteration
        spec = mcmc._speculation
        if spec is not None and len(spec) == 5 and spec[4] is self and spec[0][0] == it:
            mcmc._speculation = None
            job = spec[1]
        else:
            job = mcmc._claim_draw(bnn, it, k)       # (takes back a second batch drawn ahead, if there is one)
        idx, delta, cnt, log_u = job.result()[:4]
        w = pack_weights(bnn._w_layers)
        batch = self.batch
        cur_sigma = (np.ones(self.n_out) * bnn._error_prm) if self.regression else None
        # TWO batches are kept drawn ahead: the draws of a batch (0.6 ms of numpy's routines for 100 iterations of config 2) take
        # several times as long once in a few hundred calls, and one batch ahead of a 1-ms dispatch then makes the dispatch wait.
        # The batch after next moves up; what is missing is handed to the helper thread at the very last moment before the C call
        # (FastBatch.run), in stream order.
        second = mcmc._speculation2
        mcmc._speculation2 = None
        fresh = []
        if second is not None and second[4] is self and second[0][0] == it + k:
            ahead = second
        else:
            if second is not None:
                mcmc._speculation2 = second
                mcmc._cancel_second()
            ahead = self._draw_ahead(mcmc, it + k)
            fresh.append(ahead[1])
        further = self._draw_ahead(mcmc, it + 2 * k)
        fresh.append(further[1])
        try:
            rc = batch.run(w, idx, delta, cnt, log_u, mcmc._logLik, mcmc._logPrior, mcmc._temperature, cur_sigma, functools.partial(_enqueue_draws, fresh))
        finally:
            if ahead[1].queued:
                mcmc._speculation = ahead
                if further[1].queued:
                    mcmc._speculation2 = further
        if rc != 0:
            # (the draws of this batch are still good: hand them to the general path as the draws "made ahead")
            mcmc._cancel_speculation()
            done = _DrawJob(None)
            done.value = (idx, delta, cnt, log_u, None, None)
            done.done.release()
            mcmc._speculation = ((it, k) + self.key_tail, done, False, self.ws_src)
            return False
        res = batch.res
        sigma = np.array(res.sigma[:self.n_out]) if self.regression else None
        mcmc._absorb(bnn, k, w, batch.acc, int(res.n_accepted), res.n_passes, res.n_void_passes, res.schedule, res.loglik, res.logprior, sigma)
        self.layers = bnn._w_layers
        return True


class _Candidate:
    """The state one iteration proposes, filled in by the draw steps of :meth:`MCMC.mh_step` and scored afterwards."""
    __slots__ = ("layers", "weight_switches", "feature_switches", "override", "sigma", "slopes_moved", "log_hastings",
                 "log_prior_extra", "log_prior", "log_lik", "log_post")

    def __init__(self, bnn_obj, log_prior_extra):
        self.layers = []
        self.weight_switches = bnn_obj._indicators + 0
        self.feature_switches = None
        self.override = None                      # column override of the forward pass (feature switches)
        self.sigma = bnn_obj._error_prm
        self.slopes_moved = False
        self.log_hastings = 0
        self.log_prior_extra = log_prior_extra    # the caller's additional_prob plus the priors of slopes and sigma
        self.log_prior = self.log_lik = self.log_post = None


class MCMC():
    def __init__(self,
                 bnn_obj: npBNN,
                 update_f=None, update_ws=None,
                 temperature=1, n_iteration=100000, sampling_f=100, print_f=1000, n_post_samples=1000,
                 update_function=UpdateNormal, sample_from_prior=0, run_ID="", init_additional_prob=0,
                 likelihood_tempering=1, mcmc_id=0, randomize_seed=False, adapt_f=0, estimate_error=True,
                 adapt_fM=1, adapt_freq=1000, adapt_stop=None, likelihood_f=None, adapt_verbose=False,
                 accuracy_f=None, accuracy_lab_f=None, row_comm=None):
        # row_comm (no reference counterpart): a communicator over whose ranks the ROWS of the data are split - bnn_obj holds this
        # rank's share (npbnn_amd.rowshard.shard_rows), every rank runs this same chain, the likelihood sums are added across ranks
        n_layers = bnn_obj._n_layers
        if update_ws is None:
            update_ws = [0.075] * n_layers
        if update_f is None:
            update_f = [0.05] * n_layers
        self._runID = bnn_obj._seed if run_ID == "" else run_ID
        self._update_f = update_f[0:n_layers]
        self._update_ws = [np.ones(bnn_obj._w_layers[i].shape) * update_ws[i] for i in range(n_layers)]
        self._update_n = np.array([np.max([1, np.round(bnn_obj._w_layers[i].size * update_f[i]).astype(int)])
                                   for i in range(n_layers)])
        self._temperature = temperature
        self._n_iterations = n_iteration
        self._sampling_f = sampling_f
        self._print_f = print_f
        self._current_iteration = 0

        mode = bnn_obj._estimation_mode
        if likelihood_f is not None:
            self._likelihood_f = likelihood_f
        elif mode == "classification":
            self._likelihood_f = calc_likelihood
        elif mode == "regression":
            self._likelihood_f = calc_likelihood_regression
        elif mode == "regression-error":
            self._likelihood_f = calc_likelihood_regression_error
        regression_like = mode in ("regression", "regression-error")
        if accuracy_f is None:
            accuracy_f = CalcAccuracy if mode == "classification" else (
                CalcAccuracyRegression if regression_like else SkipAccuracy)
        if accuracy_lab_f is None:
            accuracy_lab_f = CalcLabelAccuracy if mode == "classification" else (
                CalcLabelAccuracyRegression if regression_like else SkipAccuracyVec)
        self._accuracy_f = accuracy_f
        self._accuracy_lab_f = accuracy_lab_f

        self._bnn = bnn_obj
        self._backend = get_backend(bnn_obj, self._likelihood_f)
        if row_comm is not None:
            self.shard_over(row_comm)
        self._lazy = {}
        self._accepted_override = None      # column override of the last accepted state (feature indicators)
        self._lik_temp = likelihood_tempering
        self._sample_from_prior = sample_from_prior
        if sample_from_prior:
            self._logLik = 0
        else:
            self._logLik, _ = self._log_likelihood(bnn_obj, bnn_obj._w_layers, bnn_obj._indicators, None,
                                                   likelihood_tempering, bnn_obj._error_prm, init=True)
        self._logPrior = bnn_obj.calc_prior() + init_additional_prob
        self._logPost = self._logLik + self._logPrior
        self.update_function = update_function
        self._last_accepted = 1
        self._last_accepted_mem = [self._last_accepted]
        self._acceptance_rate = 0.
        self._mcmc_id = mcmc_id
        self._randomize_seed = randomize_seed
        self._rs = np.random.default_rng(1234)
        self._counter = 0
        self._n_post_samples = n_post_samples
        self._freq_layer_update = np.ones(n_layers)
        self._adapt_f = adapt_f
        self._adapt_fM = adapt_fM
        self._adapt_verbose = adapt_verbose
        self._adapt_stop = int(self._n_iterations * 0.05) if adapt_stop is None else adapt_stop
        self._adapt_freq = adapt_freq
        self._max_n = np.array([bnn_obj._w_layers[i].size for i in range(n_layers)]).astype(int)
        # with estimate_error the regression sigma stays at 1 for the first iterations (BNN_env.py:375-379)
        self._estimate_error = np.min([20000, 0.1 * self._n_iterations]) if estimate_error else self._n_iterations

    def shard_over(self, comm):
        """The model's data are this rank's share of the rows; sum the likelihood over the ranks of ``comm`` (npbnn_amd/rowshard.py).
        Called by the constructor for ``row_comm=``; call it again on a sampler that came out of a pickle."""
        from .rowshard import RowShardedBackend
        be = self._backend if self._backend is not None else get_backend(self._bnn, self._likelihood_f)
        if isinstance(be, RowShardedBackend):
            be = be._inner
        self._backend = RowShardedBackend(be, comm, self._bnn)
        self._invalidate()

    # ------------------------------------------------------------------------------------------
    # device evaluation
    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _forward_weights(weights, indicators):
        return [weights[0] * indicators] + list(weights[1:])

    def _slopes(self, bnn_obj):
        return bnn_obj._act_fun.device_slopes(bnn_obj._n_layers - 1)

    def _accepted_slopes(self, bnn_obj):
        """Slopes of the accepted state: what its statistics and prediction matrices are computed with."""
        return bnn_obj._act_fun.device_slopes(bnn_obj._n_layers - 1, accepted=True)

    def _host_predictions(self, bnn_obj, fw, override, which=capi.TRAIN, accepted=False):
        slopes = self._accepted_slopes(bnn_obj) if accepted else self._slopes(bnn_obj)
        y = self._backend.predict(fw, slopes=slopes, col_override=override, which=which)
        if getattr(self._backend, "out_kind", 0) is None:
            y = bnn_obj._output_act_fun(y)      # user output function on the host
        return y

    def _log_likelihood(self, bnn_obj, weights, indicators, override, lik_temp, sigma, init=False):
        """logLik of a weight set and the sigma it used.  Fused device path for the built-in
        likelihoods; any other callable gets the prediction matrix (slow path)."""
        fw = self._forward_weights(weights, indicators)
        kind = likelihood_kind(self._likelihood_f)
        if getattr(self._backend, "fused_likelihood", False) and kind is not None and kind != capi.LIK_NONE:
            sig = None
            if kind == capi.LIK_GAUSS:
                empirical = bnn_obj._empirical_error and not init
                if not empirical:
                    k = bnn_obj._labels.shape[1]
                    sig = np.ones(k) * sigma if np.ndim(sigma) == 0 else np.asarray(sigma, dtype=float)
            r = self._backend.evaluate(fw, slopes=self._slopes(bnn_obj), col_override=override, lik_temp=lik_temp,
                                       sigma=sig)
            return r["loglik"], (r["sigma"] if kind == capi.LIK_GAUSS else sigma)
        y = self._host_predictions(bnn_obj, fw, override)
        if bnn_obj._estimation_mode == "regression" and bnn_obj._empirical_error and not init:
            sigma = np.std(y - bnn_obj._labels, axis=0)
        ll = self._likelihood_f(y, bnn_obj._labels, bnn_obj._sample_id, class_weight=bnn_obj._class_w,
                                instance_weight=bnn_obj._instance_weights, lik_temp=lik_temp, sig2=sigma)
        return ll, sigma

    # ------------------------------------------------------------------------------------------
    # statistics of the accepted state, produced on demand
    # ------------------------------------------------------------------------------------------
    def _invalidate(self):
        self._lazy = {}

    def _train_stats(self):
        bnn_obj = self._bnn
        fw = self._forward_weights(bnn_obj._w_layers, bnn_obj._indicators)
        ka, kl = stat_kind(self._accuracy_f), stat_kind(self._accuracy_lab_f)
        fused = getattr(self._backend, "fused_likelihood", False)
        if fused and bnn_obj._estimation_mode == "classification" and ka == "acc" and kl == "label_acc":
            r = self._backend.evaluate(fw, slopes=self._accepted_slopes(bnn_obj), col_override=self._accepted_override,
                                       want_confusion=True)
            acc, lab_acc, freq = stats_from_confusion(r["confusion"])
            self._lazy.update(_accuracy=acc, _label_acc=lab_acc, _label_freq=freq)
            return
        if fused and bnn_obj._estimation_mode == "regression" and ka == "mse" and kl == "label_mse":
            r = self._backend.evaluate(fw, slopes=self._accepted_slopes(bnn_obj), col_override=self._accepted_override,
                                       sigma=np.ones(bnn_obj._labels.shape[1]))
            mse_col = r["sum_r2"] / r["n_rows"]
            self._lazy.update(_accuracy=float(np.mean(mse_col)), _label_acc=mse_col)
            return
        y = self._y
        self._lazy.update(_accuracy=self._accuracy_f(y, bnn_obj._labels),
                          _label_acc=self._accuracy_lab_f(y, bnn_obj._labels))

    def _compute_lazy(self, name):
        bnn_obj = self._bnn
        if self._backend is None:               # (a sampler that came out of a pickle)
            self._backend = get_backend(bnn_obj, self._likelihood_f)
        fw = self._forward_weights(bnn_obj._w_layers, bnn_obj._indicators)
        if name == "_y":
            return self._host_predictions(bnn_obj, fw, self._accepted_override, accepted=True)
        if name == "_y_test":
            if len(bnn_obj._test_data) > 0:
                return self._host_predictions(bnn_obj, fw, self._accepted_override, which=capi.TEST, accepted=True)
            return []
        if name == "_test_accuracy":
            if len(bnn_obj._test_data) == 0:
                return 0
            fused = getattr(self._backend, "fused_likelihood", False)
            if fused and bnn_obj._estimation_mode == "classification" and stat_kind(self._accuracy_f) == "acc":
                r = self._backend.evaluate(fw, slopes=self._accepted_slopes(bnn_obj), col_override=self._accepted_override,
                                           which=capi.TEST, want_confusion=True)
                return stats_from_confusion(r["confusion"])[0]
            if fused and bnn_obj._estimation_mode == "regression" and stat_kind(self._accuracy_f) == "mse":
                r = self._backend.evaluate(fw, slopes=self._accepted_slopes(bnn_obj), col_override=self._accepted_override,
                                           which=capi.TEST, sigma=np.ones(bnn_obj._labels.shape[1]))
                return float(np.mean(r["sum_r2"] / r["n_rows"]))
            return self._accuracy_f(self._y_test, bnn_obj._test_labels)
        if name == "_label_freq":
            if "_label_freq" not in self._lazy:
                self._train_stats()
            if "_label_freq" not in self._lazy:      # not classification: argmax over the output columns
                y = self._y
                pred = np.argmax(y, axis=1)
                f = np.zeros(y.shape[1])
                idx, cnt = np.unique(pred, return_counts=True)
                f[idx] = cnt
                self._lazy["_label_freq"] = f / len(pred)
            return self._lazy["_label_freq"]
        self._train_stats()
        return self._lazy[name]

    def _materialize(self):
        for name in _LAZY:
            getattr(self, name)

    def __getstate__(self):
        self._cancel_speculation()
        light = self.__dict__.get("_light_pickle", False)
        if not light:
            self._materialize()
        state = dict(self.__dict__)
        state.pop("_backend", None)
        state.pop("_light_pickle", None)
        state["_speculation"] = None
        state["_speculation2"] = None
        state["_ws_copies"] = None       # (holds the pre-draw plan: pointers into this process)
        state.pop("_fast", None)         # (ctypes structs and addresses of this process)
        if light:       # a checkpoint view (postLogger): no prediction matrices, no model reference beside the pickled one
            state["_lazy"] = {k: v for k, v in self._lazy.items() if k not in ("_y", "_y_test")}
        return state

    def _light_view(self, bnn_view):
        """This sampler as a light checkpoint stores it (see postLogger): same state, bound to ``bnn_view``, pickled without the
        prediction matrices - after loading they are computed on demand from the weights, like every other statistic."""
        # draws made ahead of the chain's position go back first (the view shares the generator: a checkpoint taken with them in flight
        # would store a generator one or two batches past the iteration it claims to be at, and a run resumed from it would skip those draws)
        self._cancel_speculation()
        view = self.__class__.__new__(self.__class__)
        view.__dict__.update(self.__dict__)
        view._speculation = view._speculation2 = None
        view._fast = None
        view._bnn = bnn_view
        view._light_pickle = True
        return view

    def __setstate__(self, state):
        state = dict(state)
        if "_rs" in state:
            state["_gen"] = state.pop("_rs")
        # statistics stored as plain attributes (an upstream-format checkpoint, npbnn_amd/export.py) are the cached values here
        cached = dict(state.pop("_lazy", None) or {})
        for name in _LAZY:
            if name in state:
                cached[name] = state.pop(name)
        self.__dict__.update(state)
        self._lazy = cached
        for name in ("_backend", "_bnn", "_accepted_override"):
            self.__dict__.setdefault(name, None)

    def __deepcopy__(self, memo):
        import copy
        self._cancel_speculation()
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_ws_copies", "_fast"):    # (the pre-draw plan holds pointers into its own step-size copies: the copy builds its own)
                new.__dict__[k] = None
            else:
                new.__dict__[k] = v if k == "_backend" else copy.deepcopy(v, memo)
        return new

    # ------------------------------------------------------------------------------------------
    # adaptation of the proposal sizes (reference: BNN_env.py:392-413)
    # ------------------------------------------------------------------------------------------
    def _adapt(self, bnn_obj):
        if not (self._current_iteration % self._adapt_freq == 0 and self._current_iteration < self._adapt_stop):
            return
        for shrink in (True, False):
            if shrink:
                if not self._acceptance_rate < self._adapt_f:
                    continue
                self._freq_layer_update = self._freq_layer_update * 0.8
                self.reset_update_f(np.array(self._update_f) * .85)
                step_scale = 0.9
            else:
                if not (self._acceptance_rate > self._adapt_fM and np.sum(self._update_n) < bnn_obj._n_params):
                    continue
                self.reset_update_f(np.exp(np.log(np.array(self._update_f)) * .85))
                step_scale = 1.2
            n = (self._max_n * (self._update_f)).astype(int)
            n[n < 1] = 1
            self.reset_update_n(n)
            self.reset_update_ws([i * step_scale for i in self._update_ws])
            if self._adapt_verbose:
                print(self._acceptance_rate, self._update_n, self._update_ws[0][0][0], self._freq_layer_update,
                      self._update_f)

    # ------------------------------------------------------------------------------------------
    # one Metropolis-Hastings iteration (behaviour: np_bnn/BNN_env.py:381-532)
    # ------------------------------------------------------------------------------------------
    SLOPE_STEP, SLOPE_PRIOR_RATE = 0.05, 10          # random-walk width and Exp(rate) prior of trainable slopes (:417-419)
    SIGMA_WINDOW, SIGMA_SHARE, SIGMA_PRIOR_RATE = 1.1, 0.5, 1     # multiplier proposal and Exp(rate) prior of sigma (:439-442)
    FEATURE_FLIP_CHANCE, FEATURE_FLIP_SHARE = 0.2, 0.5            # feature switches (:425-426)

    def mh_step(self, bnn_obj, additional_prob=0, return_bnn=False):
        """One iteration: build a candidate state, score it, accept or reject it, book the outcome.

        What is contractual is the ORDER in which random numbers leave the two streams (``G`` = the chain's Generator,
        ``np.random`` = numpy's global stream), because a chain must make the reference's draws for the reference's seed:

          ====  =====================================================  ======================================  ==============
          step  when                                                   draws                                   reference
          ====  =====================================================  ======================================  ==============
          0     ``randomize_seed``                                     G = default_rng(iteration + mcmc_id)     BNN_env :383
          1     trainable activation slopes                            G.integers, G.normal                     :416-421
          2     feature switches and iteration > adapt_stop            G.random; if < 0.2: np.random.random,    :424-433
                                                                       np.random.binomial
          3     regression, sigma estimated, iteration > its start     G.binomial(k), G.random(k)               :435-444
          4     always                                                 G.random(n_layers)                       :446-447
          5     per layer, in order: the layer's proposal function     G.integers, G.integers, G.normal         :449-459
                (or, layer 0 in an indicator turn)                     np.random.random, np.random.binomial     :457-460
          6     always                                                 G.random  (the accept test)              :493
          ====  =====================================================  ======================================  ==============
        """
        self._enter_step(bnn_obj)
        cand = _Candidate(bnn_obj, additional_prob)
        self._draw_slopes(bnn_obj, cand)
        self._draw_feature_switches(bnn_obj, cand)
        self._draw_sigma(bnn_obj, cand)
        self._draw_layers(bnn_obj, cand)
        self._score(bnn_obj, cand)
        took = (cand.log_post - self._logPost) * self._temperature + cand.log_hastings >= np.log(self._gen.random())
        if took:
            self._install(bnn_obj, cand)
        self._book(took)
        if return_bnn:
            return bnn_obj, self

    def _enter_step(self, bnn_obj):
        self._cancel_speculation()
        self._bnn = bnn_obj
        if self._backend is None:
            self._backend = get_backend(bnn_obj, self._likelihood_f)
        if self._randomize_seed:
            self._gen = np.random.default_rng(self._current_iteration + self._mcmc_id)
        self._adapt(bnn_obj)

    def _draw_slopes(self, bnn_obj, cand):
        """Step 1.  The proposed slopes go into the activation object at once - accepted or not - and the forward pass of this
        iteration reads them from there."""
        act = bnn_obj._act_fun
        if not act._trainable:
            return
        moved, _, log_h = UpdateNormal1D(act._acc_prm, d=self.SLOPE_STEP, n=1, Mb=1, mb=0, rs=self._gen)
        rate = self.SLOPE_PRIOR_RATE
        cand.log_prior_extra += np.log(rate) * -np.sum(moved) * rate
        cand.log_hastings += log_h
        cand.slopes_moved = True
        act.reset_prm(moved)

    def _draw_feature_switches(self, bnn_obj, cand):
        """Step 2.  Once the switches are live every candidate is evaluated through the column override, flipped or not."""
        current = bnn_obj._feature_indicators
        cand.feature_switches = current
        if current is None or not self._current_iteration > self._adapt_stop:
            return
        cand.feature_switches = current + 0
        if self._gen.random() < self.FEATURE_FLIP_CHANCE:
            cand.feature_switches = UpdateBinomial(cand.feature_switches, self.FEATURE_FLIP_SHARE, current.shape)
        cand.override = data_transform_obj(cand.feature_switches, bnn_obj._feature_means).column_override()

    def _draw_sigma(self, bnn_obj, cand):
        """Step 3.  Outside regression, and in regression while sigma is still fixed, the candidate's sigma is the scalar 1;
        an empirical sigma comes out of the evaluation instead."""
        if not (bnn_obj._estimation_mode == "regression" and self._current_iteration > self._estimate_error):
            cand.sigma = 1
            return
        if bnn_obj._empirical_error:
            return
        cand.sigma, _, log_h = multiplier_proposal_vector(bnn_obj._error_prm, d=self.SIGMA_WINDOW, f=self.SIGMA_SHARE, rs=self._gen)
        rate = self.SIGMA_PRIOR_RATE
        cand.log_hastings += log_h
        cand.log_prior_extra += np.log(rate) * -np.sum(cand.sigma) * rate

    def _draw_layers(self, bnn_obj, cand):
        """Steps 4 and 5.  A layer is proposed when its uniform number falls below the layer's update frequency; the layer with
        the smallest number always is.  With weight indicators, layer 0 spends the turns whose number falls below
        ``freq_indicator`` on flipping indicators instead of moving weights."""
        turn = self._gen.random(bnn_obj._n_layers)
        turn[np.argmin(turn)] = 0
        mask = bnn_obj._mask
        for li, current in enumerate(bnn_obj._w_layers):
            indicator_turn = li == 0 and turn[li] < bnn_obj._freq_indicator
            if not indicator_turn and turn[li] < self._freq_layer_update[li]:
                layer, _, log_h = self.update_function(current, d=self._update_ws[li], n=self._update_n[li],
                                                       Mb=bnn_obj._w_bound, mb=-bnn_obj._w_bound, rs=self._gen)
                cand.log_hastings += log_h
            else:
                layer = current + 0
                if indicator_turn:
                    # (the flip share is read from the FOURTH update frequency, as upstream does, BNN_env.py:460: networks of
                    # fewer than four weight matrices fail here with upstream's IndexError)
                    cand.weight_switches = UpdateBinomial(bnn_obj._indicators, self._update_f[3], bnn_obj._indicators.shape)
            if mask is not None:
                layer *= mask[li]
            cand.layers.append(layer)

    def _score(self, bnn_obj, cand):
        cand.log_prior = bnn_obj.calc_prior(w=cand.layers, ind=cand.weight_switches) + cand.log_prior_extra
        if self._sample_from_prior:
            cand.log_lik = 0
        else:
            cand.log_lik, cand.sigma = self._log_likelihood(bnn_obj, cand.layers, cand.weight_switches, cand.override,
                                                            self._lik_temp, cand.sigma)
        cand.log_post = cand.log_lik + cand.log_prior

    def _install(self, bnn_obj, cand):
        """The candidate becomes the chain's state.  Statistics of the state (accuracies, predictions) are not computed here:
        they are functions of what is stored now and are produced when somebody reads them."""
        bnn_obj.reset_weights(cand.layers)
        bnn_obj.reset_indicators(cand.weight_switches)
        if bnn_obj._feature_indicators is not None:
            bnn_obj._feature_indicators = cand.feature_switches + 0
        if bnn_obj._estimation_mode == "regression":
            # (upstream stores the bare scalar 1 while sigma is still fixed and later trips over it in its own multiplier
            # proposal, BNN_env.py:444,501 -> BNN_mcmc.py:105; a vector of ones means the same and keeps working)
            sigma = cand.sigma
            bnn_obj.reset_error_prm(np.ones(bnn_obj._size_output) * sigma if np.ndim(sigma) == 0 else sigma)
        if cand.slopes_moved:
            bnn_obj._act_fun.reset_accepted_prm()
            self._slope_term_in_prior = True        # (MCMC.__init__'s prior is without the term: BNN_env.py:320 against :419)
        self._logLik, self._logPrior, self._logPost = cand.log_lik, cand.log_prior, cand.log_post
        self._accepted_override = cand.override
        self._invalidate()

    def _book(self, took):
        """Acceptance book-keeping (BNN_env.py:523-530): the rate is the mean over a window that holds up to 101 outcomes at
        the moment it is taken and is cut back to 100 afterwards."""
        self._last_accepted = 1 if took else 0
        window = self._last_accepted_mem
        window.append(self._last_accepted)
        self._acceptance_rate = np.mean(window)
        if len(window) > 100:
            self._last_accepted_mem = window[-100:]
        self._current_iteration += 1

    # ------------------------------------------------------------------------------------------
    # K iterations with the chain resident on the device
    # ------------------------------------------------------------------------------------------
    def _device_loop_ok(self, bnn_obj, k):
        """True when the next k iterations can run on the patch-list device chain (npbnn_chain_run): the default proposal and no
        feature of mh_step that changes more than a list of weights."""
        return self._device_mode(bnn_obj, k) == "patch"

    def _device_mode(self, bnn_obj, k):
        """How the next k iterations run: ``"patch"`` - the device chain whose proposals are lists of perturbed weights
        (npbnn_chain_run: speculative passes, every schedule); ``"general"`` - the general device chain (npbnn_chain_run_general:
        the full candidate built on the device every iteration - the fixed-normal and normalising proposals, weight indicators,
        feature indicators); ``None`` - one :meth:`mh_step` after the other (custom proposal / likelihood callables, sampling from
        the prior, and the settings upstream itself cannot run: UpdateUniform takes no ``rs``, BNN_mcmc.py:86, and weight indicators
        read ``_update_f[3]``, BNN_env.py:460, which needs four weight matrices)."""
        be = self._backend
        if be is None or not hasattr(be, "run_chain") or not getattr(be, "fused_likelihood", False):
            return None
        if self._sample_from_prior:
            return None
        if likelihood_kind(self._likelihood_f) in (None, capi.LIK_NONE):
            return None
        # (prior scales - one per layer, per input node or per weight, hyper_p 1-3 - change in gibbs_step only, between calls: the
        # device takes them as constants of the batch)
        if bnn_obj._estimation_mode == "regression" and not bnn_obj._empirical_error:
            # sigma fixed at 1 up to iteration _estimate_error, multiplier proposals after it (BNN_env.py:435-444): a batch
            # must lie on one side (run_steps cuts there); the proposals need the Gaussian likelihood on the device
            first, last = self._current_iteration, self._current_iteration + k - 1
            if first <= self._estimate_error < last:
                return None
            if first > self._estimate_error and likelihood_kind(self._likelihood_f) != capi.LIK_GAUSS:
                return None
        if np.isfinite(bnn_obj._w_bound) and any(np.max(np.abs(w)) > bnn_obj._w_bound for w in bnn_obj._w_layers if w.size):
            # a weight OUTSIDE the reflecting walls (initial weights under a narrow uniform prior): upstream's proposal folds every
            # entry of a layer it touches back inside, moved or not (BNN_mcmc.py:66-67); the device chains fold the moved ones.
            # mh_step does it upstream's way, and after an accepted proposal per layer nothing is outside any more
            return None
        whole_candidate = bnn_obj._feature_indicators is not None or bool(bnn_obj._freq_indicator)
        if self.update_function is UpdateNormal and not whole_candidate:
            if bnn_obj._act_fun._trainable and (bnn_obj._act_fun._function != "genReLU" or not hasattr(be, "ctx")
                                                or len(bnn_obj._act_fun._acc_prm) != bnn_obj._n_layers - 1):
                return None     # (slopes that are proposed but never used by the forward pass, or not one per hidden layer: mh_step)
            return "patch"
        if getattr(be, "row_sharded", False):
            return None         # (rows split over ranks: plain batches on the device, everything else through the sharded mh_step)
        if (self.update_function in _GENERAL_PROPOSALS and hasattr(be, "run_chain_general") and not bnn_obj._act_fun._trainable):
            if bnn_obj._freq_indicator and bnn_obj._n_layers < 4:
                return None
            if bnn_obj._feature_indicators is not None:
                boundary = self._adapt_stop          # the override switches on after it: a batch lies on one side
                first, last = self._current_iteration, self._current_iteration + k - 1
                if first <= boundary < last:
                    return None
            return "general"
        return None

    def _plain_device_batches(self, bnn_obj):
        """May the chain take part in an exchange run or a group pass?  Those entry points run plain batches only: chain state that
        rides along with extra per-iteration draws (trainable activation slopes) stays on :meth:`run_steps`."""
        return not bnn_obj._act_fun._trainable and not getattr(self._backend, "row_sharded", False)

    def _sigma_proposal_columns(self, bnn_obj, first_it):
        """Columns of the error parameter that every iteration from ``first_it`` on proposes to change (0: none)."""
        if (bnn_obj._estimation_mode == "regression" and not bnn_obj._empirical_error and first_it > self._estimate_error):
            return int(np.size(bnn_obj._error_prm))
        return 0

    def _next_adapt_boundary(self):
        if self._adapt_f <= 0 and self._adapt_fM >= 1:
            return None             # _adapt can never fire: the acceptance rate lies in [0, 1]
        it = self._current_iteration
        nxt = (it // self._adapt_freq + 1) * self._adapt_freq
        return nxt if nxt < self._adapt_stop else None

    def run_steps(self, bnn_obj, n_steps):
        """Advance the chain by ``n_steps`` iterations: exactly ``n_steps`` calls of :meth:`mh_step` (same random
        stream, same adaptation points, same bookkeeping), executed where possible as a device-resident chain
        with the random numbers pre-drawn on the host.  The draws of the next sub-batch are produced by a helper
        thread while the GPU runs the current one - also across calls: the last sub-batch of a call leaves the
        draws of the probable next call in flight (:meth:`_cancel_speculation` rewinds the generator when something
        else wants the stream first)."""
        self._bnn = bnn_obj
        if self._backend is None:
            self._backend = get_backend(bnn_obj, self._likelihood_f)
        remaining = int(n_steps)
        fast = self._fast
        if fast is not None:
            if fast.matches(self, bnn_obj, remaining) and fast.run(self, bnn_obj):
                return
            if not getattr(fast, "missed_on_boundary", False):
                self._fast = None
                self._fast_hold = 16        # (what made it fail may last: a few dispatches the general way before another is built)
        while remaining > 0:
            seg = remaining                       # iterations over which the proposal settings stay constant
            boundary = self._next_adapt_boundary()
            if boundary is not None:
                seg = min(seg, boundary - self._current_iteration)
            if bnn_obj._estimation_mode == "regression" and self._current_iteration <= self._estimate_error:
                seg = min(seg, int(np.floor(self._estimate_error)) + 1 - self._current_iteration)    # sigma proposals start after it
            if bnn_obj._feature_indicators is not None and self._current_iteration <= self._adapt_stop:
                seg = min(seg, int(self._adapt_stop) + 1 - self._current_iteration)          # the column override starts after it
            mode = self._device_mode(bnn_obj, seg)
            if mode is None:
                self.mh_step(bnn_obj)
                remaining -= 1
                continue
            if mode == "general":
                self._run_general(bnn_obj, seg)
                remaining -= seg
                continue
            self._adapt(bnn_obj)
            sizes = self._sub_batches(seg)
            it = self._current_iteration
            pending = self._claim_draw(bnn_obj, it, sizes[0])
            for n, k in enumerate(sizes):
                drawn = pending.result()
                idx, delta, cnt, log_u, smult, hast = drawn[:6]
                slope_draws = drawn[6:8] if len(drawn) > 6 else None
                it += k
                if n + 1 < len(sizes):
                    pending = self._submit_draw(bnn_obj, it, sizes[n + 1])[1]
                elif remaining == seg:            # last sub-batch of this call: draw ahead for the next call
                    self._speculation = self._submit_draw(bnn_obj, it, min(self.SUB_BATCH, int(n_steps)), rewindable=True)
                self._run_device_batch(bnn_obj, idx, delta, cnt, log_u, smult, hast, slope_draws)
            remaining -= seg
        # the same call twice in a row on the patch-list chain: the next one of this kind takes the short way (_FastDispatch)
        k = int(n_steps)
        if self._fast_hold > 0:
            self._fast_hold -= 1
        elif k == self._last_call and self._fast is None and not os.environ.get("NPBNN_NO_FAST_DISPATCH"):
            self._fast = _FastDispatch.build(self, bnn_obj, k)
            if self._fast is None:
                self._fast_hold = 64
        self._last_call = k

    _fast = None             # the kept form of the last dispatch (_FastDispatch), when calls repeat
    _fast_hold = 0
    _last_call = 0
    _speculation2 = None     # the batch after that one, same form (kept by the fast dispatch only)
    _speculation = None      # (key, job, may the draw be taken back?, step-size arrays used[, the fast dispatch that made it]) of draws made ahead
    _ws_copies = None        # (step-size arrays, private copies of them, the pre-draw plan built on the copies, the settings it holds)

    def _draw_key(self, bnn_obj, first_it, k):
        """What the draws of iterations first_it .. first_it+k-1 depend on besides the generator's state and the step sizes."""
        return (int(first_it), int(k), bool(self._randomize_seed), int(self._mcmc_id),
                np.asarray(self._update_n, dtype=np.int64).tobytes(), np.asarray(self._freq_layer_update, dtype=np.float64).tobytes(),
                tuple([w.shape for w in bnn_obj._w_layers]), self._sigma_proposal_columns(bnn_obj, first_it), self._n_trainable_slopes(bnn_obj))

    @staticmethod
    def _n_trainable_slopes(bnn_obj):
        act = bnn_obj._act_fun
        return len(act._acc_prm) if act._trainable else 0

    def _submit_draw(self, bnn_obj, first_it, k, rewindable=False):
        """Start the pre-draw of iterations first_it .. first_it+k-1 on the helper thread."""
        rs, randomize, mcmc_id = self._gen, self._randomize_seed, self._mcmc_id
        key = self._draw_key(bnn_obj, first_it, k)
        # private copies of the step sizes (the draw runs later) and the proposal description built from them (predraw.PredrawPlan),
        # made once per set of source arrays and proposal settings: _adapt and reset_update_ws install NEW arrays, so the same
        # objects mean the same values
        src = self._update_ws
        cache = self._ws_copies
        if cache is None or len(cache[0]) != len(src) or not all(map(_is, cache[0], src)) or cache[3] != key[4:7] \
                or not all(map(_same_ends, cache[1], src)):                       # (an in-place edit of _update_ws[i])
            from . import predraw as pd
            copies = [np.array(w, dtype=np.float64) for w in src]
            plan = pd.PredrawPlan([np.empty(w.shape) for w in bnn_obj._w_layers], [int(n) for n in self._update_n], copies,
                                  [float(f) for f in self._freq_layer_update])
            cache = self._ws_copies = (list(src), copies, plan, key[4:7])
        src, plan = cache[0], cache[2]
        be = self._backend
        empty = getattr(be, "host_empty", None)
        empty_group = getattr(be, "host_empty_group", None)
        keep_state = rewindable and not randomize
        sigma_k, n_slopes = key[7], key[8]

        def draw():
            if keep_state:        # (here, on the helper thread, rather than in the caller: reading the state builds a nested dictionary)
                job.saved = rs.bit_generator.state
            out = plan.run(rs, randomize, first_it, mcmc_id, k, empty=empty, sigma_k=sigma_k, n_slopes=n_slopes, slope_d=0.05,
                           empty_group=empty_group, state=job.saved)
            idx, delta, cnt, u = out[:4]
            np.log(u, out=u)                      # the accept test compares with log u (BNN_env.py:493)
            smult = hast = None
            if sigma_k:                           # multiplier_proposal_vector(q, d=1.1, f=0.5) per iteration (BNN_mcmc.py:101-113)
                chosen, u_sigma = out[5], out[6]
                smult = np.exp(2 * np.log(1.1) * (u_sigma - .5))
                smult[chosen == 0] = 1.
                hast = np.array([np.sum(np.log(row)) for row in smult])
            if n_slopes:                          # UpdateNormal1D(acc_prm, d=0.05, n=1): the entry and the step of every iteration
                return idx, delta, cnt, u, smult, hast, out[-2], out[-1]
            return idx, delta, cnt, u, smult, hast

        job = _DrawJob(draw)
        _draw_pool().enqueue(job)
        return key, job, keep_state, src

    def _claim_draw(self, bnn_obj, first_it, k):
        """The draws for iterations first_it .. first_it+k-1: the ones made ahead by the previous call when they
        are exactly these, else fresh ones."""
        self._cancel_second()        # (this path keeps one batch ahead: whatever it submits next must follow the first in the stream)
        spec = self._speculation
        if (spec is not None and spec[0] == self._draw_key(bnn_obj, first_it, k) and len(spec[3]) == len(self._update_ws)
                and all(map(_is, spec[3], self._update_ws))                     # (the very arrays the draw was made with ...
                and self._ws_copies is not None and all(map(_same_ends, self._ws_copies[1], self._update_ws))):   # ... unedited)
            self._speculation = None
            return spec[1]
        self._cancel_speculation()
        return self._submit_draw(bnn_obj, first_it, k)[1]

    def _cancel_second(self):
        """Drop the SECOND batch drawn ahead (the fast dispatch keeps two): the generator goes back to where the first one left it."""
        spec, self._speculation2 = self._speculation2, None
        if spec is None:
            return
        try:
            spec[1].result()
        finally:
            if spec[2] and spec[1].saved is not None:
                self._gen.bit_generator.state = spec[1].saved

    def _cancel_speculation(self):
        """Drop draws made ahead and put the generator back where the chain left it."""
        self._cancel_second()
        spec, self._speculation = self._speculation, None
        if spec is None:
            return
        try:
            spec[1].result()
        finally:
            if spec[2] and spec[1].saved is not None:
                self._gen.bit_generator.state = spec[1].saved

    @property
    def _rs(self):
        """The chain's numpy Generator (BNN_env.py:352).  Reading it first takes back any draws made ahead, so
        callers always see the stream exactly where the iterations done so far left it."""
        self._cancel_speculation()
        return self._gen

    @_rs.setter
    def _rs(self, value):
        self._cancel_speculation()
        self._gen = value

    n_candidates = 0         # proposals evaluated per pass over the data by the device chain: 0 = as many as fit (<= 3)
    device_schedule = 0      # 0 auto, 1 serial (evaluate, decide, evaluate ...), 2 overlapped (decide pass L-1 while pass L is evaluated),
    #                          3 overlapped with the launches alternating between two streams (they overlap; device flags order them),
    #                          4 overlapped as one persistent launch per batch round (its workgroups loop over the passes; same flags)
    _device_schedule_used = 0
    _device_accepted = 0
    _device_passes = 0
    _device_void_passes = 0
    _device_iterations = 0
    SUB_BATCH = 128          # first sub-batch of a segment (its pre-draw is not overlapped); later ones double
    # (512, not more: the draws of sub-batch n+1 are made while the GPU runs sub-batch n, so a sub-batch twice the size of the one
    # before it makes the GPU wait whenever drawing an iteration takes more than half of what evaluating it does - config 2: 5 of 9 us,
    # config 5: 10 of 11 us.  One call of 10 000 iterations with sub-batches growing to 2048 / 512: 105 / 118 k it/s on config 2,
    # 62 / 75 k on config 5, 39.5 / 39.8 k on config 4; tools/time_long_calls.py)
    SUB_BATCH_MAX = 512

    def _sub_batches(self, seg):
        sizes, k = [], self.SUB_BATCH
        while seg > 0:
            n = min(k, seg)
            sizes.append(n)
            seg -= n
            k = min(2 * k, self.SUB_BATCH_MAX)
        return sizes

    _slope_term_in_prior = False     # does _logPrior hold log(r) * -sum(accepted slopes) * r?  (from the first accepted proposal on)

    def _device_chain_cfg(self, bnn_obj, sigma_mult=None, hastings=None, slope_draws=None):
        """The chain's settings and current state as the keyword arguments of the device chain entry points."""
        regression = bnn_obj._estimation_mode == "regression"
        sigma = None
        cur_sigma = None
        if regression:
            cur_sigma = np.ones(bnn_obj._size_output) * bnn_obj._error_prm
            if not bnn_obj._empirical_error and sigma_mult is None:
                sigma = np.ones(bnn_obj._size_output)      # sigma stays 1 while it <= _estimate_error
        extra = {}
        if slope_draws is not None:
            extra["slopes"] = (slope_draws[0], slope_draws[1], np.array(bnn_obj._act_fun._acc_prm, dtype=float), self._slope_term_in_prior)
        elif bnn_obj._act_fun._function == "genReLU":
            extra["fixed_slopes"] = self._accepted_slopes(bnn_obj)       # ActFun("genReLU", prm=...) with trainable=False
        return dict(extra, sigma_mult=sigma_mult, hastings=hastings, prior_kind=bnn_obj._prior_kind() if bnn_obj._prior else 0, prior_scale=bnn_obj._prior_scale,
                    w_bound=bnn_obj._w_bound, temperature=self._temperature, lik_temp=self._lik_temp,
                    cur_loglik=self._logLik, cur_logprior=self._logPrior, cur_sigma=cur_sigma, sigma=sigma,
                    n_candidates=self.n_candidates, schedule=self.device_schedule)

    def _run_device_batch(self, bnn_obj, idx, delta, cnt, log_u, sigma_mult=None, hastings=None, slope_draws=None):
        w_new, acc, _, _, res = self._backend.run_chain(bnn_obj._w_layers, idx=idx, delta=delta, cnt=cnt, log_u=log_u, mask=bnn_obj._mask,
                                                        **self._device_chain_cfg(bnn_obj, sigma_mult, hastings, slope_draws))
        self._absorb_device_batch(bnn_obj, len(cnt), w_new, acc, res)
        if slope_draws is not None:
            # what k calls of mh_step leave in the activation object: the accepted slopes, and - installed whether accepted or not
            # (BNN_env.py:421) - the slopes the LAST iteration proposed: the accepted ones before it with its entry moved
            act = bnn_obj._act_fun
            if res["n_accepted"] > 0:
                act.reset_prm(np.array(res["slopes"][:len(act._acc_prm)], dtype=float))
                act.reset_accepted_prm()
                self._slope_term_in_prior = True
            last = np.array(act._acc_prm, dtype=float)
            if not acc[len(cnt) - 1]:
                k = int(slope_draws[0][len(cnt) - 1])
                v = last[k] + slope_draws[1][len(cnt) - 1]
                if v > 1:
                    v = 1 - (v - 1)
                if v < 0:
                    v = 0 + (0 - v)
                last[k] = v
            act.reset_prm(last)

    def _absorb_device_batch(self, bnn_obj, k, w_new, acc, res):
        """Book-keeping of k device-resident iterations: what k calls of mh_step would have left behind."""
        if k <= 0:
            return
        self._absorb(bnn_obj, k, w_new, acc, int(res.get("n_accepted", 0)), res.get("n_passes", k), res.get("n_void_passes", 0),
                     res.get("schedule", 0), res["loglik"], res["logprior"], res.get("sigma"))

    def _absorb(self, bnn_obj, k, w_new, acc, n_accepted, n_passes, n_void, schedule, loglik, logprior, sigma):
        acc = acc[:k]
        self._device_schedule_used = schedule
        self._device_accepted += n_accepted
        self._device_passes += n_passes
        self._device_void_passes += n_void
        self._device_iterations += k
        if n_accepted > 0:
            layers, off = [], 0
            for w in bnn_obj._w_layers:
                layers.append(w_new[off:off + w.size].reshape(w.shape))
                off += w.size
            if off == w_new.size and w_new.dtype == np.float64:
                note_packed_views(layers, w_new)
            bnn_obj.reset_weights(layers)
            self._logLik, self._logPrior = loglik, logprior
            self._logPost = self._logLik + self._logPrior
            if bnn_obj._estimation_mode == "regression":
                bnn_obj.reset_error_prm(sigma)
            self._accepted_override = None
            self._invalidate()
        history = self._last_accepted_mem + acc.tolist()
        self._last_accepted = history[-1]
        window = history[-101:] if len(history) > 100 else history       # (what np.mean saw in the last mh_step of these k)
        self._acceptance_rate = np.float64(sum(window)) / len(window)
        self._last_accepted_mem = history[-100:] if len(history) > 100 else history
        self._current_iteration += k
        if self._randomize_seed:        # (assigning _gen, not _rs: draws made ahead for the next call stay valid)
            self._gen = np.random.default_rng(self._current_iteration - 1 + self._mcmc_id)

    # ------------------------------------------------------------------------------------------
    # the general device chain: proposals that change more than a list of weights
    # ------------------------------------------------------------------------------------------
    def _draw_general(self, bnn_obj, first_it, k):
        """The random numbers of iterations first_it .. first_it+k-1 in the order mh_step consumes them - the chain's Generator
        for the weight proposals and the accept test, numpy's global stream for the indicator flips (UpdateBinomial) - turned into
        what npbnn_chain_run_general takes.  Nothing here depends on the chain's state."""
        n_layers = bnn_obj._n_layers
        shapes = [w.shape for w in bnn_obj._w_layers]
        offs = np.concatenate([[0], np.cumsum([w.size for w in bnn_obj._w_layers])]).astype(np.int64)
        kind = _GENERAL_PROPOSALS[self.update_function]
        M = int(max(1, sum(int(n) for n in self._update_n)))
        idx = np.full((k, M), -1, dtype=np.int32)
        val = np.zeros((k, M))
        cnt = np.zeros(k, dtype=np.int32)
        fixed = kind == capi.PROP_FIXED_NORMAL
        h_idx = np.full((k, M), -1, dtype=np.int32) if fixed else None
        h_val, h_fac = (np.zeros((k, M)), np.zeros((k, M))) if fixed else (None, None)
        h_cnt = np.zeros(k, dtype=np.int32) if fixed else None
        layer_mask = np.zeros(k, dtype=np.int32)
        log_u = np.empty(k)
        has_ind = bool(bnn_obj._freq_indicator)
        has_find = bnn_obj._feature_indicators is not None
        ind_ptr, ind_pos = np.zeros(k + 1, dtype=np.int32), []
        find_ptr, find_pos, find_use = np.zeros(k + 1, dtype=np.int32), [], np.zeros(k, dtype=np.int32)
        sigma_k = self._sigma_proposal_columns(bnn_obj, first_it)
        smult = np.ones((k, sigma_k)) if sigma_k else None
        hast = np.zeros(k) if sigma_k else None
        for t in range(k):
            it = first_it + t
            rs = np.random.default_rng(it + self._mcmc_id) if self._randomize_seed else self._gen
            if has_find and it > self._adapt_stop:
                find_use[t] = 1
                if rs.random() < 0.2:          # UpdateBinomial(ind, 0.5, shape): |ind - binomial(1, random() * 0.5, shape)|
                    flips = np.random.binomial(1, np.random.random() * 0.5, bnn_obj._feature_indicators.shape)
                    find_pos.extend(np.nonzero(flips)[0].tolist())
            find_ptr[t + 1] = len(find_pos)
            if sigma_k:                        # multiplier_proposal_vector(q, d=1.1, f=0.5, rs) (BNN_mcmc.py:101-113)
                chosen = rs.binomial(1, 0.5, sigma_k)
                u = rs.random(sigma_k)
                m = np.exp(2 * np.log(1.1) * (u - .5))
                m[chosen == 0] = 1.
                smult[t] = m
                hast[t] = np.sum(np.log(m))
            rr = rs.random(n_layers)
            rr[np.argmin(rr)] = 0
            used = h_used = 0
            for i in range(n_layers):
                if rr[i] >= bnn_obj._freq_indicator or i > 0:
                    if rr[i] < self._freq_layer_update[i]:
                        n = int(self._update_n[i])
                        d = self._update_ws[i]
                        rows = rs.integers(0, shapes[i][0], n)
                        cols = rs.integers(0, shapes[i][1], n)
                        z = rs.normal(0, d[rows, cols], n)
                        flat = offs[i] + rows * shapes[i][1] + cols
                        # numpy's indexed assignment keeps the LAST draw of a position
                        _, last = np.unique(flat[::-1], return_index=True)
                        keep = np.sort(n - 1 - last)
                        idx[t, used:used + len(keep)] = flat[keep]
                        val[t, used:used + len(keep)] = z[keep]
                        used += len(keep)
                        if fixed:              # the Hastings term runs over every draw: logpdf(old) - logpdf(drawn), width d
                            h_idx[t, h_used:h_used + n] = flat
                            h_val[t, h_used:h_used + n] = z
                            h_fac[t, h_used:h_used + n] = 0.5 / (d[rows, cols] ** 2)
                            h_used += n
                        layer_mask[t] |= 1 << i
                else:                          # layer 0 keeps its weights and proposes new indicators instead (BNN_env.py:457-460)
                    flips = np.random.binomial(1, np.random.random() * self._update_f[3], bnn_obj._indicators.shape)
                    ind_pos.extend(np.nonzero(flips.ravel())[0].tolist())
            cnt[t] = used
            if fixed:
                h_cnt[t] = h_used
            ind_ptr[t + 1] = len(ind_pos)
            log_u[t] = np.log(rs.random())
        draws = dict(kind=kind, idx=idx, val=val, cnt=cnt, layer_mask=layer_mask, h_idx=h_idx, h_val=h_val, h_fac=h_fac, h_cnt=h_cnt)
        if has_ind:
            draws.update(ind_ptr=ind_ptr, ind_pos=np.array(ind_pos, dtype=np.int32))
        if has_find:
            draws.update(find_ptr=find_ptr, find_pos=np.array(find_pos, dtype=np.int32), find_use=find_use)
        return draws, log_u, smult, hast

    def _run_general(self, bnn_obj, seg):
        """``seg`` iterations on the general device chain (proposal settings constant over them): sub-batches, the draws of the
        next one made by the helper thread while the GPU runs the current one - within this call only, because the indicator
        flips come from numpy's GLOBAL stream, which other code may use between calls."""
        self._cancel_speculation()
        self._adapt(bnn_obj)
        sizes = self._sub_batches(seg)
        it = self._current_iteration
        pending = _draw_pool().submit(self._draw_general, bnn_obj, it, sizes[0])
        for n, k in enumerate(sizes):
            draws, log_u, smult, hast = pending.result()
            it += k
            if n + 1 < len(sizes):
                pending = _draw_pool().submit(self._draw_general, bnn_obj, it, sizes[n + 1])
            has_ind = bool(bnn_obj._freq_indicator)
            has_find = bnn_obj._feature_indicators is not None
            w_new, ind, find, acc, _, _, res = self._backend.run_chain_general(
                bnn_obj._w_layers, draws=draws, log_u=log_u, mask=bnn_obj._mask,
                indicators=bnn_obj._indicators if has_ind else None,
                feature_indicators=bnn_obj._feature_indicators if has_find else None,
                feature_means=bnn_obj._feature_means if has_find else None, prior_ind1=bnn_obj._prior_ind1, has_indicator_prior=has_ind,
                **self._device_chain_cfg(bnn_obj, smult, hast))
            first_it = self._current_iteration
            self._absorb_device_batch(bnn_obj, k, w_new, acc, res)
            if res["n_accepted"] > 0:
                if has_ind:
                    bnn_obj.reset_indicators(np.array(ind).reshape(bnn_obj._indicators.shape))
                if has_find:
                    bnn_obj._feature_indicators = np.array(find).astype(bnn_obj._feature_indicators.dtype)
                    last_acc = first_it + int(np.nonzero(acc[:k])[0][-1])
                    if last_acc > self._adapt_stop:      # what mh_step keeps for the statistics of the accepted state
                        self._accepted_override = data_transform_obj(bnn_obj._feature_indicators, bnn_obj._feature_means).column_override()

    def gibbs_step(self, bnn_obj):
        self._cancel_speculation()
        bnn_obj.sample_prior_scale()
        self._logPrior = bnn_obj.calc_prior()
        self._slope_term_in_prior = False       # (calc_prior is without the slope term, as in the reference: BNN_env.py:534-538)
        self._logPost = self._logLik + self._logPrior
        self._current_iteration += 1

    def reset_update_n(self, n):
        self._update_n = n

    def reset_update_f(self, f):
        self._update_f = f

    def reset_update_ws(self, w):
        self._update_ws = w

    def reset_temperature(self, temp):
        self._temperature = temp


def _lazy_property(name):
    def getter(self):
        if name not in self._lazy:
            self._lazy[name] = self._compute_lazy(name)
        return self._lazy[name]

    def setter(self, value):
        self._lazy[name] = value

    return property(getter, setter)


for _name in _LAZY:
    setattr(MCMC, _name, _lazy_property(_name))


def predict(bnn_obj: npBNN, data: np.ndarray):
    """Predictions of the current weights for a new data matrix (reference: BNN_env.py:662-670)."""
    from .layers import RunPredict
    transform = None
    if bnn_obj._feature_indicators is not None:
        transform = data_transform_obj(bnn_obj._feature_indicators, bnn_obj._feature_means)
    return RunPredict(data, bnn_obj._w_layers, actFun=bnn_obj._act_fun, output_act_fun=bnn_obj._output_act_fun,
                      data_transform=transform)
