"""``MC3`` — Metropolis-coupled MCMC: independent heated chains + temperature swaps.

Mirror of the reference's chain dispatch (np_bnn/BNN_mc3.py:8-126) re-designed for one chain per
GPU.  The reference forks a process pool and, every ``swap_frequency`` iterations, pickles every chain
(model, data matrix, predictions) to a worker and back, although the swap decision reads two scalars per
chain.  Here chain state never moves:

  * single process (``comm`` is None / world size 1): all chains live in this process, each with its
    own resident device context; they advance one after the other;
  * one process per GPU (``comm.world_size > 1``): chain i lives on rank i % world_size; per swap
    interval the ranks all-gather ``[logPost, temperature]`` per chain (RCCL over xGMI), rank 0 draws the
    swap proposal from the parent-equivalent ``np.random`` stream (BNN_mc3.py:99,110) and broadcasts
    ``(j, k, accepted)``; only temperatures change hands.  The cold chain's log row is sent to rank 0,
    which owns the logger.
"""
import os
from copy import deepcopy

import numpy as np

from .comm import LocalComm
from .model import npBNN
from .sampler import MCMC


_SHARED = ("_data", "_test_data", "_labels", "_test_labels", "_sample_id")


def _replicate(model):
    """Per-chain copy of the model (reference: deepcopy, BNN_mc3.py:55-58).  The read-only data arrays
    are shared between the copies instead of being duplicated on the host."""
    new = model.__class__.__new__(model.__class__)
    for k, v in model.__dict__.items():
        if k == "_npbnn_backend":
            continue
        new.__dict__[k] = v if k in _SHARED else deepcopy(v)
    return new


class MC3():
    def __init__(self,
                 data: npBNN,
                 logger,
                 n_post_samples=100,
                 sampling_f=100,
                 n_chains=4,
                 swap_frequency=100,
                 verbose=1,
                 print_f=100,
                 temperatures=None,
                 min_temperature=0.8,
                 likelihood_f=None,
                 accuracy_f=None,
                 adapt_freq=50,
                 adapt_f=0.1,
                 adapt_fM=0.6,
                 adapt_stop=1000,
                 n_iteration=100000,
                 comm=None,
                 ):
        self.n_chains = n_chains
        self.swap_frequency = swap_frequency
        self.verbose = verbose
        self.print_f = print_f / swap_frequency
        self.n_post_samples = n_post_samples
        self.sampling_f = sampling_f
        self.adapt_freq, self.adapt_f, self.adapt_fM, self.adapt_stop = adapt_freq, adapt_f, adapt_fM, adapt_stop
        self.likelihood_f = likelihood_f
        self.n_mc3_iteration = np.round(n_iteration / swap_frequency).astype(int)
        self.accuracy_f = accuracy_f
        self.comm = comm if comm is not None else LocalComm()

        # chain seeds: drawn for stream parity with the reference (BNN_mc3.py:43)
        self.rseeds = np.random.choice(range(1000, 9999), n_chains, replace=False)
        if temperatures is None:
            temperatures = [1] if n_chains == 1 else np.linspace(min_temperature, 1, n_chains)
        self.temperatures = temperatures

        world, rank = self.comm.world_size, self.comm.rank
        self.local_ids = [i for i in range(n_chains) if i % world == rank]
        self.singleChainArgs = [None] * n_chains
        for i in self.local_ids:
            bnn_i = _replicate(data)
            bnn_i.reset_seed(self.rseeds[i])
            kwargs = dict(temperature=self.temperatures[i], n_iteration=self.swap_frequency,
                          sampling_f=self.sampling_f, print_f=self.swap_frequency * 10,
                          n_post_samples=self.n_post_samples, mcmc_id=i, randomize_seed=True,
                          adapt_freq=self.adapt_freq, adapt_f=self.adapt_f, adapt_fM=self.adapt_fM,
                          adapt_stop=self.adapt_stop, likelihood_f=self.likelihood_f, accuracy_f=self.accuracy_f)
            self.singleChainArgs[i] = [bnn_i, MCMC(bnn_i, **kwargs)]
        self.logger = logger
        self.swap_log = []

    def run_single_mcmc(self, arg_list):
        """Advance one chain by ``swap_frequency`` iterations (reference: BNN_mc3.py:80-85)."""
        [bnn_obj, mcmc_obj] = arg_list
        mcmc_obj.run_steps(bnn_obj, self.swap_frequency)
        return [bnn_obj, mcmc_obj]

    # -- the exchange step ---------------------------------------------------------------------
    def _local_chains(self):
        return [self.singleChainArgs[i] for i in self.local_ids]

    def _log_chain(self, i, pair, save_pickle=True):
        """Log chain ``i`` (the cold one) through rank 0's logger.  ``pair`` = (bnn, mcmc) on the rank that holds the
        chain, ignored elsewhere; with several ranks a light view of the chain (no data matrix, no predictions) is shipped
        to rank 0."""
        world = self.comm.world_size
        owner = i % world
        if world == 1:
            bnn_i, mcmc_i = pair
        else:
            view = None
            if self.comm.rank == owner:
                bnn_i, mcmc_i = pair
                small = {name: getattr(mcmc_i, name) for name in ("_accuracy", "_test_accuracy", "_label_acc",
                                                                   "_label_freq")}
                light = {k: v for k, v in bnn_i.__dict__.items() if k not in _SHARED and k != "_npbnn_backend"}
                state = {k: v for k, v in mcmc_i.__dict__.items() if k not in ("_backend", "_bnn", "_lazy", "_speculation", "_speculation2", "_ws_copies", "_fast")}
                state["_lazy"] = small
                view = (light, state)
            view = self.comm.bcast_obj(view, root=owner)
            if self.comm.rank != 0:
                return
            bnn_i = npBNN.__new__(npBNN)
            bnn_i.__dict__.update(view[0])
            bnn_i._data, bnn_i._test_data, bnn_i._labels, bnn_i._test_labels = np.zeros((0, 0)), [], [], []
            mcmc_i = MCMC.__new__(MCMC)
            mcmc_i.__setstate__(view[1])
            mcmc_i._lazy.setdefault("_y", [])
            mcmc_i._lazy.setdefault("_y_test", [])
            mcmc_i._bnn = bnn_i
        if self.comm.rank == 0 and self.logger is not None:
            self.logger.log_sample(bnn_i, mcmc_i)
            if save_pickle:
                self.logger.log_weights(bnn_i, mcmc_i)
            else:
                self.logger.log_weights(bnn_i, mcmc_i, save_pickle=False)

    def _log_cold_chains(self, scal):
        for i in range(self.n_chains):
            if scal[i, 1] == 1:
                self._log_chain(i, self.singleChainArgs[i])

    @staticmethod
    def _view_at_swap(bnn, mcmc, snap):
        """(bnn, mcmc) as the chain stood at a swap inside a device batch: the weights the device saved for the cold chain,
        its log-likelihood / prior, the iteration count and the acceptance book-keeping up to there.  The accuracy
        statistics of the view are computed on demand from those weights, as for a live chain."""
        layers, off = [], 0
        for w in bnn._w_layers:
            layers.append(np.array(snap["w"][off:off + w.size]).reshape(w.shape))
            off += w.size
        bnn_v = bnn.__class__.__new__(bnn.__class__)
        bnn_v.__dict__.update(bnn.__dict__)
        bnn_v._w_layers = layers
        if bnn._estimation_mode == "regression" and len(np.atleast_1d(snap.get("sigma", ()))) >= bnn._size_output:
            bnn_v._error_prm = np.array(snap["sigma"][:bnn._size_output], dtype=float)      # (the row's sig_* columns, BNN_env.py:611-612)
        m_v = mcmc.__class__.__new__(mcmc.__class__)
        m_v.__dict__.update(mcmc.__dict__)
        m_v._speculation = m_v._speculation2 = None
        m_v._fast = None
        m_v._bnn = bnn_v
        m_v._lazy = {}
        m_v._accepted_override = None
        k = int(snap["iterations"])
        m_v._logLik, m_v._logPrior = float(snap["loglik"]), float(snap["logprior"])
        m_v._logPost = m_v._logLik + m_v._logPrior
        m_v._current_iteration = snap["iteration0"] + k
        m_v._temperature = 1
        history = list(snap["mem_before"]) + [int(v) for v in snap["accepted"][:k]]
        m_v._last_accepted = history[-1]
        m_v._acceptance_rate = np.mean(history[-101:]) if len(history) > 100 else np.mean(history)
        m_v._last_accepted_mem = history[-100:] if len(history) > 100 else history
        return bnn_v, m_v

    exchange_batch = 20      # swap intervals per device call
    # swap intervals in device batches with the swaps decided on the GPU (True), or one device batch per interval and the swap
    # on the host (False: the reference's rhythm).  None = the library's choice: device batches for the chains of ONE process;
    # with several ranks the host path, until the in-place RCCL all-gather has been run between real ranks (set True to opt in).
    device_exchange = None
    group_passes = False     # interval-by-interval path: the local chains share their passes over the data (exchange.run_steps_batched);
    #                          "auto": whenever they have been accepting more than exchange.GROUP_PASS_ACCEPTANCE of their proposals

    def run_mcmc(self):
        """The MC3 loop (reference: BNN_mc3.py:87-126): ``n_mc3_iteration`` rounds of [swap_frequency iterations of every
        chain, one swap proposal, log the cold chain].  Rounds run in device batches (:mod:`npbnn_amd.exchange`) once the
        proposal adaptation of the chains is over; the cold chain's sample at every swap inside a batch is logged from the
        state the device saved for it."""
        from . import exchange as ex
        if getattr(self, "_swaps", None) is None:
            self._swaps = ex.SwapProposals(max(self.n_chains, 2))       # drawn from np.random, as BNN_mc3.py:99,110
        chains = self._local_chains()

        def on_interval(index, info):
            mc3_it = self._mc3_it
            self._mc3_it += 1
            scal, (j, k, r, log_u, accepted) = info["scalars"], info["swap"]
            if accepted and self.verbose > 0 and self.comm.rank == 0:
                print(mc3_it, "SWAPPED", scal[j, 0], scal[k, 0], scal[k, 1], scal[j, 1])
            self.swap_log.append((j, k, float(r), float(log_u), bool(accepted)))
            if info["cold"] is None:                    # the interval ran on the per-interval path: live chains
                self._log_cold_chains(scal)
            else:
                for i in range(self.n_chains):
                    if scal[i, 1] != 1:
                        continue
                    pair = None
                    if i in self.local_ids:
                        q = self.local_ids.index(i)
                        pair = self._view_at_swap(chains[q][0], chains[q][1], info["cold"][q])
                    self._log_chain(i, pair, save_pickle=info["last_of_batch"])
            if mc3_it % self.print_f == 0 and self.comm.rank == 0 and self.singleChainArgs[0] is not None:
                print(mc3_it, scal[0, 0], self.singleChainArgs[0][0]._w_layers[0][0][0:5])

        self._mc3_it = 0
        done = 0
        while done < self.n_mc3_iteration:
            n = min(self.exchange_batch, self.n_mc3_iteration - done)
            if self.n_chains > 1:
                got = ex.advance_intervals(chains, self.local_ids, self.n_chains, n, self.swap_frequency, self._swaps, done,
                                           comm=self.comm if self.comm.world_size > 1 else None, batch=n,
                                           device=(self.comm.world_size == 1) if self.device_exchange is None else self.device_exchange,
                                           on_interval=on_interval, group_passes=self.group_passes)
            else:                                       # a single chain: no swaps, the cold chain is logged every interval
                for _ in range(n):
                    for i in self.local_ids:
                        self.singleChainArgs[i] = self.run_single_mcmc(self.singleChainArgs[i])
                    self._log_cold_chains(self._gather_scalars())
                    if self._mc3_it % self.print_f == 0 and self.singleChainArgs[0] is not None:
                        print(self._mc3_it, self.singleChainArgs[0][1]._logPost, self.singleChainArgs[0][0]._w_layers[0][0][0:5])
                    self._mc3_it += 1
                got = n
            done += got

    def _gather_scalars(self):
        """[logPost, temperature] of every chain, on every rank."""
        from . import exchange as ex
        return ex.gather_scalars(self._local_chains(), self.local_ids, self.n_chains,
                                 self.comm if self.comm.world_size > 1 else None)


def default_comm():
    """Communicator for this process: RCCL when launched with one rank per GPU, else local."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from .comm import RcclComm
        return RcclComm()
    return LocalComm()
