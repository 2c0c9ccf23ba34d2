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
                 backend_factory=None,
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
            if backend_factory is not None:
                kwargs["backend"] = backend_factory(bnn_i)
            self.singleChainArgs[i] = [bnn_i, MCMC(bnn_i, **kwargs)]
        self.logger = logger
        self.swap_log = []

    def run_single_mcmc(self, arg_list):
        """Advance one chain by ``swap_frequency`` iterations (reference: BNN_mc3.py:80-85)."""
        [bnn_obj, mcmc_obj] = arg_list
        mcmc_obj.run_steps(bnn_obj, self.swap_frequency)
        return [bnn_obj, mcmc_obj]

    # -- the exchange step ---------------------------------------------------------------------
    def _gather_scalars(self):
        """[logPost, temperature] of every chain, on every rank."""
        world = self.comm.world_size
        per_rank = (self.n_chains + world - 1) // world
        mine = np.full((per_rank, 2), np.nan)
        for slot, i in enumerate(self.local_ids):
            m = self.singleChainArgs[i][1]
            mine[slot] = (m._logPost, m._temperature)
        allv = self.comm.allgather_f64(mine.ravel()).reshape(world, per_rank, 2)
        out = np.empty((self.n_chains, 2))
        for i in range(self.n_chains):
            out[i] = allv[i % world, i // world]
        return out

    def _swap(self, mc3_it):
        scal = self._gather_scalars()
        decision = np.zeros(3, dtype=np.int64)
        r = log_u = np.nan
        if self.comm.rank == 0:
            j, k = np.random.choice(range(self.n_chains), 2, replace=False)
            temp_j, temp_k = scal[j, 1] + 0, scal[k, 1] + 0
            r = (scal[k, 0] - scal[j, 0]) * temp_j + (scal[j, 0] - scal[k, 0]) * temp_k
            log_u = np.log(np.random.random())
            decision[:] = (j, k, 1 if r >= log_u else 0)
        j, k, accepted = (int(v) for v in self.comm.bcast_i64(decision, root=0))
        if accepted:
            temp_j, temp_k = scal[j, 1] + 0, scal[k, 1] + 0
            if self.singleChainArgs[j] is not None:
                self.singleChainArgs[j][1].reset_temperature(temp_k)
            if self.singleChainArgs[k] is not None:
                self.singleChainArgs[k][1].reset_temperature(temp_j)
            if self.verbose > 0 and self.comm.rank == 0:
                print(mc3_it, "SWAPPED", scal[j, 0], scal[k, 0], temp_j, temp_k)
            scal[j, 1], scal[k, 1] = temp_k, temp_j
        self.swap_log.append((j, k, float(r), float(log_u), bool(accepted)))
        return scal

    def _log_cold_chains(self, scal):
        world = self.comm.world_size
        for i in range(self.n_chains):
            if scal[i, 1] != 1:
                continue
            owner = i % world
            if world == 1:
                bnn_i, mcmc_i = self.singleChainArgs[i]
            else:
                view = None
                if self.comm.rank == owner:
                    bnn_i, mcmc_i = self.singleChainArgs[i]
                    small = {name: getattr(mcmc_i, name) for name in ("_accuracy", "_test_accuracy", "_label_acc",
                                                                       "_label_freq")}
                    light = {k: v for k, v in bnn_i.__dict__.items() if k not in _SHARED and k != "_npbnn_backend"}
                    state = {k: v for k, v in mcmc_i.__dict__.items() if k not in ("_backend", "_bnn", "_lazy")}
                    state["_lazy"] = small
                    view = (light, state)
                view = self.comm.bcast_obj(view, root=owner)
                if self.comm.rank != 0:
                    continue
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
                self.logger.log_weights(bnn_i, mcmc_i)

    def run_mcmc(self):
        """The MC3 loop (reference: BNN_mc3.py:87-126)."""
        for mc3_it in range(self.n_mc3_iteration):
            for i in self.local_ids:
                self.singleChainArgs[i] = self.run_single_mcmc(self.singleChainArgs[i])
            if self.n_chains > 1:
                scal = self._swap(mc3_it)
            else:
                scal = self._gather_scalars()
            self._log_cold_chains(scal)
            if mc3_it % self.print_f == 0 and self.comm.rank == 0 and self.singleChainArgs[0] is not None:
                print(mc3_it, self.singleChainArgs[0][1]._logPost, self.singleChainArgs[0][0]._w_layers[0][0][0:5])


def default_comm():
    """Communicator for this process: RCCL when launched with one rank per GPU, else local."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from .comm import RcclComm
        return RcclComm()
    return LocalComm()
