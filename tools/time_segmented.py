#!/usr/bin/env python3
"""Throughput of run_steps when the chain is advanced in short segments (the MC3 pattern: a temperature-swap point
every `seg` iterations), against one long call.  Config 2, one GPU."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import build_config2  # noqa: E402

rs = np.random.default_rng(0)
x = rs.standard_normal((100_000, 256)).astype(np.float32)
y = rs.integers(0, 10, 100_000)
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
total = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
for randomize in (False, True):
    bnn, mcmc = build_config2(x, y, [32, 8], randomize_seed=randomize, mcmc_id=1)
    mcmc.run_steps(bnn, 200)
    t0 = time.perf_counter()
    for _ in range(total // seg):
        mcmc.run_steps(bnn, seg)
    el = time.perf_counter() - t0
    t1 = time.perf_counter()
    mcmc.run_steps(bnn, total)
    el1 = time.perf_counter() - t1
    print("randomize_seed=%s: segments of %d: %.0f it/s (%.1f us per segment beyond compute); one call: %.0f it/s"
          % (randomize, seg, total / el, (el - el1) / (total // seg) * 1e6, total / el1))


class _NoSwaps:
    """Swap proposals that never swap (j = k): times the exchange machinery with a single chain on one GPU."""

    def get(self, first, n=1):
        return np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), np.zeros(n)

    def release(self, upto):
        pass


from npbnn_amd import exchange as ex  # noqa: E402

bnn, mcmc = build_config2(x, y, [32, 8], randomize_seed=True, mcmc_id=1)
mcmc.run_steps(bnn, 2000)
comm = None
if os.environ.get("NPBNN_SEG_WITH_RCCL"):        # the records of every exchange go through ncclAllGather (one rank) on the launch stream
    from npbnn_amd.comm import RcclComm
    comm = RcclComm(rank=0, world_size=1, device=0)
for batch in (10, 20, 50):
    n_seg = (2 * total // seg) // batch * batch
    ex.advance_intervals([(bnn, mcmc)], [0], 1, batch, seg, _NoSwaps(), 0, batch=batch, comm=comm)       # warm-up (buffers, draws ahead)
    t0 = time.perf_counter()
    done = ex.advance_intervals([(bnn, mcmc)], [0], 1, n_seg, seg, _NoSwaps(), 0, batch=batch, comm=comm)
    el = time.perf_counter() - t0
    t1 = time.perf_counter()
    mcmc.run_steps(bnn, n_seg * seg)
    el1 = time.perf_counter() - t1
    from npbnn_amd.hip_backend import HipBackend
    print("exchange run, %d segments of %d in batches of %d: %.0f it/s (%.1f us per segment beyond compute; launch slack %.2f); one call: %.0f it/s"
          % (done, seg, batch, n_seg * seg / el, (el - el1) / n_seg * 1e6, HipBackend.exchange_slack, n_seg * seg / el1))
