#!/usr/bin/env python3
"""Several MC3 chains on ONE GPU (config 2): aggregate iterations/s when the chains advance together through the exchange run
(every chain on its own stream: one chain's launches fill the compute units as another's drain) against one chain after the
other.  SURVEY 8(f) item 2: MC3 with more chains than GPUs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import build_config2  # noqa: E402
from npbnn_amd import exchange as ex  # noqa: E402

rs = np.random.default_rng(0)
x = rs.standard_normal((100_000, 256)).astype(np.float32)
y = rs.integers(0, 10, 100_000)
seg, n_seg = 100, 40
for n_chains in (1, 2, 3, 4):
    temps = np.linspace(0.8, 1.0, n_chains) if n_chains > 1 else [1.0]
    chains = [build_config2(x, y, [32, 8], mcmc_id=i, temperature=float(temps[i]), randomize_seed=True) for i in range(n_chains)]
    for bnn, m in chains:
        m.run_steps(bnn, 1000)
    ids = list(range(n_chains))

    class NoSwap:
        def get(self, first, n=1):
            return np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), np.zeros(n)

        def release(self, upto):
            pass
    swaps = ex.SwapProposals(n_chains, np.random.RandomState(1)) if n_chains > 1 else NoSwap()
    ex.advance_intervals(chains, ids, n_chains, 20, seg, swaps, 0, batch=20)
    t0 = time.perf_counter()
    ex.advance_intervals(chains, ids, n_chains, n_seg, seg, swaps, 20, batch=20)
    el = time.perf_counter() - t0
    t1 = time.perf_counter()
    ex.advance_intervals(chains, ids, n_chains, n_seg // 2, seg, swaps, 60, batch=20, device=False)
    el1 = time.perf_counter() - t1
    print("%d chain(s) on one GPU, swap every %d: together (exchange run) %.0f it/s aggregate = %.0f per chain; one after the other %.0f it/s aggregate"
          % (n_chains, seg, n_chains * n_seg * seg / el, n_seg * seg / el, n_chains * (n_seg // 2) * seg / el1))
    for bnn, m in chains:
        m._backend.close()
