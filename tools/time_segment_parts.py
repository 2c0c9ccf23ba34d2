#!/usr/bin/env python3
"""Where a 100-iteration run_steps segment spends its wall time (host side), config 2."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import build_config2  # noqa: E402
import npbnn_amd.sampler as smp  # noqa: E402

rs = np.random.default_rng(0)
x = rs.standard_normal((100_000, 256)).astype(np.float32)
y = rs.integers(0, 10, 100_000)
bnn, mcmc = build_config2(x, y, [32, 8], randomize_seed=True, mcmc_id=1)
mcmc.run_steps(bnn, 300)
acc = {}


def wrap(obj, name, key):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, g)


wrap(mcmc, "_claim_draw", "claim_draw")
wrap(mcmc, "_submit_draw", "submit_draw")
wrap(mcmc, "_run_device_batch", "device_batch")
wrap(mcmc._backend, "run_chain", "  backend.run_chain")
wrap(mcmc._backend.ctx, "chain_run", "    ctx.chain_run")
lib_fn = mcmc._backend.ctx._lib.npbnn_chain_run
n = 40
t0 = time.perf_counter()
for _ in range(n):
    mcmc.run_steps(bnn, 100)
el = time.perf_counter() - t0
print("segment of 100 iterations: %.0f us total" % (el / n * 1e6))
for k, v in acc.items():
    print("  %-24s %.0f us" % (k, v / n * 1e6))
