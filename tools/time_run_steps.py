#!/usr/bin/env python3
"""Where does the wall time of MCMC.run_steps go?  (config 2; run with NPBNN_CHAIN_TIMING=1 for the C side)"""
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
bnn, mcmc = build_config2(x, y, [32, 8])
mcmc.run_steps(bnn, 200)
orig = mcmc._run_device_batch
acc = {"batch": 0.0, "n": 0}


def timed(*a, **k):
    t = time.perf_counter()
    r = orig(*a, **k)
    acc["batch"] += time.perf_counter() - t
    acc["n"] += 1
    return r


mcmc._run_device_batch = timed
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
t0 = time.perf_counter()
mcmc.run_steps(bnn, steps)
el = time.perf_counter() - t0
print("run_steps(%d): %.2f ms total = %.2f us/iteration; %d batches, %.2f ms inside _run_device_batch, %.2f ms outside"
      % (steps, el * 1e3, el * 1e6 / steps, acc["n"], acc["batch"] * 1e3, (el - acc["batch"]) * 1e3))
print("iterations per pass: %.3f" % (mcmc._device_iterations / mcmc._device_passes))
