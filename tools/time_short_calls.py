#!/usr/bin/env python3
"""Cost of short run_steps calls (the MC3 rhythm: swap_frequency = 100 iterations per call, np_bnn/BNN_mc3.py:80-85) on
config 2: iterations/s as a function of the call length, after the chain has burnt in.
   python tools/time_short_calls.py [call lengths ...]        NPBNN_CHAIN_TIMING=1 adds the library's own phase timing"""
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import workload  # noqa: E402

lengths = [int(v) for v in sys.argv[1:]] or [20, 100, 500, 2000]
bnn, mcmc = workload(int(os.environ.get('NPBNN_CONFIG', '2'))).build()
mcmc.run_steps(bnn, 3000)
for n in lengths:
    calls = max(5, 4000 // n)
    mcmc.run_steps(bnn, n)
    t0 = time.perf_counter()
    for _ in range(calls):
        mcmc.run_steps(bnn, n)
    el = time.perf_counter() - t0
    print("run_steps(%d) x %d: %.1f us per call, %.2f us per iteration, %.0f it/s (acceptance %.3f, schedule %d)"
          % (n, calls, 1e6 * el / calls, 1e6 * el / calls / n, n * calls / el, mcmc._acceptance_rate, mcmc._device_schedule_used), flush=True)
mcmc._backend.close()
