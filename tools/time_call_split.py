#!/usr/bin/env python3
"""run_steps(100) on config 2: time inside npbnn_chain_run against the time of the whole call (the rest is host Python)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import workload  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bnn, mcmc = workload(int(os.environ.get('NPBNN_CONFIG', '2'))).build()
mcmc.run_steps(bnn, 3000)
ctx = mcmc._backend.ctx
inner = [0.0, 0]
from npbnn_amd import _capi as capi  # noqa: E402
real = capi.chain_run_by_address(ctx._lib)


class Timed:
    def __call__(self, *a):
        t0 = time.perf_counter()
        rc = real(*a)
        inner[0] += time.perf_counter() - t0
        inner[1] += 1
        return rc


ctx._lib._npbnn_chain_run_by_address = Timed()
for _ in range(5):
    mcmc.run_steps(bnn, n)
inner[0], inner[1] = 0.0, 0
calls = 300
t0 = time.perf_counter()
for _ in range(calls):
    mcmc.run_steps(bnn, n)
el = time.perf_counter() - t0
print("run_steps(%d): %.1f us per call, of which %.1f us inside npbnn_chain_run (%d C calls); %.0f it/s"
      % (n, 1e6 * el / calls, 1e6 * inner[0] / calls, inner[1], n * calls / el))
ctx._lib._npbnn_chain_run_by_address = real
mcmc._backend.close()
