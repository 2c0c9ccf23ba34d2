#!/usr/bin/env python3
"""Where does a chain dispatch of 100 iterations spend its host time?  cProfile over run_steps(100) calls on a burnt-in config-2
chain (tools/time_short_calls.py gives the wall clock; NPBNN_CHAIN_TIMING=1 the library's own phases).
   python tools/profile_dispatch.py [calls]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import workload  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bnn, mcmc = workload(int(os.environ.get("NPBNN_CONFIG", "2"))).build()
mcmc.run_steps(bnn, 3000)
for _ in range(20):
    mcmc.run_steps(bnn, 100)
ctx = mcmc._backend.ctx
c0 = ctx.seconds_in_chain_run
t0 = time.perf_counter()
for _ in range(calls):
    mcmc.run_steps(bnn, 100)
el = time.perf_counter() - t0
inside = ctx.seconds_in_chain_run - c0
print("run_steps(100): %.1f us per call (%d calls): %.1f us inside npbnn_chain_run, %.1f us of Python around it"
      % (1e6 * el / calls, calls, 1e6 * inside / calls, 1e6 * (el - inside) / calls))
for n in (1, 3, 10):
    c0 = ctx.seconds_in_chain_run
    t0 = time.perf_counter()
    for _ in range(calls):
        mcmc.run_steps(bnn, n)
    el = time.perf_counter() - t0
    inside = ctx.seconds_in_chain_run - c0
    print("run_steps(%d): %.1f us per call: %.1f us inside npbnn_chain_run, %.1f us of Python around it"
          % (n, 1e6 * el / calls, 1e6 * inside / calls, 1e6 * (el - inside) / calls))
pr = cProfile.Profile()
pr.enable()
for _ in range(calls):
    mcmc.run_steps(bnn, 100)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
mcmc._backend.close()
