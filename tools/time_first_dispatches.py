#!/usr/bin/env python3
"""Wall time of the first 40 dispatches of 100 iterations of a FRESH config-2 chain (the region bench.py times: warm-up 5, then 20) with the
schedule each ran on - how NPBNN_SCHED_AUTO behaves while the acceptance rate falls from 8 % to 1 %.  NPBNN_CHAIN_TIMING=1 adds the library's phases.
   python tools/time_first_dispatches.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
from bench_support import workload
wl = workload(2)
bnn, mcmc = wl.build()
ts = []
for i in range(40):
    t0 = time.perf_counter()
    mcmc.run_steps(bnn, 100)
    ts.append((time.perf_counter() - t0) * 1e6)
    if i in (4, 24, 39):
        print("after %d: accepted %d passes %d void %d sched %d" % (i + 1, mcmc._device_accepted, mcmc._device_passes, mcmc._device_void_passes, mcmc._device_schedule_used))
print(" ".join("%.0f" % t for t in ts))
