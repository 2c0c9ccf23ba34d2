#!/usr/bin/env python3
"""Wall time of each of the first run_steps(100) calls of a freshly initialised chain (the region bench.py's default run
measures), with what the device did in it.   python tools/time_first_calls.py [n_calls] [config]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import workload  # noqa: E402

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bnn, mcmc = workload(int(sys.argv[2]) if len(sys.argv) > 2 else 2).build()
if os.environ.get("NPBNN_BENCH_SCHEDULE"):
    mcmc.device_schedule = int(os.environ["NPBNN_BENCH_SCHEDULE"])
rows = []
for i in range(n_calls):
    p0, v0, a0 = mcmc._device_passes, mcmc._device_void_passes, mcmc._device_accepted
    t0 = time.perf_counter()
    mcmc.run_steps(bnn, 100)
    el = time.perf_counter() - t0
    rows.append((i, el * 1e6, mcmc._device_passes - p0, mcmc._device_void_passes - v0, mcmc._device_accepted - a0, mcmc._device_schedule_used))
for r in rows:
    print("call %2d: %7.0f us  passes %3d  void %2d  accepted %2d  schedule %d" % r)
mcmc._backend.close()
