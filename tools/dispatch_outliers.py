#!/usr/bin/env python3
"""Dispatches of 100 iterations that take far longer than their neighbours (a stall somewhere: device-side wait, allocation, the box):
   python tools/dispatch_outliers.py [config] [dispatches]     prints the median, the slowest ten and what the library's counters say."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_support import workload  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
wl = workload(cfg)
bnn, mcmc = wl.build()
ts = np.empty(n)
for i in range(n):
    t0 = time.perf_counter()
    mcmc.run_steps(bnn, 100)
    ts[i] = (time.perf_counter() - t0) * 1e3
order = np.argsort(ts)[::-1]
print("config %d: %d dispatches, median %.3f ms, mean %.3f ms; slowest: %s" % (cfg, n, np.median(ts), ts.mean(), ", ".join("#%d %.2f ms" % (i, ts[i]) for i in order[:10])))
print("dispatches over 3 x median: %d; schedule %d, device-side time-outs %d, void passes %d of %d" % (int(np.sum(ts > 3 * np.median(ts))), mcmc._device_schedule_used,
      mcmc._backend.ctx.sync_fallbacks, mcmc._device_void_passes, mcmc._device_passes))
mcmc._backend.close()
