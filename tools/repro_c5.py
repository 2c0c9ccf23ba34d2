#!/usr/bin/env python3
"""The config-5 leg of bench.py behind the other legs, in one process: per-dispatch wall clock of config 5 (and, with
NPBNN_CHAIN_TIMING=1, the library's own phase times), to see where a slow config-5 leg of a full bench run spends its time."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bench_support import workload  # noqa: E402


class A:
    steps, warmup = 20, 5


def leg(cfg, with_roofline):
    import gc
    if os.environ.get("NPBNN_REPRO_GC"):
        gc.collect()
    wl = workload(cfg)
    bnn, mcmc = wl.build()
    ts = []
    for i in range(25):
        t0 = time.perf_counter()
        mcmc.run_steps(bnn, 100)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("config %d: dispatches ms: %s" % (cfg, " ".join("%.2f" % t for t in ts)), flush=True)
    if with_roofline:
        bench.kernel_roofline(cfg, wl, mcmc._backend.ctx, bnn, mcmc, useful=2.7)
    for rep in range(3):
        t0 = time.perf_counter()
        mcmc.run_steps(bnn, 4000)
        print("config %d: one call of 4000: %.0f it/s (layer 0: %s, schedule %d)" % (cfg, 4000 / (time.perf_counter() - t0), mcmc._backend.ctx.l0_mode(),
                                                                                   mcmc._device_schedule_used), flush=True)
    mcmc._backend.close()
    mv = bench.moving_chain(wl)
    print("config %d moving:" % cfg, None if mv is None else round(mv["value"]), flush=True)


for cfg in [int(a) for a in sys.argv[1:]] or [2, 4, 5, 5]:
    leg(cfg, True)
