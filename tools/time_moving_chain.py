#!/usr/bin/env python3
"""A chain that moves (about a quarter of the proposals accepted) under each schedule: iterations/s in calls of 100 and in one long
call, on bench.py's moving-chain workload.   python tools/time_moving_chain.py [config] [schedule ...]
NPBNN_CHAIN_TIMING=1 adds the library's phase timing (and, for schedule 5, the step workgroup's per-round phases)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import workload  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scheds = [int(v) for v in sys.argv[2:]] or [0, 1, 4, 5]
wl = workload(cfg)
for sched in scheds:
    kw = dict(update_f=list(wl.moving_update_f)) if (wl.moving_update_f is not None and not os.environ.get('NPBNN_DEFAULT_PROPOSALS')) else {}
    bnn, mcmc = wl.build(**kw)
    mcmc.device_schedule = sched
    mcmc.run_steps(bnn, 2000)
    t0 = time.perf_counter()
    for _ in range(20):
        mcmc.run_steps(bnn, 100)
    el = time.perf_counter() - t0
    t0 = time.perf_counter()
    mcmc.run_steps(bnn, 4000)
    el2 = time.perf_counter() - t0
    print("config %d schedule %d (ran %d): %.0f it/s in calls of 100, %.0f it/s in one call of 4000; acceptance %.3f, %.2f iterations per pass"
          % (cfg, sched, mcmc._device_schedule_used, 2000 / el, 4000 / el2, mcmc._device_accepted / max(1, mcmc._device_iterations),
             mcmc._device_iterations / max(1, mcmc._device_passes)), flush=True)
    mcmc._backend.close()
