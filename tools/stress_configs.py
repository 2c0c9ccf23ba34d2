#!/usr/bin/env python3
"""The chain schedules against each other on bench.py's own workloads (configs 4, 5 and the default network; tools/stress_schedules.py
does config 2's shape): the same chain bit for bit under schedules 2, 4, 5 and the automatic choice - default and moving proposals - and
no device-side wait timed out.   python tools/stress_configs.py [iterations] [config ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_support import workload  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
configs = [int(v) for v in sys.argv[2:]] or [4, 5, 0]
bad = 0
for cfg in configs:
    wl = workload(cfg)
    for moving in (False, True):
        if moving and wl.moving_update_f is None:
            continue
        ref = None
        for sched in (2, 4, 5, 0):
            kw = dict(update_f=list(wl.moving_update_f)) if moving else {}
            bnn, mcmc = wl.build(**kw)
            mcmc.device_schedule = sched
            t0 = time.perf_counter()
            done = 0
            while done < n_it:                       # (calls of 100 and long calls in turn: both dispatch forms)
                k = 100 if (done // 100) % 8 else 1100
                mcmc.run_steps(bnn, k)
                done += k
            el = time.perf_counter() - t0
            state = (np.concatenate([w.ravel() for w in bnn._w_layers]), float(mcmc._logLik), float(mcmc._logPrior), int(mcmc._device_accepted))
            same = ref is None or (np.array_equal(state[0], ref[0]) and state[1:] == ref[1:])
            ref = ref or state
            fb = int(mcmc._backend.ctx.sync_fallbacks)
            bad += (not same) + (fb > 0)
            print("config %d %s schedule %d (ran %d): %.0f it/s, accepted %d of %d, time-outs %d, same chain as schedule 2: %s"
                  % (cfg, "moving" if moving else "default", sched, mcmc._device_schedule_used, done / el, state[3], done, fb, same), flush=True)
            mcmc._backend.close()
print("FAILED" if bad else "all chains identical, no time-outs")
sys.exit(1 if bad else 0)
