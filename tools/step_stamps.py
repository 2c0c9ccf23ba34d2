#!/usr/bin/env python3
"""NPBNN_STEP_STAMPS: phase times of the chain's step per pass, its period and the time it spends between two steps.
   python tools/step_stamps.py [config] [schedule ...]   (default: config 2, schedules 2 4 5; default proposals, then narrow ones)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_support import workload  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scheds = [int(a) for a in sys.argv[2:]] or [2, 4, 5]
wl = workload(cfg)
for uf in (None, [0.004] * 3):
    for sched in scheds:
        kw = {} if uf is None else dict(update_f=uf)
        bnn, m = wl.build(**kw)
        m.device_schedule = sched
        m.run_steps(bnn, 2000)
        os.environ["NPBNN_STEP_STAMPS"] = "1"
        m.run_steps(bnn, 1000)
        del os.environ["NPBNN_STEP_STAMPS"]
        print("update_f", uf, "schedule", sched, "ran", m._device_schedule_used, "acceptance", m._acceptance_rate, "its/pass",
              m._device_iterations / max(1, m._device_passes), flush=True)
        m._backend.close()
