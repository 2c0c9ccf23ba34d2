#!/usr/bin/env python3
"""NPBNN_EVAL_STAMPS=1 python tools/stamps_in_chain.py [config] [schedule]: the evaluating workgroups' phase stamps of the LAST pass of a
device batch (a pass in the middle of a persistent launch when schedule is 4 or 5: instruction and scalar caches warm), next to
tools/time_rows_sweep.py's, which are those of a launch of its own."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_support import workload  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sched = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ.pop("NPBNN_EVAL_STAMPS", None)
wl = workload(cfg)
bnn, mcmc = wl.build()
mcmc.device_schedule = sched
mcmc.run_steps(bnn, 1000)
os.environ["NPBNN_EVAL_STAMPS"] = os.environ.get("NPBNN_STAMPS_LEVEL", "1")
for _ in range(3):
    mcmc.run_steps(bnn, 100)
print("schedule used", mcmc._device_schedule_used)
mcmc._backend.close()
