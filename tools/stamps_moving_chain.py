#!/usr/bin/env python3
"""Phase stamps of the evaluating workgroups in a pass of the decision-between-passes schedule (config 2, chain that accepts a quarter of its
proposals): build the speculative translation unit with -DNPBNN_EXP_PROLOGUE_STAMPS for the prologue points and the period.  Shows the wait
for the step workgroup's decision (pass descriptor 7-8 us after a workgroup's turn starts).   python tools/stamps_moving_chain.py"""
import os, sys
sys.path.insert(0, os.getcwd())
from bench_support import workload
wl = workload(2)
bnn, mcmc = wl.build(update_f=list(wl.moving_update_f))
mcmc.device_schedule = 5
mcmc.run_steps(bnn, 2000)
os.environ["NPBNN_EVAL_STAMPS"] = os.environ.get("NPBNN_STAMPS_LEVEL", "1")
os.environ["NPBNN_CHAIN_TIMING"] = "1"
for _ in range(2):
    mcmc.run_steps(bnn, 100)
print("schedule used", mcmc._device_schedule_used)
