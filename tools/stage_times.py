#!/usr/bin/env python3
"""NPBNN_CHAIN_TIMING=2 python tools/stage_times.py: the stages of a dispatch of 100 iterations (config 2, burnt-in chain), the stream synchronised
after each one - what each costs alone (8-10 us of every figure are the synchronisation), not what a dispatch costs (they overlap)."""
import os, sys
sys.path.insert(0, os.getcwd())
from bench_support import workload
wl = workload(2)
bnn, mcmc = wl.build()
mcmc.run_steps(bnn, 3000)
os.environ["NPBNN_CHAIN_TIMING"] = "2"
for _ in range(3):
    mcmc.run_steps(bnn, 100)
