#!/usr/bin/env python3
"""One call of 10 000 iterations against the size sub-batches may grow to (MCMC.SUB_BATCH_MAX): the draws of a sub-batch are made while
the GPU runs the one before it.   python tools/time_long_calls.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from bench_support import workload
import npbnn_amd as bn
for cfg in (2, 5, 4):
    for cap in (2048, 512, 256):
        bn.MCMC.SUB_BATCH_MAX = cap
        wl = workload(cfg)
        bnn, mcmc = wl.build()
        mcmc.run_steps(bnn, 3000)
        t0 = time.perf_counter(); mcmc.run_steps(bnn, 10000); el = time.perf_counter() - t0
        print("config %d, sub-batches up to %4d: one call of 10000 -> %.0f it/s (acceptance %.3f)" % (cfg, cap, 10000 / el, mcmc._acceptance_rate), flush=True)
        mcmc._backend.close()
