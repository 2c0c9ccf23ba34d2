#!/usr/bin/env python3
"""Soak of the weight-streamed path: the bench's two streamed workloads and a moving chain on each run for a few minutes in dispatches of
100 (the MC3 rhythm) and in long calls, the chain's log-likelihood checked against a fresh device evaluation and the float64 oracle
along the way.   python tools/soak_wide.py [seconds per leg]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import workload  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
for config, kw in ((8, {}), (8, dict(update_f=[0.002, 0.01, 0.05])), (9, dict(update_f=[0.0005, 0.005, 0.05]))):
    wl = workload(config)
    bnn, mcmc = wl.build(**kw)
    ctx = mcmc._backend.ctx
    t0 = time.perf_counter()
    n_calls = 0
    while time.perf_counter() - t0 < budget:
        mcmc.run_steps(bnn, 100 if n_calls % 5 else 1000)
        n_calls += 1
        if n_calls % 50 == 0:
            dev = mcmc._backend.evaluate(bnn._w_layers, None)["loglik"]
            assert abs(dev - mcmc._logLik) <= 1e-9 * abs(dev), (dev, mcmc._logLik)
            print("  %s %s: %d iterations, %.0f it/s, accepted %.3f, %.2f iterations per pass, loglik %.3f" %
                  (wl.short, kw or "", mcmc._device_iterations, mcmc._device_iterations / (time.perf_counter() - t0),
                   mcmc._device_accepted / max(1, mcmc._device_iterations), mcmc._device_iterations / max(1, mcmc._device_passes), mcmc._logLik), flush=True)
    par = wl.parity(bnn, mcmc)
    print("%s %s: %d iterations in %d calls, path %s, parity %s" % (wl.short, kw or "", mcmc._device_iterations, n_calls,
                                                                      "weight-streamed" if ctx.is_wide() else "resident", par), flush=True)
    assert par["chain_loglik_rel_err"] < 1e-6
    mcmc._backend.close()
print("soak ok")
