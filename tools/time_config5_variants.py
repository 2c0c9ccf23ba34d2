#!/usr/bin/env python3
"""Config 5 (block-structured layer 0: 3125 row tiles over 255 x 12 waves leave 64 workgroups a 13th tile) with other candidate counts and the
general builds (14 waves at D = 2: one round of tiles): does fewer candidates on more waves beat the lone second round?  It does not.
   python tools/time_config5_variants.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
from bench_support import workload
wl = workload(5)
for cand, fast in ((3, True), (2, True), (2, False), (3, False), (1, True)):
    bnn, mcmc = wl.build()
    mcmc.n_candidates = cand
    mcmc._backend.ctx.set_fast_tails(fast)
    mcmc.run_steps(bnn, 3000)
    t0 = time.perf_counter()
    for _ in range(20):
        mcmc.run_steps(bnn, 100)
    el = time.perf_counter() - t0
    ms, used = mcmc._backend.ctx.time_pass(bnn._w_layers, n_candidates=cand, iters=100)
    print("config 5 D=%d fast=%s: %.0f it/s in calls of 100; pass kernel %.2f us (%d candidates); %.2f iterations per pass, schedule %d"
          % (cand, fast, 2000 / el, ms * 1e3, used, mcmc._device_iterations / max(1, mcmc._device_passes), mcmc._device_schedule_used), flush=True)
    mcmc._backend.close()
