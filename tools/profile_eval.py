#!/usr/bin/env python3
"""Launch only the evaluation kernels of a BASELINE.json config (for rocprofv3).
   python tools/profile_eval.py [--config 2|4|5] [--cand 1|2|3] [--iters 20] [--general]
--cand 1 times the single evaluation (npbnn_eval's kernel), --cand > 1 the chain's pass kernel with that many candidates;
--general keeps the launches on the general build (NPBNN_OPT_FAST_TAILS = 0)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_support import workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--cand", type=int, default=1, help="proposals evaluated per launch (chain pass kernel when > 1)")
ap.add_argument("--general", action="store_true")
ap.add_argument("--spin", type=float, default=0.0, help="seconds of the same launches first: the device clocks up before the counted ones "
                                                        "(a fresh process finds it clocked down; bench.py's spin_up does the same)")
a = ap.parse_args()
wl = workload(a.config)
bnn, mcmc = wl.build()
ctx = mcmc._backend.ctx
if a.general:
    ctx.set_fast_tails(False)
alg = wl.bytes_per_proposal
if a.spin > 0:
    ms0 = ctx.time_pass(bnn._w_layers, n_candidates=a.cand, iters=50)[0] if a.cand > 1 else ctx.time_eval(bnn._w_layers, iters=50)[0]
    n_spin = max(50, int(a.spin * 1e3 / max(ms0, 1e-3)))
    if a.cand > 1:
        ctx.time_pass(bnn._w_layers, n_candidates=a.cand, iters=n_spin)
    else:
        ctx.time_eval(bnn._w_layers, iters=n_spin)
if a.cand > 1:
    ms_k, cand = ctx.time_pass(bnn._w_layers, n_candidates=a.cand, iters=a.iters)
    print("config %d: pass kernel with %d candidates %.2f us (one read of X / that = %.1f%% of 8 TB/s)"
          % (a.config, cand, ms_k * 1e3, 100 * alg / (ms_k * 1e-3) / 8e12))
else:
    ms_k, ms_t = ctx.time_eval(bnn._w_layers, iters=a.iters)
    print("config %d: eval kernel %.2f us (%.0f GB/s algorithmic, %.1f%% of 8 TB/s); kernel+finalize %.2f us"
          % (a.config, ms_k * 1e3, alg / ms_k / 1e6, 100 * alg / (ms_k * 1e-3) / 8e12, ms_t * 1e3))
mcmc._backend.close()
