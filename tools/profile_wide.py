#!/usr/bin/env python3
"""Launch only the passes of a weight-streamed leg of bench.py for rocprofv3:
   python tools/profile_wide.py [--iters 200]                        20k x 4096, [256,64] (npbnn_time_wide: the first layer's product alone, then whole passes)
   python tools/profile_wide.py --config 8 --cand 3 [--iters 200]    100k x 1024, [50,5]: the fused pass with that many candidates (npbnn_time_pass)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_support import workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--l0", default="auto")
ap.add_argument("--config", type=int, default=9)
ap.add_argument("--cand", type=int, default=3)
a = ap.parse_args()
wl = workload(a.config)
bnn, mcmc = wl.build()
ctx = mcmc._backend.ctx
ctx.set_l0_precision(a.l0)
if a.config != 9:
    ms, cand = ctx.time_pass(bnn._w_layers, n_candidates=a.cand, iters=a.iters)
    print("%s (%s, streamed %d): fused pass with %d candidates %.1f us" % (wl.short, ctx.l0_mode(), ctx.is_wide(), cand, 1e3 * ms))
    mcmc._backend.close()
    sys.exit(0)
ms0, ms, geo = ctx.time_wide(bnn._w_layers, iters=a.iters)
print("wide leg (%s): first layer's product %.1f us, pass %.1f us, block %s" % (ctx.l0_mode(), 1e3 * ms0, 1e3 * ms, geo))
mcmc._backend.close()
