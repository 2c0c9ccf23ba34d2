#!/usr/bin/env python3
"""Launch only the passes of the weight-streamed leg of bench.py (20k x 4096, [256,64]) for rocprofv3:
   python tools/profile_wide.py [--iters 200]      (npbnn_time_wide: the first layer's product alone, then whole passes)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_support import workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--l0", default="auto")
a = ap.parse_args()
wl = workload(9)
bnn, mcmc = wl.build()
ctx = mcmc._backend.ctx
ctx.set_l0_precision(a.l0)
ms0, ms, geo = ctx.time_wide(bnn._w_layers, iters=a.iters)
print("wide leg (%s): first layer's product %.1f us, pass %.1f us, block %s" % (ctx.l0_mode(), 1e3 * ms0, 1e3 * ms, geo))
mcmc._backend.close()
