#!/usr/bin/env python3
"""How fast is this box?  Config 2's three-candidate pass kernel behind a spin-up (boxes of the pool read 25.3-30.7 us for the same binary:
they sustain different shader clocks).  Prints the figure; exit code 7 when it is above the limit given (us) - so that a collection
of profiles can ask for a box like the one earlier rounds' files came from:   python tools/box_speed.py 25.9 && bash tools/collect_profiles.sh r05"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import workload  # noqa: E402

bnn, mcmc = workload(2).build()
ctx = mcmc._backend.ctx
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    ctx.time_pass(bnn._w_layers, n_candidates=3, iters=200)
ms = min(ctx.time_pass(bnn._w_layers, n_candidates=3, iters=1000)[0] for _ in range(3))
print("config 2, three candidates per pass: %.2f us" % (1e3 * ms), flush=True)
mcmc._backend.close()
sys.exit(7 if len(sys.argv) > 1 and 1e3 * ms > float(sys.argv[1]) else 0)
