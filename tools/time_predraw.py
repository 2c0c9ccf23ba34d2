#!/usr/bin/env python3
"""What the host's numpy-identical pre-draw costs per iteration and per perturbed weight (one thread; npbnn_amd/csrc/npbnn_host.c):
the bench's million-weight network (53 k entries per proposal), config 2 and config 5 - outputs preallocated and touched, so that page
faults of fresh arrays are not in the figure.   python tools/time_predraw.py      (NPBNN_NO_FAST_PREDRAW=1: numpy's own routines)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npbnn_amd.predraw import PredrawPlan, load_host_library  # noqa: E402

print("inlined draws verified against numpy in this process:", bool(load_host_library().npbnn_host_fast_predraw()))
for name, shapes, f in (("20k x 4096, [256, 64] (1.07 M weights)", [(256, 4097), (64, 257), (10, 64)], 0.05),
                        ("default network on 1024 features", [(50, 1025), (5, 50), (10, 5)], 0.05),
                        ("config 2", [(32, 257), (8, 33), (10, 8)], 0.05)):
    w = [np.zeros(s) for s in shapes]
    n = [max(1, int(round(f * s[0] * s[1]))) for s in shapes]
    plan = PredrawPlan(w, n, [np.full(s, 0.05) for s in shapes], [1.0] * len(shapes))
    rs = np.random.default_rng(5)
    K = 20 if plan.M > 10000 else 200
    pool = {}

    def empty(shape, dtype):
        key = (tuple(shape) if not isinstance(shape, int) else (shape,), np.dtype(dtype))
        if key not in pool:
            pool[key] = np.zeros(shape, dtype)
        return pool[key]

    plan.run(rs, False, 0, 0, K, empty=empty, state=rs.bit_generator.state)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        plan.run(rs, False, 0, 0, K, empty=empty, state=rs.bit_generator.state)
        ts.append(time.perf_counter() - t0)
    el = min(ts)
    print("%-42s %6d entries per iteration: %8.1f us per iteration, %5.1f ns per entry" % (name, plan.M, 1e6 * el / K, 1e9 * el / K / plan.M), flush=True)
