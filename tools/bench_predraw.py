#!/usr/bin/env python3
"""Host-side cost of pre-drawing the proposals of BASELINE config 2 (us per MCMC iteration), for sizing the
device-resident chain: the device consumes one iteration every ~18 us."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npbnn_amd import predraw as pd  # noqa: E402

shapes = [np.empty((32, 257)), np.empty((8, 33)), np.empty((10, 8))]
ws = [np.ones(s.shape) * 0.075 for s in shapes]
rs = np.random.default_rng(1)
print("host cpus:", os.cpu_count())
for randomize in (False, True):
    for K in (128, 2048):
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            pd.predraw(rs, randomize, 0, 0, K, shapes, [411, 13, 4], ws, [1, 1, 1])
            best = min(best, time.perf_counter() - t)
        print("randomize_seed=%s K=%d: %.2f us/iteration" % (randomize, K, best * 1e6 / K))
