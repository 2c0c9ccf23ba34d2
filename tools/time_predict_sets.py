#!/usr/bin/env python3
"""Posterior prediction throughput at config 2: S stored weight sets on the resident 100k x 256 matrix,
npbnn_predict_sets (three sets per pass over X) against S single npbnn_predict calls."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npbnn_amd import HipContext, _capi as capi  # noqa: E402
from npbnn_amd.proposals import init_weight_prm  # noqa: E402

rs = np.random.default_rng(0)
np.random.seed(1)
n, f, c, S = 100_000, 256, 10, 30
ctx = HipContext(0)
ctx.set_data(rs.standard_normal((n, f), dtype=np.float32))
sets = [init_weight_prm([32, 8], f, c, bias_node=2) for _ in range(S)]
ctx.set_arch_from_weights(sets[0], f, capi.ACT_TANH, capi.OUT_SOFTMAX, capi.LIK_NONE)
ctx.predict(sets[0])
ctx.predict_sets(sets[:3])
t = time.perf_counter()
y1 = np.stack([ctx.predict(w) for w in sets])
t1 = time.perf_counter() - t
t = time.perf_counter()
y3 = ctx.predict_sets(sets)
t3 = time.perf_counter() - t
assert np.array_equal(y1, y3)
print("%d sets x %d rows x %d classes: single predicts %.1f ms/set, predict_sets %.1f ms/set (output copy and float64 "
      "conversion of %.0f MB per set included)" % (S, n, c, t1 / S * 1e3, t3 / S * 1e3, n * c * 8 / 1e6))
ctx.close()
