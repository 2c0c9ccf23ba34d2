#!/usr/bin/env python3
"""Launch only the evaluation kernels of a BASELINE.json config (for rocprofv3).
   python tools/profile_eval.py [--config 2|4|1] [--iters 20]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npbnn_amd import HipContext, _capi as capi  # noqa: E402
from npbnn_amd.proposals import init_weight_prm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--cand", type=int, default=1, help="proposals evaluated per launch (chain pass kernel when > 1)")
a = ap.parse_args()
rs = np.random.default_rng(0)
np.random.seed(1234)
ctx = HipContext(0)
if a.config == 2:
    n, f, c, hidden = a.rows or 100_000, 256, 10, [32, 8]
    x = rs.standard_normal((n, f), dtype=np.float32)
    ctx.set_data(x)
    ctx.set_labels(rs.integers(0, c, n))
    w = init_weight_prm(hidden, f, c, bias_node=2)
    ctx.set_arch_from_weights(w, f, capi.ACT_TANH, capi.OUT_SOFTMAX, capi.LIK_CATEGORICAL)
    alg = 4.0 * n * f + 4.0 * n
elif a.config == 4:
    n, f, k, hidden = a.rows or 1_000_000, 64, 2, [16, 4]
    x = rs.standard_normal((n, f), dtype=np.float32)
    ctx.set_data(x)
    ctx.set_targets(rs.standard_normal((n, k)))
    w = init_weight_prm(hidden, f, k, bias_node=2)
    ctx.set_arch_from_weights(w, f, capi.ACT_TANH, capi.OUT_IDENTITY, capi.LIK_GAUSS, n_targets=k)
    alg = 4.0 * n * f + 4.0 * n * k
else:
    n, f, c, hidden = a.rows or 2250, 128, 5, [5, 5]
    x = rs.standard_normal((n, f), dtype=np.float32)
    ctx.set_data(x)
    ctx.set_labels(rs.integers(0, c, n))
    w = init_weight_prm(hidden, f, c, bias_node=2)
    ctx.set_arch_from_weights(w, f, capi.ACT_TANH, capi.OUT_SOFTMAX, capi.LIK_CATEGORICAL)
    alg = 4.0 * n * f + 4.0 * n
if a.cand > 1:
    ms_k, cand = ctx.time_pass(w, n_candidates=a.cand, iters=a.iters)
    print("config %d: pass kernel with %d candidates %.2f us" % (a.config, cand, ms_k * 1e3))
    ctx.close()
    sys.exit(0)
ms_k, ms_t = ctx.time_eval(w, iters=a.iters)
print("config %d: eval kernel %.2f us (%.0f GB/s algorithmic, %.1f%% of 8 TB/s); kernel+finalize %.2f us"
      % (a.config, ms_k * 1e3, alg / ms_k / 1e6, 100 * alg / (ms_k * 1e-3) / 8e12, ms_t * 1e3))
ctx.close()
