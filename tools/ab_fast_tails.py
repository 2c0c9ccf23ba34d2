#!/usr/bin/env python3
"""A/B of the shape-specialised ("fast") builds of the evaluation kernel against the general build, in one process:
the log-likelihoods must agree bit for bit, the timings are interleaved rounds (median and min per arm).
   python tools/ab_fast_tails.py [--config 2|4|5] [--rounds 5] [--iters 200]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npbnn_amd import HipContext, _capi as capi  # noqa: E402
from npbnn_amd.proposals import init_weight_prm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--act", default="tanh")
a = ap.parse_args()
rs = np.random.default_rng(0)
np.random.seed(1234)
ctx = HipContext(0)
act = {"tanh": capi.ACT_TANH, "relu": capi.ACT_RELU, "swish": capi.ACT_SWISH, "leaky": capi.ACT_LEAKY}[a.act]
if a.config == 2:
    n, f, c, hidden = 100_000, 256, 10, [32, 8]
    x = rs.standard_normal((n, f), dtype=np.float32)
    ctx.set_data(x)
    ctx.set_labels(rs.integers(0, c, n))
    w = init_weight_prm(hidden, f, c, bias_node=2)
    ctx.set_arch_from_weights(w, f, act, capi.OUT_SOFTMAX, capi.LIK_CATEGORICAL)
    alg = 4.0 * n * f + 4.0 * n
elif a.config == 4:
    n, f, k, hidden = 1_000_000, 64, 2, [16, 4]
    x = rs.standard_normal((n, f), dtype=np.float32)
    ctx.set_data(x)
    ctx.set_targets(rs.standard_normal((n, k)))
    w = init_weight_prm(hidden, f, k, bias_node=2)
    ctx.set_arch_from_weights(w, f, act, capi.OUT_IDENTITY, capi.LIK_GAUSS, n_targets=k)
    alg = 4.0 * n * f + 4.0 * n * k
else:
    n, f, k, hidden = 50_000, 512, 1, [32, 8]
    x = rs.standard_normal((n, f), dtype=np.float32)
    ctx.set_data(x)
    ctx.set_targets(rs.standard_normal((n, k)))
    w = init_weight_prm(hidden, f, k, bias_node=-1)
    ctx.set_arch_from_weights(w, f, act, capi.OUT_IDENTITY, capi.LIK_GAUSS, n_targets=k)
    alg = 4.0 * n * f + 4.0 * n * k

print("fast builds apply to this shape:", bool(ctx.info(capi.INFO_FAST_TAILS)))
res = {}
for fast in (1, 0):
    ctx.set_fast_tails(fast)
    r = ctx.eval(w)
    res[fast] = r
    print("fast=%d loglik %.17g sigma %s" % (fast, r["loglik"], r["sigma"]))
same = res[0]["loglik"] == res[1]["loglik"] and np.array_equal(res[0]["sum_r2"], res[1]["sum_r2"])
print("BIT-IDENTICAL" if same else "DIFFERENT", "(fast vs general)")

times = {(fast, d): [] for fast in (0, 1) for d in (1, 3)}
for rnd in range(a.rounds):
    for fast in (0, 1):
        ctx.set_fast_tails(fast)
        ms1, _ = ctx.time_eval(w, iters=a.iters)
        ms3, cand = ctx.time_pass(w, n_candidates=3, iters=a.iters)
        times[(fast, 1)].append(ms1 * 1e3)
        times[(fast, cand if cand == 3 else 3)].append(ms3 * 1e3)
for (fast, d), v in sorted(times.items()):
    v = np.array(v)
    phys = alg / (np.median(v) * 1e-6) / 8e12
    print("fast=%d D=%d: median %.2f us  min %.2f us  (one read of X / median = %.3f of 8 TB/s)" % (fast, d, np.median(v), v.min(), phys))
ctx.close()
sys.exit(0 if same else 1)
