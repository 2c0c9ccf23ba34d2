#!/usr/bin/env python3
"""The weight-streamed path (npbnn_amd/csrc/npbnn_wide.hip.h) shape by shape: time of a pass (the layers' products + the likelihood
kernel, HIP events around back-to-back passes), layer-0 arithmetic rate against the matrix cores' peak, error of the last layer's
values against float64, and the chain rate in dispatches of 100.
    python tools/time_wide.py [rows,features,h1-h2-...,classes ...]      NPBNN_L0=f32: exact float32 layer 0"""
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import npbnn_amd as bn  # noqa: E402
from oracle import npbnn_oracle as orc  # noqa: E402  (diagnostics tool: float64 values to compare against)

DEFAULT = ["20000,4096,256-64,10", "2000,2000,50-5,10", "5000,1000,200-50-10,7", "100000,1024,50-5,10", "100000,256,50-5,10"]
precision = os.environ.get("NPBNN_L0", "auto")
chain = os.environ.get("NPBNN_CHAIN", "1") != "0"
for spec in sys.argv[1:] or DEFAULT:
    n, f, hid, c = spec.split(",")
    n, f, c = int(n), int(f), int(c)
    hidden = [int(v) for v in hid.split("-")]
    rs = np.random.default_rng(0)
    x = rs.standard_normal((n, f)).astype(np.float32)
    y = rs.integers(0, c, n)
    dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=hidden, actFun=bn.ActFun(fun="tanh"), use_bias_node=2)
    mcmc = bn.MCMC(bnn)
    ctx = mcmc._backend.ctx
    ctx.set_l0_precision(precision)
    ms, cand = ctx.time_pass(bnn._w_layers, n_candidates=1, iters=50)
    flops0 = 2.0 * n * f * hidden[0]
    flops = flops0 + sum(2.0 * n * a * b for a, b in zip(hidden, hidden[1:] + [c]))
    xbytes = 4.0 * n * f
    mode = ctx.l0_mode()
    mult = 3 if mode == "f16-split" else 1
    peak = 2.5e15 if mode == "f16-split" else 157.3e12
    line = ("%s wide=%d %s: pass %.1f us | %.2f GF -> %.1f TF/s algorithmic, matrix cores %.1f TF/s = %.3f of peak | X %.0f MB -> %.2f TB/s"
            % (spec, ctx.is_wide(), mode, 1e3 * ms, flops / 1e9, flops / ms / 1e9, mult * flops0 / ms / 1e9, mult * flops0 / (ms * 1e-3) / peak,
               xbytes / 1e6, xbytes / (ms * 1e-3) / 1e12))
    print(line, flush=True)
    if n * f <= 100e6:
        z = ctx.predict(bnn._w_layers, apply_out_fn=False)
        z64 = orc.forward_logits(x.astype(np.float64), bnn._w_layers, orc.Act("tanh"))
        err = np.abs(z - z64) / np.maximum(1.0, np.abs(z64))
        print("   last layer's values against float64: max scaled error %.2e (mean %.2e)" % (err.max(), err.mean()), flush=True)
    if chain:
        mcmc.run_steps(bnn, 200)
        t0 = time.perf_counter()
        for _ in range(5):
            mcmc.run_steps(bnn, 100)
        el = time.perf_counter() - t0
        print("   chain: %.0f it/s in calls of 100 (%.1f us per iteration), acceptance %.3f, schedule %d"
              % (500 / el, 1e6 * el / 500, mcmc._device_accepted / max(1, mcmc._device_iterations), mcmc._device_schedule_used), flush=True)
    mcmc._backend.close()
