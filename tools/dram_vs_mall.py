#!/usr/bin/env python3
"""Where do the pass kernel's bytes come from?  Config 2's X (102 MB) fits the 256-MiB Infinity Cache, so a launch that follows a launch
may be served from it rather than from HBM, and FETCH_SIZE counts both alike (MI355X_MICROARCH.md, HBM / Infinity Cache).  No counter
this profiler exposes separates the two, so the question is put to the clock: the same columns and network at row counts below and
above what the cache holds - if the time per byte does not change across the boundary, the kernel is not bound by where its bytes
come from.   python tools/dram_vs_mall.py > profiles/rNN_dram_vs_mall.csv"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib  # noqa: E402
import io  # noqa: E402

import npbnn_amd as bn  # noqa: E402

print("columns,network,rows,MB_of_X,fits_infinity_cache,candidates,pass_us,us_per_100MB,TB_per_s")
rs = np.random.default_rng(0)
for f, hidden, k, rows in ((256, [32, 8], 0, (50_000, 100_000, 200_000, 400_000, 800_000)), (64, [16, 4], 2, (400_000, 1_000_000, 2_000_000, 4_000_000))):
    for n in rows:
        x = rs.standard_normal((n, f)).astype(np.float32)
        if k:
            dat = dict(data=x, labels=rs.standard_normal((n, k)), test_data=np.zeros((0, f)), test_labels=np.zeros((0, k)))
            kw = dict(estimation_mode="regression", empirical_error=True)
        else:
            dat = dict(data=x, labels=rs.integers(0, 10, n), test_data=np.zeros((0, f)), test_labels=np.zeros(0))
            kw = {}
        np.random.seed(1234)
        with contextlib.redirect_stdout(io.StringIO()):
            bnn = bn.npBNN(dat, n_nodes=hidden, actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1, **kw)
        mcmc = bn.MCMC(bnn)
        ctx = mcmc._backend.ctx
        mb = n * f * 4 / 1e6
        for cand in (3, 1):
            ctx.time_pass(bnn._w_layers, n_candidates=cand, iters=100)
            ms, used = ctx.time_pass(bnn._w_layers, n_candidates=cand, iters=300)
            print("%d,%s,%d,%.1f,%s,%d,%.2f,%.2f,%.3f" % (f, "-".join(map(str, hidden)), n, mb, "yes" if mb < 256 * 1.048576 * 0.9 else "no", used,
                                                       1e3 * ms, 1e3 * ms * 100 / mb, mb / ms / 1e3 / 1e3), flush=True)
        mcmc._backend.close()
        del x, dat, bnn, mcmc
