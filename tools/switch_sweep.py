#!/usr/bin/env python3
"""Either side of the switch between the LDS-resident and the weight-streamed path: the reference's default network (npBNN(n_nodes=[50, 5]),
np_bnn/BNN_env.py:20-23) on 100k rows and a growing number of features - which path the library takes, waves per workgroup, candidates per
pass, the pass kernel's time and the chain's rate in dispatches of 100.
    python tools/switch_sweep.py [features ...]        NPBNN_FORCE_WIDE=1: the weight-streamed path at every size (the crossover)"""
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import npbnn_amd as bn  # noqa: E402
from npbnn_amd import _capi as capi  # noqa: E402

N = int(os.environ.get("NPBNN_SWEEP_ROWS", "100000"))
hidden = [int(v) for v in os.environ.get("NPBNN_SWEEP_HIDDEN", "50-5").split("-")]
print("rows,features,hidden,path,layer0,waves_per_workgroup,candidates_per_pass,pass_us,single_candidate_pass_us,TBps_of_X_per_pass,"
      "chain_it_per_s_calls_of_100,iterations_per_pass")
for f in [int(v) for v in sys.argv[1:]] or [256, 384, 512, 576, 640, 672, 704, 768, 1024, 1536, 2048]:
    rs = np.random.default_rng(0)
    x = rs.standard_normal((N, f)).astype(np.float32)
    y = rs.integers(0, 10, N)
    dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=hidden)
    mcmc = bn.MCMC(bnn)
    ctx = mcmc._backend.ctx
    mcmc.run_steps(bnn, 300)
    ms, cand = ctx.time_pass(bnn._w_layers, n_candidates=0, iters=100)
    ms1, _ = ctx.time_pass(bnn._w_layers, n_candidates=1, iters=100)
    p0, i0 = mcmc._device_passes, mcmc._device_iterations
    t0 = time.perf_counter()
    for _ in range(10):
        mcmc.run_steps(bnn, 100)
    el = time.perf_counter() - t0
    print("%d,%d,%s,%s,%s,%d,%d,%.1f,%.1f,%.2f,%.0f,%.2f" % (N, f, "-".join(map(str, hidden)), "weight-streamed" if ctx.is_wide() else "resident", ctx.l0_mode(),
                                                         ctx.info(capi.INFO_WAVES_PER_BLOCK), cand, 1e3 * ms, 1e3 * ms1, 4.0 * N * f / (ms * 1e-3) / 1e12,
                                                         1000 / el, (mcmc._device_iterations - i0) / max(1, mcmc._device_passes - p0)), flush=True)
    mcmc._backend.close()
