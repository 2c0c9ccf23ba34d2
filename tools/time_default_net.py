#!/usr/bin/env python3
"""The reference's DEFAULT network (npBNN(n_nodes=[50, 5]), ActFun() = ReLU, use_bias_node=1; np_bnn/BNN_env.py:20-23) on config-2
data (100k x 256, 10 classes) and on a narrow table (100k x 64): layer 0 of 50 nodes = four 16-unit output tiles (MT0 = 4).
Pass kernel per candidate count, chain rate in dispatches of 100 and in one call.   python tools/time_default_net.py [features ...]"""
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import npbnn_amd as bn  # noqa: E402

nodes = [int(v) for v in os.environ.get("NPBNN_NODES", "50,5").split(",")]
for f in [int(v) for v in sys.argv[1:]] or [256, 64]:
    rs = np.random.default_rng(0)
    n, c = 100_000, 10
    x = rs.standard_normal((n, f)).astype(np.float32)
    y = rs.integers(0, c, n)
    dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=nodes)
    mcmc = bn.MCMC(bnn)
    ctx = mcmc._backend.ctx
    line = ["%d features, hidden %s (%d parameters):" % (f, nodes, bnn._n_params)]
    for d in (1, 2, 3):
        ms, cand = ctx.time_pass(bnn._w_layers, n_candidates=d, iters=100)
        line.append("D=%d -> %d cand %.1f us" % (d, cand, 1e3 * ms))
    ms1, _ = ctx.time_eval(bnn._w_layers, iters=100)
    line.append("plain eval %.1f us" % (1e3 * ms1))
    print("  ".join(line), flush=True)
    for d in (0, 2, 3):
        np.random.seed(1234)
        with contextlib.redirect_stdout(io.StringIO()):
            b2 = bn.npBNN(dat, n_nodes=nodes)
        m2 = bn.MCMC(b2)
        m2.n_candidates = d
        m2.run_steps(b2, 1000)
        t0 = time.perf_counter()
        for _ in range(20):
            m2.run_steps(b2, 100)
        el = time.perf_counter() - t0
        t0 = time.perf_counter()
        m2.run_steps(b2, 4000)
        el2 = time.perf_counter() - t0
        print("   chain with n_candidates=%d: %.0f it/s in calls of 100, %.0f in one call; %.2f iterations per pass, acceptance %.3f, schedule %d"
              % (d, 2000 / el, 4000 / el2, m2._device_iterations / max(1, m2._device_passes), m2._device_accepted / max(1, m2._device_iterations),
                 m2._device_schedule_used), flush=True)
        m2._backend.close()
    mcmc._backend.close()
