#!/usr/bin/env python3
"""Throughput of the device-resident chain as a function of the acceptance rate (config-2 shapes; the proposal size
update_f sets the acceptance rate).  Speculative passes pay off most when almost everything is rejected."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import contextlib, io  # noqa: E402
import npbnn_amd as bn  # noqa: E402

rs = np.random.default_rng(0)
n, f, c = 100_000, 256, 10
x = rs.standard_normal((n, f)).astype(np.float32)
proj = rs.standard_normal((f, c)) / np.sqrt(f)
y = np.argmax(x @ proj + 0.5 * rs.standard_normal((n, c)), axis=1)          # learnable labels
schedules = [int(v) for v in sys.argv[1:]] or [0]          # 0 auto, 1 serial, 2 overlapped
ufs = [float(v) for v in os.environ.get("NPBNN_SWEEP_UPDATE_F", "0.05,0.01,0.002,0.0005").split(",")]
for uf, sched in [(u, sc) for u in ufs for sc in schedules]:
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0)), n_nodes=[32, 8],
                       actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
    mcmc = bn.MCMC(bnn, update_f=[uf] * 3)
    mcmc.device_schedule = sched
    mcmc.run_steps(bnn, 1000)
    a0 = mcmc._device_passes, mcmc._device_void_passes
    acc = []
    t = time.perf_counter()
    for _ in range(4):
        mcmc.run_steps(bnn, 1000)
        acc.append(mcmc._acceptance_rate)
    el = time.perf_counter() - t
    p, v = mcmc._device_passes - a0[0], mcmc._device_void_passes - a0[1]
    print("schedule %d, update_f %.4f (update_n %s): acceptance %.2f, %.0f it/s, %.2f iterations per decided pass, %.0f %% of the launches void"
          % (sched, uf, [int(v) for v in mcmc._update_n], float(np.mean(acc)), 4000 / el, 4000 / p, 100.0 * v / (p + v)))
