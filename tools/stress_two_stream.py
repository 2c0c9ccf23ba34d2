#!/usr/bin/env python3
"""Long-run check of the two-stream overlapped schedule against the one-stream one (config-2 shapes): the chains must be the
same bit for bit - weights, log-likelihood, acceptance record - at low and at high acceptance."""
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import npbnn_amd as bn  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rs = np.random.default_rng(0)
n, f, c = 100_000, 256, 10
x = rs.standard_normal((n, f)).astype(np.float32)
proj = rs.standard_normal((f, c)) / np.sqrt(f)
labels = {"random labels": rs.integers(0, c, n), "learnable labels": np.argmax(x @ proj + 0.5 * rs.standard_normal((n, c)), axis=1)}
for name, y in labels.items():
    for uf in (0.05, 0.004):
        out = []
        for sched in (2, 3):
            np.random.seed(1234)
            with contextlib.redirect_stdout(io.StringIO()):
                bnn = bn.npBNN(dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0)), n_nodes=[32, 8],
                               actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
            m = bn.MCMC(bnn, update_f=[uf] * 3)
            m.device_schedule = sched
            t0 = time.perf_counter()
            m.run_steps(bnn, n_it)
            el = time.perf_counter() - t0
            assert m._device_schedule_used == sched, m._device_schedule_used
            out.append((bnn, m, el))
            m._backend.close()
        (ba, ma, ea), (bb, mb, eb) = out
        same = (ma._logLik == mb._logLik and ma._logPrior == mb._logPrior and ma._last_accepted_mem == mb._last_accepted_mem
                and all(np.array_equal(u, v) for u, v in zip(ba._w_layers, bb._w_layers)) and ma._device_accepted == mb._device_accepted)
        print("%s, update_f %.3f: %d iterations, %d accepted; one stream %.0f it/s, two streams %.0f it/s; same chain: %s"
              % (name, uf, n_it, ma._device_accepted, n_it / ea, n_it / eb, same), flush=True)
        assert same
