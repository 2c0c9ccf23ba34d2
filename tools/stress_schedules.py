#!/usr/bin/env python3
"""Long-run check of the flag-ordered chain schedules (3: two launch streams, 4: one persistent launch per batch round) against the
one-stream overlapped schedule (2) on config-2 shapes: the chains must be the same bit for bit - weights, log-likelihood,
acceptance record - at low and at high acceptance, and no device-side wait may time out (HipContext.sync_fallbacks).
   python tools/stress_schedules.py [iterations] [schedule ...]      also times 100-iteration calls per schedule"""
import contextlib
import io
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import npbnn_amd as bn  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
scheds = [int(v) for v in sys.argv[2:]] or [2, 4]
rs = np.random.default_rng(0)
n, f, c = 100_000, 256, 10
x = rs.standard_normal((n, f)).astype(np.float32)
proj = rs.standard_normal((f, c)) / np.sqrt(f)
labels = {"random labels": rs.integers(0, c, n), "learnable labels": np.argmax(x @ proj + 0.5 * rs.standard_normal((n, c)), axis=1)}
bad = 0
for name, y in labels.items():
    for uf in [float(v) for v in os.environ.get("NPBNN_STRESS_UF", "0.05,0.004").split(",")]:      # (proposal sizes: acceptance 0.5-3 % and ~25 %)
        out = []
        for sched in scheds:
            np.random.seed(1234)
            with contextlib.redirect_stdout(io.StringIO()):
                bnn = bn.npBNN(dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0)), n_nodes=[32, 8],
                               actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
            m = bn.MCMC(bnn, update_f=[uf] * 3)
            m.device_schedule = sched
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                t0 = time.perf_counter()
                m.run_steps(bnn, n_it)
                el = time.perf_counter() - t0
                t1 = time.perf_counter()
                for _ in range(30):
                    m.run_steps(bnn, 100)
                el100 = time.perf_counter() - t1
            out.append((bnn, m, el, el100, m._backend.ctx.sync_fallbacks, m._device_schedule_used))
            m._backend.close()
        ref = out[0]
        for (b, m, el, el100, fb, used), sched in zip(out, scheds):
            same = (m._logLik == ref[1]._logLik and m._logPrior == ref[1]._logPrior and m._last_accepted_mem == ref[1]._last_accepted_mem
                    and all(np.array_equal(u, v) for u, v in zip(b._w_layers, ref[0]._w_layers)) and m._device_accepted == ref[1]._device_accepted)
            print("%s, update_f %.3f, schedule %d (ran %d): %d iterations, %d accepted, %.0f it/s in one call, %.0f it/s in calls of 100; "
                  "time-outs %d; same chain as schedule %d: %s" % (name, uf, sched, used, n_it, m._device_accepted, n_it / el, 3000 / el100, fb,
                                                                  scheds[0], same), flush=True)
            bad += (not same) + (fb > 0)
sys.exit(1 if bad else 0)
