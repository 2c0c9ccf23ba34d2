#!/usr/bin/env python3
"""Where the host time of a repeated dispatch goes (sampler._FastDispatch.run, stage by stage, perf_counter stamps): an instrumented
copy of the method is swapped in for the measurement.   python tools/dispatch_stages.py [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import workload  # noqa: E402
from npbnn_amd import sampler  # noqa: E402
from npbnn_amd.backend import pack_weights  # noqa: E402

now = time.perf_counter
acc = np.zeros(12)
names = ["matches (in run_steps)", "claim", "result()", "draw ahead", "pack weights", "cfg + addresses", "C call", "after C call -> absorb args",
         "absorb", "rest of run_steps"]
marks = []


def run(self, mcmc, bnn):
    t = [now()]
    k = self.k
    it = mcmc._current_iteration
    spec = mcmc._speculation
    if spec is not None and len(spec) == 5 and spec[4] is self and spec[0][0] == it:
        mcmc._speculation = None
        job = spec[1]
    else:
        job = mcmc._claim_draw(bnn, it, k)
    t.append(now())
    idx, delta, cnt, log_u = job.result()[:4]
    t.append(now())
    second = mcmc._speculation2          # (two batches are kept drawn ahead, as in the product's run)
    mcmc._speculation2 = None
    fresh = []
    if second is not None and second[4] is self and second[0][0] == it + k:
        ahead = second
    else:
        if second is not None:
            mcmc._speculation2 = second
            mcmc._cancel_second()
        ahead = self._draw_ahead(mcmc, it + k)
        fresh.append(ahead[1])
    further = self._draw_ahead(mcmc, it + 2 * k)
    fresh.append(further[1])
    t.append(now())
    w = pack_weights(bnn._w_layers)
    t.append(now())
    batch = self.batch
    cfg = batch.cfg
    cfg.cur_loglik, cfg.cur_logprior = mcmc._logLik, mcmc._logPrior
    cfg.temperature = mcmc._temperature
    from npbnn_amd.backend import _addr
    a = (_addr(w), _addr(idx), _addr(delta), _addr(cnt), _addr(log_u))
    for j in fresh:
        sampler._draw_pool().enqueue(j)
    mcmc._speculation, mcmc._speculation2 = ahead, further
    t.append(now())
    rc = batch.entry(batch.ctx._ctx, batch.cfg_ref, a[0], batch.mask_addr, batch.K, batch.M, a[1], a[2], a[3], a[4], batch.acc_addr, None, None,
                     batch.res_ref)
    t.append(now())
    assert rc == 0
    res = batch.res
    args = (int(res.n_accepted), res.n_passes, res.n_void_passes, res.schedule, res.loglik, res.logprior, None)
    t.append(now())
    mcmc._absorb(bnn, k, w, batch.acc, *args)
    self.layers = bnn._w_layers
    t.append(now())
    marks.append(t)
    return True


draw_marks = []


def _draw_ahead(self, mcmc, first_it):
    rs, randomize, mcmc_id, k, plan = mcmc._gen, mcmc._randomize_seed, mcmc._mcmc_id, self.k, self.plan
    empty, empty_group = self.empty, self.empty_group
    keep_state = not randomize
    t_q = now()

    def draw():
        t0 = now()
        if keep_state:
            job.saved = rs.bit_generator.state
        out = plan.run(rs, randomize, first_it, mcmc_id, k, empty=empty, empty_group=empty_group)
        t1 = now()
        np.log(out[3], out=out[3])
        draw_marks.append((t_q, t0, t1, now()))
        return out[0], out[1], out[2], out[3], None, None

    job = sampler._DrawJob(draw)
    return (first_it, k) + self.key_tail, job, keep_state, self.ws_src, self


calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
bnn, mcmc = workload(2).build()
mcmc.run_steps(bnn, 3000)
for _ in range(20):
    mcmc.run_steps(bnn, 100)
sampler._FastDispatch.run = run
sampler._FastDispatch._draw_ahead = _draw_ahead
outer = []
for _ in range(calls):
    t0 = now()
    mcmc.run_steps(bnn, 100)
    outer.append((t0, now()))
m = np.array(marks)
o = np.array(outer[-len(marks):])
stages = np.diff(m, axis=1) * 1e6
print("dispatches through the short way: %d of %d" % (len(marks), calls))
print("%-34s %8.1f us" % (names[0], np.mean((m[:, 0] - o[:, 0]) * 1e6)))
for n, col in zip(names[1:9], stages.T):
    print("%-34s %8.1f us (median %.1f)" % (n, col.mean(), np.median(col)))
print("%-34s %8.1f us" % (names[9], np.mean((o[:, 1] - m[:, -1]) * 1e6)))
print("%-34s %8.1f us" % ("whole call", np.mean((o[:, 1] - o[:, 0]) * 1e6)))
r = stages[:, 1]
print("result(): percentiles 50 / 90 / 99 / max = %.1f / %.1f / %.1f / %.1f us; waits over 100 us: %d of %d (their mean %.0f us)"
      % (np.percentile(r, 50), np.percentile(r, 90), np.percentile(r, 99), r.max(), (r > 100).sum(), len(r), r[r > 100].mean() if (r > 100).any() else 0))
d = np.array(draw_marks) * 1e6
if len(d):
    print("pre-draw on the helper thread: queued -> started %.1f us (max %.1f), the draw itself %.1f us (max %.1f), log %.1f us"
          % ((d[:, 1] - d[:, 0]).mean(), (d[:, 1] - d[:, 0]).max(), (d[:, 2] - d[:, 1]).mean(), (d[:, 2] - d[:, 1]).max(), (d[:, 3] - d[:, 2]).mean()))
mcmc._backend.close()
