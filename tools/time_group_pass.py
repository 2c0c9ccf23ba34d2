#!/usr/bin/env python3
"""Several chains of one model on ONE GPU (config 2 shapes): aggregate iterations/s when the chains share their passes over the data
(group passes, npbnn_chains_run_batched: one proposal per chain per streaming read of X) against the exchange run (every chain
its own speculative passes on its own stream) and against one chain alone, at two acceptance rates (learnable labels; the proposal
size sets the rate).  SURVEY 8(f) item 2.      python tools/time_group_pass.py [update_f ...]"""
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import npbnn_amd as bn  # noqa: E402
from npbnn_amd import exchange as ex  # noqa: E402

rs = np.random.default_rng(0)
n, f, c = 100_000, 256, 10
x = rs.standard_normal((n, f)).astype(np.float32)
proj = rs.standard_normal((f, c)) / np.sqrt(f)
y = np.argmax(x @ proj + 0.5 * rs.standard_normal((n, c)), axis=1)          # learnable labels
dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
seg = 100


def chains_of(n_chains, uf):
    out = []
    for i in range(n_chains):
        np.random.seed(1234 + i)
        with contextlib.redirect_stdout(io.StringIO()):
            bnn = bn.npBNN(dat, n_nodes=[32, 8], actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
        out.append((bnn, bn.MCMC(bnn, update_f=[uf] * 3, mcmc_id=i, randomize_seed=True)))
    return out


class NoSwap:
    def get(self, first, n=1):
        return np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), np.zeros(n)

    def release(self, upto):
        pass


for uf in [float(v) for v in sys.argv[1:]] or [0.05, 0.002]:
    for n_chains in (1, 3, 4):
        for mode in (("alone",) if n_chains == 1 else ("group passes", "exchange run")):
            chains = chains_of(n_chains, uf)
            for bnn, m in chains:
                m.run_steps(bnn, 1500)
            ids = list(range(n_chains))
            rounds = 20

            def advance():
                if mode == "exchange run":
                    ex.advance_intervals(chains, ids, n_chains, rounds, seg, NoSwap(), 0, batch=20, device=True)
                else:
                    for _ in range(rounds):
                        ex.run_steps_batched(chains, seg)
            advance()
            t0 = time.perf_counter()
            advance()
            el = time.perf_counter() - t0
            acc = np.mean([m._acceptance_rate for _, m in chains])
            print("update_f %.4f, %d chain(s), %-12s: %7.0f it/s aggregate (%6.0f per chain), acceptance %.2f"
                  % (uf, n_chains, mode, n_chains * rounds * seg / el, rounds * seg / el, acc), flush=True)
            for bnn, m in chains:
                m._backend.close()
