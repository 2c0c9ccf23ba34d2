#!/usr/bin/env python3
"""cProfile of the host side of short run_steps segments (config 2, one GPU)."""
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import build_config2  # noqa: E402

rs = np.random.default_rng(0)
x = rs.standard_normal((100_000, 256)).astype(np.float32)
y = rs.integers(0, 10, 100_000)
bnn, mcmc = build_config2(x, y, [32, 8], randomize_seed=True, mcmc_id=1)
mcmc.run_steps(bnn, 200)
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    mcmc.run_steps(bnn, 100)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
