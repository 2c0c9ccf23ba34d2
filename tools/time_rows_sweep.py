#!/usr/bin/env python3
"""Pass-kernel time against the number of rows (config-2 columns and network): where the per-tile cost changes tells which
level of the memory system bounds the tile loop (X of <= ~3 MB per XCD stays in its L2 between launches).
   python tools/time_rows_sweep.py [rows ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_support import build_config2  # noqa: E402

rows = [int(a) for a in sys.argv[1:]] or [4096, 12288, 24576, 49152, 100000, 200000]
rs = np.random.default_rng(0)
for n in rows:
    x = rs.standard_normal((n, 256)).astype(np.float32)
    y = rs.integers(0, 10, n)
    bnn, mcmc = build_config2(x, y, [32, 8])
    ctx = mcmc._backend.ctx
    out = []
    for cand in (3, 1):
        ms, used = ctx.time_pass(bnn._w_layers, n_candidates=cand, iters=300)
        out.append("D=%d %.2f us" % (used, ms * 1e3))
    tiles = (n + 15) // 16
    print("%7d rows (%5.1f MB, %5.2f tiles per CU): %s" % (n, n * 1024 / 1e6, tiles / 255.0, ", ".join(out)), flush=True)
    mcmc._backend.close()
