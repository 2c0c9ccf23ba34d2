#!/usr/bin/env python3
"""Where the host time of short run_steps calls goes (cProfile over calls of 100 iterations on config 2).
   python tools/profile_short_calls.py [calls]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import workload  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bnn, mcmc = workload(int(os.environ.get('NPBNN_CONFIG', '2'))).build()
mcmc.run_steps(bnn, 3000)
for _ in range(5):
    mcmc.run_steps(bnn, 100)
pr = cProfile.Profile()
pr.enable()
for _ in range(calls):
    mcmc.run_steps(bnn, 100)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
mcmc._backend.close()
