import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import synthetic_config2, HIDDEN
from bench_support import build_config2
x, y = synthetic_config2()
bnn, mcmc = build_config2(x.astype(np.float32), y, HIDDEN)
be = mcmc._backend
orig = be.run_chain
T = {"chain": 0.0, "n": 0}
def spy(weights, **kw):
    t0 = time.perf_counter(); out = orig(weights, **kw); T["chain"] += time.perf_counter() - t0; T["n"] += len(kw["cnt"]); return out
be.run_chain = spy
mcmc.run_steps(bnn, 300)
for K in (256, 1000, 1000):
    T["chain"] = 0; T["n"] = 0
    t0 = time.perf_counter(); mcmc.run_steps(bnn, K); el = time.perf_counter() - t0
    print("K=%d total %.2f ms (%.1f us/it), inside run_chain %.2f ms, outside %.2f ms" % (K, el*1e3, el/K*1e6, T["chain"]*1e3, (el-T["chain"])*1e3))
