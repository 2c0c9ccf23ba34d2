import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import npbnn_amd as bn
n, f, c = 100000, 1024, 10
rs = np.random.default_rng(0)
x = rs.standard_normal((n, f)).astype(np.float32); y = rs.integers(0, c, n)
dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
np.random.seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    bnn = bn.npBNN(dat, n_nodes=[50, 5], actFun=bn.ActFun(fun="tanh"), use_bias_node=2)
mcmc = bn.MCMC(bnn)
ctx = mcmc._backend.ctx
for d in (1, 2):
    ms, used = ctx.time_pass(bnn._w_layers, n_candidates=d, iters=100)
    print("asked %d -> %d candidates: pass %.1f us" % (d, used, 1e3 * ms))
mcmc.run_steps(bnn, 300)
print("its/pass", mcmc._device_iterations / max(1, mcmc._device_passes), "cands", mcmc._device_candidates if hasattr(mcmc, "_device_candidates") else None)
