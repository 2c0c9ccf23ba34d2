import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import npbnn_amd as bn
f = int(sys.argv[1]); mode = sys.argv[2]; sched = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, c = 100000, 10
rs = np.random.default_rng(0)
x = rs.standard_normal((n, f)).astype(np.float32); y = rs.integers(0, c, n)
dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
np.random.seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    bnn = bn.npBNN(dat, n_nodes=[50, 5], actFun=bn.ActFun(fun="tanh"), use_bias_node=2)
mcmc = bn.MCMC(bnn)
mcmc.device_schedule = sched
ctx = mcmc._backend.ctx
print(mode, "wide", ctx.is_wide(), "init logLik %.6f" % mcmc._logLik)
if mode == "mh":
    for _ in range(60): mcmc.mh_step(bnn)
else:
    mcmc.run_steps(bnn, 60)
print(" accepted", sum(mcmc._last_accepted_mem[-60:]) if hasattr(mcmc, "_last_accepted_mem") else None, "logLik %.6f" % mcmc._logLik, "sched", getattr(mcmc, "_device_schedule_used", None),
      "cands", ctx.info(7), "waves", ctx.info(2))
