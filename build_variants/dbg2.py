import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, contextlib, io
import cases, npbnn_amd as bn
cfg = cases.TRACES["cfg1"]
dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
def build():
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=cfg["n_nodes"], actFun=bn.ActFun(fun=cfg["fun"]), use_bias_node=cfg["bias"])
    return bnn, bn.MCMC(bnn, **cfg["mcmc"])
ba, ma = build(); bb, mb = build()
orig = mb._backend.run_chain
def spy(weights, **kw):
    out = orig(weights, **kw)
    print("run_chain K=%d acc=%s llp[:4]=%s cur=%.4f" % (len(kw["cnt"]), out[1][:8], out[2][:4], kw["cur_loglik"]))
    return out
mb._backend.run_chain = spy
for _ in range(8): ma.mh_step(ba)
print("host mem", ma._last_accepted_mem, ma._logLik)
mb.run_steps(bb, 8)
print("dev mem", mb._last_accepted_mem, mb._logLik)
