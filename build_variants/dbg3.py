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
oe = ma._backend.evaluate
host_ll = []
def spy_e(weights, **kw):
    r = oe(weights, **kw); host_ll.append(r["loglik"]); return r
ma._backend.evaluate = spy_e
for _ in range(6): ma.mh_step(ba)
print("host proposals", np.round(host_ll, 4), "update_n", ma._update_n, "freq", ma._freq_layer_update)
orig = mb._backend.run_chain
def spy(weights, **kw):
    out = orig(weights, **kw)
    print("dev  proposals", np.round(out[2][:6], 4), "cnt", kw["cnt"][:6], "M", kw["idx"].shape, "update_n", mb._update_n)
    return out
mb._backend.run_chain = spy
mb.run_steps(bb, 6)
print("---- second pair")
ba, ma = build(); bb, mb = build()
import npbnn_amd.sampler as S
orig2 = mb._backend.run_chain
def spy2(weights, **kw):
    out = orig2(weights, **kw)
    print("dev lpp", out[3][:4], "log_u", kw["log_u"][:4], "cur", kw["cur_loglik"], kw["cur_logprior"], "T", kw["temperature"], "acc", out[1][:6])
    return out
mb._backend.run_chain = spy2
oc = ba.calc_prior
def spy_p(w=0, ind=[]):
    r = oc(w=w, ind=ind)
    print("host prior", r)
    return r
ba.calc_prior = spy_p
rs0 = ma._rs
for _ in range(2):
    ma.mh_step(ba)
    print("host accepted", ma._last_accepted, "logPost", ma._logPost)
mb.run_steps(bb, 4)
