import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, contextlib, io
import cases, npbnn_amd as bn
from npbnn_amd import predraw as pd
cfg = cases.TRACES["cfg1"]
dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
np.random.seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    bnn = bn.npBNN(dat, n_nodes=cfg["n_nodes"], actFun=bn.ActFun(fun=cfg["fun"]), use_bias_node=cfg["bias"])
mcmc = bn.MCMC(bnn, **cfg["mcmc"])
print("init", mcmc._logLik, mcmc._logPrior)
idx, delta, cnt, u, _ = pd.predraw(np.random.default_rng(1234), False, 0, 0, 5, bnn._w_layers, mcmc._update_n, mcmc._update_ws, mcmc._freq_layer_update)
w_new, acc, llp, lpp, res = mcmc._backend.run_chain(bnn._w_layers, idx=idx, delta=delta, cnt=cnt, log_u=np.log(u), prior_kind=1, prior_scale=bnn._prior_scale, w_bound=np.inf, temperature=1, lik_temp=1, cur_loglik=mcmc._logLik, cur_logprior=mcmc._logPrior)
print("acc", acc, "llp", llp, "lpp", lpp, res)
# host evaluation of the first proposal
flat = bn.pack_weights(bnn._w_layers)
z = flat.copy(); sel = idx[0,:cnt[0]] >= 0; z[idx[0,:cnt[0]][sel]] += delta[0,:cnt[0]][sel]
ws=[]; off=0
for w in bnn._w_layers:
    ws.append(z[off:off+w.size].reshape(w.shape)); off+=w.size
print("host eval of proposal 0:", mcmc._backend.evaluate(ws)["loglik"], bnn.calc_prior(w=ws))
