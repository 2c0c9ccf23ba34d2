"""Helpers for bench.py: the config-2 model and the timed Metropolis-Hastings loop."""
import contextlib
import io

import numpy as np

import npbnn_amd as bn


def build_config2(x, y, hidden, mcmc_id=0, temperature=1.0, randomize_seed=False):
    """BASELINE.json config 2 through the reference's own call sequence: np.random.seed(1234); npBNN(n_nodes=[32,8],
    tanh, bias nodes in input+hidden layers, N(0,1) prior); MCMC defaults (update_f 0.05 -> update_n [411,13,4])."""
    dat = dict(data=x, labels=y, test_data=np.zeros((0, x.shape[1])), test_labels=np.zeros(0))
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=hidden, actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
    mcmc = bn.MCMC(bnn, temperature=temperature, mcmc_id=mcmc_id, randomize_seed=randomize_seed)
    return bnn, mcmc
