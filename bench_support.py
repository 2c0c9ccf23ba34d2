"""Helpers for bench.py: initial weights and the timed Metropolis-Hastings loop."""
import numpy as np

from npbnn_amd.proposals import UpdateNormal, init_weight_prm


def make_initial_weights(hidden, n_features, n_classes, bias_node=2):
    np.random.seed(1234)
    return init_weight_prm(hidden, n_features, n_classes, init_std=0.1, bias_node=bias_node)


def normal_log_prior(weights, scale=1.0):
    lp = 0.0
    for w in weights:
        lp += -0.5 * np.sum((w / scale) ** 2) - w.size * (np.log(scale) + 0.9189385332046727)
    return lp


class StepRunner:
    """MCMC.mh_step's default path (UpdateNormal proposals, normal prior, temperature 1)
    with the forward pass + likelihood on the GPU, one C-ABI call per proposal."""
    mode = "host-loop: one npbnn_eval call per proposal"

    def __init__(self, ctx, weights, seed=1234, update_f=0.05, update_ws=0.075):
        self.ctx = ctx
        self.weights = [w + 0 for w in weights]
        self.rs = np.random.default_rng(seed)
        self.update_n = [max(1, int(round(w.size * update_f))) for w in weights]
        self.update_ws = [np.ones(w.shape) * update_ws for w in weights]
        self.loglik = ctx.eval(self.weights)["loglik"]
        self.logprior = normal_log_prior(self.weights)
        self.n_acc = 0
        self.n_it = 0

    def run(self, n_steps):
        nl = len(self.weights)
        for _ in range(n_steps):
            rr = self.rs.random(nl)
            rr[np.argmin(rr)] = 0
            prop = []
            for i in range(nl):
                if rr[i] < 1.0:
                    z, _, _ = UpdateNormal(self.weights[i], d=self.update_ws[i], n=self.update_n[i], Mb=np.inf, mb=-np.inf, rs=self.rs)
                    prop.append(z)
                else:
                    prop.append(self.weights[i] + 0)
            ll = self.ctx.eval(prop)["loglik"]
            lp = normal_log_prior(prop)
            if (ll + lp) - (self.loglik + self.logprior) >= np.log(self.rs.random()):
                self.weights, self.loglik, self.logprior = prop, ll, lp
                self.n_acc += 1
            self.n_it += 1

    def accept_rate(self):
        return self.n_acc / max(1, self.n_it)
