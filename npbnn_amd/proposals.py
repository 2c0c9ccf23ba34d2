"""Proposal kernels and weight initialisation (host side).

These stay on numpy's ``Generator`` so that a chain consumes exactly the same
random stream as the reference for the same seed (SURVEY.md section 8a, rows
A15/A16).  Behavioural reference: np_bnn/BNN_mcmc.py:9-150.
"""
import random

import numpy as np
import scipy.stats

small_number = 1e-10


def _fresh_rng():
    # the reference seeds a throw-away generator from python's `random` when none is given
    return np.random.default_rng(random.randint(1000, 9999))


def init_weight_prm(n_nodes, n_features, size_output, init_std=0.1, bias_node=0):
    """Initial weights, one (out x in[+1]) matrix per layer with the bias in column 0;
    drawn from numpy's global RNG (reference: BNN_mcmc.py:9-25).
    bias_node: 0 none, 1 input layer, 2 input+hidden, 3 all layers, -1 last layer only."""
    first = 1 if bias_node >= 1 else 0
    hidden = 1 if bias_node >= 2 else 0
    last = 1 if bias_node in (3, -1) else 0
    sizes = [n_features] + list(n_nodes)
    layers = []
    for li in range(len(n_nodes)):
        extra = first if li == 0 else hidden
        layers.append(np.random.normal(0, init_std, (sizes[li + 1], sizes[li] + extra)))
    layers.append(np.random.normal(0, init_std, (size_output, n_nodes[-1] + last)))
    return layers


def _bounce(z, upper, lower):
    over = z > upper
    z[over] = upper - (z[over] - upper)
    under = z < lower
    z[under] = lower + (lower - z[under])
    return z


def _pick(shape, n, rs):
    return rs.integers(0, shape[0], n), rs.integers(0, shape[1], n)


def UpdateNormal(i, d=0.01, n=1, Mb=100, mb=-100, rs=0):
    """Random-walk proposal on n randomly chosen entries (with replacement; for a repeated
    entry the last draw wins), reflected at the bounds (reference: BNN_mcmc.py:57-69).
    RNG draw order: row indices, column indices, normal deviates."""
    cur = np.array(i)
    if not rs:
        rs = _fresh_rng()
    rows, cols = _pick(cur.shape, n, rs)
    new = np.zeros(cur.shape) + cur
    new[rows, cols] = new[rows, cols] + rs.normal(0, d[rows, cols], n)
    return _bounce(new, Mb, mb), (rows, cols), 0


def UpdateNormal1D(i, d=0.01, n=1, Mb=100, mb=-100, rs=0):
    """Random-walk proposal on a vector (reference: BNN_mcmc.py:44-55)."""
    cur = np.array(i)
    if not rs:
        rs = _fresh_rng()
    idx = rs.integers(0, len(cur), n)
    new = np.zeros(cur.shape) + cur
    new[idx] = new[idx] + rs.normal(0, d, n)
    return _bounce(new, Mb, mb), idx, 0


def UpdateFixedNormal(i, d=1, n=1, Mb=100, mb=-100, rs=0):
    """Independence proposal N(0, d) on n entries with its Hastings ratio
    (reference: BNN_mcmc.py:27-42)."""
    if not rs:
        rs = _fresh_rng()
    rows, cols = _pick(i.shape, n, rs)
    old = i[rows, cols]
    drawn = rs.normal(0, d[rows, cols], n)
    hastings = np.sum(scipy.stats.norm.logpdf(old, 0, d[rows, cols]) - scipy.stats.norm.logpdf(drawn, 0, d[rows, cols]))
    new = np.zeros(i.shape) + i
    new[rows, cols] = drawn
    return _bounce(new, Mb, mb), (rows, cols), hastings


def UpdateNormalNormalized(i, d=0.01, n=1, Mb=100, mb=-100, rs=0):
    """Random-walk proposal followed by renormalisation to unit sum (reference: BNN_mcmc.py:71-82)."""
    cur = np.array(i)
    if not rs:
        rs = _fresh_rng()
    rows, cols = _pick(cur.shape, n, rs)
    new = np.zeros(cur.shape) + cur
    new[rows, cols] = new[rows, cols] + rs.normal(0, d[rows, cols], n)
    return new / np.sum(new), (rows, cols), 0


def UpdateUniform(i, d=0.1, n=1, Mb=100, mb=-100):
    """Uniform sliding window on n entries, numpy global RNG (reference: BNN_mcmc.py:86-95)."""
    cur = np.array(i)
    rows = np.random.randint(0, cur.shape[0], n)
    cols = np.random.randint(0, cur.shape[1], n)
    new = np.zeros(cur.shape) + cur
    new[rows, cols] = new[rows, cols] + np.random.uniform(-d[rows, cols], d[rows, cols], n)
    return _bounce(new, Mb, mb), (rows, cols), 0


def UpdateBinomial(ind, update_f, shape_out):
    """Flip indicators with a random probability <= update_f (reference: BNN_mcmc.py:98-99)."""
    return np.abs(ind - np.random.binomial(1, np.random.random() * update_f, shape_out))


def multiplier_proposal_vector(q, d=1.05, f=1, rs=0):
    """Multiplier proposal on a random subset of a positive vector (reference: BNN_mcmc.py:101-113)."""
    if not rs:
        rs = _fresh_rng()
    shape = q.shape
    chosen = rs.binomial(1, f, shape)
    u = rs.random(shape)
    m = np.exp(2 * np.log(d) * (u - .5))
    m[chosen == 0] = 1.
    return q * m, 0, np.sum(np.log(m))


def multiplier_proposal(i, d=1.05):
    """Scalar multiplier proposal, numpy global RNG (reference: BNN_mcmc.py:116-123)."""
    m = np.exp(2 * np.log(d) * (np.random.random() - .5))
    return (i + 0) * m, 0, np.log(m)


# ---- Gibbs updates of the prior scale (reference: BNN_mcmc.py:126-150) ----------------------
def GibbsSampleNormStdGammaVector(x, a=2, b=0.1, mu=0):
    shape_post = a + len(x) / 2.
    rate_post = b + np.sum((x - mu) ** 2) / 2.
    return 1 / np.sqrt(np.random.gamma(shape_post, scale=1. / rate_post))


def GibbsSampleNormStdGamma2D(x, a=1, b=0.1, mu=0):
    shape_post = a + (x.shape[0]) / 2.
    rate_post = b + np.sum((x - mu) ** 2, axis=0) / 2.
    return 1 / np.sqrt(np.random.gamma(shape_post, scale=1. / rate_post))


def GibbsSampleNormStdGammaONE(x, a=1.5, b=0.1, mu=0):
    shape_post = a + 1 / 2.
    rate_post = b + ((x - mu) ** 2) / 2.
    return 1 / np.sqrt(np.random.gamma(shape_post, scale=1. / rate_post))


def GibbsSampleGammaRateExp(sd, a, alpha_0=1., beta_0=1.):
    tau = 1. / (sd ** 2)
    return np.random.gamma(alpha_0 + len(tau) * a, scale=1. / (beta_0 + np.sum(tau)))
