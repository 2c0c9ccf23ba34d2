"""Host side of a chain dispatch (round 3: the lean helper thread, the cached pre-draw plan, packed-weight views recognised by
identity, the settings struct refilled only when a setting changed) - on CPU, no device."""
import pickle

import numpy as np
import pytest

from npbnn_amd import backend as bk
from npbnn_amd import predraw as pd
from npbnn_amd import sampler


def test_draw_job_hands_over_results_and_errors():
    pool = sampler._draw_pool()
    assert pool.submit(lambda: 41 + 1).result() == 42
    assert pool.submit(lambda a, b: a * b, 6, 7).result() == 42
    job = pool.submit(lambda: 1 / 0)
    with pytest.raises(ZeroDivisionError):
        job.result()
    with pytest.raises(ZeroDivisionError):      # (asking twice is asking the same thing)
        job.result()
    done = [pool.submit(lambda i=i: i * i) for i in range(50)]      # order of service = order of submission (one thread)
    assert [j.result() for j in done] == [i * i for i in range(50)]


def test_a_plan_draws_what_the_one_off_call_draws():
    shapes = [np.empty((6, 9)), np.empty((4, 7)), np.empty((3, 5))]
    ws = [np.full(s.shape, 0.05 * (i + 1)) for i, s in enumerate(shapes)]
    plan = pd.PredrawPlan(shapes, [5, 3, 2], ws, [1.0, 0.5, 1.0])
    for kw in (dict(), dict(sigma_k=2), dict(n_slopes=2), dict(sigma_k=1, n_slopes=3)):
        a = plan.run(np.random.default_rng(7), False, 0, 0, 40, **kw)
        b = pd.predraw(np.random.default_rng(7), False, 0, 0, 40, shapes, [5, 3, 2], ws, [1.0, 0.5, 1.0], **kw)
        assert len(a) == len(b)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    # the same plan again: nothing of the first call is left in it
    a = plan.run(np.random.default_rng(9), True, 100, 3, 25)
    b = plan.run(np.random.default_rng(1), True, 100, 3, 25)       # (randomize_seed: the Generator handed in is not used)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_packed_views_are_recognised_by_identity_and_only_then():
    base = np.arange(20.0)
    layers = [base[0:12].reshape(3, 4), base[12:20].reshape(2, 4)]
    bk.note_packed_views(layers, base)
    out = bk.pack_weights(layers)
    np.testing.assert_array_equal(out, base)
    assert out is not base
    layers[0][0, 0] = -7.0                       # an in-place edit is an edit of the packed vector too
    assert bk.pack_weights(layers)[0] == -7.0
    other = [layers[0], np.array(layers[1])]     # one layer replaced by a copy: not the noted set any more
    other[1][0, 0] = 99.0
    assert bk.pack_weights(other)[12] == 99.0
    separate = [np.ones((2, 3)), np.zeros((1, 2))]
    np.testing.assert_array_equal(bk.pack_weights(separate), [1, 1, 1, 1, 1, 1, 0, 0])


def test_the_settings_struct_is_refilled_when_and_only_when_a_setting_changes():
    from npbnn_amd import _capi as capi
    ctx = object.__new__(bk.HipContext)

    class Arch:
        n_targets, n_layers = 2, 3
    ctx.arch = Arch()
    cfg = capi.ChainCfg()
    scale = np.ones(3)
    kw = dict(prior_kind=1, prior_scale=scale, w_bound=np.inf, temperature=1.0, lik_temp=1.0, cur_loglik=-10.0, cur_logprior=-2.0,
              n_candidates=0, schedule=0)
    ctx._fill_chain_cfg(cfg, **kw)
    key = cfg._plain_key
    assert key is not None and cfg.cur_loglik == -10.0
    cfg.w_bound = 123.0                           # (a marker: the fast path leaves everything but the chain's state alone)
    ctx._fill_chain_cfg(cfg, **dict(kw, cur_loglik=-11.0, cur_logprior=-3.0))
    assert cfg.w_bound == 123.0 and cfg.cur_loglik == -11.0 and cfg.cur_logprior == -3.0
    ctx._fill_chain_cfg(cfg, **dict(kw, temperature=0.8))
    assert cfg.w_bound == np.inf and cfg.temperature == 0.8 and cfg._plain_key != key
    scale[1] = 2.0                                # the prior scales edited in place: seen (the key holds their bytes)
    ctx._fill_chain_cfg(cfg, **dict(kw, temperature=0.8))
    assert cfg.prior_scale[1] == 2.0
    # regression: sigma and the current sigma go in every time, also on the fast path
    ctx._fill_chain_cfg(cfg, **dict(kw, cur_sigma=np.array([1.0, 2.0]), sigma=np.ones(2)))
    ctx._fill_chain_cfg(cfg, **dict(kw, cur_sigma=np.array([3.0, 4.0]), sigma=np.ones(2)))
    assert [cfg.cur_sigma[0], cfg.cur_sigma[1]] == [3.0, 4.0] and cfg.sigma_given == 1
    # per-weight scales, trainable slopes, sigma multipliers: never the fast path
    ctx._fill_chain_cfg(cfg, **dict(kw, sigma_mult=np.ones((4, 2)), hastings=np.zeros(4)))
    assert cfg._plain_key is None


def test_a_sampler_with_a_cached_plan_pickles_and_copies(tmp_path):
    import contextlib
    import copy
    import io
    import cases
    import npbnn_amd as bn
    from oracle_backend import OracleChainBackend, serve_from_oracle
    serve_from_oracle(lambda b: OracleChainBackend(b, 0))
    dat = cases.classification_data(3, 60, 5, 3)
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=[4, 3], seed=1)
        mcmc = bn.MCMC(bnn, n_iteration=100, sampling_f=10)
        mcmc.run_steps(bnn, 12)
    assert mcmc._ws_copies is not None              # (the plan with its pointers is in there)
    clone = pickle.loads(pickle.dumps(mcmc))
    assert clone._ws_copies is None and clone._current_iteration == 12
    twin = copy.deepcopy(mcmc)
    assert twin._ws_copies is None
    with contextlib.redirect_stdout(io.StringIO()):
        twin._bnn = copy.deepcopy(bnn)
        twin.run_steps(twin._bnn, 5)
        mcmc.run_steps(bnn, 5)
    for a, b in zip(twin._bnn._w_layers, bnn._w_layers):
        np.testing.assert_array_equal(a, b)
