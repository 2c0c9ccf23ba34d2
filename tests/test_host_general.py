"""The general device chain on CPU: MCMC.run_steps must hand the general chain (numpy stand-in of npbnn_chain_run_general,
tests/oracle_backend.py) exactly the random numbers MCMC.mh_step consumes - the chain's Generator for proposals and accept tests,
numpy's global stream for the indicator flips - for the sampler settings whose proposal changes more than a list of weights:
UpdateFixedNormal, UpdateNormalNormalized (np_bnn/BNN_mcmc.py:27-42,71-82), weight indicators (BNN_env.py:457-464) and feature
indicators (BNN_env.py:424-433)."""
import contextlib
import io

import numpy as np
import pytest

import cases
import npbnn_amd as bn
from oracle_backend import OracleChainBackend, serve_from_oracle


def build(update_function=bn.UpdateNormal, n_nodes=(6, 4), freq_indicator=0, feature_indicators=None, randomize_seed=False, w_bound=np.inf,
          regression=False, unit_sums=False, **mcmc_kw):
    if regression:
        dat = cases.regression_data(7, 300, 10, 2, 0)
        extra, out_kind = dict(estimation_mode="regression"), 1
    else:
        dat = cases.classification_data(7, 300, 10, 3, 40)
        extra, out_kind = {}, 0
    np.random.seed(1234)
    init = None
    if unit_sums:            # the normalising proposal keeps every layer's sum at 1: start there
        init = [np.abs(w) / np.sum(np.abs(w)) for w in bn.init_weight_prm(list(n_nodes), 10, 3, bias_node=2)]
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=list(n_nodes), use_bias_node=2, actFun=bn.ActFun(fun="tanh"), freq_indicator=freq_indicator,
                       feature_indicators=feature_indicators, w_bound=w_bound, init_weights=init, **extra)
    serve_from_oracle(lambda b: OracleChainBackend(b, out_kind))
    n_layers = bnn._n_layers
    mcmc = bn.MCMC(bnn, update_function=update_function, update_f=[0.1] * n_layers, update_ws=[0.02 if unit_sums else 0.05] * n_layers, randomize_seed=randomize_seed,
                   mcmc_id=3, n_iteration=1000, adapt_stop=20, **mcmc_kw)
    return bnn, mcmc


CASES = {
    "fixed_normal": dict(update_function=bn.UpdateFixedNormal, w_bound=0.4),
    "normal_normalized": dict(update_function=bn.UpdateNormalNormalized, unit_sums=True),
    "weight_indicators": dict(n_nodes=(6, 5, 4), freq_indicator=0.3),
    "feature_indicators": dict(feature_indicators=True),
    "feature_and_weight_indicators": dict(n_nodes=(6, 5, 4), freq_indicator=0.3, feature_indicators=True, update_function=bn.UpdateFixedNormal),
    "regression_sigma_and_features": dict(regression=True, feature_indicators=True, estimate_error=True),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("randomize_seed", [False, True])
def test_run_steps_general_equals_mh_step_loop(name, randomize_seed):
    kw = dict(CASES[name], randomize_seed=randomize_seed)
    (ba, ma), (bb, mb) = build(**kw), build(**kw)
    n = 160
    np.random.seed(77)
    for _ in range(n):
        ma.mh_step(ba)
    state_a = np.random.get_state()[1].copy()
    used = []
    real = mb._backend.run_chain_general
    mb._backend.run_chain_general = lambda w, **k: (used.append(len(k["log_u"])), real(w, **k))[1]
    mb.SUB_BATCH = 23
    np.random.seed(77)
    mb.run_steps(bb, 70)
    mb.run_steps(bb, n - 70)
    # regression: the first 100 iterations (sigma fixed at 1) run on the general chain too, in batches that end there
    assert sum(used) == n, "every iteration should have run on the general chain (%s)" % used
    assert ma._current_iteration == mb._current_iteration == n
    assert ma._last_accepted_mem == mb._last_accepted_mem and 2 <= sum(ma._last_accepted_mem) <= 100
    for wa, wb in zip(ba._w_layers, bb._w_layers):
        np.testing.assert_array_equal(wa, wb)
    np.testing.assert_array_equal(ba._indicators, bb._indicators)
    if ba._feature_indicators is not None:
        np.testing.assert_array_equal(ba._feature_indicators, bb._feature_indicators)
        assert bb._feature_indicators.dtype == ba._feature_indicators.dtype
    np.testing.assert_allclose([mb._logLik, mb._logPrior], [ma._logLik, ma._logPrior], rtol=1e-12)
    np.testing.assert_array_equal(np.random.get_state()[1], state_a)          # numpy's global stream was consumed identically
    if not randomize_seed:
        assert ma._rs.random() == mb._rs.random()
    np.testing.assert_allclose(mb._accuracy, ma._accuracy, rtol=1e-12)        # statistics of the accepted state (incl. its override)
    if "regression" in name:
        np.testing.assert_array_equal(np.asarray(ba._error_prm, dtype=float), np.asarray(bb._error_prm, dtype=float))


def test_settings_upstream_cannot_run_stay_on_mh_step():
    """UpdateUniform takes no `rs` (BNN_mcmc.py:86): the first iteration raises, here as upstream; weight indicators with three
    weight matrices read _update_f[3] (BNN_env.py:460) and raise as soon as layer 0 proposes indicators."""
    bnn, mcmc = build(update_function=bn.UpdateUniform)
    assert mcmc._device_mode(bnn, 10) is None
    with pytest.raises(TypeError):
        mcmc.run_steps(bnn, 3)
    bnn, mcmc = build(freq_indicator=0.9)
    assert mcmc._device_mode(bnn, 10) is None
    with pytest.raises(IndexError):
        mcmc.run_steps(bnn, 30)
