"""Host-side fixes of round 3 on CPU (oracle-backed stand-in): light checkpoints (feature matrices in a side file, written
once), MC3 with trainable activation slopes on its default path, in-place edits of the proposal step sizes, Gibbs steps of a
sampler with trainable slopes."""
import contextlib
import io
import os

import numpy as np

import cases
import npbnn_amd as bn
from npbnn_amd import exchange as ex
from oracle_backend import OracleChainBackend, OracleExchangeBackend, serve_from_oracle


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


def small_model(act=None, n_rows=400, n_features=24, **kw):
    dat = cases.classification_data(11, n_rows, n_features, 3, 50)
    np.random.seed(1234)
    return dat, quiet(bn.npBNN, dat, n_nodes=[5, 4], use_bias_node=2, actFun=act or bn.ActFun(fun="tanh"), **kw)


def test_checkpoints_keep_the_feature_matrices_in_a_side_file(tmp_path):
    """postLogger writes the data ONCE (``<checkpoint>_data.npz``) and the pickle without it (the reference pickles the matrices
    with every posterior sample, BNN_env.py:655-658); load_obj puts them back; pickle_data=True keeps upstream's layout."""
    dat, bnn = small_model(n_rows=3000, n_features=64)
    serve_from_oracle(lambda b: OracleChainBackend(b, 0))
    mcmc = bn.MCMC(bnn, n_iteration=60, sampling_f=20, print_f=1000)
    logger = bn.postLogger(bnn, filename="light", wdir=str(tmp_path))
    quiet(bn.run_mcmc, bnn, mcmc, logger)
    side = os.path.join(str(tmp_path), "light_l5_4_data.npz")
    assert os.path.exists(side)
    stamp = os.stat(side).st_mtime_ns
    logger.log_weights(bnn, mcmc)                       # a later sample does not rewrite the side file
    assert os.stat(side).st_mtime_ns == stamp
    assert os.path.getsize(logger._pklfile) < bnn._data.nbytes / 10
    b2, m2, lg = bn.load_obj(logger._pklfile)
    np.testing.assert_array_equal(b2._data, bnn._data)
    np.testing.assert_array_equal(b2._test_data, bnn._test_data)
    assert m2._bnn is b2 and len(lg._post_weight_samples) == len(logger._post_weight_samples)
    assert m2._accuracy == mcmc._accuracy and m2._current_iteration == mcmc._current_iteration
    for u, v in zip(b2._w_layers, bnn._w_layers):
        np.testing.assert_array_equal(u, v)
    serve_from_oracle(lambda b: OracleChainBackend(b, 0))
    np.testing.assert_array_equal(m2._y, mcmc._y)       # predictions are recomputed on demand from the weights
    # without the side file the matrices come from the caller
    os.rename(side, side + ".away")
    b3, _, _ = quiet(bn.load_obj, logger._pklfile)
    assert isinstance(b3._data, bn.DetachedMatrix) and b3._data.shape == bnn._data.shape
    b4, _, _ = bn.load_obj(logger._pklfile, dat=dat)
    np.testing.assert_array_equal(b4._data, bnn._data)
    # upstream's layout on request
    full = bn.postLogger(bnn, filename="full", wdir=str(tmp_path), pickle_data=True)
    full.log_weights(bnn, mcmc)
    assert os.path.getsize(full._pklfile) > bnn._data.nbytes
    assert not os.path.exists(os.path.join(str(tmp_path), "full_l5_4_data.npz"))


def test_mc3_with_trainable_slopes_takes_the_interval_path(tmp_path):
    """ActFun(fun='genReLU', trainable=True): the exchange run and the group pass carry no slope draws, so such chains must
    advance interval by interval (MC3's default path used to unpack 8 pre-drawn arrays into 6 names)."""
    outs = []
    for device in (None, False):
        act = bn.ActFun(fun="genReLU", prm=np.zeros(2) + 0.1, trainable=True)
        dat, bnn = small_model(act=act)
        logger = bn.postLogger(bnn, filename="slopes%s" % device, wdir=str(tmp_path))
        serve_from_oracle(lambda b: OracleExchangeBackend(b, 0))
        np.random.seed(3)
        mc3 = quiet(bn.MC3, bnn, logger=logger, n_post_samples=5, sampling_f=20, n_iteration=200, n_chains=3, swap_frequency=20,
                    verbose=0, adapt_stop=0)
        mc3.device_exchange = device
        chains = mc3._local_chains()
        assert not ex.exchange_ready(chains, 40) and not ex._batchable(chains[:2], 20)
        quiet(mc3.run_mcmc)
        outs.append(mc3)
    a, b = outs
    assert a.swap_log == b.swap_log and len(a.swap_log) == 10
    for (ba, ma), (bb, mb) in zip(a.singleChainArgs, b.singleChainArgs):
        assert ma._logPost == mb._logPost and ma._temperature == mb._temperature
        np.testing.assert_array_equal(ba._act_fun._acc_prm, bb._act_fun._acc_prm)


def test_in_place_edit_of_the_step_sizes_reaches_the_device_path():
    """mcmc._update_ws[i] *= 0.5 changes the proposals of the reference at the next iteration; the copies the pre-draw thread
    works from (and draws made ahead) must follow."""
    runs = []
    for mode in ("loop", "batch"):
        _, bnn = small_model()
        serve_from_oracle(lambda b: OracleChainBackend(b, 0))
        mcmc = bn.MCMC(bnn, n_iteration=1000)
        step = (lambda n: [mcmc.mh_step(bnn) for _ in range(n)]) if mode == "loop" else (lambda n: mcmc.run_steps(bnn, n))
        step(30)
        step(30)                              # (the second call of the batch mode leaves draws of a third in flight)
        mcmc._update_ws[0] *= 0.25
        step(30)
        step(30)
        runs.append((bnn, mcmc))
    (ba, ma), (bb, mb) = runs
    assert ma._last_accepted_mem == mb._last_accepted_mem and sum(ma._last_accepted_mem) > 5
    assert (ma._logLik, ma._logPrior) == (mb._logLik, mb._logPrior)
    for u, v in zip(ba._w_layers, bb._w_layers):
        np.testing.assert_array_equal(u, v)


def test_gibbs_step_takes_the_slope_term_out_of_the_prior():
    """gibbs_step recomputes the log prior WITHOUT the exponential prior on trainable slopes (BNN_env.py:534-538 calls calc_prior);
    the flag the device chain reads must say so, or the next batch would add the term a second time."""
    act = bn.ActFun(fun="genReLU", prm=np.zeros(2) + 0.1, trainable=True)
    _, bnn = small_model(act=act, hyper_p=1)
    serve_from_oracle(lambda b: OracleChainBackend(b, 0))
    mcmc = bn.MCMC(bnn, n_iteration=1000)
    while not mcmc._slope_term_in_prior:
        mcmc.mh_step(bnn)
    np.random.seed(5)
    mcmc.gibbs_step(bnn)
    assert mcmc._slope_term_in_prior is False
    assert mcmc._logPrior == bnn.calc_prior()


def test_light_checkpoint_stores_the_generator_where_the_chain_is():
    """run_steps leaves the draws of the probable next call in flight (one batch ahead, two on the kept dispatch); a light
    checkpoint taken then must hold the generator at the chain's position, not behind those draws - a run resumed from it makes the
    draws of the iterations that follow, as the reference's would."""
    import pickle
    dat, bnn = small_model()
    serve_from_oracle(lambda b: OracleChainBackend(b, 0))
    mcmc = bn.MCMC(bnn, n_iteration=1000, sampling_f=1000, print_f=1000)
    for _ in range(4):                                   # (the third identical call runs on the kept dispatch: two batches ahead)
        mcmc.run_steps(bnn, 25)
        view = mcmc._light_view(bnn)
        stored = pickle.loads(pickle.dumps(view))._gen.bit_generator.state
        assert stored == mcmc._rs.bit_generator.state
    # and the chain goes on as one that was never looked at
    dat2, bnn2 = small_model()
    mcmc2 = bn.MCMC(bnn2, n_iteration=1000, sampling_f=1000, print_f=1000)
    for _ in range(4):
        mcmc2.run_steps(bnn2, 25)
    mcmc.run_steps(bnn, 25)
    mcmc2.run_steps(bnn2, 25)
    assert mcmc._last_accepted_mem == mcmc2._last_accepted_mem
    for u, v in zip(bnn._w_layers, bnn2._w_layers):
        np.testing.assert_array_equal(u, v)
