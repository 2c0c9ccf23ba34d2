"""Posterior prediction on the GPU (npbnn_predict_sets / get_posterior_cat_prob) against the reference's golden
vectors (tests/golden/posterior.npz, G7) and the float64 oracle."""
import os

import numpy as np
import pytest

import contextlib
import io

import cases
import npbnn_amd as bn
import oracle as orc

pytestmark = pytest.mark.gpu


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)

TOL = 2e-5      # class probabilities, float32 forward pass vs float64


def _setup(case):
    import npbnn_amd as bn
    inp = cases.posterior_inputs(**{k: v for k, v in case.items() if k != "name"})
    act = bn.ActFun(fun=inp["fun"], prm=np.zeros(2)) if inp["fun"] == "genReLU" else bn.ActFun(fun=inp["fun"])
    return bn, inp, act


@pytest.mark.parametrize("case", cases.POSTERIOR_CASES, ids=lambda c: c["name"])
def test_get_posterior_cat_prob_matches_reference(case, golden_dir):
    bn, inp, act = _setup(case)
    g = np.load(os.path.join(golden_dir, "posterior.npz"))
    k = case["name"]
    for mode in (0, 1, 2):
        np.random.seed(4321)
        probs, summary = bn.get_posterior_cat_prob(inp["x"], post_samples=inp["samples"], post_summary_mode=mode, actFun=act,
                                                   output_act_fun=bn.SoftMax)
        assert probs.shape == g[k + "_probs"].shape
        np.testing.assert_allclose(probs, g[k + "_probs"], atol=TOL, rtol=0)
        ref = g["%s_summary%d" % (k, mode)]
        if mode == 1:
            np.testing.assert_allclose(summary, ref, atol=TOL, rtol=0)
        else:       # counts of arg-max calls / categorical draws: a float32 near-tie may move a single call
            assert np.mean(np.abs(summary - ref)) < 2e-3
            assert np.mean(np.all(summary == ref, axis=1)) > 0.97
    np.random.seed(99)
    _, summary = bn.get_posterior_cat_prob(inp["x"], post_samples=inp["samples"], post_summary_mode=1,
                                           feature_index_to_shuffle=[1, 4], unlink_features_within_block=True, actFun=act,
                                           output_act_fun=bn.SoftMax)
    np.testing.assert_allclose(summary, g[k + "_shuffled_summary1"], atol=TOL, rtol=0)


def test_predict_sets_groups_and_single_predict_agree():
    """Seven sets go through as groups of three, three and one; every set must equal its own single prediction bit for
    bit (same kernel arithmetic per weight set, whatever shares the pass) and the oracle within tolerance."""
    from npbnn_amd import HipContext, _capi as capi
    rs = np.random.default_rng(5)
    n, f, c = 1000, 40, 6
    x = rs.standard_normal((n, f))
    shapes = cases.layer_shapes(f, [12, 7], c, 2)
    sets = [[rs.normal(0, 0.5, s) for s in shapes] for _ in range(7)]
    ctx = HipContext(0)
    ctx.set_data(x)
    ctx.set_arch_from_weights(sets[0], f, capi.ACT_TANH, capi.OUT_SOFTMAX, capi.LIK_NONE)
    y = ctx.predict_sets(sets)
    assert y.shape == (7, n, c)
    for i, w in enumerate(sets):
        np.testing.assert_array_equal(y[i], ctx.predict(w))
        ref = orc.forward(x.astype(np.float32).astype(np.float64), w, orc.Act("tanh"), orc.out_softmax)
        np.testing.assert_allclose(y[i], ref, atol=TOL, rtol=0)
    ctx.close()


def test_predictbnn_writes_the_reference_files(tmp_path):
    import npbnn_amd as bn
    inp = cases.posterior_inputs(seed=31, n_rows=120, n_samples=5)
    dat = dict(data=inp["x"], labels=inp["labels"], test_data=np.zeros((0, inp["x"].shape[1])), test_labels=np.zeros(0))
    np.random.seed(1234)
    bnn = bn.npBNN(dat, n_nodes=[6, 5], actFun=bn.ActFun(fun="tanh"), use_bias_node=2)
    mcmc = bn.MCMC(bnn, n_iteration=50, sampling_f=10, print_f=1000, n_post_samples=5)
    logger = bn.postLogger(bnn, wdir=str(tmp_path), filename="run", log_all_weights=0)
    logger._post_weight_samples = inp["samples"]
    pkl = os.path.join(str(tmp_path), "run.pkl")
    bn.SaveObject([bnn, mcmc, logger], pkl)
    res = bn.predictBNN(inp["x"], pkl, test_labels=inp["labels"], post_summary_mode=1, verbose=0)
    probs = np.load(os.path.join(str(tmp_path), "run_pred_pr.npy"))
    assert probs.shape == (5, 120, 4)
    ref_probs, ref_summary = orc.posterior_cat_prob(inp["x"], inp["samples"], orc.Act("tanh"), orc.out_softmax, summary_mode=1)
    np.testing.assert_allclose(probs, ref_probs, atol=TOL, rtol=0)
    np.testing.assert_allclose(res["post_prob_predictions"], ref_summary, atol=TOL, rtol=0)
    assert os.path.exists(os.path.join(str(tmp_path), "run_pred_mean_pr.txt"))
    assert res["confusion_matrix"].sum() == 120


@pytest.mark.parametrize("tag,blocks", [("single", dict()), ("blocks", {"a": [0, 1, 2], "b": [3, 4], "c": [5, 6, 7, 8, 9, 10]})])
def test_feature_importance_matches_reference(tag, blocks, golden_dir):
    """Same permutations (numpy's global stream), same table as the reference's data frame; accuracies are multiples of
    1/N, so a float32 near-tie could move one by a single instance."""
    bn, inp, act = _setup(cases.POSTERIOR_CASES[0])
    g = np.load(os.path.join(golden_dir, "posterior.npz"))
    np.random.seed(7)
    df = bn.feature_importance(inp["x"], weights_posterior=inp["samples"], true_labels=inp["labels"], n_permutations=4,
                               feature_blocks=blocks, write_to_file=False, post_summary_mode=1, actFun=act, output_act_fun=bn.SoftMax)
    assert list(df.columns) == ['feature_block_index', 'feature_name', 'delta_acc_mean', 'delta_acc_std',
                                'acc_with_feature_randomized_mean', 'acc_with_feature_randomized_std']
    values = df.iloc[:, 2:].to_numpy().astype(float)
    order = df["feature_block_index"].to_numpy().astype(int)
    want = g["fi_%s_values" % tag]
    if np.array_equal(order, g["fi_%s_index" % tag]):
        np.testing.assert_allclose(values, want, atol=1.5 / len(inp["labels"]), rtol=0)
    else:       # a one-instance difference may swap two neighbours of the ranking
        np.testing.assert_allclose(np.sort(values[:, 0]), np.sort(want[:, 0]), atol=1.5 / len(inp["labels"]), rtol=0)


def test_get_posterior_est_of_a_regression_checkpoint(tmp_path):
    """get_posterior_est (BNN_lib.py:715-748): every stored sample's predictions on the checkpoint's own training and test
    matrices, their means, and the samples' error parameters."""
    import npbnn_amd as bn
    dat = cases.regression_data(seed=5, n_rows=150, n_features=6, k=2, n_test=30)
    np.random.seed(1234)
    bnn = bn.npBNN(dat, n_nodes=[5, 4], actFun=bn.ActFun(fun="tanh"), use_bias_node=2, estimation_mode="regression")
    mcmc = bn.MCMC(bnn, n_iteration=50, sampling_f=10, print_f=1000, n_post_samples=4)
    logger = bn.postLogger(bnn, wdir=str(tmp_path), filename="reg", log_all_weights=0)
    rs = np.random.default_rng(3)
    samples = [dict(weights=[w + rs.normal(0, 0.05, w.shape) for w in bnn._w_layers], alphas=np.zeros(3), mcmc_it=i,
                    error_prm=np.array([1.0 + 0.1 * i, 0.9])) for i in range(4)]
    logger._post_weight_samples = samples
    pkl = os.path.join(str(tmp_path), "reg.pkl")
    bn.SaveObject([bnn, mcmc, logger], pkl)
    res = bn.get_posterior_est(pkl)
    assert sorted(res) == ['error_prm', 'post_est', 'post_est_test', 'prm_mean', 'prm_mean_test']
    assert res['post_est'].shape == (4, 150, 2) and res['post_est_test'].shape == (4, 30, 2)
    for which, key in (("data", 'post_est'), ("test_data", 'post_est_test')):
        x = dat[which].astype(np.float32).astype(np.float64)
        for i, smp in enumerate(samples):
            ref = orc.forward(x, smp["weights"], orc.Act("tanh"), orc.out_identity)
            np.testing.assert_allclose(res[key][i], ref, atol=TOL, rtol=0)
    np.testing.assert_allclose(res['prm_mean'], res['post_est'].mean(axis=0), rtol=1e-14)
    np.testing.assert_array_equal(np.array(res['error_prm']), np.array([smp['error_prm'] for smp in samples]))


def test_upstream_format_checkpoint_through_this_packages_tools(golden_dir, tmp_path):
    """tests/golden/export_upstream.pkl - written by this package in upstream's format, then opened, continued and analysed by
    np_bnn itself (make_golden.py G10) - fed to THIS package's predictBNN / get_posterior_est on the GPU: the numbers np_bnn's
    own tools computed from the same file (export.npz)."""
    import shutil
    import cases
    g = np.load(os.path.join(golden_dir, "export.npz"))
    dat = cases.option_data(cases.OPTION_TRACES[cases.EXPORT_CASE])
    pkl = str(tmp_path / "export_upstream.pkl")
    shutil.copy(os.path.join(golden_dir, "export_upstream.pkl"), pkl)
    res = quiet(bn.predictBNN, dat["test_data"], pkl, test_labels=dat["test_labels"], post_summary_mode=1, verbose=0)
    np.testing.assert_allclose(res["post_prob_predictions"], g["predict_mean_prob"], atol=2e-6)
    np.testing.assert_allclose(res["mean_accuracy"], g["predict_accuracy"], atol=1e-12)
    est = bn.get_posterior_est(pkl)
    np.testing.assert_allclose(est["prm_mean"], g["est_prm_mean"], atol=2e-6)
    np.testing.assert_allclose(est["prm_mean_test"], g["est_prm_mean_test"], atol=2e-6)


def test_a_gpu_chain_writes_and_rereads_an_upstream_format_checkpoint(tmp_path):
    import cases
    from test_host_export import _globals_named
    _, bnn, mcmc = cases.option_chain(bn, cases.EXPORT_CASE)
    logger = bn.postLogger(bnn, filename="GPU", wdir=str(tmp_path), export="upstream")
    quiet(bn.run_mcmc, bnn, mcmc, logger)
    assert not [n for n in _globals_named(logger._pklfile) if n[0].startswith("npbnn_amd")]
    b2, m2, l2 = bn.load_obj(logger._pklfile)
    for wa, wb in zip(bnn._w_layers, b2._w_layers):
        np.testing.assert_array_equal(wa, wb)
    assert m2._logLik == mcmc._logLik and m2._accuracy == mcmc._accuracy and len(l2._post_weight_samples) == 6
    np.testing.assert_array_equal(np.asarray(m2._y), np.asarray(mcmc._y))
    m2.run_steps(b2, 30)
    mcmc.run_steps(bnn, 30)
    assert m2._last_accepted_mem == mcmc._last_accepted_mem and m2._logLik == mcmc._logLik
