"""Pins the CPU oracle to the reference: every oracle function is checked
against the golden vectors that tests/golden/make_golden.py produced by
running the reference (np_bnn 0.1.23) on the same seeded inputs.

float64 everywhere; forward/likelihood values must agree to ~1 ulp-level
tolerances (same numpy/scipy build gives bit-equality, the small rtol only
covers a different BLAS on the GPU box's host)."""
import os

import numpy as np
import pytest

import cases
import oracle as orc

RTOL = 1e-11


@pytest.fixture(scope="module")
def grid(golden_dir):
    return np.load(os.path.join(golden_dir, "grid.npz"))


def _act(case, inp):
    return orc.Act(case["fun"], inp["prm"]) if inp["prm"] is not None else orc.Act(case["fun"])


@pytest.mark.parametrize("case", cases.grid_cases(), ids=lambda c: c["name"])
def test_g1_forward_and_categorical(case, grid):
    inp = cases.grid_inputs(case)
    act = _act(case, inp)
    x, w, lab = inp["x"], inp["weights"], inp["labels"]
    k = case["name"]
    sid = np.arange(x.shape[0])
    z = orc.forward_logits(x, w, act)
    y = orc.forward(x, w, act, orc.out_softmax)
    h0 = orc.hidden_layer(x + 0, w[0], act, 0)
    np.testing.assert_allclose(z[:16], grid[k + "/z_head"], rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(y[:16], grid[k + "/y_head"], rtol=RTOL, atol=1e-15)
    np.testing.assert_allclose(h0[:16], grid[k + "/h0_head"], rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(z.sum(axis=0), grid[k + "/z_colsum"], rtol=1e-10, atol=1e-10)
    with np.errstate(divide="ignore"):
        lik = [orc.lik_categorical(y, lab, sid),
               orc.lik_categorical(y, lab, sid, instance_weight=inp["inst_w"]),
               orc.lik_categorical(y, lab, sid, class_weight=inp["class_w"]),
               orc.lik_categorical(y, lab, sid, lik_temp=0.5)]
    np.testing.assert_allclose(lik, grid[k + "/lik"], rtol=RTOL)
    assert np.array_equal(np.argmax(y, axis=1), grid[k + "/pred"])
    assert orc.acc_classification(y, lab) == grid[k + "/acc"]
    np.testing.assert_array_equal(orc.label_acc_classification(y, lab), grid[k + "/label_acc"])
    np.testing.assert_array_equal(orc.label_freq(y), grid[k + "/label_freq"])
    # the three statistics are functions of the confusion matrix (what the HIP epilogue emits)
    cm = orc.confusion_counts(y, lab)
    n = len(lab)
    assert np.trace(cm) / n == grid[k + "/acc"]
    np.testing.assert_allclose(cm.sum(axis=0) / n, grid[k + "/label_freq"], rtol=0, atol=0)
    present = cm.sum(axis=1) > 0
    np.testing.assert_allclose((np.diag(cm) / np.maximum(cm.sum(axis=1), 1))[present], grid[k + "/label_acc"])


def test_class_and_instance_weights_branch_is_broken_like_upstream():
    case = cases.grid_cases()[0]
    inp = cases.grid_inputs(case)
    y = orc.forward(inp["x"], inp["weights"], _act(case, inp), orc.out_softmax)
    with pytest.raises(Exception):
        orc.lik_categorical(y, inp["labels"], np.arange(len(y)), class_weight=inp["class_w"],
                            instance_weight=inp["inst_w"])


def test_g2_regression(golden_dir):
    g = np.load(os.path.join(golden_dir, "regression.npz"))
    act = orc.Act("tanh")
    inp = cases.regression_inputs()
    x, w, t = inp["x"], inp["weights"], inp["targets"]
    y = orc.forward(x, w, act, orc.out_identity)
    np.testing.assert_allclose(y, g["y"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(orc.lik_gaussian(y, t, None, sig2=1), g["lik_sig1"], rtol=RTOL)
    np.testing.assert_allclose(orc.lik_gaussian(y, t, None, sig2=inp["sig_vec"]), g["lik_sigvec"], rtol=RTOL)
    emp = np.std(y - t, axis=0)
    np.testing.assert_allclose(emp, g["emp_sigma"], rtol=RTOL)
    np.testing.assert_allclose(orc.lik_gaussian(y, t, None, sig2=emp), g["lik_emp"], rtol=RTOL)
    # closed form used by the device epilogue == scipy form
    ll, sig = orc.closed_gaussian_empirical(y, t)
    np.testing.assert_allclose(ll, g["lik_emp"], rtol=1e-10)
    np.testing.assert_allclose(sig, g["emp_sigma"], rtol=1e-10)
    ll, _ = orc.closed_gaussian_empirical(y, t, lik_temp=0.7)
    np.testing.assert_allclose(ll, g["lik_emp_temp"], rtol=1e-10)
    np.testing.assert_allclose(orc.mse_all(y, t), g["mse"], rtol=RTOL)
    np.testing.assert_allclose(orc.mse_per_column(y, t), g["mse_col"], rtol=RTOL)
    inp2 = cases.regression_inputs(seed=12, double_out=True)
    y2 = orc.forward(inp2["x"], inp2["weights"], act, orc.out_regress_error)
    np.testing.assert_allclose(y2, g["y_err"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(orc.lik_gaussian_error(y2, inp2["targets"], None), g["lik_err"], rtol=RTOL)
    np.testing.assert_allclose(orc.mse_all(y2, inp2["targets"]), g["mse_err"], rtol=RTOL)


def test_g3_count_likelihoods(golden_dir):
    g = np.load(os.path.join(golden_dir, "counts.npz"))
    act = orc.Act("swish")
    a = cases.count_inputs(seed=23, n_out=1, k=1)
    z = orc.forward(a["x"], a["weights"], act, orc.out_identity)
    np.testing.assert_allclose(z, g["poi_z"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(orc.lik_poisson(z, a["counts"]), g["poi"], rtol=RTOL)
    np.testing.assert_allclose(orc.closed_poisson(z[:, 0], a["counts"][:, 0]), g["poi"], rtol=1e-10)
    b = cases.count_inputs(seed=24, n_out=2, k=1)
    z = orc.forward(b["x"], b["weights"], act, orc.out_identity)
    np.testing.assert_allclose(orc.lik_negbin(z, b["counts"]), g["nb"], rtol=RTOL)
    np.testing.assert_allclose(orc.lik_negbin_base10(z, b["counts"]), g["nb10"], rtol=RTOL)
    np.testing.assert_allclose(orc.closed_negbin(z[:, 0], z[:, 1], b["counts"][:, 0]), g["nb"], rtol=1e-9)
    np.testing.assert_allclose(orc.closed_negbin(z[:, 0], z[:, 1], b["counts"][:, 0], base10=True), g["nb10"], rtol=1e-9)
    np.testing.assert_allclose(orc.lik_gamma(z, b["counts"] + 0.5), g["gamma"], rtol=RTOL)
    c = cases.count_inputs(seed=25, n_out=4, k=2)
    z = orc.forward(c["x"], c["weights"], act, orc.out_identity)
    np.testing.assert_allclose(orc.lik_negbin2d(z, c["counts"]), g["nb2d"], rtol=RTOL)
    np.testing.assert_allclose(orc.closed_negbin(z[:, :2], z[:, 2:], c["counts"]), g["nb2d"], rtol=1e-9)


def test_g3_product_gamma_likelihood_is_upstreams(golden_dir):
    """gamma_likelihood is a host function of the product (all-pairs sum, see its docstring): held to the reference's value."""
    import npbnn_amd as bn
    g = np.load(os.path.join(golden_dir, "counts.npz"))
    b = cases.count_inputs(seed=24, n_out=2, k=1)
    z = orc.forward(b["x"], b["weights"], orc.Act("swish"), orc.out_identity)
    np.testing.assert_allclose(bn.gamma_likelihood(z, b["counts"] + 0.5), g["gamma"], rtol=RTOL)
    np.testing.assert_allclose(bn.gamma_acc(z, b["counts"]), np.mean((np.exp(z[:, 0]) - b["counts"].flatten()) ** 2), rtol=1e-14)


def _build_oracle_chain(cfg):
    if cfg["kind"] == "classification":
        dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
        extra = {}
    else:
        dat = cases.regression_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["k"], cfg["n_test"])
        extra = dict(mode="regression", empirical_error=cfg.get("empirical_error", False))
    np.random.seed(1234)
    st = orc.make_chain(dat["data"], dat["labels"], cfg["n_nodes"], act=orc.Act(cfg["fun"]),
                        use_bias_node=cfg["bias"], prior_kind=1, p_scale=1,
                        test_data=dat["test_data"], test_labels=dat["test_labels"],
                        **extra, **cfg["mcmc"])
    return st


@pytest.mark.parametrize("name", list(cases.TRACES))
def test_g4_mh_trace(name, golden_dir):
    """Free-running chain: same seeds -> identical proposal, accept/reject and
    state sequence as the reference, bit for bit in float64."""
    cfg = cases.TRACES[name]
    g = np.load(os.path.join(golden_dir, "trace_%s.npz" % name))
    st = _build_oracle_chain(cfg)
    for i, w in enumerate(st.w):
        np.testing.assert_array_equal(w, g["w0_%d" % i])
    np.testing.assert_allclose([st.logLik, st.logPrior, st.accuracy, st.test_accuracy], g["init"], rtol=RTOL)
    np.testing.assert_array_equal(st.update_n, g["update_n"])
    st.trace = []
    rows = g["rows"]
    for it in range(cfg["steps"]):
        info = orc.mh_step(st)
        np.testing.assert_allclose(info["logLik"], rows[it, 0], rtol=RTOL, err_msg="it %d" % it)
        np.testing.assert_allclose(info["logPrior"], rows[it, 1], rtol=RTOL)
        assert info["accepted"] == bool(rows[it, 2]), "accept/reject differs at it %d" % it
        np.testing.assert_allclose([st.logLik, st.logPost, st.accuracy, st.test_accuracy, st.acceptance_rate],
                                   rows[it, 3:8], rtol=RTOL)
        if it < cfg["keep_w"]:
            for li, w in enumerate(info["w_prime"]):
                np.testing.assert_array_equal(w, g["wprime_%d_%d" % (it, li)])
    for i, w in enumerate(st.w):
        np.testing.assert_allclose(w, g["wfinal_%d" % i], rtol=0, atol=0)
    np.testing.assert_array_equal(st.update_n, g["final_update_n"])
    np.testing.assert_allclose([u.flat[0] for u in st.update_ws], g["final_update_ws0"], rtol=1e-15)
    np.testing.assert_allclose(st.label_acc, g["final_label_acc"], rtol=RTOL)
    if cfg["kind"] == "regression":
        np.testing.assert_allclose(st.error_prm, g["final_error_prm"], rtol=RTOL)


@pytest.mark.parametrize("name", list(cases.MC3_TRACES))
def test_g5_mc3_trace(name, golden_dir):
    cfg = cases.MC3_TRACES[name]
    g = np.load(os.path.join(golden_dir, "%s.npz" % name))
    dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
    np.random.seed(1234)
    w0 = orc.init_weights(cfg["n_nodes"], cfg["n_features"], cfg["n_classes"], bias_node=cfg["bias"])

    def factory(i, temp):
        a = cfg.get("act")
        act = orc.Act() if a is None else orc.Act(a["fun"], prm=np.array(a["prm"], dtype=float), trainable=a["trainable"])
        return orc.make_chain(dat["data"], dat["labels"], cfg["n_nodes"], act=act, use_bias_node=cfg["bias"],
                              init_w=[w + 0 for w in w0], test_data=dat["test_data"], test_labels=dat["test_labels"],
                              temperature=temp, n_iteration=cfg["swap_frequency"], mcmc_id=i, randomize_seed=True,
                              adapt_freq=50, adapt_f=0.1, adapt_fM=0.6, adapt_stop=1000)

    mc = orc.mc3_make(factory, n_chains=cfg["n_chains"], swap_frequency=cfg["swap_frequency"],
                      n_iteration=cfg["n_iteration"])
    np.testing.assert_array_equal(mc.rseeds, g["rseeds"])
    orc.mc3_run(mc)
    accepted = [(i, s) for i, s in enumerate(mc.swaps) if s[4]]
    assert [i for i, _ in accepted] == [int(r[0]) for r in g["swapped"]]
    np.testing.assert_allclose([st.temperature for st in mc.chains], g["final_temperature"], rtol=0)
    np.testing.assert_allclose([st.logPost for st in mc.chains], g["final_logPost"], rtol=RTOL)
    np.testing.assert_allclose([st.logLik for st in mc.chains], g["final_logLik"], rtol=RTOL)
    np.testing.assert_allclose([st.acceptance_rate for st in mc.chains], g["final_acc_rate"], rtol=RTOL)
    for ci, st in enumerate(mc.chains):
        for li, w in enumerate(st.w):
            np.testing.assert_array_equal(w, g["w_c%d_l%d" % (ci, li)])
        if st.act.trainable:
            np.testing.assert_array_equal(np.asarray(st.act.acc_prm, dtype=float), g["alphas_c%d" % ci])


def test_g6_block_masks(golden_dir):
    g = np.load(os.path.join(golden_dir, "masks.npz"))
    for bi, (nf, nodes, so, idx, npf) in enumerate(cases.BLOCK_LAYOUTS):
        shapes = cases.layer_shapes(nf, nodes, so, -1)
        m = orc.block_mask([np.ones(s) for s in shapes], idx, npf)
        for li, mm in enumerate(m):
            np.testing.assert_array_equal(mm.astype(np.int8), g["m%d_%d" % (bi, li)])


@pytest.mark.parametrize("case", cases.POSTERIOR_CASES, ids=lambda c: c["name"])
def test_g7_posterior_prediction(case, golden_dir):
    """get_posterior_cat_prob: per-sample class probabilities and the three summaries (incl. the consumption of the
    global numpy stream by the column shuffle and by the posterior-predictive resampling)."""
    g = np.load(os.path.join(golden_dir, "posterior.npz"))
    inp = cases.posterior_inputs(**{k: v for k, v in case.items() if k != "name"})
    act = orc.Act(inp["fun"], np.zeros(2)) if inp["fun"] == "genReLU" else orc.Act(inp["fun"])
    k = case["name"]
    for mode in (0, 1, 2):
        np.random.seed(4321)
        probs, summary = orc.posterior_cat_prob(inp["x"], inp["samples"], act, orc.out_softmax, summary_mode=mode)
        np.testing.assert_allclose(probs, g[k + "_probs"], rtol=RTOL, atol=1e-15)
        np.testing.assert_allclose(summary, g["%s_summary%d" % (k, mode)], rtol=RTOL, atol=1e-15)
    np.random.seed(99)
    _, summary = orc.posterior_cat_prob(inp["x"], inp["samples"], act, orc.out_softmax, summary_mode=1,
                                        feature_index_to_shuffle=[1, 4], unlink_features_within_block=True)
    np.testing.assert_allclose(summary, g[k + "_shuffled_summary1"], rtol=RTOL, atol=1e-15)


@pytest.mark.parametrize("tag,blocks", [("single", None), ("blocks", {"a": [0, 1, 2], "b": [3, 4], "c": [5, 6, 7, 8, 9, 10]})])
def test_g7_feature_importance(tag, blocks, golden_dir):
    g = np.load(os.path.join(golden_dir, "posterior.npz"))
    case = cases.POSTERIOR_CASES[0]
    inp = cases.posterior_inputs(**{k: v for k, v in case.items() if k != "name"})
    np.random.seed(7)
    order, names, table = orc.feature_importance(inp["x"], inp["samples"], orc.Act(inp["fun"]), orc.out_softmax, inp["labels"],
                                                 n_permutations=4, feature_blocks=blocks, summary_mode=1)
    np.testing.assert_array_equal(order, g["fi_%s_index" % tag])
    assert [str(n) for n in names] == [str(n) for n in g["fi_%s_names" % tag]]
    np.testing.assert_allclose(table, g["fi_%s_values" % tag], rtol=1e-12, atol=1e-15)


# ---- G9: one reference trace per sampler option beyond the default path -------------------------------------------------
_ORACLE_NAMES = dict(UpdateFixedNormal=orc.propose_fixed_normal, UpdateNormalNormalized=orc.propose_normal_normalized,
                     RegressTransformError=orc.out_regress_error, poi_likelihood=orc.lik_poisson,
                     poi_acc=lambda y, lab: np.mean((np.exp(y[:, 0]) - lab.flatten()) ** 2))        # BNN_lik.py:94-96


def _build_oracle_option_chain(name):
    cfg = cases.OPTION_TRACES[name]
    dat = cases.option_data(cfg)
    a = dict(cfg["act"])
    act = orc.Act(a["fun"], prm=np.array(a["prm"], dtype=float) if "prm" in a else None, trainable=a.get("trainable", False))
    b, m = dict(cfg["bnn"]), dict(cfg["mcmc"])
    kw = dict(act=act, use_bias_node=cfg["bias"], prior_kind=b.get("prior_f", 1), p_scale=b.get("p_scale", 1),
              mode=b.get("estimation_mode", "classification"), empirical_error=b.get("empirical_error", False),
              hyper_p=b.get("hyper_p", 0), freq_indicator=b.get("freq_indicator", 0), prior_ind1=b.get("prior_ind1", 0.5),
              feature_indicators=b.get("feature_indicators", False), size_output=b.get("size_output"),
              test_data=dat["test_data"], test_labels=dat["test_labels"])
    if "output_act_fun" in b:
        kw["out_fn"] = _ORACLE_NAMES[b["output_act_fun"]]
    if b.get("use_class_weights"):                      # BNN_env.py:98-102
        counts = np.unique(dat["labels"], return_counts=True)[1]
        cw = 1 / (counts / np.max(counts))
        kw["class_weights"] = cw / np.mean(cw)
    if cfg.get("instance_weights"):
        kw["instance_weights"] = np.random.default_rng(cfg["seed"] + 1000).uniform(0.2, 2.0, len(dat["labels"]))
    if cfg.get("init_weights") == "unit_sum":
        rs = np.random.default_rng(cfg["seed"] + 2000)
        drawn = [rs.uniform(-0.5, 1.0, sh) for sh in cases.layer_shapes(cfg["n_features"], cfg["n_nodes"], cfg["n_classes"], cfg["bias"])]
        kw["init_w"] = [w / w.sum() for w in drawn]
    for key in ("sampling_f", "print_f"):           # (driver cadence: nothing the chain itself reads)
        m.pop(key, None)
    for key in ("likelihood_f", "accuracy_f"):
        if key in m:
            kw[key] = _ORACLE_NAMES[m.pop(key)]
    if "update_function" in cfg:
        kw["update_function"] = _ORACLE_NAMES[cfg["update_function"]]
    np.random.seed(1234)
    st = orc.make_chain(dat["data"], dat["labels"], cfg["n_nodes"], **kw, **m)
    for attr, value in cfg.get("post_init", {}).items():
        setattr(st, attr.lstrip("_"), value)
    return st


def _oracle_option_state(st):
    parts = [[st.logLik, st.logPrior, st.logPost, st.last_accepted, st.acceptance_rate, st.it]]
    if st.act.trainable:
        parts += [np.ravel(st.act.prm), np.ravel(st.act.acc_prm)]
    if st.mode == "regression":
        parts.append(np.ravel(np.ones(st.size_output) * st.error_prm))
    if st.freq_indicator:
        parts.append([np.sum(st.indicators)])
    if st.feature_ind is not None:
        parts.append(np.ravel(st.feature_ind))
    if st.hyper_p:
        parts.append([np.sum([np.sum(s) for s in st.prior_scale])])
    return np.concatenate([np.asarray(p, dtype=float) for p in parts])


@pytest.mark.parametrize("name", list(cases.OPTION_TRACES))
def test_g9_option_traces(name, golden_dir):
    """The oracle's chain under every sampler option against the reference's own run: per proposal the log-likelihood, the
    prior and the Hastings terms; after every call the whole observable state; at the end weights, indicators, scales."""
    import option_traces as ot
    g = ot.load(np.load(os.path.join(golden_dir, "options.npz")), name)
    cfg = cases.OPTION_TRACES[name]
    st = _build_oracle_option_chain(name)
    for i, w in enumerate(st.w):
        np.testing.assert_array_equal(w, g["w0_%d" % i])
    np.testing.assert_array_equal(st.update_n, g["update_n"])
    np.testing.assert_allclose(_oracle_option_state(st), g["init"], rtol=RTOL)
    np.testing.assert_allclose([st.accuracy, st.test_accuracy], g["init_acc"], rtol=RTOL)
    call = 0
    for what, n in cases.option_schedule(cfg):
        for _ in range(n):
            if what == "gibbs":
                orc.gibbs_step(st)
            else:
                info = orc.mh_step(st)
                np.testing.assert_allclose(info["logLik"], g["rows"][call, 0], rtol=RTOL, err_msg="call %d" % call)
                np.testing.assert_allclose(info["hastings"], g["rows"][call, 2], rtol=RTOL, atol=1e-12)
            np.testing.assert_allclose(_oracle_option_state(st), g["states"][call], rtol=RTOL, atol=1e-12, err_msg="call %d" % call)
            np.testing.assert_allclose([st.accuracy, st.test_accuracy], g["stats"][call], rtol=RTOL)
            call += 1
    for i, w in enumerate(st.w):
        np.testing.assert_array_equal(w, g["wfinal_%d" % i])
    np.testing.assert_array_equal(st.indicators.astype(np.int8), g["final_indicators"])
    np.testing.assert_allclose(st.label_acc, g["final_label_acc"], rtol=RTOL)
    if st.hyper_p:
        for i, sc in enumerate(st.prior_scale):
            np.testing.assert_allclose(sc, g["final_prior_scale_%d" % i], rtol=1e-13)
