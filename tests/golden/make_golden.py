#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz from the REFERENCE.

Runs only in the build container, where the upstream repository is mounted at
/root/reference (np_bnn 0.1.23, imported unmodified; nothing of it is copied
into this repository -- only its numeric outputs on seeded synthetic inputs
are stored).  Usage:  python tests/golden/make_golden.py

Groups:
  G1 grid.npz        forward / categorical likelihood / accuracy statistics
  G2 regression.npz  Gaussian likelihoods (fixed, vector, empirical sigma; predicted sigma)
  G3 counts.npz      Poisson / negative-binomial / gamma plug-in likelihoods
  G4 trace_*.npz     Metropolis-Hastings traces (per-proposal logLik', logPrior', accept)
  G5 mc3.npz         MC3 (4 chains) swap sequence and final chain states
  G6 masks.npz       create_mask block layouts
  G7 posterior.npz   get_posterior_cat_prob: per-sample class probabilities and the three summaries
  G8 split.npz       get_data / randomize_data: the train / test split of seeded example tables
  G9 options.npz     one Metropolis-Hastings trace per sampler option beyond the default path (cases.OPTION_TRACES)
  G10 export.npz + export_upstream.pkl   a checkpoint written by npbnn_amd in upstream's format (on the oracle stand-in: no GPU
                     here), opened by np_bnn.load_obj, continued by np_bnn's mh_step, read by np_bnn.predictBNN / get_posterior_est
"""
import contextlib
import io
import os
import re
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import cases  # noqa: E402
import np_bnn as bn  # noqa: E402  (the reference)

HEAD = 16


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


def make_act(fun, prm=None):
    if prm is not None:
        return bn.ActFun(fun=fun, prm=prm)
    return bn.ActFun(fun=fun)


def g1_grid():
    out = {}
    for case in cases.grid_cases():
        inp = cases.grid_inputs(case)
        act = make_act(case["fun"], inp["prm"])
        x, w, lab = inp["x"], inp["weights"], inp["labels"]
        sid = np.arange(x.shape[0])
        z = bn.RunPredict(x, w, act, bn.RegressTransform)
        y = bn.RunPredict(x, w, act, bn.SoftMax)
        h0 = bn.RunHiddenLayer(x + 0, w[0], act, 0)
        with np.errstate(divide="ignore"):
            lik = [bn.calc_likelihood(y, lab, sid),
                   bn.calc_likelihood(y, lab, sid, instance_weight=inp["inst_w"]),
                   bn.calc_likelihood(y, lab, sid, class_weight=inp["class_w"]),
                   bn.calc_likelihood(y, lab, sid, lik_temp=0.5)]
        k = case["name"]
        out[k + "/z_head"] = z[:HEAD]
        out[k + "/y_head"] = y[:HEAD]
        out[k + "/h0_head"] = h0[:HEAD]
        out[k + "/z_colsum"] = z.sum(axis=0)
        out[k + "/lik"] = np.array(lik, dtype=float)
        out[k + "/pred"] = np.argmax(y, axis=1).astype(np.int16)
        out[k + "/acc"] = np.array(bn.CalcAccuracy(y, lab))
        out[k + "/label_acc"] = bn.CalcLabelAccuracy(y, lab)
        out[k + "/label_freq"] = bn.CalcLabelFreq(y)
    np.savez_compressed(os.path.join(HERE, "grid.npz"), **out)
    print("grid.npz:", len(cases.grid_cases()), "cases")


def g2_regression():
    out = {}
    act = bn.ActFun(fun="tanh")
    inp = cases.regression_inputs()
    x, w, t = inp["x"], inp["weights"], inp["targets"]
    y = bn.RunPredict(x, w, act, bn.RegressTransform)
    out["y"] = y
    out["lik_sig1"] = np.array(bn.calc_likelihood_regression(y, t, None, sig2=1))
    out["lik_sigvec"] = np.array(bn.calc_likelihood_regression(y, t, None, sig2=inp["sig_vec"]))
    emp = np.std(y - t, axis=0)
    out["emp_sigma"] = emp
    out["lik_emp"] = np.array(bn.calc_likelihood_regression(y, t, None, sig2=emp))
    out["lik_emp_temp"] = np.array(bn.calc_likelihood_regression(y, t, None, sig2=emp, lik_temp=0.7))
    out["mse"] = np.array(bn.CalcAccuracyRegression(y, t))
    out["mse_col"] = bn.CalcLabelAccuracyRegression(y, t)
    inp2 = cases.regression_inputs(seed=12, double_out=True)
    y2 = bn.RunPredict(inp2["x"], inp2["weights"], act, bn.RegressTransformError)
    out["y_err"] = y2
    out["lik_err"] = np.array(bn.calc_likelihood_regression_error(y2, inp2["targets"], None))
    out["mse_err"] = np.array(bn.CalcAccuracyRegression(y2, inp2["targets"]))
    np.savez_compressed(os.path.join(HERE, "regression.npz"), **out)
    print("regression.npz")


def g3_counts():
    out = {}
    act = bn.ActFun(fun="swish")
    a = cases.count_inputs(seed=23, n_out=1, k=1)
    z = bn.RunPredict(a["x"], a["weights"], act, bn.RegressTransform)
    out["poi_z"] = z
    out["poi"] = np.array(bn.poi_likelihood(z, a["counts"]))
    out["poi_acc"] = np.array(bn.poi_acc(z, a["counts"]))
    b = cases.count_inputs(seed=24, n_out=2, k=1)
    z = bn.RunPredict(b["x"], b["weights"], act, bn.RegressTransform)
    out["nb_z"] = z
    out["nb"] = np.array(bn.negbin_likelihood(z, b["counts"]))
    out["nb10"] = np.array(bn.negbin_likelihood_base10(z, b["counts"]))
    out["nb_acc"] = np.array(bn.negbin_acc(z, b["counts"]))
    out["nb10_acc"] = np.array(bn.negbin_acc_base10(z, b["counts"]))
    out["gamma"] = np.array(bn.gamma_likelihood(z, b["counts"] + 0.5))
    c = cases.count_inputs(seed=25, n_out=4, k=2)
    z = bn.RunPredict(c["x"], c["weights"], act, bn.RegressTransform)
    out["nb2d_z"] = z
    out["nb2d"] = np.array(bn.negbin_likelihood2d(z, c["counts"]))
    out["nb2d_acc"] = np.array(bn.negbin2d_acc(z, c["counts"]))
    np.savez_compressed(os.path.join(HERE, "counts.npz"), **out)
    print("counts.npz")


class Recorder:
    """Wraps a bound callable of a reference object and records calls."""

    def __init__(self, fn, grab):
        self.fn, self.grab, self.rows = fn, grab, []

    def __call__(self, *a, **k):
        r = self.fn(*a, **k)
        self.rows.append(self.grab(r, a, k))
        return r


def build_reference_chain(cfg):
    if cfg["kind"] == "classification":
        dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
        extra = {}
    else:
        dat = cases.regression_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["k"], cfg["n_test"])
        extra = dict(estimation_mode="regression", empirical_error=cfg.get("empirical_error", False))
    np.random.seed(1234)
    bnn = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], actFun=bn.ActFun(fun=cfg["fun"]),
                use_bias_node=cfg["bias"], prior_f=1, p_scale=1, seed=1234, init_std=0.1, **extra)
    mcmc = bn.MCMC(bnn, **cfg["mcmc"])
    return dat, bnn, mcmc


def g4_traces():
    for name, cfg in cases.TRACES.items():
        dat, bnn, mcmc = build_reference_chain(cfg)
        out = {}
        for i, w in enumerate(bnn._w_layers):
            out["w0_%d" % i] = w.copy()
        out["init"] = np.array([mcmc._logLik, mcmc._logPrior, mcmc._accuracy, mcmc._test_accuracy], dtype=float)
        out["init_label_acc"] = np.asarray(mcmc._label_acc, dtype=float)
        out["update_n"] = np.asarray(mcmc._update_n)
        lik_rec = Recorder(mcmc._likelihood_f, lambda r, a, k: float(r))
        mcmc._likelihood_f = lik_rec
        pri_rec = Recorder(bnn.calc_prior, lambda r, a, k: (float(r), [w.copy() for w in k["w"]]))
        bnn.calc_prior = pri_rec
        rows = []
        for it in range(cfg["steps"]):
            mcmc.mh_step(bnn)
            rows.append([lik_rec.rows[-1], pri_rec.rows[-1][0], mcmc._last_accepted, mcmc._logLik,
                         mcmc._logPost, mcmc._accuracy, mcmc._test_accuracy, mcmc._acceptance_rate])
            if it < cfg["keep_w"]:
                for li, w in enumerate(pri_rec.rows[-1][1]):
                    out["wprime_%d_%d" % (it, li)] = w
            pri_rec.rows[-1] = (pri_rec.rows[-1][0], None)
        out["rows"] = np.array(rows, dtype=float)   # logLik', logPrior', accepted, logLik, logPost, acc, test_acc, acc_rate
        for i, w in enumerate(bnn._w_layers):
            out["wfinal_%d" % i] = w
        out["final_update_n"] = np.asarray(mcmc._update_n)
        out["final_update_ws0"] = np.array([u.flat[0] for u in mcmc._update_ws])
        out["final_label_acc"] = np.asarray(mcmc._label_acc, dtype=float)
        if cfg["kind"] == "regression":
            out["final_error_prm"] = np.asarray(bnn._error_prm, dtype=float)
        np.savez_compressed(os.path.join(HERE, "trace_%s.npz" % name), **out)
        print("trace_%s.npz: %d steps, acceptance %.3f" % (name, cfg["steps"], np.mean(out["rows"][:, 2])))


def g5_mc3():
    for name, cfg in cases.MC3_TRACES.items():
        dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
        np.random.seed(1234)
        with tempfile.TemporaryDirectory() as tmp:
            bnn = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], use_bias_node=cfg["bias"], seed=1, init_std=0.1, **cases.mc3_act(bn, cfg))
            logger = bn.postLogger(bnn, filename="MC3", wdir=tmp, log_all_weights=0)
            mc3 = quiet(bn.MC3, bnn, logger=logger, n_post_samples=10, sampling_f=cfg["swap_frequency"],
                        n_iteration=cfg["n_iteration"], n_chains=cfg["n_chains"],
                        swap_frequency=cfg["swap_frequency"], verbose=1)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                mc3.run_mcmc()
            log_rows = np.loadtxt(logger._logfile, skiprows=1)
        swapped = []
        for line in buf.getvalue().splitlines():
            m = re.match(r"^(\d+) SWAPPED (\S+) (\S+) (\S+) (\S+)", line)
            if m:
                swapped.append([float(m.group(i)) for i in range(1, 6)])
        out = dict(rseeds=np.asarray(mc3.rseeds), temperatures0=np.asarray(mc3.temperatures, dtype=float),
                   swapped=np.array(swapped, dtype=float).reshape(-1, 5), log_rows=log_rows)
        out["final_temperature"] = np.array([c[1]._temperature for c in mc3.singleChainArgs], dtype=float)
        out["final_logPost"] = np.array([c[1]._logPost for c in mc3.singleChainArgs], dtype=float)
        out["final_logLik"] = np.array([c[1]._logLik for c in mc3.singleChainArgs], dtype=float)
        out["final_acc_rate"] = np.array([c[1]._acceptance_rate for c in mc3.singleChainArgs], dtype=float)
        for ci, c in enumerate(mc3.singleChainArgs):
            for li, w in enumerate(c[0]._w_layers):
                out["w_c%d_l%d" % (ci, li)] = w
            if c[0]._act_fun._trainable:
                out["alphas_c%d" % ci] = np.asarray(c[0]._act_fun._acc_prm, dtype=float)
        np.savez_compressed(os.path.join(HERE, "%s.npz" % name), **out)
        print("%s.npz: %d swaps accepted of %d" % (name, len(swapped), mc3.n_mc3_iteration))


def g6_masks():
    out = {}
    for bi, (nf, nodes, so, idx, npf) in enumerate(cases.BLOCK_LAYOUTS):
        shapes = cases.layer_shapes(nf, nodes, so, -1)
        w = [np.ones(s) for s in shapes]
        m = bn.create_mask(w, indx_input_list=idx, nodes_per_feature_list=npf)
        for li, mm in enumerate(m):
            out["m%d_%d" % (bi, li)] = mm.astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **out)
    print("masks.npz")


def g7_posterior():
    out = {}
    for case in cases.POSTERIOR_CASES:
        kw = {k: v for k, v in case.items() if k != "name"}
        inp = cases.posterior_inputs(**kw)
        act = bn.ActFun(fun=inp["fun"], prm=np.zeros(2)) if inp["fun"] == "genReLU" else bn.ActFun(fun=inp["fun"])
        for mode in (0, 1, 2):
            np.random.seed(4321)
            probs, summary = quiet(bn.get_posterior_cat_prob, inp["x"], post_samples=inp["samples"], post_summary_mode=mode,
                                   actFun=act, output_act_fun=bn.SoftMax)
            out["%s_summary%d" % (case["name"], mode)] = summary
        out["%s_probs" % case["name"]] = probs
        np.random.seed(99)
        probs_sh, summary_sh = quiet(bn.get_posterior_cat_prob, inp["x"], post_samples=inp["samples"], post_summary_mode=1,
                                     feature_index_to_shuffle=[1, 4], unlink_features_within_block=True, actFun=act,
                                     output_act_fun=bn.SoftMax)
        out["%s_shuffled_summary1" % case["name"]] = summary_sh
    # feature importance: accuracy lost when a feature (or a block of features) is shuffled between the instances
    inp = cases.posterior_inputs(**{k: v for k, v in cases.POSTERIOR_CASES[0].items() if k != "name"})
    act = bn.ActFun(fun=inp["fun"])
    for tag, blocks in (("single", dict()), ("blocks", {"a": [0, 1, 2], "b": [3, 4], "c": [5, 6, 7, 8, 9, 10]})):
        np.random.seed(7)
        df = quiet(bn.feature_importance, inp["x"], weights_posterior=inp["samples"], true_labels=inp["labels"], n_permutations=4,
                   feature_blocks=blocks, write_to_file=False, post_summary_mode=1, actFun=act, output_act_fun=bn.SoftMax)
        out["fi_%s_index" % tag] = df["feature_block_index"].to_numpy().astype(np.int64)
        out["fi_%s_values" % tag] = df.iloc[:, 2:].to_numpy().astype(np.float64)
        out["fi_%s_names" % tag] = df["feature_name"].to_numpy().astype(str)
    np.savez_compressed(os.path.join(HERE, "posterior.npz"), **out)
    print("posterior.npz")


def g8_split():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        f_x, f_lab, f_y = cases.write_split_tables(tmp)
        for name, kw in cases.SPLIT_CASES:
            d = quiet(bn.get_data, f_x, f_y if kw.get("label_mode") == "regression" else f_lab, **kw)
            for key in ("data", "labels", "test_data", "test_labels", "id_data", "id_test_data", "label_dict", "feature_names"):
                v = np.asarray(d[key])
                out["%s_%s" % (name, key)] = v.astype(str) if v.dtype.kind in "UO" else v
        d = quiet(bn.get_data, f_x, header=1, instance_id=1)                       # unlabelled table
        out["unlabelled_data"] = d["data"]
        out["unlabelled_id_data"] = np.asarray(d["id_data"]).astype(str)
    np.savez_compressed(os.path.join(HERE, "split.npz"), **out)
    print("split.npz")


class _Tally:
    """Wraps a proposal function of the reference and adds up the Hastings terms it returns (third element)."""

    def __init__(self, fn, book):
        self.fn, self.book = fn, book

    def __call__(self, *a, **k):
        r = self.fn(*a, **k)
        self.book[0] += float(r[2])
        return r


def g9_options():
    """One trace per entry of cases.OPTION_TRACES.  Per mh_step: the proposal's log-likelihood and calc_prior value (recorded
    at the reference's own calls), the sum of the Hastings terms its proposal functions returned, and the observable state
    after the step (cases.option_state); per gibbs_step the state after it; the final weights, indicators, prior scales."""
    import np_bnn.BNN_env as ref_env
    out = {}
    for name, cfg in cases.OPTION_TRACES.items():
        dat, bnn, mcmc = cases.option_chain(bn, name)
        book = [0.0]
        lik_rec = Recorder(mcmc._likelihood_f, lambda r, a, k: float(r))
        mcmc._likelihood_f = lik_rec
        pri_rec = Recorder(bnn.calc_prior, lambda r, a, k: float(r))
        bnn.calc_prior = pri_rec
        mcmc.update_function = _Tally(mcmc.update_function, book)
        saved = ref_env.UpdateNormal1D, ref_env.multiplier_proposal_vector
        ref_env.UpdateNormal1D = _Tally(saved[0], book)
        ref_env.multiplier_proposal_vector = _Tally(saved[1], book)
        for i, w in enumerate(bnn._w_layers):
            out["%s/w0_%d" % (name, i)] = w.copy()
        out["%s/init" % name] = cases.option_state(bnn, mcmc)
        out["%s/init_acc" % name] = np.array([mcmc._accuracy, mcmc._test_accuracy], dtype=float)
        out["%s/update_n" % name] = np.asarray(mcmc._update_n)
        rows, states, stats = [], [], []
        try:
            for what, n in cases.option_schedule(cfg):
                for _ in range(n):
                    if what == "gibbs":
                        mcmc.gibbs_step(bnn)
                        rows.append([np.nan, pri_rec.rows[-1], np.nan])
                    else:
                        book[0] = 0.0
                        mcmc.mh_step(bnn)
                        rows.append([lik_rec.rows[-1], pri_rec.rows[-1], book[0]])
                    states.append(cases.option_state(bnn, mcmc))
                    stats.append([mcmc._accuracy, mcmc._test_accuracy])
        finally:
            ref_env.UpdateNormal1D, ref_env.multiplier_proposal_vector = saved
        out["%s/rows" % name] = np.array(rows, dtype=float)        # logLik', calc_prior(w', ind'), sum of Hastings terms
        out["%s/states" % name] = np.array(states, dtype=float)    # cases.option_state after every call
        out["%s/stats" % name] = np.array(stats, dtype=float)      # accuracy, test accuracy after every call
        for i, w in enumerate(bnn._w_layers):
            out["%s/wfinal_%d" % (name, i)] = w
        out["%s/final_indicators" % name] = np.asarray(bnn._indicators, dtype=np.int8)
        out["%s/final_label_acc" % name] = np.asarray(mcmc._label_acc, dtype=float)
        if bnn._hyper_p:
            for i, sc in enumerate(bnn._prior_scale):
                out["%s/final_prior_scale_%d" % (name, i)] = np.asarray(sc, dtype=float)
        acc = np.array(states)[:, 3]
        print("  %-13s %3d calls, acceptance %.3f" % (name, len(rows), np.nanmean(acc)))
    np.savez_compressed(os.path.join(HERE, "options.npz"), **out)
    print("options.npz:", len(cases.OPTION_TRACES), "traces")


def g10_export():
    """The round trip of npbnn_amd/export.py, proven against the reference itself:
    (1) cases.EXPORT_CASE runs EXPORT_SWITCH iterations under npbnn_amd (served by the float64 oracle stand-in, the device being
        absent here) with postLogger(export="upstream");
    (2) np_bnn.load_obj opens the checkpoint as np_bnn objects; np_bnn's own mh_step continues the chain from it and must land,
        call for call, on the states of np_bnn's own uninterrupted run of the same case (options.npz) - asserted here;
    (3) np_bnn.predictBNN and np_bnn.get_posterior_est consume the file; their outputs are stored for the tests, which feed
        the committed file to this package's predictBNN / get_posterior_est."""
    import shutil
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))          # tests/
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))       # repository root
    import npbnn_amd
    from oracle_backend import OracleChainBackend, serve_from_oracle
    name = cases.EXPORT_CASE
    cfg = cases.OPTION_TRACES[name]
    golden = np.load(os.path.join(HERE, "options.npz"))
    states = golden["%s/states" % name]
    serve_from_oracle(lambda b: OracleChainBackend(b, 0))
    with tempfile.TemporaryDirectory() as tmp:
        dat, bnn, mcmc = cases.option_chain(npbnn_amd, name)
        logger = npbnn_amd.postLogger(bnn, filename="EXPORT", wdir=tmp, export="upstream")
        quiet(npbnn_amd.run_mcmc, bnn, mcmc, logger)
        assert mcmc._current_iteration == cases.EXPORT_SWITCH
        np.testing.assert_allclose(cases.option_state(bnn, mcmc), states[cases.EXPORT_SWITCH - 1], rtol=1e-10)
        b_up, m_up, l_up = bn.load_obj(logger._pklfile)
        assert type(b_up) is bn.npBNN and type(m_up) is bn.MCMC and type(l_up) is bn.postLogger and type(b_up._act_fun) is bn.ActFun
        for it in range(cases.EXPORT_SWITCH, cfg["steps"]):
            m_up.mh_step(b_up)
            np.testing.assert_allclose(cases.option_state(b_up, m_up), states[it], rtol=1e-10, err_msg="upstream continuing at %d" % it)
        out = {}
        res = quiet(bn.predictBNN, dat["test_data"], logger._pklfile, test_labels=dat["test_labels"], post_summary_mode=1, verbose=0)
        out["predict_mean_prob"] = res["post_prob_predictions"]
        out["predict_accuracy"] = np.array(res["mean_accuracy"])
        est = bn.get_posterior_est(logger._pklfile)
        out["est_prm_mean"] = est["prm_mean"]
        out["est_prm_mean_test"] = est["prm_mean_test"]
        np.random.seed(1234)
        b_restart = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], use_bias_node=cfg["bias"], pickle_file=logger._pklfile,
                          actFun=bn.ActFun(fun="genReLU", prm=np.zeros(2), trainable=True))
        out["restart_w0"] = b_restart._w_layers[0]
        out["restart_alphas"] = np.asarray(b_restart._act_fun._prm, dtype=float)
        shutil.copy(logger._pklfile, os.path.join(HERE, "export_upstream.pkl"))
        for key in ("predict_mean_prob", "est_prm_mean_test"):
            assert np.all(np.isfinite(out[key]))
    np.savez_compressed(os.path.join(HERE, "export.npz"), **out)
    print("export.npz + export_upstream.pkl: upstream continued the chain for %d iterations on np_bnn's own trace"
          % (cfg["steps"] - cases.EXPORT_SWITCH))


if __name__ == "__main__":
    print("reference np_bnn", bn.__version__, "numpy", np.__version__)
    g1_grid()
    g2_regression()
    g3_counts()
    g4_traces()
    g5_mc3()
    g6_masks()
    g7_posterior()
    g8_split()
    g9_options()
    g10_export()
