"""Seeded synthetic inputs shared by the golden-vector generator
(``make_golden.py``, runs only where /root/reference exists) and the tests
(run anywhere).  Inputs are regenerated from seeds so the committed ``.npz``
fixtures only need to hold the reference's *outputs*.
"""
import itertools

import numpy as np

ACTS = [("ReLU", None), ("genReLU", 0.01), ("swish", None), ("tanh", None)]
BIAS_MODES = [0, 1, 2, 3, -1]
SETUPS = [(7, [5, 5]), (64, [32, 8])]
N_CLASSES = [2, 5, 10]
N_ROWS = 257          # deliberately ragged (not a multiple of 16/32/64)


def layer_shapes(n_features, n_nodes, size_output, bias_node):
    """Weight shapes (out x in[+1]) for a use_bias_node setting, bias = column 0
    (reference: np_bnn/BNN_mcmc.py:9-25)."""
    bn = 1 if bias_node >= 1 else 0
    bn2 = 1 if bias_node >= 2 else 0
    bn3 = 1 if bias_node in (3, -1) else 0
    shapes = [(n_nodes[0], n_features + bn)]
    for i in range(1, len(n_nodes)):
        shapes.append((n_nodes[i], n_nodes[i - 1] + bn2))
    shapes.append((size_output, n_nodes[-1] + bn3))
    return shapes


def grid_cases():
    """G1: forward / categorical-likelihood grid."""
    out = []
    for (nf, nodes), c, (fun, alpha), bias in itertools.product(SETUPS, N_CLASSES, ACTS, BIAS_MODES):
        name = "f%d_h%s_c%d_%s_b%d" % (nf, "x".join(map(str, nodes)), c, fun, bias)
        out.append(dict(name=name, n_features=nf, n_nodes=nodes, n_classes=c, fun=fun,
                        alpha=alpha, bias=bias))
    return out


def grid_inputs(case, n_rows=N_ROWS):
    """Inputs of one G1 case.  Weights are N(0, 0.5) so that activations leave
    their linear range and softmax rows are far from uniform."""
    seed = abs(hash_name(case["name"])) % (2 ** 31)
    rs = np.random.default_rng(seed)
    x = rs.standard_normal((n_rows, case["n_features"]))
    labels = rs.integers(0, case["n_classes"], n_rows)
    labels[: case["n_classes"]] = np.arange(case["n_classes"])     # every class present
    shapes = layer_shapes(case["n_features"], case["n_nodes"], case["n_classes"], case["bias"])
    weights = [rs.normal(0, 0.5, s) for s in shapes]
    inst_w = rs.uniform(0.1, 2.0, n_rows)
    class_w = rs.uniform(0.5, 1.5, case["n_classes"])
    prm = None
    if case["fun"] == "genReLU":
        prm = np.zeros(len(case["n_nodes"])) + case["alpha"]
    return dict(x=x, labels=labels, weights=weights, inst_w=inst_w, class_w=class_w, prm=prm)


def hash_name(s):
    """Stable (process-independent) string hash."""
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


# ---- G2 / G3: regression and count-data likelihood inputs ------------------
def regression_inputs(seed=11, n_rows=301, n_features=9, n_nodes=(6, 4), k=2, bias=2, double_out=False):
    rs = np.random.default_rng(seed)
    x = rs.standard_normal((n_rows, n_features))
    size_out = 2 * k if double_out else k
    shapes = layer_shapes(n_features, list(n_nodes), size_out, bias)
    weights = [rs.normal(0, 0.4, s) for s in shapes]
    targets = rs.standard_normal((n_rows, k)) * 0.7 + 0.2
    sig_vec = rs.uniform(0.5, 1.5, k)
    return dict(x=x, weights=weights, targets=targets, sig_vec=sig_vec)


def count_inputs(seed=23, n_rows=211, n_features=6, n_nodes=(5, 3), n_out=2, k=1):
    rs = np.random.default_rng(seed)
    x = rs.standard_normal((n_rows, n_features))
    shapes = layer_shapes(n_features, list(n_nodes), n_out, 2)
    weights = [rs.normal(0, 0.3, s) for s in shapes]
    counts = rs.poisson(3.0, (n_rows, k)).astype(float)
    return dict(x=x, weights=weights, counts=counts)


# ---- G4 / G5: sampler traces ------------------------------------------------
def classification_data(seed, n_rows, n_features, n_classes, n_test=0, n_informative=None):
    """Learnable synthetic classification problem (``n_informative``: only the first so many features carry signal)."""
    rs = np.random.default_rng(seed)
    x = rs.standard_normal((n_rows + n_test, n_features))
    proj = rs.standard_normal((n_features, n_classes)) / np.sqrt(n_features)
    if n_informative is not None:
        proj[n_informative:] = 0
        proj *= np.sqrt(n_features / n_informative)
    score = x @ proj + 0.3 * rs.standard_normal((n_rows + n_test, n_classes))
    lab = np.argmax(score, axis=1)
    lab[:n_classes] = np.arange(n_classes)
    if n_test:
        lab[n_rows:n_rows + n_classes] = np.arange(n_classes)
    return dict(data=x[:n_rows], labels=lab[:n_rows], test_data=x[n_rows:], test_labels=lab[n_rows:])


def regression_data(seed, n_rows, n_features, k, n_test=0):
    rs = np.random.default_rng(seed)
    x = rs.standard_normal((n_rows + n_test, n_features))
    w1 = rs.standard_normal((n_features, 8)) / np.sqrt(n_features)
    w2 = rs.standard_normal((8, k))
    y = np.tanh(x @ w1) @ w2 + 0.5 * rs.standard_normal((n_rows + n_test, k))
    return dict(data=x[:n_rows], labels=y[:n_rows], test_data=x[n_rows:], test_labels=y[n_rows:])


TRACES = {
    # config-1 shape (bnn_classify.py settings)
    "cfg1": dict(kind="classification", seed=101, n_rows=2250, n_features=128, n_classes=5, n_test=250,
                 n_nodes=[5, 5], fun="tanh", bias=2, steps=500, keep_w=20,
                 mcmc=dict(update_f=[0.05, 0.05, 0.07], update_ws=[0.075, 0.075, 0.075], n_iteration=10000,
                           adapt_f=0.3, adapt_fM=0.6)),
    # config-2 shape scaled to N=4096
    "cfg2s": dict(kind="classification", seed=102, n_rows=4096, n_features=256, n_classes=10, n_test=0,
                  n_nodes=[32, 8], fun="tanh", bias=2, steps=200, keep_w=5, mcmc=dict()),
    # config-4 shape scaled to N=3000 (bnn_regress.py settings)
    "cfg4s": dict(kind="regression", seed=104, n_rows=3000, n_features=64, k=2, n_test=300,
                  n_nodes=[16, 4], fun="tanh", bias=2, steps=200, keep_w=5, empirical_error=True,
                  mcmc=dict(update_ws=[0.025, 0.025, 0.05], update_f=[0.005, 0.005, 0.05], n_iteration=20000,
                            adapt_f=0.3, estimate_error=False)),
}

# ---- G9: sampler options beyond the default path (one reference-generated trace each) ----------------------------------
# Every entry is built through the SAME call surface on the reference (make_golden.py) and on the product (tests), by
# ``option_chain`` below: npBNN(...) / ActFun(...) / MCMC(...) keyword for keyword.  ``gibbs_every`` = k: a gibbs_step after
# every k-th mh_step (BNN_env.py:534-538).  ``post_init``: attributes set on the sampler after construction.
_CLS = dict(kind="classification", seed=201, n_rows=1500, n_features=24, n_classes=4, n_test=150)
_REG = dict(kind="regression", seed=202, n_rows=1500, n_features=16, k=2, n_test=150)
_MC = dict(update_f=[0.05, 0.05, 0.1], update_ws=[0.075, 0.075, 0.075], n_iteration=5000)
OPTION_TRACES = {
    # (i) trainable activation slopes (BNN_env.py:416-421,502-503)
    "slopes": dict(_CLS, n_nodes=[6, 5], bias=2, steps=300, act=dict(fun="genReLU", prm=[0.1, 0.2], trainable=True),
                   bnn=dict(), mcmc=dict(_MC)),
    # slopes proposed but never used by the forward pass: ReLU + trainable (BNN_lib.py:74-87)
    "slopes_relu": dict(_CLS, n_nodes=[6, 5], bias=1, steps=200, act=dict(fun="ReLU", prm=[0.3], trainable=True),
                        bnn=dict(), mcmc=dict(_MC)),
    # (ii) feature indicators, the column override live after adapt_stop (BNN_env.py:9-17,424-433)
    "feature_ind": dict(_CLS, seed=203, n_informative=6, n_nodes=[6, 5], bias=2, steps=300, act=dict(fun="tanh"),
                        bnn=dict(feature_indicators=True), mcmc=dict(_MC, adapt_stop=10)),
    # (iii) weight indicators: four weight matrices and a four-element update_f (BNN_env.py:457-464)
    "weight_ind": dict(_CLS, seed=204, n_nodes=[6, 5, 4], bias=2, steps=300, act=dict(fun="swish"),
                       bnn=dict(freq_indicator=0.3, prior_ind1=0.4),
                       mcmc=dict(update_f=[0.05, 0.05, 0.05, 0.08], update_ws=[0.075] * 4, n_iteration=5000)),
    # (iv) hyper-prior scales, a gibbs_step every 10 iterations (BNN_env.py:196-221,534-538; BNN_mcmc.py:126-150)
    "hyper1": dict(_CLS, n_nodes=[6, 5], bias=2, steps=250, act=dict(fun="tanh"), bnn=dict(hyper_p=1), mcmc=dict(_MC),
                   gibbs_every=10),
    "hyper2": dict(_CLS, n_nodes=[6, 5], bias=3, steps=250, act=dict(fun="ReLU"), bnn=dict(hyper_p=2), mcmc=dict(_MC),
                   gibbs_every=10),
    "hyper3": dict(_REG, n_nodes=[6, 4], bias=2, steps=250, act=dict(fun="tanh"),
                   bnn=dict(hyper_p=3, estimation_mode="regression", empirical_error=True),
                   mcmc=dict(_MC, estimate_error=False), gibbs_every=7),
    # (v) the other proposal kernels (BNN_mcmc.py:27-42,71-82)
    "fixed_normal": dict(_CLS, n_nodes=[6, 5], bias=2, steps=300, act=dict(fun="tanh"), bnn=dict(),
                         mcmc=dict(update_f=[0.02, 0.05, 0.1], update_ws=[0.1, 0.1, 0.1], n_iteration=5000),
                         update_function="UpdateFixedNormal"),
    # (the normalising proposal rescales a layer to unit sum: the chain starts from layers that have it)
    "normalized": dict(_CLS, n_nodes=[6, 5], bias=0, steps=200, act=dict(fun="tanh"), bnn=dict(), init_weights="unit_sum",
                       mcmc=dict(update_f=[0.02, 0.05, 0.1], update_ws=[0.05, 0.05, 0.05], n_iteration=5000),
                       update_function="UpdateNormalNormalized"),
    # (vi) the regression error parameter with multiplier proposals (BNN_env.py:435-444; BNN_mcmc.py:101-113).  Upstream
    # stores the bare scalar 1 on an accepted step while sigma is still fixed and then fails in the multiplier proposal
    # (SURVEY 7.3 #8b), so the proposals start at iteration 0 here.
    "sigma": dict(_REG, n_nodes=[6, 4], bias=2, steps=300, act=dict(fun="tanh"), bnn=dict(estimation_mode="regression"),
                  mcmc=dict(_MC), post_init=dict(_estimate_error=-1)),
    # (vii) class weights / instance weights inside a chain (BNN_lib.py:105-119)
    "class_w": dict(_CLS, seed=205, imbalance=True, n_nodes=[6, 5], bias=2, steps=250, act=dict(fun="tanh"),
                    bnn=dict(use_class_weights=1), mcmc=dict(_MC)),
    "inst_w": dict(_CLS, seed=206, n_nodes=[6, 5], bias=2, steps=250, act=dict(fun="tanh"), bnn=dict(), instance_weights=True,
                   mcmc=dict(_MC)),
    # the other priors (BNN_env.py:135-150), a heated chain with a tempered likelihood (BNN_env.py:493-494)
    "cauchy": dict(_CLS, n_nodes=[6, 5], bias=2, steps=200, act=dict(fun="tanh"), bnn=dict(prior_f=2, p_scale=0.5), mcmc=dict(_MC)),
    "laplace": dict(_CLS, n_nodes=[6, 5], bias=2, steps=200, act=dict(fun="swish"), bnn=dict(prior_f=3, p_scale=0.7), mcmc=dict(_MC)),
    "uniform": dict(_CLS, n_nodes=[6, 5], bias=2, steps=200, act=dict(fun="tanh"), bnn=dict(prior_f=0, p_scale=0.25),
                    mcmc=dict(_MC, update_ws=[0.15, 0.15, 0.15])),
    "heated": dict(_CLS, n_nodes=[6, 5], bias=2, steps=200, act=dict(fun="tanh"), bnn=dict(),
                   mcmc=dict(_MC, temperature=0.6, likelihood_tempering=0.5)),
    # predicted-sigma regression and a count likelihood plug-in as whole chains (BNN_lib.py:134-143; BNN_lik.py:5-14,81-84)
    "reg_error": dict(_REG, n_nodes=[6, 4], bias=2, steps=200, act=dict(fun="tanh"),
                      bnn=dict(estimation_mode="regression-error", output_act_fun="RegressTransformError"), mcmc=dict(_MC)),
    "poisson": dict(kind="counts", seed=207, n_rows=1500, n_features=12, n_test=150, n_nodes=[6, 4], bias=2, steps=200,
                    act=dict(fun="swish"), bnn=dict(estimation_mode="custom", size_output=1),
                    mcmc=dict(_MC, likelihood_f="poi_likelihood", accuracy_f="poi_acc")),
}


# ---- G10: a checkpoint in upstream's format (npbnn_amd/export.py), written here, consumed by np_bnn's own tools -----------
EXPORT_CASE = "export_small"
OPTION_TRACES[EXPORT_CASE] = dict(kind="classification", seed=208, n_rows=320, n_features=10, n_classes=3, n_test=60,
                                  n_nodes=[5, 4], bias=2, steps=240, act=dict(fun="genReLU", prm=[0.05, 0.1], trainable=True),
                                  bnn=dict(), mcmc=dict(_MC, n_iteration=120, sampling_f=20, print_f=100000))
EXPORT_SWITCH = 120          # iterations under this package before the checkpoint is handed to upstream


def count_data(seed, n_rows, n_features, n_test=0):
    rs = np.random.default_rng(seed)
    x = rs.standard_normal((n_rows + n_test, n_features))
    eta = 0.8 + x[:, :3] @ np.array([0.5, -0.4, 0.3])
    y = rs.poisson(np.exp(eta)).astype(float).reshape(-1, 1)
    return dict(data=x[:n_rows], labels=y[:n_rows], test_data=x[n_rows:], test_labels=y[n_rows:])


def option_data(cfg):
    if cfg["kind"] == "classification":
        dat = classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"],
                                  n_informative=cfg.get("n_informative"))
        if cfg.get("imbalance"):       # thin out two of the classes
            keep = np.ones(cfg["n_rows"], dtype=bool)
            lab = dat["labels"]
            for c, every in ((1, 3), (2, 5)):
                rows = np.nonzero(lab == c)[0]
                keep[rows[np.arange(len(rows)) % every != 0]] = False
            keep[:cfg["n_classes"]] = True
            dat["data"], dat["labels"] = dat["data"][keep], lab[keep]
        return dat
    if cfg["kind"] == "regression":
        return regression_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["k"], cfg["n_test"])
    return count_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_test"])


def option_chain(bn, name, **mcmc_extra):
    """Model and sampler of OPTION_TRACES[name] through package ``bn`` (the reference ``np_bnn`` or ``npbnn_amd``: the same
    keywords go to the same constructors).  Returns (data, model, sampler)."""
    import contextlib
    import io
    cfg = OPTION_TRACES[name]
    dat = option_data(cfg)
    act_kw = dict(cfg["act"])
    if "prm" in act_kw:
        act_kw["prm"] = np.array(act_kw["prm"], dtype=float)
    bnn_kw = dict(cfg["bnn"])
    if isinstance(bnn_kw.get("output_act_fun"), str):
        bnn_kw["output_act_fun"] = getattr(bn, bnn_kw["output_act_fun"])
    if cfg.get("instance_weights"):
        bnn_kw["instance_weights"] = np.random.default_rng(cfg["seed"] + 1000).uniform(0.2, 2.0, len(dat["labels"]))
    if cfg.get("init_weights") == "unit_sum":
        rs = np.random.default_rng(cfg["seed"] + 2000)
        outputs = cfg["n_classes"] if cfg["kind"] == "classification" else cfg["k"]
        drawn = [rs.uniform(-0.5, 1.0, shape) for shape in layer_shapes(cfg["n_features"], cfg["n_nodes"], outputs, cfg["bias"])]
        bnn_kw["init_weights"] = [w / w.sum() for w in drawn]
    mcmc_kw = dict(cfg["mcmc"])
    for key in ("likelihood_f", "accuracy_f", "accuracy_lab_f"):
        if isinstance(mcmc_kw.get(key), str):
            mcmc_kw[key] = getattr(bn, mcmc_kw[key])
    if "update_function" in cfg:
        mcmc_kw["update_function"] = getattr(bn, cfg["update_function"])
    mcmc_kw.update(mcmc_extra)
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=cfg["n_nodes"], actFun=bn.ActFun(**act_kw), use_bias_node=cfg["bias"], seed=1234, **bnn_kw)
        mcmc = bn.MCMC(bnn, **mcmc_kw)
    for attr, value in cfg.get("post_init", {}).items():
        setattr(mcmc, attr, value)
    return dat, bnn, mcmc


def option_state(bnn, mcmc):
    """The observable state after an iteration, as one float vector (its layout depends on the case only)."""
    parts = [[mcmc._logLik, mcmc._logPrior, mcmc._logPost, mcmc._last_accepted, mcmc._acceptance_rate, mcmc._current_iteration]]
    act = bnn._act_fun
    if act._trainable:
        parts += [np.ravel(act._prm), np.ravel(act._acc_prm)]
    if bnn._estimation_mode == "regression":
        parts.append(np.ravel(np.ones(bnn._size_output) * bnn._error_prm))
    if bnn._freq_indicator:
        parts.append([np.sum(bnn._indicators)])
    if bnn._feature_indicators is not None:
        parts.append(np.ravel(bnn._feature_indicators))
    if bnn._hyper_p:
        parts.append([np.sum([np.sum(s) for s in bnn._prior_scale])])
    return np.concatenate([np.asarray(p, dtype=float) for p in parts])


def option_schedule(cfg):
    """The calls of a trace in order: ("mh", n) = n mh_step calls, ("gibbs", 1) = one gibbs_step."""
    every = cfg.get("gibbs_every")
    if not every:
        return [("mh", cfg["steps"])]
    out, left = [], cfg["steps"]
    while left > 0:
        n = min(every, left)
        out.append(("mh", n))
        left -= n
        if n == every:
            out.append(("gibbs", 1))
    return out


MC3_TRACE = dict(seed=105, n_rows=1000, n_features=32, n_classes=4, n_test=100, n_nodes=[5, 5],
                 bias=-1, n_chains=4, swap_frequency=20, n_iteration=600)

# the same with trainable activation slopes (every chain proposes its own; MC3 deep-copies the model per chain, BNN_mc3.py:55-58)
MC3_SLOPES_TRACE = dict(MC3_TRACE, seed=106, n_chains=3, n_iteration=400, act=dict(fun="genReLU", prm=[0.1, 0.15], trainable=True))
MC3_TRACES = {"mc3": MC3_TRACE, "mc3_slopes": MC3_SLOPES_TRACE}


def mc3_act(bn, cfg):
    a = cfg.get("act")
    if a is None:
        return {}
    return dict(actFun=bn.ActFun(fun=a["fun"], prm=np.array(a["prm"], dtype=float), trainable=a["trainable"]))


BLOCK_LAYOUTS = [
    # (n_features, n_nodes, size_output, indx_input_list, nodes_per_feature_list)  -- block_bnns.py:39-41,57-59,79-81
    (3, [6, 2], 2, [[0, 1, 2], [], []], [[2, 2, 2], [], []]),
    (3, [9, 6], 2, [[0, 1, 2], [0, 0, 0, 1, 1, 1, 2, 2, 2], []], [[3, 3, 3], [2, 2, 2], []]),
    (3, [9, 5], 2, [[0, 1, 1], [0, 0, 0, 1, 1, 1, 1, 1, 1], []], [[3, 6], [2, 3], []]),
    # config-5 layout: 512 features in 8 blocks of 64, 4 nodes per block
    (512, [32, 8], 1, [list(np.repeat(np.arange(8), 64)), [], []], [[4] * 8, [], []]),
]


# ---- G7: posterior prediction (get_posterior_cat_prob) ----------------------
def posterior_inputs(seed=77, n_rows=97, n_features=11, n_nodes=(6, 5), n_classes=4, n_samples=7, fun="tanh", bias=2):
    """S stored weight sets against one feature matrix.  genReLU runs carry a different slope vector per sample
    (trainable activations are logged per posterior sample, BNN_env.py:642-658)."""
    rs = np.random.default_rng(seed)
    x = rs.standard_normal((n_rows, n_features))
    shapes = layer_shapes(n_features, list(n_nodes), n_classes, bias)
    samples = []
    for i in range(n_samples):
        w = [rs.normal(0, 0.6, s) for s in shapes]
        alphas = rs.uniform(0.0, 0.3, len(n_nodes)) if fun == "genReLU" else np.zeros(1)
        samples.append(dict(weights=w, alphas=alphas, mcmc_it=100 * i))
    labels = rs.integers(0, n_classes, n_rows)
    return dict(x=x, samples=samples, labels=labels, fun=fun)


POSTERIOR_CASES = [dict(name="tanh", fun="tanh", seed=77), dict(name="genrelu", fun="genReLU", seed=78),
                   dict(name="swish_bias3", fun="swish", seed=79, bias=3)]


# ---- G8: get_data / randomize_data (split indices of seeded example tables) -------------------------------------------
SPLIT_CASES = [
    # name, keyword arguments of get_data beyond the two file names
    ("stratified", dict(seed=1234, testsize=0.2, all_class_in_testset=1, instance_id=1, header=1)),
    ("tail", dict(seed=77, testsize=0.1, all_class_in_testset=0, instance_id=1, header=1)),
    ("fold2", dict(seed=5, testsize=0.2, all_class_in_testset=1, instance_id=1, header=1, cv=2)),
    ("all_train", dict(seed=1234, testsize=0, instance_id=1, header=1)),
    ("in_order", dict(seed=9, testsize=0.2, randomize_order=False, instance_id=1, header=1)),
    ("batch", dict(seed=3, testsize=0.1, batch_training=25, instance_id=1, header=1, feature_indx=[0, 2, 5])),
    ("regression", dict(seed=11, testsize=0.2, all_class_in_testset=0, instance_id=1, header=1, label_mode="regression")),
]


def write_split_tables(directory):
    """A 50-row feature table (header line, instance names in the first column), a class-label table (5 named classes) and a
    two-column real-valued target table, in the layout of upstream's example files.  Returns their three paths."""
    import os
    rs = np.random.default_rng(2024)
    n, f = 50, 7
    names = np.array(["inst%02d" % i for i in range(n)])
    x = np.round(rs.standard_normal((n, f)), 5)
    classes = np.array(["0", "1", "4", "6", "9"])[rs.integers(0, 5, n)]
    targets = np.round(rs.standard_normal((n, 2)), 5)
    paths = [os.path.join(directory, name) for name in ("split_features.txt", "split_labels.txt", "split_targets.txt")]
    with open(paths[0], "w") as fh:
        fh.write("id\t" + "\t".join("f%d" % j for j in range(f)) + "\n")
        for i in range(n):
            fh.write(names[i] + "\t" + "\t".join(repr(float(v)) for v in x[i]) + "\n")
    with open(paths[1], "w") as fh:
        fh.write("id\tlabel\n")
        for i in range(n):
            fh.write("%s\t%s\n" % (names[i], classes[i]))
    with open(paths[2], "w") as fh:
        fh.write("id\ty0\ty1\n")
        for i in range(n):
            fh.write("%s\t%r\t%r\n" % (names[i], float(targets[i, 0]), float(targets[i, 1])))
    return paths
