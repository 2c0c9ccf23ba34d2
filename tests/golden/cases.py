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
def classification_data(seed, n_rows, n_features, n_classes, n_test=0):
    """Learnable synthetic classification problem."""
    rs = np.random.default_rng(seed)
    x = rs.standard_normal((n_rows + n_test, n_features))
    proj = rs.standard_normal((n_features, n_classes)) / np.sqrt(n_features)
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

MC3_TRACE = dict(seed=105, n_rows=1000, n_features=32, n_classes=4, n_test=100, n_nodes=[5, 5],
                 bias=-1, n_chains=4, swap_frequency=20, n_iteration=600)

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


def pdp_inputs():
    """Feature matrix for the partial-dependence vectors: column 0 continuous, column 3 ordinal, columns 8-10 one-hot."""
    inp = posterior_inputs(**{k: v for k, v in POSTERIOR_CASES[0].items() if k != "name"})
    xp = inp["x"].copy()
    xp[:, 3] = np.round(np.abs(xp[:, 3]) * 2)
    xp[:, 8:11] = np.eye(3)[np.random.default_rng(3).integers(0, 3, len(xp))]
    return inp, xp


PDP_FOCAL = (("cont", [0]), ("ord", [3]), ("ohe", [8, 9, 10]))


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
