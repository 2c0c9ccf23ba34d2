"""TEST INFRASTRUCTURE ONLY — float64 numpy restatement of npBNN's MCMC hot path.

This is the parity oracle and the timed CPU baseline ("port").  It is NOT part
of the product: ``npbnn_amd`` never imports it (see ``oracle/__init__.py``).

Every function cites the reference lines (relative to the upstream repository
root, np_bnn 0.1.23) whose behaviour it restates.  The restatement is
op-for-op where floating-point results depend on the order of operations
(``np.dot`` per layer, bias = column 0 of W, exp-form tanh,
``scipy.special.softmax``, gather-log-sum likelihood), so that float64 traces
agree bit-for-bit with the reference on the same numpy/scipy build.

Parity status: PINNED by tests/golden/*.npz (generated from the reference by
tests/golden/make_golden.py; checked in tests/test_oracle_golden.py).
"""
from types import SimpleNamespace

import numpy as np
import scipy.special
import scipy.stats

__all__ = [
    "Act", "activate", "dense", "hidden_layer", "forward", "forward_logits",
    "out_softmax", "out_identity", "out_regress_error", "softplus",
    "lik_categorical", "lik_gaussian", "lik_gaussian_error", "lik_poisson",
    "lik_negbin", "lik_negbin2d", "lik_negbin_base10", "lik_gamma",
    "closed_gaussian_empirical", "closed_poisson", "closed_negbin",
    "acc_classification", "label_acc_classification", "label_freq",
    "confusion_counts", "mse_all", "mse_per_column",
    "init_weights", "block_mask", "log_prior",
    "propose_normal", "propose_normal_1d", "propose_fixed_normal",
    "propose_normal_normalized", "propose_multiplier_vector", "propose_binomial",
    "gibbs_prior_scales", "gibbs_step",
    "make_chain", "mh_step", "run_chain", "mc3_make", "mc3_run",
    "posterior_cat_prob", "sample_from_categorical", "feature_importance",
]


# --------------------------------------------------------------------------
# activations                                   (np_bnn/BNN_lib.py:50-94)
# --------------------------------------------------------------------------
class Act:
    """Activation spec, mirrors ActFun's selection rules (BNN_lib.py:68-94).

    The reference picks ``activate`` with a chain of independent ``if``s
    (:74-81): "ReLU" -> relu; "genReLU" *or* trainable -> leaky relu; "swish";
    "tanh" (later matches override earlier ones).  ``eval`` (:83-87) passes the
    per-layer slope only when ``fun == "genReLU"``, otherwise 0.
    """

    def __init__(self, fun="ReLU", prm=None, trainable=False):
        self.fun = fun
        self.prm = np.zeros(1) if prm is None else prm
        self.acc_prm = self.prm
        self.trainable = trainable
        kind = None
        if fun == "ReLU":
            kind = "relu"
        if fun == "genReLU" or trainable is True:
            kind = "leaky"
        if fun == "swish":
            kind = "swish"
        if fun == "tanh":
            kind = "tanh"
        if kind is None:
            raise AttributeError("no activation selected for fun=%r" % (fun,))
        self.kind = kind

    def slope(self, layer_n):
        return self.prm[layer_n] if self.fun == "genReLU" else 0


def activate(z, act, layer_n):
    """Elementwise activation of a pre-activation matrix (BNN_lib.py:50-66, 83-87).

    relu / leaky mutate ``z`` in place like the reference does.
    """
    k = act.kind
    if k == "relu":
        z[z < 0] = 0                                   # :51
        return z
    if k == "leaky":
        p = act.slope(layer_n)
        z[z < 0] = p * z[z < 0]                        # :55
        return z
    if k == "swish":
        return z * (1 + np.exp(-z)) ** (-1)            # :60
    if k == "tanh":
        return 1.0 - (2.0 / (np.exp(2.0 * z) + 1.0))   # :65
    raise ValueError(k)


# --------------------------------------------------------------------------
# layer GEMM + forward                          (np_bnn/BNN_lib.py:154-193, 245-272)
# --------------------------------------------------------------------------
def dense(x, w):
    """x (N x in) times w (out x in[+1]); bias is column 0 of w when present
    (MatrixMultiplicationD, BNN_lib.py:154-162; same maths as the dead einsum
    variant MatrixMultiplication :146-152)."""
    wt = w.T
    if x.shape[1] == wt.shape[0]:
        return np.dot(x, wt)
    z = np.dot(x, wt[1:, ])
    z += wt[0, ]
    return z


def _override_columns(data, col_indicators, col_means):
    """data_transform_obj.transform (BNN_env.py:14-17): columns whose indicator
    is 0 are overwritten by the column mean."""
    d = data + 0
    d[:, col_indicators == 0] = col_means[col_indicators == 0]
    return d


def hidden_layer(z0, w, act, layer_n, col_override=None):
    """RunHiddenLayer (BNN_lib.py:184-193).  ``act`` None/False = identity."""
    if col_override is not None:
        z0 = _override_columns(z0, *col_override)
    z1 = dense(z0, w)
    if act:
        return activate(z1, act, layer_n)
    return z1


def forward_logits(data, weights, act, indicators=None, col_override=None):
    """Pre-output activations of the last layer (RunPredict / RunPredictInd
    without the output function, BNN_lib.py:245-272)."""
    if col_override is None:
        tmp = data + 0                                  # :248 (full copy)
    else:
        tmp = _override_columns(data, *col_override)    # :250
    n = len(weights)
    for i in range(n - 1):
        w = weights[i]
        if i == 0 and indicators is not None:
            w = w * indicators                          # :266
        tmp = hidden_layer(tmp, w, act, i)
    return hidden_layer(tmp, weights[n - 1], None, n - 1)


def forward(data, weights, act, out_fn, indicators=None, col_override=None):
    """RunPredict (BNN_lib.py:245-256) / RunPredictInd (:258-272)."""
    return out_fn(forward_logits(data, weights, act, indicators, col_override))


# --------------------------------------------------------------------------
# posterior prediction                          (np_bnn/BNN_lib.py:352-397, 682-713)
# --------------------------------------------------------------------------
def sample_from_categorical(post_probs):
    """sample_from_categorical (BNN_lib.py:682-713): per instance, one categorical draw per posterior sample
    (np.random.random, global stream); point estimate = class frequencies of the draws."""
    n_samples, n_instances, n_classes = post_probs.shape
    res = np.zeros((n_instances, n_samples))
    point = np.zeros((n_instances, n_classes))
    for j in range(n_instances):
        p = np.cumsum(post_probs[:, j, :], axis=1)                   # :696
        r = np.random.random(len(p))                                 # :697
        q = p - r.reshape(len(r), 1)
        q[q < 0] = 1                                                 # :699
        cls = np.argmin(q, axis=1)                                   # :700
        res[j, :] = cls
        counts = np.bincount(cls, minlength=n_classes)
        point[j, :] = counts / np.sum(counts)                        # :704
    class_counts = np.zeros((n_samples, n_classes))
    for i in range(res.shape[1]):
        class_counts[i] = np.bincount(res[:, i].astype(int), minlength=n_classes)
    return dict(predictions=point, class_counts=class_counts, post_predictions=res)


def posterior_cat_prob(features, post_samples, act, out_fn, summary_mode=0, feature_index_to_shuffle=None,
                       unlink_features_within_block=False):
    """get_posterior_cat_prob (BNN_lib.py:352-397): every stored weight set (with its own activation slopes) is
    run over the - optionally column-shuffled - feature matrix; summary 0 = frequency of the arg-max class over the
    samples, 1 = mean class probabilities, 2 = posterior-predictive resampling."""
    x = features.copy()                                              # :364
    if feature_index_to_shuffle:                                     # :366-371 (np.random.permutation, global stream)
        if unlink_features_within_block and type(feature_index_to_shuffle) == list:
            for fi in feature_index_to_shuffle:
                x[:, fi] = np.random.permutation(x[:, fi])
        else:
            x[:, feature_index_to_shuffle] = np.random.permutation(x[:, feature_index_to_shuffle])
    probs = []
    for smp in post_samples:                                         # :376-380
        a = Act(act.fun, smp["alphas"], act.trainable)
        probs.append(forward(x, smp["weights"], a, out_fn))
    probs = np.array(probs)
    if summary_mode == 0:                                            # :382-390
        calls = np.argmax(probs, axis=2).T
        n_s, n_i, n_c = probs.shape
        summary = np.zeros((n_i, n_c))
        for i, row in enumerate(calls):
            cls, cnt = np.unique(row, return_counts=True)
            summary[i, cls] = cnt
        summary = summary / n_s
    elif summary_mode == 1:                                          # :391-392
        summary = np.mean(probs, axis=0)
    else:                                                            # :393-395
        summary = sample_from_categorical(probs)["predictions"]
    return probs, summary


def feature_importance(features, post_samples, act, out_fn, true_labels, n_permutations=100, feature_blocks=None,
                       summary_mode=0, unlink_features_within_block=True, feature_names=()):
    """feature_importance (BNN_lib.py:504-597): accuracy of the posterior prediction with all features, then with each
    feature (block) shuffled between the instances, ``n_permutations`` times each.  Returns (block order, names, table)
    sorted by decreasing mean accuracy loss like the reference's data frame; table columns: delta_acc mean / std,
    accuracy-with-feature-randomised mean / std."""
    idx_all = np.arange(features.shape[1])
    names = list(feature_names) if len(feature_names) else list(idx_all.astype(str))
    if isinstance(feature_blocks, dict) and len(feature_blocks):                        # :525-531
        blocks, block_names = list(feature_blocks.values()), list(feature_blocks.keys())
    elif feature_blocks is None or isinstance(feature_blocks, dict):                    # :532-534
        blocks, block_names = [[i] for i in idx_all], names
    else:                                                                               # :536-538
        blocks, block_names = feature_blocks, ["block_" + str(i) for i in range(len(feature_blocks))]
    _, pred = posterior_cat_prob(features, post_samples, act, out_fn, summary_mode)     # :546-549
    ref_acc = acc_classification(pred, true_labels)
    acc = []
    for block in blocks:                                                                # :555-569
        row = []
        for _ in range(n_permutations):
            _, pred = posterior_cat_prob(features, post_samples, act, out_fn, summary_mode, feature_index_to_shuffle=block,
                                         unlink_features_within_block=unlink_features_within_block)
            row.append(acc_classification(pred, true_labels))
        acc.append(row)
    acc = np.array(acc)
    delta = ref_acc - acc                                                               # :571
    table = np.stack([np.mean(delta, axis=1), np.std(delta, axis=1), np.mean(acc, axis=1), np.std(acc, axis=1)], axis=1)
    # DataFrame.sort_values('delta_acc_mean', ascending=False) (:583): quicksort on the reversed column, reversed back
    key = table[:, 0]
    order = np.arange(len(key))[::-1][key[::-1].argsort(kind="quicksort")][::-1]
    return order, [block_names[i] for i in order], table[order]


# --------------------------------------------------------------------------
# output functions                              (np_bnn/BNN_lib.py:166-182)
# --------------------------------------------------------------------------
def out_softmax(z):
    return scipy.special.softmax(z, axis=1)             # :168


def softplus(z):
    return np.logaddexp(0, z)                           # :172


def out_identity(z):
    return z                                            # :174-175


def out_regress_error(z, ind=None):
    """RegressTransformError (:177-182): softplus on the second half of the
    columns, in place."""
    if ind is None:
        ind = int(z.shape[1] / 2)
    z[:, ind:] = softplus(z[:, ind:])
    return z


# --------------------------------------------------------------------------
# likelihoods                 (np_bnn/BNN_lib.py:100-143, np_bnn/BNN_lik.py:5-78)
# --------------------------------------------------------------------------
def lik_categorical(prediction, labels, sample_id, class_weight=(), instance_weight=None,
                    lik_temp=1, sig2=0):
    """calc_likelihood (BNN_lib.py:100-121).  The class-weight + instance-weight
    branch (:105-106) is broken upstream (axis=1 on a 1-D array) and is
    reproduced as the same failure."""
    picked = np.log(prediction[sample_id, labels])
    if len(class_weight):
        if instance_weight is not None:
            lik_tmp = np.sum(picked * class_weight[labels], axis=1)   # raises AxisError
            return lik_temp * (lik_tmp * instance_weight)
        return lik_temp * np.sum(picked * class_weight[labels])      # :108
    if instance_weight is not None:
        return lik_temp * (np.sum(picked * instance_weight))         # :119
    return lik_temp * np.sum(picked)                                 # :121


def lik_gaussian(prediction, true_values, _=None, class_weight=None, instance_weight=None,
                 lik_temp=1, sig2=1):
    """calc_likelihood_regression (BNN_lib.py:123-131); ``sig2`` is a std."""
    if instance_weight is not None:
        raise SystemExit("instance_weight not implemented for regression")
    return lik_temp * np.sum(scipy.stats.norm.logpdf(true_values, prediction, sig2))


def lik_gaussian_error(prediction, true_values, _=None, class_weight=None,
                       instance_weight=None, lik_temp=1, sig2=0):
    """calc_likelihood_regression_error (BNN_lib.py:134-143)."""
    if instance_weight is not None:
        raise SystemExit("instance_weight not implemented for regression")
    k = true_values.shape[1]
    return lik_temp * np.sum(scipy.stats.norm.logpdf(true_values, prediction[:, :k], prediction[:, k:]))


def lik_poisson(prediction, true_values, sample_id=None, class_weight=None,
                instance_weight=None, lik_temp=1, sig2=0):
    """poi_likelihood (BNN_lik.py:5-14)."""
    rate = np.exp(prediction[:, 0])
    return np.sum(scipy.stats.poisson.logpmf(true_values[:, 0], rate))


def lik_negbin(prediction, true_values, sample_id=None, class_weight=None,
               instance_weight=None, lik_temp=1, sig2=0):
    """negbin_likelihood (BNN_lik.py:16-30)."""
    mean = np.exp(prediction[:, 0])
    p = 1 / (1 + np.exp(-prediction[:, 1]))
    n = p * mean / (1 - p)
    return np.sum(scipy.stats.nbinom.logpmf(true_values[:, 0], n=n, p=p))


def lik_negbin2d(prediction, true_values, sample_id=None, class_weight=None,
                 instance_weight=None, lik_temp=1, sig2=0):
    """negbin_likelihood2d (BNN_lik.py:33-49)."""
    k = true_values.shape[1]
    mean = np.exp(prediction[:, :k])
    p = 1 / (1 + np.exp(-prediction[:, k:]))
    n = p * mean / (1 - p)
    return np.sum(scipy.stats.nbinom.logpmf(true_values, n=n, p=p))


def lik_negbin_base10(prediction, true_values, sample_id=None, class_weight=None,
                      instance_weight=None, lik_temp=1, sig2=0):
    """negbin_likelihood_base10 (BNN_lik.py:55-66)."""
    mean = 10 ** (prediction[:, 0])
    p = 1 / (1 + 10 ** (-prediction[:, 1]))
    n = p * mean / (1 - p)
    return np.sum(scipy.stats.nbinom.logpmf(true_values[:, 0], n=n, p=p))


def lik_gamma(prediction, true_values, sample_id=None, class_weight=None,
              instance_weight=None, lik_temp=1, sig2=0):
    """gamma_likelihood (BNN_lik.py:68-78), bug-for-bug: ``b`` is passed as
    scipy's *loc* and an (N,1) target broadcasts against (N,) to N x N."""
    a = np.exp(prediction[:, 0])
    b = np.exp(prediction[:, 1])
    return np.sum(scipy.stats.gamma.logpdf(true_values, a, b))


# closed forms the HIP kernels implement (checked against the scipy forms above)
def closed_gaussian_empirical(prediction, true_values, lik_temp=1):
    """Gaussian log-lik with the empirical per-column sigma of BNN_env.py:475-476
    (population std of the residuals) as a function of the column moments
    S1 = sum r, S2 = sum r^2."""
    r = true_values - prediction[:, :true_values.shape[1]]
    n = r.shape[0]
    s1 = np.sum(r, axis=0)
    s2 = np.sum(r * r, axis=0)
    var = s2 / n - (s1 / n) ** 2
    sig = np.sqrt(var)
    ll = np.sum(-n * (0.5 * np.log(2 * np.pi) + np.log(sig)) - s2 / (2 * var))
    return lik_temp * ll, sig


def closed_poisson(eta, k):
    return np.sum(k * eta - np.exp(eta) - scipy.special.gammaln(k + 1))


def closed_negbin(log_mean, logit_p, k, base10=False):
    if base10:
        mean = 10.0 ** log_mean
        p = 1 / (1 + 10.0 ** (-logit_p))
    else:
        mean = np.exp(log_mean)
        p = 1 / (1 + np.exp(-logit_p))
    n = p * mean / (1 - p)
    g = scipy.special.gammaln
    return np.sum(g(k + n) - g(k + 1) - g(n) + n * np.log(p) + k * np.log1p(-p))


# --------------------------------------------------------------------------
# accuracy statistics                           (np_bnn/BNN_lib.py:195-233)
# --------------------------------------------------------------------------
def acc_classification(y, lab):
    pred = np.argmax(y, axis=1)                         # :207
    return np.sum(pred == lab) / len(pred)              # :208


def label_acc_classification(y, lab):
    pred = np.argmax(y, axis=1)                         # :212
    out = []
    for label in np.unique(lab):                        # :214-218
        sel = lab == label
        out.append(np.sum(pred[sel] == lab[sel]) / len(pred[sel]))
    return np.array(out)


def label_freq(y):
    pred = np.argmax(y, axis=1)                         # :229
    f = np.zeros(y.shape[1])
    idx, cnt = np.unique(pred, return_counts=True)
    f[idx] = cnt
    return f / len(pred)                                # :233


def confusion_counts(y, lab, n_classes=None):
    """C x C matrix [true label, argmax prediction]; the three statistics above
    are all functions of it."""
    c = y.shape[1] if n_classes is None else n_classes
    pred = np.argmax(y, axis=1)
    m = np.zeros((c, c), dtype=np.int64)
    np.add.at(m, (lab, pred), 1)
    return m


def mse_all(y, lab):
    return np.mean((y[:, 0:lab.shape[1]] - lab) ** 2)           # :196


def mse_per_column(y, lab):
    return np.mean((y[:, 0:lab.shape[1]] - lab) ** 2, axis=0)   # :200


# --------------------------------------------------------------------------
# model state helpers     (np_bnn/BNN_mcmc.py:9-25, BNN_lib.py:16-47, BNN_env.py:180-194)
# --------------------------------------------------------------------------
def init_weights(n_nodes, n_features, size_output, init_std=0.1, bias_node=0):
    """init_weight_prm (BNN_mcmc.py:9-25): draws from numpy's *global* RNG;
    W_l is (out_l x in_l[+1]) with the bias in column 0."""
    bn = 1 if bias_node >= 1 else 0
    bn2 = 1 if bias_node >= 2 else 0
    bn3 = 1 if (bias_node == 3 or bias_node == -1) else 0
    n_layers = len(n_nodes) + 1
    w = [np.random.normal(0, init_std, (n_nodes[0], n_features + bn))]
    for i in range(1, n_layers - 1):
        w.append(np.random.normal(0, init_std, (n_nodes[i], n_nodes[i - 1] + bn2)))
    w.append(np.random.normal(0, init_std, (size_output, n_nodes[-1] + bn3)))
    return w


def block_mask(w_layers, indx_input_list, nodes_per_feature_list):
    """create_mask (BNN_lib.py:16-47): 0/1 masks; for layer l, consecutive input
    columns sharing a group id are wired to that group's block of rows."""
    masks = []
    for li, w in enumerate(w_layers):
        groups = indx_input_list[li]
        per_group = nodes_per_feature_list[li]
        if len(groups) == 0:
            masks.append(np.ones(w.shape))
            continue
        m = np.zeros(w.shape)
        row0 = 0          # first row of the current group's block
        next_row = 0      # one past the last row used so far
        g = 0
        for col in range(len(groups)):
            if col > 0 and groups[col] != groups[col - 1]:
                g += 1
                row0 = next_row
            rows = np.arange(per_group[g]) + row0
            m[rows, col] = 1
            next_row = np.max(rows) + 1
        masks.append(m)
    return masks


_PRIOR_LOGPDF = {1: scipy.stats.norm.logpdf, 2: scipy.stats.cauchy.logpdf,
                 3: scipy.stats.laplace.logpdf}


def log_prior(weights, prior_kind, prior_scale, indicators=None, freq_indicator=0,
              prior_ind1=0.5, n_indicators=None):
    """npBNN.calc_prior (BNN_env.py:180-194).  prior_kind 0 = uniform (0);
    1 normal, 2 Cauchy, 3 Laplace; anything else falls back to normal (:148-150)."""
    lp = 0
    if prior_kind != 0:
        f = _PRIOR_LOGPDF.get(prior_kind, scipy.stats.norm.logpdf)
        for i in range(len(weights)):
            lp += np.sum(f(weights[i], 0, scale=prior_scale[i]))
    if freq_indicator:
        s = np.sum(indicators)
        lp += s * np.log(prior_ind1) + (n_indicators - s) * np.log(1 - prior_ind1)
    return lp


# --------------------------------------------------------------------------
# proposals                                     (np_bnn/BNN_mcmc.py:27-123)
# --------------------------------------------------------------------------
def _reflect(z, Mb, mb):
    z[z > Mb] = Mb - (z[z > Mb] - Mb)
    z[z < mb] = mb + (mb - z[z < mb])
    return z


def propose_normal(w, d, n, Mb, mb, rs):
    """UpdateNormal (BNN_mcmc.py:57-69).  Draw order: integers, integers,
    normal.  Duplicate (Ix,Iy) pairs: the last write wins."""
    w = np.array(w)
    ix = rs.integers(0, w.shape[0], n)
    iy = rs.integers(0, w.shape[1], n)
    z = np.zeros(w.shape) + w
    z[ix, iy] = z[ix, iy] + rs.normal(0, d[ix, iy], n)
    return _reflect(z, Mb, mb), (ix, iy), 0


def propose_normal_1d(v, d, n, Mb, mb, rs):
    """UpdateNormal1D (BNN_mcmc.py:44-55)."""
    v = np.array(v)
    ix = rs.integers(0, len(v), n)
    z = np.zeros(v.shape) + v
    z[ix] = z[ix] + rs.normal(0, d, n)
    return _reflect(z, Mb, mb), ix, 0


def propose_fixed_normal(w, d, n, Mb, mb, rs):
    """UpdateFixedNormal (BNN_mcmc.py:27-42): independence proposal N(0,d) with
    its Hastings ratio."""
    ix = rs.integers(0, w.shape[0], n)
    iy = rs.integers(0, w.shape[1], n)
    cur = w[ix, iy]
    new = rs.normal(0, d[ix, iy], n)
    h = np.sum(scipy.stats.norm.logpdf(cur, 0, d[ix, iy]) - scipy.stats.norm.logpdf(new, 0, d[ix, iy]))
    z = np.zeros(w.shape) + w
    z[ix, iy] = new
    return _reflect(z, Mb, mb), (ix, iy), h


def propose_normal_normalized(w, d, n, Mb, mb, rs):
    """UpdateNormalNormalized (BNN_mcmc.py:71-82)."""
    w = np.array(w)
    ix = rs.integers(0, w.shape[0], n)
    iy = rs.integers(0, w.shape[1], n)
    z = np.zeros(w.shape) + w
    z[ix, iy] = z[ix, iy] + rs.normal(0, d[ix, iy], n)
    return z / np.sum(z), (ix, iy), 0


def propose_multiplier_vector(q, d, f, rs):
    """multiplier_proposal_vector (BNN_mcmc.py:101-113)."""
    shape = q.shape
    ff = rs.binomial(1, f, shape)
    u = rs.random(shape)
    lam = 2 * np.log(d)
    m = np.exp(lam * (u - .5))
    m[ff == 0] = 1.
    return q * m, 0, np.sum(np.log(m))


# --------------------------------------------------------------------------
# one Metropolis-Hastings chain    (np_bnn/BNN_env.py:19-173, 274-379, 381-532)
# --------------------------------------------------------------------------
def make_chain(data, labels, n_nodes, *, act=None, use_bias_node=1, prior_kind=1, p_scale=1,
               w_bound=np.inf, mode="classification", empirical_error=False, mask=None,
               init_w=None, instance_weights=None, class_weights=(), test_data=None,
               test_labels=None,
               update_f=None, update_ws=None, temperature=1, n_iteration=100000,
               likelihood_tempering=1, mcmc_id=0, randomize_seed=False, adapt_f=0, adapt_fM=1,
               adapt_freq=1000, adapt_stop=None, estimate_error=True, likelihood_f=None,
               with_stats=True, hyper_p=0, freq_indicator=0, prior_ind1=0.5, feature_indicators=False,
               update_function=None, out_fn=None, size_output=None, accuracy_f=None):
    """State of one chain = what npBNN.__init__ (BNN_env.py:19-173) and
    MCMC.__init__ (:274-379) set up, every sampler option included: weight
    indicators (``freq_indicator``, ``prior_ind1``), feature indicators,
    trainable activation slopes (``act.trainable``), hyper-prior scales
    (``hyper_p``, changed by :func:`gibbs_step`), the proposal function.
    Weights are drawn from numpy's global RNG exactly where the reference
    draws them (:111-115).
    """
    st = SimpleNamespace()
    st.act = act if act is not None else Act()
    st.data = data
    st.mode = mode
    st.labels = labels.astype(int) if mode == "classification" else labels
    st.test_data = test_data
    st.test_labels = test_labels
    st.error_prm = []
    if mode == "classification":
        st.size_output = len(np.unique(st.labels))
        st.out_fn = out_softmax
        st.lik = lik_categorical
    elif mode == "regression":
        st.size_output = st.labels.shape[1]
        st.out_fn = out_identity
        st.error_prm = np.ones(st.size_output)
        st.lik = lik_gaussian
    elif mode == "regression-error":
        st.size_output = st.labels.shape[1] * 2
        st.out_fn = out_regress_error
        st.lik = lik_gaussian_error
    elif mode == "custom":                                      # :72-74
        st.size_output = size_output
        st.out_fn = out_identity
        st.lik = None
    else:
        raise ValueError(mode)
    if out_fn is not None and mode != "classification":         # :57-60
        st.out_fn = out_fn
    if likelihood_f is not None:
        st.lik = likelihood_f
    st.accuracy_f = accuracy_f
    st.empirical_error = empirical_error
    st.n_layers = len(n_nodes) + 1
    st.sample_id = np.arange(data.shape[0])
    st.w_bound = p_scale if prior_kind == 0 else w_bound       # :135-137
    st.prior_kind = prior_kind
    st.prior_scale = np.ones(st.n_layers) * p_scale            # :154
    st.hyper_p = hyper_p
    st.freq_indicator = freq_indicator
    st.prior_ind1 = prior_ind1
    st.propose = propose_normal if update_function is None else update_function   # :278
    st.class_w = class_weights
    st.instance_weights = instance_weights
    st.mask = None
    if init_w is None:
        st.w = init_weights(n_nodes, data.shape[1], st.size_output, init_std=0.1,
                            bias_node=use_bias_node)           # :111-115 (init_std fixed 0.1)
    else:
        st.w = init_w
    if mask is not None:                                        # apply_mask :259-262
        st.mask = mask
        st.w = [st.w[i] * mask[i] for i in range(st.n_layers)]
    st.n_params = int(np.sum([np.size(i) for i in st.w]))
    if st.act.trainable:
        st.n_params += st.n_layers                              # :166-167
    st.indicators = np.ones(st.w[0].shape)                      # :125
    st.feature_ind = st.feature_means = None
    if feature_indicators:                                      # :170-172
        st.feature_ind = np.ones(data.shape[1]).astype(int)
        st.feature_means = np.mean(data, axis=0)
    st.test_override = None
    # ---- MCMC.__init__ ----
    if update_ws is None:
        update_ws = [0.075] * st.n_layers
    if update_f is None:
        update_f = [0.05] * st.n_layers
    st.update_f = update_f[0:st.n_layers]
    st.update_ws = [np.ones(st.w[i].shape) * update_ws[i] for i in range(st.n_layers)]
    st.update_n = np.array([np.max([1, np.round(st.w[i].size * update_f[i]).astype(int)])
                            for i in range(st.n_layers)])      # :292-293
    st.temperature = temperature
    st.n_iterations = n_iteration
    st.it = 0
    st.lik_temp = likelihood_tempering
    st.with_stats = with_stats
    st.y = forward(st.data, st.w, st.act, st.out_fn)            # :299
    st.logLik = st.lik(st.y, st.labels, st.sample_id, class_weight=st.class_w,
                       instance_weight=st.instance_weights, lik_temp=st.lik_temp,
                       sig2=st.error_prm)                       # :313-319
    st.logPrior = _chain_prior(st, st.w, st.indicators)
    st.logPost = st.logLik + st.logPrior
    _refresh_stats(st)
    st.last_accepted = 1
    st.accepted_mem = [1]
    st.acceptance_rate = 0.
    st.mcmc_id = mcmc_id
    st.randomize_seed = randomize_seed
    st.rs = np.random.default_rng(1234)                         # :362
    st.freq_layer_update = np.ones(st.n_layers)
    st.adapt_f, st.adapt_fM, st.adapt_freq = adapt_f, adapt_fM, adapt_freq
    st.adapt_stop = int(n_iteration * 0.05) if adapt_stop is None else adapt_stop
    st.max_n = np.array([st.w[i].size for i in range(st.n_layers)]).astype(int)
    st.estimate_error = np.min([20000, 0.1 * n_iteration]) if estimate_error else n_iteration
    st.trace = None
    return st


def _refresh_stats(st):
    """Accuracy bookkeeping done at init (:344-353) and on accept (:508-518)."""
    if not st.with_stats:
        return
    if st.accuracy_f is not None:
        acc_f, lab_f = st.accuracy_f, (lambda y, lab: np.ones(1))
    elif st.mode == "classification":
        acc_f, lab_f = acc_classification, label_acc_classification
    elif st.mode in ("regression", "regression-error"):
        acc_f, lab_f = mse_all, mse_per_column
    else:
        acc_f, lab_f = (lambda y, lab: 1.0), (lambda y, lab: np.ones(1))    # SkipAccuracy*, BNN_lib.py:235-239
    st.accuracy = acc_f(st.y, st.labels)
    st.label_acc = lab_f(st.y, st.labels)
    st.label_freq = label_freq(st.y)
    if st.test_data is not None and len(st.test_data) > 0:
        st.y_test = forward(st.test_data, st.w, st.act, st.out_fn, indicators=st.indicators,
                            col_override=st.test_override)     # RunPredictInd :347-349, :512-515
        st.test_accuracy = acc_f(st.y_test, st.test_labels)
    else:
        st.y_test, st.test_accuracy = [], 0


def _chain_prior(st, weights, indicators):
    return log_prior(weights, st.prior_kind, st.prior_scale, indicators=indicators,
                     freq_indicator=st.freq_indicator, prior_ind1=st.prior_ind1,
                     n_indicators=st.indicators.size)


def propose_binomial(ind, update_f, shape_out):
    """UpdateBinomial (BNN_mcmc.py:98-99): numpy's GLOBAL stream, not the chain's."""
    return np.abs(ind - np.random.binomial(1, np.random.random() * update_f, shape_out))


def gibbs_prior_scales(w_layers, hyper_p):
    """npBNN.sample_prior_scale (BNN_env.py:196-221) with the conjugate draws of
    BNN_mcmc.py:126-144: the standard deviation of a zero-mean normal under a
    gamma prior on its precision - one per layer (a=2), per input node (a=1) or
    per weight (a=1.5), b=0.1; numpy's global stream."""
    out = []
    for x in w_layers:
        if hyper_p == 1:
            v = x.flatten()
            tau = np.random.gamma(2 + len(v) / 2., scale=1. / (0.1 + np.sum(v ** 2) / 2.))
        elif hyper_p == 2:
            tau = np.random.gamma(1 + x.shape[0] / 2., scale=1. / (0.1 + np.sum(x ** 2, axis=0) / 2.))
        else:
            tau = np.random.gamma(1.5 + 1 / 2., scale=1. / (0.1 + (x ** 2) / 2.))
        out.append(1 / np.sqrt(tau))
    return out


def gibbs_step(st):
    """MCMC.gibbs_step (BNN_env.py:534-538): new prior scales, the prior of
    the current state under them (without the slope term), iteration + 1."""
    if st.hyper_p in (1, 2, 3):
        st.prior_scale = gibbs_prior_scales(st.w, st.hyper_p)
    st.logPrior = _chain_prior(st, st.w, st.indicators)
    st.logPost = st.logLik + st.logPrior
    st.it += 1


def mh_step(st):
    """One Metropolis-Hastings iteration, MCMC.mh_step (BNN_env.py:381-532).
    Returns a dict describing the proposal."""
    if st.randomize_seed:
        st.rs = np.random.default_rng(st.it + st.mcmc_id)       # :383-384
    hastings = 0
    additional_prob = 0
    w_prime = []
    tmp = st.data + 0                                            # :388
    # ---- adaptation :392-413 ----
    if st.it % st.adapt_freq == 0 and st.it < st.adapt_stop:
        if st.acceptance_rate < st.adapt_f:
            st.freq_layer_update = st.freq_layer_update * 0.8
            st.update_f = np.array(st.update_f) * .85
            n = (st.max_n * st.update_f).astype(int)
            n[n < 1] = 1
            st.update_n = n
            st.update_ws = [i * 0.9 for i in st.update_ws]
        if st.acceptance_rate > st.adapt_fM and np.sum(st.update_n) < st.n_params:
            st.update_f = np.exp(np.log(np.array(st.update_f)) * .85)
            n = (st.max_n * st.update_f).astype(int)
            n[n < 1] = 1
            st.update_n = n
            st.update_ws = [i * 1.2 for i in st.update_ws]
    # ---- trainable activation slopes :416-421 ----
    if st.act.trainable:
        prm_tmp, _, h = propose_normal_1d(st.act.acc_prm, d=0.05, n=1, Mb=1, mb=0, rs=st.rs)
        additional_prob += np.log(10) * -np.sum(prm_tmp) * 10
        hastings += h
        st.act.prm = prm_tmp
    # ---- feature indicators :424-433 ----
    col_override = None
    feature_ind_p = st.feature_ind
    if st.feature_ind is not None and st.it > st.adapt_stop:
        feature_ind_p = st.feature_ind + 0
        if st.rs.random() < 0.2:
            feature_ind_p = propose_binomial(feature_ind_p, 0.5, st.feature_ind.shape)
        col_override = (feature_ind_p, st.feature_means)
    # ---- regression error parameter :435-444 ----
    indicators_p = st.indicators + 0
    error_tmp = st.error_prm
    if st.mode == "regression" and st.it > st.estimate_error:
        if not st.empirical_error:
            error_tmp, _, h = propose_multiplier_vector(st.error_prm, d=1.1, f=0.5, rs=st.rs)
            hastings += h
            additional_prob += np.log(1) * -np.sum(error_tmp) * 1
    else:
        error_tmp = 1
    # ---- which layers :446-447 ----
    rr = st.rs.random(st.n_layers)
    rr[np.argmin(rr)] = 0
    for i in range(st.n_layers):                                 # :449-472
        if rr[i] >= st.freq_indicator or i > 0:
            if rr[i] < st.freq_layer_update[i]:
                upd, _, h = st.propose(st.w[i], d=st.update_ws[i], n=st.update_n[i],
                                       Mb=st.w_bound, mb=-st.w_bound, rs=st.rs)
                w_prime.append(upd)
                hastings += h
            else:
                w_prime.append(st.w[i] + 0)
        else:                                                    # :457-460
            w_prime.append(st.w[i] + 0)
            indicators_p = propose_binomial(st.indicators, st.update_f[3], st.indicators.shape)
        if st.mask is not None:
            w_prime[i] *= st.mask[i]
        act = st.act if i < st.n_layers - 1 else None
        if i == 0:                                               # :463-468
            tmp = hidden_layer(tmp, w_prime[i] * indicators_p, act, i, col_override)
        else:
            tmp = hidden_layer(tmp, w_prime[i], act, i)
    y_prime = st.out_fn(tmp)                                     # :473
    if st.mode == "regression" and st.empirical_error:
        error_tmp = np.std(y_prime - st.labels, axis=0)          # :475-476
    logPrior_p = _chain_prior(st, w_prime, indicators_p) + additional_prob
    logLik_p = st.lik(y_prime, st.labels, st.sample_id, class_weight=st.class_w,
                      instance_weight=st.instance_weights, lik_temp=st.lik_temp,
                      sig2=error_tmp)                            # :485-491
    logPost_p = logLik_p + logPrior_p
    log_u = np.log(st.rs.random())                               # :493
    accepted = bool((logPost_p - st.logPost) * st.temperature + hastings >= log_u)
    info = dict(logLik=float(logLik_p), logPrior=float(logPrior_p), hastings=float(hastings),
                log_u=float(log_u), accepted=accepted)
    if st.trace is not None:
        info["w_prime"] = w_prime
    if accepted:                                                 # :494-519
        st.w = w_prime
        st.indicators = indicators_p
        if st.feature_ind is not None:
            st.feature_ind = feature_ind_p + 0
        st.test_override = col_override                          # the test forward of this accept uses it (:512-515)
        if st.act.trainable:
            st.act.acc_prm = st.act.prm + 0                      # :502-503
        if st.mode == "regression":
            st.error_prm = error_tmp
        st.logPost, st.logLik, st.logPrior = logPost_p, logLik_p, logPrior_p
        st.y = y_prime
        _refresh_stats(st)
        st.last_accepted = 1
    else:
        st.last_accepted = 0
    st.accepted_mem.append(st.last_accepted)                     # :523-529
    st.acceptance_rate = np.mean(st.accepted_mem)
    if len(st.accepted_mem) > 100:
        st.accepted_mem = st.accepted_mem[-100:]
    st.it += 1
    if st.trace is not None:
        st.trace.append(info)
    return info


def run_chain(st, n_steps):
    for _ in range(n_steps):
        mh_step(st)
    return st


# --------------------------------------------------------------------------
# MC3                                           (np_bnn/BNN_mc3.py:8-126)
# --------------------------------------------------------------------------
def mc3_make(chain_factory, n_chains=4, swap_frequency=100, temperatures=None,
             min_temperature=0.8, n_iteration=100000):
    """MC3.__init__ (BNN_mc3.py:9-78).  ``chain_factory(i, temperature)`` must
    build chain i (MCMC kwargs of :61-75: n_iteration=swap_frequency,
    randomize_seed=True, mcmc_id=i, adapt_freq=50, adapt_f=0.1, adapt_fM=0.6,
    adapt_stop=1000).  The seed draw of :43 is replayed to keep numpy's global
    RNG stream aligned with the reference."""
    mc = SimpleNamespace()
    mc.n_chains = n_chains
    mc.swap_frequency = swap_frequency
    mc.n_mc3_iteration = np.round(n_iteration / swap_frequency).astype(int)   # :40
    mc.rseeds = np.random.choice(range(1000, 9999), n_chains, replace=False)  # :43
    if temperatures is None:
        temperatures = [1] if n_chains == 1 else np.linspace(min_temperature, 1, n_chains)
    mc.temperatures = temperatures
    mc.chains = [chain_factory(i, temperatures[i]) for i in range(n_chains)]
    mc.swaps = []
    return mc


def mc3_swap_decision(log_post, temps, n_chains):
    """The swap proposal of BNN_mc3.py:98-112 on gathered scalars; consumes
    numpy's global RNG exactly like the reference parent process."""
    j, k = np.random.choice(range(n_chains), 2, replace=False)                # :99-100
    tj, tk = temps[j] + 0, temps[k] + 0
    r = (log_post[k] - log_post[j]) * tj + (log_post[j] - log_post[k]) * tk   # :103-104
    log_u = np.log(np.random.random())                                        # :110
    return int(j), int(k), float(r), float(log_u), bool(r >= log_u)


def mc3_run(mc, n_mc3_iterations=None):
    """MC3.run_mcmc (BNN_mc3.py:87-126) without the process pool or logging:
    chains advance swap_frequency steps, then one temperature-swap proposal."""
    n = mc.n_mc3_iteration if n_mc3_iterations is None else n_mc3_iterations
    for _ in range(n):
        for st in mc.chains:
            run_chain(st, mc.swap_frequency)                                  # :80-85
        if mc.n_chains > 1:
            lp = [st.logPost for st in mc.chains]
            tt = [st.temperature for st in mc.chains]
            j, k, r, log_u, swapped = mc3_swap_decision(lp, tt, mc.n_chains)
            if swapped:
                mc.chains[j].temperature, mc.chains[k].temperature = tt[k], tt[j]
            mc.swaps.append((j, k, r, log_u, swapped))
    return mc


__all__.append("mc3_swap_decision")
