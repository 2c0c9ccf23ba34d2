"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle
and against the committed golden vectors of the reference.

Tolerances (float32 per-row arithmetic, float64 cross-row sums; BASELINE.json
asks for log-likelihood within 1e-4 relative):
  * last-layer values / predictions: |err| <= 2e-5 * max(1, |value|)
  * log-likelihood: relative error <= 2e-6
  * confusion counts: exact, except rows whose top-2 outputs are closer than
    1e-5 (an fp32 tie the float64 reference resolves differently)
"""
import os

import numpy as np
import pytest

import cases
import oracle as orc

pytestmark = pytest.mark.gpu

LL_RTOL = 2e-6
Z_TOL = 2e-5


@pytest.fixture(scope="module")
def hip():
    import npbnn_amd
    from npbnn_amd import _capi
    _capi.load_library()
    return npbnn_amd


ACT_KIND = {"relu": 0, "leaky": 1, "swish": 2, "tanh": 3}


def make_ctx(hip, x, weights, act, out_kind, lik_kind, labels=None, targets=None, n_targets=0):
    ctx = hip.HipContext(0)
    ctx.set_data(x)
    if labels is not None:
        ctx.set_labels(labels)
    if targets is not None:
        ctx.set_targets(targets)
    ctx.set_arch_from_weights(weights, x.shape[1], ACT_KIND[act.kind], out_kind, lik_kind, n_targets)
    return ctx


def act_prm(act, n_hidden):
    if act.kind != "leaky":
        return None
    return np.array([act.slope(i) for i in range(n_hidden)], dtype=float)


def assert_close(got, want, tol=Z_TOL):
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err.max() <= tol, "max scaled error %.3e" % err.max()


def check_confusion(conf, y64, labels):
    want = orc.confusion_counts(y64, labels)
    if np.array_equal(conf, want):
        return
    top2 = np.sort(y64, axis=1)[:, -2:]
    near_ties = int(np.sum(top2[:, 1] - top2[:, 0] < 1e-5))
    assert np.abs(conf - want).sum() <= 2 * near_ties, "confusion counts differ beyond fp32 ties"
    assert conf.sum() == want.sum()


@pytest.fixture(scope="module")
def grid(golden_dir):
    return np.load(os.path.join(golden_dir, "grid.npz"))


@pytest.mark.parametrize("case", cases.grid_cases(), ids=lambda c: c["name"])
def test_g1_grid_against_reference_golden(case, grid, hip):
    inp = cases.grid_inputs(case)
    act = orc.Act(case["fun"], inp["prm"]) if inp["prm"] is not None else orc.Act(case["fun"])
    x, w, lab = inp["x"], inp["weights"], inp["labels"]
    k = case["name"]
    ctx = make_ctx(hip, x, w, act, 0, 0, labels=lab)
    ap = act_prm(act, len(w) - 1)
    z = ctx.predict(w, act_prm=ap, apply_out_fn=False)
    y = ctx.predict(w, act_prm=ap, apply_out_fn=True)
    assert_close(z[:16], grid[k + "/z_head"])
    assert_close(y[:16], grid[k + "/y_head"])
    np.testing.assert_allclose(z.sum(axis=0), grid[k + "/z_colsum"], rtol=1e-4, atol=1e-3)
    lik = grid[k + "/lik"]
    finite = np.isfinite(lik[0])
    r = ctx.eval(w, act_prm=ap, want_confusion=True)
    if finite:   # log(softmax) underflows to -inf in the float64 reference for a few extreme cases
        np.testing.assert_allclose(r["loglik"], lik[0], rtol=LL_RTOL)
        np.testing.assert_allclose(ctx.eval(w, act_prm=ap, lik_temp=0.5)["loglik"], lik[3], rtol=LL_RTOL)
        ctx.set_row_weights(instance_w=inp["inst_w"])
        np.testing.assert_allclose(ctx.eval(w, act_prm=ap)["loglik"], lik[1], rtol=LL_RTOL)
        ctx.set_row_weights(class_w=inp["class_w"])
        np.testing.assert_allclose(ctx.eval(w, act_prm=ap)["loglik"], lik[2], rtol=LL_RTOL)
    y64 = orc.forward(x, w, act, orc.out_softmax)
    check_confusion(r["confusion"], y64, lab)
    ctx.close()


def test_g2_regression_against_reference_golden(golden_dir, hip):
    g = np.load(os.path.join(golden_dir, "regression.npz"))
    act = orc.Act("tanh")
    inp = cases.regression_inputs()
    x, w, t = inp["x"], inp["weights"], inp["targets"]
    ctx = make_ctx(hip, x, w, act, 1, 1, targets=t, n_targets=t.shape[1])
    assert_close(ctx.predict(w), g["y"])
    np.testing.assert_allclose(ctx.eval(w, sigma=1.0)["loglik"], g["lik_sig1"], rtol=LL_RTOL)
    np.testing.assert_allclose(ctx.eval(w, sigma=inp["sig_vec"])["loglik"], g["lik_sigvec"], rtol=LL_RTOL)
    r = ctx.eval(w)                                   # empirical sigma
    np.testing.assert_allclose(r["loglik"], g["lik_emp"], rtol=LL_RTOL)
    np.testing.assert_allclose(r["sigma"], g["emp_sigma"], rtol=1e-5)
    np.testing.assert_allclose(ctx.eval(w, lik_temp=0.7)["loglik"], g["lik_emp_temp"], rtol=LL_RTOL)
    np.testing.assert_allclose(r["sum_r2"] / len(x), g["mse_col"], rtol=1e-5)
    np.testing.assert_allclose(np.sum(r["sum_r2"]) / t.size, g["mse"], rtol=1e-5)
    ctx.close()
    # softplus-on-second-half output function
    inp2 = cases.regression_inputs(seed=12, double_out=True)
    ctx = make_ctx(hip, inp2["x"], inp2["weights"], act, 2, 7)
    assert_close(ctx.predict(inp2["weights"]), g["y_err"])
    ctx.close()


@pytest.mark.parametrize("name", list(cases.TRACES))
def test_g4_teacher_forced_trace(name, golden_dir, hip):
    """Replay the reference's recorded proposals: logLik' per proposal must match."""
    cfg = cases.TRACES[name]
    g = np.load(os.path.join(golden_dir, "trace_%s.npz" % name))
    act = orc.Act(cfg["fun"])
    if cfg["kind"] == "classification":
        dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
        w0 = [g["w0_%d" % i] for i in range(len(cfg["n_nodes"]) + 1)]
        ctx = make_ctx(hip, dat["data"], w0, act, 0, 0, labels=dat["labels"])
    else:
        dat = cases.regression_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["k"], cfg["n_test"])
        w0 = [g["w0_%d" % i] for i in range(len(cfg["n_nodes"]) + 1)]
        ctx = make_ctx(hip, dat["data"], w0, act, 1, 1, targets=dat["labels"], n_targets=cfg["k"])
    init_sigma = 1.0 if cfg["kind"] == "regression" else None
    np.testing.assert_allclose(ctx.eval(w0, sigma=init_sigma)["loglik"], g["init"][0], rtol=LL_RTOL)
    for it in range(cfg["keep_w"]):
        wp = [g["wprime_%d_%d" % (it, li)] for li in range(len(w0))]
        np.testing.assert_allclose(ctx.eval(wp)["loglik"], g["rows"][it, 0], rtol=LL_RTOL, err_msg="proposal %d" % it)
    ctx.close()


SHAPES = [
    # n_rows, n_features, hidden, n_classes, bias
    (1, 1, [1], 2, 0), (15, 3, [2], 2, 2), (16, 16, [16], 3, 3), (17, 17, [17, 3], 4, 2),
    (33, 48, [50, 5], 5, 1), (1000, 130, [128, 64, 32, 16, 8, 4, 3], 9, 3), (257, 300, [65, 33], 128, 2), (300, 700, [24, 40], 3, 2),
    (4099, 64, [16, 4], 2, -1),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_f%d_h%s_c%d_b%d" % (s[0], s[1], "x".join(map(str, s[2])), s[3], s[4]))
@pytest.mark.parametrize("fun", ["ReLU", "tanh"])
def test_ragged_and_extreme_shapes(shape, fun, hip):
    n, f, hidden, c, bias = shape
    rs = np.random.default_rng(n * 31 + f)
    x = rs.standard_normal((n, f))
    lab = rs.integers(0, c, n)
    w = [rs.normal(0, 0.4 / np.sqrt(s[1] / 8 + 1), s) for s in cases.layer_shapes(f, hidden, c, bias)]
    act = orc.Act(fun)
    ctx = make_ctx(hip, x, w, act, 0, 0, labels=lab)
    y64 = orc.forward(x, w, act, orc.out_softmax)
    assert_close(ctx.predict(w), y64)
    r = ctx.eval(w, want_confusion=True)
    want = orc.lik_categorical(y64, lab, np.arange(n))
    np.testing.assert_allclose(r["loglik"], want, rtol=5e-6)
    check_confusion(r["confusion"], y64, lab)
    ctx.close()


def test_column_override_and_test_set(hip):
    """data_transform (feature columns replaced by constants) folded into the layer-0 bias;
    evaluation on the resident test set (RunPredictInd path)."""
    rs = np.random.default_rng(5)
    n, f, c = 500, 37, 4
    x, xt = rs.standard_normal((n, f)), rs.standard_normal((123, f))
    lab, labt = rs.integers(0, c, n), rs.integers(0, c, 123)
    w = [rs.normal(0, 0.3, s) for s in cases.layer_shapes(f, [9, 6], c, 2)]
    act = orc.Act("swish")
    ctx = make_ctx(hip, x, w, act, 0, 0, labels=lab)
    ctx.set_data(xt, which=1)
    ctx.set_labels(labt, which=1)
    ind = (rs.random(f) < 0.6).astype(int)
    means = x.mean(axis=0)
    ov = np.where(ind == 0, means, np.nan)
    y64 = orc.forward(x, w, act, orc.out_softmax, col_override=(ind, means))
    assert_close(ctx.predict(w, col_override=ov), y64)
    np.testing.assert_allclose(ctx.eval(w, col_override=ov)["loglik"], orc.lik_categorical(y64, lab, np.arange(n)), rtol=5e-6)
    yt = orc.forward(xt, w, act, orc.out_softmax)
    assert_close(ctx.predict(w, which=1), yt)
    rt = ctx.eval(w, which=1, want_confusion=True)
    np.testing.assert_allclose(rt["loglik"], orc.lik_categorical(yt, labt, np.arange(123)), rtol=5e-6)
    check_confusion(rt["confusion"], yt, labt)
    ctx.close()


def test_error_paths(hip):
    from npbnn_amd import NpbnnError
    ctx = hip.HipContext(0)
    with pytest.raises(NpbnnError):
        ctx.set_labels(np.zeros(3, dtype=int))              # labels before data
    x = np.zeros((10, 4))
    ctx.set_data(x)
    with pytest.raises(NpbnnError):
        ctx.set_labels(np.zeros(11, dtype=int))             # wrong length
    with pytest.raises(NpbnnError):
        ctx.set_arch(4, [5000, 3], [1, 1], 0, 0, 0)          # layer wider than NPBNN_MAX_WIDTH
    ctx.set_arch(4, [200, 3], [1, 1], 0, 0, 0)               # wider than the LDS-resident builds' tiles: the weight-streamed path
    assert ctx.is_wide()
    ctx.set_arch(4, [3, 2], [1, 0], 3, 0, 0)
    assert not ctx.is_wide()
    with pytest.raises(NpbnnError):
        ctx.eval([np.zeros((3, 5)), np.zeros((2, 3))])      # categorical likelihood without labels
    ctx.close()


def test_config2_full_size(hip):
    """BASELINE.json config 2 (100k x 256, [32,8], 10 classes, tanh, bias 2): oracle comparison at
    full size plus size-independent properties (determinism, additivity over row blocks)."""
    rs = np.random.default_rng(0)
    n, f, c = 100_000, 256, 10
    x = rs.standard_normal((n, f))
    lab = rs.integers(0, c, n)
    np.random.seed(1234)
    w = orc.init_weights([32, 8], f, c, bias_node=2)
    act = orc.Act("tanh")
    ctx = make_ctx(hip, x, w, act, 0, 0, labels=lab)
    r1 = ctx.eval(w, want_confusion=True)
    r2 = ctx.eval(w)
    assert r1["loglik"] == r2["loglik"], "evaluation is not run-to-run deterministic"
    y64 = orc.forward(x, w, act, orc.out_softmax)
    want = orc.lik_categorical(y64, lab, np.arange(n))
    assert abs(r1["loglik"] - want) / abs(want) < 1e-7
    check_confusion(r1["confusion"], y64, lab)
    # larger weights: saturated tanh, peaked softmax
    w2 = [wi * 6 for wi in w]
    y64 = orc.forward(x, w2, act, orc.out_softmax)
    want = orc.lik_categorical(y64, lab, np.arange(n))
    assert abs(ctx.eval(w2)["loglik"] - want) / abs(want) < 1e-6
    ctx.close()
    # additivity: loglik(all rows) == loglik(first 60%) + loglik(rest)
    cut = 60_001
    parts = 0.0
    for sl in (slice(0, cut), slice(cut, n)):
        c2 = make_ctx(hip, x[sl], w, act, 0, 0, labels=lab[sl])
        parts += c2.eval(w)["loglik"]
        c2.close()
    assert abs(parts - r1["loglik"]) / abs(r1["loglik"]) < 1e-9


def test_config4_full_size_regression(hip):
    """BASELINE.json config 4 (1e6 x 64 -> 2 targets, [16,4], tanh, bias 2, empirical sigma): oracle comparison at
    full size, determinism, and the closed form of the empirical-sigma likelihood recomputed from the returned moments."""
    rs = np.random.default_rng(0)
    n, f, k = 1_000_000, 64, 2
    x = rs.standard_normal((n, f)).astype(np.float32)
    np.random.seed(1234)
    teacher = orc.init_weights([16, 4], f, k, bias_node=2)
    teacher = [t * 3 for t in teacher]
    act = orc.Act("tanh")
    x64 = x.astype(np.float64)
    targets = orc.forward(x64, teacher, act, orc.out_identity) + 0.5 * rs.standard_normal((n, k))
    w = orc.init_weights([16, 4], f, k, bias_node=2)
    ctx = make_ctx(hip, x, w, act, 1, 1, targets=targets, n_targets=k)
    r1 = ctx.eval(w)
    r2 = ctx.eval(w)
    assert r1["loglik"] == r2["loglik"]
    pred = orc.forward(x64, w, act, orc.out_identity)
    t32 = targets.astype(np.float32).astype(np.float64)
    want, _ = orc.closed_gaussian_empirical(pred, t32)
    assert abs(r1["loglik"] - want) / abs(want) < 1e-7
    res = t32 - pred
    np.testing.assert_allclose(r1["sigma"], np.std(res, axis=0), rtol=1e-6)
    s1, s2 = r1["sum_r"], r1["sum_r2"]
    sg = np.sqrt(s2 / n - (s1 / n) ** 2)
    closed = np.sum(-n * (0.5 * np.log(2 * np.pi) + np.log(sg)) - s2 / (2 * sg ** 2))
    assert abs(closed - r1["loglik"]) / abs(closed) < 1e-12
    # fixed sigma vector
    sig = np.array([0.7, 1.3])
    want = orc.lik_gaussian(pred, t32, sig2=sig)
    assert abs(ctx.eval(w, sigma=sig)["loglik"] - want) / abs(want) < 1e-6
    ctx.close()


def test_config5_block_masked_network(hip):
    """BASELINE.json config 5 (5e4 x 512, 8 blocks of 64 inputs with 4 nodes each, [32,8] -> 1 target, bias -1): the
    block mask zeroes 7/8 of the first layer; masked weights must behave exactly like structural zeros."""
    rs = np.random.default_rng(0)
    n, f, k = 50_000, 512, 1
    x = rs.standard_normal((n, f)).astype(np.float32)
    np.random.seed(1234)
    w = orc.init_weights([32, 8], f, k, bias_node=-1)
    mask = orc.block_mask(w, [np.repeat(np.arange(8), 64), [], []], [[4] * 8, [], []])
    assert mask[0].sum() == 32 * 64
    wm = [wi * mi for wi, mi in zip(w, mask)]
    act = orc.Act("tanh")
    x64 = x.astype(np.float64)
    targets = rs.standard_normal((n, k))
    ctx = make_ctx(hip, x, wm, act, 1, 1, targets=targets, n_targets=k)
    r = ctx.eval(wm)
    pred = orc.forward(x64, wm, act, orc.out_identity)
    t32 = targets.astype(np.float32).astype(np.float64)
    want, _ = orc.closed_gaussian_empirical(pred, t32)
    assert abs(r["loglik"] - want) / abs(want) < 1e-6
    # the masked-out entries are irrelevant: garbage there, times the mask, gives the same bits
    junk = [wi + (1 - mi) * 123.0 for wi, mi in zip(wm, mask)]
    assert ctx.eval([j * m for j, m in zip(junk, mask)])["loglik"] == r["loglik"]
    # block structure: with every later block's second-layer inputs cut, the prediction must ignore inputs 64..511
    w_cut = [wm[0], wm[1].copy(), wm[2]]
    w_cut[1][:, 4:] = 0.0                                              # keep only the 4 nodes of block 0
    y = ctx.predict(w_cut, apply_out_fn=False)
    x2 = x.copy()
    x2[:, 64:] = rs.standard_normal((n, f - 64)).astype(np.float32)
    c2 = make_ctx(hip, x2, wm, act, 1, 1, targets=targets, n_targets=k)
    np.testing.assert_array_equal(c2.predict(w_cut, apply_out_fn=False), y)
    c2.close()
    ctx.close()


# ---- layer-0 precision modes -------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_l0_modes_against_oracle(mode, hip):
    """Both layer-0 paths (exact float32 MFMA and fp16-split) meet the same tolerances; badly scaled feature
    columns (1e-3 ... 1e4) exercise the per-column power-of-two scaling of the fp16-split path."""
    rs = np.random.default_rng(11)
    n, f, c = 3001, 70, 6
    x = rs.standard_normal((n, f)) * (10.0 ** rs.uniform(-3, 4, f))
    x[:, 5] = 0.0                                           # an all-zero column
    lab = rs.integers(0, c, n)
    w = [rs.normal(0, 0.3, s) for s in cases.layer_shapes(f, [24, 9], c, 2)]
    w[0][:, 1:] /= np.maximum(np.abs(x).max(axis=0), 1e-3)  # keep the pre-activations O(1)
    act = orc.Act("tanh")
    ctx = make_ctx(hip, x, w, act, 0, 0, labels=lab)
    ctx.set_l0_precision(mode)
    y64 = orc.forward(x, w, act, orc.out_softmax)
    assert_close(ctx.predict(w), y64)
    r = ctx.eval(w, want_confusion=True)
    assert ctx.l0_mode() == ("f16-split" if mode == "f16" else "f32")
    np.testing.assert_allclose(r["loglik"], orc.lik_categorical(y64, lab, np.arange(n)), rtol=LL_RTOL)
    check_confusion(r["confusion"], y64, lab)
    ctx.close()


def test_auto_mode_falls_back_to_f32(hip):
    """auto: data with inf, or weights beyond the fp16 range after column scaling, run on the float32 path."""
    from npbnn_amd import NpbnnError
    rs = np.random.default_rng(12)
    n, f, c = 500, 40, 3
    x = rs.standard_normal((n, f))
    lab = rs.integers(0, c, n)
    w = [rs.normal(0, 0.3, s) for s in cases.layer_shapes(f, [8], c, 2)]
    act = orc.Act("ReLU")
    ctx = make_ctx(hip, x, w, act, 0, 0, labels=lab)
    r = ctx.eval(w)
    assert ctx.l0_mode() == "f16-split"
    big = [wi.copy() for wi in w]
    big[0][0, 3] = 3e5                                      # leaves the fp16 range -> this evaluation repeats in float32
    y64 = orc.forward(x, big, act, orc.out_softmax)
    with np.errstate(divide="ignore"):
        want = orc.lik_categorical(y64, lab, np.arange(n))
    got = ctx.eval(big)["loglik"]
    assert ctx.l0_mode() == "f32"
    if np.isfinite(want):
        np.testing.assert_allclose(got, want, rtol=1e-5)
    np.testing.assert_allclose(ctx.eval(w)["loglik"], r["loglik"], rtol=0)      # back on the fp16-split path, same bits
    assert ctx.l0_mode() == "f16-split"
    ctx.set_l0_precision("f16")
    with pytest.raises(NpbnnError):
        ctx.eval(big)
    ctx.close()
    x2 = x.copy()
    x2[7, 3] = np.inf
    ctx = make_ctx(hip, x2, w, act, 0, 0, labels=lab)
    ctx.eval(w)
    assert ctx.l0_mode() == "f32"
    ctx.close()


def _heavy_tailed(kind, rs, n, f):
    if kind == "outlier":           # one 1e4 outlier per column over N(0, 1e-2) data
        x = rs.normal(0, 1e-2, (n, f))
        x[rs.integers(0, n, f), np.arange(f)] = 1e4
    elif kind == "outlier12":       # one 1e12 outlier per column over N(0, 1) data: past what a moved scale can hold
        x = rs.normal(0, 1.0, (n, f))
        x[rs.integers(0, n, f), np.arange(f)] = 1e12
    elif kind == "lognormal3":      # log-normal columns, sigma = 3
        x = np.exp(3.0 * rs.standard_normal((n, f)))
    elif kind in ("student2", "student3"):   # Student t: 2 degrees of freedom are past the bound, 3 just inside it at this size
        x = rs.standard_t(int(kind[-1]), (n, f))
    elif kind == "counts":          # integer counts 0 .. 1e5: exact as a pair of fp16 numbers
        x = rs.integers(0, 100001, (n, f)).astype(float)
    elif kind == "lognormal1":      # moderate tails: the fp16 pair is still a fair picture
        x = np.exp(rs.standard_normal((n, f)))
    else:
        raise ValueError(kind)
    return x


@pytest.mark.parametrize("kind,expect,moved", [("outlier", "f16-split", True), ("lognormal3", "f16-split", True), ("student2", "f16-split", True),
                                               ("student3", "f16-split", False), ("counts", "f16-split", False), ("lognormal1", "f16-split", False),
                                               ("outlier12", "f32", True)])
def test_auto_mode_on_heavy_tailed_columns(kind, expect, moved, hip):
    """The fp16 pair scales every column by a power of two from its LARGEST entry; a column whose typical entries lie many powers
    of two below that one (an outlier, log-normal or heavy-tailed features) would keep only a few bits of them.  The library
    measures the pair's largest entry error (beyond the pair's own 22-bit rounding) against the column's mean |value| and moves the
    scale of a column past 2^-17 up by the power of two that brings it inside (fp16's range above 1 is otherwise unused; at most
    2^12: ensure_scales, NPBNN_INFO_F16_MOVED_COLUMNS); only a column that even that cannot help (one 1e12 outlier) keeps `auto` on
    float32.  Either way the result meets the tolerances of the float32 path (arithmetic of the reference: np.dot in float64,
    BNN_lib.py:154-162).  (Round 4 sent the first three kinds to float32: VERDICT r04 item 5.)"""
    from npbnn_amd import NpbnnError
    rs = np.random.default_rng(21)
    n, f, c = 20000, 48, 5
    x = _heavy_tailed(kind, rs, n, f)
    x32 = x.astype(np.float32).astype(np.float64)            # (the device holds float32 features)
    lab = rs.integers(0, c, n)
    w = [rs.normal(0, 0.3, s) for s in cases.layer_shapes(f, [24, 9], c, 2)]
    # pre-activations O(1) for the TYPICAL row: weights scaled by the columns' mean |value|
    w[0][:, 1:] /= np.maximum(np.abs(x32).mean(axis=0), 1e-12) * np.sqrt(f)
    act = orc.Act("tanh")
    y64 = orc.forward(x32, w, act, orc.out_softmax)
    z64 = orc.forward_logits(x32, w, act)
    want = orc.lik_categorical(y64, lab, np.arange(n))
    ctx = make_ctx(hip, x, w, act, 0, 0, labels=lab)
    got = ctx.eval(w)["loglik"]
    assert ctx.l0_mode() == expect
    n_moved, largest = ctx.f16_moved_columns()
    assert (n_moved > 0) == moved and 0 <= largest <= 12 and (largest > 0) == moved
    np.testing.assert_allclose(got, want, rtol=LL_RTOL)
    assert_close(ctx.predict(w, apply_out_fn=False), z64)
    ctx.set_l0_precision("f32")
    np.testing.assert_allclose(ctx.eval(w)["loglik"], want, rtol=LL_RTOL)
    ctx.set_l0_precision("f16")
    if expect == "f32":          # asked for by name, the fp16 pair is refused with the column that breaks it
        with pytest.raises(NpbnnError, match="powers of two"):
            ctx.eval(w)
    else:
        np.testing.assert_allclose(ctx.eval(w)["loglik"], want, rtol=LL_RTOL)
    ctx.close()


# ---- plug-in likelihoods (BNN_lik.py) and the predicted-sigma Gaussian ----------------------------
def test_g3_count_likelihoods_against_reference_golden(golden_dir, hip):
    g = np.load(os.path.join(golden_dir, "counts.npz"))
    act = orc.Act("swish")
    a = cases.count_inputs(seed=23, n_out=1, k=1)
    ctx = make_ctx(hip, a["x"], a["weights"], act, 1, 3, targets=a["counts"], n_targets=1)
    np.testing.assert_allclose(ctx.eval(a["weights"])["loglik"], g["poi"], rtol=LL_RTOL)
    np.testing.assert_allclose(ctx.eval(a["weights"], lik_temp=0.3)["loglik"], g["poi"], rtol=LL_RTOL)   # lik_temp is ignored upstream
    ctx.close()
    b = cases.count_inputs(seed=24, n_out=2, k=1)
    ctx = make_ctx(hip, b["x"], b["weights"], act, 1, 4, targets=b["counts"], n_targets=1)
    np.testing.assert_allclose(ctx.eval(b["weights"])["loglik"], g["nb"], rtol=5e-6)
    ctx.close()
    ctx = make_ctx(hip, b["x"], b["weights"], act, 1, 6, targets=b["counts"], n_targets=1)
    np.testing.assert_allclose(ctx.eval(b["weights"])["loglik"], g["nb10"], rtol=5e-6)
    ctx.close()
    c = cases.count_inputs(seed=25, n_out=4, k=2)
    ctx = make_ctx(hip, c["x"], c["weights"], act, 1, 5, targets=c["counts"], n_targets=2)
    np.testing.assert_allclose(ctx.eval(c["weights"])["loglik"], g["nb2d"], rtol=5e-6)
    ctx.close()


def test_g2_predicted_sigma_likelihood(golden_dir, hip):
    g = np.load(os.path.join(golden_dir, "regression.npz"))
    inp2 = cases.regression_inputs(seed=12, double_out=True)
    ctx = make_ctx(hip, inp2["x"], inp2["weights"], orc.Act("tanh"), 2, 2, targets=inp2["targets"], n_targets=2)
    np.testing.assert_allclose(ctx.eval(inp2["weights"])["loglik"], g["lik_err"], rtol=LL_RTOL)
    np.testing.assert_allclose(ctx.eval(inp2["weights"], lik_temp=0.5)["loglik"], 0.5 * g["lik_err"], rtol=LL_RTOL)
    assert_close(ctx.predict(inp2["weights"]), g["y_err"])
    ctx.close()


@pytest.mark.parametrize("l0", ["auto", "f32"])
@pytest.mark.parametrize("kind", ["classification", "regression"])
def test_layer0_block_structure_is_the_dense_result_bit_for_bit(hip, l0, kind):
    """npbnn_set_layer_mask: blocks of the first layer without weights are neither stored nor multiplied (create_mask layouts,
    BNN_lib.py:16-47).  Skipped weights are zeros, so every result - likelihood on the fast and on the general build, confusion
    counts, predictions - must be the dense one to the last bit, on the fp16-split and on the float32 path; a weight that is
    not zero where the mask is fails the call."""
    rs = np.random.default_rng(5)
    n, f = 3001, 160                      # 5 K-steps of 32 features; 48 nodes = 3 output tiles
    x = rs.standard_normal((n, f)).astype(np.float32)
    np.random.seed(7)
    n_out = 6 if kind == "classification" else 2
    w = orc.init_weights([48, 8], f, n_out, bias_node=2 if kind == "classification" else -1)
    # 10 groups of 16 features -> blocks of 4, 5, 5, 5, 5, 5, 5, 5, 5, 4 nodes: tile 0 sees features 0..63, tile 1 48..127, tile 2 112..159
    groups = np.repeat(np.arange(10), 16)
    mask = orc.block_mask(w, [groups, [], []], [[4, 5, 5, 5, 5, 5, 5, 5, 5, 4], [], []])
    if w[0].shape[1] == f + 1:            # (block_mask numbers columns from 0: with a bias column the layout shifts by one)
        m0 = np.zeros(w[0].shape)
        m0[:, 0] = 1
        m0[:, 1:] = orc.block_mask([w[0][:, 1:]], [groups], [[4, 5, 5, 5, 5, 5, 5, 5, 5, 4]])[0]
        mask[0] = m0
    wm = [wi * mi for wi, mi in zip(w, mask)]
    act = orc.Act("tanh")
    if kind == "classification":
        labels = rs.integers(0, n_out, n)
        mk = lambda: make_ctx(hip, x, wm, act, 0, 0, labels=labels)                      # noqa: E731
    else:
        targets = rs.standard_normal((n, n_out))
        mk = lambda: make_ctx(hip, x, wm, act, 1, 1, targets=targets, n_targets=n_out)   # noqa: E731
    dense, blocked = mk(), mk()
    for c in (dense, blocked):
        c.set_l0_precision(l0)
    blocked.set_layer_mask(mask)
    for fast in (1, 0):
        for c in (dense, blocked):
            c.set_fast_tails(fast)
        a, b = dense.eval(wm), blocked.eval(wm)
        assert a["loglik"] == b["loglik"] and np.array_equal(a["sum_r2"], b["sum_r2"]), (l0, kind, fast)
    if kind == "classification":
        a, b = dense.eval(wm, want_confusion=True), blocked.eval(wm, want_confusion=True)
        assert a["loglik"] == b["loglik"] and np.array_equal(a["confusion"], b["confusion"])
    np.testing.assert_array_equal(dense.predict(wm), blocked.predict(wm))
    assert blocked.l0_mode() == ("f32" if l0 == "f32" else "f16-split")
    bad = [wm[0].copy()] + wm[1:]
    bad[0][2, -1] = 0.25                   # node 2 (tile 0) x feature 159 (K-unit 4): outside tile 0's blocks
    with pytest.raises(hip.NpbnnError, match="mask"):
        blocked.eval(bad)
    blocked.set_layer_mask(None)           # dense again: the same weights are fine
    assert blocked.eval(bad)["loglik"] == dense.eval(bad)["loglik"]
    dense.close()
    blocked.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_architectures_against_the_oracle(seed, hip):
    """Seeded random networks - 2 to 4 weight matrices, hidden widths 1 .. 70, with and without bias columns, every activation,
    categorical and Gaussian heads, both layer-0 precisions - against the float64 oracle: the layouts a shape selects (narrow
    hidden layers stored transposed inside their tile, layer 1 on fp16-split products for tanh networks with two or more layer-0
    tiles, fast and general builds) must all give the same network."""
    rs = np.random.default_rng(1000 + seed)
    n = int(rs.integers(40, 400))
    f = int(rs.choice([3, 17, 32, 64, 100, 130]))
    n_hidden = int(rs.integers(1, 4))
    widths = [int(rs.choice([1, 2, 4, 5, 8, 9, 13, 16, 17, 31, 32, 33, 48, 70])) if i == 0 else int(rs.integers(1, 17))
              for i in range(n_hidden)]
    classification = bool(rs.integers(0, 2))
    n_out = int(rs.integers(2, 12)) if classification else int(rs.integers(1, 3))
    bias = int(rs.choice([-1, 0, 2, 3]))
    fun = str(rs.choice(["tanh", "tanh", "ReLU", "swish", "genReLU"]))
    act = orc.Act(fun, np.full(n_hidden, 0.2)) if fun == "genReLU" else orc.Act(fun)
    x = rs.standard_normal((n, f)) * rs.uniform(0.2, 3.0)
    np.random.seed(seed)
    w = orc.init_weights(widths, f, n_out, init_std=0.5, bias_node=bias)
    w = [wi * rs.uniform(0.5, 2.0) + 0.05 * rs.standard_normal(wi.shape) for wi in w]
    ap = act_prm(act, len(w) - 1)
    z64 = orc.forward_logits(x, w, act)
    for l0 in ("f16", "f32"):
        if classification:
            labels = rs.integers(0, n_out, n)
            ctx = make_ctx(hip, x, w, act, 0, 0, labels=labels)
            want = orc.lik_categorical(orc.out_softmax(z64), labels, np.arange(n))
        else:
            targets = rs.standard_normal((n, n_out))
            ctx = make_ctx(hip, x, w, act, 1, 1, targets=targets, n_targets=n_out)
            want = orc.closed_gaussian_empirical(z64, targets)[0]
        ctx.set_l0_precision(l0)
        assert_close(ctx.predict(w, act_prm=ap, apply_out_fn=False), z64, tol=5e-5)
        assert ctx.l0_mode() == ("f16-split" if l0 == "f16" else "f32")
        for fast in (1, 0):
            ctx.set_fast_tails(fast)
            got = ctx.eval(w, act_prm=ap)["loglik"]
            np.testing.assert_allclose(got, want, rtol=5e-6, err_msg="%s widths=%s bias=%d f=%d fast=%d l0=%s" % (fun, widths, bias, f, fast, l0))
        ctx.close()
