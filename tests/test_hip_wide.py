"""GPU tests of the weight-streamed path (npbnn_amd/csrc/npbnn_wide.hip.h): networks the LDS of a compute unit cannot hold -
layers of more than 128 nodes, the reference's default [50, 5] (np_bnn/BNN_env.py:20) on thousands of features - run layer by
layer as tiled matrix products (MatrixMultiplicationD, np_bnn/BNN_lib.py:154-162, takes any shape).

Held to the same fixtures and tolerances as the resident path (tests/test_hip_parity.py): the reference's golden grid with the
path forced on small networks (NPBNN_OPT_WIDE), the oracle at the sizes that need it, and the device chain against the mh_step
loop (same accept / reject sequence, same weights to the bit)."""
import os

import numpy as np
import pytest

import cases
import oracle as orc
import npbnn_amd as bn
from npbnn_amd import _capi as capi

pytestmark = pytest.mark.gpu

LL_RTOL = 2e-6
Z_TOL = 2e-5
ACT_KIND = {"relu": 0, "leaky": 1, "swish": 2, "tanh": 3}


def make_ctx(x, weights, act, out_kind, lik_kind, labels=None, targets=None, n_targets=0, wide=True, precision="auto", test=None):
    ctx = bn.HipContext(0)
    ctx.set_l0_precision(precision)
    if wide:
        ctx.set_wide(True)
    ctx.set_data(x)
    if test is not None:
        ctx.set_data(test, capi.TEST)
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
def test_forced_on_the_reference_grid(case, grid):
    """G1 (every activation x bias mode x shape of the reference's golden grid) with the small networks forced onto the streamed path."""
    inp = cases.grid_inputs(case)
    act = orc.Act(case["fun"], inp["prm"]) if inp["prm"] is not None else orc.Act(case["fun"])
    x, w, lab = inp["x"], inp["weights"], inp["labels"]
    k = case["name"]
    ctx = make_ctx(x, w, act, 0, 0, labels=lab)
    assert ctx.is_wide()
    ap = act_prm(act, len(w) - 1)
    z = ctx.predict(w, act_prm=ap, apply_out_fn=False)
    y = ctx.predict(w, act_prm=ap, apply_out_fn=True)
    assert_close(z[:16], grid[k + "/z_head"])
    assert_close(y[:16], grid[k + "/y_head"])
    np.testing.assert_allclose(z.sum(axis=0), grid[k + "/z_colsum"], rtol=1e-4, atol=1e-3)
    lik = grid[k + "/lik"]
    r = ctx.eval(w, act_prm=ap, want_confusion=True)
    if np.isfinite(lik[0]):
        np.testing.assert_allclose(r["loglik"], lik[0], rtol=LL_RTOL)
        np.testing.assert_allclose(ctx.eval(w, act_prm=ap, lik_temp=0.5)["loglik"], lik[3], rtol=LL_RTOL)
        ctx.set_row_weights(instance_w=inp["inst_w"])
        np.testing.assert_allclose(ctx.eval(w, act_prm=ap)["loglik"], lik[1], rtol=LL_RTOL)
        ctx.set_row_weights(class_w=inp["class_w"])
        np.testing.assert_allclose(ctx.eval(w, act_prm=ap)["loglik"], lik[2], rtol=LL_RTOL)
    check_confusion(r["confusion"], orc.forward(x, w, act, orc.out_softmax), lab)
    ctx.close()


@pytest.mark.parametrize("precision", ["auto", "f32"])
def test_forced_on_the_regression_and_count_fixtures(golden_dir, precision):
    g = np.load(os.path.join(golden_dir, "regression.npz"))
    act = orc.Act("tanh")
    inp = cases.regression_inputs()
    x, w, t = inp["x"], inp["weights"], inp["targets"]
    ctx = make_ctx(x, w, act, 1, 1, targets=t, n_targets=t.shape[1], precision=precision)
    assert ctx.is_wide() and ctx.l0_mode() == ("f32" if precision == "f32" else ctx.l0_mode())
    assert_close(ctx.predict(w), g["y"])
    np.testing.assert_allclose(ctx.eval(w, sigma=1.0)["loglik"], g["lik_sig1"], rtol=LL_RTOL)
    np.testing.assert_allclose(ctx.eval(w, sigma=inp["sig_vec"])["loglik"], g["lik_sigvec"], rtol=LL_RTOL)
    r = ctx.eval(w)
    np.testing.assert_allclose(r["loglik"], g["lik_emp"], rtol=LL_RTOL)
    np.testing.assert_allclose(r["sigma"], g["emp_sigma"], rtol=1e-5)
    np.testing.assert_allclose(ctx.eval(w, lik_temp=0.7)["loglik"], g["lik_emp_temp"], rtol=LL_RTOL)
    np.testing.assert_allclose(r["sum_r2"] / len(x), g["mse_col"], rtol=1e-5)
    ctx.close()
    inp2 = cases.regression_inputs(seed=12, double_out=True)
    ctx = make_ctx(inp2["x"], inp2["weights"], act, 2, 7, precision=precision)
    assert_close(ctx.predict(inp2["weights"]), g["y_err"])
    ctx.close()
    # predicted-sigma Gaussian and the count plug-ins: float64 row-wise terms
    ctx = make_ctx(inp2["x"], inp2["weights"], act, 2, 2, targets=inp2["targets"], n_targets=2, precision=precision)
    np.testing.assert_allclose(ctx.eval(inp2["weights"])["loglik"], g["lik_err"], rtol=LL_RTOL)
    np.testing.assert_allclose(ctx.eval(inp2["weights"], lik_temp=0.5)["loglik"], 0.5 * g["lik_err"], rtol=LL_RTOL)
    ctx.close()
    gc = np.load(os.path.join(golden_dir, "counts.npz"))
    sw = orc.Act("swish")
    a = cases.count_inputs(seed=23, n_out=1, k=1)
    ctx = make_ctx(a["x"], a["weights"], sw, 1, 3, targets=a["counts"], n_targets=1, precision=precision)
    np.testing.assert_allclose(ctx.eval(a["weights"])["loglik"], gc["poi"], rtol=LL_RTOL)
    ctx.close()
    b = cases.count_inputs(seed=24, n_out=2, k=1)
    for kind, key in ((4, "nb"), (6, "nb10")):
        ctx = make_ctx(b["x"], b["weights"], sw, 1, kind, targets=b["counts"], n_targets=1, precision=precision)
        np.testing.assert_allclose(ctx.eval(b["weights"])["loglik"], gc[key], rtol=5e-6)
        ctx.close()
    c = cases.count_inputs(seed=25, n_out=4, k=2)
    ctx = make_ctx(c["x"], c["weights"], sw, 1, 5, targets=c["counts"], n_targets=2, precision=precision)
    np.testing.assert_allclose(ctx.eval(c["weights"])["loglik"], gc["nb2d"], rtol=5e-6)
    ctx.close()


def _classification_problem(seed, n, f, c, hidden, bias=2, scale=1.0):
    rs = np.random.default_rng(seed)
    x = rs.standard_normal((n, f)).astype(np.float32)
    lab = rs.integers(0, c, n)
    np.random.seed(1234)
    if hidden:
        w = orc.init_weights(hidden, f, c, bias_node=bias)
    else:           # (the reference's initialiser wants a hidden layer; RunPredict itself takes a single matrix)
        w = [np.random.normal(0, 0.1, (c, f + 1))]
    return x, lab, [wi * scale for wi in w]


SHAPES = {
    # (rows, features, hidden, classes): the shapes VERDICT r04 names, and two that exercise the tilings' edges
    "default_net_2000x2000": (2000, 2000, [50, 5], 10),
    "f4096_h256_64": (20000, 4096, [256, 64], 10),
    "f1000_h200_50_10": (5000, 1000, [200, 50, 10], 7),
    "h300_odd_tiles": (777, 100, [300, 33], 150),
    "single_layer_wide_out": (1000, 72, [], 200),
}


@pytest.mark.parametrize("precision", ["auto", "f32"])
@pytest.mark.parametrize("name", list(SHAPES))
def test_shapes_the_lds_cannot_hold_against_the_oracle(name, precision):
    n, f, hidden, c = SHAPES[name]
    x, lab, w = _classification_problem(3, n, f, c, hidden, scale=2.0)
    act = orc.Act("tanh")
    ctx = make_ctx(x, w, act, 0, 0, labels=lab, wide=False, precision=precision)
    assert ctx.is_wide(), "this shape must pick the weight-streamed path by itself"
    r = ctx.eval(w, want_confusion=True)
    assert ctx.l0_mode() == ("f16-split" if precision == "auto" else "f32")
    assert r["loglik"] == ctx.eval(w)["loglik"], "not run-to-run deterministic"
    x64 = x.astype(np.float64)
    y64 = orc.forward(x64, w, act, orc.out_softmax)
    want = orc.lik_categorical(y64, lab, np.arange(n))
    assert abs(r["loglik"] - want) / abs(want) < LL_RTOL
    check_confusion(r["confusion"], y64, lab)
    # (float32 accumulation: the rounding error of a contraction grows with the square root of its length - the budget of the short
    # contractions, 2e-5, times sqrt(features / 1024) from there on; the log-likelihood keeps its 2e-6)
    tol = Z_TOL * max(1.0, np.sqrt(f / 1024.0))
    z = ctx.predict(w, apply_out_fn=False)
    assert_close(z, orc.forward_logits(x64, w, act), tol)
    assert_close(ctx.predict(w), y64, tol)
    ctx.close()


def _drawn_shape(seed):
    """A network np.dot would take without a thought (BNN_lib.py:154-162): ragged row counts (down to a single row), feature counts that
    are no multiple of anything, one to five weight matrices of 1-400 nodes, with and without bias nodes."""
    rs = np.random.default_rng(1000 + seed)
    n = int(rs.choice([1, 2, 15, 16, 17, 255, 257, 1000, 2999, int(rs.integers(1, 3000))]))
    f = int(rs.choice([1, 3, 31, 32, 33, 100, 255, 700, int(rs.integers(1, 700))]))
    n_hidden = int(rs.integers(0, 5))
    hidden = [int(rs.choice([1, 2, 15, 17, 50, 128, 129, 400, int(rs.integers(1, 400))])) for _ in range(n_hidden)]
    c = int(rs.choice([2, 3, 10, 129, 300]))
    bias = [int(rs.integers(0, 2)) for _ in range(n_hidden + 1)]       # (per weight matrix: a bias column or none)
    fun = ["tanh", "ReLU", "swish"][int(rs.integers(0, 3))]
    return n, f, hidden, c, bias, fun


@pytest.mark.parametrize("forced", [True, False], ids=["forced", "library's choice"])
@pytest.mark.parametrize("seed", range(30))
def test_drawn_shapes_against_the_oracle(seed, forced):
    """Thirty drawn shapes (see _drawn_shape) forced onto the weight-streamed path and on the path the library picks by itself, fp16-split
    and float32 layer 0: last layer's values, class probabilities, log-likelihood and confusion counts against the float64 oracle."""
    n, f, hidden, c, bias, fun = _drawn_shape(seed)
    rs = np.random.default_rng(seed)
    x = rs.standard_normal((n, f)).astype(np.float32)
    lab = rs.integers(0, c, n)
    dims = [f] + hidden + [c]
    w = []
    for l in range(len(dims) - 1):
        w.append(rs.normal(0, 1.0 / np.sqrt(dims[l] + 1), (dims[l + 1], dims[l] + bias[l])))
    act = orc.Act(fun)
    x64 = x.astype(np.float64)
    y64 = orc.forward(x64, w, act, orc.out_softmax)
    z64 = orc.forward_logits(x64, w, act)
    want = orc.lik_categorical(y64, lab, np.arange(n))
    for precision in ("auto", "f32"):
        ctx = make_ctx(x, w, act, 0, 0, labels=lab, precision=precision, wide=forced)
        assert ctx.is_wide() or not forced
        assert ctx.is_wide() or max(hidden + [c]) <= 128
        r = ctx.eval(w, want_confusion=True)
        assert abs(r["loglik"] - want) <= LL_RTOL * abs(want) + 1e-9, (n, f, hidden, c, bias, fun, precision)
        check_confusion(r["confusion"], y64, lab)
        assert_close(ctx.predict(w, apply_out_fn=False), z64)
        assert_close(ctx.predict(w), y64)
        ctx.close()


@pytest.mark.parametrize("forced", [True, False], ids=["forced", "library's choice"])
@pytest.mark.parametrize("seed", range(20))
def test_drawn_regression_shapes_against_the_oracle(seed, forced):
    """The same drawn networks with 1-16 Gaussian targets (calc_likelihood_regression, BNN_lib.py:123-131): predictions, the
    log-likelihood with a given sigma per column and with the empirical one (BNN_env.py:475-476), the residual moments."""
    n, f, hidden, _, bias, fun = _drawn_shape(100 + seed)
    rs = np.random.default_rng(100 + seed)
    n = max(n, 17)
    k = int(rs.choice([1, 2, 3, 4, 5, 9, 16]))
    x = rs.standard_normal((n, f)).astype(np.float32)
    dims = [f] + hidden + [k]
    w = [rs.normal(0, 1.0 / np.sqrt(dims[l] + 1), (dims[l + 1], dims[l] + bias[l])) for l in range(len(dims) - 1)]
    act = orc.Act(fun)
    x64 = x.astype(np.float64)
    y64 = orc.forward(x64, w, act, orc.out_identity)
    t = y64 + rs.normal(0, 0.7, y64.shape) + 0.1
    sig = rs.uniform(0.5, 2.0, k)
    want_emp, sig_emp = orc.closed_gaussian_empirical(y64, t)
    want_sig = orc.lik_gaussian(y64, t, sig2=sig)
    ctx = make_ctx(x, w, act, 1, 1, targets=t, n_targets=k, wide=forced)
    assert ctx.is_wide() or not forced
    r = ctx.eval(w)
    np.testing.assert_allclose(r["loglik"], want_emp, rtol=LL_RTOL)
    np.testing.assert_allclose(r["sigma"][:k], sig_emp, rtol=1e-5)
    np.testing.assert_allclose(ctx.eval(w, sigma=sig)["loglik"], want_sig, rtol=LL_RTOL)
    np.testing.assert_allclose(ctx.eval(w, sigma=sig, lik_temp=0.6)["loglik"], 0.6 * want_sig, rtol=LL_RTOL)
    np.testing.assert_allclose(r["sum_r2"][:k], np.sum((t - y64) ** 2, axis=0), rtol=1e-5)
    assert_close(ctx.predict(w), y64)
    ctx.close()


def test_the_widest_and_the_deepest_network_the_header_allows():
    """NPBNN_MAX_WIDTH = 4096 nodes in a layer, NPBNN_MAX_LAYERS = 8 weight matrices: both ends of the header's envelope, against the
    oracle (the reference itself has no limit: np.dot)."""
    rs = np.random.default_rng(9)
    for n, f, hidden, c in ((300, 64, [capi.MAX_WIDTH, 7], 3), (500, 40, [24, 130, 16, 9, 200, 5, 33], 4)):
        x = rs.standard_normal((n, f)).astype(np.float32)
        lab = rs.integers(0, c, n)
        dims = [f] + hidden + [c]
        w = [rs.normal(0, 1.0 / np.sqrt(dims[l] + 1), (dims[l + 1], dims[l] + (1 if l < len(dims) - 2 else 0))) for l in range(len(dims) - 1)]
        assert len(w) <= 8
        act = orc.Act("tanh")
        ctx = make_ctx(x, w, act, 0, 0, labels=lab, wide=False)
        assert ctx.is_wide()
        x64 = x.astype(np.float64)
        y64 = orc.forward(x64, w, act, orc.out_softmax)
        want = orc.lik_categorical(y64, lab, np.arange(n))
        assert abs(ctx.eval(w)["loglik"] - want) / abs(want) < LL_RTOL
        assert_close(ctx.predict(w, apply_out_fn=False), orc.forward_logits(x64, w, act), Z_TOL * 2)
        ctx.close()
    from npbnn_amd import NpbnnError
    w = [rs.normal(0, 0.1, (capi.MAX_WIDTH + 1, 9)), rs.normal(0, 0.1, (3, capi.MAX_WIDTH + 1))]
    ctx = bn.HipContext(0)
    ctx.set_data(rs.standard_normal((64, 8)).astype(np.float32))
    with pytest.raises(NpbnnError):
        ctx.set_arch_from_weights(w, 8, 3, 0, 0, 0)
    ctx.close()


@pytest.mark.parametrize("f", [608, 672, 704, 736, 800])
def test_either_side_of_the_switch_between_the_paths(f):
    """The reference's default [50, 5] on 600-800 features: the weight image takes 127-166 KB of a compute unit's 160 KB of LDS.  While
    four waves fit beside it the resident kernel runs (a wave per SIMD - its tile schedule needs that many; until round 5 such
    networks were planned with one to three waves and summed a quarter to three quarters of the rows); from there on the
    weight-streamed path.  Either way the log-likelihood is the oracle's."""
    n, c = 30_000, 10
    x, lab, w = _classification_problem(5, n, f, c, [50, 5], bias=1)
    act = orc.Act("ReLU")
    ctx = make_ctx(x, w, act, 0, 0, labels=lab, wide=False)
    r = ctx.eval(w, want_confusion=True)
    assert ctx.l0_mode() == "f16-split"
    assert ctx.is_wide() == (f > 672), "F = %d: weight-streamed %s" % (f, ctx.is_wide())
    if not ctx.is_wide():
        assert ctx.info(capi.INFO_WAVES_PER_BLOCK) >= 4
    x64 = x.astype(np.float64)
    y64 = orc.forward(x64, w, act, orc.out_softmax)
    want = orc.lik_categorical(y64, lab, np.arange(n))
    assert abs(r["loglik"] - want) / abs(want) < LL_RTOL
    check_confusion(r["confusion"], y64, lab)
    ctx.set_l0_precision("f32")            # the float32 image of this layer keeps its padding rows: the switch comes earlier
    r32 = ctx.eval(w)
    assert abs(r32["loglik"] - want) / abs(want) < LL_RTOL
    assert ctx.is_wide()                   # (64 rows x 600 features and more: past the LDS on that layout)
    ctx.close()


def test_wide_regression_with_a_test_set_and_column_override():
    rs = np.random.default_rng(5)
    n, f, k = 3000, 1500, 3
    x = rs.standard_normal((n, f))
    xt = rs.standard_normal((500, f))
    np.random.seed(7)
    w = orc.init_weights([64, 8], f, k, bias_node=1)
    act = orc.Act("swish")
    t = rs.standard_normal((n, k))
    ctx = make_ctx(x, w, act, 1, 1, targets=t, n_targets=k, wide=False, test=xt)
    assert ctx.is_wide()
    y64 = orc.forward(x, w, act, orc.out_identity)
    r = ctx.eval(w)
    want, sig = orc.closed_gaussian_empirical(y64, t)
    np.testing.assert_allclose(r["loglik"], want, rtol=LL_RTOL)
    np.testing.assert_allclose(r["sigma"][:k], sig, rtol=1e-6)
    assert_close(ctx.predict(w, which=capi.TEST), orc.forward(xt, w, act, orc.out_identity))
    ov = np.full(f, np.nan)
    ov[[3, 700, 1499]] = [0.25, -1.0, 2.0]
    x2 = x.copy()
    x2[:, [3, 700, 1499]] = [0.25, -1.0, 2.0]
    assert_close(ctx.predict(w, col_override=ov), orc.forward(x2, w, act, orc.out_identity))
    ctx.close()


def _chains(dat, model_kw, act_kw, sampler_kw):
    from test_hip_sampler import quiet
    out = []
    for _ in range(2):
        np.random.seed(77)
        bnn = quiet(bn.npBNN, dat, actFun=bn.ActFun(**act_kw), **model_kw)
        out.append((bnn, bn.MCMC(bnn, **sampler_kw)))
    return out


@pytest.mark.parametrize("name", ["default_net_2000x2000", "f1000_h200_50_10", "h300_odd_tiles"])
def test_run_steps_is_the_mh_step_loop_on_streamed_networks(name):
    n, f, hidden, c = SHAPES[name]
    dat = cases.classification_data(11, n, f, c, n_test=200)
    (bnn_a, mcmc_a), (bnn_b, mcmc_b) = _chains(dat, dict(n_nodes=hidden, use_bias_node=2, prior_f=1, p_scale=1), dict(fun="tanh"),
                                               dict(update_f=[0.02] * (len(hidden) + 1), update_ws=[0.05] * (len(hidden) + 1), n_iteration=100000))
    assert mcmc_a._backend.ctx.is_wide()
    np.testing.assert_allclose(mcmc_a._logLik, mcmc_b._logLik, rtol=0)
    for _ in range(90):
        mcmc_a.mh_step(bnn_a)
    for _ in range(3):
        mcmc_b.run_steps(bnn_b, 30)
    assert mcmc_b._device_iterations == 90
    assert mcmc_a._last_accepted_mem == mcmc_b._last_accepted_mem
    assert 0 < sum(mcmc_a._last_accepted_mem) or mcmc_a._acceptance_rate >= 0
    for wa, wb in zip(bnn_a._w_layers, bnn_b._w_layers):
        np.testing.assert_array_equal(wa, wb)
    np.testing.assert_allclose(mcmc_b._logLik, mcmc_a._logLik, rtol=1e-12)
    np.testing.assert_allclose(mcmc_b._logPrior, mcmc_a._logPrior, rtol=1e-11)
    # the chain's state against the oracle
    act = orc.Act("tanh")
    y64 = orc.forward(dat["data"], bnn_b._w_layers, act, orc.out_softmax)
    want = orc.lik_categorical(y64, dat["labels"], np.arange(n))
    assert abs(mcmc_b._logLik - want) / abs(want) < LL_RTOL


@pytest.mark.parametrize("shape", [(40000, 800, [50, 5], 10, "ReLU", 1, 0.01), (40000, 1300, [32, 8], 6, "tanh", 2, 0.01), (36000, 1800, [20], 4, "tanh", 2, 0.01),
                                   (40000, 800, [50, 5], 10, "ReLU", 1, 0.06), (40000, 900, [40, 6], 0, "tanh", 2, 0.01)],
                         ids=["default-net-800", "32-8-on-1300", "one-hidden-layer", "proposals-of-2.5k-entries", "regression-2-targets"])
def test_streamed_candidates_do_not_change_the_chain(shape):
    """The fused pass of the weight-streamed path evaluates up to three candidates per read of X (wide_gemm_kernel<..., D>): the chain
    is the same one - decisions, weights, log-likelihood bit for bit - with one, two or three candidates per pass (a tiling's builds
    all accumulate alike: DMAX in wide_gemm_kernel), and fewer passes are spent on it.  Proposals of more than 2048 entries have their
    candidate images kept by the launches over all compute units (wide_cand_*_kernel), narrower ones with one candidate per pass by the
    chain step itself; the regression case runs the Gaussian epilogue (two targets, empirical error)."""
    from test_hip_sampler import quiet
    n, f, hidden, c, fun, bias, uf = shape
    if c:
        dat = cases.classification_data(5, n, f, c, n_test=0)
        extra = {}
    else:
        dat = cases.regression_data(5, n, f, 2, 0)
        extra = dict(estimation_mode="regression", empirical_error=True)
    runs = []
    for d in (1, 2, 3):
        np.random.seed(31)
        bnn = quiet(bn.npBNN, dat, n_nodes=hidden, actFun=bn.ActFun(fun=fun), use_bias_node=bias, prior_f=1, p_scale=1, **extra)
        kw = dict(estimate_error=False) if not c else {}
        m = bn.MCMC(bnn, update_f=[uf] * (len(hidden) + 1), update_ws=[0.04 if uf < 0.05 else 0.01] * (len(hidden) + 1), n_iteration=100000, **kw)
        m.n_candidates = d
        for _ in range(3):
            m.run_steps(bnn, 50)
        assert m._backend.ctx.is_wide()
        runs.append((bnn, m))
    (b1, m1), (b2, m2), (b3, m3) = runs
    assert m1._device_passes == 150
    assert m3._device_passes <= m2._device_passes < 150
    assert 0 < sum(m1._last_accepted_mem) < 100
    for b, m in ((b2, m2), (b3, m3)):
        assert m._last_accepted_mem == m1._last_accepted_mem
        assert m._logLik == m1._logLik and m._logPrior == m1._logPrior
        for wa, wb in zip(b1._w_layers, b._w_layers):
            np.testing.assert_array_equal(wa, wb)


@pytest.mark.parametrize("name", list(cases.TRACES))
def test_forced_chain_follows_the_reference_trace(name, golden_dir, monkeypatch):
    """The reference's golden Metropolis-Hastings traces with the networks forced onto the streamed path: mh_step free-running
    against the reference's decisions, run_steps against mh_step."""
    from test_hip_sampler import build
    monkeypatch.setenv("NPBNN_FORCE_WIDE", "1")
    cfg = cases.TRACES[name]
    g = np.load(os.path.join(golden_dir, "trace_%s.npz" % name))
    bnn, mcmc = build(cfg)
    assert mcmc._backend.ctx.is_wide()
    np.testing.assert_allclose(mcmc._logLik, g["init"][0], rtol=2e-6)
    np.testing.assert_allclose([mcmc._accuracy, mcmc._test_accuracy], g["init"][2:4], rtol=1e-4)
    rows = g["rows"]
    n_match = 0
    for it in range(min(cfg["steps"], 200)):
        mcmc.mh_step(bnn)
        if mcmc._last_accepted != int(rows[it, 2]):
            break
        np.testing.assert_allclose(mcmc._logLik, rows[it, 3], rtol=2e-6)
        n_match += 1
    assert n_match >= min(150, cfg["steps"]), "chains diverged after %d iterations" % n_match
    bnn_b, mcmc_b = build(cfg)
    mcmc_b.run_steps(bnn_b, 100)
    mcmc_b.run_steps(bnn_b, n_match - 100)
    assert mcmc_b._last_accepted_mem[:n_match] == mcmc._last_accepted_mem[:n_match]


@pytest.mark.parametrize("seed", range(12))
def test_forced_device_chain_is_the_mh_step_loop_on_random_shapes(seed, monkeypatch):
    """tests/test_hip_sampler.py's random networks (depths, activations, bias modes, priors, both estimation modes, tempered and
    heated chains) forced onto the streamed path: run_steps against the mh_step loop, to the bit."""
    from test_hip_sampler import _random_case
    monkeypatch.setenv("NPBNN_FORCE_WIDE", "1")
    dat, model_kw, act_kw, sampler_kw = _random_case(1000 + seed)
    (bnn_a, mcmc_a), (bnn_b, mcmc_b) = _chains(dat, model_kw, act_kw, sampler_kw)
    assert mcmc_a._backend.ctx.is_wide()
    for _ in range(90):
        mcmc_a.mh_step(bnn_a)
    for _ in range(3):
        mcmc_b.run_steps(bnn_b, 30)
    assert mcmc_b._device_iterations == 90, "the device chain did not take these iterations: %s %s" % (model_kw, sampler_kw)
    assert mcmc_a._last_accepted_mem == mcmc_b._last_accepted_mem, (model_kw, act_kw, sampler_kw)
    for wa, wb in zip(bnn_a._w_layers, bnn_b._w_layers):
        np.testing.assert_array_equal(wa, wb)
    np.testing.assert_allclose(mcmc_b._logLik, mcmc_a._logLik, rtol=1e-12)
    np.testing.assert_allclose(mcmc_b._logPrior, mcmc_a._logPrior, rtol=1e-11)


# ---- the rest of the reference's fixtures with every network forced onto the streamed path (the existing tests' bodies, re-run) ----
@pytest.mark.parametrize("advance", ["mh_step", "run_steps"])
@pytest.mark.parametrize("name", list(cases.OPTION_TRACES))
def test_forced_option_traces_follow_the_reference(name, advance, golden_dir, monkeypatch):
    """G9 (one reference chain per sampler option: trainable slopes, indicators, hyper-priors, the other proposals, sigma proposals,
    weights, priors, tempering, predicted sigma, a count likelihood) on the streamed path: tests/test_hip_options.py's check."""
    import test_hip_options as t
    monkeypatch.setenv("NPBNN_FORCE_WIDE", "1")
    t.test_free_running_chain_follows_the_reference_under_every_option(name, advance, np.load(os.path.join(golden_dir, "options.npz")))


def test_forced_mc3_follows_the_reference(golden_dir, tmp_path, monkeypatch):
    """G5 (four chains, swaps every 100 iterations) on the streamed path: exchange runs on the serial schedule."""
    import test_hip_sampler as t
    monkeypatch.setenv("NPBNN_FORCE_WIDE", "1")
    t.test_mc3_four_chains_on_one_gpu_follow_reference(golden_dir, tmp_path)


@pytest.mark.parametrize("case", cases.POSTERIOR_CASES, ids=lambda c: c["name"])
def test_forced_posterior_prediction_matches_the_reference(case, golden_dir, monkeypatch):
    """G7 (posterior samples replayed: get_posterior_cat_prob) on the streamed path: one weight set per pass."""
    import test_hip_posterior as t
    monkeypatch.setenv("NPBNN_FORCE_WIDE", "1")
    t.test_get_posterior_cat_prob_matches_reference(case, golden_dir)
