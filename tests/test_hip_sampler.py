"""GPU tests of the sampler: MCMC.mh_step (one device evaluation per proposal) and MCMC.run_steps
(device-resident chain with pre-drawn random numbers) against each other and against the reference's
golden Metropolis-Hastings traces."""
import contextlib
import copy
import io
import os

import numpy as np
import pytest

import cases
import npbnn_amd as bn

pytestmark = pytest.mark.gpu


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


def build(cfg, **mcmc_extra):
    if cfg["kind"] == "classification":
        dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
        extra = {}
    else:
        dat = cases.regression_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["k"], cfg["n_test"])
        extra = dict(estimation_mode="regression", empirical_error=cfg.get("empirical_error", False))
    np.random.seed(1234)
    bnn = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], actFun=bn.ActFun(fun=cfg["fun"]), use_bias_node=cfg["bias"],
                prior_f=1, p_scale=1, seed=1234, init_std=0.1, **extra)
    kw = dict(cfg["mcmc"])
    kw.update(mcmc_extra)
    return bnn, bn.MCMC(bnn, **kw)


@pytest.mark.parametrize("name", list(cases.TRACES))
def test_free_running_chain_follows_reference_trace(name, golden_dir):
    """Same seeds -> the float32 device chain follows the float64 reference accept/reject sequence.  A decision
    may flip where |logPost' - logPost - log u| is below the float32 noise of the likelihood (~1e-3 here), after
    which the chains legitimately diverge; require a long common prefix and a matching initial state."""
    cfg = cases.TRACES[name]
    g = np.load(os.path.join(golden_dir, "trace_%s.npz" % name))
    bnn, mcmc = build(cfg)
    np.testing.assert_allclose(mcmc._logLik, g["init"][0], rtol=2e-6)
    np.testing.assert_allclose(mcmc._logPrior, g["init"][1], rtol=1e-12)
    np.testing.assert_allclose([mcmc._accuracy, mcmc._test_accuracy], g["init"][2:4], rtol=1e-4)
    rows = g["rows"]
    n_match = 0
    for it in range(cfg["steps"]):
        mcmc.mh_step(bnn)
        if mcmc._last_accepted != int(rows[it, 2]):
            break
        np.testing.assert_allclose(mcmc._logLik, rows[it, 3], rtol=2e-6)
        n_match += 1
    assert n_match >= min(150, cfg["steps"]), "chains diverged after %d iterations" % n_match
    if n_match == cfg["steps"]:
        for i, w in enumerate(bnn._w_layers):
            np.testing.assert_array_equal(w, g["wfinal_%d" % i])     # weights are float64 on the host: bit-equal
        np.testing.assert_allclose(mcmc._accuracy, rows[-1, 5], rtol=1e-3)


@pytest.mark.parametrize("name", list(cases.TRACES))
@pytest.mark.parametrize("randomize_seed", [False, True])
def test_run_steps_equals_mh_step_loop(name, randomize_seed):
    """The device-resident chain and the host loop run the same kernels on the same proposals."""
    cfg = cases.TRACES[name]
    bnn_a, mcmc_a = build(cfg, randomize_seed=randomize_seed, mcmc_id=2)
    bnn_b, mcmc_b = build(cfg, randomize_seed=randomize_seed, mcmc_id=2)
    n = 230                                   # crosses adaptation boundaries of the classification trace
    for _ in range(n):
        mcmc_a.mh_step(bnn_a)
    mcmc_b.run_steps(bnn_b, 100)
    mcmc_b.run_steps(bnn_b, n - 100)
    assert mcmc_b._current_iteration == mcmc_a._current_iteration == n
    assert mcmc_a._last_accepted_mem == mcmc_b._last_accepted_mem
    assert mcmc_a._acceptance_rate == mcmc_b._acceptance_rate
    np.testing.assert_allclose(mcmc_b._logLik, mcmc_a._logLik, rtol=1e-12)
    np.testing.assert_allclose(mcmc_b._logPrior, mcmc_a._logPrior, rtol=1e-11)
    for wa, wb in zip(bnn_a._w_layers, bnn_b._w_layers):
        np.testing.assert_array_equal(wa, wb)
    np.testing.assert_array_equal(mcmc_a._update_n, mcmc_b._update_n)
    np.testing.assert_allclose(mcmc_a._accuracy, mcmc_b._accuracy)
    if cfg["kind"] == "regression":
        np.testing.assert_allclose(bnn_a._error_prm, bnn_b._error_prm, rtol=1e-12)
    if not randomize_seed:
        assert mcmc_a._rs.random() == mcmc_b._rs.random()     # generator advanced identically


def test_run_steps_with_block_mask_and_bounds():
    """Masked (block-sparse) layers and a uniform prior with reflecting bounds, device loop vs host loop."""
    dat = cases.regression_data(9, 1500, 24, 1)
    res = []
    for mode in ("host", "device"):
        np.random.seed(3)
        bnn = quiet(bn.npBNN, dat, n_nodes=[8, 4], estimation_mode="regression", actFun=bn.ActFun(fun="tanh"),
                    use_bias_node=-1, prior_f=0, p_scale=0.6, empirical_error=True)
        m = bn.create_mask(bnn._w_layers, indx_input_list=[list(np.repeat(np.arange(4), 6)), [], []],
                           nodes_per_feature_list=[[2] * 4, [], []])
        quiet(bnn.apply_mask, m)
        mcmc = bn.MCMC(bnn, update_f=[0.2, 0.2, 0.5], update_ws=[0.3, 0.3, 0.3], n_iteration=1000, estimate_error=False)
        if mode == "host":
            for _ in range(120):
                mcmc.mh_step(bnn)
        else:
            mcmc.run_steps(bnn, 120)
        res.append((bnn, mcmc))
    (ba, ma), (bb, mb) = res
    assert ma._last_accepted_mem == mb._last_accepted_mem
    for wa, wb, mk in zip(ba._w_layers, bb._w_layers, ba._mask):
        np.testing.assert_array_equal(wa, wb)
        assert np.all(wa[mk == 0] == 0) and np.all(np.abs(wa) <= 0.6)
    np.testing.assert_allclose(ma._logLik, mb._logLik, rtol=1e-12)


def test_driver_and_logger_end_to_end(tmp_path):
    """bnn_classify.py call sequence (config 1 shape): run_mcmc + postLogger on the device backend."""
    cfg = cases.TRACES["cfg1"]
    bnn, mcmc = build(cfg, n_iteration=300, sampling_f=50, print_f=100)
    logger = bn.postLogger(bnn, filename="BNN_cv0", wdir=str(tmp_path), log_all_weights=0)
    quiet(bn.run_mcmc, bnn, mcmc, logger)
    rows = np.loadtxt(logger._logfile, skiprows=1)
    assert rows.shape[0] == 6 and rows[-1, 0] == 300
    header = open(logger._logfile).readline().split()
    assert header[:6] == ["it", "posterior", "likelihood", "prior", "accuracy", "test_accuracy"]
    b2, m2, lg = bn.load_obj(logger._pklfile)
    assert len(lg._post_weight_samples) == 6
    assert m2._accuracy == mcmc._accuracy and mcmc._accuracy > 0.22


def test_mc3_four_chains_on_one_gpu_follow_reference(golden_dir, tmp_path):
    """bnn_runner_MC3.py call sequence: 4 chains in this process (each with its own resident context), device-resident
    chains between swaps.  The float32 chains follow the float64 reference swap sequence for the first swaps."""
    cfg = cases.MC3_TRACE
    g = np.load(os.path.join(golden_dir, "mc3.npz"))
    dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
    np.random.seed(1234)
    bnn = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], use_bias_node=cfg["bias"], seed=1, init_std=0.1)
    logger = bn.postLogger(bnn, filename="MC3", wdir=str(tmp_path), log_all_weights=0)
    mc3 = quiet(bn.MC3, bnn, logger=logger, n_post_samples=10, sampling_f=cfg["swap_frequency"], n_iteration=cfg["n_iteration"],
                n_chains=cfg["n_chains"], swap_frequency=cfg["swap_frequency"], verbose=0)
    np.testing.assert_array_equal(mc3.rseeds, g["rseeds"])
    quiet(mc3.run_mcmc)
    assert len(mc3.swap_log) == 30
    accepted = [i for i, s in enumerate(mc3.swap_log) if s[4]]
    want = [int(r[0]) for r in g["swapped"]]
    n_common = 0
    for a, b in zip(accepted, want):
        if a != b:
            break
        n_common += 1
    assert n_common >= 5, "swap sequences diverged immediately: %s vs %s" % (accepted, want)
    rows = np.loadtxt(logger._logfile, skiprows=1)
    assert rows.shape[0] >= 25                      # the cold chain is logged after every swap interval
    np.testing.assert_allclose(rows[:3, 2], g["log_rows"][:3, 2], rtol=2e-6)     # likelihood column of the first samples
    temps = sorted(c[1]._temperature for c in mc3.singleChainArgs)
    np.testing.assert_allclose(temps, sorted(g["temperatures0"]))


def test_rccl_communicator_single_rank():
    """The native RCCL communicator of the C ABI (one rank: all-gather and broadcast are identities)."""
    from npbnn_amd.comm import RcclComm
    comm = RcclComm(rank=0, world_size=1, device=0)
    out = comm.allgather_f64(np.array([1.5, -2.25]))
    np.testing.assert_array_equal(out, [[1.5, -2.25]])
    np.testing.assert_array_equal(comm.bcast_i64(np.array([3, 1, 1])), [3, 1, 1])
    assert comm.bcast_obj({"a": [1, 2, 3]}) == {"a": [1, 2, 3]}
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("name", ["cfg1", "cfg4s"])
def test_speculative_passes_do_not_change_the_chain(name):
    """1, 2 or 3 candidates per pass over the data, decided between the passes (serial schedule) or inside the launch
    that already evaluates the next pass (overlapped schedule): identical accept/reject sequence, weights and
    log-likelihood - the speculation only changes how many passes the same chain needs.  Schedule 3 is the overlapped
    schedule with the launches alternating between two streams, ordered by device-side flags instead of kernel boundaries, 4 its
    persistent form (one launch loops over the passes), 5 the persistent launch with the decision between the passes."""
    cfg = cases.TRACES[name]
    out = []
    for d, sched in ((1, 1), (2, 1), (3, 1), (1, 2), (3, 2), (3, 0), (3, 3), (1, 3), (2, 3), (3, 4), (1, 4), (2, 4), (3, 5), (1, 5), (2, 5)):
        bnn, mcmc = build(cfg)
        mcmc.n_candidates = d
        mcmc.device_schedule = sched
        mcmc.SUB_BATCH = 64
        mcmc.run_steps(bnn, 300)
        out.append((bnn, mcmc, d, sched))
    (b1, m1, _, _) = out[0]
    assert m1._device_passes == 300
    assert sum(m1._last_accepted_mem) > 0
    for b, m, d, sched in out[1:]:
        assert m._last_accepted_mem == m1._last_accepted_mem
        assert m._logLik == m1._logLik and m._logPrior == m1._logPrior
        for wa, wb in zip(b1._w_layers, b._w_layers):
            np.testing.assert_array_equal(wa, wb)
        if sched in (2, 3, 4):
            assert m._device_void_passes > 0                # accepts happened, so passes were dropped ...
        if d == 1:
            assert m._device_passes == 300                  # ... and never counted
            continue
        assert m._device_passes < 300                       # fewer passes over the data for the same 300 iterations


@pytest.mark.parametrize("kind", ["regression", "classification"])
def test_device_chain_with_many_tiles_per_wave(kind):
    """200k rows (a dozen tiles per wavefront, every row-aux slot in rotation, three candidates per pass in the overlapped
    schedule) against the host loop (one single-candidate evaluation per proposal): same decisions, same state."""
    n, f = 200_000, 48
    if kind == "regression":
        dat = cases.regression_data(5, n, f, 3, 0)
        extra = dict(estimation_mode="regression", empirical_error=True)
        kw = dict(update_f=[0.02, 0.02, 0.05], estimate_error=False)
    else:
        dat = cases.classification_data(5, n, f, 7, 0)
        dat["instance_weights"] = None
        extra = {}
        kw = dict(update_f=[0.02, 0.02, 0.05])
    out = []
    for _ in range(2):
        np.random.seed(1234)
        bnn = quiet(bn.npBNN, {k: v for k, v in dat.items() if k != "instance_weights"}, n_nodes=[24, 6],
                    actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1, **extra)
        out.append((bnn, bn.MCMC(bnn, **kw)))
    (ba, ma), (bb, mb) = out
    steps = 90
    for _ in range(steps):
        ma.mh_step(ba)
    mb.run_steps(bb, steps)
    assert mb._device_void_passes + mb._device_passes > 0
    assert ma._last_accepted_mem == mb._last_accepted_mem
    assert sum(ma._last_accepted_mem) > 0
    np.testing.assert_allclose(mb._logLik, ma._logLik, rtol=1e-11)
    for wa, wb in zip(ba._w_layers, bb._w_layers):
        np.testing.assert_array_equal(wa, wb)


@pytest.mark.parametrize("randomize_seed", [False, True])
def test_run_steps_with_an_estimated_error_parameter(randomize_seed):
    """Regression with the error parameter estimated (MCMC's default): after the first 10 % of the iterations every
    proposal multiplies sigma by pre-drawn factors (BNN_env.py:435-444).  The device-resident chain carries sigma as chain
    state (sigma' = sigma * factors per candidate, Hastings term in the accept test) and must reproduce the host loop."""
    cfg = dict(cases.TRACES["cfg4s"], empirical_error=False)
    extra = dict(randomize_seed=randomize_seed, mcmc_id=2, estimate_error=True, n_iteration=1000, adapt_f=0, adapt_fM=1)
    bnn_a, mcmc_a = build(cfg, **extra)
    bnn_b, mcmc_b = build(cfg, **extra)
    assert mcmc_a._estimate_error == 100
    n = 300
    for _ in range(n):
        mcmc_a.mh_step(bnn_a)
    seen = []
    real = mcmc_b._backend.run_chain
    mcmc_b._backend.run_chain = lambda w, **kw: (seen.append((len(kw["cnt"]), kw.get("sigma_mult") is not None)), real(w, **kw))[1]
    mcmc_b.run_steps(bnn_b, 90)
    mcmc_b.run_steps(bnn_b, n - 90)
    assert sum(k for k, prop in seen) == n and sum(k for k, prop in seen if prop) == n - 101
    assert mcmc_a._last_accepted_mem == mcmc_b._last_accepted_mem and sum(mcmc_a._last_accepted_mem) > 5
    np.testing.assert_array_equal(np.asarray(bnn_a._error_prm, dtype=float), np.asarray(bnn_b._error_prm, dtype=float))
    assert not np.all(np.asarray(bnn_b._error_prm, dtype=float) == 1.0)
    np.testing.assert_allclose(mcmc_b._logLik, mcmc_a._logLik, rtol=1e-12)
    np.testing.assert_allclose(mcmc_b._logPrior, mcmc_a._logPrior, rtol=1e-11)
    for wa, wb in zip(bnn_a._w_layers, bnn_b._w_layers):
        np.testing.assert_array_equal(wa, wb)


@pytest.mark.parametrize("update_f,min_accepts", [(0.002, 300), (0.05, 20)])
def test_two_stream_schedule_at_config2_size(update_f, min_accepts):
    """The flag-ordered two-stream schedule on BASELINE config-2 shapes with learnable labels: many accepts (every one of
    them patches the global weight image while the next launch's workgroups are already arriving) and few - against the
    one-stream overlapped schedule, bit for bit."""
    rs = np.random.default_rng(0)
    n, f, c = 100_000, 256, 10
    x = rs.standard_normal((n, f)).astype(np.float32)
    proj = rs.standard_normal((f, c)) / np.sqrt(f)
    y = np.argmax(x @ proj + 0.5 * rs.standard_normal((n, c)), axis=1)
    dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
    def run(sched):
        np.random.seed(1234)
        bnn = quiet(bn.npBNN, dat, n_nodes=[32, 8], actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
        m = bn.MCMC(bnn, update_f=[update_f] * 3)
        m.device_schedule = sched
        m.run_steps(bnn, 600)
        m.run_steps(bnn, 1400)
        return bnn, m

    ba, ma = run(2)
    assert ma._device_schedule_used == 2
    # The two-stream schedule is opt-in because a device-side wait may time out when the streams do not get hardware queues of
    # their own (the batch is then repeated on one stream: same chain).  A build that ALWAYS falls back must not pass, a single
    # time-out must not fail: up to three fresh chains, one of them has to go through on two streams from start to end.
    import warnings
    clean = None
    for _ in range(3):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            bb, mb = run(3)
        assert ma._last_accepted_mem == mb._last_accepted_mem          # (fallback or not: the same chain)
        if mb._backend.ctx.sync_fallbacks == 0:
            clean = (bb, mb)
            break
    assert clean is not None, "the two-stream schedule timed out in three runs out of three"
    bb, mb = clean
    assert mb._device_schedule_used == 3
    assert ma._last_accepted_mem == mb._last_accepted_mem
    assert (ma._logLik, ma._logPrior) == (mb._logLik, mb._logPrior)
    for wa, wb in zip(ba._w_layers, bb._w_layers):
        np.testing.assert_array_equal(wa, wb)
    fresh = mb._backend.evaluate(bb._w_layers, None)["loglik"]
    np.testing.assert_allclose(mb._logLik, fresh, rtol=1e-12)       # the chain's image was the true one all along
    assert mb._device_passes < 2000 and ma._device_passes == mb._device_passes
    assert mb._device_void_passes >= min_accepts // 4       # an accept voids the pass that was being evaluated


def test_two_stream_schedule_times_out_cleanly(monkeypatch):
    """A step workgroup that never reports back (provoked through a test hook): every wait of the two-stream schedule is
    bounded, the batch ends with NPBNN_E_SYNC and its state untouched, the same batch runs on one stream, the context keeps
    that schedule off from then on - and the chain is the one it would have been."""
    cfg = cases.TRACES["cfg1"]
    bnn_a, mcmc_a = build(cfg, adapt_f=0, adapt_fM=1)
    bnn_b, mcmc_b = build(cfg, adapt_f=0, adapt_fM=1)
    mcmc_a.device_schedule = 2
    mcmc_a.run_steps(bnn_a, 400)
    ctx_b = mcmc_b._backend.ctx
    assert ctx_b._lib.npbnn_debug_sync_skip_(ctx_b._ctx, 7) == 0        # (diagnostic entry point, not part of the ABI)
    mcmc_b.device_schedule = 3
    with pytest.warns(UserWarning, match="flag-ordered"):
        mcmc_b.run_steps(bnn_b, 200)
    assert ctx_b.sync_fallbacks == 1
    assert mcmc_b._device_schedule_used == 2          # the batch was repeated on one stream
    mcmc_b.run_steps(bnn_b, 200)
    assert mcmc_b._device_schedule_used == 2          # and the two-stream schedule stays off for this context
    assert mcmc_a._last_accepted_mem == mcmc_b._last_accepted_mem
    assert (mcmc_a._logLik, mcmc_a._logPrior) == (mcmc_b._logLik, mcmc_b._logPrior)
    for wa, wb in zip(bnn_a._w_layers, bnn_b._w_layers):
        np.testing.assert_array_equal(wa, wb)


@pytest.mark.parametrize("schedule", [4, 5])
def test_persistent_schedules_time_out_cleanly(schedule):
    """The same for the two persistent forms: a step workgroup that stops reporting in the middle of a launch (test hook) leaves the
    evaluating workgroups in their bounded waits - the overlapped form's, and the decision-between-passes form's, whose wait also
    fetches the descriptor the flag names - the batch ends with NPBNN_E_SYNC, runs again on kernel boundaries, and the chain is the
    one it would have been."""
    cfg = cases.TRACES["cfg1"]
    bnn_a, mcmc_a = build(cfg, adapt_f=0, adapt_fM=1)
    bnn_b, mcmc_b = build(cfg, adapt_f=0, adapt_fM=1)
    bnn_c, mcmc_c = build(cfg, adapt_f=0, adapt_fM=1)
    mcmc_c.device_schedule = schedule
    mcmc_c.run_steps(bnn_c, 200)
    if mcmc_c._device_schedule_used != schedule:
        pytest.skip("schedule %d does not run on this chain here (ran %d)" % (schedule, mcmc_c._device_schedule_used))
    mcmc_a.device_schedule = 2
    mcmc_a.run_steps(bnn_a, 400)
    ctx_b = mcmc_b._backend.ctx
    assert ctx_b._lib.npbnn_debug_sync_skip_(ctx_b._ctx, 7) == 0        # (diagnostic entry point, not part of the ABI)
    mcmc_b.device_schedule = schedule
    with pytest.warns(UserWarning, match="flag-ordered"):
        mcmc_b.run_steps(bnn_b, 200)
    assert ctx_b.sync_fallbacks == 1
    assert mcmc_b._device_schedule_used == 2          # the batch was repeated on kernel boundaries
    mcmc_b.run_steps(bnn_b, 200)
    assert mcmc_a._last_accepted_mem == mcmc_b._last_accepted_mem
    assert (mcmc_a._logLik, mcmc_a._logPrior) == (mcmc_b._logLik, mcmc_b._logPrior)
    for wa, wb in zip(bnn_a._w_layers, bnn_b._w_layers):
        np.testing.assert_array_equal(wa, wb)


@pytest.mark.parametrize("hyper_p", [1, 2, 3])
def test_run_steps_with_hyper_priors_is_the_mh_step_loop(hyper_p):
    """Hyper-priors (npBNN.sample_prior_scale, BNN_env.py:196-221): one scale per layer, per input node or per weight, re-drawn by
    gibbs_step between batches; the device chain takes them as constants of a batch (per-weight scale vector) and must give the
    chain of the mh_step loop: same accept / reject sequence, same weights, same scales, log prior to rounding."""
    cfg = cases.TRACES["cfg1"]
    res = []
    for mode in ("host", "device"):
        dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
        np.random.seed(1234)
        bnn = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], actFun=bn.ActFun(fun=cfg["fun"]), use_bias_node=cfg["bias"],
                    prior_f=1, p_scale=1, seed=1234, init_std=0.1, hyper_p=hyper_p)
        mcmc = bn.MCMC(bnn, **dict(cfg["mcmc"], adapt_f=0, adapt_fM=1))
        np.random.seed(99)                         # gibbs_step draws from the global stream
        before = mcmc._device_iterations
        for _ in range(3):
            if mode == "host":
                for _ in range(60):
                    mcmc.mh_step(bnn)
            else:
                mcmc.run_steps(bnn, 60)
            mcmc.gibbs_step(bnn)
        assert (mcmc._device_iterations - before == 180) == (mode == "device")
        res.append((bnn, mcmc))
    (ba, ma), (bb, mb) = res
    assert ma._current_iteration == mb._current_iteration == 183
    assert ma._last_accepted_mem == mb._last_accepted_mem
    for sa, sb in zip(ba._prior_scale, bb._prior_scale):
        np.testing.assert_array_equal(np.asarray(sa, dtype=float), np.asarray(sb, dtype=float))
    for wa, wb in zip(ba._w_layers, bb._w_layers):
        np.testing.assert_array_equal(wa, wb)
    np.testing.assert_allclose([mb._logLik, mb._logPrior], [ma._logLik, ma._logPrior], rtol=1e-12)


@pytest.mark.parametrize("schedule", [0, 1, 2])
@pytest.mark.parametrize("randomize_seed", [False, True])
def test_run_steps_with_trainable_slopes_is_the_mh_step_loop(schedule, randomize_seed):
    """Parametric ReLU (ActFun(fun="genReLU", trainable=True), BNN_env.py:416-421,502-503): every iteration proposes new slopes
    from the accepted ones, evaluates its weight proposal with them and adds their exponential prior to the proposal's log
    prior.  The device chain carries the slopes as chain state (each candidate of a pass its own) and must give the chain of
    the mh_step loop: same accept / reject sequence, same weights, same accepted and last-proposed slopes, log prior to
    rounding - including the reference's quirk that the prior stored by MCMC.__init__ lacks the slope term."""
    cfg = cases.TRACES["cfg1"]
    res = []
    for mode in ("host", "device"):
        dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
        np.random.seed(1234)
        act = bn.ActFun(fun="genReLU", prm=np.full(len(cfg["n_nodes"]), 0.02), trainable=True)
        bnn = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], actFun=act, use_bias_node=cfg["bias"], prior_f=1, p_scale=1, seed=1234,
                    init_std=0.1)
        mcmc = bn.MCMC(bnn, randomize_seed=randomize_seed, mcmc_id=3, **dict(cfg["mcmc"], adapt_f=0, adapt_fM=1))
        mcmc.device_schedule = schedule
        before = mcmc._device_iterations
        if mode == "host":
            for _ in range(170):
                mcmc.mh_step(bnn)
        else:
            mcmc.run_steps(bnn, 70)
            mcmc.run_steps(bnn, 100)
        assert (mcmc._device_iterations - before == 170) == (mode == "device")
        res.append((bnn, mcmc))
    (ba, ma), (bb, mb) = res
    assert ma._current_iteration == mb._current_iteration == 170
    assert sum(ma._last_accepted_mem) > 3, "the comparison needs accepted proposals"
    assert ma._last_accepted_mem == mb._last_accepted_mem
    np.testing.assert_array_equal(ba._act_fun._acc_prm, bb._act_fun._acc_prm)
    np.testing.assert_array_equal(ba._act_fun._prm, bb._act_fun._prm)
    assert not np.array_equal(ba._act_fun._acc_prm, np.full(len(cfg["n_nodes"]), 0.02)), "the slopes should have moved"
    for wa, wb in zip(ba._w_layers, bb._w_layers):
        np.testing.assert_array_equal(wa, wb)
    np.testing.assert_allclose([mb._logLik, mb._logPrior], [ma._logLik, ma._logPrior], rtol=1e-12)
    if not randomize_seed:
        assert ma._rs.random() == mb._rs.random()


def test_persistent_schedule_never_times_out_in_a_million_iterations():
    """The persistent form of the overlapped schedule is what the library picks by itself for a chain alone on its GPU.  Its
    device-side waits are bounded, and a time-out is survivable (NPBNN_E_SYNC: the batch is repeated on kernel boundaries) - but
    on a GPU the chain has to itself none may happen: 10^6 iterations of config-2 shapes at a few per cent acceptance, in calls
    of 100 000 and of 100, and the chain is the one the schedule on kernel boundaries gives."""
    rs = np.random.default_rng(0)
    n, f, c = 100_000, 256, 10
    x = rs.standard_normal((n, f)).astype(np.float32)
    proj = rs.standard_normal((f, c)) / np.sqrt(f)
    y = np.argmax(x @ proj + 0.5 * rs.standard_normal((n, c)), axis=1)
    dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))

    def chain():
        np.random.seed(1234)
        bnn = quiet(bn.npBNN, dat, n_nodes=[32, 8], actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
        return bnn, bn.MCMC(bnn, update_f=[0.02] * 3)

    bnn, m = chain()                               # schedule left to the library
    for _ in range(9):
        m.run_steps(bnn, 100_000)
        assert m._device_schedule_used in (4, 5) and m._backend.ctx.sync_fallbacks == 0      # (5: batches after a stretch of many accepts)
    for _ in range(1000):
        m.run_steps(bnn, 100)
    assert m._device_schedule_used in (4, 5) and m._backend.ctx.sync_fallbacks == 0 and m._current_iteration == 1_000_000
    assert 0.001 < m._device_accepted / 1e6 < 0.3
    # the same chain on kernel boundaries (first 20 000 iterations)
    (ba, ma), (bb, mb) = chain(), chain()
    ma.device_schedule, mb.device_schedule = 2, 4
    ma.run_steps(ba, 20_000)
    mb.run_steps(bb, 20_000)
    assert mb._device_schedule_used == 4 and ma._last_accepted_mem == mb._last_accepted_mem
    assert (ma._logLik, ma._logPrior) == (mb._logLik, mb._logPrior)
    for wa, wb in zip(ba._w_layers, bb._w_layers):
        np.testing.assert_array_equal(wa, wb)


# ---- the general device chain (npbnn_chain_run_general): proposals that change more than a list of weights ----------------------
def _general_pair(kind, randomize_seed):
    import contextlib, io
    kw = dict(update_function=bn.UpdateNormal, n_nodes=[24, 9], freq_indicator=0, feature_indicators=None, w_bound=np.inf, unit_sums=False,
              regression=False, mcmc={})
    kw.update({"fixed_normal": dict(update_function=bn.UpdateFixedNormal, w_bound=0.4),
               "normal_normalized": dict(update_function=bn.UpdateNormalNormalized, unit_sums=True),
               "weight_indicators": dict(n_nodes=[24, 9, 6], freq_indicator=0.3),
               "feature_indicators": dict(feature_indicators=True),
               "feature_and_weight_indicators": dict(n_nodes=[24, 9, 6], freq_indicator=0.3, feature_indicators=True,
                                                     update_function=bn.UpdateFixedNormal, w_bound=0.5),
               "regression_sigma_and_features": dict(regression=True, feature_indicators=True, mcmc=dict(estimate_error=True))}[kind])
    out = []
    for _ in range(2):
        if kw["regression"]:
            dat = cases.regression_data(9, 3000, 40, 2, 200)
            extra = dict(estimation_mode="regression")
        else:
            dat = cases.classification_data(9, 3000, 40, 4, 200)
            extra = {}
        np.random.seed(1234)
        init = None
        if kw["unit_sums"]:
            init = [np.abs(w) / np.sum(np.abs(w)) for w in bn.init_weight_prm(kw["n_nodes"], 40, 4, bias_node=2)]
        bnn = quiet(bn.npBNN, dat, n_nodes=kw["n_nodes"], use_bias_node=2, actFun=bn.ActFun(fun="tanh"), freq_indicator=kw["freq_indicator"],
                    feature_indicators=kw["feature_indicators"], w_bound=kw["w_bound"], init_weights=init, **extra)
        nl = bnn._n_layers
        mcmc = bn.MCMC(bnn, update_function=kw["update_function"], update_f=[0.003 if kw["update_function"] is bn.UpdateFixedNormal else 0.05] * nl, update_ws=[0.02 if kw["unit_sums"] else 0.05] * nl,
                       randomize_seed=randomize_seed, mcmc_id=3, n_iteration=1000, adapt_stop=20, **kw["mcmc"])
        out.append((bnn, mcmc))
    return out


@pytest.mark.parametrize("kind", ["fixed_normal", "normal_normalized", "weight_indicators", "feature_indicators",
                                  "feature_and_weight_indicators", "regression_sigma_and_features"])
@pytest.mark.parametrize("randomize_seed", [False, True])
def test_run_steps_on_the_general_device_chain_is_the_mh_step_loop(kind, randomize_seed):
    """UpdateFixedNormal / UpdateNormalNormalized (np_bnn/BNN_mcmc.py:27-42,71-82), weight indicators (BNN_env.py:457-464), feature
    indicators (:424-433): run_steps builds every candidate in full on the device (npbnn_chain_run_general) and must leave what the
    same number of mh_step calls leave - weights, indicators and feature indicators exactly, log values to rounding - having
    consumed the chain's Generator and numpy's global stream identically."""
    (ba, ma), (bb, mb) = _general_pair(kind, randomize_seed)
    assert mb._device_mode(bb, 5) == "general"
    n = 150
    np.random.seed(77)
    for _ in range(n):
        ma.mh_step(ba)
    state_a = np.random.get_state()[1].copy()
    np.random.seed(77)
    mb.SUB_BATCH = 40
    mb.run_steps(bb, 60)
    mb.run_steps(bb, n - 60)
    assert ma._current_iteration == mb._current_iteration == n and mb._device_iterations == n
    assert ma._last_accepted_mem == mb._last_accepted_mem and sum(ma._last_accepted_mem) >= 2
    for wa, wb in zip(ba._w_layers, bb._w_layers):
        np.testing.assert_array_equal(wa, wb)
    np.testing.assert_array_equal(ba._indicators, bb._indicators)
    if ba._feature_indicators is not None:
        np.testing.assert_array_equal(ba._feature_indicators, bb._feature_indicators)
    np.testing.assert_allclose(mb._logLik, ma._logLik, rtol=1e-11)
    np.testing.assert_allclose(mb._logPrior, ma._logPrior, rtol=1e-11)
    np.testing.assert_array_equal(np.random.get_state()[1], state_a)
    np.testing.assert_allclose(mb._accuracy, ma._accuracy, rtol=1e-9)
    if kind.startswith("regression"):
        np.testing.assert_array_equal(np.asarray(ba._error_prm, dtype=float), np.asarray(bb._error_prm, dtype=float))


def test_proposals_wider_than_the_workgroup_on_every_schedule():
    """More perturbed weights per iteration than the evaluation kernel has threads (two or three trips through every patch loop,
    in the evaluating workgroups and in the step): all schedules give the chain of the serial one."""
    cfg = dict(cases.TRACES["cfg1"], n_nodes=[32, 8], n_rows=3000)
    out = []
    for sched in (1, 2, 4, 5):
        bnn, mcmc = build(cfg, update_f=[0.5, 0.5, 0.5], update_ws=[0.001] * 3, adapt_f=0, adapt_fM=1)
        assert int(sum(mcmc._update_n)) > 2048
        mcmc.device_schedule = sched
        mcmc.run_steps(bnn, 120)
        out.append((bnn, mcmc))
    b1, m1 = out[0]
    assert 5 < sum(m1._last_accepted_mem) < 110
    for b, m in out[1:]:
        assert m._backend.ctx.sync_fallbacks == 0
        assert m._last_accepted_mem == m1._last_accepted_mem
        assert (m._logLik, m._logPrior) == (m1._logLik, m1._logPrior)
        for wa, wb in zip(b1._w_layers, b._w_layers):
            np.testing.assert_array_equal(wa, wb)


def test_thin_second_round_runs_on_fewer_waves_and_is_the_same_chain(monkeypatch):
    """plan_launch gives a workgroup whose share is just over one tile per wave two waves fewer (config 5: 12.25 tiles for 12
    waves; the lone 13th tile ran alone behind the others).  Same proposals, same decisions, same weights as on the build's
    full count - only the order of the float64 sums changes (np_bnn/BNN_env.py:467-491 is one sum over all rows either way)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench_support import workload
    from npbnn_amd import _capi
    wl = workload(5)
    out = {}
    for tag, waves in (("rule", None), ("full", "12")):
        if waves is None:
            monkeypatch.delenv("NPBNN_WAVES", raising=False)
        else:
            monkeypatch.setenv("NPBNN_WAVES", waves)
        bnn, mcmc = wl.build()
        ll0 = mcmc._logLik
        mcmc.run_steps(bnn, 300)
        out[tag] = (ll0, mcmc._logLik, mcmc._logPrior, np.concatenate([w.ravel() for w in bnn._w_layers]).copy(), mcmc._device_accepted,
                    mcmc._backend.ctx.info(_capi.INFO_WAVES_PER_BLOCK))
        mcmc._backend.close()
    a, b = out["rule"], out["full"]
    assert a[4] == b[4] and a[4] > 0                      # the same proposals were accepted
    assert np.array_equal(a[3], b[3])                     # the same weights, bit for bit
    assert abs(a[1] - b[1]) <= 1e-9 * abs(b[1]) and abs(a[0] - b[0]) <= 1e-9 * abs(b[0])
    assert a[2] == b[2]


def test_config5_as_benched_block_build_relu_full_size():
    """BASELINE.json config 5 exactly as bench.py builds it (bench_support.Config5: 50k x 512, create_mask / apply_mask -> the
    block-structured layer-0 build through npbnn_set_layer_mask, ReLU, bias on the last layer only, one Gaussian
    target): the initial evaluation and a moved state against the float64 oracle at full size, and 300 iterations of
    the device-resident chain against the mh_step loop from the same start."""
    import oracle as orc
    from bench_support import Config5
    wl = Config5()
    bnn, mcmc = wl.build()
    assert mcmc._backend.ctx.info(bn._capi.INFO_FAST_TAILS) == 1
    x64 = wl.x.astype(np.float32).astype(np.float64)

    # (block_bnns.py's settings: the error parameter stays at 1 over these iterations, BNN_env.py:375-379,435-444)
    pred0 = orc.forward(x64, bnn._w_layers, orc.Act("ReLU"), orc.out_identity)
    want0 = orc.lik_gaussian(pred0, wl.y, sig2=np.ones(1))
    assert abs(mcmc._logLik - want0) / abs(want0) < 2e-6
    bnn_h, mcmc_h = wl.build()
    for _ in range(300):
        mcmc_h.mh_step(bnn_h)
    mcmc.run_steps(bnn, 100)
    mcmc.run_steps(bnn, 200)
    assert mcmc._device_iterations == 300 and mcmc._device_schedule_used in (1, 2, 4, 5)
    assert mcmc._last_accepted_mem == mcmc_h._last_accepted_mem
    assert sum(mcmc._last_accepted_mem) > 3
    for wa, wb, mk in zip(bnn._w_layers, bnn_h._w_layers, bnn._mask):
        np.testing.assert_array_equal(wa, wb)
        assert np.all(wa[mk == 0] == 0)
    np.testing.assert_allclose(mcmc._logLik, mcmc_h._logLik, rtol=1e-12)
    par = wl.parity(bnn, mcmc)            # the moved state against the oracle at full size (what bench.py's parity leg reads)
    assert par["chain_loglik_rel_err"] < 2e-6 and par["loglik_rel_err"] < 2e-6 and par["prediction_max_abs_err"] < 2e-5
    mcmc._backend.close()


# ---- repeated dispatches of one size: the kept form of a dispatch (sampler._FastDispatch) ------------------------------
def _fixed_sigma_variant(cfg):
    """cfg4s with the error parameter estimated (MCMC's default): sigma stays 1 for the first tenth of n_iteration, BNN_env.py:375-379"""
    return dict(cfg, empirical_error=False, mcmc=dict(cfg["mcmc"], estimate_error=True))


@pytest.mark.parametrize("name", ["cfg1", "cfg4s", "cfg4s_fixed_sigma"])
@pytest.mark.parametrize("randomize_seed", [False, True])
def test_repeated_dispatches_take_the_short_way_and_are_the_mh_step_loop(name, randomize_seed):
    cfg = _fixed_sigma_variant(cases.TRACES["cfg4s"]) if name == "cfg4s_fixed_sigma" else cases.TRACES[name]
    bnn_a, mcmc_a = build(cfg, randomize_seed=randomize_seed, mcmc_id=3)
    bnn_b, mcmc_b = build(cfg, randomize_seed=randomize_seed, mcmc_id=3)
    n_calls, k = 9, 60
    for _ in range(n_calls * k):
        mcmc_a.mh_step(bnn_a)
    short = 0
    for _ in range(n_calls):
        kept = mcmc_b._fast
        mcmc_b.run_steps(bnn_b, k)
        short += kept is not None and mcmc_b._fast is kept
    assert short >= n_calls - 3, "only %d of %d dispatches took the short way" % (short, n_calls)
    assert mcmc_b._current_iteration == mcmc_a._current_iteration == n_calls * k
    assert mcmc_a._last_accepted_mem == mcmc_b._last_accepted_mem and mcmc_a._acceptance_rate == mcmc_b._acceptance_rate
    np.testing.assert_allclose(mcmc_b._logLik, mcmc_a._logLik, rtol=1e-12)
    np.testing.assert_allclose(mcmc_b._logPrior, mcmc_a._logPrior, rtol=1e-11)
    for wa, wb in zip(bnn_a._w_layers, bnn_b._w_layers):
        np.testing.assert_array_equal(wa, wb)
    if cfg["kind"] == "regression":
        np.testing.assert_allclose(bnn_a._error_prm, bnn_b._error_prm, rtol=1e-12)
    if not randomize_seed:
        assert mcmc_a._rs.random() == mcmc_b._rs.random()


def test_the_short_way_notices_what_changes_between_dispatches():
    """Between dispatches of one size: a new temperature, step sizes edited in place, weights put back by hand, the generator
    read (draws made ahead are taken back), a pickle round trip of the sampler, another batch size - the chain stays the
    mh_step loop's through all of it."""
    import pickle
    cfg = cases.TRACES["cfg1"]
    bnn_a, mcmc_a = build(cfg)
    bnn_b, mcmc_b = build(cfg)
    k = 50

    def both(f):
        f(bnn_a, mcmc_a)
        f(bnn_b, mcmc_b)

    def advance(n_calls):
        for _ in range(n_calls * k):
            mcmc_a.mh_step(bnn_a)
        for _ in range(n_calls):
            mcmc_b.run_steps(bnn_b, k)
        assert mcmc_a._last_accepted_mem == mcmc_b._last_accepted_mem, "diverged by iteration %d" % mcmc_b._current_iteration
        for wa, wb in zip(bnn_a._w_layers, bnn_b._w_layers):
            np.testing.assert_array_equal(wa, wb)
        np.testing.assert_allclose(mcmc_b._logLik, mcmc_a._logLik, rtol=1e-12)

    advance(4)
    assert mcmc_b._fast is not None
    both(lambda b, m: m.reset_temperature(0.7))
    advance(3)
    both(lambda b, m: m._update_ws[0].__imul__(0.5))                  # in place: the arrays stay the same objects
    advance(3)
    def shrink(b, m):      # weights changed behind the sampler's back; its prior brought up to date (the device chain re-sums it
        b.reset_weights([w * 0.9 for w in b._w_layers])        # at the start of every batch), its log-likelihood stale in both alike
        m._logPrior = b.calc_prior()
        m._logPost = m._logLik + m._logPrior

    both(shrink)
    advance(3)
    both(lambda b, m: m._rs.random())                                 # consumes a number from the chain's stream
    advance(3)
    mcmc_b2 = pickle.loads(pickle.dumps(mcmc_b))
    assert mcmc_b2._fast is None
    mcmc_b2._bnn = bnn_b
    mcmc_b = mcmc_b2
    advance(3)
    for _ in range(77):
        mcmc_a.mh_step(bnn_a)
    mcmc_b.run_steps(bnn_b, 77)
    advance(20)                       # (after a call of another size the general path waits a few dispatches before it keeps one again)
    assert mcmc_b._fast is not None
    assert mcmc_a._rs.random() == mcmc_b._rs.random()


def test_mc3_with_trainable_slopes_follows_the_reference(golden_dir, tmp_path):
    """MC3 with trainable activation slopes on the GPU (three chains in this process): such chains carry extra per-iteration
    draws, stay off the exchange run and the group pass and advance interval by interval through run_steps; the
    reference's swap sequence for its first accepted swaps, its first log rows, the swap log complete."""
    cfg = cases.MC3_TRACES["mc3_slopes"]
    g = np.load(os.path.join(golden_dir, "mc3_slopes.npz"))
    dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
    np.random.seed(1234)
    bnn = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], use_bias_node=cfg["bias"], seed=1, init_std=0.1, **cases.mc3_act(bn, cfg))
    logger = bn.postLogger(bnn, filename="MC3", wdir=str(tmp_path), log_all_weights=0)
    mc3 = quiet(bn.MC3, bnn, logger=logger, n_post_samples=10, sampling_f=cfg["swap_frequency"], n_iteration=cfg["n_iteration"],
                n_chains=cfg["n_chains"], swap_frequency=cfg["swap_frequency"], verbose=0)
    np.testing.assert_array_equal(mc3.rseeds, g["rseeds"])
    quiet(mc3.run_mcmc)
    assert len(mc3.swap_log) == 20
    accepted = [i for i, s in enumerate(mc3.swap_log) if s[4]]
    want = [int(r[0]) for r in g["swapped"]]
    n_common = 0
    for a, b in zip(accepted, want):
        if a != b:
            break
        n_common += 1
    assert n_common >= 5, "swap sequences diverged immediately: %s vs %s" % (accepted, want)
    rows = np.loadtxt(logger._logfile, skiprows=1)
    np.testing.assert_allclose(rows[:3, 2], g["log_rows"][:3, 2], rtol=2e-6)
    assert all(c[1]._device_iterations > 0 for c in mc3.singleChainArgs)          # (the device chain with slopes, not the mh_step loop)


# ---- random shapes: the device chain against the mh_step loop -------------------------------------------------------------
def _random_case(seed):
    rs = np.random.default_rng(seed)
    n = int(rs.integers(40, 700))
    f = int(rs.choice([1, 3, 15, 16, 17, 31, 33, 64, 70]))
    depth = int(rs.integers(1, 4))
    hidden = [int(rs.choice([1, 2, 5, 15, 16, 17, 32, 33, 48, 50, 64])) if li == 0 else int(rs.integers(1, 17)) for li in range(depth)]
    regression = bool(rs.integers(0, 2))
    fun = str(rs.choice(["ReLU", "tanh", "swish", "genReLU"]))
    bias = int(rs.choice([0, 1, 2, 3, -1]))
    x = rs.standard_normal((n, f))
    if regression:
        k = int(rs.integers(1, 4))
        dat = dict(data=x, labels=rs.standard_normal((n, k)), test_data=np.zeros((0, f)), test_labels=np.zeros((0, k)))
        extra = dict(estimation_mode="regression", empirical_error=bool(rs.integers(0, 2)))
    else:
        c = int(rs.integers(2, 13))
        lab = rs.integers(0, c, n)
        lab[:c] = np.arange(c)
        dat = dict(data=x, labels=lab, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
        extra = {}
    prior = int(rs.choice([0, 1, 1, 2, 3]))
    act_kw = dict(fun=fun)
    if fun == "genReLU":
        act_kw["prm"] = rs.uniform(0.0, 0.3, depth)
    update_f = [float(rs.choice([0.01, 0.05, 0.2, 0.6]))] * (depth + 1)
    return dat, dict(n_nodes=hidden, use_bias_node=bias, prior_f=prior, p_scale=float(rs.choice([0.5, 1.0, 2.0])), **extra), act_kw, \
        dict(update_f=update_f, update_ws=[float(rs.choice([0.02, 0.075, 0.2]))] * (depth + 1), n_iteration=100000, estimate_error=False,
             temperature=float(rs.choice([1.0, 0.8])), likelihood_tempering=float(rs.choice([1.0, 0.6])))


@pytest.mark.parametrize("seed", range(24))
def test_device_chain_is_the_mh_step_loop_on_random_shapes(seed):
    """Random table sizes, depths (1-3 hidden layers), first layers from 1 to 64 nodes (one to four output tiles: three, two and one
    candidates per pass), every activation, bias mode and prior, both estimation modes, tempered / heated chains, proposals from 1 %
    to 60 % of a layer: 90 iterations of run_steps (in calls of 30, the last one through the kept dispatch) against 90 calls of
    mh_step - same accept / reject sequence, same weights to the bit."""
    dat, model_kw, act_kw, sampler_kw = _random_case(1000 + seed)
    chains = []
    for _ in range(2):
        np.random.seed(77)
        bnn = quiet(bn.npBNN, dat, actFun=bn.ActFun(**act_kw), **model_kw)
        chains.append((bnn, bn.MCMC(bnn, **sampler_kw)))
    (bnn_a, mcmc_a), (bnn_b, mcmc_b) = chains
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
    if model_kw.get("estimation_mode") == "regression":
        np.testing.assert_allclose(np.ones(bnn_a._size_output) * bnn_a._error_prm, np.ones(bnn_b._size_output) * bnn_b._error_prm, rtol=1e-12)


def test_automatic_schedule_goes_by_the_cost_it_measured():
    """NPBNN_SCHED_AUTO between the two persistent forms on a chain that moves (config-2 shapes, a quarter of the proposals
    accepted): the library times every batch over its iterations, per form, and runs the cheaper one; a form that has not
    run yet gets a batch of its own sooner or later.  Whatever it picks, the chain is the one a fixed schedule runs."""
    from npbnn_amd import _capi as capi
    rs = np.random.default_rng(0)
    n, f, c = 100_000, 256, 10
    x = rs.standard_normal((n, f)).astype(np.float32)
    proj = rs.standard_normal((f, c)) / np.sqrt(f)
    y = np.argmax(x @ proj + 0.5 * rs.standard_normal((n, c)), axis=1)
    dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))

    def chain(sched, calls):
        np.random.seed(1234)
        bnn = quiet(bn.npBNN, dat, n_nodes=[32, 8], actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
        m = bn.MCMC(bnn, update_f=[0.004] * 3)
        m.device_schedule = sched
        used = []
        for _ in range(calls):
            m.run_steps(bnn, 100)
            used.append(m._device_schedule_used)
        return bnn, m, used

    calls = 70                                        # (past the 48 batches after which the other form is given one)
    ba, ma, used = chain(0, calls)
    ctx = ma._backend.ctx
    cost = [ctx.info(capi.INFO_IT_NS_OVERLAPPED) / 1e3, ctx.info(capi.INFO_IT_NS_BETWEEN) / 1e3]      # us per iteration
    assert set(used) <= {4, 5} and set(used) == {4, 5}, "both persistent forms get to run: %s" % sorted(set(used))
    assert all(2.0 < v < 200.0 for v in cost), cost
    late = used[-10:]
    cheaper = 4 if cost[0] < cost[1] else 5
    if abs(cost[0] - cost[1]) > 0.1 * min(cost):      # (a clear difference: the last batches are on the cheaper form, re-probes aside)
        assert late.count(cheaper) >= 8, (late, cost)
    assert 0.1 < ma._device_accepted / ma._device_iterations < 0.5
    bb, mb, _ = chain(4, calls)
    assert ma._last_accepted_mem == mb._last_accepted_mem
    assert (ma._logLik, ma._logPrior) == (mb._logLik, mb._logPrior)
    for wa, wb in zip(ba._w_layers, bb._w_layers):
        np.testing.assert_array_equal(wa, wb)


@pytest.mark.parametrize("kind", ["lognormal3", "outlier"])
def test_chain_on_heavy_tailed_features_runs_on_the_fp16_pair_and_is_the_float32_chain(kind):
    """Heavy-tailed features (log-normal columns with sigma 3; one 1e4 outlier per column over N(0, 1e-2) data) used to send `auto`
    to the float32 layer 0 at half the speed; their columns' scales are moved up instead (ensure_scales) and the
    device chain stays on the fp16 pair: same decisions and weights as the chain on the exact float32 path and as the host loop."""
    rs = np.random.default_rng(3)
    n, f, c = 30000, 40, 4
    if kind == "lognormal3":
        x = np.exp(3.0 * rs.standard_normal((n, f)))
    else:
        x = rs.normal(0, 1e-2, (n, f))
        x[rs.integers(0, n, f), np.arange(f)] = 1e4
    score = np.tanh(np.log1p(np.abs(x[:, :6])) - np.log1p(np.abs(x[:, :6])).mean(axis=0)).sum(axis=1) + rs.normal(0, 0.5, n)
    lab = np.digitize(score, np.quantile(score, [0.25, 0.5, 0.75]))
    dat = dict(data=x, labels=lab, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
    runs = []
    for mode in ("auto", "f32", "loop"):
        np.random.seed(77)
        bnn = quiet(bn.npBNN, dat, n_nodes=[12, 6], actFun=bn.ActFun(fun="tanh"), use_bias_node=1, prior_f=1, p_scale=1, init_std=0.1)
        # (weights in units of the columns: a proposal moves a typical row's pre-activation by O(0.01))
        bnn._w_layers[0][:, 1:] /= np.abs(x).mean(axis=0)
        m = bn.MCMC(bnn, update_f=[0.05, 0.1, 0.2], update_ws=[0.01, 0.05, 0.05])
        runs.append((bnn, m, mode))
    for bnn, m, mode in runs:
        if mode == "f32":
            m._backend.ctx.set_l0_precision("f32")
        if mode == "loop":
            for _ in range(150):
                m.mh_step(bnn)
        else:
            m.run_steps(bnn, 150)
        assert m._backend.ctx.l0_mode() == ("f32" if mode == "f32" else "f16-split")
    (ba, ma, _), (bb, mb, _), (bc, mc, _) = runs
    assert ma._backend.ctx.f16_moved_columns()[0] > 0
    assert ma._last_accepted_mem == mb._last_accepted_mem == mc._last_accepted_mem
    assert 5 < sum(ma._last_accepted_mem) < 150
    np.testing.assert_allclose(ma._logLik, mb._logLik, rtol=2e-6)
    for wa, wb, wc in zip(ba._w_layers, bb._w_layers, bc._w_layers):
        np.testing.assert_array_equal(wa, wb)
        np.testing.assert_array_equal(wa, wc)
