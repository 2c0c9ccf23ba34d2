"""GPU tests of the exchange run (npbnn_chains_run_exchange): several chains advance swap intervals with the temperature
swaps of MC3.run_mcmc (np_bnn/BNN_mc3.py:94-112) decided on the GPU.  The oracle is the interval-by-interval path - the same
device chain per interval with the swap on the host - which the other suites tie to the reference: the two must agree bit for
bit in weights, log-posteriors, temperatures and acceptance book-keeping."""
import contextlib
import io

import numpy as np
import pytest

import cases
import npbnn_amd as bn
from npbnn_amd import exchange as ex

pytestmark = pytest.mark.gpu


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


def build_chains(cfg, temps, empirical=None, **mcmc_extra):
    """len(temps) chains on one data set as MC3 builds them: own weights seed, mcmc_id = chain id, reseeding every iteration."""
    if cfg["kind"] == "classification":
        dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
        extra = {}
    else:
        dat = cases.regression_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["k"], cfg["n_test"])
        extra = dict(estimation_mode="regression", empirical_error=cfg.get("empirical_error", False) if empirical is None else empirical)
    chains = []
    for i, t in enumerate(temps):
        np.random.seed(1234 + i)
        bnn = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], actFun=bn.ActFun(fun=cfg["fun"]), use_bias_node=cfg["bias"],
                    prior_f=1, p_scale=1, seed=1234 + i, init_std=0.1, **extra)
        kw = dict(cfg["mcmc"])
        kw.update(adapt_f=0, adapt_fM=1, temperature=t, mcmc_id=i, randomize_seed=True)
        kw.update(mcmc_extra)
        chains.append((bnn, bn.MCMC(bnn, **kw)))
    return chains


def state_of(chains):
    return [dict(w=[w.copy() for w in bnn._w_layers], ll=m._logLik, lp=m._logPrior, post=m._logPost, temp=m._temperature,
                 it=m._current_iteration, rate=m._acceptance_rate, mem=list(m._last_accepted_mem), err=np.array(bnn._error_prm))
            for bnn, m in chains]


def assert_same(a, b, exact=True):
    for x, y in zip(a, b):
        assert x["it"] == y["it"]
        assert x["temp"] == y["temp"]
        for u, v in zip(x["w"], y["w"]):
            np.testing.assert_array_equal(u, v)
        if exact:
            assert (x["ll"], x["lp"], x["post"]) == (y["ll"], y["lp"], y["post"])
        else:
            np.testing.assert_allclose([x["ll"], x["lp"], x["post"]], [y["ll"], y["lp"], y["post"]], rtol=1e-12)
        assert x["mem"] == y["mem"] and x["rate"] == y["rate"]
        np.testing.assert_array_equal(x["err"], y["err"])


def slow_path(chains, n_seg, seg_len, swap_seed):
    swaps = ex.SwapProposals(len(chains), np.random.RandomState(swap_seed))
    ids = list(range(len(chains)))
    log = []
    ex.advance_intervals(chains, ids, len(chains), n_seg, seg_len, swaps, 0, device=False,
                         on_interval=lambda s, info: log.append((info["swap"], info["scalars"].copy())))
    return log


@pytest.mark.parametrize("name,sched,ncand", [("cfg1", 2, 0), ("cfg1", 1, 0), ("cfg1", 2, 1), ("cfg2s", 2, 0), ("cfg2s", 1, 2), ("cfg4s", 2, 2)])
def test_exchange_run_equals_interval_by_interval(name, sched, ncand):
    """(With the schedule left to the library, the two paths may pick different ones for a batch - the choice follows the
    acceptance rate of the previous batch - and the log-likelihood sums then differ in their last bits: fixed here.)"""
    cfg = cases.TRACES[name]
    temps = [0.8, 0.9, 1.0]
    n_seg, seg_len = 6, 25
    a = build_chains(cfg, temps)
    b = build_chains(cfg, temps)
    for _, m in a + b:
        m.device_schedule, m.n_candidates = sched, ncand
    log_a = slow_path(a, n_seg, seg_len, 7)
    swaps = ex.SwapProposals(len(b), np.random.RandomState(7))
    log_b = []
    done = ex.advance_intervals(b, [0, 1, 2], 3, n_seg, seg_len, swaps, 0, batch=n_seg,
                                on_interval=lambda s, info: log_b.append((info["swap"], info["scalars"].copy(), info["cold"])))
    assert done == n_seg
    assert_same(state_of(a), state_of(b))
    assert len(log_a) == len(log_b) == n_seg
    assert any(sw[4] for sw, _ in log_a), "the case should accept at least one swap"
    for (sw_a, sc_a), (sw_b, sc_b, cold) in zip(log_a, log_b):
        assert sw_a == sw_b
        np.testing.assert_array_equal(sc_a, sc_b)
        assert cold is not None, "all intervals should have run on the device"
        assert sum(c is not None for c in cold) == 1          # exactly one chain is cold after every swap


def test_cold_chain_snapshots_are_the_states_at_the_swaps():
    cfg = cases.TRACES["cfg1"]
    temps = [0.8, 1.0]
    n_seg, seg_len = 5, 20
    a = build_chains(cfg, temps)
    b = build_chains(cfg, temps)
    snaps = []

    def keep(s, info):
        for (bnn, m), t in zip(a, info["scalars"][:, 1]):
            if t == 1.0:
                snaps.append((np.concatenate([w.ravel() for w in bnn._w_layers]), m._logLik, m._logPrior))
    swaps = ex.SwapProposals(2, np.random.RandomState(3))
    ex.advance_intervals(a, [0, 1], 2, n_seg, seg_len, swaps, 0, device=False, on_interval=keep)
    got = []
    swaps = ex.SwapProposals(2, np.random.RandomState(3))
    ex.advance_intervals(b, [0, 1], 2, n_seg, seg_len, swaps, 0, batch=n_seg,
                         on_interval=lambda s, info: got.extend(c for c in info["cold"] if c is not None))
    assert len(got) == len(snaps) == n_seg
    for (w, ll, lp), c in zip(snaps, got):
        np.testing.assert_array_equal(w, c["w"])
        assert (ll, lp) == (c["loglik"], c["logprior"])


def test_starved_interval_is_finished_on_the_slow_path():
    """Too few launches for an interval: every chain stops at that exchange, the driver completes the interval with the
    per-interval path and the chains end where they would have anyway.  (The interval that is finished in two pieces re-sums
    the log prior where the second piece starts, so log-posteriors may differ from the one-piece path in their last bits; the
    weights - a function of the accept / reject sequence alone - must not.)"""
    cfg = cases.TRACES["cfg1"]
    temps = [0.85, 1.0]
    n_seg, seg_len = 4, 30
    a = build_chains(cfg, temps)
    b = build_chains(cfg, temps)
    slow_path(a, n_seg, seg_len, 11)
    swaps = ex.SwapProposals(2, np.random.RandomState(11))
    done, records, outs = ex.run_exchange(b, [0, 1], 2, n_seg, seg_len, swaps, 0, launch_slack=0.05)
    assert done < n_seg
    for (bnn, m), out in zip(b, outs):
        assert done * seg_len <= m._current_iteration <= (done + 1) * seg_len
    # complete: the unfinished interval on the slow path, the rest on the device
    for bnn, m in b:
        rest = (done + 1) * seg_len - m._current_iteration
        if rest > 0:
            m.run_steps(bnn, rest)
    ex.host_swap(b, [0, 1], 2, swaps, done)
    left = n_seg - done - 1
    if left > 0:
        assert ex.advance_intervals(b, [0, 1], 2, left, seg_len, swaps, done + 1, batch=left) == left
    assert_same(state_of(a), state_of(b), exact=False)


def test_driver_recovers_from_a_starved_batch():
    cfg = cases.TRACES["cfg1"]
    temps = [0.85, 1.0]
    n_seg, seg_len = 6, 30
    a = build_chains(cfg, temps)
    b = build_chains(cfg, temps)
    log_a = slow_path(a, n_seg, seg_len, 5)
    cls = type(b[0][1]._backend) if b[0][1]._backend is not None else None
    from npbnn_amd.hip_backend import HipBackend
    HipBackend.exchange_slack = 0.05
    try:
        swaps = ex.SwapProposals(2, np.random.RandomState(5))
        log_b = []
        done = ex.advance_intervals(b, [0, 1], 2, n_seg, seg_len, swaps, 0, batch=3,
                                    on_interval=lambda s, info: log_b.append(info["swap"]))
    finally:
        HipBackend.exchange_slack = 1.5
    assert done == n_seg and cls is not None
    for (sw_a, _), sw_b in zip(log_a, log_b):
        assert (sw_a[0], sw_a[1], sw_a[3], sw_a[4]) == (sw_b[0], sw_b[1], sw_b[3], sw_b[4])
        np.testing.assert_allclose(sw_a[2], sw_b[2], rtol=1e-9, atol=1e-9)
    assert len(log_a) == len(log_b)
    assert_same(state_of(a), state_of(b), exact=False)


def test_single_job_and_argument_checks():
    cfg = cases.TRACES["cfg1"]
    (bnn, m), = build_chains(cfg, [1.0])
    m.run_steps(bnn, 10)
    swaps = ex.SwapProposals(2, np.random.RandomState(1))
    with pytest.raises(Exception):
        ex.run_exchange([(bnn, m)], [0], 2, 2, 10, swaps, 0)        # 2 chains announced, one job on one rank


def test_patched_weight_image_equals_a_fresh_pack():
    """The device weight image after a chain run (packed once, then patched entry by entry at every accept) is bit for bit the
    image a fresh pack of the final weights gives.  (Regression: the fp16 high/low split of a layer-0 entry once depended on
    the kernel that made it when the scaled weight fell exactly between two fp16 values.)"""
    import ctypes as C
    cfg = cases.TRACES["cfg2s"]
    chains = build_chains(cfg, [0.8, 0.9, 1.0])
    for _, m in chains:
        m.device_schedule = 2
    swaps = ex.SwapProposals(3, np.random.RandomState(7))
    done, _, _ = ex.run_exchange(chains, [0, 1, 2], 3, 2, 25, swaps, 0)
    assert done == 2

    def image(ctx):
        buf = np.zeros(1 << 16, dtype=np.float32)
        f = ctx._lib.npbnn_debug_image_
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_int]
        n = f(ctx._ctx, buf.ctypes.data, buf.size)
        assert n > 0
        return buf[:n].view(np.uint32).copy()
    for bnn, m in chains:
        ctx = m._backend.ctx
        patched = image(ctx)
        m._backend.evaluate(bnn._w_layers, None)          # packs the image from the float64 weights
        np.testing.assert_array_equal(patched, image(ctx))


def test_exchange_through_a_native_rccl_communicator():
    """The records travel through ncclAllGather (in place, on the first chain's stream) when a communicator is given: one
    rank here (this box has one GPU), three chains on it - same result as without the communicator."""
    from npbnn_amd.comm import RcclComm
    try:
        comm = RcclComm(rank=0, world_size=1, device=0)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip("no RCCL communicator on this box: %s" % e)
    cfg = cases.TRACES["cfg1"]
    temps = [0.8, 0.9, 1.0]
    n_seg, seg_len = 5, 20
    a = build_chains(cfg, temps)
    b = build_chains(cfg, temps)
    for _, m in a + b:
        m.device_schedule = 2
    try:
        swaps = ex.SwapProposals(3, np.random.RandomState(21))
        assert ex.advance_intervals(a, [0, 1, 2], 3, n_seg, seg_len, swaps, 0, batch=n_seg) == n_seg
        swaps = ex.SwapProposals(3, np.random.RandomState(21))
        done, records, _ = ex.run_exchange(b, [0, 1, 2], 3, n_seg, seg_len, swaps, 0, comm=comm)
        assert done == n_seg
        assert np.all(records[:, :, 2] == 1.0)
        assert_same(state_of(a), state_of(b))
    finally:
        comm.close()


def test_mc3_logs_are_the_same_with_and_without_device_exchange(tmp_path, monkeypatch):
    """MC3.run_mcmc (bnn_runner_MC3.py call sequence, 4 chains in this process): swap intervals in device batches against
    one device batch per interval with the swap on the host - same swap log, same rows in the cold chain's log file, same
    posterior weight samples, same pickle."""
    monkeypatch.setattr(bn.MCMC, "device_schedule", 2)       # (left to the library the two paths may pick different schedules)
    cfg = cases.MC3_TRACE
    dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
    runs = []
    for name, device in (("host", False), ("device", True)):
        np.random.seed(1234)
        bnn = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], use_bias_node=cfg["bias"], seed=1, init_std=0.1)
        logger = bn.postLogger(bnn, filename="MC3" + name, wdir=str(tmp_path), log_all_weights=0)
        mc3 = quiet(bn.MC3, bnn, logger=logger, n_post_samples=10, sampling_f=cfg["swap_frequency"], n_iteration=800,
                    n_chains=cfg["n_chains"], swap_frequency=cfg["swap_frequency"], verbose=0, adapt_stop=100)
        mc3.device_exchange = device
        mc3.exchange_batch = 7
        seen = []
        if device:
            from npbnn_amd import exchange as ex
            real = ex.run_exchange
            monkeypatch.setattr(ex, "run_exchange", lambda *a, **k: (seen.append(a[3]), real(*a, **k))[1])
        quiet(mc3.run_mcmc)
        if device:
            assert sum(seen) >= 25, "most of the 40 intervals should have run in device batches (%s)" % seen
        runs.append((mc3, logger))
    (ma, la), (mb, lb) = runs
    assert ma.swap_log == mb.swap_log and any(s[4] for s in ma.swap_log)
    assert [c[1]._temperature for c in ma.singleChainArgs] == [c[1]._temperature for c in mb.singleChainArgs]
    rows_a, rows_b = np.loadtxt(la._logfile, skiprows=1), np.loadtxt(lb._logfile, skiprows=1)
    assert rows_a.shape == rows_b.shape and rows_a.shape[0] == 40
    np.testing.assert_array_equal(rows_a, rows_b)
    assert len(la._post_weight_samples) == len(lb._post_weight_samples) == 10
    for sa, sb in zip(la._post_weight_samples, lb._post_weight_samples):
        assert sa["mcmc_it"] == sb["mcmc_it"]
        for u, v in zip(sa["weights"], sb["weights"]):
            np.testing.assert_array_equal(u, v)
    (ba, mca, _), (bb, mcb, _) = bn.load_obj(la._pklfile), bn.load_obj(lb._pklfile)
    assert mca._current_iteration == mcb._current_iteration and mca._logPost == mcb._logPost
    for u, v in zip(ba._w_layers, bb._w_layers):
        np.testing.assert_array_equal(u, v)


def test_chains_of_one_model_share_the_resident_matrices():
    """MC3 replicates the model per chain with the data arrays shared (BNN_mc3.py:55-58): their device contexts share one
    resident copy of the feature matrices; results are those of private copies, and the memory outlives the owner."""
    import gc
    cfg = cases.MC3_TRACE
    dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
    np.random.seed(1234)
    bnn = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], use_bias_node=cfg["bias"], seed=1, init_std=0.1)
    mc3 = quiet(bn.MC3, bnn, logger=None, n_chains=3, swap_frequency=20, n_iteration=200, verbose=0, adapt_stop=40)
    backends = [c[1]._backend for c in mc3.singleChainArgs]
    assert backends[0].data_shared_with is None
    assert backends[1].data_shared_with is backends[0] and backends[2].data_shared_with is backends[0]
    quiet(mc3.run_mcmc)
    state = [(c[1]._logLik, c[1]._temperature, c[0]._w_layers[0].copy()) for c in mc3.singleChainArgs]
    # the same run with private copies
    os_environ = __import__("os").environ
    os_environ["NPBNN_NO_DATA_SHARING"] = "1"
    try:
        np.random.seed(1234)
        bnn2 = quiet(bn.npBNN, dat, n_nodes=cfg["n_nodes"], use_bias_node=cfg["bias"], seed=1, init_std=0.1)
        mc3b = quiet(bn.MC3, bnn2, logger=None, n_chains=3, swap_frequency=20, n_iteration=200, verbose=0, adapt_stop=40)
        assert all(c[1]._backend.data_shared_with is None for c in mc3b.singleChainArgs)
        quiet(mc3b.run_mcmc)
    finally:
        del os_environ["NPBNN_NO_DATA_SHARING"]
    for (ll, t, w), c in zip(state, mc3b.singleChainArgs):
        assert (ll, t) == (c[1]._logLik, c[1]._temperature)
        np.testing.assert_array_equal(w, c[0]._w_layers[0])
    # the owner goes first: the borrowers keep working
    first = mc3.singleChainArgs[0]
    first[1]._backend.close()
    m = mc3.singleChainArgs[1][1]
    before = m._logLik
    r = m._backend.evaluate(mc3.singleChainArgs[1][0]._w_layers, None)
    np.testing.assert_allclose(r["loglik"], before, rtol=1e-12)
    # an owner cannot be given new data while others read its matrices; a borrower can (it lets go first)
    own, other = mc3b.singleChainArgs[0][1]._backend, mc3b.singleChainArgs[1][1]._backend
    other.ctx.share_data(own.ctx)
    with pytest.raises(Exception, match="use this one's matrices"):
        own.ctx.set_data(dat["data"])
    other.ctx.set_data(dat["data"])
    own.ctx.set_data(dat["data"])
    gc.collect()


def test_exchange_run_at_config2_size():
    """BASELINE config 2 shapes (100k x 256, [32, 8], tanh, 10 classes), two chains on one data set, swaps every 50 iterations:
    the exchange run against the per-interval path, bit for bit."""
    rs = np.random.default_rng(0)
    x = rs.standard_normal((100_000, 256)).astype(np.float32)
    y = rs.integers(0, 10, 100_000)
    dat = dict(data=x, labels=y, test_data=np.zeros((0, 256)), test_labels=np.zeros(0))

    def make():
        chains = []
        for i, t in enumerate((0.9, 1.0)):
            np.random.seed(1234 + i)
            bnn = quiet(bn.npBNN, dat, n_nodes=[32, 8], actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
            m = bn.MCMC(bnn, temperature=t, mcmc_id=i, randomize_seed=True, update_f=[0.01, 0.01, 0.01])
            m.device_schedule = 2
            chains.append((bnn, m))
        return chains
    a, b = make(), make()
    assert b[1][1]._backend.data_shared_with is not None
    n_seg, seg_len = 8, 50
    log_a = slow_path(a, n_seg, seg_len, 13)
    swaps = ex.SwapProposals(2, np.random.RandomState(13))
    log_b = []
    assert ex.advance_intervals(b, [0, 1], 2, n_seg, seg_len, swaps, 0, batch=n_seg,
                                on_interval=lambda s, info: log_b.append((info["swap"], info["cold"]))) == n_seg
    assert all(c is not None for _, c in log_b), "all intervals should have run on the device"
    assert [s for s, _ in log_a] == [s for s, _ in log_b]
    assert sum(m._last_accepted_mem.count(1) for _, m in a) > 5
    assert_same(state_of(a), state_of(b))


def test_exchange_run_with_an_estimated_error_parameter():
    """Chains whose proposals also move the regression error parameter (sigma multipliers and Hastings terms pre-drawn per
    iteration): exchange run against the per-interval path, error parameters included."""
    cfg = cases.TRACES["cfg4s"]
    temps = [0.85, 1.0]
    n_seg, seg_len = 5, 30
    kw = dict(empirical=False, estimate_error=True, n_iteration=100)        # proposals from iteration 11 on
    a = build_chains(cfg, temps, **kw)
    b = build_chains(cfg, temps, **kw)
    for bnn, m in a + b:
        m.device_schedule = 2
        m.run_steps(bnn, 20)
    slow_path(a, n_seg, seg_len, 17)
    swaps = ex.SwapProposals(2, np.random.RandomState(17))
    seen = []
    real = ex.run_exchange
    ex.run_exchange = lambda *x, **k: (lambda o: (seen.append(o[0]), o)[1])(real(*x, **k))
    try:
        assert ex.advance_intervals(b, [0, 1], 2, n_seg, seg_len, swaps, 0, batch=n_seg) == n_seg
    finally:
        ex.run_exchange = real
    assert sum(seen) >= n_seg - 2, "most intervals should have run in device batches (%s)" % seen
    assert_same(state_of(a), state_of(b), exact=seen == [n_seg])       # (an interval finished in two pieces re-sums the prior)
    assert not np.all(state_of(b)[0]["err"] == 1.0)


class _NoSwaps:
    """Swap proposals between a chain and itself: the exchange machinery with one chain (one rank of a larger run looks the
    same to the device code, apart from the all-gather)."""

    def get(self, first, n=1):
        return np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), np.zeros(n)

    def release(self, upto):
        pass


@pytest.mark.parametrize("with_comm", [False, True])
def test_single_chain_exchange_run_on_two_streams(with_comm):
    """One chain per GPU (the multi-GPU layout): its exchange run uses the two-stream schedule - exchange kernels on the stream of
    the interval's last launch, the next interval's first launch gated on the other - with or without an RCCL communicator in
    the loop.  Against plain run_steps, bit for bit."""
    cfg = cases.TRACES["cfg2s"]
    comm = None
    if with_comm:
        from npbnn_amd.comm import RcclComm
        try:
            comm = RcclComm(rank=0, world_size=1, device=0)
        except Exception as e:                                  # noqa: BLE001
            pytest.skip("no RCCL communicator on this box: %s" % e)
    (bnn_a, ma), = build_chains(cfg, [1.0])
    n_seg, seg_len = 8, 40
    ma.device_schedule = 2
    ma.run_steps(bnn_a, 64)
    for _ in range(n_seg):
        ma.run_steps(bnn_a, seg_len)
    # (opt-in schedule: a device-side wait may time out, which leaves the run short by design - the caller finishes the interval
    # the slow way.  One time-out must not fail the test, a build that never gets through must: three fresh chains at most.)
    clean = False
    try:
        for _ in range(3):
            (bnn_b, mb), = build_chains(cfg, [1.0])
            mb.device_schedule = 3
            mb.run_steps(bnn_b, 64)
            done, records, outs = ex.run_exchange([(bnn_b, mb)], [0], 1, n_seg, seg_len, _NoSwaps(), 0, comm=comm)
            if done == n_seg:
                clean = True
                break
    finally:
        if comm is not None:
            comm.close()
    assert clean, "the two-stream exchange run came out short three times out of three"
    assert outs[0]["result"]["schedule"] == 3
    assert np.all(records[:, 0, 2] == 1.0) and records[-1, 0, 3] == n_seg * seg_len
    assert_same(state_of([(bnn_a, ma)]), state_of([(bnn_b, mb)]))


def test_bench_exchange_self_check_runs_and_restores_the_chains():
    """bench.py checks the device exchange path against the interval-by-interval path before it times anything with several
    ranks; the same routine here on one GPU with two chains: it agrees, and the chains are back where they started."""
    import sys
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    from bench_support import exchange_self_check
    cfg = cases.TRACES["cfg1"]
    chains = build_chains(cfg, [0.9, 1.0])
    for bnn, m in chains:
        m.device_schedule = 2
        m.run_steps(bnn, 40)
    before = state_of(chains)
    ok, bad = exchange_self_check(chains, [0, 1], 2, None, 20, lambda: ex.SwapProposals(2, np.random.RandomState(99)))
    assert ok and bad == []
    assert_same(before, state_of(chains))
    # and the chains still run
    assert ex.advance_intervals(chains, [0, 1], 2, 3, 20, ex.SwapProposals(2, np.random.RandomState(5)), 0, batch=3) == 3


@pytest.mark.parametrize("name,n_chains", [("cfg2s", 3), ("cfg2s", 2), ("cfg4s", 3), ("cfg1", 4)])
def test_group_pass_gives_every_chain_its_own_chain(name, n_chains):
    """npbnn_chains_run_batched: the chains of one model share every streaming read of the data, one proposal
    per chain per launch.  Each chain must be the chain its own run_steps gives it: same accept / reject sequence, same weights,
    same sigma; log-likelihood to rounding (its sums come from a different number of workgroup partials).  Four chains: a group
    of three and one on its own."""
    cfg = cases.TRACES[name]
    temps = list(np.linspace(0.8, 1.0, n_chains))
    apart = build_chains(cfg, temps)
    together = build_chains(cfg, temps)
    for _ in range(3):
        for bnn, m in apart:
            m.run_steps(bnn, 70)
        ex.run_steps_batched(together, 70)
    assert_same(state_of(apart), state_of(together), exact=False)
    passes = [m._device_passes for _, m in together]
    # in a group every launch carries one proposal of every chain: a chain's passes ~ its iterations (+ one per accept)
    assert all(m._device_iterations == 210 for _, m in together)
    grouped = together[:3] if n_chains >= 3 else together
    assert all(210 <= p <= 210 + 2 * m._device_accepted + 6 for p, (_, m) in zip(passes, grouped)), passes


def test_mc3_intervals_with_group_passes_match_the_interval_path():
    """MC3's loop (BNN_mc3.py:94-112) on one GPU with the swap on the host: the local chains advance through group passes; swaps,
    temperatures and chains as when every chain runs its interval alone."""
    cfg = cases.TRACES["cfg2s"]
    temps = list(np.linspace(0.8, 1.0, 3))
    a, b = build_chains(cfg, temps), build_chains(cfg, temps)
    swaps_a, swaps_b = (ex.SwapProposals(3, np.random.RandomState(5)) for _ in range(2))
    log_a, log_b = [], []
    for i in range(6):                      # one chain at a time
        for bnn, m in a:
            m.run_steps(bnn, 40)
        log_a.append(ex.host_swap(a, [0, 1, 2], 3, swaps_a, i)[1])
    ex.advance_intervals(b, [0, 1, 2], 3, 6, 40, swaps_b, 0, device=False, on_interval=lambda s, info: log_b.append(info["swap"]),
                         group_passes=True)
    assert all(m._device_passes >= 240 for _, m in b)          # (a chain alone needs ~240 / 2.5 passes: these went through groups)
    assert [(s[0], s[1], s[4]) for s in log_a] == [(s[0], s[1], s[4]) for s in log_b]
    assert_same(state_of(a), state_of(b), exact=False)


def test_group_passes_are_taken_automatically_only_for_chains_that_move(monkeypatch):
    """advance_intervals(group_passes="auto"): chains that accept more than exchange.GROUP_PASS_ACCEPTANCE of their proposals share
    their passes over the data, interval by interval; chains that barely move run in device batches.  The same chains either way."""
    cfg = cases.TRACES["cfg2s"]
    temps = list(np.linspace(0.8, 1.0, 3))
    for threshold, expect_groups in ((0.0, True), (1.0, False)):
        monkeypatch.setattr(ex, "GROUP_PASS_ACCEPTANCE", threshold)
        a, b = build_chains(cfg, temps), build_chains(cfg, temps)
        for chains in (a, b):
            for bnn, m in chains:
                m.device_schedule = 2
                m.run_steps(bnn, 60)        # (some acceptance history)
        calls = []
        real = ex.run_steps_batched
        monkeypatch.setattr(ex, "run_steps_batched", lambda *x, **k: (calls.append(1), real(*x, **k))[1])
        swaps_a, swaps_b = (ex.SwapProposals(3, np.random.RandomState(5)) for _ in range(2))
        ex.advance_intervals(a, [0, 1, 2], 3, 4, 40, swaps_a, 0, device=False)
        ex.advance_intervals(b, [0, 1, 2], 3, 4, 40, swaps_b, 0, group_passes="auto")
        assert (len(calls) == 4) == expect_groups, (threshold, calls)
        assert_same(state_of(a), state_of(b), exact=False)
        monkeypatch.setattr(ex, "run_steps_batched", real)


def test_config5_shape_four_chains_in_one_process(tmp_path, monkeypatch):
    """BASELINE.json config 5's MC3 half in the one-GPU form: FOUR chains of the block-masked regression network (create_mask /
    apply_mask: the block-structured layer-0 builds, ReLU, bias on the last layer, bench_support.Config5's model on a tenth of its
    rows), swap intervals in device batches against one batch per interval with the swap on the host - same swap log, same cold-chain
    log rows, same final weights (masked entries still zero), all chains sharing one resident copy of X."""
    from bench_support import Config5
    monkeypatch.setattr(bn.MCMC, "device_schedule", 2)
    wl = Config5()
    n = 5000
    dat = dict(data=wl.x[:n].astype(np.float32), labels=wl.y[:n], test_data=np.zeros((0, wl.f)), test_labels=np.zeros((0, wl.k)))
    idx, per = wl._mask_args()
    runs = []
    for name, device in (("host", False), ("device", True)):
        np.random.seed(1234)
        bnn = quiet(bn.npBNN, dat, n_nodes=wl.hidden, estimation_mode="regression", p_scale=1, use_bias_node=-1)
        quiet(bnn.apply_mask, bn.create_mask(bnn._w_layers, indx_input_list=idx, nodes_per_feature_list=per))
        logger = bn.postLogger(bnn, filename="C5" + name, wdir=str(tmp_path), log_all_weights=0)
        mc3 = quiet(bn.MC3, bnn, logger=logger, n_post_samples=5, sampling_f=25, n_iteration=500, n_chains=4, swap_frequency=25,
                    verbose=0, adapt_stop=50)
        mc3.device_exchange = device
        mc3.exchange_batch = 6
        quiet(mc3.run_mcmc)
        runs.append((mc3, logger))
    (ma, la), (mb, lb) = runs
    assert ma.swap_log == mb.swap_log and len(ma.swap_log) == 20 and any(s[4] for s in ma.swap_log)
    assert [c[1]._temperature for c in ma.singleChainArgs] == [c[1]._temperature for c in mb.singleChainArgs]
    rows_a, rows_b = np.loadtxt(la._logfile, skiprows=1), np.loadtxt(lb._logfile, skiprows=1)
    head = open(la._logfile).readline().split()
    bad = sorted({head[c] for c in np.nonzero(np.any(rows_a != rows_b, axis=0))[0]}) if rows_a.shape == rows_b.shape else ["shape"]
    assert not bad, "columns that differ between the two paths: %s\n%s\n%s" % (bad, rows_a[:3], rows_b[:3])
    owners = set()
    for (ba, ca), (bb, cb) in zip(ma.singleChainArgs, mb.singleChainArgs):
        assert ca._device_iterations > 0 and cb._device_iterations > 0
        for u, v, keep in zip(ba._w_layers, bb._w_layers, ba._mask):
            np.testing.assert_array_equal(u, v)
            assert np.all(u[keep == 0] == 0)
        be = cb._backend
        owners.add(id(be.data_shared_with if be.data_shared_with is not None else be))
        assert be.ctx.info(bn._capi.INFO_FAST_TAILS) == 1
    assert len(owners) == 1
