"""One rank of a multi-process test (started by npbnn_amd.launch.spawn_ranks; RANK / WORLD_SIZE / MASTER_* in the environment).

    python tests/rank_worker.py mc3 <tmpdir> <backend: oracle|hip> <comm: socket|rccl> <device_exchange: 0|1|none> [fail=<rank>:<mode>]
    python tests/rank_worker.py idle <seconds> [fail=<rank>:<mode>]
    python tests/rank_worker.py rowshard <backend: oracle|hip> <comm: socket|rccl> <case: cls|clsw|reg|regsig>

``mc3``: the reference's golden MC3 run (tests/golden/mc3.npz: 4 chains, swaps every 20 iterations) with chain i on rank
i % world.  oracle backend (CPU stand-in, float64): the golden swap sequence, final states and log rows exactly; hip backend
(float32 rows on the GPU): the golden swap sequence for at least its first five accepted swaps, every rank agreeing on the
whole swap log.  Every rank prints ``RANK <r> OK`` at the end.

``fail=<rank>:<mode>`` plants a failure on that rank after the second swap interval: ``raise`` (exception, exit status 1),
``exit0`` (the rank leaves quietly with status 0 - its peers must notice by themselves).
"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def planted(argv):
    for a in argv:
        if a.startswith("fail="):
            r, mode = a[5:].split(":")
            return int(r), mode
    return None, None


def fail_now(mode, stamp_dir):
    if stamp_dir:
        with open(os.path.join(stamp_dir, "failed_at"), "w") as f:
            f.write(repr(time.time()))
    if mode == "exit0":
        os._exit(0)
    raise RuntimeError("planted failure")


def make_comm(kind, rank, world):
    from npbnn_amd import comm as cm
    if kind == "rccl":
        c = cm.RcclComm(rank=rank, world_size=world, device=int(os.environ.get("LOCAL_RANK", "0")))
        if rank == 0:
            print("comm:", c.describe(), flush=True)
        return c
    return cm.SocketComm(rank=rank, world_size=world, timeout=float(os.environ.get("NPBNN_SOCKET_TIMEOUT_S", "60")))


def run_mc3(argv):
    import numpy as np
    import cases
    import npbnn_amd as bn
    tmpdir, backend, comm_kind, dev_x = argv[:4]
    fail_rank, fail_mode = planted(argv)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    comm = make_comm(comm_kind, rank, world)
    cfg = cases.MC3_TRACE
    g = np.load(os.path.join(ROOT, "tests", "golden", "mc3.npz"))
    dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=cfg["n_nodes"], use_bias_node=cfg["bias"], seed=1, init_std=0.1)
        logger = bn.postLogger(bnn, filename="MC3", wdir=tmpdir, log_all_weights=0, continue_logfile=rank != 0)
        if backend == "oracle":
            from oracle_backend import OracleExchangeBackend, serve_from_oracle
            serve_from_oracle(lambda b: OracleExchangeBackend(b, 0))
        mc3 = bn.MC3(bnn, logger=logger, n_post_samples=10, sampling_f=cfg["swap_frequency"], n_iteration=cfg["n_iteration"],
                     n_chains=cfg["n_chains"], swap_frequency=cfg["swap_frequency"], verbose=0, comm=comm)
    mc3.device_exchange = None if dev_x == "none" else bool(int(dev_x))
    mc3.exchange_batch = 6
    assert mc3.local_ids == [i for i in range(cfg["n_chains"]) if i % world == rank]
    if fail_rank == rank:       # leave after the second swap interval
        from npbnn_amd import exchange as ex
        real = ex.host_swap
        seen = [0]

        def counting(*a, **k):
            out = real(*a, **k)
            seen[0] += 1
            if seen[0] == 2:
                fail_now(fail_mode, tmpdir)
            return out
        ex.host_swap = counting
    with contextlib.redirect_stdout(io.StringIO()):
        mc3.run_mcmc()
    accepted = [i for i, s in enumerate(mc3.swap_log) if s[4]]
    want = [int(r[0]) for r in g["swapped"]]
    if backend == "oracle":
        assert accepted == want, (accepted, want)
        for i, pair in enumerate(mc3.singleChainArgs):
            if pair is None:
                continue
            assert pair[1]._temperature == g["final_temperature"][i]
            np.testing.assert_allclose(pair[1]._logPost, g["final_logPost"][i], rtol=1e-9)
            for li, w in enumerate(pair[0]._w_layers):
                np.testing.assert_array_equal(w, g["w_c%d_l%d" % (i, li)])
    else:
        common = 0
        for a, b in zip(accepted, want):
            if a != b:
                break
            common += 1
        assert common >= 5, "swap sequences diverged immediately: %s vs %s" % (accepted, want)
    # every rank decided the same swaps
    mine = np.array([float(s[4]) for s in mc3.swap_log])
    allv = comm.allgather_f64(mine)
    assert np.all(allv == allv[0]), "ranks disagree on the swap log"
    temps = comm.allgather_f64(np.array([p[1]._temperature if p is not None else np.nan for p in mc3.singleChainArgs]))
    assert sorted(np.nanmax(temps, axis=0)) == sorted(g["temperatures0"]), "temperatures are a permutation of the ladder"
    comm.barrier()
    if rank == 0:
        rows = np.loadtxt(logger._logfile, skiprows=1)
        if backend == "oracle":
            np.testing.assert_allclose(rows, g["log_rows"], rtol=1e-8)
        else:
            assert rows.shape[0] >= 25
    comm.close()
    print("RANK %d OK" % rank, flush=True)


def run_rowshard(argv):
    """rowshard <backend: oracle|hip> <comm: socket|rccl> <case: cls|clsw|reg|regsig>: ONE chain whose rows are split over the ranks
    (MCMC(row_comm=...)) against the same chain on all rows in this very process: the same accept decisions, the same weights, the
    same statistics."""
    import numpy as np
    import cases
    import npbnn_amd as bn
    from npbnn_amd.rowshard import shard_rows
    backend, comm_kind, case = argv[:3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    comm = make_comm(comm_kind, rank, world)
    if backend == "oracle":
        from oracle_backend import OracleChainBackend, serve_from_oracle
        serve_from_oracle(lambda b: OracleChainBackend(b, 0 if b._estimation_mode == "classification" else 1))
    if case in ("cls", "clsw"):
        dat = cases.classification_data(5, 403, 12, 4, n_test=61)
        model = dict(n_nodes=[8, 5], use_bias_node=2, seed=3, use_class_weights=int(case == "clsw"))
        sampler = dict(update_f=[0.05, 0.1, 0.2], update_ws=[0.05, 0.05, 0.05], n_iteration=2000, sampling_f=10, adapt_f=0.3, adapt_freq=25)
    else:
        dat = cases.regression_data(7, 389, 10, 2, n_test=50)
        model = dict(n_nodes=[6, 4], use_bias_node=2, seed=3, estimation_mode="regression", empirical_error=(case == "reg"),
                     actFun=bn.ActFun(fun="tanh"))
        sampler = dict(update_f=[0.05, 0.1, 0.2], update_ws=[0.03, 0.03, 0.03], n_iteration=200, sampling_f=10, estimate_error=(case == "regsig"))

    def run(data, **extra):
        np.random.seed(1234)
        with contextlib.redirect_stdout(io.StringIO()):
            bnn = bn.npBNN(data, **model)
            mcmc = bn.MCMC(bnn, **sampler, **extra)
            trail = [float(mcmc._logLik)]
            mcmc.run_steps(bnn, 30)
            trail.append(float(mcmc._logLik))
            for _ in range(4):
                mcmc.mh_step(bnn)
            trail.append(float(mcmc._logLik))
            mcmc.run_steps(bnn, 45)
            trail.append(float(mcmc._logLik))
            stats = [float(mcmc._accuracy), float(mcmc._test_accuracy)]
        return bnn, mcmc, trail, stats

    bnn0, mcmc0, trail0, stats0 = run(dat)
    bnn1, mcmc1, trail1, stats1 = run(shard_rows(dat, rank, world), row_comm=comm)
    assert mcmc1._backend.row_sharded and mcmc1._backend.n_rows_total == len(dat["data"])
    assert len(bnn1._data) < len(dat["data"])
    assert mcmc0._current_iteration == mcmc1._current_iteration == 79
    tol = 1e-10 if backend == "oracle" else 2e-6
    np.testing.assert_allclose(trail1, trail0, rtol=tol)
    assert mcmc1._last_accepted_mem == mcmc0._last_accepted_mem, "the sharded chain took other decisions"
    assert sum(mcmc0._last_accepted_mem) >= 5, "a chain that never moves proves nothing"
    for a, b in zip(bnn1._w_layers, bnn0._w_layers):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_allclose(stats1, stats0, rtol=1e-9 if backend == "oracle" else 1e-5)
    if case == "clsw":
        np.testing.assert_allclose(bnn1._class_w, bnn0._class_w, rtol=1e-14)
    if np.size(bnn0._error_prm):
        np.testing.assert_allclose(bnn1._error_prm, bnn0._error_prm, rtol=tol * 100)
    # every rank holds the same chain
    mine = np.concatenate([w.ravel() for w in bnn1._w_layers])
    allw = comm.allgather_f64(mine)
    assert np.all(allw == allw[0]), "ranks disagree on the chain"
    if backend == "hip":
        used = getattr(mcmc1, "_device_schedule_used", None)
        assert used == 1, "a row-sharded batch runs on kernel boundaries (schedule %r)" % used
    comm.barrier()
    comm.close()
    print("RANK %d OK" % rank, flush=True)


def run_idle(argv):
    fail_rank, fail_mode = planted(argv)
    rank = int(os.environ["RANK"])
    if fail_rank == rank:
        time.sleep(0.3)
        fail_now(fail_mode, None)
    time.sleep(float(argv[0]))
    print("RANK %d OK" % rank, flush=True)


if __name__ == "__main__":
    {"mc3": run_mc3, "idle": run_idle, "rowshard": run_rowshard}[sys.argv[1]](sys.argv[2:])
