"""The exchange driver on CPU (numpy stand-in for the device entry point, tests/oracle_backend.py): swap intervals in
batches against the interval-by-interval path, shortfalls, MC3's logging from the saved cold-chain states, and a 2-rank
gloo run."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest

import cases
import npbnn_amd as bn
from npbnn_amd import exchange as ex
from oracle_backend import OracleExchangeBackend, serve_from_oracle


def build_mc3(tmpdir, name, comm=None, device=True, n_iteration=400, batch=6):
    cfg = cases.MC3_TRACE
    dat = cases.classification_data(cfg["seed"], 300, 12, cfg["n_classes"], 40)
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=[4, 3], use_bias_node=cfg["bias"], seed=1, init_std=0.1)
        rank0 = comm is None or comm.rank == 0
        logger = bn.postLogger(bnn, filename=name, wdir=str(tmpdir), log_all_weights=0, continue_logfile=not rank0)
        serve_from_oracle(lambda b: OracleExchangeBackend(b, 0))
        mc3 = bn.MC3(bnn, logger=logger, n_post_samples=10, sampling_f=20, n_iteration=n_iteration, n_chains=4, swap_frequency=20,
                     verbose=0, comm=comm, adapt_stop=60)
    mc3.device_exchange = device
    mc3.exchange_batch = batch
    return mc3, logger


def run_quiet(mc3):
    with contextlib.redirect_stdout(io.StringIO()):
        mc3.run_mcmc()


def compare(ma, la, mb, lb, rows=True):
    assert ma.swap_log == mb.swap_log
    for pa, pb in zip(ma.singleChainArgs, mb.singleChainArgs):
        if pa is None:
            assert pb is None
            continue
        assert pa[1]._temperature == pb[1]._temperature and pa[1]._current_iteration == pb[1]._current_iteration
        assert (pa[1]._logLik, pa[1]._logPrior, pa[1]._logPost) == (pb[1]._logLik, pb[1]._logPrior, pb[1]._logPost)
        assert pa[1]._last_accepted_mem == pb[1]._last_accepted_mem and pa[1]._acceptance_rate == pb[1]._acceptance_rate
        for u, v in zip(pa[0]._w_layers, pb[0]._w_layers):
            np.testing.assert_array_equal(u, v)
    if rows:
        ra, rb = np.loadtxt(la._logfile, skiprows=1), np.loadtxt(lb._logfile, skiprows=1)
        np.testing.assert_array_equal(ra, rb)
        assert [s["mcmc_it"] for s in la._post_weight_samples] == [s["mcmc_it"] for s in lb._post_weight_samples]
        for sa, sb in zip(la._post_weight_samples, lb._post_weight_samples):
            for u, v in zip(sa["weights"], sb["weights"]):
                np.testing.assert_array_equal(u, v)


def test_swap_proposals_follow_the_reference_order():
    np.random.seed(5)
    want = []
    for _ in range(7):
        j, k = np.random.choice(range(4), 2, replace=False)
        want.append((int(j), int(k), float(np.log(np.random.random()))))
    np.random.seed(5)
    sp = ex.SwapProposals(4)
    j, k, u = sp.get(0, 3)
    got = list(zip(j.tolist(), k.tolist(), u.tolist()))
    sp.release(2)
    j, k, u = sp.get(2, 5)
    got = got[:2] + list(zip(j.tolist(), k.tolist(), u.tolist()))
    assert got == want
    with pytest.raises(ValueError):
        sp.get(1, 1)


def test_device_batches_equal_interval_by_interval(tmp_path):
    ma, la = build_mc3(tmp_path, "host", device=False)      # (the swap proposals come from the global np.random stream,
    run_quiet(ma)                                            #  which build_mc3 seeds: build and run back to back)
    mb, lb = build_mc3(tmp_path, "dev", device=True)
    run_quiet(mb)
    assert any(s[4] for s in ma.swap_log)
    compare(ma, la, mb, lb)
    assert np.loadtxt(la._logfile, skiprows=1).shape[0] == 20


def test_a_starved_interval_falls_back_and_recovers(tmp_path):
    ma, la = build_mc3(tmp_path, "host", device=False)
    run_quiet(ma)
    mb, lb = build_mc3(tmp_path, "dev", device=True)
    OracleExchangeBackend.starve = {(2, 1): 7}      # (chain, interval of a batch) -> iterations it manages
    OracleExchangeBackend.exchange_slack = 1.5
    try:
        calls = []
        real = ex.run_exchange

        def spy(*a, **k):
            out = real(*a, **k)
            calls.append((a[3], out[0]))
            OracleExchangeBackend.starve = {(0, 2): 0} if len(calls) == 1 else {}
            return out
        ex.run_exchange = spy
        run_quiet(mb)
    finally:
        ex.run_exchange = real
        OracleExchangeBackend.starve = {}
    # the first 3 intervals (60 iterations) adapt the proposals and take the per-interval path; then a batch of the 3 left of
    # MC3's first 6, then batches of 6: both planted shortfalls were hit
    assert calls[0] == (3, 1) and calls[1] == (6, 2), calls
    assert OracleExchangeBackend.exchange_slack > 1.5
    OracleExchangeBackend.exchange_slack = 1.5
    compare(ma, la, mb, lb)


def _worker(rank, world, port, tmpdir, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from torch_dist_comm import TorchDistComm
        out = {}
        for name, device in (("host", False), ("dev", True)):
            comm = TorchDistComm()
            comm._comm = comm                  # the stand-in's "native handle": the communicator itself
            mc3, logger = build_mc3(tmpdir, name, comm=comm, device=device)
            if device:
                OracleExchangeBackend.starve = {(3, 2): 5}           # chain 3 lives on rank 1: rank 0 must see it in the records
                n_batches = []
                real = ex.run_exchange
                ex.run_exchange = lambda *a, **k: (lambda o: (n_batches.append(o[0]), o)[1])(real(*a, **k))
            run_quiet(mc3)
            if device:
                assert n_batches and n_batches[0] == 2, n_batches
            out[name] = (mc3, logger)
            dist.barrier()
        compare(out["host"][0], out["host"][1], out["dev"][0], out["dev"][1], rows=rank == 0)
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:      # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


def test_two_ranks_gloo_device_batches_equal_interval_by_interval(tmp_path):
    import multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
