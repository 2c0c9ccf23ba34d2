"""MC3 host logic on CPU: in-process chains and a 2-rank gloo run must both reproduce the reference's
golden swap sequence and final chain states (oracle-backed test backend, float64)."""
import contextlib
import io
import os
import socket
import sys

import numpy as np
import pytest

import cases
import npbnn_amd as bn
from oracle_backend import OracleBackend, serve_from_oracle

RTOL = 1e-9


def build_mc3(tmpdir, comm=None, name="mc3"):
    cfg = cases.MC3_TRACES[name]
    dat = cases.classification_data(cfg["seed"], cfg["n_rows"], cfg["n_features"], cfg["n_classes"], cfg["n_test"])
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=cfg["n_nodes"], use_bias_node=cfg["bias"], seed=1, init_std=0.1, **cases.mc3_act(bn, cfg))
        rank0 = comm is None or comm.rank == 0
        logger = bn.postLogger(bnn, filename="MC3", wdir=str(tmpdir), log_all_weights=0, continue_logfile=not rank0)
        serve_from_oracle(lambda b: OracleBackend(b, 0))
        mc3 = bn.MC3(bnn, logger=logger, n_post_samples=10, sampling_f=cfg["swap_frequency"],
                     n_iteration=cfg["n_iteration"], n_chains=cfg["n_chains"], swap_frequency=cfg["swap_frequency"],
                     verbose=0, comm=comm)
    return mc3, logger


def check_against_golden(mc3, g, local_only=False):
    accepted = [i for i, s in enumerate(mc3.swap_log) if s[4]]
    assert accepted == [int(r[0]) for r in g["swapped"]]
    for i, pair in enumerate(mc3.singleChainArgs):
        if pair is None:
            assert local_only
            continue
        bnn_i, mcmc_i = pair
        assert mcmc_i._temperature == g["final_temperature"][i]
        np.testing.assert_allclose(mcmc_i._logPost, g["final_logPost"][i], rtol=RTOL)
        np.testing.assert_allclose(mcmc_i._logLik, g["final_logLik"][i], rtol=RTOL)
        np.testing.assert_allclose(mcmc_i._acceptance_rate, g["final_acc_rate"][i], rtol=RTOL)
        for li, w in enumerate(bnn_i._w_layers):
            np.testing.assert_array_equal(w, g["w_c%d_l%d" % (i, li)])
        if bnn_i._act_fun._trainable:
            np.testing.assert_array_equal(np.asarray(bnn_i._act_fun._acc_prm, dtype=float), g["alphas_c%d" % i])


def test_mc3_with_trainable_slopes_matches_reference(golden_dir, tmp_path):
    """MC3 whose chains propose their own activation slopes (ActFun("genReLU", trainable=True)): the reference's swap sequence,
    final states, accepted slopes of every chain and the cold chain's log rows (alpha columns included)."""
    g = np.load(os.path.join(golden_dir, "mc3_slopes.npz"))
    mc3, logger = build_mc3(tmp_path, name="mc3_slopes")
    np.testing.assert_array_equal(mc3.rseeds, g["rseeds"])
    with contextlib.redirect_stdout(io.StringIO()):
        mc3.run_mcmc()
    check_against_golden(mc3, g)
    rows = np.loadtxt(logger._logfile, skiprows=1)
    np.testing.assert_allclose(rows, g["log_rows"], rtol=1e-8)


def test_mc3_in_process_matches_reference(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "mc3.npz"))
    mc3, logger = build_mc3(tmp_path)
    np.testing.assert_array_equal(mc3.rseeds, g["rseeds"])
    with contextlib.redirect_stdout(io.StringIO()):
        mc3.run_mcmc()
    check_against_golden(mc3, g)
    rows = np.loadtxt(logger._logfile, skiprows=1)
    np.testing.assert_allclose(rows, g["log_rows"], rtol=1e-8)
    # the pickle holds [bnn, mcmc, logger] with the posterior weight samples
    b, m, lg = bn.load_obj(logger._pklfile)
    assert len(lg._post_weight_samples) == 10 and set(lg._post_weight_samples[-1]) >= {"weights", "alphas", "mcmc_it"}


def _worker(rank, world, port, tmpdir, golden_dir, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from torch_dist_comm import TorchDistComm
        comm = TorchDistComm()
        g = np.load(os.path.join(golden_dir, "mc3.npz"))
        mc3, logger = build_mc3(tmpdir, comm)
        assert mc3.local_ids == [i for i in range(4) if i % world == rank]
        with contextlib.redirect_stdout(io.StringIO()):
            mc3.run_mcmc()
        check_against_golden(mc3, g, local_only=True)
        dist.barrier()
        if rank == 0:
            rows = np.loadtxt(logger._logfile, skiprows=1)
            np.testing.assert_allclose(rows, g["log_rows"], rtol=1e-8)
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:      # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


def test_mc3_two_ranks_gloo_matches_reference(golden_dir, tmp_path):
    """World size 2 over gloo: chains 0,2 on rank 0 and 1,3 on rank 1; only scalars are exchanged."""
    import multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
