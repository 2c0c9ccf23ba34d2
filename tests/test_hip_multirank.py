"""One process per GPU on real hardware (`-m gpu`): the MC3 swap exchange over RCCL between 2 and device_count ranks, both
the interval-by-interval path and the device exchange run, against the reference's golden swap sequence (mc3.npz) - and ranks
that fail or vanish must end the run, not hang it.  Needs two or more GPUs; on a one-GPU box the same rank flow is rehearsed
with every rank on GPU 0 and the TCP communicator in place of RCCL (RCCL refuses two ranks on one device)."""
import ctypes
import json
import os
import subprocess
import sys
import time

import pytest

from npbnn_amd.launch import spawn_ranks

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WORKER = [sys.executable, os.path.join(HERE, "rank_worker.py")]


def n_gpus():
    from npbnn_amd import _capi
    n = ctypes.c_int(0)
    _capi.load_library().npbnn_device_count(ctypes.byref(n))
    return n.value


def rank_counts():
    n = n_gpus()
    return sorted({2, n}) if n >= 2 else []


needs_two = pytest.mark.skipif("n_gpus() < 2", reason="RCCL between ranks needs two or more GPUs")


def test_rccl_runtime_is_the_one_the_library_was_built_against():
    from npbnn_amd.comm import rccl_runtime
    rt, hd, path = rccl_runtime()
    assert rt // 100 == hd // 100 and "rccl" in path
    assert "torch" not in sys.modules and "torch" not in path


@needs_two
@pytest.mark.parametrize("device_exchange", ["0", "1"])
def test_mc3_over_rccl_follows_the_reference(device_exchange, tmp_path):
    for world in rank_counts():
        d = tmp_path / ("w%d" % world)
        d.mkdir()
        status, out0, outs = spawn_ranks(WORKER + ["mc3", str(d), "hip", "rccl", device_exchange], world, capture_all=True, timeout=600)
        assert status == 0, "\n".join(outs)
        for r in range(world):
            assert "RANK %d OK" % r in outs[r]
        assert "comm: rccl" in out0 and "/opt/rocm" in out0


@needs_two
def test_a_failing_rank_ends_every_rank_within_ten_seconds(tmp_path):
    world = rank_counts()[-1]
    status, out0, outs = spawn_ranks(WORKER + ["mc3", str(tmp_path), "hip", "rccl", "0", "fail=1:raise"], world, capture_all=True, timeout=600)
    left = float(open(os.path.join(str(tmp_path), "failed_at")).read())
    assert status != 0 and time.time() - left < 10, "\n".join(outs)
    assert not any("RANK %d OK" % r in outs[r] for r in range(world))


@needs_two
def test_a_rank_that_vanishes_does_not_hang_its_peers(tmp_path):
    """Rank 1 leaves with status 0 (nothing for the launcher to react to): the others' next collective never completes; the
    bounded wait of the communicator (NPBNN_COMM_TIMEOUT_S) aborts it and the ranks end with an error."""
    env = dict(os.environ, NPBNN_COMM_TIMEOUT_S="4")
    status, out0, outs = spawn_ranks(WORKER + ["mc3", str(tmp_path), "hip", "rccl", "0", "fail=1:exit0"], 2, env=env, capture_all=True, timeout=300)
    left = float(open(os.path.join(str(tmp_path), "failed_at")).read())
    assert status not in (0, -9) and time.time() - left < 15, "\n".join(outs)
    assert "RANK 0 OK" not in out0


def test_rank_flow_rehearsal_on_one_gpu(tmp_path):
    """Two ranks, both on GPU 0, the TCP communicator in place of RCCL: the process-per-chain flow (device contexts in separate
    processes, host all-gather of [logPost, temperature], identical decisions on every rank, cold-chain logging through rank 0)
    against the golden swap sequence."""
    status, out0, outs = spawn_ranks(WORKER + ["mc3", str(tmp_path), "hip", "socket", "0"], 2, capture_all=True, timeout=600)
    assert status == 0, "\n".join(outs)
    assert "RANK 0 OK" in out0 and "RANK 1 OK" in outs[1]


def test_bench_two_ranks_without_torch(tmp_path):
    """`bench.py --gpus 2` started without a launcher: spawns its ranks, no torch in any of them, one JSON line from rank 0.
    Over RCCL when two GPUs are there, else the TCP rehearsal on one."""
    env = dict(os.environ)
    if n_gpus() < 2:
        env["NPBNN_BENCH_DIST_BACKEND"] = "socket"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["chains"] == 2 and line["config"]["swap_exchange_nranks"] == 2
    assert line["torch_imported"] is False and line["value"] > 0
    if n_gpus() >= 2:
        assert "rccl" in line["config"]["swap_exchange"] and "/opt/rocm" in line["config"]["swap_exchange"]
    leg = line["row_sharded_chain"]          # the other multi-GPU form: ONE chain of config 4, its rows split over the two ranks
    assert "error" not in leg, leg
    assert leg["ranks"] == 2 and sum(leg["rows_per_rank"]) == 1_000_000 and leg["ranks_hold_the_same_chain"] and leg["value"] > 0
    assert leg["schedule"] == 1 and ("ncclAllGather" in leg["gather"]) == (n_gpus() >= 2)


def test_bench_four_ranks_on_one_gpu_rehearsal(tmp_path):
    """`bench.py --gpus 4`'s exact code path with every rank on GPU 0 and the TCP communicator in place of RCCL (which refuses
    several ranks on one device): four chains, one per rank, the swap proposal after every step, max-over-ranks timing, then the
    row-sharded chain over the four ranks.  (Four, not eight: a GPU box admits six processes on its card; the eight-rank run is
    the driver's, on eight GPUs.)"""
    env = dict(os.environ, NPBNN_BENCH_DIST_BACKEND="socket")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 4 and line["config"]["chains"] == 4 and line["config"]["swap_exchange_nranks"] == 4
    assert line["scaling"] == "weak" and line["torch_imported"] is False and line["value"] > 0
    assert "tcp" in line["config"]["swap_exchange"] and line["config"]["swap_exchange_path"].startswith("host")
    leg = line["row_sharded_chain"]
    assert "error" not in leg, leg
    assert leg["ranks"] == 4 and sum(leg["rows_per_rank"]) == 1_000_000 and leg["ranks_hold_the_same_chain"]


def test_bench_carries_on_over_tcp_when_rccl_does_not_come_up(tmp_path):
    """`bench.py --gpus 2` whose RCCL communicator fails to come up (NPBNN_BENCH_RCCL_FAULT: every rank raises where it would build
    it): the ranks agree on it over the TCP channel, carry the swap exchange over that channel and the line says what happened -
    a node whose RCCL set-up is broken still yields its per-N line."""
    env = dict(os.environ, NPBNN_BENCH_RCCL_FAULT="1", NPBNN_BENCH_NO_ROW_SHARD="1")
    env.pop("NPBNN_BENCH_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["swap_exchange_nranks"] == 2 and line["value"] > 0
    assert "did not come up on rank(s) [0, 1]" in line["config"]["swap_exchange"]
    assert line["config"]["swap_exchange_path"].startswith("host")


def test_bench_under_the_drivers_launcher(tmp_path):
    """The driver's command for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` - the ranks take RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher's
    environment (the unique-id / TCP rendezvous of this package sits beside the launcher's own store, on MASTER_PORT + 17 / + 19)
    and still import no torch themselves.  Two ranks; on one GPU over the TCP communicator."""
    from npbnn_amd.launch import free_port
    env = dict(os.environ)
    if n_gpus() < 2:
        env["NPBNN_BENCH_DIST_BACKEND"] = "socket"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["swap_exchange_nranks"] == 2 and line["torch_imported"] is False and line["value"] > 0
    assert "error" not in line["row_sharded_chain"], line["row_sharded_chain"]


# ---- one chain, rows split over ranks (npbnn_set_row_shard; npbnn_amd/rowshard.py) ----
@pytest.mark.parametrize("case", ["cls", "clsw", "reg", "regsig"])
def test_row_sharded_chain_two_processes_on_one_gpu(case):
    """Two ranks on GPU 0, each holding half of the rows; the per-pass records travel through the host (TCP communicator - RCCL refuses
    two ranks on one device): npbnn_chain_run's sharded batches (pass, record, gather, step) and the sharded single evaluations take
    the decisions of the same chain on all rows in one context."""
    status, out0, outs = spawn_ranks(WORKER + ["rowshard", "hip", "socket", case], 2, capture_all=True, timeout=600)
    assert status == 0, "\n".join(outs)
    assert "RANK 0 OK" in out0 and "RANK 1 OK" in outs[1]


@needs_two
@pytest.mark.parametrize("case", ["cls", "reg"])
def test_row_sharded_chain_over_rccl(case):
    """The same with one GPU per rank and the records all-gathered by RCCL on the chains' streams."""
    for world in rank_counts():
        status, out0, outs = spawn_ranks(WORKER + ["rowshard", "hip", "rccl", case], world, capture_all=True, timeout=600)
        assert status == 0, "\n".join(outs)
        assert all("RANK %d OK" % r in outs[r] for r in range(world))
