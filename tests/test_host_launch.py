"""Starting ranks without a launcher and losing one of them, on CPU: npbnn_amd.launch.spawn_ranks (fail-stop), the TCP
communicator, and the reference's golden MC3 run over two torch-free ranks (oracle-backed stand-in for the device)."""
import os
import sys
import time

import numpy as np

from npbnn_amd.launch import spawn_ranks

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = [sys.executable, os.path.join(HERE, "rank_worker.py")]


def test_a_failing_rank_takes_the_others_down_at_once():
    t0 = time.time()
    status, out0, outs = spawn_ranks(WORKER + ["idle", "60", "fail=1:raise"], 3, capture_all=True)
    assert status == 1 and time.time() - t0 < 10
    assert "planted failure" in outs[1] and "OK" not in out0


def test_ranks_that_end_well_report_zero():
    status, out0, outs = spawn_ranks(WORKER + ["idle", "0.1"], 2, capture_all=True)
    assert status == 0 and "RANK 0 OK" in out0 and "RANK 1 OK" in outs[1]


def test_a_run_past_its_deadline_is_ended():
    t0 = time.time()
    status, _, _ = spawn_ranks(WORKER + ["idle", "60"], 2, timeout=1.0, capture_all=True)
    assert status == -9 and time.time() - t0 < 15


def test_mc3_two_ranks_over_tcp_match_the_reference(tmp_path):
    """World size 2 without torch: chains 0, 2 on rank 0 and 1, 3 on rank 1, [logPost, temperature] all-gathered over the TCP
    communicator; the reference's golden swap sequence, final states and log rows."""
    status, out0, outs = spawn_ranks(WORKER + ["mc3", str(tmp_path), "oracle", "socket", "0"], 2, capture_all=True, timeout=600)
    assert status == 0, "\n".join(outs)
    assert "RANK 0 OK" in out0 and "RANK 1 OK" in outs[1]


def test_mc3_one_chain_per_rank_over_tcp_matches_the_reference(tmp_path):
    """The layout of an 8-GPU node in small: as many ranks as chains (the golden run has four), ONE chain per rank, every swap
    decided from an all-gather of [logPost, temperature] over all ranks, the cold chain's row shipped to rank 0 whichever rank
    holds it - the reference's golden swap sequence, final states and log rows on every rank."""
    status, out0, outs = spawn_ranks(WORKER + ["mc3", str(tmp_path), "oracle", "socket", "0"], 4, capture_all=True, timeout=600)
    assert status == 0, "\n".join(outs)
    assert all("RANK %d OK" % r in outs[r] for r in range(4))


def test_a_rank_that_leaves_quietly_does_not_hang_its_peer(tmp_path):
    """Rank 1 exits with status 0 in the middle of the run (the launcher sees nothing wrong): rank 0 must fail by itself - its
    next exchange finds the connection closed - instead of waiting for ever."""
    env = dict(os.environ, NPBNN_SOCKET_TIMEOUT_S="5")
    status, out0, outs = spawn_ranks(WORKER + ["mc3", str(tmp_path), "oracle", "socket", "0", "fail=1:exit0"], 2, env=env,
                                     capture_all=True, timeout=120)
    assert status not in (0, -9), "\n".join(outs)
    left = float(open(os.path.join(str(tmp_path), "failed_at")).read())
    assert time.time() - left < 10 and "RANK 0 OK" not in out0
    assert "ConnectionError" in out0 or "timed out" in out0 or "Broken pipe" in out0 or "reset" in out0, out0


def test_socket_comm_collectives():
    import threading
    from npbnn_amd.comm import SocketComm
    from npbnn_amd.launch import free_port
    port, world, got = free_port(), 3, {}

    def run(r):
        c = SocketComm(rank=r, world_size=world, addr="127.0.0.1", port=port, timeout=20)
        got[r] = (c.allgather_f64([r, 10.0 * r]), c.bcast_i64([r + 5, 7], root=2), c.bcast_obj({"from": r}, root=1))
        c.barrier()
        c.close()
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    for r in range(world):
        a, b, o = got[r]
        np.testing.assert_array_equal(a, [[0, 0], [1, 10], [2, 20]])
        np.testing.assert_array_equal(b, [7, 7])
        assert o == {"from": 1}


def test_bench_deadline_helper_returns_results_errors_and_gives_up_on_time():
    """bench.py's guard around its multi-rank set-up and the leg after the line: a result comes back as it is, an exception as its
    text, and a call that never returns is left behind when the deadline passes."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert bench._with_deadline(lambda: 7, 5.0) == (7, None)
    out, why = bench._with_deadline(lambda: 1 / 0, 5.0)
    assert out is None and why.startswith("ZeroDivisionError")
    t0 = time.time()
    out, why = bench._with_deadline(lambda: time.sleep(30), 0.3)
    assert out is None and why.startswith("not back within") and time.time() - t0 < 5
