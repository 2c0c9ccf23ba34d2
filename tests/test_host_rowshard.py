"""One chain, rows split over ranks (npbnn_amd/rowshard.py), on CPU: two and three torch-free ranks over the TCP communicator, the
device stood in for by the oracle - the sharded chain takes the decisions of the same chain on all rows (SURVEY 8(e) alternative;
reference shape np_bnn/BNN_env.py:467-491: one sum over all rows per proposal)."""
import os
import sys

import numpy as np
import pytest

from npbnn_amd.launch import spawn_ranks
from npbnn_amd.rowshard import shard_bounds, shard_rows

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = [sys.executable, os.path.join(HERE, "rank_worker.py")]


def test_shares_are_contiguous_and_cover_every_row():
    for n, world in ((10, 3), (7, 7), (100000, 8), (5, 2)):
        b = [shard_bounds(n, r, world) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_shard_rows_cuts_every_per_row_array_and_nothing_else():
    dat = dict(data=np.arange(22.).reshape(11, 2), labels=np.arange(11), test_data=np.zeros((4, 2)), test_labels=np.arange(4),
               id_data=np.arange(11).astype(str), feature_names=["a", "b"], file_name="x")
    parts = [shard_rows(dat, r, 3) for r in range(3)]
    np.testing.assert_array_equal(np.concatenate([p["data"] for p in parts]), dat["data"])
    np.testing.assert_array_equal(np.concatenate([p["labels"] for p in parts]), dat["labels"])
    np.testing.assert_array_equal(np.concatenate([p["test_labels"] for p in parts]), dat["test_labels"])
    np.testing.assert_array_equal(np.concatenate([p["id_data"] for p in parts]), dat["id_data"])
    assert parts[1]["feature_names"] == ["a", "b"] and parts[2]["file_name"] == "x"
    assert len(dat["data"]) == 11          # (the caller's dictionary is untouched)


@pytest.mark.parametrize("case,world", [("cls", 2), ("clsw", 3), ("reg", 2), ("regsig", 2)])
def test_a_row_sharded_chain_is_the_chain_on_all_rows(case, world):
    status, out0, outs = spawn_ranks(WORKER + ["rowshard", "oracle", "socket", case], world, capture_all=True, timeout=600)
    assert status == 0, "\n".join(outs)
    assert all("RANK %d OK" % r in outs[r] for r in range(world))
