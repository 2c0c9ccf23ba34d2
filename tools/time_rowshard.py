#!/usr/bin/env python3
"""Wall time of the row-sharded rehearsal (two ranks on one GPU, TCP communicator): python tools/time_rowshard.py [case]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npbnn_amd.launch import spawn_ranks  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "cls"
t0 = time.time()
status, out0, outs = spawn_ranks([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), "rowshard", "hip", "socket", case], 2,
                                 capture_all=True, timeout=600)
print("status", status, "in %.1f s" % (time.time() - t0))
print(outs[0][-3000:])
