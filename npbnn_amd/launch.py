"""Starting one process per GPU without a launcher, and ending all of them when one fails.

The reference runs its chains in a ``multiprocessing.Pool`` of the parent process (np_bnn/BNN_mc3.py:87-96); here a chain lives
in a process of its own next to its GPU.  ``spawn_ranks`` is what ``bench.py --gpus N`` and the multi-rank tests use when nothing
like ``torch.distributed.run`` started the ranks: fresh child processes (never a fork or an exec of a process that has touched a
GPU), the launcher's environment variables (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT), rank 0's
standard output relayed - and fail-stop semantics: the moment any rank ends with a non-zero status the others are killed (their
own process ids only), so nobody is left waiting in a collective for a peer that is gone.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(argv, n_ranks, env=None, timeout=None, capture_all=False, poll_s=0.02):
    """Run ``argv`` as ``n_ranks`` processes.  Returns ``(status, out0, outs)``: the first non-zero exit status (0 when every rank
    ended well; -9 for a run that hit ``timeout`` seconds), rank 0's standard output, and - with ``capture_all`` - every rank's
    combined output (else the other ranks write to this process's standard error)."""
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    children, readers, chunks = [], [], []
    for r in range(n_ranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        piped = r == 0 or capture_all
        p = subprocess.Popen(list(argv), env=e, stdout=subprocess.PIPE if piped else sys.stderr,
                             stderr=subprocess.STDOUT if capture_all else sys.stderr, text=True)
        children.append(p)
        buf = []
        chunks.append(buf)
        if piped:       # a reader per pipe: nobody blocks on a full pipe, and the poll loop below never waits for output
            t = threading.Thread(target=lambda f=p.stdout, b=buf: b.extend(iter(f.readline, "")), daemon=True)
            t.start()
            readers.append(t)
    status, t0 = 0, time.time()
    pending = list(children)
    try:
        while pending:
            for c in list(pending):
                rc = c.poll()
                if rc is None:
                    continue
                pending.remove(c)
                if rc != 0 and status == 0:
                    status = rc
            if status != 0 or (timeout is not None and time.time() - t0 > timeout):
                if status == 0:
                    status = -9
                break
            time.sleep(poll_s)
    finally:
        for c in children:          # a rank failed (or the run timed out): the others would wait for it in a collective
            if c.poll() is None:
                c.kill()
        for c in children:
            try:
                c.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        for t in readers:
            t.join(timeout=5)
    outs = ["".join(b) for b in chunks]
    return status, outs[0], outs
