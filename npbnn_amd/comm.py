"""Communicators for the MC3 temperature-swap exchange.

The only data that ever crosses chains is, once per swap interval, two float64 per chain
(log-posterior, temperature) and three integers back (j, k, accepted) — reference:
np_bnn/BNN_mc3.py:98-112, where the same scalars are read out of whole pickled chain objects
returned through a multiprocessing pool.

    LocalComm      world of one process (all chains in this process)
    RcclComm       one process per GPU, RCCL over xGMI through the C ABI (libnpbnn_hip.so)
    SocketComm     plain TCP through rank 0 (no GPU, no torch): rehearsals of the rank flow on fewer GPUs than ranks, CPU tests

(The product imports no torch.  The torch.distributed wrapper the gloo tests and `bench.py --dist-backend gloo|nccl` use lives with the
tests: tests/torch_dist_comm.py.)
"""
import os
import pickle
import socket
import struct
import time

import numpy as np


class LocalComm:
    rank, world_size = 0, 1

    def allgather_f64(self, vec):
        return np.asarray(vec, dtype=np.float64).reshape(1, -1)

    def bcast_i64(self, vec, root=0):
        return np.asarray(vec, dtype=np.int64)

    def bcast_obj(self, obj, root=0):
        return obj

    def barrier(self):
        pass

    def close(self):
        pass


def _exchange_unique_id(rank, world, uid, addr, port, timeout=120.0):
    """Rank 0 hands the 128-byte RCCL unique id to every other rank over a TCP socket."""
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        for _ in range(world - 1):
            conn, _ = srv.accept()
            conn.sendall(uid)
            conn.close()
        srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            s = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.05)
    buf = b""
    while len(buf) < 128:
        chunk = s.recv(128 - len(buf))
        if not chunk:
            raise ConnectionError("unique id exchange interrupted")
        buf += chunk
    s.close()
    return buf


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the connection")
        buf += chunk
    return bytes(buf)


class SocketComm:
    """The exchange over plain TCP: every rank keeps one connection to rank 0, which gathers and hands back.  For rehearsing
    the multi-rank flow where RCCL cannot run (several ranks on one GPU, CPU tests) - the payloads are a few dozen bytes per
    swap interval, so this is not a performance path.  Every receive has a deadline (``timeout`` seconds): a rank whose peer
    has gone fails with ``ConnectionError`` / ``socket.timeout`` instead of waiting for ever."""
    _comm = None          # (no native handle: exchange runs take the interval-by-interval path)

    def __init__(self, rank=None, world_size=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else int(world_size)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = port or (int(os.environ.get("MASTER_PORT", "29500")) + 19)
        self._peers, self._up = [], None
        if self.world_size == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world_size)
            srv.settimeout(timeout)
            by_rank = {}
            try:
                while len(by_rank) < self.world_size - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    by_rank[struct.unpack("<q", _recv_exact(conn, 8))[0]] = conn
            finally:
                srv.close()
            self._peers = [by_rank[r] for r in range(1, self.world_size)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.settimeout(timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(struct.pack("<q", self.rank))
            self._up = s

    def _gather_bytes(self, blob):
        """every rank's blob, in rank order, on every rank"""
        if self.world_size == 1:
            return [blob]
        if self.rank == 0:
            parts = [blob]
            for c in self._peers:
                n = struct.unpack("<q", _recv_exact(c, 8))[0]
                parts.append(_recv_exact(c, n))
            packed = pickle.dumps(parts)
            for c in self._peers:
                c.sendall(struct.pack("<q", len(packed)) + packed)
            return parts
        self._up.sendall(struct.pack("<q", len(blob)) + blob)
        n = struct.unpack("<q", _recv_exact(self._up, 8))[0]
        return pickle.loads(_recv_exact(self._up, n))

    def allgather_f64(self, vec):
        v = np.ascontiguousarray(vec, dtype=np.float64)
        return np.stack([np.frombuffer(b, dtype=np.float64) for b in self._gather_bytes(v.tobytes())])

    def bcast_i64(self, vec, root=0):
        v = np.ascontiguousarray(vec, dtype=np.int64)
        return np.frombuffer(self._gather_bytes(v.tobytes())[root], dtype=np.int64).copy()

    def bcast_obj(self, obj, root=0):
        return pickle.loads(self._gather_bytes(pickle.dumps(obj) if self.rank == root else b"")[root])

    def barrier(self):
        self._gather_bytes(b"")

    def close(self):
        for c in self._peers + ([self._up] if self._up is not None else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._up = [], None


def rccl_runtime():
    """(runtime version, version of the rccl.h the library was built against, path of the mapped librccl) - e.g.
    (22707, 22707, '/opt/rocm/lib/librccl.so.1')."""
    import ctypes as C
    from . import _capi as capi
    lib = capi.load_library()
    rt, hd = C.c_int(0), C.c_int(0)
    path = C.create_string_buffer(512)
    capi.check(lib, None, lib.npbnn_comm_runtime(C.byref(rt), C.byref(hd), path, 512))
    return rt.value, hd.value, path.value.decode()


class RcclComm:
    """RCCL communicator owned by the C library: ncclAllGather of the per-chain scalars and
    ncclBroadcast of the decision, on this rank's GPU stream (over xGMI inside a node)."""

    @staticmethod
    def make_unique_id():
        """A fresh 128-byte RCCL unique id (rank 0 creates it, every rank passes it to the constructor)."""
        import ctypes as C
        from . import _capi as capi
        lib = capi.load_library()
        uid = (C.c_char * 128)()
        capi.check(lib, None, lib.npbnn_comm_unique_id(uid))
        return bytes(uid.raw)

    def __init__(self, rank=None, world_size=None, device=None, addr=None, port=None, uid=None):
        """``uid``: the unique id from :meth:`make_unique_id` when the launcher already offers a channel to share it (e.g. a
        torch.distributed broadcast); default: rank 0 creates one and hands it out over a TCP socket on
        MASTER_ADDR:MASTER_PORT+17."""
        import ctypes as C
        from . import _capi as capi
        from .backend import default_device
        self._C = C
        self._lib = capi.load_library()
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
        dev = default_device() if device is None else device
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = port or (int(os.environ.get("MASTER_PORT", "29500")) + 17)
        raw = uid
        if raw is None:
            raw = self.make_unique_id() if self.rank == 0 else b""
            if self.world_size > 1:
                raw = _exchange_unique_id(self.rank, self.world_size, raw, addr, port)
        if len(raw) != 128:
            raise ValueError("RCCL unique id must be 128 bytes")
        uid = (C.c_char * 128)()
        C.memmove(uid, raw, 128)
        self._comm = C.c_void_p()
        capi.check(self._lib, None, self._lib.npbnn_comm_init(dev, self.rank, self.world_size, uid, C.byref(self._comm)))

    def allgather_f64(self, vec):
        from . import _capi as capi
        v = np.ascontiguousarray(vec, dtype=np.float64)
        out = np.empty((self.world_size, v.size), dtype=np.float64)
        capi.check(self._lib, None, self._lib.npbnn_comm_allgather_f64(self._comm, capi.dptr(v), v.size, capi.dptr(out)))
        return out

    def bcast_i64(self, vec, root=0):
        from . import _capi as capi
        v = np.ascontiguousarray(vec, dtype=np.int64).copy()
        capi.check(self._lib, None, self._lib.npbnn_comm_bcast_i64(
            self._comm, v.ctypes.data_as(self._C.POINTER(self._C.c_int64)), v.size, root))
        return v

    def bcast_obj(self, obj, root=0):
        """Small python objects (the cold chain's log row) as a length-prefixed byte broadcast."""
        blob = pickle.dumps(obj) if self.rank == root else b""
        n = int(self.bcast_i64(np.array([len(blob)]), root)[0])
        words = np.zeros((n + 7) // 8, dtype=np.int64)
        if self.rank == root:
            words.view(np.uint8)[:n] = np.frombuffer(blob, dtype=np.uint8)
        words = self.bcast_i64(words, root)
        return pickle.loads(words.view(np.uint8)[:n].tobytes())

    def barrier(self):
        self.allgather_f64(np.zeros(1))

    def describe(self):
        """What carries the exchange, for logs: ranks, RCCL version and the file it was mapped from."""
        rt, hd, path = rccl_runtime()
        return "rccl %d.%d.%d (%s; header %d.%d.%d), %d ranks, C ABI" % (rt // 10000, rt // 100 % 100, rt % 100, path, hd // 10000,
                                                                         hd // 100 % 100, hd % 100, self.world_size)

    def close(self):
        if getattr(self, "_comm", None):
            self._lib.npbnn_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
