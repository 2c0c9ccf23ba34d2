"""torch.distributed as a communicator of the MC3 temperature-swap exchange (test / bench plumbing - the product package imports no
torch: its own communicators are LocalComm, RcclComm through the C ABI, and SocketComm; npbnn_amd/comm.py).  gloo on CPU for the
world-size-2 tests; "nccl" = RCCL on GPUs when bench.py is asked for it by name."""
import numpy as np


class TorchDistComm:
    """Wraps an initialised torch.distributed default group (plumbing only)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        self._device = device if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))

    def allgather_f64(self, vec):
        t = self._torch.tensor(np.asarray(vec, dtype=np.float64), device=self._device)
        out = [self._torch.empty_like(t) for _ in range(self.world_size)]
        self._dist.all_gather(out, t)
        return np.stack([o.cpu().numpy() for o in out])

    def bcast_i64(self, vec, root=0):
        t = self._torch.tensor(np.asarray(vec, dtype=np.int64), device=self._device)
        self._dist.broadcast(t, src=root)
        return t.cpu().numpy()

    def bcast_obj(self, obj, root=0):
        box = [obj if self.rank == root else None]
        self._dist.broadcast_object_list(box, src=root, device=self._device if self._device.type == "cuda" else None)
        return box[0]

    def barrier(self):
        self._dist.barrier()

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
