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
