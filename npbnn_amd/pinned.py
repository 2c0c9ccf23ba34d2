"""Page-locked host arrays for the per-batch inputs of the device-resident chain (npbnn_pinned_alloc).

A pre-drawn batch is K x M indices and deviates (5 KB per iteration on BASELINE config 2); drawn straight into
page-locked memory it is uploaded asynchronously at link speed.  Blocks are recycled through a small pool, because
locking pages costs far more than a batch upload."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

from . import _capi as capi


_TRACE = bool(os.environ.get("NPBNN_PINNED_TRACE"))      # diagnostics: every new page-locked block, its size and what locking it cost


class _Block:
    __slots__ = ("pool", "ptr", "nbytes", "__weakref__")

    def __init__(self, pool, ptr, nbytes):
        self.pool, self.ptr, self.nbytes = pool, ptr, nbytes

    def __del__(self):
        pool = self.pool
        if pool is not None:
            pool._give_back(self.ptr, self.nbytes)


class PinnedPool:
    def __init__(self, lib=None):
        self._lib = lib if lib is not None else capi.load_library()
        self._free = {}
        self._lock = threading.Lock()

    def _give_back(self, ptr, nbytes):
        try:
            with self._lock:
                self._free.setdefault(nbytes, []).append(ptr)
        except Exception:       # interpreter shutdown
            pass

    def empty_group(self, specs):
        """Uninitialised arrays ``[(shape, dtype), ...]`` in ONE page-locked block, each starting at the next multiple of 256
        bytes: npbnn_chain_run uploads neighbours laid out like this (indices, then deviates) in a single copy."""
        offsets, total = [], 0
        for shape, dtype in specs:
            total = (total + 255) // 256 * 256
            offsets.append(total)
            total += int(np.prod(shape)) * np.dtype(dtype).itemsize
        raw = self._raw(total)
        return [np.ndarray(shape, dtype=np.dtype(dtype), buffer=raw, offset=off) for (shape, dtype), off in zip(specs, offsets)]

    def empty(self, shape, dtype):
        """An uninitialised C-contiguous array in page-locked memory (returned to the pool when the last view dies)."""
        dtype = np.dtype(dtype)
        return np.ndarray(shape, dtype=dtype, buffer=self._raw(int(np.prod(shape)) * dtype.itemsize))

    def _raw(self, n):
        nbytes = 1 << max(12, (max(n, 1) - 1).bit_length())
        with self._lock:
            lst = self._free.get(nbytes)
            ptr = lst.pop() if lst else None
        if ptr is None:
            out = C.c_void_p()
            t0 = time.perf_counter() if _TRACE else 0.0
            rc = self._lib.npbnn_pinned_alloc(nbytes, C.byref(out))
            if _TRACE:
                sys.stderr.write("[npbnn pinned] %d KiB page-locked in %.2f ms (thread %s)\n"
                                 % (nbytes >> 10, (time.perf_counter() - t0) * 1e3, threading.current_thread().name))
            if rc != 0 or not out.value:
                msg = self._lib.npbnn_last_error(None)
                raise capi.NpbnnError(rc, "npbnn_pinned_alloc(%d): %s" % (nbytes, msg.decode() if msg else "?"))
            ptr = out.value
        raw = (C.c_char * nbytes).from_address(ptr)
        raw._npbnn_block = _Block(self, ptr, nbytes)       # lives as long as any array built on `raw`
        return raw
