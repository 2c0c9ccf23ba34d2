"""npbnn_amd — MI355X-native backend for npBNN's MCMC hot path.

The per-proposal forward pass + likelihood of the Metropolis-Hastings loop
runs in hand-written HIP kernels (gfx950) behind the C ABI declared in
``include/npbnn_hip.h``; host code is Python + ctypes.  There is no CPU
fallback: without the built library and an MI355X the device-backed calls
raise ``BackendUnavailable``.
"""
__version__ = "0.1.0"

from ._capi import BackendUnavailable, NpbnnError  # noqa: F401
from .backend import HipContext, pack_weights  # noqa: F401
