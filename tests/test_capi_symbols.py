"""CPU-side checks of the drop-in boundary: the shared library loads and
exports every symbol include/npbnn_hip.h declares (no compute calls: there is
no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names.update(re.findall(r"\b(npbnn_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_header_declares_entry_points():
    syms = declared_symbols()
    for must in ("npbnn_create", "npbnn_destroy", "npbnn_set_data_f64", "npbnn_set_arch", "npbnn_eval",
                 "npbnn_predict", "npbnn_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from npbnn_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), "libnpbnn_hip.so does not export %s" % s
    assert lib.npbnn_abi_version() == 1


def test_binding_covers_every_declared_symbol():
    from npbnn_amd import _capi
    assert sorted(_capi.SIGNATURES) == declared_symbols()


def test_missing_library_fails_loudly(tmp_path):
    from npbnn_amd import _capi
    with pytest.raises(_capi.BackendUnavailable):
        _capi.load_library(str(tmp_path / "nope.so"))


def test_no_device_fails_loudly():
    """On a box without a GPU the product must raise, not fall back."""
    from npbnn_amd import _capi, HipContext
    lib = _capi.load_library()
    n = ctypes.c_int(0)
    lib.npbnn_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_capi.BackendUnavailable):
        HipContext()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "npbnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), fn
