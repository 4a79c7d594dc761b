"""CPU-side checks of the C-ABI: the library builds, loads, exports every symbol the header declares,
and refuses to run without a GPU (no silent CPU fallback)."""
import ctypes
import os

import pytest

from nhd_amd import _lib, pack


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib._SIGS)
    assert lib.nhdfit_abi_version() == 5


def test_struct_sizes_match_header():
    # sizes asserted on the C side by the struct comments; here: numpy mirrors
    assert pack.REQ.itemsize == 128 and pack.DETAIL.itemsize == 128 and pack.MAPPING.itemsize == 20 and pack.PLACEMENT.itemsize == 184
    assert ctypes.sizeof(_lib.Stats) == 72


def test_create_without_gpu_fails_loudly():
    lib = _lib.load()
    if lib.nhdfit_device_count() > 0:
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    rc = lib.nhdfit_create(0, ctypes.byref(h))
    assert rc == -2 and not h.value
    assert b"no CPU path" in lib.nhdfit_last_error(None)
    from nhd_amd.matcher import HipMatcher
    with pytest.raises(_lib.NhdFitError):
        HipMatcher()
