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
    assert lib.nhdfit_abi_version() == _lib.ABI_VERSION == 9


def test_ship_build_carries_no_profiling_kernels():
    """The stand-alone role kernels (k_role, k_fit_only) and the environment knobs belong to libnhdfit_tuning.so (-DNHDFIT_TUNING)
    only: the shipped library launches none of them, so it does not hold their code either."""
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"k_role", b"k_fit_only", b"NHDFIT_FIT_SKIP", b"NHDFIT_ROLE_KERNELS"):
        assert name not in blob, name


def test_struct_sizes_match_header():
    # sizes asserted on the C side by the struct comments; here: numpy mirrors
    assert pack.REQ.itemsize == 128 and pack.DETAIL.itemsize == 128 and pack.MAPPING.itemsize == 20 and pack.PLACEMENT.itemsize == 256 and pack.WIDE.itemsize == 640 and pack.WIDE_PLACEMENT.itemsize == 480
    assert ctypes.sizeof(_lib.Stats) == 88


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


def test_null_context_is_an_error_code_not_a_crash():
    """Every entry point that takes a context answers NHDFIT_E_INVAL to a NULL one (nothing aborts across the boundary)."""
    lib = _lib.load()
    buf = (ctypes.c_uint8 * 256)()
    assert lib.nhdfit_apply_deltas(None, buf, 1, buf) == -1
    assert lib.nhdfit_upload_origin(None, 0, 1, buf) == -1
    assert lib.nhdfit_upload_nodes(None, 0, 1, buf, buf, buf, buf, buf, buf) == -1
    assert lib.nhdfit_download_nodes(None, 0, 1, buf, buf, buf, buf, buf, buf) == -1
    assert lib.nhdfit_find(None, buf, 1, 0.0, None, buf, None, None) == -1
    assert lib.nhdfit_commit(None, 0, buf, buf, 0.0, buf) == -1
    assert lib.nhdfit_stage_requests(None, buf, 1) == -1
    assert lib.nhdfit_enqueue_step(None, 0.0) == -1
    assert lib.nhdfit_sync(None) == -1
    assert lib.nhdfit_set_node_count(None, 0) == -1
    assert lib.nhdfit_reserve_nodes(None, 1, 0) == -1


def test_delta_and_origin_records_match_header():
    assert pack.ORIGIN.itemsize == 80 and pack.DELTA.itemsize == 96
    # field offsets the device code relies on (include/nhdfit.h)
    assert pack.DELTA.fields["t0"][1] == 8 and pack.DELTA.fields["gpus"][1] == 40 and pack.DELTA.fields["groups"][1] == 64
    assert pack.DELTA.fields["busy_time"][1] == 72 and pack.DELTA.fields["nic_n"][1] == 80
    assert pack.DETAIL.fields["nic_pods"][1] == 82 and pack.DETAIL.fields["gpu_sw"][1] == 96
    assert pack.ORIGIN.fields["nic_base"][1] == 32 and pack.ORIGIN.fields["hp_total"][1] == 64


def test_placement_record_matches_header():
    assert pack.PLACEMENT.itemsize == 256 and pack.WIDE.itemsize == 640 and pack.WIDE_PLACEMENT.itemsize == 480
    assert pack.PLACEMENT.fields["misc_take"][1] == 128 and pack.PLACEMENT.fields["gpu"][1] == 144 and pack.PLACEMENT.fields["numa"][1] == 176
    assert pack.PLACEMENT.fields["status"][1] == 181 and pack.PLACEMENT.fields["proc_late"][1] == 184 and pack.PLACEMENT.fields["misc_late"][1] == 248


def test_big_request_records_match_header():
    """nhdfit_big_req / nhdfit_big_mapping / nhdfit_big_placement (pods with 5..8 processing groups): sizes and the offsets
    `gcc offsetof` gives for include/nhdfit.h."""
    assert (pack.BIG_REQ.itemsize, pack.BIG_MAPPING.itemsize, pack.BIG_PLACEMENT.itemsize) == (256, 36, 904)
    f = pack.BIG_REQ.fields
    assert {k: f[k][1] for k in ("gpus", "cpu_smt", "misc_smt", "smt_bits", "n_misc", "rx", "tx", "n_proc", "n_help", "nic_use")} == \
        {"gpus": 24, "cpu_smt": 40, "misc_smt": 72, "smt_bits": 76, "n_misc": 78, "rx": 80, "tx": 144, "n_proc": 208, "n_help": 216, "nic_use": 224}
    f = pack.BIG_MAPPING.fields
    assert {k: f[k][1] for k in ("cpu", "nic_numa", "nic_idx", "valid")} == {"cpu": 8, "nic_numa": 17, "nic_idx": 25, "valid": 33}
    f = pack.BIG_PLACEMENT.fields
    assert {k: f[k][1] for k in ("help_take", "misc_take", "gpu", "numa", "status", "pod", "node")} == \
        {"help_take": 384, "misc_take": 768, "gpu": 816, "numa": 880, "status": 889, "pod": 892, "node": 896}
    lib = _lib.load()
    buf = (ctypes.c_uint8 * 1024)()
    assert lib.nhdfit_big_find(None, buf, 1, 0.0, None, buf, None) == -1
    assert lib.nhdfit_big_commit(None, 0, buf, buf, 0.0, buf) == -1


def test_binding_refuses_a_library_of_another_abi(monkeypatch):
    """nhd_amd/_lib.py compares nhdfit_abi_version() with the version its record layouts are written for."""
    assert _lib.load().nhdfit_abi_version() == _lib.ABI_VERSION
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    import pytest
    with pytest.raises(_lib.NhdFitError):
        _lib.load()
