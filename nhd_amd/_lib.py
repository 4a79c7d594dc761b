"""ctypes binding of libnhdfit.so (include/nhdfit.h).  This is the stub a maintainer of the
reference would add; there is no fallback: a missing library is an error."""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import POINTER, c_char_p, c_double, c_int, c_uint32, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# NHDFIT_LIBRARY: load another build of the same ABI instead (tools/: the tuning build libnhdfit_tuning.so)
LIB_PATH = os.environ.get("NHDFIT_LIBRARY") or os.path.join(HERE, "libnhdfit.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "nhdfit.h")
ABI_VERSION = 9                  # NHDFIT_ABI_VERSION of include/nhdfit.h this binding (and pack.py's record layouts) is written for


class NhdFitError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libnhdfit error {code}: {msg}")
        self.code = code


class Stats(ctypes.Structure):
    _fields_ = [("launches", c_uint64), ("fit_ms_total", c_double), ("fit_ms_last", c_double),
                ("digest_ms_last", c_double), ("step_ms_last", c_double), ("evals_last", c_uint64),
                ("bytes_last", c_uint64), ("nodes", c_uint32), ("nsig", c_uint32), ("ncls", c_uint32),
                ("lds_bytes", c_uint32), ("pipes", c_uint32), ("small_finds", c_uint32), ("big_nic_steps_max", c_uint32), ("batch_finds", c_uint32)]


_SIGS = {
    "nhdfit_abi_version": (c_int, []),
    "nhdfit_device_count": (c_int, []),
    "nhdfit_create": (c_int, [c_int, POINTER(c_void_p)]),
    "nhdfit_destroy": (None, [c_void_p]),
    "nhdfit_last_error": (c_char_p, [c_void_p]),
    "nhdfit_set_dictionary": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p, c_uint32, c_void_p, c_uint32, c_void_p, c_uint32, c_void_p, c_void_p, c_uint32,
                                      c_void_p, c_uint32]),
    "nhdfit_reserve_nodes": (c_int, [c_void_p, c_uint32, c_uint64]),
    "nhdfit_upload_nodes": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nhdfit_set_node_count": (c_int, [c_void_p, c_uint32]),
    "nhdfit_upload_origin": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p]),
    "nhdfit_apply_deltas": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p]),
    "nhdfit_find": (c_int, [c_void_p, c_void_p, c_uint32, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nhdfit_find_sequential": (c_int, [c_void_p, c_void_p, c_uint32, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nhdfit_schedule_batch": (c_int, [c_void_p, c_void_p, c_uint32, c_double, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      POINTER(c_uint32)]),
    "nhdfit_commit": (c_int, [c_void_p, c_uint32, c_void_p, c_void_p, c_double, c_void_p]),
    "nhdfit_wide_upload": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p, c_uint32]),
    "nhdfit_wide_share_upload": (c_int, [c_void_p, c_void_p, c_uint32]),
    "nhdfit_wide_share_download": (c_int, [c_void_p, c_void_p, c_uint32, POINTER(c_uint32)]),
    "nhdfit_wide_count": (c_int, [c_void_p, POINTER(c_uint32)]),
    "nhdfit_wide_download": (c_int, [c_void_p, c_void_p, c_uint32, POINTER(c_uint32)]),
    "nhdfit_wide_commit": (c_int, [c_void_p, c_uint32, c_void_p, c_void_p, c_double, c_void_p]),
    "nhdfit_wide_placements": (c_int, [c_void_p, c_void_p, c_uint32, POINTER(c_uint32)]),
    "nhdfit_big_find": (c_int, [c_void_p, c_void_p, c_uint32, c_double, c_void_p, c_void_p, c_void_p]),
    "nhdfit_big_commit": (c_int, [c_void_p, c_uint32, c_void_p, c_void_p, c_double, c_void_p]),
    "nhdfit_download_nodes": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nhdfit_stage_requests": (c_int, [c_void_p, c_void_p, c_uint32]),
    "nhdfit_enqueue_step": (c_int, [c_void_p, c_double]),
    "nhdfit_sync": (c_int, [c_void_p]),
    "nhdfit_fetch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "nhdfit_comm_unique_id": (c_int, [c_void_p]),
    "nhdfit_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "nhdfit_comm_destroy": (c_int, [c_void_p]),
    "nhdfit_comm_sendrecv": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p, ctypes.c_size_t, c_int]),
    "nhdfit_comm_allreduce_sum_u8": (c_int, [c_void_p, c_void_p, ctypes.c_size_t]),
    "nhdfit_comm_rank": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "nhdfit_set_outputs": (c_int, [c_void_p, c_int, c_int]),
    "nhdfit_get_stats": (c_int, [c_void_p, POINTER(Stats)]),
    "nhdfit_reset_stats": (c_int, [c_void_p]),
    "nhdfit_group_create": (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    "nhdfit_group_destroy": (None, [c_void_p]),
    "nhdfit_group_size": (c_int, [c_void_p]),
    "nhdfit_group_ctx": (c_void_p, [c_void_p, c_int]),
    "nhdfit_group_last_error": (c_char_p, [c_void_p]),
    "nhdfit_group_find": (c_int, [c_void_p, c_void_p, c_uint32, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nhdfit_digest_triad_config": (c_int, [c_char_p, ctypes.c_size_t, c_void_p, c_char_p, ctypes.c_size_t]),
    "nhdfit_digest_triad_config_big": (c_int, [c_char_p, ctypes.c_size_t, c_void_p, c_char_p, ctypes.c_size_t]),
    "nhdfit_digest_triad_configs": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p, c_void_p]),
}

_lib = None


def declared_symbols():
    """Function names declared in include/nhdfit.h (used by the ABI test)."""
    with open(HEADER) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(nhdfit_[a-z_0-9]+)\s*\(", text)))


def load():
    """Load the shared library.  Builds it with hipcc when it is missing (build container);
    raises if that is impossible - the product has no CPU implementation to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build
    if not os.path.exists(LIB_PATH):
        try:
            build.build_lib()
        except Exception as e:  # noqa: BLE001
            raise NhdFitError(-2, f"{LIB_PATH} is missing and could not be built with hipcc: {e}") from e
    elif build.stale() and os.path.exists(build.hipcc()):      # a source is newer than the library (development tree)
        try:
            build.build_lib()
        except Exception:  # noqa: BLE001 - keep the library that is there; its ABI version is checked below
            pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.nhdfit_abi_version.restype = c_int
    got = lib.nhdfit_abi_version()
    if got != ABI_VERSION:                                      # record layouts differ: never talk to it
        raise NhdFitError(-5, f"{LIB_PATH} implements ABI {got}, this binding expects {ABI_VERSION}: rebuild it (python -m nhd_amd.build --force)")
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx, rc):
    if rc != 0:
        msg = load().nhdfit_last_error(ctx)
        raise NhdFitError(rc, msg.decode() if msg else "?")
