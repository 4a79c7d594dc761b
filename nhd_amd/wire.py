"""Pod request digest straight from the wire format.

    req = wire.digest_config(text)            # numpy record of dtype pack.REQ, or None

replaces, for the matcher's purposes, the reference's

    top = TriadCfgParser(text, False).CfgToTopology(False)      # nhd/NHDScheduler.py:262-270
    ... FindNode(nodes, top) -> top.GetTotalCpusRequested() / GetTotalGpusRequested() / GetTotalNICsRequested()

without building the CfgTopology object graph: the libconfig text is read and walked by host C++ inside
libnhdfit.so (nhd_amd/csrc/wire_digest.cpp, `nhdfit_digest_triad_config`).  Return value mirrors the reference:
None where CfgToTopology returns None (the scheduler logs and skips the pod), `ConfigError` where the reference
would raise, `pack.UnsupportedNode` beyond the library's limits (same as Packer.digest)."""
import ctypes
from typing import Optional, Sequence

import numpy as np

from . import _lib, pack

WIRE_NONE, WIRE_RAISE, WIRE_LIMIT = 1, 2, 3


class ConfigError(ValueError):
    """The reference's parser would raise on this text (malformed libconfig, value of the wrong type)."""


def digest_config(text, pod_groups: Optional[Sequence[str]] = None, packer: Optional[pack.Packer] = None) -> Optional[np.ndarray]:
    """libconfig text (str or bytes) -> nhdfit_req record.  `pod_groups` (the pod's node-group annotation,
    nhd/K8SMgr.py:152-165) with `packer` makes the kernel apply InitialNodeFilter, exactly as Packer.digest does."""
    lib = _lib.load()
    raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    req = np.zeros((), pack.REQ)
    err = ctypes.create_string_buffer(256)
    rc = lib.nhdfit_digest_triad_config(raw, len(raw), req.ctypes.data_as(ctypes.c_void_p), err, len(err))
    if rc == WIRE_NONE:
        return None
    if rc == WIRE_RAISE:
        raise ConfigError(err.value.decode("utf-8", "replace"))
    if rc == WIRE_LIMIT:
        raise pack.UnsupportedNode(err.value.decode("utf-8", "replace"))
    if rc != 0:
        raise _lib.NhdFitError(rc, err.value.decode("utf-8", "replace"))
    if pod_groups is not None:
        if packer is None:
            raise ValueError("pod_groups needs the Packer that interns the group names")
        req["flags"] |= pack.RF_INITIAL_FILTER
        req["groups"] = packer.group_bits_known(pod_groups)
    return req


def digest_config_big(text) -> Optional[np.ndarray]:
    """libconfig text -> nhdfit_big_req (pack.BIG_REQ): a pod with up to eight processing groups, for the general path
    (HipMatcher.FindNodesFromConfigs sends the texts digest_configs turns away with WIRE_LIMIT here).  None / ConfigError /
    UnsupportedNode as digest_config."""
    lib = _lib.load()
    raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    req = np.zeros((), pack.BIG_REQ)
    err = ctypes.create_string_buffer(256)
    rc = lib.nhdfit_digest_triad_config_big(raw, len(raw), req.ctypes.data_as(ctypes.c_void_p), err, len(err))
    if rc == WIRE_NONE:
        return None
    if rc == WIRE_RAISE:
        raise ConfigError(err.value.decode("utf-8", "replace"))
    if rc == WIRE_LIMIT:
        raise pack.UnsupportedNode(err.value.decode("utf-8", "replace"))
    if rc != 0:
        raise _lib.NhdFitError(rc, err.value.decode("utf-8", "replace"))
    return req


def digest_configs(texts: Sequence) -> "tuple[np.ndarray, np.ndarray]":
    """Many texts in one library call: (requests [n] of dtype pack.REQ, codes [n] int32).  codes[i] is 0, WIRE_NONE,
    WIRE_RAISE or WIRE_LIMIT; requests of non-zero codes are all-zero records (= never matching)."""
    lib = _lib.load()
    raws = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in texts]
    n = len(raws)
    reqs = np.zeros(n, pack.REQ)
    codes = np.zeros(n, np.int32)
    if n:
        ptrs = (ctypes.c_char_p * n)(*raws)
        lens = (ctypes.c_size_t * n)(*[len(r) for r in raws])
        lib.nhdfit_digest_triad_configs(ptrs, lens, n, reqs.ctypes.data_as(ctypes.c_void_p), codes.ctypes.data_as(ctypes.c_void_p))
    return reqs, codes
