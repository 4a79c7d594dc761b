"""Host build of the kernels' shared arithmetic (TEST INFRASTRUCTURE, see host_harness.cpp)."""
import ctypes
import os
import subprocess

import numpy as np

from nhd_amd import pack

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_harness.cpp")
SO = os.path.join(HERE, "_host_harness.so")
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    deps = [SRC] + [os.path.join(HERE, "..", "..", "nhd_amd", "csrc", f) for f in ("fit_core.h", "winner_map.h", "seq_core.h", "set_states.h")] + \
           [os.path.join(HERE, "..", "..", "include", "nhdfit.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", SO])
    _lib = ctypes.CDLL(SO)
    _lib.hh_tuple_hash.restype = ctypes.c_uint64
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def find(packer: pack.Packer, table: pack.NodeTable, reqs: np.ndarray, now: float, cand=None, global_base=0,
         want_bitmap=True, want_map=True, force_generic=False):
    L = lib()
    caps, sig_off, pool_off, glimit, cc, ncls, nsig, npools, ncc = packer.dictionary_arrays()
    fgmax = packer.max_gpus_per_numa
    gs = packer.group_set_array()
    n, P = table.n, len(reqs)
    chunks = (n + 63) // 64
    score = np.zeros(P, np.uint64)
    bitmap = np.zeros((chunks, P), np.uint64) if want_bitmap else None
    maps = np.zeros(P, pack.MAPPING) if want_map else None
    reqs = np.ascontiguousarray(reqs)
    L.hh_find.restype = ctypes.c_int
    bad = L.hh_find(_p(table.p0), _p(table.p1), _p(table.p2), _p(table.p3), _p(table.p4), _p(table.detail),
              ctypes.c_uint32(n), ctypes.c_uint64(global_base), _p(reqs), ctypes.c_uint32(P), ctypes.c_double(now),
              ctypes.c_uint32(packer.max_cores_per_numa), ctypes.c_uint32(fgmax),
              _p(gs), ctypes.c_uint32(len(packer.group_sets)), _p(caps), ctypes.c_uint32(ncls), _p(sig_off), ctypes.c_uint32(nsig), _p(pool_off), _p(glimit), _p(cc),
              _p(cand) if cand is not None else None, _p(score), _p(bitmap) if want_bitmap else None,
              _p(maps) if want_map else None, ctypes.c_int(int(force_generic)))
    assert bad == 0, f"{bad} (node, tile) verdicts differ between the hot and the cold table section / the two forms of IsBusy"
    return score, bitmap, maps


def resolve(packer, table, reqs, now, global_base=0, cand=None):
    """Mode B on the host build: snapshot pass + sequential resolver.  Returns (node index or -1, maps, status)."""
    L = lib()
    score, bitmap, maps = find(packer, table, reqs, now, global_base=global_base, cand=cand)
    caps = np.asarray(packer.caps, dtype="<f8")
    P = len(reqs)
    node = np.zeros(P, np.int64)
    out_maps = np.zeros(P, pack.MAPPING)
    status = np.zeros(P, np.int32)
    reqs = np.ascontiguousarray(reqs)
    L.hh_resolve(_p(table.p0), _p(table.p1), _p(table.p2), _p(table.p3), _p(table.p4), _p(table.detail),
                 ctypes.c_uint32(table.n), ctypes.c_uint64(global_base), _p(reqs), ctypes.c_uint32(P), ctypes.c_double(now),
                 _p(caps), _p(score), _p(bitmap), _p(maps), _p(node), _p(out_maps), _p(status))
    return node, out_maps, status


class HarnessEngine:
    """Engine-compatible front-end of the host build (TEST ONLY): lets HipMatcher's host logic (packing,
    dirty tracking, candidate masks, result decoding) and the sharding helpers run on CPU."""

    def __init__(self, device=0):
        self.device = device
        self.n = 0
        self.global_base = 0
        self.packer = None
        self.table = None

    def close(self):
        pass

    def set_dictionary(self, packer):
        self.packer = packer

    def reset_nodes(self):
        self.n = 0
        self.table = None

    def upload(self, table, global_base=0, first=0, capacity=None):
        if first == 0 and (self.table is None or table.n >= self.n):
            self.table = pack.NodeTable(list(table.names), *[np.array(getattr(table, f)) for f in
                                                            ("p0", "p1", "p2", "p3", "p4", "detail")])
        else:
            for f in ("p0", "p1", "p2", "p3", "p4", "detail"):
                getattr(self.table, f)[first:first + table.n] = getattr(table, f)
        self.n = self.table.n
        self.global_base = global_base

    def find(self, reqs, now, cand=None, want_bitmap=True, want_map=True):
        return find(self.packer, self.table, reqs, now, cand=cand, global_base=self.global_base,
                    want_bitmap=want_bitmap, want_map=want_map)

    def find_sequential(self, reqs, now, cand=None):
        return resolve(self.packer, self.table, reqs, now, global_base=self.global_base, cand=cand)
