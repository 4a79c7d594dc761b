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
    deps = [SRC] + [os.path.join(HERE, "..", "..", "nhd_amd", "csrc", f) for f in ("fit_core.h", "winner_map.h", "seq_core.h", "set_states.h", "commit_core.h", "wide_core.h", "dict_stream.h")] + \
           [os.path.join(HERE, "..", "..", "include", "nhdfit.h")]
    other = os.environ.get("NHD_HOST_HARNESS_SO")          # another build of the same file (tools/sanitize_host_twin.sh: ASan + UBSan)
    if other:
        _lib = ctypes.CDLL(other)
        _lib.hh_tuple_hash.restype = ctypes.c_uint64
        return _lib
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        tmp = f"{SO}.{os.getpid()}.tmp"                    # (several ranks of a gloo test may find it stale at once: each builds its own, the rename is atomic)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", tmp])
        os.replace(tmp, SO)
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
    assert bad == 0, f"{bad} verdicts differ between the hot and the cold table section / the six-fetch and the pair form of the sweep / the two forms of IsBusy / the pool-by-pool and the by-pool-type form of a signature's reach family"
    return score, bitmap, maps


def typed_stream(packer: pack.Packer):
    """(words, pool types) of the dictionary's stream by pool type as nhdfit_set_dictionary builds it (dict_stream.h); 0 words: not in use."""
    L = lib()
    caps, sig_off, pool_off, glimit, cc, ncls, nsig, npools, ncc = packer.dictionary_arrays()
    nt = ctypes.c_uint32(0)
    L.hh_typed_stream.restype = ctypes.c_int
    words = L.hh_typed_stream(_p(sig_off), ctypes.c_uint32(nsig), _p(pool_off), _p(glimit), _p(cc), ctypes.byref(nt))
    return int(words), int(nt.value)


def find_lone(packer: pack.Packer, table: pack.NodeTable, reqs: np.ndarray, now: float, cand=None, global_base=0):
    """The lone-pod form of the find on the host build (fit_core.h lone_pod_fits), every pod on its own: scores, chunk-major
    bitmap and the NIC-feasible assignment bits of each winner."""
    L = lib()
    caps, sig_off, pool_off, glimit, cc, ncls, nsig, npools, ncc = packer.dictionary_arrays()
    gs = packer.group_set_array()
    n, P = table.n, len(reqs)
    score = np.zeros(P, np.uint64)
    bitmap = np.zeros(((n + 63) // 64, P), np.uint64)
    bits = np.zeros(P, np.uint32)
    reqs = np.ascontiguousarray(reqs)
    L.hh_find_lone.restype = ctypes.c_int
    rc = L.hh_find_lone(_p(table.p0), _p(table.p1), _p(table.p2), _p(table.p3), _p(table.p4), ctypes.c_uint32(n), ctypes.c_uint64(global_base),
                        _p(reqs), ctypes.c_uint32(P), ctypes.c_double(now), ctypes.c_uint32(packer.max_cores_per_numa),
                        ctypes.c_uint32(packer.max_gpus_per_numa), _p(gs), ctypes.c_uint32(len(packer.group_sets)), _p(caps), ctypes.c_uint32(ncls),
                        _p(sig_off), ctypes.c_uint32(nsig), _p(pool_off), _p(glimit), _p(cc), _p(cand) if cand is not None else None,
                        _p(score), _p(bitmap), _p(bits))
    assert rc == 0, f"{rc}: the dictionary stream does not fit / winners whose NIC-feasible assignment bits differ from the table form's"
    return score, bitmap, bits


def _dict_args(packer):
    caps, sig_off, pool_off, glimit, cc, ncls, nsig, npools, ncc = packer.dictionary_arrays()
    return caps, sig_off, pool_off, glimit, cc, ncls, nsig


def schedule(packer, table, reqs, now, global_base=0, cand=None, apply=False, close=False):
    """Mode B on the host build: snapshot pass + sequential commit on host copies of the planes.
    Returns (node index or -1, maps, placements, status, n_done); apply=True writes the committed state into `table`."""
    L = lib()
    if close:
        packer.close_signatures()
    score, bitmap, _ = find(packer, table, reqs, now, global_base=global_base, cand=cand, want_map=False)
    caps, sig_off, pool_off, glimit, cc, ncls, nsig = _dict_args(packer)
    gs = packer.group_set_array()
    P = len(reqs)
    node = np.zeros(P, np.int64)
    maps = np.zeros(P, pack.MAPPING)
    places = np.zeros(P, pack.PLACEMENT)
    status = np.zeros(P, np.int32)
    reqs = np.ascontiguousarray(reqs)
    planes = [np.array(getattr(table, f)) for f in ("p0", "p1", "p2", "p3", "p4", "detail")]
    L.hh_schedule.restype = ctypes.c_uint32
    done = L.hh_schedule(*[_p(x) for x in planes], ctypes.c_uint32(table.n), ctypes.c_uint64(global_base), _p(reqs), ctypes.c_uint32(P),
                         ctypes.c_double(now), ctypes.c_uint32(packer.max_cores_per_numa), ctypes.c_uint32(packer.max_gpus_per_numa),
                         _p(gs), ctypes.c_uint32(len(packer.group_sets)), _p(caps), ctypes.c_uint32(ncls), _p(sig_off), ctypes.c_uint32(nsig),
                         _p(pool_off), _p(glimit), _p(cc), _p(score), _p(bitmap), _p(node), _p(maps), _p(places), _p(status))
    if apply:
        for f, arr in zip(("p0", "p1", "p2", "p3", "p4", "detail"), planes):
            getattr(table, f)[...] = arr
    return node, maps, places, status, int(done)


def resolve(packer, table, reqs, now, global_base=0, cand=None):
    """Mode B, mirror untouched: (node index or -1, maps, status) - the nhdfit_find_sequential contract."""
    node, maps, _, status, done = schedule(packer, table, reqs, now, global_base=global_base, cand=cand, close=True)
    assert done == len(reqs), "a commit left a node in a NIC state the dictionary has no signature for"
    return node, maps, status


def commit(packer, table, i, req, mapping, busy_time):
    """The commit step alone on node i of `table` (modified in place).  Returns (status, placement)."""
    L = lib()
    _, sig_off, pool_off, glimit, cc, _, nsig = _dict_args(packer)
    out = np.zeros((), pack.PLACEMENT)
    req = np.ascontiguousarray(req)
    mapping = np.ascontiguousarray(mapping)
    rows = [np.ascontiguousarray(getattr(table, f)[i:i + 1]) for f in ("p0", "p1", "p2", "p3", "p4", "detail")]
    L.hh_commit.restype = ctypes.c_int
    rc = L.hh_commit(*[_p(x) for x in rows], _p(req), _p(mapping), ctypes.c_double(busy_time), _p(sig_off), ctypes.c_uint32(nsig),
                     _p(pool_off), _p(glimit), _p(cc), _p(out))
    for f, r in zip(("p0", "p1", "p2", "p3", "p4", "detail"), rows):
        getattr(table, f)[i] = r[0]
    return int(rc), out


# ---- the wavefront form of the commit step (device source text of seq2_kernel.h) under a 64-thread wavefront emulation ---------
WAVE_SRC = os.path.join(HERE, "wave_emul.cpp")
WAVE_SO = os.path.join(HERE, "_wave_emul.so")
WAVE_INC = os.path.join(HERE, "_wave_commit_block.inc")
_WAVE_FROM, _WAVE_TO = "// ---- the commit step with the wavefront's lanes", "// ---- k_decide: speculate, then retire in order"
WAVE_MAP_INC = os.path.join(HERE, "_wave_map_block.inc")
_WAVE_MAP_FROM, _WAVE_MAP_TO = "// ---- wave-cooperative forms of the mapping arithmetic", "// One block walks the batch in the caller's order"
WAVE_BIG_INC = os.path.join(HERE, "_wave_bigmap_block.inc")
_WAVE_BIG_FROM, _WAVE_BIG_TO = "// ---- the winner's mapping with the wavefront's lanes", "// ---- the kernels of the general path for requests"
_wave = None


def wave_lib():
    """Builds tests/harness/wave_emul.cpp around the commit section of nhd_amd/csrc/seq2_kernel.h (cut out unmodified)."""
    global _wave
    if _wave is not None:
        return _wave
    kernel = os.path.join(HERE, "..", "..", "nhd_amd", "csrc", "seq2_kernel.h")
    kernel1 = os.path.join(HERE, "..", "..", "nhd_amd", "csrc", "seq_kernel.h")
    kernel2 = os.path.join(HERE, "..", "..", "nhd_amd", "csrc", "big_kernel.h")
    deps = [WAVE_SRC, kernel, kernel1, kernel2] + [os.path.join(HERE, "..", "..", "nhd_amd", "csrc", f) for f in ("fit_core.h", "seq_core.h", "commit_core.h", "winner_map.h", "set_states.h", "wide_core.h")] + \
           [os.path.join(HERE, "..", "..", "include", "nhdfit.h")]
    if not os.path.exists(WAVE_SO) or any(os.path.getmtime(d) > os.path.getmtime(WAVE_SO) for d in deps):
        lines = open(kernel).read().split("\n")
        a = next(i for i, ln in enumerate(lines) if ln.startswith(_WAVE_FROM))
        b = next(i for i, ln in enumerate(lines) if ln.startswith(_WAVE_TO))
        assert a < b and any("commit_node_wave" in ln for ln in lines[a:b])
        with open(WAVE_INC, "w") as f:
            f.write("\n".join(lines[a:b]) + "\n")
        lines = open(kernel1).read().split("\n")
        a = next(i for i, ln in enumerate(lines) if ln.startswith(_WAVE_MAP_FROM))
        b = next(i for i, ln in enumerate(lines) if ln.startswith(_WAVE_MAP_TO))
        assert a < b and any("map_on_state_wave" in ln for ln in lines[a:b])
        with open(WAVE_MAP_INC, "w") as f:
            f.write("\n".join(lines[a:b]) + "\n")
        lines = open(kernel2).read().split("\n")
        a = next(i for i, ln in enumerate(lines) if ln.startswith(_WAVE_BIG_FROM))
        b = next(i for i, ln in enumerate(lines) if ln.startswith(_WAVE_BIG_TO))
        assert a < b and any("wide_map_wave" in ln for ln in lines[a:b])
        with open(WAVE_BIG_INC, "w") as f:
            f.write("\n".join(lines[a:b]) + "\n")
        tmp = f"{WAVE_SO}.{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O2", "-std=c++20", "-ffp-contract=off", "-shared", "-fPIC", "-pthread", WAVE_SRC, "-o", tmp])
        os.replace(tmp, WAVE_SO)
    _wave = ctypes.CDLL(WAVE_SO)
    return _wave


def wave_map_on_state(packer, table, i, req, tables=2, nic_bits=-1, form=1):
    """seq_core.h map_on_state (scalar) and seq_kernel.h map_on_state_wave (emulated lanes) for pod `req` on node i as it stands in
    `table`: (return code of we_map_on_state, ok bits, scalar mapping, wavefront mapping).  nic_bits >= 0: the NIC-feasible
    assignments to use (find_lone's) instead of the scalar NIC walk's."""
    L = wave_lib()
    caps = _dict_args(packer)[0]
    req = np.ascontiguousarray(req)
    rows = [np.ascontiguousarray(getattr(table, f)[i:i + 1]) for f in ("p0", "p1", "p2", "p3", "p4", "detail")]
    ms, mw = np.zeros((), pack.MAPPING), np.zeros((), pack.MAPPING)
    ok = ctypes.c_int(0)
    L.we_map_on_state.restype = ctypes.c_int
    rc = L.we_map_on_state(*[_p(x) for x in rows], _p(req), _p(caps), ctypes.c_int(tables), ctypes.c_int64(int(nic_bits)), _p(ms), _p(mw), ctypes.byref(ok), ctypes.c_int(form))
    return int(rc), ok.value, ms, mw


def wave_big_map(packer, table, v, big_req, wide=None, share=None):
    """A big request's mapping on node `v` of `table` (or on the wide record `wide`): wide_core.h wide_map (one thread) against
    big_kernel.h wide_map_wave (lane = tuple, emulated lanes).  (return code of we_big_map, (scalar rc, wave rc), scalar mapping, wave mapping)."""
    L = wave_lib()
    caps = _dict_args(packer)[0]
    req = np.ascontiguousarray(big_req)
    rows = [np.ascontiguousarray(getattr(table, f)[v:v + 1]) for f in ("p0", "p1", "p2", "p3", "p4", "detail")]
    ms, mw = np.zeros((), pack.BIG_MAPPING), np.zeros((), pack.BIG_MAPPING)
    rcs = (ctypes.c_int * 2)(0, 0)
    w = None if wide is None else np.ascontiguousarray(wide, dtype=pack.WIDE).reshape(-1)[:1]
    sh_arr, sh_ptr = _share_arg(share, 1) if share is not None else (None, None)
    L.we_big_map.restype = ctypes.c_int
    rc = L.we_big_map(*[_p(x) for x in rows], None if w is None else _p(w), _p(req), _p(caps), sh_ptr, _p(ms), _p(mw), rcs)
    return int(rc), (int(rcs[0]), int(rcs[1])), ms, mw


def wave_commit(packer, table, i, req, mapping, busy_time, form=1):
    """commit() above through the WAVEFRONT form of the commit step (commit_node_wave, emulated lanes); `table` modified in place.
    form=2: the two-stage form of k_decide's speculators for a pod without GPUs (commit_summary_wave, then commit_picks_wave)."""
    L = wave_lib()
    _, sig_off, pool_off, glimit, cc, ncls, nsig = _dict_args(packer)
    out = np.zeros((), pack.PLACEMENT)
    req = np.ascontiguousarray(req)
    mapping = np.ascontiguousarray(mapping)
    rows = [np.ascontiguousarray(getattr(table, f)[i:i + 1]) for f in ("p0", "p1", "p2", "p3", "p4", "detail")]
    L.we_commit.restype = ctypes.c_int
    rc = L.we_commit(*[_p(x) for x in rows], _p(req), _p(mapping), ctypes.c_double(busy_time), _p(sig_off), ctypes.c_uint32(nsig),
                     _p(pool_off), _p(glimit), _p(cc), ctypes.c_uint32(ncls), _p(out), ctypes.c_int(form))
    for f, r in zip(("p0", "p1", "p2", "p3", "p4", "detail"), rows):
        getattr(table, f)[i] = r[0]
    return int(rc), out


def apply_deltas(packer, table, deltas):
    """K3 on the host build: `deltas` (pack.DELTA, local node indices) applied in order to `table` (in place)."""
    L = lib()
    _, sig_off, pool_off, glimit, cc, _, nsig = _dict_args(packer)
    deltas = np.ascontiguousarray(deltas, dtype=pack.DELTA).reshape(-1)
    status = np.zeros(len(deltas), np.uint8)
    planes = [np.ascontiguousarray(getattr(table, f)) for f in ("p0", "p1", "p2", "p3", "p4", "detail", "origin")]
    L.hh_apply_deltas.restype = ctypes.c_int
    rc = L.hh_apply_deltas(*[_p(x) for x in planes], ctypes.c_uint32(table.n), _p(deltas), ctypes.c_uint32(len(deltas)),
                           _p(sig_off), ctypes.c_uint32(nsig), _p(pool_off), _p(glimit), _p(cc), _p(status))
    assert rc == 0
    for f, arr in zip(("p0", "p1", "p2", "p3", "p4", "detail", "origin"), planes):
        getattr(table, f)[...] = arr
    return status


def _share_arg(share, n=None):
    """nhdfit_wide_share records (ENABLE_SHARING) as a pointer argument, or None."""
    if share is None:
        return None, None
    arr = np.ascontiguousarray(share, dtype=pack.WIDE_SHARE).reshape(-1)
    assert n is None or len(arr) == n
    return arr, _p(arr)


def wide_eval(wide: np.ndarray, reqs: np.ndarray, now: float, packer, cand=None, global_base=0, score=None, share=None):
    """k_wide_eval on the host build: (fits [n_wide][P] bytes, score max-merged)."""
    L = lib()
    wide = np.ascontiguousarray(wide, dtype=pack.WIDE)
    reqs = np.ascontiguousarray(reqs)
    P = len(reqs)
    fits = np.zeros((len(wide), P), np.uint8)
    score = np.zeros(P, np.uint64) if score is None else np.ascontiguousarray(score, dtype=np.uint64)
    caps = np.zeros(pack.MAX_CLASSES, "<f8")
    caps[:len(packer.caps)] = packer.caps
    if len(wide):
        sh, shp = _share_arg(share, len(wide))
        L.hh_wide_eval(_p(wide), ctypes.c_uint32(len(wide)), _p(reqs), ctypes.c_uint32(P), ctypes.c_double(now), _p(caps),
                       _p(cand) if cand is not None else None, ctypes.c_uint64(global_base), _p(fits), _p(score), shp)
    return fits, score


def wide_map(rec, req, packer, share=None):
    L = lib()
    caps = np.zeros(pack.MAX_CLASSES, "<f8")
    caps[:len(packer.caps)] = packer.caps
    out = np.zeros((), pack.MAPPING)
    rec = np.ascontiguousarray(rec, dtype=pack.WIDE).reshape(1)
    req = np.ascontiguousarray(req).reshape(1)
    L.hh_wide_map.restype = ctypes.c_int
    sh, shp = _share_arg(share, 1)
    rc = L.hh_wide_map(_p(rec), _p(req), _p(caps), _p(out), shp)
    assert rc >= 0, "a set of the general model outgrew its table"
    return out


def wide_commit(rec, req, mapping, busy_time, share=None):
    """wide_commit on the host build: (status, placement, record after the commit); `share` (one nhdfit_wide_share record) is
    updated in place."""
    L = lib()
    rec = np.array(rec, dtype=pack.WIDE).reshape(1)
    out = np.zeros((), pack.WIDE_PLACEMENT)
    req = np.ascontiguousarray(req).reshape(1)
    mapping = np.ascontiguousarray(mapping).reshape(1)
    L.hh_wide_commit.restype = ctypes.c_int
    st = L.hh_wide_commit(_p(rec), _p(req), _p(mapping), ctypes.c_double(busy_time), _p(out), _p(share) if share is not None else None)
    return int(st), out, rec[0]


def _caps(packer):
    caps = np.zeros(pack.MAX_CLASSES, "<f8")
    caps[:len(packer.caps)] = packer.caps
    return caps


def big_eval(packer, table, wide: np.ndarray, reqs: np.ndarray, now: float, cand=None, global_base=0, budget=0, share=None):
    """k_big_eval on the host build: big requests (5..8 groups) against every node - the planes through wide_view, the wide
    records as they are.  Returns (fits [n][P] by node index, scores, budget_exhausted)."""
    L = lib()
    reqs = np.ascontiguousarray(reqs, dtype=pack.BIG_REQ)
    wide = np.ascontiguousarray(wide, dtype=pack.WIDE)
    P, n = len(reqs), table.n
    fits = np.zeros((n, P), np.uint8)
    score = np.zeros(P, np.uint64)
    flags = np.zeros(4, np.uint32)
    caps = _caps(packer)
    planes = [np.ascontiguousarray(getattr(table, f)) for f in ("p0", "p1", "p2", "p3", "p4", "detail")]
    L.hh_big_eval(*[_p(x) for x in planes], ctypes.c_uint32(n), _p(wide) if len(wide) else None, ctypes.c_uint32(len(wide)), _p(reqs), ctypes.c_uint32(P),
                  ctypes.c_double(now), _p(caps), _p(cand) if cand is not None else None, ctypes.c_uint64(global_base), _p(fits), _p(score), _p(flags),
                  ctypes.c_uint32(budget), _share_arg(share, len(wide))[1] if share is not None else None)
    big_eval.last_steps = int(flags[2])                               # NIC search steps of the call (diagnostics)
    return fits, score, bool(flags[1])


def big_map(packer, table, v: int, wide_rec, req, share=None):
    """wide_map for a big request on node v of `table` (wide_rec: that node's wide record, or None for an ordinary node)."""
    L = lib()
    out = np.zeros((), pack.BIG_MAPPING)
    req = np.ascontiguousarray(req, dtype=pack.BIG_REQ).reshape(1)
    caps = _caps(packer)
    planes = [np.ascontiguousarray(getattr(table, f)) for f in ("p0", "p1", "p2", "p3", "p4", "detail")]
    rec = None if wide_rec is None else np.ascontiguousarray(wide_rec, dtype=pack.WIDE).reshape(1)
    L.hh_big_map.restype = ctypes.c_int
    sh, shp = _share_arg(share, 1)
    rc = L.hh_big_map(*[_p(x) for x in planes], ctypes.c_uint32(v), _p(rec) if rec is not None else None, _p(req), _p(caps), _p(out), shp)
    assert rc >= 0, "a set of the general model outgrew its table / the NIC search budget ran out"
    return out


def big_commit(packer, table, i, req, mapping, busy_time):
    """commit_node_t<big> on node i of `table` (an ordinary node; modified in place).  Returns (status, pack.BIG_PLACEMENT)."""
    L = lib()
    _, sig_off, pool_off, glimit, cc, _, nsig = _dict_args(packer)
    out = np.zeros((), pack.BIG_PLACEMENT)
    req = np.ascontiguousarray(req, dtype=pack.BIG_REQ).reshape(1)
    mapping = np.ascontiguousarray(mapping, dtype=pack.BIG_MAPPING).reshape(1)
    rows = [np.ascontiguousarray(getattr(table, f)[i:i + 1]) for f in ("p0", "p1", "p2", "p3", "p4", "detail")]
    L.hh_big_commit.restype = ctypes.c_int
    rc = L.hh_big_commit(*[_p(x) for x in rows], _p(req), _p(mapping), ctypes.c_double(busy_time), _p(sig_off), ctypes.c_uint32(nsig),
                         _p(pool_off), _p(glimit), _p(cc), _p(out))
    for f, r in zip(("p0", "p1", "p2", "p3", "p4", "detail"), rows):
        getattr(table, f)[i] = r[0]
    out["node"] = i
    return int(rc), out


def big_commit_wide(rec, req, mapping, busy_time, share=None):
    L = lib()
    rec = np.array(rec, dtype=pack.WIDE).reshape(1)
    out = np.zeros((), pack.BIG_PLACEMENT)
    req = np.ascontiguousarray(req, dtype=pack.BIG_REQ).reshape(1)
    mapping = np.ascontiguousarray(mapping, dtype=pack.BIG_MAPPING).reshape(1)
    L.hh_big_commit_wide.restype = ctypes.c_int
    st = L.hh_big_commit_wide(_p(rec), _p(req), _p(mapping), ctypes.c_double(busy_time), _p(out), _p(share) if share is not None else None)
    return int(st), out, rec[0]


class HarnessEngine:
    """Engine-compatible front-end of the host build (TEST ONLY): lets HipMatcher's host logic (packing,
    dirty tracking, candidate masks, result decoding) and the sharding helpers run on CPU."""

    def __init__(self, device=0):
        self.device = device
        self.n = 0
        self.global_base = 0
        self.packer = None
        self.table = None
        self.wide = {}                                  # local index -> nhdfit_wide_node (the general path's records)
        self.share = {}                                 # local index -> nhdfit_wide_share (ENABLE_SHARING: one per node, all nodes wide)
        self.last_wide_places = {}

    @property
    def n_wide(self):
        return len(self.wide)

    def _wide_records(self):
        idx = sorted(self.wide)
        out = np.zeros(len(idx), pack.WIDE)
        for k, i in enumerate(idx):
            out[k] = self.wide[i]
            out[k]["index"] = i
        return out

    def _share_records(self):
        """The speed_used records in the order of _wide_records(), or None (shipped arithmetic)."""
        if not self.share:
            return None
        assert sorted(self.share) == sorted(self.wide) == list(range(self.n)), "ENABLE_SHARING: every node is a wide record with its share record"
        return np.array([self.share[i] for i in sorted(self.wide)], dtype=pack.WIDE_SHARE)

    def wide_share_download(self):
        sh = self._share_records()
        return np.zeros(0, pack.WIDE_SHARE) if sh is None else sh

    def close(self):
        pass

    def set_dictionary(self, packer):
        self.packer = packer

    def forget_dictionary(self):
        pass

    def reset_nodes(self):
        self.n = 0
        self.table = None
        self.wide = {}
        self.share = {}

    def upload(self, table, global_base=0, first=0, capacity=None):
        if first == 0 and (self.table is None or table.n >= self.n):
            self.table = pack.NodeTable(list(table.names), *[np.array(getattr(table, f)) for f in
                                                            ("p0", "p1", "p2", "p3", "p4", "detail")],
                                        np.zeros(table.n, pack.ORIGIN) if table.origin is None else np.array(table.origin))
        else:
            for f in ("p0", "p1", "p2", "p3", "p4", "detail") + (("origin",) if table.origin is not None else ()):
                getattr(self.table, f)[first:first + table.n] = getattr(table, f)
        self.n = self.table.n
        self.global_base = global_base
        for i in [i for i in self.wide if first <= i < first + table.n]:      # nhdfit_wide_upload: the range's records are replaced
            del self.wide[i]
        for i, rec in (table.wide or {}).items():
            self.wide[first + i] = np.array(rec, dtype=pack.WIDE)
        for i in [i for i in self.share if first <= i < first + table.n]:
            del self.share[i]
        for i, rec in (table.share or {}).items():
            self.share[first + i] = np.array(rec, dtype=pack.WIDE_SHARE)

    def find(self, reqs, now, cand=None, want_bitmap=True, want_map=True):
        score, bitmap, maps = find(self.packer, self.table, reqs, now, cand=cand, global_base=self.global_base,
                                   want_bitmap=want_bitmap, want_map=want_map)
        if self.wide:                                   # the general pass, merged as nhdfit.hip merges it (launch_step / nhdfit_fetch)
            recs = self._wide_records()
            shares = self._share_records()
            fits, score = wide_eval(recs, reqs, now, self.packer, cand=cand, global_base=self.global_base, score=score, share=shares)
            if want_bitmap:
                for k, rec in enumerate(recs):
                    i = int(rec["index"])
                    bitmap[i >> 6, :] |= fits[k].astype(np.uint64) << np.uint64(i & 63)
            if want_map:
                idx = np.where(score == 0, -1, (0x7FFFFFFFFFFFFFFF - (score & np.uint64(0x7FFFFFFFFFFFFFFF))).astype(np.int64) - self.global_base)
                for p in np.flatnonzero(score != 0):
                    if int(idx[p]) in self.wide:
                        k = sorted(self.wide).index(int(idx[p]))
                        maps[p] = wide_map(recs[k], reqs[p], self.packer, share=None if shares is None else shares[k:k + 1])
        return score, bitmap, maps

    def find_sequential(self, reqs, now, cand=None):
        return resolve(self.packer, self.table, reqs, now, global_base=self.global_base, cand=cand)

    def schedule_batch(self, reqs, now, packer, cand=None, apply=True):
        self.last_wide_places = {}
        if self.wide:
            return self._schedule_general(reqs, now, packer, cand, apply)
        node, maps, places, status, done = schedule(packer, self.table, reqs, now, global_base=self.global_base, cand=cand, apply=apply)
        assert done == len(reqs)
        self.n_done = done
        return node, maps, places, status

    def _schedule_general(self, reqs, now, packer, cand, apply):
        """schedule_batch_general of nhdfit.hip: FindNode for pod k, the commit step on whichever mirror holds the winner, pod k + 1."""
        P = len(reqs)
        node = np.full(P, -1, np.int64)
        maps = np.zeros(P, pack.MAPPING)
        places = np.zeros(P, pack.PLACEMENT)
        status = np.zeros(P, np.int32)
        saved_table = None if apply else pack.NodeTable(list(self.table.names), *[np.array(getattr(self.table, f)) for f in
                                                                                  ("p0", "p1", "p2", "p3", "p4", "detail")], np.array(self.table.origin))
        saved_wide = None if apply else {i: np.array(r) for i, r in self.wide.items()}
        saved_share = None if apply else {i: np.array(r) for i, r in self.share.items()}
        for i in range(P):
            r = reqs[i:i + 1]
            if not (int(r[0]["map_type"]) in (1, 2) and 1 <= int(r[0]["n_groups"]) <= pack.MAX_GROUPS):
                continue
            sc, _, mp = self.find(r, now, cand=cand, want_bitmap=False, want_map=True)
            if not sc[0]:
                continue
            v = int(0x7FFFFFFFFFFFFFFF - (int(sc[0]) & 0x7FFFFFFFFFFFFFFF)) - self.global_base
            node[i] = v + self.global_base
            maps[i] = mp[0]
            if v in self.wide:
                st, wp, rec = wide_commit(self.wide[v], r[0], mp[0], now, share=self.share.get(v))
                self.wide[v] = rec
                wp["pod"], wp["node"] = i, v
                self.last_wide_places[i] = wp
                places[i]["status"] = pack.COMMIT_WIDE
                status[i] = st if st == pack.COMMIT_WOULD_RAISE else 0
            else:
                st, pl = commit(packer, self.table, v, r[0], mp[0], now)
                places[i] = pl
                status[i] = st
        if not apply:
            for f in ("p0", "p1", "p2", "p3", "p4", "detail", "origin"):
                getattr(self.table, f)[...] = getattr(saved_table, f)
            self.wide = saved_wide
            self.share = saved_share
        self.n_done = P
        return node, maps, places, status

    def commit(self, node, req, mapping, busy_time):
        return commit(self.packer, self.table, node, req, mapping, busy_time)[1]

    def wide_commit(self, node, req, mapping, busy_time):
        st, wp, rec = wide_commit(self.wide[node], req, mapping, busy_time, share=self.share.get(node))
        self.wide[node] = rec
        wp["node"] = node
        return wp

    def big_find(self, reqs, now, cand=None, want_map=True):
        """nhdfit_big_find as nhdfit.hip does it: k_big_eval over planes + wide records, k_big_map for the winners this mirror holds."""
        reqs = np.ascontiguousarray(reqs, dtype=pack.BIG_REQ)
        recs = self._wide_records()
        shares = self._share_records()
        _, score, exhausted = big_eval(self.packer, self.table, recs, reqs, now, cand=cand, global_base=self.global_base,
                                       budget=getattr(self, "nic_budget", 0), share=shares)
        if exhausted:
            from nhd_amd._lib import NhdFitError
            raise NhdFitError(-6, "a big request's NIC stage ran out of search budget on some node")
        maps = np.zeros(len(reqs), pack.BIG_MAPPING)
        if want_map:
            order = sorted(self.wide)
            for p in np.flatnonzero(score != 0):
                v = int(0x7FFFFFFFFFFFFFFF - (int(score[p]) & 0x7FFFFFFFFFFFFFFF)) - self.global_base
                if 0 <= v < self.n:
                    k = order.index(v) if v in self.wide else -1
                    maps[p] = big_map(self.packer, self.table, v, recs[k] if k >= 0 else None, reqs[p],
                                      share=shares[k:k + 1] if shares is not None and k >= 0 else None)
        return score, (maps if want_map else None)

    def big_commit(self, node, req, mapping, busy_time):
        if node in self.wide:
            st, pl, rec = big_commit_wide(self.wide[node], req, mapping, busy_time, share=self.share.get(node))
            self.wide[node] = rec
            pl["node"] = node
            return pl
        return big_commit(self.packer, self.table, node, req, mapping, busy_time)[1]

    def apply_deltas(self, deltas):
        return apply_deltas(self.packer, self.table, deltas)

    def download(self, first=0, count=None):
        count = self.n - first if count is None else count
        t = self.table.slice(first, first + count)
        out = pack.NodeTable(list(t.names), *[np.array(getattr(t, f)) for f in ("p0", "p1", "p2", "p3", "p4", "detail")])   # a copy, as a device read-back is
        out.wide = {i - first: np.array(r) for i, r in self.wide.items() if first <= i < first + count}
        for q, r in out.wide.items():
            r["index"] = q + first
        return out
