#!/usr/bin/env python3
"""CPU only: the wire digest (nhd_amd/csrc/wire_digest.cpp: the host C++ reader of a pod's libconfig text, row f3) compiled ALONE with
AddressSanitizer + UndefinedBehaviorSanitizer and fed mutated config texts - valid configs of tests/wire_gen.py with bytes replaced,
inserted, deleted, blocks repeated, the text truncated - through nhdfit_digest_triad_config / _big / _configs.  The reader takes text
that comes out of Kubernetes ConfigMaps: whatever it is handed it must answer with a code, never touch memory it does not own.

    python tools/fuzz_wire_sanitized.py <texts> [seed]        (re-executes itself with libasan preloaded)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = "/tmp/nhdfit_wire_san.so"


def main():
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    stdcxx = subprocess.check_output(["gcc", "-print-file-name=libstdc++.so.6"], text=True).strip()   # (loaded up front: ASan's __cxa_throw interceptor looks for the real one at start-up)
    preload = asan + ":" + stdcxx
    src = os.path.join(ROOT, "nhd_amd", "csrc", "wire_digest.cpp")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "nhdfit.h"))):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=undefined", src, "-o", SO])
    if os.environ.get("LD_PRELOAD", "") != preload:
        env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
        sys.exit(subprocess.call([sys.executable] + sys.argv, env=env))
    import numpy as np
    from nhd_amd import pack
    from tests import wire_gen
    lib = ctypes.CDLL(SO)
    n_texts = int(sys.argv[1])
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    req = np.zeros(1, pack.REQ)
    big = np.zeros(1, pack.BIG_REQ)
    err = ctypes.create_string_buffer(256)
    codes = {}
    alphabet = b'{}[]()=:;,"\\/#*\n\t .-+eExXL0123456789abcdefTRUEfalse@'
    for k in range(n_texts):
        text = wire_gen.make_config(int(rng.integers(1 << 30)), types_hi=int(rng.integers(2, 11)))
        if isinstance(text, tuple):
            text = text[0]
        b = bytearray(text.encode())
        for _ in range(int(rng.integers(0, 6))):
            if not b:
                break
            r = rng.random()
            i = int(rng.integers(len(b)))
            if r < 0.3:
                b[i] = alphabet[int(rng.integers(len(alphabet)))] if rng.random() < 0.8 else int(rng.integers(256))
            elif r < 0.5:
                b[i:i] = bytes(alphabet[int(x)] for x in rng.integers(len(alphabet), size=int(rng.integers(1, 5))))
            elif r < 0.7:
                del b[i:i + int(rng.integers(1, 9))]
            elif r < 0.85:
                j = min(len(b), i + int(rng.integers(1, 200)))
                b[i:i] = b[i:j] * int(rng.integers(1, 4))
            else:
                del b[i:]
        data = bytes(b)
        buf = ctypes.create_string_buffer(data, len(data))          # exactly len bytes: a read past the end is a heap overflow ASan sees
        rc = lib.nhdfit_digest_triad_config(buf, ctypes.c_size_t(len(data)), req.ctypes.data_as(ctypes.c_void_p), err, ctypes.c_size_t(256))
        rb = lib.nhdfit_digest_triad_config_big(buf, ctypes.c_size_t(len(data)), big.ctypes.data_as(ctypes.c_void_p), err, ctypes.c_size_t(256))
        codes[(rc, rb)] = codes.get((rc, rb), 0) + 1
        ptrs = (ctypes.c_char_p * 1)(ctypes.cast(buf, ctypes.c_char_p))
        lens = (ctypes.c_size_t * 1)(len(data))
        c1 = (ctypes.c_int32 * 1)()
        lib.nhdfit_digest_triad_configs(ptrs, lens, 1, req.ctypes.data_as(ctypes.c_void_p), c1)
        if c1[0] != rc:
            print("batch form disagrees", c1[0], rc)
            sys.exit(1)
    print("texts", n_texts, "codes (record, big record):", dict(sorted(codes.items())), "- no sanitizer report")


if __name__ == "__main__":
    main()
