"""Build libnhdfit.so (hipcc, gfx950 only) in-tree.  `python -m nhd_amd.build [--force]`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "nhdfit.hip")
WIRE = os.path.join(HERE, "csrc", "wire_digest.cpp")          # host-only translation unit (libconfig reader)
DEPS = [SRC, WIRE, os.path.join(ROOT, "include", "nhdfit.h")] + \
       sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".h"))   # every header nhdfit.hip may include
LIB = os.path.join(HERE, "libnhdfit.so")
TUNING_LIB = os.path.join(HERE, "libnhdfit_tuning.so")   # -DNHDFIT_TUNING: environment knobs + ablation switches, for tools/ only
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def stale() -> bool:
    return (not os.path.exists(LIB)) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)


def build_lib(force=False, verbose=False, extra=(), tuning=False):
    out = TUNING_LIB if tuning else LIB
    if not force and not tuning and not stale():
        return out
    cmd = [hipcc()] + FLAGS + (["-DNHDFIT_TUNING"] if tuning else []) + list(extra) + [SRC, WIRE, "-o", out, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True, tuning="--tuning" in sys.argv))
