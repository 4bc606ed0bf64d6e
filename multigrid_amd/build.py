"""Builds libmg_hip.so in-tree with hipcc for gfx950.

-ffp-contract=off is part of the numerical contract: the kernels must evaluate
the reference's expressions without FMA contraction so that results match the
scalar CPU evaluation bit for bit.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmg_hip.so")
SOURCES = ["mg_kernels.hip", "mg_sweep.hip", "mg_tile.hip", "mg_resrest.hip", "mg_ef.hip", "mg_api.hip", "mg_dist.hip"]
HEADERS = [os.path.join(CSRC, "mg_kernels.h"), os.path.join(CSRC, "mg_internal.h"),
           os.path.join(HERE, "..", "include", "mg_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-pthread"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmg_hip.so cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    build(force="-f" in sys.argv, verbose=True)
    print(LIB)
