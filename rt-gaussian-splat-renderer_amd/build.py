"""In-tree builds: the gfx950 HIP library and (test infrastructure) the C oracle.

    python rt-gaussian-splat-renderer_amd/build.py          # both

Outputs are git-ignored (*.so) but travel to the GPU box with the gpurun snapshot.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "rtgs", "_lib", "librtgs_hip.so")
ORACLE_SRC = os.path.join(ROOT, "oracle", "rtgs_oracle.c")
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "librtgs_oracle.so")

HIP_SOURCES = ["rtgs_trace.hip", "rtgs_ops.hip"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_hip(force: bool = False, extra_flags=()) -> str:
    """hipcc --offload-arch=gfx950 -> rtgs/_lib/librtgs_hip.so (cross-compiles without a GPU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "rtgs_common.hpp"), os.path.join(ROOT, "include", "rtgs_hip.h")]
    if not force and _newer(LIB, deps):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    _run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *extra_flags, *srcs, "-o", LIB])
    return LIB


def build_oracle(force: bool = False) -> str | None:
    """gcc -> oracle/_build/librtgs_oracle.so: the C restatement used only as a checker."""
    if not os.path.exists(ORACLE_SRC):
        return None
    if not force and _newer(ORACLE_LIB, [ORACLE_SRC]):
        return ORACLE_LIB
    os.makedirs(os.path.dirname(ORACLE_LIB), exist_ok=True)
    _run(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off",
          ORACLE_SRC, "-o", ORACLE_LIB, "-lm"])
    return ORACLE_LIB


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_hip(force))
    print(build_oracle(force))
