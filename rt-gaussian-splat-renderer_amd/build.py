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
# the same sources with -DRTGS_COUNT=1: candidate / hit / insertion counters for bench.py's honesty
# columns (roofline.tests_executed, cull_rate); never timed, never loaded by the rtgs package
COUNT_LIB = os.path.join(HERE, "rtgs", "_lib", "librtgs_hip_count.so")
ORACLE_SRC = os.path.join(ROOT, "oracle", "rtgs_oracle.c")
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "librtgs_oracle.so")

HIP_SOURCES = ["rtgs_trace.hip", "rtgs_ops.hip"]


# -fno-slp-vectorize: the SLP vectoriser pairs fp32 FMAs into v_pk_fma_f32 with their broadcast operands DUPLICATED into
# register pairs (15 SH basis values -> 30 VGPRs); without it every k_trace form needs fewer registers and runs faster
# (round 4: c3 0.1320 -> 0.1285 ms, c5 1.304 -> 1.273, c3 + SH 0.2595 -> 0.2400)
BASE_FLAGS = ["-O3", "-fno-slp-vectorize", "--offload-arch=gfx950", "-std=c++17"]


def source_hash() -> str:
    """sha256 (16 hex digits) of everything librtgs_hip.so is compiled from: csrc/*,
    include/rtgs_hip.h and the compiler flags every build shares.  Stored in the profile summaries under profiles/ so that bench.py can tell
    whether a committed counter belongs to the build it is timing."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(BASE_FLAGS).encode() + b"\0")
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h")))
    for f in files + [os.path.join(ROOT, "include", "rtgs_hip.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def _hip_deps():
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    return srcs, srcs + [os.path.join(CSRC, "rtgs_common.hpp"), os.path.join(ROOT, "include", "rtgs_hip.h")]


def _stamp(extra_flags=()) -> str:
    """What the stamp beside a library says: the source hash, and the extra compile flags when there
    are any -- a -DRTGS_ABLATE / STAMPS / COUNT build must never pass for the product library."""
    flags = " ".join(extra_flags)
    return source_hash() + (" " + flags if flags else "")


def _stamp_ok(target, extra_flags=()) -> bool:
    """The library was built from exactly these sources WITH exactly these flags (stamp beside it)
    -- mtimes alone say nothing after a checkout or a snapshot copy."""
    try:
        with open(target + ".srchash") as f:
            return f.read().strip() == _stamp(extra_flags) and os.path.exists(target)
    except OSError:
        return False


def _hipcc(out, extra_flags=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs, _ = _hip_deps()
    os.makedirs(os.path.dirname(out), exist_ok=True)
    _run([hipcc, *BASE_FLAGS, "-fPIC", "-shared",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *extra_flags, *srcs, "-o", out])
    with open(out + ".srchash", "w") as f:
        f.write(_stamp(extra_flags) + "\n")


def build_hip(force: bool = False, extra_flags=()) -> str:
    """hipcc --offload-arch=gfx950 -> rtgs/_lib/librtgs_hip.so (cross-compiles without a GPU).
    Skipped only when the stamp written by the last build matches the tree and the flags: after a
    diagnostic build (extra_flags: tools/ab.sh and friends) the next plain build_hip() recompiles."""
    if force or not _stamp_ok(LIB, extra_flags):
        _hipcc(LIB, extra_flags)
    return LIB


def build_count(force: bool = False) -> str:
    """The counting variant (see COUNT_LIB)."""
    if force or not _stamp_ok(COUNT_LIB, ["-DRTGS_COUNT=1"]):
        _hipcc(COUNT_LIB, ["-DRTGS_COUNT=1"])
    return COUNT_LIB


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
    print(build_count(force))
    print(build_oracle(force))
