"""Build libfastenhancer_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfastenhancer_hip.so")
SOURCES = ["fe_api.hip"]
DEPS = ["fe_api.hip", "fe_kernels.hip.h", os.path.join("..", "..", "include", "fastenhancer_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[fastenhancer_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
