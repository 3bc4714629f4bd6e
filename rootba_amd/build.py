"""Builds rootba_amd/librootba_hip.so for gfx950 with hipcc (in-tree, so the
.so travels to the GPU box with the snapshot)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librootba_hip.so")
SOURCES = ["solver.hip"]
DEPS = ["solver.hip", "kernels.hpp", "device_utils.hpp", os.path.join("..", "..", "include", "rootba_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [HIPCC, *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB, "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
