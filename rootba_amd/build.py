"""Builds rootba_amd/librootba_hip.so for gfx950 with hipcc (in-tree, so the
.so travels to the GPU box with the snapshot)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librootba_hip.so")
SOURCES = ["solver.hip"]
def _deps():
    """every source the library is built from (staleness check of the in-tree .so)"""
    import glob
    files = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.hip"))
    files.append(os.path.join(HERE, "..", "include", "rootba_hip.h"))
    return files
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


APP = os.path.join(HERE, "bal_qr_hip")
def _host_deps():
    import glob
    return glob.glob(os.path.join(CSRC, "host", "*.hpp")) + glob.glob(os.path.join(CSRC, "host", "*.cpp")) + \
        [os.path.join(HERE, "..", "include", "rootba_hip.h")]


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [HIPCC, *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB, "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_app(force, verbose)
    return LIB


def build_app(force: bool = False, verbose: bool = False) -> str:
    """The C++17 host layer + `bal_qr_hip` CLI (plain g++, links the C ABI only)."""
    stale = (not os.path.exists(APP) or any(
        os.path.getmtime(d) > os.path.getmtime(APP) for d in _host_deps())
        or os.path.getmtime(LIB) > os.path.getmtime(APP))
    if force or stale:
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-pthread", os.path.join(CSRC, "host", "bal_qr_hip.cpp"),
               "-o", APP, "-L" + HERE, "-lrootba_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return APP


if __name__ == "__main__":
    print(build(force=True, verbose=True))
