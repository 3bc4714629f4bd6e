"""Builds tests/hipemu/_build/librootba_hip_emu.so: the product's solver.hip + kernel headers, UNCHANGED except for one
textual rewrite (`extern __shared__` -> `extern thread_local`: the dynamic-LDS arrays are defined by the harness) and one
substituted header (pg_record_io.hpp: the inline-assembly record accesses of the persistent PCG kernel), compiled as plain
C++ against tests/hipemu/hip/hip_runtime.h. TEST INFRASTRUCTURE ONLY - see that header.

    python tests/hipemu/build_emu.py
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get("HIPEMU_BUILD_DIR", os.path.join(HERE, "_build"))  # (a second build beside a running session)
LIB = os.path.join(OUT, "librootba_hip_emu.so")
CXX = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
# HIPEMU_SANITIZE=address: the kernels and the host side of the library under AddressSanitizer (device buffers are host
# allocations here, so an out-of-bounds access of a kernel is reported with the kernel's source line); use a build
# directory of its own (HIPEMU_BUILD_DIR) and run python with the runtime preloaded - see README.md
SANITIZE = os.environ.get("HIPEMU_SANITIZE", "")


def sources():
    return sorted(glob.glob(os.path.join(ROOT, "rootba_amd", "csrc", "*.hpp")) +
                  glob.glob(os.path.join(ROOT, "rootba_amd", "csrc", "*.hip")))


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    deps = sources() + [os.path.join(ROOT, "include", "rootba_hip.h"), os.path.join(HERE, "hip", "hip_runtime.h"),
                        os.path.join(HERE, "hipemu_runtime.cpp"), os.path.join(HERE, "fake_rccl.cpp"), os.path.join(HERE, "pg_record_io.hpp"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force: bool = False) -> str:
    if not (force or stale()):
        return LIB
    src_dir = os.path.join(OUT, "rootba_amd", "csrc")
    os.makedirs(src_dir, exist_ok=True)
    os.makedirs(os.path.join(OUT, "include"), exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "rootba_hip.h"), os.path.join(OUT, "include", "rootba_hip.h"))
    for path in sources():
        txt = open(path).read().replace("extern __shared__", "extern thread_local")
        open(os.path.join(src_dir, os.path.basename(path)), "w").write(txt)
    # the one product header the harness replaces: the inline-assembly record accesses of the persistent PCG kernel
    shutil.copy(os.path.join(HERE, "pg_record_io.hpp"), os.path.join(src_dir, "pg_record_io.hpp"))
    cmd = [CXX, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-mavx2", "-mfma", "-ffp-contract=fast",
           "-Wno-unused-value", "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-DHIPEMU=1",
           "-I", HERE, os.path.join(src_dir, "solver.hip"), "-x", "c++", os.path.join(HERE, "hipemu_runtime.cpp"),
           "-o", LIB, "-ldl", "-lpthread"]
    if SANITIZE:
        cmd[1:1] = [f"-fsanitize={SANITIZE}", "-fno-omit-frame-pointer", "-shared-libsan"]
        if "alignment" in SANITIZE:
            cmd[1:1] = ["-DHIPEMU_STRICT_ALIGN=1"]
        if SANITIZE == "thread":
            # the work-items of a workgroup are fibers on ONE OS thread: no function entry / exit events, or the
            # sanitizer's per-thread shadow call stack would see unbalanced calls across the context switches
            cmd[1:1] = ["-mllvm", "-tsan-instrument-func-entry-exit=0"]
    subprocess.check_call(cmd)
    # the file-based stand-in for librccl.so.1 (multi-process runs on the harness: put its directory on LD_LIBRARY_PATH)
    os.makedirs(os.path.join(OUT, "fake_rccl"), exist_ok=True)
    subprocess.check_call([CXX, "-std=c++17", "-O1", "-fPIC", "-shared", os.path.join(HERE, "fake_rccl.cpp"), "-o",
                           os.path.join(OUT, "fake_rccl", "librccl.so.1"), "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
