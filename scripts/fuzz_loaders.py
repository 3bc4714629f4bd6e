"""Mutation fuzzing of the host loaders (BAL text, .cereal cache, Bundler text) with an AddressSanitizer + UBSan build of the
CLI: truncations, byte flips, insertions and blown-up counts of a small valid file; every run has to end with exit code 0
(still a valid file) or 2 (rejected with a message) and without a sanitizer report.

    python scripts/fuzz_loaders.py [runs-per-format]

Round-2 record: 220 + 220 + 250 runs, no report, exit codes 2 (574) and 0 (116).
"""
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rootba_amd import build  # noqa: E402
from rootba_amd import problem as P  # noqa: E402

OUT = os.environ.get("FUZZ_DIR", "/tmp/rootba_fuzz")
APP = os.path.join(OUT, "bal_qr_hip_san")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=4096")


def build_app():
    os.makedirs(OUT, exist_ok=True)
    build.build()
    here = os.path.join(ROOT, "rootba_amd")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17",
                           "-pthread", os.path.join(here, "csrc", "host", "bal_qr_hip.cpp"), "-o", APP, "-L" + here,
                           "-lrootba_hip", "-Wl,-rpath," + here, "-Wl,-rpath-link,/opt/rocm/lib"])


def mutate(data, rng, text):
    m = bytearray(data)
    kind = rng.randrange(4)
    lo, hi = (32, 127) if text else (0, 256)
    if kind == 0:
        m = m[:rng.randrange(len(m))]
    elif kind == 1:
        for _ in range(rng.randrange(1, 8)):
            m[rng.randrange(len(m))] = rng.randrange(lo, hi)
    elif kind == 2:
        p = rng.randrange(len(m))
        m[p:p] = bytes(rng.randrange(lo, hi) for _ in range(rng.randrange(1, 40)))
    else:  # a count or an index blown up
        p = rng.randrange(max(1, len(m) - 10))
        m[p:p + 8] = b" 9999999" if text else rng.choice([2**31 - 1, 2**32 + 5, 2**62, 2**64 - 1]).to_bytes(8, "little")
    return bytes(m)


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    build_app()
    import test_oracle_vs_reference as T  # the Bundler writer of the tests
    raw = P.synthetic_problem(12, 80, 300, seed=4)
    bal = os.path.join(OUT, "problem-12-80-pre.txt")
    P.write_bal(raw, bal)
    cer = os.path.join(OUT, "problem-12-80-pre.cereal")
    subprocess.check_call([APP, "--input", bal, "--dry-run", "--no-normalize", "--save-output", "--output-optimized-path", cer],
                          stdout=subprocess.DEVNULL, env=ENV)
    bun = os.path.join(OUT, "bundle.out")
    T._write_bundler(raw, bun)
    rng = random.Random(1)
    problems, codes = 0, {}
    for src, name, extra in ((bal, "problem-m%d-pre.txt", []), (cer, "problem-m%d-pre.cereal", []),
                             (bun, "bundle_m%d.out", ["--input-type", "BUNDLER"])):
        data = open(src, "rb").read()
        for i in range(runs):
            path = os.path.join(OUT, name % i)
            open(path, "wb").write(mutate(data, rng, not src.endswith(".cereal")))
            try:
                r = subprocess.run([APP, "--input", path, "--dry-run", *extra], capture_output=True, text=True, timeout=60, env=ENV)
            except subprocess.TimeoutExpired:
                problems += 1
                print("TIMEOUT", path)
                continue
            codes[r.returncode] = codes.get(r.returncode, 0) + 1
            if "Sanitizer" in r.stderr or "runtime error" in r.stderr or r.returncode not in (0, 2):
                problems += 1
                print("PROBLEM", path, r.returncode, r.stderr[:800])
            else:
                os.remove(path)
    print("problems", problems, "exit codes", codes)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
