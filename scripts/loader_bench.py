"""Time the C++ BAL loader (rootba_amd/csrc/host/bal_problem.hpp) on a venice-1778-sized text file
for several parser thread counts. Host-only; writes gpurun_out/loader_bench.json when run on the GPU box.

    python scripts/loader_bench.py [workload] [threads ...]
"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rootba_amd import build, problem as P  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
    threads = [int(t) for t in sys.argv[2:]] or [1, 8, 32, 128]
    app = build.build_app()
    path = f"/tmp/problem-{workload}.txt"
    if not os.path.exists(path):
        t = time.time()
        P.write_bal(P.named_synthetic(workload), path)
        print(f"generated {path} in {time.time() - t:.1f}s ({os.path.getsize(path) / 1e6:.0f} MB)", flush=True)
    rows = []
    for t in threads:
        best = None
        for _ in range(3):
            out = subprocess.run([app, "--input", path, "--dry-run"], capture_output=True, text=True,
                                 env=dict(os.environ, RBA_HOST_THREADS=str(t)), check=True)
            info = json.loads(out.stdout.strip().splitlines()[-1])
            best = info["load_seconds"] if best is None else min(best, info["load_seconds"])
        rows.append({"threads": t, "load_normalize_filter_cast_seconds": best,
                     "MB_per_s": os.path.getsize(path) / 1e6 / best})
        print(rows[-1], flush=True)
    res = {"workload": workload, "file_MB": os.path.getsize(path) / 1e6, "num_observations": info["num_observations"],
           "host_cpus": os.cpu_count(), "rows": rows}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/loader_bench.json", "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
