"""HBM traffic of one H*x from rocprofv3 PMC counters (run on the GPU box).

    python scripts/pmc_traffic.py run  <outdir>   # workload profiled by rocprofv3
    python scripts/pmc_traffic.py parse <fetch_dir> <write_dir> <out.json>

`run` executes, on venice-1778: stage 1 + stage 2, then 3 x (calibration read of
the block storage with 4-byte loads, the same with 16-byte loads, one H*x) with the dense
blocks, then 3 x one H*x with the implicit-Q operator.
FETCH_SIZE / WRITE_SIZE are collected in SEPARATE rocprofv3 passes
(MI355X_MICROARCH.md: FETCH_SIZE needs 3 of 4 TCC slots, WRITE_SIZE 2) with
--kernel-trace only. Units: KiB. gfx950 correction: FETCH_SIZE under-reports wide
coalesced reads; the factor is CALIBRATED here on the two streaming reads of
known size and applied to the H*x kernels (k_hx<...> uses 4-byte loads,
k_hx_small 16-byte loads).
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import ctypes as C
    import numpy as np
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd import problem as P
    from rootba_amd.linearizor import LinearizorHIP
    prob = P.preprocess(P.named_synthetic("venice-1778"), translation_sigma=0.5, point_sigma=0.5)
    x = np.random.default_rng(0).normal(size=9 * prob.n_cams).astype(np.float32)
    meta = {}
    # dense blocks first (its kernels k_hx<..>, k_hx_small), then the implicit-Q operator (k_hx_implicit*)
    for mode, iq in (("dense", 0), ("implicit_q", 1)):
        g = LinearizorHIP(prob, np.float32, L.default_options(robust_norm=1, implicit_q=iq))
        assert g.linearize() == 0
        g.stage2(1e-4)
        nbytes = C.c_int64(0)
        for _ in range(3):
            if not iq:
                L.check(g.lib.rba_debug_read_blocks(g.h, 1, C.byref(nbytes)), "calib1")
                L.check(g.lib.rba_debug_read_blocks(g.h, 4, C.byref(nbytes)), "calib4")
            g.right_multiply(x)
        if not iq:
            meta["calib_bytes"] = nbytes.value
        meta["hx_bytes_" + mode] = g.problem_stats()["hx_bytes"]
        g.close()
    meta["hx_bytes"] = meta["hx_bytes_dense"]
    print(json.dumps(meta))


def _counter_rows(d, counter):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                out.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return out


def parse(fetch_dir, write_dir, out_path, calib_bytes, hx_bytes, hx_bytes_implicit=None):
    res = {}
    for name, d, counter in (("fetch", fetch_dir, "FETCH_SIZE"), ("write", write_dir, "WRITE_SIZE")):
        rows = _counter_rows(d, counter)
        agg = {}
        for k, v in rows:
            key = ("calib1" if "k_calib_read<1>" in k else "calib4" if "k_calib_read<4>" in k else
                   "hx_small" if "k_hx_small" in k else "hx" if "k_hx<" in k else
                   "hx_implicit" if "k_hx_implicit" in k else None)
            if key:
                agg.setdefault(key, []).append(v)
        res[name] = {k: sum(v) / 3.0 * 1024.0 for k, v in agg.items()}  # KiB -> bytes, per repetition
    f = res["fetch"]
    c1 = calib_bytes / f["calib1"]
    c4 = calib_bytes / f["calib4"]
    traffic = f["hx"] * c1 + f["hx_small"] * c4 + res["write"].get("hx", 0) + res["write"].get("hx_small", 0)
    out = {"venice-1778": {
        "traffic_bytes_per_launch": traffic,
        "algorithmic_bytes_per_launch": hx_bytes,
        "fetch_raw_bytes": f, "write_raw_bytes": res["write"],
        "fetch_correction_4B_loads": c1, "fetch_correction_16B_loads": c4,
        "calibration_bytes": calib_bytes,
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, --kernel-trace; KiB*1024; "
                  "FETCH_SIZE corrected by factors calibrated on streaming reads of the block storage",
    }}
    if hx_bytes_implicit and "hx_implicit" in f:
        # k_hx_implicit reads the stage-1 records with 4- and 16-byte loads (both calibrate to the same factor)
        out["venice-1778/implicit_q"] = {
            "traffic_bytes_per_launch": f["hx_implicit"] * c1 + res["write"].get("hx_implicit", 0),
            "algorithmic_bytes_per_launch": hx_bytes_implicit,
            "fetch_raw_bytes": f["hx_implicit"], "write_raw_bytes": res["write"].get("hx_implicit", 0),
            "fetch_correction_4B_loads": c1,
            "method": out["venice-1778"]["method"],
        }
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        meta = json.load(open(sys.argv[5]))
        parse(sys.argv[2], sys.argv[3], sys.argv[4], meta["calib_bytes"], meta["hx_bytes"],
              meta.get("hx_bytes_implicit_q"))
