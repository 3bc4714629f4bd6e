"""HBM traffic of EVERY kernel of an LM run from rocprofv3 PMC counters, against the library's byte model.

  run    (under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace): calibration reads of a known byte count
         (rba_debug_read_blocks, 4- and 16-byte loads), then `iters` LM iterations on venice-1778; prints one JSON line
         with the calibration byte count, the byte model (rba_get_byte_model), the executed PCG counts and the
         number of stage launches
  parse  <fetch_dir> <write_dir> <meta.json> <out_prefix>: per-kernel table (CSV) and per-stage sums (JSON) with the
         model beside them. FETCH_SIZE is corrected by the factor calibrated on the streaming read (the guide's gfx950
         correction: it reports half the bytes of a wide coalesced read), WRITE_SIZE is taken as reported (KiB).
Driven by scripts/run_pmc_stage_traffic.sh; outputs are committed under profiles/ and held against the byte model by
tests/test_byte_model.py."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# kernel-name substring -> launch group of rba_byte_model
GROUPS = [
    ("k_calib_read<1>", "calib1"), ("k_calib_read<4>", "calib4"),
    ("k_compute_error", "compute_error"), ("k_reduce_rows<8>", "compute_error"),
    ("k_s1_geometry", "stage1"), ("k_s1_qr", "stage1"), ("k_s1_fused", "stage1"), ("k_cam_pass_mfma<float, 1>", "stage1"), ("k_cam_pass_mfma<double, 1>", "stage1"),
    ("k_scale_gram", "stage1"), ("k_pose_scaling", "stage1"),
    ("k_s2_obs", "stage2"), ("k_cam_pass_mfma<float, 0>", "stage2"), ("k_cam_pass_mfma<double, 0>", "stage2"), ("k_invert_blocks", "stage2"),
    ("k_hx_implicit", "product_matrix_free"), ("k_scale_vec", "product_matrix_free"),
    ("k_pcgp", "persistent_solve"),  # (the persistent PCG kernel: the matrix once per solve + the exchanged records)
    ("k_pcgs_spmv", "product_assembled"),
    ("k_ex_offdiag", "assembly"), ("k_s12_cols", "assembly"), ("k_ex_set_diag", "assembly"), ("k_ex_copy_diag", "assembly"),
    ("k_a64_", "assembly"),
    ("k_pcgs_update", "pcg_vectors"), ("k_pcg_", "pcg_vectors"), ("k_pcgs_", "pcg_vectors"),
    ("k_bs_", "back_substitution"), ("k_sum_ldiff", "back_substitution"), ("k_reduce_rows<1>", "back_substitution"),
    ("k_update_cameras", "back_substitution"),
]


def group_of(kernel):
    for sub, g in GROUPS:
        if sub in kernel:
            return g
    return "other"


def run(iters):
    import ctypes as C
    import numpy as np
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd import problem as P
    from rootba_amd.linearizor import LinearizorHIP
    prob = P.preprocess(P.named_synthetic("venice-1778"), translation_sigma=0.01, point_sigma=0.01)
    g = LinearizorHIP(prob, np.float32, L.default_options(robust_norm=1, huber_parameter=1.0, max_num_iterations=iters,
                                                          function_tolerance=0.0))
    nbytes = C.c_int64(0)
    for _ in range(3):
        L.check(g.lib.rba_debug_read_blocks(g.h, 1, C.byref(nbytes)), "calib1")
        L.check(g.lib.rba_debug_read_blocks(g.h, 4, C.byref(nbytes)), "calib4")
    log, _ = g.optimize_lm()
    meta = {"calib_bytes": nbytes.value, "byte_model": g.byte_model(), "pcg": g.pcg_counters(),
            "hx_bytes": g.problem_stats()["hx_bytes"],
            "lm_iterations": len(log) - 1, "linearizations": sum(1 for r in log[1:] if r.stage1_time > 0),
            "cost_evaluations": g.pcg_counters()["cost_evaluations"],  # (counted by the library since round 5)
            "cg_iterations": [r.cg_iterations for r in log[1:]]}
    g.close()
    print("PMC_META " + json.dumps(meta))


def _rows(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                k = r["Kernel_Name"]
                a = out.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return out


def parse(fetch_dir, write_dir, meta_path, out_prefix):
    meta = json.load(open(meta_path))
    fetch, write = _rows(fetch_dir, "FETCH_SIZE"), _rows(write_dir, "WRITE_SIZE")
    kib = 1024.0
    cal = {}
    for k, (v, n) in fetch.items():
        if "k_calib_read<1>" in k:
            cal["c1"] = meta["calib_bytes"] / (v / n * kib)
        if "k_calib_read<4>" in k:
            cal["c4"] = meta["calib_bytes"] / (v / n * kib)
    corr = cal.get("c4", 2.0)  # 16-byte streaming read: the guide's factor 2 on gfx950
    table, groups = [], {}
    for k in sorted(set(fetch) | set(write)):
        fv, fn = fetch.get(k, (0.0, 0))
        wv, wn = write.get(k, (0.0, 0))
        n = max(fn, wn)
        short = k.split("(")[0].replace("void rba::", "").replace("rba::", "")
        fb, wb = fv * kib * corr, wv * kib
        g = group_of(k)
        table.append((short, g, n, fb / max(1, fn), wb / max(1, wn)))
        if not g.startswith("calib"):
            a = groups.setdefault(g, [0.0, 0.0])
            a[0] += fb
            a[1] += wb
    with open(out_prefix + ".csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launch_group", "dispatches", "fetch_bytes_per_dispatch(FETCH_SIZE*1024*corr)",
                    "write_bytes_per_dispatch(WRITE_SIZE*1024)"])
        for row in sorted(table, key=lambda r: -(r[3] + r[4]) * r[2]):
            w.writerow([row[0], row[1], row[2], f"{row[3]:.0f}", f"{row[4]:.0f}"])
    bm, pcg = dict(meta["byte_model"]), meta["pcg"]
    launches = {"compute_error": meta["cost_evaluations"], "stage1": meta["linearizations"], "stage2": meta["lm_iterations"],
                "back_substitution": meta["lm_iterations"], "product_matrix_free": pcg["products_matrix_free"],
                "product_assembled": pcg["products_assembled"] - pcg.get("products_assembled_resident", 0),
                "assembly": pcg["assemblies"], "pcg_vectors": pcg["iterations"] - pcg.get("iterations_resident", 0),
                "persistent_solve": pcg.get("solves_persistent", 0)}
    # (a persistent solve: the one load of the matrix + the records its iterations exchange)
    if pcg.get("solves_persistent", 0):
        bm = dict(bm)
        bm["persistent_solve"] = bm["persistent_solve"] + bm["persistent_iteration"] * pcg["iterations_resident"] / pcg["solves_persistent"]
    out = {"workload": "venice-1778 synthetic, float32, SQUARE_ROOT + SCHUR_JACOBI, Huber(1)", "meta": meta,
           "fetch_correction": {"measured_4B_loads": cal.get("c1"), "measured_16B_loads": cal.get("c4"), "used": corr},
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only; KiB * 1024; "
                     "FETCH_SIZE multiplied by the factor calibrated on a streaming read of a known byte count "
                     "(MI355X_MICROARCH.md: gfx950 reports half the bytes of a wide coalesced read)",
           "groups": {}}
    for g, n in launches.items():
        fb, wb = groups.get(g, [0.0, 0.0])
        out["groups"][g] = {"launches": n, "measured_bytes_per_launch": (fb + wb) / n if n else None,
                            "fetch_bytes_per_launch": fb / n if n else None, "write_bytes_per_launch": wb / n if n else None,
                            "model_bytes_per_launch": bm[g],
                            "measured_over_model": ((fb + wb) / n / bm[g]) if n and bm[g] else None}
    json.dump(out, open(out_prefix + ".json", "w"), indent=1)
    hx = out["groups"]["product_matrix_free"]
    json.dump({"venice-1778/implicit_q": {"traffic_bytes_per_launch": hx["measured_bytes_per_launch"],
                                          "algorithmic_bytes_per_launch": meta["hx_bytes"],
                                          "model_bytes_per_launch": hx["model_bytes_per_launch"],
                                          "method": out["method"], "source": os.path.basename(out_prefix) + ".json"}},
              open(os.path.join(os.path.dirname(out_prefix), "hx_traffic.json"), "w"), indent=1)
    for g, v in out["groups"].items():
        print(f"{g:22s} launches {v['launches']:5d}  measured {((v['measured_bytes_per_launch'] or 0) / 1e6):9.2f} MB  "
              f"model {v['model_bytes_per_launch'] / 1e6:9.2f} MB  ratio {v['measured_over_model']}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    else:
        parse(*sys.argv[2:6])
