"""Matrix-core counters of the kernels that use v_mfma_* (north_star: "MFMA-busy counters against gfx950 peak").

  run <float32|float64> <iters>   (under rocprofv3 --pmc ... --kernel-trace): `iters` LM iterations on venice-1778 with
                                  explicit_after = 1, so that every solve assembles the reduced matrix
  parse <dir> [<dir> ...] <out.csv>   per-kernel means of every counter found + derived columns:
         mfma_flops = SQ_INSTS_VALU_MFMA_MOPS_<F32|F64> * 512 (the definition of rocprofv3's MfmaFlops*),
         TFLOP/s = mfma_flops / kernel duration (from the kernel trace of the same pass),
         frac_of_peak against the dense matrix peak of the dtype (MI355X_MICROARCH.md: 157.3 TF f32, 78.6 TF f64),
         mfma_busy = rocprofv3's MfmaUtil: SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) / (GRBM_GUI_ACTIVE
         per XCD * 1024); the dispatch rows of GRBM_GUI_ACTIVE are sums over the 8 XCDs (checked against the kernel
         duration: 18.9 M cycles for a 1062 us kernel = 8 x 2.23 GHz x 1062 us).
Driven by scripts/run_pmc_mfma.sh; the summary is committed under profiles/."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = {"F32": 157.3e12, "F64": 78.6e12}


def run(dts, iters):
    import numpy as np
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd import problem as P
    from rootba_amd.linearizor import LinearizorHIP
    prob = P.preprocess(P.named_synthetic("venice-1778"), translation_sigma=0.01, point_sigma=0.01)
    dt = np.float64 if dts == "float64" else np.float32
    g = LinearizorHIP(prob, dt, L.default_options(robust_norm=1, huber_parameter=1.0, max_num_iterations=iters,
                                                  function_tolerance=0.0, explicit_after=1))
    log, _ = g.optimize_lm()
    print("PMC_META " + json.dumps({"dtype": dts, "cg_iterations": [r.cg_iterations for r in log[1:]],
                                    "pcg": g.pcg_counters()}))
    g.close()


def parse(dirs, out_csv):
    cnt, dur = {}, {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void rba::", "").replace("rba::", "")
                a = cnt.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void rba::", "").replace("rba::", "")
                a = dur.setdefault(k, [0.0, 0])
                a[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                a[1] += 1
    names = sorted({c for v in cnt.values() for c in v})
    with open(out_csv, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "mean_duration_us(under counter collection)"] + [n + "(mean per dispatch)" for n in names] +
                   ["mfma_TFLOPs", "frac_of_dense_matrix_peak", "mfma_busy(SQ_VALU_MFMA_BUSY_CYCLES/(GRBM_GUI_ACTIVE/8*1024))"])
        for k in sorted(cnt):
            mean = {n: (cnt[k][n][0] / cnt[k][n][1]) if n in cnt[k] else None for n in names}
            if not any((mean.get(n) or 0) > 0 for n in names if "MFMA" in n):
                continue
            d_ns = dur[k][0] / dur[k][1] if k in dur else None
            tf, frac = None, None
            for ty in ("F32", "F64"):
                mops = mean.get(f"SQ_INSTS_VALU_MFMA_MOPS_{ty}")
                if mops and d_ns:
                    tf = mops * 512 / (d_ns * 1e-9) * 1e-12
                    frac = tf * 1e12 / PEAK[ty]
            busy = None
            if mean.get("SQ_VALU_MFMA_BUSY_CYCLES") and mean.get("GRBM_GUI_ACTIVE"):
                busy = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8 * 1024)
            n_disp = max(v[1] for v in cnt[k].values())
            w.writerow([k, n_disp, f"{d_ns * 1e-3:.1f}" if d_ns else ""] +
                       [f"{mean[n]:.0f}" if mean[n] is not None else "" for n in names] +
                       [f"{tf:.2f}" if tf else "", f"{frac:.4f}" if frac else "", f"{busy:.4f}" if busy else ""])
    print(open(out_csv).read())


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]))
    else:
        parse(sys.argv[2:-1], sys.argv[-1])
