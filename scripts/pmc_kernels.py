"""Per-kernel means of whatever counters the rocprofv3 --pmc passes in the given directories collected, for the kernels
whose name contains one of the given substrings (what bounds k_a64_offdiag).
  python scripts/pmc_kernels.py <out.csv> <substr,substr,...> <dir> [<dir> ...]
Durations come from the kernel trace of the same passes (i.e. under counter collection)."""
import csv
import glob
import os
import sys


def main():
    out_csv, pats, dirs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    cnt, dur = {}, {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void rba::", "").replace("rba::", "")
                if not any(p in k for p in pats):
                    continue
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
        w.writerow(["kernel", "mean_duration_us(under counter collection)"] + [n + "(mean per dispatch)" for n in names])
        for k in sorted(cnt):
            d_ns = dur[k][0] / dur[k][1] if k in dur else None
            w.writerow([k, f"{d_ns * 1e-3:.1f}" if d_ns else ""] +
                       [f"{cnt[k][n][0] / cnt[k][n][1]:.0f}" if n in cnt[k] else "" for n in names])
    print(open(out_csv).read())


if __name__ == "__main__":
    main()
