"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.

    python scripts/pmc_summary.py <dir> [<dir> ...] [--filter substr] > table.csv
Rows: kernel, dispatches, then one column per counter (mean per dispatch), plus VGPR / LDS / grid /
workgroup size from the same file.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    filt = None
    for i, a in enumerate(sys.argv):
        if a == "--filter":
            filt = sys.argv[i + 1]
    dirs = [d for d in dirs if d != filt]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    meta = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if filt and filt not in k:
                    continue
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
                meta[k] = (r.get("VGPR_Count") or r.get("Arch_VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"),
                           r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    counters = sorted({c for k in acc for c in acc[k]})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatches", "vgpr", "agpr", "sgpr", "lds", "scratch", "grid", "wg"] + counters)
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", acc[k][counters[0]])[0]):
        n = max(v[1] for v in acc[k].values())
        short = k.split("(")[0].replace("void rba::", "").replace("rba::", "")
        w.writerow([short, n, *meta[k]] + [f"{acc[k][c][0] / acc[k][c][1]:.6g}" if c in acc[k] else "" for c in counters])


if __name__ == "__main__":
    main()
