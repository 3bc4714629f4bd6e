"""Phase times of the persistent PCG kernel from its debug stamps (RBA_PCGP_TRACE=<file>, kernels_pcgp.hpp): per
iteration eight shader-clock stamps of work-item 0 of every workgroup -
  0 loop top | 1 exchange 1 done (z, rho/Q partial sums) | 2 direction staged | 3 product + row sums | 4 p.q published |
  5 exchange 2 done | 6 step done | 7 z and partial sums published.
Prints, per solve, the median over iterations of: the mean / max over workgroups of every phase, and the skew of the
workgroups' arrival at the two exchanges.  usage: python scripts/pcgp_trace.py trace.txt"""
import sys
import numpy as np

path = sys.argv[1]
ghz = 0.1  # s_memrealtime ticks per nanosecond
solves, cur = [], None
for line in open(path):
    if line.startswith("solve"):
        cur = []
        solves.append((line.strip(), cur))
    else:
        cur.append([int(v) for v in line.split()])
names = ["exchange1 wait", "decide+stage p", "product+rowsum", "wg sum+publish pq", "exchange2 wait", "step(+refresh)", "close: z, partials"]
for head, rows in solves:
    a = np.array(rows, dtype=np.int64)
    if a.size == 0:
        continue
    G = a[:, 0].max() + 1
    n_it = min(np.bincount(a[:, 0]))
    if n_it < 4:
        continue
    t = np.zeros((G, n_it, 8), dtype=np.int64)
    for r in a:
        if r[1] < n_it:
            t[r[0], r[1]] = r[2:]
    per_it = np.median(np.diff(t[:, :, 0], axis=1), axis=1).mean()
    scale = 1.0 / (ghz * 1e3) if ghz else 1.0
    unit = "us" if ghz else "cycles"
    print(f"{head}: {n_it} traced iterations, {per_it * scale:.2f} {unit} per iteration")
    d = np.diff(t, axis=2)  # [G, it, 7]
    for k in range(7):
        mean_wg = d[:, 1:, k].mean(axis=0)
        max_wg = d[:, 1:, k].max(axis=0)
        print(f"  {names[k]:22s} mean over workgroups {np.median(mean_wg) * scale:8.2f}   slowest {np.median(max_wg) * scale:8.2f} {unit}")
    for k, nm in ((4, "publish pq"), (7, "publish z/partials")):
        skew = t[:, 1:, k].max(axis=0) - t[:, 1:, k].min(axis=0)
        print(f"  skew of '{nm}' over the workgroups: median {np.median(skew) * scale:.2f} {unit}")
    # from the LAST workgroup's publication to the first / last workgroup's completion of the exchange
    for kp, kd, nm in ((4, 5, "exchange 2"), (7, 1, "exchange 1")):
        if kd > kp:
            last_pub = t[:, 1:, kp].max(axis=0)
            done = t[:, 1:, kd]
        else:
            last_pub = t[:, 1:-1, kp].max(axis=0)
            done = t[:, 2:, kd]
        print(f"  {nm}: last publication -> first workgroup through {np.median(done.min(axis=0) - last_pub) * scale:.2f}, "
              f"last through {np.median(done.max(axis=0) - last_pub) * scale:.2f} {unit}")
