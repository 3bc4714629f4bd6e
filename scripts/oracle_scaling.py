"""CPU-oracle product scaling on this host (development aid): time per H*x vs OpenMP threads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rootba_amd import problem as P
from oracle import oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
prob = P.preprocess(P.named_synthetic(name))
k = prob.obs_per_lm().astype(np.int64); nbytes = int((2 * k * (9 * k)).sum()) * 4
x = np.random.default_rng(0).normal(size=9 * prob.n_cams).astype(np.float32)
for nt in [int(a) for a in sys.argv[2:]] or [8, 32, 64, 128, 256]:
    o = O.Oracle(prob, np.float32, O.default_options(robust_norm=1, num_threads=nt))
    assert o.linearize() == 0
    o.solve(1e-4)
    for _ in range(2): o.right_multiply(x)
    t = time.perf_counter(); n = 10
    for _ in range(n): o.right_multiply(x)
    dt = (time.perf_counter() - t) / n
    print(f"threads {nt:4d}: {dt*1e3:8.2f} ms per product, {nbytes/dt/1e9:7.1f} GB/s", flush=True)
    del o
