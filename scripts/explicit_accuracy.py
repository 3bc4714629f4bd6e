"""float32 accuracy of the two forms of the square-root operator against the float64 oracle:
matrix-free (sum_l A_l^T (A_l x)) vs explicitly assembled reduced matrix. Run on the GPU box."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401
from oracle import oracle as O
from rootba_amd import _lib as L, problem as P
from rootba_amd.linearizor import LinearizorHIP


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


for name in ("ladybug-49", "trafalgar-257"):
    prob = P.preprocess(P.named_synthetic(name), translation_sigma=0.5, point_sigma=0.5)
    o64 = O.Oracle(prob, np.float64, O.default_options(robust_norm=1))
    o32 = O.Oracle(prob, np.float32, O.default_options(robust_norm=1))
    g = LinearizorHIP(prob, np.float32, L.default_options(robust_norm=1))
    assert o64.linearize() == 0 and o32.linearize() == 0 and g.linearize() == 0
    rng = np.random.default_rng(0)
    for lam in (1e-4, 1e-7):
        for o in (o64, o32):
            o.set_pose_damping(lam)
            o.stage2(lam, o.pose_scaling() if lam == 1e-4 else None)
        g.stage2(lam)
        x = rng.uniform(-1, 1, 9 * prob.n_cams)
        y64 = o64.right_multiply(x)
        print(f"{name} lambda={lam:g}: oracle f32 {rel(o32.right_multiply(x.astype(np.float32)), y64):.2e}  "
              f"gpu matrix-free {rel(g.right_multiply(x.astype(np.float32)), y64):.2e}  "
              f"gpu explicit {rel(g.right_multiply_explicit(x.astype(np.float32)), y64):.2e}")
        ym = g.right_multiply(x.astype(np.float32)); ye = g.right_multiply_explicit(x.astype(np.float32))
        ym2 = g.right_multiply(x.astype(np.float32))
        print(f"    explicit vs matrix-free {rel(ye, ym):.2e}; matrix-free run-to-run {rel(ym2, ym):.2e}")
