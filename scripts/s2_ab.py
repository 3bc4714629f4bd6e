"""A/B timing of stage 2 under environment switches (development aid; first use: the round-2 candidates
RBA_CAM_BLOCKS=1 and RBA_S2_FUSED_LM=1, which were written without a GPU and validated on tests/hipemu only).
Every variant builds its own solver on the same problem, linearises once and runs repeated stage 2 calls (the first
one includes the Gram accumulation of a new linearisation point, the others are the re-damping form); prints the
stage-2 HIP-event time of both and the difference of b / blocks against the default variant.
usage: python scripts/s2_ab.py [workload] "ENV=val,ENV=val" ...   ('' = defaults; always run first as the reference)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from rootba_amd import problem as P
from rootba_amd.linearizor import LinearizorHIP
from rootba_amd import _lib as L


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(a) + np.linalg.norm(b)))


def run(prob, envs, ref=None, dt=np.float32, reps=5):
    saved = {}
    for kv in [e for e in envs.split(",") if e]:
        k, v = kv.split("=")
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        g = LinearizorHIP(prob, dt, L.default_options(robust_norm=1, huber_parameter=1.0))
        first, later = [], []
        for rep in range(reps):
            assert g.linearize() == 0
            b, blocks = g.stage2(1e-4)
            first.append(1e3 * g.timings().stage2_time)
            b2, blocks2 = g.stage2(1e-3)
            later.append(1e3 * g.timings().stage2_time)
        out = dict(env=envs, stage2_first_ms=round(min(first), 4), stage2_redamp_ms=round(min(later), 4))
        if ref is not None:
            out.update(b=rel(b, ref[0]), blocks=rel(blocks, ref[1]), b_redamp=rel(b2, ref[2]), blocks_redamp=rel(blocks2, ref[3]))
        return out, (b, blocks, b2, blocks2)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
    variants = sys.argv[2:] or ["RBA_CAM_BLOCKS=1", "RBA_S2_FUSED_LM=1", "RBA_CAM_BLOCKS=1,RBA_S2_FUSED_LM=1"]
    prob = P.preprocess(P.named_synthetic(name), translation_sigma=0.01, point_sigma=0.01)
    out, ref = run(prob, "")
    print(json.dumps(out), flush=True)
    for v in variants:
        print(json.dumps(run(prob, v, ref)[0]), flush=True)
