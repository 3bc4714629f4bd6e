"""The float32 landmark QR on an ENSEMBLE of nearly rank-deficient landmark blocks (helper and
rationale: tests/qr_ensemble.py). Semantics: perform_qr_householder, src/rootba/qr/impl/landmark_block_base.ipp:717-743,
after scale_Jl_cols (:571-587).

Stated bar: what the GPU's stage 1 hands on - R^T R, the signed Q1^T r, |Q2^T r| per landmark, the gradient b and a
3-iteration increment - is as close to the float64 oracle's as the float32 CPU oracle's own: the median and the 99th
percentile of every per-landmark error within 2 x the oracle's (plus one float epsilon), the two vectors within
2 x + 1e-5."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float32).eps)


@pytest.mark.parametrize("lam", [1.2e-6, 1e-4])
def test_stage1_qr_accuracy_ensemble_near_rank_deficiency(lam):
    import qr_ensemble as Q
    prob, t = Q.make_ensemble(n_lms=12288, seed=11)
    assert prob.n_lms >= 10000 and t.max() > 1e3
    r = Q.run(prob, lam=lam)
    # the ensemble is what it claims to be: a tenth of the blocks has cond(R) > 1e4 in the scaled columns
    assert r["cond_R_scaled"]["p90"] > 1e4, r["cond_R_scaled"]
    for form in ("fused", "two-kernel"):
        for metric, s in r["blocks"][form].items():
            o = r["blocks"]["oracle32"][metric]
            for p in ("median", "p99"):
                assert s[p] <= 2 * o[p] + EPS, (form, metric, p, s, o)
        assert r["b"][form] <= 2 * r["b"]["oracle32"] + 1e-5, (form, r["b"])
        assert r["inc"][form] <= 2 * r["inc"]["oracle32"] + 1e-5, (form, r["inc"])
