"""Accuracy ENSEMBLE of the float32 landmark QR (helper of tests/test_gpu_qr_accuracy.py and scripts/qr_accuracy.py;
VERDICT round 5, next 1).

The one lock-step assertion that was exempted through round 5 - final-13682, iteration 6 - sits at a state where a few
landmarks are at |p| ~ 7e4 of a scene of size 100: their 2k x 3 blocks `Jl` are nearly rank-deficient (the depth
direction is unobservable at that parallax) and the 3-iteration increment of the fused stage-1 kernel was 1.41e-2 from the
float64 iterate where the float32 CPU oracle's was 9e-4. One state says nothing about whether the fused summation order is
systematically worse. This module draws >= 10^4 such landmark blocks - tracks of k = 2 ... 8 neighbouring cameras of a
ring, the landmark pushed out along its viewing ray to a parallax down to 4e-7 - and compares what the landmark QR
hands on (landmark_block_base.ipp:717-743: `R`, `Q1^T r`, `|Q2^T r|`), the gradient `b` and a 3-iteration increment of

  (a) the fused stage-1 kernel `k_s1_fused_obs` (an observation per lane; the default),
  (b) the two-kernel form (`RBA_S1_FUSED=0`: `k_s1_geometry` + `k_s1_qr_tile`, a block row per lane, the oracle's order),
  (c) the float32 CPU oracle,

each with the FLOAT64 oracle from the identical float32-representable state (float scaling epsilon). Sign conventions
are removed (rows of `R` and `Q1^T r` multiplied by the sign of `R`'s diagonal; `Q2^T r` enters by its norm), the
landmark-Jacobian column scaling is divided out with each implementation's own stored scale."""
import os

import numpy as np

EPS_SQRT_FLOAT = 3.1622776601683794e-3


def make_ensemble(n_lms=12288, n_cams=2048, seed=11, t_lo=1.0, t_hi=5e3):
    """Ring scene of rootba_amd.problem.synthetic_problem with k = 2 ... 8, every landmark pushed out along the ray from
    the centre of its cameras by a factor log-uniform in [t_lo, t_hi] (depth 15 -> up to 7.5e4: |p| ~ 7e4 as in the
    final-13682 state), observations re-projected with 0.5 px noise, cameras perturbed by 1e-2. With 2048 cameras on
    the ring of radius 10 neighbouring cameras are 0.03 apart (final-13682: 0.005 before its normalisation): the
    parallax of the farthest landmarks is ~4e-7 and cond(R) of the scaled columns reaches 1e6 - the regime in which the
    third column of Q1 is decided by float32 rounding."""
    from rootba_amd import problem as P
    rng = np.random.default_rng(seed)
    k = rng.integers(2, 9, size=n_lms)
    raw = P.synthetic_problem(n_cams, n_lms, int(k.sum()), seed=seed, k=k)
    lm_of_obs = np.repeat(np.arange(n_lms), k)
    R = P.quat_to_rot(raw.cams[:, :4])
    centers = -np.einsum("nji,nj->ni", R, raw.cams[:, 4:7])
    c_mean = np.zeros((n_lms, 3))
    np.add.at(c_mean, lm_of_obs, centers[raw.obs_cam_idx])
    c_mean /= k[:, None]
    t = np.exp(rng.uniform(np.log(t_lo), np.log(t_hi), n_lms))
    raw.lms = c_mean + t[:, None] * (raw.lms - c_mean)
    proj, z = P.project(raw.cams, raw.lms[lm_of_obs], raw.obs_cam_idx)
    assert (z > 0.1).all()
    raw.obs_xy = proj + rng.normal(0, 0.5, proj.shape)
    prob = P.perturb(raw, 0.0, 0.01, 0.0, seed)
    # float32-representable state and observations: every implementation starts from identical numbers
    prob.cams = prob.cams.astype(np.float32).astype(np.float64)
    prob.lms = prob.lms.astype(np.float32).astype(np.float64)
    prob.obs_xy = prob.obs_xy.astype(np.float32).astype(np.float64)
    prob.name = "qr-ensemble"
    return prob, t


def _signed(R6, q1, scale):
    """R (n, 3, 3) with the column scaling divided out and rows signed so that diag >= 0; Q1^T r signed alike."""
    n = R6.shape[0]
    R = np.zeros((n, 3, 3))
    iu = np.triu_indices(3)
    R[:, iu[0], iu[1]] = np.asarray(R6, np.float64)
    sg = np.where(np.diagonal(R, axis1=1, axis2=2) < 0, -1.0, 1.0)
    R = R * sg[:, :, None] / np.asarray(scale, np.float64)[:, None, :]
    return R, np.asarray(q1, np.float64) * sg


def oracle_factors(o, k):
    """(R6, Q1^T r, |Q2^T r|, jl_scale) of every landmark from the oracle's dense blocks after stage 1."""
    n = len(k)
    R6, q1, q2 = np.zeros((n, 6)), np.zeros((n, 3)), np.zeros(n)
    iu = np.triu_indices(3)
    for l in range(n):
        blk, li = o.block(l)
        R6[l] = blk[:3, li:li + 3][iu]
        q1[l] = blk[:3, li + 3]
        q2[l] = np.linalg.norm(blk[3:2 * k[l], li + 3].astype(np.float64))
    return R6, q1, q2, np.asarray(o.jl_col_scale(), np.float64)


def gpu_factors(g):
    R6, q1 = g.landmark_R(damped=False)
    return np.asarray(R6, np.float64), np.asarray(q1, np.float64), np.asarray(g.landmark_q2tr_norm(), np.float64), \
        np.asarray(g.jl_col_scale(), np.float64)


def block_errors(f, f64):
    """Per-landmark errors of one implementation's factors against the float64 ones: R^T R (unscaled, relative),
    signed Q1^T r and |Q2^T r| relative to |r|."""
    R, q1 = _signed(f[0], f[1], f[3])
    R64, q164 = _signed(f64[0], f64[1], f64[3])
    G, G64 = np.einsum("nki,nkj->nij", R, R), np.einsum("nki,nkj->nij", R64, R64)
    rn = np.sqrt((q164 ** 2).sum(1) + f64[2] ** 2)
    return {"RtR": np.linalg.norm(G - G64, axis=(1, 2)) / np.linalg.norm(G64, axis=(1, 2)),
            "Q1tr": np.linalg.norm(q1 - q164, axis=1) / rn,
            "Q2tr_norm": np.abs(f[2] - f64[2]) / rn}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(a) + np.linalg.norm(b)))


def run(prob, lam=1.2e-6, cg_it=3, forms=("fused", "two-kernel"), log=print):
    """Returns {"blocks": {impl: {metric: {median, p90, p99, max}}}, "b": {impl: rel}, "inc": {impl: rel},
    "cond": percentiles of cond(R64)} with impl in forms + ("oracle32",)."""
    import torch  # noqa: F401
    from oracle import oracle as O
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    k = prob.obs_per_lm()
    kw = dict(robust_norm=1, huber_parameter=1.0, max_cg_it=cg_it, eta=0.0)
    o64 = O.Oracle(prob, np.float64, O.default_options(**dict(kw, jacobi_scaling_eps=EPS_SQRT_FLOAT)))
    assert o64.linearize() == 0
    f64 = oracle_factors(o64, k)
    out = {"n_landmarks": int(prob.n_lms), "lambda": lam, "cg_iterations": cg_it, "blocks": {}, "b": {}, "inc": {}}
    R64, _ = _signed(f64[0], f64[1], np.ones_like(f64[3]))  # scaled columns: what the factorisation sees
    cond = np.abs(np.diagonal(R64, axis1=1, axis2=2)).max(1) / np.maximum(np.abs(R64[:, 2, 2]), 1e-300)
    out["cond_R_scaled"] = {p: float(np.percentile(cond, q)) for p, q in (("median", 50), ("p90", 90), ("p99", 99), ("max", 100))}
    inc64, _ = o64.solve(lam)
    b64 = np.asarray(o64.last_b(), np.float64)
    impls = {}
    o32 = O.Oracle(prob, np.float32, O.default_options(**kw))
    assert o32.linearize() == 0
    f32 = oracle_factors(o32, k)
    inc32, _ = o32.solve(lam)
    impls["oracle32"] = (f32, np.asarray(o32.last_b(), np.float64), inc32)
    for form in forms:
        old = os.environ.get("RBA_S1_FUSED")
        os.environ["RBA_S1_FUSED"] = "1" if form == "fused" else "0"
        try:
            g = LinearizorHIP(prob, np.float32, L.default_options(**kw))
        finally:
            if old is None:
                del os.environ["RBA_S1_FUSED"]
            else:
                os.environ["RBA_S1_FUSED"] = old
        assert g.linearize() == 0
        fg = gpu_factors(g)
        b, _ = g.stage2(lam, blocks=False)
        inc, cg = g.solve(lam)
        assert cg.num_iterations == cg_it
        impls[form] = (fg, np.asarray(b, np.float64), inc)
        g.close()
    for name, (f, b, inc) in impls.items():
        e = block_errors(f, f64)
        out["blocks"][name] = {m: {p: float(np.percentile(v, q)) for p, q in (("median", 50), ("p90", 90), ("p99", 99), ("max", 100))}
                               for m, v in e.items()}
        out["b"][name] = rel(b, b64)
        out["inc"][name] = rel(inc, inc64)
        log(f"{name:11s} b {out['b'][name]:.3e}  inc({cg_it}) {out['inc'][name]:.3e}  " + "  ".join(
            f"{m} med {s['median']:.2e} p99 {s['p99']:.2e} max {s['max']:.2e}" for m, s in out["blocks"][name].items()))
    return out
