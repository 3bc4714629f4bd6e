"""Host-side BAL problem data model (numpy), synthetic BAL-shaped scenes, and
the reference's preprocessing pipeline.

Mirrors `rootba::BalProblem<Scalar>` (reference
src/rootba/bal/bal_problem.hpp:61-234) in the flat CSR/SoA form that the C-ABI
of include/rootba_hip.h consumes:

* cameras  `cams[n_c, 10]` = (qx, qy, qz, qw, tx, ty, tz, f, k1, k2)
  (`Camera::params()`, bal_problem.hpp:84-95; world-to-camera pose `T_c_w`)
* landmarks `lms[n_l, 3]` = p_w
* observations in CSR order: `lm_obs_offsets[n_l+1]`, `obs_cam_idx[n_o]`
  (ascending inside one landmark = `std::map` iteration order,
  bal_problem.hpp:131), `obs_xy[n_o, 2]` (image y already flipped as the
  loader does, bal_problem.cpp:243).

Everything here is one-off host work (load / generate / normalise); it is
outside the accelerated hot path.
"""
from __future__ import annotations

from dataclasses import dataclass, replace

import numpy as np

RANDOM_SEED = 38401  # reference default `random_seed` (bal_dataset_options.hpp:74-79)

# (n_cams, n_lms, target n_obs) of the BASELINE.json configs
# (reference scripts/num_ops/bal_numbers.csv:2,4,5; docs/PoBATutorial.md:162)
BAL_SIZES = {
    "ladybug-49": (49, 7776, 31843),
    "trafalgar-257": (257, 65132, 225911),
    "venice-1778": (1778, 993923, 5001946),
    "final-13682": (13682, 4456117, 28987644),
}


@dataclass
class BalProblem:
    cams: np.ndarray  # [n_c, 10] float64
    lms: np.ndarray  # [n_l, 3] float64
    lm_obs_offsets: np.ndarray  # [n_l + 1] int64
    obs_cam_idx: np.ndarray  # [n_o] int32
    obs_xy: np.ndarray  # [n_o, 2] float64
    name: str = "bal"

    @property
    def n_cams(self) -> int:
        return int(self.cams.shape[0])

    @property
    def n_lms(self) -> int:
        return int(self.lms.shape[0])

    @property
    def n_obs(self) -> int:
        return int(self.obs_cam_idx.shape[0])

    def obs_per_lm(self) -> np.ndarray:
        return np.diff(self.lm_obs_offsets)

    def copy(self) -> "BalProblem":
        return BalProblem(self.cams.copy(), self.lms.copy(), self.lm_obs_offsets.copy(),
                          self.obs_cam_idx.copy(), self.obs_xy.copy(), self.name)

    def cast(self, dtype) -> "BalProblem":
        """`copy_cast<Scalar>` (bal_problem.cpp:794-832): state arrays only."""
        return replace(self, cams=self.cams.astype(dtype), lms=self.lms.astype(dtype),
                       obs_xy=self.obs_xy.astype(dtype))

    def block_stats(self, scalar_bytes: int = 4) -> dict:
        """Algorithmic sizes of SURVEY.md §8d for this topology."""
        k = self.obs_per_lm().astype(np.int64)
        pad = (4 - (9 * k) % 4) % 4
        cols = 9 * k + pad + 4
        storage = int(((2 * k + 3) * cols).sum()) * scalar_bytes
        hx = (int((2 * k * (9 * k + pad)).sum()) * scalar_bytes + 4 * self.n_obs
              + scalar_bytes * 2 * 9 * self.n_cams)
        return {"k_mean": float(k.mean()), "k2_mean": float((k * k).mean()), "k_max": int(k.max()),
                "block_storage_bytes": storage, "hx_bytes": hx, "hx_flops": int((72 * k * k).sum())}


# ---------------------------------------------------------------------------
# SO(3) helpers (Sophus conventions, quaternion stored x,y,z,w)
# ---------------------------------------------------------------------------
def quat_to_rot(q: np.ndarray) -> np.ndarray:
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - z * w)
    R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w)
    R[..., 2, 1] = 2 * (y * z + x * w)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def rot_to_quat(R: np.ndarray) -> np.ndarray:
    """Rotation matrices [..., 3, 3] -> unit quaternions (x, y, z, w), w >= 0."""
    R = np.asarray(R, dtype=np.float64)
    flat = R.reshape(-1, 3, 3)
    out = np.empty((flat.shape[0], 4))
    for i, m in enumerate(flat):
        tr = m[0, 0] + m[1, 1] + m[2, 2]
        if tr > 0:
            s = np.sqrt(tr + 1.0) * 2
            w = 0.25 * s
            x = (m[2, 1] - m[1, 2]) / s
            y = (m[0, 2] - m[2, 0]) / s
            z = (m[1, 0] - m[0, 1]) / s
        elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
            s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
            w = (m[2, 1] - m[1, 2]) / s
            x = 0.25 * s
            y = (m[0, 1] + m[1, 0]) / s
            z = (m[0, 2] + m[2, 0]) / s
        elif m[1, 1] > m[2, 2]:
            s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
            w = (m[0, 2] - m[2, 0]) / s
            x = (m[0, 1] + m[1, 0]) / s
            y = 0.25 * s
            z = (m[1, 2] + m[2, 1]) / s
        else:
            s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
            w = (m[1, 0] - m[0, 1]) / s
            x = (m[0, 2] + m[2, 0]) / s
            y = (m[1, 2] + m[2, 1]) / s
            z = 0.25 * s
        qv = np.array([x, y, z, w])
        if w < 0:
            qv = -qv
        out[i] = qv / np.linalg.norm(qv)
    return out.reshape(R.shape[:-2] + (4,))


def so3_exp(w: np.ndarray) -> np.ndarray:
    """Rodrigues: rotation vectors [..., 3] -> rotation matrices."""
    w = np.asarray(w, dtype=np.float64)
    theta = np.linalg.norm(w, axis=-1)[..., None, None]
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1] = -w[..., 2]
    K[..., 0, 2] = w[..., 1]
    K[..., 1, 0] = w[..., 2]
    K[..., 1, 2] = -w[..., 0]
    K[..., 2, 0] = -w[..., 1]
    K[..., 2, 1] = w[..., 0]
    small = theta < 1e-8
    th = np.where(small, 1.0, theta)
    a = np.where(small, 1.0 - theta**2 / 6, np.sin(th) / th)
    b = np.where(small, 0.5 - theta**2 / 24, (1 - np.cos(th)) / th**2)
    return np.eye(3) + a * K + b * (K @ K)


def so3_log(R: np.ndarray) -> np.ndarray:
    """Rotation matrices -> rotation vectors (via the quaternion)."""
    q = rot_to_quat(R)
    v = q[..., :3]
    w = q[..., 3]
    n = np.linalg.norm(v, axis=-1)
    ang = 2 * np.arctan2(n, w)
    scale = np.where(n < 1e-12, 2.0, ang / np.where(n < 1e-12, 1.0, n))
    return v * scale[..., None]


def project(cams: np.ndarray, p_w: np.ndarray, cam_idx=None):
    """Snavely projection without the minus sign (reference
    src/rootba/bal/snavely_projection.hpp:182-190). cams [n,10], p_w [n,3] - or, with `cam_idx` [n], the camera
    TABLE [n_cams,10] (the rotation matrices are then formed once per camera and gathered: the same values bit for
    bit, 30 M quaternion conversions less at final-13682 size).
    Returns (proj [n,2], z [n])."""
    if cam_idx is not None:
        R = quat_to_rot(cams[:, :4])[cam_idx]
        cams = cams[cam_idx]
    else:
        R = quat_to_rot(cams[:, :4])
    p_c = np.einsum("nij,nj->ni", R, p_w) + cams[:, 4:7]
    m = p_c[:, :2] / p_c[:, 2:3]
    r2 = (m * m).sum(1)
    scale = 1.0 + r2 * (cams[:, 8] + r2 * cams[:, 9])
    return cams[:, 7:8] * scale[:, None] * m, p_c[:, 2]


# ---------------------------------------------------------------------------
# Synthetic BAL-shaped scenes
# ---------------------------------------------------------------------------
def synthetic_problem(n_cams: int, n_lms: int, n_obs_target: int, seed: int = RANDOM_SEED,
                      obs_noise: float = 0.5, name: str = "synthetic", k=None) -> BalProblem:
    """Cameras on a closed loop of radius 10 looking inward (+-5 deg jitter),
    landmarks in front of a random camera observed by it and its k-1 nearest
    loop neighbours, k = min(n_c, 2 + Geometric0(1/(kbar-1)))."""
    rng = np.random.default_rng(seed)
    kbar = n_obs_target / n_lms
    p = 1.0 / (kbar - 1.0)
    k_draw = np.minimum(n_cams, 2 + (rng.geometric(p, size=n_lms) - 1)).astype(np.int64)
    # `k` (optional) overrides the observation counts, e.g. to hit every k-class in tests
    k = k_draw if k is None else np.minimum(n_cams, np.asarray(k, dtype=np.int64))
    assert k.shape == (n_lms,) and k.min() >= 2

    # cameras
    theta = 2 * np.pi * np.arange(n_cams) / n_cams
    centers = np.stack([10 * np.cos(theta), 10 * np.sin(theta), np.zeros(n_cams)], 1)
    fwd = -np.stack([np.cos(theta), np.sin(theta), np.zeros(n_cams)], 1)  # looking inward
    right = np.stack([-np.sin(theta), np.cos(theta), np.zeros(n_cams)], 1)
    down = np.cross(fwd, right)
    R_w_c = np.stack([right, down, fwd], axis=2)  # columns = camera axes in world
    jitter = so3_exp(rng.normal(0.0, np.deg2rad(5.0) / 2, size=(n_cams, 3)).clip(
        -np.deg2rad(5.0), np.deg2rad(5.0)))
    R_w_c = np.einsum("nij,njk->nik", R_w_c, jitter)
    R_c_w = np.transpose(R_w_c, (0, 2, 1))
    t_c_w = -np.einsum("nij,nj->ni", R_c_w, centers)
    cams = np.empty((n_cams, 10))
    cams[:, :4] = rot_to_quat(R_c_w)
    cams[:, 4:7] = t_c_w
    cams[:, 7] = rng.uniform(500, 2000, n_cams)
    cams[:, 8] = rng.normal(0, 1e-2, n_cams) * 1e-1
    cams[:, 9] = rng.normal(0, 1e-3, n_cams) * 1e-1

    # topology: base camera + nearest loop neighbours 0,+1,-1,+2,-2,...
    off = np.zeros(n_lms + 1, dtype=np.int64)
    np.cumsum(k, out=off[1:])
    n_obs = int(off[-1])
    lm_of_obs = np.repeat(np.arange(n_lms, dtype=np.int64), k)
    j = np.arange(n_obs, dtype=np.int64) - off[lm_of_obs]
    delta = np.where(j % 2 == 1, (j + 1) // 2, -(j // 2))
    base = rng.integers(0, n_cams, size=n_lms)
    cam_idx = (base[lm_of_obs] + delta) % n_cams
    order = np.lexsort((cam_idx, lm_of_obs))  # ascending camera inside a landmark
    cam_idx = cam_idx[order].astype(np.int32)

    # landmarks: in front of the base camera, re-drawn until every obs has z >= 0.1
    lms = np.empty((n_lms, 3))
    todo = np.arange(n_lms)
    for _ in range(100):
        depth = rng.uniform(2, 20, todo.size)
        lateral = rng.uniform(-0.35, 0.35, (todo.size, 2)) * depth[:, None]
        # long tracks (seen from a large part of the ring): near the centre of the loop, where every
        # inward-looking camera has them in front
        wide = k[todo] > max(8, n_cams // 8)
        depth[wide] = rng.uniform(8, 12, int(wide.sum()))
        lateral[wide] = rng.uniform(-0.12, 0.12, (int(wide.sum()), 2)) * depth[wide, None]
        p_c = np.concatenate([lateral, depth[:, None]], 1)
        lms[todo] = np.einsum("nij,nj->ni", R_w_c[base[todo]], p_c) + centers[base[todo]]
        sel = np.isin(lm_of_obs, todo) if todo.size < n_lms else slice(None)
        lm_sel = lm_of_obs[sel]
        _, z = project(cams, lms[lm_sel], cam_idx[sel])
        bad = np.unique(lm_sel[z < 0.1])
        if bad.size == 0:
            break
        todo = bad
    else:
        raise RuntimeError("could not place all landmarks in front of their cameras")

    proj, _ = project(cams, lms[lm_of_obs], cam_idx)
    obs_xy = proj + rng.normal(0, obs_noise, proj.shape)
    return BalProblem(cams, lms, off, cam_idx, obs_xy, name)


def heavy_tail_counts(n_cams: int, n_lms: int, n_obs_target: int, seed: int = RANDOM_SEED) -> np.ndarray:
    """Observation counts with a realistic heavy tail: the geometric bulk of `synthetic_problem` plus
    max(8, n_lms / 5000) long tracks with a Pareto(alpha = 1) length, P(k > x) = 40 / x, capped at
    max(120, n_cams / 4) (real BAL tracks: a few landmarks are seen by hundreds of cameras; one seen by
    EVERY camera would make the reduced camera matrix dense, which is a different workload)."""
    rng = np.random.default_rng(seed + 1)
    kbar = n_obs_target / n_lms
    k = np.minimum(n_cams, 2 + (rng.geometric(1.0 / (kbar - 1.0), size=n_lms) - 1)).astype(np.int64)
    n_tail = max(8, n_lms // 5000)
    cap = min(n_cams, max(120, n_cams // 4))
    tail = np.minimum(cap, (40.0 / rng.uniform(40.0 / (4 * n_cams), 1.0, n_tail)).astype(np.int64))
    tail[0] = cap
    k[rng.choice(n_lms, n_tail, replace=False)] = np.maximum(2, tail)
    return k


def named_synthetic(config: str, seed: int = RANDOM_SEED) -> BalProblem:
    """`<bal size name>` or `<bal size name>+tail` (same sizes, heavy-tailed track lengths)."""
    base, _, variant = config.partition("+")
    n_c, n_l, n_o = BAL_SIZES[base]
    k = heavy_tail_counts(n_c, n_l, n_o, seed) if variant == "tail" else None
    return synthetic_problem(n_c, n_l, n_o, seed=seed, name=f"synthetic-{config}", k=k)


# ---------------------------------------------------------------------------
# Reference preprocessing (bal_problem.cpp:428-554, 794-832)
# ---------------------------------------------------------------------------
def _median_upper(x: np.ndarray) -> float:
    """`median_destructive` (bal_problem.cpp:116-122): element n/2 of the sorted data."""
    n = x.shape[0]
    return float(np.partition(x, n // 2)[n // 2])


def normalize(prob: BalProblem, new_scale: float = 100.0) -> BalProblem:
    """`BalProblem::normalize` (bal_problem.cpp:428-469)."""
    med = np.array([_median_upper(prob.lms[:, j].copy()) for j in range(3)])
    mad = _median_upper(np.abs(prob.lms - med).sum(1))
    scale = new_scale / mad
    out = prob.copy()
    out.lms = scale * (prob.lms - med)
    R = quat_to_rot(prob.cams[:, :4])
    centers = -np.einsum("nji,nj->ni", R, prob.cams[:, 4:7])  # T_w_c translation
    centers = scale * (centers - med)
    out.cams[:, 4:7] = -np.einsum("nij,nj->ni", R, centers)
    return out


def perturb(prob: BalProblem, rotation_sigma: float, translation_sigma: float,
            point_sigma: float, seed: int = RANDOM_SEED) -> BalProblem:
    """`BalProblem::perturb` (bal_problem.cpp:508-554). The reference draws from
    `std::default_random_engine`, which is not reproducible across standard
    libraries; same distributions, numpy generator here."""
    rng = np.random.default_rng(seed)
    out = prob.copy()
    if rotation_sigma > 0 or translation_sigma > 0:
        R = quat_to_rot(out.cams[:, :4])
        if translation_sigma > 0:
            centers = -np.einsum("nji,nj->ni", R, out.cams[:, 4:7])
            centers = centers + rng.normal(0, translation_sigma, centers.shape)
            out.cams[:, 4:7] = -np.einsum("nij,nj->ni", R, centers)
        if rotation_sigma > 0:
            dR = so3_exp(rng.normal(0, rotation_sigma, (prob.n_cams, 3)))
            R = np.einsum("nij,njk->nik", dR, R)
            out.cams[:, :4] = rot_to_quat(R)
    if point_sigma > 0:
        out.lms = out.lms + rng.normal(0, point_sigma, out.lms.shape)
    return out


def filter_obs(prob: BalProblem, threshold: float) -> BalProblem:
    """`BalProblem::filter_obs` (bal_problem.cpp:471-506): drop observations with
    depth < threshold, then landmarks with fewer than 2 observations."""
    if threshold <= 0:
        return prob
    k = prob.obs_per_lm()
    lm_of_obs = np.repeat(np.arange(prob.n_lms, dtype=np.int64), k)
    _, z = project(prob.cams, prob.lms[lm_of_obs], prob.obs_cam_idx)
    keep = z >= threshold
    k_new = np.bincount(lm_of_obs[keep], minlength=prob.n_lms)
    lm_keep = k_new >= 2
    keep &= lm_keep[lm_of_obs]
    off = np.zeros(int(lm_keep.sum()) + 1, dtype=np.int64)
    np.cumsum(k_new[lm_keep], out=off[1:])
    return BalProblem(prob.cams.copy(), prob.lms[lm_keep].copy(), off,
                      prob.obs_cam_idx[keep].copy(), prob.obs_xy[keep].copy(), prob.name)


def preprocess(prob: BalProblem, normalization_scale: float = 100.0, rotation_sigma: float = 0.0,
               translation_sigma: float = 0.01, point_sigma: float = 0.01,
               init_depth_threshold: float = 0.1, seed: int = RANDOM_SEED) -> BalProblem:
    """`load_normalized_bal_problem` order (bal_problem.cpp:794-832) with the
    CVPR'21 common dataset settings as defaults (docs/Configuration.md:283-296)."""
    out = normalize(prob, normalization_scale) if normalization_scale > 0 else prob.copy()
    out = perturb(out, rotation_sigma, translation_sigma, point_sigma, seed)
    return filter_obs(out, init_depth_threshold)


# ---------------------------------------------------------------------------
# BAL text format (bal_problem.cpp:190-282)
# ---------------------------------------------------------------------------
_AXIS_INV = np.diag([1.0, -1.0, -1.0])


def write_bal(prob: BalProblem, path: str) -> None:
    k = prob.obs_per_lm()
    lm_of_obs = np.repeat(np.arange(prob.n_lms, dtype=np.int64), k)
    R = quat_to_rot(prob.cams[:, :4])
    rvec = so3_log(np.einsum("ij,njk->nik", _AXIS_INV, R))
    t_bal = prob.cams[:, 4:7] @ _AXIS_INV.T
    with open(path, "w") as f:
        f.write(f"{prob.n_cams} {prob.n_lms} {prob.n_obs}\n")
        for c, l, (x, y) in zip(prob.obs_cam_idx, lm_of_obs, prob.obs_xy):
            f.write(f"{int(c)} {int(l)} {x:.17g} {-y:.17g}\n")
        for i in range(prob.n_cams):
            for v in (*rvec[i], *t_bal[i], *prob.cams[i, 7:10]):
                f.write(f"{v:.17g}\n")
        for p in prob.lms:
            for v in p:
                f.write(f"{v:.17g}\n")


def read_bal(path: str) -> BalProblem:
    """`BalProblem::load_bal` (bal_problem.cpp:190-282)."""
    with open(path) as f:
        tok = f.read().split()
    n_c, n_l, n_o = int(tok[0]), int(tok[1]), int(tok[2])
    assert n_c > 0 and n_l > 0 and n_o > 0
    obs = np.array(tok[3:3 + 4 * n_o], dtype=np.float64).reshape(n_o, 4)
    cam_idx = obs[:, 0].astype(np.int64)
    lm_idx = obs[:, 1].astype(np.int64)
    assert cam_idx.min() >= 0 and cam_idx.max() < n_c and lm_idx.min() >= 0 and lm_idx.max() < n_l
    xy = obs[:, 2:4].copy()
    xy[:, 1] = -xy[:, 1]
    order = np.lexsort((cam_idx, lm_idx))
    cam_idx, lm_idx, xy = cam_idx[order], lm_idx[order], xy[order]
    dup = (np.diff(lm_idx) == 0) & (np.diff(cam_idx) == 0)
    if dup.any():
        raise ValueError(f"Invalid file '{path}': duplicate (camera, landmark) observation")
    pos = 3 + 4 * n_o
    cp = np.array(tok[pos:pos + 9 * n_c], dtype=np.float64).reshape(n_c, 9)
    pos += 9 * n_c
    lms = np.array(tok[pos:pos + 3 * n_l], dtype=np.float64).reshape(n_l, 3)
    R = np.einsum("ij,njk->nik", _AXIS_INV, so3_exp(cp[:, :3]))
    cams = np.empty((n_c, 10))
    cams[:, :4] = rot_to_quat(R)
    cams[:, 4:7] = cp[:, 3:6] @ _AXIS_INV.T
    cams[:, 7:10] = cp[:, 6:9]
    off = np.zeros(n_l + 1, dtype=np.int64)
    np.cumsum(np.bincount(lm_idx, minlength=n_l), out=off[1:])
    return BalProblem(cams, lms, off, cam_idx.astype(np.int32), xy, name=path)
