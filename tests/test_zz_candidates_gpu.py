"""Default-OFF kernel variants that were written when no GPU time was left (round 2) and have so far run on the CPU
execution harness of tests/hipemu only: they stay behind their switches until they have been measured on an MI355X.
These tests hold each of them to the oracle exactly like the default path (tests/test_gpu_parity.py), so the first GPU
run that includes this file says whether they are correct on the hardware; sorted last on purpose.

Not part of the acceptance suite: code that has never run on an MI355X does not belong in the run that certifies the
default path, so the file runs only with RBA_TEST_CANDIDATES=1 (scripts/run_round3_first_call.sh sets it, each
invocation under its own `timeout`) or on the CPU harness (RBA_EMU=1).

  RBA_S2_FUSED_LM=1   k_s2_w8_fused (kernels_s1.hpp): the landmark damping pass folded into the per-observation W8 pass
  RBA_CAM_BLOCKS=1    k_cam_block_accumulate / k_cam_block_finish (kernels.hpp): the stage-2 camera pass over the merged,
                      address-sorted observation lists of blocks of 8 cameras (float only)
  RBA_HX_THREADS=512  k_hx_implicit_lds<S, 512> (kernels.hpp): the LDS-private product with 512-thread workgroups -
                      156 VGPRs and no scratch in double (the 1024-thread instance is capped at 128 and spills 104
                      bytes per lane), at half the waves per CU
"""
import numpy as np
import pytest

from conftest import rel_err

import os

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.environ.get("RBA_TEST_CANDIDATES") == "1" or os.environ.get("RBA_EMU")),
                                 reason="default-off candidates: RBA_TEST_CANDIDATES=1 runs them")]
TOL = {np.float32: 1e-4, np.float64: 1e-10}


def _pair(prob, dtype, **kw):
    import torch  # noqa: F401  (HIP runtime first, as in bench.py)
    from oracle import oracle as O
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    base = dict(robust_norm=1, huber_parameter=1.0)
    base.update(kw)
    return LinearizorHIP(prob, dtype, L.default_options(**base)), O.Oracle(prob, dtype, O.default_options(**base))


def _assert_increment(prob, dtype, o, lam, ig, cg, io, co, tol, **okw):
    """Unconditional (tests/test_gpu_parity.py::_assert_increment): when the truncated solves stop one iteration apart,
    the oracle's iterate after EXACTLY the product's count is the comparison."""
    from oracle import oracle as O
    assert abs(cg.num_iterations - co.num_iterations) <= (1 if dtype == np.float32 else 0)
    ref = io
    if cg.num_iterations != co.num_iterations:
        base = dict(robust_norm=1, huber_parameter=1.0, max_cg_it=cg.num_iterations, eta=0.0)
        base.update(okw)
        o_n = O.Oracle(prob, dtype, O.default_options(**base))
        o_n.set_state(*o.get_state())
        assert o_n.linearize() == 0
        ref, cn = o_n.solve(lam)
        assert cn.num_iterations == cg.num_iterations
    assert rel_err(ig, ref) < tol, (rel_err(ig, ref), cg.num_iterations, co.num_iterations)


@pytest.fixture(scope="module")
def mixed_k_problem():
    from rootba_amd import problem as P
    k = np.concatenate([np.arange(2, 61), np.random.default_rng(21).integers(2, 30, 141)])
    raw = P.synthetic_problem(90, k.size, int(k.sum()), seed=21, k=k)
    return P.preprocess(raw, seed=21, translation_sigma=0.3, point_sigma=0.3)


SWITCHES = [{"RBA_S2_FUSED_LM": "1"}, {"RBA_CAM_BLOCKS": "1"}, {"RBA_S2_FUSED_LM": "1", "RBA_CAM_BLOCKS": "1"}]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("which", ["small", "mixed"])
@pytest.mark.parametrize("env", SWITCHES, ids=["fused-landmark-pass", "camera-blocks", "both"])
def test_stage2_variants(small_problem, mixed_k_problem, dtype, which, env, monkeypatch):
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    prob = small_problem if which == "small" else mixed_k_problem
    tol = TOL[dtype]
    g, o = _pair(prob, dtype)
    assert g.linearize() == 0 and o.linearize() == 0
    x = np.random.default_rng(0).uniform(-1, 1, 9 * prob.n_cams).astype(dtype)
    first = True
    for lam in (0.1, 0.0, 1e-6, 10.0):  # re-damping on one linearisation point
        o.set_pose_damping(lam)
        b_o, bl_o = o.stage2(lam, o.pose_scaling() if first else None)
        first = False
        b_g, bl_g = g.stage2(lam)
        assert rel_err(b_g, b_o) < tol and rel_err(bl_g, bl_o) < tol
        assert rel_err(g.right_multiply(x), o.right_multiply(x)) < tol
    inc = (np.random.default_rng(1).uniform(-1, 1, 9 * prob.n_cams) * 0.01).astype(dtype)
    lg, lo = g.back_substitute(inc), o.back_substitute(inc)
    assert abs(lg - lo) / (abs(lg) + abs(lo)) < tol
    assert rel_err(g.get_state()[1], o.get_state()[1]) < tol
    # a whole solve + update and a short LM run
    g2, o2 = _pair(prob, dtype, max_num_iterations=5)
    assert g2.linearize() == 0 and o2.linearize() == 0
    ig, cg = g2.solve(1e-4)
    io, co = o2.solve(1e-4)
    _assert_increment(prob, dtype, o2, 1e-4, ig, cg, io, co, 10 * tol)
    g3, o3 = _pair(prob, dtype, max_num_iterations=5)
    a, _ = g3.optimize_lm()
    b, _ = o3.optimize_lm()
    for r, q in zip(a[:4], b[:4]):
        assert bool(r.step_is_successful) == bool(q.step_is_successful)
        assert abs(r.cost - q.cost) <= (1e-4 if dtype == np.float32 else 1e-10) * q.cost


@pytest.mark.parametrize("env", SWITCHES[1:], ids=["camera-blocks", "both"])
def test_camera_blocks_with_invalid_projections_and_odd_camera_count(small_problem, env, monkeypatch):
    """A camera count that is not a multiple of the block size, cameras without valid observations (zero Jacobian
    blocks) and the JACOBI preconditioner / sdiag outputs of the pass."""
    from rootba_amd.problem import BalProblem
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    p = small_problem
    keep = p.obs_cam_idx < 37  # 37 cameras: the last block holds 5
    off = np.concatenate([[0], np.cumsum(np.add.reduceat(keep.astype(np.int64), p.lm_obs_offsets[:-1]))])
    ok = np.diff(off) >= 2
    sel = keep & np.repeat(ok, np.diff(p.lm_obs_offsets))
    off2 = np.concatenate([[0], np.cumsum(np.add.reduceat(sel.astype(np.int64), p.lm_obs_offsets[:-1])[ok])])
    cams = p.cams[:37].copy()
    x, y, z, w = cams[3, :4]
    cams[3, :4] = [w, -z, y, -x]
    cams[3, 4:7] *= np.array([1.0, -1.0, -1.0])
    prob = BalProblem(cams, p.lms[ok].copy(), off2, p.obs_cam_idx[sel], p.obs_xy[sel], "37-cameras")
    for kw in (dict(optimized_cost=1, use_valid_projections_only=1), dict(preconditioner_type=0), dict()):
        g, o = _pair(prob, np.float32, **kw)
        assert g.linearize() == 0 and o.linearize() == 0
        for lam in (1e-2, 1e-4):
            ig, cg = g.solve(lam)
            io, co = o.solve(lam)
            _assert_increment(prob, np.float32, o, lam, ig, cg, io, co, 2e-3, **kw)
        lg, lo = g.apply(io), o.apply(io)
        assert abs(lg - lo) / (abs(lg) + abs(lo)) < 1e-4


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("which", ["small", "mixed"])
@pytest.mark.parametrize("env", [{"RBA_HX_LDS": "2"}, {"RBA_HX_LDS": "2", "RBA_HX_WIN": "7"}], ids=["lds-private", "lds-window"])
def test_product_with_512_thread_workgroups(small_problem, mixed_k_problem, dtype, which, env, monkeypatch):
    """tests/test_gpu_parity.py::test_implicit_q_product_kernels with RBA_HX_THREADS=512, then a solve."""
    prob = {"small": small_problem, "mixed": mixed_k_problem}[which]
    monkeypatch.setenv("RBA_HX_THREADS", "512")
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    g, o = _pair(prob, dtype, implicit_q=1)
    assert g.linearize() == 0 and o.linearize() == 0
    rng = np.random.default_rng(5)
    for lam in (1e-4, 1e-6):
        o.set_pose_damping(lam)
        o.stage2(lam, o.pose_scaling() if lam == 1e-4 else None)
        g.stage2(lam)
        for _ in range(2):
            x = rng.uniform(-1, 1, 9 * prob.n_cams).astype(dtype)
            assert rel_err(g.right_multiply(x), o.right_multiply(x)) < TOL[dtype]
    g2, o2 = _pair(prob, dtype, implicit_q=1)
    assert g2.linearize() == 0 and o2.linearize() == 0
    ig, cg = g2.solve(1e-4)
    io, co = o2.solve(1e-4)
    _assert_increment(prob, dtype, o2, 1e-4, ig, cg, io, co, 10 * TOL[dtype])
