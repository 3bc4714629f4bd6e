"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package `rootba_amd`.
See oracle/rootba_oracle.hpp for the parity status (pinned against the reference's own code
through oracle/_ref, see oracle/ref.py; third-party arithmetic unpinned).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class Options(C.Structure):
    _fields_ = [
        ("use_householder", C.c_int),
        ("use_valid_projections_only", C.c_int),
        ("robust_norm", C.c_int),
        ("huber_parameter", C.c_double),
        ("jacobi_scaling_eps", C.c_double),
        ("preconditioner_type", C.c_int),
        ("reduction_alg", C.c_int),
        ("power_order", C.c_int),
        ("min_cg_it", C.c_int),
        ("max_cg_it", C.c_int),
        ("eta", C.c_double),
        ("num_threads", C.c_int),
        ("max_num_iterations", C.c_int),
        ("min_relative_decrease", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("function_tolerance", C.c_double),
        ("initial_vee", C.c_double),
        ("vee_factor", C.c_double),
        ("optimized_cost", C.c_int),
        ("staged_execution", C.c_int),
        ("implicit_q", C.c_int),  # product-only switch; the oracle has one operator
        ("solver_type", C.c_int),  # 0 SQUARE_ROOT, 1 SCHUR_COMPLEMENT, 2 SCHUR_COMPLEMENT matrix-free (large-problem referee; oracle only)
        ("explicit_after", C.c_int),  # product-only
    ]


class ResidualInfo(C.Structure):
    _fields_ = [
        ("all_num_obs", C.c_int),
        ("all_error", C.c_double),
        ("all_residual_sum", C.c_double),
        ("valid_num_obs", C.c_int),
        ("valid_error", C.c_double),
        ("valid_residual_sum", C.c_double),
        ("is_numerically_valid", C.c_int),
    ]


class CgSummary(C.Structure):
    _fields_ = [("termination_type", C.c_int), ("num_iterations", C.c_int)]


class LmIteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int),
        ("step_is_valid", C.c_int),
        ("step_is_successful", C.c_int),
        ("cg_iterations", C.c_int),
        ("cg_termination", C.c_int),
        ("cost", C.c_double),
        ("cost_valid", C.c_double),
        ("lambda_", C.c_double),
        ("relative_decrease", C.c_double),
        ("l_diff", C.c_double),
        ("inc_norm", C.c_double),
        ("iteration_time", C.c_double),
        ("stage1_time", C.c_double),
        ("stage2_time", C.c_double),
        ("precond_time", C.c_double),
        ("pcg_time", C.c_double),
        ("backsub_time", C.c_double),
        ("residual_time", C.c_double),
        ("num_obs", C.c_int),
        ("num_obs_valid", C.c_int),
        ("residual_sum", C.c_double),
        ("residual_sum_valid", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with the committed Makefile."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("rootba_oracle.hpp", "oracle_capi.cpp", "Makefile")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        assert _lib.orc_sizeof_lm_iteration() == C.sizeof(LmIteration)
    return _lib


def default_options(**kw) -> Options:
    o = Options()
    lib().orc_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class Oracle:
    """One `LinearizorQR<Scalar>` worth of state on the CPU (oracle)."""

    def __init__(self, prob, dtype=np.float32, options: Options | None = None):
        self.dtype = np.dtype(dtype)
        self.suf = "f32" if self.dtype == np.float32 else "f64"
        self.ct = C.c_float if self.dtype == np.float32 else C.c_double
        self.L = lib()
        self.n_cams, self.n_lms, self.n_obs = prob.n_cams, prob.n_lms, prob.n_obs
        self.options = options or default_options()
        off = np.ascontiguousarray(prob.lm_obs_offsets, dtype=np.int64)
        cam = np.ascontiguousarray(prob.obs_cam_idx, dtype=np.int32)
        xy = np.ascontiguousarray(prob.obs_xy, dtype=self.dtype)
        f = self._fn("create")
        f.restype = C.c_void_p
        self.h = C.c_void_p(f(C.c_int(self.n_cams), C.c_int(self.n_lms), _ptr(off, C.c_int64),
                              _ptr(cam, C.c_int32), _ptr(xy, self.ct), C.byref(self.options)))
        self.set_state(prob.cams, prob.lms)

    def _fn(self, name):
        return getattr(self.L, f"orc_{name}_{self.suf}")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self._fn("destroy")(self.h)
                self.h = None
        except Exception:
            pass

    def _vec(self, n):
        return np.zeros(n, dtype=self.dtype)

    def _in(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    def num_threads(self) -> int:
        return int(self._fn("num_threads")(self.h))

    def set_state(self, cams, lms):
        c, l = self._in(cams).ravel(), self._in(lms).ravel()
        assert c.size == 10 * self.n_cams and l.size == 3 * self.n_lms
        self._fn("set_state")(self.h, _ptr(c, self.ct), _ptr(l, self.ct))

    def get_state(self):
        c, l = self._vec(10 * self.n_cams), self._vec(3 * self.n_lms)
        self._fn("get_state")(self.h, _ptr(c, self.ct), _ptr(l, self.ct))
        return c.reshape(-1, 10), l.reshape(-1, 3)

    def backup(self):
        self._fn("backup")(self.h)

    def restore(self):
        self._fn("restore")(self.h)

    def compute_error(self) -> ResidualInfo:
        ri = ResidualInfo()
        self._fn("compute_error")(self.h, C.byref(ri))
        return ri

    def stage1(self, jacobi_blocks: bool = False):
        d = self._vec(9 * self.n_cams)
        blocks = self._vec(81 * self.n_cams) if jacobi_blocks else None
        rc = self._fn("stage1")(self.h, _ptr(d, self.ct),
                                _ptr(blocks, self.ct) if jacobi_blocks else None)
        return rc, d, blocks

    def set_pose_damping(self, lam):
        self._fn("set_pose_damping")(self.h, self.ct(lam))

    def stage2(self, lam, jacobian_scaling=None, blocks: bool = True):
        b = self._vec(9 * self.n_cams)
        bl = self._vec(81 * self.n_cams) if blocks else None
        js = self._in(jacobian_scaling) if jacobian_scaling is not None else None
        self._fn("stage2")(self.h, self.ct(lam), _ptr(js, self.ct) if js is not None else None,
                           _ptr(b, self.ct), _ptr(bl, self.ct) if blocks else None)
        return b, (bl.reshape(-1, 9, 9) if blocks else None)

    def right_multiply(self, x):
        x = self._in(x)
        y = self._vec(9 * self.n_cams)
        self._fn("right_multiply")(self.h, _ptr(x, self.ct), _ptr(y, self.ct))
        return y

    def back_substitute(self, pose_inc):
        f = self._fn("back_substitute")
        f.restype = self.ct
        x = self._in(pose_inc)
        return float(f(self.h, _ptr(x, self.ct)))

    def linearize(self) -> int:
        return int(self._fn("linearize")(self.h))

    def solve(self, lam):
        if self.options.solver_type == 2 and self.options.preconditioner_type != 1:
            raise ValueError("solver_type 2 (matrix-free Schur complement, the large-problem referee): SCHUR_JACOBI only")
        inc = self._vec(9 * self.n_cams)
        cg = CgSummary()
        self._fn("solve")(self.h, self.ct(lam), _ptr(inc, self.ct), C.byref(cg))
        return inc, cg

    def apply(self, inc):
        f = self._fn("apply")
        f.restype = self.ct
        x = self._in(inc)
        return float(f(self.h, _ptr(x, self.ct)))

    def optimize_lm(self, max_rows: int = 256):
        log = (LmIteration * max_rows)()
        term = C.c_int(0)
        n = self._fn("optimize_lm")(self.h, log, C.c_int(max_rows), C.byref(term))
        return [log[i] for i in range(min(n, max_rows))], term.value

    def block(self, l: int):
        r, c, li = C.c_int(), C.c_int(), C.c_int()
        self._fn("block_shape")(self.h, C.c_int(l), C.byref(r), C.byref(c), C.byref(li))
        out = self._vec(r.value * c.value)
        self._fn("get_block")(self.h, C.c_int(l), _ptr(out, self.ct))
        return out.reshape(r.value, c.value), li.value

    def jl_col_scale(self):
        out = self._vec(3 * self.n_lms)
        self._fn("get_jl_col_scale")(self.h, _ptr(out, self.ct))
        return out.reshape(-1, 3)

    def pose_scaling(self):
        out = self._vec(9 * self.n_cams)
        self._fn("get_pose_scaling")(self.h, _ptr(out, self.ct))
        return out

    def last_b(self):
        out = self._vec(9 * self.n_cams)
        self._fn("get_last_b")(self.h, _ptr(out, self.ct))
        return out

    def precond_blocks(self):
        out = self._vec(81 * self.n_cams)
        self._fn("get_precond_blocks")(self.h, _ptr(out, self.ct))
        return out.reshape(-1, 9, 9)

    def power_precond(self, lam, b):
        """PowerSCPreconditioner::solve_assign at the current linearisation point."""
        x = self._vec(9 * self.n_cams)
        bb = self._in(b)
        self._fn("power_precond")(self.h, self.ct(lam), _ptr(bb, self.ct), _ptr(x, self.ct))
        return x

    def sc_build(self, lam, pose_lambda=0.0, pose_scaling=None, want_H=True):
        n = 9 * self.n_cams
        H = self._vec(n * n) if want_H else None
        b, d = self._vec(n), self._vec(n)
        ps = self._in(pose_scaling) if pose_scaling is not None else None
        self._fn("sc_build")(self.h, self.ct(lam), self.ct(pose_lambda),
                             _ptr(ps, self.ct) if ps is not None else None,
                             _ptr(H, self.ct) if want_H else None, _ptr(b, self.ct),
                             _ptr(d, self.ct))
        return (H.reshape(n, n) if want_H else None), b, d

    def sc_back_substitute(self, lam, pose_scaling, pose_inc):
        f = self._fn("sc_back_substitute")
        f.restype = self.ct
        ps = self._in(pose_scaling) if pose_scaling is not None else None
        x = self._in(pose_inc)
        return float(f(self.h, self.ct(lam), _ptr(ps, self.ct) if ps is not None else None,
                       _ptr(x, self.ct)))


def linearize_point(obs, p_w, cam, dtype=np.float64, ignore_validity_check=True):
    dt = np.dtype(dtype)
    suf, ct = ("f32", C.c_float) if dt == np.float32 else ("f64", C.c_double)
    a = [np.ascontiguousarray(v, dtype=dt) for v in (obs, p_w, cam)]
    res, Jp, Ji, Jl = (np.zeros(n, dtype=dt) for n in (2, 12, 6, 6))
    valid = getattr(lib(), f"orc_linearize_point_{suf}")(
        _ptr(a[0], ct), _ptr(a[1], ct), _ptr(a[2], ct), C.c_int(int(ignore_validity_check)),
        _ptr(res, ct), _ptr(Jp, ct), _ptr(Ji, ct), _ptr(Jl, ct))
    return bool(valid), res, Jp.reshape(2, 6), Ji.reshape(2, 3), Jl.reshape(2, 3)


def apply_inc_camera(cam, inc9, dtype=np.float64):
    dt = np.dtype(dtype)
    suf, ct = ("f32", C.c_float) if dt == np.float32 else ("f64", C.c_double)
    c = np.array(cam, dtype=dt).copy()
    i = np.ascontiguousarray(inc9, dtype=dt)
    getattr(lib(), f"orc_apply_inc_camera_{suf}")(_ptr(c, ct), _ptr(i, ct))
    return c
