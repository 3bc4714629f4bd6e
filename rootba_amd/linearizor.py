"""`LinearizorHIP`: Python mirror of the reference's `Linearizor<Scalar>`
interface (reference src/rootba/solver/linearizor.hpp:48-83) bound to the
MI355X-native C ABI (include/rootba_hip.h).

Same call sequence as the reference's LM loop
(src/rootba/solver/bal_bundle_adjustment.cpp:291-521):
``start_iteration -> compute_error -> linearize -> {solve(lambda) -> apply(inc)
-> compute_error} x (1 + backtracks) -> finish_iteration``; `optimize_lm`
runs that loop inside the library (host C++), as `optimize_lm_ours` does.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .problem import BalProblem


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class LinearizorHIP:
    """One `LinearizorQR<Scalar>` worth of state, resident on one MI355X."""

    def __init__(self, prob: BalProblem, dtype=np.float32, options: L.RbaOptions | None = None,
                 device: int = 0, devices=None):
        """`devices`: a list of device indices -> ONE handle over several GPUs of this process (rba_create_sharded: the
        library shards the landmarks itself; an index may repeat, e.g. [0, 0] on a single-GPU box)."""
        # "mixed" (RBA_MIXED): double state / observations / costs, float linear algebra; camera-sized
        # vectors cross the boundary as float32, the state as float64
        self.mixed = isinstance(dtype, str) and dtype == "mixed"
        self.dtype = np.dtype(np.float32 if self.mixed else dtype)
        self.state_dtype = np.dtype(np.float64) if self.mixed else self.dtype
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32, float64 or 'mixed'")
        self.lib = L.lib()
        if L.device_count() <= 0:
            raise RuntimeError("LinearizorHIP needs a HIP device (no CPU fallback): " + L.last_error())
        self.options = options or L.default_options()
        self.n_cams, self.n_lms, self.n_obs = prob.n_cams, prob.n_lms, prob.n_obs
        off = np.ascontiguousarray(prob.lm_obs_offsets, dtype=np.int64)
        cam = np.ascontiguousarray(prob.obs_cam_idx, dtype=np.int32)
        xy = np.ascontiguousarray(prob.obs_xy, dtype=self.state_dtype)
        self.h = C.c_void_p()
        dt_code = C.c_int(L.RBA_MIXED if self.mixed else L.RBA_F32 if self.dtype == np.float32 else L.RBA_F64)
        if devices is not None:
            ids = (C.c_int * len(devices))(*devices)
            L.check(self.lib.rba_create_sharded(dt_code, C.c_int(len(devices)), ids, C.c_int32(self.n_cams),
                                                C.c_int32(self.n_lms), _ptr(off), _ptr(cam), _ptr(xy),
                                                C.byref(self.options), C.byref(self.h)), "rba_create_sharded")
        else:
            L.check(self.lib.rba_create(dt_code, C.c_int(device), C.c_int32(self.n_cams), C.c_int32(self.n_lms),
                                        _ptr(off), _ptr(cam), _ptr(xy), C.byref(self.options), C.byref(self.h)),
                    "rba_create")
        self.set_state(prob.cams, prob.lms)
        self.it_summary = None

    # -- factory with the reference's name -------------------------------------
    @classmethod
    def create(cls, bal_problem: BalProblem, options: L.RbaOptions | None = None, dtype=np.float32,
               device: int = 0) -> "LinearizorHIP":
        """`Linearizor<Scalar>::create` (reference src/rootba/solver/linearizor.cpp:133-150)."""
        return cls(bal_problem, dtype, options, device)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.rba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _vec(self, n):
        return np.zeros(n, dtype=self.dtype)

    def _in(self, a, n):
        a = np.ascontiguousarray(a, dtype=self.dtype).ravel()
        if a.size != n:
            raise ValueError(f"expected {n} scalars, got {a.size}")
        return a

    def shard_ranges(self):
        """Landmark ranges of a sharded handle: device r holds landmarks cuts[r] .. cuts[r + 1] - 1."""
        n = C.c_int(0)
        L.check(self.lib.rba_get_shard_ranges(self.h, C.byref(n), None, C.c_int(0)), "rba_get_shard_ranges")
        cuts = (C.c_int32 * (n.value + 1))()
        L.check(self.lib.rba_get_shard_ranges(self.h, C.byref(n), cuts, C.c_int(n.value + 1)), "rba_get_shard_ranges")
        return list(cuts) if n.value > 1 else [0, self.n_lms]

    # -- multi-GPU ----------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        L.check(L.lib().rba_comm_unique_id(buf), "rba_comm_unique_id")
        return buf.raw

    def comm_init(self, rank: int, nranks: int, unique_id: bytes):
        L.check(self.lib.rba_comm_init(self.h, C.c_int(rank), C.c_int(nranks),
                                       C.c_char_p(unique_id)), "rba_comm_init")

    def comm_init_callback(self, rank: int, nranks: int, allreduce):
        """`allreduce(array, op)` all-reduces a numpy array in place (op: 'sum' | 'max')."""
        np_dt = {0: np.float32, 1: np.float64, 2: np.int32}

        def _cb(_ctx, buf, count, dtype, op):
            try:
                dt = np.dtype(np_dt[dtype])
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_char)), shape=(count * dt.itemsize,)).view(dt)
                allreduce(arr, "max" if op == 1 else "sum")
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        self._allreduce_cb = L.ALLREDUCE_FN(_cb)  # keep alive
        L.check(self.lib.rba_comm_init_callback(self.h, C.c_int(rank), C.c_int(nranks), self._allreduce_cb,
                                                None), "rba_comm_init_callback")

    def comm_info(self) -> dict:
        r, n, t = C.c_int(0), C.c_int(0), C.c_int(0)
        L.check(self.lib.rba_comm_info(self.h, C.byref(r), C.byref(n), C.byref(t)), "rba_comm_info")
        return {"rank": r.value, "nranks": n.value, "transport": ("none", "rccl", "callback")[t.value]}

    def comm_stats(self) -> dict:
        c, b, s = C.c_int64(0), C.c_int64(0), C.c_double(0)
        L.check(self.lib.rba_get_comm_stats(self.h, C.byref(c), C.byref(b), C.byref(s)), "rba_get_comm_stats")
        return {"calls": c.value, "bytes": b.value, "seconds": s.value}

    # -- BalProblem state -----------------------------------------------------------
    def set_state(self, cams, lms):
        c = np.ascontiguousarray(cams, dtype=self.state_dtype).ravel()
        l = np.ascontiguousarray(lms, dtype=self.state_dtype).ravel()
        if c.size != 10 * self.n_cams or l.size != 3 * self.n_lms:
            raise ValueError("state arrays have the wrong size")
        L.check(self.lib.rba_set_state(self.h, _ptr(c), _ptr(l)), "rba_set_state")

    def get_state(self):
        c = np.zeros(10 * self.n_cams, dtype=self.state_dtype)
        l = np.zeros(3 * self.n_lms, dtype=self.state_dtype)
        L.check(self.lib.rba_get_state(self.h, _ptr(c), _ptr(l)), "rba_get_state")
        return c.reshape(-1, 10), l.reshape(-1, 3)

    def backup(self):
        L.check(self.lib.rba_backup(self.h), "rba_backup")

    def restore(self):
        L.check(self.lib.rba_restore(self.h), "rba_restore")

    # -- Linearizor interface ---------------------------------------------------------
    def start_iteration(self, it_summary=None):
        self.it_summary = it_summary

    def finish_iteration(self):
        self.it_summary = None

    def compute_error(self) -> L.RbaResidualInfo:
        ri = L.RbaResidualInfo()
        L.check(self.lib.rba_compute_error(self.h, C.byref(ri)), "rba_compute_error")
        return ri

    def linearize(self, want_jp_diag2: bool = False):
        """Returns status (0 ok, 1 numerical failure) [and Jp_diag2]."""
        d = self._vec(9 * self.n_cams) if want_jp_diag2 else None
        st = L.check(self.lib.rba_linearize(self.h, _ptr(d) if want_jp_diag2 else None),
                     "rba_linearize", allow_numerical_failure=True)
        return (st, d) if want_jp_diag2 else st

    def solve(self, lam: float):
        inc = self._vec(9 * self.n_cams)
        cg = L.RbaCgSummary()
        L.check(self.lib.rba_solve(self.h, C.c_double(lam), _ptr(inc), C.byref(cg)), "rba_solve")
        return inc, cg

    def apply(self, inc) -> float:
        x = self._in(inc, 9 * self.n_cams)
        l_diff = C.c_double(0)
        L.check(self.lib.rba_apply(self.h, _ptr(x), C.byref(l_diff)), "rba_apply",
                allow_numerical_failure=True)
        return l_diff.value

    # -- LinearizationQR pieces the reference's tests call directly --------------------
    def stage2(self, lam: float, blocks: bool = True):
        b = self._vec(9 * self.n_cams)
        bl = self._vec(81 * self.n_cams) if blocks else None
        L.check(self.lib.rba_stage2(self.h, C.c_double(lam), _ptr(b), _ptr(bl) if blocks else None),
                "rba_stage2")
        return b, (bl.reshape(-1, 9, 9) if blocks else None)

    def right_multiply(self, x):
        xi = self._in(x, 9 * self.n_cams)
        y = self._vec(9 * self.n_cams)
        L.check(self.lib.rba_right_multiply(self.h, _ptr(xi), _ptr(y)), "rba_right_multiply")
        return y

    def right_multiply_explicit(self, x):
        """The same product through the explicitly assembled reduced matrix (`explicit_after`)."""
        xi = self._in(x, 9 * self.n_cams)
        y = self._vec(9 * self.n_cams)
        L.check(self.lib.rba_right_multiply_explicit(self.h, _ptr(xi), _ptr(y)), "rba_right_multiply_explicit")
        return y

    def back_substitute(self, inc) -> float:
        x = self._in(inc, 9 * self.n_cams)
        l_diff = C.c_double(0)
        L.check(self.lib.rba_back_substitute(self.h, _ptr(x), C.byref(l_diff)),
                "rba_back_substitute", allow_numerical_failure=True)
        return l_diff.value

    # -- optimize_lm_ours ------------------------------------------------------------------
    def optimize_lm(self, max_rows: int = 512):
        log = (L.RbaLmIteration * max_rows)()
        n, term = C.c_int(0), C.c_int(0)
        L.check(self.lib.rba_optimize_lm(self.h, log, C.c_int(max_rows), C.byref(n), C.byref(term)),
                "rba_optimize_lm")
        return [log[i] for i in range(min(n.value, max_rows))], term.value

    def lm_begin(self):
        L.check(self.lib.rba_lm_begin(self.h), "rba_lm_begin")

    def lm_step(self):
        """One LM iteration. Returns (row, more)."""
        row, more = L.RbaLmIteration(), C.c_int(0)
        L.check(self.lib.rba_lm_step(self.h, C.byref(row), C.byref(more)), "rba_lm_step")
        return row, bool(more.value)

    def lm_termination(self) -> int:
        t = C.c_int(0)
        L.check(self.lib.rba_lm_termination(self.h, C.byref(t)), "rba_lm_termination")
        return t.value

    def synchronize(self):
        L.check(self.lib.rba_synchronize(self.h), "rba_synchronize")

    # -- introspection ------------------------------------------------------------------------
    def substage_timings(self) -> L.RbaSubstageTimings:
        """Sub-stage timers of the reference's unstaged execution (options.staged_execution = 0)."""
        t = L.RbaSubstageTimings()
        L.check(self.lib.rba_get_substage_timings(self.h, C.byref(t)), "rba_get_substage_timings")
        return t

    def timings(self) -> L.RbaIterTimings:
        t = L.RbaIterTimings()
        L.check(self.lib.rba_get_timings(self.h, C.byref(t)), "rba_get_timings")
        return t

    def jl_col_scale(self):
        out = self._vec(3 * self.n_lms)
        L.check(self.lib.rba_get_jl_col_scale(self.h, _ptr(out)), "rba_get_jl_col_scale")
        return out.reshape(-1, 3)

    def pose_scaling(self):
        out = self._vec(9 * self.n_cams)
        L.check(self.lib.rba_get_pose_scaling(self.h, _ptr(out)), "rba_get_pose_scaling")
        return out

    def landmark_R(self, damped: bool = False):
        R, q = self._vec(6 * self.n_lms), self._vec(3 * self.n_lms)
        L.check(self.lib.rba_get_landmark_R(self.h, C.c_int(int(damped)), _ptr(R), _ptr(q)),
                "rba_get_landmark_R")
        return R.reshape(-1, 6), q.reshape(-1, 3)

    def landmark_q2tr_norm(self):
        out = self._vec(self.n_lms)
        L.check(self.lib.rba_get_landmark_q2tr_norm(self.h, _ptr(out)), "rba_get_landmark_q2tr_norm")
        return out

    def byte_model(self) -> dict:
        m = L.RbaByteModel()
        L.check(self.lib.rba_get_byte_model(self.h, C.byref(m)), "rba_get_byte_model")
        return {n: getattr(m, n) for n, _ in L.RbaByteModel._fields_}

    def pcg_counters(self) -> dict:
        m = L.RbaPcgCounters()
        L.check(self.lib.rba_get_pcg_counters(self.h, C.byref(m)), "rba_get_pcg_counters")
        return {n: getattr(m, n) for n, _ in L.RbaPcgCounters._fields_}

    def reduced_matrix_info(self) -> dict:
        m = L.RbaReducedMatrixInfo()
        L.check(self.lib.rba_get_reduced_matrix_info(self.h, C.byref(m)), "rba_get_reduced_matrix_info")
        return {n: getattr(m, n) for n, _ in m._fields_}

    def problem_stats(self) -> dict:
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L.check(self.lib.rba_get_problem_stats(self.h, C.byref(a), C.byref(b), C.byref(c)),
                "rba_get_problem_stats")
        return {"block_storage_bytes": a.value, "hx_bytes": b.value, "hx_flops": c.value}
