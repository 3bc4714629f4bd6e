"""ctypes front-end of oracle/_ref/librootba_ref.so: the REFERENCE'S OWN solver sources, compiled by
oracle/build_ref.sh against the third-party stand-ins of oracle/ref_shims/.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and tests/golden/make_ref_golden.py) to pin the restated
oracle (oracle/rootba_oracle.hpp) - never by the product package `rootba_amd`.
`Reference` has the interface of `oracle.Oracle`, so a test can run the same code against both.
What this build does and does not pin is spelled out in the header of oracle/ref_driver.cpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .oracle import CgSummary, LmIteration, Options, ResidualInfo, _ptr

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "librootba_ref.so")
_BINDING_PATH = os.path.join(_HERE, "_ref", "librootba_ref_binding.so")
_MOCK_PATH = os.path.join(_HERE, "_ref", "librootba_hip_mock.so")
REFERENCE_ROOT = os.environ.get("REF", "/root/reference")


def build(force: bool = False) -> str | None:
    """Run oracle/build_ref.sh when the reference tree is present; returns the library path or None."""
    have_ref = os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "rootba"))
    if have_ref:
        root = os.path.dirname(_HERE)
        deps = [os.path.join(_HERE, "ref_driver.cpp"), os.path.join(_HERE, "build_ref.sh"),
                os.path.join(_HERE, "mock_rootba_hip.cpp"), os.path.join(_HERE, "rootba_oracle.hpp"),
                os.path.join(root, "include", "rootba_hip.h"),
                os.path.join(root, "integration", "linearizor_factory_hip.cpp"),
                os.path.join(root, "integration", "rootba", "solver", "linearizor_hip.hpp")]
        for root, _, files in os.walk(os.path.join(_HERE, "ref_shims")):
            deps += [os.path.join(root, f) for f in files]
        newest = min((os.path.getmtime(p) for p in (_LIB_PATH, _BINDING_PATH, _MOCK_PATH) if os.path.exists(p)), default=0)
        stale = force or not all(os.path.exists(p) for p in (_LIB_PATH, _BINDING_PATH, _MOCK_PATH)) or any(
            os.path.getmtime(d) > newest for d in deps)
        if stale:
            subprocess.check_call(["sh", os.path.join(_HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)
    return _LIB_PATH if os.path.exists(_LIB_PATH) else None


def available() -> bool:
    return build() is not None


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref/librootba_ref.so is not built and /root/reference is not present")
        _lib = C.CDLL(path)
        assert _lib.ref_sizeof_lm_iteration() == C.sizeof(LmIteration)
        assert _lib.ref_sizeof_options() == C.sizeof(Options)
    return _lib


_binding = {}


def binding_available() -> bool:
    return build() is not None and os.path.exists(_BINDING_PATH)


def binding_lib(provider: str):
    """oracle/_ref/librootba_ref_binding.so: the reference's sources + the reference-side binding of the HIP
    library (integration/rootba/solver/linearizor_hip.hpp) behind the wrapped factory. Its rba_* entry points
    are resolved from `provider`, loaded first with RTLD_GLOBAL:
      "hip"  - rootba_amd/librootba_hip.so (needs a GPU),
      "mock" - oracle/_ref/librootba_hip_mock.so, the oracle-backed test double of the C ABI (no GPU).
    One provider per process (the first global definition of rba_* wins)."""
    if _binding and provider not in _binding:
        raise RuntimeError(f"this process already bound the C ABI to {list(_binding)}")
    if provider not in _binding:
        build()
        if provider == "mock":
            path = _MOCK_PATH
        else:
            from rootba_amd import _lib as product  # (test sessions with RBA_EMU=1 point LIB_PATH at the CPU harness build)
            path = product.LIB_PATH
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        lib_ = C.CDLL(_BINDING_PATH)
        assert lib_.ref_sizeof_lm_iteration() == C.sizeof(LmIteration) and lib_.ref_sizeof_options() == C.sizeof(Options)
        _binding[provider] = lib_
    return _binding[provider]


def default_options(**kw) -> Options:
    """The defaults of the reference's own SolverOptions declaration (src/rootba/bal/solver_options.hpp)."""
    o = Options()
    lib().ref_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class Reference:
    """`Linearizor<Scalar>` / `LinearizationQR<Scalar, 9>` of the reference on one problem."""

    def __init__(self, prob, dtype=np.float32, options: Options | None = None, library=None):
        self.dtype = np.dtype(dtype)
        self.suf = "f32" if self.dtype == np.float32 else "f64"
        self.ct = C.c_float if self.dtype == np.float32 else C.c_double
        self.L = library or lib()
        self.n_cams, self.n_lms, self.n_obs = prob.n_cams, prob.n_lms, prob.n_obs
        self.options = options or default_options()
        off = np.ascontiguousarray(prob.lm_obs_offsets, dtype=np.int64)
        cam = np.ascontiguousarray(prob.obs_cam_idx, dtype=np.int32)
        xy = np.ascontiguousarray(prob.obs_xy, dtype=self.dtype)
        f = self._fn("create")
        f.restype = C.c_void_p
        h = f(C.c_int(self.n_cams), C.c_int(self.n_lms), _ptr(off, C.c_int64),
              _ptr(cam, C.c_int32), _ptr(xy, self.ct), C.byref(self.options))
        if not h:
            raise ValueError("not a configuration of the reference (use_valid_projections_only must equal "
                             "optimized_cost != ERROR, linearizor_qr.cpp:58-68)")
        self.h = C.c_void_p(h)
        self.set_state(prob.cams, prob.lms)

    def _fn(self, name):
        return getattr(self.L, f"ref_{name}_{self.suf}")

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

    # ---- LinearizationQR, stage by stage (square-root solver) ------------------------------------
    def stage1(self, jacobi_blocks: bool = False):
        d = self._vec(9 * self.n_cams)
        blocks = self._vec(81 * self.n_cams) if jacobi_blocks else None
        rc = self._fn("stage1")(self.h, _ptr(d, self.ct), _ptr(blocks, self.ct) if jacobi_blocks else None)
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

    # ---- Linearizor (the solver type of the options, through the reference's factory) -------------
    def linearize(self) -> int:
        return int(self._fn("linearize")(self.h))

    def solve(self, lam):
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
        """bundle_adjust_manual. Rows carry what the reference's IterationSummary holds (no l_diff /
        inc_norm); `lambda_` of a row is 1 / trust_region_radius, i.e. the damping of the NEXT solve."""
        log = (LmIteration * max_rows)()
        term = C.c_int(0)
        n = self._fn("optimize_lm")(self.h, log, C.c_int(max_rows), C.byref(term))
        return [log[i] for i in range(min(n, max_rows))], term.value


class ReferenceOnHip(Reference):
    """The REFERENCE'S Linearizor interface and LM loop on top of the C ABI of include/rootba_hip.h: the wrapped
    factory (integration/linearizor_factory_hip.cpp) returns rootba::LinearizorHIP for the square-root solver while
    ROOTBA_LINEARIZOR=hip is set. `linearize / solve / apply / compute_error / optimize_lm` then run
    reference code (Linearizor calls, optimize_lm_ours) -> binding -> rba_* of `provider` ("hip" | "mock")."""

    def __init__(self, prob, dtype=np.float32, options: Options | None = None, provider="hip"):
        super().__init__(prob, dtype, options, library=binding_lib(provider))

    def _with_hip(self, f, *a):
        old = os.environ.get("ROOTBA_LINEARIZOR")
        os.environ["ROOTBA_LINEARIZOR"] = "hip"
        try:
            return f(*a)
        finally:
            if old is None:
                del os.environ["ROOTBA_LINEARIZOR"]
            else:
                os.environ["ROOTBA_LINEARIZOR"] = old

    def linearize(self):
        return self._with_hip(super().linearize)  # (creates the linearizor on first use)

    def compute_error(self) -> ResidualInfo:
        """Linearizor::compute_error of the binding (rba_compute_error on the uploaded host state)."""
        ri = ResidualInfo()
        self._with_hip(self._fn("linearizor_compute_error"), self.h, C.byref(ri))
        return ri

    def optimize_lm(self, max_rows: int = 256):
        return self._with_hip(super().optimize_lm, max_rows)


def linearize_point(obs, p_w, cam, dtype=np.float64, ignore_validity_check=True):
    dt = np.dtype(dtype)
    suf, ct = ("f32", C.c_float) if dt == np.float32 else ("f64", C.c_double)
    a = [np.ascontiguousarray(v, dtype=dt) for v in (obs, p_w, cam)]
    res, Jp, Ji, Jl = (np.zeros(n, dtype=dt) for n in (2, 12, 6, 6))
    valid = getattr(lib(), f"ref_linearize_point_{suf}")(
        _ptr(a[0], ct), _ptr(a[1], ct), _ptr(a[2], ct), C.c_int(int(ignore_validity_check)),
        _ptr(res, ct), _ptr(Jp, ct), _ptr(Ji, ct), _ptr(Jl, ct))
    return bool(valid), res, Jp.reshape(2, 6), Ji.reshape(2, 3), Jl.reshape(2, 3)


def apply_inc_camera(cam, inc9, dtype=np.float64):
    dt = np.dtype(dtype)
    suf, ct = ("f32", C.c_float) if dt == np.float32 else ("f64", C.c_double)
    c = np.array(cam, dtype=dt).copy()
    i = np.ascontiguousarray(inc9, dtype=dt)
    getattr(lib(), f"ref_apply_inc_camera_{suf}")(_ptr(c, ct), _ptr(i, ct))
    return c


def load_bal(path, normalize=True, normalization_scale=100.0, rotation_sigma=0.0, translation_sigma=0.0,
             point_sigma=0.0, seed=-1, init_depth_threshold=0.0, input_type="BAL"):
    """load_normalized_bal_problem<double> of the reference (input_type BAL, BUNDLER or AUTO = by file name).
    Returns a dict of arrays."""
    L = lib()
    L.ref_load_bal.restype = C.c_void_p
    h = L.ref_load_bal(os.fsencode(path), C.c_int(int(normalize)), C.c_double(normalization_scale),
                       C.c_double(rotation_sigma), C.c_double(translation_sigma), C.c_double(point_sigma),
                       C.c_int(seed), C.c_double(init_depth_threshold),
                       C.c_int({"AUTO": 0, "BAL": 2, "BUNDLER": 3}[input_type]))
    if not h:
        raise RuntimeError(f"the reference's loader rejected {path}")
    h = C.c_void_p(h)
    nc, nl, no = C.c_int(), C.c_int(), C.c_int64()
    L.ref_loaded_sizes(h, C.byref(nc), C.byref(nl), C.byref(no))
    cams, lms = np.zeros((nc.value, 10)), np.zeros((nl.value, 3))
    off, cam = np.zeros(nl.value + 1, dtype=np.int64), np.zeros(no.value, dtype=np.int32)
    xy = np.zeros((no.value, 2))
    L.ref_loaded_get(h, _ptr(cams, C.c_double), _ptr(lms, C.c_double), _ptr(off, C.c_int64),
                     _ptr(cam, C.c_int32), _ptr(xy, C.c_double))
    sparsity, kmax = C.c_double(), C.c_int()
    L.ref_loaded_stats(h, C.byref(sparsity), C.byref(kmax))
    L.ref_loaded_destroy(h)
    return dict(cams=cams, lms=lms, lm_obs_offsets=off, obs_cam_idx=cam, obs_xy=xy, rcs_sparsity=sparsity.value,
                max_obs_per_lm=kmax.value)
