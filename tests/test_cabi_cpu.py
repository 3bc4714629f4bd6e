"""CPU-side checks of the product boundary (no GPU needed):
the C-ABI library loads, exports every symbol include/rootba_hip.h declares,
its POD structs match the header, and it refuses to run without a GPU (there is
no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rootba_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rootba_hip.h")


@pytest.fixture(scope="module")
def lib():
    from rootba_amd import build
    build.build()
    return L.lib()


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rba_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    declared = _declared_functions()
    assert len(declared) >= 25
    missing = [f for f in declared if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(L.EXPORTS) == declared


def test_default_options_match_reference_defaults(lib):
    # reference examples/config/rootba_config_default.toml
    o = L.default_options()
    assert (o.use_householder, o.preconditioner_type, o.max_cg_it, o.min_cg_it) == (1, 1, 500, 0)
    assert (o.max_num_iterations, o.robust_norm, o.optimized_cost, o.staged_execution) == (20, 0, 0, 1)
    assert o.eta == 0.1 and o.jacobi_scaling_eps == 0.0 and o.function_tolerance == 1e-6
    assert o.initial_trust_region_radius == 1e4 and o.min_trust_region_radius == 1e-32
    assert o.max_trust_region_radius == 1e16 and o.initial_vee == 2.0 and o.vee_factor == 2.0


def test_default_options_are_the_reference_builds(lib):
    """rba_default_options (include/rootba_hip.h) field by field against the defaults of the reference's OWN
    SolverOptions declaration, read out of the reference build (oracle/_ref, see oracle/ref_driver.cpp)."""
    from oracle import ref as R
    if not R.available():
        pytest.skip("oracle/_ref is not built and /root/reference is not present")
    a, b = L.default_options(), R.default_options()
    for name, _ in L.RbaOptions._fields_:
        if name in ("implicit_q", "explicit_after", "num_threads"):  # product-only switches / host thread count
            continue
        assert getattr(a, name) == getattr(b, name), name


def test_option_struct_layout_matches_oracle_mirror():
    from oracle import oracle as O
    assert [f[0] for f in L.RbaOptions._fields_] == [f[0] for f in O.Options._fields_]
    assert C.sizeof(L.RbaOptions) == C.sizeof(O.Options)
    assert [f[0] for f in L.RbaLmIteration._fields_] == [f[0] for f in O.LmIteration._fields_]
    assert C.sizeof(L.RbaLmIteration) == C.sizeof(O.LmIteration)


@pytest.mark.skipif(L.device_count() > 0 if os.path.exists(L.LIB_PATH) else False,
                    reason="only meaningful on a box without a GPU")
def test_create_fails_loudly_without_gpu(lib, small_problem):
    from rootba_amd.linearizor import LinearizorHIP
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
        LinearizorHIP(small_problem, np.float32)
    # and through the raw C ABI
    h = C.c_void_p()
    off = np.ascontiguousarray(small_problem.lm_obs_offsets, dtype=np.int64)
    cam = np.ascontiguousarray(small_problem.obs_cam_idx, dtype=np.int32)
    xy = np.ascontiguousarray(small_problem.obs_xy, dtype=np.float32)
    o = L.default_options()
    st = lib.rba_create(0, 0, small_problem.n_cams, small_problem.n_lms, off.ctypes.data_as(C.c_void_p),
                        cam.ctypes.data_as(C.c_void_p), xy.ctypes.data_as(C.c_void_p), C.byref(o), C.byref(h))
    assert st < 0 and "fallback" in L.last_error()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rootba_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("oracle/", "").replace("(oracle)", "") or \
                    "never imported" in txt or "test infrastructure" in txt, f


def test_nothing_shipped_knows_the_test_double_of_the_c_abi():
    """oracle/mock_rootba_hip.cpp (the oracle-backed stand-in of the C ABI that lets tests/test_reference_loop_on_hip.py
    exercise the reference-side binding without a GPU) is reachable from tests/ only: the product package, the public
    header, the integration sources, bench.py and __graft_entry__.py never name it, and the binding library is not
    linked against it (its rba_* symbols are unresolved until a test picks the provider)."""
    import subprocess
    needles = ("mock_rootba", "librootba_hip_mock", "hip_mock")
    paths = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for top in ("rootba_amd", "include", "integration"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            paths += [os.path.join(dirpath, f) for f in files if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp"))]
    for path in paths:
        txt = open(path).read()
        if path.endswith(os.path.join("solver", "linearizor_hip.hpp")):
            txt = txt.replace("oracle/mock_rootba_hip.cpp", "")  # its header comment says where the CPU test lives
        assert not any(n in txt for n in needles), path
    binding = os.path.join(ROOT, "oracle", "_ref", "librootba_ref_binding.so")
    if os.path.exists(binding):
        needed = subprocess.run(["readelf", "-d", binding], capture_output=True, text=True).stdout
        assert "mock" not in needed and "librootba_hip" not in needed


def _header_struct_fields(name):
    """Field names of `typedef struct <name> { ... } <name>;` in the header, in order (comments stripped)."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\}\s*%s;" % (name, name), src, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, rest = decl.split(None, 1)
        fields += [(ctype, n.strip()) for n in rest.split(",")]
    return fields


@pytest.mark.parametrize("cname,mirror", [("rba_options", "RbaOptions"), ("rba_residual_info", "RbaResidualInfo"),
                                          ("rba_cg_summary", "RbaCgSummary"), ("rba_iter_timings", "RbaIterTimings"),
                                          ("rba_substage_timings", "RbaSubstageTimings"),
                                          ("rba_lm_iteration", "RbaLmIteration"), ("rba_byte_model", "RbaByteModel")])
def test_ctypes_mirrors_match_the_header(cname, mirror):
    """Every POD struct of include/rootba_hip.h against its ctypes mirror: same field order, names and C types."""
    ctype_of = {"int": C.c_int, "double": C.c_double, "int64_t": C.c_int64}
    want = _header_struct_fields(cname)
    got = getattr(L, mirror)._fields_
    rename = {"lambda": "lambda_"}  # Python keyword
    assert [rename.get(n, n) for _, n in want] == [f[0] for f in got]
    assert [ctype_of[t] for t, _ in want] == [f[1] for f in got]
    assert C.sizeof(getattr(L, mirror)) > 0


def test_dtype_constants_match_the_header():
    src = open(HEADER).read()
    for name in ("RBA_F32", "RBA_F64", "RBA_MIXED", "RBA_OK", "RBA_NUMERICAL_FAILURE"):
        assert int(re.search(r"#define %s \(?(-?\d+)\)?" % name, src).group(1)) == getattr(L, name)
