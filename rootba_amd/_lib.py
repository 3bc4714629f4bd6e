"""ctypes binding of librootba_hip.so (C ABI: include/rootba_hip.h).

There is NO fallback: if the HIP library is missing or no GPU is present the
solver raises. (The CPU restatement under oracle/ is test infrastructure and is
never imported from this package.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librootba_hip.so")

RBA_OK = 0
RBA_NUMERICAL_FAILURE = 1
RBA_F32, RBA_F64, RBA_MIXED = 0, 1, 2


class RbaOptions(C.Structure):
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
        ("implicit_q", C.c_int),
        ("solver_type", C.c_int),  # 0 SQUARE_ROOT, 1 SCHUR_COMPLEMENT
        ("explicit_after", C.c_int),
    ]


class RbaResidualInfo(C.Structure):
    _fields_ = [
        ("all_num_obs", C.c_int),
        ("all_error", C.c_double),
        ("all_residual_sum", C.c_double),
        ("valid_num_obs", C.c_int),
        ("valid_error", C.c_double),
        ("valid_residual_sum", C.c_double),
        ("is_numerically_valid", C.c_int),
    ]


class RbaCgSummary(C.Structure):
    _fields_ = [("termination_type", C.c_int), ("num_iterations", C.c_int)]


class RbaIterTimings(C.Structure):
    _fields_ = [
        ("residual_evaluation_time", C.c_double),
        ("stage1_time", C.c_double),
        ("stage2_time", C.c_double),
        ("compute_preconditioner_time", C.c_double),
        ("solve_reduced_system_time", C.c_double),
        ("back_substitution_time", C.c_double),
        ("update_cameras_time", C.c_double),
        ("hx_time", C.c_double),
        ("hx_calls", C.c_int),
    ]


class RbaSubstageTimings(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "jacobian_evaluation_time", "scale_landmark_jacobian_time", "stage1_preconditioner_time",
        "perform_qr_time", "landmark_damping_time", "scale_pose_jacobian_time",
        "stage2_preconditioner_and_gradient_time")]


class RbaLmIteration(C.Structure):
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


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int)

# every symbol include/rootba_hip.h declares
EXPORTS = [
    "rba_default_options", "rba_last_error", "rba_device_count", "rba_create", "rba_create_sharded", "rba_get_shard_ranges",
    "rba_destroy",
    "rba_comm_unique_id", "rba_comm_init", "rba_comm_init_callback", "rba_comm_info", "rba_get_comm_stats",
    "rba_set_state", "rba_get_state", "rba_backup",
    "rba_restore", "rba_compute_error", "rba_linearize", "rba_solve", "rba_stage2",
    "rba_right_multiply", "rba_right_multiply_explicit", "rba_apply", "rba_back_substitute", "rba_optimize_lm", "rba_lm_begin", "rba_lm_step", "rba_lm_termination", "rba_synchronize",
    "rba_get_timings", "rba_get_substage_timings", "rba_debug_read_blocks",
    "rba_get_jl_col_scale", "rba_get_pose_scaling", "rba_get_landmark_R", "rba_get_landmark_q2tr_norm", "rba_get_problem_stats",
    "rba_get_byte_model", "rba_get_pcg_counters", "rba_get_reduced_matrix_info",
]


class RbaByteModel(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("compute_error", "stage1", "stage2", "product_matrix_free",
                                         "product_assembled", "assembly", "pcg_vectors", "back_substitution",
                                         "persistent_solve", "persistent_iteration")]

class RbaPcgCounters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("products_matrix_free", "products_assembled", "assemblies", "iterations",
                                         "solves_repeated_matrix_free", "early_switches", "solves_persistent",
                                         "products_assembled_resident", "iterations_resident", "cost_evaluations")]

class RbaReducedMatrixInfo(C.Structure):
    _fields_ = [("blocks_stored", C.c_int64), ("blocks_full", C.c_int64), ("density", C.c_double),
                ("bytes_stored", C.c_int64), ("resident_in_registers", C.c_int), ("persistent_workgroups", C.c_int)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m rootba_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch ships its own ROCm runtime (libamdhip64.so.7, libhsa-runtime64, librccl.so.1)
        # under the same SONAMEs as /opt/rocm: whichever is loaded first serves the whole
        # process. A mixed pair breaks RCCL ("no ROCm-capable device is detected" inside
        # ncclCommInitRank), so when PyTorch is installed it goes first, as in bench.py.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.rba_last_error.restype = C.c_char_p
    return _lib


def last_error() -> str:
    return (lib().rba_last_error() or b"").decode()


def check(status: int, what: str, allow_numerical_failure: bool = False) -> int:
    if status == RBA_OK or (allow_numerical_failure and status == RBA_NUMERICAL_FAILURE):
        return status
    raise RuntimeError(f"{what} failed with status {status}: {last_error()}")


def device_count() -> int:
    n = C.c_int(0)
    lib().rba_device_count(C.byref(n))
    return n.value


def default_options(**kw) -> RbaOptions:
    o = RbaOptions()
    lib().rba_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o
