"""The kernels' LOGIC on a machine without a GPU: a few GPU tests re-run in a subprocess on the CPU execution harness of
tests/hipemu (the product's solver.hip and kernel headers compiled unchanged as C++; a development tool, not a
backend - see tests/hipemu/README.md). The subprocess keeps the harness build out of this test session, whose loader
stays on the real library. Skipped when the harness cannot be built here (it needs the ROCm clang++)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_slice_of_the_gpu_suite_on_the_cpu_harness():
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    if not os.path.exists(build_emu.CXX):
        pytest.skip(f"{build_emu.CXX} is not available")
    try:
        build_emu.build()
    except subprocess.CalledProcessError as e:
        pytest.skip(f"the harness does not build here: {e}")
    env = dict(os.environ, RBA_EMU="1")
    sel = ("test_compute_error or (test_solve_and_apply and sqrt-schur_jacobi) or (test_invalid_projections and ERROR_VALID-float32)")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-k", sel,
                          os.path.join(ROOT, "tests", "test_reference_gpu.py")],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    tail = out.stdout[-1500:] + out.stderr[-500:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout, tail
    # the two forms of stage 1 (two kernels / fused with an observation per lane) and the assembled reduced matrix built on the neighbour lists, on small problems
    sel = "(test_fused_stage1 and small) or (test_explicit_reduced_matrix_is_the_same_operator and small-float32)"
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-k", sel,
                          os.path.join(ROOT, "tests", "test_gpu_parity.py")],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    tail = out.stdout[-1500:] + out.stderr[-500:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout, tail


def test_bench_with_two_ranks_on_the_cpu_harness():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), on the harness with the
    file-based stand-in for librccl: the library's RCCL code path with more than one rank (rba_comm_init, the union of the
    block structure, every all-reduce site), the collective transport decision and the JSON line. Both ranks walk the same
    trajectory as one rank does (the per-iteration costs of the log)."""
    import json
    import socket
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    if not os.path.exists(build_emu.CXX):
        pytest.skip(f"{build_emu.CXX} is not available")
    try:
        build_emu.build()
    except subprocess.CalledProcessError as e:
        pytest.skip(f"the harness does not build here: {e}")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    harness = os.path.join(ROOT, "tests", "hipemu", "bench_on_harness.py")
    common = ["--workload", "ladybug-49", "--steps", "2", "--warmup", "2", "--cpu-baseline-iters", "0",
              "--no-reference-semantics"]
    env = dict(os.environ, HIPEMU_THREADS="4")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), harness, "--gpus", "2", *common],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert two.returncode == 0, two.stderr[-3000:]
    line2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and "transport=rccl, nranks=2" in line2["config"]["parallelism"]
    assert line2["config"]["comm_per_step"]["all_reduces"] > 0
    one = subprocess.run([sys.executable, harness, *common], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert one.returncode == 0, one.stderr[-3000:]
    line1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert line1["n_gpus"] == 1 and line1["config"]["comm_per_step"] is None
    assert abs(line2["config"]["final_cost"] - line1["config"]["final_cost"]) <= 2e-6 * line1["config"]["final_cost"]
    assert line2["config"]["cg_iterations_per_step"] == line1["config"]["cg_iterations_per_step"]
    # ... and with the products on the assembled matrix split over the two ranks (what large problems do by themselves,
    # Solver::decide_product_split: one all-reduce of the product vector per PCG iteration through the communicator)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    split = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), harness, "--gpus", "2", *common],
                           cwd=ROOT, env=dict(env, RBA_PCG_SPLIT="1", RBA_EXPLICIT_AFTER="2"), capture_output=True, text=True,
                           timeout=1500)
    assert split.returncode == 0, split.stderr[-3000:]
    line3 = json.loads([ln for ln in split.stdout.splitlines() if ln.startswith("{")][-1])
    assert line3["config"]["comm_per_step"]["all_reduces"] > line2["config"]["comm_per_step"]["all_reduces"]
    assert abs(line3["config"]["final_cost"] - line1["config"]["final_cost"]) <= 2e-6 * line1["config"]["final_cost"]
