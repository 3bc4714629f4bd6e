"""Real multi-rank RCCL: `bench.py --gpus 2` launched WITHOUT a launcher must start its two ranks
itself, create the library's RCCL communicator over both, and report n_gpus == 2 with the
communicator's own rank count. Auto-skips on boxes with fewer than two GPUs (the single-GPU
development boxes): there the two-rank logic is covered by tests/test_gpu_sharded.py (callback
transport, two ranks on one GPU) and the RCCL call path by test_rccl_call_path_with_one_rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.timeout(900)
def test_bench_self_launches_two_rccl_ranks():
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--workload", "trafalgar-257", "--cpu-baseline-iters", "0"],
                       env=env, capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2
    assert "transport=rccl, nranks=2" in line["config"]["parallelism"]
    assert line["config"]["comm_per_step"]["all_reduces"] > 0
    # same problem, same trajectory as one GPU (replicated reductions are deterministic per rank count)
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                         "--workload", "trafalgar-257", "--cpu-baseline-iters", "0"],
                        env=env, capture_output=True, text=True, timeout=850)
    one = json.loads([l for l in p1.stdout.splitlines() if l.startswith("{")][-1])
    assert abs(line["config"]["final_cost"] - one["config"]["final_cost"]) < 2e-6 * one["config"]["final_cost"]


@pytest.mark.timeout(600)
def test_bench_single_gpu_line_is_complete():
    """The N = 1 line carries everything the driver and the judge read."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2",
                        "--workload", "ladybug-49", "--cpu-baseline-iters", "2"],
                       env=env, capture_output=True, text=True, timeout=550)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 0
    assert "transport=none, nranks=1" in line["config"]["parallelism"]
    assert line["config"]["value_reference_semantics"]["termination"] in ("CONVERGED", "NO_CONVERGENCE")
    assert set(line["roofline"]["stages"]) >= {"stage1", "stage2", "back_substitution", "compute_error", "pcg"}
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
