"""bench.py on the CPU execution harness (development tool, see tests/hipemu/README.md): exercises the control flow of the
benchmark script - also for N > 1 ranks, with tests/hipemu/fake_rccl.cpp standing in for librccl - where there is no
GPU. The numbers it prints mean nothing.

  one rank:   python tests/hipemu/bench_on_harness.py --workload ladybug-49 --steps 3 --warmup 2 --cpu-baseline-iters 1
  two ranks:  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \\
                  --master-addr 127.0.0.1 --master-port 29611 tests/hipemu/bench_on_harness.py --gpus 2 \\
                  --workload ladybug-49 --steps 3 --warmup 2 --cpu-baseline-iters 0
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import build_emu  # noqa: E402
import rootba_amd._lib as L  # noqa: E402

L.LIB_PATH = build_emu.build()
os.environ.setdefault("HIPEMU_RCCL", os.path.join(os.path.dirname(L.LIB_PATH), "fake_rccl", "librccl.so.1"))

# the pieces of torch.cuda / NCCL the script touches, on the CPU
torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 8
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None
_tensor = torch.tensor


def _cpu_tensor(*a, **k):
    if str(k.get("device", "")).startswith("cuda"):
        k["device"] = "cpu"
    return _tensor(*a, **k)


torch.tensor = _cpu_tensor
torch.Tensor.cuda = lambda self, *a, **k: self
_init = dist.init_process_group


def _gloo(backend=None, **k):
    k.pop("device_id", None)
    return _init("gloo", **k)


dist.init_process_group = _gloo

if __name__ == "__main__":
    sys.argv[0] = os.path.join(ROOT, "bench.py")
    os.chdir(ROOT)
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
