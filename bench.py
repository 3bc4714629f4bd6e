#!/usr/bin/env python
"""bench.py — LM iterations/sec of the MI355X-native square-root BA solver.

Metric (BASELINE.json): LM iterations/sec (linearize + QR + PCG + back-sub) on
BAL venice-1778 (synthetic stand-in of the same size/shape; the
real BAL file is used when present). A "step" is ONE LM iteration = one row of
the reference's iteration log (bal_bundle_adjustment.cpp:291-521): compute_error,
stage 1 (when the linearisation point moved), stage 2 + preconditioner + PCG,
back-substitution + camera update, compute_error.

    python bench.py --gpus N --steps K --warmup W

N > 1 shards the landmarks of the SAME problem over N ranks (strong scaling); the
camera-sized vectors are all-reduced with RCCL inside the library. When WORLD_SIZE is
not set and N > 1 the script launches its N ranks itself (torch.distributed.run on
127.0.0.1); under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
it runs as one of them. Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

# CPUs of this process BEFORE any OpenMP runtime starts: with OMP_PLACES set, libgomp binds the main
# thread to its place as soon as it is loaded (with numpy / torch), and the affinity mask read later
# would be that single core
_CPUS_AT_START = len(os.sched_getaffinity(0))
# the CPU baseline's OpenMP threads stay on their cores / NUMA node (must be set before libgomp starts)
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DTYPE = np.float32
GPU_DTYPE = np.float32  # what LinearizorHIP is created with: DTYPE, or "mixed" (--mixed: double state and costs,
                        # float linear algebra; the CPU baseline then runs the float32 oracle)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def shard_ranges(k: np.ndarray, n: int):
    """Contiguous landmark ranges balanced by the bytes a landmark moves per LM iteration in this layout - ~120 per
    observation (rows, records, reflectors) + ~100 per landmark - the weights rba_create_sharded uses inside the
    library (rounds 1-4 balanced sum(k^2), the bytes of dense blocks that no longer exist); every rank gets >= 1."""
    if k.size < n:
        raise SystemExit(f"cannot shard {k.size} landmarks over {n} ranks")
    w = np.cumsum(120.0 * k.astype(np.float64) + 100.0)
    cuts = [0]
    for r in range(1, n):
        c = int(np.searchsorted(w, w[-1] * r / n))
        cuts.append(min(max(c, cuts[-1] + 1), k.size - (n - r)))
    cuts.append(k.size)
    return [(cuts[i], cuts[i + 1]) for i in range(n)]


def take_landmarks(prob, lo: int, hi: int):
    from rootba_amd.problem import BalProblem
    off = prob.lm_obs_offsets
    o0, o1 = int(off[lo]), int(off[hi])
    return BalProblem(prob.cams, prob.lms[lo:hi].copy(), (off[lo:hi + 1] - o0).copy(),
                      prob.obs_cam_idx[o0:o1].copy(), prob.obs_xy[o0:o1].copy(), prob.name)


REAL_FILES = {
    "venice-1778": ("venice", "problem-1778-993923-pre.txt"),
    "ladybug-49": ("ladybug", "problem-49-7776-pre.txt"),
    "trafalgar-257": ("trafalgar", "problem-257-65132-pre.txt"),
    "final-13682": ("final", "problem-13682-4456117-pre.txt"),
}


def make_problem(workload: str, args):
    from rootba_amd import problem as P
    base = workload.split("+")[0]
    real = os.path.join(ROOT, "..", "rootba_data", "bal", *REAL_FILES.get(base, ("none", "none")))
    if os.path.exists(real):
        raw, data = P.read_bal(real), "real"
    else:
        raw, data = P.named_synthetic(workload), "synthetic"
    prob = P.preprocess(raw, translation_sigma=args.translation_sigma, point_sigma=args.point_sigma,
                        rotation_sigma=args.rotation_sigma)
    return prob, data


PRECOND = {"JACOBI": 0, "SCHUR_JACOBI": 1, "POWER_SCHUR_COMPLEMENT": 2}
_SOLVER_KW = {}
_GPU_KW = {}  # switches that exist only in the HIP library


def solver_options(mod, n_iter: int, function_tolerance: float = 0.0):
    # reference defaults + CVPR'21 common settings (Huber 1.0); function_tolerance = 0 so that
    # exactly warmup+steps LM iterations are executed in the timed run
    return mod.default_options(robust_norm=1, huber_parameter=1.0, max_num_iterations=n_iter,
                               function_tolerance=function_tolerance, **_SOLVER_KW)


def effective_cpus() -> tuple[int, str]:
    """CPUs this process may actually use: the cgroup CPU quota when there is one (the GPU boxes run the
    container with cpu.max = 16 CPUs on a 256-thread host: more OpenMP threads than that only get throttled
    - measured 37 GB/s with 16 threads, 31 with 128, 8.5 with 256), else the affinity mask."""
    n = _CPUS_AT_START
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            q = max(1, int(float(quota) / float(period) + 0.5))
            if q < n:
                return q, f"cgroup cpu.max quota {q} of {n} hardware threads"
    except (OSError, ValueError):
        pass
    return n, f"{n} hardware threads (no quota)"


def cpu_baseline(prob, n_iter: int, gpu_rows):
    """Oracle (CPU restatement of the reference path) on a bounded sample of the same workload: the
    first LM iterations, all host cores; the number of iterations is chosen from a one-iteration probe so
    that the sample stays within about 25 s on any host (at most 2 * n_iter iterations)."""
    from oracle import oracle as O

    n_cpu, cpu_note = effective_cpus()

    def run(n):
        t0 = time.perf_counter()
        opts = solver_options(O, n)
        opts.num_threads = n_cpu
        o = O.Oracle(prob, DTYPE, opts)
        log(f"[cpu_baseline] oracle set up in {time.perf_counter() - t0:.1f}s, threads={o.num_threads()}")
        rows, _ = o.optimize_lm()
        its = [r for r in rows if r.iteration >= 1]
        return o.num_threads(), its, sum(r.iteration_time for r in its)

    # Bounded sample (about 10-30 s of CPU work whatever the host is): a one-iteration probe gives the
    # cost of a linearisation and of one CG iteration on this host; the GPU run's CG counts of the same
    # iterations (the oracle follows the same trajectory) then predict the time of iterations 1..n, and n
    # is the largest count <= 2 * n_iter whose prediction stays below the budget.
    budget = 25.0
    threads, its, t = run(1)
    if its:
        cg1 = max(1, its[0].cg_iterations)
        t_cg = its[0].pcg_time / cg1
        t_lin = max(0.0, its[0].iteration_time - its[0].pcg_time)
        pred, n_fit = 0.0, 0
        for r in [r for r in gpu_rows if r.iteration >= 1][:2 * n_iter]:
            pred += t_lin + t_cg * r.cg_iterations * (1 + 1.0 / 10)
            if pred > budget and n_fit >= 1:
                break
            n_fit += 1
        log(f"[cpu_baseline] probe: {t_lin:.2f} s per linearisation + {t_cg:.3f} s per CG iteration -> "
            f"sampling LM iterations 1..{n_fit}")
        n_iter = max(1, n_fit)
        if n_iter > 1:
            threads, its, t = run(n_iter)
    else:
        n_iter = 1
    g = [r for r in gpu_rows if 1 <= r.iteration <= n_iter]
    tg = sum(r.iteration_time for r in g)
    for r in its:
        log(f"[cpu_baseline] it {r.iteration} cg {r.cg_iterations} cost {r.cost:.6e} t {r.iteration_time:.2f}s")
    n_cg = sum(r.cg_iterations for r in its)
    t_pcg = sum(r.pcg_time for r in its)
    # the oracle streams the dense (2k) x (9k + pad) operand of every landmark per product
    k = prob.obs_per_lm().astype(np.int64)
    dense_bytes = int((2 * k * (9 * k + (4 - (9 * k) % 4) % 4)).sum()) * np.dtype(DTYPE).itemsize
    return {
        "value": len(its) / t if t > 0 else None,
        "unit": "LM iterations/s",
        "cores": threads,
        "cores_note": cpu_note,
        "kind": "port",
        "kind_note": "OpenMP restatement (oracle/liboracle.so), pinned against the reference's own sources compiled "
                     "with third-party stand-ins (oracle/_ref, tests/test_oracle_vs_reference.py); that build is serial "
                     "and scalar, so it is the checker of this port, not a timing baseline",
        "sample": (f"LM iterations 1..{n_iter} of the same problem, {np.dtype(DTYPE).name}, "
                   f"{n_cg} CG iterations in total, {t:.1f} s; "
                   f"the GPU path runs the same {n_iter} iterations at {len(g) / tg if tg > 0 else 0:.1f} it/s"),
        "product_GBps": ((n_cg + sum(r.cg_iterations // 10 for r in its)) * dense_bytes / 1e9 / t_pcg)
        if t_pcg > 0 else None,
        "product_note": "achieved bandwidth of the CPU's H*x (dense blocks, first-touch NUMA placement, "
                        "per-thread accumulators, pinned OpenMP threads)",
        "final_cost_rel_diff_vs_gpu": (abs(its[-1].cost - g[-1].cost) / its[-1].cost) if g and its else None,
    }


def pmc_child(args):
    """Child of measure_product_traffic(), run under `rocprofv3 --pmc ...`: calibration reads of a known byte count, then
    `args.pmc_child` matrix-free products of the workload (what `roofline.achieved` times)."""
    import ctypes as C
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    prob, _ = make_problem(args.workload, args)
    g = LinearizorHIP(prob, GPU_DTYPE, solver_options(L, 2), device=0)
    nbytes = C.c_int64(0)
    for _ in range(3):
        L.check(g.lib.rba_debug_read_blocks(g.h, 4, C.byref(nbytes)), "calibration read")
    assert g.linearize() == 0
    g.stage2(1e-4)
    x = np.random.default_rng(0).uniform(-1, 1, 9 * prob.n_cams).astype(DTYPE)
    for _ in range(args.pmc_child):
        g.right_multiply(x)
    print("PMC_CHILD " + json.dumps({"calib_bytes": nbytes.value, "products": args.pmc_child}), flush=True)
    g.close()


def measure_product_traffic(argv):
    """HBM traffic of ONE matrix-free product, measured in this run: rocprofv3 PMC counters FETCH_SIZE and WRITE_SIZE in
    SEPARATE passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section) over a child process that repeats the
    product on the same workload; FETCH_SIZE is corrected by the factor measured on a streaming read of a known byte
    count in the same pass (gfx950 reports half the bytes of a wide coalesced read), WRITE_SIZE taken as reported
    (KiB). Returns (bytes per product or None, description)."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this process already runs under rocprofv3 (no nested counter collection)"
    n_prod = 12
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", str(n_prod)] + \
        [a for a in argv if a not in ("--no-reference-semantics",)]
    per = {}
    meta = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rba_pmc_", dir="/tmp")
        try:
            r = subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
            for ln in r.stdout.splitlines():
                if ln.startswith("PMC_CHILD "):
                    meta = json.loads(ln[len("PMC_CHILD "):])
            if r.returncode != 0 or meta is None:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr[-300:]}"
            acc = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter:
                        a = acc.setdefault(row["Kernel_Name"], [0.0, 0])
                        a[0] += float(row["Counter_Value"])
                        a[1] += 1
            per[counter] = acc
        finally:
            shutil.rmtree(d, ignore_errors=True)
    calib = [v for k, v in per["FETCH_SIZE"].items() if "k_calib_read<4>" in k]
    if not calib:
        return None, "no calibration kernel in the counter output"
    corr = meta["calib_bytes"] / (calib[0][0] / calib[0][1] * 1024.0)
    fetch = sum(v[0] for k, v in per["FETCH_SIZE"].items() if "k_hx_implicit" in k) * 1024.0 * corr
    write = sum(v[0] for k, v in per["WRITE_SIZE"].items() if "k_hx_implicit" in k) * 1024.0
    return (fetch + write) / meta["products"], (
        f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
        f"{meta['products']} products of a child process on the same workload; FETCH_SIZE x {corr:.4f} (calibrated on a "
        f"streaming read of {meta['calib_bytes']} bytes in the same pass), WRITE_SIZE as reported; KiB x 1024")


def self_launch(args_list, n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks on this node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *args_list]
    log(f"[bench] WORLD_SIZE is not set: launching {n} ranks: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dense_companion(args):
    """The headline workload's other regime: `venice-1778+tail` - the same cameras,
    landmark count and observation count with heavy-tailed (Pareto) track lengths, whose reduced camera matrix is nearly
    dense (524 MB in half storage) and does NOT fit the register files: long PCG solves stream it from HBM in two
    launches per iteration. The synthetic scene of SURVEY.md 8d gives a banded matrix; real BAL venice is an internet
    photo collection and may be in either regime. One repetition of the same warm-up + steps in a child process."""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "venice-1778+tail", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--cpu-baseline-iters", "0", "--no-pmc", "--no-reference-semantics",
           "--repeats", "1", "--no-dense-companion", "--preconditioner", args.preconditioner,
           "--power-order", str(args.power_order)] + (["--mixed"] if args.mixed else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        c = d["config"]
        return {"value": d["value"], "unit": d["unit"], "workload": c["workload"], "ms_per_step": d["ms_per_step"],
                "reduced_matrix": c["reduced_matrix"], "solves_persistent": c["solves_persistent"],
                "successful_steps": c["successful_steps"], "cg_iterations_per_step": c["cg_iterations_per_step"],
                "executed": d["roofline"]["stages"]["pcg"]["executed"]}
    except Exception as e:  # the companion must never take the headline down
        log(f"[value_dense_covisibility] failed: {e!r}")
        return None


def reference_semantics_run(local, prob_name, rank, world, local_rank, comm_setup, devices=None):
    """The same LM run with the reference's own stopping rule (function_tolerance = 1e-6,
    bal_bundle_adjustment.cpp:174-201, 476-481): it/s over iterations 1..(iteration where it fires)."""
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    opts = solver_options(L, 50, function_tolerance=1e-6)
    for key, val in _GPU_KW.items():
        setattr(opts, key, val)
    lin = LinearizorHIP(local, GPU_DTYPE, opts, device=local_rank, devices=devices)
    comm_setup(lin)
    lin.lm_begin()
    lin.lm_step()  # iteration 0: evaluation only
    lin.synchronize()
    t0 = time.perf_counter()
    n, cg, more, cost = 0, 0, True, None
    while more:
        row, more = lin.lm_step()
        n += 1
        cg += row.cg_iterations
        if row.step_is_successful:
            cost = row.cost
    lin.synchronize()
    t = time.perf_counter() - t0
    term = lin.lm_termination()
    lin.close()
    return {"value": n / t, "unit": "LM iterations/s", "iterations": n, "seconds": t, "cg_iterations": cg,
            "termination": {0: "NO_CONVERGENCE", 1: "CONVERGED", -1: "FAILURE"}.get(term, str(term)),
            "final_cost": cost,
            "rule": "iterations 1..k, k = first iteration with |cost change| <= 1e-6 cost (the reference's "
                    "function_tolerance); the headline `value` runs exactly --steps iterations with the rule off"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=18)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="venice-1778")
    ap.add_argument("--cpu-baseline-iters", type=int, default=4,
                    help="LM iterations of the CPU oracle timed beside the GPU (0 = skip)")
    ap.add_argument("--translation-sigma", type=float, default=0.01,
                    help="camera-centre perturbation (SURVEY.md 8d / CVPR'21 common settings: 0.01)")
    ap.add_argument("--point-sigma", type=float, default=0.01)
    ap.add_argument("--rotation-sigma", type=float, default=0.0)
    ap.add_argument("--preconditioner", choices=sorted(PRECOND), default="SCHUR_JACOBI",
                    help="reference default: SCHUR_JACOBI")
    ap.add_argument("--power-order", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=3,
                    help="repetitions of the whole measurement (warm-up + steps, from the same initial state): `value` "
                         "is the MEDIAN repetition (SURVEY.md 8d: median of >= 3 runs), all of them are reported")
    ap.add_argument("--solver-type", choices=["SQUARE_ROOT", "SCHUR_COMPLEMENT"], default="SQUARE_ROOT",
                    help="SCHUR_COMPLEMENT: explicit reduced camera matrix + SpMV (SURVEY.md 8f #4), 1 GPU")
    ap.add_argument("--use-double", action="store_true", help="float64 (BASELINE metric is float32)")
    ap.add_argument("--mixed", action="store_true",
                    help="RBA_MIXED: double state / observations / costs, float linear algebra (BASELINE config 5)")
    ap.add_argument("--no-reference-semantics", action="store_true",
                    help="skip the second run with the reference's function_tolerance stopping rule")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 counter passes over a child process, "
                         "~30 s); the committed profiles/hx_traffic.json is quoted instead")
    ap.add_argument("--no-dense-companion", action="store_true",
                    help="skip `value_dense_covisibility` (the same workload with heavy-tailed track lengths, whose reduced "
                         "camera matrix is nearly dense and does not fit the register files; one repetition)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N through ONE handle of ONE process (rba_create_sharded: the library shards the landmarks "
                         "over devices 0..N-1 itself, one host thread and one RCCL rank per device) instead of one "
                         "process per GPU - the entry a drop-in binding of the reference's one-process driver uses")
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--companion-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.single_process and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        raise SystemExit("--single-process is ONE process over --gpus devices: do not start it under a multi-rank launcher")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.single_process:
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    global DTYPE, GPU_DTYPE
    DTYPE = np.float64 if args.use_double else np.float32
    GPU_DTYPE = "mixed" if args.mixed else DTYPE
    if args.mixed and args.use_double:
        raise SystemExit("--mixed and --use-double exclude each other")
    _SOLVER_KW.update(preconditioner_type=PRECOND[args.preconditioner], power_order=args.power_order)
    _GPU_KW.update(solver_type=int(args.solver_type == "SCHUR_COMPLEMENT"))
    if args.pmc_child:
        pmc_child(args)
        return

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sp_devices = list(range(args.gpus)) if args.single_process and args.gpus > 1 else None
    if world != args.gpus and sp_devices is None:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong "
                         f"number of ranks")
    if sp_devices is not None and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"--single-process --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the solver has no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP

    t0 = time.perf_counter()
    prob, data = make_problem(args.workload, args)
    log(f"[rank {rank}] problem {prob.name}: {prob.n_cams} cams, {prob.n_lms} lms, {prob.n_obs} obs "
        f"({time.perf_counter() - t0:.1f}s)")
    lo, hi = shard_ranges(prob.obs_per_lm(), world)[rank]
    local = prob if world == 1 else take_landmarks(prob, lo, hi)

    n_iter = args.warmup + args.steps - 1  # the first warmup step is iteration 0 (evaluation)
    gpu_opts = solver_options(L, max(n_iter, 1))
    for key, val in _GPU_KW.items():
        setattr(gpu_opts, key, val)
    t0 = time.perf_counter()
    # HIP events around the matrix-free products (rba_iter_timings.hx_time -> roofline.avg_launch_ms) are marker packets
    # on the solver stream: the timed repetitions run WITHOUT them, a separate pass on a second handle measures the
    # product's launch time. An explicit RBA_HX_TIMING_STRIDE in the environment wins.
    stride_env = os.environ.get("RBA_HX_TIMING_STRIDE")
    os.environ["RBA_HX_TIMING_STRIDE"] = stride_env if stride_env is not None else "0"
    lin = LinearizorHIP(local, GPU_DTYPE, gpu_opts, device=local_rank, devices=sp_devices)
    log(f"[rank {rank}] solver set up in {time.perf_counter() - t0:.2f}s (rba_create: sort by track length, "
        f"CSC index, block structure of the reduced matrix, launch graphs, device allocation)")

    # ---- transport: decided COLLECTIVELY before anybody enters ncclCommInitRank ------------------
    use_callback = [False]
    if world > 1:
        probe = 1
        try:
            LinearizorHIP.comm_unique_id()  # can this rank load librccl and talk to it at all?
        except Exception as e:
            log(f"[rank {rank}] RCCL probe failed: {e!r}")
            probe = 0
        flag = torch.tensor([probe], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        use_callback[0] = int(flag.item()) == 0
        if use_callback[0]:
            log(f"[rank {rank}] RCCL is not usable on every rank: ALL ranks use the callback transport "
                f"(torch.distributed all-reduce, host staging)")

    def comm_setup(solver):
        if world == 1:
            return
        if use_callback[0]:
            def allreduce(arr, op):
                t = torch.from_numpy(arr).cuda()
                dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
                arr[...] = t.cpu().numpy()
            solver.comm_init_callback(rank, world, allreduce)
        else:
            uid = [LinearizorHIP.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            solver.comm_init(rank, world, uid[0])  # a failure here is fatal (no silent fallback)
        info = solver.comm_info()
        if info["nranks"] != world or info["rank"] != rank:
            raise SystemExit(f"rank {rank}: the library's communicator reports {info}, expected {world} ranks")

    comm_setup(lin)
    info = lin.comm_info()
    stats = lin.problem_stats()
    bytes_model = lin.byte_model()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    state0 = lin.get_state()

    def measure(lin=lin):
        """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides."""
        rows, hx_time, hx_calls = [], 0.0, 0
        lin.set_state(*state0)
        lin.lm_begin()
        for _ in range(args.warmup):
            row, more = lin.lm_step()
            rows.append(row)
            if not more:
                raise SystemExit(f"the LM loop terminated during warm-up (iteration {row.iteration}): no valid measurement")
        lin.synchronize()
        comm0, pcg0 = lin.comm_stats(), lin.pcg_counters()
        barrier()
        t_start = time.perf_counter()
        for s_ in range(args.steps):
            row, more = lin.lm_step()
            rows.append(row)
            tm = lin.timings()
            hx_time += tm.hx_time
            hx_calls += tm.hx_calls
            if not more and s_ + 1 < args.steps:
                raise SystemExit(f"the LM loop terminated after {s_ + 1} of {args.steps} timed steps (iteration "
                                 f"{row.iteration}, lambda {row.lambda_:.2e}): refusing to count no-op steps")
        lin.synchronize()
        barrier()
        elapsed = time.perf_counter() - t_start
        comm1, pcg1 = lin.comm_stats(), lin.pcg_counters()
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return dict(elapsed=elapsed, rows=rows, hx_time=hx_time, hx_calls=hx_calls, comm0=comm0, comm1=comm1,
                    pcg={k: pcg1[k] - pcg0[k] for k in pcg1})

    reps = [measure() for _ in range(max(1, args.repeats))]
    order = sorted(range(len(reps)), key=lambda i: reps[i]["elapsed"])
    med = reps[order[len(order) // 2]]  # the median repetition is the one reported in detail
    elapsed, rows, hx_time, hx_calls = med["elapsed"], med["rows"], med["hx_time"], med["hx_calls"]
    comm0, comm1, pcg_cnt = med["comm0"], med["comm1"], med["pcg"]
    matrix_info = lin.reduced_matrix_info()
    lin.close()
    hx_pass = "events around every product of the timed repetitions (RBA_HX_TIMING_STRIDE from the environment)"
    if stride_env is None:
        # the product's launch time: the same warm-up + steps once more on a handle with HIP events around every 8th
        # matrix-free product (not part of `value`)
        os.environ["RBA_HX_TIMING_STRIDE"] = "8"
        lin_t = LinearizorHIP(local, GPU_DTYPE, gpu_opts, device=local_rank, devices=sp_devices)
        comm_setup(lin_t)
        tp = measure(lin_t)
        hx_time, hx_calls = tp["hx_time"], tp["hx_calls"]
        lin_t.close()
        os.environ["RBA_HX_TIMING_STRIDE"] = "0"
        hx_pass = ("a separate pass of the same warm-up + steps on a second handle with HIP events on the solver stream "
                   "around every 8th matrix-free product; the timed repetitions carry no events")

    ref_sem = None
    if not args.no_reference_semantics:
        ref_sem = reference_semantics_run(local, prob.name, rank, world, local_rank, comm_setup, sp_devices)

    if rank == 0:
        timed = rows[args.warmup:]
        for r in rows:
            log(f"  it {r.iteration:2d} ok {r.step_is_successful} cg {r.cg_iterations:3d} cost {r.cost:.8e} "
                f"lambda {r.lambda_:.2e} t {r.iteration_time * 1e3:8.2f} ms (s1 {r.stage1_time * 1e3:.2f} "
                f"s2 {r.stage2_time * 1e3:.2f} pcg {r.pcg_time * 1e3:.2f} bs {r.backsub_time * 1e3:.2f} "
                f"err {r.residual_time * 1e3:.2f})")
        sc = args.solver_type == "SCHUR_COMPLEMENT"
        avg_hx = hx_time / hx_calls if hx_calls else None
        # (one handle over several devices: the timed launches are device 0's, over its share of the landmarks - the ranges
        #  are balanced by bytes, rba_get_shard_ranges)
        hx_bytes_per_launch = stats["hx_bytes"] / (len(sp_devices) if sp_devices else 1)
        achieved = hx_bytes_per_launch / avg_hx / 1e9 if avg_hx else None
        traffic, traffic_source = None, None
        if world == 1 and sp_devices is None and not sc and not args.no_pmc:
            try:
                traffic, traffic_source = measure_product_traffic(sys.argv[1:])
                if traffic is None:
                    log(f"[roofline.traffic] not measured in this run: {traffic_source}")
            except Exception as e:  # the counter passes must never take the bench line down
                log(f"[roofline.traffic] in-run measurement failed: {e!r}")
                traffic = None
        tpath = os.path.join(ROOT, "profiles", "hx_traffic.json")
        if traffic is None and os.path.exists(tpath) and world == 1 and sp_devices is None and not sc:
            try:
                with open(tpath) as f:
                    traffic = json.load(f).get(args.workload + "/implicit_q", {}).get("traffic_bytes_per_launch")
                traffic_source = ("profiles/hx_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  "scripts/run_pmc_traffic.sh on this kernel and workload; not measured in this run)")
            except Exception:
                traffic = None

        # per-stage and whole-iteration rooflines from the stage timers of the timed rows (device clock stamps, solver.hip: time_begin) and
        # the library's byte model (include/rootba_hip.h: rba_byte_model)
        def frac(nbytes, secs):
            return nbytes / secs / 1e9 / HBM_PEAK_GBS if secs > 0 else None
        n_t = max(1, len(timed))
        n_lin = sum(1 for r in timed if r.stage1_time > 0)
        t_s1 = sum(r.stage1_time for r in timed)
        t_s2 = sum(r.stage2_time for r in timed)
        t_bs = sum(r.backsub_time for r in timed)
        t_err = sum(r.residual_time for r in timed)
        t_pcg = sum(r.pcg_time for r in timed)
        n_cg = sum(r.cg_iterations for r in timed)
        # (cost evaluations the timed steps launched, counted by the library: since round 5 the LM loop reuses the trial
        #  evaluation of an accepted step - one per step)
        n_err = pcg_cnt.get("cost_evaluations") or sum(2 if r.stage1_time > 0 else 1 for r in timed)
        b_iter = (bytes_model["stage1"] * n_lin + bytes_model["stage2"] * n_t + bytes_model["back_substitution"] * n_t +
                  bytes_model["compute_error"] * n_err)
        # the PCG phase priced with what it executed (rba_get_pcg_counters): products on either operator, assemblies
        # of the reduced matrix, the vector / preconditioner work of every iteration
        # (products the persistent kernel executed out of the register files move no matrix bytes: such a solve is
        #  priced with the one load of the matrix and the records its iterations exchange)
        res_p, res_i = pcg_cnt.get("products_assembled_resident", 0), pcg_cnt.get("iterations_resident", 0)
        b_pcg = (pcg_cnt["products_matrix_free"] * bytes_model["product_matrix_free"] +
                 (pcg_cnt["products_assembled"] - res_p) * bytes_model["product_assembled"] +
                 pcg_cnt.get("solves_persistent", 0) * bytes_model.get("persistent_solve", 0) +
                 res_i * bytes_model.get("persistent_iteration", 0) +
                 pcg_cnt["assemblies"] * bytes_model["assembly"] + (pcg_cnt["iterations"] - res_i) * bytes_model["pcg_vectors"])
        stages = {
            "stage1": {"bytes_per_launch": bytes_model["stage1"], "ms": 1e3 * t_s1 / max(1, n_lin),
                       "frac": frac(bytes_model["stage1"] * n_lin, t_s1)},
            "stage2": {"bytes_per_launch": bytes_model["stage2"], "ms": 1e3 * t_s2 / n_t,
                       "frac": frac(bytes_model["stage2"] * n_t, t_s2)},
            "back_substitution": {"bytes_per_launch": bytes_model["back_substitution"], "ms": 1e3 * t_bs / n_t,
                                  "frac": frac(bytes_model["back_substitution"] * n_t, t_bs)},
            "compute_error": {"bytes_per_launch": bytes_model["compute_error"], "ms": 1e3 * t_err / max(1, n_err),
                              "frac": frac(bytes_model["compute_error"] * n_err, t_err)},
            "pcg": {"ms_per_step": 1e3 * t_pcg / n_t, "cg_iterations": n_cg, "frac": frac(b_pcg, t_pcg),
                    "bytes_per_step": b_pcg / n_t, "executed": pcg_cnt,
                    "bytes_per_matrix_free_product": bytes_model["product_matrix_free"],
                    "bytes_per_assembled_product": bytes_model["product_assembled"],
                    "bytes_per_assembly": bytes_model["assembly"]},
        }
        out = {
            # BASELINE.json's metric string for the headline workload; other workloads are named as what they are
            "metric": "LM iterations/sec (linearize+QR+PCG+back-sub) on BAL " + args.workload.split("+")[0],
            "value": args.steps / elapsed,
            "value_repeats": {"what": f"{len(reps)} repetitions of warm-up + steps from the same initial state on one "
                                      "handle; `value` is the median repetition",
                              "values": [args.steps / r["elapsed"] for r in reps],
                              "spread_rel": (max(r["elapsed"] for r in reps) - min(r["elapsed"] for r in reps)) / elapsed},
            "unit": "LM iterations/s",
            "n_gpus": args.gpus if sp_devices is not None else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 (state, observations and costs f64)" if args.mixed else "f64" if DTYPE == np.float64 else "f32",
            "data": data,
            "config": {
                "workload": f"BAL {args.workload} ({data}): {prob.n_cams} cams, {prob.n_lms} lms, {prob.n_obs} obs, "
                            f"solver={args.solver_type}, {args.preconditioner}, Huber(1), {'mixed float32/float64' if args.mixed else 'float64' if DTYPE == np.float64 else 'float32'}",
                "parallelism": (f"landmarks sharded over {info['nranks']} GPU(s)"
                                + (" behind ONE handle of one process (rba_create_sharded)" if sp_devices is not None else "")
                                + f"; library communicator: transport={info['transport']}, nranks={info['nranks']}"
                                + ("" if info['nranks'] == 1 else "; all-reduce of camera-sized vectors / the assembled matrix")),
                "comm_per_step": None if info['nranks'] == 1 else {
                    "all_reduces": (comm1["calls"] - comm0["calls"]) / args.steps,
                    "bytes": (comm1["bytes"] - comm0["bytes"]) / args.steps,
                    "ms": 1e3 * (comm1["seconds"] - comm0["seconds"]) / args.steps},
                "explicit_after": int(os.environ.get("RBA_EXPLICIT_AFTER", gpu_opts.explicit_after)),
                # which regime this workload is in: a banded reduced camera matrix that fits the
                # register files (long PCG solves = one persistent kernel) or a nearly dense one that streams from HBM
                "reduced_matrix": {"blocks": matrix_info["blocks_full"], "blocks_stored": matrix_info["blocks_stored"],
                                   "density": matrix_info["density"], "bytes": matrix_info["bytes_stored"],
                                   "resident_in_registers": bool(matrix_info["resident_in_registers"]),
                                   "persistent_workgroups": matrix_info["persistent_workgroups"]},
                "solves_persistent": pcg_cnt.get("solves_persistent", 0),
                "solves_assembled": pcg_cnt.get("assemblies", 0),
                "function_tolerance": 0.0,
                "compute_error_per_iteration": {
                    "launched": n_err / n_t,
                    "what": "cost evaluations LAUNCHED per timed step (rba_pcg_counters.cost_evaluations). The reference "
                            "evaluates twice per iteration (start of every outer iteration + after the step, "
                            "bal_bundle_adjustment.cpp:297-301, 417); here the evaluation at the start of an iteration "
                            "that follows an accepted step IS the trial evaluation of that step - same kernel, same "
                            "state, same fixed summation order, bit-identical - and is reused, not repeated"},
                "cg_iterations_per_step": sum(r.cg_iterations for r in timed) / max(1, len(timed)),
                "successful_steps": sum(r.step_is_successful for r in timed),
                "initial_cost": rows[0].cost,
                "final_cost": [r.cost for r in rows if r.step_is_successful][-1],
                "value_reference_semantics": ref_sem,
            },
            "roofline": {
                "kernel": ("k_pcgs_spmv (S*x on the explicit block-CSR reduced camera matrix; algorithmic bytes = "
                           "81 s + 4 per block + 2 x 9 n_c s)" if sc else
                           "k_hx_implicit_lds (+ k_hx_implicit_wide for 32 < k <= 112): H*x from the QR factors, workgroup-private "
                           "double y in LDS where 9 n_c doubles fit, else k_hx_implicit; algorithmic bytes = SURVEY.md 8d "
                           "implicit-Q formula"),
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None,
                # context, not the roofline: what a plain 16-byte-load read kernel streams on this part
                # (scripts/microbench/stream_read.hip, profiles/r5_microbench_stream_read.txt: 6.5 TB/s; copy 5.9)
                "frac_of_measured_stream_read": achieved / 6500.0 if achieved else None,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": hx_bytes_per_launch,
                "avg_launch_ms": avg_hx * 1e3 if avg_hx else None,
                "launches_timed": hx_calls,
                "launch_timing": hx_pass,
                "whole_iteration": {
                    "what": "EVERYTHING an LM iteration launches: stage 1, stage 2, the PCG (executed products on "
                            "either operator, assemblies, vector work - rba_get_pcg_counters), back-substitution, "
                            "cost evaluations: the bytes each must move in this layout (rba_get_byte_model) / their "
                            "summed stage times (rba_iter_timings: device clock stamps at the stage boundaries inside "
                            "rba_lm_step) / 8 TB/s",
                    "frac": frac(b_iter + b_pcg, t_s1 + t_s2 + t_bs + t_err + t_pcg),
                    "bytes_per_step": (b_iter + b_pcg) / n_t,
                    "frac_without_pcg": frac(b_iter, t_s1 + t_s2 + t_bs + t_err),
                },
                "stages": stages,
            },
        }
        if (world == 1 and sp_devices is None and args.workload == "venice-1778" and data == "synthetic" and not args.no_dense_companion
                and not args.use_double and args.solver_type == "SQUARE_ROOT"):
            out["config"]["value_dense_covisibility"] = dense_companion(args)
        if world == 1 and sp_devices is None and args.cpu_baseline_iters > 0:
            try:
                out["cpu_baseline"] = cpu_baseline(prob, args.cpu_baseline_iters, rows)
            except Exception as e:  # the baseline must never take the GPU number down
                log(f"[cpu_baseline] failed: {e!r}")
                out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
