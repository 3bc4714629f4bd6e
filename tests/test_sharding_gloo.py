"""Multi-GPU decomposition, exercised on CPU with world_size = 2 over gloo.

The HIP library shards LANDMARKS across ranks and all-reduces the camera-sized
accumulators: Jp_diag2 after stage 1, [b | block diagonal] after
stage 2 (the pose-damping lambda*I must count once), H*x per PCG iteration (the
lambda*x term is added after the reduction), 7 residual sums, l_diff. This test
replays exactly that decomposition with the CPU oracle standing in for the
per-rank kernels and checks that the reduced quantities equal the unsharded
ones. (RCCL itself cannot be exercised without GPUs; the rank-local math and
the reduction points are what is covered here.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAMBDA = 0.1


def _allreduce(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
    dist.all_reduce(t)
    return t.numpy()


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bench import shard_ranges, take_landmarks
    from oracle import oracle as O
    from rootba_amd import problem as P
    prob = P.preprocess(P.synthetic_problem(40, 400, 1700, seed=7), seed=7, translation_sigma=0.5,
                        point_sigma=0.5)
    lo, hi = shard_ranges(prob.obs_per_lm(), world)[rank]
    local = take_landmarks(prob, lo, hi)
    opts = O.default_options(robust_norm=1, num_threads=1)
    o = O.Oracle(local, np.float64, opts)
    ri = o.compute_error()
    err = _allreduce([ri.all_error, ri.all_num_obs, ri.valid_error])
    rc, d2, _ = o.stage1()
    d2 = _allreduce(d2)
    scaling = 1.0 / (o_eps() + np.sqrt(d2))
    o.set_pose_damping(LAMBDA)
    b, blocks = o.stage2(LAMBDA, scaling)
    b, blocks = _allreduce(b), _allreduce(blocks)
    blocks[:, np.arange(9), np.arange(9)] -= LAMBDA * (world - 1)  # lambda*I counted once
    x = np.random.default_rng(0).uniform(-1, 1, 9 * prob.n_cams)
    hx = _allreduce(o.right_multiply(x) - LAMBDA * x) + LAMBDA * x
    inc = np.random.default_rng(1).uniform(-1, 1, 9 * prob.n_cams) * 0.01
    l_diff = _allreduce([o.back_substitute(inc)])[0]
    lms = o.get_state()[1]
    if rank == 0:
        ret.update(err=err, d2=d2, b=b, blocks=blocks, hx=hx, l_diff=l_diff)
    ret[f"lms{rank}"] = (lo, hi, lms)
    dist.barrier()
    dist.destroy_process_group()


def o_eps():
    return 1e-5  # epsilonSqrt<double>


@pytest.mark.timeout(300)
def test_landmark_sharding_reductions_match_unsharded():
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from rootba_amd import problem as P
    world = 2
    port = 29500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)

    prob = P.preprocess(P.synthetic_problem(40, 400, 1700, seed=7), seed=7, translation_sigma=0.5,
                        point_sigma=0.5)
    o = O.Oracle(prob, np.float64, O.default_options(robust_norm=1))
    ri = o.compute_error()
    assert np.allclose(ret["err"], [ri.all_error, ri.all_num_obs, ri.valid_error], rtol=1e-12)
    rc, d2, _ = o.stage1()
    assert np.allclose(ret["d2"], d2, rtol=1e-12)
    scaling = 1.0 / (1e-5 + np.sqrt(d2))
    o.set_pose_damping(LAMBDA)
    b, blocks = o.stage2(LAMBDA, scaling)
    assert np.allclose(ret["b"], b, rtol=1e-9, atol=1e-9 * np.abs(b).max())
    assert np.allclose(ret["blocks"], blocks, rtol=1e-9, atol=1e-9 * np.abs(blocks).max())
    x = np.random.default_rng(0).uniform(-1, 1, 9 * prob.n_cams)
    hx = o.right_multiply(x)
    assert np.allclose(ret["hx"], hx, rtol=1e-9, atol=1e-9 * np.abs(hx).max())
    inc = np.random.default_rng(1).uniform(-1, 1, 9 * prob.n_cams) * 0.01
    l = o.back_substitute(inc)
    assert abs(ret["l_diff"] - l) < 1e-9 * abs(l)
    lms = o.get_state()[1]
    for r in range(world):
        lo, hi, part = ret[f"lms{r}"]
        assert np.allclose(part, lms[lo:hi], rtol=1e-10, atol=1e-10)


def test_shard_ranges_balance_bytes():
    sys.path.insert(0, ROOT)
    from bench import shard_ranges
    k = np.random.default_rng(0).geometric(0.25, 10000) + 1
    for n in (1, 2, 4, 8):
        r = shard_ranges(k, n)
        assert r[0][0] == 0 and r[-1][1] == k.size and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        # (the layout's bytes per landmark: ~120 per observation + ~100, as rba_create_sharded balances inside the library)
        w = np.array([(120.0 * k[a:b].astype(float) + 100.0).sum() for a, b in r])
        assert w.max() / w.mean() < 1.01


def test_shard_ranges_never_empty():
    sys.path.insert(0, ROOT)
    from bench import shard_ranges
    k = np.array([2, 2, 2, 2000, 2, 2, 2, 2])  # one landmark holds almost all the bytes
    r = shard_ranges(k, 8)
    assert all(b > a for a, b in r) and r[0][0] == 0 and r[-1][1] == k.size
    with pytest.raises(SystemExit):
        shard_ranges(k[:3], 8)


def test_bench_refuses_a_world_size_mismatch():
    """`bench.py --gpus N` must never print a line for a different number of ranks."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and p.stdout.strip() == ""
