"""Landmark sharding of the HIP solver itself, two ranks sharing ONE GPU.

RCCL refuses two ranks on one device, so the ranks all-reduce through the
library's callback transport (`rba_comm_init_callback`, here backed by gloo).
Everything else is the production multi-GPU path: per-rank landmark shards, the
all-reduce points of SURVEY.md §8e inside the library, the lambda*I bookkeeping,
the lazily polled PCG state that must stay identical on all ranks. The "-split-products" cases force what large
problems do by themselves (Solver::decide_product_split): after two matrix-free iterations a solve runs on the assembled
matrix with each rank multiplying HALF of the block work items and the product vector all-reduced per iteration."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, dtype_name, env, opts, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.update(env)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bench import shard_ranges, take_landmarks
    from rootba_amd import _lib as L
    if os.environ.get("RBA_EMU") == "1":  # development runs on the CPU execution harness (tests/hipemu), as tests/conftest.py
        sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
        import build_emu
        L.LIB_PATH = build_emu.LIB
    from rootba_amd import problem as P
    from rootba_amd.linearizor import LinearizorHIP
    dtype = "mixed" if dtype_name == "mixed" else np.dtype(dtype_name)
    vec_dtype = np.float32 if dtype_name == "mixed" else dtype
    prob = P.preprocess(P.named_synthetic("ladybug-49"), translation_sigma=0.5, point_sigma=0.5)
    lo, hi = shard_ranges(prob.obs_per_lm(), world)[rank]
    g = LinearizorHIP(take_landmarks(prob, lo, hi), dtype,
                      L.default_options(robust_norm=1, max_num_iterations=6, **opts), device=0)

    def allreduce(arr, op):
        t = torch.from_numpy(arr)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)

    g.comm_init_callback(rank, world, allreduce)
    err = g.compute_error()
    assert g.linearize() == 0
    b, blocks = g.stage2(0.1)
    x = np.random.default_rng(0).uniform(-1, 1, 9 * prob.n_cams).astype(vec_dtype)
    hx = g.right_multiply(x)
    inc, cg = g.solve(1e-4)
    l_diff = g.apply(inc)
    cams, _ = g.get_state()
    g2 = LinearizorHIP(take_landmarks(prob, lo, hi), dtype,
                       L.default_options(robust_norm=1, max_num_iterations=6, **opts), device=0)
    g2.comm_init_callback(rank, world, allreduce)
    log, term = g2.optimize_lm()
    ret[rank] = dict(err=(err.all_error, err.all_num_obs), b=b, blocks=blocks, hx=hx, inc=inc,
                     cg=cg.num_iterations, l_diff=l_diff, cams=cams,
                     lm=[(r.cost, r.cg_iterations, r.step_is_successful) for r in log], term=term,
                     pcg=g2.pcg_counters())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dtype,env", [(np.float64, {}), (np.float32, {}),
                                       (np.float32, {"RBA_HX_LDS": "2", "RBA_HX_WIN": "9"}), ("mixed", {}),
                                       (np.float64, {"RBA_PCG_SPLIT": "1", "RBA_EXPLICIT_AFTER": "2"}),
                                       (np.float32, {"RBA_PCG_SPLIT": "1", "RBA_EXPLICIT_AFTER": "2"}),
                                       (np.float32, {"RBA_PCG_SPLIT": "1", "RBA_EXPLICIT_AFTER": "2",
                                                     "RBA_HALF_LOWER_MAX": "3"}),
                                       (np.float64, {"_opts": "schur_complement"}),
                                       (np.float32, {"_opts": "schur_complement"}),
                                       (np.float32, {"_opts": "schur_complement_power"})],
                         ids=["float64", "float32", "float32-lds-window", "mixed", "float64-split-products",
                              "float32-split-products", "float32-split-products-heavy-rows",
                              "float64-schur-complement", "float32-schur-complement", "float32-schur-complement-power"])
def test_two_ranks_one_gpu_match_unsharded(dtype, env, monkeypatch):
    import torch  # noqa: F401
    import torch.multiprocessing as mp
    from conftest import rel_err
    from rootba_amd import _lib as L
    from rootba_amd import problem as P
    from rootba_amd.linearizor import LinearizorHIP
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    env = dict(env)
    # the explicit Schur-complement backend sharded over landmarks (the reference's LinearizationSC sums the same
    # per-landmark terms, linearization_sc.hpp:232-347): the ranks' sums of S and b are all-reduced in the united
    # structure, the PCG on S is replicated
    opts = {"schur_complement": dict(solver_type=1),
            "schur_complement_power": dict(solver_type=1, preconditioner_type=2, power_order=4)}.get(env.pop("_opts", ""), {})
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    mixed = isinstance(dtype, str)
    vec_dtype = np.float32 if mixed else dtype
    mp.spawn(_worker, args=(world, 29600 + os.getpid() % 2000, dtype if mixed else np.dtype(dtype).name, env, opts, ret),
             nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    # replicated quantities are bit-identical on both ranks
    for key in ("b", "blocks", "hx", "inc", "cams"):
        assert np.array_equal(r0[key], r1[key]), key
    assert r0["cg"] == r1["cg"] and r0["lm"] == r1["lm"] and r0["term"] == r1["term"]
    if "RBA_PCG_SPLIT" in env:  # (the split products did run: most iterations of the LM run's solves used them)
        assert r0["pcg"]["products_assembled"] > r0["pcg"]["products_matrix_free"] > 0, r0["pcg"]

    prob = P.preprocess(P.named_synthetic("ladybug-49"), translation_sigma=0.5, point_sigma=0.5)
    g = LinearizorHIP(prob, dtype, L.default_options(robust_norm=1, max_num_iterations=6, **opts))
    tol = 1e-5 if vec_dtype == np.float32 else 1e-12
    err = g.compute_error()
    assert r0["err"][1] == err.all_num_obs and \
        abs(r0["err"][0] - err.all_error) < (1e-12 if mixed else tol) * err.all_error  # mixed: the cost is double
    assert g.linearize() == 0
    b, blocks = g.stage2(0.1)
    assert rel_err(r0["b"], b) < tol and rel_err(r0["blocks"], blocks) < tol
    x = np.random.default_rng(0).uniform(-1, 1, 9 * prob.n_cams).astype(vec_dtype)
    assert rel_err(r0["hx"], g.right_multiply(x)) < tol
    inc, cg = g.solve(1e-4)
    assert abs(cg.num_iterations - r0["cg"]) <= (1 if vec_dtype == np.float32 else 0)
    assert rel_err(r0["inc"], inc) < (1e-3 if vec_dtype == np.float32 else 1e-9)
    g3 = LinearizorHIP(prob, dtype, L.default_options(robust_norm=1, max_num_iterations=6, **opts))
    log, term = g3.optimize_lm()
    assert len(log) == len(r0["lm"])
    for a, (cost, cgi, ok) in zip(log, r0["lm"]):
        assert a.step_is_successful == ok
        # float32: two summation orders (shards, atomics) on truncated PCG solves - the late iterations
        # agree to the few 1e-5 that separate any two float32 runs of this problem
        assert abs(a.cost - cost) <= ((1e-5 if a.iteration <= 2 else 5e-5) if vec_dtype == np.float32 else 1e-9) * cost


def test_rccl_call_path_with_one_rank(ladybug_problem):
    """A one-rank RCCL communicator: dlopen, ncclGetUniqueId, ncclCommInitRank and
    every ncclAllReduce site of the library run on this single-GPU box; the
    results must equal the run without a communicator."""
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    opts = dict(robust_norm=1, max_num_iterations=4)
    a = LinearizorHIP(ladybug_problem, np.float32, L.default_options(**opts))
    b = LinearizorHIP(ladybug_problem, np.float32, L.default_options(**opts))
    uid = LinearizorHIP.comm_unique_id()
    assert len(uid) == 128
    b.comm_init(0, 1, uid)
    assert a.linearize() == 0 and b.linearize() == 0
    ba, bla = a.stage2(0.1)
    bb, blb = b.stage2(0.1)
    assert np.array_equal(ba, bb) and np.array_equal(bla, blb)
    la, _ = a.optimize_lm()
    lb, _ = b.optimize_lm()
    assert len(la) == len(lb)
    # two float32 runs differ by the order of the atomic scatter-adds: cost resolution ~1e-6 (see
    # test_lm_trajectory_matches_oracle)
    assert np.allclose([r.cost for r in la], [r.cost for r in lb], rtol=3e-6)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dtype", [np.float64, np.float32, "mixed"], ids=["float64", "float32", "mixed"])
@pytest.mark.parametrize("n_dev", [2, 3])
def test_single_process_sharded_handle(dtype, n_dev):
    """rba_create_sharded (ONE process, n_gpus devices behind one handle - the entry the reference's one-process
    driver can reach several GPUs through): the library shards the landmarks itself and fans the calls out on a host
    thread per device. Here the devices repeat (one GPU: the ranks exchange through host memory); every call of the
    Linearizor interface and whole LM runs must equal the unsharded handle's, state and per-landmark outputs in the
    caller's landmark order."""
    import torch  # noqa: F401
    from conftest import rel_err
    from rootba_amd import _lib as L
    from rootba_amd import problem as P
    from rootba_amd.linearizor import LinearizorHIP
    mixed = isinstance(dtype, str)
    vec_dtype = np.float32 if mixed else dtype
    prob = P.preprocess(P.named_synthetic("ladybug-49"), translation_sigma=0.5, point_sigma=0.5)
    opts = dict(robust_norm=1, max_num_iterations=6)
    g = LinearizorHIP(prob, dtype, L.default_options(**opts))
    s = LinearizorHIP(prob, dtype, L.default_options(**opts), devices=[0] * n_dev)
    cuts = s.shard_ranges()
    assert len(cuts) == n_dev + 1 and cuts[0] == 0 and cuts[-1] == prob.n_lms and all(b > a for a, b in zip(cuts, cuts[1:]))
    # (balanced by bytes per landmark: 120 per observation + 100)
    w = 120.0 * prob.obs_per_lm() + 100.0
    shares = [w[a:b].sum() / w.sum() for a, b in zip(cuts, cuts[1:])]
    assert max(shares) - min(shares) < 0.02, shares
    tol = 1e-5 if vec_dtype == np.float32 else 1e-12
    eg, es = g.compute_error(), s.compute_error()
    assert es.all_num_obs == eg.all_num_obs and abs(es.all_error - eg.all_error) < (1e-12 if mixed else tol) * eg.all_error
    assert g.linearize() == 0 and s.linearize() == 0
    assert rel_err(s.jl_col_scale(), g.jl_col_scale()) < tol  # (per landmark, in the caller's order)
    (bg, kg), (bs, ks) = g.stage2(0.1), s.stage2(0.1)
    assert rel_err(bs, bg) < tol and rel_err(ks, kg) < tol
    x = np.random.default_rng(0).uniform(-1, 1, 9 * prob.n_cams).astype(vec_dtype)
    assert rel_err(s.right_multiply(x), g.right_multiply(x)) < tol
    (ig, cg), (is_, cs) = g.solve(1e-4), s.solve(1e-4)
    assert abs(cg.num_iterations - cs.num_iterations) <= (1 if vec_dtype == np.float32 else 0)
    assert rel_err(is_, ig) < (1e-3 if vec_dtype == np.float32 else 1e-9)
    ld_g, ld_s = g.apply(ig), s.apply(ig)
    assert abs(ld_g - ld_s) <= (1e-4 if vec_dtype == np.float32 else 1e-10) * abs(ld_g)
    (cg_, lg_), (cs_, ls_) = g.get_state(), s.get_state()
    assert rel_err(cs_, cg_) < tol and rel_err(ls_, lg_) < tol and ls_.shape == lg_.shape
    s.set_state(prob.cams, prob.lms)  # (the caller's arrays, sliced per device inside)
    g.set_state(prob.cams, prob.lms)
    lg, tg = g.optimize_lm()
    lsh, ts = s.optimize_lm()
    assert tg == ts and len(lg) == len(lsh)
    for a, b in zip(lg, lsh):
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= ((1e-5 if a.iteration <= 2 else 5e-5) if vec_dtype == np.float32 else 1e-9) * a.cost
    _, lms_g = g.get_state()
    _, lms_s = s.get_state()
    assert rel_err(lms_s, lms_g) < (1e-4 if vec_dtype == np.float32 else 1e-9)
    info = s.comm_info()
    assert info["nranks"] == n_dev and info["transport"] == "callback", info  # (the devices repeat: host memory)
    s.close()
    g.close()


@pytest.mark.parametrize("split", [False, True])
def test_single_process_sharded_handle_over_distinct_devices(small_problem, split, monkeypatch):
    """The same entry over DISTINCT devices: one RCCL communicator inside the process (ncclCommInitRank from the ranks'
    own host threads). Needs two devices - the MI355X development boxes have one; on the CPU execution harness of
    tests/hipemu (eight stand-in devices, file-based stand-in for RCCL) this is where the path runs."""
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    if L.device_count() < 2:
        pytest.skip("one device: the RCCL transport of rba_create_sharded needs two")
    # (a fixed operator switch: the default is a MEASURED break-even - assembly time / product time -, which differs
    #  between the two handles. Three iterations: the fourth solve of this run, 47 PCG iterations, amplifies the 1e-15
    #  between two summation orders to 1e-7 of the cost even in float64 - measured, with identical counters on both
    #  handles; the first three agree to 1e-14.)
    opts = dict(robust_norm=1, max_num_iterations=3, explicit_after=3)
    g = LinearizorHIP(small_problem, np.float64, L.default_options(**opts))
    if split:
        # products on the assembled matrix split over the ranks: every rank receives the sums of ITS range of the matrix
        # only (Solver::reduce_ranges: one ncclReduce per root in a group instead of an all-reduce of the whole matrix)
        monkeypatch.setenv("RBA_PCG_SPLIT", "1")
    s = LinearizorHIP(small_problem, np.float64, L.default_options(**opts), devices=[0, 1])
    assert s.comm_info()["transport"] == "rccl" and s.comm_info()["nranks"] == 2
    a, ta = g.optimize_lm()
    b, tb = s.optimize_lm()
    if split:
        assert s.pcg_counters()["assemblies"] > 0
    assert ta == tb and len(a) == len(b)
    for x, y in zip(a, b):
        assert x.step_is_successful == y.step_is_successful and abs(x.cost - y.cost) <= 1e-9 * x.cost


def test_persistent_pcg_fallback_is_collective(ladybug_problem, monkeypatch):
    """A sharded run replicates the PCG on the assembled matrix on every rank; should ONE rank's persistent kernel give up
    waiting (kernels_pcgp.hpp: bounded polling), all ranks must continue alike - the ranks whose kernel finished go back
    to the solve's entry state and everybody takes the two-launch path (Solver::pcg). Test hook RBA_PCGP_TEST_GIVE_UP=1
    on a one-rank RCCL communicator: the kernel finishes, the rank behaves as if it had given up; the LM run must be,
    bit by bit, the run with RBA_PCG_PERSISTENT=0 (deterministic mode: no floating-point atomics in either)."""
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    opts = dict(robust_norm=1, max_num_iterations=6, function_tolerance=0.0)
    monkeypatch.setenv("RBA_DETERMINISTIC", "1")
    runs = []
    for env in ({"RBA_PCGP_TEST_GIVE_UP": "1"}, {"RBA_PCG_PERSISTENT": "0"}):
        for k in ("RBA_PCGP_TEST_GIVE_UP", "RBA_PCG_PERSISTENT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = LinearizorHIP(ladybug_problem, np.float32, L.default_options(**opts))
        g.comm_init(0, 1, LinearizorHIP.comm_unique_id())
        rows, _ = g.optimize_lm()
        runs.append((rows, g.get_state(), g.pcg_counters()))
        g.close()
    (ra, sa, ca), (rb, sb, cb) = runs
    assert ca["products_assembled"] > 0 and ca["solves_persistent"] == 0 and cb["solves_persistent"] == 0, (ca, cb)
    assert [r.cg_iterations for r in ra] == [r.cg_iterations for r in rb]
    assert [r.cost for r in ra] == [r.cost for r in rb]
    assert np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1])


@pytest.mark.timeout(120)
def test_sharded_handle_fails_instead_of_hanging_when_one_rank_throws(monkeypatch):
    """a rank of a sharded handle that throws while its peers are inside a collective must not leave them
    waiting for ever. Test hook RBA_TEST_FAIL_RANK=1: rank 1 throws at its next linearisation while rank 0 enters the
    all-reduce of the failure flag; the call returns an error (the waiter of the host transport is woken and fails, an
    RCCL communicator would be aborted), and every later call on the handle fails at once."""
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd import problem as P
    from rootba_amd.linearizor import LinearizorHIP
    prob = P.preprocess(P.named_synthetic("ladybug-49"), translation_sigma=0.5, point_sigma=0.5)
    monkeypatch.setenv("RBA_TEST_FAIL_RANK", "1")
    s = LinearizorHIP(prob, np.float32, L.default_options(robust_norm=1), devices=[0, 0])
    assert s.compute_error().all_error > 0  # (collectives work until the failure)
    with pytest.raises(RuntimeError, match="fails on purpose|aborted|callback failed"):
        s.linearize()
    with pytest.raises(RuntimeError, match="broken"):
        s.compute_error()
    s.close()


@pytest.mark.timeout(300)
def test_collectives_per_lm_iteration_of_a_sharded_run():
    """a sharded LM iteration enters ONE collective per reduction the reference performs over
    landmarks (linearization_qr.hpp:677-683 Jp_diag2, :770-775 b + block diagonal, :406-429 one per product) plus ONE for
    everything the end of the iteration hands to the host - the eight cost sums of the trial point, l_diff and the
    failure bits of linearisation / back-substitution / block inversion travel in one block (rba::kEndRed). A floor
    iteration (two PCG iterations) is 5 collectives (<= 8 asked; rounds 1-5: 8 + the flag reductions)."""
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd import problem as P
    from rootba_amd.linearizor import LinearizorHIP
    prob = P.preprocess(P.named_synthetic("ladybug-49"), translation_sigma=0.5, point_sigma=0.5)
    s = LinearizorHIP(prob, np.float32, L.default_options(robust_norm=1, max_num_iterations=6, function_tolerance=0.0,
                                                         explicit_after=0), devices=[0, 0])
    s.lm_begin()
    s.lm_step()
    calls = s.comm_stats()["calls"]
    seen = 0
    more = True
    while more:
        row, more = s.lm_step()
        now = s.comm_stats()["calls"]
        products = row.cg_iterations + row.cg_iterations // 10
        # Jp_diag2 (only when the step linearises: after an accepted one), [b | blocks], the products, the end
        assert now - calls <= 3 + products, (row.iteration, row.cg_iterations, now - calls)
        assert now - calls >= 2 + products
        calls = now
        seen += 1
    assert seen >= 4
    s.close()
