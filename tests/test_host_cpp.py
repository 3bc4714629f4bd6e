"""C++17 host layer (rootba_amd/csrc/host): BAL loader + reference preprocessing
on CPU, and the `bal_qr_hip` CLI end to end on the GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

from rootba_amd import build
from rootba_amd import problem as P


@pytest.fixture(scope="module")
def app():
    build.build()
    return build.APP


@pytest.fixture(scope="module")
def bal_file(tmp_path_factory):
    raw = P.synthetic_problem(16, 150, 600, seed=9)
    path = str(tmp_path_factory.mktemp("bal") / "problem-16-150-pre.txt")
    P.write_bal(raw, path)
    return path, raw


def test_cli_help(app):
    out = subprocess.run([app, "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--input" in out.stdout and "--max-num-iterations" in out.stdout


def test_cpp_loader_and_preprocessing_match_numpy_mirror(app, bal_file):
    path, raw = bal_file
    thr = 60.0  # on the normalised scale (100) this drops a good part of the observations
    out = subprocess.run([app, "--input", path, "--dry-run", "--init-depth-threshold", str(thr)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    info = json.loads(out.stdout.strip().splitlines()[-1])
    ref = P.filter_obs(P.normalize(P.read_bal(path), 100.0), thr)
    assert ref.n_obs < raw.n_obs
    assert (info["num_cameras"], info["num_landmarks"], info["num_observations"]) == (ref.n_cams, ref.n_lms, ref.n_obs)
    assert np.allclose(info["landmark_sum"], ref.lms.sum(0), rtol=1e-9, atol=1e-6)
    # reduced-camera-system sparsity (reference bal_problem.cpp:647-712)
    mask = np.zeros((ref.n_cams, ref.n_cams), bool)
    off = ref.lm_obs_offsets
    for l in range(ref.n_lms):
        c = ref.obs_cam_idx[off[l]:off[l + 1]]
        mask[np.ix_(c, c)] = True
    np.fill_diagonal(mask, True)
    assert np.isclose(info["rcs_sparsity"], 1.0 - mask.sum() / ref.n_cams ** 2, atol=1e-12)
    q = np.array(info["cam0"][:4])
    assert np.allclose(P.quat_to_rot(q), P.quat_to_rot(ref.cams[0, :4]), atol=1e-9)
    assert np.allclose(info["cam0"][4:], ref.cams[0, 4:7], rtol=1e-9, atol=1e-7)


def test_number_parser_matches_strtod(app):
    """The loader's own decimal parser (Clinger / x87 fast paths, strtod otherwise) is bit-exact
    against strtod, i.e. reads what the reference's fscanf("%lf") reads."""
    out = subprocess.run([app, "--self-test-parser", "200000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["tokens"] > 1_000_000 and info["mismatches"] == 0


def _dry(app, path, *extra):
    return subprocess.run([app, "--input", path, "--dry-run", "--no-normalize", *extra], capture_output=True, text=True)


def _obs_checksum(p):
    off = p.lm_obs_offsets
    pos = np.arange(p.n_obs) - np.repeat(off[:-1], np.diff(off)) + 1
    return float(np.sum(pos * ((p.obs_cam_idx + 1.0) * p.obs_xy[:, 0] + p.obs_xy[:, 1])))


def test_loader_any_observation_order_and_thread_count(app, bal_file, tmp_path):
    """Observations may come in any order in the file (the reference buckets them in a std::map per
    landmark): shuffled lines load to the same CSR as the sorted file, for any number of parser threads."""
    path, raw = bal_file
    ref = P.read_bal(path)
    lines = open(path).read().split("\n")
    n_obs = raw.n_obs
    rng = np.random.default_rng(3)
    perm = rng.permutation(n_obs)
    shuffled = str(tmp_path / "shuffled.txt")
    with open(shuffled, "w") as f:
        # also exercise tabs, CRLF and several tokens per line in the parameter section
        f.write(lines[0] + "\r\n")
        f.write("\n".join(lines[1 + i].replace(" ", "\t") for i in perm) + "\n")
        f.write(" ".join(lines[1 + n_obs:]) + "\n")
    want = _obs_checksum(ref)
    for threads in ("1", "3", "8"):
        for fpath in (path, shuffled):
            env = dict(os.environ, RBA_HOST_THREADS=threads)
            out = subprocess.run([app, "--input", fpath, "--dry-run", "--no-normalize"], capture_output=True,
                                 text=True, env=env)
            assert out.returncode == 0, out.stderr
            info = json.loads(out.stdout.strip().splitlines()[-1])
            assert (info["num_cameras"], info["num_landmarks"], info["num_observations"]) == (ref.n_cams, ref.n_lms, ref.n_obs)
            assert np.isclose(info["obs_checksum"], want, rtol=1e-12)
            assert np.allclose(info["landmark_sum"], ref.lms.sum(0), rtol=1e-12)


def test_loader_rejects_malformed_files(app, bal_file, tmp_path):
    """Duplicate (camera, landmark) pairs, out-of-range indices, non-numeric and missing tokens are
    fatal like in the reference (bal_problem.cpp:229-230, :199-205)."""
    path, raw = bal_file
    lines = open(path).read().split("\n")

    def variant(name, edit):
        ls = list(lines)
        edit(ls)
        fpath = str(tmp_path / name)
        open(fpath, "w").write("\n".join(ls))
        return fpath

    def dup(ls):
        ls[2] = ls[1]
    def bad_cam(ls):
        ls[1] = "9999 " + ls[1].split(" ", 1)[1]
    def bad_token(ls):
        ls[5] = ls[5].replace(ls[5].split()[2], "1.2.3")
    def float_index(ls):
        ls[3] = "0.5 " + ls[3].split(" ", 1)[1]
    def truncated(ls):
        del ls[-5:]

    for name, edit in (("dup", dup), ("bad_cam", bad_cam), ("bad_token", bad_token), ("float_index", float_index),
                       ("truncated", truncated)):
        out = _dry(app, variant(name, edit))
        assert out.returncode == 2, (name, out.stdout, out.stderr)
        assert "FATAL" in out.stderr


# BaLog::BaIteration members, reference src/rootba/bal/ba_log.hpp:139-237 (one array each in ba_log.json)
BA_ITERATION_KEYS = """iteration linear_solver_type step_is_valid step_is_nonmonotonic step_is_successful num_obs
num_obs_valid num_obs_valid_change cost cost_change cost_valid cost_valid_change cost_avg_valid cost_avg_valid_change
grad_projected_norm grad_projected_max_norm grad_norm grad_max_norm residual_block_mean residual_block_valid_mean
step_norm relative_decrease trust_region_radius linear_solver_iterations iteration_time cumulative_time logging_time
step_solver_time residual_evaluation_time jacobian_evaluation_time scale_landmark_jacobian_time perform_qr_time
stage1_time scale_pose_jacobian_time landmark_damping_time compute_preconditioner_time compute_gradient_time
stage2_time prepare_time solve_reduced_system_time back_substitution_time update_cameras_time resident_memory
resident_memory_peak""".split()
# BaLog::BaSolver / ProblemInfo / PipelineTiming members (ba_log.hpp:45-137)
BA_SOLVER_KEYS = """solver_type termination_type message num_successful_steps num_unsuccessful_steps
logging_time_in_seconds preprocessor_time_in_seconds minimizer_time_in_seconds postprocessor_time_in_seconds
total_time_in_seconds linear_solver_time_in_seconds num_linear_solves residual_evaluation_time_in_seconds
num_residual_evaluations jacobian_evaluation_time_in_seconds num_jacobian_evaluations num_threads_given
num_threads_used num_threads_available resident_memory_peak""".split()
BA_PROBLEM_KEYS = "type input_path num_cameras num_landmarks num_observations rcs_sparsity per_lm_obs per_host_lms".split()


def check_ba_log_layout(log):
    """The flat layout of reference ba_log.cpp:62-149 that python/rootba/log.py + plot_logs.py consume."""
    assert log["_type"] == "rootba"
    n = len(log["iteration"])
    for k in BA_ITERATION_KEYS:
        assert k in log and len(log[k]) == n, k
    assert set(log) == set(BA_ITERATION_KEYS) | {"_type", "_static"}
    st = log["_static"]
    assert set(st) == {"problem_info", "timing", "solver"}
    assert set(BA_SOLVER_KEYS) <= set(st["solver"])
    assert set(st["problem_info"]) == set(BA_PROBLEM_KEYS)
    assert set(st["problem_info"]["per_lm_obs"]) == {"mean", "min", "max", "stddev"}
    assert set(st["timing"]) == {"total", "load", "preprocess", "optimize", "postprocess"}
    assert not st["solver"]["solver_type"].endswith("_ceres")
    assert all(isinstance(v, bool) for v in log["step_is_successful"])


def test_ba_log_layout_and_rejected_step_rows(app, tmp_path):
    path = str(tmp_path / "ba_log.json")
    assert subprocess.run([app, "--self-test-log", path]).returncode == 0
    log = json.load(open(path))
    check_ba_log_layout(log)
    # rejected step (iteration 2) repeats the previous row's cost columns (ba_log_utils.cpp:119-137)
    assert log["cost"] == [100.0, 60.0, 60.0, 50.0]
    assert log["cost_change"] == [0.0, -40.0, 0.0, -25.0]      # w.r.t. the previous summary, 75 -> 50
    assert log["step_is_successful"] == [True, True, False, True]
    assert log["step_norm"][2] == 0 and log["relative_decrease"][2] == 0
    assert log["num_obs_valid"] == [990, 991, 991, 993] and log["num_obs_valid_change"] == [0, 1, 0, 1]
    assert np.allclose(log["residual_block_mean"], [2.0, 2.001, 2.001, 2.003])
    assert np.allclose(log["step_solver_time"], 0.009)
    s = log["_static"]
    assert s["solver"]["num_successful_steps"] == 2 and s["solver"]["num_unsuccessful_steps"] == 1
    assert s["solver"]["num_linear_solves"] == 3 and s["timing"]["total"] == 6
    assert s["problem_info"]["input_path"] == 'self "test"'


def _ubjson_decode(buf):
    """Decoder for the subset nlohmann::json::to_ubjson(j) emits without size/type optimisation
    (UBJSON draft 12): Z T F i U I l L D S [ ] { }; object keys are size-prefixed strings without 'S'."""
    import struct
    pos = 0

    def take(n):
        nonlocal pos
        b = buf[pos:pos + n]
        assert len(b) == n
        pos += n
        return b

    def integer(marker):
        fmt = {b"i": ">b", b"U": ">B", b"I": ">h", b"l": ">i", b"L": ">q"}[marker]
        return struct.unpack(fmt, take(struct.calcsize(fmt)))[0]

    def value(marker=None):
        marker = marker or take(1)
        if marker == b"Z":
            return None
        if marker in (b"T", b"F"):
            return marker == b"T"
        if marker in b"iUIlL":
            return integer(marker)
        if marker == b"D":
            return struct.unpack(">d", take(8))[0]
        if marker == b"S":
            return take(integer(take(1))).decode()
        if marker == b"[":
            out = []
            while True:
                m = take(1)
                if m == b"]":
                    return out
                out.append(value(m))
        if marker == b"{":
            out = {}
            while True:
                m = take(1)
                if m == b"}":
                    return out
                key = take(integer(m)).decode()
                out[key] = value()
        raise AssertionError(f"unexpected UBJSON marker {marker!r} at {pos}")

    v = value()
    assert pos == len(buf)
    return v


def test_ba_log_ubjson_twin(app, tmp_path):
    """SaveLogFlag::UBJSON (reference ba_log.cpp:127-145): `<log>.ubjson` holds the same object as the JSON file,
    in nlohmann's to_ubjson byte layout (smallest integer type, big-endian doubles, sorted keys)."""
    path = str(tmp_path / "ba_log.json")
    assert subprocess.run([app, "--self-test-log", path]).returncode == 0
    log = json.load(open(path))
    raw = open(str(tmp_path / "ba_log.ubjson"), "rb").read()
    ub = _ubjson_decode(raw)
    check_ba_log_layout(ub)
    assert list(ub) == sorted(ub) and list(log) == sorted(log)  # std::map order of nlohmann::json
    assert ub == log
    # spot checks of the encoding itself: "_type" is the second key; small ints are int8, doubles 'D'
    assert raw[:1] == b"{" and raw[-1:] == b"}"
    assert b"i\x05_typeSi\x06rootba" in raw                      # key: size-prefixed, no 'S'; value: 'S' + size
    assert b"i\x09iteration[i\x00i\x01i\x02i\x03]" in raw         # small integers as int8
    assert b"i\x04cost[D" + __import__("struct").pack(">d", 100.0) in raw  # doubles big-endian


def test_cli_rejects_bad_input(app, tmp_path):
    assert subprocess.run([app, "--input", str(tmp_path / "missing.txt")], capture_output=True).returncode == 2
    assert subprocess.run([app, "--input", "x", "--preconditioner-type", "POWER_VARIABLE_PROJECTION"],
                          capture_output=True).returncode == 1


@pytest.mark.gpu
def test_bal_qr_hip_end_to_end(app, bal_file, tmp_path):
    """CLI (C++ host layer -> C ABI) and the Python binding run the same solve."""
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    path, _ = bal_file
    log_path = str(tmp_path / "ba_log.json")
    out = subprocess.run([app, "--input", path, "--max-num-iterations", "6", "--robust-norm", "HUBER",
                          "--log-path", log_path], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "Final Cost" in out.stdout and "Iteration 1" in out.stdout
    log = json.load(open(log_path))
    check_ba_log_layout(log)
    assert log["iteration"][0] == 0
    prob = P.normalize(P.read_bal(path), 100.0)
    g = LinearizorHIP(prob, np.float64, L.default_options(robust_norm=1, max_num_iterations=6))
    rows, _ = g.optimize_lm()
    assert len(rows) == len(log["iteration"])
    # rejected steps repeat the previous cost in the log (monotonic plots)
    want, last = [], None
    for r in rows:
        last = r.cost if (r.step_is_successful or last is None) else last
        want.append(last)
    assert np.allclose(want, log["cost"], rtol=1e-7)
    assert [r.cg_iterations for r in rows] == log["linear_solver_iterations"]
    assert log["num_obs"][0] == prob.n_obs and log["_static"]["problem_info"]["num_observations"] == prob.n_obs
    assert np.isclose(log["cumulative_time"][-1], log["_static"]["solver"]["minimizer_time_in_seconds"], rtol=0.2)


@pytest.mark.gpu
def test_bal_qr_hip_mixed_precision_and_unstaged_timers(app, bal_file, tmp_path):
    """`--mixed-precision --no-staged-execution`: the CLI runs RBA_MIXED (same costs as the Python binding's
    mixed run) and the log carries the reference's unstaged sub-stage timers."""
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    path, _ = bal_file
    log_path = str(tmp_path / "ba_log.json")
    out = subprocess.run([app, "--input", path, "--max-num-iterations", "5", "--robust-norm", "HUBER", "--use-double",
                          "--mixed-precision", "--no-staged-execution", "--log-path", log_path],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    log = json.load(open(log_path))
    check_ba_log_layout(log)
    prob = P.normalize(P.read_bal(path), 100.0)
    g = LinearizorHIP(prob, "mixed", L.default_options(robust_norm=1, max_num_iterations=5))
    rows, _ = g.optimize_lm()
    want, last = [], None
    for r in rows:
        last = r.cost if (r.step_is_successful or last is None) else last
        want.append(last)
    assert len(rows) == len(log["iteration"]) and np.allclose(want, log["cost"], rtol=1e-6)
    for key in ("jacobian_evaluation_time", "perform_qr_time", "scale_pose_jacobian_time", "scale_landmark_jacobian_time"):
        assert all(v > 0 for v in log[key][1:]), key
    # (timed inside scale_pose_jacobian_time: the damping rotations run in the per-observation pass of stage 2)
    assert all(v == 0 for v in log["landmark_damping_time"])
    # staged (default) run: the same columns are zero, as in the reference
    out = subprocess.run([app, "--input", path, "--max-num-iterations", "2", "--log-path", log_path],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    log = json.load(open(log_path))
    assert all(v == 0 for v in log["perform_qr_time"])
    # mixed precision without a double host problem is refused
    out = subprocess.run([app, "--input", path, "--no-use-double", "--mixed-precision"], capture_output=True, text=True)
    assert out.returncode != 0


GOLDEN_BAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_bal.txt")


def test_loaders_against_hand_derived_values(app):
    """tests/golden/tiny_bal.txt (2 cameras, 3 landmarks, 5 observations in scrambled order) against values derived BY
    HAND from the reference's conventions (bal_problem.cpp:190-282) - not from either loader: observations grouped by
    landmark and sorted by camera (std::map order), image y inverted, T_c_w = diag(1,-1,-1) * exp(rodrigues) and
    diag(1,-1,-1) * t, intrinsics (f, k1, k2) unchanged. Camera 0 has the Rodrigues vector (0, 0, pi/2)."""
    # --- numpy mirror -------------------------------------------------------------------------
    p = P.read_bal(GOLDEN_BAL)
    assert (p.n_cams, p.n_lms, p.n_obs) == (2, 3, 5)
    assert p.lm_obs_offsets.tolist() == [0, 2, 4, 5]
    assert p.obs_cam_idx.tolist() == [0, 1, 0, 1, 0]
    assert np.array_equal(p.obs_xy, [[1.0, -2.0], [10.5, 20.25], [0.25, 0.75], [7.0, -8.0], [-3.5, -4.5]])
    assert np.array_equal(p.lms, [[1.0, 1.0, 10.0], [-2.0, 0.5, 12.0], [3.0, -1.5, 8.0]])
    R0 = np.array([[0.0, -1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])  # diag(1,-1,-1) * rot_z(90 deg)
    R1 = np.diag([1.0, -1.0, -1.0])                                        # zero Rodrigues vector
    assert np.allclose(P.quat_to_rot(p.cams[0, :4]), R0, atol=1e-15)
    assert np.allclose(P.quat_to_rot(p.cams[1, :4]), R1, atol=1e-15)
    assert np.allclose(np.linalg.norm(p.cams[:, :4], axis=1), 1.0, atol=1e-15)
    assert np.array_equal(p.cams[0, 4:], [1.0, -2.0, -3.0, 500.0, 1e-2, -1e-3])
    assert np.array_equal(p.cams[1, 4:], [-4.0, -5.0, -6.5, 650.5, 0.0, 2.5e-4])
    # --- C++ loader (parallel tokeniser + own decimal parser), no normalisation, no filtering --------
    out = subprocess.run([app, "--input", GOLDEN_BAL, "--dry-run", "--no-normalize", "--init-depth-threshold", "0"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert (info["num_cameras"], info["num_landmarks"], info["num_observations"]) == (2, 3, 5)
    assert info["landmark_sum"] == [2.0, 0.0, 30.0]
    # sum over landmarks and their camera-sorted observations i = 1, 2, ... of i * ((cam + 1) * x + y), y inverted:
    # -1 + 82.5 + 1 + 12 - 8
    assert info["obs_checksum"] == 86.5
    assert info["rcs_sparsity"] == 0.0
    assert np.allclose(P.quat_to_rot(np.array(info["cam0"][:4])), R0, atol=1e-12)
    assert info["cam0"][4:] == [1.0, -2.0, -3.0]


# ---- the `.cereal` problem cache (reference BalProblem::save_rootba / load_rootba) ------------------------------
def _cereal_bytes(prob, translation_first=True, matrix_dims=False, intrinsics_dims=False, ftype=b"rootba::BalProblem",
                  version=b"1.0"):
    """The documented layout (rootba_amd/csrc/host/bal_problem.hpp), written independently with struct.pack."""
    import struct
    out = [struct.pack("<Q", len(ftype)), ftype, struct.pack("<Q", len(version)), version]
    dims = lambda on, r, c: struct.pack("<ii", r, c) if on else b""  # noqa: E731
    out.append(struct.pack("<Q", prob.n_cams))
    for cam in prob.cams:
        se3 = list(cam[4:7]) + list(cam[0:4]) if translation_first else list(cam[0:7])
        out.append(struct.pack("<7d", *se3) + dims(matrix_dims and intrinsics_dims, 3, 1) + struct.pack("<3d", *cam[7:10]))
    out.append(struct.pack("<Q", prob.n_lms))
    off = prob.lm_obs_offsets
    for l in range(prob.n_lms):
        out.append(dims(matrix_dims, 3, 1) + struct.pack("<3d", *prob.lms[l]) + struct.pack("<Q", off[l + 1] - off[l]))
        for o in range(off[l], off[l + 1]):
            out.append(struct.pack("<i", prob.obs_cam_idx[o]) + dims(matrix_dims, 2, 1) + struct.pack("<2d", *prob.obs_xy[o]))
    return b"".join(out)


def _stats(app, path, *extra):
    out = subprocess.run([app, "--input", path, "--dry-run", "--no-normalize", *extra], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    info = json.loads(out.stdout.strip().splitlines()[-1])
    return {k: info[k] for k in ("num_cameras", "num_landmarks", "num_observations", "obs_checksum", "landmark_sum", "cam0",
                                 "rcs_sparsity")}


def _same(a, b):
    assert (a["num_cameras"], a["num_landmarks"], a["num_observations"]) == (b["num_cameras"], b["num_landmarks"], b["num_observations"])
    assert np.isclose(a["obs_checksum"], b["obs_checksum"], rtol=1e-12) and a["rcs_sparsity"] == b["rcs_sparsity"]
    assert np.allclose(a["landmark_sum"], b["landmark_sum"], rtol=1e-11) and np.allclose(a["cam0"], b["cam0"], rtol=1e-11)


def test_cereal_cache_round_trip_and_byte_layout(app, bal_file, tmp_path):
    """BAL text -> preprocess -> save as .cereal -> load: same problem; the file is byte for byte the documented layout
    (file_info strings, uint64 counts, SE3 as px py pz qx qy qz qw, raw fixed-size matrices, int32 map keys)."""
    path, raw = bal_file
    cache = str(tmp_path / "pre.cereal")
    first = subprocess.run([app, "--input", path, "--dry-run", "--init-depth-threshold", "60", "--save-output",
                            "--output-optimized-path", cache], capture_output=True, text=True)
    assert first.returncode == 0, first.stderr
    want = json.loads(first.stdout.strip().splitlines()[-1])
    _same(_stats(app, cache), want)                       # autodetected by the extension
    other = str(tmp_path / "pre.bin")
    os.replace(cache, other)
    _same(_stats(app, other, "--input-type", "ROOTBA"), want)
    mirror = P.filter_obs(P.normalize(P.read_bal(path), 100.0), 60.0)
    data = open(other, "rb").read()
    mine = _cereal_bytes(mirror)
    assert len(data) == len(mine)
    assert data[:45] == mine[:45]  # file_info + camera count
    a, b = np.frombuffer(data[45:45 + 80 * mirror.n_cams], "<f8"), np.frombuffer(mine[45:45 + 80 * mirror.n_cams], "<f8")
    assert np.allclose(a.reshape(-1, 10)[:, :3], b.reshape(-1, 10)[:, :3], rtol=1e-9, atol=1e-9)    # translations
    assert np.allclose(a.reshape(-1, 10)[:, 7:], b.reshape(-1, 10)[:, 7:], rtol=0, atol=0)          # intrinsics: untouched
    for qa, qb in zip(a.reshape(-1, 10)[:, 3:7], b.reshape(-1, 10)[:, 3:7]):
        assert min(np.linalg.norm(qa - qb), np.linalg.norm(qa + qb)) < 1e-9


@pytest.mark.parametrize("variant", [dict(), dict(translation_first=False), dict(matrix_dims=True),
                                     dict(matrix_dims=True, translation_first=False),
                                     dict(matrix_dims=True, intrinsics_dims=True),
                                     dict(matrix_dims=True, intrinsics_dims=True, translation_first=False)],
                         ids=["t-first", "q-first", "dims", "dims-q-first", "dims+intr", "dims+intr-q-first"])
def test_cereal_cache_reader_accepts_the_layout_variants(app, bal_file, tmp_path, variant):
    """The third-party part of the layout (basalt-headers' serialisers) is unpinned here, so the reader tries its
    plausible variants and takes the one that validates - each variant, written independently, loads to the same problem."""
    path, raw = bal_file
    prob = P.read_bal(path)
    want = _stats(app, path)
    f = str(tmp_path / "v.cereal")
    open(f, "wb").write(_cereal_bytes(prob, **variant))
    _same(_stats(app, f), want)


def test_cereal_cache_reader_rejects_what_it_cannot_validate(app, bal_file, tmp_path):
    path, raw = bal_file
    prob = P.read_bal(path)
    good = _cereal_bytes(prob)
    bad_q = P.BalProblem(prob.cams.copy(), prob.lms, prob.lm_obs_offsets, prob.obs_cam_idx, prob.obs_xy, "bad")
    bad_q.cams[3, :4] *= 1.5
    desc = P.BalProblem(prob.cams, prob.lms, prob.lm_obs_offsets, prob.obs_cam_idx.copy(), prob.obs_xy, "desc")
    o0 = prob.lm_obs_offsets[5]
    desc.obs_cam_idx[o0], desc.obs_cam_idx[o0 + 1] = desc.obs_cam_idx[o0 + 1], desc.obs_cam_idx[o0]
    cases = {"type": (_cereal_bytes(prob, ftype=b"rootba::Something"), "different type"),
             "version": (_cereal_bytes(prob, version=b"2.0"), "unknown version"),
             "truncated": (good[:-9], "accepted layout"), "trailing": (good + b"\0" * 8, "accepted layout"),
             "quaternion": (_cereal_bytes(bad_q), "accepted layout"), "order": (_cereal_bytes(desc), "accepted layout"),
             "empty": (b"", "Failed")}
    for name, (blob, msg) in cases.items():
        f = str(tmp_path / (name + ".cereal"))
        open(f, "wb").write(blob)
        out = subprocess.run([app, "--input", f, "--dry-run"], capture_output=True, text=True)
        assert out.returncode != 0 and msg in out.stderr, (name, out.stderr)


def _cereal_parse(data):
    """Inverse of _cereal_bytes for the layout the writer emits."""
    import struct
    at = 0

    def take(fmt):
        nonlocal at
        v = struct.unpack_from(fmt, data, at)
        at += struct.calcsize(fmt)
        return v
    for want in (b"rootba::BalProblem", b"1.0"):
        (n,) = take("<Q")
        assert data[at:at + n] == want
        at += n
    (nc,) = take("<Q")
    cams = np.zeros((nc, 10))
    for i in range(nc):
        v = take("<10d")
        cams[i] = list(v[3:7]) + list(v[0:3]) + list(v[7:10])
    (nl,) = take("<Q")
    lms, off, cam, xy = np.zeros((nl, 3)), [0], [], []
    for l in range(nl):
        lms[l] = take("<3d")
        (k,) = take("<Q")
        for _ in range(k):
            c, x, y = take("<idd")
            cam.append(c)
            xy.append((x, y))
        off.append(len(cam))
    assert at == len(data)
    return P.BalProblem(cams, lms, np.array(off, dtype=np.int64), np.array(cam, dtype=np.int32), np.array(xy), "cereal")


def test_loaders_end_cleanly_on_mutated_files(app, tmp_path):
    """Truncations, byte flips, insertions and blown-up counts of a small valid BAL text / .cereal cache: every run ends
    with exit code 0 (still a valid file) or 2 (rejected with a message) - never a signal, never an unbounded allocation.
    (scripts/fuzz_loaders.py is the long form with an AddressSanitizer + UBSan build, incl. the Bundler loader.)"""
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from fuzz_loaders import mutate
    raw = P.synthetic_problem(12, 80, 300, seed=4)
    bal = str(tmp_path / "problem-12-80-pre.txt")
    P.write_bal(raw, bal)
    cer = str(tmp_path / "problem-12-80-pre.cereal")
    assert subprocess.run([app, "--input", bal, "--dry-run", "--no-normalize", "--save-output", "--output-optimized-path", cer],
                          capture_output=True).returncode == 0
    rng = random.Random(7)
    codes = set()
    for src, name in ((bal, "problem-m%d-pre.txt"), (cer, "problem-m%d-pre.cereal")):
        data = open(src, "rb").read()
        for i in range(40):
            path = str(tmp_path / (name % i))
            open(path, "wb").write(mutate(data, rng, src.endswith(".txt")))
            out = subprocess.run([app, "--input", path, "--dry-run"], capture_output=True, text=True, timeout=60)
            assert out.returncode in (0, 2), (path, out.returncode, out.stderr[-300:])
            codes.add(out.returncode)
            os.remove(path)
    assert 2 in codes
