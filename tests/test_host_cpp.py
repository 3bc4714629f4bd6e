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
    assert log["_type"] == "rootba" and log["iteration"][0] == 0
    prob = P.normalize(P.read_bal(path), 100.0)
    g = LinearizorHIP(prob, np.float64, L.default_options(robust_norm=1, max_num_iterations=6))
    rows, _ = g.optimize_lm()
    assert len(rows) == len(log["iteration"])
    assert np.allclose([r.cost for r in rows], log["cost_all_error"], rtol=1e-7)
    assert [r.cg_iterations for r in rows] == log["linear_solver_iterations"]
