"""Host-side data model: synthetic generator, reference preprocessing, BAL text
round trip (reference src/rootba/bal/bal_problem.cpp:190-282, 428-554)."""
import numpy as np

from rootba_amd import problem as P


def test_synthetic_matches_survey_statistics():
    prob = P.named_synthetic("ladybug-49")
    k = prob.obs_per_lm()
    assert prob.n_cams == 49 and prob.n_lms == 7776
    assert k.min() >= 2 and k.max() <= 49
    assert abs(k.mean() - 4.07) < 0.05 and abs((k * k).mean() - 22.9) < 0.5
    # ascending, unique cameras inside every landmark
    lm = np.repeat(np.arange(prob.n_lms), k)
    same = lm[1:] == lm[:-1]
    assert np.all(np.diff(prob.obs_cam_idx.astype(np.int64))[same] > 0)
    # all points in front of their cameras
    _, z = P.project(prob.cams[prob.obs_cam_idx], prob.lms[lm])
    assert z.min() >= 0.1


def test_normalize_scale_and_center():
    prob = P.named_synthetic("ladybug-49")
    out = P.normalize(prob, 100.0)
    med = np.array([np.partition(out.lms[:, j], out.n_lms // 2)[out.n_lms // 2] for j in range(3)])
    assert np.abs(med).max() < 1e-9
    mad = np.partition(np.abs(out.lms).sum(1), out.n_lms // 2)[out.n_lms // 2]
    assert abs(mad - 100.0) < 1e-9
    # projections are invariant under the similarity transform
    lm = np.repeat(np.arange(prob.n_lms), prob.obs_per_lm())
    a, _ = P.project(prob.cams[prob.obs_cam_idx], prob.lms[lm])
    b, _ = P.project(out.cams[out.obs_cam_idx], out.lms[lm])
    assert np.allclose(a, b, rtol=1e-9, atol=1e-7)


def test_filter_obs_drops_close_points_and_weak_landmarks():
    prob = P.normalize(P.synthetic_problem(20, 200, 800, seed=3))
    lm = np.repeat(np.arange(prob.n_lms), prob.obs_per_lm())
    _, z = P.project(prob.cams[prob.obs_cam_idx], prob.lms[lm])
    thr = float(np.quantile(z, 0.3))
    out = P.filter_obs(prob, thr)
    lm2 = np.repeat(np.arange(out.n_lms), out.obs_per_lm())
    _, z2 = P.project(out.cams[out.obs_cam_idx], out.lms[lm2])
    assert z2.min() >= thr and out.obs_per_lm().min() >= 2 and out.n_obs < prob.n_obs


def test_bal_text_round_trip(tmp_path):
    prob = P.synthetic_problem(6, 30, 100, seed=5)
    path = str(tmp_path / "problem-6-30-pre.txt")
    P.write_bal(prob, path)
    back = P.read_bal(path)
    assert back.n_cams == prob.n_cams and back.n_lms == prob.n_lms and back.n_obs == prob.n_obs
    assert np.array_equal(back.lm_obs_offsets, prob.lm_obs_offsets)
    assert np.array_equal(back.obs_cam_idx, prob.obs_cam_idx)
    assert np.allclose(back.obs_xy, prob.obs_xy, atol=1e-12)
    assert np.allclose(back.lms, prob.lms, atol=1e-12)
    assert np.allclose(P.quat_to_rot(back.cams[:, :4]), P.quat_to_rot(prob.cams[:, :4]), atol=1e-10)
    assert np.allclose(back.cams[:, 4:], prob.cams[:, 4:], atol=1e-10)


def test_duplicate_observation_is_rejected(tmp_path):
    path = str(tmp_path / "dup.txt")
    with open(path, "w") as f:
        f.write("1 1 2\n0 0 1.0 2.0\n0 0 1.0 2.0\n" + "0\n" * 9 + "0\n0\n1\n")
    try:
        P.read_bal(path)
        raise AssertionError("expected ValueError")
    except ValueError as e:
        assert "Invalid file" in str(e)  # reference: CHECK(inserted) << "Invalid file" (bal_problem.cpp:229-230)


def test_heavy_tail_variant_has_long_tracks():
    from rootba_amd import problem as P
    raw = P.named_synthetic("trafalgar-257+tail")
    k = raw.obs_per_lm()
    assert (raw.n_cams, raw.n_lms) == (257, 65132)
    assert k.max() == 120 and (k > 60).sum() >= 5 and k.min() >= 2
    prob = P.preprocess(raw)  # every observation stays in front of its camera
    assert prob.obs_per_lm().max() >= 115
