"""The library's byte model (rba_get_byte_model: compulsory HBM bytes per launch group, the numerator of the per-stage
rooflines bench.py prints) against the traffic MEASURED with rocprofv3 PMC counters on an MI355X
(profiles/r6_pmc_stage_traffic.json, made by scripts/run_pmc_stage_traffic.sh from the same run that recorded the
model). A compulsory-bytes model can never exceed what the hardware moved; round 2's back-substitution model did
 and nothing checked it. The LIVE model of the library is held to the recorded one on the GPU
(the table alone cannot fail when rba_get_byte_model changes)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "profiles", "r6_pmc_stage_traffic.json")


@pytest.fixture(scope="module")
def table():
    with open(PATH) as f:
        return json.load(f)


def test_model_never_exceeds_measured_traffic(table):
    for name, g in table["groups"].items():
        if name == "product_assembled" and g["launches"] == 0:
            # (round 5: every solve of the recorded run that reached the assembled matrix ran in the persistent kernel -
            #  no product of the two-launch form was executed; its model row is checked on final-13682 size elsewhere)
            continue
        assert g["launches"] > 0, name
        # 3 %: FETCH_SIZE is calibrated on a streaming read, per-kernel access patterns deviate slightly
        assert g["model_bytes_per_launch"] <= 1.03 * g["measured_bytes_per_launch"], (name, g)


def test_measured_traffic_is_close_to_the_model_where_the_kernels_stream(table):
    """Streams (cost evaluation, stage 1, the products, the back-substitution) move within 15 % of the model; the
    camera-major gather of stage 2 fetches whole cache lines (round 6, split rows: one line of rows + one with the 32-byte
    stage-2 record per observation - 1.32 x by FETCH_SIZE, whose calibration is that of wide coalesced reads; rounds 3-5,
    72-byte rows across two lines: 1.38 x); the assembly of the double
    matrix gathers one-cache-line records, of which L2 serves most repeats (round 3's float assembly: 2.26 x; VERDICT
    round 3 asked for < 1.6); the vector kernels of a PCG iteration include the slots of half storage read back."""
    g = table["groups"]
    for name in ("compute_error", "stage1", "product_matrix_free", "back_substitution"):
        assert g[name]["measured_over_model"] < 1.15, (name, g[name])
    assert g["stage2"]["measured_over_model"] < 1.5, g["stage2"]
    assert g["assembly"]["measured_over_model"] < 1.5, g["assembly"]
    # the vector kernels of a MATRIX-FREE iteration (what is left outside the persistent kernel since round 5): three
    # launches of a few hundred KB each + the start / closing kernels of short solves - 2.8 MB measured for 1.2 MB of
    # vectors and M^-1 per iteration, of a 4 GB LM iteration
    assert g["pcg_vectors"]["measured_over_model"] < 2.6, g["pcg_vectors"]
    # the persistent PCG kernel: the model is the matrix once per solve + the exchanged records; MEASURED traffic includes
    # every polling sweep of every workgroup (L1-bypassing loads of records that are not there yet): 3.4 x - the kernel is
    # a latency chain, the price of no grid barrier is read traffic nobody waits for
    assert 1.0 <= g["persistent_solve"]["measured_over_model"] < 4.5, g["persistent_solve"]


def test_fetch_size_calibration_matches_the_guide(table):
    """gfx950: FETCH_SIZE reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section)."""
    c = table["fetch_correction"]
    assert 1.9 < c["measured_16B_loads"] < 2.1 and 1.9 < c["measured_4B_loads"] < 2.1
    assert c["used"] == c["measured_16B_loads"]


def test_bench_traffic_file_comes_from_the_same_pass(table):
    with open(os.path.join(ROOT, "profiles", "hx_traffic.json")) as f:
        hx = json.load(f)["venice-1778/implicit_q"]
    assert hx["traffic_bytes_per_launch"] == table["groups"]["product_matrix_free"]["measured_bytes_per_launch"]
    assert hx["model_bytes_per_launch"] == table["groups"]["product_matrix_free"]["model_bytes_per_launch"]


@pytest.mark.gpu
def test_live_byte_model_is_the_recorded_one(table):
    """rba_get_byte_model of the library as built, for the workload of the committed table."""
    import numpy as np
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd import problem as P
    from rootba_amd.linearizor import LinearizorHIP
    prob = P.preprocess(P.named_synthetic("venice-1778"), translation_sigma=0.01, point_sigma=0.01)
    g = LinearizorHIP(prob, np.float32, L.default_options(robust_norm=1, huber_parameter=1.0))
    live = g.byte_model()
    g.close()
    for k, v in table["meta"]["byte_model"].items():
        assert live[k] == v, (k, live[k], v)
