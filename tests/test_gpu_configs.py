"""BASELINE.json's configs at FULL size on the GPU, every output buffer against the CPU oracle.

tools/bench_configs.py builds each config's synthetic inputs, runs the HIP passes through the C ABI, and compares with
the multithreaded oracle before it reports a time; it raises SystemExit on any difference.  These tests run the same
functions with a handful of iterations, so the numbers in profiles/ and the parity bar come from one piece of code:

    config 2   1 000 000 draws, drawcull<0,0> (commands, count, drawVisibility) and drawcull<1,0> + HiZ (2048^2 pyramid)
    config 3B  62 500 draws -> drawcull<0,1> -> tasksubmit -> clustercull<0> -> clustersubmit, every buffer, both launch forms
    config 4   4096^2 depth -> 2048^2 x 12 pyramid, then clustercull<1> over 10 M meshlets with random visibility bits and
               lateDrawVisibility: IDs and the rewritten meshletVisibility words
    dense      config 3A's 10 M meshlets as a cloud of radius 40 seen from outside (87 % of the commands have survivors), cone on / off
    frame      niagara's dependent frame (early cull -> pyramid -> late cull) at BASELINE scale: 1 M draws, ~10 M meshlets tested per
               cluster pass, 4096^2 depth, three rotated scene copies, both launch forms; every buffer of a frame's two phases
"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bc():
    spec = importlib.util.spec_from_file_location("bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture()
def ctx():
    from niagara_amd import pipeline as P
    c = P.Context()
    yield c
    c.close()


def test_config2_one_million_draws(bc, ctx):
    r = bc.config2(ctx, 3)
    assert r["parity"] == "bit-identical" and r["draws"] == 1_000_000 and 0.01 * r["draws"] < r["visible"] < 0.2 * r["draws"]


def test_config2_late_with_hiz(bc, ctx):
    r = bc.config2_late(ctx, 3)
    assert r["parity"] == "bit-identical" and r["visible"] > 0


@pytest.mark.parametrize("fused", [False, True])
def test_config3b_contract_chain(bc, ctx, fused):
    r = bc.config3b(ctx, 3, fused=fused)
    assert r["parity"] == "bit-identical" and r["draws"] == 62500 and r["meshlets_tested"] > 4_000_000


def test_config4_pyramid_and_late_pass(bc, ctx):
    r = bc.config4(ctx, 3)
    assert r["parity"] == "bit-identical" and r["late_visible"] > 0


@pytest.mark.parametrize("backface", [1, 0])
def test_dense_visibility(bc, ctx, backface):
    r = bc.cluster_config(ctx, 3, "dense", 15625, 10, scene_radius=40.0, backface=backface, cam_pos=(0, 0, 60))
    assert r["parity"] == "bit-identical" and r["commands_with_survivors"] > 0.3


@pytest.mark.parametrize("fused", [True, False])
def test_frame_at_baseline_scale(bc, fused):
    """the frame chain at 1 M draws / 4096^2 / ~10 M meshlets per cluster pass, 3 timed frames over 3 rotated scene
    copies, frame N's visibility feeding frame N + 1; the buffers of both phases of one more frame against the oracle"""
    from niagara_amd import pipeline as P
    c = P.Context()
    try:
        r = bc.config_frame(c, 3, fused=fused)
    finally:
        c.close()
    assert r["parity"] == "bit-identical" and r["draws"] == 1_000_000
    assert r["early"]["meshlets_tested"] > 8_000_000 and r["late"]["meshlets_tested"] > 8_000_000
    assert r["early"]["visible"] > 1_000_000 and r["late"]["draws_visible"] > 10_000
    assert r["frames_on_checked_copy"] >= 3
