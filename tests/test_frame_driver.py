"""examples/frame_driver.cpp — the reference's frame loop written in C++ on the C ABI (no Python or torch in that process) —
against the oracle driven over the same scene: every buffer of every phase, byte for byte."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle
import passes
from scenes import make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "examples", "frame_driver")

TAG_PYRAMID, TAG_COUNT4, TAG_COMMANDS, TAG_CC4, TAG_CIB, TAG_DVB, TAG_MVB = range(1, 8)


def write_scene(path, scene, cd):
    vw, vh = scene["viewport"]
    with open(path, "wb") as f:
        f.write(struct.pack("<6I", 0x4353564E, len(scene["meshes"]), len(scene["meshlets"]), len(scene["draws"]), vw, vh))
        f.write(cd.tobytes())
        for key in ("meshes", "meshlets", "draws"):
            f.write(np.ascontiguousarray(scene[key]).tobytes())
        depth = np.ascontiguousarray(scene["depth"], dtype=np.float32)
        assert depth.shape == (vh, vw)
        f.write(depth.tobytes())


def expected_stream(records):
    out = []

    def rec(tag, arr):
        b = np.ascontiguousarray(arr).tobytes()
        out.append(struct.pack("<2I", tag, len(b)) + b)

    for frame in records:
        for phase in [p for p in ("early", "late", "post") if p in frame]:
            if phase == "late":
                rec(TAG_PYRAMID, frame["pyramid"])
            r = frame[phase]
            rec(TAG_COUNT4, r["count4"])
            rec(TAG_COMMANDS, r["commands"])
            rec(TAG_CC4, r["cc4"])
            rec(TAG_CIB, r["cib"])
            rec(TAG_DVB, r["dvb"])
            rec(TAG_MVB, r["mvb"])
    return out


def test_driver_is_built_and_prints_usage():
    assert os.path.exists(DRIVER), "examples/frame_driver is missing: run __graft_entry__.build() (make -C examples)"
    p = subprocess.run([DRIVER], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "usage" in p.stderr


def compare_stream(got, want):
    pos = 0
    for i, w in enumerate(want):
        assert got[pos:pos + 8] == w[:8], ("record header", i, struct.unpack("<2I", got[pos:pos + 8]), struct.unpack("<2I", w[:8]))
        assert got[pos:pos + len(w)] == w, ("record payload", i, struct.unpack("<2I", w[:8]))
        pos += len(w)
    assert pos == len(got)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("flags, post", [((1, 1, 1, 1, 1), 0.0), ((1, 0, 1, 0, 1), 0.0), ((1, 1, 1, 1, 1), 0.2), ((1, 1, 1, 1, 0), 0.2)])
def test_cpp_frame_loop_matches_the_oracle(tmp_path, flags, post, fused):
    """post > 0: a fifth of the draws are postPass draws, so every frame runs the third phase cull(late, postPass=1) +
    render(late, postPass=1) (src/niagara.cpp:1781-1787) — driven from C++"""
    scene = make_scene(seed=31, n_draws=1200, meshlets_lod0=150, zero_radius_fraction=0.02, post_pass_fraction=post)
    cd = passes.set_flags(scene["cull"], flags)
    write_scene(tmp_path / "scene.bin", scene, cd)
    p = subprocess.run([DRIVER, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin"), "3"] + (["fused"] if fused else []),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    records = passes.run_frames(oracle, scene, flags, frames=3)
    assert all(("post" in r) == (post > 0) for r in records)
    compare_stream(open(tmp_path / "out.bin", "rb").read(), expected_stream(records))
    assert "visible clusters" in p.stdout and ("post" in p.stdout) == (post > 0)


@pytest.mark.gpu
def test_cpp_frame_loop_timed_mode(tmp_path):
    """`time N`: N frames back to back without read-backs, one JSON line with the wall time per frame, and the frame recorded
    after them (frame 2 + 3 warm-up + N) still equals the oracle's"""
    import json
    scene = make_scene(seed=33, n_draws=1500, meshlets_lod0=150, post_pass_fraction=0.1)
    flags = (1, 1, 1, 1, 1)
    cd = passes.set_flags(scene["cull"], flags)
    write_scene(tmp_path / "scene.bin", scene, cd)
    n = 7
    p = subprocess.run([DRIVER, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin"), "2", "fused", "time", str(n)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert line["frames"] == n and line["frame_us"] > 0 and line["phases_per_frame"] == 3
    records = passes.run_frames(oracle, scene, flags, frames=2 + 3 + n + 1)
    compare_stream(open(tmp_path / "out.bin", "rb").read(), expected_stream(records[:2] + records[-1:]))
