"""Golden fixtures produced by the reference's own shaders (tests/golden/generate.py, via oracle/_ref): the CPU oracle
must reproduce them bit for bit everywhere (CPU suite), and so must the HIP passes through the C ABI (GPU suite)."""
import glob
import os

import numpy as np
import pytest

import oracle

import passes

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def load(path):
    z = np.load(path)
    scene = dict(meshes=z["meshes"], meshlets=z["meshlets"], draws=z["draws"], cull=z["cull"], depth=z["depth"], slots=int(z["slots"][0]),
                 viewport=tuple(int(x) for x in z["viewport"]))
    return z, scene, tuple(int(x) for x in z["flags"])


def compare(z, frames):
    for f, rec in enumerate(frames):
        assert rec["pyramid"].tobytes() == z["f%d_pyramid" % f].tobytes(), (f, "pyramid")
        for phase in ("early", "late"):
            for key in ("count4", "commands", "cc4", "cib", "dvb", "mvb"):
                assert rec[phase][key].tobytes() == z["f%d_%s_%s" % (f, phase, key)].tobytes(), (f, phase, key)


def test_fixtures_exist():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_reference_fixtures(path):
    z, scene, flags = load(path)
    compare(z, passes.run_frames(oracle, scene, flags, frames=2))


@pytest.mark.gpu
@pytest.mark.parametrize("use_soa", [True, False])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_reproduces_reference_fixtures(path, use_soa):
    from niagara_amd import pipeline as P

    import gpu_passes as G
    z, scene, flags = load(path)
    ctx = P.Context()
    try:
        compare(z, G.run_frames(ctx, scene, flags, frames=2, use_soa=use_soa))
    finally:
        ctx.close()
