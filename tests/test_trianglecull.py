"""SURVEY.md §8f N4 — the mesh stage's per-triangle cull (src/shaders/meshlet.mesh.glsl:91-198 with the reference's own
MESH_CULL switch, src/config.h:10-11, set to 1).

CPU suite: the oracle's restatement equals the reference's mesh shader executing on the CPU (oracle/_ref, two runs per
workgroup instead of the barrier, see oracle/ref_runner.cpp) bit for bit, and replays the committed fixture the
reference generated (tests/golden/mesh/, tests/golden/generate.py).  GPU suite: the HIP kernel equals the oracle and the
fixture bit for bit, through the C ABI (nv_trianglecull)."""
import os

import numpy as np
import pytest

import oracle
from oracle import ref as R
from niagara_amd import layouts as L
from scenes import make_triangle_scene

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mesh", "trianglecull.npz")


def cluster_list(impl, s, backface=0):
    """clustercull<LATE=0> -> clustersubmit with `impl` (oracle or oracle.ref): the consumer's grid and index list"""
    cd = s["cull"].copy()
    cd["clusterBackfaceEnabled"] = backface
    cd["cullingEnabled"] = 1
    cib = np.zeros(s["n"] * 64 + 256, np.uint32)
    cc4 = np.zeros(4, np.uint32)
    impl.clustercull(cd, 0, s["commands"], s["count4"], s["draws"], s["meshlets"], None, None, cib, cc4)
    impl.clustersubmit(cc4, cib)
    return cib, cc4


def run(fn, s, cib, cc4):
    slots = int(cc4[1]) * int(cc4[2]) * int(cc4[3])
    masks = np.zeros(slots, L.TRIMASK)
    totals = np.zeros(3, np.uint64)
    fn(s["globals"], s["commands"], s["draws"], s["meshlets"], s["data"], s["vertices"], cib, cc4, masks, totals)
    return masks, totals


CAMERAS = [dict(), dict(cam_pos=(3.0, -2.0, 5.0), cam_quat=(0.0, 0.3826834, 0.0, 0.9238795)), dict(cam_pos=(0, 0, -8.0), viewport=(1920, 1080)),
           dict(scene_radius=3.0),  # puts the camera inside the cloud: vertices behind the perspective plane
           dict(scene_radius=6.0, specials=True)]  # NaN / inf / denormal vertices, non-finite draw fields


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (no reference tree and no prebuilt .so)")
@pytest.mark.parametrize("case", range(len(CAMERAS)))
def test_oracle_equals_reference_mesh_shader(case):
    s = make_triangle_scene(seed=40 + case, **CAMERAS[case])
    cib, cc4 = cluster_list(R, s)
    mo, to = run(oracle.trianglecull, s, cib, cc4)
    mr, tr = run(R.meshlet_mesh, s, cib, cc4)
    assert to.tolist() == tr.tolist() and to[0] > 50
    assert mo.tobytes() == mr.tobytes()
    kept = (mo["counts"] >> 16).sum()
    assert 0 < kept < (mo["counts"] & 0xff).sum()  # both decisions occur
    # padding slots (~0) of the 256-aligned list produce no outputs
    pad = cib[:len(mo)] == 0xffffffff
    assert pad.any() and not mo["counts"][pad].any()


def test_oracle_replays_reference_fixture():
    z = np.load(FIXTURE)
    s = {k: z[k] for k in ("globals", "commands", "draws", "meshlets", "data", "vertices")}
    mo, to = run(oracle.trianglecull, s, z["cib"], z["cc4"])
    assert to.tolist() == z["totals"].tolist()
    assert mo.tobytes() == z["masks"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(CAMERAS)))
def test_hip_equals_oracle(case):
    import torch
    from niagara_amd import pipeline as P
    s = make_triangle_scene(seed=40 + case, n_draws=400, commands_per_draw=5, **CAMERAS[case])
    cib, cc4 = cluster_list(oracle, s)
    mo, to = run(oracle.trianglecull, s, cib, cc4)
    ctx = P.Context()
    try:
        dev = ctx.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
        masks = torch.zeros(len(mo) * 16, dtype=torch.uint8, device=dev)
        totals = torch.zeros(3, dtype=torch.int64, device=dev)
        for _ in range(2):  # totals accumulate: zeroed by the caller
            totals.zero_()
            ctx.trianglecull(s["globals"], t(s["commands"]), t(s["draws"]), t(s["meshlets"]), t(s["data"]), t(s["vertices"]), t(cib), t(cc4), masks, len(mo), totals)
        assert totals.cpu().numpy().astype(np.uint64).tolist() == to.tolist()
        assert masks.cpu().numpy().tobytes() == mo.tobytes()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_hip_replays_reference_fixture():
    import torch
    from niagara_amd import pipeline as P
    z = np.load(FIXTURE)
    ctx = P.Context()
    try:
        dev = ctx.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
        masks = torch.zeros(len(z["masks"]) * 16, dtype=torch.uint8, device=dev)
        totals = torch.zeros(3, dtype=torch.int64, device=dev)
        ctx.trianglecull(z["globals"], t(z["commands"]), t(z["draws"]), t(z["meshlets"]), t(z["data"]), t(z["vertices"]), t(z["cib"]), t(z["cc4"]), masks,
                         len(z["masks"]), totals)
        assert totals.cpu().numpy().astype(np.uint64).tolist() == z["totals"].tolist()
        assert masks.cpu().numpy().tobytes() == z["masks"].tobytes()
    finally:
        ctx.close()
