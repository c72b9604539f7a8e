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


def _run_hip(s, cib, cc4, slots):
    import torch
    from niagara_amd import pipeline as P
    ctx = P.Context()
    try:
        dev = ctx.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
        masks = torch.full((slots * 16,), 0x5a, dtype=torch.uint8, device=dev)
        totals = torch.zeros(3, dtype=torch.int64, device=dev)
        ctx.trianglecull(s["globals"], t(s["commands"]), t(s["draws"]), t(s["meshlets"]), t(s["data"]), t(s["vertices"]), t(cib), t(cc4), masks, slots, totals)
        return masks.cpu().numpy(), totals.cpu().numpy().astype(np.uint64)
    finally:
        ctx.close()


def _every_meshlet_list(s, repeat=1, holes=0.0, seed=0):
    """A cluster list naming every meshlet of the scene `repeat` times (the list a fully visible scene produces), optionally with
    ~0 holes (clustersubmit's padding value in the middle of the list: slots that produce nothing), plus the consumer's grid."""
    m = s["n"] * 64
    ids = (np.arange(m, dtype=np.uint32) // 64) | ((np.arange(m, dtype=np.uint32) % 64) << 24)
    ids = np.tile(ids, repeat)
    if holes:
        rng = np.random.default_rng(seed)
        ids[rng.random(len(ids)) < holes] = 0xffffffff
    cc4 = np.array([len(ids), 0, 0, 0], np.uint32)
    cib = np.concatenate([ids, np.zeros(512, np.uint32)])
    oracle.clustersubmit(cc4, cib)
    return cib, cc4


@pytest.mark.gpu
def test_hip_malformed_meshlets_read_as_the_oracle_defines():
    """Counts the packed streams must not trip over: no vertices, no triangles, more triangles than MESH_MAXTRI (the loops stop at
    96, the totals count the raw byte), index bytes that name a vertex the meshlet does not have (the shader's zero-initialised
    slot: oracle memset), ~0 slots between live ones."""
    s = make_triangle_scene(seed=77, n_draws=300, commands_per_draw=4, scene_radius=6.0)
    ml = s["meshlets"]
    rng = np.random.default_rng(78)
    n = len(ml)
    pick = rng.random(n)
    safe = np.arange(n) < n - 64  # raised triangle counts read index words past the meshlet's own: keep those inside the buffer
    ml["triangleCount"][(pick < 0.05)] = 0
    # fewer vertices than the index bytes name.  (indexOffset follows the mutated count on both sides; whatever bytes lie there are
    # masked to 6 bits, and the ones at or above the count read the zero slot)
    ml["vertexCount"][(pick >= 0.05) & (pick < 0.10)] //= 3
    ml["triangleCount"][(pick >= 0.10) & (pick < 0.15) & safe] = 200
    ml["triangleCount"][(pick >= 0.15) & (pick < 0.17) & safe] = 255
    zero_v = (pick >= 0.17) & (pick < 0.20)
    ml["vertexCount"][zero_v] = 0
    cib, cc4 = _every_meshlet_list(s, holes=0.03, seed=79)
    mo, to = run(oracle.trianglecull, s, cib, cc4)
    mg, tg = _run_hip(s, cib, cc4, len(mo))
    assert tg.tolist() == to.tolist()
    assert mg.tobytes() == mo.tobytes()
    assert (mo["counts"] >> 16).sum() > 0 and (mo["counts"][cib[:len(mo)] == 0xffffffff] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("specials", [False, True])
def test_hip_long_list_several_chunks_per_wave(specials):
    """More slots than 64 per wave of the launch (8 workgroups x 4 waves per CU): every wave walks several 64-slot chunks, batches
    end on chunk boundaries, the last wave's run is ragged."""
    s = make_triangle_scene(seed=91, n_draws=120, commands_per_draw=3, scene_radius=5.0, specials=specials)
    m = s["n"] * 64
    repeat = (8 * 4 * 256 * 64 * 2) // m + 1  # > 128 slots per wave on a 256-CU part
    cib, cc4 = _every_meshlet_list(s, repeat=repeat, holes=0.01, seed=92)
    mo, to = run(oracle.trianglecull, s, cib, cc4)
    mg, tg = _run_hip(s, cib, cc4, len(mo))
    assert tg.tolist() == to.tolist()
    assert mg.tobytes() == mo.tobytes()


@pytest.mark.gpu
def test_hip_mask_capacity_is_respected():
    """slots past `maskCapacity` are counted in the totals and not written (meshlet.mesh.glsl has no such bound: the ABI's)"""
    s = make_triangle_scene(seed=93, n_draws=50, commands_per_draw=2)
    cib, cc4 = _every_meshlet_list(s)
    mo, to = run(oracle.trianglecull, s, cib, cc4)
    cap = len(mo) // 2 + 7
    mg, tg = _run_hip(s, cib, cc4, len(mo))
    import torch
    from niagara_amd import pipeline as P
    ctx = P.Context()
    try:
        dev = ctx.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
        masks = torch.full((len(mo) * 16,), 0x5a, dtype=torch.uint8, device=dev)
        totals = torch.zeros(3, dtype=torch.int64, device=dev)
        ctx.trianglecull(s["globals"], t(s["commands"]), t(s["draws"]), t(s["meshlets"]), t(s["data"]), t(s["vertices"]), t(cib), t(cc4), masks, cap, totals)
        got = masks.cpu().numpy()
        assert got[:cap * 16].tobytes() == mo[:cap].tobytes() and (got[cap * 16:] == 0x5a).all()
        assert totals.cpu().numpy().astype(np.uint64).tolist() == to.tolist()
    finally:
        ctx.close()
