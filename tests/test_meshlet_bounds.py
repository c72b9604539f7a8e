"""meshlet bounds (src/scene.cpp:69-85).  PARITY UNPINNED — the arithmetic is meshoptimizer's, which
the reference does not vendor; oracle.c restates the library's published algorithm.  What CAN be pinned without the
library is pinned here:

  * the properties niagara's cull relies on: every vertex inside the sphere, every triangle normal inside the cone, the
    s8 cone conservative; for random cameras the cone test with the generated bounds never culls a meshlet that has a
    front-facing triangle, and the frustum test never culls a meshlet with a vertex inside the frustum;
  * the quantisers against IEEE fp16 / the documented rounding;
  * GPU suite: the HIP kernel equals the oracle bit for bit (random payloads, coherent geometry, degenerate triangles,
    empty meshlets, both reference widths), and a scene whose bounds were generated on the GPU goes through
    drawcull -> clustercull identically to the oracle."""
import numpy as np
import pytest

import oracle
from niagara_amd import host, synth
from niagara_amd import layouts as L

from meshlet_builder import build_meshlets, meshlet_triangles, torus


@pytest.fixture(scope="module")
def mesh():
    pos, tris = torus()
    meshlets, data, vertices = build_meshlets(pos * 0.9, tris)
    return meshlets, data, vertices


def _half(u16):
    return np.asarray(u16, np.uint16).view(np.float16).astype(np.float64)


def test_sphere_and_cone_contain_the_geometry(mesh):
    meshlets, data, vertices = (x.copy() for x in mesh)
    f = oracle.meshlet_bounds(vertices, data, meshlets, want_float=True)
    assert len(meshlets) > 60
    cones = 0
    for k in range(len(meshlets)):
        pos, tri = meshlet_triangles(meshlets, data, vertices, k)
        c, r, axis, cutoff = f[k, :3].astype(np.float64), float(f[k, 3]), f[k, 4:7].astype(np.float64), float(f[k, 7])
        used = tri.reshape(-1, 3)
        assert (np.linalg.norm(used - c, axis=1) <= r * (1 + 1e-5) + 1e-7).all()           # every corner inside the sphere
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        ln = np.linalg.norm(n, axis=1)
        n = n[ln > 0] / ln[ln > 0, None]
        if cutoff < 1.0:
            cones += 1
            mindp = np.sqrt(max(0.0, 1 - cutoff * cutoff))
            assert (n @ axis >= mindp - 1e-5).all()                                         # every normal inside the cone
            assert abs(np.linalg.norm(axis) - 1) < 1e-5
        # quantised record: fp16 sphere (rounded to nearest, like niagara stores it) and a conservative s8 cone
        cq = _half(meshlets[k]["center"])
        rq = float(_half(meshlets[k]["radius"]))
        slack = 3 * 2.0 ** -11 * (np.abs(c).max() + r) + 1e-6
        assert (np.linalg.norm(used - cq, axis=1) <= rq + slack).all()
        aq, cutq = meshlets[k]["cone_axis"].astype(np.float64) / 127, float(meshlets[k]["cone_cutoff"]) / 127
        if cutoff < 1.0:
            assert cutq >= cutoff and cutq <= 1.0
            # the 8-bit test culls only what the float cone culls: for unit view directions d, dot(d, aq) >= cutq  =>  dot(d, axis) >= cutoff
            d = np.random.default_rng(k).normal(size=(2000, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            assert ((d @ aq < cutq) | (d @ axis >= cutoff - 1e-6)).all()
        else:
            assert meshlets[k]["cone_cutoff"] == 127 and not meshlets[k]["cone_axis"].any()
    assert cones > len(meshlets) // 2  # coherent geometry: most meshlets have a real cone


def test_cull_with_generated_bounds_is_a_superset_of_per_triangle_truth(mesh):
    """random cameras: coneCull (math.h:41-44) on the quantised record may only reject meshlets whose triangles all face
    away; the frustum test (clustercull.comp.glsl:103-108) may only reject meshlets with no vertex inside the frustum"""
    meshlets, data, vertices = (x.copy() for x in mesh)
    oracle.meshlet_bounds(vertices, data, meshlets)
    rng = np.random.default_rng(3)
    culled_cone = culled_frustum = 0
    for trial in range(30):
        cam = rng.normal(size=3)
        cam *= rng.uniform(1.6, 4.0) / np.linalg.norm(cam)
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        cd = host.build_cull_data(cam_pos=tuple(cam), cam_quat=tuple(q), draw_distance=50.0, draw_count=1, cullingEnabled=1, clusterBackfaceEnabled=1)
        V = cd["view"][0].astype(np.float64).reshape(4, 4).T
        fr, zn, zf = cd["frustum"][0].astype(np.float64), float(cd["znear"][0]), float(cd["zfar"][0])
        for k in range(len(meshlets)):
            pos, tri = meshlet_triangles(meshlets, data, vertices, k)
            c = V[:3, :3] @ _half(meshlets[k]["center"]) + V[:3, 3]
            r = float(_half(meshlets[k]["radius"]))
            axis = V[:3, :3] @ (meshlets[k]["cone_axis"].astype(np.float64) / 127)
            cutoff = float(meshlets[k]["cone_cutoff"]) / 127
            if c @ axis >= cutoff * np.linalg.norm(c) + r:
                culled_cone += 1
                n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
                facing = np.einsum("ij,ij->i", tri[:, 0] - cam, n)  # >= 0: the triangle faces away from the camera
                assert (facing >= -1e-9).all(), (trial, k)
            vis = c[2] * fr[1] - abs(c[0]) * fr[0] > -r and c[2] * fr[3] - abs(c[1]) * fr[2] > -r and c[2] + r > zn and c[2] - r < zf
            if not vis:
                culled_frustum += 1
                pv = (V[:3, :3] @ pos.T).T + V[:3, 3]
                inside = (pv[:, 2] * fr[1] - np.abs(pv[:, 0]) * fr[0] > 1e-6) & (pv[:, 2] * fr[3] - np.abs(pv[:, 1]) * fr[2] > 1e-6) & (pv[:, 2] > zn + 1e-6) & (pv[:, 2] < zf - 1e-6)
                assert not inside.any(), (trial, k)
    assert culled_cone > 50 and culled_frustum > 100


def test_quantisers():
    import ctypes as C
    lib = oracle.lib()
    rng = np.random.default_rng(1)
    meshlets = np.zeros(1, L.MESHLET)
    # through the public function: a one-triangle meshlet whose sphere centre we control is awkward; check the rounding rule
    # on its own instead, restated from the header of the library: round to nearest on the 13 dropped bits (ties up in
    # magnitude), flush below 2^-14, saturate to infinity
    def qh(v):
        ui = np.float32(v).view(np.uint32).item()
        s, em = (ui >> 16) & 0x8000, ui & 0x7fffffff
        h = (em - (112 << 23) + (1 << 12)) >> 13
        h = 0 if em < (113 << 23) else h
        h = 0x7c00 if em >= (143 << 23) else h
        h = 0x7e00 if em > (255 << 23) else h
        return (s | h) & 0xffff
    vals = np.concatenate([rng.uniform(-70000, 70000, 2000), rng.uniform(-1, 1, 2000), 10.0 ** rng.uniform(-9, 5, 2000)]).astype(np.float32)
    mine = np.array([qh(v) for v in vals], np.uint16)
    ieee = vals.astype(np.float16).view(np.uint16)
    normal = (np.abs(vals) >= 2.0 ** -14) & (np.abs(vals) < 65504)
    # identical to IEEE round-to-nearest-even except on exact ties (none in random data) — and flushes where IEEE goes subnormal
    assert (mine[normal] == ieee[normal]).all()
    assert (mine[np.abs(vals) < 2.0 ** -14] & 0x7fff == 0).all()
    assert qh(1e6) == 0x7c00 and qh(-1e6) == 0xfc00 and qh(float("nan")) & 0x7fff == 0x7e00
    # the oracle's own copy of the rule, through a degenerate meshlet (no triangles -> all-zero bounds)
    oracle.meshlet_bounds(np.zeros(4, L.VERTEX), np.zeros(8, np.uint32), meshlets)
    assert meshlets["center"].tolist() == [[0, 0, 0]] and meshlets["radius"][0] == 0 and meshlets["cone_cutoff"][0] == 0


def _random_payload_scene(n=20000, seed=5):
    draws, meshlets, commands, ncmd = synth.cluster_scene(max(1, n // 64), 1, seed=seed, scene_radius=40.0)
    meshlets = meshlets[:n].copy()
    data, vertices = synth.make_geometry(meshlets, seed=seed + 1)
    # degenerate and tiny meshlets
    meshlets["triangleCount"][::97] = 0
    meshlets["vertexCount"][5::101] = 1
    return meshlets, data, vertices


def test_oracle_handles_degenerate_input():
    meshlets, data, vertices = _random_payload_scene(3000)
    f = oracle.meshlet_bounds(vertices, data, meshlets, want_float=True)
    assert np.isfinite(f).all()
    assert (meshlets["radius"][::97] == 0).all() and (meshlets["cone_cutoff"][::97] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["random payloads", "torus", "torus, 32-bit references"])
def test_hip_bounds_equal_the_oracle(which):
    import torch
    from niagara_amd import pipeline as P
    if which == "random payloads":
        meshlets, data, vertices = _random_payload_scene(131072)
    else:
        pos, tris = torus(160, 80)
        meshlets, data, vertices = build_meshlets(pos * 0.9, tris, force_long_refs="32-bit" in which)
    mo = meshlets.copy()
    fo = oracle.meshlet_bounds(vertices, data, mo, want_float=True)
    ctx = P.Context()
    try:
        dev = ctx.device
        mlb = P.to_device(meshlets, dev)
        vb, dd = P.to_device(vertices, dev), torch.from_numpy(data.view(np.int32)).to(dev)
        f = torch.zeros((len(meshlets), 8), dtype=torch.float32, device=dev)
        ctx.meshlet_bounds(vb, dd, mlb, len(meshlets), f)
        ctx.status()
        got = P.from_device(mlb, L.MESHLET)
        assert got.tobytes() == mo.tobytes()
        assert f.cpu().numpy().tobytes() == fo.tobytes()
        # without the float output, and idempotent
        mlb2 = P.to_device(meshlets, dev)
        ctx.meshlet_bounds(vb, dd, mlb2, len(meshlets))
        ctx.meshlet_bounds(vb, dd, mlb2, len(meshlets))
        ctx.status()
        assert P.from_device(mlb2, L.MESHLET).tobytes() == mo.tobytes()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_scene_with_gpu_generated_bounds_through_the_passes():
    """torus instances: bounds from nv_meshlet_bounds, then the two-frame protocol (drawcull -> tasksubmit -> clustercull, both
    phases) against the oracle fed with the oracle's bounds"""
    import torch
    import gpu_passes as G
    import passes
    from niagara_amd import pipeline as P
    from scenes import make_scene
    pos, tris = torus(128, 64)
    meshlets, data, vertices = build_meshlets(pos * 0.9, tris)
    scene = make_scene(seed=77, n_draws=600, n_meshes=1, lods=1, meshlets_lod0=len(meshlets), scene_radius=12.0)
    scene["meshes"]["center"], scene["meshes"]["radius"] = (0.0, 0.0, 0.0), 1.3
    mo = meshlets.copy()
    oracle.meshlet_bounds(vertices, data, mo)
    ctx = P.Context()
    try:
        dev = ctx.device
        mlb = P.to_device(meshlets, dev)
        ctx.meshlet_bounds(P.to_device(vertices, dev), torch.from_numpy(data.view(np.int32)).to(dev), mlb, len(meshlets))
        ctx.status()
        generated = P.from_device(mlb, L.MESHLET).copy()
        assert generated.tobytes() == mo.tobytes()
        want = passes.run_frames(oracle, dict(scene, meshlets=mo), (1, 1, 1, 1, 1), frames=2)
        got = G.run_frames(ctx, dict(scene, meshlets=generated), (1, 1, 1, 1, 1), frames=2)
        seen = 0
        for g, w in zip(got, want):
            for phase in ("early", "late"):
                for key in ("count4", "cc4", "cib", "dvb", "mvb"):
                    assert (g[phase][key] == w[phase][key]).all(), (phase, key)
                seen += int(w[phase]["cc4"][0])
        assert seen > 1000
    finally:
        ctx.close()


KITTEN = "/root/reference/data/kitten.obj"


@pytest.mark.skipif(not __import__("os").path.exists(KITTEN), reason="the reference tree (data/kitten.obj) is not mounted here")
def test_kitten_meshlets_through_the_cluster_cull():
    """niagara's own asset.  kitten.obj (read where the reference tree is mounted, never copied) is cut into
    meshlets in face order, the bounds come from orc_meshlet_bounds, and 600 instances go through the oracle's cluster cull:
    every meshlet the cull drops has no front-facing triangle inside the frustum (per-triangle ground truth in fp64), and
    sphere / cone contain the geometry.  The GPU suite generates the same bounds with nv_meshlet_bounds on the torus; the
    asset itself does not travel to the GPU box."""
    pos, faces = [], []
    for line in open(KITTEN):
        if line.startswith("v "):
            pos.append([float(x) for x in line.split()[1:4]])
        elif line.startswith("f "):
            faces.append([int(t.split("/")[0]) - 1 for t in line.split()[1:4]])
    pos, faces = np.array(pos, np.float32), np.array(faces, np.int64)
    assert len(pos) == 14472 and len(faces) == 28944
    # spatially coherent order (the clusterizer's job): sort faces by the Morton code of their centroid
    cen = pos[faces].mean(axis=1)
    g = np.clip(((cen - cen.min(0)) / (np.ptp(cen, axis=0) + 1e-9) * 1023).astype(np.int64), 0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; return (v | (v << 2)) & 0x09249249
    order = np.argsort(spread(g[:, 0]) | (spread(g[:, 1]) << 1) | (spread(g[:, 2]) << 2), kind="stable")
    meshlets, data, vertices = build_meshlets(pos, faces[order])
    f = oracle.meshlet_bounds(vertices, data, meshlets, want_float=True)
    assert 300 < len(meshlets) < 2000 and (f[:, 7] < 1).mean() > 0.5
    for k in range(0, len(meshlets), 7):
        p, tri = meshlet_triangles(meshlets, data, vertices, k)
        c, r = f[k, :3].astype(np.float64), float(f[k, 3])
        assert (np.linalg.norm(tri.reshape(-1, 3) - c, axis=1) <= r * (1 + 1e-5) + 1e-7).all()
    # 600 kitten instances around the camera through clustercull (commands: one per 64 meshlets of an instance)
    n_draws = 600
    draws = host.synth_draws(n_draws, 1, 12.0)
    cpd = (len(meshlets) + 63) // 64
    commands = np.zeros((n_draws * cpd + 63) // 64 * 64, dtype=L.TASKCMD)
    k = np.arange(n_draws * cpd)
    commands["drawId"][:len(k)] = k // cpd
    commands["taskOffset"][:len(k)] = (k % cpd) * 64
    commands["taskCount"][:len(k)] = np.minimum(64, len(meshlets) - (k % cpd) * 64)
    cd = host.build_cull_data(draw_count=n_draws, cullingEnabled=1, clusterBackfaceEnabled=1, draw_distance=40.0)
    cib, cc4 = np.zeros(len(commands) * 64, np.uint32), np.zeros(4, np.uint32)
    oracle.clustercull(cd, 0, commands, synth.count4_for(len(k)), draws, meshlets, None, None, cib, cc4)
    visible = set(int(x) for x in cib[:cc4[0]])
    assert 0.02 * len(k) * 64 < len(visible) < 0.9 * n_draws * len(meshlets)
    V = cd["view"][0].astype(np.float64).reshape(4, 4).T
    fr, zn, zf = cd["frustum"][0].astype(np.float64), float(cd["znear"][0]), float(cd["zfar"][0])
    rng = np.random.default_rng(2)
    checked = 0
    for ci in rng.choice(len(k), 400, replace=False):
        d = draws[commands["drawId"][ci]]
        q, s, t = d["orientation"].astype(np.float64), float(d["scale"]), d["position"].astype(np.float64)
        for lane in range(int(commands["taskCount"][ci])):
            if (int(ci) | (lane << 24)) in visible:
                continue
            mi = int(commands["taskOffset"][ci]) + lane
            _, tri = meshlet_triangles(meshlets, data, vertices, mi)
            qv, w = q[:3], q[3]
            pts = tri.reshape(-1, 3)
            rot = pts + 2.0 * np.cross(qv, np.cross(qv, pts) + w * pts)
            world = rot * s + t
            view = (V[:3, :3] @ world.T).T + V[:3, 3]
            tv, tw = view.reshape(-1, 3, 3), world.reshape(-1, 3, 3)
            # facing in world space (the view matrix mirrors z, which would flip a normal recomputed from view-space corners;
            # the shader transforms the object-space axis as a vector): camera at the origin
            n = np.cross(tw[:, 1] - tw[:, 0], tw[:, 2] - tw[:, 0])
            front = np.einsum("ij,ij->i", tw[:, 0], n) < -1e-7
            inside = ((tv[..., 2] * fr[1] - np.abs(tv[..., 0]) * fr[0] > 1e-5) & (tv[..., 2] * fr[3] - np.abs(tv[..., 1]) * fr[2] > 1e-5)
                      & (tv[..., 2] > zn + 1e-5) & (tv[..., 2] < zf - 1e-5)).any(axis=1)
            assert not (front & inside).any(), (int(ci), lane)          # a culled meshlet hides nothing that could be seen
            checked += 1
    assert checked > 3000
