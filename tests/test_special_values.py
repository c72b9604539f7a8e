"""Non-finite and degenerate inputs.  The reference's shaders run on whatever is in the buffers; a drop-in has to take the
same decisions for NaN / inf / denormal bounds, zero or negative scales and un-normalised quaternions, because one
different comparison changes the visible list.  Random bit patterns go through the reference shaders on the CPU
(oracle/_ref), the oracle, and — GPU suite — the HIP kernels; all three lists must be identical."""
import numpy as np
import pytest

import oracle
from oracle import ref as R
from niagara_amd import host, synth
from niagara_amd import layouts as L

SPECIAL = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-42, -1e-42, 3.4e38, -3.4e38, 1.0, -1.0, 65504.0], np.float32)


def special_depth(w, h, seed):
    rng = np.random.default_rng(seed)
    d = rng.random((h, w)).astype(np.float32)
    k = rng.random((h, w))
    d[k < 0.02] = np.nan
    d[(k >= 0.02) & (k < 0.04)] = np.inf
    d[(k >= 0.04) & (k < 0.06)] = -np.inf
    d[(k >= 0.06) & (k < 0.08)] = -0.0
    d[(k >= 0.08) & (k < 0.10)] = 1e-42
    return d


def special_scene(seed, n_draws=48, cpd=2):
    rng = np.random.default_rng(seed)
    draws, meshlets, commands, n = synth.cluster_scene(n_draws, cpd, seed=seed, scene_radius=15.0)
    # meshlet bounds: a third of the pool gets raw 16-bit patterns (every fp16 class: NaN, inf, denormals, -0)
    raw = rng.random(len(meshlets)) < 0.33
    meshlets["center"][raw] = rng.integers(0, 1 << 16, (int(raw.sum()), 3)).astype(np.uint16)
    meshlets["radius"][raw] = rng.integers(0, 1 << 16, int(raw.sum())).astype(np.uint16)
    meshlets["cone_axis"][raw] = rng.integers(-128, 128, (int(raw.sum()), 3)).astype(np.int8)
    meshlets["cone_cutoff"][raw] = rng.integers(-128, 128, int(raw.sum())).astype(np.int8)
    # draws: a quarter gets special floats in position / scale / orientation
    for i in np.nonzero(rng.random(n_draws) < 0.25)[0]:
        field = rng.integers(0, 3)
        if field == 0:
            draws["position"][i][rng.integers(0, 3)] = rng.choice(SPECIAL)
        elif field == 1:
            draws["scale"][i] = rng.choice(SPECIAL)
        else:
            draws["orientation"][i][rng.integers(0, 4)] = rng.choice(SPECIAL)
    # ragged commands
    commands["taskCount"][:n] = rng.integers(0, 65, n)
    cd = host.build_cull_data(draw_count=n_draws, viewport=(256, 192), pyramid=(128, 128), cullingEnabled=1, clusterBackfaceEnabled=1,
                              clusterOcclusionEnabled=1, occlusionEnabled=1)
    slots = n * 64 + 64
    mvb = rng.integers(0, 2 ** 32, slots // 32 + 2, dtype=np.uint64).astype(np.uint32)
    commands["lateDrawVisibility"][:n] = rng.integers(0, 2, n)
    depth = synth.make_depth(256, 192, seed=seed)
    bad = special_depth(256, 192, seed)  # NaN / inf / -0 texels among the occluders: the HiZ comparison must agree on them too
    k = np.random.default_rng(seed + 1).random((192, 256)) < 0.15
    depth[k] = bad[k]
    return dict(draws=draws, meshlets=meshlets, commands=commands, n=n, cull=cd, mvb=mvb, depth=depth, count4=synth.count4_for(n))


def run_cpu(impl, s, late, post_pass=0):
    pyr = oracle.Pyramid(256, 192)
    impl.depthreduce(s["depth"], pyr)
    cd = s["cull"].copy()
    cd["postPass"] = post_pass
    cd["pyramidWidth"], cd["pyramidHeight"] = pyr.width, pyr.height
    cib = np.zeros(s["n"] * 64 + 256, np.uint32)
    cc4 = np.zeros(4, np.uint32)
    mvb = s["mvb"].copy()
    impl.clustercull(cd, late, s["commands"], s["count4"], s["draws"], s["meshlets"], mvb, pyr, cib, cc4)
    return cc4.copy(), cib[:int(cc4[0])].copy(), mvb, cd, pyr


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (no reference tree and no prebuilt .so)")
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("late", [0, 1])
@pytest.mark.parametrize("post_pass", [0, 1])
def test_oracle_equals_reference_on_special_values(seed, late, post_pass):
    s = special_scene(900 + seed)
    co, io, mo, _, _ = run_cpu(oracle, s, late, post_pass)
    cr, ir, mr, _, _ = run_cpu(R, s, late, post_pass)
    assert co.tolist() == cr.tolist()
    assert (io == ir).all()
    assert (mo == mr).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("late", [0, 1])
@pytest.mark.parametrize("soa", [True, False])
@pytest.mark.parametrize("post_pass", [0, 1])
def test_hip_equals_oracle_on_special_values(seed, late, soa, post_pass):
    import torch
    from niagara_amd import pipeline as P
    s = special_scene(900 + seed)
    co, io, mo, cd, pyr = run_cpu(oracle, s, late, post_pass)
    ctx = P.Context()
    try:
        dev = ctx.device
        gp = P.DepthPyramid(dev, 256, 192)
        ctx.depthreduce(torch.from_numpy(s["depth"]).to(dev), 256, 192, gp.desc)
        db, mlb, dcb = P.to_device(s["draws"], dev), P.to_device(s["meshlets"], dev), P.to_device(s["commands"], dev)
        if soa:
            ctx.upload_meshlets(mlb, len(s["meshlets"]))
        dccb = torch.from_numpy(s["count4"].view(np.int32).copy()).to(dev)
        mvb = torch.from_numpy(s["mvb"].view(np.int32).copy()).to(dev)
        cib = torch.zeros(s["n"] * 64 + 256, dtype=torch.int32, device=dev)
        ccb = torch.zeros(4, dtype=torch.int32, device=dev)
        ctx.clustercull(cd, late, dcb, dccb, db, mlb, mvb, gp.desc, cib, ccb)
        total = int(ccb[0].item())
        assert total == int(co[0])
        assert (cib.cpu().numpy().view(np.uint32)[:total] == io).all()
        assert (mvb.cpu().numpy().view(np.uint32) == mo).all()
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("soa", [True, False])
@pytest.mark.parametrize("bits", [0, 1])
def test_taskcull_hip_equals_oracle_on_special_values(seed, soa, bits):
    """nv_taskcull's early pass over the SoA mirror walks 64-command segments through the conservative frustum filter (round 3):
    NaN / inf / denormal bounds and draw fields, dummy commands and partial commands must leave the oracle's payloads"""
    import torch
    from niagara_amd import pipeline as P
    s = special_scene(950 + seed)
    pyr = oracle.Pyramid(256, 192)
    oracle.depthreduce(s["depth"], pyr)
    cd = s["cull"].copy()
    cd["pyramidWidth"], cd["pyramidHeight"] = pyr.width, pyr.height
    cd["clusterOcclusionEnabled"] = bits
    ncmd = len(s["commands"])
    pay_o, cnt_o = np.zeros((ncmd, 64), np.uint32), np.zeros(ncmd, np.uint32)
    mvb_o = s["mvb"].copy()
    oracle.taskcull(cd, 0, s["commands"], s["count4"], s["draws"], s["meshlets"], mvb_o, pyr, pay_o, cnt_o)
    ctx = P.Context()
    try:
        dev = ctx.device
        gp = P.DepthPyramid(dev, 256, 192)
        ctx.depthreduce(torch.from_numpy(s["depth"]).to(dev), 256, 192, gp.desc)
        db, mlb, dcb = P.to_device(s["draws"], dev), P.to_device(s["meshlets"], dev), P.to_device(s["commands"], dev)
        if soa:
            ctx.upload_meshlets(mlb, len(s["meshlets"]))
        dccb = torch.from_numpy(s["count4"].view(np.int32).copy()).to(dev)
        mvb = torch.from_numpy(s["mvb"].view(np.int32).copy()).to(dev)
        d_pay = torch.full((ncmd * 64,), -1, dtype=torch.int32, device=dev)
        d_cnt = torch.full((ncmd,), -1, dtype=torch.int32, device=dev)
        ctx.taskcull(cd, 0, dcb, dccb, db, mlb, mvb, gp.desc, d_pay, d_cnt)
        n = int(s["count4"][1]) * 64
        cnt_g = d_cnt.cpu().numpy().view(np.uint32)
        assert (cnt_g[:n] == cnt_o[:n]).all()
        pay_g = d_pay.cpu().numpy().view(np.uint32).reshape(ncmd, 64)
        for c in range(n):
            assert (pay_g[c, :cnt_o[c]] == pay_o[c, :cnt_o[c]]).all(), c
        assert (mvb.cpu().numpy().view(np.uint32) == mvb_o).all()
        assert cnt_o[:n].sum() > 0
    finally:
        ctx.close()


# ---------------------------------------------------------------------------------------------------------------- drawcull
def special_draw_scene(seed):
    from scenes import make_scene
    rng = np.random.default_rng(seed)
    scene = make_scene(seed=seed, n_draws=400, n_meshes=4, lods=5, meshlets_lod0=90)
    d, m = scene["draws"], scene["meshes"]
    for i in np.nonzero(rng.random(len(d)) < 0.3)[0]:
        field = rng.integers(0, 3)
        if field == 0:
            d["position"][i][rng.integers(0, 3)] = rng.choice(SPECIAL)
        elif field == 1:
            d["scale"][i] = rng.choice(SPECIAL)
        else:
            d["orientation"][i][rng.integers(0, 4)] = rng.choice(SPECIAL)
    m["radius"][1] = rng.choice(SPECIAL)
    m["center"][2][rng.integers(0, 3)] = rng.choice(SPECIAL)
    m["lods"][3][2]["error"] = np.nan
    m["lods"][3][3]["error"] = -np.inf
    scene["dvb0"] = rng.integers(0, 2, len(d)).astype(np.uint32)
    w, h = scene["viewport"]
    bad = special_depth(w, h, seed)
    k = rng.random((h, w)) < 0.15
    scene["depth"] = scene["depth"].copy()
    scene["depth"][k] = bad[k]
    return scene


def draw_pass(impl, scene, late, task):
    import passes
    pyr = oracle.Pyramid(*scene["viewport"])
    impl.depthreduce(scene["depth"], pyr)
    cd = scene["cull"].copy()
    cd["occlusionEnabled"] = 1
    cd["pyramidWidth"], cd["pyramidHeight"] = pyr.width, pyr.height
    dvb = scene["dvb0"].copy()
    commands, count4 = passes.run_drawcull(impl, scene, cd, late, task, dvb, pyr)
    return commands, count4, dvb, cd


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (no reference tree and no prebuilt .so)")
@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("late", [0, 1])
@pytest.mark.parametrize("task", [0, 1])
def test_drawcull_oracle_equals_reference_on_special_values(seed, late, task):
    scene = special_draw_scene(700 + seed)
    co, c4o, dvo, _ = draw_pass(oracle, scene, late, task)
    cr, c4r, dvr, _ = draw_pass(R, scene, late, task)
    assert c4o.tolist() == c4r.tolist() and (dvo == dvr).all()
    # the reference appends in atomics order = invocation order when serialised: same as the oracle
    assert co[:int(c4o[0])].tobytes() == cr[:int(c4r[0])].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("late", [0, 1])
@pytest.mark.parametrize("task", [0, 1])
def test_drawcull_hip_equals_oracle_on_special_values(seed, late, task):
    import torch
    import gpu_passes as G
    from niagara_amd import pipeline as P
    scene = special_draw_scene(700 + seed)
    co, c4o, dvo, cd = draw_pass(oracle, scene, late, task)
    ctx = P.Context()
    try:
        g = G.GpuScene(ctx, scene, True)
        g.depthreduce(scene["depth"])
        dcb, dccb, dvb = g.drawcull(cd, late, task, scene["dvb0"])
        n = int(c4o[0])
        assert G.host_u32(dccb)[0] == n
        dt = L.TASKCMD if task else L.DRAWCMD
        assert P.from_device(dcb, dt)[:n].tobytes() == co[:n].tobytes()
        assert (G.host_u32(dvb) == dvo).all()
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------------------- depthreduce
@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (no reference tree and no prebuilt .so)")
@pytest.mark.parametrize("size", [(64, 64), (100, 75), (512, 256), (33, 2)])
def test_depthreduce_oracle_equals_reference_with_nonfinite_depth(size):
    """min(x, y) = y < x ? y : x keeps x when y is NaN: which NaNs survive a level depends on the operand order"""
    w, h = size
    depth = special_depth(w, h, w * 7 + h)
    po, pr = oracle.Pyramid(w, h), oracle.Pyramid(w, h)
    oracle.depthreduce(depth, po)
    R.depthreduce(depth, pr)
    assert po.data.tobytes() == pr.data.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(64, 64), (100, 75), (512, 256), (33, 2), (1024, 512), (4096, 4096)])
def test_depthreduce_hip_equals_oracle_with_nonfinite_depth(size):
    import torch
    from niagara_amd import pipeline as P
    w, h = size
    depth = special_depth(w, h, w * 7 + h)
    po = oracle.Pyramid(w, h)
    oracle.depthreduce(depth, po)
    ctx = P.Context()
    try:
        pg = P.DepthPyramid(ctx.device, w, h)
        ctx.depthreduce(torch.from_numpy(depth).to(ctx.device), w, h, pg.desc)
        assert pg.data.cpu().numpy().tobytes() == po.data.tobytes()
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------------ degenerate cameras
WEIRD_CAMERAS = ["nan_view", "inf_translation", "zero_frustum", "inverted_near_far", "huge_scale_view"]


def weird_camera(cd, case):
    cd = cd.copy()
    if case == "nan_view":
        cd["view"][0][5] = np.nan
    elif case == "inf_translation":
        cd["view"][0][14] = np.inf
    elif case == "zero_frustum":
        cd["frustum"][0][:] = 0
    elif case == "inverted_near_far":
        cd["znear"], cd["zfar"] = 50.0, 0.01
    elif case == "huge_scale_view":
        cd["view"][0][:12] *= np.float32(1e30)
    return cd


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (no reference tree and no prebuilt .so)")
@pytest.mark.parametrize("case", WEIRD_CAMERAS)
@pytest.mark.parametrize("late", [0, 1])
def test_oracle_equals_reference_with_degenerate_cameras(case, late):
    s = special_scene(950)
    s["cull"] = weird_camera(s["cull"], case)
    co, io, mo, _, _ = run_cpu(oracle, s, late)
    cr, ir, mr, _, _ = run_cpu(R, s, late)
    assert co.tolist() == cr.tolist() and (io == ir).all() and (mo == mr).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", WEIRD_CAMERAS)
@pytest.mark.parametrize("late", [0, 1])
def test_hip_equals_oracle_with_degenerate_cameras(case, late):
    import torch
    from niagara_amd import pipeline as P
    s = special_scene(950)
    s["cull"] = weird_camera(s["cull"], case)
    co, io, mo, cd, pyr = run_cpu(oracle, s, late)
    ctx = P.Context()
    try:
        dev = ctx.device
        gp = P.DepthPyramid(dev, 256, 192)
        ctx.depthreduce(torch.from_numpy(s["depth"]).to(dev), 256, 192, gp.desc)
        db, mlb, dcb = P.to_device(s["draws"], dev), P.to_device(s["meshlets"], dev), P.to_device(s["commands"], dev)
        ctx.upload_meshlets(mlb, len(s["meshlets"]))
        dccb = torch.from_numpy(s["count4"].view(np.int32).copy()).to(dev)
        mvb = torch.from_numpy(s["mvb"].view(np.int32).copy()).to(dev)
        cib = torch.zeros(s["n"] * 64 + 256, dtype=torch.int32, device=dev)
        ccb = torch.zeros(4, dtype=torch.int32, device=dev)
        ctx.clustercull(cd, late, dcb, dccb, db, mlb, mvb, gp.desc, cib, ccb)
        total = int(ccb[0].item())
        assert total == int(co[0])
        assert (cib.cpu().numpy().view(np.uint32)[:total] == io).all()
        assert (mvb.cpu().numpy().view(np.uint32) == mo).all()
    finally:
        ctx.close()


# --------------------------------------------------------------------------------------------- magnitudes (filter error bound)
MAGNITUDES = [(1e4, 1.0, 1e9), (1.0, 1e-4, 200.0), (1e4, 1e3, 1e9), (1e-3, 1e-3, 200.0), (3e5, 1e-2, 1e9)]


def magnitude_scene(case, seed=31):
    pos_scale, draw_scale, zfar = MAGNITUDES[case]
    draws, meshlets, commands, n = synth.cluster_scene(400, 4, seed=seed, scene_radius=30.0)
    draws["position"] *= np.float32(pos_scale)
    draws["scale"] *= np.float32(draw_scale)
    cd = host.build_cull_data(draw_count=len(draws), cullingEnabled=1, clusterBackfaceEnabled=1)
    cd["zfar"] = zfar
    return draws, meshlets, commands, n, cd


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(MAGNITUDES)))
def test_filter_is_exact_across_magnitudes(case):
    """the conservative frustum filter's error bound scales with the draw's position and scale: scenes 1e-3 .. 3e5 units
    across, scales 1e-4 .. 1e3, with and without the filter, against the oracle"""
    import torch
    from niagara_amd import pipeline as P
    draws, meshlets, commands, n, cd = magnitude_scene(case)
    c4 = synth.count4_for(n)
    cib_o, cc4_o = np.zeros(n * 64, np.uint32), np.zeros(4, np.uint32)
    oracle.clustercull(cd, 0, commands, c4, draws, meshlets, None, None, cib_o, cc4_o)
    total_o = int(cc4_o[0])
    ctx = P.Context()
    try:
        dev = ctx.device
        db, mlb, dcb = P.to_device(draws, dev), P.to_device(meshlets, dev), P.to_device(commands, dev)
        ctx.upload_meshlets(mlb, len(meshlets))
        dccb = torch.from_numpy(c4.view(np.int32).copy()).to(dev)
        cib = torch.zeros(n * 64 + 256, dtype=torch.int32, device=dev)
        ccb = torch.zeros(4, dtype=torch.int32, device=dev)
        ctx.clustercull(cd, 0, dcb, dccb, db, mlb, None, None, cib, ccb)
        total = int(ccb[0].item())
        assert total == total_o
        assert (cib.cpu().numpy().view(np.uint32)[:total] == cib_o[:total_o]).all()
    finally:
        ctx.close()
