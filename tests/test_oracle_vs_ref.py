"""Pins the CPU oracle against the reference's own shader sources executing on the CPU (oracle/_ref).

oracle/_ref = /root/reference/src/shaders/{drawcull,tasksubmit,clustercull,clustersubmit,depthreduce}.comp.glsl and
math.h (and meshlet.task.glsl, meshlet.mesh.glsl), rewritten syntactically by oracle/ref_translate.py and compiled against oracle/glsl_shim.h; plus verbatim
host helpers (src/niagara.cpp:424-481, src/resources.cpp:280-292).  Bit-exact agreement is required everywhere except
the one documented place where the oracle deliberately differs: ceil(log2(x)) is exact in the oracle and libm-rounded
in the shim (test_occlusion_mip_differs_only_just_above_powers_of_two).
"""
import numpy as np
import pytest

import oracle
import oracle.ref as R
from niagara_amd import layouts as L

import passes
from scenes import flag_matrix, make_scene

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (no reference tree and no prebuilt .so)")


def _bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


# ---------------------------------------------------------------- scalar helpers (src/shaders/math.h:2-49)

def test_rotate_quat_bit_exact():
    rng = np.random.default_rng(1)
    import ctypes as C
    out = np.zeros(3, np.float32)
    for _ in range(2000):
        v = rng.normal(size=3).astype(np.float32) * np.float32(10 ** rng.uniform(-3, 3))
        q = rng.normal(size=4).astype(np.float32)
        q /= np.linalg.norm(q)
        oracle.lib().orc_rotate_quat(v.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert (_bits(out) == _bits(R.rotate_quat(v, q))).all()


def _orc_project(c, r, znear, p00, p11):
    import ctypes as C
    aabb = np.zeros(4, np.float32)
    cc = np.asarray(c, np.float32)
    ok = oracle.lib().orc_project_sphere(cc.ctypes.data_as(C.c_void_p), C.c_float(r), C.c_float(znear), C.c_float(p00), C.c_float(p11),
                                         aabb.ctypes.data_as(C.c_void_p))
    return bool(ok), aabb


def test_project_sphere_bit_exact_including_near_plane_straddle():
    rng = np.random.default_rng(2)
    for i in range(4000):
        c = (rng.normal(size=3) * 10).astype(np.float32)
        c[2] = np.float32(abs(c[2]) + rng.uniform(0, 30))
        r = np.float32(rng.uniform(0, 3)) if i % 7 else np.float32(0)
        if i % 11 == 0:
            c[2] = np.float32(r + np.float32(0.1))  # exactly on the c.z < r + znear boundary
        ok_o, a_o = _orc_project(c, r, 0.1, 1.07, 1.43)
        ok_r, a_r = R.project_sphere(c, r, 0.1, 1.07, 1.43)
        assert ok_o == ok_r
        if ok_o:
            assert (_bits(a_o) == _bits(a_r)).all()


def _orc_mip(aabb, pw, ph):
    import ctypes as C
    a = np.asarray(aabb, np.float32)
    return float(oracle.lib().orc_occlusion_mip(a.ctypes.data_as(C.c_void_p), C.c_float(pw), C.c_float(ph)))


def test_occlusion_mip_matches_reference_on_random_boxes():
    rng = np.random.default_rng(3)
    mism = 0
    for _ in range(20000):
        x0, y0 = rng.uniform(-0.2, 1.0, 2)
        w, h = 10 ** rng.uniform(-4, 0.3, 2)
        aabb = np.array([x0, y0, x0 + w, y0 + h], np.float32)
        pw, ph = (2048.0, 2048.0) if rng.random() < 0.5 else (512.0, 256.0)
        lo, lr = _orc_mip(aabb, pw, ph), R.occlusion_mip(aabb, pw, ph)
        mism += min(lo, 16.0) != min(lr, 16.0)
    assert mism == 0


def test_occlusion_mip_edge_cases():
    # zero-size box (radius 0), negative size, huge size: the sampler clamps levels to [0, levels-1] (maxLod 16)
    for aabb in ([0.3, 0.3, 0.3, 0.3], [0, 0, 1e30, 1e30], [0.25, 0.25, 0.25 + 2 ** -11, 0.25 + 2 ** -11]):
        lo, lr = _orc_mip(aabb, 2048.0, 2048.0), R.occlusion_mip(aabb, 2048.0, 2048.0)
        assert min(max(lo, 0.0), 16.0) == min(max(lr, 0.0), 16.0), aabb
    # inverted box (cannot come out of projectSphere): the reference's log2(negative) is NaN, i.e. an undefined LOD;
    # the oracle DEFINES it as level 0
    assert np.isnan(R.occlusion_mip([0.5, 0.5, 0.4, 0.4], 2048.0, 2048.0)) and _orc_mip([0.5, 0.5, 0.4, 0.4], 2048.0, 2048.0) == 0.0
    assert _orc_mip([0.0, 0.0, 1.0, 1.0], 2048.0, 2048.0) == R.occlusion_mip([0.0, 0.0, 1.0, 1.0], 2048.0, 2048.0)


def test_occlusion_mip_differs_only_just_above_powers_of_two():
    """DEFINED SEMANTIC.  For a footprint a few ulp above 2^k texels, ceil(log2(x)) is mathematically k+1; libm's
    log2f rounds log2(x) to exactly k, so the shim (and any GPU's approximate log2) may answer k.  The oracle uses the
    exact value (the conservative, coarser mip).  This test documents that the two differ there and nowhere else."""
    pw = 2048.0
    diffs = []
    for k in range(1, 11):
        for ulps in (-2, -1, 0, 1, 2, 3):
            size_texels = np.float32(2.0 ** k)
            for _ in range(abs(ulps)):
                size_texels = np.nextafter(size_texels, np.float32(np.inf if ulps > 0 else 0), dtype=np.float32)
            w = np.float32(size_texels / np.float32(pw))  # exact: power-of-two divide
            aabb = np.array([0.0, 0.0, w, w], np.float32)
            # off-grid origin so that the "fits in 2x2 at the finer mip" refinement is false and the raw level shows
            aabb[:2] += np.float32(0.3)
            aabb[2:] = aabb[:2] + w
            got_w = np.float32(aabb[2] - aabb[0]) * np.float32(pw)
            lo, lr = _orc_mip(aabb, pw, pw), R.occlusion_mip(aabb, pw, pw)
            if lo != lr:
                diffs.append((k, float(got_w), lo, lr))
    for k, x, lo, lr in diffs:
        assert 2.0 ** round(np.log2(x)) < x < 2.0 ** round(np.log2(x)) * (1 + 1e-6), (k, x)  # only just above a power of two
        assert lo == lr + 1  # oracle is the exact (coarser) answer


def test_cone_cull_bit_exact():
    import ctypes as C
    rng = np.random.default_rng(4)
    for _ in range(4000):
        c = (rng.normal(size=3) * 5).astype(np.float32)
        axis = rng.normal(size=3).astype(np.float32)
        axis /= np.linalg.norm(axis)
        cutoff = np.float32(rng.integers(-127, 128) / 127.0)
        r = np.float32(rng.uniform(0, 1))
        o = oracle.lib().orc_cone_cull(c.ctypes.data_as(C.c_void_p), C.c_float(r), axis.ctypes.data_as(C.c_void_p), C.c_float(cutoff))
        assert bool(o) == R.cone_cull(c, r, axis, cutoff)


# ---------------------------------------------------------------- host helpers

def test_previous_pow2_and_mip_levels():
    for v in list(range(1, 70)) + [255, 256, 257, 1023, 1024, 1025, 2047, 2048, 4096, 4097, 1 << 20]:
        assert oracle.lib().orc_previous_pow2(v) == R.lib().ref_previous_pow2(v)
    assert R.lib().ref_previous_pow2(4096) == 2048  # strictly below
    for w, h in [(1, 1), (2, 1), (1, 2), (512, 512), (512, 256), (2048, 2048), (1000, 3)]:
        assert oracle.lib().orc_image_mip_levels(w, h) == R.lib().ref_image_mip_levels(w, h)


def test_pcg32_stream_and_synth_draw_positions():
    import ctypes as C
    R.lib().ref_rng_seed(C.c_uint64(0x42))
    state = C.c_uint64(0x42)
    for _ in range(1000):
        assert oracle.lib().orc_pcg32(C.byref(state), C.c_uint64(0xda3e39cb94b95bdb)) == R.lib().ref_rand32()
    # the draw generator consumes: rand32 (mesh), 3x rand01 (position), rand01 (scale), 3x rand01 (axis), rand01 (angle)
    R.lib().ref_rng_seed(C.c_uint64(0x42))
    draws = oracle.synth_draws(50, 5, 300.0)
    for i in range(50):
        assert draws[i]["meshIndex"] == R.lib().ref_rand32() % 5
        for k in range(3):
            assert draws[i]["position"][k] == np.float32(np.float32(np.float32(R.lib().ref_rand01()) * np.float32(300)) * np.float32(2)) - np.float32(300)
        assert draws[i]["scale"] == (np.float32(R.lib().ref_rand01()) + np.float32(1)) * np.float32(2)
        for _ in range(4):
            R.lib().ref_rand01()


def test_projection_and_frustum_planes():
    import ctypes as C
    for vw, vh, fov in [(1024, 768, np.radians(70.0)), (4096, 4096, np.radians(70.0)), (1920, 1080, 1.0)]:
        proj = np.zeros(16, np.float32)
        fr = np.zeros(4, np.float32)
        R.lib().ref_perspective(C.c_float(fov), C.c_float(np.float32(vw) / np.float32(vh)), C.c_float(0.1), proj.ctypes.data_as(C.c_void_p),
                                fr.ctypes.data_as(C.c_void_p))
        cd = oracle.make_cull_data(fovy=fov, viewport=(vw, vh))
        assert _bits(cd["P00"][0]) == _bits(proj[0]) and _bits(cd["P11"][0]) == _bits(proj[5])
        assert (_bits(cd["frustum"][0]) == _bits(fr)).all()


# ---------------------------------------------------------------- the passes

@pytest.mark.parametrize("late", [0, 1])
@pytest.mark.parametrize("task", [0, 1])
def test_drawcull_flag_matrix(late, task):
    scene = make_scene(seed=5 + late * 2 + task, n_draws=400, post_pass_fraction=0.1)
    rng = np.random.default_rng(9)
    pyr_o, pyr_r = oracle.Pyramid(*scene["viewport"]), oracle.Pyramid(*scene["viewport"])
    oracle.depthreduce(scene["depth"], pyr_o)
    R.depthreduce(scene["depth"], pyr_r)
    assert pyr_o.data.tobytes() == pyr_r.data.tobytes()
    for flags in flag_matrix():
        for post in (0, 1):
            cd = passes.set_flags(scene["cull"], flags)
            dvb0 = (rng.random(len(scene["draws"])) < 0.6).astype(np.uint32)
            dvb_o, dvb_r = dvb0.copy(), dvb0.copy()
            co, c4o = passes.run_drawcull(oracle, scene, cd, late, task, dvb_o, pyr_o, post)
            cr, c4r = passes.run_drawcull(R, scene, cd, late, task, dvb_r, pyr_r, post)
            assert c4o[0] == c4r[0], (flags, post)
            assert co.tobytes() == cr.tobytes(), (flags, post)
            assert (dvb_o == dvb_r).all(), (flags, post)


def test_tasksubmit_and_clustersubmit_padding_and_clamps():
    for count in [0, 1, 63, 64, 65, 1000, 4095, 4096]:
        a, b = np.full(count + 80, 7, dtype=L.TASKCMD), np.full(count + 80, 7, dtype=L.TASKCMD)
        c4a, c4b = np.array([count, 9, 9, 9], np.uint32), np.array([count, 9, 9, 9], np.uint32)
        oracle.tasksubmit(c4a, a)
        R.tasksubmit(c4b, b)
        assert (c4a == c4b).all() and a.tobytes() == b.tobytes()
    # overflow clamp: count > TASK_WGLIMIT writes nothing past the limit and clamps X to 65535
    big = L.TASK_WGLIMIT + 5
    a, b = np.zeros(L.TASK_WGLIMIT + 64, dtype=L.TASKCMD), np.zeros(L.TASK_WGLIMIT + 64, dtype=L.TASKCMD)
    c4a, c4b = np.array([big, 0, 0, 0], np.uint32), np.array([big, 0, 0, 0], np.uint32)
    oracle.tasksubmit(c4a, a)
    R.tasksubmit(c4b, b)
    assert (c4a == c4b).all() and c4a[1] == 65535
    for count in [0, 1, 255, 256, 257, 5000]:
        a, b = np.full(count + 300, 5, np.uint32), np.full(count + 300, 5, np.uint32)
        c4a, c4b = np.array([count, 9, 9, 9], np.uint32), np.array([count, 9, 9, 9], np.uint32)
        oracle.clustersubmit(c4a, a)
        R.clustersubmit(c4b, b)
        assert (c4a == c4b).all() and (a == b).all()
        assert tuple(c4a[1:]) == (16, (count + 255) // 256, 16)


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_two_frame_protocol_all_cluster_flags(seed):
    # seed 23: a quarter of the draws are postPass draws, so every frame has the third phase (src/niagara.cpp:1781-1787)
    scene = make_scene(seed=seed, n_draws=250, meshlets_lod0=130, zero_radius_fraction=0.02, post_pass_fraction=0.25 if seed == 23 else 0.0)
    for flags in [(1, 1, 1, 1, 1), (1, 1, 1, 0, 1), (1, 0, 0, 0, 0), (0, 1, 1, 1, 0), (1, 1, 0, 1, 1)]:
        fo = passes.run_frames(oracle, scene, flags, frames=2)
        fr = passes.run_frames(R, scene, flags, frames=2)
        for a, b in zip(fo, fr):
            assert a["pyramid"].tobytes() == b["pyramid"].tobytes()
            assert ("post" in a) == (seed == 23)
            for phase in [p for p in ("early", "late", "post") if p in a]:
                for key in ("commands", "count4", "cib", "cc4", "dvb", "mvb"):
                    assert a[phase][key].tobytes() == b[phase][key].tobytes(), (flags, phase, key)
        # the protocol does something: frame 0 early emits nothing, frame 0 late establishes the visible set
        assert fo[0]["early"]["cc4"][0] == 0 and fo[0]["late"]["cc4"][0] > 0


@pytest.mark.parametrize("size", [(64, 64), (128, 32), (100, 75), (257, 130), (33, 2), (5, 3)])
def test_depthreduce_pow2_and_ragged_sizes(size):
    w, h = size
    rng = np.random.default_rng(w * 1000 + h)
    depth = rng.random((h, w)).astype(np.float32)
    po, pr = oracle.Pyramid(w, h), oracle.Pyramid(w, h)
    oracle.depthreduce(depth, po)
    R.depthreduce(depth, pr)
    assert po.levels == pr.levels and po.data.tobytes() == pr.data.tobytes()
    assert po.level(po.levels - 1).shape == (1, 1)


def test_cluster_pass_postpass_matrix():
    """clustercull with postPass != 0 (bits are updated but not read) and every cluster flag, oracle vs reference shader"""
    from niagara_amd import host, synth
    rng = np.random.default_rng(91)
    draws, meshlets, commands, n = synth.cluster_scene(200, 5, seed=8)
    draws["position"] *= np.float32(0.2)
    commands["lateDrawVisibility"][:n] = rng.integers(0, 2, n)
    cd = host.build_cull_data(draw_count=len(draws), cullingEnabled=1, pyramid=(128, 128))
    mvb0 = rng.integers(0, 2 ** 32, n * 2 + 3, dtype=np.uint64).astype(np.uint32)
    pyr = oracle.Pyramid(256, 192)
    oracle.depthreduce(make_scene(seed=3)["depth"], pyr)
    c4 = synth.count4_for(n)
    for late in (0, 1):
        for coe in (0, 1):
            for cbe in (0, 1):
                for post in (0, 1):
                    c = cd.copy()
                    c["clusterOcclusionEnabled"], c["clusterBackfaceEnabled"], c["postPass"] = coe, cbe, post
                    outs = []
                    for impl in (oracle, R):
                        cib, cc4, mvb = np.zeros(len(commands) * 64 + 256, np.uint32), np.zeros(4, np.uint32), mvb0.copy()
                        impl.clustercull(c, late, commands, c4, draws, meshlets, mvb, pyr, cib, cc4)
                        outs.append((cib, cc4, mvb))
                    for a, b in zip(*outs):
                        assert a.tobytes() == b.tobytes(), (late, coe, cbe, post)


def test_taskcull_equals_the_reference_task_shader():
    """meshlet.task.glsl (TASK_CULL = 1, src/config.h:8) executing on the CPU vs the oracle's taskcull: EmitMeshTasksEXT count,
    payload entries and visibility words over LATE x clusterOcclusion x backface x postPass, on a scene's own task commands and on
    the config-3 style command list"""
    from niagara_amd import host, synth
    rng = np.random.default_rng(17)
    cases = []
    # (a) commands produced by drawcull<TASK> -> tasksubmit of a frame scene
    scene = make_scene(seed=23, n_draws=500, meshlets_lod0=130, zero_radius_fraction=0.02)
    pyr = oracle.Pyramid(*scene["viewport"])
    oracle.depthreduce(scene["depth"], pyr)
    cd = passes.set_flags(scene["cull"], (1, 1, 1, 1, 1))
    cmds, c4 = passes.run_drawcull(oracle, scene, cd, 0, 1, np.ones(len(scene["draws"]), np.uint32), pyr)
    oracle.tasksubmit(c4, cmds)
    cmds["lateDrawVisibility"][:int(c4[0])] = rng.integers(0, 2, int(c4[0]))
    cases.append((cd, cmds, c4, scene["draws"], scene["meshlets"], (scene["slots"] + 31) // 32 + 2, pyr))
    # (b) dense command list
    draws, meshlets, commands, n = synth.cluster_scene(150, 4, seed=8)
    draws["position"] *= np.float32(0.2)
    commands["lateDrawVisibility"][:n] = rng.integers(0, 2, n)
    cd2 = host.build_cull_data(draw_count=len(draws), cullingEnabled=1, pyramid=(pyr.width, pyr.height))
    cases.append((cd2, commands, synth.count4_for(n), draws, meshlets, n * 2 + 3, pyr))
    emitted = 0
    for cd0, cmds, c4, draws, meshlets, words, pyr in cases:
        ncmd = int(c4[1]) * 64
        mvb0 = rng.integers(0, 2 ** 32, words, dtype=np.uint64).astype(np.uint32)
        for late in (0, 1):
            for coe in (0, 1):
                for cbe in (0, 1):
                    for post in (0, 1):
                        c = cd0.copy()
                        c["clusterOcclusionEnabled"], c["clusterBackfaceEnabled"], c["postPass"] = coe, cbe, post
                        outs = []
                        for fn in (oracle.taskcull, R.meshlet_task):
                            pay, cnt, mvb = np.zeros((ncmd, 64), np.uint32), np.zeros(ncmd, np.uint32), mvb0.copy()
                            fn(c, late, cmds, c4, draws, meshlets, mvb, pyr, pay, cnt)
                            outs.append((pay, cnt, mvb))
                        (po, co, mo), (pr, cr, mr) = outs
                        assert (co == cr).all(), (late, coe, cbe, post)
                        assert mo.tobytes() == mr.tobytes(), (late, coe, cbe, post)
                        live = np.arange(64)[None, :] < co[:, None]
                        assert (po[live] == pr[live]).all(), (late, coe, cbe, post)
                        emitted += int(co.sum())
    assert emitted > 10000
