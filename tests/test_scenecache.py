"""the scene cache reader (nv_scenecache_info / nv_scenecache_read) against a writer that restates
saveSceneCache's layout, then the arrays it returns through the visibility passes."""
import numpy as np
import pytest

from niagara_amd import host
from niagara_amd import layouts as L

import oracle
import passes
from scenecache_writer import write_scene_cache
from scenes import make_scene


@pytest.fixture(scope="module")
def scene():
    return make_scene(seed=41, n_draws=900, meshlets_lod0=120, zero_radius_fraction=0.02)


@pytest.mark.parametrize("compressed", [False, True])
def test_reader_returns_the_raw_arrays_and_the_header(tmp_path, scene, compressed):
    path = tmp_path / "scene.cache"
    sizes = write_scene_cache(path, scene["meshes"], scene["meshlets"], scene["draws"], compressed=compressed)
    info, meshes, meshlets, draws = host.scenecache_read(str(path))
    assert meshes.tobytes() == scene["meshes"].tobytes()
    assert meshlets.tobytes() == scene["meshlets"].tobytes()
    assert draws.tobytes() == scene["draws"].tobytes()
    assert (info.version, info.compressed, info.clrtMode, info.hashMeta) == (7, int(compressed), 0, 0x1122334455667788)
    assert (info.meshletMaxVertices, info.meshletMaxTriangles) == (64, 96)
    assert (info.meshCount, info.meshletCount, info.drawCount) == (len(meshes), len(meshlets), len(draws))
    assert (info.vertexCount, info.indexCount, info.materialCount, info.lightCount, info.texturePathCount) == (100, 300, 3, 2, 2)
    assert list(info.cameraPosition) == [1.0, 2.0, 3.0] and list(info.cameraOrientation) == [0.0, 0.0, 0.0, 1.0]
    assert (info.cameraFovY, info.cameraZnear) == (np.float32(1.2), 0.5) and list(info.sunDirection) == [0.0, -1.0, 0.0]
    assert (info.vertexBytes, info.indexBytes, info.meshletdataBytes) == (sizes["v"], sizes["i"], sizes["md"])
    assert info.vertexOffset == 160 and info.meshletOffset == 160 + sizes["v"] + sizes["i"]


def test_empty_scene_and_partial_reads(tmp_path):
    path = tmp_path / "empty.cache"
    write_scene_cache(path, np.zeros(0, L.MESH), np.zeros(0, L.MESHLET), np.zeros(0, L.MESHDRAW), vertex_count=0, index_count=0,
                      meshletdata_count=0, meshletvtx0_count=0, material_count=0, light_count=0, animation_count=0, keyframe_count=0,
                      texture_paths=0, omm=(0, 0, 0))
    info, meshes, meshlets, draws = host.scenecache_read(str(path))
    assert info.fileSize == 160 and len(meshes) == len(meshlets) == len(draws) == 0


@pytest.mark.parametrize("bad", [dict(magic=0x12345678), dict(version=6), dict(max_vertices=128), dict(max_triangles=124)])
def test_header_checks_of_loadSceneCache(tmp_path, scene, bad):
    """src/scenecache.cpp:284-293: any mismatch rejects the file"""
    path = tmp_path / "bad.cache"
    write_scene_cache(path, scene["meshes"], scene["meshlets"], scene["draws"], **bad)
    with pytest.raises(RuntimeError, match="NV_EFORMAT"):
        host.scenecache_info(str(path))


def test_truncated_and_missing_files(tmp_path, scene):
    path = tmp_path / "t.cache"
    write_scene_cache(path, scene["meshes"], scene["meshlets"], scene["draws"])
    data = open(path, "rb").read()
    info = host.scenecache_info(str(path))
    for cut in (100, 160, int(info.meshletOffset) + 10, int(info.drawOffset) + 47):
        open(tmp_path / "cut.cache", "wb").write(data[:cut])
        with pytest.raises(RuntimeError, match="NV_EFORMAT"):
            host.scenecache_info(str(tmp_path / "cut.cache"))
    with pytest.raises(RuntimeError, match="NV_EIO"):
        host.scenecache_info(str(tmp_path / "nothing.cache"))


def test_cache_arrays_feed_the_oracle_passes(tmp_path, scene):
    """what comes out of the file drives the passes exactly like the arrays that went in"""
    path = tmp_path / "scene.cache"
    write_scene_cache(path, scene["meshes"], scene["meshlets"], scene["draws"], compressed=True)
    _, meshes, meshlets, draws = host.scenecache_read(str(path))
    loaded = dict(scene, meshes=meshes, meshlets=meshlets, draws=draws)
    a = passes.run_frames(oracle, scene, (1, 1, 1, 1, 1), frames=2)
    b = passes.run_frames(oracle, loaded, (1, 1, 1, 1, 1), frames=2)
    for fa, fb in zip(a, b):
        for phase in ("early", "late"):
            for key in ("count4", "commands", "cc4", "cib", "dvb", "mvb"):
                assert fa[phase][key].tobytes() == fb[phase][key].tobytes()


@pytest.mark.gpu
def test_cache_arrays_through_the_hip_passes(tmp_path, scene):
    import gpu_passes as G
    from niagara_amd import pipeline as P
    path = tmp_path / "scene.cache"
    write_scene_cache(path, scene["meshes"], scene["meshlets"], scene["draws"], compressed=True)
    _, meshes, meshlets, draws = host.scenecache_read(str(path))
    loaded = dict(scene, meshes=meshes, meshlets=meshlets, draws=draws)
    ctx = P.Context(0)
    fo = passes.run_frames(oracle, scene, (1, 1, 1, 1, 1), frames=2)
    fg = G.run_frames(ctx, loaded, (1, 1, 1, 1, 1), frames=2)
    for a, b in zip(fo, fg):
        for phase in ("early", "late"):
            for key in ("count4", "commands", "cc4", "cib", "dvb", "mvb"):
                assert a[phase][key].tobytes() == b[phase][key].tobytes(), (phase, key)
    ctx.close()


# ---- pinned against the reference's own scene-cache code (src/scenecache.cpp compiled in place: oracle/ref_scenecache.cpp)
from oracle import ref as R  # noqa: E402

needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (no reference tree and no prebuilt .so)")


@needs_ref
def test_header_layout_is_the_reference_compilers():
    """SceneHeader (src/scenecache.cpp:16-55) as the reference's own translation unit lays it out, against the offsets the
    reader (niagara_amd/csrc/host.cpp SceneCacheHeader) and the test-side writer assume"""
    want = {0: 160, 1: 8, 2: 16, 3: 24, 4: 25, 5: 28, 6: 40, 7: 44, 8: 64, 9: 68, 10: 88, 11: 92, 12: 104, 13: 108, 14: 144,  # SceneHeader
            15: 36, 26: 12, 27: 28,                                                                                        # Camera
            16: 16, 17: 64, 18: 32, 19: 24, 20: 32, 21: 24, 22: 208, 23: 48,                                               # array elements
            24: 0x434E4353, 25: 7}
    got = {k: R.scene_sizeof(k) for k in want}
    assert got == want


@needs_ref
def test_reader_reads_a_file_the_reference_wrote(tmp_path, scene):
    """saveSceneCache itself writes the file; nv_scenecache_read must return the arrays that went in and the header's
    counts / camera, and place every section where the reference put it"""
    path = tmp_path / "ref.cache"
    R.save_scene_cache(path, scene["meshes"], scene["meshlets"], scene["draws"], vertex_count=777, index_count=999, meshletdata_count=1234,
                       meshletvtx0_count=64, material_count=5, light_count=3, animation_count=2, keyframe_count=6, texture_paths=4,
                       camera=((4.0, 5.0, 6.0), (0.1, 0.2, 0.3, 0.9), 0.9, 0.25), sun=(0.3, -0.8, 0.5), hash_meta=0xdeadbeefcafef00d, clrt_mode=True)
    info, meshes, meshlets, draws = host.scenecache_read(str(path))
    assert meshes.tobytes() == scene["meshes"].tobytes()
    assert meshlets.tobytes() == scene["meshlets"].tobytes()
    assert draws.tobytes() == scene["draws"].tobytes()
    assert (info.version, info.compressed, info.clrtMode, info.hashMeta) == (7, 0, 1, 0xdeadbeefcafef00d)
    assert (info.vertexCount, info.indexCount, info.meshletdataCount, info.meshletvtx0Count) == (777, 999, 1234, 64)
    assert (info.materialCount, info.lightCount, info.animationCount, info.keyframeCount, info.texturePathCount) == (5, 3, 2, 6, 4)
    assert np.allclose(list(info.cameraPosition), [4, 5, 6]) and np.allclose(list(info.cameraOrientation), [0.1, 0.2, 0.3, 0.9])
    assert (info.cameraFovY, info.cameraZnear) == (np.float32(0.9), 0.25) and np.allclose(list(info.sunDirection), [0.3, -0.8, 0.5])
    # the file ends where the reader's walk ends: header + every section of saveSceneCache, nothing unaccounted for
    size = path.stat().st_size
    expect = (160 + 777 * 16 + 999 * 4 + len(meshlets) * 24 + 1234 * 4 + 64 * 2 + len(meshes) * 208 + 5 * 64 + len(draws) * 48 + 3 * 32 + 2 * 24
              + 6 * 32 + 4 * 256)
    assert size == expect == info.fileSize
    # and the reference's loader agrees with the reader on the same file
    rm, rl, rd, cam, sun = R.load_scene_cache(path, L.MESH, L.MESHLET, L.MESHDRAW, hash_meta=0xdeadbeefcafef00d, clrt_mode=True)
    assert rm.tobytes() == meshes.tobytes() and rl.tobytes() == meshlets.tobytes() and rd.tobytes() == draws.tobytes()


@needs_ref
def test_reference_loader_accepts_the_test_writer(tmp_path, scene):
    """the test-side writer (which also produces the `compressed` layouts the reference cannot write here) is itself held
    to the reference: loadSceneCache reads its uncompressed files and returns the same arrays as the reader; and it
    rejects what the reader rejects"""
    path = tmp_path / "w.cache"
    write_scene_cache(path, scene["meshes"], scene["meshlets"], scene["draws"], omm=(0, 0, 0))
    got = R.load_scene_cache(path, L.MESH, L.MESHLET, L.MESHDRAW)
    assert got is not None
    rm, rl, rd, cam, sun = got
    _, meshes, meshlets, draws = host.scenecache_read(str(path))
    assert rm.tobytes() == meshes.tobytes() and rl.tobytes() == meshlets.tobytes() and rd.tobytes() == draws.tobytes()
    assert list(cam) == [1.0, 2.0, 3.0, 0.0, 0.0, 0.0, 1.0, np.float32(1.2), 0.5] and list(sun) == [0.0, -1.0, 0.0]
    for bad in (dict(magic=0x12345678), dict(version=6), dict(max_vertices=128), dict(max_triangles=124)):
        write_scene_cache(tmp_path / "bad.cache", scene["meshes"], scene["meshlets"], scene["draws"], omm=(0, 0, 0), **bad)
        assert R.load_scene_cache(tmp_path / "bad.cache", L.MESH, L.MESHLET, L.MESHDRAW) is None
