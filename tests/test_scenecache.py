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
