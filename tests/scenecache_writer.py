"""Test-side writer of niagara's scene cache file, restating saveSceneCache (src/scenecache.cpp:120-203): the header of
:16-55 and the sections in the order they are written.  TEST INFRASTRUCTURE — the product only reads such files
(nv_scenecache_info / nv_scenecache_read).

The meshopt-coded streams of a compressed file are opaque to the reader (it steps over them by the sizes in the
header), so `compressed=True` writes arbitrary bytes of the stated sizes there.  Parity: the reader is pinned against the
reference's own saveSceneCache / loadSceneCache (src/scenecache.cpp compiled in place, oracle/ref_scenecache.cpp) for the
raw layout, and this writer is itself held to the reference's loader (tests/test_scenecache.py).  What stays unpinned is
the `compressed` layout only — the codecs are meshoptimizer's, which the reference does not vendor, so niagara's code
cannot produce such a file here; for it this writer restates :163-182 (same sections, sizes from the header).
"""
import struct

import numpy as np

MAGIC, VERSION = 0x434E4353, 7          # src/scenecache.cpp:12-13
VERTEX_BYTES, MATERIAL_BYTES, LIGHT_BYTES, ANIMATION_BYTES, KEYFRAME_BYTES = 16, 64, 32, 24, 32  # src/scene.h:25-66,119-136


def write_scene_cache(path, meshes, meshlets, draws, *, compressed=False, vertex_count=100, index_count=300, meshletdata_count=500,
                      meshletvtx0_count=64, material_count=3, light_count=2, animation_count=1, keyframe_count=4, texture_paths=2,
                      omm=(10, 20, 3), camera=((1.0, 2.0, 3.0), (0.0, 0.0, 0.0, 1.0), 1.2, 0.5), sun=(0.0, -1.0, 0.0), hash_meta=0x1122334455667788,
                      clrt_mode=False, omm_states=0, magic=MAGIC, version=VERSION, max_vertices=64, max_triangles=96, seed=5):
    rng = np.random.default_rng(seed)

    def junk(n):
        return rng.integers(0, 256, int(n), dtype=np.uint8).tobytes()

    sizes = dict(v=vertex_count * VERTEX_BYTES, i=index_count * 4, md=meshletdata_count * 4, v0=meshletvtx0_count * 2)
    if compressed:  # meshopt streams: shorter, odd-sized
        sizes = {k: max(1, v // 3 + 1) for k, v in sizes.items()}
    (pos, quat, fovy, znear) = camera
    header = struct.pack("<IIQII??2x4I6I6I4I3f4fff3f4x", magic, version, hash_meta, max_vertices, max_triangles, clrt_mode, compressed,
                         *([sizes["v"], sizes["i"], sizes["md"], sizes["v0"]] if compressed else [0, 0, 0, 0]),
                         vertex_count, index_count, len(meshlets), meshletdata_count, meshletvtx0_count, len(meshes),
                         material_count, len(draws), texture_paths, light_count, animation_count, keyframe_count,
                         omm[0], omm[1], omm[2], omm_states, *pos, *quat, fovy, znear, *sun)
    assert len(header) == 160
    with open(path, "wb") as f:
        f.write(header)                                          # :161
        f.write(junk(sizes["v"]))                                # :163-166 vertices
        f.write(junk(sizes["i"]))                                # :168-171 indices
        f.write(np.ascontiguousarray(meshlets).tobytes())        # :173 raw Meshlet array, also in compressed mode
        f.write(junk(sizes["md"]))                               # :174-177 meshlet data
        f.write(junk(sizes["v0"]))                               # :179-182 RT vertices
        f.write(np.ascontiguousarray(meshes).tobytes())          # :184 raw Mesh array
        f.write(junk(material_count * MATERIAL_BYTES))           # :185
        f.write(np.ascontiguousarray(draws).tobytes())           # :186 raw MeshDraw array
        f.write(junk(light_count * LIGHT_BYTES))                 # :187
        f.write(junk(animation_count * ANIMATION_BYTES))         # :188
        f.write(junk(keyframe_count * KEYFRAME_BYTES))           # :189
        f.write(junk(omm[0]) + junk(omm[1]) + junk(omm[2] * 4))  # :191-193
        f.write(junk(texture_paths * 256))                       # :195-200
    return sizes
