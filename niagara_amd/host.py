"""Host-side helpers (no device work): numpy views over the C ABI's host functions.

Mirrors the parts of niagara's main() that prepare cull inputs: CullData (src/niagara.cpp:1487-1516), the depth
pyramid geometry (:1340-1344), the synthetic scene (:969-998) and the visibility-slot prefix (:1002-1020).
"""
import ctypes as C
import os

import numpy as np

from . import layouts as L
from ._lib import PyramidDesc, check, lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def previous_pow2(v):
    return int(lib.nv_previous_pow2(v))


def image_mip_levels(w, h):
    return int(lib.nv_image_mip_levels(w, h))


def pyramid_desc(depth_w, depth_h):
    d = PyramidDesc()
    check(lib.nv_pyramid_desc_init(C.byref(d), depth_w, depth_h), "nv_pyramid_desc_init")
    return d


def build_cull_data(cam_pos=(0, 0, 0), cam_quat=(0, 0, 0, 1), fovy=float(np.radians(70.0)), znear=0.1, draw_distance=200.0,
                    viewport=(1024, 768), pyramid=(512, 512), draw_count=0, lod_step=0, **flags):
    """defaults = niagara's camera (src/niagara.cpp:836-837,1000,1184)"""
    cd = np.zeros(1, dtype=L.CULLDATA)
    pos = np.asarray(cam_pos, dtype=np.float32)
    q = np.asarray(cam_quat, dtype=np.float32)
    check(lib.nv_build_cull_data(_p(cd), _p(pos), _p(q), fovy, znear, draw_distance, viewport[0], viewport[1], pyramid[0], pyramid[1],
                                 draw_count, lod_step), "nv_build_cull_data")
    for k, v in flags.items():
        cd[k] = v
    return cd


def synth_draws(n, mesh_count, scene_radius=300.0):
    d = np.zeros(n, dtype=L.MESHDRAW)
    check(lib.nv_synth_draws(_p(d), n, mesh_count, scene_radius), "nv_synth_draws")
    return d


def assign_visibility_offsets(draws, meshes):
    slots, mask = C.c_uint32(0), C.c_uint32(0)
    check(lib.nv_assign_visibility_offsets(_p(draws), len(draws), _p(meshes), len(meshes), C.byref(slots), C.byref(mask)),
          "nv_assign_visibility_offsets")
    return slots.value, mask.value


def shard_range(total, rank, world):
    b, e = C.c_uint64(0), C.c_uint64(0)
    lib.nv_shard_range(total, rank, world, C.byref(b), C.byref(e))
    return b.value, e.value


def scenecache_info(path):
    """header of a niagara .cache file (src/scenecache.cpp:16-55) + where its Meshlet / Mesh / MeshDraw arrays sit"""
    from ._lib import SceneCacheInfo
    info = SceneCacheInfo()
    check(lib.nv_scenecache_info(os.fsencode(path), C.byref(info)), "nv_scenecache_info")
    return info


def scenecache_read(path):
    """(info, meshes, meshlets, draws) of a niagara .cache file; the arrays are the raw struct arrays of the file"""
    info = scenecache_info(path)
    meshes = np.zeros(info.meshCount, dtype=L.MESH)
    meshlets = np.zeros(info.meshletCount, dtype=L.MESHLET)
    draws = np.zeros(info.drawCount, dtype=L.MESHDRAW)
    check(lib.nv_scenecache_read(os.fsencode(path), C.byref(info), _p(meshes), _p(meshlets), _p(draws)), "nv_scenecache_read")
    return info, meshes, meshlets, draws


def mesh_bounds(positions):
    """Mesh.center / Mesh.radius of src/scene.cpp:207-220 over an (n, 3) float32 array"""
    pos = np.ascontiguousarray(positions, np.float32)
    center, radius = np.zeros(3, np.float32), np.zeros(1, np.float32)
    check(lib.nv_mesh_bounds(_p(pos), len(pos), _p(center), _p(radius)), "nv_mesh_bounds")
    return center, radius[0]
