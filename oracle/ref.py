"""oracle.ref — ctypes binding over oracle/_ref/libniagara_ref.so: the reference's own shader sources
(translated syntactically by oracle/ref_translate.py) executing on the CPU.  TEST INFRASTRUCTURE ONLY.

The library is built by `make -C oracle ref` where /root/reference exists; the prebuilt .so travels to the
GPU box with the snapshot.  available() is False when neither is possible.
"""
import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libniagara_ref.so")
_lib = None


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
        _lib.ref_occlusion_mip.restype = C.c_float
        _lib.ref_sizeof.restype = C.c_uint32
        _lib.ref_previous_pow2.restype = C.c_uint32
        _lib.ref_image_mip_levels.restype = C.c_uint32
        _lib.ref_rand32.restype = C.c_uint32
        _lib.ref_rand01.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _pyr(p):
    return None if p is None else p.ref()  # same struct layout as oracle.Pyramid


def drawcull(cd, late, task, draws, meshes, commands, count4, dvb, pyr=None):
    lib().ref_drawcull(_p(cd), int(late), int(task), _p(draws), _p(meshes), _p(commands), _p(count4), _p(dvb), _pyr(pyr))


def tasksubmit(count4, commands):
    lib().ref_tasksubmit(_p(count4), _p(commands))


def clustercull(cd, late, commands, count4, draws, meshlets, mvb, pyr, cib, cc4):
    lib().ref_clustercull(_p(cd), int(late), _p(commands), _p(count4), _p(draws), _p(meshlets), _p(mvb), _pyr(pyr), _p(cib), _p(cc4))


def meshlet_mesh(globals_, commands, draws, meshlets, meshlet_data, vertices, cib, cc4, masks, totals3):
    """src/shaders/meshlet.mesh.glsl (MESH_CULL = 1, TASK = false) over the grid in cc4; see oracle/ref_runner.cpp"""
    lib().ref_meshlet_mesh(_p(globals_), _p(commands), _p(draws), _p(meshlets), _p(meshlet_data), _p(vertices), _p(cib), _p(cc4), _p(masks),
                           C.c_uint32(len(masks)), _p(totals3))


def meshlet_task(cd, late, commands, count4, draws, meshlets, mvb, pyr, payloads, payload_counts):
    """src/shaders/meshlet.task.glsl (TASK_CULL = 1) over the grid in count4; see oracle/ref_runner.cpp"""
    lib().ref_meshlet_task(_p(cd), int(late), _p(commands), _p(count4), _p(draws), _p(meshlets), _p(mvb), _pyr(pyr), _p(payloads), _p(payload_counts))


def mesh_bounds(positions):
    """src/scene.cpp:207-220 (mean centre, max distance) over an (n, 3) float32 array -> (center[3], radius)"""
    import numpy as np
    pos = np.ascontiguousarray(positions, np.float32)
    center, radius = np.zeros(3, np.float32), np.zeros(1, np.float32)
    lib().ref_mesh_bounds(_p(pos), C.c_uint32(len(pos)), _p(center), _p(radius))
    return center, radius[0]


def clustersubmit(cc4, cib):
    lib().ref_clustersubmit(_p(cc4), _p(cib))


def depthreduce(depth, pyr):
    h, w = depth.shape
    offs = (C.c_uint32 * 16)(*pyr.mip_offset)
    lib().ref_depthreduce(_p(depth), C.c_uint32(w), C.c_uint32(h), _p(pyr.data), C.c_uint32(pyr.width), C.c_uint32(pyr.height),
                          C.c_uint32(pyr.levels), offs)


def rotate_quat(v, q):
    out = np.zeros(3, np.float32)
    lib().ref_rotate_quat(_p(np.asarray(v, np.float32)), _p(np.asarray(q, np.float32)), _p(out))
    return out


def project_sphere(c, r, znear, p00, p11):
    aabb = np.zeros(4, np.float32)
    ok = lib().ref_project_sphere(_p(np.asarray(c, np.float32)), C.c_float(r), C.c_float(znear), C.c_float(p00), C.c_float(p11), _p(aabb))
    return bool(ok), aabb


def occlusion_mip(aabb, pw, ph):
    return float(lib().ref_occlusion_mip(_p(np.asarray(aabb, np.float32)), C.c_float(pw), C.c_float(ph)))


def cone_cull(c, r, axis, cutoff):
    return bool(lib().ref_cone_cull(_p(np.asarray(c, np.float32)), C.c_float(r), _p(np.asarray(axis, np.float32)), C.c_float(cutoff)))


# ---- the reference's scene-cache code (src/scenecache.cpp compiled in place, oracle/ref_scenecache.cpp) ----
def scene_sizeof(what):
    lib().ref_scene_sizeof.restype = C.c_uint32
    return int(lib().ref_scene_sizeof(int(what)))


def save_scene_cache(path, meshes, meshlets, draws, *, vertex_count=100, index_count=300, meshletdata_count=500, meshletvtx0_count=64,
                     material_count=3, light_count=2, animation_count=1, keyframe_count=4, texture_paths=2,
                     camera=((1.0, 2.0, 3.0), (0.0, 0.0, 0.0, 1.0), 1.2, 0.5), sun=(0.0, -1.0, 0.0), hash_meta=0x1122334455667788, clrt_mode=False,
                     omm_states=0):
    """saveSceneCache (src/scenecache.cpp:120-260) itself, uncompressed (the codecs are meshoptimizer's: not vendored)"""
    cam = np.array(list(camera[0]) + list(camera[1]) + [camera[2], camera[3]], np.float32)
    s3 = np.array(sun, np.float32)
    rc = lib().ref_save_scene_cache(os.fsencode(str(path)), _p(np.ascontiguousarray(meshes)), C.c_uint32(len(meshes)), _p(np.ascontiguousarray(meshlets)),
                                    C.c_uint32(len(meshlets)), _p(np.ascontiguousarray(draws)), C.c_uint32(len(draws)), C.c_uint32(vertex_count),
                                    C.c_uint32(index_count), C.c_uint32(meshletdata_count), C.c_uint32(meshletvtx0_count), C.c_uint32(material_count),
                                    C.c_uint32(light_count), C.c_uint32(animation_count), C.c_uint32(keyframe_count), C.c_uint32(texture_paths), _p(cam), _p(s3),
                                    C.c_uint64(hash_meta), int(bool(clrt_mode)), C.c_uint32(omm_states))
    if rc:
        raise RuntimeError("saveSceneCache failed")


def load_scene_cache(path, mesh_dtype, meshlet_dtype, draw_dtype, hash_meta=0x1122334455667788, clrt_mode=False, omm_states=0):
    """loadSceneCache (src/scenecache.cpp:273-370) itself -> (meshes, meshlets, draws, camera9, sun3), or None if it rejects the file"""
    counts, cam, sun = np.zeros(3, np.uint32), np.zeros(9, np.float32), np.zeros(3, np.float32)
    rc = lib().ref_load_scene_cache(os.fsencode(str(path)), C.c_uint64(hash_meta), int(bool(clrt_mode)), int(omm_states), _p(counts), _p(cam), _p(sun))
    if rc:
        return None
    meshes, meshlets, draws = np.zeros(counts[0], mesh_dtype), np.zeros(counts[1], meshlet_dtype), np.zeros(counts[2], draw_dtype)
    lib().ref_loaded_arrays(_p(meshes), _p(meshlets), _p(draws))
    return meshes, meshlets, draws, cam, sun
