"""ctypes binding of the C ABI in include/niagara_vis.h (niagara_amd/libniagara_vis.so).

The library is the product: there is no Python or CPU fallback.  If the shared object is missing this module raises
at import time, and every device entry point raises NvError on a non-zero status.
"""
import ctypes as C
import os

# The Python host layer hands torch-allocated device pointers and torch streams to the library, so both must sit on
# ONE HIP runtime.  torch bundles its own libamdhip64.so.7; loading it first makes the library's NEEDED entry resolve
# to that same copy (same SONAME).  A C++ host that does not use torch simply links /opt/rocm's runtime.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover - the C ABI itself does not need torch
    pass

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("NV_LIBRARY_PATH") or os.path.join(_HERE, "libniagara_vis.so")  # override: A/B builds during development


class NvError(RuntimeError):
    pass


class PyramidDesc(C.Structure):
    _fields_ = [("d_base", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("levels", C.c_uint32),
                ("mipOffset", C.c_uint32 * 16), ("totalTexels", C.c_uint32)]


class SceneCacheInfo(C.Structure):
    """NvSceneCacheInfo (include/niagara_vis.h)"""
    _fields_ = ([(n, C.c_uint32) for n in ("version", "compressed", "clrtMode", "ommStates")] + [("hashMeta", C.c_uint64)] +
                [(n, C.c_uint32) for n in ("meshletMaxVertices", "meshletMaxTriangles", "vertexCount", "indexCount", "meshletCount",
                                           "meshletdataCount", "meshletvtx0Count", "meshCount", "materialCount", "drawCount",
                                           "texturePathCount", "lightCount", "animationCount", "keyframeCount")] +
                [("cameraPosition", C.c_float * 3), ("cameraOrientation", C.c_float * 4), ("cameraFovY", C.c_float),
                 ("cameraZnear", C.c_float), ("sunDirection", C.c_float * 3)] +
                [(n, C.c_uint64) for n in ("fileSize", "vertexOffset", "indexOffset", "meshletOffset", "meshletdataOffset", "meshOffset",
                                           "drawOffset", "vertexBytes", "indexBytes", "meshletdataBytes")])


if not os.path.exists(SO_PATH):
    raise ImportError("niagara_amd: %s is missing — build it with `make -C niagara_amd/csrc` (or __graft_entry__.build()); "
                      "there is no fallback path" % SO_PATH)

lib = C.CDLL(SO_PATH)

_vp, _u32, _i, _f = C.c_void_p, C.c_uint32, C.c_int, C.c_float
_SIGS = {
    "nv_create": (_i, [C.POINTER(_vp), _i]),
    "nv_destroy": (None, [_vp]),
    "nv_version": (C.c_char_p, []),
    "nv_status": (_i, [_vp, _vp]),
    "nv_set_option": (_i, [_vp, _i, _i]),
    "nv_reserve": (_i, [_vp, _u32, _u32]),
    "nv_share_scene": (_i, [_vp, _vp]),
    "nv_profile_enable": (_i, [_vp, _i]),
    "nv_profile_read": (_i, [_vp, C.POINTER(C.c_float * 5), C.POINTER(C.c_uint32 * 5)]),
    "nv_profile_variants": (_i, [_vp, C.POINTER(C.c_uint32 * 10)]),  # NV_VARIANT_SLOTS
    "nv_upload_meshlets": (_i, [_vp, _vp, _vp, _u32]),
    "nv_upload_meshes": (_i, [_vp, _vp, _vp, _u32]),
    "nv_upload_draws": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "nv_meshlet_bounds": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "nv_update_draws": (_i, [_vp, _vp, _vp, _u32, _u32]),
    "nv_drawcull": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, C.POINTER(PyramidDesc)]),
    "nv_reset_count": (_i, [_vp, _vp, _vp, _vp]),
    "nv_tasksubmit": (_i, [_vp, _vp, _vp, _vp]),
    "nv_clustercull": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, C.POINTER(PyramidDesc), _vp, _vp]),
    "nv_clustersubmit": (_i, [_vp, _vp, _vp, _vp]),
    "nv_taskcull": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, C.POINTER(PyramidDesc), _vp, _vp]),
    "nv_cluster_expand": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "nv_trianglecull": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "nv_depthreduce": (_i, [_vp, _vp, _vp, _u32, _u32, C.POINTER(PyramidDesc)]),
    "nv_previous_pow2": (_u32, [_u32]),
    "nv_division_magic": (_u32, [_u32]),
    "nv_image_mip_levels": (_u32, [_u32, _u32]),
    "nv_pyramid_desc_init": (_i, [C.POINTER(PyramidDesc), _u32, _u32]),
    "nv_build_cull_data": (_i, [_vp, _vp, _vp, _f, _f, _f, _u32, _u32, _u32, _u32, _u32, _i]),
    "nv_assign_visibility_offsets": (_i, [_vp, _u32, _vp, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "nv_synth_draws": (_i, [_vp, _u32, _u32, _f]),
    "nv_mesh_bounds": (_i, [_vp, _u32, _vp, _vp]),
    "nv_shard_range": (None, [C.c_uint64, _u32, _u32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nv_pack_counts": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "nv_set_counts_sink": (_i, [_vp, _vp]),
    "nv_scenecache_info": (_i, [C.c_char_p, C.POINTER(SceneCacheInfo)]),
    "nv_scenecache_read": (_i, [C.c_char_p, C.POINTER(SceneCacheInfo), _vp, _vp, _vp]),
    "nv_probe_cluster_scalars": (_i, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, C.POINTER(PyramidDesc), _vp]),
}
EXPORTS = sorted(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = header/library mismatch, fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc, what):
    if rc != 0:
        names = {-1: "NV_EINVAL", -2: "NV_ENOMEM", -3: "NV_ESTATE", -4: "NV_ENODEV", -5: "NV_EIO", -6: "NV_EFORMAT"}
        raise NvError("%s failed: %s" % (what, names.get(rc, "hipError %d" % rc)))
