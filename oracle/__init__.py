"""oracle — CPU restatement of niagara's visibility passes (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (niagara_amd/) never does.  See oracle/oracle.h for the parity-pinning status.

ctypes binding over oracle/liboracle.so (built by `make -C oracle`); numpy structured dtypes mirror
the reference layouts (src/scene.h:10-93, src/niagara.cpp:227-260).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# ---- layouts (independent restatement; tests cross-check against niagara_amd.layouts and oracle/_ref) ----
MESHLET = np.dtype([("center", "<u2", 3), ("radius", "<u2"), ("cone_axis", "i1", 3), ("cone_cutoff", "i1"),
                    ("dataOffset", "<u4"), ("baseVertex", "<u4"), ("vertexCount", "u1"), ("triangleCount", "u1"),
                    ("shortRefs", "u1"), ("padding", "u1")])
MESHDRAW = np.dtype([("position", "<f4", 3), ("scale", "<f4"), ("orientation", "<f4", 4), ("meshIndex", "<u4"),
                     ("meshletVisibilityOffset", "<u4"), ("postPass", "<u4"), ("materialIndex", "<u4")])
MESHLOD = np.dtype([("indexOffset", "<u4"), ("indexCount", "<u4"), ("meshletOffset", "<u4"), ("meshletCount", "<u4"),
                    ("error", "<f4")])
MESH = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("vertexOffset", "<u4"), ("vertexCount", "<u4"),
                 ("ommIndexData", "<u4"), ("ommIndexBase", "<u4"), ("lodCount", "<u4"), ("lodRT", "<u4"),
                 ("padding", "<u4", 2), ("lods", MESHLOD, 8)])
DRAWCMD = np.dtype([("drawId", "<u4"), ("indexCount", "<u4"), ("instanceCount", "<u4"), ("firstIndex", "<u4"),
                    ("vertexOffset", "<u4"), ("firstInstance", "<u4")])
TASKCMD = np.dtype([("drawId", "<u4"), ("taskOffset", "<u4"), ("taskCount", "<u4"), ("lateDrawVisibility", "<u4"),
                    ("meshletVisibilityOffset", "<u4")])
CULLDATA = np.dtype([("view", "<f4", 16), ("P00", "<f4"), ("P11", "<f4"), ("znear", "<f4"), ("zfar", "<f4"),
                     ("frustum", "<f4", 4), ("lodTarget", "<f4"), ("pyramidWidth", "<f4"), ("pyramidHeight", "<f4"),
                     ("drawCount", "<u4"), ("cullingEnabled", "<i4"), ("lodEnabled", "<i4"), ("occlusionEnabled", "<i4"),
                     ("clusterOcclusionEnabled", "<i4"), ("clusterBackfaceEnabled", "<i4"), ("postPass", "<u4"),
                     ("_pad", "<u4", 2)])
assert (MESHLET.itemsize, MESHDRAW.itemsize, MESHLOD.itemsize, MESH.itemsize) == (24, 48, 20, 208)
assert (DRAWCMD.itemsize, TASKCMD.itemsize, CULLDATA.itemsize) == (24, 20, 144)
VERTEX = np.dtype([("vx", "<u2"), ("vy", "<u2"), ("vz", "<u2"), ("tp", "<u2"), ("np", "<u4"), ("tu", "<u2"), ("tv", "<u2")])  # src/shaders/mesh.h:3-9
GLOBALS = np.dtype([("projection", "<f4", 16), ("cullData", CULLDATA), ("screenWidth", "<f4"), ("screenHeight", "<f4"), ("_pad", "<f4", 2)])  # mesh.h:46-51
TRIMASK = np.dtype([("keep", "<u4", 3), ("counts", "<u4")])
assert (VERTEX.itemsize, GLOBALS.itemsize, TRIMASK.itemsize) == (16, 224, 16)

TASK_WGLIMIT = 1 << 22
CLUSTER_LIMIT = 1 << 24


class _Pyr(C.Structure):
    _fields_ = [("base", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("levels", C.c_uint32),
                ("mipOffset", C.c_uint32 * 16), ("totalTexels", C.c_uint32)]


def build(force=False):
    """compile liboracle.so (and oracle/_ref when the reference tree is present)"""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if os.path.isdir("/root/reference/src/shaders"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _lib = C.CDLL(so)
        _lib.orc_occlusion_mip.restype = C.c_float
        _lib.orc_sample_min.restype = C.c_float
        _lib.orc_half_to_float.restype = C.c_float
        _lib.orc_previous_pow2.restype = C.c_uint32
        _lib.orc_image_mip_levels.restype = C.c_uint32
        _lib.orc_pcg32.restype = C.c_uint32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Pyramid:
    """host-memory depth pyramid (one linear fp32 buffer + mip offsets)"""

    def __init__(self, depth_w, depth_h):
        self.s = _Pyr()
        lib().orc_pyramid_init(C.byref(self.s), C.c_uint32(depth_w), C.c_uint32(depth_h))
        self.data = np.zeros(self.s.totalTexels, dtype=np.float32)
        self.s.base = self.data.ctypes.data
        self.width, self.height, self.levels = self.s.width, self.s.height, self.s.levels
        self.mip_offset = [int(x) for x in self.s.mipOffset]

    def level(self, i):
        w, h = max(1, self.width >> i), max(1, self.height >> i)
        return self.data[self.mip_offset[i]:self.mip_offset[i] + w * h].reshape(h, w)

    def ref(self):
        return C.byref(self.s)


def _pyr(p):
    return None if p is None else p.ref()


def make_cull_data(cam_pos=(0, 0, 0), cam_quat=(0, 0, 0, 1), fovy=np.radians(70.0), znear=0.1, draw_distance=200.0,
                   viewport=(1024, 768), pyramid=(512, 512), draw_count=0, lod_step=0):
    cd = np.zeros(1, dtype=CULLDATA)
    pos = np.asarray(cam_pos, dtype=np.float32)
    q = np.asarray(cam_quat, dtype=np.float32)
    lib().orc_build_cull_data(_p(cd), _p(pos), _p(q), C.c_float(fovy), C.c_float(znear), C.c_float(draw_distance),
                              C.c_uint32(viewport[0]), C.c_uint32(viewport[1]), C.c_uint32(pyramid[0]), C.c_uint32(pyramid[1]),
                              C.c_uint32(draw_count), C.c_int(lod_step))
    return cd


def synth_draws(n, mesh_count, scene_radius=300.0):
    d = np.zeros(n, dtype=MESHDRAW)
    lib().orc_synth_draws(_p(d), C.c_uint32(n), C.c_uint32(mesh_count), C.c_float(scene_radius))
    return d


def assign_visibility_offsets(draws, meshes):
    slots, mask = C.c_uint32(0), C.c_uint32(0)
    lib().orc_assign_visibility_offsets(_p(draws), C.c_uint32(len(draws)), _p(meshes), C.byref(slots), C.byref(mask))
    return slots.value, mask.value


def drawcull(cd, late, task, draws, meshes, commands, count4, dvb, pyr=None, threads=0):
    if threads:
        lib().orc_drawcull_mt(_p(cd), int(late), int(task), _p(draws), _p(meshes), _p(commands), _p(count4), _p(dvb), _pyr(pyr),
                              int(threads))
    else:
        lib().orc_drawcull(_p(cd), int(late), int(task), _p(draws), _p(meshes), _p(commands), _p(count4), _p(dvb), _pyr(pyr))


def tasksubmit(count4, commands):
    lib().orc_tasksubmit(_p(count4), _p(commands))


def clustercull(cd, late, commands, count4, draws, meshlets, mvb, pyr, cib, cc4, threads=0):
    if threads:
        lib().orc_clustercull_mt(_p(cd), int(late), _p(commands), _p(count4), _p(draws), _p(meshlets), _p(mvb), _pyr(pyr), _p(cib),
                                 _p(cc4), int(threads))
    else:
        lib().orc_clustercull(_p(cd), int(late), _p(commands), _p(count4), _p(draws), _p(meshlets), _p(mvb), _pyr(pyr), _p(cib), _p(cc4))


def clustersubmit(cc4, cib):
    lib().orc_clustersubmit(_p(cc4), _p(cib))


def taskcull(cd, late, commands, count4, draws, meshlets, mvb, pyr, payloads, payload_counts):
    lib().orc_taskcull(_p(cd), int(late), _p(commands), _p(count4), _p(draws), _p(meshlets), _p(mvb), _pyr(pyr), _p(payloads),
                       _p(payload_counts))


def depthreduce(depth, pyr):
    h, w = depth.shape
    lib().orc_depthreduce(_p(depth), C.c_uint32(w), C.c_uint32(h), pyr.ref())


def probe_cluster_scalars(cd, commands, draws, meshlets, pyr=None):
    out = np.zeros((len(commands), 64, 16), dtype=np.float32)
    lib().orc_probe_cluster_scalars(_p(cd), _p(commands), C.c_uint32(len(commands)), _p(draws), _p(meshlets), _pyr(pyr), _p(out))
    return out


def cluster_expand(commands, meshlets, cib, cc4, records, totals3):
    lib().orc_cluster_expand(_p(commands), _p(meshlets), _p(cib), _p(cc4), _p(records), C.c_uint32(len(records)), _p(totals3))


def trianglecull(globals_, commands, draws, meshlets, meshlet_data, vertices, cib, cc4, masks, totals3):
    lib().orc_trianglecull(_p(globals_), _p(commands), _p(draws), _p(meshlets), _p(meshlet_data), _p(vertices), _p(cib), _p(cc4), _p(masks),
                           C.c_uint32(len(masks)), _p(totals3))


def meshlet_bounds(vertices, meshlet_data, meshlets, want_float=False):
    """fills center / radius / cone_axis / cone_cutoff of `meshlets` in place (src/scene.cpp:69-85; parity unpinned: oracle.c);
    optionally returns the unquantised {center, radius, axis, cutoff} rows"""
    out = np.zeros((len(meshlets), 8), np.float32) if want_float else None
    lib().orc_meshlet_bounds(_p(vertices), _p(meshlet_data), _p(meshlets), C.c_uint32(len(meshlets)), _p(out))
    return out


def max_threads():
    return int(lib().orc_max_threads())
