"""Byte layouts of the buffers crossing the boundary, as numpy structured dtypes.

Same bytes as the reference structs (src/scene.h:10-93, src/niagara.cpp:227-260, src/shaders/mesh.h:11-123) and
as include/niagara_vis.h; tests cross-check all three.
"""
import numpy as np

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

CLUSTERRECORD = np.dtype([("drawId", "<u4"), ("meshletIndex", "<u4"), ("vertexCount", "<u4"), ("triangleCount", "<u4"), ("vertexOffset", "<u4"),
                          ("indexOffset", "<u4"), ("baseVertex", "<u4"), ("shortRefs", "<u4")])
VERTEX = np.dtype([("vx", "<u2"), ("vy", "<u2"), ("vz", "<u2"), ("tp", "<u2"), ("np", "<u4"), ("tu", "<u2"), ("tv", "<u2")])  # src/shaders/mesh.h:3-9
GLOBALS = np.dtype([("projection", "<f4", 16), ("cullData", CULLDATA), ("screenWidth", "<f4"), ("screenHeight", "<f4"), ("_pad", "<f4", 2)])  # mesh.h:46-51
TRIMASK = np.dtype([("keep", "<u4", 3), ("counts", "<u4")])
assert (VERTEX.itemsize, GLOBALS.itemsize, TRIMASK.itemsize) == (16, 224, 16)

TASK_WGSIZE = 64
TASK_WGLIMIT = 1 << 22
CLUSTER_LIMIT = 1 << 24
CLUSTER_TILE = 16

assert (MESHLET.itemsize, MESHDRAW.itemsize, MESHLOD.itemsize, MESH.itemsize) == (24, 48, 20, 208)
assert (DRAWCMD.itemsize, TASKCMD.itemsize, CULLDATA.itemsize) == (24, 20, 144)
