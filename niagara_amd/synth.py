"""Synthetic scenes of the BASELINE.json configs (numpy, host side).

The reference generates its stress scene in main() (src/niagara.cpp:969-998); meshlet bounds come from meshoptimizer
there (src/scene.cpp:69-85), which is not vendored, so meshlet pools are drawn from a seeded generator with the
distributions SURVEY.md §8(d) fixes.
"""
import numpy as np

from . import host
from . import layouts as L


def make_meshlets(count, seed=2):
    """centre ~U[-1,1]^3 and radius ~U[0.02,0.1] as fp16; cone axis = random unit vector -> round(127 x) s8;
    cutoff ~U{0..127}"""
    rng = np.random.default_rng(seed)
    m = np.zeros(count, dtype=L.MESHLET)
    m["center"] = rng.uniform(-1, 1, (count, 3)).astype(np.float16).view(np.uint16)
    m["radius"] = rng.uniform(0.02, 0.1, count).astype(np.float16).view(np.uint16)
    axis = rng.normal(size=(count, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    m["cone_axis"] = np.rint(axis * 127).astype(np.int8)
    m["cone_cutoff"] = rng.integers(0, 128, count).astype(np.int8)
    m["vertexCount"] = 64
    m["triangleCount"] = 96
    return m


def make_meshes(mesh_count, lod_count, meshlets_lod0, center=(0.0, 0.0, 0.0), radius=1.8, seed=3):
    """meshes with `lod_count` LODs; LOD i has ceil(meshlets_lod0 / 2^i) meshlets, error 0.002*2^i (error 0 for LOD 0),
    indexCount 86832>>i.  Returns (meshes, total_meshlets); meshlet ranges are packed back to back."""
    rng = np.random.default_rng(seed)
    meshes = np.zeros(mesh_count, dtype=L.MESH)
    offset = 0
    index_offset = 0
    for i in range(mesh_count):
        meshes[i]["center"] = np.asarray(center, np.float32) + rng.uniform(-0.05, 0.05, 3).astype(np.float32) * (mesh_count > 1)
        meshes[i]["radius"] = radius
        meshes[i]["vertexOffset"] = i * 1000
        meshes[i]["vertexCount"] = 1000
        meshes[i]["lodCount"] = lod_count
        for l in range(lod_count):
            mc = max(1, -(-meshlets_lod0 // (1 << l))) if meshlets_lod0 else 0
            lod = meshes[i]["lods"][l]
            lod["indexOffset"] = index_offset
            lod["indexCount"] = 86832 >> l
            lod["meshletOffset"] = offset
            lod["meshletCount"] = mc
            lod["error"] = 0.0 if l == 0 else 0.002 * (1 << l)
            offset += mc
            index_offset += 86832 >> l
    return meshes, offset


def make_task_commands(draw_count, commands_per_draw, late_draw_visibility=None, meshlet_base=0):
    """config 3A: full commands (taskCount 64), command k covers meshlets [64k, 64k+64) and visibility slots alike.
    The array is padded with zeroed dummy commands to a multiple of 64, exactly what tasksubmit leaves behind
    (tasksubmit.comp.glsl:40-46); use count4_for() for the matching {count, X, 64, 1} words."""
    n = draw_count * commands_per_draw
    c = np.zeros((n + 63) // 64 * 64, dtype=L.TASKCMD)
    k = np.arange(n, dtype=np.uint32)
    c["drawId"][:n] = k // commands_per_draw
    c["taskOffset"][:n] = k * 64 + meshlet_base
    c["taskCount"][:n] = 64
    c["meshletVisibilityOffset"][:n] = k * 64
    if late_draw_visibility is not None:
        c["lateDrawVisibility"][:n] = late_draw_visibility[c["drawId"][:n]]
    return c


def count4_for(command_count):
    """the dccb words tasksubmit writes for `command_count` commands (tasksubmit.comp.glsl:30-38)"""
    count = min(command_count, L.TASK_WGLIMIT)
    return np.array([command_count, min((count + 63) // 64, 65535), 64, 1], np.uint32)


def make_depth(width, height, znear=0.1, rects=64, seed=4):
    """reverse-Z depth target: background 0 (far), `rects` axis-aligned rectangles with depth = znear / z, z~U[5,100]"""
    rng = np.random.default_rng(seed)
    depth = np.zeros((height, width), dtype=np.float32)
    for _ in range(rects):
        w = int(rng.integers(max(2, width // 64), max(3, width // 4)))
        h = int(rng.integers(max(2, height // 64), max(3, height // 4)))
        x = int(rng.integers(0, max(1, width - w)))
        y = int(rng.integers(0, max(1, height - h)))
        z = np.float32(rng.uniform(5, 100))
        depth[y:y + h, x:x + w] = np.maximum(depth[y:y + h, x:x + w], np.float32(znear) / z)
    return depth


def cluster_scene(draw_count, commands_per_draw=10, seed=2, scene_radius=300.0):
    """config 3A / 5 inputs: draws (niagara generator), meshlet pool, padded task commands, real command count"""
    draws = host.synth_draws(draw_count, 1, scene_radius)
    n_cmd = draw_count * commands_per_draw
    meshlets = make_meshlets(n_cmd * 64, seed)
    draws["meshletVisibilityOffset"] = np.arange(draw_count, dtype=np.uint32) * (commands_per_draw * 64)
    commands = make_task_commands(draw_count, commands_per_draw)
    return draws, meshlets, commands, n_cmd


def make_globals(cd, viewport):
    """the mesh pipeline's push constants (src/shaders/mesh.h:46-51): niagara's reverse-Z infinite projection
    (perspectiveProjection, src/niagara.cpp:424-431) rebuilt from the P00 / P11 / znear the CullData already carries"""
    g = np.zeros(1, dtype=L.GLOBALS)
    p = np.zeros(16, np.float32)
    p[0], p[5], p[11], p[14] = cd["P00"][0], cd["P11"][0], 1.0, cd["znear"][0]
    g["projection"][0] = p
    g["cullData"][0] = cd[0]
    g["screenWidth"], g["screenHeight"] = viewport
    return g


def make_geometry(meshlets, seed=5, vertices_per_mesh=4096):
    """Synthetic meshlet payloads in niagara's packed form (src/scene.cpp:24-115, src/shaders/meshlet.mesh.glsl:107-127):
    per meshlet `dataOffset` words = vertex references (u16 pairs when shortRefs, else u32) followed by 3 index bytes per
    triangle; vertices = fp16 positions scattered around the meshlet's centre with about its radius.  Fills vertexCount
    (<= 64), triangleCount (<= 96), dataOffset, baseVertex, shortRefs of `meshlets` in place; returns (meshlet_data u32[],
    vertices).  meshoptimizer builds the real thing and is not vendored: distributions only, like the bounds."""
    rng = np.random.default_rng(seed)
    n = len(meshlets)
    vc = rng.integers(3, 65, n).astype(np.uint32)
    tc = np.minimum(rng.integers(1, 97, n), 96).astype(np.uint32)
    short = rng.integers(0, 2, n).astype(np.uint32)
    ref_words = np.where(short == 1, (vc + 1) // 2, vc)
    idx_words = (tc * 3 + 3) // 4
    words = ref_words + idx_words
    offsets = np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.uint32)
    data = np.zeros(int(words.sum()) + 4, np.uint32)
    d16, d8 = data.view(np.uint16), data.view(np.uint8)
    base = (np.arange(n, dtype=np.uint32) * 61) % max(1, vertices_per_mesh - 64 * 4)  # overlapping windows of a shared pool per "mesh"
    pool = (np.arange(n, dtype=np.uint32) // 4096) * vertices_per_mesh
    base = base + pool
    total_vertices = int(pool.max()) + vertices_per_mesh if n else vertices_per_mesh
    vertices = np.zeros(total_vertices, dtype=L.VERTEX)
    centers = meshlets["center"].view(np.float16).astype(np.float32).reshape(n, 3)
    radii = meshlets["radius"].view(np.float16).astype(np.float32)
    # positions: every vertex of the pool gets a position near the centre of the first meshlet whose window covers it
    vpos = rng.normal(size=(total_vertices, 3)).astype(np.float32)
    owner = np.minimum(np.searchsorted(base, np.arange(total_vertices, dtype=np.uint32), side="right").clip(1) - 1, n - 1) if n else np.zeros(total_vertices, int)
    vpos = centers[owner] + vpos * (radii[owner][:, None] * 0.6)
    vertices["vx"], vertices["vy"], vertices["vz"] = (vpos[:, k].astype(np.float16).view(np.uint16) for k in range(3))
    vertices["np"] = rng.integers(0, 2 ** 32, total_vertices, dtype=np.uint64).astype(np.uint32)
    vertices["tp"] = rng.integers(0, 2 ** 16, total_vertices).astype(np.uint16)
    for i in range(n):
        o, v, t = int(offsets[i]), int(vc[i]), int(tc[i])
        refs = rng.integers(0, 192, v).astype(np.uint32)  # window of 192 pool vertices behind baseVertex
        if short[i]:
            d16[o * 2:o * 2 + v] = refs
        else:
            data[o:o + v] = refs
        io = (o + int(ref_words[i])) * 4
        d8[io:io + t * 3] = rng.integers(0, v, t * 3).astype(np.uint8)
    meshlets["vertexCount"] = vc
    meshlets["triangleCount"] = tc
    meshlets["dataOffset"] = offsets
    meshlets["baseVertex"] = base
    meshlets["shortRefs"] = short
    return data, vertices
