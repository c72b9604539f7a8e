"""Test-side geometry for SURVEY.md §8(f) N2: procedural meshes with coherent triangles, cut into meshlets in niagara's
packed form (src/scene.cpp:24-47: vertex references — u16 pairs when shortRefs — followed by 3 index bytes per triangle).
meshoptimizer's clusterizer is not vendored; a greedy cut in triangle order (<= 64 vertices, <= 96 triangles) stands in
for meshopt_buildMeshlets — the bounds code under test does not care how a meshlet was formed.  TEST INFRASTRUCTURE."""
import numpy as np

from niagara_amd import layouts as L


def torus(nu=96, nv=48, R=1.0, r=0.35, bump=0.05, seed=0):
    """bumpy torus: (positions float32 [n,3], triangles int [m,3]); a closed surface, so every view direction has
    back-facing meshlets"""
    rng = np.random.default_rng(seed)
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    rr = r + bump * np.sin(5 * uu) * np.cos(3 * vv) + 0.01 * rng.standard_normal(uu.shape)
    x = (R + rr * np.cos(vv)) * np.cos(uu)
    y = (R + rr * np.cos(vv)) * np.sin(uu)
    z = rr * np.sin(vv)
    pos = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    idx = lambda i, j: (i % nu) * nv + (j % nv)
    # triangles ordered in patches of 6 x 8 quads (63 vertices, 96 triangles), so that a cut in order gives compact
    # meshlets like a real clusterizer's rather than full rings of the tube
    tris = []
    for bi in range(0, nu, 6):
        for bj in range(0, nv, 8):
            for i in range(bi, min(bi + 6, nu)):
                for j in range(bj, min(bj + 8, nv)):
                    a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
                    tris += [(a, b, c), (a, c, d)]
    return pos, np.array(tris, np.int64)


def build_meshlets(positions, triangles, max_vertices=64, max_triangles=96, base_vertex=0, force_long_refs=False):
    """greedy cut in triangle order -> (meshlets L.MESHLET [k] with dataOffset / baseVertex / counts / shortRefs set,
    meshlet_data uint32 [], vertices L.VERTEX [n] with fp16 positions)"""
    vertices = np.zeros(len(positions), dtype=L.VERTEX)
    h = positions.astype(np.float16).view(np.uint16)
    vertices["vx"], vertices["vy"], vertices["vz"] = h[:, 0], h[:, 1], h[:, 2]
    groups, cur_v, cur_t = [], {}, []
    for t in triangles:
        new = [int(x) for x in t if int(x) not in cur_v]
        if len(cur_v) + len(set(new)) > max_vertices or len(cur_t) >= max_triangles:
            groups.append((cur_v, cur_t))
            cur_v, cur_t = {}, []
        for x in t:
            cur_v.setdefault(int(x), len(cur_v))
        cur_t.append([cur_v[int(x)] for x in t])
    if cur_t:
        groups.append((cur_v, cur_t))
    meshlets = np.zeros(len(groups), dtype=L.MESHLET)
    words = []
    for k, (vmap, tl) in enumerate(groups):
        refs = np.array(sorted(vmap, key=vmap.get), np.uint32)
        lo = int(refs.min())
        short = (int(refs.max()) - lo < (1 << 16)) and not force_long_refs
        meshlets[k]["dataOffset"] = len(words)
        meshlets[k]["baseVertex"] = base_vertex + lo
        meshlets[k]["vertexCount"], meshlets[k]["triangleCount"], meshlets[k]["shortRefs"] = len(refs), len(tl), int(short)
        rel = refs - lo
        if short:
            padded = np.concatenate([rel, np.zeros(len(rel) % 2, np.uint32)]).astype(np.uint16)
            words += padded.view(np.uint32).tolist()
        else:
            words += rel.tolist()
        idx = np.array(tl, np.uint8).reshape(-1)
        idx = np.concatenate([idx, np.zeros((-len(idx)) % 4, np.uint8)])
        words += idx.view(np.uint32).tolist()
    return meshlets, np.array(words + [0, 0, 0, 0], np.uint32), vertices


def meshlet_triangles(meshlets, data, vertices, k):
    """float64 corner positions [t,3,3] of meshlet k, decoded like the mesh shader does (meshlet.mesh.glsl:107-127)"""
    m = meshlets[k]
    vc, tc, off = int(m["vertexCount"]), int(m["triangleCount"]), int(m["dataOffset"])
    if m["shortRefs"] == 1:
        refs = data.view(np.uint16)[off * 2:off * 2 + vc].astype(np.int64)
        ioff = off + (vc + 1) // 2
    else:
        refs = data[off:off + vc].astype(np.int64)
        ioff = off + vc
    vi = refs + int(m["baseVertex"])
    pos = np.stack([vertices["vx"][vi], vertices["vy"][vi], vertices["vz"][vi]], -1).view(np.float16).astype(np.float64)
    idx = data.view(np.uint8)[ioff * 4:ioff * 4 + tc * 3].reshape(tc, 3).astype(np.int64)
    return pos, pos[idx]
