"""Small seeded scenes for the parity tests (test infrastructure)."""
import numpy as np

from niagara_amd import host, synth
from niagara_amd import layouts as L


def random_quat(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return q.astype(np.float32)


def make_scene(seed=0, n_draws=300, n_meshes=3, lods=4, meshlets_lod0=150, scene_radius=25.0, viewport=(256, 192),
               random_camera=True, post_pass_fraction=0.0, zero_radius_fraction=0.0):
    """draws scattered in a cube around a camera; enough of them visible to exercise every branch"""
    rng = np.random.default_rng(seed)
    meshes, total = synth.make_meshes(n_meshes, lods, meshlets_lod0, seed=seed + 11)
    meshlets = synth.make_meshlets(total, seed=seed + 12)
    if zero_radius_fraction:
        z = rng.random(total) < zero_radius_fraction
        meshlets["radius"][z] = 0
    draws = host.synth_draws(n_draws, n_meshes, scene_radius)
    if post_pass_fraction:
        draws["postPass"] = (rng.random(n_draws) < post_pass_fraction).astype(np.uint32)
    slots, mask = host.assign_visibility_offsets(draws, meshes)
    cam_pos = rng.uniform(-3, 3, 3).astype(np.float32) if random_camera else np.zeros(3, np.float32)
    cam_q = random_quat(rng) if random_camera else np.array([0, 0, 0, 1], np.float32)
    pw, ph = host.previous_pow2(viewport[0]), host.previous_pow2(viewport[1])
    cd = host.build_cull_data(cam_pos, cam_q, draw_distance=60.0, viewport=viewport, pyramid=(pw, ph), draw_count=n_draws,
                              cullingEnabled=1, lodEnabled=1)
    # occluders that matter at this scale: big rectangles at z in [3, 30] (reverse-Z: depth = znear / z, far = 0)
    depth = np.zeros((viewport[1], viewport[0]), np.float32)
    for _ in range(10):
        w = int(rng.integers(viewport[0] // 6, viewport[0] // 2))
        h = int(rng.integers(viewport[1] // 6, viewport[1] // 2))
        x, y = int(rng.integers(0, viewport[0] - w)), int(rng.integers(0, viewport[1] - h))
        depth[y:y + h, x:x + w] = np.maximum(depth[y:y + h, x:x + w], np.float32(0.1) / np.float32(rng.uniform(3, 30)))
    return dict(meshes=meshes, meshlets=meshlets, draws=draws, slots=slots, post_mask=mask, cull=cd, depth=depth, viewport=viewport)


def task_capacity(scene):
    """upper bound on task commands any pass can emit for this scene (+64 for the tasksubmit padding)"""
    meshes, draws = scene["meshes"], scene["draws"]
    per_mesh = np.array([max((int(m["lods"][l]["meshletCount"]) + 63) // 64 for l in range(int(m["lodCount"]))) for m in meshes])
    return int(per_mesh[draws["meshIndex"]].sum()) + 64


def flag_matrix():
    """(cullingEnabled, lodEnabled, occlusionEnabled, clusterOcclusionEnabled, clusterBackfaceEnabled)"""
    out = []
    for ce in (0, 1):
        for le in (0, 1):
            for oe in (0, 1):
                for coe in (0, 1):
                    for cbe in (0, 1):
                        out.append((ce, le, oe, coe, cbe))
    return out


def make_triangle_scene(seed=0, n_draws=60, commands_per_draw=3, viewport=(640, 480), scene_radius=20.0, cam_pos=(0, 0, 0), cam_quat=(0, 0, 0, 1),
                        full_meshlets=False, specials=False):
    """Inputs of the mesh stage's triangle cull: a config-3 style cluster scene plus synthetic meshlet
    payloads (vertex references, index bytes, fp16 vertices).  The cluster list is NOT part of the scene: the caller
    produces it with the implementation under test (clustercull -> clustersubmit)."""
    from niagara_amd import host, synth
    draws, meshlets, commands, n = synth.cluster_scene(n_draws, commands_per_draw, seed=seed, scene_radius=scene_radius)
    data, vertices = synth.make_geometry(meshlets, seed=seed + 1000)
    if specials:  # every fp16 class in the vertex stream (NaN, inf, denormals, -0) and non-finite / zero / negative draw fields
        rng = np.random.default_rng(seed + 5)
        raw = rng.random(len(vertices)) < 0.2
        for f in ("vx", "vy", "vz"):
            vertices[f][raw] = rng.integers(0, 1 << 16, int(raw.sum())).astype(np.uint16)
        sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-42, 3.4e38, -1.0], np.float32)
        for i in np.nonzero(rng.random(n_draws) < 0.3)[0]:
            k = rng.integers(0, 3)
            if k == 0:
                draws["position"][i][rng.integers(0, 3)] = rng.choice(sp)
            elif k == 1:
                draws["scale"][i] = rng.choice(sp)
            else:
                draws["orientation"][i][rng.integers(0, 4)] = rng.choice(sp)
    cd = host.build_cull_data(cam_pos=cam_pos, cam_quat=cam_quat, draw_count=n_draws, viewport=viewport, cullingEnabled=1, clusterBackfaceEnabled=0)
    return dict(draws=draws, meshlets=meshlets, commands=commands, n=n, cull=cd, data=data, vertices=vertices,
                globals=synth.make_globals(cd, viewport), count4=synth.count4_for(n), viewport=viewport)


def random_case(seed):
    """(make_scene kwargs, flags, use_soa, fused) drawn from a wide range: tiny and ragged draw counts, 1-8 meshes and LODs,
    up to six task groups per draw, odd viewports down to 2 pixels, post-pass draws, zero-radius meshlets"""
    rng = np.random.default_rng(seed)
    kw = dict(seed=seed, n_draws=int(rng.integers(1, 3000)), n_meshes=int(rng.integers(1, 9)), lods=int(rng.integers(1, 9)),
              meshlets_lod0=int(rng.integers(1, 400)), scene_radius=float(rng.uniform(5, 60)),
              viewport=(int(rng.integers(2, 700)), int(rng.integers(2, 500))), post_pass_fraction=float(rng.choice([0.0, 0.0, 0.3])),
              zero_radius_fraction=float(rng.choice([0.0, 0.05])))
    flags = tuple(int(x) for x in rng.integers(0, 2, 5))
    return kw, flags, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
