#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE'S OWN shader sources executing on the CPU (oracle/_ref, built by
`make -C oracle ref` from /root/reference/src/shaders/*.glsl).  Run in the build container, where the reference tree is
mounted; the fixtures travel with the repository, the reference does not.

    python tests/golden/generate.py

Each fixture stores the inputs (meshes, meshlets, draws, CullData, depth target, flags) and every buffer the
reference's passes leave behind over two frames of the early -> pyramid -> late protocol (src/niagara.cpp:1765-1788).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle.ref as R  # noqa: E402
import passes  # noqa: E402
from scenes import make_scene  # noqa: E402

CASES = [
    # name, scene kwargs, flags (cullingEnabled, lodEnabled, occlusionEnabled, clusterOcclusionEnabled, clusterBackfaceEnabled)
    ("all_on", dict(seed=101, n_draws=300, meshlets_lod0=130, zero_radius_fraction=0.02), (1, 1, 1, 1, 1)),
    ("no_cluster_occlusion", dict(seed=102, n_draws=300, meshlets_lod0=100), (1, 1, 1, 0, 1)),
    ("frustum_only", dict(seed=103, n_draws=400, meshlets_lod0=70, lods=1), (1, 0, 0, 0, 0)),
    ("culling_off_backface_off", dict(seed=104, n_draws=120, meshlets_lod0=90, random_camera=False), (0, 1, 1, 1, 0)),
]


def main():
    assert R.available(), "oracle/_ref is not built: run `make -C oracle ref` where /root/reference exists"
    for name, kw, flags in CASES:
        scene = make_scene(**kw)
        frames = passes.run_frames(R, scene, flags, frames=2)
        out = dict(meshes=scene["meshes"], meshlets=scene["meshlets"], draws=scene["draws"], cull=scene["cull"], depth=scene["depth"],
                   flags=np.array(flags, np.int32), slots=np.array([scene["slots"]], np.uint32), viewport=np.array(scene["viewport"], np.uint32))
        for f, rec in enumerate(frames):
            out["f%d_pyramid" % f] = rec["pyramid"]
            for phase in ("early", "late"):
                for key, val in rec[phase].items():
                    out["f%d_%s_%s" % (f, phase, key)] = val
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        vis = int(frames[0]["late"]["cc4"][0])
        print("%-28s %6d bytes, frame-0 late visible clusters: %d" % (name + ".npz", os.path.getsize(path), vis))


def mesh_fixture():
    """tests/golden/mesh/trianglecull.npz: inputs + the reference mesh shader's own triangle-cull decisions (MESH_CULL = 1)"""
    import test_trianglecull as T
    from scenes import make_triangle_scene
    s = make_triangle_scene(seed=77, n_draws=20, commands_per_draw=1, scene_radius=8.0, cam_pos=(1.0, 0.5, -2.0))
    cib, cc4 = T.cluster_list(R, s)
    masks, totals = T.run(R.meshlet_mesh, s, cib, cc4)
    os.makedirs(os.path.join(HERE, "mesh"), exist_ok=True)
    path = os.path.join(HERE, "mesh", "trianglecull.npz")
    np.savez_compressed(path, globals=s["globals"], commands=s["commands"], draws=s["draws"], meshlets=s["meshlets"], data=s["data"],
                        vertices=s["vertices"], cib=cib[:len(masks)], cc4=cc4, masks=masks, totals=totals)
    print("%-28s %6d bytes, clusters %d, triangles %d, kept %d" % ("mesh/trianglecull.npz", os.path.getsize(path), *totals))


if __name__ == "__main__":
    main()
    mesh_fixture()
