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


def task_fixture():
    """tests/golden/mesh/taskcull.npz: task commands of a frame scene + what the reference TASK shader (meshlet.task.glsl, TASK_CULL = 1)
    emits for them in the early and in the late pass: EmitMeshTasksEXT counts, payload entries, visibility words"""
    scene = make_scene(seed=105, n_draws=900, meshlets_lod0=110, zero_radius_fraction=0.02)
    from oracle import Pyramid
    pyr = Pyramid(*scene["viewport"])
    R.depthreduce(scene["depth"], pyr)
    cd = passes.set_flags(scene["cull"], (1, 1, 1, 1, 1))
    cmds, c4 = passes.run_drawcull(R, scene, cd, 0, 1, np.ones(len(scene["draws"]), np.uint32), pyr)
    R.tasksubmit(c4, cmds)
    rng = np.random.default_rng(9)
    cmds["lateDrawVisibility"][:int(c4[0])] = rng.integers(0, 2, int(c4[0]))
    ncmd = int(c4[1]) * 64
    mvb0 = rng.integers(0, 2 ** 32, (scene["slots"] + 31) // 32 + 2, dtype=np.uint64).astype(np.uint32)
    out = dict(meshlets=scene["meshlets"], draws=scene["draws"], cull=cd, depth=scene["depth"], viewport=np.array(scene["viewport"], np.uint32),
               commands=cmds[:ncmd], count4=c4, mvb0=mvb0)
    for late in (0, 1):
        pay, cnt, mvb = np.zeros((ncmd, 64), np.uint32), np.zeros(ncmd, np.uint32), mvb0.copy()
        R.meshlet_task(cd, late, cmds, c4, scene["draws"], scene["meshlets"], mvb, pyr, pay, cnt)
        pay[np.arange(64)[None, :] >= cnt[:, None]] = 0  # entries past the emitted count are not part of the result
        out["late%d_payloads" % late], out["late%d_counts" % late], out["late%d_mvb" % late] = pay, cnt, mvb
    path = os.path.join(HERE, "mesh", "taskcull.npz")
    np.savez_compressed(path, **out)
    print("%-28s %6d bytes, %d commands, emitted early %d late %d" % ("mesh/taskcull.npz", os.path.getsize(path), ncmd, int(out["late0_counts"].sum()),
                                                                     int(out["late1_counts"].sum())))


if __name__ == "__main__":
    main()
    mesh_fixture()
    task_fixture()
