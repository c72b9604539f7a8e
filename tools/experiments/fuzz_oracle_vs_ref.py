#!/usr/bin/env python3
"""Randomised soak on the CPU: the oracle against the REFERENCE'S OWN shaders executing on the CPU (oracle/_ref), over the
same random cases the GPU soaks draw (scenes.random_case): three frames of the early / pyramid / late protocol with every
buffer, drawcull over LATE x TASK with random history and postPass, the task shader's payloads, pyramids of random sizes,
the mesh shader's triangle cull.  Needs the reference tree (or a prebuilt oracle/_ref/libniagara_ref.so); no GPU.

    python tools/experiments/fuzz_oracle_vs_ref.py [seconds=120] [first_seed=1000]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
import oracle.ref as R  # noqa: E402
import passes  # noqa: E402
from niagara_amd import layouts as L  # noqa: E402
from scenes import make_scene, make_triangle_scene, random_case  # noqa: E402
from test_trianglecull import cluster_list, run as run_triangles  # noqa: E402

assert R.available(), "oracle/_ref is not built"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t0 = time.time()
counts = dict(frames=0, drawcull=0, taskcull=0, pyramid=0, triangles=0)
bad = []
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    kw, flags, _, _ = random_case(seed)
    kw["n_draws"] = min(kw["n_draws"], 800)  # the reference shaders run one invocation at a time
    scene = make_scene(**kw)
    # ---- frames
    fo = passes.run_frames(oracle, scene, flags, frames=3)
    fr = passes.run_frames(R, scene, flags, frames=3)
    for f, (a, b) in enumerate(zip(fo, fr)):
        if a["pyramid"].tobytes() != b["pyramid"].tobytes():
            bad.append(("frames pyramid", seed, f))
        for phase in ("early", "late"):
            for key in ("count4", "commands", "cc4", "cib", "dvb", "mvb"):
                if a[phase][key].tobytes() != b[phase][key].tobytes():
                    bad.append(("frames", seed, f, phase, key, kw, flags))
    counts["frames"] += 1
    # ---- drawcull matrix + task shader
    pyr = oracle.Pyramid(*scene["viewport"])
    oracle.depthreduce(scene["depth"], pyr)
    cd = passes.set_flags(scene["cull"], flags)
    for late in (0, 1):
        for task in (0, 1):
            post = int(rng.integers(0, 2))
            dvb0 = (rng.random(len(scene["draws"])) < rng.random()).astype(np.uint32)
            outs = []
            for impl in (oracle, R):
                dvb = dvb0.copy()
                co, c4 = passes.run_drawcull(impl, scene, cd, late, task, dvb, pyr, post)
                outs.append((co[:int(c4[0])].tobytes(), c4.tobytes(), dvb.tobytes()))
            if outs[0] != outs[1]:
                bad.append(("drawcull", seed, late, task, post, kw, flags))
            counts["drawcull"] += 1
    cmds, c4 = passes.run_drawcull(oracle, scene, cd, 0, 1, np.ones(len(scene["draws"]), np.uint32), pyr)
    oracle.tasksubmit(c4, cmds)
    cmds["lateDrawVisibility"][:int(c4[0])] = rng.integers(0, 2, int(c4[0]))
    ncmd = int(c4[1]) * 64
    mvb0 = rng.integers(0, 2 ** 32, (scene["slots"] + 31) // 32 + 2, dtype=np.uint64).astype(np.uint32)
    for late in (0, 1):
        c = cd.copy()
        c["postPass"] = int(rng.integers(0, 2))
        outs = []
        for fn in (oracle.taskcull, R.meshlet_task):
            pay, cnt, mvb = np.zeros((ncmd, 64), np.uint32), np.zeros(ncmd, np.uint32), mvb0.copy()
            fn(c, late, cmds, c4, scene["draws"], scene["meshlets"], mvb, pyr, pay, cnt)
            pay[np.arange(64)[None, :] >= cnt[:, None]] = 0
            outs.append((pay.tobytes(), cnt.tobytes(), mvb.tobytes()))
        if outs[0] != outs[1]:
            bad.append(("taskcull", seed, late, kw, flags))
        counts["taskcull"] += 1
    # ---- pyramid of a random size
    w, h = int(rng.integers(1, 400)), int(rng.integers(1, 300))
    depth = rng.random((h, w), dtype=np.float32)
    po, pr = oracle.Pyramid(w, h), oracle.Pyramid(w, h)
    oracle.depthreduce(depth, po)
    R.depthreduce(depth, pr)
    if po.data.tobytes() != pr.data.tobytes():
        bad.append(("pyramid", seed, w, h))
    counts["pyramid"] += 1
    # ---- triangle cull
    s = make_triangle_scene(seed=seed, n_draws=int(rng.integers(1, 60)), commands_per_draw=int(rng.integers(1, 4)),
                            viewport=(int(rng.integers(16, 2000)), int(rng.integers(16, 1200))), scene_radius=float(rng.uniform(2, 40)),
                            cam_pos=tuple(float(x) for x in rng.uniform(-5, 5, 3)), specials=bool(rng.random() < 0.3))
    cib, cc4 = cluster_list(oracle, s, backface=int(rng.integers(0, 2)))
    mo, to = run_triangles(oracle.trianglecull, s, cib, cc4)
    mr, tr = run_triangles(R.meshlet_mesh, s, cib, cc4)
    if to.tolist() != tr.tolist() or mo.tobytes() != mr.tobytes():
        bad.append(("triangles", seed))
    counts["triangles"] += 1
    seed += 1
for b in bad[:20]:
    print("MISMATCH", b)
print("fuzz_oracle_vs_ref:", counts, "mismatches:", len(bad), "in %.0f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
