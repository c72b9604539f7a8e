#!/usr/bin/env python3
"""Randomised soak of the single passes the frame soak (fuzz_frames.py) does not reach: drawcull in all four LATE x TASK
specialisations with random flags / postPass / visibility history, depth pyramids of random (odd, tiny, wide) sizes, and
the triangle cull over random cluster scenes and cameras.  Every output compared with the oracle.  Needs a GPU.

    python tools/experiments/fuzz_passes.py [seconds=90] [first_seed=5000]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402  (checker)
import passes  # noqa: E402
import gpu_passes as G  # noqa: E402
from niagara_amd import layouts as L  # noqa: E402
from niagara_amd import pipeline as P  # noqa: E402
from scenes import make_scene, make_triangle_scene, random_case  # noqa: E402
from test_trianglecull import cluster_list, run as run_triangles  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
ctx = P.Context(0)
dev = ctx.device
t0 = time.time()
counts = {"drawcull": 0, "pyramid": 0, "triangles": 0}
bad = []
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    # ---- drawcull
    kw, flags, use_soa, _ = random_case(seed)
    scene = make_scene(**kw)
    g = G.GpuScene(ctx, scene, use_soa)
    pyr = oracle.Pyramid(*scene["viewport"])
    oracle.depthreduce(scene["depth"], pyr)
    if g.depthreduce(scene["depth"]).tobytes() != pyr.data.tobytes():
        bad.append(("scene pyramid", seed, kw))
    def drawcull_passes(scene, g, pyr, flags, what):
        for late in (0, 1):
            for task in (0, 1):
                post = int(rng.integers(0, 2))
                # the early pass's request order and the task pass's emission form (round 6: the list form reads the decide launch's records)
                opts = (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
                ctx.set_option(P.NV_OPT_DRAW_RECORDS, opts[0])
                ctx.set_option(P.NV_OPT_TASK_EMIT, opts[1])
                cd = passes.set_flags(scene["cull"], flags)
                dvb0 = (rng.random(len(scene["draws"])) < rng.random() ** 2).astype(np.uint32)
                dvb_o = dvb0.copy()
                co, c4o = passes.run_drawcull(oracle, scene, cd, late, task, dvb_o, pyr, post)
                dcb, dccb, dvb = g.drawcull(cd, late, task, dvb0, post)
                n = int(c4o[0])
                dt = L.TASKCMD if task else L.DRAWCMD
                if G.host_u32(dccb)[0] != n or P.from_device(dcb, dt)[:n].tobytes() != co[:n].tobytes() or not (G.host_u32(dvb) == dvb_o).all():
                    bad.append(("drawcull", seed, what, flags, late, task, post, opts))
                counts["drawcull"] += 1
        ctx.set_option(P.NV_OPT_DRAW_RECORDS, 0)
        ctx.set_option(P.NV_OPT_TASK_EMIT, 0)

    drawcull_passes(scene, g, pyr, flags, kw)
    # ---- and over enough draws that a wave of the decide launch walks several 64-draw units (more than 131 072)
    big = dict(seed=seed, n_draws=int(rng.integers(131_073, 600_000)), n_meshes=int(rng.integers(1, 9)), lods=int(rng.integers(1, 9)), meshlets_lod0=int(rng.integers(1, 400)),
               scene_radius=float(rng.uniform(5, 60)), post_pass_fraction=float(rng.choice([0.0, 0.1])))
    scene = make_scene(**big)
    g = G.GpuScene(ctx, scene, bool(rng.integers(0, 2)))
    pyr = oracle.Pyramid(*scene["viewport"])
    oracle.depthreduce(scene["depth"], pyr)
    g.depthreduce(scene["depth"])
    drawcull_passes(scene, g, pyr, tuple(int(x) for x in rng.integers(0, 2, 5)), big)
    del g, scene
    # ---- pyramid of a random size
    w, h = (int(rng.integers(1, 1200)), int(rng.integers(1, 900))) if rng.random() < 0.8 else (int(2 ** rng.integers(0, 12)), int(2 ** rng.integers(0, 12)))
    depth = rng.random((h, w), dtype=np.float32)
    po = oracle.Pyramid(w, h)
    oracle.depthreduce(depth, po)
    pg = P.DepthPyramid(dev, w, h)
    ctx.depthreduce(torch.from_numpy(depth).to(dev), w, h, pg.desc)
    if pg.data.cpu().numpy().tobytes() != po.data.tobytes():
        bad.append(("pyramid", seed, w, h))
    counts["pyramid"] += 1
    # ---- triangle cull
    s = make_triangle_scene(seed=seed, n_draws=int(rng.integers(1, 300)), commands_per_draw=int(rng.integers(1, 7)),
                            viewport=(int(rng.integers(16, 2000)), int(rng.integers(16, 1200))), scene_radius=float(rng.uniform(2, 40)),
                            cam_pos=tuple(float(x) for x in rng.uniform(-5, 5, 3)), specials=bool(rng.random() < 0.3))
    cib, cc4 = cluster_list(oracle, s, backface=int(rng.integers(0, 2)))
    mo, to = run_triangles(oracle.trianglecull, s, cib, cc4)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    masks = torch.zeros(max(1, len(mo)) * 16, dtype=torch.uint8, device=dev)
    totals = torch.zeros(3, dtype=torch.int64, device=dev)
    ctx.trianglecull(s["globals"], t(s["commands"]), t(s["draws"]), t(s["meshlets"]), t(s["data"]), t(s["vertices"]), t(cib), t(cc4), masks, len(mo), totals)
    if totals.cpu().numpy().astype(np.uint64).tolist() != to.tolist() or masks.cpu().numpy()[:len(mo) * 16].tobytes() != mo.tobytes():
        bad.append(("triangles", seed))
    counts["triangles"] += 1
    ctx.status()
    seed += 1
for b in bad[:20]:
    print("MISMATCH", b)
print("fuzz_passes:", counts, "mismatches:", len(bad), "in %.0f s" % (time.time() - t0))
ctx.close()
sys.exit(1 if bad else 0)
