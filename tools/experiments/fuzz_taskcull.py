#!/usr/bin/env python3
"""Randomised soak of nv_taskcull against the oracle (GPU): random scenes, flags, histories and pinned kernel forms — the early pass
is nv_clustercull's cull launch with the payload epilogue (filter / direct / lane-per-set-bit form), the late pass the
one-command-per-wave kernel.  Every command's count, its payload entries and the visibility words are compared.

    python tools/experiments/fuzz_taskcull.py [seconds=60] [first_seed=1000]
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
from niagara_amd import pipeline as P  # noqa: E402
from scenes import make_scene, random_case  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = P.Context(0)
dev = ctx.device
t0 = time.time()
runs = bad = 0
while time.time() - t0 < budget:
    kw, flags, use_soa, _ = random_case(seed)
    rng = np.random.default_rng(seed)
    scene = make_scene(**kw)
    g = G.GpuScene(ctx, scene, use_soa=use_soa)
    pyr = oracle.Pyramid(*scene["viewport"])
    oracle.depthreduce(scene["depth"], pyr)
    g.depthreduce(scene["depth"])
    form = int(rng.integers(0, 6))
    ctx.set_option(P.NV_OPT_CULL_FORM, form)
    for late in (0, 1, 0):
        fl = tuple(int(x) for x in rng.integers(0, 2, 5)) if late == 0 else flags
        cd = passes.set_flags(scene["cull"], fl)
        dvb = (rng.random(len(scene["draws"])) < 0.8).astype(np.uint32)
        cmds, c4 = passes.run_drawcull(oracle, scene, cd, 0, 1, dvb, pyr)
        oracle.tasksubmit(c4, cmds)
        ncmd = int(c4[1]) * 64
        if ncmd == 0:
            continue
        mvb0 = rng.integers(0, 2 ** 32, (scene["slots"] + 31) // 32 + 2, dtype=np.uint64).astype(np.uint32)
        mvb_o = mvb0.copy()
        pay_o, cnt_o = np.zeros((ncmd, 64), np.uint32), np.zeros(ncmd, np.uint32)
        oracle.taskcull(cd, late, cmds, c4, scene["draws"], scene["meshlets"], mvb_o, pyr, pay_o, cnt_o)
        d_pay = torch.zeros(ncmd * 64, dtype=torch.int32, device=dev)
        d_cnt = torch.full((ncmd,), -1, dtype=torch.int32, device=dev)
        d_mvb = torch.from_numpy(mvb0.view(np.int32).copy()).to(dev)
        ctx.taskcull(cd, late, P.to_device(cmds[:ncmd], dev), torch.from_numpy(c4.view(np.int32).copy()).to(dev), g.db, g.mlb, d_mvb, g.pyramid.desc, d_pay, d_cnt)
        cnt_g, pay_g = G.host_u32(d_cnt), G.host_u32(d_pay).reshape(ncmd, 64)
        keep = np.arange(64)[None, :] < cnt_o[:, None]
        ok = (cnt_g == cnt_o).all() and (pay_g[keep] == pay_o[keep]).all() and (G.host_u32(d_mvb) == mvb_o).all()
        runs += 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "late", late, "form", form, fl, kw, use_soa)
    ctx.set_option(P.NV_OPT_CULL_FORM, 0)
    seed += 1
print("fuzz_taskcull: %d passes, %d with mismatches, %.0f s" % (runs, bad, time.time() - t0))
ctx.close()
sys.exit(1 if bad else 0)
