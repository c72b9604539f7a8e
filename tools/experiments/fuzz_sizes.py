#!/usr/bin/env python3
"""Randomised soak of nv_clustercull over SIZES: command counts from a handful to ~600 k (38 M meshlets), random commands per
draw, both passes, visibility bits on / off, SoA mirror or AoS in place, each size twice in a row (the second launch picks
its filter-ring depth from the first one's count).  Count, ID list and visibility words against the multithreaded oracle.
Needs a GPU.

    python tools/experiments/fuzz_sizes.py [seconds=90] [first_seed=7000]
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
import gpu_passes as G  # noqa: E402
from niagara_amd import host, synth  # noqa: E402
from niagara_amd import pipeline as P  # noqa: E402
from scenes import make_scene  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
ctx = P.Context(0)
if os.environ.get("FUZZ_SCATTER_WAVES"):  # the throughput knobs must not change a result
    ctx.set_option(P.NV_OPT_SCATTER_WAVES, int(os.environ["FUZZ_SCATTER_WAVES"]))
if os.environ.get("FUZZ_CULL_WG"):
    ctx.set_option(P.NV_OPT_CULL_WORKGROUPS_PER_CU, int(os.environ["FUZZ_CULL_WG"]))
dev = ctx.device
threads = oracle.max_threads()
pyr = oracle.Pyramid(256, 192)
depth = make_scene(seed=3)["depth"]
oracle.depthreduce(depth, pyr)
gp = P.DepthPyramid(dev, 256, 192)
ctx.depthreduce(torch.from_numpy(depth).to(dev), 256, 192, gp.desc)
t0 = time.time()
cases, bad, meshlets_total = 0, [], 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    cpd = int(rng.integers(1, 40))
    n_cmd_target = int(10 ** rng.uniform(0.5, 5.78))
    n_draws = max(1, n_cmd_target // cpd)
    late, coe, cbe, soa = (int(x) for x in rng.integers(0, 2, 4))
    draws, meshlets, commands, n = synth.cluster_scene(n_draws, cpd, seed=seed)
    draws["position"] *= np.float32(rng.choice([0.1, 0.3, 1.0]))
    commands["lateDrawVisibility"][:n] = rng.integers(0, 2, n)
    # (round 6) half of the cases with ragged commands — any taskCount in 0 .. 64, empty ones included: what the direct form's packed walk packs — and every
    # case with a random NV_OPT_CULL_FORM (0 = by the statistics the previous cases left .. 4 = one command per wave iteration, 5 = the packed walk also with visibility bits)
    if rng.random() < 0.5:
        commands["taskCount"][:n] = rng.integers(0, 65, n)
    ctx.set_option(P.NV_OPT_CULL_FORM, int(rng.integers(0, 6)))
    cd = host.build_cull_data(draw_count=n_draws, cullingEnabled=1, clusterBackfaceEnabled=cbe, clusterOcclusionEnabled=coe, occlusionEnabled=1)
    cd["pyramidWidth"], cd["pyramidHeight"] = pyr.width, pyr.height
    mvb0 = rng.integers(0, 2 ** 32, n * 2 + 3, dtype=np.uint64).astype(np.uint32) if coe else None
    c4 = synth.count4_for(n)
    cib_o, cc4_o = np.zeros(len(commands) * 64 + 256, np.uint32), np.zeros(4, np.uint32)
    mvb_o = None if mvb0 is None else mvb0.copy()
    oracle.clustercull(cd, late, commands, c4, draws, meshlets, mvb_o, pyr, cib_o, cc4_o, threads=threads)
    total = int(cc4_o[0])
    db, mlb, dcb = P.to_device(draws, dev), P.to_device(meshlets, dev), P.to_device(commands, dev)
    if soa:
        ctx.upload_meshlets(mlb, len(meshlets))
    else:
        ctx.upload_meshlets(None, 0)
    dccb = torch.from_numpy(c4.view(np.int32).copy()).to(dev)
    cib = torch.zeros(len(commands) * 64 + 256, dtype=torch.int32, device=dev)
    ccb = torch.zeros(4, dtype=torch.int32, device=dev)
    for rep in range(2):
        d_mvb = None if mvb0 is None else torch.from_numpy(mvb0.view(np.int32).copy()).to(dev)
        ccb.zero_()
        ctx.clustercull(cd, late, dcb, dccb, db, mlb, d_mvb, gp.desc, cib, ccb)
        ok = int(ccb[0].item()) == total and (G.host_u32(cib)[:total] == cib_o[:total]).all() and (mvb0 is None or (G.host_u32(d_mvb) == mvb_o).all())
        if not ok:
            bad.append((seed, n, cpd, late, coe, cbe, soa, rep))
    ctx.status()
    cases += 1
    meshlets_total += n * 64
    seed += 1
for b in bad[:20]:
    print("MISMATCH (seed, commands, cpd, late, clusterOcclusion, backface, soa, repeat):", b)
print("fuzz_sizes: %d sizes (%.0f M meshlets in total), mismatches: %d, %.0f s" % (cases, meshlets_total / 1e6, len(bad), time.time() - t0))
ctx.close()
sys.exit(1 if bad else 0)
