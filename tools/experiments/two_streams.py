#!/usr/bin/env python3
"""Throughput of config 3A passes issued alternately on S HIP streams (one nv_context per stream, own scratch and outputs):
the scatter launch of pass i (latency-bound, 16 waves per CU) overlaps the start of the cull launch of pass i+1.
    python tools/experiments/two_streams.py [streams=2] [steps=200]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from niagara_amd import host, synth  # noqa: E402
from niagara_amd import layouts as L  # noqa: E402
from niagara_amd import pipeline as P  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
n_draws, cpd, copies = 15625, 10, 4
n_cmd = n_draws * cpd
n_meshlets = n_cmd * 64
draws = host.synth_draws(n_draws, 1, 300.0)
draws["meshletVisibilityOffset"] = np.arange(n_draws, dtype=np.uint32) * (cpd * 64)
meshlets = synth.make_meshlets(n_meshlets, seed=2)
cd = host.build_cull_data(draw_count=n_draws, cullingEnabled=1, clusterBackfaceEnabled=1)
db = P.to_device(draws, dev)
mlb = torch.empty(copies * n_meshlets * L.MESHLET.itemsize, dtype=torch.uint8, device=dev)
one = torch.from_numpy(meshlets.view(np.uint8).reshape(-1))
for c in range(copies):
    mlb[c * one.numel():(c + 1) * one.numel()].copy_(one)
dcbs = [P.to_device(synth.make_task_commands(n_draws, cpd, meshlet_base=c * n_meshlets), dev) for c in range(copies)]
dccb = torch.from_numpy(synth.count4_for(n_cmd).view(np.int32).copy()).to(dev)
ctxs, streams, cibs, ccbs = [], [], [], []
for s in range(S):
    c = P.Context(0)
    c.set_option(P.NV_OPT_FUSED_COUNT_RESET, 1)
    if os.environ.get("TS_SCATTER_WAVES"):
        c.set_option(P.NV_OPT_SCATTER_WAVES, int(os.environ["TS_SCATTER_WAVES"]))
    if os.environ.get("TS_CULL_WG"):
        c.set_option(P.NV_OPT_CULL_WORKGROUPS_PER_CU, int(os.environ["TS_CULL_WG"]))
    c.upload_meshlets(mlb, copies * n_meshlets)
    ctxs.append(c)
    streams.append(torch.cuda.Stream())
    cibs.append(torch.zeros(n_meshlets + 256, dtype=torch.int32, device=dev))
    ccbs.append(torch.zeros(4, dtype=torch.int32, device=dev))
torch.cuda.synchronize()


def run(n):
    for i in range(n):
        s = i % S
        with torch.cuda.stream(streams[s]):
            ctxs[s].clustercull(cd, 0, dcbs[i % copies], dccb, db, mlb, None, None, cibs[s], ccbs[s])
    torch.cuda.synchronize()


run(20)
t0 = time.perf_counter()
run(steps)
el = time.perf_counter() - t0
print("streams", S, "steps", steps, "us/step", round(el / steps * 1e6, 2), "G meshlets/s", round(n_meshlets * steps / el / 1e9, 1), "visible", [int(c[0].item()) for c in ccbs])
