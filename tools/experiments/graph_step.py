"""development aid: does a captured hipGraph of the config-3A pass (4 rotating input sets per graph) beat eager launches?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from niagara_amd import host, synth
from niagara_amd import layouts as L
from niagara_amd import pipeline as P

ctx = P.Context(0)
dev = ctx.device
ctx.set_option(P.NV_OPT_FUSED_COUNT_RESET, 1)
n_draws, cpd, copies = 15625, 10, 4
n_cmd = n_draws * cpd
n_meshlets = n_cmd * 64
draws = host.synth_draws(n_draws, 1, 300.0)
meshlets = synth.make_meshlets(n_meshlets, seed=2)
cd = host.build_cull_data(draw_count=n_draws, cullingEnabled=1, clusterBackfaceEnabled=1)
db = P.to_device(draws, dev)
mlb = torch.empty(copies * n_meshlets * 24, dtype=torch.uint8, device=dev)
one = torch.from_numpy(meshlets.view(np.uint8).reshape(-1))
for c in range(copies):
    mlb[c * one.numel():(c + 1) * one.numel()].copy_(one)
dcbs = [P.to_device(synth.make_task_commands(n_draws, cpd, meshlet_base=c * n_meshlets), dev) for c in range(copies)]
dccb = torch.from_numpy(synth.count4_for(n_cmd).view(np.int32).copy()).to(dev)
cib = torch.zeros(n_meshlets + 256, dtype=torch.int32, device=dev)
ccb = torch.zeros(4, dtype=torch.int32, device=dev)
ctx.upload_meshlets(mlb, copies * n_meshlets)

def step(i):
    ctx.clustercull(cd, 0, dcbs[i % copies], dccb, db, mlb, None, None, cib, ccb)

for i in range(20):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(400):
    step(i)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 400 * 1e6
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(copies):
            step(i)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 400 * 1e6
print("eager %.2f us/pass, graph replay %.2f us/pass, visible %d" % (eager, graph, int(ccb[0].item())))
