"""one-off: the 100 M-meshlet pass of tools/bench_configs.py (`big`) against the multithreaded oracle: count and full list"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from niagara_amd import host, synth
from niagara_amd import layouts as L
from niagara_amd import pipeline as P
ctx = P.Context(0); dev = ctx.device
t = time.time()
draws, meshlets, commands, n = synth.cluster_scene(156250, 10)
cd = host.build_cull_data(draw_count=156250, cullingEnabled=1, clusterBackfaceEnabled=1)
print("scene %.1fs" % (time.time() - t), flush=True)
db, mlb, dcb = P.to_device(draws, dev), P.to_device(meshlets, dev), P.to_device(commands, dev)
ctx.upload_meshlets(mlb, len(meshlets))
c4 = synth.count4_for(n)
dccb = torch.from_numpy(c4.view(np.int32).copy()).to(dev)
cib = torch.zeros(L.CLUSTER_LIMIT + 256, dtype=torch.int32, device=dev); ccb = torch.zeros(4, dtype=torch.int32, device=dev)
for rep in range(2):
    ccb.zero_()
    ctx.clustercull(cd, 0, dcb, dccb, db, mlb, None, None, cib, ccb)
total = int(ccb[0].item())
t = time.time()
cib_o, cc4_o = np.zeros(n * 64, np.uint32), np.zeros(4, np.uint32)
oracle.clustercull(cd, 0, commands, c4, draws, meshlets, None, None, cib_o, cc4_o, threads=oracle.max_threads())
print("oracle %.1fs" % (time.time() - t))
ids = cib.cpu().numpy().view(np.uint32)[:total]
print("gpu", total, "oracle", int(cc4_o[0]), "identical list:", bool(total == int(cc4_o[0]) and (ids == cib_o[:total]).all()))
