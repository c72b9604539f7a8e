"""development aid: which cluster ids differ between the HIP path and the oracle on a special-values scene"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import oracle
from niagara_amd import pipeline as P
import test_special_values as T
seed, late, soa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = T.special_scene(900 + seed)
co, io, mo, cd, pyr = T.run_cpu(oracle, s, late)
ctx = P.Context(); dev = ctx.device
gp = P.DepthPyramid(dev, 256, 192)
ctx.depthreduce(torch.from_numpy(s["depth"]).to(dev), 256, 192, gp.desc)
db, mlb, dcb = P.to_device(s["draws"], dev), P.to_device(s["meshlets"], dev), P.to_device(s["commands"], dev)
if soa: ctx.upload_meshlets(mlb, len(s["meshlets"]))
dccb = torch.from_numpy(s["count4"].view(np.int32).copy()).to(dev)
mvb = torch.from_numpy(s["mvb"].view(np.int32).copy()).to(dev)
cib = torch.zeros(s["n"] * 64 + 256, dtype=torch.int32, device=dev); ccb = torch.zeros(4, dtype=torch.int32, device=dev)
ctx.clustercull(cd, late, dcb, dccb, db, mlb, mvb, gp.desc, cib, ccb)
total = int(ccb[0].item()); ig = cib.cpu().numpy().view(np.uint32)[:total]
print("mode", os.environ.get("NV_DEBUG_MODE"), "gpu", total, "oracle", int(co[0]))
so, sg = set(io.tolist()), set(ig.tolist())
for ci in sorted(so ^ sg):
    c, l = ci & 0xffffff, ci >> 24
    cmd = s["commands"][c]; d = s["draws"][cmd["drawId"]]; m = s["meshlets"][cmd["taskOffset"] + l]
    print("  id cmd %d lane %d in %s: draw pos %s scale %s q %s | center %s radius %s cone %s %d" % (c, l, "oracle" if ci in so else "gpu", d["position"], d["scale"], d["orientation"],
          m["center"].view(np.float16), m["radius"].view(np.float16) if hasattr(m["radius"], "view") else m["radius"], m["cone_axis"], m["cone_cutoff"]))
