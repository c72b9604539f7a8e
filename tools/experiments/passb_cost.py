#!/usr/bin/env python3
"""passb_cost.py — development aid (experiments build, NV_DEBUG_MODE = 8 | 134217728): what a candidate command costs a wave in pass B of the
headline cull launch — cycles of the exact pass against the number of candidates the wave's segment had (slope = per candidate, intercept =
what the first one waits for)."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["NV_DEBUG_MODE"] = str(8 | 134217728 | int(os.environ.get("NV_DEBUG_MODE", "0")))  # (further bits from the caller: 4096 = pass-B loads only, 1048576 = reference arithmetic only)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from niagara_amd import host, synth  # noqa: E402
from niagara_amd import pipeline as P  # noqa: E402
from niagara_amd._lib import lib  # noqa: E402

ctx = P.Context(0)
dev = ctx.device
ctx.set_option(P.NV_OPT_FUSED_COUNT_RESET, 1)
draws, meshlets, commands, n = synth.cluster_scene(15625, 10)
cd = host.build_cull_data(draw_count=len(draws), cullingEnabled=1, clusterBackfaceEnabled=1)
db, mlb, dcb = P.to_device(draws, dev), P.to_device(meshlets, dev), P.to_device(commands, dev)
ctx.upload_meshlets(mlb, len(meshlets))
dccb = torch.from_numpy(synth.count4_for(n).view(np.int32).copy()).to(dev)
cib = torch.zeros(n * 64 + 256, dtype=torch.int32, device=dev)
ccb = torch.zeros(4, dtype=torch.int32, device=dev)
for _ in range(6):
    ctx.clustercull(cd, 0, dcb, dccb, db, mlb, None, None, cib, ccb)
torch.cuda.synchronize()
waves = 256 * 6 * 4
out = np.zeros((waves, 8), np.uint64)
lib.nv_debug_read_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
assert lib.nv_debug_read_timing(ctx.h, out.ctypes.data_as(C.c_void_p), waves) == 0
t = out.astype(np.int64)
cand = t[:, 2]
pb = t[:, 4] - t[:, 3]
end_us = (t[:, 7] - t[:, 6].min()) / 100.0
print("visible", int(ccb[0].item()), "candidates in all", int(cand.sum()), "waves with candidates", int((cand > 0).sum()))
for k in range(0, int(cand.max()) + 1):
    sel = cand == k
    if sel.sum():
        print("candidates %2d: %4d waves  pass B cycles p10 %6d p50 %6d p90 %6d  | wave end us p50 %.2f max %.2f" %
              (k, sel.sum(), *np.percentile(pb[sel], [10, 50, 90]), np.percentile(end_us[sel], 50), end_us[sel].max()))
sel = cand > 0
A = np.vstack([cand[sel], np.ones(sel.sum())]).T
slope, icpt = np.linalg.lstsq(A, pb[sel], rcond=None)[0]
print("fit: pass B cycles = %.0f x candidates + %.0f" % (slope, icpt))
