#!/usr/bin/env python3
"""wave_timeline.py — development aid: per-wave s_memtime stamps of one clustercull launch (NV_DEBUG_MODE=8).

stamps (cluster_mask_kernel, last segment of each wave): 0 kernel entry · 1 segment loaded (commands, draws) and the
lane-parallel filters derived · 2 filter ring filled (issue only) · 3 pass A (filter stream) done · 4 pass B (exact tests
of the surviving commands) done · 5 ballots stored / wave end
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ["NV_DEBUG_MODE"] = str(8 | int(os.environ.get("NV_DEBUG_MODE", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from niagara_amd import host, synth  # noqa: E402
from niagara_amd import pipeline as P  # noqa: E402
from niagara_amd._lib import lib  # noqa: E402

ctx = P.Context(0)
dev = ctx.device
draws, meshlets, commands, n = synth.cluster_scene(15625, 10)
LATE = int(os.environ.get("NV_TL_LATE", "0"))  # 1: the late pass of tools/bench_configs.py config 4 (HiZ, visibility bits)
if LATE:
    size = 4096
    depth = torch.from_numpy(synth.make_depth(size, size)).to(dev)
    pyr = P.DepthPyramid(dev, size, size)
    ctx.depthreduce(depth, size, size, pyr.desc)
    cd = host.build_cull_data(draw_count=len(draws), viewport=(size, size), pyramid=(pyr.width, pyr.height), cullingEnabled=1,
                              clusterBackfaceEnabled=1, clusterOcclusionEnabled=1, occlusionEnabled=1)
    rng = np.random.default_rng(7)
    commands["lateDrawVisibility"][:n] = rng.integers(0, 2, n)
else:
    cd = host.build_cull_data(draw_count=len(draws), cullingEnabled=1, clusterBackfaceEnabled=1)
db, mlb, dcb = P.to_device(draws, dev), P.to_device(meshlets, dev), P.to_device(commands, dev)
ctx.upload_meshlets(mlb, len(meshlets))
dccb = torch.from_numpy(synth.count4_for(n).view(np.int32).copy()).to(dev)
cib = torch.zeros(n * 64 + 256, dtype=torch.int32, device=dev)
ccb = torch.zeros(4, dtype=torch.int32, device=dev)
if LATE:
    mvb0 = torch.from_numpy(rng.integers(0, 2 ** 32, n * 2 + 4, dtype=np.uint64).astype(np.uint32).view(np.int32)).to(dev)
    mvb = mvb0.clone()
for _ in range(5):
    ccb.zero_()
    if LATE:
        mvb.copy_(mvb0)
        ctx.clustercull(cd, 1, dcb, dccb, db, mlb, mvb, pyr.desc, cib, ccb)
    else:
        ctx.clustercull(cd, 0, dcb, dccb, db, mlb, None, None, cib, ccb)
torch.cuda.synchronize()
waves = 256 * int(os.environ.get("NV_CC_BLOCKS_PER_CU", "6")) * 4
out = np.zeros((waves, 8), np.uint64)
lib.nv_debug_read_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
rc = lib.nv_debug_read_timing(ctx.h, out.ctypes.data_as(C.c_void_p), waves)
assert rc == 0, rc
t = out.astype(np.int64)
t0 = t[:, 0].min()
rel = (t - t0)[:, :6]
names = ["entry", "segment+filters ready", "ring A issued", "pass A done", "pass B done", "wave end"]
print("s_memtime ticks relative to the earliest wave entry (min / median / max over %d waves)" % waves)
for i, nm in enumerate(names):
    col = rel[:, i]
    print("%-16s %9d %9d %9d" % (nm, col.min(), np.median(col), col.max()))
d = np.diff(rel, axis=1)
print("per-wave phase durations (median):", dict(zip(names[1:], np.median(d, axis=0).astype(int))))
for i, nm in enumerate(names[1:]):
    col = d[:, i]
    print("%-16s p0 %7d p10 %7d p50 %7d p90 %7d p99 %7d p100 %7d" % ((nm,) + tuple(np.percentile(col, [0, 10, 50, 90, 99, 100]).astype(int))))
tot = rel[:, 5] - rel[:, 0]
print("entry->end      ", np.percentile(tot, [0, 10, 50, 90, 99, 100]).astype(int))
busy = rel[:, 4] - rel[:, 0]
print("entry->loop done", np.percentile(busy, [0, 10, 50, 90, 99, 100]).astype(int))

# per-workgroup span (stamps of one workgroup share an XCD clock): balance across workgroups
blk = rel.reshape(-1, 4, 6)
span = blk[:, :, 5].max(axis=1) - blk[:, :, 0].min(axis=1)
print("per-workgroup span", np.percentile(span, [0, 10, 50, 90, 99, 100]).astype(int))

# per-XCD view (each XCD has its own s_memtime base): when does the LAST wave of the XCD pass each stamp, relative to
# the XCD's first wave entry
xcd = (t[:, 0] // 10**9)
for x in np.unique(xcd):
    sel = t[xcd == x][:, :6]
    base = sel[:, 0].min()
    print("xcd@%d: %5d waves  last entry %6d | last passA-done %6d | last passB-done %6d | last end %6d | median passA-done %6d" %
          (x, len(sel), sel[:, 0].max() - base, sel[:, 3].max() - base, sel[:, 4].max() - base, sel[:, 5].max() - base, np.median(sel[:, 3]) - base))

# what do the slowest waves spend their time on?
order = np.argsort(tot)
for name, idx in (("fastest 10%", order[:len(order) // 10]), ("middle 10%", order[len(order) * 45 // 100:len(order) * 55 // 100]),
                  ("slowest 10%", order[-(len(order) // 10):]), ("slowest 1%", order[-(len(order) // 100):])):
    print("%-12s total %6d = " % (name, tot[idx].mean()) + "  ".join("%s %6d" % (nm.split()[0] + nm.split()[1][:1] if " " in nm else nm, d[idx, i].mean()) for i, nm in enumerate(names[1:])))
# per-workgroup: when does the workgroup's last wave end relative to its first entry, vs the mean over its waves
print("workgroups: span p50 %d p90 %d p100 %d" % tuple(np.percentile(span, [50, 90, 100])))

# spread of the pass-A end within a workgroup (what a workgroup-level pooling of the exact pass would wait for), and the
# exact-pass time summed per workgroup vs the longest single wave
a_end = blk[:, :, 3]
print("pass-A end spread within a workgroup (max - min): p50 %d p90 %d p100 %d" % tuple(np.percentile(a_end.max(axis=1) - a_end.min(axis=1), [50, 90, 100])))
pb = (blk[:, :, 4] - blk[:, :, 3])
print("exact pass per workgroup: longest wave p50 %d p90 %d p99 %d p100 %d | mean over its 4 waves p50 %d p90 %d p99 %d p100 %d" %
      (tuple(np.percentile(pb.max(axis=1), [50, 90, 99, 100])) + tuple(np.percentile(pb.mean(axis=1), [50, 90, 99, 100]))))
end_w = blk[:, :, 5].max(axis=1) - blk[:, :, 0].min(axis=1)
ideal = (a_end.max(axis=1) - blk[:, :, 0].min(axis=1)) + pb.mean(axis=1)
print("workgroup lifetime now p50 %d p90 %d p99 %d p100 %d | pooled estimate p50 %d p90 %d p99 %d p100 %d" %
      (tuple(np.percentile(end_w, [50, 90, 99, 100])) + tuple(np.percentile(ideal, [50, 90, 99, 100]))))

# chip-wide 100 MHz clock (stamps 6, 7): launch ramp and finish order across the whole grid, in microseconds
rt0 = t[:, 6].min()
entry_us = (t[:, 6] - rt0) / 100.0
end_us = (t[:, 7] - rt0) / 100.0
print("wave entry  (us after the first wave): p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(entry_us, [10, 50, 90, 99, 100])))
print("wave end    (us after the first wave): p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(end_us, [10, 50, 90, 99, 100])))
life = end_us - entry_us
print("wave lifetime (us): p10 %.2f p50 %.2f p90 %.2f max %.2f ; corr(entry, lifetime) = %.2f" % (tuple(np.percentile(life, [10, 50, 90, 100])) + (np.corrcoef(entry_us, life)[0, 1],)))
late = np.argsort(end_us)[-60:]
print("last 1%% of waves to finish: entry %.2f lifetime %.2f exact-pass cycles %d (all waves: %.2f %.2f %d)" %
      (entry_us[late].mean(), life[late].mean(), d[late, 3].mean(), entry_us.mean(), life.mean(), d[:, 3].mean()))

# is the finish time systematic? by XCD (workgroup index mod 8, the usual round-robin) and by dispatch order
bidx = np.arange(waves) // 4
for x in range(8):
    sel = (bidx % 8) == x
    print("xcd~%d: end p50 %.2f p90 %.2f max %.2f | lifetime p50 %.2f | exact-pass cycles mean %d" %
          (x, np.percentile(end_us[sel], 50), np.percentile(end_us[sel], 90), end_us[sel].max(), np.percentile(life[sel], 50), d[sel, 3].mean()))
q = np.argsort(bidx)
for lo in range(0, 1536, 256):
    sel = (bidx >= lo) & (bidx < lo + 256)
    print("workgroups %4d..%4d: entry p50 %.2f end p50 %.2f p90 %.2f max %.2f" % (lo, lo + 255, np.percentile(entry_us[sel], 50), np.percentile(end_us[sel], 50), np.percentile(end_us[sel], 90), end_us[sel].max()))
nocand = d[:, 3] < 400
print("waves without exact-pass work: %d, end p50 %.2f p90 %.2f p99 %.2f max %.2f" % ((nocand.sum(),) + tuple(np.percentile(end_us[nocand], [50, 90, 99, 100]))))
print("waves with exact-pass work:    %d, end p50 %.2f p90 %.2f p99 %.2f max %.2f" % (((~nocand).sum(),) + tuple(np.percentile(end_us[~nocand], [50, 90, 99, 100]))))
for lo in range(0, 1536, 256):
    sel = (bidx >= lo) & (bidx < lo + 256)
    print("workgroups %4d..%4d phases (median cycles): " % (lo, lo + 255) + "  ".join("%s %6d" % (nm[:12], np.median(d[sel, i])) for i, nm in enumerate(names[1:])) + "  | lifetime us %.2f" % np.median(life[sel]))
