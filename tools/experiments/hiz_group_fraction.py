#!/usr/bin/env python3
"""VERDICT r5 item 3, the measurement it asks for first: of the frame's late cluster pass (tools/bench_configs.py frame_scene, BASELINE scale), what fraction of
the task commands that reach the occlusion stage (>= 1 frustum / cone survivor) end with ALL their probed lanes occluded, and what fraction of the stage's
probes those commands hold.  Runs on the CPU oracle alone (no GPU): the survivors of frustum + cone are the visible list of the same commands culled with
clusterOcclusionEnabled = 0 / LATE = 0; what the late pass leaves visible is its meshletVisibility bits (clustercull.comp.glsl:125-131: bit = visible for
every valid lane)."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
import bench_configs as B  # noqa: E402
from niagara_amd import layouts as L  # noqa: E402

n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
meshes, meshlets, draws, slots, depth, cd = B.frame_scene(n_draws, 2200, 4096)
want, frames = B.oracle_frames(meshes, meshlets, draws, slots, depth, cd, 4096, 3)
late = want["late"]
cmds = late["commands"]
ncmd = int(late["count4"][0])
c4 = late["count4"].copy()
T = oracle.max_threads()
# frustum + cone survivors of the late pass's commands
cd0 = cd.copy()
cd0["clusterOcclusionEnabled"] = 0
cib, cc4 = np.zeros(ncmd * 64 + 256, np.uint32), np.zeros(4, np.uint32)
oracle.clustercull(cd0, 0, cmds, c4, draws, meshlets, None, None, cib, cc4, threads=T)
ids = cib[:int(cc4[0])]
probes = np.bincount(ids & 0xffffff, minlength=len(cmds))[:ncmd]
# visible after the occlusion test = the bits the late pass left
mvb = late["mvb"]
bits = np.unpackbits(mvb.view(np.uint8), bitorder="little")
vis = np.zeros(ncmd, np.int64)
off, cnt = cmds["meshletVisibilityOffset"][:ncmd].astype(np.int64), cmds["taskCount"][:ncmd].astype(np.int64)
csum = np.concatenate([[0], np.cumsum(bits, dtype=np.int64)])
vis = csum[off + cnt] - csum[off]
has = probes > 0
all_occ = has & (vis == 0)
print("late pass: %d commands, %d with frustum / cone survivors, %d probes, %d clusters visible after HiZ" % (ncmd, has.sum(), probes.sum(), vis.sum()))
print("commands with survivors that end with ALL probed lanes occluded: %d = %.1f %% of the listed commands, holding %d = %.1f %% of the probes"
      % (all_occ.sum(), 100.0 * all_occ.sum() / max(1, has.sum()), probes[all_occ].sum(), 100.0 * probes[all_occ].sum() / max(1, probes.sum())))
# per draw: the same question one level up (a draw-level group test)
d = cmds["drawId"][:ncmd]
pd, vd = np.bincount(d, weights=probes, minlength=n_draws), np.bincount(d, weights=vis, minlength=n_draws)
hd = pd > 0
print("draws with probes: %d; of them fully occluded at cluster level: %d (%.1f %%), holding %.1f %% of the probes" % (hd.sum(), (hd & (vd == 0)).sum(), 100.0 * (hd & (vd == 0)).sum() / max(1, hd.sum()), 100.0 * pd[hd & (vd == 0)].sum() / max(1, pd.sum())))
hist = np.bincount(np.minimum((vis[has] * 8 // np.maximum(probes[has], 1)), 8), minlength=9)
print("listed commands by visible / probed eighths (0, 1/8 .. 8/8):", hist.tolist())

# ---- what a conservative GROUP test could certify (the upper bounds above are what IS fully occluded, not what a group test can prove).  Prototype on a
# sample of the listed commands: group = bounding sphere (view space) of the command's valid meshlets' spheres; its projected box, grown by one texel of the
# level it is sampled at on every side (a member's 2 x 2 footprint at its own, finer-or-equal level stays inside), against the MIN of ALL texels of that
# level the grown box touches; certified when the group's nearest depth znear / (cz - R) <= that min (then every member's depthSphere <= its own sample).
rng = np.random.default_rng(1)
listed = np.flatnonzero(has)
sample = np.sort(rng.choice(listed, size=min(40000, len(listed)), replace=False))
sc = cmds[:ncmd][sample]
pyr = oracle.Pyramid(4096, 4096)
oracle.depthreduce(depth, pyr)
sc_probe = oracle.probe_cluster_scalars(cd, sc, draws, meshlets)  # (n, 64, 16): c = [..., 0:3], r = [..., 3]
lane = np.arange(64)[None, :]
valid = lane < sc["taskCount"][:, None]
c, r = sc_probe[..., 0:3].astype(np.float64), sc_probe[..., 3].astype(np.float64)
w = valid[..., None].astype(np.float64)
cen = (c * w).sum(1) / np.maximum(w.sum(1), 1)
R = np.where(valid, np.linalg.norm(c - cen[:, None, :], axis=-1) + r, 0).max(1)
znear, P00, P11 = float(cd["znear"][0]), float(cd["P00"][0]), float(cd["P11"][0])
pw, ph = float(cd["pyramidWidth"][0]), float(cd["pyramidHeight"][0])
cz = cen[:, 2]
ok = cz >= R + znear  # projectSphere's own precondition (otherwise the group is not tested: members go the usual way)
cx, cy = cen[:, 0], cen[:, 1]
with np.errstate(all="ignore"):
    czr2 = cz * cz - R * R
    vx, vy = np.sqrt(cx * cx + czr2), np.sqrt(cy * cy + czr2)
    minx, maxx = (vx * cx - cz * R) / (vx * cz + cx * R), (vx * cx + cz * R) / (vx * cz - cx * R)
    miny, maxy = (vy * cy - cz * R) / (vy * cz + cy * R), (vy * cy + cz * R) / (vy * cz - cy * R)
a0, a2 = minx * P00 * 0.5 + 0.5, maxx * P00 * 0.5 + 0.5
a1, a3 = maxy * P11 * -0.5 + 0.5, miny * P11 * -0.5 + 0.5
size = np.maximum((a2 - a0) * pw, (a3 - a1) * ph)
lvl = np.clip(np.ceil(np.log2(np.maximum(size, 1e-9))), 0, pyr.levels - 1).astype(int)
depthG = znear / (cz - R)
cert = np.zeros(len(sample), bool)
reads = np.zeros(len(sample), np.int64)
for L in range(pyr.levels):
    idx = np.flatnonzero(ok & (lvl == L))
    if not len(idx):
        continue
    img = pyr.level(L)
    hL, wL = img.shape
    x0 = np.clip(np.floor(a0[idx] * wL - 1.0), 0, wL - 1).astype(int); x1 = np.clip(np.floor(a2[idx] * wL + 1.0), 0, wL - 1).astype(int)
    y0 = np.clip(np.floor(a1[idx] * hL - 1.0), 0, hL - 1).astype(int); y1 = np.clip(np.floor(a3[idx] * hL + 1.0), 0, hL - 1).astype(int)
    for k, i in enumerate(idx):
        m = img[y0[k]:y1[k] + 1, x0[k]:x1[k] + 1].min()
        reads[i] = (y1[k] - y0[k] + 1) * (x1[k] - x0[k] + 1)
        cert[i] = depthG[i] <= m
sp, sv = probes[sample], vis[sample]
assert not (cert & (sv > 0)).any(), "the prototype's group test certified a command that has a visible cluster"
print("group-test prototype on %d listed commands: certifies %d (%.1f %% of them; %.1f %% of the truly all-occluded ones), holding %.1f %% of the sample's probes; %.1f texel reads per group test (members: 4 per probe)"
      % (len(sample), cert.sum(), 100.0 * cert.mean(), 100.0 * cert.sum() / max(1, ((sv == 0)).sum()), 100.0 * sp[cert].sum() / sp.sum(), reads[ok].mean()))
print("levels of the group tests:", np.bincount(lvl[ok], minlength=pyr.levels).tolist())
