#!/usr/bin/env python3
"""ranged_dealing.py — round 3 experiment: would a cost-balanced, contiguous dealing of the cull kernel's chunks cut its tail?

cluster_mask_kernel deals chunks of 4 commands round-robin over its 6144 waves (weighted by the workgroup's generation); a chunk
whose commands pass the conservative filter costs a wave several times a chunk the filter finishes, those chunks cluster per visible
draw, and the launch ends with the few waves that drew several of them (waves leave between 16 and 21 us).  Here the
HOST computes, with numpy, which commands can have a frustum survivor, gives every chunk the cost 4 + kappa * (its candidate commands),
and hands the experiments build a table of W + 1 chunk boundaries with equal cost per wave (optionally weighted by generation like
the static dealing); the kernel then walks contiguous ranges (ClusterArgs::dealRanges).  Prints the cull kernel's time (HIP events,
cache-cold rotation as in bench.py) for the static dealing and for every kappa.

    NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so python tools/experiments/ranged_dealing.py
(needs tools/experiments/ranged_dealing_r3.diff applied: the kernel side of the experiment is not in the product.  Result, in the
patch header: no variant beats the static dealing.)
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import niagara_amd  # noqa: E402
from niagara_amd import host, synth  # noqa: E402
from niagara_amd import layouts as L  # noqa: E402
from niagara_amd import pipeline as P  # noqa: E402
from niagara_amd._lib import lib  # noqa: E402


def candidate_commands(cd, draws, meshlets, n_cmd, cpd):
    """bool per command: some meshlet's sphere is inside (or touches) the frustum — what the conservative filter cannot finish"""
    c = meshlets["center"].view(np.float16).astype(np.float64).reshape(-1, 3)
    r = meshlets["radius"].view(np.float16).astype(np.float64)
    d = draws[np.arange(n_cmd * 64) // 64 // cpd]
    q, w = d["orientation"][:, :3].astype(np.float64), d["orientation"][:, 3].astype(np.float64)
    t = np.cross(q, c) + w[:, None] * c
    rot = c + 2.0 * np.cross(q, t)
    world = rot * d["scale"].astype(np.float64)[:, None] + d["position"].astype(np.float64)
    V = cd["view"][0].astype(np.float64).reshape(4, 4).T  # column-major -> V[r, k]
    v = world @ V[:3, :3].T + V[:3, 3]
    rad = r * np.abs(d["scale"].astype(np.float64))
    f = cd["frustum"][0].astype(np.float64)
    ok = (v[:, 2] * f[1] - np.abs(v[:, 0]) * f[0] > -rad) & (v[:, 2] * f[3] - np.abs(v[:, 1]) * f[2] > -rad)
    ok &= (v[:, 2] + rad > float(cd["znear"][0])) & (v[:, 2] - rad < float(cd["zfar"][0]))
    return ok.reshape(n_cmd, 64).any(axis=1)


def boundaries(cost, waves, gen_weights):
    """chunk boundaries with (weighted) equal cost per wave: wave w owns [b[w], b[w + 1])"""
    share = np.ones(waves)
    if gen_weights is not None:
        per_gen = waves // 6
        mean_cost = cost.sum() / waves
        delay = np.array([0, 0.5, 2.4, 4.2, 7.1, 13.1])  # make_dealing's start delays in commands (pass A cost 1 each)
        for g in range(6):
            share[g * per_gen:(g + 1) * per_gen] = max(0.2, (mean_cost + gen_weights * (delay.mean() - delay[g])) / mean_cost)
    target = np.concatenate([[0.0], np.cumsum(share)]) / share.sum() * cost.sum()
    prefix = np.concatenate([[0.0], np.cumsum(cost)])
    b = np.searchsorted(prefix, target, side="left").astype(np.uint32)
    b[0], b[-1] = 0, len(cost)
    return np.maximum.accumulate(b)


def main():
    if "exp" not in niagara_amd.SO_PATH:
        raise SystemExit("needs the experiments build: NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so")
    lib.nv_debug_set_deal_ranges.restype = C.c_int
    lib.nv_debug_set_deal_ranges.argtypes = [C.c_void_p, C.c_void_p]
    n_draws, cpd, copies, iters = 15625, 10, 4, 100
    ctx = P.Context(0)
    dev = ctx.device
    ctx.set_option(P.NV_OPT_FUSED_COUNT_RESET, 1)
    draws, meshlets, commands, n = synth.cluster_scene(n_draws, cpd)
    cd = host.build_cull_data(draw_count=n_draws, cullingEnabled=1, clusterBackfaceEnabled=1)
    m = n * 64
    db = P.to_device(draws, dev)
    mlb = torch.cat([P.to_device(meshlets, dev) for _ in range(copies)])
    dcbs = [P.to_device(synth.make_task_commands(n_draws, cpd, meshlet_base=c * m), dev) for c in range(copies)]
    ctx.upload_meshlets(mlb, copies * m)
    dccb = torch.from_numpy(synth.count4_for(n).view(np.int32).copy()).to(dev)
    cib = torch.zeros(m + 256, dtype=torch.int32, device=dev)
    ccb = torch.zeros(4, dtype=torch.int32, device=dev)

    cand = candidate_commands(cd, draws, meshlets, n, cpd)
    chunks = (n + 3) // 4
    per_chunk = np.add.reduceat(cand.astype(np.float64), np.arange(0, n, 4))
    waves = 256 * 6 * 4  # the cull launch: 6 workgroups of 4 waves per CU (MI355X: 256 CUs)
    print("commands %d, candidates %d (%.1f %%), chunks %d, waves %d" % (n, cand.sum(), 100.0 * cand.mean(), chunks, waves))

    def measure(label):
        for i in range(8):
            ctx.clustercull(cd, 0, dcbs[i % copies], dccb, db, mlb, None, None, cib, ccb)
        torch.cuda.synchronize()
        ctx.profile(True)
        for i in range(iters):
            ctx.clustercull(cd, 0, dcbs[i % copies], dccb, db, mlb, None, None, cib, ccb)
        torch.cuda.synchronize()
        pr = ctx.profile_read()
        ctx.profile(False)
        print("%-44s cull %.2f us  scatter %.2f us  visible %d" % (label, pr["cluster_cull"][0] / iters * 1e3, pr["cluster_scatter"][0] / iters * 1e3, int(ccb[0].item())))

    measure("static dealing (product)")
    keep = []
    for kappa in (0.0, 0.5, 1.0, 2.0, 4.0, 8.0):
        for gw in (None, 1.0):
            cost = 4.0 + kappa * per_chunk
            b = boundaries(cost, waves, gw)
            table = torch.from_numpy(np.concatenate([[chunks, waves], b]).astype(np.uint32).view(np.int32)).to(dev)
            keep.append(table)
            lib.nv_debug_set_deal_ranges(ctx.h, C.c_void_p(table.data_ptr()))
            measure("ranged, kappa %.1f, generation weights %s" % (kappa, "on" if gw else "off"))
    lib.nv_debug_set_deal_ranges(ctx.h, None)
    measure("static dealing again")
    ctx.close()


if __name__ == "__main__":
    main()
