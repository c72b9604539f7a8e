#!/usr/bin/env python3
"""bench_configs.py — the other BASELINE.json configs (2, 3B, 4, roofline size), one JSON line each.

Not the driver's bench (that is bench.py = config 3A); these numbers feed DESIGN.md.  Per-kernel times come from the
library's HIP events on the launch stream (nv_profile_*), median over `--iters` launches, inputs rotated so that every
launch is cache-cold where the working set would otherwise fit the 256 MiB Infinity Cache.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import oracle  # noqa: E402  (the checker: every config's output is compared with the CPU oracle before its time is reported)
from niagara_amd import host, synth  # noqa: E402
from niagara_amd import layouts as L  # noqa: E402
from niagara_amd import pipeline as P  # noqa: E402

HBM = 8000.0


def valu_roofline(kernel, measured_us):
    """the vector-issue roofline of an issue-bound kernel next to its HBM figure (tools/valu_roofline.py; None while no
    profiles/rNN_valu_counters.json is committed or hipcc is not at hand)"""
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import valu_roofline as V
        return V.roofline_valu(kernel, measured_us)
    except Exception as e:  # noqa: BLE001 — a reporting aid must not fail a measurement
        return {"error": str(e)[:200]}


def timed(ctx, fn, iters, slot):
    """(wall time per call in us with NO event brackets between the launches, average of the library's own event pair for
    `slot` in us, all slots); the wall time of the instrumented loop — every event record is a barrier packet and costs a few
    microseconds — is returned as prof["wall_with_events_us"]"""
    wall = plain_wall(fn, iters)
    ctx.profile(True)
    t0 = time.perf_counter()
    for i in range(iters):
        fn(i)
    torch.cuda.synchronize()
    wall_events = (time.perf_counter() - t0) / iters
    prof = ctx.profile_read()
    ctx.profile(False)
    ms, n = prof[slot]
    prof["wall_with_events_us"] = wall_events * 1e6
    return wall, ms / max(1, n) * 1e3, prof


def plain_wall(fn, iters, repeats=3):
    """wall time per call with no event brackets between the launches: the SHORTEST of `repeats` timed loops of `iters` calls.  (One loop of 100 phases is 4 ms
    of wall time: a single scheduling hiccup of the launching thread — seen in the driver-style line of round 6: contract chain 61.5 us where five other runs
    on the same box gave 37.9-38.3 — is half of it.  bench.py's own timed region is not this function.)"""
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        for i in range(iters):
            fn(i)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / iters * 1e6
        best = t if best is None or t < best else best
    return best


def graph_wall(fn, iters):
    """the same calls captured once into a HIP graph (the C ABI is stream-ordered and keeps its cross-launch state in device
    memory, so a captured frame phase replays) and replayed: what is left of the launch gaps"""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn(0)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / iters * 1e6
    return us


def config2(ctx, iters, n_draws=1_000_000, copies=6, soa=True, fused_reset=False):
    """1 M MeshDraw spheres, frustum cull + LOD + ordered compaction: drawcull<LATE=0,TASK=0>.  fused_reset: the pass absorbs the
    caller's vkCmdFillBuffer(dccb, 0, 4, 0) (NV_OPT_FUSED_COUNT_RESET) instead of nv_reset_count as its own launch"""
    dev = ctx.device
    ctx.set_option(P.NV_OPT_FUSED_COUNT_RESET, int(fused_reset))
    meshes, _ = synth.make_meshes(1, 8, 1 << 12)
    meshes["center"] = (-0.016, -0.028, -0.034)
    meshes["radius"] = 0.598  # kitten sphere (tests/golden/kitten_bounds.json)
    draws = host.synth_draws(n_draws, 1, 300.0)
    host.assign_visibility_offsets(draws, meshes)
    cd = host.build_cull_data(draw_count=n_draws, cullingEnabled=1, lodEnabled=1)
    mb = P.to_device(meshes, dev)
    ctx.upload_meshes(mb, len(meshes))
    one = P.to_device(draws, dev)
    db_all = torch.cat([one] * copies)  # `copies` shards of one buffer, rotated: every pass streams from HBM
    dbs = [db_all[c * one.numel():(c + 1) * one.numel()] for c in range(copies)]
    if soa:
        ctx.upload_draws(db_all, copies * n_draws, mb)
    dvbs = [torch.ones(n_draws, dtype=torch.int32, device=dev) for _ in range(copies)]
    dcb = torch.zeros(n_draws * 24 + 64, dtype=torch.uint8, device=dev)
    dccb = torch.zeros(4, dtype=torch.int32, device=dev)

    def step(i):
        if not fused_reset:
            ctx.reset_count(dccb)
        ctx.drawcull(cd, 0, 0, dbs[i % copies], mb, dcb, dccb, dvbs[i % copies], None)

    wall, k_us, prof = timed(ctx, step, iters, "drawcull")
    ctx.set_option(P.NV_OPT_FUSED_COUNT_RESET, 0)
    v = int(dccb[0].item())
    # parity: commands, count and (untouched by the early pass) drawVisibility against the oracle
    co, c4o, dvo = np.zeros(n_draws + 1, dtype=L.DRAWCMD), np.zeros(4, np.uint32), np.ones(n_draws, np.uint32)
    oracle.drawcull(cd, 0, 0, draws, meshes, co, c4o, dvo, None, threads=oracle.max_threads())
    same = (v == int(c4o[0]) and dcb[:v * 24].cpu().numpy().tobytes() == co[:v].tobytes()
            and (dvbs[(iters - 1) % copies].cpu().numpy().view(np.uint32) == dvo).all())
    algo = n_draws * 52 + v * 24 + 208 + 4
    return dict(config="2: 1M draws, drawcull<0,0>" + ("" if soa else " (AoS records in place)") + (" (count reset fused)" if fused_reset else ""), draws=n_draws, visible=v, kernel_us=k_us, step_us=wall, step_us_with_events=prof["wall_with_events_us"], draws_per_s=n_draws / (k_us * 1e-6),
                algorithmic_bytes=algo, achieved_GBs=algo / k_us / 1e3, frac=algo / k_us / 1e3 / HBM, parity=verdict(same),
                **({"roofline_valu": {"draw_decide_kernel (measured_us = the decide AND the scatter launch)": valu_roofline("draw_decide_kernel", k_us)}} if soa and not fused_reset else {}))


def config2_late(ctx, iters, n_draws=1_000_000, copies=6, size=2048):
    """the same 1 M draws through drawcull<LATE=1,TASK=0> with the HiZ test against a 2048^2 pyramid"""
    dev = ctx.device
    meshes, _ = synth.make_meshes(1, 8, 1 << 12)
    meshes["center"] = (-0.016, -0.028, -0.034)
    meshes["radius"] = 0.598
    draws = host.synth_draws(n_draws, 1, 300.0)
    host.assign_visibility_offsets(draws, meshes)
    depth = torch.from_numpy(synth.make_depth(size, size)).to(dev)
    pyr = P.DepthPyramid(dev, size, size)
    ctx.depthreduce(depth, size, size, pyr.desc)
    cd = host.build_cull_data(draw_count=n_draws, viewport=(size, size), pyramid=(pyr.width, pyr.height), cullingEnabled=1, lodEnabled=1, occlusionEnabled=1)
    mb = P.to_device(meshes, dev)
    ctx.upload_meshes(mb, len(meshes))
    one = P.to_device(draws, dev)
    db_all = torch.cat([one] * copies)
    dbs = [db_all[c * one.numel():(c + 1) * one.numel()] for c in range(copies)]
    ctx.upload_draws(db_all, copies * n_draws, mb)
    rng = np.random.default_rng(3)
    dvb0 = torch.from_numpy(rng.integers(0, 2, n_draws).astype(np.int32)).to(dev)
    dvbs = [dvb0.clone() for _ in range(copies)]
    dcb = torch.zeros(n_draws * 24 + 64, dtype=torch.uint8, device=dev)
    dccb = torch.zeros(4, dtype=torch.int32, device=dev)

    def step(i):
        dvbs[i % copies].copy_(dvb0)
        ctx.reset_count(dccb)
        ctx.drawcull(cd, 1, 0, dbs[i % copies], mb, dcb, dccb, dvbs[i % copies], pyr.desc)

    wall, k_us, prof = timed(ctx, step, iters, "drawcull")
    v = int(dccb[0].item())
    # parity: pyramid, commands, count and the rewritten drawVisibility against the oracle
    po = oracle.Pyramid(size, size)
    oracle.depthreduce(depth.cpu().numpy(), po)
    co, c4o, dvo = np.zeros(n_draws + 1, dtype=L.DRAWCMD), np.zeros(4, np.uint32), dvb0.cpu().numpy().view(np.uint32).copy()
    oracle.drawcull(cd, 1, 0, draws, meshes, co, c4o, dvo, po, threads=oracle.max_threads())
    same = ((pyr.data.cpu().numpy() == po.data).all() and v == int(c4o[0]) and dcb[:v * 24].cpu().numpy().tobytes() == co[:v].tobytes()
            and (dvbs[(iters - 1) % copies].cpu().numpy().view(np.uint32) == dvo).all())
    return dict(config="2L: 1M draws, drawcull<1,0> with HiZ", draws=n_draws, visible=v, kernel_us=k_us, step_us=wall, step_us_with_events=prof["wall_with_events_us"],
                draws_per_s=n_draws / (k_us * 1e-6), parity=verdict(same))


def config3b(ctx, iters, n_draws=15625 * 4, fused=False):
    """contract path: drawcull<0,TASK> (LOD on, 64 meshes x 4 LODs) -> tasksubmit -> clustercull<0>"""
    dev = ctx.device
    meshes, total = synth.make_meshes(64, 4, 640)
    meshlets = synth.make_meshlets(total)
    draws = host.synth_draws(n_draws, 64, 300.0)
    slots, _ = host.assign_visibility_offsets(draws, meshes)
    cd = host.build_cull_data(draw_count=n_draws, cullingEnabled=1, lodEnabled=1, clusterBackfaceEnabled=1)
    cd["cullingEnabled"] = 0  # every draw emits commands: the cluster pass sees the whole pool
    pipe = P.VisibilityPipeline(meshes, meshlets, draws, (1024, 768), ctx=ctx, task_capacity=n_draws * 10 + 64, fused=fused)
    pipe.dvb.fill_(1)

    def step(i):
        pipe.cull(cd, late=False, task=True)
        pipe.render_clusters(cd, late=False)

    wall_plain, k_us, prof = timed(ctx, step, iters, "cluster_cull")
    wall = prof["wall_with_events_us"]
    wall_graph = graph_wall(step, iters)
    cmds = int(pipe.dccb[0].item())
    tested = int((P.from_device(pipe.dcb, L.TASKCMD)[:cmds]["taskCount"]).sum())
    # parity: every buffer of the chain against the oracle (the pipeline rewrote the draws' visibility offsets: use its copy)
    T = oracle.max_threads()
    pdc = cd.copy()
    pdc["clusterBackfaceEnabled"] = 1  # what VisibilityPipeline.cull passes for postPass 0 (src/niagara.cpp:1549)
    co, c4o = np.zeros(n_draws * 10 + 128, dtype=L.TASKCMD), np.zeros(4, np.uint32)
    oracle.drawcull(pdc, 0, 1, pipe.draws_host, meshes, co, c4o, np.ones(n_draws, np.uint32), None, threads=T)
    oracle.tasksubmit(c4o, co)
    ncmd = int(c4o[1]) * 64
    cib_o, cc4_o = np.zeros(ncmd * 64 + 256, np.uint32), np.zeros(4, np.uint32)
    oracle.clustercull(cd, 0, co, c4o, pipe.draws_host, meshlets, None, None, cib_o, cc4_o, threads=T)
    oracle.clustersubmit(cc4_o, cib_o)
    nv = (int(cc4_o[0]) + 255) // 256 * 256
    same = ((pipe.dccb.cpu().numpy().view(np.uint32) == c4o).all() and P.from_device(pipe.dcb, L.TASKCMD)[:ncmd].tobytes() == co[:ncmd].tobytes()
            and (pipe.ccb.cpu().numpy().view(np.uint32) == cc4_o).all() and (pipe.cib[:nv].cpu().numpy().view(np.uint32) == cib_o[:nv]).all()
            and bool((pipe.dvb == 1).all().item()))
    return dict(config="3B: drawcull<0,1> -> tasksubmit -> clustercull<0> -> clustersubmit" + (" (NV_OPT_FUSED_SUBMIT + FUSED_COUNT_RESET: 4 launches)" if fused else " (8 launches)"), draws=n_draws, task_commands=cmds, meshlets_tested=tested,
                visible=int(pipe.ccb[0].item()), step_us=wall_plain, step_us_with_events=wall, step_us_graph_replay=wall_graph, cluster_cull_us=k_us, cluster_scatter_us=prof["cluster_scatter"][0] / max(1, prof["cluster_scatter"][1]) * 1e3,
                drawcull_us=prof["drawcull"][0] / max(1, prof["drawcull"][1]) * 1e3, meshlets_per_s=tested / (wall_plain * 1e-6), parity=verdict(same),
                kernel_variants=ctx.profile_variants(),
                **({"roofline_valu": {"cull launch (direct form's packed walk)": valu_roofline("cluster_mask_kernel<false, true, false, 8, true, false, true>", k_us)}} if n_draws >= 100000 else {}))


def config4(ctx, iters, size=4096, n_draws=15625, cpd=10, copies=4):
    """two-phase HiZ: 4096^2 depth -> 2048^2 x 12 pyramid, then clustercull<LATE=1> with cluster occlusion over 10 M meshlets.
    Cache-cold like configs 2 and 3A: `copies` depth targets (64 MiB each) with their own pyramids, and `copies` meshlet
    pools + command lists + visibility words, are rotated, so that neither the 64 MiB depth image nor the 120 MB of cull bytes is
    found in the 256 MiB Infinity Cache."""
    dev = ctx.device
    depth_h = synth.make_depth(size, size)
    depths = [torch.from_numpy(depth_h).to(dev) for _ in range(copies)]
    pyrs = [P.DepthPyramid(dev, size, size) for _ in range(copies)]

    def build(i):
        ctx.depthreduce(depths[i % copies], size, size, pyrs[i % copies].desc)

    wall_p, k_p, _ = timed(ctx, build, iters, "depthreduce")
    for i in range(copies):  # (every pyramid complete, whatever `iters` was)
        build(i)
    pyr = pyrs[0]
    pyr_bytes = 4 * size * size + pyr.desc.totalTexels * 4
    draws, meshlets, commands, n = synth.cluster_scene(n_draws, cpd)
    cd = host.build_cull_data(draw_count=n_draws, viewport=(size, size), pyramid=(pyr.width, pyr.height), cullingEnabled=1, clusterBackfaceEnabled=1,
                              clusterOcclusionEnabled=1, occlusionEnabled=1)
    rng = np.random.default_rng(7)
    ldv = rng.integers(0, 2, n)
    commands["lateDrawVisibility"][:n] = ldv
    m = n * 64
    db = P.to_device(draws, dev)
    mlb = torch.cat([P.to_device(meshlets, dev) for _ in range(copies)])
    dcbs = []
    for c in range(copies):
        cc = synth.make_task_commands(n_draws, cpd, meshlet_base=c * m)
        cc["lateDrawVisibility"][:n] = ldv
        dcbs.append(P.to_device(cc, dev))
    ctx.upload_meshlets(mlb, copies * m)
    dccb = torch.from_numpy(synth.count4_for(n).view(np.int32).copy()).to(dev)
    mvb0 = torch.from_numpy(rng.integers(0, 2 ** 32, n * 2 + 4, dtype=np.uint64).astype(np.uint32).view(np.int32)).to(dev)
    mvbs = [mvb0.clone() for _ in range(copies)]
    cib = torch.zeros(n * 64 + 256, dtype=torch.int32, device=dev)
    ccb = torch.zeros(4, dtype=torch.int32, device=dev)

    def late(i):
        c = i % copies
        mvbs[c].copy_(mvb0)
        ctx.reset_count(ccb)
        ctx.clustercull(cd, 1, dcbs[c], dccb, db, mlb, mvbs[c], pyrs[c].desc, cib, ccb)

    # the whole late pass back to back (count reset + cull + occlusion stage + scatter), no event brackets between the kernels;
    # the visibility words are not restored in this loop (same work per pass: every command is tested either way)
    for i in range(3):
        late(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        c = i % copies
        ctx.reset_count(ccb)
        ctx.clustercull(cd, 1, dcbs[c], dccb, db, mlb, mvbs[c], pyrs[c].desc, cib, ccb)
    torch.cuda.synchronize()
    pass_us = (time.perf_counter() - t0) / iters * 1e6
    wall_c, k_c, prof = timed(ctx, late, iters, "cluster_cull")
    k_h = prof["cluster_hiz"][0] / max(1, prof["cluster_hiz"][1]) * 1e3
    k_s = prof["cluster_scatter"][0] / max(1, prof["cluster_scatter"][1]) * 1e3
    algo = m * 12 + n * 68 + n * 8 + m // 4
    # parity: every pyramid, and the visible IDs + the rewritten visibility words of the last timed pass, against the oracle
    po = oracle.Pyramid(size, size)
    oracle.depthreduce(depth_h, po)
    mvo = mvb0.cpu().numpy().view(np.uint32).copy()
    cib_o, cc4_o = np.zeros(m, np.uint32), np.zeros(4, np.uint32)
    oracle.clustercull(cd, 1, commands, synth.count4_for(n), draws, meshlets, mvo, po, cib_o, cc4_o, threads=oracle.max_threads())
    total = int(ccb[0].item())
    last = (iters - 1) % copies
    same = (all((p.data.cpu().numpy() == po.data).all() for p in pyrs) and total == int(cc4_o[0]) and (cib[:total].cpu().numpy().view(np.uint32) == cib_o[:total]).all()
            and (mvbs[last].cpu().numpy().view(np.uint32) == mvo).all())
    return dict(config="4: 4096^2 depth pyramid + 10M-meshlet late clustercull with HiZ", input_copies_rotated=copies, parity=verdict(same), pyramid_us=k_p, pyramid_wall_us=wall_p, pyramid_bytes=pyr_bytes,
                pyramid_GBs=pyr_bytes / k_p / 1e3, pyramid_frac=pyr_bytes / k_p / 1e3 / HBM, texels_per_s=size * size / (k_p * 1e-6),
                late_cull_us=k_c, late_hiz_us=k_h, late_scatter_us=k_s, late_pass_us=pass_us, late_visible=int(ccb[0].item()),
                late_algorithmic_bytes=algo, late_frac=algo / k_c / 1e3 / HBM, late_cull_plus_hiz_frac=algo / (k_c + k_h) / 1e3 / HBM,
                late_pass_frac=algo / pass_us / 1e3 / HBM, meshlets_per_s=m / (pass_us * 1e-6))


def hiz_phase_sums(ctx, waves=4096):
    """experiments build, NV_DEBUG_MODE bit 28: per-wave cycle sums of cluster_hiz_kernel's phases (clustercull.hip HizTimes) of the last late pass"""
    import ctypes as C
    from niagara_amd._lib import lib
    out = np.zeros((6144, 8), dtype=np.uint64)
    lib.nv_debug_read_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    if lib.nv_debug_read_timing(ctx.h, out.ctypes.data_as(C.c_void_p), 6144):
        return None
    t = out[:waves].astype(np.float64)
    t = t[t[:, 6] > 0]
    names = ("chunk_setup", "bounds_wait", "probe_arithmetic", "texel_wait", "finish_atomics", "epilogue")
    tot = t[:, 6]
    r = {"waves": int(len(t)), "rounds_per_wave_mean": float(t[:, 7].mean()), "lifetime_cycles_mean": float(tot.mean()), "lifetime_cycles_p95": float(np.percentile(tot, 95)),
         "lifetime_cycles_max": float(tot.max())}
    for i, n in enumerate(names):
        r[n + "_share"] = float(t[:, i].sum() / tot.sum())
        r[n + "_cycles_per_round"] = float(t[:, i].sum() / max(1.0, t[:, 7].sum()))
    r["unaccounted_share"] = float(1.0 - t[:, :6].sum() / tot.sum())
    return r


def frame_scene(n_draws, lod0, size):
    """the frame benchmark's scene: niagara's draw generator over 64 meshes x 4 LODs, a seeded meshlet pool, a synthetic depth
    target; flags as niagara ships them with every culling feature on"""
    meshes, total = synth.make_meshes(64, 4, lod0)
    meshlets = synth.make_meshlets(total)
    draws = host.synth_draws(n_draws, 64, 300.0)
    slots, _ = host.assign_visibility_offsets(draws, meshes)
    depth = synth.make_depth(size, size)
    pw = host.previous_pow2(size)
    cd = host.build_cull_data(draw_count=n_draws, viewport=(size, size), pyramid=(pw, pw), cullingEnabled=1, lodEnabled=1, occlusionEnabled=1,
                              clusterOcclusionEnabled=1, clusterBackfaceEnabled=1)
    return meshes, meshlets, draws, slots, depth, cd


def oracle_frames(meshes, meshlets, draws, slots, depth, cd, size, frames, first_cleared=False):
    """the reference's frame order (src/niagara.cpp:1765-1788) on the CPU oracle; returns the last frame's buffers per phase and
    the per-phase counts; stops early once a frame reproduces the previous frame's state AND outputs (a static camera reaches
    that fixed point with frame 1: every later frame is identical).  first_cleared: frame 0 reduces a cleared depth target, like
    examples/frame_driver and the reference's first frame (its visibility bits outlive the frame: history matters)"""
    T = oracle.max_threads()
    n = len(draws)
    dvb, mvb = np.zeros(n, np.uint32), np.zeros((slots + 31) // 32 + 2, np.uint32)
    pyr = oracle.Pyramid(size, size)
    cap = L.TASK_WGLIMIT + 64
    prev = None
    for f in range(frames):
        rec = {}
        for phase, late in (("early", 0), ("late", 1)):
            if late:
                oracle.depthreduce(np.zeros_like(depth) if first_cleared and f == 0 else depth, pyr)
            co, c4 = np.zeros(cap, dtype=L.TASKCMD), np.zeros(4, np.uint32)
            pd = cd.copy()
            pd["clusterBackfaceEnabled"] = 1
            oracle.drawcull(pd, late, 1, draws, meshes, co, c4, dvb, pyr, threads=T)
            oracle.tasksubmit(c4, co)
            ncmd = int(c4[1]) * 64
            cib, cc4 = np.zeros(min(ncmd * 64, L.CLUSTER_LIMIT) + 256, np.uint32), np.zeros(4, np.uint32)
            oracle.clustercull(cd, late, co, c4, draws, meshlets, mvb, pyr, cib, cc4, threads=T)
            oracle.clustersubmit(cc4, cib)
            nv = (min(int(cc4[0]), L.CLUSTER_LIMIT) + 255) // 256 * 256
            rec[phase] = dict(count4=c4, commands=co[:ncmd].copy(), cc4=cc4, cib=cib[:nv].copy(), dvb=dvb.copy(), mvb=mvb.copy(),
                              tested=int(co[:min(int(c4[0]), L.TASK_WGLIMIT)]["taskCount"].sum()))
        rec["pyramid"] = pyr.data.copy()
        if prev is not None and all(prev[ph][k].tobytes() == rec[ph][k].tobytes() for ph in ("early", "late") for k in ("count4", "commands", "cc4", "cib", "dvb", "mvb")):
            return rec, f + 1
        prev = rec
    return prev, frames


def config_frame(ctx, iters, n_draws=1_000_000, lod0=2200, size=4096, copies=3, fused=True, cpp_driver=False):
    """VERDICT r2 item 2 — the drop-in number: niagara's dependent frame at BASELINE scale, one frame after the other on one
    stream (src/niagara.cpp:1765-1788):

        drawcull<0,TASK> -> tasksubmit -> clustercull<0> -> clustersubmit -> depthreduce (4096^2 -> 2048^2 x 12)
        -> drawcull<1,TASK> -> tasksubmit -> clustercull<1> -> clustersubmit

    1 M draws over 64 meshes x 4 LODs (LOD 0 = `lod0` meshlets: ~10 M meshlets tested per cluster pass), every culling flag on,
    frame N's late visibility (drawVisibility, meshletVisibility) feeding frame N + 1's early pass.  `copies` complete scene copies
    (draws + mirror, visibility buffers, depth target, pyramid) are rotated so that no frame finds its 48 MB of draws, its
    visibility words or its depth target in the 256 MiB Infinity Cache.  fused = NV_OPT_FUSED_COUNT_RESET + NV_OPT_FUSED_SUBMIT
    (11 launches per frame); otherwise the reference's dispatch sequence one to one (19 launches)."""
    dev = ctx.device
    meshes, meshlets, draws, slots, depth_h, cd = frame_scene(n_draws, lod0, size)
    ctx.set_option(P.NV_OPT_FUSED_COUNT_RESET, int(fused))
    ctx.set_option(P.NV_OPT_FUSED_SUBMIT, int(fused))
    ctx.reserve(n_draws, L.TASK_WGLIMIT)
    mb, mlb = P.to_device(meshes, dev), P.to_device(meshlets, dev)
    ctx.upload_meshes(mb, len(meshes))
    ctx.upload_meshlets(mlb, len(meshlets))
    one = P.to_device(draws, dev)
    db_all = torch.cat([one] * copies)
    dbs = [db_all[c * one.numel():(c + 1) * one.numel()] for c in range(copies)]
    ctx.upload_draws(db_all, copies * n_draws, mb)
    mvb_words = (slots + 31) // 32 + 2
    dvbs = [torch.zeros(n_draws, dtype=torch.int32, device=dev) for _ in range(copies)]
    mvbs = [torch.zeros(mvb_words, dtype=torch.int32, device=dev) for _ in range(copies)]
    depths = [torch.from_numpy(depth_h).to(dev) for _ in range(copies)]
    pyrs = [P.DepthPyramid(dev, size, size) for _ in range(copies)]
    dcb = torch.zeros((L.TASK_WGLIMIT + 64) * L.TASKCMD.itemsize, dtype=torch.uint8, device=dev)
    dccb = torch.zeros(4, dtype=torch.int32, device=dev)
    cib = torch.zeros(L.CLUSTER_LIMIT + 256, dtype=torch.int32, device=dev)
    ccb = torch.zeros(4, dtype=torch.int32, device=dev)
    pd = cd.copy()
    pd["clusterBackfaceEnabled"] = 1  # cull(): src/niagara.cpp:1549, postPass 0
    frames_of = [0] * copies

    def phase(c, late):
        if not fused:
            ctx.reset_count(dccb)
        ctx.drawcull(pd, late, 1, dbs[c], mb, dcb, dccb, dvbs[c], pyrs[c].desc)
        if not fused:
            ctx.tasksubmit(dccb, dcb)
            ctx.reset_count(ccb)
        ctx.clustercull(cd, late, dcb, dccb, dbs[c], mlb, mvbs[c], pyrs[c].desc, cib, ccb)
        if not fused:
            ctx.clustersubmit(ccb, cib)

    def frame(i):
        c = i % copies
        phase(c, 0)
        ctx.depthreduce(depths[c], size, size, pyrs[c].desc)
        phase(c, 1)
        frames_of[c] += 1

    k = 0
    for _ in range(2 * copies):  # every copy reaches the steady state (frame 0 establishes the visible set, frame 1 is the first real one)
        frame(k)
        k += 1
    torch.cuda.synchronize()
    frame_us = None
    for _ in range(3):  # (the shortest of three timed loops, as plain_wall: a loop of 30 frames is 6 ms of wall time and one scheduling hiccup of the launching thread is a tenth of it)
        t0 = time.perf_counter()
        for _ in range(iters):
            frame(k)
            k += 1
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / iters * 1e6
        frame_us = t if frame_us is None or t < frame_us else frame_us

    # per-pass breakdown: the library's event pairs, read (and thereby synchronised) after every phase
    names = ("early_drawcull", "early_cluster_cull", "early_cluster_scatter", "pyramid", "late_drawcull", "late_cluster_cull", "late_cluster_hiz", "late_cluster_scatter")
    acc = dict.fromkeys(names, 0.0)
    ctx.profile(True)
    ctx.profile_variants()  # (reset)
    n_b = max(3, min(iters, 10))
    for _ in range(n_b):
        c = k % copies
        phase(c, 0)
        pr = ctx.profile_read()
        acc["early_drawcull"] += pr["drawcull"][0]
        acc["early_cluster_cull"] += pr["cluster_cull"][0]
        acc["early_cluster_scatter"] += pr["cluster_scatter"][0]
        ctx.depthreduce(depths[c], size, size, pyrs[c].desc)
        acc["pyramid"] += ctx.profile_read()["depthreduce"][0]
        phase(c, 1)
        pr = ctx.profile_read()
        acc["late_drawcull"] += pr["drawcull"][0]
        acc["late_cluster_cull"] += pr["cluster_cull"][0]
        acc["late_cluster_hiz"] += pr["cluster_hiz"][0]
        acc["late_cluster_scatter"] += pr["cluster_scatter"][0]
        frames_of[c] += 1
        k += 1
    ctx.profile(False)
    kernel_variants = ctx.profile_variants()  # which forms the host's per-launch choices resolved to in the frames of the breakdown
    breakdown = {n + "_us": acc[n] / n_b * 1e3 for n in names}
    hiz_phases = hiz_phase_sums(ctx) if int(os.environ.get("NV_DEBUG_MODE", "0")) & 268435456 else None  # experiments build, bit 28

    # ---- parity: one more frame on the next copy with every buffer of both phases read back, against the oracle after the same
    # number of frames on that copy (errors of the timed frames would sit in its visibility state)
    c = k % copies
    got = {}
    for name, late in (("early", 0), ("late", 1)):
        if late:
            ctx.depthreduce(depths[c], size, size, pyrs[c].desc)
        phase(c, late)
        if fused:  # the fused passes leave what the submit launches would; nothing else to run
            pass
        torch.cuda.synchronize()
        c4, cc4 = dccb.cpu().numpy().view(np.uint32).copy(), ccb.cpu().numpy().view(np.uint32).copy()
        ncmd = int(c4[1]) * 64
        nv = (min(int(cc4[0]), L.CLUSTER_LIMIT) + 255) // 256 * 256
        got[name] = dict(count4=c4, commands=P.from_device(dcb, L.TASKCMD)[:ncmd].copy(), cc4=cc4, cib=cib[:nv].cpu().numpy().view(np.uint32).copy(),
                         dvb=dvbs[c].cpu().numpy().view(np.uint32).copy(), mvb=mvbs[c].cpu().numpy().view(np.uint32).copy())
    got["pyramid"] = pyrs[c].data.cpu().numpy()
    frames_of[c] += 1
    want, simulated = oracle_frames(meshes, meshlets, draws, slots, depth_h, cd, size, frames_of[c])
    same = got["pyramid"].tobytes() == want["pyramid"].tobytes()
    for ph in ("early", "late"):
        for key in ("count4", "commands", "cc4", "cib", "dvb", "mvb"):
            same = same and got[ph][key].tobytes() == want[ph][key].tobytes()

    # ---- algorithmic bytes of the frame (what the reference's shaders read and write, SURVEY.md §8d per pass): early drawcull
    # reads postPass + the visibility word of every draw and the rest of the record only for last frame's visible draws; the late one
    # reads every record and rewrites every visibility word; a cluster pass reads 12 cull bytes + 1 visibility bit per meshlet and
    # command + draw per command (the late pass also writes the bit); the pyramid reads the depth target once and writes its levels.
    e, l = want["early"], want["late"]
    vis_prev = int(e["dvb"].sum())  # (the early pass leaves drawVisibility alone: these are last frame's visible draws)

    def cluster_bytes(r, late):
        m, ncmd = r["tested"], int(r["count4"][0])
        return m * 12 + ncmd * 68 + m // 8 * (2 if late else 1) + int(r["cc4"][0]) * 4 + 4

    bytes_ = dict(early_drawcull=n_draws * 8 + vis_prev * 44 + int(e["count4"][0]) * 20 + 4 + 208 * len(meshes), early_cluster=cluster_bytes(e, 0),
                  pyramid=4 * size * size + pyrs[0].desc.totalTexels * 4,
                  late_drawcull=n_draws * 56 + int(l["count4"][0]) * 20 + 4 + 208 * len(meshes), late_cluster=cluster_bytes(l, 1))
    algo = sum(bytes_.values())
    tested = e["tested"] + l["tested"]
    out = dict(config="frame: 1M draws, early cull -> pyramid -> late cull at BASELINE scale" + (" (fused: 11 launches)" if fused else " (reference dispatch sequence: 19 launches)"),
               draws=n_draws, lod0_meshlets=lod0, depth=size, scene_copies_rotated=copies, frames_timed=iters, frame_us=frame_us, **breakdown,
               sum_of_kernels_us=sum(breakdown.values()), kernel_variants=kernel_variants, **({"hiz_phases": hiz_phases} if hiz_phases else {}),
               early=dict(task_commands=int(e["count4"][0]), meshlets_tested=e["tested"], visible=int(e["cc4"][0])),
               late=dict(task_commands=int(l["count4"][0]), meshlets_tested=l["tested"], visible=int(l["cc4"][0]), draws_visible=int(l["dvb"].sum())),
               algorithmic_bytes=algo, algorithmic_bytes_by_pass=bytes_, achieved_GBs=algo / frame_us / 1e3, frac=algo / frame_us / 1e3 / HBM,
               meshlets_tested_per_frame=tested, meshlets_per_s=tested / (frame_us * 1e-6), draws_per_s=2 * n_draws / (frame_us * 1e-6),
               frames_per_s=1e6 / frame_us, oracle_frames_simulated=simulated, frames_on_checked_copy=frames_of[c], parity=verdict(same))
    out["roofline_valu"] = {"cluster_hiz_kernel": valu_roofline("cluster_hiz_kernel", breakdown["late_cluster_hiz_us"]),
                            "late cull launch (direct form's packed walk, frustum / cone ballots)": valu_roofline("cluster_mask_kernel<false, true, false, 8, true, true, true>", breakdown["late_cluster_cull_us"])}
    if cpp_driver:
        out["cpp_driver"] = frame_driver_timed(meshes, meshlets, draws, slots, depth_h, cd, size, fused, iters)
    ctx.set_option(P.NV_OPT_FUSED_COUNT_RESET, 0)
    ctx.set_option(P.NV_OPT_FUSED_SUBMIT, 0)
    ctx.upload_draws(None, 0)
    return out


def frame_driver_timed(meshes, meshlets, draws, slots, depth, cd, size, fused, iters):
    """the same scene through examples/frame_driver (C++ on the C ABI, no Python or torch in that process) in its timed mode (one
    scene copy, not rotated; frame 0 reduces a cleared depth target); the frame it records after the timed ones is held against the
    oracle after the same history"""
    import struct
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        scene, outp = os.path.join(tmp, "scene.bin"), os.path.join(tmp, "out.bin")
        with open(scene, "wb") as f:
            f.write(struct.pack("<6I", 0x4353564E, len(meshes), len(meshlets), len(draws), size, size))
            f.write(cd.tobytes())
            for a in (meshes, meshlets, draws):
                f.write(np.ascontiguousarray(a).tobytes())
            f.write(np.ascontiguousarray(depth, dtype=np.float32).tobytes())
        p = subprocess.run([os.path.join(root, "examples", "frame_driver"), scene, outp, "0"] + (["fused"] if fused else []) + ["time", str(iters)],
                           capture_output=True, text=True, timeout=1800)
        if p.returncode != 0:
            raise SystemExit("frame_driver failed: " + p.stderr[-2000:])
        line = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][0])
        blob = open(outp, "rb").read()
    want, _ = oracle_frames(meshes, meshlets, draws, slots, depth, cd, size, 3 + iters + 1, first_cleared=True)
    # records of the one frame written after the timed ones: early {count4, commands, cc4, cib, dvb, mvb}, pyramid, late {...}
    pos, recs = 0, []
    while pos < len(blob):
        tag, nbytes = struct.unpack_from("<2I", blob, pos)
        recs.append((tag, blob[pos + 8:pos + 8 + nbytes]))
        pos += 8 + nbytes
    keys = ("count4", "commands", "cc4", "cib", "dvb", "mvb")
    expect = [want["early"][k].tobytes() for k in keys] + [want["pyramid"].tobytes()] + [want["late"][k].tobytes() for k in keys]
    same = len(recs) == len(expect) and all(r[1] == w for r, w in zip(recs, expect))
    return dict(frame_us=line["frame_us"], frames=line["frames"], parity=verdict(same))


def config_n4(ctx, iters, n_draws=2048, cpd=1):
    """the mesh stage's triangle cull (meshlet.mesh.glsl with MESH_CULL = 1) over a cluster list that
    names every meshlet of a 131 k-meshlet pool once (the list a fully visible scene would produce)"""
    dev = ctx.device
    draws, meshlets, commands, n = synth.cluster_scene(n_draws, cpd, seed=9, scene_radius=60.0)
    data, vertices = synth.make_geometry(meshlets, seed=11)
    cd = host.build_cull_data(draw_count=n_draws, viewport=(1920, 1080), cullingEnabled=1)
    g = synth.make_globals(cd, (1920, 1080))
    m = n * 64
    ids = (np.arange(m, dtype=np.uint32) // 64) | ((np.arange(m, dtype=np.uint32) % 64) << 24)
    cib = torch.from_numpy(np.concatenate([ids, np.zeros(512, np.uint32)]).view(np.int32)).to(dev)
    ccb = torch.from_numpy(np.array([m, 0, 0, 0], np.uint32).view(np.int32)).to(dev)
    ctx.clustersubmit(ccb, cib)
    db, mlb, dcb = P.to_device(draws, dev), P.to_device(meshlets, dev), P.to_device(commands, dev)
    dd = torch.from_numpy(data.view(np.int32)).to(dev)
    vb = P.to_device(vertices, dev)
    slots = (m + 255) // 256 * 256
    masks = torch.zeros(slots * 16, dtype=torch.uint8, device=dev)
    totals = torch.zeros(3, dtype=torch.int64, device=dev)

    def step(i):
        ctx.trianglecull(g, dcb, db, mlb, dd, vb, cib, ccb, masks, slots, totals)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    totals.zero_()
    e0.record()
    for i in range(iters):
        step(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    vc, tc = meshlets["vertexCount"].astype(np.int64), meshlets["triangleCount"].astype(np.int64)
    refs = np.where(meshlets["shortRefs"] == 1, 2, 4) * vc
    algo = int((4 + 20 + 48 + 24 + 16) * m + refs.sum() + 3 * tc.sum() + 8 * vc.sum())
    t = totals.cpu().numpy() // iters
    # parity: every slot's keep mask and the totals against the oracle
    mo, to = np.zeros(slots, dtype=L.TRIMASK), np.zeros(3, np.uint64)
    cib_h = cib.cpu().numpy().view(np.uint32)
    oracle.trianglecull(g, commands, draws, meshlets, data, vertices, cib_h, ccb.cpu().numpy().view(np.uint32), mo, to)
    same = masks.cpu().numpy().tobytes() == mo.tobytes() and [int(x) for x in t] == [int(x) for x in to]
    return dict(config="N4: mesh-stage triangle cull, %d clusters" % m, clusters=int(t[0]), triangles=int(t[1]), kept=int(t[2]), call_us=us,
                algorithmic_bytes=algo, achieved_GBs=algo / us / 1e3, frac=algo / us / 1e3 / HBM, triangles_per_s=int(t[1]) / (us * 1e-6), parity=verdict(same),
                roofline_valu={"trianglecull_kernel": valu_roofline("trianglecull_kernel", us)})


def cluster_config(ctx, iters, label, n_draws=156250, cpd=10, aos=False, scene_radius=300.0, backface=1, copies=1, cam_pos=(0, 0, 0)):
    """clustercull<0> over a pre-built command list (config 3A's shape): the roofline size (x10), the AoS-in-place read,
    and the dense-visibility variants (camera inside the cloud: most commands reach pass B)"""
    dev = ctx.device
    draws, meshlets, commands, n = synth.cluster_scene(n_draws, cpd, scene_radius=scene_radius)
    cd = host.build_cull_data(cam_pos=cam_pos, draw_count=n_draws, cullingEnabled=1, clusterBackfaceEnabled=backface)
    m = n * 64
    db, dcb = P.to_device(draws, dev), P.to_device(commands, dev)
    mlbs = [P.to_device(meshlets, dev) for _ in range(copies)]  # rotated: cache-cold passes when one copy fits the Infinity Cache
    mlb = torch.cat(mlbs) if copies > 1 else mlbs[0]
    del mlbs
    dcbs = [P.to_device(synth.make_task_commands(n_draws, cpd, meshlet_base=c * m), dev) for c in range(copies)]
    if not aos:
        ctx.upload_meshlets(mlb, copies * m)
    dccb = torch.from_numpy(synth.count4_for(n).view(np.int32).copy()).to(dev)
    cib = torch.zeros(min(m, L.CLUSTER_LIMIT) + 256, dtype=torch.int32, device=dev)
    ccb = torch.zeros(4, dtype=torch.int32, device=dev)

    def step(i):
        ctx.reset_count(ccb)
        ctx.clustercull(cd, 0, dcbs[i % copies], dccb, db, mlb, None, None, cib, ccb)

    wall, k_us, prof = timed(ctx, step, iters, "cluster_cull")
    total = int(ccb[0].item())
    # parity: the whole visible list of the last timed pass against the multithreaded oracle
    cib_o, cc4_o = np.zeros(m, np.uint32), np.zeros(4, np.uint32)
    oracle.clustercull(cd, 0, commands, synth.count4_for(n), draws, meshlets, None, None, cib_o, cc4_o, threads=oracle.max_threads())
    ids = cib[:min(total, L.CLUSTER_LIMIT)].cpu().numpy().view(np.uint32)
    same = total == int(cc4_o[0]) and (ids == cib_o[:min(total, L.CLUSTER_LIMIT)]).all()
    # commands that reach pass B = commands with a meshlet the frustum test keeps or the filter cannot exclude; reported as
    # the share of commands with at least one visible-or-cone-culled meshlet is not observable from outside, so: share of
    # commands with a survivor (lower bound of the candidates)
    with_survivor = len(np.unique(cib_o[:int(cc4_o[0])] & 0xffffff)) / n
    algo = m * (24 if aos else 12) + n * 76
    scat = prof["cluster_scatter"][0] / max(1, prof["cluster_scatter"][1]) * 1e3
    return dict(config=label, meshlets=m, visible=total, commands_with_survivors=round(with_survivor, 4), cull_us=k_us, scatter_us=scat, step_us=wall, step_us_with_events=prof["wall_with_events_us"],
                algorithmic_bytes=algo, achieved_GBs=algo / k_us / 1e3, frac=algo / k_us / 1e3 / HBM, pass_frac=(algo - n * 8 + total * 4) / (k_us + scat) / 1e3 / HBM,
                meshlets_per_s=m / (wall * 1e-6), parity=verdict(same),
                **({"roofline_valu": {"cull launch (direct form's packed walk)": valu_roofline("cluster_mask_kernel<false, true, false, 8, true, false, true>", k_us)}} if with_survivor > 0.5 and not aos else {}))


def config_task(ctx, iters, n_draws=15625, cpd=10, late=0, copies=4):
    """the task-shader form of the cluster cull (meshlet.task.glsl:53-149, nv_taskcull): per command a compacted 64-entry payload
    + count instead of the global ordered list; config 3A's 10 M meshlets, `copies` meshlet pools rotated (cache-cold)"""
    dev = ctx.device
    draws, meshlets, commands, n = synth.cluster_scene(n_draws, cpd)
    cd = host.build_cull_data(draw_count=n_draws, cullingEnabled=1, clusterBackfaceEnabled=1)
    m = n * 64
    db = P.to_device(draws, dev)
    mlb = torch.cat([P.to_device(meshlets, dev) for _ in range(copies)])
    dcbs = [P.to_device(synth.make_task_commands(n_draws, cpd, meshlet_base=c * m), dev) for c in range(copies)]
    ctx.upload_meshlets(mlb, copies * m)
    dccb = torch.from_numpy(synth.count4_for(n).view(np.int32).copy()).to(dev)
    ncmd = len(commands)
    payloads = torch.zeros(ncmd * 64, dtype=torch.int32, device=dev)
    counts = torch.zeros(ncmd, dtype=torch.int32, device=dev)

    def step(i):
        ctx.taskcull(cd, late, dcbs[i % copies], dccb, db, mlb, None, None, payloads, counts)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        step(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    po, co = np.zeros(ncmd * 64, np.uint32), np.zeros(ncmd, np.uint32)
    oracle.taskcull(cd, late, commands, synth.count4_for(n), draws, meshlets, None, None, po, co)
    cg, pg = counts.cpu().numpy().view(np.uint32), payloads.cpu().numpy().view(np.uint32).reshape(ncmd, 64)
    same = (cg == co).all() and all((pg[i, :co[i]] == po.reshape(ncmd, 64)[i, :co[i]]).all() for i in np.nonzero(co)[0][:20000])
    algo = m * 12 + n * 68 + int(co.sum()) * 4 + ncmd * 4
    return dict(config="T: task-shader form (nv_taskcull), %d meshlets" % m, input_copies_rotated=copies, visible=int(co.sum()), call_us=us, algorithmic_bytes=algo,
                achieved_GBs=algo / us / 1e3, frac=algo / us / 1e3 / HBM, meshlets_per_s=m / (us * 1e-6), parity=verdict(same))


ALLOW_MISMATCH = False  # --allow-mismatch: timing of experiments-build debug modes that switch work off (their lines say "DIFFERENT")


def verdict(same):
    if not same:
        if ALLOW_MISMATCH:
            return "DIFFERENT (--allow-mismatch: a debug mode that skips work; timing only)"
        raise SystemExit("parity FAILURE against the CPU oracle")
    return "bit-identical"


# NV_BENCH_CULL_FORM=n (A/B runs only): every context this module — or bench.contract_chain through it — creates pins NV_OPT_CULL_FORM to n
# NV_BENCH_TASK_EMIT=n likewise NV_OPT_TASK_EMIT (1 = per draw, 2 = the list form)
if os.environ.get("NV_BENCH_CULL_FORM") or os.environ.get("NV_BENCH_TASK_EMIT"):
    class _PinnedContext(P.Context):
        def __init__(self, *args, **kw):
            super().__init__(*args, **kw)
            if os.environ.get("NV_BENCH_CULL_FORM"):
                self.set_option(P.NV_OPT_CULL_FORM, int(os.environ["NV_BENCH_CULL_FORM"]))
            if os.environ.get("NV_BENCH_TASK_EMIT"):
                self.set_option(P.NV_OPT_TASK_EMIT, int(os.environ["NV_BENCH_TASK_EMIT"]))
    P.Context = _PinnedContext


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--allow-mismatch", action="store_true", help="report a time even if the output differs from the oracle (NV_DEBUG_MODE timing experiments only)")
    a = ap.parse_args()
    ALLOW_MISMATCH = a.allow_mismatch
    ctx = P.Context(0)
    runs = {"2": lambda: config2(ctx, a.iters), "2_fused": lambda: config2(P.Context(0), a.iters, fused_reset=True), "2_aos": lambda: config2(P.Context(0), a.iters, soa=False), "2l": lambda: config2_late(ctx, a.iters), "3b": lambda: config3b(ctx, a.iters), "3b_fused": lambda: config3b(P.Context(0), a.iters, fused=True),
            # (3b_chain = bench.py's `contract_chain`: BASELINE configs[2] with LOD select at 10 M meshlets)
            "3b_chain": lambda: config3b(P.Context(0), a.iters, n_draws=125000, fused=True),
            "4": lambda: config4(ctx, a.iters),
            "4b": lambda: config4(ctx, a.iters, size=1024), "n4": lambda: config_n4(ctx, a.iters),
            "frame": lambda: config_frame(P.Context(0), a.iters, cpp_driver=True), "frame_py": lambda: config_frame(P.Context(0), a.iters), "frame_contract": lambda: config_frame(P.Context(0), a.iters, fused=False),
            "task": lambda: config_task(ctx, a.iters),
            "big": lambda: cluster_config(ctx, max(5, a.iters // 3), "3A x10 (SoA mirror)"),
            "big_aos": lambda: cluster_config(P.Context(0), max(5, a.iters // 3), "3A x10 (AoS in place)", aos=True),
            # dense visibility: the same 10 M meshlets in a cloud of radius 40 seen from z = +60 (87 % of the
            # commands have survivors) or +30 (53 %), with and without the cone test (the reference ships
            # clusterBackfaceEnabled = 0 on this path, src/niagara.cpp:1595-1596)
            "3a_dense": lambda: cluster_config(ctx, a.iters, "3A dense: 10 M meshlets, radius 40, camera z=60", 15625, 10, scene_radius=40.0, copies=4, cam_pos=(0, 0, 60)),
            "3a_dense_nocone": lambda: cluster_config(ctx, a.iters, "3A dense, cone off", 15625, 10, scene_radius=40.0, backface=0, copies=4, cam_pos=(0, 0, 60)),
            "3a_half": lambda: cluster_config(ctx, a.iters, "3A half dense: radius 40, camera z=30", 15625, 10, scene_radius=40.0, copies=4, cam_pos=(0, 0, 30)),
            "3a": lambda: cluster_config(ctx, a.iters, "3A: 10 M meshlets, scene radius 300 (bench.py's workload)", 15625, 10, copies=4)}
    for k, fn in runs.items():
        if a.only and k not in a.only.split(","):
            continue
        print(json.dumps(fn()), flush=True)
    ctx.status()
    ctx.close()
