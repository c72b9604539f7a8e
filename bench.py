#!/usr/bin/env python3
"""bench.py — meshlets culled+compacted per second (BASELINE.json metric) on N MI355X GPUs of one node.

Workload (config.workload = "config3A"): BASELINE.json configs[2] — synthetic 10 M meshlets (bistro scale), i.e.
156 250 full MeshTaskCommands over 15 625 draws, per-meshlet cone + frustum cull (`clustercull`, LATE = 0,
clusterBackfaceEnabled = 1) with ordered compaction of the visible IDs.  One "step" = one pass of the hot path over
one batch: reset of the count word (the caller's vkCmdFillBuffer, src/niagara.cpp:1586; fused into the pass through
NV_OPT_FUSED_COUNT_RESET unless --explicit-reset) + nv_clustercull, with inputs resident in HBM.  Steps rotate over
`--copies` distinct input sets so that every pass streams from HBM rather than from the 256 MiB Infinity Cache.

ONE regime: `value` / `ms_per_step` = one pass after the other on ONE stream — what a frame's
dependent cull pass costs end to end — and `roofline` times the dominant kernel in that same mode (HIP events on the launch
stream, a second loop of the same passes).  What several independent passes in flight reach (three views on three
streams, contexts sharing one scene mirror) is reported next to it as `throughput_overlapped`, never as `value`.

N > 1: the pool shards by contiguous command ranges and the only collective is the all-reduce of the passes' visible
counts (RCCL; the rows of `--counts-batch` passes, written by the scatter launches, share one asynchronous all-reduce).
Default = weak scaling (10 M meshlets per GPU); `--total-meshlets T` = strong scaling (the T-meshlet pool split over the
ranks; `--gpus 8 --total-meshlets 100000000` is BASELINE config 5).

Prints ONE JSON line (rank 0):  metric/value/unit as in BASELINE.json + "roofline" (dominant kernel) + "cpu_baseline"
(the CPU oracle timed on this host's cores; a reported baseline, not a target) + "parity" (every rank's visible-ID list of the
benchmarked pass against the CPU oracle on its shard, outside the timed region; a difference on any rank is a non-zero exit).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--draws", type=int, default=15625, help="weak scaling: draws per GPU (x commands-per-draw x 64 = meshlets per GPU)")
    ap.add_argument("--commands-per-draw", type=int, default=10)
    ap.add_argument("--total-meshlets", type=int, default=0,
                    help="strong scaling: the size of the WHOLE pool, split over the ranks by contiguous command ranges (100000000 with --gpus 8 = BASELINE config 5)")
    ap.add_argument("--copies", type=int, default=4, help="distinct input sets rotated through (cache-cold passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    ap.add_argument("--aos", action="store_true", help="read the 24-B AoS meshlets in place (no SoA mirror)")
    ap.add_argument("--counts-batch", type=int, default=8, help="N > 1: passes whose counts share one all-reduce (1 = one collective per pass)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only for functional tests)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="take every N > 1 branch also with ONE rank (process group, warm-up all-reduce, the batched asynchronous counts all-reduce, barrier, "
                         "MAX over the ranks, per-rank oracle check + flag all-reduce): the sharded path over RCCL on a one-GPU box (VERDICT r4 item 2)")
    ap.add_argument("--shared-device", action="store_true", help="functional test only: all ranks use cuda:0 (needs --backend gloo)")
    ap.add_argument("--streams", type=int, default=1,
                    help="streams of the timed region: 1 (default) = strictly one pass after the other; > 1 issues the steps round-robin on that many HIP streams "
                         "(independent passes in flight) and says so in the line")
    ap.add_argument("--overlap-streams", type=int, default=3, help="streams of the `throughput_overlapped` side leg (N = 1 only; 0 = skip it)")
    ap.add_argument("--scatter-waves", type=int, default=0, help="NV_OPT_SCATTER_WAVES (4, 8 or 16) of the timed region; 0 = 16 with one stream, 8 with several")
    ap.add_argument("--no-contract-chain", action="store_true", help="skip the `contract_chain` side field (N = 1 with the CPU baseline only)")
    ap.add_argument("--no-frame", action="store_true", help="skip the `frame` side field (N = 1 with the CPU baseline only)")
    ap.add_argument("--frame-iters", type=int, default=30, help="frames timed for the `frame` side field")
    ap.add_argument("--dump-ids", default="", help="directory: every rank saves the visible-ID list of its last profiled pass, rebased to pool-wide command "
                                                   "indices (shard.to_global_ids), as ids_<rank>.npy — tests/test_distributed_gpu.py concatenates them")
    ap.add_argument("--explicit-reset", action="store_true",
                    help="reference contract: zero the count word with a separate launch (nv_reset_count) instead of NV_OPT_FUSED_COUNT_RESET")
    return ap.parse_args()


def make_commands(b, e, cpd, meshlet_base=0):
    """full task commands (taskCount 64) for the global command range [b, e) of the pool: command g belongs to draw g // cpd;
    drawId is local to the rank's draw slice (which starts at draw b // cpd), meshlets and visibility slots are the rank's own,
    packed from 0.  Padded with zeroed dummy commands to a multiple of 64 like tasksubmit leaves it."""
    from niagara_amd import layouts as L
    n = e - b
    c = np.zeros((n + 63) // 64 * 64, dtype=L.TASKCMD)
    k = np.arange(n, dtype=np.uint32)
    c["drawId"][:n] = (k + np.uint32(b)) // cpd - np.uint32(b // cpd)
    c["taskOffset"][:n] = k * 64 + meshlet_base
    c["taskCount"][:n] = 64
    c["meshletVisibilityOffset"][:n] = k * 64
    return c


def make_inputs(args, rank, world):
    """the rank's shard of the synthetic pool: (draws, meshlets, CullData, command range).  The pool's commands shard by
    contiguous ranges (shard.command_range); a rank's draws are the matching slice of niagara's draw generator
    (src/niagara.cpp:978-997) — a draw whose commands straddle a shard boundary is present on both sides — and its meshlets
    come from a per-rank seed."""
    from niagara_amd import host, shard, synth
    cpd = args.commands_per_draw
    if args.total_meshlets:
        total_cmd = args.total_meshlets // 64
    else:
        total_cmd = args.draws * cpd * world
    b, e = shard.command_range(total_cmd, rank, world)
    d0, d1 = b // cpd, (e + cpd - 1) // cpd
    total_draws = (total_cmd + cpd - 1) // cpd
    draws = host.synth_draws(total_draws, 1, 300.0)[d0:d1].copy()
    draws["meshletVisibilityOffset"] = np.arange(d1 - d0, dtype=np.uint32) * (cpd * 64)
    meshlets = synth.make_meshlets((e - b) * 64, seed=2 + rank)
    cd = host.build_cull_data(draw_count=d1 - d0, cullingEnabled=1, clusterBackfaceEnabled=1)
    return draws, meshlets, cd, (b, e), total_cmd


def self_launch_command(gpus, argv, port=None):
    """the command `python bench.py --gpus N ...` re-executes itself as when no launcher started it (WORLD_SIZE unset): the task contract's
    own launcher line — one rank per GPU under torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container's hostname may not
    resolve) at a free port — with the caller's arguments unchanged."""
    if port is None:
        import socket
        s0 = socket.socket()
        s0.bind(("127.0.0.1", 0))
        port = s0.getsockname()[1]
        s0.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + list(argv)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's command is `python3 bench.py --gpus N --steps S --warmup W` with no launcher in front.  Start the N
        # ranks here; rank 0 of the child job prints the ONE JSON line on this process's stdout, and its exit status becomes this one's.
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))  # (the per-rank oracle checks share the host; torchrun would set 1)
        raise SystemExit(subprocess.call(self_launch_command(args.gpus, sys.argv[1:]), env=env))
    import torch
    import torch.distributed as dist

    import niagara_amd  # raises if libniagara_vis.so is missing: there is no fallback
    from niagara_amd import shard, synth
    from niagara_amd import layouts as L
    from niagara_amd import pipeline as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if args.shared_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # `sharded`: the N > 1 code path.  --force-sharded takes it with one rank too (under torch.distributed.run --nproc-per-node 1, or
    # stand-alone: the rendezvous variables then default to this host and a free port)
    sharded = world > 1 or args.force_sharded
    if args.force_sharded and "MASTER_ADDR" not in os.environ:
        import socket
        s0 = socket.socket()
        s0.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s0.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        s0.close()
    if sharded:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        # communicator set-up (lazy in RCCL) belongs to neither the warm-up nor the timed steps
        dist.all_reduce(torch.zeros(1, dtype=torch.int64, device=dev))
        torch.cuda.synchronize()

    # ---- inputs: the pool shards by contiguous command ranges (niagara_amd/shard.py)
    cpd = args.commands_per_draw
    copies = max(1, args.copies)
    draws, meshlets, cd, (cmd_b, cmd_e), total_cmd = make_inputs(args, rank, world)
    n_cmd = cmd_e - cmd_b
    n_meshlets = n_cmd * 64
    n_draws = len(draws)
    total_meshlets = total_cmd * 64
    count4 = synth.count4_for(n_cmd)

    S = max(1, args.streams)
    OS = max(0, args.overlap_streams) if not sharded else 0
    ctxs = [P.Context(local_rank) for _ in range(max(S, OS, 1))]
    ctx = ctxs[0]
    for c in ctxs[1:]:
        c.share_scene(ctx)  # ONE SoA mirror for all streams' contexts (scratch stays per context)
    if not args.explicit_reset:
        for c in ctxs:
            c.set_option(P.NV_OPT_FUSED_COUNT_RESET, 1)  # the pass absorbs the caller's vkCmdFillBuffer(ccb, 0, 4, 0)
    scatter_waves = args.scatter_waves if args.scatter_waves else (8 if S > 1 else 16)
    for c in ctxs:
        c.set_option(P.NV_OPT_SCATTER_WAVES, scatter_waves)
    extra_streams = [torch.cuda.Stream(device=dev) for _ in range(len(ctxs))]
    streams = extra_streams[:S] if S > 1 else [torch.cuda.current_stream(dev)]
    db = P.to_device(draws, dev)
    mlb = torch.empty(copies * n_meshlets * L.MESHLET.itemsize, dtype=torch.uint8, device=dev)
    one = torch.from_numpy(meshlets.view(np.uint8).reshape(-1))
    for c in range(copies):
        mlb[c * one.numel():(c + 1) * one.numel()].copy_(one)
    dcbs = [P.to_device(make_commands(cmd_b, cmd_e, cpd, meshlet_base=c * n_meshlets), dev) for c in range(copies)]
    dccb = torch.from_numpy(count4.view(np.int32).copy()).to(dev)
    cibs = [torch.zeros(min(n_meshlets, L.CLUSTER_LIMIT) + 256, dtype=torch.int32, device=dev) for _ in ctxs]
    ccbs = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in ctxs]
    cib, ccb = cibs[0], ccbs[0]
    # N > 1: the passes' counts are summed over the ranks; batched, asynchronous, written by the scatter launch (shard.CountsReducer,
    # one per stream: a stream's reducer sees that stream's passes)
    B = max(1, args.counts_batch)
    reds = [shard.CountsReducer(ctxs[s], dev, B, stream=streams[s] if S > 1 else None, force_collective=args.force_sharded) for s in range(S)]
    if not args.aos:
        ctx.upload_meshlets(mlb, copies * n_meshlets)
    torch.cuda.synchronize()
    # every argument of a pass marshalled once (Context.bind_clustercull): at ~25-30 us per step the per-call marshalling of the
    # Python layer (stream lookup, eight data_ptr() calls, a stream context) is what the host cannot afford, not the launch
    calls = [[ctxs[s].bind_clustercull(streams[s] if S > 1 else None, cd, 0, dcbs[c], dccb, db, mlb, None, None, cibs[s], ccbs[s]) for c in range(copies)]
             for s in range(S)]

    def step(i):
        """pass i of the run = pass i // S of stream i % S"""
        s = i % S
        if args.explicit_reset:  # the caller's vkCmdFillBuffer(ccb, 0, 4, 0) as its own launch, on the pass's stream
            with torch.cuda.stream(streams[s]):
                ctxs[s].reset_count(ccbs[s])
        reds[s].before_pass(i // S)
        calls[s][i % copies]()
        reds[s].after_pass(i // S)

    def passes_of(s, n):
        return (n - s + S - 1) // S if n > s else 0

    def drain(n):
        for s in range(S):
            reds[s].drain(passes_of(s, n))

    for i in range(args.warmup):
        step(i)
    drain(args.warmup)
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: exactly `steps` passes, no instrumentation.  It ends when this rank's passes AND the reductions of all
    # of their counts have completed; the MAX over the ranks of that time is the job's time (no closing barrier: the MAX
    # all-reduce below does what it would, outside the clock).
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if sharded:
        for s in streams:
            s.synchronize()
    t_passes = time.perf_counter()
    drain(args.steps)  # every pass's reduction completes inside the timed region
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # what the job waited for collectives after its last pass had finished (N = 1: no collective, and no separate wait for the passes either —
    # the closing device synchronisation is the contract's bracket; a stream synchronisation in front of it was one more driver call per region)
    collective_wait = time.perf_counter() - t_passes if sharded else 0.0
    s_last = (args.steps - 1) % S
    last_counts = reds[s_last].last(passes_of(s_last, args.steps))
    visible_by_stream = [int(c[0].item()) for c in ccbs[:S]]
    # the visible-ID list the TIMED loop's last pass left, copied out before the profiled loop below writes the same buffer:
    # this is the list the oracle check further down holds
    timed_visible = visible_by_stream[s_last]
    timed_ids = cibs[s_last][:min(timed_visible, L.CLUSTER_LIMIT)].cpu().numpy().view(np.uint32).copy()

    # ---- roofline leg: the same passes again with the library's HIP events bracketing each kernel on the launch stream
    # (nv_profile_*), one pass after the other on ONE stream — the mode `value` is measured in when --streams is 1.  It is a
    # separate loop because an event record is itself a barrier packet: three records per pass cost ~10 us per pass.
    single = shard.CountsReducer(ctx, dev, B, force_collective=args.force_sharded)
    serial_calls = [ctx.bind_clustercull(None, cd, 0, dcbs[c], dccb, db, mlb, None, None, cib, ccb) for c in range(copies)]

    def serial_step(i):
        if args.explicit_reset:
            ctx.reset_count(ccb)
        single.before_pass(i)
        serial_calls[i % copies]()
        single.after_pass(i)

    n_prof = max(args.steps, 100)  # (a short run still averages the kernels over 100 launches)
    serial = None
    if S > 1:  # the timed region overlapped passes: also report what one pass after the other takes
        for i in range(5):
            serial_step(i)
        single.drain(5)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(n_prof):
            serial_step(i)
        single.drain(n_prof)
        torch.cuda.synchronize()
        serial = (time.perf_counter() - t2) / n_prof
    ctx.profile_variants()  # (reset: the counts below are the profiled loop's)
    ctx.profile(True)
    t1 = time.perf_counter()
    for i in range(n_prof):
        serial_step(i)
    single.drain(n_prof)
    torch.cuda.synchronize()
    profiled = time.perf_counter() - t1
    prof = ctx.profile_read()
    variants = ctx.profile_variants()  # which kernel form the host's per-launch choices resolved to in the timed launches
    ctx.profile(False)
    ctx.status()

    # the profiled loop's last pass ran the same code on the same pool (the rotated copies hold the same records): its list must be the timed loop's,
    # which is the one checked against the oracle below
    profiled_visible = int(ccb[0].item())
    profiled_ids = cib[:min(profiled_visible, L.CLUSTER_LIMIT)].cpu().numpy().view(np.uint32)
    if profiled_visible != timed_visible or profiled_ids.tobytes() != timed_ids.tobytes():
        raise SystemExit("rank %d: the profiled loop's visible-ID list differs from the timed loop's (%d against %d IDs)" % (rank, profiled_visible, timed_visible))
    visible, visible_ids = timed_visible, timed_ids
    if args.dump_ids:
        np.save(os.path.join(args.dump_ids, "ids_%d.npy" % rank), shard.to_global_ids(visible_ids, cmd_b))

    # ---- side leg (N = 1): independent passes in flight on `--overlap-streams` streams, contexts sharing one scene mirror, the
    # scatter launch with 8 waves per workgroup.  A throughput figure for multi-view callers; never `value`.
    overlapped = None
    if OS > 1:
        for c in ctxs[:OS]:
            c.set_option(P.NV_OPT_SCATTER_WAVES, 8)
        ocalls = [[ctxs[s].bind_clustercull(extra_streams[s], cd, 0, dcbs[c], dccb, db, mlb, None, None, cibs[s], ccbs[s]) for c in range(copies)] for s in range(OS)]
        for i in range(max(OS * 2, 6)):
            ocalls[i % OS][i % copies]()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for i in range(n_prof):
            ocalls[i % OS][i % copies]()
        torch.cuda.synchronize()
        o_step = (time.perf_counter() - t3) / n_prof
        overlapped = {"value": n_meshlets / o_step, "unit": "meshlets/s", "ms_per_step": o_step * 1e3, "streams": OS, "scatter_waves_per_workgroup": 8,
                      "what": "%d independent passes in flight (contexts on %d HIP streams sharing one scene mirror): the scatter launch of one pass overlaps "
                              "the next pass's cull launch; a frame's dependent passes cannot do this" % (OS, OS)}
        for c in ctxs[:OS]:
            c.set_option(P.NV_OPT_SCATTER_WAVES, scatter_waves)

    if sharded:
        t = torch.tensor([elapsed, collective_wait], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, collective_wait = float(t[0].item()), float(t[1].item())
    total_visible = int(last_counts[2].item())

    cull_ms, cull_n = prof["cluster_cull"]
    scat_ms, scat_n = prof["cluster_scatter"]
    kernel_avg_s = max(cull_ms / max(1, cull_n) * 1e-3, 1e-9)      # dominant kernel: cluster_mask_kernel
    scatter_avg_s = scat_ms / max(1, scat_n) * 1e-3

    # algorithmic bytes.  Whole pass: 12 cull bytes per meshlet + (20 + 48) per command + 4 per survivor
    # + 4.  The dominant kernel moves the first two terms plus its 8-byte ballot per command; the survivors' 4 bytes
    # belong to the scatter kernel.
    per_meshlet = 24 if args.aos else 12
    pass_bytes = n_meshlets * per_meshlet + n_cmd * 68 + visible * 4 + 4
    algo_bytes = n_meshlets * per_meshlet + n_cmd * 68 + n_cmd * 8
    achieved = algo_bytes / kernel_avg_s / 1e9
    step_s = elapsed / args.steps
    single_s = step_s if S == 1 else serial

    traffic, traffic_note = pmc_traffic(n_meshlets, args)

    if rank == 0:
        if args.total_meshlets:
            workload = ("config5: %d meshlets sharded over %d GPUs by contiguous command ranges (%d on rank 0)" % (total_meshlets, world, n_meshlets)
                        if world > 1 else "config3A shape at %d meshlets" % total_meshlets)
        else:
            workload = "config3A: %d meshlets/GPU, %d task commands over %d draws" % (n_meshlets, n_cmd, n_draws)
        out = {
            "metric": "meshlets culled+compacted /sec",
            "value": total_meshlets * args.steps / elapsed,
            "unit": "meshlets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.total_meshlets else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload + ", cone+frustum clustercull (LATE=0) + ordered compaction",
                       "meshlets_total": total_meshlets, "meshlets_rank0": n_meshlets, "commands_rank0": n_cmd, "draws_rank0": n_draws,
                       "streams": S, "scatter_waves_per_workgroup": scatter_waves,
                       "regime": "one pass after the other on one stream" if S == 1 else "steps issued round-robin on %d HIP streams (independent passes in flight)" % S,
                       "input_copies_rotated": copies, "count_reset": "explicit launch" if args.explicit_reset else "fused (NV_OPT_FUSED_COUNT_RESET)", "meshlet_layout": "AoS24" if args.aos else "SoA12",
                       "force_sharded": bool(args.force_sharded), "backend": args.backend if sharded else None,
                       "visible_rank0": visible, "visible_per_stream": visible_by_stream, "visible_total": total_visible, "sharding": "commands x%d" % world,
                       "counts_allreduce": ("none (N=1)" if not sharded else "one async all-reduce of [%d, 3] int64 per %d passes, rows written by the scatter launch" % (B, B))},
            "collective_wait_ms": collective_wait * 1e3,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_measured_in_run": False, "traffic_source": traffic_note,
                         "kernel": "cluster_mask_kernel", "kernel_avg_us": kernel_avg_s * 1e6, "timed_in": "one pass after the other on one stream (HIP events on the launch stream)",
                         "algorithmic_bytes": algo_bytes, "launches_timed": cull_n, "kernel_variants": variants,
                         "scatter_kernel_avg_us": scatter_avg_s * 1e6, "ms_per_step_with_events": profiled / n_prof * 1e3,
                         "pass_algorithmic_bytes": pass_bytes,
                         # the whole pass (cull + scatter launches) against the roofline, one pass after the other, un-instrumented
                         "ms_per_pass_single_stream": single_s * 1e3,
                         "pass_frac": pass_bytes / single_s / 1e9 / HBM_PEAK_GBS},
            "throughput_overlapped": overlapped,
            "library": niagara_amd.SO_PATH,
        }
        if S == 1 and kernel_avg_s > step_s * 1.02:
            out["note"] = "inconsistent: the dominant kernel's event time exceeds ms_per_step"

    # ---- parity + CPU baseline, outside every timed region.  Every rank holds its visible-ID list of the last profiled pass against
    # the CPU oracle on ITS shard (same inputs, same command range); the line says "bit-identical" only if all ranks agree.  The timed
    # CPU baseline runs on rank 0's host cores after the other ranks' checks have finished (they share the host): at N = 1 over the
    # whole batch, at N > 1 over rank 0's shard — a bounded sample of the same workload, reported as a rate.
    if not args.no_cpu_baseline:
        ok, detail = True, ""
        if sharded:
            try:  # (a rank that fails INSIDE the check must still reach the all-reduce: the others would wait for it until the rendezvous times out)
                ok, detail = oracle_check(args, cd, draws, meshlets, cmd_b, cmd_e, visible, visible_ids)
            except Exception as exc:  # noqa: BLE001
                ok, detail = False, "oracle check raised %s: %s" % (type(exc).__name__, exc)
        if sharded:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.SUM)
            agreeing = int(flag.item())
            if not ok:
                print("rank %d: %s" % (rank, detail), file=sys.stderr, flush=True)
            dist.barrier()
            if agreeing != world:
                raise SystemExit("parity failure: %d of %d ranks differ from the CPU oracle on their shard" % (world - agreeing, world))
        if rank == 0:
            out["cpu_baseline"] = cpu_baseline(args, cd, draws, meshlets, cmd_b, cmd_e, visible, visible_ids, world)
            out["parity"] = "bit-identical"
            out["parity_checked"] = ("visible-ID list and count of the TIMED loop's last pass (and, identical to it, the profiled loop's) against the CPU oracle: " +
                                     ("the whole batch" if not sharded else "every rank on its own shard, %d of %d ranks agree" % (world, world)))
    elif rank == 0:
        out["parity"] = "not checked (--no-cpu-baseline)"
    # ---- side field (N = 1): BASELINE.json configs[2] as literally worded — "cone + frustum cull + LOD select" — i.e. the reference's
    # chain drawcull<0,1> (LOD select) -> tasksubmit -> clustercull<0> -> clustersubmit (src/niagara.cpp:1530-1611), timed in this
    # process after everything above, with its own parity check of every buffer of the chain.  Never `value`.
    if rank == 0 and not sharded and not args.no_cpu_baseline and not args.no_contract_chain:
        out["contract_chain"] = contract_chain(local_rank)
    # ---- side field (N = 1): niagara's dependent FRAME at BASELINE scale (src/niagara.cpp:1765-1788: early cull -> pyramid -> late cull, 1 M draws, ~10 M
    # meshlets per cluster pass, 4096^2 depth), the production-shaped figure, timed by this process and held against the CPU oracle.  Never `value`.
    if rank == 0 and not sharded and not args.no_cpu_baseline and not args.no_frame:
        for c in ctxs:  # (the main leg's contexts and their scratch are done)
            c.close()
        ctxs = []
        out["frame"] = frame_field(local_rank, args.frame_iters)
    if rank == 0:
        print(json.dumps(out), flush=True)

    for c in ctxs:
        c.close()
    if sharded:
        dist.destroy_process_group()


def frame_field(device_index, iters):
    """tools/bench_configs.py config_frame (fused: 11 launches per frame), three scene copies rotated so that no frame finds its draws, visibility words or
    depth target in the Infinity Cache; every buffer of both phases and the pyramid of one more frame against the CPU oracle after the same history
    (oracle/ only as the checker; a difference is a non-zero exit).  `frac` = SURVEY §8(d)'s algorithmic bytes of the five passes / frame time / 8 TB/s."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs
    from niagara_amd import pipeline as P
    ctx = P.Context(device_index)
    try:
        r = bench_configs.config_frame(ctx, iters)
    finally:
        ctx.close()
    if r["parity"] != "bit-identical":
        raise SystemExit("parity failure in the frame: " + str(r["parity"]))
    keep = ("frame_us", "frac", "achieved_GBs", "algorithmic_bytes", "algorithmic_bytes_by_pass", "sum_of_kernels_us", "kernel_variants", "early", "late", "frames_timed",
            "scene_copies_rotated", "meshlets_tested_per_frame", "meshlets_per_s", "draws_per_s", "oracle_frames_simulated", "parity", "roofline_valu")
    f = {"what": r["config"], "regime": "one frame after the other on one stream, the shortest of three loops of %d frames; per-launch times = the library's HIP event pairs in separate frames" % iters}
    f.update({k: r[k] for k in keep if k in r})
    f["launch_us"] = {k[:-3]: r[k] for k in r if k.endswith("_us") and k not in ("frame_us", "sum_of_kernels_us")}
    f["parity_checked"] = "task commands, count words, visible-ID lists + submit padding, drawVisibility, meshletVisibility of both phases and the pyramid of one frame after the same history"
    return f


def contract_chain(device_index, n_draws=125000, iters=100):
    """config 3B at BASELINE configs[2]'s scale (tools/bench_configs.py config3b, ~10 M meshlets tested after LOD select): drawcull<0,1> over
    `n_draws` draws of 64 meshes x 4 LODs -> tasksubmit -> clustercull<0> -> clustersubmit with the two fusion options (count reset and
    submit words inside the passes: 4 launches per phase), one phase after the other on one stream; every buffer of the chain is held
    against the CPU oracle (a difference is a non-zero exit) — oracle/ only as the checker."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs
    from niagara_amd import pipeline as P
    ctx = P.Context(device_index)
    try:
        r = bench_configs.config3b(ctx, iters, n_draws=n_draws, fused=True)
    finally:
        ctx.close()
    # drawcull<0,1> reads 48 + 4 B per draw and writes 20 B per command; the cluster pass reads 12 B per meshlet tested + (20 + 48) B per
    # command, writes 4 B per survivor + the count word
    N, Cn, M, Vm = r["draws"], r["task_commands"], r["meshlets_tested"], r["visible"]
    b_draw, b_cluster = 52 * N + 20 * Cn, 12 * M + 68 * Cn + 4 * Vm + 4
    b_cull_launch = 12 * M + 68 * Cn + 8 * Cn  # (the cull launch alone: + its 8-byte ballot per command, the survivors' 4 B belong to the scatter launch)
    roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes": b_draw + b_cluster,
            "per_pass_bytes": {"drawcull": b_draw, "cluster": b_cluster}, "achieved": (b_draw + b_cluster) / r["step_us"] / 1e3,
            "frac": (b_draw + b_cluster) / r["step_us"] / 1e3 / HBM_PEAK_GBS,
            "cluster_launches_frac": b_cluster / max(r["cluster_cull_us"] + r["cluster_scatter_us"], 1e-9) / 1e3 / HBM_PEAK_GBS,
            "cull_launch_frac": b_cull_launch / max(r["cluster_cull_us"], 1e-9) / 1e3 / HBM_PEAK_GBS,
            "formulas": "drawcull<0,1>: 52 N + 20 C; cluster pass: 12 M + 68 C + 4 Vm + 4 (N draws, C commands, M meshlets tested, Vm visible)"}
    if r["parity"] != "bit-identical":
        raise SystemExit("parity failure in the contract chain: " + str(r["parity"]))
    return {"what": "BASELINE configs[2] with LOD select: " + r["config"], "roofline": roof, "draws": r["draws"], "task_commands": r["task_commands"],
            "meshlets_tested": r["meshlets_tested"], "visible": r["visible"], "us_per_phase": r["step_us"], "meshlets_per_s": r["meshlets_per_s"],
            "drawcull_us": r["drawcull_us"], "cluster_cull_us": r["cluster_cull_us"], "cluster_scatter_us": r["cluster_scatter_us"],
            "options": "NV_OPT_FUSED_COUNT_RESET + NV_OPT_FUSED_SUBMIT (4 launches per phase)", "regime": "one phase after the other on one stream, the shortest of three loops of %d phases" % iters,
            "parity": r["parity"], "parity_checked": "task commands + count words, visible-ID list + count + submit padding, drawVisibility against the CPU oracle"}


def pmc_traffic(n_meshlets, args):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside the process being timed:
    they come from separate `rocprofv3 --pmc` passes over this same command (tools/pmc_traffic.sh), whose per-launch
    means are committed as profiles/rNN_pmc_traffic.json (the newest round's file is used) with the guide's gfx950
    corrections already applied.  Used only when the committed measurement is for this workload; otherwise null."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                rec = json.load(f)
            if rec.get("meshlets_per_gpu") == n_meshlets and rec.get("meshlet_layout") == ("AoS24" if args.aos else "SoA12"):
                return rec["cluster_mask_kernel"]["traffic_bytes"], "profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; FETCH_SIZE x2 per the gfx950 calibration)" % os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def oracle_check(args, cd, draws, meshlets, cmd_b, cmd_e, gpu_visible, gpu_ids):
    """one untimed pass of the CPU oracle over this rank's shard (oracle/ = test infrastructure; here ONLY as a checker): (ok, detail)"""
    import oracle
    from niagara_amd import synth
    n_cmd = cmd_e - cmd_b
    commands = make_commands(cmd_b, cmd_e, args.commands_per_draw)
    cib, cc4 = np.zeros(n_cmd * 64, np.uint32), np.zeros(4, np.uint32)
    oracle.clustercull(cd, 0, commands, synth.count4_for(n_cmd), draws, meshlets, None, None, cib, cc4, threads=max(1, oracle.max_threads() // max(1, args.gpus)))
    if int(cc4[0]) != gpu_visible:
        return False, "CPU oracle sees %d visible meshlets on commands [%d, %d), GPU %d" % (int(cc4[0]), cmd_b, cmd_e, gpu_visible)
    if not (gpu_ids == cib[:len(gpu_ids)]).all():
        return False, "the visible-ID lists differ on commands [%d, %d)" % (cmd_b, cmd_e)
    return True, ""


def cpu_baseline(args, cd, draws, meshlets, cmd_b, cmd_e, gpu_visible, gpu_ids, world=1):
    """the CPU oracle (oracle/ = test infrastructure; here ONLY as the timed baseline and as a checker) on this host"""
    import oracle
    from niagara_amd import synth
    n_cmd = cmd_e - cmd_b
    commands = make_commands(cmd_b, cmd_e, args.commands_per_draw)
    c4 = synth.count4_for(n_cmd)
    threads = oracle.max_threads()
    cib = np.zeros(n_cmd * 64, np.uint32)
    times = []
    spent = 0.0
    while spent < args.cpu_seconds or len(times) < 5:
        cc4 = np.zeros(4, np.uint32)
        t = time.perf_counter()
        oracle.clustercull(cd, 0, commands, c4, draws, meshlets, None, None, cib, cc4, threads=threads)
        dt = time.perf_counter() - t
        times.append(dt)
        spent += dt
    if int(cc4[0]) != gpu_visible or not (gpu_ids == cib[:len(gpu_ids)]).all():
        raise SystemExit("parity failure: CPU oracle sees %d visible meshlets, GPU %d (or the ID lists differ)" % (int(cc4[0]), gpu_visible))
    # median of >= 5 passes.  The host is shared (passes scatter between the quiet-machine time and 20x
    # that), so the best pass — what the cores can do — is reported next to it.
    best, med = min(times), sorted(times)[len(times) // 2]
    return {"value": n_cmd * 64 / med, "unit": "meshlets/s", "cores": threads, "kind": "port", "best_pass_value": n_cmd * 64 / best,
            "visible_list": "bit-identical to the GPU's (%d IDs)" % len(gpu_ids),
            "sample": "%d passes of %s (%d meshlets), OpenMP oracle: median pass %.1f ms (value), best pass %.1f ms"
                      % (len(times), "the full batch" if world == 1 else "rank 0's shard, 1 / %d of the pool" % world, n_cmd * 64, med * 1e3, best * 1e3)}


if __name__ == "__main__":
    main()
