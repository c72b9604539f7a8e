#!/usr/bin/env python3
"""bench.py — meshlets culled+compacted per second (BASELINE.json metric) on N MI355X GPUs of one node.

Workload (config.workload = "config3A"): BASELINE.json configs[2] — synthetic 10 M meshlets (bistro scale), i.e.
156 250 full MeshTaskCommands over 15 625 draws, per-meshlet cone + frustum cull (`clustercull`, LATE = 0,
clusterBackfaceEnabled = 1) with ordered compaction of the visible IDs.  One "step" = one pass of the hot path over
one batch: reset of the count word (the caller's vkCmdFillBuffer, src/niagara.cpp:1586; fused into the pass through
NV_OPT_FUSED_COUNT_RESET unless --explicit-reset) + nv_clustercull, with inputs resident in HBM.  Steps rotate over `--copies` distinct input sets so that every pass streams from HBM rather
than from the 256 MiB Infinity Cache.  The steps are independent passes (different batches), so they are issued round-robin on
`--streams` HIP streams (default 3; one nv_context, output list and count word per stream): the latency-bound scatter launch
of one pass overlaps the ramp of the next pass's cull launch.  `value` / `ms_per_step` are that throughput;
`roofline.ms_per_pass_single_stream` is one pass after the other on one stream, and the kernels are timed that way.  N > 1: the pool shards by contiguous command ranges, every rank culls its own
10 M meshlets (weak scaling) and the only collective is the all-reduce of the passes' visible counts (RCCL; the rows of
`--counts-batch` passes, written by the scatter launches, share one asynchronous all-reduce; 1 = one collective per pass).

Prints ONE JSON line (rank 0):  metric/value/unit as in BASELINE.json + "roofline" (dominant kernel, HIP events on
the launch stream) + "cpu_baseline" (the CPU oracle timed on this host's cores; a reported baseline, not a target).
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
    ap.add_argument("--draws", type=int, default=15625, help="draws per GPU (x commands-per-draw x 64 = meshlets)")
    ap.add_argument("--commands-per-draw", type=int, default=10)
    ap.add_argument("--copies", type=int, default=4, help="distinct input sets rotated through (cache-cold passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    ap.add_argument("--aos", action="store_true", help="read the 24-B AoS meshlets in place (no SoA mirror)")
    ap.add_argument("--counts-batch", type=int, default=8, help="N > 1: passes whose counts share one all-reduce (1 = one collective per pass)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only for functional tests)")
    ap.add_argument("--shared-device", action="store_true", help="functional test only: all ranks use cuda:0 (needs --backend gloo)")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent passes in flight: steps are issued round-robin on this many HIP streams, one nv_context (scratch, outputs) per stream; 1 = strictly one pass after the other")
    ap.add_argument("--scatter-waves", type=int, default=0, help="NV_OPT_SCATTER_WAVES (4, 8 or 16); 0 = 8 with several streams, 16 with one")
    ap.add_argument("--explicit-reset", action="store_true",
                    help="reference contract: zero the count word with a separate launch (nv_reset_count) instead of NV_OPT_FUSED_COUNT_RESET")
    return ap.parse_args()


def make_inputs(n_draws, cpd, rank, world):
    """the rank's shard of the synthetic pool: (draws, meshlets, CullData, dccb words).  The pool's commands are
    [0, world * n_draws * cpd); shard.command_range gives the rank its contiguous range, whose draws are the matching
    slice of niagara's draw generator (src/niagara.cpp:978-997) and whose meshlets come from a per-rank seed."""
    from niagara_amd import host, shard, synth
    n_cmd = n_draws * cpd
    b, e = shard.command_range(n_cmd * world, rank, world)
    assert (b, e) == (rank * n_cmd, (rank + 1) * n_cmd)
    draws = host.synth_draws(n_draws * world, 1, 300.0)[b // cpd:e // cpd].copy()
    draws["meshletVisibilityOffset"] = np.arange(n_draws, dtype=np.uint32) * (cpd * 64)
    meshlets = synth.make_meshlets(n_cmd * 64, seed=2 + rank)
    cd = host.build_cull_data(draw_count=n_draws, cullingEnabled=1, clusterBackfaceEnabled=1)
    return draws, meshlets, cd, synth.count4_for(n_cmd)


def main():
    args = parse()
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
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        # communicator set-up (lazy in RCCL) belongs to neither the warm-up nor the timed steps
        dist.all_reduce(torch.zeros(1, dtype=torch.int64, device=dev))
        torch.cuda.synchronize()

    # ---- inputs: the pool of world x C commands shards by contiguous command ranges (niagara_amd/shard.py);
    # rank r owns [r C, (r+1) C) and the 10 M meshlets they reference (weak scaling); draws follow their commands
    n_draws, cpd = args.draws, args.commands_per_draw
    n_cmd = n_draws * cpd
    n_meshlets = n_cmd * 64
    copies = max(1, args.copies)
    draws, meshlets, cd, count4 = make_inputs(n_draws, cpd, rank, world)

    # Independent passes overlap: a pass is a bandwidth-bound cull launch followed by a latency-bound scatter launch, so the
    # scatter of pass i (16 waves per CU, ~6 us) and the ramp of the cull launch of pass i + 1 can share the chip.  Steps are
    # issued round-robin on `--streams` HIP streams, each with its own nv_context (ballot scratch, tile counts, hints), output
    # list and count word; every step still does the whole pass, and all of them complete inside the timed region.
    S = max(1, args.streams)
    ctxs = [P.Context(local_rank) for _ in range(S)]
    ctx = ctxs[0]
    if not args.explicit_reset:
        for c in ctxs:
            c.set_option(P.NV_OPT_FUSED_COUNT_RESET, 1)  # the pass absorbs the caller's vkCmdFillBuffer(ccb, 0, 4, 0)
    scatter_waves = args.scatter_waves if args.scatter_waves else (8 if S > 1 else 16)
    for c in ctxs:
        # several passes in flight: the scatter launch with 8 instead of 16 waves per workgroup leaves more of the CUs' wave slots to the
        # neighbour pass's cull launch (the same setting in every leg of this run, also the single-stream ones)
        c.set_option(P.NV_OPT_SCATTER_WAVES, scatter_waves)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    db = P.to_device(draws, dev)
    mlb = torch.empty(copies * n_meshlets * L.MESHLET.itemsize, dtype=torch.uint8, device=dev)
    one = torch.from_numpy(meshlets.view(np.uint8).reshape(-1))
    for c in range(copies):
        mlb[c * one.numel():(c + 1) * one.numel()].copy_(one)
    dcbs = [P.to_device(synth.make_task_commands(n_draws, cpd, meshlet_base=c * n_meshlets), dev) for c in range(copies)]
    dccb = torch.from_numpy(count4.view(np.int32).copy()).to(dev)
    cibs = [torch.zeros(min(n_meshlets, L.CLUSTER_LIMIT) + 256, dtype=torch.int32, device=dev) for _ in range(S)]
    ccbs = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(S)]
    cib, ccb = cibs[0], ccbs[0]
    # N > 1: the passes' counts are summed over the ranks; batched, asynchronous, written by the scatter launch (shard.CountsReducer,
    # one per stream: a stream's reducer sees that stream's passes)
    B = max(1, args.counts_batch)
    reds = [shard.CountsReducer(ctxs[s], dev, B, stream=streams[s] if S > 1 else None) for s in range(S)]
    if not args.aos:
        for c in ctxs:
            c.upload_meshlets(mlb, copies * n_meshlets)
    torch.cuda.synchronize()
    # every argument of a pass marshalled once (Context.bind_clustercull): at ~24 us per step the per-call marshalling of the
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
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: exactly `steps` passes, no instrumentation
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain(args.steps)  # every pass's reduction has completed inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    s_last = (args.steps - 1) % S
    last_counts = reds[s_last].last(passes_of(s_last, args.steps)) if world > 1 else None
    visible_by_stream = [int(c[0].item()) for c in ccbs]

    # ---- one pass after the other on ONE stream (what a single pass costs end to end), same number of steps, untimed by `value`
    single = shard.CountsReducer(ctx, dev, B)

    serial_calls = [ctx.bind_clustercull(None, cd, 0, dcbs[c], dccb, db, mlb, None, None, cib, ccb) for c in range(copies)]

    def serial_step(i):
        if args.explicit_reset:
            ctx.reset_count(ccb)
        single.before_pass(i)
        serial_calls[i % copies]()
        single.after_pass(i)

    n_prof = max(args.steps, 100)  # (a short run still averages the single-stream pass and the kernels over 100 launches)
    for i in range(5):
        serial_step(i)
    single.drain(5)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for i in range(n_prof):
        serial_step(i)
    single.drain(n_prof)
    torch.cuda.synchronize()
    serial = time.perf_counter() - t2

    # ---- roofline leg: the same `steps` passes again with the library's HIP events bracketing each kernel on the
    # launch stream (nv_profile_*).  It is a separate loop because an event record is itself a barrier packet: three
    # records per pass serialise the launches and cost ~10 us per pass, which would deflate `value` by ~25 %.
    ctx.profile(True)
    t1 = time.perf_counter()
    for i in range(n_prof):
        serial_step(i)  # one stream: the kernels are timed without a neighbour pass on the chip
    single.drain(n_prof)
    torch.cuda.synchronize()
    profiled = time.perf_counter() - t1
    prof = ctx.profile_read()
    ctx.profile(False)
    ctx.status()

    visible = int(ccb[0].item())
    visible_ids = cib[:min(visible, L.CLUSTER_LIMIT)].cpu().numpy().view(np.uint32)  # the list the profiled run's last pass left: checked against the oracle below
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        total_visible = int(last_counts[2].item())
    else:
        total_visible = visible

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

    traffic, traffic_note = pmc_traffic(n_meshlets, args)

    if rank == 0:
        out = {
            "metric": "meshlets culled+compacted /sec",
            "value": n_meshlets * world * args.steps / elapsed,
            "unit": "meshlets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "config3A: %d meshlets/GPU, %d task commands over %d draws, cone+frustum clustercull (LATE=0) + ordered compaction"
                                   % (n_meshlets, n_cmd, n_draws),
                       "meshlets_per_gpu": n_meshlets, "commands_per_gpu": n_cmd, "draws_per_gpu": n_draws,
                       "streams": S, "scatter_waves_per_workgroup": scatter_waves, "passes_in_flight": "steps issued round-robin on %d HIP streams, one nv_context / output list per stream; the scatter launch of a pass overlaps the next pass's cull launch" % S if S > 1 else "one pass after the other",
                       "input_copies_rotated": copies, "count_reset": "explicit launch" if args.explicit_reset else "fused (NV_OPT_FUSED_COUNT_RESET)", "meshlet_layout": "AoS24" if args.aos else "SoA12",
                       "visible_per_gpu": visible, "visible_per_stream": visible_by_stream, "visible_total": total_visible, "sharding": "commands x%d" % world,
                       "counts_allreduce": ("none (N=1)" if world == 1 else "one async all-reduce of [%d, 3] int64 per %d passes, rows written by the scatter launch" % (B, B))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_measured_in_run": False, "traffic_source": traffic_note,
                         "kernel": "cluster_mask_kernel", "kernel_avg_us": kernel_avg_s * 1e6,
                         "algorithmic_bytes": algo_bytes, "launches_timed": cull_n,
                         "scatter_kernel_avg_us": scatter_avg_s * 1e6, "ms_per_step_with_events": profiled / n_prof * 1e3,
                         "pass_algorithmic_bytes": pass_bytes,
                         # one pass after the other on one stream, un-instrumented: what a single pass takes end to end
                         "ms_per_pass_single_stream": serial / n_prof * 1e3,
                         # the whole pass (cull + scatter launches) against the roofline: from that single-stream time, and
                         # from the timed region's throughput (passes overlapping on `streams` streams)
                         "pass_frac": pass_bytes / (serial / n_prof) / 1e9 / HBM_PEAK_GBS,
                         "pass_frac_overlapped": pass_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
            "note": ("value and ms_per_step are THROUGHPUT figures: the %d steps are independent passes issued round-robin on %d HIP streams, so consecutive "
                     "passes overlap (scatter launch of one under the cull launch of the next) and ms_per_step can be shorter than one pass; a single pass end to end takes "
                     "roofline.ms_per_pass_single_stream, and the kernels were timed one pass after the other" % (args.steps, S)) if S > 1 else "one pass after the other on one stream",
            "library": niagara_amd.SO_PATH,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, cd, draws, meshlets, n_cmd, visible, visible_ids)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


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


def cpu_baseline(args, cd, draws, meshlets, n_cmd, gpu_visible, gpu_ids):
    """the CPU oracle (oracle/ = test infrastructure; here ONLY as the timed baseline and as a checker) on this host"""
    import oracle
    from niagara_amd import synth
    commands = synth.make_task_commands(len(draws), args.commands_per_draw)
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
            "sample": "%d passes of the full %d-meshlet config3A batch, OpenMP oracle: median pass %.1f ms (value), best pass %.1f ms"
                      % (len(times), n_cmd * 64, med * 1e3, best * 1e3)}


if __name__ == "__main__":
    main()
