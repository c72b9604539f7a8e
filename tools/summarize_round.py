#!/usr/bin/env python3
"""summarize_round.py <tag> — profiles/<tag>_* from what tools/round_evidence.sh <tag> and tools/direct_form_insts.sh <tag> left under gpurun_out/ (final_<tag>/, evidence_<tag>/,
insts_<tag>/, pmct_<tag>/): the bench lines, the per-kernel traffic, the counters, and the four markdown summaries (headline, direct form before / after, frame, plain build).
Runs on the build host (no GPU); round 6's files were written by it."""
import json
import os
import re
import shutil
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E, F, I, P = (os.path.join(R, "gpurun_out", d % TAG) + "/" for d in ("evidence_%s", "final_%s", "insts_%s", "pmct_%s"))
OUT = os.path.join(R, "profiles", TAG + "_")
sys.path.insert(0, os.path.join(R, "tools"))
COST = {"full": 2.7, "half": 4.5, "quarter": 8.3}


def lines(path):
    return [json.loads(l) for l in open(path) if l.startswith("{")]


# ---- raw files
shutil.copy(F + "bench_default.json", OUT + "bench_default.json")
shutil.copy(F + "bench_20.jsonl", OUT + "bench_driver_style_20_steps.jsonl")
shutil.copy(E + "valu_counters.json", OUT + "valu_counters.json")
traffic = json.load(open(F + "pmc_traffic.json"))
if "cluster_mask_kernel" not in traffic:  # (a run of tools/pmc_traffic.sh from before its kernel-name fix: take the launch from the raw table)
    raw = json.load(open(P + "traffic_raw.json"))
    k = max((v for n, v in raw.items() if "mask_kernel" in n), key=lambda v: v["FETCH_SIZE"]["launches"])
    m = lambda c: k.get(c, {}).get("mean_per_launch")  # noqa: E731
    f, w = m("FETCH_SIZE"), m("WRITE_SIZE")
    traffic["cluster_mask_kernel"] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w, "read_bytes": 2 * f * 1024, "write_bytes": w * 1024, "traffic_bytes": 2 * f * 1024 + w * 1024,
                                      "TCC_EA0_RDREQ": m("TCC_EA0_RDREQ_sum"), "TCC_EA0_WRREQ": m("TCC_EA0_WRREQ_sum"), "TCC_HIT": m("TCC_HIT_sum"), "TCC_MISS": m("TCC_MISS_sum"),
                                      "launches": k["FETCH_SIZE"]["launches"]}
json.dump(traffic, open(OUT + "pmc_traffic.json", "w"), indent=1)
cfg = {k: json.load(open(E + "pmc_traffic_%s.json" % k)) for k in ("frame_py", "2_fused", "4", "3b_chain", "3a_dense")}
json.dump({"what": "HBM-side traffic per kernel (FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_config.sh: one rocprofv3 --pmc pass per counter, kernel-trace only) and kernel-trace durations of "
                   "tools/bench_configs.py configs on the round's final tree", "configs": cfg}, open(OUT + "configs_pmc_traffic.json", "w"), indent=1)

# ---- bench_configs lines with roofline_valu against this round's counters
import valu_roofline as V  # noqa: E402

PACKED, PACKED_DEFER = "cluster_mask_kernel<false, true, false, 8, true, false, true>", "cluster_mask_kernel<false, true, false, 8, true, true, true>"
cl = lines(F + "bench_configs.jsonl")
for d in cl:
    c = d["config"]
    if c.startswith("frame"):
        d["roofline_valu"] = {"cluster_hiz_kernel": V.roofline_valu("cluster_hiz_kernel", d["late_cluster_hiz_us"]),
                              "late cull launch (direct form's packed walk, frustum / cone ballots)": V.roofline_valu(PACKED_DEFER, d["late_cluster_cull_us"]),
                              "early cull launch (one lane per set bit)": V.roofline_valu("cluster_bits_kernel", d["early_cluster_cull_us"])}
    elif c.startswith("N4"):
        d["roofline_valu"] = {"trianglecull_kernel": V.roofline_valu("trianglecull_kernel", d["call_us"])}
    elif "roofline_valu" in d and (c.startswith("3A") or c.startswith("3B")):
        del d["roofline_valu"]  # (the counters file averages the packed kernel over two configs: the per-config figures are in <tag>_direct_form_instructions.md)
with open(OUT + "bench_configs.jsonl", "w") as f:
    for d in cl:
        f.write(json.dumps(d) + "\n")

# ---- headline
bd = json.load(open(F + "bench_default.json"))
b20 = lines(F + "bench_20.jsonl")
kt = open(F + "kt.txt").read()


def ktrow(name):
    m = re.search(re.escape(name) + r".*?calls\s+(\d+) avg\s+(\d+) ns\s+min\s+(\d+) max\s+(\d+)", kt)
    return tuple(int(x) for x in m.groups())


cm, sc = ktrow("cluster_mask_kernel<false, true, false, 4, false, false"), ktrow("cluster_scatter_kernel<16>")
m = re.search(r"cluster_mask_kernel<false, true, false, 4, false, false\s+n=(\d+)\s+(\{.*?\})", open(F + "pmc_insts.txt").read())
cnt, n = eval(m.group(2)), int(m.group(1))
cc, fr, rf = bd["contract_chain"], bd["frame"], bd["roofline"]
ch = [d["contract_chain"] for d in b20]
tr = traffic["cluster_mask_kernel"]["traffic_bytes"]
rng = lambda xs, fmt: "%s-%s" % (fmt % min(xs), fmt % max(xs))  # noqa: E731
md = f"""# {TAG} — `clustercull` on config 3A (10 M meshlets, 156 250 task commands), one MI355X

Collected through gpurun on the round's final tree by `tools/round_numbers.sh {TAG}` inside `tools/round_evidence.sh {TAG}` (one box, one call: `tools/pmc_traffic.sh`, `bench.py`,
`bench.py --steps 20` five times, `tools/kt.sh`, `tools/pmc.sh`, `tools/bench_configs.py`) and summarised by `tools/summarize_round.py {TAG}`; the raw csv is not kept.  The command profiled is
`bench.py`'s default regime — **one pass after the other on one stream**, four input sets rotated (cache-cold), count reset fused.  Files: `{TAG}_bench_default.json`,
`{TAG}_bench_driver_style_20_steps.jsonl`, `{TAG}_pmc_traffic.json`, `{TAG}_configs_pmc_traffic.json`, `{TAG}_bench_configs.jsonl`, `{TAG}_valu_counters.json`,
`{TAG}_direct_form_instructions.md`, `{TAG}_frame.md`, `{TAG}_plain_build.md`.

The headline kernel (the filter form) is round 5's: round 6 worked on the form niagara's own pipeline runs (`{TAG}_direct_form_instructions.md`) and had to leave this pass no slower.

## bench line of the same build

| | 200 steps | 20 steps (driver style, 5 runs) | round 5 |
|---|---|---|---|
| value | **{bd['value'] / 1e9:.1f} G meshlets/s** | {rng([d['value'] / 1e9 for d in b20], '%.0f')} G meshlets/s | 383.0 / 374-378 |
| ms_per_step (= one pass, end to end) | **{bd['ms_per_step'] * 1e3:.2f} us** | {rng([d['ms_per_step'] * 1e3 for d in b20], '%.2f')} us | 26.11 / 26.48-26.74 |
| dominant kernel by HIP events (`roofline.kernel_avg_us`) | {rf['kernel_avg_us']:.2f} us -> frac **{rf['frac']:.3f}** | {rng([d['roofline']['frac'] for d in b20], '%.3f')} | 23.70 -> 0.696 |
| pass_frac ({rf['pass_algorithmic_bytes'] / 1e6:.2f} MB / ms_per_step / 8 TB/s) | **{rf['pass_frac']:.3f}** | **{rng([d['roofline']['pass_frac'] for d in b20], '%.3f')}** | 0.630 / 0.615-0.622 |
| scatter launch by events | {rf['scatter_kernel_avg_us']:.2f} us | | 7.45 |
| `roofline.kernel_variants` | {rf['kernel_variants']} | | |
| `parity` | {bd['parity']} (the whole visible-ID list against the CPU oracle) | {b20[0]['parity']} | |
| `contract_chain` (configs[2] with LOD select, 4 launches per phase) | {cc['us_per_phase']:.1f} us per phase ({cc['roofline']['frac']:.2f} of HBM), cull launch {cc['cluster_cull_us']:.1f}, scatter {cc['cluster_scatter_us']:.1f}, drawcull {cc['drawcull_us']:.1f}; {cc['parity']} | **{rng([c['us_per_phase'] for c in ch], '%.1f')} us per phase, {rng([c['roofline']['frac'] for c in ch], '%.2f')} of HBM** ({cc['roofline']['algorithmic_bytes'] / 1e6:.1f} MB), cull launch {rng([c['cluster_cull_us'] for c in ch], '%.1f')} | 50.3 us, 0.37 |
| `frame` (1 M draws, early cull -> pyramid -> late cull) | **{fr['frame_us']:.1f} us**, {fr['frac']:.3f} of HBM; variants {fr['kernel_variants']} | {rng([d['frame']['frame_us'] for d in b20], '%.1f')} us | 194.0-194.3 (`r05_frame.md`) |

## kernel-trace (`rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0`)

| kernel | calls | average | min | max |
|---|---|---|---|---|
| `nv::cluster_mask_kernel<false, true, false, 4, false, false, false>` | {cm[0]} | **{cm[1] / 1e3:.2f} us** | {cm[2] / 1e3:.2f} | {cm[3] / 1e3:.2f} |
| `nv::cluster_scatter_kernel<16>` | {sc[0]} | {sc[1] / 1e3:.2f} us | {sc[2] / 1e3:.2f} | {sc[3] / 1e3:.2f} |

131 875 000 algorithmic bytes / {cm[1] / 1e3:.2f} us = {131.875 / (cm[1] / 1e3):.2f} TB/s = **{131.875 / (cm[1] / 1e3) / 8:.3f} of 8 TB/s** (round 5: 20.74 us = 0.795).  Cull + scatter by kernel-trace =
{(cm[1] + sc[1]) / 1e3:.1f} us inside the {bd['ms_per_step'] * 1e3:.1f} us pass.  HBM-side traffic of the launch (`{TAG}_pmc_traffic.json`, FETCH_SIZE x 2 + WRITE_SIZE, separate `--pmc` passes of
the same command): {tr / 1e6:.1f} MB = {tr / 131875000:.2f} of the algorithmic bytes (cones of frustum-rejected meshlets are never read).

## instruction counters of the cull launch (`rocprofv3 --pmc`, separate pass, mean of {n} launches)

| | round 5 | this tree |
|---|---|---|
| `SQ_INSTS_VALU` per launch / per command | 6.18 M / 39.6 | {cnt['SQ_INSTS_VALU'] / 1e6:.2f} M / {cnt['SQ_INSTS_VALU'] / 156250:.1f} |
| `SQ_INSTS_SALU` per launch / per command | 3.43 M / 22.0 | {cnt['SQ_INSTS_SALU'] / 1e6:.2f} M / {cnt['SQ_INSTS_SALU'] / 156250:.1f} |

## what was tried on this pass in round 6 (EXPERIMENTS.md (A6) 9, 12)

* the filter form's exact pass as a packed walk over the lanes the filter could not finish: 26.0 -> 27.9 us (as a hybrid from four candidates per segment: 31-32 us, spills) — archived;
* a measured fill statistic beside the filter statistic: the scatter launch 4.71 -> 5.09 us by kernel-trace whichever part of its plumbing was removed (20 builds) — replaced by a
  host-side estimate; this pass's kernels are round 5's again (scatter {sc[1] / 1e3:.2f} us above).
"""
open(OUT + "clustercull_config3A.md", "w").write(md)


# ---- the direct form before / after
def cnts(path, sub):
    for l in open(path):
        if sub in l:
            return eval(re.search(r"(\{.*\})", l).group(1))


cfgs = [("contract chain (`3b_chain`: 250 114 commands, 10.01 M meshlets, cache-resident pool)", "3b_chain", "8, true, false,", 250114, 10008880),
        ("3A dense (156 250 full commands, 10 M meshlets from HBM)", "3a_dense", "8, true, false,", 156250, 10000000),
        ("frame, late cull launch (`DEFER`: 174 760 commands, 10.16 M meshlets)", "frame_py", "8, true, true,", 174760, 10156300)]
kt_us = {k: [v["kernel_trace_avg_us"] for n_, v in cfg[k].items() if isinstance(v, dict) and "mask_kernel" in n_ and "8, true" in n_][0] for _, k, _, _, _ in cfgs}
vc = json.load(open(OUT + "valu_counters.json"))
t = "| config | form | `SQ_INSTS_VALU` | `SQ_INSTS_SALU` | `SQ_INSTS_LDS` | VALU + SALU per command | per window of 64 meshlets |\n|---|---|---|---|---|---|---|\n"
fl = ""
for label, key, sub, C, M in cfgs:
    W = M / 64
    for form, name in ((4, "one command per wave iteration (`NV_OPT_CULL_FORM` 4 = round 5's direct form)"), (0, "**packed walk** (the host's choice)")):
        c = cnts(I + "%s_form%d.txt" % (key, form), sub)
        t += f"| {label} | {name} | {c['SQ_INSTS_VALU'] / 1e6:.2f} M | {c['SQ_INSTS_SALU'] / 1e6:.2f} M | {c['SQ_INSTS_LDS'] / 1e6:.2f} M | {c['SQ_INSTS_VALU'] / C:.1f} + {c['SQ_INSTS_SALU'] / C:.1f} | {c['SQ_INSTS_VALU'] / W:.1f} + {c['SQ_INSTS_SALU'] / W:.1f} |\n"
    c, c1, c2 = cnts(I + "%s_form0.txt" % key, sub), cnts(I + "%s_cls1.txt" % key, sub), cnts(I + "%s_cls2.txt" % key, sub)
    Vn = c["SQ_INSTS_VALU"]
    cls = {k.replace("SQ_INSTS_VALU_", ""): v for k, v in {**c1, **c2}.items()}
    rest = Vn - sum(cls.values())
    lb = (COST["full"] * (cls["FMA_F32"] + cls["MUL_F32"] + cls["ADD_F32"] + cls["INT32"] + rest) + COST["quarter"] * cls["TRANS_F32"] + COST["half"] * cls["CVT"]) / 1024 / 2400
    mix = vc["class_mix"][PACKED_DEFER if key == "frame_py" else PACKED]
    w = sum(COST[k] * mix[k] for k in COST) / (mix["full"] + mix["half"] + mix["quarter"])
    st = Vn * w / 1024 / 2400
    fl += (f"| {label} | {Vn / 1e6:.2f} M | FMA {cls['FMA_F32'] / 1e6:.2f}, MUL {cls['MUL_F32'] / 1e6:.2f}, ADD {cls['ADD_F32'] / 1e6:.2f}, INT32 {cls['INT32'] / 1e6:.2f}, CVT {cls['CVT'] / 1e6:.2f}, "
           f"TRANS {cls['TRANS_F32'] / 1e6:.3f}, unclassified {rest / 1e6:.2f} M | {kt_us[key]:.2f} us | {st:.1f} us = **{st / kt_us[key]:.2f}** | {lb:.1f} us = {lb / kt_us[key]:.2f} |\n")
dense = [d for d in cl if d["config"].startswith("3A dense: ")][0]
frames = [d for d in cl if d["config"].startswith("frame")]
md = f"""# {TAG} — the direct form of the cluster cull launch before / after the packed walk (VERDICT r5 item 2)

`tools/direct_form_insts.sh {TAG}` through gpurun on the final tree: one `rocprofv3 --kernel-trace --pmc` pass per configuration and form over `tools/bench_configs.py --iters 20 --only <config>`
(`NV_BENCH_CULL_FORM` pins `NV_OPT_CULL_FORM`: 4 = one command per wave iteration — round 5's direct form, still in the library —, 0 = the host's choice, which is the packed
walk on all three), per-launch means; two more passes per configuration for the executed class counters (`SQ_INSTS_VALU_{{FMA,MUL,ADD,TRANS}}_F32`, `_CVT`, `_INT32`).

{t}
What the walk changes: every lane of every window is live (the chain's commands average 40 valid lanes: 250 k command iterations become 156 k windows), the draw's coefficients come
from an 80-byte LDS table entry per lane instead of 23 `v_readlane` per draw change + 5 per command, the margin is the draw's `tK` (four multiply-adds less), the walk is a
counted loop without per-command broadcasts.

## times and the vector-issue roofline of the packed walk

Kernel-trace averages from `{TAG}_configs_pmc_traffic.json` (same tree; HIP-event times of the same launches are ~2-3 us longer: `{TAG}_bench_configs.jsonl`).  Floor = instructions x
class cost / (1024 SIMDs x 2.4 GHz), class costs from `tools/experiments/valu_classes.hip` (2.7 / 4.5 / 8.3 cycles); "static mix" = the kernel's compiled text (`tools/valu_roofline.py`),
"executed, lower bound" = the executed class counters with everything they do not classify (compares, selects, min / max, `v_fma_mix`, moves, `v_mbcnt`) priced at the CHEAPEST class.

| config | `SQ_INSTS_VALU` | executed classes | launch (kernel-trace) | floor, static mix = frac | floor, executed lower bound = frac |
|---|---|---|---|---|---|
{fl}
`tools/wave_timeline.py` (`NV_DIRECT=1`, 3A geometry) on the same kernel: a wave lives 17-19 us for ~25 windows, 267 cycles per window and SIMD with six waves per SIMD — 65 vector
instructions at 4 cycles each: the launch is bound by vector issue; what is left above the floor is the ramp (all waves in after 1.7 us, first data ~2 us later) and the spread of the
waves' ends, which the walk's own delay table in the dealing narrowed (EXPERIMENTS (A6) 11, 19).

| launch | round 5 | this tree | VERDICT r5's bar |
|---|---|---|---|
| contract chain, cull launch (events) | 33.8 us (lane-per-valid-cluster form) | **{rng([c['cluster_cull_us'] for c in ch], '%.1f')} us** ({kt_us['3b_chain']:.1f} by kernel-trace) | <= 28 |
| contract chain, us per phase | 50.0 (0.37 of HBM) | **{rng([c['us_per_phase'] for c in ch], '%.1f')}** ({rng([c['roofline']['frac'] for c in ch], '%.2f')} of HBM; five driver-style runs) | <= 42 |
| 3A dense, cull launch (events) | 36.7 | **{dense['cull_us']:.1f}** ({kt_us['3a_dense']:.1f} by kernel-trace) | <= 31 |
| frame, late cull launch (events) | 36.6-38.2 | **{rng([d['late_cluster_cull_us'] for d in frames[:2]], '%.1f')}** ({kt_us['frame_py']:.1f} by kernel-trace) | <= 32 |
"""
open(OUT + "direct_form_instructions.md", "w").write(md)

# ---- frame
rows = []
for l in open(F + "trace_frame.txt"):
    m = re.match(r"(.*?)\s+calls\s+(\d+) avg_us\s+([\d.]+) min_us\s+([\d.]+) max_us\s+([\d.]+)", l)
    if m and "nv::" in m.group(1) and int(m.group(2)) >= 30:
        rows.append(m.groups())
tc = cfg["frame_py"]
md = f"""# {TAG} — niagara's dependent frame at BASELINE scale (1 M draws, early cull -> pyramid -> late cull), one MI355X

`tools/bench_configs.py --only frame,frame_py,frame_contract` (inside `tools/round_numbers.sh {TAG}`), `tools/trace_config.sh frame_py` and `tools/pmc_config.sh {TAG}_frame_py frame_py` through
gpurun on the final tree; every buffer of both phases against the CPU oracle before a time is reported.  Lines: `{TAG}_bench_configs.jsonl`; per-kernel traffic: `{TAG}_configs_pmc_traffic.json`.

| line | frame | early drawcull | early cull | early scatter | pyramid | late drawcull | late cull | occlusion stage | late scatter |
|---|---|---|---|---|---|---|---|---|---|
"""
for d, name in zip(frames, ("`frame` (C++ driver timed too)", "`frame_py`", "`frame_contract` (no fusion options: 19 launches)")):
    md += "| %s | **%.1f us** | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f |\n" % (name, d["frame_us"], d["early_drawcull_us"], d["early_cluster_cull_us"], d["early_cluster_scatter_us"],
                                                                                             d["pyramid_us"], d["late_drawcull_us"], d["late_cluster_cull_us"], d["late_cluster_hiz_us"], d["late_cluster_scatter_us"])
md += f"""
(per-launch columns = the library's HIP event pairs, in separate frames; round 5: 194.0-194.3 us, late cull launch 37.6-38.2.)  The driver-style bench lines of the same box
(`{TAG}_bench_driver_style_20_steps.jsonl`, `frame.frame_us`): {', '.join('%.1f' % d['frame']['frame_us'] for d in b20)} us.  `frac` = 444.9 MB / frame time / 8 TB/s = {frames[1]['frac']:.3f}.

## kernel-trace of `frame_py` (`rocprofv3 --kernel-trace --stats`, averages) and HBM-side traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate `--pmc` passes)

| kernel | calls | average | min | max | traffic per launch | of that, GB/s |
|---|---|---|---|---|---|---|
"""
for n_, c, a, mi, ma in rows:
    short = n_.split("(")[0].replace("void ", "").strip()
    tt = next((v for k, v in tc.items() if isinstance(v, dict) and k.replace("void ", "").strip()[-40:] == short[-40:]), None)
    md += "| `%s` | %s | **%s us** | %s | %s | %s | %s |\n" % (short, c, a, mi, ma, ("%.1f MB" % (tt["traffic_bytes"] / 1e6)) if tt else "", ("%d" % tt["traffic_GBs"]) if tt and tt["traffic_GBs"] else "")
md += """
The sum of the averages is the frame: the launches are back to back.  What changed against round 5: the late cull launch (frustum / cone ballots of 174 760 commands, `DEFER`) walks packed
windows of 64 valid meshlets (`cluster_mask_kernel<false, true, false, 8, true, true, true>`: 28-29 us against 37.6-38.2); the early cull launch (one lane per set bit) tests with the draw's
margin `tK`; `drawcull`'s decide launch is a fixed grid of waves walking 64-draw units behind a ring of requests (late: 12.1-12.2 us against 12.8-13.3; early, visibility words ahead
of the records and only last frame's visible draws' records fetched: 8.3 against 9.2-10.4, 20.7 MB per launch instead of 33) and hands 16-byte records of the emitting draws to the TASK
scatter launch (`draw_scatter_kernel<true, true, 1u, true, true>`: 6.2-6.5 us against 8.5-8.8).  The occlusion stage (58 us, 0.77 of its vector-issue floor) reads 209 MB per launch for 6.1 M probes; a group test in front of it was measured on the CPU and not built
(`tools/experiments/hiz_group_fraction.py`: 26 % of the listed commands end all-occluded, a conservative test certifies 0.1 % of them).
"""
open(OUT + "frame.md", "w").write(md)

# ---- plain build
pb = lines(E + "plain_bench.json")[-1]
pc, ac = lines(E + "plain_configs.jsonl"), lines(E + "asm_configs.jsonl")
md = f"""# {TAG} — what the plain-loads build costs (VERDICT r5 item 6d)

`libniagara_vis_plain.so` is the product with ordinary loads and the compiler's own waits in `clustercull.hip` instead of the inline-asm load rings with hand-counted `s_waitcnt vmcnt(N)`
(`-DNV_PLAIN_LOADS`; the escape hatch for a hipcc the ISA scan was not validated on).  Same box, same call (`tools/round_evidence.sh {TAG}`), results bit-identical to the oracle in both builds.

| workload | asm rings (the product) | plain loads |
|---|---|---|
| config 3A headline, `bench.py --steps 100` | {bd['ms_per_step'] * 1e3:.2f} us per pass ({bd['value'] / 1e9:.1f} G meshlets/s) | **{pb['ms_per_step'] * 1e3:.2f} us** ({pb['value'] / 1e9:.1f} G meshlets/s), cull launch by events {pb['roofline']['kernel_avg_us']:.2f} against {bd['roofline']['kernel_avg_us']:.2f} us |
"""
for a, p in zip(ac, pc):
    if a["config"].startswith("3B"):
        md += "| contract chain (`3b_chain`) | %.1f us per phase, cull launch %.1f | **%.1f us**, cull launch %.1f |\n" % (a["step_us"], a["cluster_cull_us"], p["step_us"], p["cluster_cull_us"])
    elif a["config"].startswith("frame"):
        md += "| frame (`frame_py`) | %.1f us (late cull launch %.1f, early %.1f, occlusion stage %.1f) | **%.1f us** (late cull launch %.1f, early %.1f, occlusion stage %.1f) |\n" % (
            a["frame_us"], a["late_cluster_cull_us"], a["early_cluster_cull_us"], a["late_cluster_hiz_us"], p["frame_us"], p["late_cluster_cull_us"], p["early_cluster_cull_us"], p["late_cluster_hiz_us"])
    elif a["config"].startswith("3A dense"):
        md += "| 3A dense | %.1f us per pass, cull launch %.1f | **%.1f us**, cull launch %.1f |\n" % (a["step_us"], a["cull_us"], p["step_us"], p["cull_us"])
md += ("\nhipcc collapses a loop-carried prefetch ring to ~`vmcnt(0)`, which the six waves per SIMD mostly hide: the unvalidated-compiler fallback costs a few per cent on the streaming forms; "
       "the lane-per-item kernels (occlusion stage, early pass with visibility bits) have no ring and do not change.\n")
open(OUT + "plain_build.md", "w").write(md)
print("profiles/%s_* written" % TAG)
