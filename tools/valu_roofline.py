#!/usr/bin/env python3
"""valu_roofline.py — a VECTOR-ISSUE roofline for the kernels that are bound by instruction issue, next to the HBM one.

"0.21 of the HBM roofline" is the only figure the bench lines had for kernels that sit at 0.8 VALU-busy.  This gives them the bound they are
actually held by:

    floor_us = SQ_INSTS_VALU x (class-weighted cycles per wave-instruction) / (SIMDs x clock)        frac = floor_us / measured_us

  * SQ_INSTS_VALU per launch: a rocprofv3 --pmc pass (kernel-trace only) over tools/bench_configs.py, committed as profiles/rNN_valu_counters.json
    (tools/pmc_valu.sh writes it; the newest round's file is used);
  * the class weights: what a wave-instruction of each class costs one SIMD on this chip, measured by tools/experiments/valu_classes.hip
    (eight independent chains per wave, four waves per SIMD): fp32 multiply-adds, integer adds / logic, moves 2.4-3.0 cycles; compares, selects,
    min / max, conversions, shifts, v_fma_mix, DPP moves, readlane / writelane, the v_div_* helpers 4.2-4.8; reciprocal / root, 32-bit integer
    multiplies 8.2-8.3.  The MIX of a kernel is taken from its compiled ISA (every VALU instruction of the kernel's text, statically — loops and
    branches weigh what they weigh in the text): an approximation, stated as such in the line;
  * `reference_insts`: the instructions of the reference's own arithmetic in that kernel (207 per occlusion probe, 122 per vertex
    pass ...), so that the head-room left in the glue around it is a figure.

Library use: roofline_valu(kernel_substring, measured_us[, units]) -> dict or None.  CLI: python tools/valu_roofline.py  (prints the mixes).
"""
import glob
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "niagara_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S"]
SIMDS, CLOCK_GHZ = 1024, 2.4
COST = {"full": 2.7, "half": 4.5, "quarter": 8.3}  # cycles per wave-instruction and SIMD (midpoints of valu_classes.hip's ranges)
QUARTER = re.compile(r"^v_(rcp|rsq|sqrt|log|exp|sin|cos|mul_lo_u32|mul_hi_u32|mul_hi_i32|mad_u64_u32|mad_i64_i32)")
HALF = re.compile(r"^v_(cmp|cndmask|min|max|med3|floor|ceil|trunc|rndne|fract|cvt|frexp|ldexp|lshl|lshr|ashr|mad_u32_u24|mul_u32_u24|lshl_add|lshl_or|mov_b64|fma_mix|readlane|"
                  r"readfirstlane|writelane|div_scale|div_fmas|div_fixup|bfe|bfi|alignbit|perm|mbcnt|bcnt|ffbh|ffbl|pk_)")

# kernel (substring of the demangled name) -> (source file, substring of the MANGLED name of the instance the frame / config runs, what the reference's own arithmetic is)
KERNELS = {
    "cluster_hiz_kernel": ("clustercull.hip", "cluster_hiz_kernelILb1ELb1E", "207 per probe (reference sphere + projectSphere + the four texel tests)"),
    # (round 6: the direct form's packed walk — windows of 64 valid meshlets; the last template argument)
    "cluster_mask_kernel<false, true, false, 8, true, true, true>": ("clustercull.hip", "cluster_mask_kernelILb0ELb1ELb0ELi8ELb1ELb1ELb1E", "~50 per window of 64 meshlets (certified frustum + cone test)"),
    "cluster_mask_kernel<false, true, false, 8, true, false, true>": ("clustercull.hip", "cluster_mask_kernelILb0ELb1ELb0ELi8ELb1ELb0ELb1E", "~50 per window of 64 meshlets (certified frustum + cone test)"),
    "cluster_mask_kernel<false, true, false, 8, true, true, false>": ("clustercull.hip", "cluster_mask_kernelILb0ELb1ELb0ELi8ELb1ELb1ELb0E", "~60 per command (certified frustum + cone test)"),
    "cluster_mask_kernel<false, true, false, 8, true, false, false>": ("clustercull.hip", "cluster_mask_kernelILb0ELb1ELb0ELi8ELb1ELb0ELb0E", "~60 per command (certified frustum + cone test)"),
    "cluster_bits_kernel": ("clustercull.hip", "cluster_bits_kernelILb1ELb1E", "~60 per set bit (certified frustum + cone test)"),
    "trianglecull_kernel": ("trianglecull.hip", "trianglecull_kernel", "122 per vertex + 45 per triangle: 13.5 M per 131 072-cluster pass"),
    "draw_decide_kernel": ("drawcull.hip", "draw_decide_kernel", "~95 per draw (sphere, frustum, LOD select; + projectSphere and the probe in the late pass)"),
}
_isa_cache = {}


def _isa(src):
    if src not in _isa_cache:
        out = os.path.join("/tmp", "nv_valu_" + src.replace(".", "_") + ".s")
        subprocess.run(["hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, src), "-o", out], check=True, capture_output=True)
        _isa_cache[src] = open(out).read()
    return _isa_cache[src]


def class_mix(src, mangled_part):
    """static VALU class counts of the first kernel of `src` whose mangled name contains `mangled_part`"""
    text = _isa(src)
    m = re.search(r"^(_Z\w*%s\w*):.*?$(.*?)^\s*\.end_amdhsa_kernel" % re.escape(mangled_part), text, re.S | re.M)
    if not m:
        return None
    counts = {"full": 0, "half": 0, "quarter": 0}
    for line in m.group(2).split("\n"):
        ins = line.strip().split(" ")[0]
        if not ins.startswith("v_") or ins.startswith("v_nop"):
            continue
        dpp = " row_" in line or "quad_perm" in line or "wave_sh" in line or "row_bcast" in line
        cls = "quarter" if QUARTER.match(ins) else ("half" if (HALF.match(ins) or dpp) else "full")
        counts[cls] += 1
    counts["kernel"] = m.group(1)
    return counts


def counters():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_valu_counters.json")))
    return (json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)) if files else ({}, None)


def roofline_valu(kernel, measured_us, units=None):
    """kernel: a key of KERNELS; measured_us: the launch's duration in this run; units: (name, count) for a per-unit instruction figure"""
    data, source = counters()
    entry = next((v for k, v in data.get("kernels", {}).items() if kernel in k), None)
    if entry is None or kernel not in KERNELS:
        return None
    src, mangled, ref = KERNELS[kernel]
    mix = data.get("class_mix", {}).get(kernel) or class_mix(src, mangled)  # (stored with the counters by tools/pmc_valu.sh: no compiler needed at bench time)
    if not mix:
        return None
    total = mix["full"] + mix["half"] + mix["quarter"]
    weight = sum(COST[c] * mix[c] for c in COST) / total
    insts = entry["SQ_INSTS_VALU"]
    cycles = insts * weight
    floor_us = cycles / SIMDS / (CLOCK_GHZ * 1e3)
    out = {"bound": "valu-issue", "insts": insts, "insts_source": source + " (SQ_INSTS_VALU per launch, mean of %d launches of a separate rocprofv3 --pmc pass)" % entry.get("launches", 0),
           "class_mix_static": {c: round(mix[c] / total, 3) for c in COST}, "class_cost_cycles": COST, "class_weighted_cycles": cycles,
           "floor_us": floor_us, "measured_us": measured_us, "frac": floor_us / measured_us if measured_us else None,
           "peak": "%d SIMDs x %.1f GHz, one wave-instruction per class cost (tools/experiments/valu_classes.hip)" % (SIMDS, CLOCK_GHZ), "reference_insts": ref,
           "note": "class mix from the kernel's compiled text (static), instruction count from the counters (dynamic)"}
    if "SQ_INSTS_SALU" in entry:
        out["salu_insts"] = entry["SQ_INSTS_SALU"]
    # (round 6) the EXECUTED class counts, where the counters pass collected them: fp32 multiply-adds / multiplies / adds at the full-rate
    # cost, transcendentals at the quarter-rate cost, conversions at the half-rate cost — and everything the counters do not classify (compares, selects,
    # min / max, v_fma_mix, moves, lane operations: SQ_INSTS_VALU minus the classified ones) at the CHEAPEST cost, so that this floor can only be too low
    cls = {c: entry.get("SQ_INSTS_VALU_" + c) for c in ("FMA_F32", "MUL_F32", "ADD_F32", "TRANS_F32", "CVT", "INT32")}
    if all(v is not None for v in cls.values()):
        rest = max(0.0, insts - sum(cls.values()))
        cyc = COST["full"] * (cls["FMA_F32"] + cls["MUL_F32"] + cls["ADD_F32"] + cls["INT32"] + rest) + COST["quarter"] * cls["TRANS_F32"] + COST["half"] * cls["CVT"]
        fl = cyc / SIMDS / (CLOCK_GHZ * 1e3)
        out["executed"] = {"class_counts": dict(cls, unclassified=rest), "floor_us_lower_bound": fl, "frac_lower_bound": fl / measured_us if measured_us else None,
                           "note": "class counts from the SQ_INSTS_VALU_* counters (executed); unclassified instructions priced at the cheapest class"}
    if units:
        out["insts_per_" + units[0]] = insts * 64.0 / units[1] if units[0] in ("probe", "lane") else insts / units[1]
    return out


def all_mixes():
    return {k: class_mix(src, mangled) for k, (src, mangled, ref) in KERNELS.items()}


if __name__ == "__main__":
    data, source = counters()
    print("counters:", source)
    for k, (src, mangled, ref) in KERNELS.items():
        mix = class_mix(src, mangled)
        entry = next((v for n, v in data.get("kernels", {}).items() if k in n), None)
        print(k, "->", mix, "| counters:", entry)
