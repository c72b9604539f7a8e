#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory (gpurun_out/prof_<tag>) into a markdown summary under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_r01 profiles/r01_clustercull_config3A.md "title"
"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(dst)
lines = ["# %s — rocprofv3 summaries (MI355X, gfx950, ROCm 7.2)" % title, "",
         "Collected by `tools/profile.sh` through gpurun:",
         "`rocprofv3 --kernel-trace --stats -f csv -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline` and, in separate",
         "runs (one per counter group, never combined with other trace domains),",
         "`rocprofv3 --kernel-trace --pmc <counters> -f csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline`.", "",
         "## kernel stats (kernel-trace)", "", "```"]
def newest(pattern):
    """a directory may hold the files of earlier runs (gpurun merges): keep the most recent one"""
    fs = glob.glob(pattern, recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []


for f in newest(src + "/kt/**/*kernel_stats.csv"):
    lines += [l[:230] for l in open(f).read().strip().split("\n")]
lines += ["```", "", "## kernel stats, one pass after the other (`bench.py --streams 1 ...`, kernel-trace)", "",
          "The default command issues its timed steps on three HIP streams, so the table above averages launches that shared the",
          "chip with neighbour passes (timed region: longer individually, shorter per step) and launches that did not (the",
          "single-stream and event-instrumented legs).  bench.py's `roofline` times the kernels one pass after the other; the trace",
          "of that mode:", "", "```"]
for f in newest(src + "/kt1/**/*kernel_stats.csv"):
    lines += [l[:230] for l in open(f).read().strip().split("\n")[:4]]
lines += ["```", "", "## bench.py line of the traced run", "", "```"]
log = os.path.join(src, "kt.log")
if os.path.exists(log):
    lines += [l.strip()[:3000] for l in open(log) if l.startswith("{")]
lines += ["```", "", "## PMC averages per dispatch", "",
          "FETCH_SIZE / WRITE_SIZE are in KiB-like units of the tool (x1024 B); on gfx950 FETCH_SIZE reports half of the bytes of a",
          "wide coalesced read (MI355X_MICROARCH.md, HBM) — doubled before comparing with algorithmic bytes.", "", "```"]
for d in sorted(glob.glob(src + "/pmc*")):
    if not os.path.isdir(d):
        continue
    for f in newest(d + "/**/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "nv::" in k:
                lines.append("%s | %s | %s | dispatches=%d" % (os.path.basename(d), k, json.dumps({c: round(sum(v) / len(v), 1) for c, v in cs.items()}),
                                                              len(next(iter(cs.values())))))
lines += ["```", ""]
open(dst, "w").write("\n".join(lines))
print("wrote", dst)
