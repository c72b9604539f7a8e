#!/bin/bash
# usage (through gpurun): bash tools/pmc_config.sh <tag> <bench_configs key> — HBM-side traffic (FETCH_SIZE x2 + WRITE_SIZE, the
# gfx950 correction of tools/pmc_traffic.sh) and kernel-trace durations of the kernels of one tools/bench_configs.py config.
# One counter per rocprofv3 pass, kernel-trace only.
tag=$1; cfg=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmcc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pmc -f csv -d $out/$pmc -- python $R/tools/bench_configs.py --iters 20 --only $cfg > $out/$pmc.log 2>&1
done
rocprofv3 --kernel-trace --stats -f csv -d $out/kt -- python $R/tools/bench_configs.py --iters 40 --only $cfg > $out/kt.log 2>&1
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if "nv::" in row["Kernel_Name"]:
                acc[row["Kernel_Name"].split("(")[0][-70:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = {}
for f in glob.glob("$out/kt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Name"].split("(")[0][-70:]] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
res = {"config": "$cfg"}
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    if n < 10:
        continue
    f = sum(cs.get("FETCH_SIZE", [0])) / max(1, len(cs.get("FETCH_SIZE", [0])))
    w = sum(cs.get("WRITE_SIZE", [0])) / max(1, len(cs.get("WRITE_SIZE", [0])))
    t = dur.get(k, (None, 0))[0]
    res[k] = {"read_bytes": 2 * f * 1024, "write_bytes": w * 1024, "traffic_bytes": 2 * f * 1024 + w * 1024, "kernel_trace_avg_us": t,
              "traffic_GBs": (2 * f * 1024 + w * 1024) / t / 1e3 if t else None, "launches": n}
json.dump(res, open("$out/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
