#!/bin/bash
# usage (through gpurun): bash tools/trace_config.sh <bench_configs keys> [tag]
# rocprofv3 kernel-trace of tools/bench_configs.py --only <keys>: average duration per kernel -> gpurun_out/trace_<tag>.txt
keys=$1; tag=${2:-$1}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/trace_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $out -- python $R/tools/bench_configs.py --iters 60 --only $keys > $out/run.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$out/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("$R/gpurun_out/trace_$tag.txt", "w") as o:
        for r in rows:
            line = "%-90s calls %6s avg_us %9.2f min_us %9.2f max_us %9.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
            print(line); o.write(line + "\n")
PY
