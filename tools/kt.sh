#!/bin/bash
# usage (on the GPU box): bash tools/kt.sh <tag> [ENV=VAL ...] -- <command ...>
# rocprofv3 --kernel-trace --stats of the command; prints the per-kernel table (calls, average / min / max ns) and keeps the csv under
# gpurun_out/kt_<tag>/
tag=$1; shift
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/kt_$tag
rm -rf $out; mkdir -p $out
cmd=("$@")
cd /tmp && export TMPDIR=/tmp
( cd $R && env "${envs[@]}" rocprofv3 --kernel-trace --stats -f csv -d $out -- "${cmd[@]}" > $out/run.log 2>&1 )
python3 - <<PY
import csv, glob
for f in glob.glob("$out/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("== kernel stats [$tag]")
    for r in rows:
        name = r["Name"]
        if any(k in name for k in ("nv::", "reduce", "cluster", "draw")) :
            print("%-70s calls %6s avg %9.0f ns  min %9s max %9s  %5s%%" % (name[:70], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
PY
