#!/bin/bash
# usage: bash tools/kt.sh <tag> <command...> — rocprofv3 kernel-trace stats of any command; prints the per-kernel table
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/kt_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --stats -f csv -d $out -- "$@" > $out/run.log 2>&1)
python3 - <<PY
import csv, glob
for f in glob.glob("$out/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-70s calls %6s avg %9.2f us  min %9.2f  max %9.2f  %5s%%" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3, row["Percentage"]))
PY
