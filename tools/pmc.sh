#!/bin/bash
# usage (on the GPU box): bash tools/pmc.sh <tag> "<COUNTER ...>" [ENV=VAL ...] -- <command ...>
# one rocprofv3 --pmc pass (kernel-trace only, as the pool requires) of the command; prints per-kernel averages of the counters
tag=$1; shift
pmc=$1; shift
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cmd=("$@")
cd /tmp && export TMPDIR=/tmp
( cd $R && env "${envs[@]}" rocprofv3 --kernel-trace --pmc $pmc -f csv -d $out -- "${cmd[@]}" > $out/run.log 2>&1 ) || echo "pmc pass failed: $pmc"
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:64]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== pmc [$tag]")
    for k, cs in acc.items():
        if any(x in k for x in ("nv::", "reduce", "cluster", "draw")):
            print("%-64s n=%-5d" % (k, len(next(iter(cs.values())))), {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
PY
