#!/bin/bash
# usage: bash tools/pmc_traffic.sh <tag> — HBM-side traffic of bench.py's kernels from the L2 fabric counters.
# One counter group per rocprofv3 pass (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2), kernel-trace only.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmct_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_HIT[A-Z_]*\|TCC_MISS[A-Z_]*\|FETCH_SIZE\|WRITE_SIZE\|TCC_REQ[A-Z_]*" | sort -u > $out/counters_available.txt
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $pmc -f csv -d $out/pmc$i -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/pmc$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os, json
res = collections.defaultdict(dict)
for d in sorted(glob.glob("$out/pmc*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            short = k.split("(")[0][-60:]
            for c, v in cs.items():
                res[short][c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
json.dump(res, open("$out/traffic_raw.json", "w"), indent=1)
for k, v in res.items():
    print(k, {c: round(x["mean_per_launch"], 1) for c, x in v.items()})
PY
