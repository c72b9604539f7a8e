#!/bin/bash
# usage: bash tools/pmc_quick.sh <tag> [ENV=VAL ...] — two PMC passes (instruction mix, stall split) of bench.py
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmcq_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $pmc -f csv -d $out/pmc$i -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/pmc$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
for d in sorted(glob.glob("$out/pmc*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "nv::" in k and "soa" not in k:
                print(os.path.basename(d), k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in cs.items()}, "(millions)")
PY
