#!/bin/bash
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmcs_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32"; do
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
            if "cluster_mask" in k:
                print(os.path.basename(d), {c: round(sum(v) / len(v) / 1e6, 3) for c, v in cs.items()}, "(millions)")
PY
