#!/bin/bash
# round 3, call 22: sensitivity of the headline cull launch to 8 more scalar / 8 more vector instructions per command (experiments build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so
for rep in 1 2; do
for mode in 0 134217728 268435456 402653184; do
  NV_DEBUG_MODE=$mode timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('mode $mode', 'pass us %.2f' % (d['ms_per_step']*1e3), 'cull us %.2f' % r['kernel_avg_us'], 'scatter us %.2f' % r['scatter_kernel_avg_us'], 'visible', d['config']['visible_total'])"
done
done
cd /tmp && export TMPDIR=/tmp
for mode in 0 134217728 268435456; do
  NV_DEBUG_MODE=$mode timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES -f csv -d $GRAFT_REPO_ROOT/gpurun_out/salu_$mode -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --overlap-streams 0 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/salu_$mode/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cluster_mask" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("mode $mode", {k: round(sum(v)/len(v)/156250, 1) for k, v in acc.items() if k != "SQ_WAVES"}, "per command")
PY
done
