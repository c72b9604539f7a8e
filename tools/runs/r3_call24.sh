#!/bin/bash
# round 3, call 24: A/B of the product library against the library saved before a change (niagara_amd/libniagara_vis_base.so), one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for lib in libniagara_vis_base.so libniagara_vis.so; do
    NV_LIBRARY_PATH=niagara_amd/$lib timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-26s' % '$lib', 'pass us %.2f' % (d['ms_per_step']*1e3), 'cull us %.2f' % r['kernel_avg_us'], 'scatter us %.2f' % r['scatter_kernel_avg_us'], 'visible', d['config']['visible_total'])"
  done
done
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for lib in libniagara_vis_base.so libniagara_vis.so; do
  out=$R/gpurun_out/ab_$lib; rm -rf $out
  NV_LIBRARY_PATH=$R/niagara_amd/$lib timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES -f csv -d $out -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --overlap-streams 0 > $out.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cluster_mask" in row["Kernel_Name"] and "4, false, false" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("$lib", {k: round(sum(v)/len(v)/156250, 1) for k, v in acc.items() if k != "SQ_WAVES"}, "per command")
PY
done
