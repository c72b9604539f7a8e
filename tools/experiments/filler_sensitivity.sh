#!/bin/bash
# filler_sensitivity.sh — round 3: how does the headline cull launch react to MORE instructions per command?
#   bash tools/experiments/filler_sensitivity.sh build      (here: cross-compiles niagara_amd/libniagara_vis_f{s,v}{8,16}.so)
#   bash tools/experiments/filler_sensitivity.sh run        (on the GPU box, through gpurun)
# The libraries are the product with n register-free scalar (s_cmp_eq_u32 0, 0) or vector (v_nop) instructions more per task
# command in the filter pass's issue step (clustercull.hip, NV_FILLER_S / NV_FILLER_V), or 200 scalar / 100 vector instructions more per WAVE
# in front of its first segment (NV_FILLER_PS / NV_FILLER_PV); bench.py measures each with the same inputs.
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R/niagara_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function"
OTHERS="build/drawcull.o build/submit.o build/depthreduce.o build/trianglecull.o build/bounds.o build/context.o build/host.o"
if [ "$1" = build ]; then
  mkdir -p build/filler
  for v in S=8 S=16 V=8 V=16 PS=200 PV=100; do
    tag=$(echo $v | tr 'SVP=' 'svp_' | tr -d _)
    hipcc $FLAGS -DNV_FILLER_$v -c clustercull.hip -o build/filler/cc_$tag.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libniagara_vis_f$tag.so build/filler/cc_$tag.o $OTHERS && echo built libniagara_vis_f$tag.so
  done
else
  cd $R
  for rep in 1 2; do
    for lib in libniagara_vis.so libniagara_vis_fs8.so libniagara_vis_fs16.so libniagara_vis_fv8.so libniagara_vis_fv16.so libniagara_vis_fps200.so libniagara_vis_fpv100.so; do
      NV_LIBRARY_PATH=niagara_amd/$lib timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-26s' % '$lib', 'pass us %.2f' % (d['ms_per_step']*1e3), 'cull us %.2f' % r['kernel_avg_us'], 'scatter us %.2f' % r['scatter_kernel_avg_us'], 'visible', d['config']['visible_total'])"
    done
  done
  cd /tmp && export TMPDIR=/tmp
  for lib in libniagara_vis.so libniagara_vis_fs8.so libniagara_vis_fv8.so; do
    out=$R/gpurun_out/filler_$lib; rm -rf $out
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
fi
