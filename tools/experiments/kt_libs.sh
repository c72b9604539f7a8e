#!/bin/bash
# usage (through gpurun): bash tools/experiments/kt_libs.sh [-c configs] lib ... — kernel-trace averages of the drawcull kernels per library ("product" or a
# tools/build_variant.sh name) over tools/bench_configs.py lines, plus the lines' own wall figures
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cfgs="2_fused,2l,frame"
while getopts "c:" o; do case $o in c) cfgs=$OPTARG;; esac; done
shift $((OPTIND - 1))
for l in "$@"; do
  if [ $l = product ]; then e=NV_X=1; else e=NV_LIBRARY_PATH=$R/variants/$l.so; fi
  rm -rf gpurun_out/kt_lib_$l
  bash tools/kt.sh lib_$l $e -- python tools/bench_configs.py --iters 40 --only $cfgs 2>&1 | grep -i "draw_\|==" | grep -v split | cut -c1-160
  grep "^{" gpurun_out/kt_lib_$l/run.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:30], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('step_us','kernel_us','frame_us','parity')})"
done
