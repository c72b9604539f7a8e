#!/bin/bash
# tools/ab.sh — A/B runs on ONE box in ONE gpurun call (box-to-box variance is ~5 %).
# usage: bash tools/ab.sh [-r rounds] [-s steps] [-m "<NV_DEBUG_MODE values>"] [-k "<workgroups per CU>"] [-c "<bench_configs keys>"] lib.so ...
#   every lib (NV_LIBRARY_PATH; builds kept under the git-ignored variants/) x every mode x every blocks-per-CU value goes
#   through bench.py `rounds` times, interleaved; -c adds tools/bench_configs.py --only <keys> per lib.
#   -m / -k only act on a build with -DNV_EXPERIMENTS (libniagara_vis_exp.so): the product library reads no environment.
rounds=2; steps=200; modes="0"; blocks="6"; cfgs=""
while getopts "r:s:m:k:c:" o; do
  case $o in r) rounds=$OPTARG;; s) steps=$OPTARG;; m) modes=$OPTARG;; k) blocks=$OPTARG;; c) cfgs=$OPTARG;; esac
done
shift $((OPTIND - 1))
for round in $(seq $rounds); do
for so in "$@"; do for m in $modes; do for b in $blocks; do
  NV_DEBUG_MODE=$m NV_CC_BLOCKS_PER_CU=$b NV_LIBRARY_PATH=$PWD/$so timeout 100 python bench.py --steps $steps --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$so', 'mode', $m, 'wg/CU', $b, 'round', $round, '| G/s', round(d['value']/1e9,1), 'step_us', round(d['ms_per_step']*1e3,2), 'cull_us', round(r['kernel_avg_us'],2), 'scatter_us', round(r['scatter_kernel_avg_us'],2), 'frac', round(r['frac'],3), 'visible', d['config']['visible_total'])"
done; done; done
done
if [ -n "$cfgs" ]; then
for so in "$@"; do
  echo "== $so (configs $cfgs)"
  NV_LIBRARY_PATH=$PWD/$so timeout 600 python tools/bench_configs.py --iters 60 --only $cfgs 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:40], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_us') or k in ('visible','late_visible','parity','candidates')})"
done
fi
