#!/bin/bash
# usage (through gpurun): bash tools/experiments/undecided_cost.sh — round 6: what the lanes the certified test leaves undecided cost the lane-parallel forms (the packed
# direct walk, the lane-per-set-bit early pass): NV_DEBUG_MODE bit 29 of the experiments build skips their reference arithmetic (results then differ: timing only).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in 1 2; do
for m in 0 536870912; do
  echo "== NV_DEBUG_MODE=$m (round $round)"
  NV_DEBUG_MODE=$m NV_LIBRARY_PATH=$R/niagara_amd/libniagara_vis_exp.so timeout 900 python tools/bench_configs.py --iters 60 --allow-mismatch --only ${1:-3a_dense,frame_py,3b_fused} 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:40], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_us') or k in ('visible','parity')})"
done
done
