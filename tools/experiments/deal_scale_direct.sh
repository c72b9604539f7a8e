#!/bin/bash
# usage (through gpurun): bash tools/experiments/deal_scale_direct.sh [configs] [scales] — round 6: the weighted dealing's scale (ClusterArgs::dealScale; NV_DEAL_SCALE in the
# experiments build) for the DIRECT form of the cull launch.  The weights were calibrated on the sparse filter form; the packed walk's waves live 18-23 us and the
# later generations end 3 us after the first (tools/wave_timeline.py with NV_DIRECT=1).  Prints cull / scatter / step times per scale.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cfgs=${1:-3b_chain,frame_py,3a_dense}
for round in 1 2; do
for s in ${2:-100 200 300 400 0}; do
  echo "== NV_DEAL_SCALE=$s (round $round)"
  NV_DEAL_SCALE=$s NV_LIBRARY_PATH=$R/niagara_amd/libniagara_vis_exp.so timeout 900 python tools/bench_configs.py --iters 60 --only $cfgs 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:40], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_us') or k in ('visible','parity')})"
done
done
