#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in 1 2; do for b in 6 5 4; do
  echo "== NV_CC_BLOCKS_PER_CU=$b (round $round)"
  NV_CC_BLOCKS_PER_CU=$b NV_LIBRARY_PATH=$R/niagara_amd/libniagara_vis_exp.so timeout 900 python tools/bench_configs.py --iters 60 --only 3b_chain,frame_py,3a_dense 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:40], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('step_us','cull_us','cluster_cull_us','late_cluster_cull_us','frame_us')})"
done; done
