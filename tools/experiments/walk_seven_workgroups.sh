#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== $1 blocks=$2 scale=$3"; NV_DEAL_SCALE=$3 NV_CC_BLOCKS_PER_CU=$2 NV_LIBRARY_PATH=$R/$1 timeout 900 python tools/bench_configs.py --iters 60 --only 3b_chain,frame_py,3a_dense 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:40], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('step_us','cull_us','cluster_cull_us','late_cluster_cull_us','frame_us','parity')})"; }
for round in 1 2; do
run variants/cc_e_walk6n.so 6 100
run variants/cc_e_walk6n.so 6 0
run variants/cc_e_walk7.so 7 100
done
