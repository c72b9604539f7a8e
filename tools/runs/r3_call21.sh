#!/bin/bash
# round 3, call 21: A/B of the list expansion of cluster_bits_kernel (experiments build, one box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
for mode in 0 67108864; do
  echo "== NV_DEBUG_MODE=$mode"
  NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so NV_DEBUG_MODE=$mode timeout 300 python tools/bench_configs.py --iters 30 --only frame_py 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:round(d[k],1) for k in ('frame_us','early_cluster_cull_us','early_cluster_scatter_us','late_cluster_cull_us','late_cluster_hiz_us')}, d['parity'])"
done
done
