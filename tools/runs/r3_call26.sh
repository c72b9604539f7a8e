#!/bin/bash
# round 3, call 26: the scatter's LDS-staged step: parity, then A/B against the owner loop (experiments build, NV_DEBUG_MODE bit 30) on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_special_values.py tests/test_plain_loads.py -x -q -m gpu 2>&1 | tail -3
export NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so
for rep in 1 2; do
for mode in 0 1073741824; do
  echo "== NV_DEBUG_MODE=$mode"
  NV_DEBUG_MODE=$mode timeout 400 python tools/bench_configs.py --iters 30 --only 3a,3a_dense,3a_half,frame_py 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print('  ', d['config'][:44], {k:round(v,1) for k,v in d.items() if k in ('cull_us','scatter_us','step_us','frame_us','early_cluster_scatter_us','late_cluster_scatter_us','early_cluster_cull_us')}, d['parity'])"
done
done
