#!/bin/bash
# round 3, call 31: the lane-per-cluster form for dense passes over a cache-resident pool (early without bits, late first stage): parity, then
# A/B against the one-command-per-wave direct form (NV_DIRECT=2 of the experiments build = NV_OPT_CULL_FORM 3) on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_special_values.py tests/test_gpu_configs.py tests/test_plain_loads.py -x -q -m gpu 2>&1 | tail -3
export NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so
for rep in 1 2; do
for d in "" 2; do
  echo "== NV_DIRECT=$d"
  NV_DIRECT=$d timeout 400 python tools/bench_configs.py --iters 30 --only 3b_fused,frame_py 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print('  ', d['config'][:44], {k:round(v,1) for k,v in d.items() if k in ('step_us','cluster_cull_us','cluster_scatter_us','frame_us','early_cluster_cull_us','late_cluster_cull_us','late_cluster_hiz_us','late_cluster_scatter_us')}, d['parity'])"
done
done
