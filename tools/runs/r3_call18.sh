#!/bin/bash
# round 3, call 18: the pipelined bit-expanding early form: parity, then a sweep of its grid / iteration size at frame scale
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_expanding or cone_test_is_exact or dense_passes or two_frame or pinned_kernel or flag_and_postpass" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "frame" 2>&1 | tail -3
for cfg in "4 128" "4 64" "3 128" "5 64" "6 64" "8 128" "8 64" "2 128"; do
  set -- $cfg
  echo "== blocks/CU $1, commands/iteration $2"
  NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so NV_BITS_BLOCKS=$1 NV_BITS_COMMANDS=$2 timeout 300 python tools/bench_configs.py --iters 30 --only frame_py 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:round(d[k],1) for k in ('frame_us','early_cluster_cull_us','early_cluster_scatter_us','late_cluster_cull_us','late_cluster_hiz_us')}, d['parity'])"
done
