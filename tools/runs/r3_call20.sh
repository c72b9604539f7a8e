#!/bin/bash
# round 3, call 20: lane-per-cluster early form for every dense early pass: parity, dense configs, frame
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_expanding or cone_test_is_exact or dense_passes or two_frame or pinned_kernel or flag_and_postpass or frustum_filter or frustum_coeff or ragged or special" 2>&1 | tail -5
for c in 3a_dense 3a_half 3b frame; do
  timeout 300 python tools/bench_configs.py --iters 30 --only $c 2>&1 | tail -1 | cut -c1-900
done
for c in 3a_dense 3a_half; do
  echo "== pinned one-command-per-wave direct form ($c)"
  NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so NV_DIRECT=2 timeout 300 python tools/bench_configs.py --iters 30 --only $c 2>&1 | tail -1 | cut -c1-600
done
