#!/bin/bash
# round 3, call 17: the bit-expanding early form (cluster_bits_kernel): parity, then the frame
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_expanding or cone_test_is_exact or dense_passes or two_frame or pinned_kernel or flag_and_postpass" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "frame" 2>&1 | tail -5
timeout 600 python tools/bench_configs.py --iters 30 --only frame 2>&1 | tail -4
