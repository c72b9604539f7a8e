#!/bin/bash
# round 3, call 28: nv_taskcull's early pass as cull launch + payload launch: parity, then A/B against the one-launch kernel (experiments build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_special_values.py tests/test_gpu_configs.py -x -q -m gpu -k "taskcull or task or special or two_frame or dense_passes" 2>&1 | tail -3
export NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so
for rep in 1 2; do
for one in 0 1; do
  NV_TASKCULL_ONE_LAUNCH=$one timeout 300 python tools/bench_configs.py --iters 30 --only task 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('one-launch=$one', d['config'][:50], {k:round(v,2) for k,v in d.items() if k.endswith('_us') or k=='frac'}, d['parity'])"
done
done
