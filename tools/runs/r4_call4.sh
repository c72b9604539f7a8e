#!/bin/bash
# round 4, call 4: the late pass fused into the lane kernel (cull + occlusion probes + epilogue in one launch): parity, then A/B of the three
# late forms on one box (NV_LANE_LATE 0 = one command per wave + cluster_hiz_kernel, 1 = lane first stage + cluster_hiz_kernel, 2 = fused)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4d; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -X faulthandler -m pytest tests/test_lane_form.py tests/test_gpu_parity.py tests/test_special_values.py tests/test_golden.py tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
export NV_LIBRARY_PATH=$PWD/niagara_amd/libniagara_vis_exp.so
for rep in 1 2; do
for mode in 0 1 2; do
  echo "== NV_LANE_LATE=$mode"
  NV_LANE_LATE=$mode timeout 400 python tools/bench_configs.py --iters 30 --only frame_py 2>>$O/bc.err | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  ', d['config'][:40], {k:round(v,1) for k,v in d.items() if isinstance(v,(int,float)) and k.endswith('_us')}, d.get('parity'))"
done
done
tail -3 $O/bc.err
