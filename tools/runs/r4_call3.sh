#!/bin/bash
# round 4, call 3: the HiZ probe's selection logic restated (aliased weightless texels, frexp mip level, integer clamps: 338 -> 293 VALU per
# probe, no floating-point operation of the reference touched): parity suite, then the occlusion stage's time against round 3's library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4c; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -X faulthandler -m pytest tests/test_special_values.py tests/test_gpu_parity.py tests/test_lane_form.py tests/test_golden.py tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for rep in 1 2; do
for so in variants/r3_exp.so niagara_amd/libniagara_vis_exp.so; do
  echo "== $so"
  NV_LIBRARY_PATH=$PWD/$so timeout 400 python tools/bench_configs.py --iters 30 --only frame_py,4,2l 2>>$O/bc.err | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  ', d['config'][:40], {k:round(v,1) for k,v in d.items() if isinstance(v,(int,float)) and k.endswith('_us')}, d.get('parity'))"
done
done
