#!/bin/bash
# round 4, call 2: lane forms A/B (call 1 set NV_DIRECT to the empty string = forced filter form: nothing was measured), scratch
# canaries through the experiments build, what a dependent launch costs in wall time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 120 tools/launch_cost 200 > $O/launch_cost.jsonl 2>&1; cat $O/launch_cost.jsonl
NV_LIBRARY_PATH=$PWD/niagara_amd/libniagara_vis_exp.so timeout 900 python -X faulthandler -m pytest tests/test_lane_form.py tests/test_gpu_parity.py tests/test_special_values.py -x -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_extension_is_the_hip_library --deselect tests/test_gpu_parity.py::test_environment_cannot_change_results > $O/exp.log 2>&1; echo "exp rc=$?"; tail -4 $O/exp.log
export NV_LIBRARY_PATH=$PWD/niagara_amd/libniagara_vis_exp.so
for rep in 1 2; do
  echo "== auto (lane forms)"
  env -u NV_DIRECT timeout 400 python tools/bench_configs.py --iters 30 --only 3b_fused,frame_py,4 2>>$O/bc.err | grep "^{" >> $O/bc_auto.jsonl
  echo "== NV_DIRECT=2 (direct, one command per wave)"
  NV_DIRECT=2 timeout 400 python tools/bench_configs.py --iters 30 --only 3b_fused,frame_py,4 2>>$O/bc.err | grep "^{" >> $O/bc_2.jsonl
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/bc_*.jsonl')):
    print(f)
    for l in open(f):
        d=json.loads(l); print('  ', d['config'][:40], {k:round(v,1) for k,v in d.items() if isinstance(v,(int,float)) and k.endswith('_us')}, d.get('parity'))
PY
tail -5 $O/bc.err
