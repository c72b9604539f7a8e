#!/bin/bash
# round 4, call 1: the lane-per-cluster dense forms (r3's patch re-applied) — the new deterministic tests, then the four-file process
# that aborted in round 3, on poisoned device memory, logs kept; then the same through the experiments build (scratch canaries); A/B timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4a; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -X faulthandler -m pytest tests/test_lane_form.py -x -v -m gpu -p no:cacheprovider > $O/lane.log 2>&1; echo "lane rc=$?"; tail -5 $O/lane.log
for i in 1 2; do
  timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_special_values.py tests/test_gpu_configs.py tests/test_plain_loads.py tests/test_lane_form.py -x -v -m gpu -p no:cacheprovider > $O/four_$i.log 2>&1; echo "four_$i rc=$?"; tail -4 $O/four_$i.log
done
NV_LIBRARY_PATH=$PWD/niagara_amd/libniagara_vis_exp.so timeout 900 python -X faulthandler -m pytest tests/test_lane_form.py tests/test_gpu_parity.py tests/test_special_values.py -x -v -m gpu -p no:cacheprovider > $O/exp.log 2>&1; echo "exp rc=$?"; tail -4 $O/exp.log
export NV_LIBRARY_PATH=$PWD/niagara_amd/libniagara_vis_exp.so
for rep in 1 2; do
for d in "" 2; do
  echo "== NV_DIRECT=$d"
  NV_DIRECT=$d timeout 400 python tools/bench_configs.py --iters 30 --only 3b_fused,frame_py,3a_dense 2>>$O/bc.err | grep "^{" >> $O/bc_$d.jsonl
done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4a/bc_*.jsonl')):
    print(f)
    for l in open(f):
        d=json.loads(l); print('  ', d['config'][:40], {k:round(v,1) for k,v in d.items() if isinstance(v,(int,float)) and k.endswith('_us')}, d.get('parity'))
PY
