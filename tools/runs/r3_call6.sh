#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3f; mkdir -p $O
EXP=$PWD/niagara_amd/libniagara_vis_exp.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "depthreduce" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for m in 0 1; do bash tools/kt.sh pyr$m NV_LIBRARY_PATH=$EXP NV_PYRAMID_MODE=$m -- python tools/bench_configs.py --iters 40 --only 4 > $O/kt_pyr$m.txt 2>&1; grep -E "reduce|stats" $O/kt_pyr$m.txt; grep -o '"pyramid_us": [0-9.]*' gpurun_out/kt_pyr$m/run.log; done
