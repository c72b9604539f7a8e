#!/bin/bash
# round 3, GPU call 1: regression suite + the new plumbing + first numbers in the new regime
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; tail -c 600 $O/bench20.err
timeout 300 python bench.py > $O/bench200.json 2> $O/bench200.err
timeout 600 python tools/bench_configs.py --iters 30 --only frame,frame_contract > $O/frame.jsonl 2> $O/frame.err; tail -c 800 $O/frame.err
timeout 600 python tools/bench_configs.py --iters 30 --only 2,2l,4,task,3a,3a_dense > $O/configs.jsonl 2> $O/configs.err; tail -c 800 $O/configs.err
# pyramid A/B on the experiments build: tail inside the first launch (0) vs one launch per stage (1)
for m in 0 1; do NV_LIBRARY_PATH=$PWD/niagara_amd/libniagara_vis_exp.so NV_PYRAMID_MODE=$m timeout 300 python tools/bench_configs.py --iters 40 --only 4 > $O/pyr_mode$m.jsonl 2>&1; done
timeout 120 ./examples/shard_driver --steps 50 > $O/shard_driver.log 2>&1; tail -3 $O/shard_driver.log
cat $O/bench20.json | cut -c 1-1500
