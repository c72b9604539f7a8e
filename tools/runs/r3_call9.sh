#!/bin/bash
cd "$GRAFT_REPO_ROOT"
EXP=$PWD/niagara_amd/libniagara_vis_exp.so
for seg in 0 1 2; do for bpc in 8 10 16 24; do
  dm=$((seg << 27))
  echo -n "seg code $seg blocks/CU $bpc: "
  NV_LIBRARY_PATH=$EXP NV_DEBUG_MODE=$dm NV_TASK_BLOCKS_PER_CU=$bpc timeout 300 python tools/bench_configs.py --iters 40 --only task 2>/dev/null | grep -o '"call_us": [0-9.]*'
done; done
