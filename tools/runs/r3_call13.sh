#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/kt.sh f13 -- python tools/bench_configs.py --iters 20 --only frame_py 2>&1 | grep -E "draw_|stats"
grep -o '"frame_us": [0-9.]*, "early_drawcull_us": [0-9.]*' gpurun_out/kt_f13/run.log
timeout 300 python tools/bench_configs.py --iters 30 --only frame_py 2>/dev/null | grep -o '"frame_us": [0-9.]*, "early_drawcull_us": [0-9.]*, "early_cluster_cull_us": [0-9.]*, "early_cluster_scatter_us": [0-9.]*, "pyramid_us": [0-9.]*, "late_drawcull_us": [0-9.]*'
