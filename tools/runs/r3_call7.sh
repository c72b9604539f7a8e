#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3g; mkdir -p $O
EXP=$PWD/niagara_amd/libniagara_vis_exp.so
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['config'][:50], {k: round(v, 1) for k, v in d.items() if k.endswith('_us')})"; }
timeout 300 python tools/bench_configs.py --iters 30 --only 2,2_fused | show
for t in 1 2; do echo "scatter tiles per CU $t"; NV_LIBRARY_PATH=$EXP NV_SCATTER_TILES_PER_CU=$t timeout 300 python tools/bench_configs.py --iters 30 --only 3a_dense,frame_py 2>/dev/null | show; done
