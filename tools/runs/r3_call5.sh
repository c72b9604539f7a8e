#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e; mkdir -p $O
EXP=$PWD/niagara_amd/libniagara_vis_exp.so
for sl in 0 2 4 5 6 7; do
  dm=$((sl << 26))
  NV_LIBRARY_PATH=$EXP NV_DEBUG_MODE=$dm timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 --riding-scatter 1 > $O/b_$sl.json 2> $O/b_$sl.err
  python3 -c "
import json
d=json.loads(open('$O/b_$sl.json').read().strip().splitlines()[-1]); r=d['roofline']
print('pre-sleep $sl x 3.4us: ms_per_step %.5f kernel %.2f us visible %d' % (d['ms_per_step'], r['kernel_avg_us'], d['config']['visible_total']))"
done
