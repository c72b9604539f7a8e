#!/bin/bash
# round 3, call 23: fewer waves with more commands each (per-wave fixed instructions are ~45 % of the launch's instruction count) x ring depth
cd ${GRAFT_REPO_ROOT:-/root/repo}
export NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so
for wg in 6 5 4 3 2; do
for mode in 0 65536; do
  NV_CC_BLOCKS_PER_CU=$wg NV_DEBUG_MODE=$mode timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('wg/CU $wg deep-ring-bit $mode', 'pass us %.2f' % (d['ms_per_step']*1e3), 'cull us %.2f' % r['kernel_avg_us'], 'scatter us %.2f' % r['scatter_kernel_avg_us'], 'visible', d['config']['visible_total'])"
done
done
