#!/bin/bash
# round 3, call 27: 7 and 8 cull workgroups per CU (the kernel is at 64 VGPRs now: 8 waves per SIMD fit)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so
for rep in 1 2; do
for wg in 6 7 8; do
  NV_CC_BLOCKS_PER_CU=$wg timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('wg/CU $wg', 'pass us %.2f' % (d['ms_per_step']*1e3), 'cull us %.2f' % r['kernel_avg_us'], 'scatter us %.2f' % r['scatter_kernel_avg_us'], 'visible', d['config']['visible_total'])"
done
done
