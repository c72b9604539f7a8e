#!/bin/bash
# usage: bash tools/ab_blocks.sh "<blocks per CU list>" lib.so ... — bench.py per NV_CC_BLOCKS_PER_CU and library build
bl=$1; shift
for so in "$@"; do for b in $bl; do
  NV_CC_BLOCKS_PER_CU=$b NV_LIBRARY_PATH=$PWD/$so timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', 'blocks/CU', $b, 'step', round(d['ms_per_step']*1e3,2), 'K1', round(d['roofline']['kernel_avg_us'],2))"
done; done
