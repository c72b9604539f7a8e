#!/bin/bash
# usage: bash tools/ab_modes.sh "<modes>" lib1.so lib2.so ... — bench.py K1 time per NV_DEBUG_MODE and library build
modes=$1; shift
for so in "$@"; do for m in $modes; do
  NV_DEBUG_MODE=$m NV_LIBRARY_PATH=$PWD/$so timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', 'mode', $m, 'step', round(d['ms_per_step']*1e3,2), 'K1', round(d['roofline']['kernel_avg_us'],2))"
done; done
