#!/bin/bash
# usage: bash tools/ab_all.sh lib1.so lib2.so ... — bench.py (10 M) + late pass + 100 M per library build, same box
for so in "$@"; do
  NV_LIBRARY_PATH=$PWD/$so timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', '10M step', round(d['ms_per_step']*1e3,2), 'K1', round(d['roofline']['kernel_avg_us'],2))"
  bash tools/ab_cfg_lib1.sh "4,big" $so
done
