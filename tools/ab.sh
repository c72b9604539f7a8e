#!/bin/bash
# usage: bash tools/ab.sh variants/libA.so variants/libB.so ... — bench.py + the 100 M config per library build
for so in "$@"; do
  echo "== $so"
  NV_LIBRARY_PATH=$PWD/$so timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,1), round(d['ms_per_step']*1e3,2), round(d['roofline']['kernel_avg_us'],2), d['config']['visible_total'])"
  NV_LIBRARY_PATH=$PWD/$so timeout 100 python tools/bench_configs.py --only big 2>&1 | tail -1 | cut -c1-150
done
