#!/bin/bash
# usage (through gpurun): bash tools/refresh_essentials.sh <tag> — the numbers DESIGN.md / profiles/ quote, in one call:
# PMC traffic of bench.py's kernels, the default bench line (reads that traffic), kernel-trace stats of the same command,
# and the other BASELINE configs.  Everything lands in gpurun_out/refresh_<tag>/.
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/refresh_$tag; mkdir -p $out
cd $R
bash tools/pmc_traffic.sh $tag > $out/pmc_traffic.log 2>&1
cp gpurun_out/pmct_$tag/pmc_traffic_summary.json $out/pmc_traffic.json && cp $out/pmc_traffic.json profiles/${tag}_pmc_traffic.json
python bench.py 2> $out/bench_default.err | tail -1 > $out/bench_default.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $out/kt -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/kt.log 2>&1)
python tools/bench_configs.py --iters 100 2> $out/bench_configs.err | grep "^{" > $out/bench_configs.jsonl
tail -c 600 $out/bench_default.json; echo; wc -l $out/bench_configs.jsonl
