#!/bin/bash
# usage (through gpurun): bash tools/round_numbers.sh <tag> — the numbers DESIGN.md / README / profiles/ quote, on one box in one call:
# PMC traffic, the default + five 20-step bench lines, the single-stream kernel trace + instruction counters of the same command, every other config
# (tools/bench_configs.py), the frame's kernel trace.  Everything lands in gpurun_out/final_<tag>/ — copy what is quoted into profiles/<tag>_*.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
tag=${1:-r05}
out=$R/gpurun_out/final_$tag; mkdir -p $out
bash tools/pmc_traffic.sh $tag > $out/pmc_traffic.log 2>&1
cp gpurun_out/pmct_$tag/pmc_traffic_summary.json $out/pmc_traffic.json && cp $out/pmc_traffic.json profiles/${tag}_pmc_traffic.json
python bench.py 2> $out/bench_default.err | tail -1 > $out/bench_default.json
for i in 1 2 3 4 5; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 >> $out/bench_20.jsonl; done
bash tools/kt.sh ${tag}bench -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 > $out/kt.txt 2>&1
bash tools/pmc.sh ${tag}insts "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --overlap-streams 0 > $out/pmc_insts.txt 2>&1
python tools/bench_configs.py --iters 100 2> $out/bench_configs.err | grep "^{" > $out/bench_configs.jsonl
bash tools/trace_config.sh frame_py ${tag}frame > /dev/null 2>&1; cp gpurun_out/trace_${tag}frame.txt $out/trace_frame.txt 2>/dev/null
cat $out/kt.txt | head -8; cat $out/pmc_insts.txt | tail -4
python - <<PY
import json
for l in open("$out/bench_20.jsonl"):
    d=json.loads(l); print("20 steps:", round(d["ms_per_step"]*1e3,2), round(d["roofline"]["pass_frac"],3), round(d["roofline"]["frac"],3), d.get("parity"))
d=json.load(open("$out/bench_default.json")); print("default:", round(d["ms_per_step"]*1e3,2), round(d["roofline"]["pass_frac"],3), round(d["roofline"]["frac"],3), d.get("parity"), d["roofline"]["traffic"])
PY
wc -l $out/bench_configs.jsonl
