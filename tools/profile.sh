#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/profile.sh <tag> [ENV=VAL ...]
# kernel-trace stats + separate PMC passes of bench.py; summaries land in gpurun_out/prof_<tag>/
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { env "$@" python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --overlap-streams 0; }
env "$@" rocprofv3 --kernel-trace --stats -f csv -d $out/kt -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 > $out/kt.log 2>&1
# the same with three independent passes in flight (the `throughput_overlapped` side figure of the bench line)
env "$@" rocprofv3 --kernel-trace --stats -f csv -d $out/kt3 -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 --streams 3 > $out/kt3.log 2>&1
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $pmc -f csv -d $out/pmc$i -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --overlap-streams 0 > $out/pmc$i.log 2>&1 || echo "pmc pass $i failed: $pmc"
done
python3 - <<PY
import csv, glob, collections, os
out = "$out"
# kernel stats
for sub, what in (("kt", "one pass after the other on one stream (bench.py's default: the mode the bench line's value is measured in)"), ("kt3", "--streams 3: three independent passes in flight")):
    for f in glob.glob(out + "/" + sub + "/**/*kernel_stats.csv", recursive=True):
        print("== kernel stats,", what)
        print(open(f).read()[:3000])
# PMC averages per kernel
for d in sorted(glob.glob(out + "/pmc*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "cluster" in k or "drawcull" in k or "reduce" in k:
                print(os.path.basename(d), k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
