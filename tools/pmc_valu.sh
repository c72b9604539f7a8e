#!/bin/bash
# usage (through gpurun): bash tools/pmc_valu.sh <tag> — SQ_INSTS_VALU / SQ_INSTS_SALU / busy counters per launch of the issue-bound kernels
# (occlusion stage, direct-form cull launch, triangle cull, drawcull decide) over tools/bench_configs.py: one rocprofv3 --pmc pass,
# kernel-trace only.  Writes gpurun_out/pmcv_<tag>/valu_counters.json — copy it to profiles/rNN_valu_counters.json (tools/valu_roofline.py reads the newest).
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmcv_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CFGS=${2:-frame_py,n4,2,3a_dense,3b_chain}
( cd $R && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -f csv -d $out/pmc -- python tools/bench_configs.py --iters 20 --only $CFGS > $out/run.log 2>&1 ) || echo "pmc pass failed"
# (round 6) the EXECUTED class mix: the per-class instruction counters this rocprofv3 offers, three per pass
( cd $R && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 -f csv -d $out/pmc_cls1 -- python tools/bench_configs.py --iters 20 --only $CFGS > $out/run_cls1.log 2>&1 ) || echo "class pass 1 failed"
( cd $R && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 -f csv -d $out/pmc_cls2 -- python tools/bench_configs.py --iters 20 --only $CFGS > $out/run_cls2.log 2>&1 ) || echo "class pass 2 failed"
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {"command": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES (+ two passes of SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F32 / _CVT / _INT32) -- python tools/bench_configs.py --iters 20 --only $CFGS",
       "units": "per launch (mean over the profiled launches); wave-instructions", "kernels": {}}
for k, cs in acc.items():
    if any(x in k for x in ("cluster_hiz", "cluster_mask", "cluster_bits", "trianglecull", "draw_decide", "draw_scatter", "cluster_scatter", "reduce_")):
        res["kernels"][k.split("(")[0]] = dict({c: sum(v) / len(v) for c, v in cs.items()}, launches=len(next(iter(cs.values()))))
import sys
sys.path.insert(0, "$R/tools")
import valu_roofline as V
res["class_mix"] = V.all_mixes()  # static VALU class counts of the kernels' compiled text (the build the counters were taken on)
json.dump(res, open("$out/valu_counters.json", "w"), indent=1)
for k, v in res["kernels"].items():
    print("%-80s" % k[-80:], {c: round(x) for c, x in v.items()})
PY
