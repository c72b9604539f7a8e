#!/bin/bash
# round 3, call 19: instruction counts / VALU busy of cluster_bits_kernel at frame scale
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/bits_pmc; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  ( cd $R && timeout 150 rocprofv3 --kernel-trace --pmc $pmc -f csv -d $out/p$i -- python tools/bench_configs.py --iters 8 --only frame_py > $out/p$i.log 2>&1 ) || echo "pass $i failed: $(grep -i -m2 'error' $out/p$i.log)"
done
python3 - <<PY
import csv, glob, collections, os
for d in sorted(glob.glob("$out/p*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if any(x in k for x in ("cluster_bits", "cluster_hiz")):
                print(os.path.basename(d), "%-50s" % k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        t = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            t[row["Kernel_Name"].split("(")[0][-50:]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, v in t.items():
            if "cluster_bits" in k: print(os.path.basename(d), k, "avg ns", sum(v) / len(v), "n", len(v))
PY
