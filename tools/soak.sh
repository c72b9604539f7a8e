#!/bin/bash
# usage (through gpurun): bash tools/soak.sh [seconds=420] — five randomised soaks side by side on one GPU, fresh seeds, every result against the oracle
# (fuzz_frames, fuzz_sizes, fuzz_taskcull, fuzz_passes; the GPU suite runs fixed-seed samples of the same tools)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
out=$R/gpurun_out/${SOAK_TAG:-r6soak}; mkdir -p $out
T=${1:-420}
python tools/experiments/fuzz_frames.py $T ${SOAK_SEED:-605000} > $out/frames.txt 2>&1 &
FUZZ_CULL_FORM=${SOAK_FORM:-5} python tools/experiments/fuzz_frames.py $T $((${SOAK_SEED:-605000} + 5000)) > $out/frames_form2.txt 2>&1 &
python tools/experiments/fuzz_sizes.py $T $((${SOAK_SEED:-605000} + 2000)) > $out/sizes.txt 2>&1 &
python tools/experiments/fuzz_taskcull.py $T $((${SOAK_SEED:-605000} - 4000)) > $out/taskcull.txt 2>&1 &
python tools/experiments/fuzz_passes.py $T $((${SOAK_SEED:-605000} + 1000)) > $out/passes.txt 2>&1 &
wait
for f in frames frames_form2 sizes taskcull passes; do echo "== $f"; tail -3 $out/$f.txt; done
