#!/usr/bin/env python3
"""Randomised soak: many small scenes with random sizes, viewports, flags, layouts and fusion options through three
frames of the HIP passes, every buffer of every phase compared with the oracle (same comparison as
tests/test_gpu_parity.py::test_two_frame_protocol, wider parameter space).  Needs a GPU.

    python tools/experiments/fuzz_frames.py [seconds=120] [first_seed=1000]
    FUZZ_CULL_FORM=2 ...   pins NV_OPT_CULL_FORM (2: the early passes with visibility bits run cluster_bits_kernel from the first frame on)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402  (checker)
import passes  # noqa: E402
import gpu_passes as G  # noqa: E402
from niagara_amd import pipeline as P  # noqa: E402
from scenes import make_scene, random_case  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = P.Context(0)
if os.environ.get("FUZZ_CULL_FORM"):
    ctx.set_option(P.NV_OPT_CULL_FORM, int(os.environ["FUZZ_CULL_FORM"]))
t0 = time.time()
runs = bad = 0
while time.time() - t0 < budget:
    kw, flags, use_soa, fused = random_case(seed)
    if os.environ.get("FUZZ_VERBOSE"):
        print("seed", seed, kw, flags, use_soa, fused, flush=True)
    scene = make_scene(**kw)
    ctx.set_option(P.NV_OPT_DRAW_RECORDS, seed % 3)      # drawcull: the early pass's request order ...
    ctx.set_option(P.NV_OPT_TASK_EMIT, seed // 3 % 3)    # ... and the task pass's emission form (2: the list form, fed by the decide launch's records)
    fo = passes.run_frames(oracle, scene, flags, frames=3)
    fg = G.run_frames(ctx, scene, flags, frames=3, use_soa=use_soa, fused=fused)
    ok = True
    for f, (a, b) in enumerate(zip(fo, fg)):
        if a["pyramid"].tobytes() != b["pyramid"].tobytes():
            ok = False
            print("MISMATCH seed", seed, "frame", f, "pyramid", kw, flags, use_soa, fused)
        for phase in ("early", "late"):
            for key in ("count4", "commands", "cc4", "cib", "dvb", "mvb"):
                if a[phase][key].tobytes() != b[phase][key].tobytes():
                    ok = False
                    print("MISMATCH seed", seed, "frame", f, phase, key, kw, flags, use_soa, fused)
    runs += 1
    bad += 0 if ok else 1
    seed += 1
print("fuzz_frames: %d scenes, %d with mismatches, %.0f s" % (runs, bad, time.time() - t0))
ctx.close()
sys.exit(1 if bad else 0)
