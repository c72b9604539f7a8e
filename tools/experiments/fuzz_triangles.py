#!/usr/bin/env python3
"""Randomised soak of nv_trianglecull alone (the packed form of round 4): random cluster scenes, cameras (inside the cloud, behind it),
fp16 specials in the vertex stream, malformed meshlet counts (no vertices, no triangles, more triangles than MESH_MAXTRI, index bytes above
the vertex count), ~0 holes in the list, lists long enough for several chunks per wave, random mask capacities.  Every mask and the totals
compared with the oracle.  Needs a GPU.

    python tools/experiments/fuzz_triangles.py [seconds=120] [first_seed=9000]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402  (checker)
from niagara_amd import layouts as L  # noqa: E402
from niagara_amd import pipeline as P  # noqa: E402
from scenes import make_triangle_scene  # noqa: E402
from test_trianglecull import _every_meshlet_list, cluster_list, run as run_triangles  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
ctx = P.Context(0)
dev = ctx.device
t0 = time.time()
runs = slots_total = 0
bad = []
t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    radius = float(rng.choice([3.0, 6.0, 20.0, 60.0]))
    cam = dict(cam_pos=tuple(float(x) for x in rng.uniform(-8, 8, 3))) if rng.random() < 0.5 else {}
    s = make_triangle_scene(seed=seed, n_draws=int(rng.integers(1, 400)), commands_per_draw=int(rng.integers(1, 7)), scene_radius=radius,
                            viewport=(int(rng.integers(2, 2000)), int(rng.integers(2, 1200))), specials=bool(rng.random() < 0.3), **cam)
    ml = s["meshlets"]
    n = len(ml)
    if rng.random() < 0.5:  # malformed counts (tests/test_trianglecull.py: test_hip_malformed_meshlets_read_as_the_oracle_defines)
        pick = rng.random(n)
        safe = np.arange(n) < n - 64
        ml["triangleCount"][pick < 0.05] = 0
        ml["vertexCount"][(pick >= 0.05) & (pick < 0.10)] //= 3
        ml["triangleCount"][(pick >= 0.10) & (pick < 0.15) & safe] = rng.integers(97, 256)
        ml["vertexCount"][(pick >= 0.17) & (pick < 0.20)] = 0
    if rng.random() < 0.5:
        cib, cc4 = cluster_list(oracle, s, backface=int(rng.integers(0, 2)))
    else:
        cib, cc4 = _every_meshlet_list(s, repeat=int(rng.choice([1, 1, 2, 7])), holes=float(rng.choice([0.0, 0.02, 0.5])), seed=seed)
    mo, to = run_triangles(oracle.trianglecull, s, cib, cc4)
    cap = len(mo) if rng.random() < 0.7 else int(rng.integers(0, len(mo) + 1))
    masks = torch.full((len(mo) * 16,), 0x5a, dtype=torch.uint8, device=dev)
    totals = torch.zeros(3, dtype=torch.int64, device=dev)
    ctx.trianglecull(s["globals"], t(s["commands"]), t(s["draws"]), t(ml), t(s["data"]), t(s["vertices"]), t(cib), t(cc4), masks, cap, totals)
    got = masks.cpu().numpy()
    ok = got[:cap * 16].tobytes() == mo[:cap].tobytes() and (got[cap * 16:] == 0x5a).all() and totals.cpu().numpy().astype(np.uint64).tolist() == to.tolist()
    if not ok:
        bad.append(seed)
    runs += 1
    slots_total += len(mo)
    seed += 1
print("fuzz_triangles: %d scenes, %d slots, %.0f s, differences: %s" % (runs, slots_total, time.time() - t0, bad or "none"))
sys.exit(1 if bad else 0)
