"""Runs a fixed set of clustercull passes with whatever library NV_LIBRARY_PATH names and prints one sha256 per pass
(tests/test_plain_loads.py compares the product build with the plain-loads build).  Test infrastructure."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from niagara_amd import host, synth  # noqa: E402
from niagara_amd import layouts as L  # noqa: E402
from niagara_amd import pipeline as P  # noqa: E402


def main():
    ctx = P.Context(0)
    dev = ctx.device
    rng = np.random.default_rng(5)
    for n_draws, cpd, radius in ((900, 7, 60.0), (15625, 10, 300.0), (4000, 10, 40.0)):
        draws, meshlets, commands, n = synth.cluster_scene(n_draws, cpd, seed=4, scene_radius=radius)
        commands["taskCount"][:n:7] = rng.integers(0, 65, len(commands["taskCount"][:n:7]))  # ragged commands too
        commands["lateDrawVisibility"][:n] = rng.integers(0, 2, n)
        size = 512
        depth = torch.from_numpy(synth.make_depth(size, size)).to(dev)
        pyr = P.DepthPyramid(dev, size, size)
        ctx.depthreduce(depth, size, size, pyr.desc)
        db, mlb, dcb = P.to_device(draws, dev), P.to_device(meshlets, dev), P.to_device(commands, dev)
        ctx.upload_meshlets(mlb, len(meshlets))
        dccb = torch.from_numpy(synth.count4_for(n).view(np.int32).copy()).to(dev)
        mvb0 = torch.from_numpy(rng.integers(0, 2 ** 32, n * 2 + 4, dtype=np.uint64).astype(np.uint32).view(np.int32)).to(dev)
        cib = torch.zeros(n * 64 + 256, dtype=torch.int32, device=dev)
        ccb = torch.zeros(4, dtype=torch.int32, device=dev)
        for late in (0, 1):
            for coe in (0, 1):
                for post in (0, 1):
                    cd = host.build_cull_data(draw_count=n_draws, viewport=(size, size), pyramid=(pyr.width, pyr.height), cullingEnabled=1,
                                              clusterBackfaceEnabled=1, clusterOcclusionEnabled=coe, postPass=post)
                    mvb = mvb0.clone()
                    ccb.zero_()
                    ctx.clustercull(cd, late, dcb, dccb, db, mlb, mvb, pyr.desc, cib, ccb)
                    ctx.status()
                    total = int(ccb[0].item())
                    h = hashlib.sha256()
                    h.update(cib[:total].cpu().numpy().tobytes())
                    h.update(mvb.cpu().numpy().tobytes())
                    print("%d %d %d %d %d %d %s" % (n_draws, cpd, late, coe, post, total, h.hexdigest()), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
