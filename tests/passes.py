"""Drives one implementation of the passes (oracle, oracle.ref — same call signatures) over a scene, in the
reference's frame order (src/niagara.cpp:1765-1788).  Test infrastructure."""
import numpy as np

import oracle
from niagara_amd import layouts as L

from scenes import task_capacity


def set_flags(cd, flags):
    cd = cd.copy()
    for k, v in zip(("cullingEnabled", "lodEnabled", "occlusionEnabled", "clusterOcclusionEnabled", "clusterBackfaceEnabled"), flags):
        cd[k] = v
    return cd


def run_drawcull(impl, scene, cd, late, task, dvb, pyr, post_pass=0):
    cap = task_capacity(scene) if task else len(scene["draws"]) + 1
    commands = np.zeros(cap, dtype=L.TASKCMD if task else L.DRAWCMD)
    count4 = np.zeros(4, np.uint32)
    pd = cd.copy()
    pd["clusterBackfaceEnabled"] = 1 if post_pass == 0 else 0  # cull(): src/niagara.cpp:1549 (drawcull does not read it)
    pd["postPass"] = post_pass
    impl.drawcull(pd, late, task, scene["draws"], scene["meshes"], commands, count4, dvb, pyr)
    return commands, count4


def run_cluster(impl, scene, cd, late, commands, count4, mvb, pyr, post_pass=0):
    ncmd = int(count4[1]) * 64
    cib = np.zeros(ncmd * 64 + 256, np.uint32)
    cc4 = np.zeros(4, np.uint32)
    pd = cd.copy()
    pd["postPass"] = post_pass
    impl.clustercull(pd, late, commands, count4, scene["draws"], scene["meshlets"], mvb, pyr, cib, cc4)
    impl.clustersubmit(cc4, cib)
    return cib, cc4


def run_frames(impl, scene, flags, frames=2, backface_on_cluster=True):
    """full frame protocol with the task/cluster path (src/niagara.cpp:1765-1788): early, pyramid, late and — when a draw of the
    scene has a postPass bit >= 1 (`meshPostPasses >> 1`, :1781) — the post phase cull(late, postPass=1) + render(late,
    postPass=1); returns every intermediate for comparison"""
    cd = set_flags(scene["cull"], flags)
    n = len(scene["draws"])
    dvb = np.zeros(n, np.uint32)
    mvb = np.zeros((scene["slots"] + 31) // 32 + 2, np.uint32)
    vw, vh = scene["viewport"]
    pyr = oracle.Pyramid(vw, vh)
    out = []
    for f in range(frames):
        rec = {}
        phases = [("early", 0, 0), ("late", 1, 0)] + ([("post", 1, 1)] if int(scene.get("post_mask", 0)) >> 1 else [])
        for phase, late, post in phases:
            if phase == "late":
                # pyramid from a depth buffer (frame 0: cleared depth = everything passes, like the reference's first frame)
                depth = scene["depth"] if f > 0 else np.zeros_like(scene["depth"])
                impl.depthreduce(depth, pyr)
                rec["pyramid"] = pyr.data.copy()
            cmds, c4 = run_drawcull(impl, scene, cd, late, 1, dvb, pyr, post_pass=post)
            impl.tasksubmit(c4, cmds)
            cib, cc4 = run_cluster(impl, scene, cd, late, cmds, c4, mvb, pyr, post_pass=post)
            rec[phase] = dict(commands=cmds[:int(c4[1]) * 64].copy(), count4=c4.copy(), cib=cib[:(int(cc4[0]) + 255) // 256 * 256].copy(),
                              cc4=cc4.copy(), dvb=dvb.copy(), mvb=mvb.copy())
        out.append(rec)
    return out
