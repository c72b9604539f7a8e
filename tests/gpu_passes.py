"""HIP-side counterparts of tests/passes.py: same call shapes, device memory through niagara_amd.pipeline."""
import numpy as np
import torch

from niagara_amd import layouts as L
from niagara_amd import pipeline as P

from scenes import task_capacity


class GpuScene:
    """device copies of a scenes.make_scene() dict"""

    def __init__(self, ctx, scene, use_soa=True, fused=False):
        self.ctx, self.scene = ctx, scene
        # fused: NV_OPT_FUSED_COUNT_RESET + NV_OPT_FUSED_SUBMIT — the count words start dirty and no submit launch is issued
        self.fused = fused
        dev = ctx.device
        self.mb = P.to_device(scene["meshes"], dev)
        self.mlb = P.to_device(scene["meshlets"], dev)
        self.db = P.to_device(scene["draws"], dev)
        vw, vh = scene["viewport"]
        self.pyramid = P.DepthPyramid(dev, vw, vh)
        self.use_soa = use_soa
        if use_soa:
            ctx.upload_meshlets(self.mlb, len(scene["meshlets"]))
            ctx.upload_meshes(self.mb, len(scene["meshes"]))  # Mesh table staged in LDS; the other variant gathers it
            ctx.upload_draws(self.db, len(scene["draws"]), self.mb)    # decision inputs from the SoA mirror; the other variant reads the records
        else:
            # the registrations are by device pointer: a context shared between scenes must drop the previous scene's, or a
            # reused allocation would be taken for the buffer the mirror was built from
            ctx.upload_meshlets(None, 0)
            ctx.upload_meshes(None, 0)
            ctx.upload_draws(None, 0)

    def depthreduce(self, depth):
        d = torch.from_numpy(np.ascontiguousarray(depth)).to(self.ctx.device)
        self.ctx.depthreduce(d, depth.shape[1], depth.shape[0], self.pyramid.desc)
        return self.pyramid.data.cpu().numpy()

    def drawcull(self, cd, late, task, dvb_host, post_pass=0, with_pyramid=True):
        dev = self.ctx.device
        cap = task_capacity(self.scene) if task else len(self.scene["draws"]) + 1
        dt = L.TASKCMD if task else L.DRAWCMD
        dcb = torch.zeros(cap * dt.itemsize, dtype=torch.uint8, device=dev)
        dccb = torch.full((4,), 12345 if self.fused else 0, dtype=torch.int32, device=dev)
        dvb = torch.from_numpy(dvb_host.view(np.int32).copy()).to(dev)
        pd = cd.copy()
        pd["clusterBackfaceEnabled"] = 1 if post_pass == 0 else 0  # cull(): src/niagara.cpp:1549
        pd["postPass"] = post_pass
        self.ctx.drawcull(pd, late, task, self.db, self.mb, dcb, dccb, dvb, self.pyramid.desc if with_pyramid else None)
        return dcb, dccb, dvb

    def clustercull(self, cd, late, dcb, dccb, mvb, post_pass=0):
        dev = self.ctx.device
        ncmd = int(dccb[1].item()) * 64
        cib = torch.zeros(ncmd * 64 + 256, dtype=torch.int32, device=dev)
        ccb = torch.full((4,), 54321 if self.fused else 0, dtype=torch.int32, device=dev)
        pd = cd.copy()
        pd["postPass"] = post_pass
        self.ctx.clustercull(pd, late, dcb, dccb, self.db, self.mlb, mvb, self.pyramid.desc, cib, ccb)
        if not self.fused:
            self.ctx.clustersubmit(ccb, cib)
        return cib, ccb


def host_u32(t):
    return t.cpu().numpy().view(np.uint32)


def run_frames(ctx, scene, flags, frames=2, use_soa=True, fused=False):
    """same record structure as passes.run_frames, produced by the HIP passes"""
    from passes import set_flags
    g = GpuScene(ctx, scene, use_soa, fused)
    ctx.set_option(P.NV_OPT_FUSED_COUNT_RESET, int(fused))
    ctx.set_option(P.NV_OPT_FUSED_SUBMIT, int(fused))
    dev = ctx.device
    cd = set_flags(scene["cull"], flags)
    n = len(scene["draws"])
    dvb_host = np.zeros(n, np.uint32)
    mvb = torch.zeros((scene["slots"] + 31) // 32 + 2, dtype=torch.int32, device=dev)
    out = []
    for f in range(frames):
        rec = {}
        phases = [("early", 0, 0), ("late", 1, 0)] + ([("post", 1, 1)] if int(scene.get("post_mask", 0)) >> 1 else [])
        for phase, late, post in phases:
            if phase == "late":
                depth = scene["depth"] if f > 0 else np.zeros_like(scene["depth"])
                rec["pyramid"] = g.depthreduce(depth).copy()
            dcb, dccb, dvb = g.drawcull(cd, late, 1, dvb_host, post_pass=post)
            if not fused:
                ctx.tasksubmit(dccb, dcb)
            cib, ccb = g.clustercull(cd, late, dcb, dccb, mvb, post_pass=post)
            dvb_host = host_u32(dvb).copy()
            c4, cc4 = host_u32(dccb), host_u32(ccb)
            rec[phase] = dict(commands=P.from_device(dcb, L.TASKCMD)[:int(c4[1]) * 64].copy(), count4=c4.copy(),
                              cib=host_u32(cib)[:(int(cc4[0]) + 255) // 256 * 256].copy(), cc4=cc4.copy(), dvb=dvb_host.copy(),
                              mvb=host_u32(mvb).copy())
        out.append(rec)
    ctx.status()
    ctx.set_option(P.NV_OPT_FUSED_COUNT_RESET, 0)
    ctx.set_option(P.NV_OPT_FUSED_SUBMIT, 0)
    return out
