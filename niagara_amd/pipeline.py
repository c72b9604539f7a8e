"""Host-side mirror of niagara's cull/render/pyramid lambdas (src/niagara.cpp:1530-1611,1703-1733,1765-1788)
driving the HIP passes through the C ABI.  torch is plumbing only: device memory, streams, torch.distributed.

Buffer names follow the reference (src/niagara.cpp:1027-1093):
    mb meshes · mlb meshlets · db draws · dvb drawVisibility · dcb draw/task commands · dccb command count + indirect
    args · mvb meshletVisibility · cib clusterIndices · ccb cluster count + indirect args
"""
import ctypes as C

import numpy as np
import torch

from . import host
from . import layouts as L
from ._lib import NvError, PyramidDesc, check, lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_device(arr, device):
    """numpy (structured) array -> flat uint8 device tensor with the same bytes"""
    a = np.ascontiguousarray(arr)
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(device)


def from_device(t, dtype, count=None):
    a = t.detach().cpu().numpy().view(np.uint8).reshape(-1)
    out = a.view(dtype)
    return out[:count] if count is not None else out


NV_OPT_FUSED_COUNT_RESET = 1
NV_OPT_FUSED_SUBMIT = 2
NV_OPT_CULL_WORKGROUPS_PER_CU = 3
NV_OPT_SCATTER_WAVES = 4
NV_OPT_CULL_FORM = 5
NV_OPT_CULL_RING = 6
NV_OPT_TASK_EMIT = 7
NV_OPT_DRAW_RECORDS = 8


class Context:
    """one nv_context per device; not re-entrant (one stream at a time)"""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise NvError("niagara_amd needs a HIP device: torch.cuda.is_available() is False and there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        h = C.c_void_p()
        check(lib.nv_create(C.byref(h), self.device.index), "nv_create")
        self.h = h

    def close(self):
        if self.h:
            lib.nv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def status(self):
        check(lib.nv_status(self.h, _stream()), "nv_status")

    def set_option(self, option, value):
        check(lib.nv_set_option(self.h, int(option), int(value)), "nv_set_option")

    def reserve(self, max_draws, max_commands=0):
        """scratch for passes over up to max_draws draws (nv_create reserves 1 M): pass entry points never allocate"""
        check(lib.nv_reserve(self.h, int(max_draws), int(max_commands)), "nv_reserve")

    def share_scene(self, other):
        """use `other`'s scene mirrors (meshlet SoA, draw mirror, mesh table) instead of building this context's own"""
        check(lib.nv_share_scene(self.h, other.h), "nv_share_scene")

    def profile(self, enabled):
        check(lib.nv_profile_enable(self.h, int(enabled)), "nv_profile_enable")

    def profile_read(self):
        """{slot: (total_ms, launches)} for cluster_cull, cluster_scatter, drawcull, depthreduce, cluster_hiz"""
        ms, cnt = (C.c_float * 5)(), (C.c_uint32 * 5)()
        check(lib.nv_profile_read(self.h, C.byref(ms), C.byref(cnt)), "nv_profile_read")
        names = ("cluster_cull", "cluster_scatter", "drawcull", "depthreduce", "cluster_hiz")
        return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(names)}

    VARIANTS = ("cull_filter_ring4", "cull_filter_ring8", "cull_direct", "cull_lanes_bits", "cull_lanes", "cull_aos", "hiz_stage", "task_list", "task_per_draw", "cull_direct_packed")

    def profile_variants(self):
        """{variant: launches since the last call}: which kernel form the host's per-launch choices resolved to (non-zero entries only)"""
        cnt = (C.c_uint32 * len(self.VARIANTS))()
        check(lib.nv_profile_variants(self.h, C.byref(cnt)), "nv_profile_variants")
        return {n: int(cnt[i]) for i, n in enumerate(self.VARIANTS) if cnt[i]}

    # ---- passes (argument order = descriptor order of the reference dispatches)
    def upload_meshlets(self, mlb, count):
        check(lib.nv_upload_meshlets(self.h, _stream(), _ptr(mlb), count), "nv_upload_meshlets")

    def upload_meshes(self, mb, count):
        check(lib.nv_upload_meshes(self.h, _stream(), _ptr(mb), count), "nv_upload_meshes")

    def meshlet_bounds(self, vertices, meshlet_data, mlb, count, bounds8=None):
        """src/scene.cpp:69-85 on the GPU: fills center / radius / cone of the Meshlet records in `mlb` (parity unpinned: header)"""
        check(lib.nv_meshlet_bounds(self.h, _stream(), _ptr(vertices), _ptr(meshlet_data), _ptr(mlb), count, _ptr(bounds8)), "nv_meshlet_bounds")

    def upload_draws(self, db, count, mb=None):
        """mirror of what a draw decision reads: world-space spheres (mesh bounds of table `mb` folded in), scale, meshIndex,
        postPass (None, 0 drops the registration)"""
        check(lib.nv_upload_draws(self.h, _stream(), _ptr(db), count, _ptr(mb)), "nv_upload_draws")

    def update_draws(self, db, first, count):
        """re-transposes draws [first, first + count) after the caller rewrote them (animation, src/niagara.cpp:1385-1391)"""
        check(lib.nv_update_draws(self.h, _stream(), _ptr(db), first, count), "nv_update_draws")

    def drawcull(self, cull, late, task, db, mb, dcb, dccb, dvb, pyramid=None):
        check(lib.nv_drawcull(self.h, _stream(), C.c_void_p(cull.ctypes.data), int(late), int(task), _ptr(db), _ptr(mb), _ptr(dcb),
                              _ptr(dccb), _ptr(dvb), None if pyramid is None else C.byref(pyramid)), "nv_drawcull")

    def reset_count(self, a, b=None):
        check(lib.nv_reset_count(self.h, _stream(), _ptr(a), _ptr(b)), "nv_reset_count")

    def tasksubmit(self, dccb, dcb):
        check(lib.nv_tasksubmit(self.h, _stream(), _ptr(dccb), _ptr(dcb)), "nv_tasksubmit")

    def clustercull(self, cull, late, dcb, dccb, db, mlb, mvb, pyramid, cib, ccb):
        check(lib.nv_clustercull(self.h, _stream(), C.c_void_p(cull.ctypes.data), int(late), _ptr(dcb), _ptr(dccb), _ptr(db), _ptr(mlb),
                                 _ptr(mvb), None if pyramid is None else C.byref(pyramid), _ptr(cib), _ptr(ccb)), "nv_clustercull")

    def bind_clustercull(self, stream, cull, late, dcb, dccb, db, mlb, mvb, pyramid, cib, ccb):
        """nv_clustercull with every argument marshalled once: returns a zero-argument callable that launches the pass on
        `stream` (a torch.cuda.Stream, or None for the current one).  For callers that issue many passes back to back: the
        per-call marshalling (stream lookup, data_ptr() of eight tensors) costs more host time than a 25 us pass leaves."""
        st = C.c_void_p((torch.cuda.current_stream() if stream is None else stream).cuda_stream)
        keep = (stream, cull, dcb, dccb, db, mlb, mvb, pyramid, cib, ccb)  # the closure keeps the buffers alive
        args = (self.h, st, C.c_void_p(cull.ctypes.data), int(late), _ptr(dcb), _ptr(dccb), _ptr(db), _ptr(mlb), _ptr(mvb),
                None if pyramid is None else C.byref(pyramid), _ptr(cib), _ptr(ccb))
        fn = lib.nv_clustercull

        def launch(_keep=keep):
            rc = fn(*args)
            if rc:
                check(rc, "nv_clustercull")
        return launch

    def clustersubmit(self, ccb, cib):
        check(lib.nv_clustersubmit(self.h, _stream(), _ptr(ccb), _ptr(cib)), "nv_clustersubmit")

    def taskcull(self, cull, late, dcb, dccb, db, mlb, mvb, pyramid, payloads, payload_counts):
        check(lib.nv_taskcull(self.h, _stream(), C.c_void_p(cull.ctypes.data), int(late), _ptr(dcb), _ptr(dccb), _ptr(db), _ptr(mlb),
                              _ptr(mvb), None if pyramid is None else C.byref(pyramid), _ptr(payloads), _ptr(payload_counts)), "nv_taskcull")

    def cluster_expand(self, dcb, mlb, cib, ccb, records, capacity, totals3):
        check(lib.nv_cluster_expand(self.h, _stream(), _ptr(dcb), _ptr(mlb), _ptr(cib), _ptr(ccb), _ptr(records), capacity, _ptr(totals3)),
              "nv_cluster_expand")

    def trianglecull(self, globals_, dcb, db, mlb, meshlet_data, vertices, cib, ccb, masks, capacity, totals3):
        """meshlet.mesh.glsl:91-198 with MESH_CULL = 1: one NvTriangleMask per slot of the consumer's grid"""
        check(lib.nv_trianglecull(self.h, _stream(), C.c_void_p(globals_.ctypes.data), _ptr(dcb), _ptr(db), _ptr(mlb), _ptr(meshlet_data), _ptr(vertices),
                                  _ptr(cib), _ptr(ccb), _ptr(masks), capacity, _ptr(totals3)), "nv_trianglecull")

    def depthreduce(self, depth, width, height, pyramid):
        check(lib.nv_depthreduce(self.h, _stream(), _ptr(depth), width, height, C.byref(pyramid)), "nv_depthreduce")

    def set_counts_sink(self, out3):
        """the next clustercull calls also write {0, dccb[0], cluster count} (3 x u64) to out3; None turns it off"""
        check(lib.nv_set_counts_sink(self.h, _ptr(out3)), "nv_set_counts_sink")

    def pack_counts(self, a, b, c, out3):
        check(lib.nv_pack_counts(self.h, _stream(), _ptr(a), _ptr(b), _ptr(c), _ptr(out3)), "nv_pack_counts")

    def probe_cluster_scalars(self, cull, dcb, command_count, db, mlb, pyramid=None):
        out = torch.zeros((command_count, 64, 16), dtype=torch.float32, device=self.device)
        check(lib.nv_probe_cluster_scalars(self.h, _stream(), C.c_void_p(cull.ctypes.data), _ptr(dcb), command_count, _ptr(db), _ptr(mlb),
                                           None if pyramid is None else C.byref(pyramid), _ptr(out)), "nv_probe_cluster_scalars")
        return out


class DepthPyramid:
    """replaces the R32F mip-chain image + MIN sampler (src/niagara.cpp:629,1339-1350)"""

    def __init__(self, device, depth_w, depth_h):
        self.desc = host.pyramid_desc(depth_w, depth_h)
        self.data = torch.zeros(self.desc.totalTexels, dtype=torch.float32, device=device)
        self.desc.d_base = self.data.data_ptr()
        self.width, self.height, self.levels = self.desc.width, self.desc.height, self.desc.levels
        self.mip_offset = [int(x) for x in self.desc.mipOffset]

    def level(self, i):
        w, h = max(1, self.width >> i), max(1, self.height >> i)
        return self.data[self.mip_offset[i]:self.mip_offset[i] + w * h].view(h, w)


class VisibilityPipeline:
    """niagara's GPU-driven visibility front-end for one scene on one device."""

    def __init__(self, meshes, meshlets, draws, depth_size, ctx=None, task_capacity=None, cluster_capacity=None, use_soa=True, fused=False):
        self.ctx = ctx or Context()
        # fused=True: the passes absorb the count-word resets and the tasksubmit / clustersubmit fix-ups (same buffer
        # contents, four launches less per phase); fused=False issues the reference's dispatch sequence one to one
        self.fused = bool(fused)
        self.ctx.set_option(NV_OPT_FUSED_COUNT_RESET, int(self.fused))
        self.ctx.set_option(NV_OPT_FUSED_SUBMIT, int(self.fused))
        dev = self.ctx.device
        self.mesh_count, self.meshlet_count, self.draw_count = len(meshes), len(meshlets), len(draws)
        self.draws_host = draws.copy()
        self.slots, self.post_mask = host.assign_visibility_offsets(self.draws_host, meshes)  # src/niagara.cpp:1002-1020
        self.mb = to_device(meshes, dev)
        self.mlb = to_device(meshlets, dev)
        self.db = to_device(self.draws_host, dev)
        self.dvb = torch.zeros(max(1, self.draw_count), dtype=torch.int32, device=dev)         # zeroed once (:1450-1457)
        self.mvb = torch.zeros(max(1, (self.slots + 31) // 32 + 2), dtype=torch.int32, device=dev)  # (:1459-1468)
        tcap = task_capacity or L.TASK_WGLIMIT
        ccap = cluster_capacity or L.CLUSTER_LIMIT
        # The kernels clamp at the reference's limits (TASK_WGLIMIT commands, CLUSTER_LIMIT indices: the sizes niagara
        # allocates, src/niagara.cpp:1070,1088), not at the size of a smaller buffer.  Smaller buffers are accepted only when
        # no pass over this scene can fill them: every draw emitting its largest LOD (+ the submit kernels' padding).
        lod_counts = meshes["lods"]["meshletCount"].astype(np.int64)
        lod_valid = np.arange(lod_counts.shape[1])[None, :] < meshes["lodCount"][:, None]
        per_mesh = np.where(lod_valid, lod_counts, 0).max(axis=1) if len(meshes) else np.zeros(0, np.int64)
        mi = np.minimum(draws["meshIndex"].astype(np.int64), max(0, len(meshes) - 1))
        worst_meshlets = int(per_mesh[mi].sum()) if len(draws) and len(meshes) else 0
        worst_tasks = int(((per_mesh[mi] + 63) // 64).sum()) if len(draws) and len(meshes) else 0
        if tcap < min(worst_tasks, L.TASK_WGLIMIT) or ccap < min(worst_meshlets, L.CLUSTER_LIMIT):
            raise NvError("task_capacity %d / cluster_capacity %d cannot hold what this scene can emit (%d task commands, %d meshlets): the passes "
                          "drop output only at the reference's limits, never at a smaller buffer's end" % (tcap, ccap, worst_tasks, worst_meshlets))
        self.dcb = torch.zeros(tcap * L.TASKCMD.itemsize + 64 * L.TASKCMD.itemsize, dtype=torch.uint8, device=dev)
        self.dccb = torch.zeros(4, dtype=torch.int32, device=dev)
        self.cib = torch.zeros(ccap + 256, dtype=torch.int32, device=dev)
        self.ccb = torch.zeros(4, dtype=torch.int32, device=dev)
        self.depth_w, self.depth_h = depth_size
        self.pyramid = DepthPyramid(dev, *depth_size)
        self.ctx.reserve(self.draw_count, tcap)
        self.ctx.upload_meshes(self.mb, self.mesh_count)
        if use_soa and self.meshlet_count:
            self.ctx.upload_meshlets(self.mlb, self.meshlet_count)
        if use_soa and self.draw_count:
            self.ctx.upload_draws(self.db, self.draw_count, self.mb)

    # src/niagara.cpp:1530-1574
    def cull(self, cull_data, late, task=True, post_pass=0):
        if not self.fused:
            self.ctx.reset_count(self.dccb)                         # vkCmdFillBuffer(dccb, 0, 4, 0)  (:1541)
        pass_data = cull_data.copy()
        pass_data["clusterBackfaceEnabled"] = 1 if post_pass == 0 else 0   # (:1549)
        pass_data["postPass"] = post_pass
        self.ctx.drawcull(pass_data, late, task, self.db, self.mb, self.dcb, self.dccb, self.dvb, self.pyramid.desc)
        if task and not self.fused:
            self.ctx.tasksubmit(self.dccb, self.dcb)                # (:1563-1568)

    # src/niagara.cpp:1582-1611 (cluster branch of render())
    def render_clusters(self, cull_data, late, post_pass=0):
        if not self.fused:
            self.ctx.reset_count(self.ccb)                          # vkCmdFillBuffer(ccb, 0, 4, 0)  (:1586)
        pass_data = cull_data.copy()
        pass_data["postPass"] = post_pass                           # (:1595-1596)
        self.ctx.clustercull(pass_data, late, self.dcb, self.dccb, self.db, self.mlb, self.mvb, self.pyramid.desc, self.cib, self.ccb)
        if not self.fused:
            self.ctx.clustersubmit(self.ccb, self.cib)

    # src/niagara.cpp:1703-1733
    def build_pyramid(self, depth):
        self.ctx.depthreduce(depth, self.depth_w, self.depth_h, self.pyramid.desc)

    def visible_clusters(self):
        n = int(self.ccb[0].item())
        return self.cib[:min(n, L.CLUSTER_LIMIT)].cpu().numpy().view(np.uint32), n
