"""The lane-per-cluster form of the cluster passes (clustercull.hip cluster_bits_kernel<SOA, BITS, DEFER>): one LANE per cluster
that can be visible at all instead of one wave per task command — the early pass with visibility bits (candidates = set bits), and,
while the registered meshlet pool is small enough to stay in the caches (an instanced scene), the early pass without bits and the
late pass's first stage (DEFER: frustum / cone ballots, the survivors' commands listed for cluster_hiz_kernel).

Round 3 wrote the dense forms, saw one abort and one dead box in a four-file pytest process and took them back unmeasured.  These
tests pin the states that process could reach and a single-file run could not:
  * the direct form as the FIRST launch of a fresh context (no statistic of an earlier launch, no list, parity word 0);
  * sub-lists of the late pass's survivor-command list filled to exactly their capacity, one entry short of it, and past it
    (the occlusion stage's scan fallback);
  * every pass on poisoned device memory (tests/conftest.py): what a fresh process reads as zero beyond a buffer's end is garbage here.
"""
import numpy as np
import pytest
import torch

import oracle
from niagara_amd import host, synth
from niagara_amd import layouts as L
from niagara_amd import pipeline as P

import gpu_passes as G
from scenes import make_scene

pytestmark = pytest.mark.gpu


def _instanced_scene(draw_count, commands_per_draw, pool_commands, seed=3, ragged=False):
    """draws x commands over a SMALL meshlet pool: command k reads meshlets [64 (k mod pool_commands), +64) — an instanced scene, the
    pool stays in the caches; every command has its own visibility slots"""
    rng = np.random.default_rng(seed)
    draws = host.synth_draws(draw_count, 1, 300.0)
    n = draw_count * commands_per_draw
    meshlets = synth.make_meshlets(pool_commands * 64, seed)
    commands = synth.make_task_commands(draw_count, commands_per_draw)
    k = np.arange(n, dtype=np.uint32)
    commands["taskOffset"][:n] = (k % pool_commands) * 64
    if ragged:
        commands["taskCount"][:n:7] = rng.integers(0, 65, len(commands["taskCount"][:n:7]))
        tc = commands["taskCount"][:n].astype(np.int64)
        commands["meshletVisibilityOffset"][:n] = (np.concatenate([[0], np.cumsum(tc)[:-1]]) + 5).astype(np.uint32)  # words shared by neighbours
    commands["lateDrawVisibility"][:n] = rng.integers(0, 2, n)
    draws["meshletVisibilityOffset"] = np.arange(draw_count, dtype=np.uint32) * (commands_per_draw * 64)
    return draws, meshlets, commands, n


def _pyramid(ctx):
    pyr = oracle.Pyramid(256, 192)
    depth = make_scene(seed=3)["depth"]
    oracle.depthreduce(depth, pyr)
    gp = P.DepthPyramid(ctx.device, 256, 192)
    ctx.depthreduce(torch.from_numpy(depth).to(ctx.device), 256, 192, gp.desc)
    return pyr, gp


class _Pass:
    """device copies of one scene; run(cd, late, mvb0) compares one nv_clustercull pass with the oracle"""

    def __init__(self, ctx, draws, meshlets, commands, n, pyr=None, gp=None):
        self.ctx, self.draws, self.meshlets, self.commands, self.n, self.pyr, self.gp = ctx, draws, meshlets, commands, n, pyr, gp
        dev = ctx.device
        self.db, self.mlb, self.dcb = P.to_device(draws, dev), P.to_device(meshlets, dev), P.to_device(commands, dev)
        ctx.upload_meshlets(self.mlb, len(meshlets))
        self.c4 = synth.count4_for(n)
        self.dccb = torch.from_numpy(self.c4.view(np.int32).copy()).to(dev)
        self.cib = torch.zeros(len(commands) * 64 + 256, dtype=torch.int32, device=dev)
        self.ccb = torch.zeros(4, dtype=torch.int32, device=dev)

    def run(self, cd, late, mvb0, n=None):
        n = self.n if n is None else n
        c4 = synth.count4_for(n)
        self.dccb.copy_(torch.from_numpy(c4.view(np.int32).copy()))
        cib_o, cc4_o = np.zeros(len(self.commands) * 64 + 256, np.uint32), np.zeros(4, np.uint32)
        mvb_o = None if mvb0 is None else mvb0.copy()
        oracle.clustercull(cd, late, self.commands, c4, self.draws, self.meshlets, mvb_o, self.pyr if late else None, cib_o, cc4_o, threads=oracle.max_threads())
        d_mvb = None if mvb0 is None else torch.from_numpy(mvb0.view(np.int32).copy()).to(self.ctx.device)
        self.ccb.zero_()
        self.ctx.clustercull(cd, late, self.dcb, self.dccb, self.db, self.mlb, d_mvb, None if self.gp is None else self.gp.desc, self.cib, self.ccb)
        total = int(cc4_o[0])
        assert int(self.ccb[0].item()) == total
        assert (G.host_u32(self.cib)[:min(total, L.CLUSTER_LIMIT)] == cib_o[:min(total, L.CLUSTER_LIMIT)]).all()
        if mvb0 is not None:
            assert (G.host_u32(d_mvb) == mvb_o).all()
        self.ctx.status()
        return total


@pytest.mark.parametrize("first", ["early", "late", "early_bits", "taskcull"])
def test_direct_form_as_the_first_launch_of_a_fresh_context(first):
    """NV_OPT_CULL_FORM 2 on a context that has never launched anything: the lane form of the pass named by `first` runs before any
    other launch has left a statistic, a list or a flipped parity word; then the other passes in turn, twice (both banks)."""
    rng = np.random.default_rng(5)
    ctx = P.Context()
    try:
        ctx.set_option(P.NV_OPT_CULL_FORM, 2)
        draws, meshlets, commands, n = _instanced_scene(600, 7, 300, ragged=True)
        draws["position"] *= np.float32(0.05)
        pyr = oracle.Pyramid(256, 192)
        oracle.depthreduce(make_scene(seed=3)["depth"], pyr)
        cd = host.build_cull_data(cam_pos=(0, 0, 25), draw_count=len(draws), cullingEnabled=1, clusterBackfaceEnabled=1)
        cd["pyramidWidth"], cd["pyramidHeight"] = pyr.width, pyr.height
        bits = cd.copy()
        bits["clusterOcclusionEnabled"] = 1
        words = int(commands["meshletVisibilityOffset"][:n].max() + 64) // 32 + 4
        mvb0 = rng.integers(0, 2 ** 32, words, dtype=np.uint64).astype(np.uint32)
        p = _Pass(ctx, draws, meshlets, commands, n, pyr, None)

        def late_pass():
            # (the pyramid is built by a launch of its own: for `first == "late"` it must not come before the pass — the oracle's
            # pyramid is uploaded instead)
            if p.gp is None:
                gp = P.DepthPyramid(ctx.device, 256, 192)
                gp.data.copy_(torch.from_numpy(pyr.data.copy()))
                p.gp = gp
            return p.run(bits, 1, mvb0)

        def taskcull():
            payload_o, counts_o = np.zeros(len(commands) * 64, np.uint32), np.zeros(len(commands), np.uint32)
            oracle.taskcull(cd, 0, commands, synth.count4_for(n), draws, meshlets, None, None, payload_o, counts_o)
            dev = ctx.device
            pay = torch.zeros(len(commands) * 64, dtype=torch.int32, device=dev)
            cnt = torch.zeros(len(commands), dtype=torch.int32, device=dev)
            p.dccb.copy_(torch.from_numpy(synth.count4_for(n).view(np.int32).copy()))
            ctx.taskcull(cd, 0, p.dcb, p.dccb, p.db, p.mlb, None, None, pay, cnt)
            ctx.status()
            c = G.host_u32(cnt)[:n]
            assert (c == counts_o[:n]).all()
            got, want = G.host_u32(pay).reshape(-1, 64), payload_o.reshape(-1, 64)
            for i in np.nonzero(c)[0][:2000]:
                assert (got[i, :c[i]] == want[i, :c[i]]).all()
            return int(c.sum())

        passes = {"early": lambda: p.run(cd, 0, None), "late": late_pass, "early_bits": lambda: p.run(bits, 0, mvb0), "taskcull": taskcull}
        order = [first] + [k for k in passes if k != first]
        seen = [passes[k]() for k in order + order]
        assert max(seen) > 0.2 * n * 64
    finally:
        ctx.close()


@pytest.mark.parametrize("n_cmds", [140_000, 524_288, 524_288 + 4096])
def test_sub_lists_filled_to_exactly_their_capacity(n_cmds):
    """The late pass's first stage lists the commands that have frustum / cone survivors in 256 sub-lists of 2048 entries.  In the lane
    form a block lists up to 128 commands per iteration into two sub-lists shared with three other blocks each: with 1024 blocks and
    four iterations of 128 commands (524 288 commands) every sub-list holds EXACTLY 2048 entries; one iteration more and they overflow,
    and the occlusion stage scans all commands instead.  Every command has survivors (whole cloud inside the frustum, no cone test);
    IDs, count and visibility words against the oracle, direct form pinned, twice in a row (both banks), then the same through the
    automatic choice."""
    rng = np.random.default_rng(11)
    cpd = 8
    draws, meshlets, commands, n = _instanced_scene((n_cmds + cpd - 1) // cpd, cpd, 500)
    n = n_cmds
    draws["position"] *= np.float32(0.01)
    ctx = P.Context()
    try:
        pyr, gp = _pyramid(ctx)
        cd = host.build_cull_data(cam_pos=(0, 0, 12), draw_count=len(draws), cullingEnabled=1, clusterBackfaceEnabled=0, clusterOcclusionEnabled=1)
        cd["pyramidWidth"], cd["pyramidHeight"] = pyr.width, pyr.height
        mvb0 = rng.integers(0, 2 ** 32, len(commands) * 2 + 3, dtype=np.uint64).astype(np.uint32)
        p = _Pass(ctx, draws, meshlets, commands, n, pyr, gp)
        for form in (2, 2, 0, 0):
            ctx.set_option(P.NV_OPT_CULL_FORM, form)
            total = p.run(cd, 1, mvb0)
            assert 0.02 * n * 64 < total < 0.98 * n * 64
        sparse = cd.copy()
        sparse["view"][0][14] += np.float32(500.0)
        assert p.run(sparse, 1, mvb0) == 0     # nothing listed: the flag and the counts of the dense passes must be gone
        p.run(cd, 1, mvb0)
    finally:
        ctx.close()


def test_profile_variants_name_the_form_each_launch_took():
    """nv_profile_variants: a profile can say which kernel variant it timed.  Pinned forms on a sparse and a dense view of
    an instanced pool: the counts name the filter form with the ring depth the command count selects, the direct form, the two lane forms,
    the occlusion stage and the AoS path; reading resets them."""
    ctx = P.Context()
    try:
        draws, meshlets, commands, n = _instanced_scene(600, 7, 300)
        draws["position"] *= np.float32(0.05)
        pyr, gp = _pyramid(ctx)
        cd = host.build_cull_data(cam_pos=(0, 0, 25), draw_count=len(draws), cullingEnabled=1, clusterBackfaceEnabled=1)
        cd["pyramidWidth"], cd["pyramidHeight"] = pyr.width, pyr.height
        bits = cd.copy()
        bits["clusterOcclusionEnabled"] = 1
        mvb0 = np.random.default_rng(1).integers(0, 2 ** 32, len(commands) * 2 + 3, dtype=np.uint64).astype(np.uint32)
        p = _Pass(ctx, draws, meshlets, commands, n, pyr, gp)
        assert ctx.profile_variants() == {}
        ctx.set_option(P.NV_OPT_CULL_FORM, 1)
        ctx.set_option(P.NV_OPT_CULL_RING, 4)
        p.run(cd, 0, None)
        ctx.set_option(P.NV_OPT_CULL_RING, 8)
        p.run(cd, 0, None)
        assert ctx.profile_variants() == {"cull_filter_ring4": 1, "cull_filter_ring8": 1}
        ctx.set_option(P.NV_OPT_CULL_FORM, 2)
        p.run(cd, 0, None)       # no visibility bits: the direct form's packed walk (windows of 64 valid meshlets)
        p.run(bits, 0, mvb0)     # with bits: one lane per set bit
        p.run(bits, 1, mvb0)     # late with HiZ: the packed walk as the first stage + the occlusion stage
        assert ctx.profile_variants() == {"cull_direct_packed": 2, "cull_lanes_bits": 1, "hiz_stage": 1}
        ctx.set_option(P.NV_OPT_CULL_FORM, 3)
        p.run(bits, 0, mvb0)     # one wave per command also with visibility bits
        assert ctx.profile_variants() == {"cull_direct": 1}
        ctx.set_option(P.NV_OPT_CULL_FORM, 4)
        p.run(cd, 0, None)       # one command per wave iteration also where the packed walk applies
        p.run(bits, 1, mvb0)
        assert ctx.profile_variants() == {"cull_direct": 2, "hiz_stage": 1}
        ctx.upload_meshlets(None, 0)  # no mirror: the records are read in place
        p.run(cd, 0, None)
        assert ctx.profile_variants() == {"cull_aos": 1} and ctx.profile_variants() == {}
    finally:
        ctx.close()


def test_a_fresh_contexts_first_frame_takes_the_direct_forms_for_its_own_drawculls_commands():
    """No launch has left a filter statistic yet: a cluster pass over the command buffer THIS context's nv_drawcull(task) wrote runs on the commands of
    draws the draw-level cull already found visible, and takes the direct / lane forms instead of the filter form (context.hip taskCommandsFrom; VERDICT r4
    item 3c: a renderer's first frames hitched through the filter form).  A pre-built command list on a fresh context keeps the filter form.  Results
    are the oracle's either way."""
    import passes
    from gpu_passes import run_frames
    scene = make_scene(seed=31, n_draws=1200, meshlets_lod0=130)
    flags = (1, 1, 1, 1, 1)
    fo = passes.run_frames(oracle, scene, flags, frames=2)
    ctx = P.Context()
    try:
        fg = run_frames(ctx, scene, flags, frames=1)
        v = ctx.profile_variants()
        # frame 0: the early cluster pass is the context's first cluster launch ever
        assert v.get("cull_direct", 0) + v.get("cull_lanes_bits", 0) + v.get("cull_direct_packed", 0) >= 1, v
        for phase in ("early", "late"):
            for key in ("count4", "commands", "cc4", "cib", "dvb", "mvb"):
                assert fo[0][phase][key].tobytes() == fg[0][phase][key].tobytes(), (phase, key)
    finally:
        ctx.close()
    # a command list that did not come from this context's drawcull: no statistic, no provenance — the filter form
    ctx = P.Context()
    try:
        draws, meshlets, commands, n = _instanced_scene(300, 5, 200)
        cd = host.build_cull_data(cam_pos=(0, 0, 25), draw_count=len(draws), cullingEnabled=1, clusterBackfaceEnabled=1)
        _Pass(ctx, draws, meshlets, commands, n).run(cd, 0, None)
        v = ctx.profile_variants()
        assert v.get("cull_filter_ring4", 0) + v.get("cull_filter_ring8", 0) == 1 and "cull_direct" not in v and "cull_direct_packed" not in v, v
    finally:
        ctx.close()


@pytest.mark.parametrize("shape", ["ragged", "sixteens", "ones", "empties", "full", "over64", "single"])
def test_packed_walk_command_shapes(shape):
    """The direct form's packed walk (cluster_mask_kernel PACK, round 6) lays the VALID meshlets of a wave's 64 commands end to end and tests them in windows
    of 64: command sizes decide where the windows' borders fall.  Shapes that put them everywhere: any taskCount in 0 .. 64 (empty commands inside a segment,
    in front of it, behind it), sizes that divide 64 (a border on every fourth command), one meshlet per command (64 commands = one window), nothing but empty
    commands, full commands only (every window = one command), taskCount above 64 (the reference tests lanes 0 .. 63: clustercull.comp.glsl:66-70) and a pass
    of ONE command.  Early pass without visibility bits (the walk's ballots and statistic) and the late pass with HiZ (the walk as the first stage: DEFER),
    each against the oracle, pinned direct (NV_OPT_CULL_FORM 2) and — the same lists — one command per wave iteration (4)."""
    rng = np.random.default_rng(11)
    ctx = P.Context()
    try:
        draws, meshlets, commands, n = _instanced_scene(700, 9, 300)
        draws["position"] *= np.float32(0.05)
        tc = commands["taskCount"][:n]
        if shape == "ragged":
            tc[:] = rng.integers(0, 65, n)
        elif shape == "sixteens":
            tc[:] = rng.choice([16, 32, 48, 64], n)
        elif shape == "ones":
            tc[:] = 1
        elif shape == "empties":
            tc[:] = 0
            tc[rng.integers(0, n, 5)] = rng.integers(1, 65, 5)
        elif shape == "over64":
            tc[::3] = rng.integers(65, 1000, len(tc[::3]))
        elif shape == "single":
            n = 1
        t64 = np.minimum(commands["taskCount"][:len(commands)].astype(np.int64), 64)
        commands["meshletVisibilityOffset"][:] = (np.concatenate([[0], np.cumsum(t64)[:-1]]) + 3).astype(np.uint32)  # words shared by neighbours
        pyr, gp = _pyramid(ctx)
        cd = host.build_cull_data(cam_pos=(0, 0, 25), draw_count=len(draws), cullingEnabled=1, clusterBackfaceEnabled=1)
        cd["pyramidWidth"], cd["pyramidHeight"] = pyr.width, pyr.height
        late_cd = cd.copy()
        late_cd["clusterOcclusionEnabled"] = 1
        mvb0 = rng.integers(0, 2 ** 32, int(t64.sum()) // 32 + 8, dtype=np.uint64).astype(np.uint32)
        p = _Pass(ctx, draws, meshlets, commands, n, pyr, gp)
        for form, variant in ((2, "cull_direct_packed"), (4, "cull_direct")):
            ctx.set_option(P.NV_OPT_CULL_FORM, form)
            ctx.profile_variants()
            for rep in range(2):  # (both banks of the tile counters)
                p.run(cd, 0, None)
                p.run(late_cd, 1, mvb0)
            v = ctx.profile_variants()
            assert v.get(variant, 0) == 4 and v.get("hiz_stage", 0) == 2, (form, v)
        nocone = cd.copy()
        nocone["clusterBackfaceEnabled"] = 0
        ctx.set_option(P.NV_OPT_CULL_FORM, 2)
        p.run(nocone, 0, None)
    finally:
        ctx.close()
