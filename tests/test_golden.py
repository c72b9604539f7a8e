"""Golden fixtures produced by the reference's own shaders (tests/golden/generate.py, via oracle/_ref): the CPU oracle
must reproduce them bit for bit everywhere (CPU suite), and so must the HIP passes through the C ABI (GPU suite)."""
import glob
import os

import numpy as np
import pytest

import oracle

import passes

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def load(path):
    z = np.load(path)
    scene = dict(meshes=z["meshes"], meshlets=z["meshlets"], draws=z["draws"], cull=z["cull"], depth=z["depth"], slots=int(z["slots"][0]),
                 viewport=tuple(int(x) for x in z["viewport"]))
    return z, scene, tuple(int(x) for x in z["flags"])


def compare(z, frames):
    for f, rec in enumerate(frames):
        assert rec["pyramid"].tobytes() == z["f%d_pyramid" % f].tobytes(), (f, "pyramid")
        for phase in ("early", "late"):
            for key in ("count4", "commands", "cc4", "cib", "dvb", "mvb"):
                assert rec[phase][key].tobytes() == z["f%d_%s_%s" % (f, phase, key)].tobytes(), (f, phase, key)


def test_fixtures_exist():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_reference_fixtures(path):
    z, scene, flags = load(path)
    compare(z, passes.run_frames(oracle, scene, flags, frames=2))


@pytest.mark.gpu
@pytest.mark.parametrize("use_soa", [True, False])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_reproduces_reference_fixtures(path, use_soa):
    from niagara_amd import pipeline as P

    import gpu_passes as G
    z, scene, flags = load(path)
    ctx = P.Context()
    try:
        compare(z, G.run_frames(ctx, scene, flags, frames=2, use_soa=use_soa))
    finally:
        ctx.close()


TASK_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mesh", "taskcull.npz")


def _task_inputs(z):
    pyr = oracle.Pyramid(*(int(x) for x in z["viewport"]))
    oracle.depthreduce(z["depth"], pyr)
    return pyr, int(z["count4"][1]) * 64


def _check_task(z, late, pay, cnt, mvb):
    assert (cnt == z["late%d_counts" % late]).all()
    live = np.arange(64)[None, :] < cnt[:, None]
    assert (pay[live] == z["late%d_payloads" % late][live]).all()
    assert mvb.tobytes() == z["late%d_mvb" % late].tobytes()


def test_oracle_reproduces_the_reference_task_shader_fixture():
    """tests/golden/mesh/taskcull.npz was written by meshlet.task.glsl executing on the CPU (generate.py::task_fixture)"""
    z = np.load(TASK_FIXTURE)
    pyr, ncmd = _task_inputs(z)
    for late in (0, 1):
        pay, cnt, mvb = np.zeros((ncmd, 64), np.uint32), np.zeros(ncmd, np.uint32), z["mvb0"].copy()
        oracle.taskcull(z["cull"], late, z["commands"], z["count4"], z["draws"], z["meshlets"], mvb, pyr, pay, cnt)
        _check_task(z, late, pay, cnt, mvb)
        assert cnt.sum() > 100


@pytest.mark.gpu
@pytest.mark.parametrize("use_soa", [True, False])
def test_hip_reproduces_the_reference_task_shader_fixture(use_soa):
    import torch
    from niagara_amd import pipeline as P
    z = np.load(TASK_FIXTURE)
    _, ncmd = _task_inputs(z)
    vw, vh = (int(x) for x in z["viewport"])
    ctx = P.Context()
    try:
        dev = ctx.device
        pyr = P.DepthPyramid(dev, vw, vh)
        ctx.depthreduce(torch.from_numpy(np.ascontiguousarray(z["depth"])).to(dev), vw, vh, pyr.desc)
        db, mlb, dcb = P.to_device(z["draws"], dev), P.to_device(z["meshlets"], dev), P.to_device(z["commands"], dev)
        if use_soa:
            ctx.upload_meshlets(mlb, len(z["meshlets"]))
        dccb = torch.from_numpy(z["count4"].view(np.int32).copy()).to(dev)
        for late in (0, 1):
            d_pay = torch.zeros(ncmd * 64, dtype=torch.int32, device=dev)
            d_cnt = torch.zeros(ncmd, dtype=torch.int32, device=dev)
            d_mvb = torch.from_numpy(z["mvb0"].view(np.int32).copy()).to(dev)
            ctx.taskcull(z["cull"], late, dcb, dccb, db, mlb, d_mvb, pyr.desc, d_pay, d_cnt)
            _check_task(z, late, d_pay.cpu().numpy().view(np.uint32).reshape(ncmd, 64), d_cnt.cpu().numpy().view(np.uint32),
                        d_mvb.cpu().numpy().view(np.uint32))
    finally:
        ctx.close()
