"""The cull launch's static dealing (niagara_amd/csrc/dealing.h) on the CPU: the header the kernels compile is built HERE with g++ (tests/dealing_shim.cpp)
and held to what the kernel relies on — every chunk of a pass goes to exactly one wave, whatever the command count; the plan the HOST prepares (division
magics) is the plan a wave derives itself (plain divisions); the weighted rounds apply at the headline's shape; and a command's scatter tile by one mulhi
(DealPlan::tileMul31) is index / tileCmds for every command of the pass."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORDER = ["cmds", "flags", "weighted", "r0", "r1", "r2", "r3", "r4", "r5", "weightedTotal", "restPerWave", "restRem", "perWaveChunks", "evenRem", "tileCmds", "tileMul31",
         "numTiles", "waves"]


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("deal") / "dealing_shim.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "dealing_shim.cpp"), "-o", so], check=True)
    lib = C.CDLL(so)
    lib.shim_deal_all.restype = C.c_uint32
    lib.shim_tile_of.restype = C.c_uint32
    return lib


def plan_of(lib, n, grid=1536, chunk=4, weighted=1, scale=100, tiles=256, magic=1, generations=6):
    """generations = workgroups per CU of the launch (ClusterArgs::generations; the weighted dealing is calibrated for six)"""
    out = np.zeros(len(ORDER), np.uint32)
    lib.shim_deal_plan(C.c_uint32(n), C.c_uint32(chunk), C.c_int(weighted), C.c_uint32(grid), C.c_uint32(generations), C.c_uint32(scale), C.c_uint32(tiles), C.c_int(magic),
                       out.ctypes.data_as(C.c_void_p))
    return out


def field(plan, name):
    return int(plan[ORDER.index(name)])


COUNTS = [0, 1, 3, 4, 5, 63, 64, 100, 6143, 6144 * 4, 6144 * 4 + 1, 24577, 100000, 156250, 175000, 250114, 262144, 262145, 1_000_000, 1_562_500, 65535 * 64]


@pytest.mark.parametrize("grid", [1536, 768, 1530, 256 * 8])
def test_every_chunk_goes_to_exactly_one_wave(shim, grid):
    rng = np.random.default_rng(grid)
    for n in COUNTS + [int(x) for x in rng.integers(1, 3_000_000, 12)]:
        for weighted, scale in ((1, 100), (0, 100), (1, 1100), (1, 1250)):  # (1000 + s: the packed walk's delay table at s %: dealing.h DEAL_PACKED_TABLE)
            plan = plan_of(shim, n, grid=grid, weighted=weighted, scale=scale)
            chunks = (n + 3) // 4
            owner, per_wave = np.zeros(max(chunks, 1), np.uint32), np.zeros(grid * 4, np.uint32)
            bad = shim.shim_deal_all(plan.ctypes.data_as(C.c_void_p), C.c_uint32(grid), C.c_uint32(chunks), owner.ctypes.data_as(C.c_void_p), per_wave.ctypes.data_as(C.c_void_p))
            assert bad == 0, (n, grid, weighted, scale)
            assert (owner[:chunks] != 0xFFFFFFFF).all(), (n, grid, weighted, scale, int((owner[:chunks] == 0xFFFFFFFF).sum()))
            assert chunks <= int(per_wave.sum()) <= chunks + grid * 4  # at most one slot per wave past the end
            if field(plan, "weighted"):
                assert per_wave.max() <= 64  # a wave's chunk table is one VGPR


def test_the_headline_shape_is_dealt_weighted_and_later_generations_take_fewer_rounds(shim):
    plan = plan_of(shim, 156250)
    assert field(plan, "weighted") == 1 and field(plan, "waves") == 6144 and field(plan, "cmds") == 156250
    rounds = [field(plan, "r%d" % k) for k in range(6)]
    assert rounds == sorted(rounds, reverse=True) and rounds[0] > rounds[5]
    packed = plan_of(shim, 156250, scale=1100)  # the packed walk's table: steeper, the last generation still takes part
    rp = [field(packed, "r%d" % k) for k in range(6)]
    assert field(packed, "weighted") == 1 and rp == sorted(rp, reverse=True) and rp[0] - rp[5] > rounds[0] - rounds[5]
    assert field(plan, "tileCmds") == 768 and field(plan, "numTiles") == 204
    # other grid shapes, tiny passes and passes of more than 64 chunks per wave fall back to plain round-robin
    assert field(plan_of(shim, 156250, grid=768, generations=3), "weighted") == 0
    assert field(plan_of(shim, 156250, grid=1537), "weighted") == 0  # (a grid that is not six generations of equal size)
    assert field(plan_of(shim, 1000), "weighted") == 0
    assert field(plan_of(shim, 65535 * 64), "weighted") == 0
    assert field(plan_of(shim, 156250, weighted=0), "weighted") == 0


def test_the_hosts_plan_is_the_plan_a_wave_derives(shim):
    """the host divides through its prepared multipliers, a wave whose count word disagrees with the plan divides plainly: the same plan either way"""
    rng = np.random.default_rng(1)
    for n in COUNTS + [int(x) for x in rng.integers(1, 4_000_000, 200)]:
        for scale in (100, 70, 0, 130, 1100, 1070):
            a, b = plan_of(shim, n, scale=scale, magic=1), plan_of(shim, n, scale=scale, magic=0)
            assert (a == b).all(), (n, scale, a, b)


def test_the_largest_grid_the_magic_is_offered_for(shim):
    """deal_magic is enabled up to a divisor of 8192 waves (a grid of 2048 workgroups), exact for dividends below 2^39 / 8192 = 67.1 M.  The
    weighted branch divides numChunks x 64 — but only for passes of fewer than 60 chunks per wave (below 3904 W = 32 M at W = 8192), the plain branch
    divides at most 2^22 chunks: host plan (magic) and wave-derived plan (plain division) agree over the whole range of counts at that grid, and every
    chunk still goes to exactly one wave."""
    grid = 2046  # (six generations of 341 workgroups: the largest weighted shape under the magic's limit of 8192 waves)
    W = grid * 4
    assert W <= 8192 < (grid + 6) * 4
    rng = np.random.default_rng(7)
    edge = [40 * W * 4, 59 * W * 4 - 1, 59 * W * 4, 60 * W * 4 - 4, 60 * W * 4 - 1, 60 * W * 4, 60 * W * 4 + 1, 61 * W * 4, 65535 * 64, 65535 * 64 - 1, 4 * W * 4, 4 * W * 4 - 1]
    for n in edge + [int(x) for x in rng.integers(1, 65535 * 64, 300)]:
        for scale in (100, 70, 130, 1100):
            a, b = plan_of(shim, n, grid=grid, scale=scale, magic=1), plan_of(shim, n, grid=grid, scale=scale, magic=0)
            assert (a == b).all(), (n, scale)
    assert field(plan_of(shim, 40 * W * 4, grid=grid), "weighted") == 1  # (this shape does reach the weighted division)
    for n in (40 * W * 4, 59 * W * 4, 60 * W * 4 - 1, 65535 * 64):
        plan = plan_of(shim, n, grid=grid)
        chunks = (n + 3) // 4
        owner, per_wave = np.zeros(chunks, np.uint32), np.zeros(W, np.uint32)
        assert shim.shim_deal_all(plan.ctypes.data_as(C.c_void_p), C.c_uint32(grid), C.c_uint32(chunks), owner.ctypes.data_as(C.c_void_p), per_wave.ctypes.data_as(C.c_void_p)) == 0
        assert (owner != 0xFFFFFFFF).all()


def test_a_commands_tile_by_one_mulhi(shim):
    rng = np.random.default_rng(2)
    for n in [1, 255, 256, 257, 65535, 65536, 65537, 156250, 250114, 1_562_500, 65535 * 64] + [int(x) for x in rng.integers(1, 65535 * 64, 30)]:
        plan = plan_of(shim, n)
        t, mul, tiles = field(plan, "tileCmds"), field(plan, "tileMul31"), field(plan, "numTiles")
        assert t % 256 == 0 and t >= 256 and tiles == (n + t - 1) // t and tiles <= 256
        idx = np.unique(np.concatenate([np.arange(0, min(n, 4096)), rng.integers(0, n, 4096), np.arange(max(0, n - 4096), n),
                                        (np.arange(1, tiles + 1) * t - 1).clip(0, n - 1), (np.arange(0, tiles) * t).clip(0, n - 1)])).astype(np.uint64)
        got = ((idx >> np.uint64(8) << np.uint64(1)) * np.uint64(mul)) >> np.uint64(32)
        assert (got == idx // np.uint64(t)).all(), n
        assert shim.shim_tile_of(C.c_uint32(n - 1), C.c_uint32(mul)) == (n - 1) // t
