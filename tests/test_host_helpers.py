"""Host-side helpers of the C ABI (no GPU): bit-identical to the CPU oracle and, through it, to the reference's own
host code (src/niagara.cpp:424-481,969-1020; src/resources.cpp:280-292)."""
import numpy as np

import oracle
from niagara_amd import host, synth
from niagara_amd import layouts as L


def test_cull_data_matches_oracle_for_random_cameras():
    rng = np.random.default_rng(0)
    for _ in range(200):
        pos = rng.uniform(-50, 50, 3)
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        vw, vh = int(rng.integers(64, 4097)), int(rng.integers(64, 4097))
        kw = dict(cam_pos=pos, cam_quat=q, fovy=float(rng.uniform(0.3, 2.0)), znear=float(rng.uniform(0.01, 1)),
                  draw_distance=float(rng.uniform(50, 500)), viewport=(vw, vh), pyramid=(host.previous_pow2(vw), host.previous_pow2(vh)),
                  draw_count=int(rng.integers(0, 1 << 20)), lod_step=int(rng.integers(0, 4)))
        assert host.build_cull_data(**kw).tobytes() == oracle.make_cull_data(**kw).tobytes()


def test_default_camera_is_niagaras():
    cd = host.build_cull_data()
    assert (cd["view"][0] == np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1], np.float32)).all()
    assert cd["znear"][0] == np.float32(0.1) and cd["zfar"][0] == np.float32(200)


def test_synthetic_scene_and_visibility_slots_match_oracle():
    d1, d2 = host.synth_draws(5000, 7, 300.0), oracle.synth_draws(5000, 7, 300.0)
    assert d1.tobytes() == d2.tobytes()
    meshes, _ = synth.make_meshes(7, 5, 300)
    d1["postPass"] = d2["postPass"] = (np.arange(5000) % 11 == 0)
    assert host.assign_visibility_offsets(d1, meshes) == oracle.assign_visibility_offsets(d2, meshes)
    assert d1.tobytes() == d2.tobytes()
    # slot base = running sum of max-over-LODs meshletCount (src/niagara.cpp:1008-1016)
    widest = np.array([m["lods"]["meshletCount"][:m["lodCount"]].max() for m in meshes])
    assert (d1["meshletVisibilityOffset"] == np.r_[0, np.cumsum(widest[d1["meshIndex"]])[:-1]]).all()


def test_pyramid_geometry():
    for w, h in [(1024, 768), (4096, 4096), (1920, 1080), (2, 2), (3, 1), (1, 1)]:
        d = host.pyramid_desc(w, h)
        p = oracle.Pyramid(w, h)
        assert (d.width, d.height, d.levels, d.totalTexels) == (p.width, p.height, p.levels, p.s.totalTexels)
        assert list(d.mipOffset) == p.mip_offset
    assert host.previous_pow2(4096) == 2048 and host.pyramid_desc(4096, 4096).levels == 12


def test_count4_for_matches_tasksubmit():
    for n in [0, 1, 63, 64, 65, 156250, L.TASK_WGLIMIT, L.TASK_WGLIMIT + 9]:
        c4 = np.array([n, 0, 0, 0], np.uint32)
        oracle.tasksubmit(c4, np.zeros(min(n, L.TASK_WGLIMIT) + 64, dtype=L.TASKCMD))
        assert (synth.count4_for(n) == c4).all()


def test_s8_over_127_is_exact():
    """The HIP cone unpack replaces int8 / 127.0f by  q = k*c,  q' = fma(fma(-127, q, k), c, q)  (clustercull.hip
    s8_over_127).  Exact-rational re-evaluation: the result equals the correctly rounded quotient for all 256 inputs,
    i.e. what the reference's division (clustercull.comp.glsl:78-80) produces."""
    from fractions import Fraction

    def rn(fr):
        f32 = np.float32(float(fr))
        cands = [np.nextafter(f32, np.float32(-np.inf)), f32, np.nextafter(f32, np.float32(np.inf))]
        return np.float32(min(cands, key=lambda c: (abs(Fraction(float(c)) - fr), int(np.float32(c).view(np.uint32)) & 1)))

    c = np.float32(0.00787401571869850158691406250)
    assert c.view(np.uint32) == 0x3c010204 and c == rn(Fraction(1, 127))
    fc = Fraction(float(c))
    for k in range(-128, 128):
        q = rn(Fraction(k) * fc)
        r = rn(Fraction(k) - 127 * Fraction(float(q)))
        q2 = rn(Fraction(float(q)) + Fraction(float(r)) * fc)
        assert q2 == np.float32(k) / np.float32(127.0), k
        assert q2 == rn(Fraction(k, 127)), k


def test_division_magic_is_exact_where_the_kernels_use_it():
    """nv_division_magic (host.cpp): mulhi(n, m) >> 7 == n // d for every d it serves (256 .. 8192) and every n < 2^39 / d — the
    cull launch divides chunk counts (< 2^20), 64 x chunk counts (< 2^26), workgroup indices and command counts (< 2^23) by its grid
    constants this way (clustercull.hip div_launch_constant).  Checked at the multiples of d and their neighbours, at random n and
    at the top of the range; outside the served divisors the magic is 0 and the kernels divide."""
    from niagara_amd._lib import lib
    rng = np.random.default_rng(5)
    for d in list(range(256, 8193, 7)) + [256, 257, 1024, 1536, 4096, 6144, 8191, 8192]:
        m = int(lib.nv_division_magic(d))
        assert 0 < m < 2 ** 32
        top = (1 << 39) // d
        q = rng.integers(0, top // d, 64, dtype=np.int64)
        n = np.concatenate([q * d, q * d + d - 1, q * d + 1, rng.integers(0, top, 256, dtype=np.int64), [0, 1, d - 1, d, top - 1, (1 << 26) - 1]])
        n = n[(n >= 0) & (n < top)]
        got = [((int(x) * m) >> 32) >> 7 for x in n]
        assert got == [int(x) // d for x in n], d
    for d in (0, 1, 42, 255, 8193, 1 << 20):
        assert int(lib.nv_division_magic(d)) == 0
