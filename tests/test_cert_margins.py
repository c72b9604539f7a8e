"""The margins behind clustercull.hip's conservative frustum filter and certified two-sided test, checked on the CPU against the
constants actually compiled.

niagara_amd/csrc/filtermath.h — the per-draw derivation (filter_make), the margin scale (filter_k) and every constant the error analysis
fixes (K = 48, the 1.001 slack, aR = 2^-20, coneK = 2.02 ||V|| rot + 1 ...) — is compiled HERE with g++ through tests/cert_shim.cpp; nothing
is re-typed.  The per-meshlet arithmetic of certainly_outside / certified_visible (FMA chains over the packed halfs, clustercull.hip) is
restated once with the FMAs emulated in fp64 (product + addend rounded to fp32: within an ulp of a true FMA) for the bulk, and in EXACT
rational arithmetic (fractions.Fraction, correctly rounded to fp32 after every operation) for a sample — and held against the reference
arithmetic's own intermediates (oracle.probe_cluster_scalars: the view-space centre, dot(c, axis), cutoff |c| + r, the frustum and
cone decisions, src/shaders/clustercull.comp.glsl:72-108, math.h:41-44):

    |c~ - c_ref|_inf <= E = T / 4,   |D~ - (lhs_ref - rhs_ref)| <= T coneK / 2,   and no certain decision contradicts the reference's.

The bounded form of tools/experiments/cert_margin.py (which prints the slack per scene class); runs in seconds, needs no GPU.  The dropped
certified HiZ probe (tools/experiments/hiz_cert_margin.py) is not in the product and has no test."""
import ctypes as C
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

import oracle
from niagara_amd import host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("cert") / "cert_shim.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", os.path.join(ROOT, "tests", "cert_shim.cpp"), "-o", so], check=True)
    lib = C.CDLL(so)
    lib.shim_filter_k.restype = C.c_float
    lib.shim_filter_k.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.shim_fold_sound.restype = C.c_int
    lib.shim_fold_sound.argtypes = [C.c_void_p]
    lib.shim_make_filters.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_float, C.c_float, C.c_float, C.c_void_p]
    return lib


def pool_bounds(meshlets):
    """{3 x the largest |centre component|, the largest |radius|} of a pool, as clustercull.hip pool_bounds_kernel / pool_bounds_finish find them"""
    c = meshlets["center"].view(np.uint16) & 0x7fff
    r = meshlets["radius"].view(np.uint16) & 0x7fff
    v, rr = int(c.max(initial=0)), int(r.max(initial=0))
    to_f = lambda bits: f32(np.inf) if bits >= 0x7c00 else f32(np.uint16(bits).view(np.float16))  # noqa: E731
    return f32(3.0) * to_f(v), to_f(rr)


def filters_of(shim, cd, draws, vmax3=np.inf, rmax=np.inf):
    """(filterK, FilterDraw rows [n, 19]) exactly as fill_cluster_args / make_filter derive them"""
    fr = np.ascontiguousarray(cd["frustum"][0], f32)
    k = shim.shim_filter_k(fr.ctypes.data, float(cd["znear"][0]), float(cd["zfar"][0]))
    view = np.ascontiguousarray(cd["view"][0], f32)
    out = np.zeros((len(draws), 19), f32)
    d = np.ascontiguousarray(draws)
    shim.shim_make_filters(view.ctypes.data, d.ctypes.data, d.dtype.itemsize // 4, len(d), C.c_float(k), C.c_float(vmax3), C.c_float(rmax), out.ctypes.data)
    return k, out


def test_constants_are_the_analysed_ones(shim):
    """the numbers the written error analysis (clustercull.hip, above make_filter / certified_visible) was done for; changing one means
    re-doing the analysis and this line"""
    c = (C.c_float * 12)()
    shim.shim_constants(c)
    assert list(c) == [48.0, 2.0 ** -24, f32(1.001), f32(1e-30), 2.0 ** -20, f32(1e12), f32(1e-15), f32(1e3), f32(1e30), f32(2.02), 1.0, f32(1.0 / 127.0)]
    unit = np.array([0.6, 0.8, 0.28, 0.96], f32)
    assert shim.shim_filter_k(unit.ctypes.data, 0.1, 200.0) == f32(f32(f32(f32(4.0) * f32(48.0)) * f32(2.0 ** -24)) * f32(1.001)) * f32(f32(0.6) + f32(0.8))
    for bad in (np.array([np.nan, 0, 0, 0], f32), np.array([np.inf, 0, 0, 0], f32), np.array([2e3, 0, 0, 0], f32)):
        assert shim.shim_filter_k(bad.ctypes.data, 0.1, 200.0) == 0.0
    assert shim.shim_filter_k(unit.ctypes.data, np.inf, 200.0) == 0.0 and shim.shim_filter_k(unit.ctypes.data, 0.1, np.nan) == 0.0


def fma64(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def rn32(x):
    """a Fraction correctly rounded (nearest, ties to even) to fp32, as a Fraction"""
    if x == 0:
        return Fraction(0)
    s, a = (-1 if x < 0 else 1), abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    q = Fraction(2) ** (max(e, -126) - 23)
    n, r = divmod(a, q)
    n = int(n)
    if r * 2 > q or (r * 2 == q and n & 1):
        n += 1
    return s * n * q


def scene(radius, scale_mul, qmul, cam, camq, n_draws, cpd, seed):
    rng = np.random.default_rng(seed)
    draws = host.synth_draws(n_draws, 1, radius)
    draws["scale"] *= f32(scale_mul)
    draws["orientation"] *= f32(qmul)
    q = np.asarray(camq, np.float64)
    cd = host.build_cull_data(cam_pos=cam, cam_quat=tuple(q / np.linalg.norm(q)), draw_count=n_draws, cullingEnabled=1, clusterBackfaceEnabled=1,
                              draw_distance=max(200.0, radius))
    commands = synth.make_task_commands(n_draws, cpd)[:n_draws * cpd]
    meshlets = synth.make_meshlets(n_draws * cpd * 64, seed=int(rng.integers(1 << 30)))
    meshlets["cone_axis"][::7] = rng.integers(-128, 128, (len(meshlets[::7]), 3)).astype(np.int8)  # all int8 values incl. -128
    meshlets["cone_cutoff"][::5] = rng.integers(-128, 128, len(meshlets[::5])).astype(np.int8)
    return draws, meshlets, commands, cd


CASES = [("config 3A geometry (radius 300)", 300.0, 1.0, 1.0, (0, 0, 0), (0, 0, 0, 1)),
         ("dense (radius 40)", 40.0, 1.0, 1.0, (0, 0, 0), (0, 0, 0, 1)),
         ("tiny scene 1e-3", 1e-3, 1e-3, 1.0, (0, 0, 0), (0, 0, 0, 1)),
         ("huge scene 3e5", 3e5, 100.0, 1.0, (1e4, -2e4, 3e3), (0.3, -0.5, 0.2, 0.78)),
         ("non-unit quaternions x7.5", 100.0, 1.0, 7.5, (5, 6, 7), (0.1, 0.7, -0.1, 0.69)),
         ("camera far from origin", 300.0, 1.0, 1.0, (5e3, 5e3, -5e3), (0.5, 0.5, 0.5, 0.5)),
         ("small quaternions x0.01", 50.0, 2.0, 0.01, (0, 0, 0), (0, 0, 0, 1))]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_margins_cover_the_distance_to_the_reference(shim, case):
    name, radius, scale_mul, qmul, cam, camq = case
    draws, meshlets, commands, cd = scene(radius, scale_mul, qmul, cam, camq, n_draws=400, cpd=2, seed=11)
    n = len(commands)
    probe = oracle.probe_cluster_scalars(cd, commands, draws, meshlets)  # (n, 64, 16): the reference arithmetic's intermediates
    filterK, F = filters_of(shim, cd, draws, *pool_bounds(meshlets))
    assert filterK > 0
    d = commands["drawId"]
    ml = meshlets[commands["taskOffset"][:, None] + np.arange(64, dtype=np.uint32)[None, :]]
    v = ml["center"].view(np.float16).astype(f32)  # (n, 64, 3)
    rad = ml["radius"].view(np.float16).astype(f32)
    Fd = F[d][:, None, :] + 0 * rad[..., None]     # (n, 64, 19)
    m, b, aK, bK, aR, scale, coneK, is127, tK = Fd[..., 0:9], Fd[..., 9:12], Fd[..., 12], Fd[..., 13], Fd[..., 14], Fd[..., 15], Fd[..., 16], Fd[..., 17], Fd[..., 18]
    # certainly_outside / certified_visible (clustercull.hip), FMA = fp64 product-sum rounded to fp32
    c = np.stack([fma64(m[..., 3 * r], v[..., 0], fma64(m[..., 3 * r + 1], v[..., 1], fma64(m[..., 3 * r + 2], v[..., 2], b[..., r]))) for r in range(3)], axis=-1)
    T = fma64(aK, np.abs(v[..., 0]), bK)
    T = fma64(aK, np.abs(v[..., 1]), T)
    T = fma64(aK, np.abs(v[..., 2]), T)
    T = fma64(aR, np.abs(rad), T)
    c_ref = probe[..., 0:3]
    ok = np.isfinite(c_ref).all(axis=-1) & np.isfinite(T)
    assert ok.mean() > 0.99
    ratio_c = np.where(ok, np.max(np.abs(c.astype(np.float64) - c_ref), axis=-1) / (T.astype(np.float64) / 4.0), 0.0)
    assert ratio_c.max() < 0.5, "%s: |c~ - c_ref| reaches %.3f of E" % (name, ratio_c.max())
    # frustum: the two certain decisions against the reference's own predicate (probe[..., 14])
    fr, znear, zfar = cd["frustum"][0].astype(f32), f32(cd["znear"][0]), f32(cd["zfar"][0])
    thr_hi, thr_lo = fma64(scale, rad, T), fma64(scale, rad, -T)
    g1 = fma64(c[..., 2], fr[1], -(np.abs(c[..., 0]) * fr[0]).astype(f32))
    g2 = fma64(c[..., 2], fr[3], -(np.abs(c[..., 1]) * fr[2]).astype(f32))
    g = np.minimum(np.minimum(g1, g2), np.minimum(c[..., 2] - znear, zfar - c[..., 2]))
    out_m, in_m, vis_ref = g < -thr_hi, g > -thr_lo, probe[..., 14] != 0
    assert not (ok & out_m & vis_ref).any() and not (ok & in_m & ~vis_ref).any(), name
    # the filter pass (certainly_outside) uses the per-draw margin tK = T at the pool's largest |centre component| and |radius|: never
    # smaller than a meshlet's own T by more than its three roundings, so it can only decide LESS; and it must still decide nearly everything
    assert np.isfinite(tK).all() and (tK.astype(np.float64) >= T.astype(np.float64) * (1.0 - 2.0 ** -21))[ok].all(), name
    out_f = g < -fma64(scale, rad, tK)
    assert not (ok & out_f & vis_ref).any(), name
    # ... and in the form the early pass's filter loop evaluates (clustercull.hip certainly_outside<FOLD>): the x / y rows of M and b pre-multiplied by the side
    # planes' coefficients (one fp32 rounding per entry), a side plane's distance one FMA  cz f1 - |f0 cx|
    ms, bs = m.copy(), b.copy()
    ms[..., 0:3] = (fr[0] * m[..., 0:3]).astype(f32)
    ms[..., 3:6] = (fr[2] * m[..., 3:6]).astype(f32)
    bs[..., 0] = (fr[0] * b[..., 0]).astype(f32)
    bs[..., 1] = (fr[2] * b[..., 1]).astype(f32)
    cs = np.stack([fma64(ms[..., 3 * r], v[..., 0], fma64(ms[..., 3 * r + 1], v[..., 1], fma64(ms[..., 3 * r + 2], v[..., 2], bs[..., r]))) for r in range(3)], axis=-1)
    g1s = fma64(cs[..., 2], fr[1], -np.abs(cs[..., 0]))
    g2s = fma64(cs[..., 2], fr[3], -np.abs(cs[..., 1]))
    gs = np.minimum(np.minimum(g1s, g2s), np.minimum(cs[..., 2] - znear, zfar - cs[..., 2]))
    out_fold = gs < -fma64(scale, rad, tK)
    assert not (ok & out_fold & vis_ref).any(), name
    assert (ok & out_fold).sum() >= 0.97 * (ok & out_m).sum(), name
    # the folded distances stay within a small part of the margin of the unfolded ones (what one more rounding per entry may cost)
    drift = np.where(ok, np.maximum(np.abs(g1s.astype(np.float64) - g1), np.abs(g2s.astype(np.float64) - g2)) / T.astype(np.float64), 0.0)
    assert drift.max() < 0.05, "%s: folding the plane coefficients moves a distance by %.3f of T" % (name, drift.max())
    assert (ok & out_f).sum() >= 0.97 * (ok & out_m).sum(), "%s: the pool-wide margin gives up %d of %d certain rejections" % (name, (ok & out_m & ~out_f).sum(), (ok & out_m).sum())
    assert (ok & (out_m | in_m)).mean() > 0.9  # the margins must not be so wide that nothing is decided
    # the direct form's packed walk (clustercull.hip packed_walk, round 6) runs the TWO-SIDED test with the per-draw margin tK as well: both of its certain
    # frustum decisions against the reference's predicate (its cone decisions with tK coneK: below)
    in_t = g > -fma64(scale, rad, -tK)
    assert not (ok & in_t & ~vis_ref).any(), name
    assert (ok & (out_f | in_t)).sum() >= 0.97 * (ok & (out_m | in_m)).sum(), name
    # cone
    k, kc = ml["cone_axis"].astype(f32), ml["cone_cutoff"].astype(f32)
    w = np.stack([fma64(m[..., 3 * r], k[..., 0], fma64(m[..., 3 * r + 1], k[..., 1], (m[..., 3 * r + 2] * k[..., 2]).astype(f32))) for r in range(3)], axis=-1)
    lhs = (fma64(c[..., 0], w[..., 0], fma64(c[..., 1], w[..., 1], (c[..., 2] * w[..., 2]).astype(f32))) * is127).astype(f32)
    len2 = fma64(c[..., 0], c[..., 0], fma64(c[..., 1], c[..., 1], (c[..., 2] * c[..., 2]).astype(f32)))
    root = np.sqrt(len2.astype(np.float64)).astype(f32)
    D_ref = probe[..., 4].astype(np.float64) - probe[..., 5].astype(np.float64)
    Tc = (T * coneK).astype(f32)
    okc = ok & np.isfinite(D_ref) & np.isfinite(Tc)
    cull_ref = probe[..., 15] != 0
    inv127 = f32(1.0 / 127.0)
    for ln in (root, np.nextafter(root, f32(np.inf)), np.nextafter(root, f32(-np.inf))):  # v_sqrt_f32: within an ulp
        D = (lhs - fma64(kc * inv127, ln, (scale * rad).astype(f32))).astype(f32)
        ratio_d = np.where(okc, np.abs(D.astype(np.float64) - D_ref) / Tc.astype(np.float64), 0.0)
        assert ratio_d.max() < 0.5, "%s: |D~ - D_ref| reaches %.3f of the cone margin" % (name, ratio_d.max())
        assert not (okc & (((D > Tc) & ~cull_ref) | ((D < -Tc) & cull_ref))).any(), name
        Tct = (tK * coneK).astype(f32)  # the packed walk's cone margin
        assert not (okc & np.isfinite(Tct) & (((D > Tct) & ~cull_ref) | ((D < -Tct) & cull_ref))).any(), name

    # ---- the same for a sample of lanes in exact rational arithmetic: every FMA = the exact a b + c rounded once to fp32
    rng = np.random.default_rng(3)
    Fr = Fraction
    for ci, li in zip(rng.integers(0, n, 48), rng.integers(0, 64, 48)):
        if not okc[ci, li]:
            continue
        fd = [Fr(float(x)) for x in F[d[ci]]]
        vx, vy, vz, r_ = (Fr(float(x)) for x in (*v[ci, li], rad[ci, li]))
        fma = lambda a, b_, c_: rn32(a * b_ + c_)  # noqa: E731
        ce = [fma(fd[3 * r], vx, fma(fd[3 * r + 1], vy, fma(fd[3 * r + 2], vz, fd[9 + r]))) for r in range(3)]
        Te = fma(fd[14], abs(r_), fma(fd[12], abs(vz), fma(fd[12], abs(vy), fma(fd[12], abs(vx), fd[13]))))
        cr = [Fr(float(x)) for x in c_ref[ci, li]]
        assert max(abs(a - b_) for a, b_ in zip(ce, cr)) * 4 * 2 < Te, (name, int(ci), int(li))
        kx, ky, kz, kce = (Fr(int(x)) for x in (*ml["cone_axis"][ci, li], ml["cone_cutoff"][ci, li]))
        we = [fma(fd[3 * r], kx, fma(fd[3 * r + 1], ky, rn32(fd[3 * r + 2] * kz))) for r in range(3)]
        lhs_e = rn32(fma(ce[0], we[0], fma(ce[1], we[1], rn32(ce[2] * we[2]))) * fd[17])
        len2_e = fma(ce[0], ce[0], fma(ce[1], ce[1], rn32(ce[2] * ce[2])))
        ln_e = Fr(float(np.sqrt(np.float64(float(len2_e))).astype(f32)))  # fp32 root of an fp32 value (exactly representable input)
        De = rn32(lhs_e - fma(rn32(kce * Fr(float(inv127))), ln_e, rn32(fd[15] * r_)))
        Tce = rn32(Te * fd[16])
        assert abs(De - Fr(float(D_ref[ci, li]))) * 2 < Tce, (name, int(ci), int(li))
        if De > Tce:
            assert cull_ref[ci, li]
        if De < -Tce:
            assert not cull_ref[ci, li]


@pytest.mark.parametrize("signs", [(-1, 1), (1, -1), (-1, -1)], ids=["f0<0", "f2<0", "both<0"])
def test_negative_side_plane_coefficients(shim, signs):
    """CullData.frustum may hold anything (a mirrored / flipped projection negates a side plane's coefficient).  The reference
    computes cz f1 - |cx| f0 with the coefficient's sign; pass B's forms (certified_visible, the unfolded filter) do the same and stay sound; the FOLDED
    filter of the early pass's loop computes cz f1 - |f0 cx| and would reject clusters the reference keeps — which is why filtermath.h
    filter_fold_sound switches the filter off for such input (cluster_mask_kernel: foldSound)."""
    draws, meshlets, commands, cd = scene(300.0, 1.0, 1.0, (0, 0, 0), (0, 0, 0, 1), n_draws=400, cpd=2, seed=11)
    fr = cd["frustum"][0].astype(f32).copy()
    assert shim.shim_fold_sound(np.ascontiguousarray(fr).ctypes.data) == 1
    fr[0] *= f32(signs[0])
    fr[2] *= f32(signs[1])
    cd = cd.copy()
    cd["frustum"][0] = fr
    assert shim.shim_fold_sound(np.ascontiguousarray(fr).ctypes.data) == 0
    for special in (np.array([np.nan, 1, 1, 1], f32), np.array([1, 1, np.nan, 1], f32), np.array([-0.0, 0, 0.5, 0], f32)):
        assert shim.shim_fold_sound(special.ctypes.data) == (0 if np.isnan(special).any() else 1)  # (-0.0 >= 0: the product is the reference's, a signed zero)
    probe = oracle.probe_cluster_scalars(cd, commands, draws, meshlets)
    filterK, F = filters_of(shim, cd, draws, *pool_bounds(meshlets))
    assert filterK > 0  # (|f| is what filter_k looks at: the filter stays on as far as the host is concerned)
    d = commands["drawId"]
    ml = meshlets[commands["taskOffset"][:, None] + np.arange(64, dtype=np.uint32)[None, :]]
    v = ml["center"].view(np.float16).astype(f32)
    rad = ml["radius"].view(np.float16).astype(f32)
    Fd = F[d][:, None, :] + 0 * rad[..., None]
    m, b, aK, bK, aR, scale, tK = Fd[..., 0:9], Fd[..., 9:12], Fd[..., 12], Fd[..., 13], Fd[..., 14], Fd[..., 15], Fd[..., 18]
    c = np.stack([fma64(m[..., 3 * r], v[..., 0], fma64(m[..., 3 * r + 1], v[..., 1], fma64(m[..., 3 * r + 2], v[..., 2], b[..., r]))) for r in range(3)], axis=-1)
    T = fma64(aR, np.abs(rad), fma64(aK, np.abs(v[..., 2]), fma64(aK, np.abs(v[..., 1]), fma64(aK, np.abs(v[..., 0]), bK))))
    ok = np.isfinite(probe[..., 0:3]).all(axis=-1) & np.isfinite(T)
    znear, zfar = f32(cd["znear"][0]), f32(cd["zfar"][0])
    g1 = fma64(c[..., 2], fr[1], -(np.abs(c[..., 0]) * fr[0]).astype(f32))
    g2 = fma64(c[..., 2], fr[3], -(np.abs(c[..., 1]) * fr[2]).astype(f32))
    g = np.minimum(np.minimum(g1, g2), np.minimum(c[..., 2] - znear, zfar - c[..., 2]))
    vis_ref = probe[..., 14] != 0
    assert vis_ref.sum() > 1000
    # the forms that keep the sign: sound in both directions
    out_m, in_m = g < -fma64(scale, rad, T), g > -fma64(scale, rad, -T)
    assert not (ok & out_m & vis_ref).any() and not (ok & in_m & ~vis_ref).any()
    assert not (ok & (g < -fma64(scale, rad, tK)) & vis_ref).any()
    # the folded form: rejects what the reference keeps
    ms, bs = m.copy(), b.copy()
    ms[..., 0:3] = (fr[0] * m[..., 0:3]).astype(f32)
    ms[..., 3:6] = (fr[2] * m[..., 3:6]).astype(f32)
    bs[..., 0] = (fr[0] * b[..., 0]).astype(f32)
    bs[..., 1] = (fr[2] * b[..., 1]).astype(f32)
    cs = np.stack([fma64(ms[..., 3 * r], v[..., 0], fma64(ms[..., 3 * r + 1], v[..., 1], fma64(ms[..., 3 * r + 2], v[..., 2], bs[..., r]))) for r in range(3)], axis=-1)
    gs = np.minimum(np.minimum(fma64(cs[..., 2], fr[1], -np.abs(cs[..., 0])), fma64(cs[..., 2], fr[3], -np.abs(cs[..., 1]))), np.minimum(cs[..., 2] - znear, zfar - cs[..., 2]))
    wrong = ok & (gs < -fma64(scale, rad, tK)) & vis_ref
    assert wrong.sum() > 100, "the folded form no longer over-rejects with a negated coefficient: is the guard still needed?"


def test_unsound_inputs_make_nothing_certain(shim):
    """non-finite or absurd draw fields: the margin becomes inf / NaN, every `certain` comparison false (clustercull.hip make_filter)"""
    draws = host.synth_draws(8, 1, 300.0)
    draws["position"][0, 1] = np.nan
    draws["scale"][1] = np.inf
    draws["orientation"][2, 3] = np.nan
    draws["scale"][3] = 1e-20
    draws["position"][4, 0] = 1e20
    draws["scale"][5] = 1e14
    cd = host.build_cull_data(draw_count=8, cullingEnabled=1, clusterBackfaceEnabled=1)
    _, F = filters_of(shim, cd, draws, 3.0, 0.1)
    for i in range(6):
        assert not np.isfinite(F[i, 13]) or not np.isfinite(F[i, 12]), i  # bK (or aK) poisons T
    assert not np.isfinite(F[:6, 18]).any() and np.isfinite(F[6:, :19]).all()
    # a pool with a non-finite centre or radius: no finite filter margin for any draw
    m = synth.make_meshlets(256, seed=1)
    assert all(np.isfinite(x) for x in pool_bounds(m))
    m["center"].view(np.uint16)[17, 1] = 0x7c00
    assert not np.isfinite(pool_bounds(m)[0])
    _, F = filters_of(shim, cd, draws, *pool_bounds(m))
    assert not np.isfinite(F[:, 18]).any()
    bad_view = cd.copy()
    bad_view["view"][0][5] = np.nan
    _, F = filters_of(shim, bad_view, draws)
    assert not np.isfinite(F[:, 13]).any()
