#!/usr/bin/env python3
"""hiz_cert_margin.py — CPU check of the certified HiZ probe of cluster_hiz_kernel (clustercull.hip, round 3).

The occlusion stage evaluates src/shaders/math.h:2-39 (projectSphere, getOcclusionMip) and the MIN sampler's footprint with
FAST arithmetic (v_rcp_f32 / v_sqrt_f32 instead of the IEEE division and square root, same formulas) and carries an absolute
error bound for every quantity a discrete decision hangs on: the mip level, the `fits` refinement, the footprint's texel
indices, and the final depth comparison.  A probe whose every decision is farther from its boundary than the bound is CERTAIN:
its texel addresses and its verdict are the reference's.  Any other probe is redone with the reference arithmetic.

This script emulates both paths in numpy fp32 (one rounding per operation, like the kernels), perturbs every fast
reciprocal / square root by up to +-1.5 ulp, and checks over millions of random spheres that NO certain probe differs from the
reference in level, texel indices, use flags or verdict; it prints the share of probes that fall back.

    python tools/experiments/hiz_cert_margin.py [--n 2000000] [--seed 1]
"""
import argparse

import numpy as np

f32 = np.float32
EPS = f32(2.0 ** -23)  # one ulp, relative


def perturb(x, rng, ulps=1.5):
    """a fast transcendental's result: within `ulps` ulp of the exact one"""
    return (x.astype(np.float64) * (1.0 + rng.uniform(-ulps, ulps, x.shape) * 2.0 ** -24)).astype(f32)


def ceil_log2_exact(x):
    u = x.view(np.uint32)
    e = ((u >> 23) & 0xff).astype(np.int32)
    m = u & 0x7fffff
    return e - 127 + (m != 0)  # normal numbers only (the callers guarantee x > 2^-100)


def reference(c, r, znear, P00, P11, pw, ph, width, height, levels):
    """fp32, one rounding per op, in the reference's order; returns dict of decisions"""
    cx, cy, cz = c
    active = ~(cz < r + znear)
    crx, cry, crz = cx * r, cy * r, cz * r
    czr2 = cz * cz - r * r
    out = {}
    with np.errstate(all="ignore"):
        vx = np.sqrt(cx * cx + czr2)
        minx = (vx * cx - crz) / (vx * cz + crx)
        maxx = (vx * cx + crz) / (vx * cz - crx)
        vy = np.sqrt(cy * cy + czr2)
        miny = (vy * cy - crz) / (vy * cz + cry)
        maxy = (vy * cy + crz) / (vy * cz - cry)
        ax, ay, az, aw = minx * P00, miny * P11, maxx * P00, maxy * P11
        A0 = ax * f32(0.5) + f32(0.5)
        A1 = aw * f32(-0.5) + f32(0.5)
        A2 = az * f32(0.5) + f32(0.5)
        A3 = ay * f32(-0.5) + f32(0.5)
        sx, sy = A2 - A0, A3 - A1
        m = np.maximum(sx * pw, sy * ph)
        pos = m > 0
        lvl = np.where(pos, ceil_log2_exact(np.where(pos, m, f32(1))), 0)
        lvl = np.clip(lvl, None, 32)
        has = lvl > 0
        scale = np.exp2((1 - np.where(has, lvl, 1)).astype(np.float64)).astype(f32)
        fx, fy = pw * scale, ph * scale
        u0, u1 = A0 * fx, A1 * fy
        fits = ((u0 - np.floor(u0)) + sx * fx <= f32(2.0)) & ((u1 - np.floor(u1)) + sy * fy <= f32(2.0))
        lvl = np.where(has, lvl - fits, 0)
        l = np.clip(lvl, 0, levels - 1)
        w = np.maximum(1, width >> l).astype(np.int64)
        h = np.maximum(1, height >> l).astype(np.int64)
        u = (A0 + A2) * f32(0.5)
        v = (A1 + A3) * f32(0.5)
        tx = u * w.astype(f32) - f32(0.5)
        ty = v * h.astype(f32) - f32(0.5)
        depth = znear / (cz - r)

    def foot(t, size):
        f0 = np.floor(t)
        fr = t - f0
        lim = size.astype(f32)
        f0 = np.where(f0 >= -1, f0, f32(-1))
        f0 = np.where(f0 > lim, lim, f0)
        a = f0.astype(np.int64)
        b = a + 1
        hi = size - 1
        return np.clip(a, 0, hi), np.clip(b, 0, hi), (f32(1) - fr) != 0, fr != 0

    x0, x1, ux0, ux1 = foot(tx, w)
    y0, y1, uy0, uy1 = foot(ty, h)
    out.update(active=active, level=l, x0=x0, x1=x1, y0=y0, y1=y1, use=(ux0 & uy0) * 1 + (ux1 & uy0) * 2 + (ux0 & uy1) * 4 + (ux1 & uy1) * 8, depth=depth)
    return out


def fast(c, r, znear, P00, P11, pw, ph, width, height, levels, rng, K=8.0):
    """the kernel's fast path + its certificates.  K = constant of the error bounds, in ulps (the derivation gives 3: DESIGN.md)"""
    cx, cy, cz = c
    active = ~(cz < r + znear)  # exact (one add, one compare)
    crx, cry, crz = cx * r, cy * r, cz * r
    czr2 = cz * cz - r * r
    kE = f32(K) * EPS
    with np.errstate(all="ignore"):
        def axis(ca, cra):
            va = perturb(np.sqrt((ca * ca + czr2).astype(np.float64)).astype(f32), rng)
            p, g = va * ca, va * cz
            nmin, nmax = p - crz, p + crz
            dmin, dmax = g + cra, g - cra
            rmin, rmax = perturb((1.0 / dmin.astype(np.float64)).astype(f32), rng), perturb((1.0 / dmax.astype(np.float64)).astype(f32), rng)
            qmin, qmax = nmin * rmin, nmax * rmax
            Nt, Dt = np.abs(p) + np.abs(crz), np.abs(g) + np.abs(cra)
            emin = (Nt + np.abs(qmin) * Dt) * np.abs(rmin) * kE + np.abs(qmin) * kE
            emax = (Nt + np.abs(qmax) * Dt) * np.abs(rmax) * kE + np.abs(qmax) * kE
            return qmin, qmax, emin, emax

        minx, maxx, eminx, emaxx = axis(cx, crx)
        miny, maxy, eminy, emaxy = axis(cy, cry)
        hP0, hP1 = np.abs(P00) * f32(0.5), np.abs(P11) * f32(0.5)
        A0 = minx * P00 * f32(0.5) + f32(0.5)
        A1 = maxy * P11 * f32(-0.5) + f32(0.5)
        A2 = maxx * P00 * f32(0.5) + f32(0.5)
        A3 = miny * P11 * f32(-0.5) + f32(0.5)
        E0 = eminx * hP0 + (np.abs(A0) + f32(0.5)) * kE
        E1 = emaxy * hP1 + (np.abs(A1) + f32(0.5)) * kE
        E2 = emaxx * hP0 + (np.abs(A2) + f32(0.5)) * kE
        E3 = eminy * hP1 + (np.abs(A3) + f32(0.5)) * kE
        sx, sy = A2 - A0, A3 - A1
        Esx, Esy = E0 + E2 + np.abs(sx) * kE, E1 + E3 + np.abs(sy) * kE
        mx, my = sx * pw, sy * ph
        m = np.maximum(mx, my)
        Em = np.maximum(Esx * pw + np.abs(mx) * kE, Esy * ph + np.abs(my) * kE)
        lo, hi = m - Em, m + Em
        ok = np.isfinite(m) & np.isfinite(Em)
        pos_lo, pos_hi = lo > f32(2.0 ** -100), hi > f32(2.0 ** -100)
        l_lo = np.where(pos_lo, ceil_log2_exact(np.where(pos_lo, lo, f32(1))), -1000)
        l_hi = np.where(pos_hi, ceil_log2_exact(np.where(pos_hi, hi, f32(1))), -1000)
        # certain level: both ends positive with the same ceil(log2), or both ends non-positive-ish (level 0 either way)
        ok &= (pos_lo & (l_lo == l_hi)) | (hi <= 0)
        lvl = np.clip(np.where(pos_lo, l_lo, 0), None, 32)
        has = lvl > 0
        scale = np.exp2((1 - np.where(has, lvl, 1)).astype(np.float64)).astype(f32)
        fx, fy = pw * scale, ph * scale

        def fit(A, EA, s, Es, f):
            u = A * f
            Eu = EA * f + np.abs(u) * kE
            fl = np.floor(u)
            fr = u - fl
            T = fr + s * f
            ET = Eu + Es * f + (np.abs(u) + np.abs(T) + f32(2)) * kE
            good = (fr > Eu) & (fr < f32(1) - Eu) & (np.abs(T - f32(2)) > ET)
            return T <= f32(2.0), good

        fit0, g0 = fit(A0, E0, sx, Esx, fx)
        fit1, g1 = fit(A1, E1, sy, Esy, fy)
        ok &= ~has | (g0 & g1)
        lvl = np.where(has, lvl - (fit0 & fit1), 0)
        l = np.clip(lvl, 0, levels - 1)
        w = np.maximum(1, width >> l).astype(np.int64)
        h = np.maximum(1, height >> l).astype(np.int64)
        u = (A0 + A2) * f32(0.5)
        v = (A1 + A3) * f32(0.5)
        Eu = (E0 + E2) * f32(0.5) + np.abs(u) * kE
        Ev = (E1 + E3) * f32(0.5) + np.abs(v) * kE

        def foot(uv, Euv, size):
            sf = size.astype(f32)
            t = uv * sf - f32(0.5)
            Et = Euv * sf + (np.abs(t) + f32(1)) * kE
            f0 = np.floor(t)
            fr = t - f0
            good = (fr > Et) & (fr < f32(1) - Et)
            f0 = np.where(f0 >= -1, f0, f32(-1))
            f0 = np.where(f0 > sf, sf, f0)
            a = f0.astype(np.int64)
            hi_ = size - 1
            return np.clip(a, 0, hi_), np.clip(a + 1, 0, hi_), good

        x0, x1, gx = foot(u, Eu, w)
        y0, y1, gy = foot(v, Ev, h)
        ok &= gx & gy
        rd = perturb((1.0 / (cz - r).astype(np.float64)).astype(f32), rng)
        depth = znear * rd
    return dict(active=active, certain=ok | ~active, level=l, x0=x0, x1=x1, y0=y0, y1=y1, depth=depth, depth_margin=np.abs(depth) * kE)


def fast_lean(c, r, znear, P00, P11, pw, ph, width, height, levels, rng, K=8.0):
    """the lean variant: ONE bound E_A for all four aabb coordinates instead of one per quantity.
        |q_fast - q_ref| <= 3 eps Q (2 + K'),   Q = (|c_a| + r c_z / v_a) / (c_z - r) >= |q|,   K' = (c_z + r) / (c_z - r)
    (D = v_a c_z +- c_a r >= v_a (c_z - r) because |c_a| <= v_a), E_A = 0.5 max|P| max_a E_q + rounding of the affine map."""
    cx, cy, cz = c
    active = ~(cz < r + znear)
    crz = cz * r
    czr2 = cz * cz - r * r
    kE = f32(K) * EPS
    with np.errstate(all="ignore"):
        rd = perturb((1.0 / (cz - r).astype(np.float64)).astype(f32), rng)
        Kp = (cz + r) * rd

        def axis(ca):
            cra = ca * r
            va = perturb(np.sqrt((ca * ca + czr2).astype(np.float64)).astype(f32), rng)
            rva = perturb((1.0 / va.astype(np.float64)).astype(f32), rng)
            p, g = va * ca, va * cz
            qmin = (p - crz) * perturb((1.0 / (g + cra).astype(np.float64)).astype(f32), rng)
            qmax = (p + crz) * perturb((1.0 / (g - cra).astype(np.float64)).astype(f32), rng)
            Q = (np.abs(ca) + crz * rva) * rd
            return qmin, qmax, Q

        minx, maxx, Qx = axis(cx)
        miny, maxy, Qy = axis(cy)
        Pm = np.maximum(np.abs(P00), np.abs(P11)) * f32(0.5)
        A0 = minx * P00 * f32(0.5) + f32(0.5)
        A1 = maxy * P11 * f32(-0.5) + f32(0.5)
        A2 = maxx * P00 * f32(0.5) + f32(0.5)
        A3 = miny * P11 * f32(-0.5) + f32(0.5)
        Amax = np.maximum(np.maximum(np.abs(A0), np.abs(A1)), np.maximum(np.abs(A2), np.abs(A3)))
        EA = (np.maximum(Qx, Qy) * (f32(2) + Kp) * Pm + Amax + f32(0.5)) * kE
        sx, sy = A2 - A0, A3 - A1
        mx, my = sx * pw, sy * ph
        m = np.maximum(mx, my)
        pmax = np.maximum(pw, ph)
        Em = f32(2) * EA * pmax + np.abs(m) * kE
        lo, hi = m - Em, m + Em
        pos_lo = lo > f32(2.0 ** -100)
        pos_hi = hi > f32(2.0 ** -100)
        l_lo = np.where(pos_lo, ceil_log2_exact(np.where(pos_lo, lo, f32(1))), -1000)
        l_hi = np.where(pos_hi, ceil_log2_exact(np.where(pos_hi, hi, f32(1))), -1000)
        ok = np.isfinite(m) & np.isfinite(Em) & ((pos_lo & (l_lo == l_hi)) | (hi <= 0))
        lvl = np.clip(np.where(pos_lo, l_lo, 0), None, 32)
        has = lvl > 0
        scale = np.exp2((1 - np.where(has, lvl, 1)).astype(np.float64)).astype(f32)
        fx, fy = pw * scale, ph * scale
        fmax = np.maximum(fx, fy)
        Ef = EA * fmax  # bound of any aabb coordinate in texels of the `fits` grid

        def fit(A, s, f):
            u = A * f
            fl = np.floor(u)
            fr = u - fl
            T = fr + s * f
            Eu = Ef + np.abs(u) * kE
            ET = f32(3) * Ef + (np.abs(u) + np.abs(T) + f32(2)) * kE
            good = (fr > Eu) & (fr < f32(1) - Eu) & (np.abs(T - f32(2)) > ET)
            return T <= f32(2.0), good

        fit0, g0 = fit(A0, sx, fx)
        fit1, g1 = fit(A1, sy, fy)
        ok &= ~has | (g0 & g1)
        lvl = np.where(has, lvl - (fit0 & fit1), 0)
        l = np.clip(lvl, 0, levels - 1)
        w = np.maximum(1, width >> l).astype(np.int64)
        h = np.maximum(1, height >> l).astype(np.int64)
        u = (A0 + A2) * f32(0.5)
        v = (A1 + A3) * f32(0.5)

        def foot(uv, size):
            sf = size.astype(f32)
            t = uv * sf - f32(0.5)
            Et = EA * sf + (np.abs(t) + f32(1)) * kE
            f0 = np.floor(t)
            fr = t - f0
            good = (fr > Et) & (fr < f32(1) - Et)
            f0 = np.where(f0 >= -1, f0, f32(-1))
            f0 = np.where(f0 > sf, sf, f0)
            a = f0.astype(np.int64)
            hi_ = size - 1
            return np.clip(a, 0, hi_), np.clip(a + 1, 0, hi_), good

        x0, x1, gx = foot(u, w)
        y0, y1, gy = foot(v, h)
        ok &= gx & gy
        depth = znear * rd
    return dict(active=active, certain=ok | ~active, level=l, x0=x0, x1=x1, y0=y0, y1=y1, depth=depth, depth_margin=np.abs(depth) * kE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--K", type=float, default=8.0)
    ap.add_argument("--lean", action="store_true", help="the one-bound variant")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    worst = 0.0
    for name, zr, rr, size in (("frame scene", (0.2, 200.0), (0.02, 1.0), 4096), ("tiny spheres far away", (20.0, 400.0), (0.001, 0.05), 4096),
                               ("huge spheres near the camera", (0.15, 5.0), (0.05, 3.0), 2048), ("small viewport", (0.2, 60.0), (0.01, 0.5), 256)):
        n = a.n
        znear = f32(0.1)
        f = f32(1.0 / np.tan(np.radians(70.0) / 2))
        P00, P11 = f32(f), f32(f)
        pw = ph = f32(size // 2)
        width = height = size // 2
        levels = int(np.log2(width)) + 1
        cz = np.exp(rng.uniform(np.log(zr[0]), np.log(zr[1]), n)).astype(f32)
        cx = (rng.uniform(-1.3, 1.3, n) * cz / P00).astype(f32)
        cy = (rng.uniform(-1.3, 1.3, n) * cz / P11).astype(f32)
        r = np.exp(rng.uniform(np.log(rr[0]), np.log(rr[1]), n)).astype(f32)
        ref = reference((cx, cy, cz), r, znear, P00, P11, pw, ph, width, height, levels)
        fa = (fast_lean if a.lean else fast)((cx, cy, cz), r, znear, P00, P11, pw, ph, width, height, levels, rng, a.K)
        act = ref["active"]
        assert (act == fa["active"]).all()
        cert = fa["certain"] & act
        bad = np.zeros(n, bool)
        for k in ("level", "x0", "x1", "y0", "y1"):
            bad |= cert & (ref[k] != fa[k])
        bad |= cert & (ref["use"] != 15)  # a certain probe asserts that all four texels carry weight
        # the verdict: certain only when |depth - texel| > margin for the texel value it meets; check the bound itself
        dd = np.abs(ref["depth"].astype(np.float64) - fa["depth"].astype(np.float64))
        bad |= act & (dd > fa["depth_margin"].astype(np.float64))
        worst = max(worst, float((dd[act] / np.maximum(1e-30, fa["depth_margin"][act].astype(np.float64))).max()))
        print("%-32s probes %8d  active %5.1f %%  fall back %6.3f %%  wrong certain decisions %d" %
              (name, n, 100.0 * act.mean(), 100.0 * (act & ~fa["certain"]).sum() / max(1, act.sum()), int(bad.sum())))
        assert not bad.any()
    print("largest |depth_fast - depth_ref| / margin: %.3f" % worst)


if __name__ == "__main__":
    main()
