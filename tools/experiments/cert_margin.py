#!/usr/bin/env python3
"""cert_margin.py — CPU check of the error margins behind clustercull.hip's conservative filter / certified test.

Emulates make_filter() and certified_visible() in numpy fp32 (FMA = fp64 product-sum rounded once more, which differs
from a true FMA by at most one ulp of the result) and compares against the reference arithmetic's own intermediates
(oracle.probe_cluster_scalars: view-space centre, radius, dot(c, axis), cutoff * |c| + r):

    |c~ - c_ref|_inf                 / E        (E = T / 4: the bound the frustum margins rest on)
    |D~ - (lhs_ref - rhs_ref)|       / (T coneK)

over random scenes of very different magnitudes, orientations (unit and non-unit quaternions), scales and cameras.
Every ratio must stay well below 1 (the analysis leaves > 2x slack); the script exits non-zero otherwise.
Needs no GPU.  Test infrastructure (imports the oracle)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from niagara_amd import host, synth  # noqa: E402

f32 = np.float32
U = f32(5.9604644775390625e-8)


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def make_filter(cd, draws, filterK):
    """numpy restatement of make_filter (clustercull.hip), one row per draw"""
    V = cd["view"][0].astype(f32)
    x, y, z, w = (draws["orientation"][:, i].astype(f32) for i in range(4))
    s = draws["scale"].astype(f32)
    p = draws["position"].astype(f32)
    two, one = f32(2), f32(1)
    R = [one - two * (y * y + z * z), two * (x * y - w * z), two * (x * z + w * y),
         two * (x * y + w * z), one - two * (x * x + z * z), two * (y * z - w * x),
         two * (x * z - w * y), two * (y * z + w * x), one - two * (x * x + y * y)]
    m = np.zeros((len(draws), 9), f32)
    b = np.zeros((len(draws), 3), f32)
    for r in range(3):
        for c in range(3):
            m[:, 3 * r + c] = s * ((V[r] * R[c] + V[4 + r] * R[3 + c]) + V[8 + r] * R[6 + c])
        b[:, r] = ((V[r] * p[:, 0] + V[4 + r] * p[:, 1]) + V[8 + r] * p[:, 2]) + V[12 + r]
    Qa = np.abs(x) + np.abs(y) + np.abs(z)
    rotAbs = one + two * Qa * (Qa + np.abs(w))
    Vn = max(abs(V[r]) + abs(V[4 + r]) + abs(V[8 + r]) for r in range(3))
    V3n = max(abs(V[12 + r]) for r in range(3))
    pn = np.max(np.abs(p), axis=1)
    alpha = Vn * np.abs(s) * rotAbs
    beta = Vn * pn + V3n
    aK = f32(filterK) * alpha
    bK = f32(filterK) * beta + f32(1e-30)
    aR = f32(9.5367431640625e-7) * np.abs(s)
    coneK = f32(2.02) * (Vn * rotAbs) + one
    is127 = (one / s) * f32(0.00787401574803149606)
    return m, b, aK, bK, aR, s, coneK, is127


def check(name, draws, meshlets, commands, cd):
    n = len(commands)
    probe = oracle.probe_cluster_scalars(cd, commands, draws, meshlets)  # (n, 64, 16)
    filterK = 4.0 * 48.0 * 5.9604644775390625e-8 * 1.001
    m, b, aK, bK, aR, s, coneK, is127 = make_filter(cd, draws, filterK)
    d = commands["drawId"][:n]
    mi = commands["taskOffset"][:n, None] + np.arange(64, dtype=np.uint32)[None, :]
    ml = meshlets[mi]
    v = ml["center"].view(np.float16).astype(f32)          # (n, 64, 3)
    rad = ml["radius"].view(np.float16).astype(f32)
    M = m[d][:, None, :]
    B = b[d][:, None, :]
    c = np.zeros(v.shape, f32)
    for r in range(3):
        c[..., r] = fma(M[..., 3 * r] + 0 * v[..., 0], v[..., 0], fma(M[..., 3 * r + 1] + 0 * v[..., 0], v[..., 1], fma(M[..., 3 * r + 2] + 0 * v[..., 0], v[..., 2], B[..., r] + 0 * v[..., 0])))
    T = fma(aK[d][:, None] + 0 * rad, np.abs(v[..., 0]), bK[d][:, None] + 0 * rad)
    T = fma(aK[d][:, None] + 0 * rad, np.abs(v[..., 1]), T)
    T = fma(aK[d][:, None] + 0 * rad, np.abs(v[..., 2]), T)
    T = fma(aR[d][:, None] + 0 * rad, np.abs(rad), T)
    c_ref = probe[..., 0:3]
    ok = np.isfinite(c_ref).all(axis=-1) & np.isfinite(T)
    err_c = np.max(np.abs(c.astype(np.float64) - c_ref.astype(np.float64)), axis=-1)
    ratio_c = np.where(ok, err_c / (T.astype(np.float64) / 4.0), 0.0)
    # cone
    k = ml["cone_axis"].astype(f32)
    kc = ml["cone_cutoff"].astype(f32)
    wv = np.zeros(v.shape, f32)
    for r in range(3):
        wv[..., r] = fma(M[..., 3 * r] + 0 * k[..., 0], k[..., 0], fma(M[..., 3 * r + 1] + 0 * k[..., 0], k[..., 1], (M[..., 3 * r + 2] * k[..., 2]).astype(f32)))
    lhs = fma(c[..., 0], wv[..., 0], fma(c[..., 1], wv[..., 1], (c[..., 2] * wv[..., 2]).astype(f32))) * is127[d][:, None]
    len2 = fma(c[..., 0], c[..., 0], fma(c[..., 1], c[..., 1], (c[..., 2] * c[..., 2]).astype(f32)))
    ln = np.sqrt(len2.astype(np.float64)).astype(f32)
    ln = np.nextafter(ln, f32(np.inf))  # v_sqrt_f32: up to 1 ulp off
    rhs = fma(kc * f32(0.00787401574803149606), ln, (s[d][:, None] * rad).astype(f32))
    D = (lhs - rhs).astype(f32)
    D_ref = probe[..., 4].astype(np.float64) - probe[..., 5].astype(np.float64)
    Tc = (T * coneK[d][:, None]).astype(np.float64)
    okc = ok & np.isfinite(D_ref) & np.isfinite(Tc)
    ratio_d = np.where(okc, np.abs(D.astype(np.float64) - D_ref) / Tc, 0.0)
    # decisions the certified test would take, against the reference's
    cull_ref = probe[..., 15] != 0
    wrong = okc & (((D > Tc) & ~cull_ref) | ((D < -Tc) & cull_ref))
    undec = okc & (np.abs(D) <= Tc)
    print("%-34s centre error / E: max %.3f   cone error / margin: max %.4f   undecided cone lanes %.3f %%   wrong %d" %
          (name, ratio_c.max(), ratio_d.max(), 100.0 * undec.mean(), int(wrong.sum())))
    return ratio_c.max(), ratio_d.max(), int(wrong.sum())


def main():
    rng = np.random.default_rng(11)
    worst_c = worst_d = 0.0
    wrong = 0
    cases = [("config 3A geometry (radius 300)", 300.0, 1.0, 1.0, (0, 0, 0), (0, 0, 0, 1)),
             ("dense (radius 40)", 40.0, 1.0, 1.0, (0, 0, 0), (0, 0, 0, 1)),
             ("tiny scene 1e-3", 1e-3, 1e-3, 1.0, (0, 0, 0), (0, 0, 0, 1)),
             ("huge scene 3e5", 3e5, 100.0, 1.0, (1e4, -2e4, 3e3), (0.3, -0.5, 0.2, 0.78)),
             ("non-unit quaternions x7.5", 100.0, 1.0, 7.5, (5, 6, 7), (0.1, 0.7, -0.1, 0.69)),
             ("camera far from origin", 300.0, 1.0, 1.0, (5e3, 5e3, -5e3), (0.5, 0.5, 0.5, 0.5)),
             ("small quaternions x0.01", 50.0, 2.0, 0.01, (0, 0, 0), (0, 0, 0, 1))]
    for name, radius, scale_mul, qmul, cam, camq in cases:
        n_draws, cpd = 1500, 2
        draws = host.synth_draws(n_draws, 1, radius)
        draws["scale"] *= f32(scale_mul)
        draws["orientation"] *= f32(qmul)
        q = np.asarray(camq, np.float64)
        q = q / np.linalg.norm(q)
        cd = host.build_cull_data(cam_pos=cam, cam_quat=tuple(q), draw_count=n_draws, cullingEnabled=1, clusterBackfaceEnabled=1,
                                  draw_distance=max(200.0, radius))
        commands = synth.make_task_commands(n_draws, cpd)
        meshlets = synth.make_meshlets(n_draws * cpd * 64, seed=int(rng.integers(1 << 30)))
        # all int8 values incl. -128, extreme cutoffs
        meshlets["cone_axis"][::7] = rng.integers(-128, 128, (len(meshlets[::7]), 3)).astype(np.int8)
        meshlets["cone_cutoff"][::5] = rng.integers(-128, 128, len(meshlets[::5])).astype(np.int8)
        a, b, w = check(name, draws, meshlets, commands[:n_draws * cpd], cd)
        worst_c, worst_d, wrong = max(worst_c, a), max(worst_d, b), wrong + w
    print("worst: centre %.3f of E, cone %.4f of its margin, wrong decisions %d" % (worst_c, worst_d, wrong))
    return 0 if worst_c < 0.5 and worst_d < 0.5 and wrong == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
