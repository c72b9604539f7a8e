// filtermath.h — the per-draw derivation behind clustercull.hip's conservative frustum filter and certified two-sided test, and the
// constants its soundness rests on, in ONE place that compiles for the device (hipcc) and for the host (g++).
//
// Why a header of its own: the margins K = 48, the 1.001 slack, aR = 2^-20, coneK = 2.02 ||V|| rot + 1 are what makes a
// "certain" decision certain.  tests/test_cert_margins.py compiles THIS file with g++ (tests/cert_shim.cpp), feeds it random and adversarial
// draws and holds the resulting margins against the reference arithmetic (the oracle's intermediates) in exact rational arithmetic — so a
// changed constant fails a CPU test instead of waiting for a GPU soak.  The kernels (clustercull.hip make_filter) and the host
// (context.hip fill_cluster_args) call the same functions; -ffp-contract=off on both sides: one IEEE fp32 operation per source operation.
//
// The analysis itself is written above make_filter / certified_visible in clustercull.hip.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define NV_FM __host__ __device__ __forceinline__
#else
#define NV_FM static inline
#endif

namespace nv
{

// ---- the constants
constexpr float FILTER_K = 48.0f;                          // K of E = K u (alpha max|v_i| + beta): the chains are <= 13 + 8 roundings deep, > 2x slack
constexpr float FILTER_U = 5.9604644775390625e-8f;         // u = 2^-24
constexpr float FILTER_SLACK = 1.001f;                     // rounds 4 K u S up
constexpr float FILTER_FLOOR = 1e-30f;                     // absolute floor of the margin's constant part
constexpr float FILTER_RADIUS_U = 9.5367431640625e-7f;     // aR = 2^-20 |scale|: the roundings of radius * scale and of the sums with r
constexpr float FILTER_MAGNITUDE_MAX = 1e12f;              // alpha, beta beyond this: nothing is certain for the draw
constexpr float FILTER_SCALE_MIN = 1e-15f;                 // |scale| below this: likewise
constexpr float FILTER_COEFF_MAX = 1e3f;                   // |f0| + |f1|, |f2| + |f3| beyond this (or non-finite): filter and certified test off
constexpr float FILTER_PLANE_MAX = 1e30f;                  // |znear|, |zfar| beyond this (or non-finite): likewise
constexpr float CONE_K_SLOPE = 2.02f, CONE_K_OFFSET = 1.0f; // coneK = 2.02 ||V|| (1 + 2 Qa (Qa + |qw|)) + 1: T coneK >= 2 E_cone
constexpr float INV_127 = 0.00787401574803149606f;         // RN(1 / 127)

struct FilterDraw
{
	float m[9];  // M = scale V R, row-major
	float b[3];  // V p + V3
	float aK, bK; // 4 K u S alpha, 4 K u S beta (+ the absolute floor; inf / NaN when nothing is certain)
	float aR;     // 2^-20 |scale|
	float tK;     // bK + aK 3 Vmax + aR Rmax >= T of every meshlet of the pool: the FILTER's margin, one value per draw (see filter_make)
	float scale;
	float coneK;  // the certified cone test's margin is T * coneK
	float is127;  // 1 / (127 scale): takes M = scale V R back to V R and the int8 axis to [-1, 1] in one factor
};

// filterK = 4 K u S (rounded up), S = max(1, |f0| + |f1|, |f2| + |f3|) of the frustum coefficients: the margins assume |f| <= 1; other
// finite coefficients scale them, non-finite or absurd ones (or near / far planes that are not finite) give 0 = filter and certified test off
NV_FM float filter_k(const float frustum[4], float znear, float zfar)
{
	const float s01 = __builtin_fabsf(frustum[0]) + __builtin_fabsf(frustum[1]), s23 = __builtin_fabsf(frustum[2]) + __builtin_fabsf(frustum[3]);
	float S = 1.0f;
	S = s01 > S ? s01 : S;
	S = s23 > S ? s23 : S;
	const bool finite = s01 <= FILTER_COEFF_MAX && s23 <= FILTER_COEFF_MAX && __builtin_fabsf(znear) <= FILTER_PLANE_MAX && __builtin_fabsf(zfar) <= FILTER_PLANE_MAX; // false on NaN
	return finite ? 4.0f * FILTER_K * FILTER_U * FILTER_SLACK * S : 0.0f;
}

// The early pass's filter loop evaluates a side plane's distance with the plane's coefficient folded into the row (clustercull.hip certainly_outside<FOLD>):
// cz f1 - |f0 cx| = cz f1 - |f0| |cx|, where the reference (clustercull.comp.glsl:104-105) computes cz f1 - |cx| f0.  The two agree only for f0, f2 >= 0; for
// a negative coefficient (a mirrored / flipped projection) the folded distance is too small by 2 |f0| |cx| and the filter would reject what the reference
// keeps.  The folded variants therefore run without the filter unless this holds; false on NaN.
NV_FM bool filter_fold_sound(const float frustum[4]) { return frustum[0] >= 0.0f && frustum[2] >= 0.0f; }

// the view-only terms of filter_make: Vn = max over rows r of |V(r,0)| + |V(r,1)| + |V(r,2)|, V3n = max_r |V(r,3)|, sumV = the sum of the
// twelve entries (0 x it is 0, or NaN for a non-finite view).  V column-major: V(r,k) = V[4k + r].
NV_FM void filter_view_norms(const float* V, float* Vn_, float* V3n_, float* sumV_)
{
	float Vn = 0.0f, V3n = 0.0f;
	for (int r = 0; r < 3; ++r)
	{
		Vn = __builtin_fmaxf(Vn, __builtin_fabsf(V[r]) + __builtin_fabsf(V[4 + r]) + __builtin_fabsf(V[8 + r]));
		V3n = __builtin_fmaxf(V3n, __builtin_fabsf(V[12 + r]));
	}
	float sumV = 0.0f;
	for (int i = 0; i < 15; ++i)
		sumV += (i & 3) == 3 ? 0.0f : V[i];
	*Vn_ = Vn;
	*V3n_ = V3n;
	*sumV_ = sumV;
}

// one draw's filter: q = (x, y, z, w), s = scale, p = position; (Vn, V3n, sumV) = filter_view_norms(V).
// vmax3 >= |vx| + |vy| + |vz| and rmax >= |radius| for EVERY meshlet the pass can touch (3 x the largest |centre component| and the largest
// |radius| of the registered pool, found once by nv_upload_meshlets; inf / NaN when the pool holds a non-finite record): the filter pass uses
// tK = bK + aK vmax3 + aR rmax >= T = bK + aK (|vx| + |vy| + |vz|) + aR |radius| as its margin — per draw instead of per meshlet, four
// FMAs less per meshlet in the stream (round 5).  tK only ever ENLARGES the margin (by < 4 % where beta dominates alpha, the usual
// case: positions are much larger than a mesh's own extent), so every "certainly outside" stays certain; the three fp32 roundings of its
// evaluation are relative 2^-24 against the analysis' > 2x slack.  The certified test (pass B) keeps the per-meshlet T.
// 1 / (127 scale), the one IEEE division of the derivation (~12 instructions): only the certified cone test reads it, so the cull kernel derives it
// when a segment has a candidate at all (filter_make<false> + filter_is127 in front of pass B) instead of on every wave's start-up path.
NV_FM float filter_is127(float s) { return (1.0f / s) * INV_127; }

template <bool WITH_IS127 = true>
NV_FM FilterDraw filter_make(const float* V, float x, float y, float z, float w, float s, float px, float py, float pz, float filterK, float Vn, float V3n, float sumV,
                             float vmax3, float rmax)
{
	// R = (1 - 2|q_xyz|^2) I + 2 q q^T + 2 w [q]x  (valid for any q, unit or not) — same map as rotateQuat.
	// Written with explicit fused multiply-adds (round 5): the derivation runs once per wave in front of its first filter result, on the launch's
	// start-up path, where an instruction costs what 1 / 25 of an instruction per command costs in the stream; every FMA is one rounding where
	// the un-fused pair had two, so the approximation's chains only get shorter than the <= 8 roundings the analysis allows them.
	const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
	float R[9];
	R[0] = __builtin_fmaf(-2.0f, yy + zz, 1.0f);
	R[1] = 2.0f * __builtin_fmaf(-w, z, xy);
	R[2] = 2.0f * __builtin_fmaf(w, y, xz);
	R[3] = 2.0f * __builtin_fmaf(w, z, xy);
	R[4] = __builtin_fmaf(-2.0f, xx + zz, 1.0f);
	R[5] = 2.0f * __builtin_fmaf(-w, x, yz);
	R[6] = 2.0f * __builtin_fmaf(-w, y, xz);
	R[7] = 2.0f * __builtin_fmaf(w, x, yz);
	R[8] = __builtin_fmaf(-2.0f, xx + yy, 1.0f);
	FilterDraw f;
#pragma unroll
	for (int r = 0; r < 3; ++r)
	{
#pragma unroll
		for (int c = 0; c < 3; ++c)
			f.m[3 * r + c] = s * __builtin_fmaf(V[8 + r], R[6 + c], __builtin_fmaf(V[4 + r], R[3 + c], V[r] * R[c]));
		f.b[r] = __builtin_fmaf(V[8 + r], pz, __builtin_fmaf(V[4 + r], py, __builtin_fmaf(V[r], px, V[12 + r])));
	}
	const float Qa = __builtin_fabsf(x) + __builtin_fabsf(y) + __builtin_fabsf(z);
	const float rotAbs = 1.0f + 2.0f * Qa * (Qa + __builtin_fabsf(w));
	const float pn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(px), __builtin_fabsf(py)), __builtin_fabsf(pz));
	const float alpha = Vn * __builtin_fabsf(s) * rotAbs;
	const float beta = Vn * pn + V3n;
	// The analysis assumes finite inputs and no overflow / harmful underflow in either evaluation.  fmaxf drops NaNs, so
	// non-finite draw or view fields are caught by a sum that is 0 or NaN, and magnitudes outside a generous range
	// (every intermediate of both chains, including the certified cone test's products, then stays far from the fp32
	// limits) make the margin infinite: nothing is certain for such a draw and the reference arithmetic decides.
	const float poison = 0.0f * ((((x + y) + (z + w)) + (s + ((px + py) + pz))) + sumV); // 0, or NaN
	const bool sane = alpha <= FILTER_MAGNITUDE_MAX && beta <= FILTER_MAGNITUDE_MAX && __builtin_fabsf(s) >= FILTER_SCALE_MIN;
	f.aK = filterK * alpha;
	f.bK = sane ? (filterK * beta + FILTER_FLOOR) + poison : __builtin_inff();
	f.aR = FILTER_RADIUS_U * __builtin_fabsf(s);
	f.tK = __builtin_fmaf(f.aR, rmax, __builtin_fmaf(f.aK, vmax3, f.bK));
	f.scale = s;
	f.coneK = CONE_K_SLOPE * (Vn * rotAbs) + CONE_K_OFFSET;
	f.is127 = WITH_IS127 ? filter_is127(s) : 0.0f;
	return f;
}

} // namespace nv
