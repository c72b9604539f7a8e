// cullmath.h — device-side cull arithmetic for gfx950, one IEEE fp32 operation per source operation.
//
// Build with -ffp-contract=off: a visibility decision must be the same bit pattern the CPU reference of the
// same math produces, so no FMA contraction, IEEE sqrt/divide (hipcc default), and the operation order of the
// reference shaders is kept as written:
//   rotateQuat / coneCull / projectSphere / getOcclusionMip  src/shaders/math.h:2-49
//   sphere transform + frustum test                          src/shaders/drawcull.comp.glsl:73-84,
//                                                            src/shaders/clustercull.comp.glsl:72-108
//   HiZ test                                                 src/shaders/drawcull.comp.glsl:86-99
// GLSL leaves mat*vec association, dot/length summation order and log2/exp2 accuracy open; they are fixed here
// as ((c0*x + c1*y) + c2*z) + c3, x->y->z, and exact exponent arithmetic (DESIGN.md "Defined semantics").
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/niagara_vis.h"

#define NV_DEV __device__ __forceinline__

namespace nv
{

struct f3
{
	float x, y, z;
};

// GLSL: min(x,y) = y<x ? y : x ; max(x,y) = x<y ? y : x (NaN behaviour follows from the comparison)
NV_DEV float gl_min(float x, float y) { return y < x ? y : x; }
NV_DEV float gl_max(float x, float y) { return x < y ? y : x; }

// cross(a,b) per the GLSL spec: (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
NV_DEV f3 cross3(f3 a, f3 b)
{
	f3 o;
	o.x = a.y * b.z - b.y * a.z;
	o.y = a.z * b.x - b.z * a.x;
	o.z = a.x * b.y - b.x * a.y;
	return o;
}

NV_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
NV_DEV float length3(f3 a) { return __builtin_sqrtf(dot3(a, a)); }

// math.h:46-49  v + 2.0 * cross(q.xyz, cross(q.xyz, v) + q.w * v)
NV_DEV f3 rotate_quat(f3 v, f3 q, float qw)
{
	f3 t = cross3(q, v);
	t.x = t.x + qw * v.x;
	t.y = t.y + qw * v.y;
	t.z = t.z + qw * v.z;
	f3 u = cross3(q, t);
	f3 o;
	o.x = v.x + 2.0f * u.x;
	o.y = v.y + 2.0f * u.y;
	o.z = v.z + 2.0f * u.z;
	return o;
}

// (view * vec4(p,1)).xyz, view column-major
NV_DEV f3 view_point(const float* m, f3 p)
{
	f3 o;
	o.x = ((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12];
	o.y = ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13];
	o.z = ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14];
	return o;
}

// mat3(view) * v
NV_DEV f3 view_dir(const float* m, f3 v)
{
	f3 o;
	o.x = (m[0] * v.x + m[4] * v.y) + m[8] * v.z;
	o.y = (m[1] * v.x + m[5] * v.y) + m[9] * v.z;
	o.z = (m[2] * v.x + m[6] * v.y) + m[10] * v.z;
	return o;
}

// rotateQuat(c, q) * scale + position, then into view space (drawcull.comp.glsl:73-74)
NV_DEV f3 sphere_center(const NvCullData& cd, f3 local, f3 q, float qw, float scale, f3 pos)
{
	f3 r = rotate_quat(local, q, qw);
	f3 w;
	w.x = r.x * scale + pos.x;
	w.y = r.y * scale + pos.y;
	w.z = r.z * scale + pos.z;
	return view_point(cd.view, w);
}

// drawcull.comp.glsl:77-82 / clustercull.comp.glsl:103-108
NV_DEV bool frustum_test(const NvCullData& cd, f3 c, float r)
{
	// all four comparisons are evaluated (they are side-effect free), then ANDed: no exec-mask branching
	const bool sx = c.z * cd.frustum[1] - __builtin_fabsf(c.x) * cd.frustum[0] > -r;
	const bool sy = c.z * cd.frustum[3] - __builtin_fabsf(c.y) * cd.frustum[2] > -r;
	const bool zn = c.z + r > cd.znear;
	const bool zf = c.z - r < cd.zfar;
	return (sx & sy) & (zn & zf);
}

// math.h:41-44 with camera_position = 0
NV_DEV bool cone_cull(f3 c, float r, f3 axis, float cutoff)
{
	return dot3(c, axis) >= cutoff * length3(c) + r;
}

// math.h:2-22
NV_DEV bool project_sphere(f3 c, float r, float znear, float P00, float P11, float aabb[4])
{
	if (c.z < r + znear)
		return false;

	float crx = c.x * r, cry = c.y * r, crz = c.z * r;
	float czr2 = c.z * c.z - r * r;

	float vx = __builtin_sqrtf(c.x * c.x + czr2);
	float minx = (vx * c.x - crz) / (vx * c.z + crx);
	float maxx = (vx * c.x + crz) / (vx * c.z - crx);

	float vy = __builtin_sqrtf(c.y * c.y + czr2);
	float miny = (vy * c.y - crz) / (vy * c.z + cry);
	float maxy = (vy * c.y + crz) / (vy * c.z - cry);

	float ax = minx * P00, ay = miny * P11, az = maxx * P00, aw = maxy * P11;
	aabb[0] = ax * 0.5f + 0.5f;
	aabb[1] = aw * -0.5f + 0.5f;
	aabb[2] = az * 0.5f + 0.5f;
	aabb[3] = ay * -0.5f + 0.5f;
	return true;
}

// exact ceil(log2(x)) for x > 0 from the exponent bits; +inf -> 129
NV_DEV int ceil_log2_exact(float x)
{
	uint32_t u = __float_as_uint(x);
	uint32_t e = (u >> 23) & 0xffu;
	uint32_t m = u & 0x7fffffu;
	if (e == 0)
	{
		int hb = 31 - __builtin_clz(m);
		bool pow2 = (m & (m - 1)) == 0;
		return (hb - 149) + (pow2 ? 0 : 1);
	}
	if (e == 255)
		return 129;
	return (int)e - 127 + (m != 0 ? 1 : 0);
}

NV_DEV float fract1(float x) { return x - __builtin_floorf(x); }

// math.h:24-39; integer-valued result in [0, 32]
NV_DEV float occlusion_mip(const float aabb[4], float pw, float ph)
{
	float sx = aabb[2] - aabb[0];
	float sy = aabb[3] - aabb[1];
	float m = gl_max(sx * pw, sy * ph);

	if (!(m > 0.0f))
		return 0.0f;

	int level = ceil_log2_exact(m);
	if (level <= 0)
		return 0.0f;
	if (level > 32)
		level = 32;

	float scale = __uint_as_float((uint32_t)(127 + 1 - level) << 23); // exp2(1 - level), exact
	float fx = pw * scale, fy = ph * scale;
	bool fits = (fract1(aabb[0] * fx) + sx * fx <= 2.0f) && (fract1(aabb[1] * fy) + sy * fy <= 2.0f);
	level -= fits ? 1 : 0;

	return (float)level;
}

// one axis of the bilinear footprint: clamped texel indices + which of the two carry non-zero weight
NV_DEV void footprint(float t, uint32_t size, int& i0, int& i1, bool& u0, bool& u1)
{
	float f0 = __builtin_floorf(t);
	float fr = t - f0;
	float lim = (float)size;
	if (!(f0 >= -1.0f))
		f0 = -1.0f;
	if (f0 > lim)
		f0 = lim;
	int a = (int)f0, b = a + 1;
	int hi = (int)size - 1;
	i0 = a < 0 ? 0 : (a > hi ? hi : a);
	i1 = b < 0 ? 0 : (b > hi ? hi : b);
	u0 = (1.0f - fr) != 0.0f;
	u1 = fr != 0.0f;
}

// texture()/textureLod() on one mip: LINEAR, CLAMP_TO_EDGE, MIN reduction (src/niagara.cpp:629)
NV_DEV float sample_min_image(const float* __restrict__ img, uint32_t w, uint32_t h, float u, float v)
{
	int x0, x1, y0, y1;
	bool ux0, ux1, uy0, uy1;
	footprint(u * (float)w - 0.5f, w, x0, x1, ux0, ux1);
	footprint(v * (float)h - 0.5f, h, y0, y1, uy0, uy1);

	// the four texels are always loaded (indices are clamped, so the loads are in range); unused ones are
	// dropped from the min. Order of the min chain matches the oracle: (x0,y0) (x1,y0) (x0,y1) (x1,y1).
	float t00 = img[(size_t)y0 * w + x0];
	float t10 = img[(size_t)y0 * w + x1];
	float t01 = img[(size_t)y1 * w + x0];
	float t11 = img[(size_t)y1 * w + x1];

	float best = 0.0f;
	bool have = false;
	if (ux0 && uy0)
	{
		best = t00;
		have = true;
	}
	if (ux1 && uy0)
	{
		best = have ? gl_min(best, t10) : t10;
		have = true;
	}
	if (ux0 && uy1)
	{
		best = have ? gl_min(best, t01) : t01;
		have = true;
	}
	if (ux1 && uy1)
	{
		best = have ? gl_min(best, t11) : t11;
		have = true;
	}
	return best;
}

NV_DEV uint32_t mip_dim(uint32_t d, uint32_t level)
{
	uint32_t r = d >> level;
	return r ? r : 1u;
}

NV_DEV float sample_min(const NvPyramidDesc& p, float u, float v, float level)
{
	int l = (int)level;
	int top = (int)p.levels - 1;
	l = l < 0 ? 0 : (l > top ? top : l);
	return sample_min_image(p.d_base + p.mipOffset[l], mip_dim(p.width, (uint32_t)l), mip_dim(p.height, (uint32_t)l), u, v);
}

// The HiZ test in two halves, so that a caller can put other work between requesting the four texels and comparing them
// (clustercull.hip's occlusion stage, drawcull.hip's compacted probes).  hiz_prepare: projection, mip selection and the footprint's
// texel offsets relative to the pyramid base; hiz_finish: MIN over the texels that carry weight, in the oracle's order
// (x0,y0) (x1,y0) (x0,y1) (x1,y1), and the comparison.  Offsets are always in range (clamped), also for an inactive probe.
//
// Round 4: the stage that runs these is bound by VALU issue (6.9 M probes x 348 instructions at frame scale), and
// 140 of the 348 were not the reference's arithmetic but the selection logic around it.  Three restatements of that logic, none of
// which touches a floating-point operation of the reference:
//   * a texel WITHOUT weight is not flagged and skipped in hiz_finish — its offset is replaced by the offset of the texel of the
//     same row / column that has weight (an axis always has one: fr cannot be 0 and 1 at once; NaN gives both).  The MIN chain
//     `best = t < best ? t : best` over a sequence with such repeats equals the chain over the distinct weighted texels in the
//     oracle's order: the first element is the same (so a NaN first texel still poisons the chain, a later NaN is still skipped),
//     a repeat never wins `t < best`, and among equal values (-0 / +0) the first occurrence is kept in both;
//   * ceil(log2(m)) from v_frexp_exp / v_frexp_mant (m = f 2^e, f in [0.5, 1): e - (f == 0.5)) instead of field extraction with a
//     branch per class — same integer for every positive finite m including denormals; +inf and everything above 2^32 clamp to 32
//     like before, non-positive and NaN sizes give level 0 like before;
//   * the footprint's index clamps with integer min / max on the converted floor (the conversion saturates) instead of compares and
//     selects on the float.
struct HizProbe
{
	uint32_t o00, o10, o01, o11; // texel offsets (floats) from pyr.d_base: the weighted texels in the oracle's order, weightless ones aliased
	uint32_t use;                // bit 4: the probe is active (projectSphere succeeded); bits 0..3 unused since round 4
	float depthSphere;
};

// math.h:24-39 as occlusion_mip above, integer result in [0, 32], branch-free
NV_DEV int occlusion_level(const float aabb[4], float pw, float ph)
{
	const float sx = aabb[2] - aabb[0];
	const float sy = aabb[3] - aabb[1];
	const float m = gl_max(sx * pw, sy * ph);
	// ceil_log2_exact(m) for 0 < m < inf; frexp of NaN / inf / 0 gives exponent 0
	int level = __builtin_amdgcn_frexp_expf(m) - (__builtin_amdgcn_frexp_mantf(m) == 0.5f ? 1 : 0);
	level = m > 4294967296.0f ? 32 : level; // (level > 32 -> 32; +inf -> 129 -> 32)
	level = m > 0.0f ? level : 0;            // (!(m > 0) -> 0: negative sizes have a positive exponent)
	level = level < 0 ? 0 : level;           // (level <= 0 -> 0)
	const float scale = __uint_as_float((uint32_t)(127 + 1 - level) << 23); // exp2(1 - level), exact (level 0: 2.0, result unused)
	const float fx = pw * scale, fy = ph * scale;
	const bool fits = (fract1(aabb[0] * fx) + sx * fx <= 2.0f) && (fract1(aabb[1] * fy) + sy * fy <= 2.0f);
	return level - (fits && level > 0 ? 1 : 0);
}

// one axis of the bilinear footprint as footprint() above, the two indices already aliased: a0 / a1 = the index of the first /
// second texel of the axis if it has weight, else the other one's
NV_DEV void footprint_aliased(float t, uint32_t size, uint32_t& a0, uint32_t& a1)
{
	float f0 = __builtin_floorf(t);
	const float fr = t - f0;
	f0 = __builtin_amdgcn_fmed3f(f0, -1.0f, 16777216.0f); // !(f0 >= -1) -> -1 (a NaN too: v_med3_f32 then returns the MIN3); the upper bound
	                                                      // only keeps the conversion in range (size <= 2^24: the index clamp below decides)
	const int hi = (int)size - 1;
	int a = (int)f0;
	a = a < hi ? a : hi;             // -1 <= a <= hi
	const int b = a + 1;
	const int i0 = a < 0 ? 0 : a;
	const int i1 = b < hi ? b : hi;
	const bool u0 = (1.0f - fr) != 0.0f, u1 = fr != 0.0f;
	a0 = (uint32_t)(u0 ? i0 : i1);
	a1 = (uint32_t)(u1 ? i1 : i0);
}

// What a probe needs to know about its mip level, as a table a kernel keeps in LDS (two 16-byte reads per probe instead of the
// shift / max / convert / decrement sequence per axis: 7 VALU instructions of a 293-instruction probe)
struct MipRecord
{
	float wf, hf;      // (float)w, (float)h
	uint32_t w, base;  // row pitch, first texel of the level (from pyr.d_base)
	uint32_t hx, hy;   // w - 1, h - 1
	uint32_t pad0, pad1;
};

NV_DEV MipRecord make_mip_record(const NvPyramidDesc& pyr, uint32_t level, uint32_t base)
{
	MipRecord m;
	m.w = mip_dim(pyr.width, level);
	const uint32_t h = mip_dim(pyr.height, level);
	m.wf = (float)m.w;
	m.hf = (float)h;
	m.base = base;
	m.hx = m.w - 1u;
	m.hy = h - 1u;
	m.pad0 = m.pad1 = 0u;
	return m;
}

// footprint_aliased with the level's float size and last index given
NV_DEV void footprint_aliased_rec(float t, int hi, uint32_t& a0, uint32_t& a1)
{
	float f0 = __builtin_floorf(t);
	const float fr = t - f0;
	f0 = __builtin_amdgcn_fmed3f(f0, -1.0f, 16777216.0f);
	int a = (int)f0;
	a = a < hi ? a : hi;
	const int b = a + 1;
	const int i0 = a < 0 ? 0 : a;
	const int i1 = b < hi ? b : hi;
	const bool u0 = (1.0f - fr) != 0.0f, u1 = fr != 0.0f;
	a0 = (uint32_t)(u0 ? i0 : i1);
	a1 = (uint32_t)(u1 ? i1 : i0);
}

// mipOffsets = pyr.mipOffset, or a copy of it in LDS: indexed per lane, the kernel-argument array costs a vector load
// (and a full memory latency) per probe.  RECORDS: `table` is a MipRecord[levels] in LDS instead (clustercull.hip's occlusion stage).
template <bool RECORDS = false>
NV_DEV HizProbe hiz_prepare(const NvCullData& cd, const NvPyramidDesc& pyr, f3 c, float r, const void* table)
{
	HizProbe p = { 0, 0, 0, 0, 0, 0.0f };
	float aabb[4];
	if (project_sphere(c, r, cd.znear, cd.P00, cd.P11, aabb))
	{
		int l = occlusion_level(aabb, cd.pyramidWidth, cd.pyramidHeight);
		const int top = (int)pyr.levels - 1;
		l = l > top ? top : l; // (l >= 0)
		const float u = (aabb[0] + aabb[2]) * 0.5f, v = (aabb[1] + aabb[3]) * 0.5f;
		uint32_t x0, x1, y0, y1, w, base;
		if (RECORDS)
		{
			const uint4* rec = reinterpret_cast<const uint4*>(static_cast<const MipRecord*>(table) + l);
			const uint4 r0 = rec[0], r1 = rec[1];
			w = r0.z;
			base = r0.w;
			footprint_aliased_rec(u * __uint_as_float(r0.x) - 0.5f, (int)r1.x, x0, x1);
			footprint_aliased_rec(v * __uint_as_float(r0.y) - 0.5f, (int)r1.y, y0, y1);
		}
		else
		{
			w = mip_dim(pyr.width, (uint32_t)l);
			const uint32_t h = mip_dim(pyr.height, (uint32_t)l);
			footprint_aliased(u * (float)w - 0.5f, w, x0, x1);
			footprint_aliased(v * (float)h - 0.5f, h, y0, y1);
			base = static_cast<const uint32_t*>(table)[l];
		}
		// rows and widths are below 2^24 (a mip chain addressed with 32-bit texel offsets): the 24-bit multiply-add is exact,
		// and unlike the 64-bit form hipcc otherwise picks it reads no register pair (tools/check_asm_hazards.py, check 2)
		const uint32_t row0 = __umul24(y0, w) + base, row1 = __umul24(y1, w) + base;
		p.o00 = row0 + x0;
		p.o10 = row0 + x1;
		p.o01 = row1 + x0;
		p.o11 = row1 + x1;
		p.use = 16u;
		p.depthSphere = cd.znear / (c.z - r);
	}
	return p;
}

NV_DEV bool hiz_finish(const HizProbe& p, float t00, float t10, float t01, float t11)
{
	float best = t00;
	best = gl_min(best, t10);
	best = gl_min(best, t01);
	best = gl_min(best, t11);
	return !(p.use & 16u) || p.depthSphere > best;
}

// drawcull.comp.glsl:86-99 / clustercull.comp.glsl:110-123: returns the sphere's visibility against the pyramid
// TABLE: mipOffsets is a copy of pyr.mipOffset in LDS (see hiz_prepare); otherwise the descriptor's own array is indexed.
// A compile-time choice: selecting between the two pointers at run time would make the access a flat load and put the
// kernel-argument array in scratch.
template <bool TABLE = false>
NV_DEV bool hiz_test(const NvCullData& cd, const NvPyramidDesc& pyr, f3 c, float r, const uint32_t* mipOffsets = nullptr)
{
	const HizProbe p = TABLE ? hiz_prepare<false>(cd, pyr, c, r, mipOffsets) : hiz_prepare<false>(cd, pyr, c, r, pyr.mipOffset);
	if (!(p.use & 16u))
		return true;
	const float* base = pyr.d_base;
	return hiz_finish(p, base[p.o00], base[p.o10], base[p.o01], base[p.o11]);
}

// fp16 bits -> fp32 (exact)
NV_DEV float half_bits_to_float(uint32_t h)
{
	_Float16 v;
	unsigned short b = (unsigned short)h;
	__builtin_memcpy(&v, &b, 2);
	return (float)v;
}

} // namespace nv
