// bounds.hip — meshlet bounds on the GPU: bounding sphere, normal cone, fp16 / s8 quantisation of
// every Meshlet from the scene's own vertex and meshlet-data buffers, so that real geometry can feed the visibility
// passes without meshoptimizer.
//
// Replaces the call site src/scene.cpp:69-85 (meshopt_computeMeshletBounds + meshopt_quantizeHalf + cone_axis_s8 /
// cone_cutoff_s8).  PARITY UNPINNED: the arithmetic is meshoptimizer's (an un-vendored submodule of the reference, no
// pinned commit); this file and the CPU oracle (orc_meshlet_bounds) restate the library's published algorithm —
// meshopt_computeClusterBounds with the three-axis Ritter sphere — with defined semantics (fp32, one IEEE operation per
// source operation, sums left to right, points in triangle order), and the two agree bit for bit.
//
// Mapping to CDNA4: one wavefront per meshlet (<= 64 vertices, <= 96 triangles).  Vertices, the compacted triangle list
// and its normals live in a 2.3 KiB LDS slice per wave.  The algorithm's sequential loops are kept exact without being
// run sequentially:
//   * extreme points: per-lane scan of its points in ascending order, then a wave reduction on (value, index) — equal
//     values resolve to the lower index, which is what the strict comparisons of the sequential scan keep;
//   * Ritter's growth pass "for every point: if outside, grow" only changes state at the points that ARE outside: all
//     lanes test their remaining points against the current sphere, the lowest violating index wins (wave min), the
//     sphere grows by exactly that point, and the search resumes behind it — the same sequence of updates as the loop,
//     in a handful of rounds instead of up to 288 dependent steps.
#include "cullmath.h"
#include "args.h"

namespace nv
{

constexpr int MB_WAVES = 4;
constexpr uint32_t MB_MAXVTX = 64, MB_MAXTRI = 96;

struct BoundsLds
{
	float pos[MB_MAXVTX][3];
	float nrm[MB_MAXTRI][3];
	uint32_t tri[MB_MAXTRI]; // corners a | b << 8 | c << 16 of the non-degenerate triangles, in order
};

NV_DEV uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
	{
		const uint32_t t = __shfl_xor(v, o, 64);
		v = t < v ? t : v;
	}
	return v;
}

// (value, index) reductions of the extreme-point search: LESS: smaller value wins, else larger; ties -> lower index
template <bool LESS>
NV_DEV void wave_extreme(float& v, uint32_t& idx)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
	{
		const float tv = __shfl_xor(v, o, 64);
		const uint32_t ti = __shfl_xor(idx, o, 64);
		const bool better = LESS ? (tv < v) : (tv > v);
		if (better || (tv == v && ti < idx))
		{
			v = tv;
			idx = ti;
		}
	}
}

// point p of a set: POINTS: corner (p % 3) of triangle (p / 3); else normal p
template <bool POINTS>
NV_DEV f3 fetch_point(const BoundsLds& s, uint32_t p)
{
	if (POINTS)
	{
		const uint32_t t = p / 3u, k = p - t * 3u;
		const uint32_t v = (s.tri[t] >> (8u * k)) & 63u;
		return f3{ s.pos[v][0], s.pos[v][1], s.pos[v][2] };
	}
	return f3{ s.nrm[p][0], s.nrm[p][1], s.nrm[p][2] };
}

// computeBoundingSphere (three-axis Ritter) over `count` >= 1 points; every lane returns the same sphere
template <bool POINTS>
NV_DEV void bounding_sphere(const BoundsLds& s, uint32_t count, uint32_t lane, f3& center, float& radius)
{
	constexpr uint32_t PER = POINTS ? (MB_MAXTRI * 3 + 63) / 64 : (MB_MAXTRI + 63) / 64; // points per lane: 5 / 2
	// ---- extreme points per axis (strict comparisons, first occurrence)
	float mn[3], mx[3];
	uint32_t imn[3], imx[3];
	{
		const f3 p0 = fetch_point<POINTS>(s, 0);
		mn[0] = mx[0] = p0.x, mn[1] = mx[1] = p0.y, mn[2] = mx[2] = p0.z;
#pragma unroll
		for (int a = 0; a < 3; ++a)
			imn[a] = imx[a] = 0;
	}
#pragma unroll
	for (uint32_t j = 0; j < PER; ++j)
	{
		const uint32_t p = j * 64u + lane;
		if (p < count)
		{
			const f3 q = fetch_point<POINTS>(s, p);
			const float c[3] = { q.x, q.y, q.z };
#pragma unroll
			for (int a = 0; a < 3; ++a)
			{
				if (c[a] < mn[a])
					mn[a] = c[a], imn[a] = p;
				if (c[a] > mx[a])
					mx[a] = c[a], imx[a] = p;
			}
		}
	}
#pragma unroll
	for (int a = 0; a < 3; ++a)
	{
		wave_extreme<true>(mn[a], imn[a]);
		wave_extreme<false>(mx[a], imx[a]);
	}
	// ---- the longest of the three extreme segments is the first diameter
	float paxisd2 = 0.0f;
	uint32_t i1 = imn[0], i2 = imx[0];
#pragma unroll
	for (int a = 0; a < 3; ++a)
	{
		const f3 p1 = fetch_point<POINTS>(s, imn[a]), p2 = fetch_point<POINTS>(s, imx[a]);
		const float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
		const float d2 = (dx * dx + dy * dy) + dz * dz;
		if (d2 > paxisd2)
		{
			paxisd2 = d2;
			i1 = imn[a], i2 = imx[a];
		}
	}
	{
		const f3 p1 = fetch_point<POINTS>(s, i1), p2 = fetch_point<POINTS>(s, i2);
		center = f3{ (p1.x + p2.x) / 2.0f, (p1.y + p2.y) / 2.0f, (p1.z + p2.z) / 2.0f };
		radius = __builtin_sqrtf(paxisd2) / 2.0f;
	}
	// ---- growth pass: the lowest point index outside the current sphere grows it; resume behind that point
	uint32_t from = 0;
	for (;;)
	{
		uint32_t first = ~0u;
#pragma unroll
		for (uint32_t j = 0; j < PER; ++j)
		{
			const uint32_t p = j * 64u + lane;
			if (p >= from && p < count && first == ~0u)
			{
				const f3 q = fetch_point<POINTS>(s, p);
				const float dx = q.x - center.x, dy = q.y - center.y, dz = q.z - center.z;
				const float d2 = (dx * dx + dy * dy) + dz * dz;
				if (d2 > radius * radius)
					first = p;
			}
		}
		first = wave_min_u32(first);
		if (first == ~0u)
			break;
		const f3 q = fetch_point<POINTS>(s, first);
		const float dx = q.x - center.x, dy = q.y - center.y, dz = q.z - center.z;
		const float d = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
		const float k = 0.5f + (radius / d) / 2.0f;
		center.x = center.x * k + q.x * (1.0f - k);
		center.y = center.y * k + q.y * (1.0f - k);
		center.z = center.z * k + q.z * (1.0f - k);
		radius = (radius + d) / 2.0f;
		from = first + 1u;
	}
}

// meshopt_quantizeHalf: round to nearest in the mantissa sum, flush below 2^-14, saturate to infinity, NaN -> qNaN
NV_DEV uint32_t quantize_half(float v)
{
	const uint32_t ui = __float_as_uint(v);
	const int s = (int)((ui >> 16) & 0x8000u);
	const int em = (int)(ui & 0x7fffffffu);
	int h = (em - (112 << 23) + (1 << 12)) >> 13;
	h = (em < (113 << 23)) ? 0 : h;
	h = (em >= (143 << 23)) ? 0x7c00 : h;
	h = (em > (255 << 23)) ? 0x7e00 : h;
	return (uint32_t)(s | h) & 0xffffu;
}

// meshopt_quantizeSnorm(v, 8)
NV_DEV int quantize_snorm8(float v)
{
	const float round = v >= 0.0f ? 0.5f : -0.5f;
	v = (v >= -1.0f) ? v : -1.0f;
	v = (v <= 1.0f) ? v : 1.0f;
	return (int)(v * 127.0f + round);
}

__global__ __launch_bounds__(MB_WAVES * 64) void meshlet_bounds_kernel(const NvVertex* __restrict__ vertices, const uint32_t* __restrict__ data,
                                                                      NvMeshlet* __restrict__ meshlets, uint32_t count, float* __restrict__ out8)
{
	__shared__ BoundsLds s_all[MB_WAVES];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	BoundsLds& s = s_all[wave];
	const uint16_t* data16 = reinterpret_cast<const uint16_t*>(data);
	const uint8_t* data8 = reinterpret_cast<const uint8_t*>(data);

	for (uint32_t mi = blockIdx.x * MB_WAVES + wave; mi < count; mi += gridDim.x * MB_WAVES)
	{
		const uint32_t* hdr = reinterpret_cast<const uint32_t*>(meshlets + mi);
		const uint32_t dataOffset = __builtin_amdgcn_readfirstlane(hdr[3]), baseVertex = __builtin_amdgcn_readfirstlane(hdr[4]);
		const uint32_t counts = __builtin_amdgcn_readfirstlane(hdr[5]);
		const uint32_t vcRaw = counts & 0xffu, tcRaw = (counts >> 8) & 0xffu;
		const uint32_t vertexCount = vcRaw < MB_MAXVTX ? vcRaw : MB_MAXVTX, triangleCount = tcRaw < MB_MAXTRI ? tcRaw : MB_MAXTRI;
		const bool shortRefs = ((counts >> 16) & 0xffu) == 1u;
		const uint32_t indexOffset = dataOffset + (shortRefs ? (vertexCount + 1u) / 2u : vertexCount);

		// ---- vertices (src/scene.cpp:193-198: fp16 positions, dequantised); unused slots are zero like the oracle's
		{
			float x = 0.0f, y = 0.0f, z = 0.0f;
			if (lane < vertexCount)
			{
				const uint32_t vi = (shortRefs ? (uint32_t)data16[dataOffset * 2u + lane] : data[dataOffset + lane]) + baseVertex;
				const uint2 v = *reinterpret_cast<const uint2*>(vertices + vi); // vx, vy | vz, tp
				x = half_bits_to_float(v.x & 0xffffu);
				y = half_bits_to_float(v.x >> 16);
				z = half_bits_to_float(v.y & 0xffffu);
			}
			s.pos[lane][0] = x, s.pos[lane][1] = y, s.pos[lane][2] = z;
		}
		// (one wave reads and writes its own LDS slice: program order is enough, no barrier)

		// ---- triangle normals, degenerate triangles dropped, order kept
		uint32_t triangles = 0;
#pragma unroll
		for (uint32_t j = 0; j < 2; ++j)
		{
			const uint32_t t = j * 64u + lane;
			bool keep = false;
			float nx = 0.0f, ny = 0.0f, nz = 0.0f, area = 1.0f;
			uint32_t packed = 0;
			if (t < triangleCount)
			{
				const uint32_t o = indexOffset * 4u + t * 3u;
				const uint32_t a = data8[o] & 63u, b = data8[o + 1] & 63u, c = data8[o + 2] & 63u;
				const float p10x = s.pos[b][0] - s.pos[a][0], p10y = s.pos[b][1] - s.pos[a][1], p10z = s.pos[b][2] - s.pos[a][2];
				const float p20x = s.pos[c][0] - s.pos[a][0], p20y = s.pos[c][1] - s.pos[a][1], p20z = s.pos[c][2] - s.pos[a][2];
				nx = p10y * p20z - p10z * p20y;
				ny = p10z * p20x - p10x * p20z;
				nz = p10x * p20y - p10y * p20x;
				area = __builtin_sqrtf((nx * nx + ny * ny) + nz * nz);
				keep = !(area == 0.0f);
				packed = a | (b << 8) | (c << 16);
			}
			const uint64_t m = __ballot(keep);
			if (keep)
			{
				const uint32_t idx = triangles + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
				s.nrm[idx][0] = nx / area, s.nrm[idx][1] = ny / area, s.nrm[idx][2] = nz / area;
				s.tri[idx] = packed;
			}
			triangles += (uint32_t)__builtin_popcountll(m);
		}

		f3 center = { 0.0f, 0.0f, 0.0f }, axis = { 0.0f, 0.0f, 0.0f };
		float radius = 0.0f, cutoff = 0.0f;
		int a8[3] = { 0, 0, 0 }, c8 = 0;
		if (triangles)
		{
			bounding_sphere<true>(s, triangles * 3u, lane, center, radius);
			float nr;
			bounding_sphere<false>(s, triangles, lane, axis, nr);
			const float axislength = __builtin_sqrtf((axis.x * axis.x + axis.y * axis.y) + axis.z * axis.z);
			const float inv = axislength == 0.0f ? 0.0f : 1.0f / axislength;
			axis.x *= inv, axis.y *= inv, axis.z *= inv;
			float mindp = 1.0f;
#pragma unroll
			for (uint32_t j = 0; j < 2; ++j)
			{
				const uint32_t t = j * 64u + lane;
				if (t < triangles)
				{
					const float dp = (s.nrm[t][0] * axis.x + s.nrm[t][1] * axis.y) + s.nrm[t][2] * axis.z;
					mindp = dp < mindp ? dp : mindp;
				}
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1)
			{
				const float t = __shfl_xor(mindp, o, 64);
				mindp = t < mindp ? t : mindp;
			}
			if (mindp <= 0.1f)
			{
				axis = f3{ 0.0f, 0.0f, 0.0f };
				cutoff = 1.0f;
				c8 = 127;
			}
			else
			{
				cutoff = __builtin_sqrtf(1.0f - mindp * mindp);
				const float ax[3] = { axis.x, axis.y, axis.z };
				float e = 0.0f;
#pragma unroll
				for (int k = 0; k < 3; ++k)
				{
					a8[k] = quantize_snorm8(ax[k]);
					e += __builtin_fabsf((float)a8[k] / 127.0f - ax[k]);
				}
				const int q = (int)(127.0f * (cutoff + e) + 1.0f);
				c8 = q > 127 ? 127 : q;
			}
		}
		if (lane == 0)
		{
			uint32_t* out = reinterpret_cast<uint32_t*>(meshlets + mi);
			out[0] = quantize_half(center.x) | (quantize_half(center.y) << 16);
			out[1] = quantize_half(center.z) | (quantize_half(radius) << 16);
			out[2] = ((uint32_t)a8[0] & 0xffu) | (((uint32_t)a8[1] & 0xffu) << 8) | (((uint32_t)a8[2] & 0xffu) << 16) | (((uint32_t)c8 & 0xffu) << 24);
			if (out8)
			{
				float* o = out8 + (size_t)mi * 8;
				o[0] = center.x, o[1] = center.y, o[2] = center.z, o[3] = radius, o[4] = axis.x, o[5] = axis.y, o[6] = axis.z, o[7] = cutoff;
			}
		}
	}
}

int launch_meshlet_bounds(hipStream_t stream, const NvVertex* vertices, const uint32_t* data, NvMeshlet* meshlets, uint32_t count, float* out8, uint32_t gridBlocks)
{
	if (count)
		hipLaunchKernelGGL(meshlet_bounds_kernel, dim3(gridBlocks), dim3(MB_WAVES * 64), 0, stream, vertices, data, meshlets, count, out8);
	return (int)hipGetLastError();
}

} // namespace nv
