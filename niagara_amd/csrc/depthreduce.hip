// depthreduce.hip — hierarchical depth (HiZ) pyramid build for gfx950.
//
// Replaces src/shaders/depthreduce.comp.glsl:14-22 and its per-level dispatch loop (src/niagara.cpp:1703-1733:
// one dispatch + full barrier per mip, 12 dependent launches for a 2048^2 pyramid).
//
// Every level is out(x,y) = MIN-reduction sample of the previous level at ((x,y)+0.5)/size.  For the pyramid's own
// power-of-two levels that footprint is exactly the 2x2 block {2x,2x+1}x{2y,2y+1} (clamped when a dimension has
// bottomed out at 1), and min is exact and associative, so the chain can be fused: one workgroup reads a 128x128
// source tile once (16-B loads, 64 KiB) and emits SEVEN levels — three from registers (each lane owns an 8x8 source
// patch), four more through a 1 KiB LDS tile.  A 4096^2 depth target becomes a 2048^2 x 12-level pyramid in two
// launches that read every source texel once: 4*W*H + (4/3)*4*pw*ph bytes instead of re-reading each mip.
// Level 0 from a depth target that is not exactly 2x the pyramid (e.g. 1024x768 -> 512x512) goes through the
// generic sampler kernel first.
#include "cullmath.h"

namespace nv
{

// ---- generic: one level through the sampler (any source size)
__global__ __launch_bounds__(256) void reduce_generic_kernel(const float* __restrict__ src, uint32_t sw, uint32_t sh,
                                                            float* __restrict__ dst, uint32_t lw, uint32_t lh)
{
	uint32_t x = blockIdx.x * 32 + (threadIdx.x & 31);
	uint32_t y = blockIdx.y * 8 + (threadIdx.x >> 5);
	if (x >= lw || y >= lh)
		return;
	float u = ((float)x + 0.5f) / (float)lw;
	float v = ((float)y + 0.5f) / (float)lh;
	dst[(size_t)y * lw + x] = sample_min_image(src, sw, sh, u, v);
}

struct ChainArgs
{
	const float* src;
	uint32_t sw, sh;
	float* base; // pyramid base
	uint32_t pw, ph;
	uint32_t firstLevel, numLevels; // numLevels <= 7
	uint32_t mipOffset[NV_MAX_MIPS];
};

// MIN over a 2 x 2 footprint in the sampler's defined order (x0,y0) (x1,y0) (x0,y1) (x1,y1), as a chain: min(x, y) =
// y < x ? y : x is neither commutative nor associative once NaN or -0 / +0 are among the texels, and the pyramid must
// be bit-identical for those too (tests/test_special_values.py)
NV_DEV float min4(float a, float b, float c, float d) { return gl_min(gl_min(gl_min(a, b), c), d); }

__global__ __launch_bounds__(256) void reduce_chain_kernel(ChainArgs a)
{
	__shared__ float s_l2[16][17];
	__shared__ float s_l3[8][9];
	__shared__ float s_l4[4][5];
	__shared__ float s_l5[2][3];

	const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
	const uint32_t L = a.firstLevel;

	// ---- 8x8 source patch -> registers
	const uint32_t sx = (blockIdx.x * 64 + tx * 4) * 2;
	const uint32_t sy = (blockIdx.y * 64 + ty * 4) * 2;
	float p[8][8];
	if (sx + 7 < a.sw && sy + 7 < a.sh)
	{
#pragma unroll
		for (int r = 0; r < 8; ++r)
		{
			const float4* row = reinterpret_cast<const float4*>(a.src + (size_t)(sy + r) * a.sw + sx);
			float4 lo = row[0], hi = row[1];
			p[r][0] = lo.x, p[r][1] = lo.y, p[r][2] = lo.z, p[r][3] = lo.w;
			p[r][4] = hi.x, p[r][5] = hi.y, p[r][6] = hi.z, p[r][7] = hi.w;
		}
	}
	else
	{
		// edge / tiny levels: clamp-to-edge reads (duplicates do not change a min)
#pragma unroll
		for (int r = 0; r < 8; ++r)
		{
			uint32_t yy = sy + r < a.sh ? sy + r : a.sh - 1;
#pragma unroll
			for (int c = 0; c < 8; ++c)
			{
				uint32_t xx = sx + c < a.sw ? sx + c : a.sw - 1;
				p[r][c] = a.src[(size_t)yy * a.sw + xx];
			}
		}
	}

	// ---- level L: 4x4 per lane
	float q[4][4];
#pragma unroll
	for (int r = 0; r < 4; ++r)
#pragma unroll
		for (int c = 0; c < 4; ++c)
			q[r][c] = min4(p[2 * r][2 * c], p[2 * r][2 * c + 1], p[2 * r + 1][2 * c], p[2 * r + 1][2 * c + 1]);
	{
		const uint32_t lw = mip_dim(a.pw, L), lh = mip_dim(a.ph, L);
		const uint32_t x0 = blockIdx.x * 64 + tx * 4, y0 = blockIdx.y * 64 + ty * 4;
		float* dst = a.base + a.mipOffset[L];
		if (x0 + 3 < lw && y0 + 3 < lh)
		{
#pragma unroll
			for (int r = 0; r < 4; ++r)
				*reinterpret_cast<float4*>(dst + (size_t)(y0 + r) * lw + x0) = make_float4(q[r][0], q[r][1], q[r][2], q[r][3]);
		}
		else
		{
#pragma unroll
			for (int r = 0; r < 4; ++r)
#pragma unroll
				for (int c = 0; c < 4; ++c)
					if (x0 + c < lw && y0 + r < lh)
						dst[(size_t)(y0 + r) * lw + x0 + c] = q[r][c];
		}
	}
	if (a.numLevels < 2)
		return;

	// ---- level L+1: 2x2 per lane
	float h[2][2];
#pragma unroll
	for (int r = 0; r < 2; ++r)
#pragma unroll
		for (int c = 0; c < 2; ++c)
			h[r][c] = min4(q[2 * r][2 * c], q[2 * r][2 * c + 1], q[2 * r + 1][2 * c], q[2 * r + 1][2 * c + 1]);
	{
		const uint32_t lw = mip_dim(a.pw, L + 1), lh = mip_dim(a.ph, L + 1);
		const uint32_t x0 = blockIdx.x * 32 + tx * 2, y0 = blockIdx.y * 32 + ty * 2;
		float* dst = a.base + a.mipOffset[L + 1];
		if (x0 + 1 < lw && y0 + 1 < lh)
		{
			*reinterpret_cast<float2*>(dst + (size_t)y0 * lw + x0) = make_float2(h[0][0], h[0][1]);
			*reinterpret_cast<float2*>(dst + (size_t)(y0 + 1) * lw + x0) = make_float2(h[1][0], h[1][1]);
		}
		else
		{
#pragma unroll
			for (int r = 0; r < 2; ++r)
#pragma unroll
				for (int c = 0; c < 2; ++c)
					if (x0 + c < lw && y0 + r < lh)
						dst[(size_t)(y0 + r) * lw + x0 + c] = h[r][c];
		}
	}
	if (a.numLevels < 3)
		return;

	// ---- level L+2: one texel per lane, staged in LDS for the rest of the chain
	const float t2 = min4(h[0][0], h[0][1], h[1][0], h[1][1]);
	{
		const uint32_t lw = mip_dim(a.pw, L + 2), lh = mip_dim(a.ph, L + 2);
		const uint32_t x = blockIdx.x * 16 + tx, y = blockIdx.y * 16 + ty;
		if (x < lw && y < lh)
			a.base[a.mipOffset[L + 2] + (size_t)y * lw + x] = t2;
	}
	if (a.numLevels < 4)
		return;
	s_l2[ty][tx] = t2;
	__syncthreads();

	// ---- levels L+3 .. L+6 from LDS (8x8, 4x4, 2x2, 1x1 per workgroup)
	if (threadIdx.x < 64)
	{
		const uint32_t x = threadIdx.x & 7u, y = threadIdx.x >> 3;
		const float t = min4(s_l2[2 * y][2 * x], s_l2[2 * y][2 * x + 1], s_l2[2 * y + 1][2 * x], s_l2[2 * y + 1][2 * x + 1]);
		s_l3[y][x] = t;
		const uint32_t lw = mip_dim(a.pw, L + 3), lh = mip_dim(a.ph, L + 3);
		const uint32_t gx = blockIdx.x * 8 + x, gy = blockIdx.y * 8 + y;
		if (gx < lw && gy < lh)
			a.base[a.mipOffset[L + 3] + (size_t)gy * lw + gx] = t;
	}
	if (a.numLevels < 5)
		return;
	__syncthreads();
	if (threadIdx.x < 16)
	{
		const uint32_t x = threadIdx.x & 3u, y = threadIdx.x >> 2;
		const float t = min4(s_l3[2 * y][2 * x], s_l3[2 * y][2 * x + 1], s_l3[2 * y + 1][2 * x], s_l3[2 * y + 1][2 * x + 1]);
		s_l4[y][x] = t;
		const uint32_t lw = mip_dim(a.pw, L + 4), lh = mip_dim(a.ph, L + 4);
		const uint32_t gx = blockIdx.x * 4 + x, gy = blockIdx.y * 4 + y;
		if (gx < lw && gy < lh)
			a.base[a.mipOffset[L + 4] + (size_t)gy * lw + gx] = t;
	}
	if (a.numLevels < 6)
		return;
	__syncthreads();
	if (threadIdx.x < 4)
	{
		const uint32_t x = threadIdx.x & 1u, y = threadIdx.x >> 1;
		const float t = min4(s_l4[2 * y][2 * x], s_l4[2 * y][2 * x + 1], s_l4[2 * y + 1][2 * x], s_l4[2 * y + 1][2 * x + 1]);
		s_l5[y][x] = t;
		const uint32_t lw = mip_dim(a.pw, L + 5), lh = mip_dim(a.ph, L + 5);
		const uint32_t gx = blockIdx.x * 2 + x, gy = blockIdx.y * 2 + y;
		if (gx < lw && gy < lh)
			a.base[a.mipOffset[L + 5] + (size_t)gy * lw + gx] = t;
	}
	if (a.numLevels < 7)
		return;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		const float t = min4(s_l5[0][0], s_l5[0][1], s_l5[1][0], s_l5[1][1]);
		const uint32_t lw = mip_dim(a.pw, L + 6), lh = mip_dim(a.ph, L + 6);
		if (blockIdx.x < lw && blockIdx.y < lh)
			a.base[a.mipOffset[L + 6] + (size_t)blockIdx.y * lw + blockIdx.x] = t;
	}
}


// ---- first stage for large sources: row-coalesced variant of the chain.  A wave reads 8 source rows x 256 columns,
// one 16-B load per lane and row (1 KiB contiguous per wave-instruction), and reduces 4 x 8 -> 2 x 4 -> 1 x 2 in
// registers; level +2 pairs neighbouring lanes with a DPP shuffle, levels +3 and +4 pair the workgroup's four waves
// (32 source rows) through 512 B of LDS.  Five levels per launch, every store a contiguous run.
// WAVES = 8 (round 5): 64 source rows per workgroup and a SIXTH level, so that the single-workgroup tail launch behind it starts
// from a 64 x 64 level (16 KB) instead of 128 x 128 (64 KB) and has one LDS stage less.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void reduce_rows_kernel(ChainArgs a)
{
	__shared__ float s_l2[WAVES][32];

	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t L = a.firstLevel;
	const uint32_t col0 = blockIdx.x * 256u + lane * 4u; // source column of this lane
	const uint32_t row0 = blockIdx.y * (WAVES * 8u) + wave * 8u;  // first source row of this wave

	// the depth target is read once: non-temporal loads, which leave the caches to the pyramid the late passes probe (round 4: the two launches
	// 29.7-29.8 -> 27.7-27.9 us by events, the frame 202.4-203.2 -> 198.0-200.2 us — the passes behind the pyramid find more of their data in the L2)
	typedef float v4f __attribute__((ext_vector_type(4)));
	float4 v[8];
#pragma unroll
	for (int r = 0; r < 8; ++r)
	{
		const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a.src + (size_t)(row0 + r) * a.sw + col0));
		v[r] = make_float4(t.x, t.y, t.z, t.w);
	}

	// level L: 2 x 4 per lane
	float q[4][2];
#pragma unroll
	for (int i = 0; i < 4; ++i)
	{
		q[i][0] = min4(v[2 * i].x, v[2 * i].y, v[2 * i + 1].x, v[2 * i + 1].y);
		q[i][1] = min4(v[2 * i].z, v[2 * i].w, v[2 * i + 1].z, v[2 * i + 1].w);
	}
	{
		const uint32_t lw = a.sw / 2;
		float* dst = a.base + a.mipOffset[L] + (size_t)(row0 / 2) * lw + col0 / 2;
#pragma unroll
		for (int i = 0; i < 4; ++i)
			*reinterpret_cast<float2*>(dst + (size_t)i * lw) = make_float2(q[i][0], q[i][1]); // (cacheable: stored non-temporal, level 0 costs the late pass's occlusion stage 6 us — 58.7 -> 65.1 at frame scale)
	}
	if (a.numLevels < 2)
		return;

	// level L+1: 1 x 2 per lane
	float h[2];
	h[0] = min4(q[0][0], q[0][1], q[1][0], q[1][1]);
	h[1] = min4(q[2][0], q[2][1], q[3][0], q[3][1]);
	{
		const uint32_t lw = a.sw / 4;
		float* dst = a.base + a.mipOffset[L + 1] + (size_t)(row0 / 4) * lw + col0 / 4;
		dst[0] = h[0];
		dst[lw] = h[1];
	}
	if (a.numLevels < 3)
		return;

	// level L+2: lane pairs; (x0,y0) = even lane's h[0], (x1,y0) = odd lane's h[0], (x0,y1) = even h[1], (x1,y1) = odd h[1]
	// (the neighbour lane's values through DPP quad_perm:[1,0,3,2] — a VALU move — where __shfl_xor goes through LDS: two round trips on
	// every wave's load -> reduce -> store chain)
	const float n0 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(h[0]), 0xB1, 0xf, 0xf, false));
	const float n1 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(h[1]), 0xB1, 0xf, 0xf, false));
	const float t = min4(h[0], n0, h[1], n1); // meaningful on even lanes
	if ((lane & 1u) == 0)
	{
		const uint32_t lw = a.sw / 8;
		a.base[a.mipOffset[L + 2] + (size_t)(row0 / 8) * lw + col0 / 8] = t;
		s_l2[wave][lane >> 1] = t;
	}
	if (a.numLevels < 4)
		return;
	__syncthreads();

	// levels L+3 (WAVES / 2 x 16 per workgroup), L+4 (WAVES / 4 x 8) and, with eight waves, L+5 (1 x 4): lanes 0 .. 8 WAVES - 1 of wave 0, lane = y * 16 + x.
	// The four texels of a footprint always enter min4 in the sampler's order (x0,y0) (x1,y0) (x0,y1) (x1,y1).
	if (threadIdx.x < 8u * WAVES)
	{
		const uint32_t x = threadIdx.x & 15u, y = threadIdx.x >> 4;
		float m = min4(s_l2[2 * y][2 * x], s_l2[2 * y][2 * x + 1], s_l2[2 * y + 1][2 * x], s_l2[2 * y + 1][2 * x + 1]);
		{
			const uint32_t lw = a.sw / 16;
			a.base[a.mipOffset[L + 3] + (size_t)(blockIdx.y * (WAVES / 2) + y) * lw + blockIdx.x * 16 + x] = m;
		}
		if (a.numLevels >= 5)
		{
			m = min4(m, __shfl_xor(m, 1, 64), __shfl_xor(m, 16, 64), __shfl_xor(m, 17, 64)); // meaningful on lanes with y and x even
			if ((y & 1u) == 0 && (x & 1u) == 0)
			{
				const uint32_t lw = a.sw / 32;
				a.base[a.mipOffset[L + 4] + (size_t)(blockIdx.y * (WAVES / 4) + y / 2) * lw + blockIdx.x * 8 + x / 2] = m;
			}
			if (WAVES == 8 && a.numLevels >= 6)
			{
				m = min4(m, __shfl_xor(m, 2, 64), __shfl_xor(m, 32, 64), __shfl_xor(m, 34, 64)); // meaningful on lanes with y == 0 and x a multiple of 4
				if (y == 0 && (x & 3u) == 0)
				{
					const uint32_t lw = a.sw / 64;
					a.base[a.mipOffset[L + 5] + (size_t)blockIdx.y * lw + blockIdx.x * 4 + x / 4] = m;
				}
			}
		}
	}
}

static bool halves(uint32_t s, uint32_t d) { return s == 2 * d || (s == 1 && d == 1); }

int launch_depthreduce(hipStream_t stream, const float* depth, uint32_t w, uint32_t h, const NvPyramidDesc& pyr)
{
	const float* src = depth;
	uint32_t sw = w, sh = h;
	uint32_t L = 0;

	if (!(halves(w, pyr.width) && halves(h, pyr.height)))
	{
		dim3 grid((pyr.width + 31) / 32, (pyr.height + 7) / 8);
		hipLaunchKernelGGL(reduce_generic_kernel, grid, dim3(256), 0, stream, src, sw, sh, pyr.d_base + pyr.mipOffset[0], pyr.width, pyr.height);
		src = pyr.d_base + pyr.mipOffset[0];
		sw = pyr.width;
		sh = pyr.height;
		L = 1;
	}

	while (L < pyr.levels)
	{
		ChainArgs a;
		a.src = src;
		a.sw = sw;
		a.sh = sh;
		a.base = pyr.d_base;
		a.pw = pyr.width;
		a.ph = pyr.height;
		a.firstLevel = L;
		a.numLevels = pyr.levels - L < 7 ? pyr.levels - L : 7;
		for (uint32_t i = 0; i < NV_MAX_MIPS; ++i)
			a.mipOffset[i] = pyr.mipOffset[i];
		uint32_t lw = pyr.width >> L, lh = pyr.height >> L;
		lw = lw ? lw : 1;
		lh = lh ? lh : 1;
		// big sources: row-coalesced five-level stage (needs whole 256 x 32 source tiles and exact halving down to its
		// last level); everything else: the seven-level 128 x 128 tile chain
		const bool rows = sw % 256 == 0 && sh % 32 == 0 && sw == 2 * lw && sh == 2 * lh && sw >= 512 && sh >= 64 && pyr.levels - L >= 5;
		const bool rows6 = rows && sh % 64 == 0 && pyr.levels - L >= 6; // 256 x 64 source tiles, six levels
		if (rows6)
		{
			a.numLevels = 6;
			hipLaunchKernelGGL(reduce_rows_kernel<8>, dim3(sw / 256, sh / 64), dim3(512), 0, stream, a);
		}
		else if (rows)
		{
			a.numLevels = 5;
			hipLaunchKernelGGL(reduce_rows_kernel<4>, dim3(sw / 256, sh / 32), dim3(256), 0, stream, a);
		}
		else
		{
			dim3 grid((lw + 63) / 64, (lh + 63) / 64);
			hipLaunchKernelGGL(reduce_chain_kernel, grid, dim3(256), 0, stream, a);
		}

		uint32_t last = L + a.numLevels - 1;
		src = pyr.d_base + pyr.mipOffset[last];
		sw = pyr.width >> last;
		sh = pyr.height >> last;
		sw = sw ? sw : 1;
		sh = sh ? sh : 1;
		L += a.numLevels;
	}
	return (int)hipGetLastError();
}

} // namespace nv
