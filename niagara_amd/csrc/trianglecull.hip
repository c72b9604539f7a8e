// trianglecull.hip — the mesh stage's per-triangle cull for gfx950.
//
// Replaces src/shaders/meshlet.mesh.glsl:91-198 with MESH_CULL = 1 (src/config.h:10-11), minus the shading attributes:
// per slot of the grid clustersubmit wrote, the meshlet's vertices go to screen space exactly like the mesh shader
// computes them and each triangle gets its gl_CullPrimitiveEXT decision.
//
// Mapping to CDNA4: MESH_WGSIZE = 64 = one wavefront per meshlet, MESH_MAXVTX = 64 = one vertex per lane, MESH_MAXTRI =
// 96 = two triangle rounds per lane.  The reference's `shared vec3 vertexClip[]` + barrier() become a 768-byte LDS
// slice per wave and slot and nothing else: the wave is the workgroup.  The cost of a slot is its dependent chain
// (cluster index -> task command -> {draw, meshlet header} -> {vertex refs, index bytes} -> vertices: five round trips
// for ~0.6 KB of payload), so the chain is walked for many slots at once: a wave owns a contiguous run of slots, fetches
// their headers lane-parallel (lane = slot: three trips per 64 slots), then requests the references / index bytes of
// four slots together, then their vertices together, and only then computes (first version, one slot at a time:
// 348 us for 131 k clusters).  Bound: HBM (gather-heavy); algorithmic bytes per slot =
// 4 + 20 + 48 + 24 + refs (2 or 4 B x vertexCount) + 3 B x triangleCount + 8 B x vertexCount (position half of the 16-B
// vertex) + 16 B out.
#include "cullmath.h"
#include "args.h"

namespace nv
{

constexpr int TC_WAVES = 4;
constexpr int TC_THREADS = TC_WAVES * 64;

// mat4 * vec4 in the reference's association: ((c0*x + c1*y) + c2*z) + c3*w
NV_DEV void mat4_mul(const float* m, float x, float y, float z, float w, float out[4])
{
#pragma unroll
	for (int r = 0; r < 4; ++r)
		out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

NV_DEV uint32_t rl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
NV_DEV float rlf(float v, uint32_t l) { return __uint_as_float(rl(__float_as_uint(v), l)); }

constexpr uint32_t TC_BATCH = 4; // slots whose payload loads are in flight together

__global__ __launch_bounds__(TC_THREADS) void trianglecull_kernel(TriangleArgs a)
{
	__shared__ float4 s_clip[TC_WAVES][TC_BATCH][64]; // screen x, y, clip w (16-byte slots: one ds_read_b128 per corner)
	__shared__ uint32_t s_idx[TC_WAVES][TC_BATCH][80]; // 72 dwords of index bytes per slot (MESH_MAXTRI * 3 / 4)

	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// grid of the consumer: {cc4[1], cc4[2], cc4[3]} = {16, Y, 16}; index = x + 256 y + 16 z enumerates [0, 256 Y)
	const uint32_t slots = a.cc4[1] * a.cc4[2] * a.cc4[3];
	const uint32_t numWaves = gridDim.x * TC_WAVES;
	const uint32_t w = blockIdx.x * TC_WAVES + wave;
	const uint32_t per = (slots + numWaves - 1) / numWaves; // contiguous slots per wave
	const uint32_t begin = w * per < slots ? w * per : slots;
	const uint32_t end = begin + per < slots ? begin + per : slots;
	const uint16_t* data16 = reinterpret_cast<const uint16_t*>(a.meshletData);

	unsigned long long clusters = 0, triangles = 0, keptTotal = 0;

	for (uint32_t chunk = begin; chunk < end; chunk += 64)
	{
		const uint32_t cnt = end - chunk < 64u ? end - chunk : 64u;

		// ---- stage 0, lane = slot: cluster index -> task command -> {meshlet header, draw}.  Three dependent round trips for
		// up to 64 slots at once instead of per slot.
		uint32_t hCi = ~0u, hDataOffset = 0, hBaseVertex = 0, hCounts = 0; // counts = vertexCount | triangleCount << 8 | shortRefs << 16
		float4 hD0 = make_float4(0, 0, 0, 0), hD1 = make_float4(0, 0, 0, 1);
		if (lane < cnt)
			hCi = a.clusterIndices[chunk + lane];
		if (hCi != ~0u)
		{
			const uint32_t* cmd = reinterpret_cast<const uint32_t*>(a.commands + (hCi & 0xffffffu));
			const uint32_t drawId = cmd[0], taskOffset = cmd[1];
			const uint32_t mi = taskOffset + (hCi >> 24);
			const uint32_t* mw = reinterpret_cast<const uint32_t*>(a.meshlets + mi);
			hDataOffset = mw[3];
			hBaseVertex = mw[4];
			hCounts = mw[5] & 0xffffffu;
			const float4* dp = reinterpret_cast<const float4*>(a.draws + drawId);
			hD0 = dp[0];
			hD1 = dp[1];
		}

		for (uint32_t b = 0; b < cnt; b += TC_BATCH)
		{
			// ---- stage 1: vertex references and index bytes of TC_BATCH slots, all requested before any is used
			uint32_t vc[TC_BATCH], tc[TC_BATCH], ref[TC_BATCH], iw0[TC_BATCH], iw1[TC_BATCH];
			bool live[TC_BATCH];
#pragma unroll
			for (uint32_t k = 0; k < TC_BATCH; ++k)
			{
				const uint32_t s = b + k < cnt ? b + k : cnt - 1;
				const uint32_t counts = rl(hCounts, s);
				live[k] = b + k < cnt && rl(hCi, s) != ~0u;
				vc[k] = live[k] ? counts & 0xffu : 0u;
				tc[k] = live[k] ? counts >> 8 & 0xffu : 0u;
				const bool shortRefs = (counts >> 16 & 0xffu) == 1u;
				const uint32_t dataOffset = rl(hDataOffset, s);
				const uint32_t indexOffset = dataOffset + (shortRefs ? (vc[k] + 1) / 2 : vc[k]);
				const uint32_t idxWords = (tc[k] * 3u + 3u) / 4u;
				ref[k] = 0;
				if (lane < vc[k])
					ref[k] = shortRefs ? (uint32_t)data16[dataOffset * 2 + lane] : a.meshletData[dataOffset + lane];
				iw0[k] = lane < idxWords ? a.meshletData[indexOffset + lane] : 0u;
				iw1[k] = lane < 16u && lane + 64u < idxWords ? a.meshletData[indexOffset + 64u + lane] : 0u;
			}
			// ---- stage 2: the vertices those references name (position half of the 16-byte record)
			uint2 pv[TC_BATCH];
#pragma unroll
			for (uint32_t k = 0; k < TC_BATCH; ++k)
			{
				const uint32_t s = b + k < cnt ? b + k : cnt - 1;
				pv[k] = make_uint2(0, 0);
				if (lane < vc[k])
					pv[k] = *reinterpret_cast<const uint2*>(a.vertices + (ref[k] + rl(hBaseVertex, s)));
				s_idx[wave][k][lane] = iw0[k];
				if (lane < 16u)
					s_idx[wave][k][64 + lane] = iw1[k];
			}
			// ---- stage 3: per slot, vertex phase (meshlet.mesh.glsl:121-160, lane = vertex) then triangle phase (:166-205)
#pragma unroll
			for (uint32_t k = 0; k < TC_BATCH; ++k)
			{
				if (b + k >= cnt)
					break;
				const uint32_t s = b + k;
				NvTriangleMask out = { { 0, 0, 0 }, 0 };
				if (live[k])
				{
					float4* clipv = s_clip[wave][k];
					if (lane < vc[k])
					{
						const f3 position = { half_bits_to_float(pv[k].x & 0xffffu), half_bits_to_float(pv[k].x >> 16), half_bits_to_float(pv[k].y & 0xffffu) };
						const f3 q = { rlf(hD1.x, s), rlf(hD1.y, s), rlf(hD1.z, s) };
						const f3 rot = rotate_quat(position, q, rlf(hD1.w, s));
						const float scale = rlf(hD0.w, s);
						const float wx = rot.x * scale + rlf(hD0.x, s);
						const float wy = rot.y * scale + rlf(hD0.y, s);
						const float wz = rot.z * scale + rlf(hD0.z, s);
						float v4[4], clip[4];
#pragma unroll
						for (int r = 0; r < 4; ++r) // view * vec4(wpos, 1): c3 * 1.0f is c3 exactly
							v4[r] = ((a.globals.cullData.view[r] * wx + a.globals.cullData.view[4 + r] * wy) + a.globals.cullData.view[8 + r] * wz) + a.globals.cullData.view[12 + r];
						mat4_mul(a.globals.projection, v4[0], v4[1], v4[2], v4[3], clip);
						// vertexClip[i] = vec3((clip.xy / clip.w * 0.5 + vec2(0.5)) * screen, clip.w)
						clipv[lane] = make_float4(((clip[0] / clip[3]) * 0.5f + 0.5f) * a.globals.screenWidth,
						                          ((clip[1] / clip[3]) * 0.5f + 0.5f) * a.globals.screenHeight, clip[3], 0.0f);
					}
					// barrier() of the reference: the wave is the workgroup, its LDS accesses are ordered
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

					const uint8_t* idx8 = reinterpret_cast<const uint8_t*>(s_idx[wave][k]);
					uint32_t kept = 0;
#pragma unroll
					for (uint32_t round = 0; round < 2; ++round)
					{
						const uint32_t i = round * 64u + lane;
						bool keep = false;
						if (i < tc[k] && i < 96u) // MESH_MAXTRI: a larger count is malformed input; the mask has 96 bits (oracle: same clamp)
						{
							const uint32_t ia = idx8[i * 3], ib = idx8[i * 3 + 1], ic = idx8[i * 3 + 2];
							const float4 pa = clipv[ia & 63u], pb = clipv[ib & 63u], pc = clipv[ic & 63u];
							bool culled = false;
							const float ebx = pb.x - pa.x, eby = pb.y - pa.y;
							const float ecx = pc.x - pa.x, ecy = pc.y - pa.y;
							culled = culled || (ebx * ecy <= eby * ecx); // backface + zero-area
							const float bminx = gl_min(pa.x, gl_min(pb.x, pc.x)), bminy = gl_min(pa.y, gl_min(pb.y, pc.y));
							const float bmaxx = gl_max(pa.x, gl_max(pb.x, pc.x)), bmaxy = gl_max(pa.y, gl_max(pb.y, pc.y));
							const float sbprec = 1.0f / 256.0f;
							// round(): half-to-even (v_rndne_f32), the definition the oracle and the shim share
							culled = culled || (__builtin_rintf(bminx - sbprec) == __builtin_rintf(bmaxx) || __builtin_rintf(bminy) == __builtin_rintf(bmaxy + sbprec));
							culled = culled && (pa.z > 0 && pb.z > 0 && pc.z > 0);
							keep = !culled;
						}
						const uint64_t ballot = __ballot(keep);
						if (round == 0)
						{
							out.keep[0] = (uint32_t)ballot;
							out.keep[1] = (uint32_t)(ballot >> 32);
						}
						else
							out.keep[2] = (uint32_t)ballot;
						kept += (uint32_t)__builtin_popcountll(ballot);
					}
					out.counts = (tc[k] & 0xffu) | (vc[k] & 0xffu) << 8 | kept << 16;
					clusters += 1;
					triangles += tc[k];
					keptTotal += kept;
				}
				if (lane == 0 && chunk + s < a.capacity)
					*reinterpret_cast<uint4*>(a.masks + chunk + s) = make_uint4(out.keep[0], out.keep[1], out.keep[2], out.counts);
			}
			__builtin_amdgcn_wave_barrier(); // the next batch overwrites this wave's LDS slices
		}
	}

	// totals: per-workgroup partial sums, plain stores; a one-workgroup kernel adds them up.  (Atomics from every wave
	// into the caller's three adjacent counters — one cache line — serialise in its L2 channel: 25 k of them took ~280 us
	// of a 320 us launch.)
	__shared__ unsigned long long s_tot[TC_WAVES][3];
	if (lane == 0)
	{
		s_tot[wave][0] = clusters;
		s_tot[wave][1] = triangles;
		s_tot[wave][2] = keptTotal;
	}
	__syncthreads();
	if (threadIdx.x < 3)
	{
		unsigned long long t = 0;
#pragma unroll
		for (int k = 0; k < TC_WAVES; ++k)
			t += s_tot[k][threadIdx.x];
		a.partials[(size_t)blockIdx.x * 3 + threadIdx.x] = t;
	}
}

int launch_totals3(hipStream_t stream, const unsigned long long* partials, uint32_t blocks, unsigned long long* totals); // submit.hip

int launch_trianglecull(hipStream_t stream, const TriangleArgs& a, uint32_t gridBlocks)
{
	hipLaunchKernelGGL(trianglecull_kernel, dim3(gridBlocks), dim3(TC_THREADS), 0, stream, a);
	return launch_totals3(stream, a.partials, gridBlocks, a.totals);
}

} // namespace nv
