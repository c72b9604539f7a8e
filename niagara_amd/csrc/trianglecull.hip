// trianglecull.hip — the mesh stage's per-triangle cull for gfx950.
//
// Replaces src/shaders/meshlet.mesh.glsl:91-198 with MESH_CULL = 1 (src/config.h:10-11), minus the shading attributes:
// per slot of the grid clustersubmit wrote, the meshlet's vertices go to screen space exactly like the mesh shader
// computes them and each triangle gets its gl_CullPrimitiveEXT decision.
//
// Mapping to CDNA4: MESH_WGSIZE = 64 = one wavefront, and the wave is the workgroup: the reference's `shared vec3 vertexClip[]` + barrier()
// become LDS that one wave writes and reads in order.  The cost of a slot is (a) its dependent chain (cluster index -> task command ->
// {draw, meshlet header} -> {vertex refs, index bytes} -> vertices: five round trips for ~0.6 KB of payload), so the chain is walked for
// many slots at once — a wave owns a contiguous run of slots, fetches their headers lane-parallel (lane = slot: three trips per 64 slots),
// then requests the references / index bytes of a whole batch of slots together, then their vertices together, and only then computes
// (first version, one slot at a time: 348 us for 131 k clusters) — and (b) VALU issue: ~120 instructions per vertex pass, ~45 per triangle
// pass, so the passes are PACKED: the vertices (triangles) of a batch form one stream and a pass takes 64 consecutive positions of it,
// whatever slots they belong to (round 4; through round 3 a pass was one slot's vertices or 64 of its triangles, about half of the lanes).
// Algorithmic bytes per slot = 4 + 20 + 48 + 24 + refs (2 or 4 B x vertexCount) + 3 B x triangleCount + 8 B x vertexCount (position half
// of the 16-B vertex) + 16 B out.
#include "cullmath.h"
#include "args.h"

namespace nv
{

constexpr int TC_WAVES = 4;
constexpr int TC_THREADS = TC_WAVES * 64;

#ifndef TC_VCAP
#define TC_VCAP 256 // vertices of one batch (a multiple of 64)
#endif
#ifndef TC_TCAP
#define TC_TCAP 384 // triangles of one batch (a multiple of 64)
#endif
#ifndef TC_CHUNK
#define TC_CHUNK 32 // slots whose headers a wave fetches together (lane = slot): 64 or 32
#endif
#ifndef TC_BLOCKS_PER_CU
#define TC_BLOCKS_PER_CU 6 // <= 8: the partial totals are sized for 8 workgroups per CU (context.hip)
#endif

// mat4 * vec4 in the reference's association: ((c0*x + c1*y) + c2*z) + c3*w
NV_DEV void mat4_mul(const float* m, float x, float y, float z, float w, float out[4])
{
#pragma unroll
	for (int r = 0; r < 4; ++r)
		out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

NV_DEV uint32_t rl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
NV_DEV uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// the wave is the reference's workgroup: its LDS accesses are ordered, the compiler is told so
NV_DEV void wave_lds_order()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The 64 consecutive stream positions of a pass -> the batch slot each belongs to.  `starts` holds one bit per slot but the batch's first,
// at (first stream position of the slot) - 1: the slots that begin at or before position g are the set bits below g — the pass's own
// 64 bits through v_mbcnt, the earlier passes' as a running count.
NV_DEV uint32_t stream_slot(const uint32_t* starts, uint32_t pass, uint32_t& before)
{
	const uint2 m = *reinterpret_cast<const uint2*>(starts + 2 * pass); // one address for the wave
	const uint32_t lo = rfl(m.x), hi = rfl(m.y);
	const uint32_t k = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, before));
	before += (uint32_t)__builtin_popcount(lo) + (uint32_t)__builtin_popcount(hi);
	return k;
}

// Round 4: vertices and triangles PACKED across the slots of a batch.  Through round 3 a wave took one slot per pass — lane = vertex, then
// lane = triangle, 64 + 32 lanes — and a meshlet fills about half of that (the synthetic payloads: 33 of 64 vertex lanes, 48 of 128 triangle
// lanes; real meshlets are fuller, never full), while the launch is bound by VALU issue.  Now a batch is as many consecutive slots as fit
// TC_VCAP vertices and TC_TCAP triangles; its vertices form one stream, its triangles another, and a pass takes 64 consecutive positions of
// a stream whatever slots they belong to.  Per-slot data (draw transform, meshlet header, where the slot's streams begin) sits in a 48-byte
// LDS record per slot that a lane reads by the slot its position falls into; a triangle lane fetches its three index bytes itself (two
// aligned words + v_alignbyte) instead of through an LDS copy; the keep bits leave as ballots into an LDS bit stream from which lane = slot
// cuts its 96 bits (three funnel shifts) and stores its 16-byte mask: one coalesced store per batch.
template <uint32_t VCAP, uint32_t TCAP, uint32_t CHUNK>
__global__ __launch_bounds__(TC_THREADS) void trianglecull_kernel(TriangleArgs a)
{
	constexpr uint32_t VP = VCAP / 64, TP = TCAP / 64;
	constexpr uint32_t VSTART = 0, TSTART = VCAP / 32, KEEP = TSTART + TCAP / 32, NBITS = KEEP + TCAP / 32 + 4;
	static_assert((CHUNK == 64 || CHUNK == 32) && VCAP % 64 == 0 && TCAP % 64 == 0 && VCAP >= 64 && TCAP >= 128 && VCAP <= 1024, "batch capacities");
	// per slot of the chunk: [0] {dataOffset, baseVertex, ve | shortRefs << 8, first vertex position}, [1] [2] draw words 0-3, 4-7,
	// [3] {indexOffset, last index word | ve << 8 | te << 16, first triangle position, first vertex position}
	__shared__ uint4 s_rec[TC_WAVES][CHUNK][4];
	__shared__ float4 s_clip[TC_WAVES][VCAP]; // screen x, y, clip w per vertex of the batch (16-byte slots: one ds_read_b96 per corner)
	__shared__ __attribute__((aligned(8))) uint32_t s_bits[TC_WAVES][NBITS];

	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// grid of the consumer: {cc4[1], cc4[2], cc4[3]} = {16, Y, 16}; index = x + 256 y + 16 z enumerates [0, 256 Y)
	const uint32_t slots = a.cc4[1] * a.cc4[2] * a.cc4[3];
	const uint32_t numWaves = gridDim.x * TC_WAVES;
	const uint32_t w = blockIdx.x * TC_WAVES + wave;
	const uint32_t per = (slots + numWaves - 1) / numWaves; // contiguous slots per wave
	const uint32_t begin = w * per < slots ? w * per : slots;
	const uint32_t end = begin + per < slots ? begin + per : slots;
	uint32_t* bits = s_bits[wave];
	float4* clipv = s_clip[wave];

	uint32_t clusters = 0, triangles = 0, keptTotal = 0; // per lane (= slot of a chunk); a wave's run of slots keeps them far below 2^32

	for (uint32_t chunk = begin; chunk < end; chunk += CHUNK)
	{
		const uint32_t cnt = end - chunk < CHUNK ? end - chunk : CHUNK;

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
		const bool live = hCi != ~0u;
		const uint32_t vcRaw = hCounts & 0xffu, tcRaw = hCounts >> 8 & 0xffu;
		// what the shader's loops cover: `i < vertexCount` over 64 lanes, `i < triangleCount` over MESH_MAXTRI = 96 (a larger count is
		// malformed input; the mask has 96 bits; oracle: the same clamps)
		const uint32_t ve = live ? (vcRaw < 64u ? vcRaw : 64u) : 0u;
		const uint32_t te = live ? (tcRaw < 96u ? tcRaw : 96u) : 0u;
		// stream lengths: every slot of the chunk takes at least one position, so that no two slots begin at the same one
		const uint32_t vs = lane < cnt ? (ve ? ve : 1u) : 0u;
		const uint32_t ts = lane < cnt ? (te ? te : 1u) : 0u;
		const uint32_t svIncl = wave_scan_inclusive_u32(vs), stIncl = wave_scan_inclusive_u32(ts);
		const uint32_t svEx = svIncl - vs, stEx = stIncl - ts; // < 64 * 64 = 2^12, < 64 * 96 < 2^13
		wave_lds_order(); // the previous chunk's readers are done
		const uint32_t shortRefs = (hCounts >> 16 & 0xffu) == 1u ? 1u : 0u;
		const uint32_t indexOffset = hDataOffset + (shortRefs ? (vcRaw + 1) / 2 : vcRaw);
		const uint32_t idxLast = ((tcRaw * 3u + 3u) / 4u - 1u) & 0xffu; // (tc = 0: no triangle reads it)
		if (CHUNK == 64 || lane < CHUNK)
		{
			s_rec[wave][lane][0] = make_uint4(hDataOffset, hBaseVertex, ve | shortRefs << 8, svEx);
			s_rec[wave][lane][3] = make_uint4(indexOffset, idxLast | ve << 8 | te << 16, stEx, svEx);
			s_rec[wave][lane][1] = make_uint4(__float_as_uint(hD0.x), __float_as_uint(hD0.y), __float_as_uint(hD0.z), __float_as_uint(hD0.w));
			s_rec[wave][lane][2] = make_uint4(__float_as_uint(hD1.x), __float_as_uint(hD1.y), __float_as_uint(hD1.z), __float_as_uint(hD1.w));
		}

		// ---- the chunk's batches.  A batch [b, e) is the longest run of slots from b whose streams fit (one slot always does).
		// (Measured and not kept, tools/experiments/trianglecull_pipelined_r4.diff: the loads of batch n + 1 issued before the arithmetic of
		// batch n — 117 VGPRs, one idle batch of loads behind every chunk, 55.9 us against 52.1; batches of 384 / 576 and 512 / 768: the
		// instruction count falls and the launch gets slower, 63-68 us, it is the waves' phases that must interleave; chunks handed out
		// through a ticket counter instead of a contiguous run per wave, for the tail: 164-202 us — 4-10 k returning atomics on one
		// address are the whole launch.)
		struct Batch
		{
			uint32_t b, e, svB, stB, vTotal, tTotal;      // uniform
			uint32_t vSlot[VP], vJ[VP], vX[VP], vY[VP];   // slot; vertex within the slot | shortRefs << 8 (~0: idle lane); {baseVertex, reference word}, then the vertex's position words
			uint32_t tW0[TP], tW1[TP], tMisc[TP];         // misc = byte shift | first clip slot of the meshlet << 2 | ve << 12 | active << 31
		};
		auto plan = [&](Batch& g, uint32_t b) {
			g.b = b;
			g.svB = rl(svEx, b), g.stB = rl(stEx, b);
			const uint64_t fit = __ballot(lane >= b && lane < cnt && svIncl - g.svB <= VCAP && stIncl - g.stB <= TCAP); // a prefix of [b, cnt)
			g.e = b + (uint32_t)__builtin_popcountll(fit);
			g.vTotal = rl(svIncl, g.e - 1) - g.svB, g.tTotal = rl(stIncl, g.e - 1) - g.stB;
		};
		// stages 1 + 2: where the slots begin in the two streams (bit maps), then per stream position the slot record and the first loads
		auto issue_refs = [&](Batch& g) {
			wave_lds_order();
			for (uint32_t i = lane; i < KEEP; i += 64)
				bits[i] = 0;
			wave_lds_order();
			if (lane > g.b && lane < g.e)
			{
				const uint32_t pv = svEx - g.svB - 1, pt = stEx - g.stB - 1;
				atomicOr(&bits[VSTART + (pv >> 5)], 1u << (pv & 31u));
				atomicOr(&bits[TSTART + (pt >> 5)], 1u << (pt & 31u));
			}
			wave_lds_order();
			// lane = vertex of the batch: slot record -> vertex reference.  Every pass and every lane loads (idle ones word 0): behind a
			// branch hipcc cannot count a load, and the first use of any load then waits for all of them
			uint32_t before = 0;
#pragma unroll
			for (uint32_t p = 0; p < VP; ++p)
			{
				const uint32_t slot = g.b + stream_slot(bits + VSTART, p, before);
				const uint4 r0 = s_rec[wave][slot][0];
				const uint32_t j = p * 64 + lane + g.svB - r0.w;
				const bool active = j < (r0.z & 0xffu);
				g.vSlot[p] = slot;
				g.vX[p] = r0.y;
				g.vJ[p] = active ? j | (r0.z & 256u) : ~0u;
				g.vY[p] = a.meshletData[active ? r0.x + (j >> (r0.z >> 8)) : 0u];
			}
			// lane = triangle of the batch: slot record -> the two words that hold its three index bytes
			before = 0;
#pragma unroll
			for (uint32_t t = 0; t < TP; ++t)
			{
				const uint32_t slot = g.b + stream_slot(bits + TSTART, t, before);
				const uint4 r3 = s_rec[wave][slot][3];
				const uint32_t i = t * 64 + lane + g.stB - r3.z;
				const bool active = i < r3.y >> 16;
				const uint32_t w0 = i * 3u >> 2, last = r3.y & 0xffu;
				const uint32_t w1 = w0 < last ? w0 + 1 : last; // (the bytes of triangle i < tc end inside the meshlet's words)
				g.tW0[t] = a.meshletData[active ? r3.x + w0 : 0u];
				g.tW1[t] = a.meshletData[active ? r3.x + w1 : 0u];
				g.tMisc[t] = active ? (i * 3u & 3u) | (r3.w - g.svB) << 2 | (r3.y >> 8 & 0xffu) << 12 | 1u << 31 : 0u;
			}
		};
		// stage 3: the vertices those references name (position half of the 16-byte record)
		auto issue_vertices = [&](Batch& g) {
#pragma unroll
			for (uint32_t p = 0; p < VP; ++p)
			{
				const uint32_t ref = g.vJ[p] & 256u ? ((g.vJ[p] & 1u) ? g.vY[p] >> 16 : g.vY[p] & 0xffffu) : g.vY[p];
				const uint2 pv = *reinterpret_cast<const uint2*>(a.vertices + (g.vJ[p] != ~0u ? ref + g.vX[p] : 0u)); // (idle lanes: vertex 0)
				g.vX[p] = pv.x, g.vY[p] = pv.y;
			}
		};
		// stage 4: vertex phase (meshlet.mesh.glsl:121-160), a pass = 64 vertices of the batch; returns the lanes that produced a NaN
		auto vertex_phase = [&](const Batch& g) -> uint64_t {
			uint64_t nanSeen = 0;
#pragma unroll
			for (uint32_t p = 0; p < VP; ++p)
			{
				if (p * 64 < g.vTotal)
				{
					bool isNan = false;
					if (g.vJ[p] != ~0u)
					{
						const uint4 r1 = s_rec[wave][g.vSlot[p]][1], r2 = s_rec[wave][g.vSlot[p]][2];
						const f3 position = { half_bits_to_float(g.vX[p] & 0xffffu), half_bits_to_float(g.vX[p] >> 16), half_bits_to_float(g.vY[p] & 0xffffu) };
						const f3 q = { __uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z) };
						const f3 rot = rotate_quat(position, q, __uint_as_float(r2.w));
						const float scale = __uint_as_float(r1.w);
						const float wx = rot.x * scale + __uint_as_float(r1.x);
						const float wy = rot.y * scale + __uint_as_float(r1.y);
						const float wz = rot.z * scale + __uint_as_float(r1.z);
						float v4[4], clip[4];
#pragma unroll
						for (int r = 0; r < 4; ++r) // view * vec4(wpos, 1): c3 * 1.0f is c3 exactly
							v4[r] = ((a.globals.cullData.view[r] * wx + a.globals.cullData.view[4 + r] * wy) + a.globals.cullData.view[8 + r] * wz) + a.globals.cullData.view[12 + r];
						mat4_mul(a.globals.projection, v4[0], v4[1], v4[2], v4[3], clip);
						// vertexClip[i] = vec3((clip.xy / clip.w * 0.5 + vec2(0.5)) * screen, clip.w)
						const float sx = ((clip[0] / clip[3]) * 0.5f + 0.5f) * a.globals.screenWidth;
						const float sy = ((clip[1] / clip[3]) * 0.5f + 0.5f) * a.globals.screenHeight;
						clipv[p * 64 + lane] = make_float4(sx, sy, clip[3], 0.0f);
						isNan = sx != sx || sy != sy;
					}
					nanSeen |= __ballot(isNan);
				}
			}
			return nanSeen;
		};
		// stage 5: triangle phase (:166-205), a pass = 64 triangles of the batch; the keep bits go into the LDS bit stream
		auto triangle_phase = [&](const Batch& g, uint64_t nanSeen) {
			for (uint32_t i = KEEP + lane; i < NBITS; i += 64)
				bits[i] = 0;
			wave_lds_order(); // also barrier() of the reference: the vertex phase's LDS writes
#pragma unroll
			for (uint32_t t = 0; t < TP; ++t)
			{
				if (t * 64 < g.tTotal)
				{
					const bool active = (g.tMisc[t] >> 31) != 0;
					const uint32_t idx = __builtin_amdgcn_alignbyte(g.tW1[t], g.tW0[t], g.tMisc[t] & 3u);
					const uint32_t ia = idx & 63u, ib = idx >> 8 & 63u, ic = idx >> 16 & 63u;
					const uint32_t base = g.tMisc[t] >> 2 & 0x3ffu, ve = g.tMisc[t] >> 12 & 0x7fu;
					const uint32_t top = ia > ib ? (ia > ic ? ia : ic) : (ib > ic ? ib : ic);
					bool keep = false;
					// usual case: every vertex a finite screen position and every index below the vertex count.  Then the 2 x 2 chains of
					// gl_min / gl_max are plain minima / maxima (v_min3 / v_max3: only the sign of a zero can differ from the chain's, and
					// round() and == do not see it)
					if (nanSeen == 0 && __ballot(active && top >= ve) == 0)
					{
						if (active)
						{
							const float4 pa = clipv[base + ia], pb = clipv[base + ib], pc = clipv[base + ic];
							const float ebx = pb.x - pa.x, eby = pb.y - pa.y;
							const float ecx = pc.x - pa.x, ecy = pc.y - pa.y;
							bool culled = ebx * ecy <= eby * ecx; // backface + zero-area
							const float bminx = __builtin_fminf(__builtin_fminf(pa.x, pb.x), pc.x), bminy = __builtin_fminf(__builtin_fminf(pa.y, pb.y), pc.y);
							const float bmaxx = __builtin_fmaxf(__builtin_fmaxf(pa.x, pb.x), pc.x), bmaxy = __builtin_fmaxf(__builtin_fmaxf(pa.y, pb.y), pc.y);
							const float sbprec = 1.0f / 256.0f;
							culled = culled | (__builtin_rintf(bminx - sbprec) == __builtin_rintf(bmaxx)) | (__builtin_rintf(bminy) == __builtin_rintf(bmaxy + sbprec));
							culled = culled & (pa.z > 0) & (pb.z > 0) & (pc.z > 0);
							keep = !culled;
						}
					}
					else if (active)
					{
						// a vertex the meshlet does not have reads as the shader's zero-initialised slot (oracle: memset)
						auto corner = [&](uint32_t i) {
							float4 v = clipv[i < ve ? base + i : base];
							v.x = i < ve ? v.x : 0.0f, v.y = i < ve ? v.y : 0.0f, v.z = i < ve ? v.z : 0.0f;
							return v;
						};
						const float4 pa = corner(ia), pb = corner(ib), pc = corner(ic);
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
					if (lane == 0)
						*reinterpret_cast<uint2*>(bits + KEEP + 2 * t) = make_uint2((uint32_t)ballot, (uint32_t)(ballot >> 32));
				}
			}
			wave_lds_order();
		};
		// stage 6, lane = slot: the slot's 96 bits out of the batch's bit stream, one 16-byte store per slot
		auto store_masks = [&](const Batch& g) {
			if (lane >= g.b && lane < g.e)
			{
				uint4 out = make_uint4(0, 0, 0, 0);
				if (live)
				{
					const uint32_t at = stEx - g.stB;
					const uint32_t* q = bits + KEEP + (at >> 5);
					const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], sh = at & 31u;
					const uint32_t n0 = te < 32u ? te : 32u, n1 = te < 32u ? 0u : (te < 64u ? te - 32u : 32u), n2 = te < 64u ? 0u : te - 64u;
					out.x = __builtin_amdgcn_alignbit(q1, q0, sh) & (n0 == 32u ? ~0u : (1u << n0) - 1u);
					out.y = __builtin_amdgcn_alignbit(q2, q1, sh) & (n1 == 32u ? ~0u : (1u << n1) - 1u);
					out.z = __builtin_amdgcn_alignbit(q3, q2, sh) & (n2 == 32u ? ~0u : (1u << n2) - 1u);
					const uint32_t kept = (uint32_t)__builtin_popcount(out.x) + (uint32_t)__builtin_popcount(out.y) + (uint32_t)__builtin_popcount(out.z);
					out.w = tcRaw | vcRaw << 8 | kept << 16;
					clusters += 1;
					triangles += tcRaw;
					keptTotal += kept;
				}
				if (chunk + lane < a.capacity)
					*reinterpret_cast<uint4*>(a.masks + chunk + lane) = out;
			}
		};

		Batch g;
		for (uint32_t b = 0; b < cnt; b = g.e)
		{
			plan(g, b);
			issue_refs(g);
			issue_vertices(g);
			const uint64_t nanSeen = vertex_phase(g);
			triangle_phase(g, nanSeen);
			store_masks(g);
		}
	}

	// totals: per-workgroup partial sums, plain stores; a one-workgroup kernel adds them up.  (Atomics from every wave
	// into the caller's three adjacent counters — one cache line — serialise in its L2 channel: 25 k of them took ~280 us
	// of a 320 us launch.)
	__shared__ unsigned long long s_tot[TC_WAVES][3];
	const uint32_t wc = wave_sum_u32(clusters), wt = wave_sum_u32(triangles), wk = wave_sum_u32(keptTotal);
	if (lane == 0)
	{
		s_tot[wave][0] = wc;
		s_tot[wave][1] = wt;
		s_tot[wave][2] = wk;
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
	gridBlocks = gridBlocks / 8 * TC_BLOCKS_PER_CU; // the caller passes 8 workgroups per CU, the size of `partials`
	hipLaunchKernelGGL((trianglecull_kernel<TC_VCAP, TC_TCAP, TC_CHUNK>), dim3(gridBlocks), dim3(TC_THREADS), 0, stream, a);
	return launch_totals3(stream, a.partials, gridBlocks, a.totals);
}

} // namespace nv
