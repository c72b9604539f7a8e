// drawcull.hip — per-draw frustum / HiZ cull, LOD selection and ordered indirect-command compaction for gfx950.
//
// Replaces src/shaders/drawcull.comp.glsl:54-156 (4 pipelines LATE x TASK, src/niagara.cpp:724-727).
//
// Mapping to CDNA4 — two wait-free launches on the stream, the same scheme as clustercull.hip:
//   K1 draw_decide_kernel   one lane per draw; a grid that does not grow with the draw count, each wave walking its own range of
//                           64-draw units with the records of the next DS_DEPTH units in flight.  Decides visibility / LOD
//                           (reference arithmetic), writes one result byte per draw (LOD | emit << 3 | old visibility << 4),
//                           the new drawVisibility word (LATE), and adds its command counts to the counts of the scatter tiles
//                           they fall in (one fire-and-forget atomic per wave and tile).
//   K2 draw_scatter_kernel  one workgroup per CU owns a contiguous range of draws; its append base is the count word
//                           plus the counts of the tiles before it, so no workgroup waits on another.  Each lane owns
//                           16 consecutive draws (one 16-B load of result bytes); the append index of a draw is the
//                           exclusive prefix sum of the emit counts in draw order — one valid serialisation of the
//                           reference's atomicAdd (drawcull.comp.glsl:123,143), and bit-reproducible.
//                           TASK mode, per-draw form (draws of one or two task groups): a draw's commands are written by its
//                           own lane or, beyond four, wave-cooperatively (the owning lane's draw, LOD range and dci broadcast
//                           with readlane, 64 lanes writing consecutive 20-B MeshTaskCommands) instead of one lane looping over
//                           up to hundreds of commands (drawcull.comp.glsl:131-138).  TASK mode, list form (many task groups per
//                           draw): one LANE per output command; since round 6 fed by 16-byte records of the emitting draws that
//                           K1 leaves per wave, so that K2 reads neither result bytes nor draw words nor the Mesh table.
#include "cullmath.h"
#include "args.h"

namespace nv
{

constexpr int DC_WAVES = 4;
constexpr int DC_THREADS = DC_WAVES * 64;
constexpr uint32_t DC_MESH_LDS = 64;  // meshes staged in LDS (13 KiB) when the table is registered and small enough


struct DrawResult
{
	uint32_t count;   // commands this draw appends (0 if none)
	uint32_t lodWord; // lodIndex | emit<<8
	uint32_t oldVis;  // drawVisibility[di] before this pass (lateDrawVisibility)
};

NV_DEV uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane)
{
	(void)lane;
	return wave_scan_inclusive_u32(v); // six DPP adds (args.h) instead of six ds_bpermute round trips
}

// drawcull.comp.glsl:56-118 + :154-155 for one draw
// What a decision needs of a Mesh, 20 words (the decide kernel's LDS table when the Mesh table is registered):
//   [0..3] center.xyz, radius   [4] lodCount   [5..11] lods[1..7].error   [12..19] lods[0..7].meshletCount   [20..27] lods[0..7].meshletOffset
constexpr uint32_t DC_LOD_WORDS = 28;
// word k of that record = word lod_table_source(k) of the 52-word NvMesh
NV_DEV uint32_t lod_table_source(uint32_t k)
{
	return k < 4u ? k : (k == 4u ? 8u : (k < 12u ? 12u + 5u * (k - 4u) + 4u : (k < 20u ? 12u + 5u * (k - 12u) + 3u : 12u + 5u * (k - 20u) + 2u)));
}

// drawcull.comp.glsl:56-118 + :154-155 for one draw.  COMPACT: meshBase is the LDS table above, else the NvMesh array.
// WORLD: the draw comes from the mirror of nv_upload_draws — d0 = the world-space sphere {rotateQuat(center, q) * scale +
// position, radius * scale}, i.e. the view-independent prefix of drawcull.comp.glsl:73-75 evaluated once at upload in the
// reference's own operation order (bit-identical intermediates), d1.x = scale; only the view transform is left per pass.
// The decision in three parts, so that the late pass can run the occlusion probe of a whole workgroup's visible draws on
// compacted lanes (draw_decide_kernel): decide_pre = drawcull.comp.glsl:56-84 (the early-outs, the sphere in view space, the
// frustum test), draw_probe = :86-99 (HiZ), decide_post = :101-118 + :154-155 (LOD, counts, the new drawVisibility word).
struct DrawPre
{
	f3 c;
	float radius, scale;
	const char* mesh;
	bool skip;    // the draw is not part of this pass (postPass mismatch / early pass and invisible last frame): nothing is written
	bool visible; // frustum (or culling disabled)
};

template <bool LATE, bool COMPACT, bool WORLD>
NV_DEV DrawPre decide_pre(const DrawArgs& a, const char* meshBase, const float4& d0, const float4& d1, const uint4& d2, uint32_t oldVis)
{
	const NvCullData& cd = a.cd;
	DrawPre pre = { { 0.0f, 0.0f, 0.0f }, 0.0f, 0.0f, meshBase, true, false };
	if (d2.z != cd.postPass) // drawData.postPass
		return pre;
	if (!LATE && oldVis == 0)
		return pre;
	pre.skip = false;

	const uint32_t meshIndex = d2.x;
	pre.mesh = meshBase + (size_t)meshIndex * (COMPACT ? DC_LOD_WORDS * 4u : sizeof(NvMesh));
	if (WORLD)
	{
		pre.c = view_point(cd.view, f3{ d0.x, d0.y, d0.z });
		pre.radius = d0.w;
		pre.scale = d1.x;
	}
	else
	{
		const float4 cr = *reinterpret_cast<const float4*>(pre.mesh); // center.xyz, radius
		f3 q = { d1.x, d1.y, d1.z };
		pre.c = sphere_center(cd, f3{ cr.x, cr.y, cr.z }, q, d1.w, d0.w, f3{ d0.x, d0.y, d0.z });
		pre.radius = cr.w * d0.w;
		pre.scale = d0.w;
	}
	pre.visible = frustum_test(cd, pre.c, pre.radius) || cd.cullingEnabled == 0;
	return pre;
}

// drawcull.comp.glsl:86-99
NV_DEV bool draw_probe(const DrawArgs& a, f3 c, float radius)
{
	const HizProbe p = hiz_prepare(a.cd, a.pyr, c, radius, a.pyr.mipOffset);
	if (!(p.use & 16u))
		return true;
	const float* base = a.pyr.d_base;
	return hiz_finish(p, base[p.o00], base[p.o10], base[p.o01], base[p.o11]);
}

template <bool LATE, bool TASK, bool COMPACT>
NV_DEV DrawResult decide_post(const DrawArgs& a, const DrawPre& pre, bool visible, uint32_t di, uint32_t oldVis)
{
	const NvCullData& cd = a.cd;
	DrawResult res = { 0, 0, oldVis };
	if (pre.skip)
		return res;
	const char* mesh = pre.mesh;
	const f3 c = pre.c;
	const float radius = pre.radius, scale = pre.scale;

	// TASK_CULL == 1 (src/config.h:8)
	if (visible && (!LATE || cd.clusterOcclusionEnabled == 1 || oldVis == 0 || cd.postPass != 0))
	{
		uint32_t lodIndex = 0;
		if (cd.lodEnabled == 1 && !NV_DBG(a, 2u))
		{
			float distance = gl_max(length3(c) - radius, 0.0f);
			float threshold = distance * cd.lodTarget / scale;
			// drawcull.comp.glsl:108-110: the last LOD below the threshold.  All eight slots exist in the struct, so the
			// errors are fetched together and the loop bound becomes part of the condition.
			uint32_t lodCount;
			float err[NV_MAX_LODS];
			if (COMPACT)
			{
				const float4 e0 = reinterpret_cast<const float4*>(mesh)[1], e1 = reinterpret_cast<const float4*>(mesh)[2];
				lodCount = __float_as_uint(e0.x);
				err[1] = e0.y, err[2] = e0.z, err[3] = e0.w;
				err[4] = e1.x, err[5] = e1.y, err[6] = e1.z, err[7] = e1.w;
			}
			else
			{
				lodCount = *reinterpret_cast<const uint32_t*>(mesh + 32);
#pragma unroll
				for (uint32_t i = 1; i < NV_MAX_LODS; ++i)
					err[i] = *reinterpret_cast<const float*>(mesh + 48 + 20 * i + 16);
			}
			// "the last i in [1, lodCount) with err[i] < threshold" = the highest set bit of the comparisons' mask under the count's mask.  As the
			// loop reads (`if (i < lodCount && err[i] < threshold) lodIndex = i`) hipcc ANDs the two compares on the scalar unit into VCC and
			// selects on it — seven s_and_b64 vcc / v_cndmask pairs at ~23 cycles each on this chip (a VCC written by the scalar unit stalls the
			// vector instruction that reads it: tools/experiments/valu_classes.hip `vcc`) against three vector instructions per LOD here.
			uint32_t below = 1u; // bit 0: LOD 0 is always a candidate
#pragma unroll
			for (uint32_t i = 1; i < NV_MAX_LODS; ++i)
				below |= err[i] < threshold ? 1u << i : 0u;
			const uint32_t counted = lodCount < NV_MAX_LODS ? lodCount : NV_MAX_LODS; // (i < lodCount for every i of the loop once lodCount >= 8)
			below &= (1u << counted) - 1u | 1u;
			lodIndex = 31u - (uint32_t)__builtin_clz(below);
		}
		res.lodWord = lodIndex | 0x100u;
		if (TASK)
		{
			uint32_t meshletCount = COMPACT ? reinterpret_cast<const uint32_t*>(mesh)[12 + lodIndex]
			                                : *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lodIndex + 12);
			res.count = (meshletCount + NV_TASK_WGSIZE - 1) / NV_TASK_WGSIZE;
		}
		else
			res.count = 1;
	}

	if (LATE)
		a.dvb[di] = visible ? 1u : 0u;
	return res;
}

struct DrawLoad
{
	float4 d0, d1; // position.xyz, scale | orientation
	uint4 d2;      // meshIndex, meshletVisibilityOffset, postPass, materialIndex
	uint32_t oldVis;
};

// SOA: the streams of the mirror (a wave reads 1 KiB + 512 B + 256 B contiguous); otherwise the 48-B record in place
// (three 16-B loads at a 48-B stride)
template <bool SOA>
NV_DEV DrawLoad load_draw_fields(const DrawArgs& a, uint32_t di)
{
	DrawLoad l;
	if (SOA)
	{
		l.d0 = a.soaWorld[di];
		const uint2 sm = a.soaScaleMesh[di];
		l.d1 = make_float4(__uint_as_float(sm.x), 0.0f, 0.0f, 0.0f);
		l.d2 = make_uint4(sm.y, 0u, a.soaPostPass[di], 0u); // the decision reads meshIndex and postPass only
	}
	else
	{
		const float4* p = reinterpret_cast<const float4*>(a.draws + di);
		l.d0 = p[0];
		l.d1 = p[1];
		l.d2 = *reinterpret_cast<const uint4*>(p + 2);
	}
	l.oldVis = 0u;
	return l;
}

template <bool SOA>
NV_DEV DrawLoad load_draw_record(const DrawArgs& a, uint32_t di)
{
	DrawLoad l = load_draw_fields<SOA>(a, di);
	l.oldVis = a.dvb[di];
	return l;
}

// draws per scatter tile: an even split over the scatter grid, rounded up to whole waves of the decide kernel
static uint32_t scatter_tile_draws(uint32_t drawCount, uint32_t tiles)
{
	const uint32_t t = (drawCount + tiles - 1) / tiles;
	return t < 64u ? 64u : (t + 63u) / 64u * 64u;
}

// The Mesh table (center/radius, LOD errors, LOD ranges) is read by every draw: staged in LDS once per workgroup when
// nv_upload_meshes registered a table of at most DC_MESH_LDS meshes, otherwise gathered from global memory.
// Two halves, because s_waitcnt vmcnt counts in issue order: the table's loads are ISSUED before the workgroup's draw
// records and committed to LDS after them, so that waiting for the table does not wait for the records.
constexpr uint32_t DC_STAGE_WORDS = (DC_MESH_LDS * sizeof(NvMesh) / 4 + DC_THREADS - 1) / DC_THREADS;
struct MeshStage
{
	uint32_t w[DC_STAGE_WORDS];
};

template <bool MESH_LDS>
NV_DEV MeshStage stage_mesh_issue(const DrawArgs& a)
{
	MeshStage st;
	if (MESH_LDS)
	{
		const uint32_t words = a.meshCount * (uint32_t)(sizeof(NvMesh) / 4);
		const uint32_t* src = reinterpret_cast<const uint32_t*>(a.meshes);
#pragma unroll
		for (uint32_t k = 0; k < DC_STAGE_WORDS; ++k)
		{
			const uint32_t i = k * DC_THREADS + threadIdx.x;
			st.w[k] = src[i < words ? i : 0u]; // clamped: unconditional loads
		}
	}
	return st;
}

template <bool MESH_LDS>
NV_DEV const char* stage_mesh_commit(const DrawArgs& a, const MeshStage& st, uint32_t* s_meshTable)
{
	if (!MESH_LDS)
		return reinterpret_cast<const char*>(a.meshes);
	const uint32_t words = a.meshCount * (uint32_t)(sizeof(NvMesh) / 4);
#pragma unroll
	for (uint32_t k = 0; k < DC_STAGE_WORDS; ++k)
	{
		const uint32_t i = k * DC_THREADS + threadIdx.x;
		if (i < words)
			s_meshTable[i] = st.w[k];
	}
	__syncthreads();
	return reinterpret_cast<const char*>(s_meshTable);
}

// The decide kernel's table: only the 20 words per mesh a decision reads (DC_LOD_WORDS), gathered from the NvMesh records
// — 5 loads per thread instead of 13, and the LOD errors of a mesh in two 16-byte LDS reads.
constexpr uint32_t DC_LOD_STAGE_WORDS = (DC_MESH_LDS * DC_LOD_WORDS + DC_THREADS - 1) / DC_THREADS;
struct LodStage
{
	uint32_t w[DC_LOD_STAGE_WORDS];
};

template <bool MESH_LDS>
NV_DEV LodStage stage_lod_issue(const DrawArgs& a)
{
	LodStage st;
	if (MESH_LDS)
	{
		const uint32_t words = a.meshCount * DC_LOD_WORDS;
		const uint32_t* src = reinterpret_cast<const uint32_t*>(a.meshes);
#pragma unroll
		for (uint32_t k = 0; k < DC_LOD_STAGE_WORDS; ++k)
		{
			const uint32_t i = k * DC_THREADS + threadIdx.x;
			const uint32_t j = i < words ? i : 0u; // clamped: unconditional loads
			st.w[k] = src[(j / DC_LOD_WORDS) * (uint32_t)(sizeof(NvMesh) / 4) + lod_table_source(j % DC_LOD_WORDS)];
		}
	}
	return st;
}

template <bool MESH_LDS>
NV_DEV const char* stage_lod_commit(const DrawArgs& a, const LodStage& st, uint32_t* s_lodTable)
{
	if (!MESH_LDS)
		return reinterpret_cast<const char*>(a.meshes);
	const uint32_t words = a.meshCount * DC_LOD_WORDS;
#pragma unroll
	for (uint32_t k = 0; k < DC_LOD_STAGE_WORDS; ++k)
	{
		const uint32_t i = k * DC_THREADS + threadIdx.x;
		if (i < words)
			s_lodTable[i] = st.w[k];
	}
	__syncthreads();
	return reinterpret_cast<const char*>(s_lodTable);
}

// K1: a grid that does not grow with the draw count (at most DS_MAX_BLOCKS workgroups = two waves per SIMD), each wave walking its own
// contiguous range of 64-draw units with the records of the next DS_DEPTH units requested ahead of the one it decides — for 1 M draws
// the dispatcher starts 2 k waves instead of the 7.8 k of one workgroup per 512 draws (rounds 1-5), and no wave's loads wait behind
// its own arithmetic.  The decision is split in decide_pre / draw_probe / decide_post: what follows the frustum test (the occlusion
// probe, the LOD choice, the counts) runs on the wave's own queue of survivors, up to 64 at a time on consecutive lanes — the probe's
// ~360 instructions are executed once per 64 survivors instead of once per wave that holds one (round 2 compacted them per workgroup
// behind two barriers) and no workgroup barrier follows the staging of the Mesh table.  The survivors of a wave are queued in draw
// order, so the scatter tiles their counts fall in are non-decreasing and a wave adds to each tile's counter once.
#ifndef DS_MAX_BLOCKS
#define DS_MAX_BLOCKS 512
#endif
#ifndef DS_DEPTH
#define DS_DEPTH 4
#endif
#ifndef DS_DEFER_SM
#define DS_DEFER_SM 1 // (0: tools/build_variant.sh A/B — the early pass's walk requests {scale, meshIndex} of every draw with the other streams)
#endif
constexpr uint32_t DS_QUEUE = 128; // survivors a wave may hold: a unit adds at most 64 to fewer than 64

struct TileRun
{
	uint32_t tile, sum, emit;
};

template <bool TASK>
NV_DEV void tile_run_flush(const DrawArgs& a, uint32_t bank, const TileRun& run, uint32_t lane)
{
	if (lane == 0 && !NV_DBG(a, 4u))
	{
		if (run.sum)
			atomicAdd(&a.tileCounts->counts[bank][run.tile * CC_COUNT_STRIDE], run.sum);
		if (TASK && run.emit)
			atomicAdd(&a.tileCounts->counts[bank][run.tile * CC_COUNT_STRIDE + 1], run.emit);
	}
}

template <bool LATE, bool TASK, bool MESH_LDS, bool SOA, bool VISFIRST, bool RECORDS>
__global__ __launch_bounds__(DC_THREADS) void draw_decide_kernel(DrawArgs a)
{
	__shared__ __attribute__((aligned(16))) uint32_t s_lodTable[MESH_LDS ? DC_MESH_LDS * DC_LOD_WORDS : 4];
	// (DS_DEFER_SM: the survivors of the last DS_DEPTH + 1 units may wait for their {scale, meshIndex} behind fewer than 64 whole entries)
	constexpr uint32_t QUEUE = DS_DEFER_SM && SOA && !LATE && !(VISFIRST && DS_DEFER_SM == 1) ? 64u * (DS_DEPTH + 2u) : DS_QUEUE;
	__shared__ float4 s_q0[DC_WAVES][QUEUE]; // view-space centre, radius
	__shared__ uint4 s_q1[DC_WAVES][QUEUE];  // scale, meshIndex, draw, previous visibility

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

	const uint32_t drawCount = a.cd.drawCount;
	const uint32_t T2 = a.tileDraws;
	const uint32_t bank = load_uniform_u32(&a.tileCounts->parity) & 1u;
	if (blockIdx.x == 0 && tid == 0)
	{
		a.tileCounts->k2parity = bank;
		a.tileCounts->base = a.fusedReset ? 0u : a.count4[0];
	}

	// the wave's units: [u0, u1)
	const uint32_t units = (drawCount + 63u) / 64u;
	const uint32_t per = a.unitsPerWave;
	const uint32_t w = blockIdx.x * DC_WAVES + wave;
	static_assert(!RECORDS || TASK, "records feed the TASK scatter's list form");
	constexpr bool records = RECORDS;
	uint32_t recorded = 0;
	const uint32_t u0 = w * per < units ? w * per : units;
	const uint32_t u1 = u0 + per < units ? u0 + per : units;

	// Indices past the range are clamped, not branched: unconditional loads the compiler can count (s_waitcnt vmcnt(N)).
	// VISFIRST (early pass, when few draws were visible last frame): the visibility words run DS_DEPTH units ahead of the records, and
	// a lane whose draw was not visible — it leaves at drawcull.comp.glsl:66 whatever its record holds — requests the record of the
	// unit's first draw instead of its own: the instruction stays unconditional, the lines it fetches are those of the visible draws.
	static_assert(!(LATE && VISFIRST), "the late pass decides every draw");
	const uint32_t lastDraw = drawCount ? drawCount - 1u : 0u;
	auto draw_of = [&](uint32_t u, uint32_t l) {
		const uint32_t uc = u < u1 ? u : (u1 ? u1 - 1u : 0u);
		const uint32_t di = uc * 64u + l;
		return di < lastDraw ? di : lastDraw;
	};
	auto request = [&](DrawLoad& slot, uint32_t u) { slot = load_draw_record<SOA>(a, draw_of(u, lane)); };
	auto request_visible = [&](DrawLoad& slot, uint32_t u, uint32_t vis) {
		slot = load_draw_fields<SOA>(a, draw_of(u, vis != 0 ? lane : 0u));
		slot.oldVis = vis != 0 ? 1u : 0u; // (a value of its own: the register of the word just read is free for the next one)
	};
	const char* meshBase = nullptr; // (set once the Mesh table is staged; the functions below are called after that)
	const bool probe = LATE && a.cd.occlusionEnabled == 1; // (uniform)
	uint32_t queued = 0;
	TileRun run = { u0 * 64u / T2, 0u, 0u };

	// the queue's first `cnt` survivors on lanes 0..cnt-1: drawcull.comp.glsl:86-118 + :154-155
	auto drain = [&](uint32_t cnt) {
		const bool mine = lane < cnt;
		uint32_t count = 0, di = 0;
		uint32_t mvo = 0, lodRange = 0; // records: the draw's meshletVisibilityOffset (requested ahead of the probe), the LOD's {meshletOffset, meshletCount} at mesh + lodRange
		const char* recMesh = meshBase;
		if (RECORDS)
			mvo = a.draws[s_q1[wave][mine ? lane : 0u].z].meshletVisibilityOffset;
		// late pass over the mirror: {scale, meshIndex} of the survivors only, requested here beside the probe's texels (the walk leaves them out: 8 of 32 B per draw)
		uint2 lateSm = make_uint2(0u, 0u);
		if (LATE && SOA)
			lateSm = a.soaScaleMesh[s_q1[wave][mine ? lane : 0u].z];
		if (mine)
		{
			const float4 q0 = s_q0[wave][lane];
			const uint4 q1 = s_q1[wave][lane];
			di = q1.z;
			DrawPre pre;
			pre.c = f3{ q0.x, q0.y, q0.z };
			pre.radius = q0.w;
			pre.scale = __uint_as_float(LATE && SOA ? lateSm.x : q1.x);
			pre.mesh = meshBase + (size_t)(LATE && SOA ? lateSm.y : q1.y) * (MESH_LDS ? DC_LOD_WORDS * 4u : sizeof(NvMesh));
			pre.skip = false;
			pre.visible = true;
			const bool seen = probe ? draw_probe(a, pre.c, pre.radius) : true;
			const DrawResult res = decide_post<LATE, TASK, MESH_LDS>(a, pre, seen, di, q1.w);
			a.results[di] = (uint8_t)((res.lodWord & 7u) | ((res.lodWord >> 8 & 1u) << 3) | ((q1.w != 0 ? 1u : 0u) << 4));
			count = res.count;
			recMesh = pre.mesh;
			lodRange = res.lodWord & 7u;
			di |= records && q1.w != 0 ? 0x80000000u : 0u; // (stripped again below)
		}
		if (records)
		{
			// one record per emitting draw, in draw order, at the front of the wave's own range of the record array
			const uint64_t emits = __ballot(count != 0);
			if (count != 0)
			{
				uint32_t meshletOffset, meshletCount;
				if (MESH_LDS)
				{
					meshletOffset = reinterpret_cast<const uint32_t*>(recMesh)[20u + lodRange];
					meshletCount = reinterpret_cast<const uint32_t*>(recMesh)[12u + lodRange];
				}
				else
				{
					meshletOffset = *reinterpret_cast<const uint32_t*>(recMesh + 48 + 20 * lodRange + 8);
					meshletCount = *reinterpret_cast<const uint32_t*>(recMesh + 48 + 20 * lodRange + 12);
				}
				const uint32_t at = recorded + __builtin_amdgcn_mbcnt_hi((uint32_t)(emits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)emits, 0u));
				a.records[(size_t)u0 * 64u + at] = make_uint4(meshletOffset, meshletCount, mvo, di);
			}
			recorded += (uint32_t)__builtin_popcountll(emits);
			di &= 0x7fffffffu;
		}
		// (The request of `mvo` is issued and used on every path of the drain — RECORDS is a template parameter for that: a request that is
		// consumed only under a condition hipcc must assume still in flight when the walk goes on, and it then waits for nearly the whole
		// ring — vmcnt(1) — before the next unit's code overwrites the register: +1 us per decide launch at 1 M draws.)
		if (RECORDS)
			asm volatile("" ::"v"(mvo));
		if (LATE && SOA)
			asm volatile("" ::"v"(lateSm.x), "v"(lateSm.y)); // (likewise)
		// counts -> the wave's run of scatter tiles (draws ascend along the queue, so do their tiles)
		uint64_t rest = __ballot(mine);
		while (rest)
		{
			const uint32_t head = (uint32_t)__builtin_ctzll(rest);
			const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)di, (int)head) / T2;
			const bool in = (rest >> lane & 1u) != 0 && di < (t + 1u) * T2;
			const uint32_t sum = wave_sum_u32(in ? count : 0u);
			const uint32_t emit = (uint32_t)__builtin_popcountll(__ballot(in && count != 0));
			if (t != run.tile)
			{
				tile_run_flush<TASK>(a, bank, run, lane);
				run = TileRun{ t, 0u, 0u };
			}
			run.sum += sum;
			run.emit += emit;
			rest &= ~__ballot(in);
		}
		// what is left moves to the front (64 entries at a time, lowest first: a destination is never ahead of a source not yet read)
		for (uint32_t base = 0; cnt + base < queued; base += 64u)
		{
			float4 m0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			uint4 m1 = make_uint4(0u, 0u, 0u, 0u);
			const bool moves = cnt + base + lane < queued;
			if (moves)
			{
				m0 = s_q0[wave][cnt + base + lane];
				m1 = s_q1[wave][cnt + base + lane];
			}
			if (moves)
			{
				s_q0[wave][base + lane] = m0;
				s_q1[wave][base + lane] = m1;
			}
		}
		queued -= cnt;
	};

	// The early pass's walk over the mirror leaves {scale, meshIndex} out like the late pass's (only the LOD select reads them: 8 of 32 B per draw, 24 of 33 MB
	// at 1 M draws).  With no probe to hide the request behind, a unit's survivors request theirs right after the frustum test and the words are put into the
	// queue entries when the slot comes round again (commit_pending); only whole entries drain.
	constexpr bool deferSm = DS_DEFER_SM && SOA && !LATE && !(VISFIRST && DS_DEFER_SM == 1); // (DS_DEFER_SM=2: also in the visibility-first form, where the records are fetched for last frame's visible draws only anyway)
	// one pending request per ring slot: issued when the slot's unit is decided, put into the queue when the slot comes round again (DS_DEPTH units later:
	// consumed any earlier, the in-order return of the loads would make it a wait for every request of the ring issued before it)
	uint2 pendSm[DS_DEPTH];
	uint32_t pendSlot[DS_DEPTH], pendCount[DS_DEPTH], uncommitted = 0;
#pragma unroll
	for (uint32_t k = 0; k < DS_DEPTH; ++k)
	{
		pendSm[k] = make_uint2(0u, 0u);
		pendSlot[k] = ~0u;
		pendCount[k] = 0u;
	}
	auto commit_pending = [&](uint32_t k) {
		if (deferSm)
		{
			asm volatile("" ::"v"(pendSm[k].x), "v"(pendSm[k].y)); // (used on every path: see the drain's `mvo`)
			if (pendSlot[k] != ~0u)
				*reinterpret_cast<uint2*>(&s_q1[wave][pendSlot[k]]) = pendSm[k];
			pendSlot[k] = ~0u;
			uncommitted -= pendCount[k];
			pendCount[k] = 0u;
		}
	};
	auto decide = [&](const DrawLoad& ld, uint32_t u, auto ringed, uint32_t k) {
		constexpr bool deferred = deferSm && decltype(ringed)::value;
		const uint32_t di = u * 64u + lane;
		const bool valid = u < u1 && di < drawCount;
		DrawPre pre = decide_pre<LATE, MESH_LDS, SOA>(a, meshBase, ld.d0, ld.d1, ld.d2, ld.oldVis);
		const bool survives = valid && !pre.skip && pre.visible;
		// a draw that is not part of the pass or fails the frustum test ends here: no command, LATE: drawVisibility 0 (decide_post with visible = false)
		if (valid && !survives)
		{
			a.results[di] = (uint8_t)((ld.oldVis != 0 ? 1u : 0u) << 4);
			if (LATE && !pre.skip)
				a.dvb[di] = 0u;
		}
		const uint64_t want = __ballot(survives);
		if (deferred)
			pendSm[k] = a.soaScaleMesh[survives ? di : (u < u1 ? u : u0) * 64u]; // (every lane, the others the unit's first draw: one unconditional instruction)
		if (want)
		{
			const uint32_t slot = queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
			if (survives)
			{
				s_q0[wave][slot] = make_float4(pre.c.x, pre.c.y, pre.c.z, pre.radius);
				s_q1[wave][slot] = (LATE && SOA) || deferred ? make_uint4(0u, 0u, di, ld.oldVis) : make_uint4(__float_as_uint(pre.scale), ld.d2.x, di, ld.oldVis);
				if (deferred)
					pendSlot[k] = slot;
			}
			queued += (uint32_t)__builtin_popcountll(want);
			if (deferred)
			{
				pendCount[k] = (uint32_t)__builtin_popcountll(want);
				uncommitted += pendCount[k];
			}
		}
	};

	const LodStage st = stage_lod_issue<MESH_LDS>(a);
	// A pass of at most DS_MAX_BLOCKS * DC_WAVES units has one unit per wave and nothing to run ahead of: request, decide, finish — in a
	// block of its own that ends the kernel, so that the ring below has one definition in front of the walk (a value requested under a
	// condition and used under a later one is "maybe in flight" for hipcc on the other path, and the walk then waits for it in every round).
	if (per == 1u) // (uniform over the grid)
	{
		DrawLoad one;
		if (VISFIRST)
			request_visible(one, u0, a.dvb[draw_of(u0, lane)]);
		else
			request(one, u0);
		meshBase = stage_lod_commit<MESH_LDS>(a, st, s_lodTable);
		if (u0 < u1)
		{
			decide(one, u0, std::false_type{}, 0u);
			if (queued)
				drain(queued);
			tile_run_flush<TASK>(a, bank, run, lane);
		}
		if (records && lane == 0)
			a.recordCounts[w] = recorded;
		return;
	}
	DrawLoad ring[DS_DEPTH];
	uint32_t visRing[DS_DEPTH]; // VISFIRST: visRing[k] = the visibility words of the unit DS_DEPTH after ring[k]'s
	if (VISFIRST)
	{
		uint32_t first[DS_DEPTH];
#pragma unroll
		for (uint32_t k = 0; k < DS_DEPTH; ++k)
			first[k] = a.dvb[draw_of(u0 + k, lane)];
#pragma unroll
		for (uint32_t k = 0; k < DS_DEPTH; ++k)
			visRing[k] = a.dvb[draw_of(u0 + DS_DEPTH + k, lane)];
#pragma unroll
		for (uint32_t k = 0; k < DS_DEPTH; ++k)
			request_visible(ring[k], u0 + k, first[k]);
	}
	else
	{
#pragma unroll
		for (uint32_t k = 0; k < DS_DEPTH; ++k)
			request(ring[k], u0 + k);
	}
	meshBase = stage_lod_commit<MESH_LDS>(a, st, s_lodTable);
	if (u0 >= u1)
	{
		if (records && lane == 0)
			a.recordCounts[w] = 0u;
		return;
	}

	// Every slot of the ring is decided and requested again in every round (units past the wave's range: clamped loads, nothing
	// written), and a slot's new loads are issued only after the last use of its old values — so that the loop carries the ring in
	// the same registers and the back edge holds no copy of a register with a load in flight (hipcc waits for a load before it
	// moves its destination: with the request ahead of the decision the ring drained to vmcnt(0) once per round).
	const uint32_t rounds = (u1 - u0 + DS_DEPTH - 1u) / DS_DEPTH;
	for (uint32_t r = 0, u = u0; r < rounds; ++r, u += DS_DEPTH)
	{
#pragma unroll
		for (uint32_t k = 0; k < DS_DEPTH; ++k)
		{
			if (deferSm)
			{
				commit_pending(k); // the survivors of this slot's last unit are whole now ...
				if (queued - uncommitted >= 64u) // ... and only whole entries drain (the queue is in draw order: the whole ones are at its front)
				{
					drain(64u);
#pragma unroll
					for (uint32_t j = 0; j < DS_DEPTH; ++j)
						pendSlot[j] = pendSlot[j] != ~0u ? pendSlot[j] - 64u : ~0u; // (what was left moved to the front)
				}
			}
			if (u + k < u1) // (uniform; a pass of few draws has one unit per wave and DS_DEPTH - 1 idle slots)
				decide(ring[k], u + k, std::true_type{}, k);
			asm volatile("" ::: "memory");
			if (VISFIRST)
			{
				request_visible(ring[k], u + k + DS_DEPTH, visRing[k]);
				asm volatile("" ::: "memory");
				visRing[k] = a.dvb[draw_of(u + k + 2u * DS_DEPTH, lane)];
			}
			else
				request(ring[k], u + k + DS_DEPTH);
			asm volatile("" ::: "memory");
			if (!deferSm && queued >= 64u)
				drain(64u);
		}
	}
#pragma unroll
	for (uint32_t k = 0; k < DS_DEPTH; ++k)
		commit_pending(k);
	while (deferSm && queued > 64u)
		drain(64u);
	if (queued)
		drain(queued);
	tile_run_flush<TASK>(a, bank, run, lane);
	if (records && lane == 0)
		a.recordCounts[w] = recorded;
}

// result bytes of PER consecutive draws, packed little-endian into words
template <uint32_t PER>
NV_DEV void load_result_bytes(const uint8_t* p, uint32_t (&w)[4])
{
	w[0] = w[1] = w[2] = w[3] = 0;
	if (PER == 1)
		w[0] = *p;
	else if (PER == 4)
		w[0] = *reinterpret_cast<const uint32_t*>(p);
	else
	{
		const uint4 v = *reinterpret_cast<const uint4*>(p);
		w[0] = v.x;
		w[1] = v.y;
		w[2] = v.z;
		w[3] = v.w;
	}
}

// K2.  DC_PER_LANE = consecutive draws per lane (1, 4 or 16, chosen by the launcher so that a tile is one step and all
// lanes have work: 16 for >= 1 M draws, 1 for the few thousand draws of a small scene).
// RECORDS (TASK, list form): the decide launch left a 16-byte record per emitting draw (args.h) — no result bytes, no draw words and no
// Mesh table are read here; the records of a tile's waves are requested together with the tile counts (DC_THREADS / waves-per-tile of
// each wave's, which holds them all unless a wave emitted from more draws than that: then the tile's records are walked in order).
template <bool TASK, bool MESH_LDS, uint32_t DC_PER_LANE, bool LIST = false, bool RECORDS = false>
__global__ __launch_bounds__(DC_THREADS) void draw_scatter_kernel(DrawArgs a)
{
	static_assert(!RECORDS || (TASK && LIST), "records feed the list form");
	constexpr uint32_t DC_STEP = DC_THREADS * DC_PER_LANE;
	__shared__ __attribute__((aligned(16))) uint32_t s_meshTable[MESH_LDS && !RECORDS ? DC_MESH_LDS * sizeof(NvMesh) / 4 : 4];
	__shared__ uint32_t s_part[DC_WAVES];
	__shared__ uint32_t s_sum[DC_WAVES];
	// TASK: the step's emitting draws in draw order, {first command of the draw relative to the step's first command,
	// draw within the step | lod << 12 | previous visibility << 15}, + one sentinel (round 3: one LANE per output command)
	__shared__ uint2 s_emit[TASK && LIST && !RECORDS ? DC_STEP + 1 : 1];
	__shared__ uint32_t s_partE[DC_WAVES];
	__shared__ uint4 s_rec[RECORDS ? DC_THREADS : 1];           // RECORDS: the step's records in draw order ...
	__shared__ uint32_t s_recFirst[RECORDS ? DC_THREADS + 1 : 1]; // ... and each one's first command relative to the step's, + one sentinel
	__shared__ uint32_t s_recCount[RECORDS ? DC_THREADS + 1 : 1]; // records per wave of the tile; then their exclusive prefix

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

	const uint32_t drawCount = a.cd.drawCount;
	const uint32_t T = a.tileDraws;
	const uint32_t numTiles = (drawCount + T - 1) / T; // <= gridDim.x
	const uint32_t tile = blockIdx.x;

	// Everything below was written by the decide kernel, i.e. before this launch: plain loads, all issued together.
	// Both banks of tile counts are read speculatively so that no load waits for the parity word.
	const uint32_t k2parity = load_uniform_u32(&a.tileCounts->k2parity);
	const uint32_t base0 = load_uniform_u32(&a.tileCounts->base);
	uint32_t cnt0[2] = { 0, 0 }, cnt1[2] = { 0, 0 }; // this thread's tiles tid and tid + 256, per bank
	uint32_t em0[2] = { 0, 0 }, em1[2] = { 0, 0 };   // TASK, last tile: likewise the tiles' emitting draws (word 1 of the line)
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		const uint32_t i = j * DC_THREADS + tid;
		if (i < numTiles)
		{
			cnt0[j] = a.tileCounts->counts[0][i * CC_COUNT_STRIDE];
			cnt1[j] = a.tileCounts->counts[1][i * CC_COUNT_STRIDE];
			if (TASK && tile == numTiles - 1)
			{
				em0[j] = a.tileCounts->counts[0][i * CC_COUNT_STRIDE + 1];
				em1[j] = a.tileCounts->counts[1][i * CC_COUNT_STRIDE + 1];
			}
		}
	}
	const uint32_t first = tile * T;
	const uint32_t n = tile < numTiles ? (drawCount - first < T ? drawCount - first : T) : 0u;
	uint32_t bw[4] = { 0, 0, 0, 0 }; // first step's result bytes (usually the only step)
	if (!RECORDS && tid * DC_PER_LANE < n)
		load_result_bytes<DC_PER_LANE>(a.results + first + tid * DC_PER_LANE, bw);
	// RECORDS: the tile is wavesPerTile whole ranges of the decide launch; thread tid holds record tid % share of wave tid / share
	const uint32_t waveDraws = a.unitsPerWave * 64u;
	const uint32_t wavesPerTile = RECORDS ? T / waveDraws : 1u;
	const uint32_t share = RECORDS ? DC_THREADS / wavesPerTile : 1u; // >= 1 (the host takes this form for at most DC_THREADS waves per tile)
	uint4 myRec = make_uint4(0u, 0u, 0u, 0u);
	uint32_t myWaveRecords = 0;
	if (RECORDS && tile < numTiles)
	{
		const uint32_t j = tid / share, i = tid - j * share;
		if (j < wavesPerTile && i < waveDraws && (size_t)first + (size_t)j * waveDraws < drawCount)
			myRec = a.records[(size_t)first + (size_t)j * waveDraws + i];
		if (tid < wavesPerTile && (size_t)first + (size_t)tid * waveDraws < drawCount)
			myWaveRecords = a.recordCounts[first / waveDraws + tid];
	}

	const uint32_t bank = k2parity & 1u;
	// every workgroup clears its entries of the other bank for the next pass; one thread flips the parity the next
	// decide kernel will read (this pass reads k2parity only)
	for (uint32_t i = tile * DC_THREADS + tid; i < CC_MAX_SCATTER_TILES; i += gridDim.x * DC_THREADS)
	{
		a.tileCounts->counts[bank ^ 1u][i * CC_COUNT_STRIDE] = 0;
		a.tileCounts->counts[bank ^ 1u][i * CC_COUNT_STRIDE + 1] = 0; // (a TASK pass's emitting draws: cleared by whatever pass comes next)
	}
	if (tile == 0 && tid == 0)
	{
		a.tileCounts->parity = bank ^ 1u;
		if (numTiles == 0 && a.fusedReset)
			a.count4[0] = 0; // no draws: the fused reset still leaves a zero count
		if (numTiles == 0 && TASK && a.fusedSubmit)
		{
			const uint32_t raw = a.fusedReset ? 0u : a.count4[0];
			const uint32_t count = raw < NV_TASK_WGLIMIT ? raw : NV_TASK_WGLIMIT;
			const uint32_t gx = (count + 63u) / 64u;
			a.count4[1] = gx < 65535u ? gx : 65535u;
			a.count4[2] = 64;
			a.count4[3] = 1;
			for (uint32_t i = count; i < ((count + 63u) & ~63u); ++i)
				static_cast<NvMeshTaskCommand*>(a.commands)[i] = NvMeshTaskCommand{ 0, 0, 0, 0, 0 };
		}
	}
	if (tile >= numTiles)
		return;

	const char* meshBase = RECORDS ? nullptr : stage_mesh_commit<MESH_LDS>(a, stage_mesh_issue<MESH_LDS && !RECORDS>(a), s_meshTable);
	if (RECORDS)
		s_recCount[tid] = myWaveRecords; // (0 past the tile's waves; ordered by the barrier below)

	uint32_t before = 0, all = 0;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		const uint32_t i = j * DC_THREADS + tid;
		const uint32_t v = bank ? cnt1[j] : cnt0[j];
		all += v;
		before += i < tile ? v : 0u;
	}
	const uint32_t wBefore = wave_sum_u32(before), wAll = wave_sum_u32(all);
	if (lane == 0)
	{
		s_part[wave] = wBefore;
		s_sum[wave] = wAll;
	}
	__syncthreads();
	uint32_t running = base0, total = base0;
#pragma unroll
	for (int w = 0; w < DC_WAVES; ++w)
	{
		running += s_part[w];
		total += s_sum[w];
	}
	if (tid == 0 && tile == numTiles - 1)
		a.count4[0] = total; // what the chain of atomicAdds leaves in the count word
	if (TASK && tile == numTiles - 1 && a.hostHint)
	{
		// the pass's emitting draws and commands for the host's choice of a later launch's form (mapped words; speed only)
		__shared__ uint32_t s_emitters;
		if (tid == 0)
			s_emitters = 0;
		__syncthreads();
		const uint32_t we = wave_sum_u32(bank ? em1[0] + em1[1] : em0[0] + em0[1]);
		if (lane == 0 && we)
			atomicAdd(&s_emitters, we);
		__syncthreads();
		if (tid == 0)
		{
			__hip_atomic_store(a.hostHint + 2, s_emitters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(a.hostHint + 3, total - base0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	if (TASK && a.fusedSubmit && tile == numTiles - 1)
	{
		// NV_OPT_FUSED_SUBMIT: tasksubmit.comp.glsl:27-47 from the workgroup that knows the final count.  The dummy commands
		// [count, next multiple of 64) lie past every emitted command, so no other workgroup writes there.
		const uint32_t count = total < NV_TASK_WGLIMIT ? total : NV_TASK_WGLIMIT;
		if (tid == 0)
		{
			const uint32_t gx = (count + 63u) / 64u;
			a.count4[1] = gx < 65535u ? gx : 65535u;
			a.count4[2] = 64;
			a.count4[3] = 1;
		}
		const uint32_t boundary = (count + 63u) & ~63u;
		if (tid < 64u && count + tid < boundary)
			static_cast<NvMeshTaskCommand*>(a.commands)[count + tid] = NvMeshTaskCommand{ 0, 0, 0, 0, 0 };
	}

	if constexpr (RECORDS)
	{
		NvMeshTaskCommand* tc = static_cast<NvMeshTaskCommand*>(a.commands);
		// <= DC_THREADS records in draw order, one per thread: drawcull.comp.glsl:120-139 with one LANE per output command (as the list form below)
		auto emit_records = [&](bool valid, const uint4& rec) {
			const uint32_t mine = valid ? (rec.y + NV_TASK_WGSIZE - 1) / NV_TASK_WGSIZE : 0u;
			const uint32_t incl = wave_inclusive_scan(mine, lane);
			const uint32_t mineE = valid ? 1u : 0u; // (an emitting draw has >= 1 command: the decide launch recorded only those)
			const uint32_t inclE = wave_inclusive_scan(mineE, lane);
			__syncthreads(); // s_part and the lists are free
			if (lane == 63)
			{
				s_part[wave] = incl;
				s_partE[wave] = inclE;
			}
			__syncthreads();
			const uint32_t stepBase = running;
			uint32_t dci = running + incl - mine, eIdx = inclE - mineE, stepEmitters = 0;
#pragma unroll
			for (int w = 0; w < DC_WAVES; ++w)
			{
				const uint32_t p = s_part[w], pe = s_partE[w];
				dci += w < (int)wave ? p : 0u;
				eIdx += w < (int)wave ? pe : 0u;
				running += p;
				stepEmitters += pe;
			}
			const uint32_t stepTotal = running - stepBase;
			if (valid)
			{
				s_rec[eIdx] = rec;
				s_recFirst[eIdx] = dci - stepBase;
			}
			if (tid == 0)
				s_recFirst[stepEmitters] = stepTotal;
			__syncthreads();
			for (uint32_t k = tid; k < stepTotal; k += DC_THREADS)
			{
				uint32_t lo = 0, hi = stepEmitters; // invariant: s_recFirst[lo] <= k < s_recFirst[hi]
				while (hi - lo > 1u)
				{
					const uint32_t mid = (lo + hi) >> 1;
					if (s_recFirst[mid] <= k)
						lo = mid;
					else
						hi = mid;
				}
				const uint32_t e = s_recFirst[lo];
				const uint32_t groups = s_recFirst[lo + 1u] - e;
				const uint32_t i = k - e;
				const uint4 r = s_rec[lo];
				if (stepBase + e + groups <= NV_TASK_WGLIMIT) // drop the whole draw on overflow (:128)
				{
					NvMeshTaskCommand cmd;
					cmd.drawId = r.w & 0x7fffffffu;
					cmd.taskOffset = r.x + i * NV_TASK_WGSIZE;
					const uint32_t rest = r.y - i * NV_TASK_WGSIZE;
					cmd.taskCount = rest < NV_TASK_WGSIZE ? rest : NV_TASK_WGSIZE;
					cmd.lateDrawVisibility = r.w >> 31;
					cmd.meshletVisibilityOffset = r.z + i * NV_TASK_WGSIZE;
					tc[stepBase + k] = cmd;
				}
			}
		};
		// (s_recCount was written before the barriers of the base reduction)
		const uint32_t j = tid / share, i = tid - j * share;
		const bool held = !__syncthreads_or(tid < wavesPerTile && myWaveRecords > share ? 1 : 0);
		if (held)
			emit_records(j < wavesPerTile && i < s_recCount[j], myRec);
		else
		{
			// a wave emitted from more draws than the tile's threads hold of it: the tile's records in order, DC_THREADS at a time
			__syncthreads();
			if (tid == 0)
			{
				uint32_t sum = 0;
				for (uint32_t w = 0; w <= wavesPerTile && w <= DC_THREADS; ++w)
				{
					const uint32_t c = w < wavesPerTile ? s_recCount[w] : 0u;
					s_recCount[w] = sum;
					sum += c;
				}
			}
			__syncthreads();
			const uint32_t all = s_recCount[wavesPerTile];
			for (uint32_t k0 = 0; k0 < all; k0 += DC_THREADS)
			{
				const uint32_t k = k0 + tid;
				uint4 rec = make_uint4(0u, 0u, 0u, 0u);
				if (k < all)
				{
					uint32_t lo = 0, hi = wavesPerTile; // invariant: s_recCount[lo] <= k < s_recCount[hi]
					while (hi - lo > 1u)
					{
						const uint32_t mid = (lo + hi) >> 1;
						if (s_recCount[mid] <= k)
							lo = mid;
						else
							hi = mid;
					}
					rec = a.records[(size_t)first + (size_t)lo * waveDraws + (k - s_recCount[lo])];
				}
				emit_records(k < all, rec);
			}
		}
		return;
	}

	for (uint32_t c0 = 0; c0 < n; c0 += DC_STEP)
	{
		const uint32_t c = c0 + tid * DC_PER_LANE; // this lane's first draw within the tile
		if (c0)
		{
			bw[0] = bw[1] = bw[2] = bw[3] = 0;
			if (c < n)
				load_result_bytes<DC_PER_LANE>(a.results + first + c, bw);
		}
		uint32_t res[DC_PER_LANE];  // result byte, 0 past the tile
		uint32_t meshIndex[DC_PER_LANE], mvo[DC_PER_LANE], cnt[DC_PER_LANE];
#pragma unroll
		for (uint32_t j = 0; j < DC_PER_LANE; ++j)
		{
			res[j] = c + j < n ? (bw[j / 4] >> (8 * (j % 4)) & 0xffu) : 0u;
			meshIndex[j] = 0;
			mvo[j] = 0;
			if (res[j] & 8u) // emitting draws only: meshIndex, meshletVisibilityOffset (8 B of the 48-B record)
			{
				const uint2 d2 = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.draws + first + c + j) + 32);
				meshIndex[j] = d2.x;
				mvo[j] = d2.y;
			}
		}
		uint32_t mine = 0;
#pragma unroll
		for (uint32_t j = 0; j < DC_PER_LANE; ++j)
		{
			cnt[j] = 0;
			if (res[j] & 8u)
			{
				if (TASK)
				{
					const char* mesh = meshBase + (size_t)meshIndex[j] * sizeof(NvMesh);
					const uint32_t meshletCount = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * (res[j] & 7u) + 12);
					cnt[j] = (meshletCount + NV_TASK_WGSIZE - 1) / NV_TASK_WGSIZE;
				}
				else
					cnt[j] = 1;
			}
			mine += cnt[j];
		}
		const uint32_t incl = wave_inclusive_scan(mine, lane);
		uint32_t mineE = 0, inclE = 0;
		if (TASK)
		{
#pragma unroll
			for (uint32_t j = 0; j < DC_PER_LANE; ++j)
				mineE += cnt[j] != 0 ? 1u : 0u;
			inclE = wave_inclusive_scan(mineE, lane);
		}
		__syncthreads(); // s_part free (base reduction / previous step's readers are done)
		if (lane == 63)
		{
			s_part[wave] = incl;
			if (TASK)
				s_partE[wave] = inclE;
		}
		__syncthreads();
		const uint32_t stepBase = running; // append index of the step's first command
		uint32_t dci = running + incl - mine;
		uint32_t eIdx = inclE - mineE, stepEmitters = 0;
#pragma unroll
		for (int w = 0; w < DC_WAVES; ++w)
		{
			const uint32_t p = s_part[w];
			dci += w < (int)wave ? p : 0u;
			running += p;
			if (TASK)
			{
				const uint32_t pe = s_partE[w];
				eIdx += w < (int)wave ? pe : 0u;
				stepEmitters += pe;
			}
		}

		// TASK, two forms (LIST, chosen per launch by the host from the previous launches' statistic: frame coherence, speed only).
		// Passes whose emitting draws have many task groups (a 1 M-draw frame over meshes of thousands of meshlets: 7 commands per
		// emitting draw) take the list form below; passes of small draws (one or two commands each: the synthetic contract scene,
		// most real meshes) keep round 2's form further down — every lane writes its own draws' commands, the few large ones are
		// expanded owner by owner — which is 1.5 us faster for them (config 3B: 13.1 against 14.6 us).  One kernel with a per-step
		// choice between the two measured 14.2 us where the list form alone takes 9.4 (151 VGPRs, both paths' state live).
		if constexpr (TASK && LIST)
		{
			// ---- drawcull.comp.glsl:120-139, one LANE per output command (round 3).  Through round 2 a draw's commands were
			// written by its owning lane (<= 4 task groups) or wave-cooperatively, one owning draw at a time; with one scatter
			// workgroup per CU that is one wave per SIMD walking ~25 owners at single-wave latency — 18.5 us for the 175 k commands
			// of a 1 M-draw frame against 11 us for the decision itself.  Now the step's emitting draws go into an LDS list in draw
			// order and every output command finds its draw by binary search over the list's (sorted) first-command column: all
			// lanes busy whatever the draws' sizes, consecutive lanes write consecutive 20-B commands.
			const uint32_t stepTotal = running - stepBase;
#pragma unroll
			for (uint32_t j = 0; j < DC_PER_LANE; ++j)
			{
				if (cnt[j])
				{
					s_emit[eIdx] = make_uint2(dci - stepBase, (tid * DC_PER_LANE + j) | (res[j] & 7u) << 12 | (res[j] >> 4 & 1u) << 15);
					++eIdx;
				}
				dci += cnt[j];
			}
			if (tid == 0)
				s_emit[stepEmitters] = make_uint2(stepTotal, 0u);
			__syncthreads();
			NvMeshTaskCommand* tc = static_cast<NvMeshTaskCommand*>(a.commands);
			for (uint32_t k = tid; k < stepTotal; k += DC_THREADS)
			{
				// largest e with s_emit[e].x <= k (s_emit[0].x == 0; draws emit >= 1 command, so the column is strictly increasing)
				uint32_t lo = 0, hi = stepEmitters; // invariant: s_emit[lo].x <= k < s_emit[hi].x
				while (hi - lo > 1u)
				{
					const uint32_t mid = (lo + hi) >> 1;
					if (s_emit[mid].x <= k)
						lo = mid;
					else
						hi = mid;
				}
				const uint2 e = s_emit[lo];
				const uint32_t groups = s_emit[lo + 1u].x - e.x;
				const uint32_t i = k - e.x;
				const uint32_t di = first + c0 + (e.y & 0xfffu);
				const uint32_t lod = e.y >> 12 & 7u;
				// the draw's meshIndex / meshletVisibilityOffset (8 B of its record; one request per draw and wave) and its LOD's range
				const uint2 d2 = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.draws + di) + 32);
				const char* mesh = meshBase + (size_t)d2.x * sizeof(NvMesh);
				const uint32_t meshletOffset = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lod + 8);
				const uint32_t meshletCount = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lod + 12);
				if (stepBase + e.x + groups <= NV_TASK_WGLIMIT) // drop the whole draw on overflow (:128)
				{
					NvMeshTaskCommand cmd;
					cmd.drawId = di;
					cmd.taskOffset = meshletOffset + i * NV_TASK_WGSIZE;
					const uint32_t rest = meshletCount - i * NV_TASK_WGSIZE;
					cmd.taskCount = rest < NV_TASK_WGSIZE ? rest : NV_TASK_WGSIZE;
					cmd.lateDrawVisibility = e.y >> 15 & 1u;
					cmd.meshletVisibilityOffset = d2.y + i * NV_TASK_WGSIZE;
					tc[stepBase + k] = cmd;
				}
			}
			continue; // (the next step's first barrier orders the list's reuse)
		}

#pragma unroll
		for (uint32_t j = 0; j < DC_PER_LANE; ++j)
		{
			const uint32_t lodIndex = res[j] & 7u;
			if (TASK) // (!LIST: the list form left the step above)
			{
				// drawcull.comp.glsl:120-139.  Draws with a handful of task groups write their own commands (all lanes
				// in parallel); larger ones are expanded wave-cooperatively, one owning lane at a time, so that no lane
				// loops over hundreds of commands while 63 others wait.
				constexpr uint32_t DC_SMALL = 4;
				uint64_t owners = __ballot(cnt[j] > DC_SMALL);
				const uint32_t oldVis = res[j] >> 4 & 1u;
				NvMeshTaskCommand* tc = static_cast<NvMeshTaskCommand*>(a.commands);
				if (cnt[j] != 0 && cnt[j] <= DC_SMALL && dci + cnt[j] <= NV_TASK_WGLIMIT) // drop the whole draw on overflow (:128)
				{
					const char* mesh = meshBase + (size_t)meshIndex[j] * sizeof(NvMesh);
					const uint32_t meshletOffset = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lodIndex + 8);
					const uint32_t meshletCount = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lodIndex + 12);
					for (uint32_t i = 0; i < cnt[j]; ++i)
					{
						NvMeshTaskCommand cmd;
						cmd.drawId = first + c + j;
						cmd.taskOffset = meshletOffset + i * NV_TASK_WGSIZE;
						const uint32_t rest = meshletCount - i * NV_TASK_WGSIZE;
						cmd.taskCount = rest < NV_TASK_WGSIZE ? rest : NV_TASK_WGSIZE;
						cmd.lateDrawVisibility = oldVis;
						cmd.meshletVisibilityOffset = mvo[j] + i * NV_TASK_WGSIZE;
						tc[dci + i] = cmd;
					}
				}
				while (owners)
				{
					const int src = __builtin_ctzll(owners);
					owners &= owners - 1;
					const uint32_t oDraw = first + c0 + (wave * 64 + src) * DC_PER_LANE + j;
					const uint32_t oDci = __builtin_amdgcn_readlane(dci, src);
					const uint32_t oGroups = __builtin_amdgcn_readlane(cnt[j], src);
					const uint32_t oLod = __builtin_amdgcn_readlane(lodIndex, src);
					const uint32_t oVis = __builtin_amdgcn_readlane(oldVis, src);
					const uint32_t oMesh = __builtin_amdgcn_readlane(meshIndex[j], src);
					const uint32_t oMvo = __builtin_amdgcn_readlane(mvo[j], src);
					if (oDci + oGroups <= NV_TASK_WGLIMIT) // drop the whole draw on overflow (:128)
					{
						const char* mesh = meshBase + (size_t)oMesh * sizeof(NvMesh);
						const uint32_t meshletOffset = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * oLod + 8);
						const uint32_t meshletCount = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * oLod + 12);
						for (uint32_t i = lane; i < oGroups; i += 64)
						{
							NvMeshTaskCommand cmd;
							cmd.drawId = oDraw;
							cmd.taskOffset = meshletOffset + i * NV_TASK_WGSIZE;
							uint32_t rest = meshletCount - i * NV_TASK_WGSIZE;
							cmd.taskCount = rest < NV_TASK_WGSIZE ? rest : NV_TASK_WGSIZE;
							cmd.lateDrawVisibility = oVis;
							cmd.meshletVisibilityOffset = oMvo + i * NV_TASK_WGSIZE;
							tc[oDci + i] = cmd;
						}
					}
				}
			}
			else if (cnt[j])
			{
				// drawcull.comp.glsl:141-150
				const char* mesh = meshBase + (size_t)meshIndex[j] * sizeof(NvMesh);
				const uint32_t indexOffset = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lodIndex + 0);
				const uint32_t indexCount = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lodIndex + 4);
				const uint32_t vertexOffset = *reinterpret_cast<const uint32_t*>(mesh + 16);
				uint2* dc = reinterpret_cast<uint2*>(static_cast<NvMeshDrawCommand*>(a.commands) + dci);
				dc[0] = make_uint2(first + c + j, indexCount);
				dc[1] = make_uint2(1u, indexOffset);
				dc[2] = make_uint2(vertexOffset, 0u);
			}
			dci += cnt[j];
		}
	}
}

// workgroups of the decide launch: one 64-draw unit per wave until DS_MAX_BLOCKS workgroups are out, more units per wave beyond
static uint32_t decide_blocks(uint32_t drawCount)
{
	const uint32_t units = (drawCount + 63u) / 64u, want = (units + DC_WAVES - 1u) / DC_WAVES;
	return want < 1u ? 1u : (want < DS_MAX_BLOCKS ? want : DS_MAX_BLOCKS);
}

template <bool MESH_LDS, bool SOA>
static void launch_decide(hipStream_t stream, const DrawArgs& a, int late, int task)
{
	const dim3 grid(decide_blocks(a.cd.drawCount)), block(DC_THREADS);
#ifdef DS_FORCE_VISFIRST // (tools/build_variant.sh: A/B of the two early forms)
	const bool visFirst = DS_FORCE_VISFIRST;
#else
	const bool visFirst = a.visFirst != 0;
#endif
#define LAUNCH_DECIDE(L, T, V, R) hipLaunchKernelGGL((draw_decide_kernel<L, T, MESH_LDS, SOA, V, R>), grid, block, 0, stream, a)
	if (late)
	{
		if (!task)
			LAUNCH_DECIDE(true, false, false, false);
		else if (a.recordsOn)
			LAUNCH_DECIDE(true, true, false, true);
		else
			LAUNCH_DECIDE(true, true, false, false);
	}
	else if (visFirst)
	{
		if (!task)
			LAUNCH_DECIDE(false, false, true, false);
		else if (a.recordsOn)
			LAUNCH_DECIDE(false, true, true, true);
		else
			LAUNCH_DECIDE(false, true, true, false);
	}
	else
	{
		if (!task)
			LAUNCH_DECIDE(false, false, false, false);
		else if (a.recordsOn)
			LAUNCH_DECIDE(false, true, false, true);
		else
			LAUNCH_DECIDE(false, true, false, false);
	}
#undef LAUNCH_DECIDE
}

template <bool MESH_LDS>
static void launch_dc(hipStream_t stream, const DrawArgs& a, int late, int task)
{
	dim3 block(DC_THREADS);
	if (a.soaWorld)
		launch_decide<MESH_LDS, true>(stream, a, late, task);
	else
		launch_decide<MESH_LDS, false>(stream, a, late, task);
	const uint32_t t = (a.cd.drawCount + a.scatterTiles - 1) / a.scatterTiles; // scatter_tile_draws, before rounding
	const dim3 sgrid(a.scatterTiles);
#define LAUNCH_TASK_SCATTER(PER)                                                                                      \
	do                                                                                                                 \
	{                                                                                                                  \
		if (a.recordsOn)                                                                                               \
			hipLaunchKernelGGL((draw_scatter_kernel<true, MESH_LDS, 1, true, true>), sgrid, block, 0, stream, a);      \
		else if (a.taskList)                                                                                                \
			hipLaunchKernelGGL((draw_scatter_kernel<true, MESH_LDS, PER, true>), sgrid, block, 0, stream, a);          \
		else                                                                                                           \
			hipLaunchKernelGGL((draw_scatter_kernel<true, MESH_LDS, PER, false>), sgrid, block, 0, stream, a);         \
	} while (0)
	if (task)
	{
		if (t <= DC_THREADS)
			LAUNCH_TASK_SCATTER(1);
		else if (t <= DC_THREADS * 4)
			LAUNCH_TASK_SCATTER(4);
		else
			LAUNCH_TASK_SCATTER(16);
	}
	else
	{
		if (t <= DC_THREADS)
			hipLaunchKernelGGL((draw_scatter_kernel<false, MESH_LDS, 1>), sgrid, block, 0, stream, a);
		else if (t <= DC_THREADS * 4)
			hipLaunchKernelGGL((draw_scatter_kernel<false, MESH_LDS, 4>), sgrid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((draw_scatter_kernel<false, MESH_LDS, 16>), sgrid, block, 0, stream, a);
	}
}

int launch_drawcull(hipStream_t stream, const DrawArgs& a0, int late, int task)
{
	DrawArgs a = a0;
	const uint32_t units = (a.cd.drawCount + 63u) / 64u, waves = decide_blocks(a.cd.drawCount) * DC_WAVES;
	a.unitsPerWave = units ? (units + waves - 1u) / waves : 1u;
	a.tileDraws = scatter_tile_draws(a.cd.drawCount, a.scatterTiles);
	// the list form of a TASK pass reads the decide launch's records of the emitting draws: tiles of whole waves' ranges, at most one wave per thread of a tile
	const uint32_t waveDraws = a.unitsPerWave * 64u, wavesPerTile = (a.tileDraws + waveDraws - 1u) / waveDraws;
#ifdef DC_NO_RECORDS // (tools/build_variant.sh: A/B against the list form that reads result bytes, draw words and the Mesh table)
	a.recordsOn = 0u;
#else
	a.recordsOn = task && a.taskList && a.cd.drawCount && a.cd.drawCount < 0x80000000u && wavesPerTile <= (uint32_t)DC_THREADS ? 1u : 0u;
#endif
	if (a.recordsOn)
		a.tileDraws = wavesPerTile * waveDraws;
	const size_t bytes = ((size_t)a.cd.drawCount + 64u + 255u) & ~(size_t)255u;
	a.recordCounts = reinterpret_cast<uint32_t*>(a.results + bytes);
	a.records = reinterpret_cast<uint4*>(a.results + bytes + (size_t)DS_MAX_BLOCKS * DC_WAVES * sizeof(uint32_t));
	if (a.meshCount && a.meshCount <= DC_MESH_LDS)
		launch_dc<true>(stream, a, late, task);
	else
		launch_dc<false>(stream, a, late, task);
	return (int)hipGetLastError();
}

// Mirror of the decision's inputs (nv_upload_draws / nv_update_draws), draws [first, first + count): the world-space
// sphere — drawcull.comp.glsl:73-75 up to the view transform, in the reference's operation order — plus {scale, meshIndex}
// and postPass as their own streams.
__global__ __launch_bounds__(256) void draw_split_kernel(const NvMeshDraw* __restrict__ draws, const NvMesh* __restrict__ meshes, uint32_t meshCount, uint32_t first, uint32_t count,
                                                        float4* __restrict__ world, uint2* __restrict__ scaleMesh, uint32_t* __restrict__ postPass)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= count)
		return;
	const float4* p = reinterpret_cast<const float4*>(draws + first + i);
	const float4 d0 = p[0], d1 = p[1];
	const uint4 ids = *reinterpret_cast<const uint4*>(p + 2);
	// A draw the passes always skip (postPass mismatch, never visible) may carry any meshIndex: the decide kernel never reads
	// its Mesh, so the mirror must not either.  With a registered table the index is clamped — the entry of such a
	// draw is never used; a valid index is untouched.
	const uint32_t mi = meshCount && ids.x >= meshCount ? 0u : ids.x;
	const float4 cr = *reinterpret_cast<const float4*>(meshes + mi); // center.xyz, radius
	const f3 r = rotate_quat(f3{ cr.x, cr.y, cr.z }, f3{ d1.x, d1.y, d1.z }, d1.w);
	// the same three statements as sphere_center() (cullmath.h) before view_point()
	world[first + i] = make_float4(r.x * d0.w + d0.x, r.y * d0.w + d0.y, r.z * d0.w + d0.z, cr.w * d0.w);
	scaleMesh[first + i] = make_uint2(__float_as_uint(d0.w), ids.x);
	postPass[first + i] = ids.z;
}

int launch_draw_split(hipStream_t stream, const NvMeshDraw* draws, const NvMesh* meshes, uint32_t meshCount, uint32_t first, uint32_t count, float4* world, uint2* scaleMesh,
                      uint32_t* postPass)
{
	if (count)
		hipLaunchKernelGGL(draw_split_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, draws, meshes, meshCount, first, count, world, scaleMesh, postPass);
	return (int)hipGetLastError();
}

// bytes of the per-draw result scratch (padded so that the scatter kernel's 16-B loads stay in range)
// + the record counts of the decide launch's waves and one 16-byte record per draw (the TASK list form; a wave's records lie at the front of its own range)
size_t drawcull_result_bytes(uint32_t drawCount)
{
	const size_t bytes = ((size_t)drawCount + 64u + 255u) & ~(size_t)255u;
	return bytes + (size_t)DS_MAX_BLOCKS * DC_WAVES * sizeof(uint32_t) + ((size_t)drawCount + 64u) * sizeof(uint4);
}

} // namespace nv
