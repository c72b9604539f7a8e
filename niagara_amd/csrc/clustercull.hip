// clustercull.hip — per-meshlet frustum / cone / HiZ cull + ordered compaction for gfx950.
//
// Replaces src/shaders/clustercull.comp.glsl:56-149 (and the cull half of src/shaders/meshlet.task.glsl:53-149).
//
// Mapping to CDNA4 (not a translation of the one-workgroup-per-command Vulkan grid):
//   * TASK_WGSIZE = 64 = one wavefront: a MeshTaskCommand and its MeshDraw are wave-uniform, so they are fetched
//     with scalar loads into SGPRs together with CullData (kernarg); only the 12 cull bytes of each meshlet travel
//     through the vector memory path: bounds (4 x fp16, 8 B/lane = 512 B/wave) and cone (4 x s8, 4 B/lane), both
//     perfectly coalesced from the SoA mirror built by nv_upload_meshlets (the 24-B AoS records are read in place
//     when no mirror exists);
//   * a workgroup (4 waves) owns a TILE of 4*K consecutive commands; each wave issues the loads of its K commands
//     up front (K*2 independent vector loads in flight per lane), then runs the tests; a command's result is one
//     64-bit ballot held in SGPRs — no LDS staging of survivors;
//   * survivors are appended in command-major, lane-minor order through ordered.cuh (chained scan across tiles),
//     1 ticket + 2 eight-byte publishes per 256*K meshlets instead of 1 global atomic per survivor;
//   * visibility bits (late pass) are updated with <= 3 word-level atomics per wave built from the ballots
//     instead of one atomicOr/atomicAnd per lane (clustercull.comp.glsl:125-131);
//   * tests are pure predicates ANDed together, so the cheap frustum test runs first and the cone / HiZ tests are
//     skipped wave-wide when no lane survives — identical result, fewer VALU cycles.
#include "cullmath.cuh"
#include "ordered.cuh"
#include "args.cuh"

namespace nv
{

constexpr int CC_WAVES = 4;
constexpr int CC_THREADS = CC_WAVES * 64;


struct LaneData
{
	uint32_t b0, b1, cone; // center.xy | center.z,radius | cone axis xyz,cutoff
	uint32_t mvbWord;
};

struct DrawUniform
{
	f3 q;
	float qw;
	float scale;
	f3 pos;
};

struct LaneScalars
{
	f3 c;
	float r;
	f3 axis;
	float cutoff;
};

// Wave-uniform records are read through the constant address space so that they become scalar (SMEM) loads into
// SGPRs; both arrays are read-only for the duration of the kernel and the scalar cache is invalidated per dispatch.
typedef __attribute__((address_space(4))) const uint32_t* k_u32p;
typedef __attribute__((address_space(4))) const float* k_f32p;

NV_DEV DrawUniform load_draw(const NvMeshDraw* draws, uint32_t drawId)
{
	k_f32p d = (k_f32p)(uintptr_t)(draws + drawId);
	DrawUniform u;
	u.pos = { d[0], d[1], d[2] };
	u.scale = d[3];
	u.q = { d[4], d[5], d[6] };
	u.qw = d[7];
	return u;
}

NV_DEV NvMeshTaskCommand load_command(const NvMeshTaskCommand* commands, uint32_t ci)
{
	k_u32p c = (k_u32p)(uintptr_t)(commands + ci);
	return NvMeshTaskCommand{ c[0], c[1], c[2], c[3], c[4] };
}

// The reference reads meshlets[mi] for all 64 lanes and masks the result with `valid` afterwards
// (clustercull.comp.glsl:72-82).  Here invalid lanes re-read the command's first meshlet instead (taskCount == 0:
// meshlet 0), so that every load is unconditional and in range: no branch sits between a load and its use, which
// lets the compiler keep several commands' loads in flight (counted s_waitcnt vmcnt(N) instead of vmcnt(0)).
template <bool SOA>
NV_DEV LaneData load_lane(const ClusterArgs& a, uint32_t taskOffset, uint32_t taskCount, uint32_t lane)
{
	const uint32_t mi = (taskCount ? taskOffset : 0u) + (lane < taskCount ? lane : 0u);
	LaneData l;
	if (SOA)
	{
		uint2 b = a.soaBounds[mi];
		l.b0 = b.x;
		l.b1 = b.y;
		l.cone = a.soaCones[mi];
	}
	else
	{
		const uint32_t* p = reinterpret_cast<const uint32_t*>(a.meshlets + mi);
		l.b0 = p[0];
		l.b1 = p[1];
		l.cone = p[2];
	}
	l.mvbWord = 0;
	return l;
}

// clustercull.comp.glsl:72-76: centre / radius in view space
NV_DEV void lane_sphere(const NvCullData& cd, const DrawUniform& u, const LaneData& l, f3& c, float& r)
{
	f3 lc = { half_bits_to_float(l.b0 & 0xffffu), half_bits_to_float(l.b0 >> 16), half_bits_to_float(l.b1 & 0xffffu) };
	c = sphere_center(cd, lc, u.q, u.qw, u.scale, u.pos);
	r = half_bits_to_float(l.b1 >> 16) * u.scale;
}

// clustercull.comp.glsl:78-80: cone axis / cutoff
NV_DEV void lane_cone(const NvCullData& cd, const DrawUniform& u, const LaneData& l, f3& axis, float& cutoff)
{
	f3 la;
	la.x = (float)(int)(int8_t)(l.cone & 0xffu) / 127.0f;
	la.y = (float)(int)(int8_t)((l.cone >> 8) & 0xffu) / 127.0f;
	la.z = (float)(int)(int8_t)((l.cone >> 16) & 0xffu) / 127.0f;
	f3 ra = rotate_quat(la, u.q, u.qw);
	axis = view_dir(cd.view, ra);
	cutoff = (float)(int)(int8_t)(l.cone >> 24) / 127.0f;
}

// One command on one wave.  Returns the ballot of lanes that append (visible && !skip) and, for the late pass,
// applies the visibility-bit update.  All arguments except `l` are wave-uniform.
// BITS = (clusterOcclusionEnabled == 1 && postPass == 0), resolved on the host so that the variant without
// visibility bits carries no load for them.
template <bool LATE, bool BITS>
NV_DEV uint64_t cull_command(const ClusterArgs& a, const NvMeshTaskCommand& cmd, const DrawUniform& u, const LaneData& l, uint32_t lane)
{
	const NvCullData& cd = a.cd;
	const bool valid = lane < cmd.taskCount;
	const uint32_t mvi = lane + cmd.meshletVisibilityOffset;

	bool visible = valid;
	bool skip = false;

	if (BITS)
	{
		// clustercull.comp.glsl:86-99
		bool bit = (l.mvbWord & (1u << (mvi & 31))) != 0;
		if (!LATE && !bit)
			visible = false;
		if (LATE && cmd.lateDrawVisibility == 1 && bit)
			skip = true;
	}

	if (__ballot(visible) != 0)
	{
		f3 c;
		float r;
		lane_sphere(cd, u, l, c, r);
		visible = visible && frustum_test(cd, c, r);

		if (cd.clusterBackfaceEnabled != 0 && __ballot(visible) != 0)
		{
			f3 axis;
			float cutoff;
			lane_cone(cd, u, l, axis, cutoff);
			visible = visible && !cone_cull(c, r, axis, cutoff);
		}

		if (LATE && cd.clusterOcclusionEnabled == 1 && visible)
			visible = hiz_test(cd, a.pyr, c, r);
	}

	const uint64_t visMask = __ballot(visible);

	if (LATE && cd.clusterOcclusionEnabled == 1)
	{
		// clustercull.comp.glsl:125-131, as <= 3 word-level atomics per wave: lane j carries word (off>>5)+j
		const uint64_t validMask = __ballot(valid);
		const uint32_t off = cmd.meshletVisibilityOffset;
		const uint32_t sh = off & 31u;
		if (lane < 3)
		{
			// bits of word j come from lanes [32*j - sh, 32*j - sh + 32)
			int lo = 32 * (int)lane - (int)sh;
			uint64_t setAll = visMask & validMask, clrAll = ~visMask & validMask;
			uint32_t setw, clrw;
			if (lo >= 0)
			{
				setw = lo < 64 ? (uint32_t)(setAll >> lo) : 0u;
				clrw = lo < 64 ? (uint32_t)(clrAll >> lo) : 0u;
			}
			else
			{
				setw = (uint32_t)(setAll << (-lo));
				clrw = (uint32_t)(clrAll << (-lo));
			}
			uint32_t* word = a.mvb + (off >> 5) + lane;
			if (clrw)
				atomicAnd(word, ~clrw);
			if (setw)
				atomicOr(word, setw);
		}
	}

	return __ballot(visible && !skip);
}

NV_DEV uint32_t load_mvb_word(const ClusterArgs& a, uint32_t meshletVisibilityOffset, uint32_t taskCount, uint32_t lane)
{
	const uint32_t mvi = taskCount ? meshletVisibilityOffset + (lane < taskCount ? lane : 0u) : 0u;
	// late pass: other waves update neighbouring bits of shared words concurrently; an agent-scope load keeps the
	// read out of a stale L1 line (our own bit is only ever written by this lane, later)
	return __hip_atomic_load(a.mvb + (mvi >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

NV_DEV uint32_t indirect_command_count(const ClusterArgs& a)
{
	// vkCmdDispatchIndirect(dccb, 4): grid = (groupCountX, 64, 1), commandId = x*64 + y (clustercull.comp.glsl:59)
	return a.commandCountOverride ? a.commandCountOverride : a.count4[1] * 64u;
}

// ---------------------------------------------------------------------------------------------------------------
// ordered cluster append (clustercull.comp.glsl)
//
// Static tiles: the grid is a fixed, co-resident set of G workgroups (context.hip sizes it to 4 per CU); tile t covers
// commands [t*T, (t+1)*T) with T = ceil(numCmds / G) (device-computed from the indirect words, clamped to CC_TMAX),
// so a pass is ONE tile per workgroup and the chained scan runs once per workgroup at the tail, after its whole
// range has been culled: phase 1 cull -> 64-bit ballots in LDS, phase 2 tile total, phase 3 look-back across tiles,
// phase 4 ordered scatter from the LDS ballots.  No tickets, no per-tile atomics.
constexpr uint32_t CC_TMAX = 1024; // commands per tile: 8 KiB of ballots in LDS
constexpr int CC_D = 6;            // ring slots per wave: CC_D - 1 commands' meshlet loads in flight behind the one being tested

// lane l of a wave holds the l-th command of the wave's current 64-command segment (one coalesced 1280-B read
// instead of 64 dependent scalar loads) and the MeshDraw it points at; fields are broadcast with v_readlane as the
// wave walks the segment, so the walk itself contains no scalar-memory wait.
struct SegmentRegs
{
	uint32_t drawId, taskOffset, taskCount, lateDrawVisibility, meshletVisibilityOffset;
	float4 d0, d1; // position.xyz, scale | orientation.xyzw of draws[drawId]
};

NV_DEV NvMeshTaskCommand segment_command(const SegmentRegs& r, uint32_t c)
{
	NvMeshTaskCommand cmd;
	cmd.drawId = (uint32_t)__builtin_amdgcn_readlane(r.drawId, c);
	cmd.taskOffset = (uint32_t)__builtin_amdgcn_readlane(r.taskOffset, c);
	cmd.taskCount = (uint32_t)__builtin_amdgcn_readlane(r.taskCount, c);
	cmd.lateDrawVisibility = (uint32_t)__builtin_amdgcn_readlane(r.lateDrawVisibility, c);
	cmd.meshletVisibilityOffset = (uint32_t)__builtin_amdgcn_readlane(r.meshletVisibilityOffset, c);
	return cmd;
}

NV_DEV float readlane_f(float v, uint32_t c) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), c)); }

NV_DEV DrawUniform segment_draw(const SegmentRegs& r, uint32_t c)
{
	DrawUniform u;
	u.pos = { readlane_f(r.d0.x, c), readlane_f(r.d0.y, c), readlane_f(r.d0.z, c) };
	u.scale = readlane_f(r.d0.w, c);
	u.q = { readlane_f(r.d1.x, c), readlane_f(r.d1.y, c), readlane_f(r.d1.z, c) };
	u.qw = readlane_f(r.d1.w, c);
	return u;
}

// ---- software-pipelined meshlet stream (SoA mirror only)
// hipcc's s_waitcnt insertion collapses a loop-carried prefetch ring to (almost) vmcnt(0): measured, every command then
// costs one full memory latency.  The ring's loads are therefore issued from inline asm, which hipcc does not count,
// and waited for by hand with a counted vmcnt (cdna_hip_programming.md §5.7, form (ii): "=v" loads, then a wait
// statement that names every destination "+v" before its first consumer).  Rules that keep this safe:
//   * a slot's registers are touched by nothing but its issue / wait statements until the wait has passed;
//   * slots are reissued in a fixed rotation with unconditional loads (indices clamped, never branched), so exactly
//     (CC_D - 1) * LOADS younger ring loads are outstanding at every wait;
//   * VMEM operations hipcc issues itself in between (HiZ texels, visibility-bit atomics) are younger than the slot
//     being waited for and are consumed before the next ring issue (the issue statement takes the command's ballot as
//     an operand), so they can only make a wait stricter, never too weak.
struct RingSlot
{
	uint64_t bounds; // center.xy | center.z, radius (4 x fp16)
	uint32_t cone;
	uint32_t mvbWord;
};

template <bool BITS>
NV_DEV void ring_issue(RingSlot& s, const ClusterArgs& a, uint32_t taskOffset, uint32_t taskCount, uint32_t mvo, uint32_t lane, uint64_t order)
{
	const uint32_t li = lane < taskCount ? lane : 0u;
	const uint32_t mi = (taskCount ? taskOffset : 0u) + li;
	const uint32_t off8 = mi * 8u, off4 = mi * 4u;
	if (BITS)
	{
		const uint32_t offw = taskCount ? ((mvo + li) >> 5) * 4u : 0u;
		asm volatile("global_load_dwordx2 %0, %3, %4\n\tglobal_load_dword %1, %5, %6\n\tglobal_load_dword %2, %7, %8 sc1"
		             : "=&v"(s.bounds), "=&v"(s.cone), "=&v"(s.mvbWord)
		             : "v"(off8), "s"(a.soaBounds), "v"(off4), "s"(a.soaCones), "v"(offw), "s"(a.mvb), "s"(order)
		             : "memory");
	}
	else
	{
		asm volatile("global_load_dwordx2 %0, %2, %3\n\tglobal_load_dword %1, %4, %5"
		             : "=&v"(s.bounds), "=&v"(s.cone)
		             : "v"(off8), "s"(a.soaBounds), "v"(off4), "s"(a.soaCones), "s"(order)
		             : "memory");
		s.mvbWord = 0;
	}
}

template <bool BITS, int YOUNGER>
NV_DEV void ring_wait(RingSlot& s)
{
	if (BITS)
		asm volatile("s_waitcnt vmcnt(%3)" : "+v"(s.bounds), "+v"(s.cone), "+v"(s.mvbWord) : "i"(YOUNGER * 3) : "memory");
	else
		asm volatile("s_waitcnt vmcnt(%2)" : "+v"(s.bounds), "+v"(s.cone) : "i"(YOUNGER * 2) : "memory");
}

template <bool LATE, bool SOA, bool BITS>
__global__ __launch_bounds__(CC_THREADS) void clustercull_kernel(ClusterArgs a)
{
	__shared__ uint64_t s_mask[CC_TMAX];
	__shared__ uint32_t s_part[CC_WAVES];
	__shared__ uint32_t s_scratch[16];

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t G = gridDim.x;

	const uint32_t numCmds = indirect_command_count(a);
	uint32_t T = (numCmds + G - 1) / G;
	T = T < 1 ? 1 : (T > CC_TMAX ? CC_TMAX : T);
	const uint32_t numTiles = (numCmds + T - 1) / T;
	const uint32_t epoch = load_epoch(a.ctl);
	const uint32_t base0 = a.clusterCount4[0];
	const bool dbgNoScan = a.debugMode & 1u, dbgNoScatter = a.debugMode & 4u; // experiments only
	// debugMode bit 3: per-wave s_memtime stamps into probeOut (tools/wave_timeline.py); never set in production
	const bool dbgTime = (a.debugMode & 8u) && a.probeOut;
	unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.probeOut) + (size_t)(blockIdx.x * CC_WAVES + wave) * 8;
#define NV_STAMP(i) do { if (dbgTime && lane == 0) stamps[i] = __builtin_readcyclecounter(); } while (0)
	NV_STAMP(0);

	for (uint32_t tile = blockIdx.x; tile < numTiles; tile += G)
	{
		const uint32_t first = tile * T;
		const uint32_t n = numCmds - first < T ? numCmds - first : T;

		// ---- phase 1: each wave walks a contiguous quarter of the tile, CC_D commands' meshlet loads in flight
		const uint32_t cw = (n + CC_WAVES - 1) / CC_WAVES;
		const uint32_t wbeg = wave * cw < n ? wave * cw : n;
		const uint32_t wend = wbeg + cw < n ? wbeg + cw : n;
		uint32_t waveCount = 0;

		for (uint32_t seg = wbeg; seg < wend; seg += 64)
		{
			const uint32_t cnt = wend - seg < 64u ? wend - seg : 64u;

			SegmentRegs r = {};
			if (lane < cnt)
			{
				const uint32_t* p = reinterpret_cast<const uint32_t*>(a.commands + first + seg + lane);
				r.drawId = p[0];
				r.taskOffset = p[1];
				r.taskCount = p[2];
				r.lateDrawVisibility = p[3];
				r.meshletVisibilityOffset = p[4];
				if (r.taskCount)
				{
					const float4* d = reinterpret_cast<const float4*>(a.draws + r.drawId);
					r.d0 = d[0];
					r.d1 = d[1];
				}
			}

			uint32_t curDraw = ~0u;
			DrawUniform du = {};
			NV_STAMP(1);

			// body of one command, shared by both load paths
			auto run_command = [&](uint32_t c, const LaneData& cur) -> uint64_t
			{
				const NvMeshTaskCommand cmd = segment_command(r, c);
				uint64_t m = 0;
				if (cmd.taskCount)
				{
					if (cmd.drawId != curDraw) // a draw's task commands are consecutive: usually a hit
					{
						curDraw = cmd.drawId;
						du = segment_draw(r, c);
					}
					m = cull_command<LATE, BITS>(a, cmd, du, cur, lane);
				}
				if (lane == 0)
					s_mask[seg + c] = m;
				waveCount += (uint32_t)__builtin_popcountll(m);
				return m;
			};

			if (SOA)
			{
				// make sure hipcc has waited for its own segment loads before the first uncounted load is issued
				asm volatile("" : "+v"(r.d0.x), "+v"(r.d1.x), "+v"(r.taskOffset), "+v"(r.meshletVisibilityOffset));

				// indices past the segment are clamped to its last command: redundant but unconditional loads
				RingSlot ring[CC_D];
#pragma unroll
				for (int k = 0; k < CC_D; ++k)
				{
					const uint32_t c = (uint32_t)k < cnt ? k : cnt - 1;
					ring_issue<BITS>(ring[k], a, __builtin_amdgcn_readlane(r.taskOffset, c), __builtin_amdgcn_readlane(r.taskCount, c),
					                 __builtin_amdgcn_readlane(r.meshletVisibilityOffset, c), lane, 0);
				}
				NV_STAMP(2);
				for (uint32_t i = 0; i < cnt; i += CC_D)
				{
#pragma unroll
					for (int k = 0; k < CC_D; ++k)
					{
						const uint32_t c = i + k;
						ring_wait<BITS, CC_D - 1>(ring[k]);
						if (i == 0 && k == 0)
							NV_STAMP(3);
						uint64_t m = 0;
						if (c < cnt)
						{
							LaneData cur;
							cur.b0 = (uint32_t)ring[k].bounds;
							cur.b1 = (uint32_t)(ring[k].bounds >> 32);
							cur.cone = ring[k].cone;
							cur.mvbWord = ring[k].mvbWord;
							m = run_command(c, cur);
						}
						const uint32_t cn = c + CC_D < cnt ? c + CC_D : cnt - 1;
						ring_issue<BITS>(ring[k], a, __builtin_amdgcn_readlane(r.taskOffset, cn), __builtin_amdgcn_readlane(r.taskCount, cn),
						                 __builtin_amdgcn_readlane(r.meshletVisibilityOffset, cn), lane, m);
					}
				}
				// drain: nothing of the ring may be in flight when the registers are reused
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
			else
			{
				// AoS records read in place: compiler-scheduled loads, one command ahead
				LaneData nxt = load_lane<false>(a, __builtin_amdgcn_readlane(r.taskOffset, 0), __builtin_amdgcn_readlane(r.taskCount, 0), lane);
				if (BITS)
					nxt.mvbWord = load_mvb_word(a, __builtin_amdgcn_readlane(r.meshletVisibilityOffset, 0), __builtin_amdgcn_readlane(r.taskCount, 0), lane);
				for (uint32_t c = 0; c < cnt; ++c)
				{
					const LaneData cur = nxt;
					const uint32_t cn = c + 1 < cnt ? c + 1 : cnt - 1;
					nxt = load_lane<false>(a, __builtin_amdgcn_readlane(r.taskOffset, cn), __builtin_amdgcn_readlane(r.taskCount, cn), lane);
					if (BITS)
						nxt.mvbWord = load_mvb_word(a, __builtin_amdgcn_readlane(r.meshletVisibilityOffset, cn), __builtin_amdgcn_readlane(r.taskCount, cn), lane);
					run_command(c, cur);
				}
			}
		}

		// ---- phase 2 + 3: tile total, chained scan across tiles
		NV_STAMP(4);
		if (lane == 0)
			s_part[wave] = waveCount;
		__syncthreads();
		NV_STAMP(5);
		uint32_t aggregate = 0;
#pragma unroll
		for (int w = 0; w < CC_WAVES; ++w)
			aggregate += s_part[w];
		const uint32_t exclusive = dbgNoScan ? 0u : lookback_exclusive(a.state, a.ctl, tile, epoch, aggregate, base0, s_scratch);
		if (tid == 0 && tile == numTiles - 1)
		{
			a.clusterCount4[0] = exclusive + aggregate; // what the chain of atomicAdds leaves in clusterCount
			advance_epoch(a.ctl, epoch);                 // every tile has published: nobody polls any more
		}
		__syncthreads(); // s_part is reused by the scatter

		NV_STAMP(6);
		// ---- phase 4: ordered scatter, 256 commands per step, one command per lane for the scan and one command
		// per iteration for the (coalesced) stores; clustercull.comp.glsl:137-138 drops entries past CLUSTER_LIMIT
		uint32_t running = exclusive;
		for (uint32_t c0 = 0; c0 < n; c0 += CC_THREADS)
		{
			const uint32_t c = c0 + tid;
			const uint64_t m = c < n ? s_mask[c] : 0ull;
			const uint32_t pc = (uint32_t)__builtin_popcountll(m);
			uint32_t incl = pc;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1)
			{
				uint32_t t = __shfl_up(incl, o, 64);
				if ((int)lane >= o)
					incl += t;
			}
			__syncthreads(); // previous step's s_part readers are done
			if (lane == 63)
				s_part[wave] = incl;
			__syncthreads();
			uint32_t waveBase = running;
#pragma unroll
			for (int w = 0; w < CC_WAVES; ++w)
			{
				uint32_t p = s_part[w];
				waveBase += w < (int)wave ? p : 0u;
				running += p;
			}
			const uint32_t excl = waveBase + incl - pc;

			uint64_t owners = dbgNoScatter ? 0ull : __ballot(pc != 0);
			while (owners)
			{
				const int src = __builtin_ctzll(owners);
				owners &= owners - 1;
				const uint32_t mlo = __builtin_amdgcn_readlane((uint32_t)m, src);
				const uint32_t mhi = __builtin_amdgcn_readlane((uint32_t)(m >> 32), src);
				const uint32_t off = __builtin_amdgcn_readlane(excl, src);
				const uint64_t ms = ((uint64_t)mhi << 32) | mlo;
				if (ms >> lane & 1ull)
				{
					uint32_t rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
					uint32_t index = off + rank;
					if (index < NV_CLUSTER_LIMIT)
						a.clusterIndices[index] = (first + c0 + wave * 64 + src) | (lane << 24);
				}
			}
		}
		__syncthreads(); // s_mask is rewritten by the next tile
		NV_STAMP(7);

		if (tile == numTiles - 1 && ((epoch + 1) & 0x3fffffffu) == 0)
			for (uint32_t i = tid; i < a.stateCapacity; i += CC_THREADS) // epoch wrapped: drop 2^30-launch-old tags
				a.state[i] = 0;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// task-shader form (meshlet.task.glsl:135-143): survivors compacted per command into its 64-entry payload
template <bool LATE, bool SOA>
__global__ __launch_bounds__(CC_THREADS) void taskcull_kernel(ClusterArgs a)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t numCmds = indirect_command_count(a);

	for (uint32_t ci = blockIdx.x * CC_WAVES + wave; ci < numCmds; ci += gridDim.x * CC_WAVES)
	{
		const NvMeshTaskCommand cmd = load_command(a.commands, __builtin_amdgcn_readfirstlane(ci));
		LaneData ld = load_lane<SOA>(a, cmd.taskOffset, cmd.taskCount, lane);
		const bool bits = a.cd.clusterOcclusionEnabled == 1 && a.cd.postPass == 0;
		if (bits)
			ld.mvbWord = load_mvb_word(a, cmd.meshletVisibilityOffset, cmd.taskCount, lane);
		const DrawUniform du = load_draw(a.draws, cmd.drawId);
		uint64_t m = 0;
		if (cmd.taskCount)
			m = bits ? cull_command<LATE, true>(a, cmd, du, ld, lane) : cull_command<LATE, false>(a, cmd, du, ld, lane);
		if (m >> lane & 1ull)
		{
			uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			a.clusterIndices[(size_t)ci * 64 + rank] = ci | (lane << 24);
		}
		if (lane == 0)
			a.payloadCounts[ci] = (uint32_t)__builtin_popcountll(m);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// verification probe: the same device functions, intermediates written out (16 floats per lane)
template <bool SOA>
__global__ __launch_bounds__(CC_THREADS) void probe_kernel(ClusterArgs a)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t numCmds = a.commandCountOverride;
	const NvCullData& cd = a.cd;

	for (uint32_t ci = blockIdx.x * CC_WAVES + wave; ci < numCmds; ci += gridDim.x * CC_WAVES)
	{
		const NvMeshTaskCommand cmd = load_command(a.commands, __builtin_amdgcn_readfirstlane(ci));
		LaneData ld = load_lane<SOA>(a, cmd.taskOffset, 64u, lane);
		const DrawUniform du = load_draw(a.draws, cmd.drawId);

		f3 c, axis;
		float r, cutoff;
		lane_sphere(cd, du, ld, c, r);
		lane_cone(cd, du, ld, axis, cutoff);

		float o[16];
#pragma unroll
		for (int i = 0; i < 16; ++i)
			o[i] = 0.0f;
		o[0] = c.x;
		o[1] = c.y;
		o[2] = c.z;
		o[3] = r;
		o[4] = dot3(c, axis);
		o[5] = cutoff * length3(c) + r;
		float aabb[4];
		bool proj = project_sphere(c, r, cd.znear, cd.P00, cd.P11, aabb);
		if (proj)
		{
			o[6] = aabb[0];
			o[7] = aabb[1];
			o[8] = aabb[2];
			o[9] = aabb[3];
			o[10] = occlusion_mip(aabb, cd.pyramidWidth, cd.pyramidHeight);
			if (a.pyr.d_base)
				o[11] = sample_min(a.pyr, (aabb[0] + aabb[2]) * 0.5f, (aabb[1] + aabb[3]) * 0.5f, o[10]);
			o[12] = cd.znear / (c.z - r);
		}
		o[13] = proj ? 1.0f : 0.0f;
		o[14] = frustum_test(cd, c, r) ? 1.0f : 0.0f;
		o[15] = cone_cull(c, r, axis, cutoff) ? 1.0f : 0.0f;

		float4* dst = reinterpret_cast<float4*>(a.probeOut + ((size_t)ci * 64 + lane) * 16);
		dst[0] = make_float4(o[0], o[1], o[2], o[3]);
		dst[1] = make_float4(o[4], o[5], o[6], o[7]);
		dst[2] = make_float4(o[8], o[9], o[10], o[11]);
		dst[3] = make_float4(o[12], o[13], o[14], o[15]);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// SoA mirror of the 12 cull bytes (nv_upload_meshlets)
__global__ __launch_bounds__(256) void soa_split_kernel(const NvMeshlet* __restrict__ meshlets, uint32_t count, uint32_t padded,
                                                       uint2* __restrict__ bounds, uint32_t* __restrict__ cones)
{
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= padded)
		return;
	uint2 b = make_uint2(0, 0);
	uint32_t c = 0;
	if (i < count)
	{
		const uint32_t* p = reinterpret_cast<const uint32_t*>(meshlets + i);
		b.x = p[0];
		b.y = p[1];
		c = p[2];
	}
	bounds[i] = b;
	cones[i] = c;
}

// ---------------------------------------------------------------------------------------------------------------
// launchers (called from context.hip)

template <bool LATE, bool SOA>
static void launch_cc(hipStream_t stream, const ClusterArgs& a, uint32_t gridBlocks)
{
	dim3 grid(gridBlocks), block(CC_THREADS);
	if (a.cd.clusterOcclusionEnabled == 1 && a.cd.postPass == 0)
		hipLaunchKernelGGL((clustercull_kernel<LATE, SOA, true>), grid, block, 0, stream, a);
	else
		hipLaunchKernelGGL((clustercull_kernel<LATE, SOA, false>), grid, block, 0, stream, a);
}

int launch_clustercull(hipStream_t stream, const ClusterArgs& a, int late, bool soa, uint32_t gridBlocks)
{
	if (late)
	{
		if (soa)
			launch_cc<true, true>(stream, a, gridBlocks);
		else
			launch_cc<true, false>(stream, a, gridBlocks);
	}
	else
	{
		if (soa)
			launch_cc<false, true>(stream, a, gridBlocks);
		else
			launch_cc<false, false>(stream, a, gridBlocks);
	}
	return (int)hipGetLastError();
}

uint32_t clustercull_max_tiles(uint32_t gridBlocks) { return (gridBlocks > NV_TASK_WGLIMIT / CC_TMAX ? gridBlocks : NV_TASK_WGLIMIT / CC_TMAX) + 2; }

int launch_taskcull(hipStream_t stream, const ClusterArgs& a, int late, bool soa, uint32_t gridBlocks)
{
	dim3 grid(gridBlocks), block(CC_THREADS);
	if (late)
	{
		if (soa)
			hipLaunchKernelGGL((taskcull_kernel<true, true>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((taskcull_kernel<true, false>), grid, block, 0, stream, a);
	}
	else
	{
		if (soa)
			hipLaunchKernelGGL((taskcull_kernel<false, true>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((taskcull_kernel<false, false>), grid, block, 0, stream, a);
	}
	return (int)hipGetLastError();
}

int launch_probe(hipStream_t stream, const ClusterArgs& a, bool soa, uint32_t gridBlocks)
{
	dim3 grid(gridBlocks), block(CC_THREADS);
	if (soa)
		hipLaunchKernelGGL((probe_kernel<true>), grid, block, 0, stream, a);
	else
		hipLaunchKernelGGL((probe_kernel<false>), grid, block, 0, stream, a);
	return (int)hipGetLastError();
}

int launch_soa_split(hipStream_t stream, const NvMeshlet* meshlets, uint32_t count, uint32_t padded, uint2* bounds, uint32_t* cones)
{
	hipLaunchKernelGGL(soa_split_kernel, dim3((padded + 255) / 256), dim3(256), 0, stream, meshlets, count, padded, bounds, cones);
	return (int)hipGetLastError();
}

} // namespace nv
