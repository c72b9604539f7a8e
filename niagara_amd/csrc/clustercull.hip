// clustercull.hip — per-meshlet frustum / cone / HiZ cull + ordered compaction for gfx950.
//
// Replaces src/shaders/clustercull.comp.glsl:56-149 (and the cull half of src/shaders/meshlet.task.glsl:53-149).
//
// Mapping to CDNA4 (not a translation of the one-workgroup-per-command Vulkan grid):
//   * TASK_WGSIZE = 64 = one wavefront, so a MeshTaskCommand and its MeshDraw are wave-uniform.  A wave fetches 64 of
//     its commands with one coalesced vector load (lane = command), gathers their MeshDraws the same way and derives
//     the per-draw quantities lane-parallel; fields are broadcast with v_readlane where a command is visited.  Only the
//     12 cull bytes of a meshlet travel per lane: bounds (4 x fp16, 512 B per wave) and cone (4 x s8), both perfectly
//     coalesced from the SoA mirror built by nv_upload_meshlets (the 24-B AoS records are read in place otherwise);
//   * two launches, no workgroup ever waits on another (scheme: args.h).  K1 cluster_mask_kernel is a pure map dealt for
//     balance: every command becomes one 64-bit ballot, and the survivors per scatter tile are counted on the way;
//     K2 cluster_scatter_kernel owns contiguous command ranges and stores the IDs in command-major, lane-minor order;
//   * K1 works in two passes per 64-command segment.  Pass A streams the 8 bounds bytes through a conservative frustum
//     filter (23 VALU per command) with CC_DA commands' loads in flight (counted vmcnt waits); pass B visits only the
//     commands that may have survivors: a certified two-sided test (frustum + cone evaluated through one affine map
//     per draw with proven error margins: ~60 VALU) decides almost all of them, and the reference's own un-fused
//     arithmetic (~190 VALU) runs for a command only when some lane falls inside a margin — the result is the
//     reference's for every lane either way;
//   * visibility bits (late pass) are updated lane-parallel per segment from the ballots — <= 3 words per command, a
//     plain store for a word the command owns, an atomic only for a shared edge word — instead of one atomicOr/And per
//     lane (clustercull.comp.glsl:125-131);
//   * the late pass with HiZ is three launches: K1 in its early form (DEFER: frustum / cone ballots, the commands that have
//     survivors listed), cluster_hiz_kernel with ONE LANE PER SURVIVOR for the occlusion probe, the visibility bits and the
//     final ballots, then K2.  Inside K1 the probe cost three dependent memory latencies per command with only the
//     command's survivors active, and the launch ended with the few waves that drew the visible part of the scene.
#include <type_traits>

#include "cullmath.h"
#include "args.h"
#include "filtermath.h"
#include "dealing.h"

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
	return NvMeshTaskCommand{ c[0], c[1], c[2] < 64u ? c[2] : 64u, c[3], c[4] }; // (taskCount above TASK_WGSIZE: the reference's invocations are lanes 0 .. 63, clustercull.comp.glsl:66-70)
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

// int8 / 127.0 exactly as IEEE division rounds it, in 3 instructions instead of the ~10 of a full-range fp32 division:
// q = k * c with c = RN(1/127), one residual correction q' = fma(fma(-127, q, k), c, q).  Verified against the division
// for all 256 inputs (tests/test_host_helpers.py::test_s8_over_127_is_exact does the same arithmetic in exact rationals;
// the GPU parity tests exercise every int8 value through the cone axes).
NV_DEV float s8_over_127(uint32_t byte)
{
	const float k = (float)(int)(int8_t)byte;
	const float c = 0.00787401571869850158691406250f; // 0x3c010204
	const float q = k * c;
	const float r = __builtin_fmaf(-127.0f, q, k);
	return __builtin_fmaf(r, c, q);
}

// clustercull.comp.glsl:78-80: cone axis / cutoff
NV_DEV void lane_cone(const NvCullData& cd, const DrawUniform& u, const LaneData& l, f3& axis, float& cutoff)
{
	f3 la;
	la.x = s8_over_127(l.cone & 0xffu);
	la.y = s8_over_127((l.cone >> 8) & 0xffu);
	la.z = s8_over_127((l.cone >> 16) & 0xffu);
	f3 ra = rotate_quat(la, u.q, u.qw);
	axis = view_dir(cd.view, ra);
	cutoff = s8_over_127(l.cone >> 24);
}

// Visibility-bit update of one command (clustercull.comp.glsl:125-131): lanes in setAll become 1, lanes in clrAll become 0.
// Instead of one atomicOr/atomicAnd per lane, lane j < 3 handles word (offset >> 5) + j:
//   * bits that already have the wanted value are left alone (needs the loaded words, BITS);
//   * a word that lies entirely inside this command's slot range has no other writer: plain store of the new value;
//   * edge words shared with a neighbouring command: one atomicAnd / atomicOr with only the bits that change.
template <bool BITS>
NV_DEV void update_visibility_words(const ClusterArgs& a, uint32_t off, uint32_t taskCount, const LaneData& l, uint32_t lane, uint64_t setAll,
                                    uint64_t clrAll)
{
	if ((setAll | clrAll) == 0)
		return;
	const uint32_t sh = off & 31u;
	// word j's old value sits in any lane whose slot falls into it: the first such lane is max(0, 32 j - sh)
	const int lo = 32 * (int)(lane < 3 ? lane : 0u) - (int)sh;
	uint32_t old = 0;
	if (BITS)
		old = __shfl(l.mvbWord, lo > 0 ? (lo < 64 ? lo : 63) : 0, 64);
	if (lane < 3)
	{
		// bits of word j come from lanes [32 j - sh, 32 j - sh + 32)
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
		if (BITS)
		{
			setw &= ~old; // only bits that change
			clrw &= old;
			const bool owned = lo >= 0 && (uint32_t)lo + 32u <= taskCount;
			if (owned)
			{
				if (setw | clrw)
					*word = (old & ~clrw) | setw;
				return;
			}
		}
		if (clrw)
			atomicAnd(word, ~clrw);
		if (setw)
			atomicOr(word, setw);
	}
}

// One command on one wave.  Returns the ballot of lanes that append (visible && !skip) and, for the late pass,
// applies the visibility-bit update (or hands the visible ballot to the caller through visOut).  All arguments except `l` are wave-uniform.
// BITS = (clusterOcclusionEnabled == 1 && postPass == 0), resolved on the host so that the variant without
// visibility bits carries no load for them.
template <bool LATE, bool BITS, bool TABLE = false>
NV_DEV uint64_t cull_command(const ClusterArgs& a, const NvMeshTaskCommand& cmd, const DrawUniform& u, const LaneData& l, uint32_t lane,
                             uint64_t* visOut = nullptr, const uint32_t* mipOffsets = nullptr)
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

		if (LATE && cd.clusterOcclusionEnabled == 1 && visible && !NV_DBG(a, 512u)) // bit 9 (experiments): no HiZ
			visible = hiz_test<TABLE>(cd, a.pyr, c, r, mipOffsets); // TABLE: the pyramid's level offsets from LDS
	}

	const uint64_t visMask = __ballot(visible);

	if (visOut)
		*visOut = visMask; // the caller applies the visibility-bit update for its whole segment (update_segment_visibility)
	else if (LATE && cd.clusterOcclusionEnabled == 1)
	{
		const uint64_t validMask = __ballot(valid);
		update_visibility_words<BITS>(a, cmd.meshletVisibilityOffset, cmd.taskCount, l, lane, visMask & validMask, ~visMask & validMask);
	}

	return __ballot(visible && !skip);
}

// Late pass, lane-parallel form of the same update for a whole segment: lane c owns the segment's c-th command and
// holds its `visible` ballot in vis (0 for commands the filter rejected).  Runs after the load rings have drained, so
// its stores and atomics never sit between counted loads (a store issued inside a ring lengthens every s_waitcnt
// vmcnt(N) behind it by its own latency).  The <= 3 words per command are re-read here: they were fetched by this wave
// moments ago, and only this command's own bits of them are used.
NV_DEV void update_segment_visibility(const ClusterArgs& a, bool active, uint32_t off, uint32_t taskCount, uint64_t vis)
{
	if (!active || taskCount == 0)
		return;
	const uint64_t valid = taskCount >= 64u ? ~0ull : (1ull << taskCount) - 1ull;
	const uint64_t setAll = vis & valid, clrAll = valid & ~vis;
	const uint32_t sh = off & 31u;
	uint32_t* words = a.mvb + (off >> 5);
#pragma unroll
	for (int j = 0; j < 3; ++j)
	{
		const int lo = 32 * j - (int)sh; // bits of word j come from lanes [lo, lo + 32)
		if (lo >= (int)taskCount)
			break;
		uint32_t setw, clrw;
		if (lo >= 0)
		{
			setw = (uint32_t)(setAll >> lo);
			clrw = (uint32_t)(clrAll >> lo);
		}
		else
		{
			setw = (uint32_t)(setAll << (-lo));
			clrw = (uint32_t)(clrAll << (-lo));
		}
		const uint32_t old = words[j];
		setw &= ~old; // only bits that change
		clrw &= old;
		if ((setw | clrw) == 0)
			continue;
		if (lo >= 0 && (uint32_t)lo + 32u <= taskCount) // the whole word belongs to this command: no other writer
			words[j] = (old & ~clrw) | setw;
		else
		{
			if (clrw)
				atomicAnd(words + j, ~clrw);
			if (setw)
				atomicOr(words + j, setw);
		}
	}
}

NV_DEV uint32_t load_mvb_word(const ClusterArgs& a, uint32_t meshletVisibilityOffset, uint32_t taskCount, uint32_t lane)
{
	const uint32_t mvi = taskCount ? meshletVisibilityOffset + (lane < taskCount ? lane : 0u) : 0u;
	// Other waves update neighbouring bits of shared words concurrently (late pass), but only this lane's own bit is
	// extracted from the word and nobody else writes it during the pass, so a plain (possibly L1-served) load is exact.
	return a.mvb[mvi >> 5];
}

NV_DEV uint32_t indirect_command_count(const ClusterArgs& a)
{
	// vkCmdDispatchIndirect(dccb, 4): grid = (groupCountX, 64, 1), commandId = x*64 + y (clustercull.comp.glsl:59).
	// groupCountX is at most 65535 in a valid dispatch (tasksubmit.comp.glsl:36 clamps to it; maxComputeWorkGroupCount);
	// a larger word is clamped rather than trusted, because the ballot scratch is sized for TASK_WGLIMIT commands
	const uint32_t raw = load_uniform_u32(a.count4 + 1);
	const uint32_t groups = raw < 65535u ? raw : 65535u;
	return a.commandCountOverride ? a.commandCountOverride : groups * 64u;
}

// ---------------------------------------------------------------------------------------------------------------
// ordered cluster append (clustercull.comp.glsl) = two launches on the stream:
//
//   K1 cluster_mask_kernel    cull every command -> one 64-bit ballot per command in a scratch array (8 B/command).
//                             A pure map, so work is dealt out for BALANCE, not for order: commands are cut into chunks
//                             of CC_CHUNK and wave w takes chunks w, w+W, w+2W, ...  (a tile-per-workgroup assignment
//                             measured 2.5x spread between the fastest and slowest wave, because the commands of a
//                             visible draw run the cone test on top of the frustum test and visible draws cluster).
//                             Besides the ballot, K1 adds each command's survivor count to the counter of the scatter
//                             tile the command falls in (one counter per 64-byte line, quad-aggregated adds).
//   K2 cluster_scatter_kernel one workgroup per scatter tile (a contiguous range of commands, <= 512 tiles): its append
//                             base = the count word as K1 found it + the counters of all earlier tiles, so no workgroup
//                             waits on another; one scan over the tile's ballots, then the IDs are stored in
//                             command-major, lane-minor order.  The last tile writes the final count (and, on request,
//                             clustersubmit's words and the all-reduce payload).
//
// The ballots cost 8 B per 64 meshlets of extra traffic (1 %), the extra launch boundary ~4 us.
constexpr uint32_t CC_CHUNK = 4; // consecutive commands per dealt chunk (early pass)
// Late pass: a candidate command costs ~3 k cycles there (certified test + the reference's sphere + the HiZ probe), so the
// launch ends with the waves that drew the most candidates.  Dealing smaller chunks spreads a draw's commands over more
// waves, but measured slower: 42.0 us with 4, 45.3 with 2, 49.7 with 1 (config 4) — every command then starts a new draw
// in pass A (16 v_readlane + 4 moves per command) and the command loads stop coalescing.
#ifndef NV_CC_CHUNK_LATE
#define NV_CC_CHUNK_LATE 4
#endif
constexpr uint32_t CC_CHUNK_LATE = NV_CC_CHUNK_LATE;
// CC_DA (template parameter of the cull kernel) = ring slots of the filter pass: CC_DA - 1 commands' bounds in flight behind
// the one being filtered.  8 keep HBM saturated through the segment boundaries of a long stream (100 M meshlets: 198 us
// vs 208 us with 4) and suit the late pass; for a pass of a few hundred thousand commands — one segment per wave — 4
// measure 3 us faster (10 M meshlets: 27.8 vs 31 us): the queues stay shorter, so every dependent step of the later
// workgroups' start-up chains is served sooner.  The host picks per launch from the command count of the PREVIOUS
// clustercull (the kernel leaves it in a mapped host word; frame coherence; a wrong guess only costs speed).
constexpr uint32_t CC_SHALLOW_COMMANDS = 500000;
#ifndef NV_CC_DB
#define NV_CC_DB 3
#endif
// the DIRECT form's pass B walks EVERY command of the segment through the ring: it is the launch's whole stream (12 bytes per meshlet), not the
// short tail it is behind the filter — its depth is a constant of its own (round 6 sweep: DESIGN.md §4.1)
#ifndef NV_CC_DB_DIRECT
#define NV_CC_DB_DIRECT 3
#endif
constexpr int CC_DB_DIRECT = NV_CC_DB_DIRECT;
// the packed direct form (round 6): windows of 64 ENTRIES in flight, and the room of the per-wave window words (64 windows of a full segment + the ring's run-out)
#ifndef NV_CP_DB
#define NV_CP_DB 3
#endif
constexpr int CP_DB = NV_CP_DB;
constexpr int CP_WINDOWS = 72;
static_assert(64 + 2 * CP_DB + 1 <= CP_WINDOWS, "the ring issues up to 2 CP_DB - 2 windows past a full segment's last, the map runs one window ahead of it");
constexpr int CC_DB = NV_CC_DB;         // ring slots of the exact pass (r2 sweep on 3A: 6 slots 30.2 us / step, 3 slots 29.3; pass B is
                                 // short in the sparse case and a deep ring is mostly redundant loads at its end)

// ---- conservative frustum filter (exactness-preserving early-out)
// The reference's sphere transform costs ~58 un-fused fp32 operations per meshlet and must be reproduced bit for bit
// wherever a decision depends on it.  But most commands lie entirely outside the frustum, and THAT can be proven with
// a much cheaper computation: per draw, the whole chain  view * (rotateQuat(v, q) * scale + position)  is one affine
// map  c = M v + b  (9 FMAs per meshlet), and the distance between this approximation and the reference's rounded
// result is bounded by  E = K u (alpha * max|v_i| + beta)  with  alpha = ||V||_inf,row * |s| * (1 + 2 Qa (Qa + |qw|)),
// beta = ||V||_inf,row * max|p_i| + max|V3_i|,  Qa = |qx|+|qy|+|qz|,  u = 2^-24  (standard forward error analysis:
// every intermediate of either evaluation is bounded by the same expression with absolute values; the reference
// chain is <= 13 roundings deep, the approximation <= 8 including the roundings inside M and b; K = 48 leaves > 2x
// slack).  Each frustum predicate of the reference has the form  g > -r  (or  z + r > znear,  z - r < zfar)  with
// coefficients |f| <= 1, so its margin moves by at most 4E when evaluated on the approximation.  A lane is dropped
// only when some margin is below -4E, i.e. when the reference's own test is certainly false; NaNs and infinities
// fail that comparison and fall through.  If any valid lane of the wave is not certainly out, the wave runs the
// exact path for all lanes — results are identical to the unfiltered kernel (tests/test_gpu_parity.py compares both
// against the oracle, including adversarial meshlets placed on the planes).
// FilterDraw, the constants and the derivation itself live in filtermath.h, which also compiles for the host: tests/test_cert_margins.py
// holds the margins it produces against the reference arithmetic on the CPU, context.hip takes filterK and the view norms from it.
// filterK = 4 K u S (rounded up), S = max(1, |f0| + |f1|, |f2| + |f3|) of the frustum coefficients: the margins above
// assume |f| <= 1; the host scales them for other coefficients and passes 0 (no filter, no certified test) for
// non-finite or absurd ones (filter_k, fill_cluster_args).
template <bool WITH_IS127 = true>
NV_DEV FilterDraw make_filter(const NvCullData& cd, const DrawUniform& u, float filterK, float Vn, float V3n, float sumV, float vmax3, float rmax)
{
	// (Vn, V3n = the row norms of the view matrix's linear part and of its translation, sumV = the sum of its twelve entries:
	// the host's, ClusterArgs::viewRowNorm ..., filter_view_norms; vmax3, rmax = the registered pool's bounds, ClusterArgs::poolBounds)
	return filter_make<WITH_IS127>(cd.view, u.q.x, u.q.y, u.q.z, u.qw, u.scale, u.pos.x, u.pos.y, u.pos.z, filterK, Vn, V3n, sumV, vmax3, rmax);
}

// Wave-uniform copy of one draw's filter.  M, aK, aR and scale live in SGPRs; the addends of the four FMA chains are
// pinned in VGPRs (a VOP3P instruction reads at most one SGPR, so an SGPR addend would cost a v_mov per use).
struct FilterUniform
{
	float m[9];
	float scale;
	float b0, b1, b2, tK; // VGPR-resident
};

NV_DEV float pin_vgpr(float x)
{
	float v;
	// x usually comes straight from a v_readlane: gfx950 needs 2 wait states between a VALU write of an SGPR and a VALU
	// read of it, and the hazard recognizer does not look inside inline asm
	asm("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v) : "s"(x));
	return v;
}

// true for lanes whose sphere is certainly outside the frustum (see above); never true on NaN.
// 19 VALU instructions per 64 meshlets (round 5; 23 before): 9 + 1 mixed-precision FMAs straight from the packed halfs, 8 for the plane
// distances and their minimum, one compare.  The threshold carries the radius:  every predicate of the reference reads
// g > -r (or z + r > znear, z - r < zfar; unit-length plane coefficients), so  min(g_i) < -(r + T)  proves all of them
// false at once;  T = bK + aK (|vx| + |vy| + |vz|) + aR |radius|  over-estimates 4E (sum >= max) plus the roundings of
// radius * scale and of the sums with r — and the filter uses tK >= T, the same expression at the registered pool's largest
// |centre component| and |radius| (filtermath.h filter_make): one value per draw, no per-meshlet arithmetic for the margin.
// FOLD (round 5, the early pass): the x and y rows of f arrive pre-multiplied by the side planes' coefficients (SegmentRegs::fold: f.m[0..2], f.b0 carry
// frustum[0], f.m[3..5], f.b1 carry frustum[2]), so a side plane's distance is one FMA instead of a multiply and an FMA — 17 instead of 19 instructions
// per 64 meshlets in a loop that is bound by vector issue.  One more rounding per entry of two of the three chains (relative u each, against the > 2x the
// margin's K = 48 leaves over the ~21 roundings it counts); tests/test_cert_margins.py holds this very form against the reference in its emulation.
// The folded distance is the reference's only for NON-NEGATIVE side-plane coefficients (|f0 cx| = f0 |cx| needs f0 >= 0): the kernel switches the filter off
// otherwise (foldSound, cluster_mask_kernel) — ADVICE r5; tests/test_cert_margins.py shows the folded form over-rejecting with a negated coefficient.
template <bool FOLD>
NV_DEV bool certainly_outside(const NvCullData& cd, const FilterUniform& f, uint32_t b0, uint32_t b1)
{
	const float vx = half_bits_to_float(b0 & 0xffffu), vy = half_bits_to_float(b0 >> 16), vz = half_bits_to_float(b1 & 0xffffu);
	const float rad = half_bits_to_float(b1 >> 16);
	const float cx = __builtin_fmaf(f.m[0], vx, __builtin_fmaf(f.m[1], vy, __builtin_fmaf(f.m[2], vz, f.b0)));
	const float cy = __builtin_fmaf(f.m[3], vx, __builtin_fmaf(f.m[4], vy, __builtin_fmaf(f.m[5], vz, f.b1)));
	const float cz = __builtin_fmaf(f.m[6], vx, __builtin_fmaf(f.m[7], vy, __builtin_fmaf(f.m[8], vz, f.b2)));
	const float thr = __builtin_fmaf(f.scale, rad, f.tK);
	const float g1 = FOLD ? __builtin_fmaf(cz, cd.frustum[1], -__builtin_fabsf(cx)) : __builtin_fmaf(cz, cd.frustum[1], -(__builtin_fabsf(cx) * cd.frustum[0]));
	const float g2 = FOLD ? __builtin_fmaf(cz, cd.frustum[3], -__builtin_fabsf(cy)) : __builtin_fmaf(cz, cd.frustum[3], -(__builtin_fabsf(cy) * cd.frustum[2]));
	const float gn = cz - cd.znear;
	const float gf = cd.zfar - cz;
	const float g = __builtin_fminf(__builtin_fminf(g1, g2), __builtin_fminf(gn, gf)); // minNum: a NaN distance is ignored, as a false compare was
	return g < -thr;
}

// lane l of a wave holds the l-th command of the wave's current 64-command segment (one coalesced 1280-B read
// instead of 64 dependent scalar loads) and the MeshDraw it points at; fields are broadcast with v_readlane as the
// wave walks the segment, so the walk itself contains no scalar-memory wait.
struct SegmentRegs
{
	uint32_t drawId, taskOffset, taskCount, lateDrawVisibility, meshletVisibilityOffset;
	float4 d0, d1; // position.xyz, scale | orientation.xyzw of draws[drawId]
	FilterDraw f;  // the frustum filter of that draw, derived lane-parallel (64 draws per ~130 VALU instructions)
	float fold[8]; // FOLD (the early pass): f.m[0..2], f.b[0] times frustum[0] | f.m[3..5], f.b[1] times frustum[2] — what the filter loop broadcasts for its x / y rows
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

// vec[lane c] = val (wave-uniform val and c)
NV_DEV uint32_t writelane_u32(uint32_t vec, uint32_t val, uint32_t c)
{
	return (threadIdx.x & 63u) == c ? val : vec;
}

NV_DEV float readlane_f(float v, uint32_t c) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), c)); }

template <bool FOLD>
NV_DEV FilterUniform segment_filter(const SegmentRegs& r, uint32_t c)
{
	FilterUniform f;
#pragma unroll
	for (int i = 0; i < 3; ++i)
	{
		f.m[i] = readlane_f(FOLD ? r.fold[i] : r.f.m[i], c);
		f.m[3 + i] = readlane_f(FOLD ? r.fold[4 + i] : r.f.m[3 + i], c);
		f.m[6 + i] = readlane_f(r.f.m[6 + i], c);
	}
	f.scale = readlane_f(r.f.scale, c);
	f.b0 = pin_vgpr(readlane_f(FOLD ? r.fold[3] : r.f.b[0], c));
	f.b1 = pin_vgpr(readlane_f(FOLD ? r.fold[7] : r.f.b[1], c));
	f.b2 = pin_vgpr(readlane_f(r.f.b[2], c));
	f.tK = pin_vgpr(readlane_f(r.f.tK, c));
	return f;
}

NV_DEV DrawUniform lane_draw(const SegmentRegs& r)
{
	DrawUniform u;
	u.pos = { r.d0.x, r.d0.y, r.d0.z };
	u.scale = r.d0.w;
	u.q = { r.d1.x, r.d1.y, r.d1.z };
	u.qw = r.d1.w;
	return u;
}

NV_DEV DrawUniform segment_draw(const SegmentRegs& r, uint32_t c)
{
	DrawUniform u;
	u.pos = { readlane_f(r.d0.x, c), readlane_f(r.d0.y, c), readlane_f(r.d0.z, c) };
	u.scale = readlane_f(r.d0.w, c);
	u.q = { readlane_f(r.d1.x, c), readlane_f(r.d1.y, c), readlane_f(r.d1.z, c) };
	u.qw = readlane_f(r.d1.w, c);
	return u;
}

// ---- certified two-sided test (pass B)
// The filter's approximation  c~ = M v + b  decides a predicate of the reference in BOTH directions once its distance to
// the threshold exceeds the margin:  |c~ - c_ref| <= E,  every frustum predicate moves by at most 4E (above), so
//     min_i g_i(c~) + r  < -T   =>  some predicate of the reference is false   (the filter),
//     min_i g_i(c~) + r  >  T   =>  all four are true,
// with T >= 4E + the roundings of radius * scale.  The cone test (math.h:41-44)  dot(c, axis) >= cutoff |c| + r  gets
// the same treatment: axis~ = (M k) / (127 scale)  for the int8 axis k  (M / scale = V R, the linear part of the
// reference's mat3(view) * rotateQuat(k / 127, q)), |c~| through v_sqrt_f32 (1 ulp), and
//     D = dot(c~, axis~) - (cutoff~ |c~| + r~),    |D - (lhs_ref - rhs_ref)| <= E_cone,
//     E_cone <= u (163 A Aax + 63 A) + 4.1 u |r|,   A = alpha max|v_i| + beta (the magnitude bound behind E = 48 u A),
//     Aax = 1.01 ||V||_inf,row (1 + 2 Qa (Qa + |qw|))  (the same bound for the axis chain),
// from the same forward analysis: the reference's axis chain is <= 9 roundings deep, its dot 3, its length 2.6 u relative
// plus |c_ref - c*|_2, the approximation's M 7, its FMA chains 3 each, the factor 1 / (127 scale) 3.  With T = 192 u A + ...
// the margin  T coneK,  coneK = 2.02 ||V|| (1 + 2 Qa (Qa + |qw|)) + 1,  is >= 2 E_cone.  D > T coneK: certainly culled;
// D < -T coneK: certainly kept.  NaN / inf anywhere makes T NaN / inf (make_filter), every comparison false, the lane
// undecided.  A command with an undecided lane that matters runs the reference arithmetic for the whole wave, so the
// result is the reference's in every case; tests/test_gpu_parity.py places meshlets within ulps of the cone threshold
// and of the planes, tools/experiments/cert_margin.py measures the bound's slack against exact rational arithmetic.
struct CertUniform
{
	float m[9];
	float aK, aR, scale, coneK, is127;
	float b0, b1, b2, bK; // VGPR-resident (addends of the FMA chains)
};

NV_DEV CertUniform segment_cert(const SegmentRegs& r, uint32_t c)
{
	CertUniform f;
#pragma unroll
	for (int i = 0; i < 9; ++i)
		f.m[i] = readlane_f(r.f.m[i], c);
	f.aK = readlane_f(r.f.aK, c);
	f.aR = readlane_f(r.f.aR, c);
	f.scale = readlane_f(r.f.scale, c);
	f.coneK = readlane_f(r.f.coneK, c);
	f.is127 = readlane_f(r.f.is127, c);
	f.b0 = pin_vgpr(readlane_f(r.f.b[0], c));
	f.b1 = pin_vgpr(readlane_f(r.f.b[1], c));
	f.b2 = pin_vgpr(readlane_f(r.f.b[2], c));
	f.bK = pin_vgpr(readlane_f(r.f.bK, c));
	return f;
}

NV_DEV float s8_to_float(uint32_t word, int byte)
{
	return (float)(int)(int8_t)(word >> (8 * byte));
}

// need = the lanes whose decision matters (valid, and for the early pass with visibility bits: bit set).  Returns true
// when every such lane is decided; *vis = ballot of the lanes that pass frustum and cone (clustercull.comp.glsl:102-108).
NV_DEV bool certified_visible(const NvCullData& cd, const CertUniform& f, uint32_t b0, uint32_t b1, uint32_t cone, uint64_t need, uint64_t* vis,
                              bool* filterRejects = nullptr)
{
	const float vx = half_bits_to_float(b0 & 0xffffu), vy = half_bits_to_float(b0 >> 16), vz = half_bits_to_float(b1 & 0xffffu);
	const float rad = half_bits_to_float(b1 >> 16);
	const float cx = __builtin_fmaf(f.m[0], vx, __builtin_fmaf(f.m[1], vy, __builtin_fmaf(f.m[2], vz, f.b0)));
	const float cy = __builtin_fmaf(f.m[3], vx, __builtin_fmaf(f.m[4], vy, __builtin_fmaf(f.m[5], vz, f.b1)));
	const float cz = __builtin_fmaf(f.m[6], vx, __builtin_fmaf(f.m[7], vy, __builtin_fmaf(f.m[8], vz, f.b2)));
	float T = __builtin_fmaf(f.aK, __builtin_fabsf(vx), f.bK);
	T = __builtin_fmaf(f.aK, __builtin_fabsf(vy), T);
	T = __builtin_fmaf(f.aK, __builtin_fabsf(vz), T);
	T = __builtin_fmaf(f.aR, __builtin_fabsf(rad), T);
	const float thrHi = __builtin_fmaf(f.scale, rad, T), thrLo = __builtin_fmaf(f.scale, rad, -T);
	const float g1 = __builtin_fmaf(cz, cd.frustum[1], -(__builtin_fabsf(cx) * cd.frustum[0]));
	const float g2 = __builtin_fmaf(cz, cd.frustum[3], -(__builtin_fabsf(cy) * cd.frustum[2]));
	const float gn = cz - cd.znear;
	const float gf = cd.zfar - cz;
	// a finite T implies finite, bounded c~ and (host-checked) finite coefficients: no NaN is dropped by these minima
	const float g = __builtin_fminf(__builtin_fminf(g1, g2), __builtin_fminf(gn, gf));
	const uint64_t outM = __ballot(g < -thrHi), inM = __ballot(g > -thrLo);
	if (filterRejects) // what pass A's filter would have said about this command (the same comparison)
		*filterRejects = (need & ~outM) == 0;
	if (need & ~(outM | inM))
		return false;
	uint64_t alive = need & inM;
	if (cd.clusterBackfaceEnabled != 0 && alive)
	{
		const float kx = s8_to_float(cone, 0), ky = s8_to_float(cone, 1), kz = s8_to_float(cone, 2), kc = s8_to_float(cone, 3);
		const float wx = __builtin_fmaf(f.m[0], kx, __builtin_fmaf(f.m[1], ky, f.m[2] * kz));
		const float wy = __builtin_fmaf(f.m[3], kx, __builtin_fmaf(f.m[4], ky, f.m[5] * kz));
		const float wz = __builtin_fmaf(f.m[6], kx, __builtin_fmaf(f.m[7], ky, f.m[8] * kz));
		const float lhs = __builtin_fmaf(cx, wx, __builtin_fmaf(cy, wy, cz * wz)) * f.is127;
		const float len = __builtin_amdgcn_sqrtf(__builtin_fmaf(cx, cx, __builtin_fmaf(cy, cy, cz * cz)));
		const float rhs = __builtin_fmaf(kc * INV_127, len, f.scale * rad);
		const float D = lhs - rhs;
		const float Tc = T * f.coneK;
		const uint64_t cullM = __ballot(D > Tc), keepM = __ballot(D < -Tc);
		if (alive & ~(cullM | keepM))
			return false;
		alive &= keepM;
	}
	*vis = alive;
	return true;
}

// ---- software-pipelined meshlet stream (SoA mirror only)
// hipcc's s_waitcnt insertion collapses a loop-carried prefetch ring to (almost) vmcnt(0): measured, every command then
// costs one full memory latency.  The ring's loads are therefore issued from inline asm, which hipcc does not count,
// and waited for by hand with a counted vmcnt (cdna_hip_programming.md §5.7, form (ii): "=v" loads, then a wait
// statement that names every destination "+v" before its first consumer).  Rules that keep this safe:
//   * a slot's registers are touched by nothing but its issue / wait statements until the wait has passed;
//   * slots are reissued in a fixed rotation with unconditional loads (indices clamped, never branched), so exactly
//     (CC_D - 1) * LOADS younger ring loads are outstanding at every wait;
//   * VMEM operations hipcc issues itself in between (HiZ texels) are younger than the slot being waited for and are
//     consumed before the next ring issue (the issue statement takes the command's ballot as an operand), so they can
//     only make a wait stricter, never too weak.  Stricter is slower, though: a store issued inside a ring lengthens
//     every counted wait behind it by its own latency, so ballots, tile counts and visibility-bit updates are written
//     per segment, after the rings have drained;
//   * each slot has exactly ONE issue site and ONE wait site in the loop body.  With a second copy of the loop (a fast
//     path next to a general path) hipcc joins the slots' registers at the merge point with v_mov — copying registers
//     whose loads are still in flight (observed: memory faults from garbage addresses);
//   * the hazard recognizer does not look inside inline asm: an SGPR written by v_readlane needs s_nop 1 before an
//     asm VALU instruction reads it (pin_vgpr), and 5 wait states before an asm VMEM instruction reads it — which
//     happens even to kernel arguments: under SGPR pressure hipcc spills the base pointers to VGPR lanes and reloads
//     them with v_readlane directly in front of the asm statement (observed: memory faults from a stale base).  Every
//     asm load that takes an SGPR base therefore starts with s_nop 4.
// ring A (filter pass): the 8 bounds bytes per meshlet (+ the visibility word when BITS)
struct SlotA
{
	uint64_t bounds; // center.xy | center.z, radius (4 x fp16)
	uint32_t mvbWord;
};

// ring B (exact pass over the surviving commands): bounds + cone (+ visibility word)
struct SlotB
{
	uint64_t bounds;
	uint32_t cone;
	uint32_t mvbWord;
};

NV_DEV uint32_t lane_meshlet(uint32_t taskOffset, uint32_t taskCount, uint32_t lane)
{
	return (taskCount ? taskOffset : 0u) + (lane < taskCount ? lane : 0u);
}

// off8 = byte offset of this lane's bounds record; offw = byte offset of its visibility word (BITS)
// NV_PLAIN_LOADS (libniagara_vis_plain.so): the same kernels with ordinary loads and the compiler's own waits —
// slower (hipcc waits for a loop-carried ring with ~vmcnt(0)), but free of every assumption the counted waits make about
// register allocation and instruction order.  tests/test_plain_loads.py holds the asm build to its results.
// Cache policy of the kernel's streams (round 4): pass A's bounds and pass B's cones are read once per pass, so they are loaded non-temporal
// (`nt`) and leave the L2 to what is read again — the ballots and tile counts the scatter launch picks up, pass B's re-read of the candidates'
// bounds.  Headline pass, three alternating rounds on one box: 28.4-29.3 -> 28.0-28.3 us (the scatter launch 7.8 -> 7.4-7.5 us by events); the frame,
// the dense pass and config 4 do not notice (202-204 us either way).  Pass B's bounds stay cacheable: the late pass's occlusion stage gathers the
// survivors' bounds again.  Also measured: `nt` on pass B's bounds as well (no further gain), non-temporal stores of the scatter launch's IDs
// (early scatter of the frame 12.8 -> 12.3 us, dense scatter 15.2 -> 15.5: not adopted).
#define NV_POLICY_A " nt"
#define NV_POLICY_CONE " nt"
#define NV_POLICY_B ""

template <bool BITS>
NV_DEV void ringA_issue(SlotA& s, const ClusterArgs& a, uint32_t off8, uint32_t offw, uint64_t order)
{
#ifdef NV_PLAIN_LOADS
	(void)order;
	s.bounds = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(a.soaBounds) + off8);
	s.mvbWord = BITS ? *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.mvb) + offw) : 0u;
#else
	if (BITS)
	{
		asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %2, %3" NV_POLICY_A "\n\tglobal_load_dword %1, %4, %5"
		             : "=&v"(s.bounds), "=&v"(s.mvbWord)
		             : "v"(off8), "s"(a.soaBounds), "v"(offw), "s"(a.mvb), "s"(order)
		             : "memory");
	}
	else
	{
		asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" NV_POLICY_A : "=&v"(s.bounds) : "v"(off8), "s"(a.soaBounds), "s"(order) : "memory");
		s.mvbWord = 0;
	}
#endif
}

// every counted wait of the file goes through here: a no-op in the plain build (the compiler waits where it must)
#ifdef NV_PLAIN_LOADS
#define NV_COUNTED_WAIT(...) do { } while (0)
#else
#define NV_COUNTED_WAIT(...) asm volatile(__VA_ARGS__)
#endif

template <bool BITS, int YOUNGER>
NV_DEV void ringA_wait(SlotA& s)
{
	if (BITS)
		NV_COUNTED_WAIT("s_waitcnt vmcnt(%2) ; nv_ready %0 %1" : "+v"(s.bounds), "+v"(s.mvbWord) : "i"(YOUNGER * 2) : "memory");
	else
		NV_COUNTED_WAIT("s_waitcnt vmcnt(%1) ; nv_ready %0" : "+v"(s.bounds) : "i"(YOUNGER) : "memory");
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool BITS>
NV_DEV void ringB_issue(SlotB& s, const ClusterArgs& a, uint32_t taskOffset, uint32_t taskCount, uint32_t mvo, uint32_t lane, uint64_t order)
{
	const uint32_t li = lane < taskCount ? lane : 0u;
	const uint32_t mi = lane_meshlet(taskOffset, taskCount, lane);
	const uint32_t off8 = mi * 8u, off4 = mi * 4u;
	const uint32_t offw = BITS && taskCount ? ((mvo + li) >> 5) * 4u : 0u;
#ifdef NV_PLAIN_LOADS
	(void)order;
	s.bounds = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(a.soaBounds) + off8);
	s.cone = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.soaCones) + off4);
	s.mvbWord = BITS ? *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.mvb) + offw) : 0u;
#else
	if (BITS)
	{
		asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %3, %4" NV_POLICY_B "\n\tglobal_load_dword %1, %5, %6" NV_POLICY_CONE "\n\tglobal_load_dword %2, %7, %8"
		             : "=&v"(s.bounds), "=&v"(s.cone), "=&v"(s.mvbWord)
		             : "v"(off8), "s"(a.soaBounds), "v"(off4), "s"(a.soaCones), "v"(offw), "s"(a.mvb), "s"(order)
		             : "memory");
	}
	else
	{
		asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %2, %3" NV_POLICY_B "\n\tglobal_load_dword %1, %4, %5" NV_POLICY_CONE
		             : "=&v"(s.bounds), "=&v"(s.cone)
		             : "v"(off8), "s"(a.soaBounds), "v"(off4), "s"(a.soaCones), "s"(order)
		             : "memory");
		s.mvbWord = 0;
	}
#endif
}

// the packed direct form: the lane's meshlet index comes from the wave's entry -> command map (one lane = one ENTRY of the segment's flattened meshlet list)
template <bool BITS>
NV_DEV void ringP_issue(SlotB& s, const ClusterArgs& a, uint32_t mi, uint32_t offw, uint64_t order)
{
	const uint32_t off8 = mi * 8u, off4 = mi * 4u;
#ifdef NV_PLAIN_LOADS
	(void)order;
	s.bounds = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(a.soaBounds) + off8);
	s.cone = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.soaCones) + off4);
	s.mvbWord = BITS ? *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.mvb) + offw) : 0u;
#else
	if (BITS)
	{
		asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %3, %4" NV_POLICY_B "\n\tglobal_load_dword %1, %5, %6" NV_POLICY_CONE "\n\tglobal_load_dword %2, %7, %8"
		             : "=&v"(s.bounds), "=&v"(s.cone), "=&v"(s.mvbWord)
		             : "v"(off8), "s"(a.soaBounds), "v"(off4), "s"(a.soaCones), "v"(offw), "s"(a.mvb), "s"(order)
		             : "memory");
	}
	else
	{
		asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %2, %3" NV_POLICY_B "\n\tglobal_load_dword %1, %4, %5" NV_POLICY_CONE
		             : "=&v"(s.bounds), "=&v"(s.cone)
		             : "v"(off8), "s"(a.soaBounds), "v"(off4), "s"(a.soaCones), "s"(order)
		             : "memory");
		s.mvbWord = 0;
	}
#endif
}

template <bool BITS, int YOUNGER>
NV_DEV void ringB_wait(SlotB& s)
{
	if (BITS)
		NV_COUNTED_WAIT("s_waitcnt vmcnt(%3) ; nv_ready %0 %1 %2" : "+v"(s.bounds), "+v"(s.cone), "+v"(s.mvbWord) : "i"(YOUNGER * 3) : "memory");
	else
		NV_COUNTED_WAIT("s_waitcnt vmcnt(%2) ; nv_ready %0 %1" : "+v"(s.bounds), "+v"(s.cone) : "i"(YOUNGER * 2) : "memory");
}

// End of a ring: wait for everything, THEN release the slots.  The last loads of a ring are redundant (clamped re-reads
// whose values nobody uses), so for the compiler the slots are dead as soon as their last consumer has run — and it
// reuses their registers for whatever comes next (observed: a division sunk below the filter loop computed in two slot
// registers while their loads were still in flight, and was overwritten when they landed).  Naming every slot in a
// statement BEHIND the drain keeps the registers reserved until the loads have landed (tools/check_asm_hazards.py, check 2).
NV_DEV void ring_drain() { asm volatile("s_waitcnt vmcnt(0) ; nv_ready all" ::: "memory"); }
NV_DEV void ring_release(SlotA& s) { asm volatile("; released %0 %1" : "+v"(s.bounds), "+v"(s.mvbWord)); }
NV_DEV void ring_release(SlotB& s) { asm volatile("; released %0 %1 %2" : "+v"(s.bounds), "+v"(s.cone), "+v"(s.mvbWord)); }

// ---------------------------------------------------------------------------------------------------------------
// The packed walk of the direct form (round 6).  One command per wave iteration runs the certified test on the command's VALID lanes
// only — 40 of 64 on average behind drawcull's LOD select (a draw's meshlets end in a partial command) — and the direct form is bound by instruction
// issue.  Here the valid meshlets of ALL the segment's commands are ONE list of E entries (command-major: the order of the commands, the lanes of a
// command in order) and a wave iteration takes a WINDOW of 64 consecutive entries, whatever commands they belong to: every lane of every window but the
// segment's last is live.  What was wave-uniform per command becomes per-lane: the coefficients come from the wave's table in LDS (16-byte reads, mostly
// broadcasts: a window spans one to three commands), the entry -> command map is a bit mask of the commands' first entries and one v_mbcnt pair per
// window.  Decisions are certified_visible's, per lane; a lane it leaves undecided runs the reference arithmetic (bits_round's form).  A window's
// `visible` ballot goes to LDS and is cut back into the commands' ballots lane-parallel at the segment's end (cluster_mask_kernel), and so is the
// `not certainly outside` ballot behind the launch's statistic.
// (The same walk over the lanes the FILTER could not finish — the filter form's exact pass, the candidates' need masks as the entries, the lane's position
// the n-th set bit of its command's mask — was built and measured in round 6 and is not here: tools/experiments/sparse_exact_pass_r6.diff.  The walk's
// fixed price, ~500 instructions and half a dozen LDS round trips per segment, is more than the one or two candidates of a usual segment cost one at a
// time: headline pass 26.0 -> 27.9 us; as a hybrid — the walk for segments with four or more candidates — both rings in one kernel spill: 31-32 us.)
// The margin is the draw's tK >= T of every meshlet of the pool (filtermath.h filter_make: what the filter pass uses) instead of the meshlet's own T: four
// multiply-adds and a 16-byte LDS read less per window, a margin wider by a few per cent (tests/test_cert_margins.py holds both two-sided tests, with T and
// with tK, against the reference's decisions); a pool with a non-finite record has tK = inf and every lane takes the reference arithmetic.
// Table entry (CP_ENTRY bytes per non-empty command, in the order of the commands; LDS is what bounds the six workgroups per CU):
//   +0 the coefficients: m[0..2], b0 | m[3..5], b1 | m[6..8], b2 | tK, scale, tK coneK, is127  (64 bytes)
//   +64 taskOffset - the command's first entry   +68 BITS: meshletVisibilityOffset - the command's first entry   +72 first entry | the command's lane in the segment << 16
constexpr uint32_t CP_ENTRY = 80, CP_ENTRY_HEAD = 64;
static_assert(CP_ENTRY % 16 == 0, "the coefficients are read with ds_read_b128");

NV_DEV void walk_coefficients(char* at, const FilterDraw& f)
{
	float4* t4 = reinterpret_cast<float4*>(at);
	t4[0] = make_float4(f.m[0], f.m[1], f.m[2], f.b[0]);
	t4[1] = make_float4(f.m[3], f.m[4], f.m[5], f.b[1]);
	t4[2] = make_float4(f.m[6], f.m[7], f.m[8], f.b[2]);
	t4[3] = make_float4(f.tK, f.scale, f.tK * f.coneK, f.is127);
}

// heads: bit p set = a command other than the first starts at entry p + 1 (so the commands in front of entry e, the first not counted, are the set
// bits at positions < e); CP_WINDOWS words, zero past the list.  PACK overwrites word j with window j's `visible` ballot once the map has passed it.
// segDrawId: lane c = the drawId of the segment's c-th command (the reference arithmetic's draw comes through a lane permutation: rare).
// between(): what the caller does between the ring's first requests and the first window's test (PACK: the MeshDraw gather's wait and the
// coefficients — the map needs the commands only, so the dependent chain of a wave's start is commands -> {draws, first windows} -> coefficients).
template <bool BITS, class Between>
NV_DEV void packed_walk(const ClusterArgs& a, char* tabBytes, uint64_t* heads, uint64_t* nout, uint32_t lane, uint32_t E, uint32_t segDrawId, bool useCertP, Between&& between)
{
	const NvCullData& cd = a.cd;
	const uint32_t nW = (E + 63u) >> 6, eLast = E ? E - 1u : 0u;
	// the map runs one window AHEAD of the ring's issue: window j's {table entry, meshlet} are in registers when its loads are issued, and the heads
	// word of the window after it is already requested — no LDS round trip sits between a landed window and the next request
	uint32_t startsBefore = 0; // commands that start in front of the mapped window, the first one not counted (scalar)
	uint32_t ePos = lane;      // the lane's entry in the window being mapped
	uint32_t jMap = 0;         // the window being mapped
	uint64_t H = heads[0];
	uint32_t rkNext = 0; // the lane's table entry in the mapped window (byte offset: an LDS POINTER kept across the ring decays to a generic one — flat
	                     // loads, which count on vmcnt too, and hipcc then waits for the whole ring in front of every window: 3A dense 33 -> 50 us)
	uint32_t miNext = 0, wNext = 0; // the lane's meshlet (and, BITS, the byte offset of its visibility word) in the mapped window
	auto map_next = [&]()
	{
		// (the heads word is the same in every lane: through the scalar unit, so that the running count of commands costs no vector instruction)
		const uint32_t hlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)H), hhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(H >> 32));
		rkNext = __umul24(__builtin_amdgcn_mbcnt_hi(hhi, __builtin_amdgcn_mbcnt_lo(hlo, startsBefore)), CP_ENTRY);
		startsBefore += (uint32_t)__builtin_popcount(hlo) + (uint32_t)__builtin_popcount(hhi);
		const uint32_t e = ePos < eLast ? ePos : eLast; // the lanes past the list's end (and the ring's windows past its last) re-read the last entry: in range, unconditional
		if (BITS)
		{
			const uint2 h = *reinterpret_cast<const uint2*>(tabBytes + rkNext + CP_ENTRY_HEAD); // {taskOffset - first entry, meshletVisibilityOffset - first entry}
			miNext = h.x + e;
			wNext = ((h.y + e) >> 5) * 4u;
		}
		else
			miNext = *reinterpret_cast<const uint32_t*>(tabBytes + rkNext + CP_ENTRY_HEAD) + e;
		ePos += 64u;
		++jMap;
		H = heads[jMap]; // (jMap <= 64 + 2 CP_DB: zero past the segment's last window — the lanes stay on the last command)
	};
	map_next();
	SlotB ring[CP_DB];
	uint32_t entryOf[CP_DB]; // the lane's command in the slot's window: the byte offset of its table entry
	auto issueP = [&](SlotB& slot, uint32_t& te, uint64_t order)
	{
		te = rkNext;
		ringP_issue<BITS>(slot, a, miNext, wNext, order);
		map_next();
	};
#pragma unroll
	for (int k = 0; k < CP_DB; ++k)
		issueP(ring[k], entryOf[k], 0);
	between();
#ifdef NV_EXPERIMENTS
	const uint64_t certM = useCertP ? ~0ull : 0ull; // (bit 20: no certified test — every valid lane takes the reference arithmetic)
#else
	(void)useCertP; // (the host takes the direct form only with filterK > 0: launch_cluster_mask)
#endif
	for (uint32_t j0 = 0; j0 < nW; j0 += CP_DB)
	{
#pragma unroll
		for (int k = 0; k < CP_DB; ++k)
		{
			const uint32_t j = j0 + k;
			// the coefficients, requested in front of the wait for the window's meshlets
			const char* te = tabBytes + entryOf[k];
			const float4* tc = reinterpret_cast<const float4*>(te);
			const float4 r0 = tc[0], r1 = tc[1], r2 = tc[2], r3 = tc[3];
			// (BITS: the lane's bit within its visibility word — the entry's bit base again, and the lane's entry: cheaper than a register per ring slot)
			uint32_t bitShift = 0;
			if (BITS)
			{
				const uint32_t e = j * 64u + lane;
				bitShift = (*reinterpret_cast<const uint32_t*>(te + CP_ENTRY_HEAD + 4) + (e < eLast ? e : eLast)) & 31u;
			}
			ringB_wait<BITS, CP_DB - 1>(ring[k]);
			uint64_t visM = 0;
			if (j < nW)
			{
				const uint32_t b0 = (uint32_t)ring[k].bounds, b1 = (uint32_t)(ring[k].bounds >> 32), cone = ring[k].cone;
				const uint32_t left = E - j * 64u;
				uint64_t validM = left >= 64u ? ~0ull : (1ull << left) - 1ull;
				if (BITS) // the early pass with visibility bits: only last frame's visible clusters can be visible (clustercull.comp.glsl:91-92) — the others decide nothing
					validM &= __ballot((ring[k].mvbWord >> bitShift & 1u) != 0);
				// certified_visible, one lane = one cluster
				const float vx = half_bits_to_float(b0 & 0xffffu), vy = half_bits_to_float(b0 >> 16), vz = half_bits_to_float(b1 & 0xffffu);
				const float rad = half_bits_to_float(b1 >> 16);
				const float cx = __builtin_fmaf(r0.x, vx, __builtin_fmaf(r0.y, vy, __builtin_fmaf(r0.z, vz, r0.w)));
				const float cy = __builtin_fmaf(r1.x, vx, __builtin_fmaf(r1.y, vy, __builtin_fmaf(r1.z, vz, r1.w)));
				const float cz = __builtin_fmaf(r2.x, vx, __builtin_fmaf(r2.y, vy, __builtin_fmaf(r2.z, vz, r2.w)));
				const float T = r3.x, scale = r3.y, Tc = r3.z, is127 = r3.w; // (the draw's tK and tK coneK)
				const float thrHi = __builtin_fmaf(scale, rad, T), thrLo = __builtin_fmaf(scale, rad, -T);
				const float g1 = __builtin_fmaf(cz, cd.frustum[1], -(__builtin_fabsf(cx) * cd.frustum[0]));
				const float g2 = __builtin_fmaf(cz, cd.frustum[3], -(__builtin_fabsf(cy) * cd.frustum[2]));
				const float gn = cz - cd.znear;
				const float gf = cd.zfar - cz;
				const float g = __builtin_fminf(__builtin_fminf(g1, g2), __builtin_fminf(gn, gf));
#ifdef NV_EXPERIMENTS
				const uint64_t outM = __ballot(g < -thrHi) & certM, inM = __ballot(g > -thrLo) & certM;
#else
				const uint64_t outM = __ballot(g < -thrHi), inM = __ballot(g > -thrLo);
#endif
				uint64_t decidedM = outM | inM;
				visM = inM;
				if (cd.clusterBackfaceEnabled != 0 && (inM & validM))
				{
					const float kx = s8_to_float(cone, 0), ky = s8_to_float(cone, 1), kz = s8_to_float(cone, 2), kc = s8_to_float(cone, 3);
					const float wx = __builtin_fmaf(r0.x, kx, __builtin_fmaf(r0.y, ky, r0.z * kz));
					const float wy = __builtin_fmaf(r1.x, kx, __builtin_fmaf(r1.y, ky, r1.z * kz));
					const float wz = __builtin_fmaf(r2.x, kx, __builtin_fmaf(r2.y, ky, r2.z * kz));
					const float lhs = __builtin_fmaf(cx, wx, __builtin_fmaf(cy, wy, cz * wz)) * is127;
					const float len = __builtin_amdgcn_sqrtf(__builtin_fmaf(cx, cx, __builtin_fmaf(cy, cy, cz * cz)));
					const float rhs = __builtin_fmaf(kc * INV_127, len, scale * rad);
					const float D = lhs - rhs;
					const uint64_t cullM = __ballot(D > Tc), keepM = __ballot(D < -Tc);
					decidedM = outM | (inM & (cullM | keepM)); // (a cluster outside the frustum is decided whatever its cone says)
					visM = inM & keepM;
				}
				visM &= validM;
				const uint64_t undecidedM = validM & ~decidedM;
				if (undecidedM && !NV_DBG(a, 536870912u)) // the reference's arithmetic (clustercull.comp.glsl:72-80,102-108) for the lanes inside a margin, as cull_command evaluates it.  (bit 29, experiments: skipped — what the undecided lanes cost)
				{
					// the lane's draw: the drawId of its command, from the lane of the segment that holds the command (all lanes take part in the permutation)
					const uint32_t cmdLane = *reinterpret_cast<const uint32_t*>(te + CP_ENTRY_HEAD + 8) >> 16;
					const uint32_t drawId = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(cmdLane << 2), (int)segDrawId);
					bool visible = false;
					if (undecidedM >> lane & 1ull)
					{
						const float4* dp = reinterpret_cast<const float4*>(a.draws + drawId);
						const float4 q0 = dp[0], q1 = dp[1];
						DrawUniform u;
						u.pos = { q0.x, q0.y, q0.z };
						u.scale = q0.w;
						u.q = { q1.x, q1.y, q1.z };
						u.qw = q1.w;
						LaneData l;
						l.b0 = b0;
						l.b1 = b1;
						l.cone = cone;
						l.mvbWord = 0;
						f3 c;
						float rr;
						lane_sphere(cd, u, l, c, rr);
						visible = frustum_test(cd, c, rr);
						if (cd.clusterBackfaceEnabled != 0 && visible)
						{
							f3 axis;
							float cutoff;
							lane_cone(cd, u, l, axis, cutoff);
							visible = !cone_cull(c, rr, axis, cutoff);
						}
					}
					visM = (visM & ~undecidedM) | (__ballot(visible) & undecidedM);
				}
				if (lane == 0)
				{
					heads[j] = visM; // (window j's heads were consumed CP_DB + 1 windows ago)
					if (!BITS)
						nout[j] = validM & ~outM;
				}
			}
			issueP(ring[k], entryOf[k], visM);
		}
	}
	ring_drain();
#pragma unroll
	for (int k = 0; k < CP_DB; ++k)
		ring_release(ring[k]);
}

// commands per scatter tile: the same function of the indirect words in both kernels
// n / d for a launch constant d whose magic the host prepared (ClusterArgs): one s_mul_hi_u32 and a shift instead of the ~18
// instructions of a 32-bit division by a run-time value
NV_DEV uint32_t div_launch_constant(uint32_t n, uint32_t d, uint32_t magic)
{
	return magic ? __umulhi(n, magic) >> 7 : n / d;
}

NV_DEV uint32_t scatter_tile_commands(uint32_t numCmds, uint32_t tiles, uint32_t tilesMagic)
{
	static_assert(DEAL_TILE_THREADS == CC_THREADS, "dealing.h");
	return deal_tile_commands(numCmds, tiles, tilesMagic);
}

// wave w's c-th command (c counts through the wave's chunks in order)
// true in every lane of a quad iff all four lanes of the quad have v == ref (ref = the quad's first lane's v)
NV_DEV bool __all_quad_same(uint32_t v, uint32_t ref)
{
	const uint64_t diff = __ballot(v != ref);
	const uint32_t lane = threadIdx.x & 63u;
	return (diff >> (lane & ~3u) & 0xfull) == 0;
}

// Static work assignment of the cull kernel: chunks of CC_CHUNK commands dealt round-robin over the waves — but not
// evenly.  The workgroups that share a CU start within a microsecond of each other, yet the SIMD arbitrates
// oldest-first and the later a workgroup was dispatched, the slower it streams (measured, 10 M meshlets, 6 workgroups
// per CU, even dealing: the first workgroup of a CU finished after 13 us, the sixth after 18-20 us, and the launch
// ended with a quarter of the waves running alone).  Generation g (= workgroup index / workgroups per generation;
// dispatch is round-robin over the CUs — an assumption that only affects speed) therefore takes part in fewer rounds:
// per-wave commands = mean + mean(delay) - delay[g], the delays in units of the time a command takes to stream.
// Round r hands one chunk to every wave of the generations that still take part (a prefix of the wave index space,
// because later generations leave first), so the chunks of a round stay contiguous in memory like in plain round-robin;
// what is left after the weighted rounds is dealt evenly.
// The weighted rounds of a wave are computed ONCE, lane-parallel (lane j = the wave's j-th chunk), and kept in one VGPR;
// keeping the parameters live as scalars instead cost the late-pass variants 30 VGPRs of spill code and with them two
// of the six resident workgroups per CU.  Returns the number of chunks of this wave; *chunkOf = index of its lane-th
// chunk, or the plain round-robin when the pass is not weighted (other grid shapes, tiny passes, or more than 64
// chunks per wave — where a start-up delay of a few commands no longer matters).
// The part that depends only on the pass's command count and the launch shape is a PLAN (dealing.h deal_plan): the host derives it for the
// previous launch's count and passes it with the arguments, a wave whose count word says otherwise derives it itself (round 5: ~150 scalar
// instructions off every wave's start-up path).  What is left per wave: its own rounds, its chunk count and the lane-parallel chunk table.
// the chunk of this lane's command in segment s (<= 3: the table holds 64 chunks) of a wave that deals from its table (deal_wave)
NV_DEV uint32_t segment_chunk(uint32_t chunkOf, uint32_t s)
{
	switch (s)
	{
	case 0: return dpp_move_u32<0x00>(chunkOf); // quad_perm:[0,0,0,0]
	case 1: return dpp_move_u32<0x55>(chunkOf);
	case 2: return dpp_move_u32<0xAA>(chunkOf);
	default: return dpp_move_u32<0xFF>(chunkOf);
	}
}

NV_DEV uint32_t deal_wave(const DealPlan& p, uint32_t w, uint32_t lane, uint32_t gen, uint32_t genWaves, uint32_t* chunkOf)
{
	if (!p.weighted)
	{
		*chunkOf = deal_wave_entry(p, w, 0u, genWaves, lane);
		return deal_wave_chunks(p, w, gen);
	}
	// table entry j lives in lane ((j & 15) << 2) | (j >> 4): the lanes of a segment's quad q all want entry 16 s + q (s = the segment's
	// number, four commands per chunk), which is lane s of their own quad — one DPP quad broadcast (segment_chunk) where a table in lane order
	// needed a ds_bpermute_b32 round trip through LDS on every wave's start-up path.  The arithmetic is dealing.h's (tests/test_dealing.py).
	static_assert(CC_CHUNK == 4, "the chunk table's quad layout");
	const uint32_t j = (lane >> 2) + 16u * (lane & 3u);
	*chunkOf = deal_wave_entry(p, w, deal_wave_rounds(p, gen), genWaves, j);
	return deal_wave_chunks(p, w, gen);
}

// DIRECT (SoA mirror only): no pass A — every valid command goes straight to pass B, bounds and cone read once.  The
// host picks it for a launch when the previous launch found that most commands pass the filter (frame coherence; the
// kernels count what the filter rejects, or would have rejected, and leave the count in a mapped host word): a pass over
// the commands of draws that drawcull already found visible — the production case — has nothing for the filter to remove,
// and streaming its 8 bytes first only to re-read them with the cone costs a third of the launch.
// DEFER (late pass with HiZ: the early form, LATE = BITS = false): frustum / cone ballots only, no tile counts — the occlusion
// stage (cluster_hiz_kernel) finishes the commands that have survivors; the visibility bits of the commands without any are
// cleared here (clustercull.comp.glsl:125-131 with visible == false), so that the stage touches 3 % of the commands, not all.
template <bool LATE, bool SOA, bool BITS, int CC_DA, bool DIRECT = false, bool DEFER = false, bool PACK = false>
__global__ __launch_bounds__(CC_THREADS, 6) void cluster_mask_kernel(ClusterArgs a)
{
	static_assert(!PACK || (DIRECT && SOA && !LATE), "the packed walk is a form of the direct early pass");
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t w = blockIdx.x * CC_WAVES + wave;

	// The packed walk's per-wave words (packed_walk): the table of the segment's non-empty commands (CP_ENTRY bytes each), the heads of the entry -> command
	// map, which PACK's windows overwrite with their `visible` ballots as the walk passes, and PACK's `not certainly outside` ballots (the launch's
	// statistic): 6.3 KB per wave, 25.1 KB per workgroup.
	__shared__ float4 s_tab[PACK ? CC_WAVES : 1][PACK ? 64 : 1][CP_ENTRY / 16];
	__shared__ uint64_t s_heads[PACK ? CC_WAVES : 1][PACK ? CP_WINDOWS : 1];
	__shared__ uint64_t s_nout[PACK && !BITS ? CC_WAVES : 1][PACK && !BITS ? CP_WINDOWS : 1];

	// late pass: the pyramid's level offsets in LDS, one copy per wave (written and read by the same wave: no barrier).
	// Built from the scalar kernel arguments with constant indices — a per-lane index into the argument array would be
	// a vector load, per probe a full memory latency in front of the texel fetch.
	__shared__ uint32_t s_mipTable[CC_WAVES][NV_MAX_MIPS];
	const uint32_t* s_mipOffset = s_mipTable[wave];
	if (LATE)
	{
		uint32_t off = 0;
#pragma unroll
		for (uint32_t i = 0; i < NV_MAX_MIPS; ++i)
			off = lane == i ? a.pyr.mipOffset[i] : off;
		if (lane < NV_MAX_MIPS)
			s_mipTable[wave][lane] = off;
	}

	// the start-up chain (count -> commands -> draws / first bounds) is latency-critical and a few dozen instructions
	// long: it must not queue behind the older waves' filter arithmetic (measured: without this the sixth workgroup of
	// a CU got its first data 11 k cycles after the first one)
	if (!NV_DBG(a, 262144u)) // bit 18 (experiments)
		__builtin_amdgcn_s_setprio(3);
	const uint32_t gen = a.genBlocks ? div_launch_constant(blockIdx.x, a.genBlocks, a.genBlocksMagic) : 0u; // (workgroups are numbered generation-major)
	// The three words of device memory every wave needs before it can ask for its commands — the indirect count, the pass's bank of tile counters,
	// the registered pool's bounds for the filter's per-draw margin (filtermath.h filter_make; no mirror: nothing is known, nothing is certain) —
	// requested together and waited for ONCE (round 5).  Left to hipcc each became a load at its first use with an s_waitcnt lgkmcnt(0) of its
	// own: two or three L2 round trips in a row on every wave's start-up path, where the launch is most sensitive.
	float poolVmax3 = __builtin_inff(), poolRmax = __builtin_inff();
	uint32_t rawGroups, bankWord;
#ifdef NV_PLAIN_LOADS
	bankWord = load_uniform_u32(&a.tileCounts->parity);
	rawGroups = load_uniform_u32(a.count4 + 1);
	if (SOA)
	{
		k_f32p pb = (k_f32p)(uintptr_t)a.poolBounds;
		poolVmax3 = pb[0];
		poolRmax = pb[1];
	}
#else
	if (SOA)
	{
		uint64_t pool;
		asm volatile("s_nop 4\n\ts_load_dword %0, %3, 0x4\n\ts_load_dword %1, %4, 0x0\n\ts_load_dwordx2 %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
		             : "=&s"(rawGroups), "=&s"(bankWord), "=&s"(pool)
		             : "s"(a.count4), "s"(&a.tileCounts->parity), "s"(a.poolBounds)
		             : "memory");
		poolVmax3 = __uint_as_float((uint32_t)pool);
		poolRmax = __uint_as_float((uint32_t)(pool >> 32));
	}
	else
		asm volatile("s_nop 4\n\ts_load_dword %0, %2, 0x4\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
		             : "=&s"(rawGroups), "=&s"(bankWord)
		             : "s"(a.count4), "s"(&a.tileCounts->parity)
		             : "memory");
#endif
	// vkCmdDispatchIndirect(dccb, 4): see indirect_command_count
	const uint32_t numCmds = a.commandCountOverride ? a.commandCountOverride : (rawGroups < 65535u ? rawGroups : 65535u) * 64u;
	const uint32_t bank = bankWord & 1u;
	// (nv_taskcull's payload form has its own word: no scatter launch follows it that would refresh the filter statistic in word 1, so
	// its count must not become the denominator of nv_clustercull's next filter / direct choice — ADVICE r3)
	if (a.hostHint && blockIdx.x == 0 && threadIdx.x == 0)
		__hip_atomic_store(a.hostHint + (a.payloadCounts ? 4 : 0), numCmds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	constexpr uint32_t CH = LATE ? CC_CHUNK_LATE : CC_CHUNK;
	// (late pass: even dealing — the weights are calibrated on the early pass, and the extra state costs the late variants two resident workgroups per CU)
	const bool weightedWanted = !LATE && !NV_DBG(a, 32768u); // bit 15 (experiments): even dealing
	const uint32_t planFlags = (weightedWanted ? DEAL_WEIGHTED_WANTED : 0u) | CH << 8;
	uint32_t chunkOf, myChunks, tileMul31, numTiles, gridWaves;
	bool dealtWeighted;
	{
		// (the plan's words merge here and die in deal_wave; with a deal_wave call per branch instead, the experiments build's early variants ran out of VGPRs)
		DealPlan plan;
		if (numCmds == a.plan.cmds && planFlags == a.plan.flags && !NV_DBG(a, 1073741824u)) // the host's guess holds (bit 30, experiments: never)
			plan = a.plan;
		else
			plan = deal_plan(numCmds, CH, weightedWanted, gridDim.x * CC_WAVES, a.cullWavesMagic, a.generations, a.genBlocks, gridDim.x, a.dealScale, a.scatterTiles, a.tilesMagic);
		myChunks = deal_wave(plan, w, lane, gen, a.genBlocks * CC_WAVES, &chunkOf);
		gridWaves = plan.waves;
		dealtWeighted = plan.weighted != 0;
		tileMul31 = plan.tileMul31;
		numTiles = plan.numTiles;
	}
	const uint32_t myCmds = myChunks * CH; // the last chunk of the pass may run past numCmds: guarded below
	if (w == 0 && lane == 0)
	{
		a.tileCounts->k2parity = bank;
		a.tileCounts->base = a.fusedReset ? 0u : a.clusterCount4[0];
	}

	// debugMode bit 3: per-wave s_memtime stamps into probeOut (tools/wave_timeline.py); never set in production
	const bool dbgTime = NV_DBG(a, 8u) && a.probeOut;
	unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.probeOut) + (size_t)w * 8;
#define NV_STAMP(i) do { if (dbgTime && lane == 0) stamps[i] = __builtin_readcyclecounter(); } while (0)
	NV_STAMP(0);
	if (dbgTime && lane == 0)
		stamps[6] = wall_clock64(); // 100 MHz, chip-wide: comparable across CUs (the cycle counter is not)

	uint32_t passedFilter = 0; // commands of this wave the conservative filter does not finish (DIRECT: would not have finished)
	// (the same per WAVE, in front of its first segment: NV_FILLER_PS / NV_FILLER_PV)
#if defined(NV_FILLER_PS)
#pragma unroll
	for (int f = 0; f < NV_FILLER_PS; ++f)
		asm volatile("s_cmp_eq_u32 0, 0" : : : "scc");
#endif
#if defined(NV_FILLER_PV)
#pragma unroll
	for (int f = 0; f < NV_FILLER_PV; ++f)
		asm volatile("v_nop");
#endif
	for (uint32_t seg = 0; seg < myCmds; seg += 64)
	{
		const uint32_t cnt = myCmds - seg < 64u ? myCmds - seg : 64u;

		// lane l holds the wave's (seg + l)-th command and (below) the MeshDraw it points at
		const uint32_t cidx = (seg + lane) / CH;
		const uint32_t chunk = dealtWeighted ? segment_chunk(chunkOf, seg >> 6) : cidx * gridWaves + w;
		const uint32_t myIdx = chunk * CH + (seg + lane) % CH;
		SegmentRegs r = {};
		if (lane < cnt && myIdx < numCmds)
		{
			const uint32_t* p = reinterpret_cast<const uint32_t*>(a.commands + myIdx);
			r.drawId = p[0];
			r.taskOffset = p[1];
			r.taskCount = p[2] < 64u ? p[2] : 64u; // (a taskCount above TASK_WGSIZE means 64: the reference's invocations are lanes 0 .. 63, clustercull.comp.glsl:66-70 — drawcull never writes one, a caller's list may)
			r.lateDrawVisibility = p[3];
			r.meshletVisibilityOffset = p[4];
		}
		if (!SOA)
		{
			if (r.taskCount)
			{
				const float4* d = reinterpret_cast<const float4*>(a.draws + r.drawId);
				r.d0 = d[0];
				r.d1 = d[1];
			}
			r.f = make_filter(a.cd, lane_draw(r), a.filterK, a.viewRowNorm, a.viewTransNorm, a.viewSum, poolVmax3, poolRmax); // lane-parallel: one filter per command of the segment
		}
		NV_STAMP(1);

		// SOA path: the MeshDraw gather is issued uncounted, AHEAD of the filter ring's first loads, and waited for
		// behind them (counted), so the dependent chain is commands -> {draws, first bounds} instead of
		// commands -> draws -> filters -> first bounds.
		constexpr bool FOLD_A = !LATE; // (the late variants have no eight VGPRs to spare under the six-workgroup bound)
		// (the folded rows evaluate cz f1 - |f0 cx| = cz f1 - |f0| |cx|, the reference cz f1 - |cx| f0: the same number only for f0, f2 >= 0.  A mirrored
		// or flipped projection — a negative side-plane coefficient — therefore takes no filter at all in the folded variants: every valid command goes to pass B, whose
		// certified test and reference arithmetic keep the coefficient's sign.  Written so that a NaN coefficient also fails it.)
		const bool foldSound = !FOLD_A || filter_fold_sound(a.cd.frustum);
		const bool useFilter = a.filterK > 0.0f && foldSound && !NV_DBG(a, 32u);   // filterK 0: coefficients outside the proven range (host); bit 5 (experiments): every valid command goes to the exact pass
		u32x4 g0 = {}, g1 = {};
		auto gather_issue = [&]()
		{
			const char* dp = reinterpret_cast<const char*>(a.draws + (r.taskCount ? r.drawId : 0u));
#ifdef NV_PLAIN_LOADS
			g0 = reinterpret_cast<const u32x4*>(dp)[0];
			g1 = reinterpret_cast<const u32x4*>(dp)[1];
#else
			asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(g0), "=&v"(g1) : "v"(dp) : "memory");
#endif
		};
		auto gather_finish = [&]()
		{
			r.d0 = make_float4(__uint_as_float(g0.x), __uint_as_float(g0.y), __uint_as_float(g0.z), __uint_as_float(g0.w));
			r.d1 = make_float4(__uint_as_float(g1.x), __uint_as_float(g1.y), __uint_as_float(g1.z), __uint_as_float(g1.w));
			r.f = make_filter<false>(a.cd, lane_draw(r), a.filterK, a.viewRowNorm, a.viewTransNorm, a.viewSum, poolVmax3, poolRmax); // (is127: in front of pass B)
			if (FOLD_A && !DIRECT) // (the direct forms have no filter loop)
			{
#pragma unroll
				for (int i = 0; i < 3; ++i)
				{
					r.fold[i] = a.cd.frustum[0] * r.f.m[i];
					r.fold[4 + i] = a.cd.frustum[2] * r.f.m[3 + i];
				}
				r.fold[3] = a.cd.frustum[0] * r.f.b[0];
				r.fold[7] = a.cd.frustum[2] * r.f.b[1];
			}
			if (!useFilter && !DIRECT) // an infinite margin: nothing is certainly outside.  (Here, lane-parallel once per segment: as a test where the filter loop
				r.f.tK = __builtin_inff(); // broadcasts a draw's filter, hipcc materialised the uniform flag with two vector instructions per COMMAND.  The direct forms have no filter loop; the packed walk's certified test reads tK.)
		};

		const bool useCert = a.filterK > 0.0f && !NV_DBG(a, 1048576u); // bit 20 (experiments): pass B with the reference arithmetic only
		const bool streamOnly = NV_DBG(a, 64u); // bit 6 (experiments): no arithmetic at all
		const bool updateBits = LATE && a.cd.clusterOcclusionEnabled == 1;
		constexpr bool BITS_A = BITS && !LATE; // the filter pass needs the visibility words only for the early pass's bit test
		uint32_t maskLo = 0, maskHi = 0; // lane c ends up holding the ballot of the segment's c-th command (v_writelane)
		uint32_t visLo = 0, visHi = 0;   // late pass: likewise the `visible` ballot, for the visibility-bit update

		if (SOA)
		{
			// hipcc must have waited for its own segment loads before the first uncounted load is issued
			asm volatile("" : "+v"(r.drawId), "+v"(r.taskOffset), "+v"(r.meshletVisibilityOffset), "+v"(r.taskCount), "+v"(r.lateDrawVisibility));
			// (PACK: the scan over the commands' sizes in front of the first uncounted load — in the experiments build it carries an exec-mask assertion, i.e. a
			// path out of the kernel, which must not start with loads in flight)
			uint32_t packIncl = 0;
			if constexpr (PACK && !BITS)
				packIncl = wave_scan_inclusive_u32(r.taskCount < 64u ? r.taskCount : 64u);
			gather_issue();

			// ---- pass A: stream the 8 bounds bytes of every command through the conservative frustum filter.
			// Commands with no possible survivor are finished here (ballot 0); the rest are queued in candMask.
			// ---- per-segment scalar summaries (one ballot each) so that the walk needs no per-command v_readlane for
			// control: which commands are full (64 meshlets), empty (dummy), or start a new draw
			const uint64_t fullMask = __ballot(r.taskCount >= 64u);
			// "the draw changes" must be judged on the draw whose filter the lane actually holds: an empty command
			// (taskCount 0) gathered draw 0 above, whatever its drawId says (tests/test_special_values.py)
			const uint32_t heldDraw = r.taskCount ? r.drawId : 0u;
			const uint32_t prevDraw = wave_shift_up1_u32(heldDraw);
			const uint64_t changeMask = __ballot(lane == 0 || heldDraw != prevDraw) | 1ull;
			const uint32_t base8 = (r.taskCount ? r.taskOffset : 0u) * 8u; // lane-parallel: byte offset of each command's bounds
			const uint32_t lane8 = lane * 8u;

			uint64_t candMask = 0;
			FilterUniform fd = {};

			// `full` = the command in lane c has 64 meshlets (the usual case: no per-lane clamp of the offsets).  The filter loop passes the
			// bit from a per-round nibble of fullMask (one 64-bit shift per four commands instead of one per command: the loop is bound by
			// scalar and vector ISSUE, and of its ~30 scalar instructions per command a third were 64-bit mask tests and index clamps — round 5).
			auto issueA = [&](SlotA& slot, uint32_t c, bool full, uint64_t order)
			{
				uint32_t off8 = __builtin_amdgcn_readlane(base8, c) + lane8, offw = 0;
				if (!full) // (one rarely taken branch; an if / else here became two flag tests per command)
				{
					const uint32_t tc = __builtin_amdgcn_readlane(r.taskCount, c);
					off8 = __builtin_amdgcn_readlane(base8, c) + (lane < tc ? lane8 : 0u);
				}
				if (BITS_A)
				{
					const uint32_t tc = __builtin_amdgcn_readlane(r.taskCount, c);
					const uint32_t mvo = __builtin_amdgcn_readlane(r.meshletVisibilityOffset, c);
					offw = tc ? ((mvo + (lane < tc ? lane : 0u)) >> 5) * 4u : 0u;
				}
				// Sensitivity of the launch to its instruction mix ("no experiment has isolated SALU count"): builds with
				// -DNV_FILLER_S=n / -DNV_FILLER_V=n issue n more scalar / vector instructions per command here (register-free, results
				// unchanged; tools/experiments/filler_sensitivity.sh).  Compile-time: a run-time switch in this lambda — even around
				// an empty statement — costs the kernel 13 VGPRs and a scratch frame.
#if defined(NV_FILLER_S)
#pragma unroll
				for (int f = 0; f < NV_FILLER_S; ++f)
					asm volatile("s_cmp_eq_u32 0, 0" : : : "scc");
#endif
#if defined(NV_FILLER_V)
#pragma unroll
				for (int f = 0; f < NV_FILLER_V; ++f)
					asm volatile("v_nop");
#endif
				ringA_issue<BITS_A>(slot, a, off8, offw, order);
			};

			if constexpr (PACK)
			{
				// ---- the direct form as a packed walk (packed_walk above): the segment's valid meshlets in windows of 64.
				// (Unconditional, also for a segment of empty commands only — E = 0, no window, the ring's requests re-read meshlet 0: a branch around the walk
				// would give the gather a second wait site, and hipcc joins the two with copies of registers whose loads are still in flight.)
				uint32_t tcc = r.taskCount; // (<= 64; lanes without a command hold 0)
				if constexpr (BITS)
				{
					// The early pass with visibility bits (clustercull.comp.glsl:86-95): a command none of whose clusters was visible last frame has no entry at
					// all, and a window's lanes whose bit is clear decide nothing.  The commands' visibility words — three per command at most — are requested
					// behind the MeshDraw gather and waited for together with it: the map needs them (chain: commands -> {draws, words} -> windows).
					const uint32_t mvo = r.meshletVisibilityOffset;
					const uint32_t wFirst = tcc ? mvo >> 5 : 0u, wLast = tcc ? (mvo + tcc - 1u) >> 5 : 0u;
					uint32_t w0, w1, w2;
					{
						const uint32_t o0 = wFirst * 4u, o1 = (wFirst + 1u < wLast ? wFirst + 1u : wLast) * 4u, o2 = (wFirst + 2u < wLast ? wFirst + 2u : wLast) * 4u;
#ifdef NV_PLAIN_LOADS
						w0 = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.mvb) + o0);
						w1 = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.mvb) + o1);
						w2 = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.mvb) + o2);
#else
						asm volatile("s_nop 4\n\tglobal_load_dword %0, %3, %6\n\tglobal_load_dword %1, %4, %6\n\tglobal_load_dword %2, %5, %6"
						             : "=&v"(w0), "=&v"(w1), "=&v"(w2)
						             : "v"(o0), "v"(o1), "v"(o2), "s"(a.mvb)
						             : "memory");
#endif
					}
					NV_COUNTED_WAIT("s_waitcnt vmcnt(%5) ; nv_ready %0 %1 %2 %3 %4" : "+v"(g0), "+v"(g1), "+v"(w0), "+v"(w1), "+v"(w2) : "i"(0) : "memory"); // the gather and the words
					const uint32_t sh = mvo & 31u;
					const uint32_t lo = sh ? (w0 >> sh) | (w1 << (32u - sh)) : w0, hi = sh ? (w1 >> sh) | (w2 << (32u - sh)) : w1;
					const uint64_t valid = tcc >= 64u ? ~0ull : (1ull << tcc) - 1ull;
					if (((((uint64_t)hi << 32) | lo) & valid) == 0)
						tcc = 0; // nothing of this command can be visible
					packIncl = wave_scan_inclusive_u32(tcc); // (every uncounted load has landed: the experiments build's exec-mask assertion may leave here)
				}
				candMask = __ballot(tcc != 0); // every command that has an entry
				const uint32_t excl = packIncl - tcc; // the command's first entry
				const uint32_t E = (uint32_t)__builtin_amdgcn_readlane((int)packIncl, 63);
				const uint32_t rankC = __builtin_amdgcn_mbcnt_hi((uint32_t)(candMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)candMask, 0u)); // among the commands with entries
				char* tabBytes = reinterpret_cast<char*>(s_tab[wave]);
				char* myEntry = tabBytes + rankC * CP_ENTRY;
				uint64_t* heads = s_heads[wave];
				uint64_t* nout = BITS ? nullptr : s_nout[wave];
				if (tcc || (candMask == 0 && lane == 0)) // (no command at all: entry 0 = meshlet 0, visibility word 0)
				{
					*reinterpret_cast<uint2*>(myEntry + CP_ENTRY_HEAD) = make_uint2(tcc ? r.taskOffset - excl : 0u, tcc ? r.meshletVisibilityOffset - excl : 0u);
					*reinterpret_cast<uint32_t*>(myEntry + CP_ENTRY_HEAD + 8) = excl | lane << 16;
				}
				heads[lane] = 0ull;
				if (lane < (uint32_t)CP_WINDOWS - 64u)
					heads[64u + lane] = 0ull;
				// (LDS serves a wave's operations in order; the statements only keep hipcc from reordering what it sees as accesses of different lanes)
				asm volatile("" ::: "memory");
				if (tcc && excl)
					atomicOr(reinterpret_cast<uint32_t*>(heads) + ((excl - 1u) >> 5), 1u << ((excl - 1u) & 31u));
				asm volatile("" ::: "memory");
				packed_walk<BITS>(a, tabBytes, heads, nout, lane, E, r.drawId, a.filterK > 0.0f && !NV_DBG(a, 1048576u), [&]() // bit 20 (experiments): the reference arithmetic only
				{
					if constexpr (!BITS)
						NV_COUNTED_WAIT("s_waitcnt vmcnt(%2) ; nv_ready %0 %1" : "+v"(g0), "+v"(g1) : "i"(CP_DB * 2) : "memory"); // the gather
					gather_finish();
					r.f.is127 = filter_is127(r.f.scale);
					if (tcc)
						walk_coefficients(myEntry, r.f);
					asm volatile("" ::: "memory");
				});
				// ---- the commands' ballots, cut out of the windows' (lane c = the segment's c-th command)
				asm volatile("" ::: "memory");
				{
					const uint32_t jc = excl >> 6, sh = excl & 63u;
					const uint64_t need = tcc >= 64u ? ~0ull : (1ull << tcc) - 1ull;
					const uint64_t vlo = heads[jc], vhi = heads[jc + 1u];
					const uint64_t m = (sh ? (vlo >> sh) | (vhi << (64u - sh)) : vlo) & need;
					maskLo = (uint32_t)m;
					maskHi = (uint32_t)(m >> 32);
					if constexpr (BITS)
						passedFilter += (uint32_t)__builtin_popcountll(candMask); // (cluster_bits_kernel's statistic: a command with a set bit nearly always still has a cluster the filter cannot finish)
					else
					{
						const uint64_t nlo = nout[jc], nhi = nout[jc + 1u];
						const uint64_t nm = (sh ? (nlo >> sh) | (nhi << (64u - sh)) : nlo) & need;
						passedFilter += (uint32_t)__builtin_popcountll(__ballot(nm != 0)); // what pass A's filter would not have finished
					}
				}
				asm volatile("" ::: "memory");
			}
			else if (DIRECT)
			{
				NV_COUNTED_WAIT("s_waitcnt vmcnt(%2) ; nv_ready %0 %1" : "+v"(g0), "+v"(g1) : "i"(0) : "memory"); // the gather
				gather_finish();
				candMask = __ballot(lane < cnt && myIdx < numCmds && r.taskCount != 0); // every valid command
			}
			else
			{
				SlotA ring[CC_DA];
#pragma unroll
				for (int k = 0; k < CC_DA; ++k)
					issueA(ring[k], k, (fullMask >> k & 1ull) != 0, 0); // lanes >= cnt hold an empty command (meshlet 0): redundant but unconditional, in-range loads
				NV_COUNTED_WAIT("s_waitcnt vmcnt(%2) ; nv_ready %0 %1" : "+v"(g0), "+v"(g1) : "i"(CC_DA * (BITS_A ? 2 : 1)) : "memory"); // the gather
				gather_finish();
				NV_STAMP(2);
				for (uint32_t i = 0; i < cnt; i += CC_DA)
				{
					// The SIMD arbitrates oldest-first, which lets the oldest resident workgroup run ahead and leaves
					// the youngest to finish alone at single-wave issue rate.  Rotating the priority, offset by the
					// workgroup's generation so that the workgroups sharing a CU hold different levels at any time,
					// evens the progress of the waves that share a SIMD (speed only).
					if (!NV_DBG(a, 256u)) // bit 8 (experiments) turns the rotation off
					{
						switch ((NV_DBG(a, 131072u) ? blockIdx.x + i / CC_DA : gen + i / CC_DA) & 3u) // bit 17 (experiments): rotation without the generation offset
						{
						case 0: __builtin_amdgcn_s_setprio(0); break;
						case 1: __builtin_amdgcn_s_setprio(1); break;
						case 2: __builtin_amdgcn_s_setprio(2); break;
						default: __builtin_amdgcn_s_setprio(3); break;
						}
					}
					// One straight body per command (pass A is bound by scalar and vector ISSUE, not by memory): the filter
					// runs on all 64 lanes of every slot — clamped loads make that harmless for partial, dummy and
					// out-of-range commands — and validity is applied to the ballot with scalar mask arithmetic.
					// the round's bits of the three per-command masks, as nibbles: the commands of this round (change, full) and of the next
					// one, whose loads this round issues (lanes >= cnt hold empty commands — meshlet 0, in range — so the index of a load
					// needs no clamp, only the wrap at the end of a full segment: (i + CC_DA) & 63 re-reads the segment's first commands)
					const uint32_t nextBase = (i + CC_DA) & 63u;
					const uint32_t chgN = (uint32_t)(changeMask >> i), fullN = (uint32_t)(fullMask >> i), fullNext = (uint32_t)(fullMask >> nextBase);
					uint32_t candN = 0;
#pragma unroll
					for (int k = 0; k < CC_DA; ++k)
					{
						const uint32_t c = i + k;
						ringA_wait<BITS_A, CC_DA - 1>(ring[k]);
						const uint32_t b0 = (uint32_t)ring[k].bounds, b1 = (uint32_t)(ring[k].bounds >> 32);
						if (chgN & (1u << k)) // first command of a draw within this segment
						{
							fd = segment_filter<FOLD_A>(r, c);
						}
						uint64_t cand = ~__ballot(certainly_outside<FOLD_A>(a.cd, fd, b0, b1));
						if (!(fullN & (1u << k))) // partial, dummy (taskCount 0) or past the wave's last command (lanes >= cnt hold 0)
							cand &= (1ull << (uint32_t)__builtin_amdgcn_readlane(r.taskCount, c)) - 1ull;
						if (BITS_A) // early pass: only last frame's visible clusters (clustercull.comp.glsl:91-92)
							cand &= __ballot((ring[k].mvbWord >> ((lane + (uint32_t)__builtin_amdgcn_readlane(r.meshletVisibilityOffset, c)) & 31u) & 1u) != 0);
						if (streamOnly)
							cand = 0;
						candN |= cand ? 1u << k : 0u;
						issueA(ring[k], nextBase + k, (fullNext & (1u << k)) != 0, cand);
					}
					candMask |= (uint64_t)candN << i;
				}
				ring_drain();
#pragma unroll
				for (int k = 0; k < CC_DA; ++k)
					ring_release(ring[k]);
				passedFilter += (uint32_t)__builtin_popcountll(candMask);
			}
			NV_STAMP(3);
			if (dbgTime && lane == 0 && NV_DBG(a, 134217728u)) // bit 27 (experiments): slot 2 = the segment's candidate count instead of the ring-filled stamp (tools/experiments/passb_cost.py)
				stamps[2] = (unsigned long long)__builtin_popcountll(candMask);

			// ---- pass B: the commands that can have survivors, bounds + cone.  (Measured and dropped: raising a wave to the
			// top priority for pass B, because it is on the launch's critical path — late pass 44.8 -> 48.8 us, early pass
			// unchanged: the boosted wave takes issue slots from the streaming waves that keep HBM busy; likewise
			// keeping level 3 out of the streaming waves' rotation: early pass +1.5 us.)
			if (!PACK && candMask && !NV_DBG(a, 1024u)) // bit 10 (experiments): no exact pass.  (PACK: the walk above was the exact pass)
			{
				uint32_t curDraw = ~0u, certDraw = ~0u;
				DrawUniform du = {};
				CertUniform cf = {};
				r.f.is127 = filter_is127(r.f.scale); // lane-parallel, for the certified cone test: only segments that have a candidate pay the division
				const bool certFinal = useCert && !(LATE && a.cd.clusterOcclusionEnabled == 1); // (HiZ decides after frustum and cone)
				// one candidate command c with its landed lane data: the certified test, the reference arithmetic if a lane that matters sits inside a
				// margin; returns the ballot (also written to lane c of maskLo / maskHi, and of visLo / visHi)
				auto exact_command = [&](uint32_t c, const LaneData& cur) -> uint64_t
				{
					const NvMeshTaskCommand cmd = segment_command(r, c);
					uint64_t vis = 0, m = 0;
					bool decided = NV_DBG(a, 4096u); // bit 12 (experiments): exact pass loads only
					if (certFinal && !decided)
					{
						if (cmd.drawId != certDraw)
						{
							certDraw = cmd.drawId;
							cf = segment_cert(r, c);
						}
						uint64_t need = cmd.taskCount >= 64u ? ~0ull : (1ull << cmd.taskCount) - 1ull, skipM = 0;
						if (BITS) // clustercull.comp.glsl:86-99
						{
							const uint64_t bitM = __ballot((cur.mvbWord >> ((lane + cmd.meshletVisibilityOffset) & 31u) & 1u) != 0);
							if (!LATE)
								need &= bitM;
							else if (cmd.lateDrawVisibility == 1)
								skipM = bitM;
						}
						bool rejects = true;
						if (need == 0) // early pass: none of the command's clusters was visible last frame (clustercull.comp.glsl:91-92) — nothing to test
							decided = true;
						else
							decided = certified_visible(a.cd, cf, cur.b0, cur.b1, cur.cone, need, &vis, &rejects);
						m = vis & ~skipM;
						if (DIRECT && !rejects)
							++passedFilter;
					}
					else if (DIRECT)
						++passedFilter;
					if (!decided) // some lane sits inside a margin (or the test is off): the reference arithmetic for the whole wave
					{
						if (cmd.drawId != curDraw)
						{
							curDraw = cmd.drawId;
							du = load_draw(a.draws, cmd.drawId); // scalar loads: the gathered copy lives only until the filters are derived
						}
						m = cull_command<LATE, BITS, LATE>(a, cmd, du, cur, lane, &vis, s_mipOffset);
					}
					maskLo = writelane_u32(maskLo, (uint32_t)m, c);
					maskHi = writelane_u32(maskHi, (uint32_t)(m >> 32), c);
					if (updateBits)
					{
						visLo = writelane_u32(visLo, (uint32_t)vis, c);
						visHi = writelane_u32(visHi, (uint32_t)(vis >> 32), c);
					}
					return m;
				};
				{
				constexpr int DB = DIRECT ? CC_DB_DIRECT : CC_DB;
				SlotB ring[DB];
				if (DIRECT)
				{
					// Every command of the segment is a candidate: the walk is c = 0, 1, 2, ... — no pending mask, no find-first-bit, no per-slot
					// command registers, no "one more round" bookkeeping (round 5; VERDICT r4 item 3b: 63 scalar instructions per command in this
					// form).  Dummy commands and the lanes past the wave's last command hold taskCount 0: nothing to test, their loads
					// re-read meshlet 0; a slot index past the segment wraps to its first commands (redundant, in range, like the filter ring's).
#pragma unroll
					for (int k = 0; k < DB; ++k)
						ringB_issue<BITS>(ring[k], a, __builtin_amdgcn_readlane(r.taskOffset, k), __builtin_amdgcn_readlane(r.taskCount, k),
						                  __builtin_amdgcn_readlane(r.meshletVisibilityOffset, k), lane, 0);
					for (uint32_t c0 = 0; c0 < cnt; c0 += DB)
					{
#pragma unroll
						for (int k = 0; k < DB; ++k)
						{
							ringB_wait<BITS, DB - 1>(ring[k]);
							const uint32_t c = c0 + k;
							uint64_t m = 0;
							// (an empty command is skipped as a whole, not tested with need == 0: its lane holds the filter of draw 0 — what its gather
							// read — under its own drawId, which the next command of that draw must not take for its own: tests/test_lane_form.py)
							if (c < cnt && __builtin_amdgcn_readlane(r.taskCount, c & 63u) != 0)
							{
								LaneData cur;
								cur.b0 = (uint32_t)ring[k].bounds;
								cur.b1 = (uint32_t)(ring[k].bounds >> 32);
								cur.cone = ring[k].cone;
								cur.mvbWord = ring[k].mvbWord;
								m = exact_command(c, cur);
							}
							const uint32_t nx = (c + DB) & 63u;
							ringB_issue<BITS>(ring[k], a, __builtin_amdgcn_readlane(r.taskOffset, nx), __builtin_amdgcn_readlane(r.taskCount, nx),
							                  __builtin_amdgcn_readlane(r.meshletVisibilityOffset, nx), lane, m);
						}
					}
				}
				else
				{
				uint64_t pending = candMask; // commands not yet issued into the ring
				uint32_t cIssued[CC_DB];
				uint32_t last = (uint32_t)__builtin_ctzll(candMask);
#pragma unroll
				for (int k = 0; k < CC_DB; ++k)
				{
					if (pending)
					{
						last = (uint32_t)__builtin_ctzll(pending);
						pending &= pending - 1;
						cIssued[k] = last;
					}
					else
						cIssued[k] = ~0u; // nothing left: the slot re-reads the last command and is ignored
					ringB_issue<BITS>(ring[k], a, __builtin_amdgcn_readlane(r.taskOffset, last), __builtin_amdgcn_readlane(r.taskCount, last),
					                  __builtin_amdgcn_readlane(r.meshletVisibilityOffset, last), lane, 0);
				}
				for (bool more = true; more;)
				{
					more = false;
#pragma unroll
					for (int k = 0; k < CC_DB; ++k)
					{
						ringB_wait<BITS, CC_DB - 1>(ring[k]);
						const uint32_t c = cIssued[k];
						uint64_t m = 0;
						if (c != ~0u)
						{
							LaneData cur;
							cur.b0 = (uint32_t)ring[k].bounds;
							cur.b1 = (uint32_t)(ring[k].bounds >> 32);
							cur.cone = ring[k].cone;
							cur.mvbWord = ring[k].mvbWord;
							m = exact_command(c, cur);
						}
						if (pending)
						{
							last = (uint32_t)__builtin_ctzll(pending);
							pending &= pending - 1;
							cIssued[k] = last;
							more = true;
						}
						else
							cIssued[k] = ~0u;
						ringB_issue<BITS>(ring[k], a, __builtin_amdgcn_readlane(r.taskOffset, last), __builtin_amdgcn_readlane(r.taskCount, last),
						                  __builtin_amdgcn_readlane(r.meshletVisibilityOffset, last), lane, m);
					}
					// slots issued in this round still hold commands: one more round consumes them
					if (!more)
#pragma unroll
						for (int k = 0; k < CC_DB; ++k)
							more = more || cIssued[k] != ~0u;
				}
				}
				ring_drain();
#pragma unroll
				for (int k = 0; k < DB; ++k)
					ring_release(ring[k]);
				}
			}
		}
		else
		{
			// AoS records read in place (no mirror registered): compiler-scheduled loads, one command ahead
			uint32_t curDraw = ~0u;
			DrawUniform du = {};
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
				const NvMeshTaskCommand cmd = segment_command(r, c);
				uint64_t m = 0;
				if (cmd.taskCount)
				{
					if (cmd.drawId != curDraw)
					{
						curDraw = cmd.drawId;
						du = segment_draw(r, c);
					}
					uint64_t vis = 0;
					m = cull_command<LATE, BITS, LATE>(a, cmd, du, cur, lane, &vis, s_mipOffset);
					if (updateBits)
					{
						visLo = writelane_u32(visLo, (uint32_t)vis, c);
						visHi = writelane_u32(visHi, (uint32_t)(vis >> 32), c);
					}
				}
				maskLo = writelane_u32(maskLo, (uint32_t)m, c);
				maskHi = writelane_u32(maskHi, (uint32_t)(m >> 32), c);
			}
		}
		NV_STAMP(4);

		// the segment's ballots: one 8-B store per lane (32-B runs per chunk), after the rings have drained so that no
		// store sits between counted loads
		if (updateBits) // clustercull.comp.glsl:125-131 for the whole segment
			update_segment_visibility(a, lane < cnt && myIdx < numCmds, r.meshletVisibilityOffset, r.taskCount, ((uint64_t)visHi << 32) | visLo);
		if (DEFER)
		{
			// no survivor of frustum and cone: nothing is visible, whatever the pyramid holds
			const bool liveCmd = lane < cnt && myIdx < numCmds;
			update_segment_visibility(a, liveCmd && (maskLo | maskHi) == 0, r.meshletVisibilityOffset, r.taskCount, 0ull);
			// the others go on the occlusion stage's list: one returning add per segment that has any.  CC_LISTS sub-lists
			// with a counter each (one list head measured +10 us on this kernel: ~2000 adds to one address are served one
			// after the other at the memory side); the order within a sub-list is whatever the adds make it — the stage
			// writes per command, so the results do not depend on it.  A sub-list that runs out of room raises a flag
			// and the stage scans all commands instead: the list only saves time.
			const uint64_t has = __ballot(liveCmd && (maskLo | maskHi) != 0);
			if (has)
			{
				const uint32_t n = (uint32_t)__builtin_popcountll(has);
				const uint32_t sub = w & (CC_LISTS - 1u);
				uint32_t slot = 0;
				if (lane == 0)
					slot = atomicAdd(&a.tileCounts->listCount[bank][sub * CC_COUNT_STRIDE], n);
				slot = __builtin_amdgcn_readfirstlane(slot);
				if (slot + n <= a.listStride)
				{
					if (has >> lane & 1ull) // an entry = the command's index, its five words and its ballot (32 B): the stage starts from it alone
					{
						uint4* e = a.candList + (size_t)(sub * a.listStride + slot + __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0u))) * 2u;
						e[0] = make_uint4(myIdx, r.drawId, r.taskOffset, r.taskCount);
						e[1] = make_uint4(r.lateDrawVisibility, r.meshletVisibilityOffset, maskLo, maskHi);
					}
				}
				else if (lane == 0)
					atomicOr(&a.tileCounts->listOverflow[bank], 1u);
			}
		}
		if (lane < cnt && myIdx < numCmds)
		{
			const uint64_t m = ((uint64_t)maskHi << 32) | maskLo;
			a.masks[myIdx] = m;
		}
		const bool payloadForm = !LATE && !DEFER && a.payloadCounts != nullptr; // (uniform) nv_taskcull's early pass: see below
		if (payloadForm)
		{
			// The task-shader form (meshlet.task.glsl:135-143) straight from this launch: a command's survivors compacted into its own
			// 64-entry payload, its count beside it — no ordered append, so no tile counts and no second launch.  The segment's ballots
			// are final here and the rings have drained; the commands that have survivors are walked with the lanes on a command's
			// 64 bits (rank = v_mbcnt: one contiguous store per command).
			const bool liveCmd = lane < cnt && myIdx < numCmds;
			const uint32_t mlo = liveCmd ? maskLo : 0u, mhi = liveCmd ? maskHi : 0u;
			if (liveCmd)
				a.payloadCounts[myIdx] = (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi);
			for (uint64_t owners = __ballot((mlo | mhi) != 0); owners; owners &= owners - 1)
			{
				const uint32_t src = (uint32_t)__builtin_ctzll(owners);
				const uint32_t lo = __builtin_amdgcn_readlane(mlo, src), hi = __builtin_amdgcn_readlane(mhi, src);
				const uint32_t ci = __builtin_amdgcn_readlane(myIdx, src);
				if ((lane < 32u ? lo >> lane : hi >> (lane - 32u)) & 1u)
					a.clusterIndices[(size_t)ci * 64 + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u))] = ci | (lane << 24);
			}
		}
		// survivors per scatter tile: fire-and-forget adds.  The four commands of a chunk sit in four neighbouring lanes
		// and nearly always in one tile: their counts are summed across the quad first (3x fewer atomics).
		{
			const bool live = lane < cnt && myIdx < numCmds;
			const uint64_t m = live ? ((uint64_t)maskHi << 32) | maskLo : 0ull;
			const uint32_t tileOf = live ? deal_tile_of(myIdx, tileMul31) : ~0u;
			uint32_t pc = (uint32_t)__builtin_popcountll(m);
			const uint32_t tile0 = quad_first_u32(tileOf);
			const bool quadUniform = __all_quad_same(tileOf, tile0);
			if (quadUniform)
			{
				pc += quad_xor1_u32(pc);
				pc += quad_xor2_u32(pc);
				if (lane & 3u)
					pc = 0;
			}
			if (pc && !DEFER && !payloadForm && !NV_DBG(a, 2048u)) // DEFER: the occlusion stage counts the final ballots; payloads: nothing is appended; bit 11 (experiments): no tile counts
				atomicAdd(&a.tileCounts->counts[bank][tileOf * CC_COUNT_STRIDE], pc);
		}
	}
	// the launch's filter statistic for the host's choice of the next launch's form: one add per wave, spread over the tile
	// counters' lines (word 1 of a line; the scatter kernel sums them)
	if (lane == 0 && passedFilter && !(!LATE && !DEFER && a.payloadCounts != nullptr)) // (payloads: no scatter launch follows that would sum and clear them)
	{
		// (any of the lines of the tiles that hold commands — the scatter launch sums them all: the largest power of two of them, a mask instead of a remainder)
		const uint32_t spread = numTiles ? (1u << (31 - __builtin_clz(numTiles))) - 1u : 0u;
		atomicAdd(&a.tileCounts->counts[bank][(w & spread) * CC_COUNT_STRIDE + 1], passedFilter);
	}
	NV_STAMP(5);
	if (dbgTime && lane == 0)
		stamps[7] = wall_clock64();
#undef NV_STAMP
}

// K2: contiguous ranges, ordered scatter.  Tile t's append base = count word + survivors of tiles < t, which the cull
// kernel has already accumulated per tile: no workgroup waits on another, the kernel is a handful of parallel loads,
// one scan and the stores.  A step covers 1024 commands: 16 waves with one command per lane (the usual shape: the stores
// are emitted one owning command at a time per wave, so with one wave per SIMD — 4 waves x 4 commands per lane — a
// workgroup spent 3.7 of its 8 us walking ~25 owners per wave at single-wave issue latency), or 4 waves with four.
template <int SC_WAVES>
__global__ __launch_bounds__(SC_WAVES * 64) void cluster_scatter_kernel(ClusterArgs a)
{
	constexpr uint32_t SC_THREADS = SC_WAVES * 64;
	__shared__ uint32_t s_part[SC_WAVES];
	__shared__ uint32_t s_sum[SC_WAVES];
	constexpr uint32_t STEP = 1024, PER_LANE = STEP / SC_THREADS;
	constexpr int TILE_LOADS = (int)((CC_MAX_SCATTER_TILES + SC_THREADS - 1) / SC_THREADS); // tile counters per thread and bank

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

	const uint32_t numCmds = indirect_command_count(a);
	const uint32_t T = scatter_tile_commands(numCmds, a.scatterTiles, a.tilesMagic);
	const uint32_t numTiles = (numCmds + T - 1) / T; // <= gridDim.x
	const uint32_t tile = blockIdx.x;
	const bool dbgNoScatter = NV_DBG(a, 4u); // experiments only

	// Everything below was written by the cull kernel, i.e. before this launch: plain loads, all issued together.
	// Both banks of tile counts are read speculatively so that no load waits for the parity word.
	const uint32_t k2parity = load_uniform_u32(&a.tileCounts->k2parity);
	const uint32_t base0 = load_uniform_u32(&a.tileCounts->base);
	uint32_t cnt0[TILE_LOADS] = {}, cnt1[TILE_LOADS] = {}; // this thread's tiles tid, tid + SC_THREADS, ..., per bank
	uint32_t pf0[TILE_LOADS] = {}, pf1[TILE_LOADS] = {};   // likewise the cull kernel's filter statistic (word 1 of the line)
#pragma unroll
	for (int j = 0; j < TILE_LOADS; ++j)
	{
		const uint32_t i = j * SC_THREADS + tid;
		if (i < numTiles)
		{
			cnt0[j] = a.tileCounts->counts[0][i * CC_COUNT_STRIDE];
			cnt1[j] = a.tileCounts->counts[1][i * CC_COUNT_STRIDE];
			pf0[j] = a.tileCounts->counts[0][i * CC_COUNT_STRIDE + 1];
			pf1[j] = a.tileCounts->counts[1][i * CC_COUNT_STRIDE + 1];
		}
	}
	const uint32_t first = tile * T;
	const uint32_t n = tile < numTiles ? (numCmds - first < T ? numCmds - first : T) : 0u;
	// first step's ballots (the only step for the usual T <= 1024); the ballot array is padded, so the 16-B loads of
	// a partially valid quad stay in range and are masked afterwards
	uint64_t m4[PER_LANE];
	if (PER_LANE == 4)
	{
		const uint32_t c = tid * PER_LANE;
		const ulonglong2* src = reinterpret_cast<const ulonglong2*>(a.masks + first + c);
		ulonglong2 lo = make_ulonglong2(0, 0), hi = make_ulonglong2(0, 0);
		if (c < n)
		{
			lo = src[0];
			hi = src[1];
		}
		m4[0] = c + 0 < n ? lo.x : 0ull;
		m4[PER_LANE > 1 ? 1 : 0] = c + 1 < n ? lo.y : 0ull;
		m4[PER_LANE > 2 ? 2 : 0] = c + 2 < n ? hi.x : 0ull;
		m4[PER_LANE > 3 ? 3 : 0] = c + 3 < n ? hi.y : 0ull;
	}
	else
	{
#pragma unroll
		for (uint32_t j = 0; j < PER_LANE; ++j)
			m4[j] = tid * PER_LANE + j < n ? a.masks[first + tid * PER_LANE + j] : 0ull;
	}

	const uint32_t bank = k2parity & 1u;
	// every workgroup clears its entries of the other bank for the next pass; one thread flips the parity the next
	// cull kernel will read (this pass reads k2parity only)
	for (uint32_t i = tile * SC_THREADS + tid; i < CC_MAX_SCATTER_TILES; i += gridDim.x * SC_THREADS)
	{
		a.tileCounts->counts[bank ^ 1u][i * CC_COUNT_STRIDE] = 0;
		a.tileCounts->counts[bank ^ 1u][i * CC_COUNT_STRIDE + 1] = 0;
		if (i < CC_LISTS)
			a.tileCounts->listCount[bank ^ 1u][i * CC_COUNT_STRIDE] = 0;
	}
	if (tile == 0 && tid == 0)
	{
		a.tileCounts->parity = bank ^ 1u;
		a.tileCounts->listOverflow[bank ^ 1u] = 0;
		if (numTiles == 0 && a.hostHint)
			__hip_atomic_store(a.hostHint + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		if (numTiles == 0) // no commands at all: the count word keeps its base, the submit words describe an empty grid
		{
			if (a.fusedReset)
				a.clusterCount4[0] = 0;
			if (a.countsSink)
			{
				a.countsSink[0] = 0;
				a.countsSink[1] = a.count4[0];
				a.countsSink[2] = a.fusedReset ? 0u : a.clusterCount4[0];
			}
			if (a.fusedSubmit)
			{
				const uint32_t raw = a.fusedReset ? 0u : a.clusterCount4[0];
				const uint32_t count = raw < NV_CLUSTER_LIMIT ? raw : NV_CLUSTER_LIMIT;
				const uint32_t gy = (count + 255u) / 256u;
				a.clusterCount4[1] = NV_CLUSTER_TILE;
				a.clusterCount4[2] = gy < 65535u ? gy : 65535u;
				a.clusterCount4[3] = 256u / NV_CLUSTER_TILE;
				for (uint32_t i = count; i < ((count + 255u) & ~255u); ++i)
					a.clusterIndices[i] = ~0u;
			}
		}
	}
	if (tile >= numTiles)
		return;

	uint32_t before = 0, all = 0, passed = 0;
#pragma unroll
	for (int j = 0; j < TILE_LOADS; ++j)
	{
		const uint32_t i = j * SC_THREADS + tid;
		const uint32_t v = bank ? cnt1[j] : cnt0[j];
		all += v;
		before += i < tile ? v : 0u;
		passed += bank ? pf1[j] : pf0[j];
	}
	if (tile == numTiles - 1 && a.hostHint) // (one workgroup: the statistic is a tuning hint, its sum need not be fast; its mapped-host store at the launch's end costs nothing measurable: round 4, NV_DEBUG_MODE A/B)
	{
		__shared__ uint32_t s_passed;
		if (tid == 0)
			s_passed = 0;
		__syncthreads();
		const uint32_t wp = wave_sum_u32(passed);
		if (lane == 0 && wp)
			atomicAdd(&s_passed, wp);
		__syncthreads();
		if (tid == 0)
			__hip_atomic_store(a.hostHint + 1, s_passed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
	for (int w = 0; w < SC_WAVES; ++w)
	{
		running += s_part[w];
		total += s_sum[w];
	}
	if (tid == 0 && tile == numTiles - 1)
	{
		a.clusterCount4[0] = total; // what the chain of atomicAdds leaves in clusterCount
		if (a.countsSink) // nv_set_counts_sink: the payload of a sharded caller's all-reduce
		{
			a.countsSink[0] = 0;
			a.countsSink[1] = a.count4[0];
			a.countsSink[2] = total;
		}
	}
	if (a.fusedSubmit && tile == numTiles - 1)
	{
		// NV_OPT_FUSED_SUBMIT: clustersubmit.comp.glsl:25-45 from the workgroup that knows the final count.  The padding
		// slots [count, next multiple of 256) are past every survivor, so no other workgroup writes there.
		const uint32_t count = total < NV_CLUSTER_LIMIT ? total : NV_CLUSTER_LIMIT;
		if (tid == 0)
		{
			const uint32_t gy = (count + 255u) / 256u;
			a.clusterCount4[1] = NV_CLUSTER_TILE;
			a.clusterCount4[2] = gy < 65535u ? gy : 65535u;
			a.clusterCount4[3] = 256u / NV_CLUSTER_TILE;
		}
		const uint32_t boundary = (count + 255u) & ~255u;
		if (count + tid < boundary)
			a.clusterIndices[count + tid] = ~0u;
	}

	// ---- ordered scatter, 1024 commands per step: one scan per step, then one command per iteration for the
	// (coalesced) stores; clustercull.comp.glsl:137-138 drops entries past CLUSTER_LIMIT
	for (uint32_t c0 = 0; c0 < n; c0 += STEP)
	{
		if (c0)
		{
			const uint32_t c = c0 + tid * PER_LANE;
#pragma unroll
			for (uint32_t j = 0; j < PER_LANE; ++j)
				m4[j] = c + j < n ? a.masks[first + c + j] : 0ull;
		}
		uint32_t pc[PER_LANE], mine = 0;
#pragma unroll
		for (uint32_t j = 0; j < PER_LANE; ++j)
		{
			pc[j] = (uint32_t)__builtin_popcountll(m4[j]);
			mine += pc[j];
		}
		uint32_t incl = wave_scan_inclusive_u32(mine);
		__syncthreads(); // s_part free (prefix reduction / previous step's readers are done)
		if (lane == 63)
			s_part[wave] = incl;
		__syncthreads();
		uint32_t excl = running + incl - mine;
#pragma unroll
		for (int w = 0; w < SC_WAVES; ++w)
		{
			uint32_t p = s_part[w];
			excl += w < (int)wave ? p : 0u;
			running += p;
		}

		// The owner loop is what a dense pass's scatter spends its instructions on (one iteration per command with survivors and wave;
		// the CU's scalar unit serves all 16 waves' loops): the CLUSTER_LIMIT test of clustercull.comp.glsl:137 is hoisted to one uniform
		// test per step (the step's last index is `running`), and the owner bit is cleared with one s_bitset0_b64.
		const bool mayOverflow = running > NV_CLUSTER_LIMIT; // (uniform; `running` already includes this step's survivors)
		auto emit = [&](auto checked)
		{
#pragma unroll
			for (uint32_t j = 0; j < PER_LANE; ++j)
			{
				uint64_t owners = dbgNoScatter ? 0ull : __ballot(pc[j] != 0);
				while (owners)
				{
					const int src = __builtin_ctzll(owners);
					asm("s_bitset0_b64 %0, %1" : "+s"(owners) : "s"(src));
					const uint32_t mlo = __builtin_amdgcn_readlane((uint32_t)m4[j], src);
					const uint32_t mhi = __builtin_amdgcn_readlane((uint32_t)(m4[j] >> 32), src);
					const uint32_t off = __builtin_amdgcn_readlane(excl, src);
					const uint64_t ms = ((uint64_t)mhi << 32) | mlo;
					if (ms >> lane & 1ull)
					{
						uint32_t rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
						uint32_t index = off + rank;
						if (!decltype(checked)::value || index < NV_CLUSTER_LIMIT)
							a.clusterIndices[index] = (first + c0 + (wave * 64 + src) * PER_LANE + j) | (lane << 24);
					}
				}
				excl += pc[j];
			}
		};
		if (mayOverflow)
			emit(std::true_type{});
		else
			emit(std::false_type{});
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Late pass with HiZ, stage 2 (between the cull kernel and the scatter kernel): clustercull.comp.glsl:110-131 with one
// LANE per frustum / cone survivor.
//
// Inside the cull kernel the occlusion probe costs a wave three dependent memory latencies per COMMAND (bounds -> texel
// addresses -> texels) with only the command's survivors active, and the launch ended with the few waves that drew the
// visible part of the scene (r1 / r2 measurements: 40-43 us against 25 us for the same pass without the probe, whatever
// the ring depths, chunk sizes or priorities).  So the late pass defers it: the cull kernel runs in its early form
// (frustum + cone only, no visibility words, no tile counts: ClusterArgs::deferHiz) and leaves the survivors' ballots;
// this kernel compacts the survivors of CH_CMDS commands through LDS, probes them lane-parallel with the reference's own
// sphere arithmetic (lane_sphere + hiz_test: the same functions the one-stage form called), clears the ballot bits of
// the occluded ones, and then does per command — one command per lane — what the cull kernel's epilogue did: the
// visibility-bit update, the `skip` of clusters the early pass already drew, the final ballot and the tile count.
constexpr int CH_THREADS = 256; // lanes probing
constexpr int CH_CMDS = 128;    // commands per block (threads 0 .. CH_CMDS - 1 own one each)
// (round 4, frame scale, experiments flavour, lanes x commands x blocks per sub-list: 256 x 128 x 4 as built 59.7 us; 256 x 64 x 4 / 8: 61.4-64.4; 512 x 128 x 2 / 4:
//  67.6-69.2; 512 x 256 x 2 / 3: 60.7-62.6; 128 x 64 x 8 / 12: 63.5-66.6; 128 x 128 x 6 / 8: 71.5-72.5; blocks per sub-list 4 / 5 / 6 / 7 / 8 / 12 at 256 x 128: 59.6 / 61.8 / 61.2 / 62.7 / 60.3 / 60.1)
#ifndef NV_CH_U
#define NV_CH_U 4
#endif
// survivors per lane in flight: a round probes CH_THREADS * CH_U of the block's survivors.  Round 4 (tools/build_cc_variants.sh, the frame at
// BASELINE scale): 8 (121 VGPRs, four waves per SIMD) 62.3-64.0 us, 6 / 5: 60.2-61.2, **4 (80 VGPRs, six waves) 58.4-59.4**, 3: 59.4; config 4's
// short lists do not care (14.6-14.7 us at 8 and at 4)
constexpr int CH_U = NV_CH_U;

// LDS-only barrier: __syncthreads() also waits for the global loads in flight (vmcnt counts them on gfx9), and the point
// of this kernel is to keep them in flight across the LDS hand-overs
#define NV_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// The kernel is a chain of memory latencies, not of bytes (10 M meshlets, config 4: 198 k survivors in 3 % of the
// commands, up to ~1600 in one block), so it is laid out to be three latencies long whatever a block holds:
//   commands + ballots  ->  { MeshDraws of the commands with survivors, their visibility words, bounds of CH_U survivors
//   per lane }  ->  4 x CH_U texels per lane  ->  stores.
// One round: U survivors per lane, straight-line — every load is unconditional (slots past the block's last survivor re-read
// it, inactive probes fetch clamped texels) so that hipcc can count its waits and all U bounds, then all 4 U texels, are in
// flight together.  The first round also hands the MeshDraws over through LDS, behind the bounds loads.
// Experiments build, NV_DEBUG_MODE bit 28: where a wave of the stage spends its cycles (tools/experiments/hiz_timeline.py).  Phase sums per wave:
// 0 list / draws / compaction of a chunk, 1 a round's bounds (issue to arrival), 2 its probes' arithmetic (the texel loads issued behind each),
// 3 the texels (the wait that is left), 4 comparisons and LDS atomics, 5 a chunk's per-command epilogue; the stamped build waits for ALL loads at
// the phase boundaries, which the product does not.
struct HizTimes
{
	bool on;
	uint64_t prev;
	uint64_t acc[6];
	uint32_t rounds;
};
#ifdef NV_EXPERIMENTS
#define NV_HIZ_T(T, i, drain)                                                  \
	do                                                                         \
	{                                                                          \
		if ((T).on)                                                            \
		{                                                                      \
			if (drain)                                                         \
				asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    \
			const uint64_t now = __builtin_readcyclecounter();                 \
			(T).acc[i] += now - (T).prev;                                      \
			(T).prev = now;                                                    \
		}                                                                      \
	} while (0)
#else
#define NV_HIZ_T(T, i, drain) do { } while (0)
#endif

template <int U, bool SOA, bool FIRST>
NV_DEV void hiz_round(const ClusterArgs& a, uint32_t base, uint32_t total, uint32_t tid, bool ownsDraw, const float4& d0, const float4& d1,
                      const uint16_t* s_list, const uint32_t* s_taskOffset, float4 (*s_draw)[2], const MipRecord* s_mip, uint32_t* s_visLo, uint32_t* s_visHi,
                      HizTimes& T)
{
	const float* __restrict__ texels = a.pyr.d_base;
	uint32_t e[U];
	uint2 b[U];
#pragma unroll
	for (int k = 0; k < U; ++k)
	{
		const uint32_t s = base + k * CH_THREADS + tid;
		e[k] = s_list[s < total ? s : total - 1u];
		const uint32_t mi = s_taskOffset[e[k] >> 6] + (e[k] & 63u);
		if (SOA)
			b[k] = a.soaBounds[mi];
		else
			b[k] = *reinterpret_cast<const uint2*>(a.meshlets + mi);
	}
	if (FIRST)
	{
		if (ownsDraw)
		{
			s_draw[tid][0] = d0;
			s_draw[tid][1] = d1;
		}
		NV_LDS_BARRIER();
	}
	NV_HIZ_T(T, 1, true);
	float depth[U];
	uint32_t use[U];
	float t00[U], t10[U], t01[U], t11[U];
#pragma unroll
	for (int k = 0; k < U; ++k)
	{
		const float4 q0 = s_draw[e[k] >> 6][0], q1 = s_draw[e[k] >> 6][1];
		DrawUniform u;
		u.pos = { q0.x, q0.y, q0.z };
		u.scale = q0.w;
		u.q = { q1.x, q1.y, q1.z };
		u.qw = q1.w;
		LaneData l;
		l.b0 = b[k].x;
		l.b1 = b[k].y;
		l.cone = 0;
		l.mvbWord = 0;
		f3 c;
		float r;
		lane_sphere(a.cd, u, l, c, r);
		const HizProbe p = hiz_prepare<true>(a.cd, a.pyr, c, r, s_mip);
		use[k] = p.use;
		depth[k] = p.depthSphere;
		const uint32_t zero = NV_DBG(a, 16777216u) ? 0u : ~0u; // bit 24 (experiments): every probe reads texel 0
		t00[k] = texels[p.o00 & zero]; // in range also for an inactive probe (hiz_prepare)
		t10[k] = texels[p.o10 & zero];
		t01[k] = texels[p.o01 & zero];
		t11[k] = texels[p.o11 & zero];
	}
	NV_HIZ_T(T, 2, false);
	NV_HIZ_T(T, 3, true);
#pragma unroll
	for (int k = 0; k < U; ++k)
	{
		const uint32_t s = base + k * CH_THREADS + tid;
		const HizProbe p = { 0, 0, 0, 0, use[k], depth[k] };
		if (!hiz_finish(p, t00[k], t10[k], t01[k], t11[k]) && s < total)
		{
			const uint32_t owner = e[k] >> 6, bit = e[k] & 63u;
			atomicAnd(bit < 32u ? &s_visLo[owner] : &s_visHi[owner], ~(1u << (bit & 31u)));
		}
	}
	NV_HIZ_T(T, 4, true);
	T.rounds += 1u;
}

template <bool SOA, bool BITS>
__global__ __launch_bounds__(CH_THREADS) void cluster_hiz_kernel(ClusterArgs a)
{
	__shared__ __attribute__((aligned(16))) MipRecord s_mip[NV_MAX_MIPS];
	__shared__ uint32_t s_visLo[CH_CMDS], s_visHi[CH_CMDS], s_taskOffset[CH_CMDS];
	__shared__ float4 s_draw[CH_CMDS][2];
	__shared__ uint16_t s_list[CH_CMDS * 64]; // (command within the block << 6) | lane, in command-major order
	__shared__ uint32_t s_part[CH_THREADS / 64];

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	HizTimes T = { NV_DBG(a, 268435456u) && a.probeOut != nullptr, 0, { 0, 0, 0, 0, 0, 0 }, 0 }; // bit 28 (experiments): phase stamps
#ifdef NV_EXPERIMENTS
	const uint64_t tEntry = T.on ? __builtin_readcyclecounter() : 0;
	T.prev = tEntry;
#endif
	{
		// per-level records for the probes (cullmath.h MipRecord), built from the scalar kernel arguments with constant indices
		uint32_t off = 0;
#pragma unroll
		for (uint32_t i = 0; i < NV_MAX_MIPS; ++i)
			off = tid == i ? a.pyr.mipOffset[i] : off;
		if (tid < NV_MAX_MIPS)
			s_mip[tid] = make_mip_record(a.pyr, tid, off);
	}
	// The cull kernel listed the commands that have survivors in CC_LISTS sub-lists; block b works on sub-list b % CC_LISTS
	// together with the gridDim.x / CC_LISTS - 1 other blocks of that sub-list, in chunks: listMinPer (8) listed commands per
	// block while that covers the sub-list (<= 512 survivors: two per lane), more for a longer one.  (Measured, config 4:
	// 4 blocks x 8 commands 12.3 us, 5 x 4 14.2, 2 x 16 12.9, 1 x 32 13.6 — the stage is bound by its fixed chain of
	// latencies and the probes' arithmetic, not by the balance.)  The first chunk's entries are fetched
	// together with the sub-list's length (entries past its end are ignored).  If a sub-list overflowed — a pass with more
	// than a few hundred thousand such commands — every block scans its share of ALL commands instead: then nearly every
	// command has survivors and contiguous ranges are balanced by themselves.
	const uint32_t sub = blockIdx.x % CC_LISTS, rank0 = blockIdx.x / CC_LISTS, sharers = gridDim.x / CC_LISTS;
	const uint4* __restrict__ list = a.candList + (size_t)sub * a.listStride * 2u;
	uint4 spec0 = make_uint4(0, 0, 0, 0), spec1 = spec0;
	if (tid < a.listMinPer) // (in range: the buffer is padded)
	{
		spec0 = list[(rank0 * a.listMinPer + tid) * 2u];
		spec1 = list[(rank0 * a.listMinPer + tid) * 2u + 1u];
	}
	const uint32_t bank = load_uniform_u32(&a.tileCounts->k2parity) & 1u; // the cull kernel of this pass wrote it
	const uint32_t count0 = load_uniform_u32(&a.tileCounts->listCount[0][sub * CC_COUNT_STRIDE]);
	const uint32_t count1 = load_uniform_u32(&a.tileCounts->listCount[1][sub * CC_COUNT_STRIDE]);
	const uint32_t over0 = load_uniform_u32(&a.tileCounts->listOverflow[0]), over1 = load_uniform_u32(&a.tileCounts->listOverflow[1]);
	const uint32_t numCmds = indirect_command_count(a);
	const uint32_t T2 = scatter_tile_commands(numCmds, a.scatterTiles, a.tilesMagic);
	const bool scan = (bank ? over1 : over0) != 0;
	const uint32_t items = scan ? numCmds : (bank ? count1 : count0);
	const uint32_t share = scan ? gridDim.x : sharers;
	uint32_t per = (items + share - 1u) / share;
	per = per < a.listMinPer ? a.listMinPer : (per > (uint32_t)CH_CMDS ? (uint32_t)CH_CMDS : per);
	const uint32_t firstChunk = scan ? blockIdx.x : rank0;

	for (uint32_t chunk = firstChunk; chunk * per < items; chunk += share)
	{
		const uint32_t li = chunk * per + tid;
		const bool live = tid < per && li < items;
		if (NV_DBG(a, 4194304u)) // bit 22 (experiments): the list only
			continue;
		uint32_t idx = 0;
		uint64_t cand = 0;
		uint32_t drawId = 0, taskOffset = 0, taskCount = 0, lateDrawVisibility = 0, mvo = 0;
		if (scan)
		{
			idx = li;
			if (live)
			{
				cand = a.masks[idx];
				const uint32_t* p = reinterpret_cast<const uint32_t*>(a.commands + idx);
				drawId = p[0];
				taskOffset = p[1];
				taskCount = p[2] < 64u ? p[2] : 64u;
				lateDrawVisibility = p[3];
				mvo = p[4];
			}
		}
		else if (live)
		{
			uint4 e0 = spec0, e1 = spec1;
			if (!(per == a.listMinPer && chunk == rank0))
			{
				e0 = list[li * 2u];
				e1 = list[li * 2u + 1u];
			}
			if (e0.x < numCmds)
			{
				idx = e0.x;
				drawId = e0.y;
				taskOffset = e0.z;
				taskCount = e0.w;
				lateDrawVisibility = e1.x;
				mvo = e1.y;
				cand = ((uint64_t)e1.w << 32) | e1.z;
			}
		}
		// second latency, part 1: the command's MeshDraw and its <= 3 visibility words.  Unconditional, clamped loads (dummy
		// commands and the lanes without a command read element 0): a branch around a load makes hipcc fall back to
		// s_waitcnt vmcnt(0) at the join, and the loads behind it would wait for these
		const float4* dp = reinterpret_cast<const float4*>(a.draws + (taskCount ? drawId : 0u));
		const float4 d0 = dp[0], d1 = dp[1];
		const uint32_t sh = mvo & 31u;
		uint32_t oldw[3];
		{
			const uint32_t w0 = taskCount ? mvo >> 5 : 0u, wLast = taskCount ? (mvo + taskCount - 1u) >> 5 : 0u;
#pragma unroll
			for (uint32_t j = 0; j < 3; ++j) // word j holds the bits of lanes [32 j - sh, 32 j - sh + 32)
				oldw[j] = a.mvb[w0 + j < wLast ? w0 + j : wLast];
		}

		const uint32_t pc = (uint32_t)__builtin_popcountll(cand);
		uint32_t incl = wave_scan_inclusive_u32(pc);
		if (lane == 63)
			s_part[wave] = incl;
		if (tid < CH_CMDS)
		{
			s_visLo[tid] = (uint32_t)cand;
			s_visHi[tid] = (uint32_t)(cand >> 32);
			s_taskOffset[tid] = taskOffset;
		}
		NV_LDS_BARRIER();
		uint32_t excl = incl - pc, total = 0;
#pragma unroll
		for (int w = 0; w < CH_THREADS / 64; ++w)
		{
			const uint32_t part = s_part[w];
			excl += w < (int)wave ? part : 0u;
			total += part;
		}
		for (uint64_t rest = cand; rest; rest &= rest - 1)
			s_list[excl++] = (uint16_t)((tid << 6) | (uint32_t)__builtin_ctzll(rest));
		NV_LDS_BARRIER();

		// ---- one survivor per lane and slot: clustercull.comp.glsl:72-76 (sphere) and :110-123 (probe).  The first round is
		// not a loop body: at a loop header hipcc merges the wait state of the back edge into it and waits for everything
		// in flight (the MeshDraw and visibility words) before the first bounds load.
#define NV_HIZ_ARGS(base) a, base, total, tid, cand != 0, d0, d1, s_list, s_taskOffset, s_draw, s_mip, s_visLo, s_visHi, T
#define NV_HIZ_ROUND(FIRST, base)                                                                                  \
	do                                                                                                             \
	{                                                                                                              \
		const uint32_t rem = total - (base);                                                                       \
		if (CH_U > 4 && rem > 4u * CH_THREADS)                                                                     \
			hiz_round<CH_U, SOA, FIRST>(NV_HIZ_ARGS(base));                                                        \
		else if (CH_U >= 4 && rem > 2u * CH_THREADS)                                                               \
			hiz_round<4, SOA, FIRST>(NV_HIZ_ARGS(base));                                                           \
		else if (CH_U == 3 && rem > 2u * CH_THREADS)                                                               \
			hiz_round<3, SOA, FIRST>(NV_HIZ_ARGS(base));                                                           \
		else if (CH_U >= 2 && rem > CH_THREADS)                                                                    \
			hiz_round<2, SOA, FIRST>(NV_HIZ_ARGS(base));                                                           \
		else                                                                                                       \
			hiz_round<1, SOA, FIRST>(NV_HIZ_ARGS(base));                                                           \
	} while (0)
		if (total == 0) // (uniform) nothing survived frustum and cone in this block: the cull kernel has cleared the bits
			continue;
		NV_HIZ_T(T, 0, false);
		if (!NV_DBG(a, 8388608u)) // bit 23 (experiments): no probes
		{
			NV_HIZ_ROUND(true, 0u);
			for (uint32_t base = CH_THREADS * CH_U; base < total; base += CH_THREADS * CH_U)
				NV_HIZ_ROUND(false, base);
			if (NV_DBG(a, 33554432u)) // bit 25 (experiments): every round a second time (same result) — what the same code costs warm
				for (uint32_t base = 0; base < total; base += CH_THREADS * CH_U)
					NV_HIZ_ROUND(false, base);
		}
#undef NV_HIZ_ROUND
#undef NV_HIZ_ARGS
		NV_LDS_BARRIER();

		// ---- one command per lane: `visible` is final.  clustercull.comp.glsl:97-99 (skip what the early pass drew),
		// :125-131 (visibility bits: as update_segment_visibility, from the words loaded above — only this command's own
		// bits of them are used, and nobody else writes those), the final ballot and the tile count.
		uint64_t m = 0;
		if (cand) // (commands without survivors: done by the cull kernel)
		{
			const uint64_t vis = ((uint64_t)s_visHi[tid] << 32) | s_visLo[tid];
			const uint64_t valid = taskCount >= 64u ? ~0ull : (1ull << taskCount) - 1ull;
			const uint64_t setAll = vis & valid, clrAll = valid & ~vis;
			uint64_t old = 0;
			uint32_t* words = a.mvb + (mvo >> 5);
#pragma unroll
			for (int j = 0; j < 3; ++j)
			{
				const int lo = 32 * j - (int)sh;
				if (lo >= (int)taskCount)
					break;
				uint32_t setw, clrw;
				if (lo >= 0)
				{
					old |= (uint64_t)oldw[j] << lo;
					setw = (uint32_t)(setAll >> lo);
					clrw = (uint32_t)(clrAll >> lo);
				}
				else
				{
					old |= (uint64_t)(oldw[j] >> (-lo));
					setw = (uint32_t)(setAll << (-lo));
					clrw = (uint32_t)(clrAll << (-lo));
				}
				setw &= ~oldw[j]; // only bits that change
				clrw &= oldw[j];
				if ((setw | clrw) == 0)
					continue;
				if (lo >= 0 && (uint32_t)lo + 32u <= taskCount) // the whole word belongs to this command: no other writer
					words[j] = (oldw[j] & ~clrw) | setw;
				else
				{
					if (clrw)
						atomicAnd(words + j, ~clrw);
					if (setw)
						atomicOr(words + j, setw);
				}
			}
			m = vis;
			if (BITS && lateDrawVisibility == 1)
				m &= ~old;
			a.masks[idx] = m;
		}
		if (m) // the listed commands come in no particular order: one add per command
			atomicAdd(&a.tileCounts->counts[bank][(idx / T2) * CC_COUNT_STRIDE], (uint32_t)__builtin_popcountll(m));
		NV_HIZ_T(T, 5, false);
	}
#ifdef NV_EXPERIMENTS
	if (T.on && lane == 0)
	{
		unsigned long long* out = reinterpret_cast<unsigned long long*>(a.probeOut) + (size_t)(blockIdx.x * (CH_THREADS / 64) + wave) * 8u;
		for (int i = 0; i < 6; ++i)
			out[i] = T.acc[i];
		out[6] = __builtin_readcyclecounter() - tEntry;
		out[7] = T.rounds;
	}
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Early pass with visibility bits, dense form: one LANE per cluster whose bit is set.
//
// clustercull.comp.glsl:86-95 lets only last frame's visible clusters through, so a wave that walks one command at a time
// runs the sphere / cone arithmetic with most of its lanes off (a frame of the 1 M-draw scene: 29 % of the slots are set;
// the direct form of cluster_mask_kernel spends 85 VALU + 60 SALU instructions per command whatever the bits say, 38-40 us).
// This kernel is the occlusion stage's layout applied to the early pass: a block owns a contiguous range of commands (one per
// lane), expands their set bits into an LDS list — one entry per cluster that can be visible at all — and tests the entries
// with one LANE each: the certified two-sided test of pass B (certified_visible, same arithmetic and margins, decided per
// lane instead of per wave), the reference's own arithmetic for the lanes it leaves undecided.  The per-draw coefficients
// (make_filter) are computed once per command, lane-parallel, and read from LDS per entry.  Output as cluster_mask_kernel's:
// one ballot per command, survivors per scatter tile, the statistic for the host's choice of the next launch's form.
//
// Measured on the frame's early pass (175 k commands, 2.98 M set bits): 27 us against 38-40 us, bound by instruction issue
// (12 M wave-instructions, 4.5 cycles each per SIMD outside the launch's 4 us) — which is why the variants that only moved
// latencies changed nothing: the entries' loads kept in flight by a counted ring instead of rounds (27.7 us), four
// independent waves per block with a list each and no barrier (33-34 us: the per-command phases then run on 22 lanes
// of every wave instead of on the full first wave of a block).  Without visibility bits every valid cluster is an entry and
// the form loses to one command per wave (10 M clusters: 52-62 us against 40, whatever the variant), so it is used for the
// early pass with bits only.
constexpr int CB_THREADS = 256;
constexpr int CB_CMDS = 128; // commands per block and iteration, at most (lanes 0 .. per - 1 own one each)
constexpr int CB_U = 4;      // list entries per lane in flight
// (round 4, frame scale, cull 25.6-25.9 us as built: 96 commands with 5 blocks per CU 25.9-26.2, 96 / 64 commands with 6 blocks — 80 VGPRs, spills — 39-40,
//  2 entries per lane 26.1-27.7, 8 with 3 blocks 27.6-28.5, 3 blocks per CU 27.0-28.9)

// One round = CB_U list entries per lane, slot-major (slot k of lane t = entry base + k CB_THREADS + t), straight-line: all
// loads unconditional (slots past the last entry re-read it) and issued together; a slot no lane of the wave holds an entry
// for is skipped as a whole.  ONE instantiation per layout: with a copy per round size hipcc gave the copies' registers
// to each other and waited for everything in flight (the prefetches) before the largest copy's first load.
template <bool SOA, bool STAT>
NV_DEV void bits_round(const ClusterArgs& a, uint32_t base, uint32_t total, uint32_t tid, bool first, bool ownsCommand, const float4& d0, const float4& d1,
                       const uint16_t* s_list, const uint32_t* s_taskOffset, float4 (*s_draw)[2], float4 (*s_cert)[5], uint32_t* s_visLo, uint32_t* s_visHi,
                       uint32_t* s_passed)
{
	constexpr int U = CB_U;
	uint32_t e[U];
	uint2 b[U];
	uint32_t cone[U];
#pragma unroll
	for (int k = 0; k < U; ++k)
	{
		const uint32_t s = base + k * CB_THREADS + tid;
		e[k] = s_list[s < total ? s : total - 1u];
		const uint32_t mi = s_taskOffset[e[k] >> 6] + (e[k] & 63u);
		if (SOA)
		{
			b[k] = a.soaBounds[mi];
			cone[k] = a.soaCones[mi];
		}
		else
		{
			const uint32_t* p = reinterpret_cast<const uint32_t*>(a.meshlets + mi);
			b[k] = make_uint2(p[0], p[1]);
			cone[k] = p[2];
		}
	}
	if (first) // (uniform)
	{
		// behind the first entries' loads: the per-command coefficients of the certified test (one command per lane).  The empty
		// statement keeps them there: left alone hipcc hoists the arithmetic (it only depends on the prefetched draw) in front of
		// the loads, together with a wait for the youngest prefetches — one full latency before the entries are even asked for.
		float4 q0 = d0, q1 = d1;
		asm volatile("; coefficients behind the entry loads" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w)::"memory");
		if (ownsCommand)
		{
			DrawUniform u;
			u.pos = { q0.x, q0.y, q0.z };
			u.scale = q0.w;
			u.q = { q1.x, q1.y, q1.z };
			u.qw = q1.w;
			// (SOA: the margin is the draw's tK over the registered pool's bounds, as in packed_walk — four multiply-adds and a 16-byte LDS read less per entry;
			// records read in place have no pool bounds and keep the per-meshlet margin)
			float vmax3 = 0.0f, rmax = 0.0f;
			if (SOA)
			{
				k_f32p pb = (k_f32p)(uintptr_t)a.poolBounds;
				vmax3 = pb[0];
				rmax = pb[1];
			}
			const FilterDraw f = make_filter(a.cd, u, a.filterK, a.viewRowNorm, a.viewTransNorm, a.viewSum, vmax3, rmax);
			s_draw[tid][0] = q0;
			s_draw[tid][1] = q1;
			s_cert[tid][0] = make_float4(f.m[0], f.m[1], f.m[2], f.b[0]);
			s_cert[tid][1] = make_float4(f.m[3], f.m[4], f.m[5], f.b[1]);
			s_cert[tid][2] = make_float4(f.m[6], f.m[7], f.m[8], f.b[2]);
			if (SOA)
				s_cert[tid][3] = make_float4(f.tK, f.scale, f.tK * f.coneK, f.is127);
			else
			{
				s_cert[tid][3] = make_float4(f.aK, f.bK, f.aR, f.scale);
				s_cert[tid][4] = make_float4(f.coneK, f.is127, 0.0f, 0.0f);
			}
		}
		NV_LDS_BARRIER();
	}
	const NvCullData& cd = a.cd;
	const bool useCert = a.filterK > 0.0f;
#pragma unroll
	for (int k = 0; k < U; ++k)
	{
		const uint32_t s = base + k * CB_THREADS + tid;
		if (__ballot(s < total) == 0) // (uniform) nobody in this wave holds an entry in this slot
			continue;
		const uint32_t owner = e[k] >> 6, bit = e[k] & 63u;
		const float4 r0 = s_cert[owner][0], r1 = s_cert[owner][1], r2 = s_cert[owner][2], r3 = s_cert[owner][3];
		const uint32_t b0 = b[k].x, b1 = b[k].y;
		// certified_visible, one lane = one cluster
		const float vx = half_bits_to_float(b0 & 0xffffu), vy = half_bits_to_float(b0 >> 16), vz = half_bits_to_float(b1 & 0xffffu);
		const float rad = half_bits_to_float(b1 >> 16);
		const float cx = __builtin_fmaf(r0.x, vx, __builtin_fmaf(r0.y, vy, __builtin_fmaf(r0.z, vz, r0.w)));
		const float cy = __builtin_fmaf(r1.x, vx, __builtin_fmaf(r1.y, vy, __builtin_fmaf(r1.z, vz, r1.w)));
		const float cz = __builtin_fmaf(r2.x, vx, __builtin_fmaf(r2.y, vy, __builtin_fmaf(r2.z, vz, r2.w)));
		float T, scale, Tc, is127;
		if (SOA)
			T = r3.x, scale = r3.y, Tc = r3.z, is127 = r3.w;
		else
		{
			const float4 r4 = s_cert[owner][4];
			const float aK = r3.x, bK = r3.y, aR = r3.z;
			scale = r3.w;
			is127 = r4.y;
			T = __builtin_fmaf(aK, __builtin_fabsf(vx), bK);
			T = __builtin_fmaf(aK, __builtin_fabsf(vy), T);
			T = __builtin_fmaf(aK, __builtin_fabsf(vz), T);
			T = __builtin_fmaf(aR, __builtin_fabsf(rad), T);
			Tc = T * r4.x;
		}
		const float thrHi = __builtin_fmaf(scale, rad, T), thrLo = __builtin_fmaf(scale, rad, -T);
		const float g1 = __builtin_fmaf(cz, cd.frustum[1], -(__builtin_fabsf(cx) * cd.frustum[0]));
		const float g2 = __builtin_fmaf(cz, cd.frustum[3], -(__builtin_fabsf(cy) * cd.frustum[2]));
		const float gn = cz - cd.znear;
		const float gf = cd.zfar - cz;
		const float g = __builtin_fminf(__builtin_fminf(g1, g2), __builtin_fminf(gn, gf));
		const bool out = g < -thrHi, in = g > -thrLo;
		bool decided = useCert && (out || in);
		bool visible = in;
		if (cd.clusterBackfaceEnabled != 0)
		{
			const float kx = s8_to_float(cone[k], 0), ky = s8_to_float(cone[k], 1), kz = s8_to_float(cone[k], 2), kc = s8_to_float(cone[k], 3);
			const float wx = __builtin_fmaf(r0.x, kx, __builtin_fmaf(r0.y, ky, r0.z * kz));
			const float wy = __builtin_fmaf(r1.x, kx, __builtin_fmaf(r1.y, ky, r1.z * kz));
			const float wz = __builtin_fmaf(r2.x, kx, __builtin_fmaf(r2.y, ky, r2.z * kz));
			const float lhs = __builtin_fmaf(cx, wx, __builtin_fmaf(cy, wy, cz * wz)) * is127;
			const float len = __builtin_amdgcn_sqrtf(__builtin_fmaf(cx, cx, __builtin_fmaf(cy, cy, cz * cz)));
			const float rhs = __builtin_fmaf(kc * 0.00787401574803149606f, len, scale * rad);
			const float D = lhs - rhs;
			const bool cull = D > Tc, keep = D < -Tc;
			decided = decided && (out || cull || keep); // (a cluster outside the frustum is decided whatever its cone says)
			visible = visible && keep;
		}
		if (!decided && !NV_DBG(a, 536870912u)) // the reference's arithmetic (clustercull.comp.glsl:72-80,102-108), as cull_command evaluates it.  (bit 29, experiments: skipped)
		{
			const float4 q0 = s_draw[owner][0], q1 = s_draw[owner][1];
			DrawUniform u;
			u.pos = { q0.x, q0.y, q0.z };
			u.scale = q0.w;
			u.q = { q1.x, q1.y, q1.z };
			u.qw = q1.w;
			LaneData l;
			l.b0 = b0;
			l.b1 = b1;
			l.cone = cone[k];
			l.mvbWord = 0;
			f3 c;
			float r;
			lane_sphere(cd, u, l, c, r);
			visible = frustum_test(cd, c, r);
			if (cd.clusterBackfaceEnabled != 0 && visible)
			{
				f3 axis;
				float cutoff;
				lane_cone(cd, u, l, axis, cutoff);
				visible = !cone_cull(c, r, axis, cutoff);
			}
		}
		if (STAT)
		{
			// the statistic of a pass WITHOUT visibility bits: what pass A's filter would have let through (the same comparison) —
			// with bits a set bit stands in for it (kernel), here every valid cluster is an entry and the host must see a pass whose
			// commands lie outside the frustum turn sparse.  One store per streak of such lanes within a command's entries, not one per lane.
			const bool passes = s < total && !(useCert && out);
			const uint64_t passM = __ballot(passes);
			const uint32_t lane = tid & 63u;
			const uint32_t prevOwner = wave_shift_up1_u32(owner);
			if (passes && (lane == 0u || prevOwner != owner || !(passM >> (lane - 1u) & 1ull)))
				s_passed[owner] = 1u;
		}
		if (s < total && !visible)
			atomicAnd(bit < 32u ? &s_visLo[owner] : &s_visHi[owner], ~(1u << (bit & 31u)));
	}
}

// The block's share of the pass is cut into iterations of `per` <= CB_CMDS commands, and the loop is software-pipelined over
// them: while the entries of iteration i are fetched and tested, the MeshDraws and visibility words of iteration i + 1 and
// the commands of iteration i + 2 are in flight, so an iteration starts with its list instead of two memory latencies.
// Plain loads: the prefetches are older than the entries' loads, so hipcc's in-order wait for the latter covers them, and
// at the loop's back edge they have had the whole iteration to arrive.
struct BitsCommand
{
	uint32_t drawId, taskOffset, taskCount, mvo;
};

// (unconditional: a lane without a command reads command 0 and its taskCount is zeroed where the record is USED — a load under
// a branch is copied out of the branch by hipcc, behind a wait that also covers the prefetches issued just before it)
NV_DEV BitsCommand bits_load_command(const ClusterArgs& a, uint32_t idx, bool live)
{
	const uint32_t* p = reinterpret_cast<const uint32_t*>(a.commands + (live ? idx : 0u));
	BitsCommand c;
	c.drawId = p[0];
	c.taskOffset = p[1];
	c.taskCount = p[2] < 64u ? p[2] : 64u;
	c.mvo = p[4];
	return c;
}

// BITS: the early pass with visibility bits (candidates = set bits).  Without: every valid cluster is a candidate — the form for a dense
// EARLY pass over a meshlet pool that stays in the caches (instanced scenes: the host decides, context.hip; round 4: config 3B's cluster
// pass 27.0 against 28.4 us).  The two LATE applications of the form were built and measured in round 4 and are not here
// (tools/experiments/lane_late_forms_r4.diff): as the late pass's first stage (the survivors' commands listed for cluster_hiz_kernel)
// it takes 48.7 us where one command per wave takes 40.2 us at frame scale (10.2 M candidates: 86 bytes of LDS reads and two gathered
// loads per entry, four waves per SIMD), and with the occlusion probes fused in (second compaction of the survivors, hiz_round in the
// same block, no list and no second launch) 109.3 us against 40.2 + 63.9 — the block's phases serialise behind its barriers and the
// other blocks of the CU do not fill the gaps.
template <bool SOA, bool BITS>
__global__ __launch_bounds__(CB_THREADS, 4) void cluster_bits_kernel(ClusterArgs a)
{
	__shared__ uint32_t s_visLo[CB_CMDS], s_visHi[CB_CMDS], s_taskOffset[CB_CMDS], s_excl[CB_CMDS], s_passed[BITS ? 1 : CB_CMDS];
	__shared__ float4 s_draw[CB_CMDS][2];
	__shared__ float4 s_cert[CB_CMDS][5];
	__shared__ uint16_t s_list[CB_CMDS * 64]; // (command within the iteration << 6) | lane, in command-major order
	__shared__ uint32_t s_part[CB_THREADS / 64];

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t numCmds = indirect_command_count(a);
	if (a.hostHint && blockIdx.x == 0 && tid == 0)
		__hip_atomic_store(a.hostHint + (a.payloadCounts ? 4 : 0), numCmds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (word 4: see cluster_mask_kernel)
	const uint32_t T2 = scatter_tile_commands(numCmds, a.scatterTiles, a.tilesMagic);
	const uint32_t bank = load_uniform_u32(&a.tileCounts->parity) & 1u;
	if (blockIdx.x == 0 && tid == 0)
	{
		a.tileCounts->k2parity = bank;
		a.tileCounts->base = a.fusedReset ? 0u : a.clusterCount4[0];
	}
	if (numCmds == 0) // an empty pass: the pipeline prologue below reads commands[0], draws[0] and mvb[0] unconditionally
		return;
	// every block the same number of iterations (the grid is resident at once: the last iteration is the launch's tail),
	// iteration i of block b = commands [(i G + b) per, (i G + b + 1) per): neighbouring blocks read neighbouring commands
	const uint32_t G = gridDim.x;
	const uint32_t iters = (numCmds + G * (uint32_t)CB_CMDS - 1u) / (G * (uint32_t)CB_CMDS);
	uint32_t per = iters ? (numCmds + G * iters - 1u) / (G * iters) : 0u;
	per = per < 16u ? 16u : per; // (<= CB_CMDS by construction)
	uint32_t passedAcc = 0;

	// pipeline prologue: commands of iterations 0 and 1, MeshDraw and words of iteration 0
	uint32_t chunk = blockIdx.x;
	BitsCommand cur = bits_load_command(a, chunk * per + tid, tid < per && chunk * per + tid < numCmds);
	BitsCommand nxt = bits_load_command(a, (chunk + G) * per + tid, tid < per && (chunk + G) * per + tid < numCmds);
	float4 d0, d1;
	uint32_t oldw[3];
	auto load_dependents = [&](const BitsCommand& c, bool liveC, float4& o0, float4& o1, uint32_t* w)
	{
		// unconditional, clamped loads (see cluster_hiz_kernel): the command's MeshDraw and its <= 3 visibility words
		const uint32_t tc = liveC ? c.taskCount : 0u;
		const float4* dp = reinterpret_cast<const float4*>(a.draws + (tc ? c.drawId : 0u));
		o0 = dp[0];
		o1 = dp[1];
		const uint32_t w0 = tc ? c.mvo >> 5 : 0u, wLast = tc ? (c.mvo + tc - 1u) >> 5 : 0u;
#pragma unroll
		for (uint32_t j = 0; j < 3; ++j) // word j holds the bits of lanes [32 j - sh, 32 j - sh + 32)
			w[j] = BITS ? a.mvb[w0 + j < wLast ? w0 + j : wLast] : 0u;
	};
	load_dependents(cur, tid < per && chunk * per + tid < numCmds, d0, d1, oldw);
	// (complete before the loop: otherwise the loop body must assume they may still be in flight, and the wait hipcc then puts
	// in front of their use covers the youngest prefetches of every later iteration)
	asm volatile("; prologue loads landed" : "+v"(d0.x), "+v"(d0.y), "+v"(d0.z), "+v"(d0.w), "+v"(d1.x), "+v"(d1.y), "+v"(d1.z), "+v"(d1.w), "+v"(oldw[0]), "+v"(oldw[1]), "+v"(oldw[2]),
	             "+v"(nxt.drawId), "+v"(nxt.taskOffset), "+v"(nxt.taskCount), "+v"(nxt.mvo));

	for (; chunk * per < numCmds; chunk += G)
	{
		const uint32_t idx = chunk * per + tid;
		const bool live = tid < per && idx < numCmds;
		const uint32_t taskCount = live ? cur.taskCount : 0u;
		// clustercull.comp.glsl:86-95: only the clusters whose bit is set can be visible in the early pass
		uint64_t cand = 0;
		{
			const uint32_t sh = cur.mvo & 31u;
			uint64_t old = 0;
#pragma unroll
			for (int j = 0; j < 3; ++j)
			{
				const int lo = 32 * j - (int)sh;
				if (lo >= (int)taskCount)
					break;
				old |= lo >= 0 ? (uint64_t)oldw[j] << lo : (uint64_t)(oldw[j] >> (-lo));
			}
			const uint64_t valid = taskCount >= 64u ? ~0ull : (1ull << taskCount) - 1ull;
			cand = BITS ? old & valid : valid; // (without visibility bits: every valid cluster)
		}

		const uint32_t pc = (uint32_t)__builtin_popcountll(cand);
		uint32_t incl = wave_scan_inclusive_u32(pc);
		if (lane == 63)
			s_part[wave] = incl;
		if (tid < CB_CMDS)
		{
			s_visLo[tid] = (uint32_t)cand;
			s_visHi[tid] = (uint32_t)(cand >> 32);
			s_taskOffset[tid] = cur.taskOffset;
			s_excl[tid] = incl - pc; // (within the wave: the waves before it are added below)
			if (!BITS)
				s_passed[tid] = 0u;
		}

		// the prefetches, in front of this iteration's entry loads: iteration i + 1's MeshDraw and words, iteration i + 2's command
		const float4 c0 = d0, c1 = d1; // (this iteration's draw: consumed in the first round)
		float4 n0, n1;
		uint32_t neww[3];
		load_dependents(nxt, tid < per && idx + G * per < numCmds, n0, n1, neww);
		const uint32_t idx2 = (chunk + 2u * G) * per + tid;
		const BitsCommand nn = bits_load_command(a, idx2, tid < per && idx2 < numCmds);
		NV_LDS_BARRIER();

		// the list: the block's waves take the commands in turn, the LANES of a wave on a command's 64 bits — consecutive
		// addresses.  (A loop over the set bits per lane, every lane expanding its own command, writes with a stride of one
		// command's entries — at full density 128 B, all lanes on one LDS bank — and leaves the waves that own no command idle.)
		uint32_t total = 0;
		{
			const uint32_t p0 = s_part[0], p1 = s_part[1];
			total = p0 + p1; // (only commands 0 .. CB_CMDS - 1, i.e. the first two waves, have candidates)
			if (NV_DBG(a, 67108864u)) // bit 26 (experiments): every lane expands its own command, one set bit at a time
			{
				uint32_t at = tid < (uint32_t)CB_CMDS ? s_excl[tid] + (tid >= 64u ? p0 : 0u) : 0u;
				for (uint64_t rest = cand; rest; rest &= rest - 1)
					s_list[at++] = (uint16_t)((tid << 6) | (uint32_t)__builtin_ctzll(rest));
			}
			else
			for (uint32_t c = wave; c < per; c += CB_THREADS / 64)
			{
				const uint32_t lo = s_visLo[c], hi = s_visHi[c];
				const uint32_t at = s_excl[c] + (c >= 64u ? p0 : 0u);
				if ((lane < 32u ? lo >> lane : hi >> (lane - 32u)) & 1u)
					s_list[at + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u))] = (uint16_t)((c << 6) | lane);
			}
		}
		NV_LDS_BARRIER();

		for (uint32_t base = 0; base < total; base += CB_THREADS * CB_U)
			bits_round<SOA, !BITS>(a, base, total, tid, base == 0, cand != 0, c0, c1, s_list, s_taskOffset, s_draw, s_cert, s_visLo, s_visHi, s_passed);
		NV_LDS_BARRIER();
		// (the prefetches are consumed HERE, in front of the stores below: vmcnt counts stores too, and a wait for the prefetched
		// registers at the loop's end would also wait for this iteration's store and atomic to be acknowledged)
		float4 p0 = n0, p1 = n1;
		uint32_t pw0 = neww[0], pw1 = neww[1], pw2 = neww[2];
		BitsCommand pn = nn;
		asm volatile("; prefetches landed" : "+v"(p0.x), "+v"(p0.y), "+v"(p0.z), "+v"(p0.w), "+v"(p1.x), "+v"(p1.y), "+v"(p1.z), "+v"(p1.w), "+v"(pw0), "+v"(pw1), "+v"(pw2),
		             "+v"(pn.drawId), "+v"(pn.taskOffset), "+v"(pn.taskCount), "+v"(pn.mvo));

		// ---- one command per lane: the ballot (every command of the range, also those with no bit set) and the tile count
		uint64_t m = 0;
		if (live)
		{
			m = cand ? ((uint64_t)s_visHi[tid] << 32) | s_visLo[tid] : 0ull;
			a.masks[idx] = m;
			if (a.payloadCounts)
				a.payloadCounts[idx] = (uint32_t)__builtin_popcountll(m);
			// The statistic for the host's choice of the next launch's form (cluster_mask_kernel counts the commands its filter
			// does not finish): a command with a set bit was visible a frame ago, and nearly always still has a cluster the
			// filter cannot finish — counting those per entry cost 8 instructions per cluster for a tuning hint.
			passedAcc += BITS ? (cand ? 1u : 0u) : (cand ? s_passed[tid] : 0u);
		}
		if (a.payloadCounts) // (uniform) nv_taskcull's early pass: the payloads straight from here (see cluster_mask_kernel), no tile counts
		{
			const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
			for (uint64_t owners = __ballot(m != 0); owners; owners &= owners - 1)
			{
				const uint32_t src = (uint32_t)__builtin_ctzll(owners);
				const uint32_t lo = __builtin_amdgcn_readlane(mlo, src), hi = __builtin_amdgcn_readlane(mhi, src);
				const uint32_t ci = __builtin_amdgcn_readlane(idx, src);
				if ((lane < 32u ? lo >> lane : hi >> (lane - 32u)) & 1u)
					a.clusterIndices[(size_t)ci * 64 + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u))] = ci | (lane << 24);
			}
		}
		else
		{
			const uint32_t tileOf = idx / T2;
			const uint32_t tile0 = __builtin_amdgcn_readfirstlane(tileOf);
			const uint32_t pcm = (uint32_t)__builtin_popcountll(m);
			if (__ballot(pcm != 0 && tileOf != tile0) == 0) // the wave's commands with survivors sit in one tile (lanes are consecutive commands)
			{
				const uint32_t sum = wave_sum_u32(pcm);
				if (lane == 0 && sum)
					atomicAdd(&a.tileCounts->counts[bank][tile0 * CC_COUNT_STRIDE], sum);
			}
			else if (pcm)
				atomicAdd(&a.tileCounts->counts[bank][tileOf * CC_COUNT_STRIDE], pcm);
		}
		NV_LDS_BARRIER(); // the lists are rebuilt by the next iteration
		cur = nxt;
		nxt = pn;
		d0 = p0;
		d1 = p1;
		oldw[0] = pw0;
		oldw[1] = pw1;
		oldw[2] = pw2;
	}
	// the launch's statistic (cluster_mask_kernel: word 1 of a tile counter's line; the scatter kernel sums them)
	{
		const uint32_t sum = wave_sum_u32(passedAcc);
		if (lane == 0 && sum && !a.payloadCounts)
		{
			const uint32_t numTiles = (numCmds + T2 - 1) / T2;
			atomicAdd(&a.tileCounts->counts[bank][((blockIdx.x * (CB_THREADS / 64) + wave) % (numTiles ? numTiles : 1u)) * CC_COUNT_STRIDE + 1], sum);
		}
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
// The registered pool's largest |centre component| and |radius| (fp16 bit patterns without the sign: ordered like the magnitudes for finite
// values; infinities and NaNs sort above every finite value), for the filter's per-draw margin (filtermath.h filter_make): one grid-stride
// sweep over the mirror's bounds at upload time, one atomicMax pair per workgroup, then pool_bounds_finish turns the two words into
// {3 Vmax, Rmax} as floats (inf for a pool that holds a non-finite record).
__global__ __launch_bounds__(256) void pool_bounds_kernel(const uint2* __restrict__ bounds, uint32_t count, uint32_t* __restrict__ maxBits)
{
	__shared__ uint32_t s_v[4], s_r[4];
	uint32_t mv = 0, mr = 0;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u)
	{
		const uint2 b = bounds[i];
		const uint32_t x = b.x & 0x7fffu, y = (b.x >> 16) & 0x7fffu, z = b.y & 0x7fffu, r = (b.y >> 16) & 0x7fffu;
		mv = mv > x ? mv : x;
		mv = mv > y ? mv : y;
		mv = mv > z ? mv : z;
		mr = mr > r ? mr : r;
	}
	for (int o = 32; o; o >>= 1)
	{
		const uint32_t ov = (uint32_t)__shfl_xor((int)mv, o, 64), orr = (uint32_t)__shfl_xor((int)mr, o, 64);
		mv = mv > ov ? mv : ov;
		mr = mr > orr ? mr : orr;
	}
	if ((threadIdx.x & 63u) == 0)
	{
		s_v[threadIdx.x >> 6] = mv;
		s_r[threadIdx.x >> 6] = mr;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		for (int w = 1; w < 4; ++w)
		{
			mv = mv > s_v[w] ? mv : s_v[w];
			mr = mr > s_r[w] ? mr : s_r[w];
		}
		atomicMax(maxBits, mv);
		atomicMax(maxBits + 1, mr);
	}
}

__global__ void pool_bounds_finish(const uint32_t* __restrict__ maxBits, float* __restrict__ out2)
{
	const uint32_t v = maxBits[0], r = maxBits[1];
	out2[0] = v >= 0x7c00u ? __builtin_inff() : 3.0f * half_bits_to_float(v); // (3 x an 11-bit significand: exact)
	out2[1] = r >= 0x7c00u ? __builtin_inff() : half_bits_to_float(r);
}

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

template <bool LATE, bool SOA, int DEPTH, bool DIRECT = false>
static void launch_cc(hipStream_t stream, const ClusterArgs& a, uint32_t gridBlocks)
{
	dim3 grid(gridBlocks), block(CC_THREADS);
	// the direct form's packed walk (windows of 64 valid meshlets instead of one command per iteration): early form without visibility bits, over the mirror
	constexpr bool CAN_PACK = DIRECT && SOA && !LATE;
	const bool pack = CAN_PACK && a.packDirect != 0;
	if (!LATE && a.deferHiz)
	{
		if (pack)
			hipLaunchKernelGGL((cluster_mask_kernel<false, SOA, false, DEPTH, DIRECT, true, CAN_PACK>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((cluster_mask_kernel<false, SOA, false, DEPTH, DIRECT, true>), grid, block, 0, stream, a);
	}
	else if (a.cd.clusterOcclusionEnabled == 1 && a.cd.postPass == 0)
	{
		if (pack && a.packBits != 0)
			hipLaunchKernelGGL((cluster_mask_kernel<LATE, SOA, true, DEPTH, DIRECT, false, CAN_PACK>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((cluster_mask_kernel<LATE, SOA, true, DEPTH, DIRECT>), grid, block, 0, stream, a);
	}
	else if (pack)
		hipLaunchKernelGGL((cluster_mask_kernel<LATE, SOA, false, DEPTH, DIRECT, false, CAN_PACK>), grid, block, 0, stream, a);
	else
		hipLaunchKernelGGL((cluster_mask_kernel<LATE, SOA, false, DEPTH, DIRECT>), grid, block, 0, stream, a);
}

// what launch_cluster_mask resolves (late, soa, direct) and the arguments to: the packed walk? (context.hip count_cull_variant)
bool clustercull_takes_packed(const ClusterArgs& a, int late, bool soa, bool direct)
{
	return soa && direct && a.filterK > 0.0f && !late && a.packDirect != 0 && (a.deferHiz || a.packBits != 0 || !(a.cd.clusterOcclusionEnabled == 1 && a.cd.postPass == 0));
}

// any grid size (pure map); shallow = use the 4-deep filter ring (early pass over the SoA mirror only); direct = no filter
// pass (SoA mirror only; the host's guess from the previous launch's statistic — a wrong guess only costs speed)
// expectedCmds: the host's guess of the indirect command count (the previous launch's, or the explicit override): the dealing plan travels
// with the arguments for it (dealing.h); 0 = no guess
bool clustercull_takes_packed(const ClusterArgs& a, int late, bool soa, bool direct);

int launch_cluster_mask(hipStream_t stream, const ClusterArgs& args, int late, bool soa, uint32_t maskBlocks, bool shallow, bool direct, uint32_t expectedCmds)
{
	ClusterArgs a = args;
	// The dealing's delay compensation was calibrated on the filter form (a wave of config 3A lives ~11 us).  A wave of the packed walk lives 18-23 us and
	// is bound by vector issue, so a later generation — the younger waves of every SIMD — falls behind by more than its start delay: the walk has a delay
	// table of its own (dealing.h DEAL_PACKED_TABLE; round 6: 0 / 100 / 200 / 300 / 400 % of the filter form's — contract chain's cull launch 26.8 / 25.9 /
	// 24.8 / 24.7 / 25.3 us — then the last generation's entry by the wave timeline)
	if (clustercull_takes_packed(a, late, soa, direct))
		a.dealScale += DEAL_PACKED_TABLE;
	a.plan = deal_plan(expectedCmds, late ? CC_CHUNK_LATE : CC_CHUNK, !late, maskBlocks * CC_WAVES, a.cullWavesMagic, a.generations, a.genBlocks, maskBlocks, a.dealScale,
	                   a.scatterTiles, a.tilesMagic);
	if (expectedCmds == 0)
		a.plan.flags = ~0u; // (no plan: an empty pass derives its own, which costs nothing)
	if (soa && direct && a.filterK > 0.0f)
	{
		if (late)
			launch_cc<true, true, 8, true>(stream, a, maskBlocks);
		else
			launch_cc<false, true, 8, true>(stream, a, maskBlocks);
	}
	else if (late)
	{
		// (the 4-deep ring measured slower for the late pass: 46.7 vs 42.9 us, config 4)
		if (soa)
			launch_cc<true, true, 8>(stream, a, maskBlocks);
		else
			launch_cc<true, false, 8>(stream, a, maskBlocks);
	}
	else
	{
		if (soa && shallow)
			launch_cc<false, true, 4>(stream, a, maskBlocks);
		else if (soa)
			launch_cc<false, true, 8>(stream, a, maskBlocks);
		else
			launch_cc<false, false, 8>(stream, a, maskBlocks);
	}
	return (int)hipGetLastError();
}

bool clustercull_prefers_shallow(uint32_t previousCommandCount) { return previousCommandCount != 0 && previousCommandCount <= CC_SHALLOW_COMMANDS; }

// the filter pass pays for itself while it finishes more than about half of the commands (measured: DESIGN.md §4.1)
bool clustercull_prefers_direct(uint32_t previousCommandCount, uint32_t previousPassedFilter, uint32_t percent)
{
	return previousCommandCount != 0 && (uint64_t)previousPassedFilter * 100u > (uint64_t)previousCommandCount * percent;
}

// The filter form's time goes with the number of COMMANDS (its stream is bound by instruction issue: ~60 instructions per command whatever the command's
// size), the packed direct walk's with the number of valid MESHLETS / 64.  Behind drawcull's LOD select a draw's meshlets end in a partial command — config
// 3B at BASELINE scale: 250 k commands, 40 valid lanes on average — and the packed walk then wins even where the filter rejects nearly everything (22.5 against
// 34 us there, a cache-resident pool; 3A's full commands streamed from HBM: filter 21 us, packed walk 32).  The pass's FILL — valid meshlets per command slot —
// is ESTIMATED, at no cost to any kernel, from what the task pass that wrote the commands left for the host anyway (context.hip: emitting draws and commands,
// hint words 2 and 3): every emitting draw ends in one command that is half full on average, fill ~ 1 - draws / (2 commands).  (Round 6 first MEASURED it — the
// valid meshlets summed by the cull kernels, a second word beside the filter statistic, summed by the scatter launch: the scatter launch of the headline
// pass took 5.09 instead of 4.71 us by kernel-trace, whichever part of the plumbing was taken out again; the estimate decides the same way on every config.)
bool clustercull_prefers_packed(uint32_t taskCommands, uint32_t emittingDraws, uint32_t fillPercent)
{
	return taskCommands != 0 && emittingDraws != 0 && (uint64_t)(2u * (uint64_t)taskCommands - emittingDraws) * 100u < (uint64_t)taskCommands * 2u * fillPercent;
}

// early pass with visibility bits, dense form (one lane per set bit): any grid size (equal contiguous shares per block, grid-stride beyond CB_CMDS commands per block)
template <bool SOA>
static void launch_cb(hipStream_t stream, const ClusterArgs& a, uint32_t gridBlocks)
{
	// (the form without visibility bits — every valid cluster an entry, cluster_bits_kernel<SOA, false> — was the host's choice for a cache-resident pool in
	// rounds 4-5; the direct form's packed walk took its place in round 6 and it is no longer instantiated)
	dim3 grid(gridBlocks), block(CB_THREADS);
	hipLaunchKernelGGL((cluster_bits_kernel<SOA, true>), grid, block, 0, stream, a);
}

int launch_cluster_bits(hipStream_t stream, const ClusterArgs& a, bool soa, uint32_t gridBlocks)
{
	if (soa)
		launch_cb<true>(stream, a, gridBlocks);
	else
		launch_cb<false>(stream, a, gridBlocks);
	return (int)hipGetLastError();
}

// late pass with HiZ, stage 2: any grid size (grid-stride over blocks of CH_CMDS commands)
int launch_cluster_hiz(hipStream_t stream, const ClusterArgs& a, bool soa, uint32_t gridBlocks)
{
	dim3 grid(gridBlocks), block(CH_THREADS);
	const bool bits = a.cd.clusterOcclusionEnabled == 1 && a.cd.postPass == 0;
	if (soa)
	{
		if (bits)
			hipLaunchKernelGGL((cluster_hiz_kernel<true, true>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((cluster_hiz_kernel<true, false>), grid, block, 0, stream, a);
	}
	else
	{
		if (bits)
			hipLaunchKernelGGL((cluster_hiz_kernel<false, true>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((cluster_hiz_kernel<false, false>), grid, block, 0, stream, a);
	}
	return (int)hipGetLastError();
}

// one workgroup per scatter tile (context.hip: one per CU, at most CC_MAX_SCATTER_TILES); no workgroup waits on another.
// waves = 16 (one command per lane: the shortest launch) or 4 / 8 (NV_OPT_SCATTER_WAVES: fewer wave slots per CU, for callers
// that keep several passes in flight — the launch then shares the chip with a neighbour pass's cull launch)
int launch_cluster_scatter(hipStream_t stream, const ClusterArgs& a, uint32_t scatterBlocks, uint32_t waves)
{
	if (waves == 4u)
		hipLaunchKernelGGL((cluster_scatter_kernel<4>), dim3(scatterBlocks), dim3(4 * 64), 0, stream, a);
	else if (waves == 8u)
		hipLaunchKernelGGL((cluster_scatter_kernel<8>), dim3(scatterBlocks), dim3(8 * 64), 0, stream, a);
	else
		hipLaunchKernelGGL((cluster_scatter_kernel<16>), dim3(scatterBlocks), dim3(16 * 64), 0, stream, a);
	return (int)hipGetLastError();
}

size_t clustercull_mask_bytes() { return (size_t)(NV_TASK_WGLIMIT + 64) * sizeof(uint64_t); }

// Room per sub-list of the late pass's survivor-command list (32-B entries): 512 k listed commands in all.  The list serves the
// sparse passes — a few per cent of the commands have survivors and they cluster per draw, so contiguous command ranges
// are badly balanced; a pass that lists more than fits raises the overflow flag and the stage scans contiguous ranges,
// which is as good when most commands have survivors.
uint32_t clustercull_list_stride() { return 2048u; }
size_t clustercull_list_bytes() { return ((size_t)CC_LISTS * clustercull_list_stride() + 64) * 2 * sizeof(uint4); }

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

// poolWords: 2 x u32 scratch + 2 x float result (ClusterArgs::poolBounds points at the floats)
int launch_soa_split(hipStream_t stream, const NvMeshlet* meshlets, uint32_t count, uint32_t padded, uint2* bounds, uint32_t* cones, uint32_t* poolWords)
{
	hipLaunchKernelGGL(soa_split_kernel, dim3((padded + 255) / 256), dim3(256), 0, stream, meshlets, count, padded, bounds, cones);
	(void)hipMemsetAsync(poolWords, 0, 2 * sizeof(uint32_t), stream);
	const uint32_t blocks = (count + 255) / 256 < 2048u ? ((count + 255) / 256 ? (count + 255) / 256 : 1u) : 2048u;
	hipLaunchKernelGGL(pool_bounds_kernel, dim3(blocks), dim3(256), 0, stream, bounds, count, poolWords);
	hipLaunchKernelGGL(pool_bounds_finish, dim3(1), dim3(1), 0, stream, poolWords, reinterpret_cast<float*>(poolWords + 2));
	return (int)hipGetLastError();
}

} // namespace nv
