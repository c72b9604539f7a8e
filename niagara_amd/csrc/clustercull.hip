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

template <bool SOA>
NV_DEV LaneData load_lane(const ClusterArgs& a, uint32_t mi, bool valid)
{
	LaneData l = { 0, 0, 0, 0 };
	if (valid)
	{
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
	}
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
template <bool LATE>
NV_DEV uint64_t cull_command(const ClusterArgs& a, const NvMeshTaskCommand& cmd, const DrawUniform& u, const LaneData& l, uint32_t lane)
{
	const NvCullData& cd = a.cd;
	const bool valid = lane < cmd.taskCount;
	const bool useBits = cd.clusterOcclusionEnabled == 1 && cd.postPass == 0;
	const uint32_t mvi = lane + cmd.meshletVisibilityOffset;

	bool visible = valid;
	bool skip = false;

	if (useBits)
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

template <bool LATE>
NV_DEV uint32_t load_mvb_word(const ClusterArgs& a, const NvMeshTaskCommand& cmd, uint32_t lane, bool valid)
{
	if (a.cd.clusterOcclusionEnabled == 1 && a.cd.postPass == 0 && valid)
	{
		uint32_t mvi = lane + cmd.meshletVisibilityOffset;
		// late pass: other waves update neighbouring bits of shared words concurrently; an agent-scope load keeps
		// the read out of a stale L1 line (our own bit is only ever written by this lane, later)
		return __hip_atomic_load(a.mvb + (mvi >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	return 0;
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
constexpr uint32_t CC_TMAX = 4096; // commands per tile: 32 KiB of ballots in LDS

template <int K>
struct Batch
{
	NvMeshTaskCommand cmd[K];
	LaneData ld[K];
};

template <bool LATE, bool SOA, int K>
NV_DEV void load_batch(const ClusterArgs& a, Batch<K>& b, uint32_t first, uint32_t i, uint32_t n, uint32_t lane)
{
#pragma unroll
	for (int k = 0; k < K; ++k)
		b.cmd[k] = (i + k < n) ? load_command(a.commands, first + i + k) : NvMeshTaskCommand{ 0, 0, 0, 0, 0 };
#pragma unroll
	for (int k = 0; k < K; ++k)
	{
		const bool valid = lane < b.cmd[k].taskCount;
		b.ld[k] = load_lane<SOA>(a, b.cmd[k].taskOffset + lane, valid);
		b.ld[k].mvbWord = load_mvb_word<LATE>(a, b.cmd[k], lane, valid);
	}
}

template <bool LATE, bool SOA, int K>
__global__ __launch_bounds__(CC_THREADS) void clustercull_kernel(ClusterArgs a)
{
	__shared__ uint64_t s_mask[CC_TMAX];
	__shared__ uint32_t s_part[CC_WAVES];
	__shared__ uint32_t s_base;

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t G = gridDim.x;
	constexpr uint32_t STEP = CC_WAVES * K;

	const uint32_t numCmds = indirect_command_count(a);
	uint32_t T = ((numCmds + G - 1) / G + STEP - 1) / STEP * STEP;
	T = T < STEP ? STEP : (T > CC_TMAX ? CC_TMAX : T);
	const uint32_t numTiles = (numCmds + T - 1) / T;
	const uint32_t epoch = load_epoch(a.ctl);
	const uint32_t base0 = a.clusterCount4[0];
	const bool dbgNoScan = a.debugMode & 1u, dbgNoScatter = a.debugMode & 4u; // experiments only

	for (uint32_t tile = blockIdx.x; tile < numTiles; tile += G)
	{
		const uint32_t first = tile * T;
		const uint32_t n = numCmds - first < T ? numCmds - first : T;

		// ---- phase 1: cull; the loads of batch i+1 are in flight while batch i is tested
		uint32_t waveCount = 0;
		uint32_t curDraw = ~0u;
		DrawUniform du = {};
		Batch<K> cur, nxt;
		load_batch<LATE, SOA, K>(a, cur, first, wave * K, n, lane);
		for (uint32_t i = wave * K; i < n; i += STEP)
		{
			if (i + STEP < n)
				load_batch<LATE, SOA, K>(a, nxt, first, i + STEP, n, lane);
#pragma unroll
			for (int k = 0; k < K; ++k)
			{
				if (i + k < n)
				{
					uint64_t m = 0;
					if (cur.cmd[k].taskCount)
					{
						if (cur.cmd[k].drawId != curDraw) // a draw's task commands are consecutive: usually a hit
						{
							curDraw = cur.cmd[k].drawId;
							du = load_draw(a.draws, curDraw);
						}
						m = cull_command<LATE>(a, cur.cmd[k], du, cur.ld[k], lane);
					}
					if (lane == 0)
						s_mask[i + k] = m;
					waveCount += (uint32_t)__builtin_popcountll(m);
				}
			}
			cur = nxt;
		}

		// ---- phase 2 + 3: tile total, chained scan across tiles
		if (lane == 0)
			s_part[wave] = waveCount;
		__syncthreads();
		if (wave == 0)
		{
			uint32_t aggregate = 0;
#pragma unroll
			for (int w = 0; w < CC_WAVES; ++w)
				aggregate += s_part[w];
			uint32_t exclusive = dbgNoScan ? 0u : lookback_exclusive(a.state, a.ctl, tile, epoch, aggregate, base0);
			if (lane == 0)
			{
				s_base = exclusive;
				if (tile == numTiles - 1)
				{
					a.clusterCount4[0] = exclusive + aggregate; // what the chain of atomicAdds leaves in clusterCount
					advance_epoch(a.ctl, epoch);                 // every tile has published: nobody polls any more
				}
			}
		}
		__syncthreads();

		// ---- phase 4: ordered scatter, 256 commands per step, one command per lane for the scan and one command
		// per iteration for the (coalesced) stores; clustercull.comp.glsl:137-138 drops entries past CLUSTER_LIMIT
		uint32_t running = s_base;
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
		const bool valid = lane < cmd.taskCount;
		LaneData ld = load_lane<SOA>(a, cmd.taskOffset + lane, valid);
		ld.mvbWord = load_mvb_word<LATE>(a, cmd, lane, valid);
		const DrawUniform du = load_draw(a.draws, cmd.drawId);
		const uint64_t m = cmd.taskCount ? cull_command<LATE>(a, cmd, du, ld, lane) : 0ull;
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
		LaneData ld = load_lane<SOA>(a, cmd.taskOffset + lane, true);
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

template <int K>
static void launch_cc_k(hipStream_t stream, const ClusterArgs& a, int late, bool soa, uint32_t gridBlocks)
{
	dim3 grid(gridBlocks), block(CC_THREADS);
	if (late)
	{
		if (soa)
			hipLaunchKernelGGL((clustercull_kernel<true, true, K>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((clustercull_kernel<true, false, K>), grid, block, 0, stream, a);
	}
	else
	{
		if (soa)
			hipLaunchKernelGGL((clustercull_kernel<false, true, K>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((clustercull_kernel<false, false, K>), grid, block, 0, stream, a);
	}
}

// K = commands per wave per tile (tile = 4*K commands); chosen per context (NV_CC_K, default CC_K_DEFAULT)
constexpr int CC_K_DEFAULT = 2;

int launch_clustercull(hipStream_t stream, const ClusterArgs& a, int late, bool soa, uint32_t gridBlocks, int k)
{
	switch (k)
	{
	case 1: launch_cc_k<1>(stream, a, late, soa, gridBlocks); break;
	case 2: launch_cc_k<2>(stream, a, late, soa, gridBlocks); break;
	case 3: launch_cc_k<3>(stream, a, late, soa, gridBlocks); break;
	case 4: launch_cc_k<4>(stream, a, late, soa, gridBlocks); break;
	default: launch_cc_k<CC_K_DEFAULT>(stream, a, late, soa, gridBlocks); break;
	}
	return (int)hipGetLastError();
}

uint32_t clustercull_max_tiles(uint32_t gridBlocks) { return (gridBlocks > NV_TASK_WGLIMIT / CC_TMAX ? gridBlocks : NV_TASK_WGLIMIT / CC_TMAX) + 1; }
int clustercull_default_k() { return CC_K_DEFAULT; }

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
