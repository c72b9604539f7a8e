// drawcull.hip — per-draw frustum / HiZ cull, LOD selection and ordered indirect-command compaction for gfx950.
//
// Replaces src/shaders/drawcull.comp.glsl:54-156 (4 pipelines LATE x TASK, src/niagara.cpp:724-727).
//
// Mapping to CDNA4:
//   * one lane per draw, 4 draws per lane per tile (tile = 1024 consecutive draws per 256-thread workgroup): the
//     12 independent 16-B loads of the four 48-B MeshDraw records are issued before any arithmetic;
//   * the emit count of a draw (1 command, or ceil(meshletCount/64) task commands) is scanned inside the wave with
//     DPP shuffles, across waves through 16 LDS words, and across tiles through ordered.cuh — append index =
//     exclusive prefix in draw order, no per-draw global atomic (drawcull.comp.glsl:123,143);
//   * TASK mode expands a draw's commands wave-cooperatively: the owning lane's (draw, LOD range, dci) is broadcast
//     with readlane and all 64 lanes write consecutive 20-B MeshTaskCommands, instead of one lane looping over
//     up to hundreds of commands (drawcull.comp.glsl:131-138).
#include "cullmath.cuh"
#include "ordered.cuh"
#include "args.cuh"

namespace nv
{

constexpr int DC_WAVES = 4;
constexpr int DC_THREADS = DC_WAVES * 64;
constexpr int DC_ITEMS = 4; // draws per lane per tile
constexpr uint32_t DC_TILE = DC_THREADS * DC_ITEMS;


struct DrawResult
{
	uint32_t count;   // commands this draw appends (0 if none)
	uint32_t lodWord; // lodIndex | emit<<8
	uint32_t oldVis;  // drawVisibility[di] before this pass (lateDrawVisibility)
};

NV_DEV uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
	for (int o = 1; o < 64; o <<= 1)
	{
		uint32_t t = __shfl_up(v, o, 64);
		if ((int)lane >= o)
			v += t;
	}
	return v;
}

// drawcull.comp.glsl:56-118 + :154-155 for one draw
template <bool LATE, bool TASK>
NV_DEV DrawResult decide_draw(const DrawArgs& a, uint32_t di, const float4& d0, const float4& d1, const uint4& d2, uint32_t oldVis)
{
	const NvCullData& cd = a.cd;
	DrawResult res = { 0, 0, oldVis };

	if (d2.z != cd.postPass) // drawData.postPass
		return res;
	if (!LATE && oldVis == 0)
		return res;

	const uint32_t meshIndex = d2.x;
	const char* mesh = reinterpret_cast<const char*>(a.meshes + meshIndex);
	const float4 cr = *reinterpret_cast<const float4*>(mesh); // center.xyz, radius

	f3 q = { d1.x, d1.y, d1.z };
	f3 c = sphere_center(cd, f3{ cr.x, cr.y, cr.z }, q, d1.w, d0.w, f3{ d0.x, d0.y, d0.z });
	float radius = cr.w * d0.w;

	bool visible = frustum_test(cd, c, radius);
	visible = visible || cd.cullingEnabled == 0;

	if (LATE && visible && cd.occlusionEnabled == 1)
		visible = hiz_test(cd, a.pyr, c, radius);

	// TASK_CULL == 1 (src/config.h:8)
	if (visible && (!LATE || cd.clusterOcclusionEnabled == 1 || oldVis == 0 || cd.postPass != 0))
	{
		uint32_t lodIndex = 0;
		if (cd.lodEnabled == 1)
		{
			float distance = gl_max(length3(c) - radius, 0.0f);
			float threshold = distance * cd.lodTarget / d0.w;
			const uint32_t lodCount = *reinterpret_cast<const uint32_t*>(mesh + 32);
			for (uint32_t i = 1; i < lodCount; ++i)
			{
				float err = *reinterpret_cast<const float*>(mesh + 48 + 20 * i + 16);
				if (err < threshold)
					lodIndex = i;
			}
		}
		res.lodWord = lodIndex | 0x100u;
		if (TASK)
		{
			uint32_t meshletCount = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lodIndex + 12);
			res.count = (meshletCount + NV_TASK_WGSIZE - 1) / NV_TASK_WGSIZE;
		}
		else
			res.count = 1;
	}

	if (LATE)
		a.dvb[di] = visible ? 1u : 0u;
	return res;
}

template <bool LATE, bool TASK>
__global__ __launch_bounds__(DC_THREADS) void drawcull_kernel(DrawArgs a)
{
	__shared__ uint32_t s_tile;
	__shared__ uint32_t s_part[DC_ITEMS * DC_WAVES];
	__shared__ uint32_t s_base;
	__shared__ uint32_t s_wrapped;

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t shard = blockIdx.x % NV_SHARDS;

	const uint32_t epoch = load_epoch(a.ctl);
	const uint32_t drawCount = a.cd.drawCount;
	const uint32_t numTiles = (drawCount + DC_TILE - 1) / DC_TILE;
	const uint32_t base0 = a.count4[0];

	uint32_t nextTile = 0;
	if (tid == 0)
		nextTile = draw_ticket(a.ctl, shard);

	for (;;)
	{
		if (tid == 0)
			s_tile = nextTile;
		__syncthreads();
		const uint32_t tile = __builtin_amdgcn_readfirstlane(s_tile);
		if (tile >= numTiles)
			break;
		if (tid == 0)
			nextTile = draw_ticket(a.ctl, shard);

		// ---- loads: item j of this lane is draw tile*1024 + j*256 + tid (coalesced across the workgroup)
		float4 d0[DC_ITEMS], d1[DC_ITEMS];
		uint4 d2[DC_ITEMS];
		uint32_t oldVis[DC_ITEMS];
		uint32_t di[DC_ITEMS];
#pragma unroll
		for (int j = 0; j < DC_ITEMS; ++j)
		{
			di[j] = tile * DC_TILE + j * DC_THREADS + tid;
			if (di[j] < drawCount)
			{
				const float4* p = reinterpret_cast<const float4*>(a.draws + di[j]);
				d0[j] = p[0];
				d1[j] = p[1];
				d2[j] = *reinterpret_cast<const uint4*>(p + 2);
				oldVis[j] = a.dvb[di[j]];
			}
		}

		// ---- decisions
		DrawResult res[DC_ITEMS];
		uint32_t incl[DC_ITEMS];
#pragma unroll
		for (int j = 0; j < DC_ITEMS; ++j)
		{
			res[j] = DrawResult{ 0, 0, 0 };
			if (di[j] < drawCount)
				res[j] = decide_draw<LATE, TASK>(a, di[j], d0[j], d1[j], d2[j], oldVis[j]);
			incl[j] = wave_inclusive_scan(res[j].count, lane);
			if (lane == 63)
				s_part[j * DC_WAVES + wave] = incl[j];
		}
		__syncthreads();

		// ---- tile aggregate + chained scan
		if (wave == 0)
		{
			uint32_t aggregate = 0;
#pragma unroll
			for (int i = 0; i < DC_ITEMS * DC_WAVES; ++i)
				aggregate += s_part[i];
			uint32_t exclusive = lookback_exclusive(a.state, a.ctl, tile, epoch, aggregate, base0);
			if (lane == 0)
			{
				s_base = exclusive;
				if (tile == numTiles - 1)
					a.count4[0] = exclusive + aggregate;
			}
		}
		__syncthreads();

		// ---- emit in draw order: parts are ordered item-major, wave-minor
		uint32_t partBase = s_base;
#pragma unroll
		for (int j = 0; j < DC_ITEMS; ++j)
		{
			uint32_t before = partBase;
#pragma unroll
			for (int w = 0; w < DC_WAVES; ++w)
			{
				uint32_t p = s_part[j * DC_WAVES + w];
				before += w < (int)wave ? p : 0u;
				partBase += p;
			}
			const uint32_t dci = before + incl[j] - res[j].count;
			const bool emit = (res[j].lodWord & 0x100u) != 0;
			const uint32_t lodIndex = res[j].lodWord & 0xffu;

			if (TASK)
			{
				// wave-cooperative expansion, one owning lane at a time (drawcull.comp.glsl:120-139)
				uint64_t owners = __ballot(emit && res[j].count != 0);
				NvMeshTaskCommand* tc = static_cast<NvMeshTaskCommand*>(a.commands);
				while (owners)
				{
					const int src = __builtin_ctzll(owners);
					owners &= owners - 1;
					const uint32_t oDraw = __shfl(di[j], src, 64);
					const uint32_t oDci = __shfl(dci, src, 64);
					const uint32_t oGroups = __shfl(res[j].count, src, 64);
					const uint32_t oLod = __shfl(lodIndex, src, 64);
					const uint32_t oVis = __shfl(res[j].oldVis, src, 64);
					const uint32_t oMesh = __shfl(d2[j].x, src, 64);
					const uint32_t oMvo = __shfl(d2[j].y, src, 64);
					if (oDci + oGroups <= NV_TASK_WGLIMIT) // drop the whole draw on overflow (:128)
					{
						const char* mesh = reinterpret_cast<const char*>(a.meshes + oMesh);
						const uint32_t meshletOffset = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * oLod + 8);
						const uint32_t meshletCount = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * oLod + 12);
						for (uint32_t i = lane; i < oGroups; i += 64)
						{
							NvMeshTaskCommand c;
							c.drawId = oDraw;
							c.taskOffset = meshletOffset + i * NV_TASK_WGSIZE;
							uint32_t rest = meshletCount - i * NV_TASK_WGSIZE;
							c.taskCount = rest < NV_TASK_WGSIZE ? rest : NV_TASK_WGSIZE;
							c.lateDrawVisibility = oVis;
							c.meshletVisibilityOffset = oMvo + i * NV_TASK_WGSIZE;
							tc[oDci + i] = c;
						}
					}
				}
			}
			else if (emit)
			{
				// drawcull.comp.glsl:141-150
				const char* mesh = reinterpret_cast<const char*>(a.meshes + d2[j].x);
				const uint32_t indexOffset = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lodIndex + 0);
				const uint32_t indexCount = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lodIndex + 4);
				const uint32_t vertexOffset = *reinterpret_cast<const uint32_t*>(mesh + 16);
				uint2* dc = reinterpret_cast<uint2*>(static_cast<NvMeshDrawCommand*>(a.commands) + dci);
				dc[0] = make_uint2(di[j], indexCount);
				dc[1] = make_uint2(1u, indexOffset);
				dc[2] = make_uint2(vertexOffset, 0u);
			}
		}
	}

	if (tid == 0)
		s_wrapped = leave_and_maybe_reset(a.ctl, epoch) ? 1u : 0u;
	__syncthreads();
	if (s_wrapped)
		for (uint32_t i = tid; i < a.stateCapacity; i += DC_THREADS)
			a.state[i] = 0;
}

int launch_drawcull(hipStream_t stream, const DrawArgs& a, int late, int task, uint32_t gridBlocks)
{
	dim3 grid(gridBlocks), block(DC_THREADS);
	if (late)
	{
		if (task)
			hipLaunchKernelGGL((drawcull_kernel<true, true>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((drawcull_kernel<true, false>), grid, block, 0, stream, a);
	}
	else
	{
		if (task)
			hipLaunchKernelGGL((drawcull_kernel<false, true>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((drawcull_kernel<false, false>), grid, block, 0, stream, a);
	}
	return (int)hipGetLastError();
}

uint32_t drawcull_tile_draws() { return DC_TILE; }

} // namespace nv
