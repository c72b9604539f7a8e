// drawcull.hip — per-draw frustum / HiZ cull, LOD selection and ordered indirect-command compaction for gfx950.
//
// Replaces src/shaders/drawcull.comp.glsl:54-156 (4 pipelines LATE x TASK, src/niagara.cpp:724-727).
//
// Mapping to CDNA4:
//   * one lane per draw; a workgroup owns one contiguous tile of draws per pass and keeps the next step's 48-B
//     MeshDraw record (3 x 16-B loads) + visibility word in flight while it tests the current one;
//   * the emit count of a draw (1 command, or ceil(meshletCount/64) task commands) is scanned inside the wave with
//     DPP shuffles, across waves through LDS, and across tiles through ordered.cuh — append index = exclusive prefix
//     in draw order, no per-draw global atomic (drawcull.comp.glsl:123,143);
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
constexpr uint32_t DC_TMAX = 1024;    // draws per tile: 5 x 4 KiB of per-draw results in LDS
constexpr uint32_t DC_MESH_LDS = 64;  // meshes staged in LDS (13 KiB) when the table is registered and small enough


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
NV_DEV DrawResult decide_draw(const DrawArgs& a, const char* meshBase, uint32_t di, const float4& d0, const float4& d1, const uint4& d2, uint32_t oldVis)
{
	const NvCullData& cd = a.cd;
	DrawResult res = { 0, 0, oldVis };

	if (d2.z != cd.postPass) // drawData.postPass
		return res;
	if (!LATE && oldVis == 0)
		return res;

	const uint32_t meshIndex = d2.x;
	const char* mesh = meshBase + (size_t)meshIndex * sizeof(NvMesh);
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

struct DrawLoad
{
	float4 d0, d1; // position.xyz, scale | orientation
	uint4 d2;      // meshIndex, meshletVisibilityOffset, postPass, materialIndex
	uint32_t oldVis;
};

NV_DEV DrawLoad load_draw_record(const DrawArgs& a, uint32_t di)
{
	DrawLoad l;
	const float4* p = reinterpret_cast<const float4*>(a.draws + di);
	l.d0 = p[0];
	l.d1 = p[1];
	l.d2 = *reinterpret_cast<const uint4*>(p + 2);
	l.oldVis = a.dvb[di];
	return l;
}

// Static tiles, one per workgroup per pass (see ordered.cuh and clustercull.hip): phase 1 decide -> per-draw emit
// count / LOD / old visibility in LDS, phase 2 tile total, phase 3 look-back across tiles, phase 4 ordered emit.
template <bool LATE, bool TASK, bool MESH_LDS>
__global__ __launch_bounds__(DC_THREADS) void drawcull_kernel(DrawArgs a)
{
	__shared__ uint32_t s_count[DC_TMAX];
	__shared__ uint32_t s_flags[DC_TMAX];
	__shared__ uint32_t s_old[DC_TMAX];
	__shared__ uint32_t s_mesh[DC_TMAX]; // meshIndex and meshletVisibilityOffset of the tile's draws, for the emit phase
	__shared__ uint32_t s_mvo[DC_TMAX];
	// the Mesh table (center/radius, LOD errors, LOD ranges) is read by every draw: staged once per workgroup when
	// nv_upload_meshes registered a table of at most DC_MESH_LDS meshes, otherwise gathered from global memory
	__shared__ __attribute__((aligned(16))) uint32_t s_meshTable[MESH_LDS ? DC_MESH_LDS * sizeof(NvMesh) / 4 : 4];
	__shared__ uint32_t s_part[DC_WAVES];
	__shared__ uint32_t s_scratch[16];

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t G = gridDim.x;

	const uint32_t drawCount = a.cd.drawCount;
	uint32_t T = ((drawCount + G - 1) / G + DC_THREADS - 1) / DC_THREADS * DC_THREADS;
	T = T < DC_THREADS ? DC_THREADS : (T > DC_TMAX ? DC_TMAX : T);
	const uint32_t numTiles = (drawCount + T - 1) / T;
	const uint32_t epoch = load_epoch(a.ctl);
	const uint32_t base0 = a.fusedReset ? 0u : a.count4[0];

	const char* meshBase = reinterpret_cast<const char*>(a.meshes);
	if (MESH_LDS)
	{
		const uint32_t words = a.meshCount * (uint32_t)(sizeof(NvMesh) / 4);
		const uint32_t* src = reinterpret_cast<const uint32_t*>(a.meshes);
		for (uint32_t i = tid; i < words; i += DC_THREADS)
			s_meshTable[i] = src[i];
		meshBase = reinterpret_cast<const char*>(s_meshTable);
		__syncthreads();
	}

	for (uint32_t tile = blockIdx.x; tile < numTiles; tile += G)
	{
		const uint32_t first = tile * T;
		const uint32_t n = drawCount - first < T ? drawCount - first : T;

		// ---- phase 1: one draw per lane per step, four steps' records (4 x 52 bytes per lane) requested before any of
		// them is used.  Indices past the tile are clamped, not branched, so that the loads are unconditional and the
		// compiler can count them (s_waitcnt vmcnt(N)) instead of draining after each one.
		uint32_t threadSum = 0;
		constexpr int DC_BATCH = 4;
		for (uint32_t c0 = 0; c0 < n; c0 += DC_THREADS * DC_BATCH)
		{
			DrawLoad ld[DC_BATCH];
#pragma unroll
			for (int j = 0; j < DC_BATCH; ++j)
			{
				const uint32_t c = c0 + j * DC_THREADS + tid;
				ld[j] = load_draw_record(a, first + (c < n ? c : n - 1));
			}
#pragma unroll
			for (int j = 0; j < DC_BATCH; ++j)
			{
				const uint32_t c = c0 + j * DC_THREADS + tid;
				if (c < n)
				{
					DrawResult res = decide_draw<LATE, TASK>(a, meshBase, first + c, ld[j].d0, ld[j].d1, ld[j].d2, ld[j].oldVis);
					s_count[c] = res.count;
					s_flags[c] = res.lodWord;
					s_old[c] = ld[j].oldVis;
					s_mesh[c] = ld[j].d2.x;
					s_mvo[c] = ld[j].d2.y;
					threadSum += res.count;
				}
			}
		}

		// ---- phase 2 + 3
		uint32_t waveSum = wave_sum_u32(threadSum);
		if (lane == 0)
			s_part[wave] = waveSum;
		__syncthreads();
		uint32_t aggregate = 0;
#pragma unroll
		for (int w = 0; w < DC_WAVES; ++w)
			aggregate += s_part[w];
		const uint32_t exclusive = lookback_exclusive(a.state, a.ctl, tile, epoch, aggregate, base0, s_scratch);
		if (tid == 0 && tile == numTiles - 1)
		{
			a.count4[0] = exclusive + aggregate;
			advance_epoch(a.ctl, epoch);
		}
		__syncthreads(); // s_part is reused by the emit scan

		// ---- phase 4: ordered emit, 256 draws per step
		uint32_t running = exclusive;
		for (uint32_t c0 = 0; c0 < n; c0 += DC_THREADS)
		{
			const uint32_t c = c0 + tid;
			const uint32_t cnt = c < n ? s_count[c] : 0u;
			const uint32_t flags = c < n ? s_flags[c] : 0u;
			const uint32_t incl = wave_inclusive_scan(cnt, lane);
			__syncthreads();
			if (lane == 63)
				s_part[wave] = incl;
			__syncthreads();
			uint32_t waveBase = running;
#pragma unroll
			for (int w = 0; w < DC_WAVES; ++w)
			{
				uint32_t p = s_part[w];
				waveBase += w < (int)wave ? p : 0u;
				running += p;
			}
			const uint32_t dci = waveBase + incl - cnt;
			const bool emit = (flags & 0x100u) != 0;
			const uint32_t lodIndex = flags & 0xffu;
			const uint32_t di = first + c;

			if (TASK)
			{
				// wave-cooperative expansion, one owning lane at a time (drawcull.comp.glsl:120-139)
				uint64_t owners = __ballot(emit && cnt != 0);
				NvMeshTaskCommand* tc = static_cast<NvMeshTaskCommand*>(a.commands);
				uint32_t meshIndex = 0, mvo = 0, oldVis = 0;
				if (emit && cnt != 0)
				{
					meshIndex = s_mesh[c];
					mvo = s_mvo[c];
					oldVis = s_old[c];
				}
				while (owners)
				{
					const int src = __builtin_ctzll(owners);
					owners &= owners - 1;
					const uint32_t oDraw = first + c0 + wave * 64 + src;
					const uint32_t oDci = __builtin_amdgcn_readlane(dci, src);
					const uint32_t oGroups = __builtin_amdgcn_readlane(cnt, src);
					const uint32_t oLod = __builtin_amdgcn_readlane(lodIndex, src);
					const uint32_t oVis = __builtin_amdgcn_readlane(oldVis, src);
					const uint32_t oMesh = __builtin_amdgcn_readlane(meshIndex, src);
					const uint32_t oMvo = __builtin_amdgcn_readlane(mvo, src);
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
			else if (emit)
			{
				// drawcull.comp.glsl:141-150
				const uint32_t meshIndex = s_mesh[c];
				const char* mesh = meshBase + (size_t)meshIndex * sizeof(NvMesh);
				const uint32_t indexOffset = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lodIndex + 0);
				const uint32_t indexCount = *reinterpret_cast<const uint32_t*>(mesh + 48 + 20 * lodIndex + 4);
				const uint32_t vertexOffset = *reinterpret_cast<const uint32_t*>(mesh + 16);
				uint2* dc = reinterpret_cast<uint2*>(static_cast<NvMeshDrawCommand*>(a.commands) + dci);
				dc[0] = make_uint2(di, indexCount);
				dc[1] = make_uint2(1u, indexOffset);
				dc[2] = make_uint2(vertexOffset, 0u);
			}
		}
		__syncthreads(); // LDS is rewritten by the next tile

		if (tile == numTiles - 1 && ((epoch + 1) & 0x3fffffffu) == 0)
			for (uint32_t i = tid; i < a.stateCapacity; i += DC_THREADS)
				a.state[i] = 0;
	}
}

template <bool MESH_LDS>
static void launch_dc(hipStream_t stream, const DrawArgs& a, int late, int task, uint32_t gridBlocks)
{
	dim3 grid(gridBlocks), block(DC_THREADS);
	if (late)
	{
		if (task)
			hipLaunchKernelGGL((drawcull_kernel<true, true, MESH_LDS>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((drawcull_kernel<true, false, MESH_LDS>), grid, block, 0, stream, a);
	}
	else
	{
		if (task)
			hipLaunchKernelGGL((drawcull_kernel<false, true, MESH_LDS>), grid, block, 0, stream, a);
		else
			hipLaunchKernelGGL((drawcull_kernel<false, false, MESH_LDS>), grid, block, 0, stream, a);
	}
}

int launch_drawcull(hipStream_t stream, const DrawArgs& a, int late, int task, uint32_t gridBlocks)
{
	if (a.meshCount && a.meshCount <= DC_MESH_LDS)
		launch_dc<true>(stream, a, late, task, gridBlocks);
	else
		launch_dc<false>(stream, a, late, task, gridBlocks);
	return (int)hipGetLastError();
}

uint32_t drawcull_max_tiles(uint32_t drawCount, uint32_t gridBlocks)
{
	uint32_t byCap = (drawCount + DC_TMAX - 1) / DC_TMAX;
	return (byCap > gridBlocks ? byCap : gridBlocks) + 1;
}

} // namespace nv
