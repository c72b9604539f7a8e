// context.hip — the extern "C" boundary declared in include/niagara_vis.h: context lifetime, argument packing and
// kernel launches.  No torch types, no exceptions across the ABI; every entry point only enqueues on the caller's
// stream (nv_status and scratch growth are the only synchronising calls).
#include <hip/hip_runtime.h>

#include <atomic>
#include <new>
#include <algorithm>
#include <vector>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>

#include "../../include/niagara_vis.h"
#include "cullmath.h"
#include "args.h"
#include "filtermath.h"

namespace nv
{

int launch_cluster_mask(hipStream_t, const ClusterArgs&, int late, bool soa, uint32_t maskBlocks, bool shallow, bool direct, uint32_t expectedCmds);
bool clustercull_prefers_shallow(uint32_t previousCommandCount);
bool clustercull_prefers_direct(uint32_t previousCommandCount, uint32_t previousPassedFilter, uint32_t percent);
bool clustercull_takes_packed(const ClusterArgs&, int late, bool soa, bool direct);
bool clustercull_prefers_packed(uint32_t taskCommands, uint32_t emittingDraws, uint32_t fillPercent);
int launch_cluster_scatter(hipStream_t, const ClusterArgs&, uint32_t scatterBlocks, uint32_t waves);
int launch_cluster_hiz(hipStream_t, const ClusterArgs&, bool soa, uint32_t gridBlocks);
int launch_cluster_bits(hipStream_t, const ClusterArgs&, bool soa, uint32_t gridBlocks);
size_t clustercull_mask_bytes();
size_t clustercull_list_bytes();
uint32_t clustercull_list_stride();
int launch_taskcull(hipStream_t, const ClusterArgs&, int late, bool soa, uint32_t gridBlocks);
int launch_probe(hipStream_t, const ClusterArgs&, bool soa, uint32_t gridBlocks);
int launch_soa_split(hipStream_t, const NvMeshlet*, uint32_t count, uint32_t padded, uint2* bounds, uint32_t* cones, uint32_t* poolWords);
int launch_drawcull(hipStream_t, const DrawArgs&, int late, int task);
int launch_draw_split(hipStream_t, const NvMeshDraw*, const NvMesh*, uint32_t meshCount, uint32_t first, uint32_t count, float4* world, uint2* scaleMesh, uint32_t* postPass);
size_t drawcull_result_bytes(uint32_t drawCount);
int launch_tasksubmit(hipStream_t, uint32_t* count4, NvMeshTaskCommand* commands);
int launch_reset_count(hipStream_t, uint32_t* a, uint32_t* b);
int launch_cluster_expand(hipStream_t, const NvMeshTaskCommand*, const NvMeshlet*, const uint32_t* clusterIndices, const uint32_t* cc4,
                          NvClusterRecord* records, uint32_t capacity, uint64_t* totals, unsigned long long* partials, uint32_t gridBlocks);
int launch_clustersubmit(hipStream_t, uint32_t* cc4, uint32_t* clusterIndices);
int launch_pack_counts(hipStream_t, const uint32_t*, const uint32_t*, const uint32_t*, uint64_t*);
int launch_depthreduce(hipStream_t, const float* depth, uint32_t w, uint32_t h, const NvPyramidDesc& pyr);
int launch_trianglecull(hipStream_t, const TriangleArgs& a, uint32_t gridBlocks);
int launch_meshlet_bounds(hipStream_t, const NvVertex* vertices, const uint32_t* data, NvMeshlet* meshlets, uint32_t count, float* out8, uint32_t gridBlocks);

} // namespace nv

struct ProfRecord
{
	int slot;
	hipEvent_t begin, end;
};

// What the upload hooks build from the caller's scene buffers: the SoA mirrors and the Mesh-table registration.  One
// per context by default; nv_share_scene makes several contexts of one device (several streams, several views in
// flight) use ONE instance, so that three contexts do not hold three identical 480 MB mirrors.  Reference-counted.
struct nv_scene
{
	std::atomic<int> refs;
	int device;
	// SoA mirror of the meshlet cull bytes
	const NvMeshlet* mirroredFrom;
	uint32_t mirroredCount;
	uint2* soaBounds;
	uint32_t* soaCones;
	uint32_t* poolWords; // 2 x u32 (largest |centre component| / |radius| as fp16 bits) + 2 x float {3 Vmax, Rmax}: clustercull.hip pool_bounds_kernel
	uint32_t soaCapacity;
	// SoA mirror of the MeshDraw fields a draw decision reads (nv_upload_draws)
	const NvMeshDraw* drawsFrom;
	const NvMesh* drawsMeshes; // the Mesh table whose centres / radii are folded into the mirror
	uint32_t drawsCount;
	float4* soaWorld;
	uint2* soaScaleMesh;
	uint32_t* soaPostPass;
	uint32_t drawsCapacity;
	// Mesh table registered by nv_upload_meshes (pointer identity + count): lets drawcull stage it in LDS
	const NvMesh* meshesFrom;
	uint32_t meshCount;
};

struct nv_context
{
	int device;
	int numCUs;
	uint64_t* masks; // per-command ballots between the two clustercull launches
	nv::ClusterCounts* tileCounts;
	uint4* candList; // late pass with HiZ: the commands with survivors (cull kernel -> occlusion stage)
	uint32_t listStride; // room per sub-list (entries)
	uint32_t listSharers, listMinPer; // occlusion stage: blocks per sub-list, listed commands per block at least
	// drawcull: per-draw result bytes between its two launches, and its own per-tile counts
	uint8_t* drawResults;
	size_t drawResultsCapacity;
	nv::ClusterCounts* drawTileCounts;
	unsigned long long* totalsPartials; // nv_trianglecull / nv_cluster_expand: per-workgroup partial totals (3 x u64 x grid)
	nv_scene* scene; // mirrors + registrations (shared between contexts by nv_share_scene)
	// launch shape of the cull kernel (workgroups per CU) and the dealing's start-delay compensation in percent; constants
	// in the product, environment-tunable (with the NV_DEBUG_MODE bit mask) only in the NV_EXPERIMENTS build
	uint32_t debugMode;
	uint32_t ccBlocksPerCU;
	uint32_t dealScale;
	uint32_t scatterTilesPerCU;
	uint32_t scatterWaves; // NV_OPT_SCATTER_WAVES: waves per workgroup of the cluster scatter launch (16; 4 / 8)
	uint32_t scatterTilesAbs; // experiments: absolute number of scatter tiles (0 = per CU)
	uint32_t directPercent; // share of commands passing the filter above which the next launch skips the filter pass
	uint32_t bitsBlocksPerCU; // cluster_bits_kernel: grid (blocks per CU)
	uint32_t taskcullOneLaunch; // experiments: nv_taskcull's early pass as the one-command-per-wave kernel (the round-1 form)
	int forceDirect;        // NV_OPT_CULL_FORM: -1 = by the previous launch's statistic, 0 / 1 = always filter / always direct, 2 = direct and never the bit-expanding early form
	int forceVisFirst;      // NV_OPT_DRAW_RECORDS: -1 = by the statistic of the last TASK pass, 0 / 1 = records with / after the visibility words in drawcull's early pass
	int forceTaskList;      // NV_OPT_TASK_EMIT: -1 = by the statistic of earlier TASK passes, 0 / 1 = per-draw / list form of drawcull's TASK scatter
	int forceShallow;       // NV_OPT_CULL_RING: -1 = by the previous launch's command count, 0 / 1 = always the 8-deep / the 4-deep ring
	// command count of the previous clustercull launch, written by its kernel into mapped host memory (tuning hint)
	volatile uint32_t* hintHost;
	uint32_t* hintDevice;
	// the command buffer the last nv_drawcull(task) of this context wrote: a cluster pass over it runs on the commands of draws the draw-level cull
	// already found visible — nothing for the conservative filter to remove — so, while no launch has left a filter statistic yet (a fresh
	// context's first frames: the hint words lag the launches that fill them), such a pass takes the direct / lane forms instead of the filter form
	const void* taskCommandsFrom;
	uint32_t fusedReset;
	uint32_t fusedSubmit;
	uint64_t* countsSink; // nv_set_counts_sink
	// nv_profile_*: event pairs recorded on the launch stream, drained by nv_profile_read
	int profiling;
	std::vector<ProfRecord>* prof;
	float* timing; // NV_DEBUG_MODE bit 3: 8 x u64 stamps per wave of the last clustercull
	size_t timingWaves; // its room, in waves
	uint32_t variants[NV_VARIANT_SLOTS]; // nv_profile_variants: launches per kernel variant since the last read
};

namespace
{

struct DeviceGuard
{
	int prev;
	bool switched;
	explicit DeviceGuard(int dev) : prev(-1), switched(false)
	{
		if (hipGetDevice(&prev) == hipSuccess && prev != dev)
			switched = hipSetDevice(dev) == hipSuccess;
	}
	~DeviceGuard()
	{
		if (switched)
			(void)hipSetDevice(prev);
	}
};

uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// Library-owned device memory goes through these two.  Product: hipMalloc / hipFree.  Experiments build (tests with
// NV_LIBRARY_PATH = libniagara_vis_exp.so, tools/): every block sits between two 4 KiB canary zones (0xC5) that
// nv_debug_check_scratch compares, and starts out as 0xAB bytes instead of whatever the driver hands out — an out-of-range store of a
// kernel into the library's own scratch, or a kernel that reads a scratch word nobody wrote, shows up in ANY process instead of only
// in a long one whose allocations are recycled.
#ifdef NV_EXPERIMENTS
constexpr size_t NV_GUARD_BYTES = 4096;
struct GuardedBlock
{
	char* base;
	size_t bytes; // payload
};
std::vector<GuardedBlock>& guarded_blocks()
{
	static std::vector<GuardedBlock> v;
	return v;
}
std::atomic_flag g_guardLock = ATOMIC_FLAG_INIT;
struct GuardLock
{
	GuardLock() { while (g_guardLock.test_and_set(std::memory_order_acquire)) { } }
	~GuardLock() { g_guardLock.clear(std::memory_order_release); }
};
#endif

template <class T>
hipError_t scratch_alloc(T** out, size_t bytes)
{
#ifdef NV_EXPERIMENTS
	char* base = nullptr;
	const size_t padded = (bytes + 255) / 256 * 256;
	hipError_t e = hipMalloc(&base, padded + 2 * NV_GUARD_BYTES);
	if (e != hipSuccess)
		return e;
	(void)hipMemset(base, 0xC5, NV_GUARD_BYTES);
	(void)hipMemset(base + NV_GUARD_BYTES, 0xAB, padded);
	(void)hipMemset(base + NV_GUARD_BYTES + padded, 0xC5, NV_GUARD_BYTES);
	{
		GuardLock lock;
		guarded_blocks().push_back(GuardedBlock{ base, padded });
	}
	*out = reinterpret_cast<T*>(base + NV_GUARD_BYTES);
	return hipSuccess;
#else
	return hipMalloc(out, bytes);
#endif
}

void scratch_free(void* p)
{
	if (!p)
		return;
#ifdef NV_EXPERIMENTS
	char* base = static_cast<char*>(p) - NV_GUARD_BYTES;
	{
		GuardLock lock;
		std::vector<GuardedBlock>& v = guarded_blocks();
		for (size_t i = 0; i < v.size(); ++i)
			if (v[i].base == base)
			{
				v[i] = v.back();
				v.pop_back();
				break;
			}
	}
	(void)hipFree(base);
#else
	(void)hipFree(p);
#endif
}

// profiling: returns an event recorded on `stream` now, or nullptr when profiling is off
hipEvent_t prof_mark(nv_context* ctx, hipStream_t stream)
{
	if (!ctx->profiling)
		return nullptr;
	hipEvent_t e = nullptr;
	if (hipEventCreate(&e) != hipSuccess)
		return nullptr;
	(void)hipEventRecord(e, stream);
	return e;
}

void prof_push(nv_context* ctx, int slot, hipEvent_t b, hipEvent_t e)
{
	if (b && e && ctx->prof)
		ctx->prof->push_back(ProfRecord{ slot, b, e });
}

NvPyramidDesc null_pyramid()
{
	NvPyramidDesc p;
	memset(&p, 0, sizeof(p));
	return p;
}

// Per-draw result bytes between drawcull's two launches.  Sized by nv_create (1 M draws) and nv_reserve only: a pass entry
// point never allocates or synchronises (it would stall the device behind an "asynchronous enqueue" and break a stream
// capture); a pass over more draws than were reserved returns NV_ENOMEM.
int reserve_draw_results(nv_context* ctx, uint32_t drawCount)
{
	const size_t need = nv::drawcull_result_bytes(drawCount);
	if (need <= ctx->drawResultsCapacity)
		return NV_OK;
	hipError_t e = hipDeviceSynchronize(); // a drawcull still in flight on any stream may be reading the old scratch
	if (e != hipSuccess)
		return (int)e;
	if (ctx->drawResults)
		scratch_free(ctx->drawResults);
	ctx->drawResults = nullptr;
	ctx->drawResultsCapacity = 0;
	const size_t cap = need < (size_t(1) << 21) ? (size_t(1) << 21) : need;
	if (scratch_alloc(&ctx->drawResults, cap) != hipSuccess)
		return NV_ENOMEM;
	ctx->drawResultsCapacity = cap;
	return NV_OK;
}

nv_scene* scene_new(int device)
{
	nv_scene* sc = new (std::nothrow) nv_scene();
	if (!sc)
		return nullptr;
	sc->refs.store(1);
	sc->device = device;
	sc->mirroredFrom = nullptr;
	sc->mirroredCount = 0;
	sc->soaBounds = nullptr;
	sc->soaCones = nullptr;
	sc->soaCapacity = 0;
	sc->drawsFrom = nullptr;
	sc->drawsMeshes = nullptr;
	sc->drawsCount = 0;
	sc->soaWorld = nullptr;
	sc->soaScaleMesh = nullptr;
	sc->soaPostPass = nullptr;
	sc->drawsCapacity = 0;
	sc->meshesFrom = nullptr;
	sc->meshCount = 0;
	return sc;
}

// drops one reference; the last one frees the mirrors (the caller has made the owning device current)
void scene_release(nv_scene* sc)
{
	if (!sc || sc->refs.fetch_sub(1) != 1)
		return;
	if (sc->soaBounds)
		scratch_free(sc->soaBounds);
	if (sc->soaCones)
		scratch_free(sc->soaCones);
	if (sc->poolWords)
		scratch_free(sc->poolWords);
	if (sc->soaWorld)
		scratch_free(sc->soaWorld);
	if (sc->soaScaleMesh)
		scratch_free(sc->soaScaleMesh);
	if (sc->soaPostPass)
		scratch_free(sc->soaPostPass);
	delete sc;
}

// one scatter workgroup per CU, capped by the per-tile count table
uint32_t scatter_grid(const nv_context* ctx)
{
	uint32_t g = ctx->scatterTilesAbs ? ctx->scatterTilesAbs : (uint32_t)ctx->numCUs * ctx->scatterTilesPerCU;
	return g > nv::CC_MAX_SCATTER_TILES ? nv::CC_MAX_SCATTER_TILES : g;
}

// grid of the streaming kernels: blocksPerCU workgroups of 256 threads per CU
uint32_t persistent_grid(const nv_context* ctx, uint32_t blocksPerCU)
{
	return (uint32_t)ctx->numCUs * blocksPerCU;
}

// nv_profile_variants: what launch_cluster_bits / launch_cluster_mask resolve the host's choices to (clustercull.hip)
void count_cull_variant(nv_context* ctx, const nv::ClusterArgs& a, bool lanes, bool bits, bool late, bool shallow, bool direct)
{
	int v;
	if (lanes)
		v = bits ? NV_VARIANT_CULL_LANES_BITS : NV_VARIANT_CULL_LANES;
	else if (!a.soaBounds)
		v = NV_VARIANT_CULL_AOS;
	else if (direct && a.filterK > 0.0f)
		v = nv::clustercull_takes_packed(a, late, true, true) ? NV_VARIANT_CULL_DIRECT_PACKED : NV_VARIANT_CULL_DIRECT;
	else
		v = !late && shallow ? NV_VARIANT_CULL_FILTER_RING4 : NV_VARIANT_CULL_FILTER_RING8;
	ctx->variants[v] += 1u;
}

} // namespace

extern "C" {

const char* nv_version(void) { return "niagara_vis 0.4 (gfx950)"; } // 0.2: NV_PROF_SLOTS 4 -> 5; 0.3: nv_reserve, nv_share_scene; 0.4: NV_VARIANT_SLOTS 9 -> 10

int nv_create(nv_context** out_ctx, int device)
{
	if (!out_ctx)
		return NV_EINVAL;
	*out_ctx = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
		return NV_ENODEV;
	if (device < 0 || device >= count)
		return NV_EINVAL;
	DeviceGuard guard(device);

	nv_context* ctx = new (std::nothrow) nv_context();
	if (!ctx)
		return NV_ENOMEM;
	memset(ctx, 0, sizeof(*ctx));
	ctx->device = device;
	ctx->scene = scene_new(device);
	if (!ctx->scene)
	{
		delete ctx;
		return NV_ENOMEM;
	}

	hipDeviceProp_t prop;
	hipError_t e = hipGetDeviceProperties(&prop, device);
	if (e != hipSuccess)
	{
		scene_release(ctx->scene);
		delete ctx;
		return (int)e;
	}
	ctx->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	ctx->ccBlocksPerCU = 6;
	ctx->dealScale = 100;
	ctx->scatterTilesPerCU = 1;
	ctx->scatterWaves = 16;
	ctx->directPercent = 35; // measured crossover (config 3A geometry at several densities): ~36 % of the commands passing the filter
	ctx->forceDirect = -1;
	ctx->bitsBlocksPerCU = 4;
	ctx->taskcullOneLaunch = 0;
	ctx->forceShallow = -1;
	ctx->forceTaskList = -1;
	ctx->forceVisFirst = -1;
	ctx->listStride = nv::clustercull_list_stride();
	ctx->listSharers = 4;
	ctx->listMinPer = 8;
#ifdef NV_EXPERIMENTS
	if (const char* v = getenv("NV_DIRECT"))
		ctx->forceDirect = atoi(v);
	if (const char* v = getenv("NV_TASKCULL_ONE_LAUNCH"))
		ctx->taskcullOneLaunch = (uint32_t)atoi(v);
	if (const char* v = getenv("NV_BITS_BLOCKS"))
		ctx->bitsBlocksPerCU = (uint32_t)atoi(v);
	if (const char* v = getenv("NV_DIRECT_PERCENT"))
		ctx->directPercent = (uint32_t)atoi(v);
	if (const char* v = getenv("NV_SCATTER_TILES_PER_CU"))
		ctx->scatterTilesPerCU = (uint32_t)atoi(v) ? (uint32_t)atoi(v) : 1;
	if (const char* v = getenv("NV_SCATTER_TILES"))
		ctx->scatterTilesAbs = (uint32_t)atoi(v);
	// the experiments build only (tools/): the product library reads no environment variable
	if (const char* v = getenv("NV_DEBUG_MODE"))
		ctx->debugMode = (uint32_t)atoi(v);
	if (const char* v = getenv("NV_HIZ_LIST_STRIDE")) // a small value forces the occlusion stage's scan fallback (tests)
		ctx->listStride = (uint32_t)atoi(v) < nv::clustercull_list_stride() ? (uint32_t)atoi(v) : nv::clustercull_list_stride();
	if (const char* v = getenv("NV_HIZ_SHARERS"))
		ctx->listSharers = (uint32_t)atoi(v) ? (uint32_t)atoi(v) : 4;
	if (const char* v = getenv("NV_HIZ_MIN_PER"))
		ctx->listMinPer = (uint32_t)atoi(v) >= 4 && (uint32_t)atoi(v) <= 64 ? (uint32_t)atoi(v) : 8;
	if (const char* v = getenv("NV_DEAL_SCALE"))
		ctx->dealScale = (uint32_t)atoi(v);
	if (const char* v = getenv("NV_CC_BLOCKS_PER_CU"))
		ctx->ccBlocksPerCU = (uint32_t)atoi(v) ? ((uint32_t)atoi(v) > 8 ? 8u : (uint32_t)atoi(v)) : 6; // (the stamp buffer holds 8 per CU)
#endif

	if (scratch_alloc(&ctx->masks, nv::clustercull_mask_bytes()) != hipSuccess || scratch_alloc(&ctx->tileCounts, sizeof(nv::ClusterCounts)) != hipSuccess ||
	    scratch_alloc(&ctx->candList, nv::clustercull_list_bytes()) != hipSuccess ||
	    hipMemset(ctx->tileCounts, 0, sizeof(nv::ClusterCounts)) != hipSuccess ||
	    scratch_alloc(&ctx->drawTileCounts, sizeof(nv::ClusterCounts)) != hipSuccess ||
	    hipMemset(ctx->drawTileCounts, 0, sizeof(nv::ClusterCounts)) != hipSuccess || reserve_draw_results(ctx, 1u << 20) != NV_OK ||
	    scratch_alloc(&ctx->totalsPartials, (size_t)persistent_grid(ctx, 8) * 3 * sizeof(unsigned long long)) != hipSuccess)
	{
		nv_destroy(ctx);
		return NV_ENOMEM;
	}
	// optional: without it the kernels simply keep the deep ring
	{
		void* h = nullptr;
		if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess)
		{
			memset(h, 0, 64);
			void* d = nullptr;
			if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess)
			{
				ctx->hintHost = static_cast<volatile uint32_t*>(h);
				ctx->hintDevice = static_cast<uint32_t*>(d);
			}
			else
				(void)hipHostFree(h);
		}
	}

	*out_ctx = ctx;
	return NV_OK;
}

void nv_destroy(nv_context* ctx)
{
	if (!ctx)
		return;
	DeviceGuard guard(ctx->device);
	if (ctx->hintHost)
		(void)hipHostFree(const_cast<uint32_t*>(ctx->hintHost));
	if (ctx->drawResults)
		scratch_free(ctx->drawResults);
	if (ctx->drawTileCounts)
		scratch_free(ctx->drawTileCounts);
	if (ctx->totalsPartials)
		scratch_free(ctx->totalsPartials);
	if (ctx->masks)
		scratch_free(ctx->masks);
	if (ctx->tileCounts)
		scratch_free(ctx->tileCounts);
	if (ctx->candList)
		scratch_free(ctx->candList);
	scene_release(ctx->scene);
	if (ctx->timing)
		scratch_free(ctx->timing);
	delete ctx->prof;
	delete ctx;
}

int nv_status(nv_context* ctx, void* stream)
{
	if (!ctx)
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	hipError_t e = hipStreamSynchronize((hipStream_t)stream);
	if (e != hipSuccess)
		return (int)e;
	return NV_OK;
}

int nv_set_option(nv_context* ctx, int option, int value)
{
	if (!ctx)
		return NV_EINVAL;
	switch (option)
	{
	case NV_OPT_FUSED_COUNT_RESET:
		ctx->fusedReset = value ? 1u : 0u;
		return NV_OK;
	case NV_OPT_FUSED_SUBMIT:
		ctx->fusedSubmit = value ? 1u : 0u;
		return NV_OK;
	case NV_OPT_SCATTER_WAVES:
		if (value != 4 && value != 8 && value != 16)
			return NV_EINVAL;
		ctx->scatterWaves = (uint32_t)value;
		return NV_OK;
	case NV_OPT_CULL_FORM:
		if (value < 0 || value > 5)
			return NV_EINVAL;
		ctx->forceDirect = value - 1; // 0 -> -1 (by statistic), 1 -> 0 (filter form), 2 -> 1 (direct form), 3 -> 2 (direct, one command per wave also with visibility bits), 4 -> 3 (as 3, and no packed walk), 5 -> 4 (direct, the packed walk also in the early pass with visibility bits)
		return NV_OK;
	case NV_OPT_TASK_EMIT:
		if (value < 0 || value > 2)
			return NV_EINVAL;
		ctx->forceTaskList = value - 1;
		return NV_OK;
	case NV_OPT_DRAW_RECORDS:
		if (value < 0 || value > 2)
			return NV_EINVAL;
		ctx->forceVisFirst = value - 1;
		return NV_OK;
	case NV_OPT_CULL_RING:
		if (value != 0 && value != 4 && value != 8)
			return NV_EINVAL;
		ctx->forceShallow = value == 0 ? -1 : (value == 4 ? 1 : 0);
		return NV_OK;
	case NV_OPT_CULL_WORKGROUPS_PER_CU:
		if (value < 1 || value > 8)
			return NV_EINVAL;
		ctx->ccBlocksPerCU = (uint32_t)value;
		return NV_OK;
	default:
		return NV_EINVAL;
	}
}

int nv_reserve(nv_context* ctx, uint32_t maxDraws, uint32_t maxCommands)
{
	if (!ctx)
		return NV_EINVAL;
	(void)maxCommands; // the per-command scratch (ballots, survivor lists) is sized for NV_TASK_WGLIMIT by nv_create: the passes never process more
	DeviceGuard guard(ctx->device);
	return reserve_draw_results(ctx, maxDraws);
}

int nv_share_scene(nv_context* dst, nv_context* src)
{
	if (!dst || !src || dst->device != src->device)
		return NV_EINVAL;
	if (dst->scene == src->scene)
		return NV_OK;
	DeviceGuard guard(dst->device);
	// passes of dst still in flight read dst's current mirrors
	hipError_t e = hipDeviceSynchronize();
	if (e != hipSuccess)
		return (int)e;
	src->scene->refs.fetch_add(1);
	scene_release(dst->scene);
	dst->scene = src->scene;
	return NV_OK;
}

int nv_profile_enable(nv_context* ctx, int enabled)
{
	if (!ctx)
		return NV_EINVAL;
	if (!ctx->prof)
		ctx->prof = new (std::nothrow) std::vector<ProfRecord>();
	if (!ctx->prof)
		return NV_ENOMEM;
	ctx->profiling = enabled ? 1 : 0;
	return NV_OK;
}

int nv_profile_read(nv_context* ctx, float out_ms[NV_PROF_SLOTS], uint32_t out_count[NV_PROF_SLOTS])
{
	if (!ctx || !out_ms || !out_count)
		return NV_EINVAL;
	for (int i = 0; i < NV_PROF_SLOTS; ++i)
	{
		out_ms[i] = 0.0f;
		out_count[i] = 0;
	}
	if (!ctx->prof)
		return NV_OK;
	DeviceGuard guard(ctx->device);
	int rc = NV_OK;
	for (ProfRecord& r : *ctx->prof)
	{
		float ms = 0.0f;
		hipError_t e = hipEventSynchronize(r.end);
		if (e == hipSuccess)
			e = hipEventElapsedTime(&ms, r.begin, r.end);
		if (e == hipSuccess && r.slot >= 0 && r.slot < NV_PROF_SLOTS)
		{
			out_ms[r.slot] += ms;
			out_count[r.slot] += 1;
		}
		else if (e != hipSuccess)
			rc = (int)e;
	}
	// events are shared between adjacent records (end of one = begin of the next): destroy each once
	std::vector<hipEvent_t> seen;
	for (ProfRecord& r : *ctx->prof)
		for (hipEvent_t ev : { r.begin, r.end })
		{
			bool dup = false;
			for (hipEvent_t s : seen)
				dup = dup || s == ev;
			if (!dup)
			{
				seen.push_back(ev);
				(void)hipEventDestroy(ev);
			}
		}
	ctx->prof->clear();
	return rc;
}

int nv_profile_variants(nv_context* ctx, uint32_t out_count[NV_VARIANT_SLOTS])
{
	if (!ctx || !out_count)
		return NV_EINVAL;
	for (int i = 0; i < NV_VARIANT_SLOTS; ++i)
	{
		out_count[i] = ctx->variants[i];
		ctx->variants[i] = 0;
	}
	return NV_OK;
}

int nv_upload_meshlets(nv_context* ctx, void* stream, const NvMeshlet* d_meshlets, uint32_t meshletCount)
{
	if (!ctx || (!d_meshlets && meshletCount))
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	// one extra 64-entry block so that a command's 64-lane window never leaves the mirror
	uint32_t padded = round_up(meshletCount, 64) + 64;
	if (padded > ctx->scene->soaCapacity)
	{
		// a pass of this or of a sharing context may still be reading the old mirror on another stream
		hipError_t e = hipDeviceSynchronize();
		if (e != hipSuccess)
			return (int)e;
		if (ctx->scene->soaBounds)
			scratch_free(ctx->scene->soaBounds);
		if (ctx->scene->soaCones)
			scratch_free(ctx->scene->soaCones);
		ctx->scene->soaBounds = nullptr;
		ctx->scene->soaCones = nullptr;
		ctx->scene->soaCapacity = 0;
		ctx->scene->mirroredFrom = nullptr;
		if (scratch_alloc(&ctx->scene->soaBounds, (size_t)padded * sizeof(uint2)) != hipSuccess ||
		    scratch_alloc(&ctx->scene->soaCones, (size_t)padded * sizeof(uint32_t)) != hipSuccess)
			return NV_ENOMEM;
		ctx->scene->soaCapacity = padded;
	}
	if (!ctx->scene->poolWords && scratch_alloc(&ctx->scene->poolWords, 4 * sizeof(uint32_t)) != hipSuccess)
	{
		ctx->scene->mirroredFrom = nullptr; // (no pool bounds: no SoA pass may name the old mirror either)
		return NV_ENOMEM;
	}
	int rc = nv::launch_soa_split((hipStream_t)stream, d_meshlets, meshletCount, padded, ctx->scene->soaBounds, ctx->scene->soaCones, ctx->scene->poolWords);
	if (rc)
		return rc;
	ctx->scene->mirroredFrom = d_meshlets;
	ctx->scene->mirroredCount = meshletCount;
	return NV_OK;
}

int nv_upload_meshes(nv_context* ctx, void* stream, const NvMesh* d_meshes, uint32_t meshCount)
{
	(void)stream;
	if (!ctx || (!d_meshes && meshCount))
		return NV_EINVAL;
	ctx->scene->meshesFrom = d_meshes;
	ctx->scene->meshCount = meshCount;
	return NV_OK;
}

int nv_upload_draws(nv_context* ctx, void* stream, const NvMeshDraw* d_draws, uint32_t drawCount, const NvMesh* d_meshes)
{
	if (!ctx || (!d_draws && drawCount) || (d_draws && !d_meshes))
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	if (!d_draws)
	{
		ctx->scene->drawsFrom = nullptr;
		ctx->scene->drawsMeshes = nullptr;
		ctx->scene->drawsCount = 0;
		return NV_OK;
	}
	if (drawCount > ctx->scene->drawsCapacity)
	{
		//  a drawcull of this or of a sharing context may still be reading the old mirror on another stream
		hipError_t e = hipDeviceSynchronize();
		if (e != hipSuccess)
			return (int)e;
		if (ctx->scene->soaWorld)
			scratch_free(ctx->scene->soaWorld);
		if (ctx->scene->soaScaleMesh)
			scratch_free(ctx->scene->soaScaleMesh);
		if (ctx->scene->soaPostPass)
			scratch_free(ctx->scene->soaPostPass);
		ctx->scene->soaWorld = nullptr;
		ctx->scene->soaScaleMesh = nullptr;
		ctx->scene->soaPostPass = nullptr;
		ctx->scene->drawsCapacity = 0;
		ctx->scene->drawsFrom = nullptr;
		const size_t cap = (size_t)drawCount + 64;
		if (scratch_alloc(&ctx->scene->soaWorld, cap * sizeof(float4)) != hipSuccess || scratch_alloc(&ctx->scene->soaScaleMesh, cap * sizeof(uint2)) != hipSuccess ||
		    scratch_alloc(&ctx->scene->soaPostPass, cap * sizeof(uint32_t)) != hipSuccess)
			return NV_ENOMEM;
		ctx->scene->drawsCapacity = drawCount;
	}
	int rc = nv::launch_draw_split((hipStream_t)stream, d_draws, d_meshes, ctx->scene->meshesFrom == d_meshes ? ctx->scene->meshCount : 0u, 0, drawCount, ctx->scene->soaWorld, ctx->scene->soaScaleMesh, ctx->scene->soaPostPass);
	if (rc)
		return rc;
	ctx->scene->drawsFrom = d_draws;
	ctx->scene->drawsMeshes = d_meshes;
	ctx->scene->drawsCount = drawCount;
	return NV_OK;
}

int nv_update_draws(nv_context* ctx, void* stream, const NvMeshDraw* d_draws, uint32_t first, uint32_t count)
{
	if (!ctx || !d_draws)
		return NV_EINVAL;
	if (!ctx->scene->drawsFrom || d_draws < ctx->scene->drawsFrom || d_draws >= ctx->scene->drawsFrom + ctx->scene->drawsCount || ctx->scene->drawsFrom + (d_draws - ctx->scene->drawsFrom) != d_draws)
		return NV_OK; // nothing registered for this buffer: the passes read it in place anyway
	const size_t base = (size_t)(d_draws - ctx->scene->drawsFrom) + first;
	if (base > ctx->scene->drawsCount || count > ctx->scene->drawsCount - base)
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	return nv::launch_draw_split((hipStream_t)stream, ctx->scene->drawsFrom, ctx->scene->drawsMeshes,
	                             ctx->scene->meshesFrom == ctx->scene->drawsMeshes ? ctx->scene->meshCount : 0u, (uint32_t)base, count, ctx->scene->soaWorld, ctx->scene->soaScaleMesh, ctx->scene->soaPostPass);
}

int nv_drawcull(nv_context* ctx, void* stream, const NvCullData* cull, int late, int task, const NvMeshDraw* d_draws,
                const NvMesh* d_meshes, void* d_commands, uint32_t* d_count4, uint32_t* d_drawVisibility,
                const NvPyramidDesc* pyramid)
{
	if (!ctx || !cull || !d_count4 || !d_drawVisibility || (cull->drawCount && (!d_draws || !d_meshes || !d_commands)))
		return NV_EINVAL;
	if (late && cull->occlusionEnabled == 1 && !(pyramid && pyramid->d_base))
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);

	if (nv::drawcull_result_bytes(cull->drawCount) > ctx->drawResultsCapacity)
		return NV_ENOMEM; // nv_reserve(ctx, maxDraws, ...) first: a pass never allocates
	int rc = NV_OK;

	nv::DrawArgs a;
	a.cd = *cull;
	a.pyr = pyramid ? *pyramid : null_pyramid();
	a.draws = d_draws;
	// the mirror serves the registered buffer and any sub-range of it that starts on a record (a pass over a shard of the draws)
	const size_t drawOffset = ctx->scene->drawsFrom && d_draws >= ctx->scene->drawsFrom ? (size_t)(d_draws - ctx->scene->drawsFrom) : ~size_t(0);
	const bool mirrored = ctx->scene->soaWorld && drawOffset != ~size_t(0) && ctx->scene->drawsFrom + drawOffset == d_draws &&
	                      drawOffset + cull->drawCount <= ctx->scene->drawsCount && ctx->scene->drawsMeshes == d_meshes; // (the mirror folds THAT table's bounds in)
	a.soaWorld = mirrored ? ctx->scene->soaWorld + drawOffset : nullptr;
	a.soaScaleMesh = mirrored ? ctx->scene->soaScaleMesh + drawOffset : nullptr;
	a.soaPostPass = mirrored ? ctx->scene->soaPostPass + drawOffset : nullptr;
	a.meshes = d_meshes;
	a.commands = d_commands;
	a.count4 = d_count4;
	a.dvb = d_drawVisibility;
	a.results = ctx->drawResults;
	a.tileCounts = ctx->drawTileCounts;
	a.scatterTiles = scatter_grid(ctx);
#ifdef NV_EXPERIMENTS
	a.debugMode = ctx->debugMode;
#endif
	a.fusedReset = ctx->fusedReset;
	a.fusedSubmit = ctx->fusedSubmit;
	a.meshCount = ctx->scene->meshesFrom == d_meshes ? ctx->scene->meshCount : 0u;
	a.hostHint = ctx->hintDevice;
	// TASK scatter form: the list form (one lane per output command) when an earlier TASK pass emitted more than 4 commands per
	// emitting draw, the per-draw form otherwise (and until a pass has been seen); NV_OPT_TASK_EMIT pins it
	a.taskList = ctx->forceTaskList >= 0 ? (uint32_t)ctx->forceTaskList : (ctx->hintHost && ctx->hintHost[3] > 4u * ctx->hintHost[2] ? 1u : 0u);
	// early pass: visibility words first, and only the records of last frame's visible draws, when the last TASK pass emitted from fewer than one draw in eight
	a.visFirst = late ? 0u : (ctx->forceVisFirst >= 0 ? (uint32_t)ctx->forceVisFirst : (ctx->hintHost && ctx->hintHost[2] != 0u && (uint64_t)ctx->hintHost[2] * 8u < cull->drawCount ? 1u : 0u));
	if (task)
	{
		ctx->variants[a.taskList ? NV_VARIANT_TASK_LIST : NV_VARIANT_TASK_PER_DRAW] += 1u;
		ctx->taskCommandsFrom = d_commands;
	}
	hipEvent_t e0 = prof_mark(ctx, (hipStream_t)stream);
	rc = nv::launch_drawcull((hipStream_t)stream, a, late, task);
	prof_push(ctx, NV_PROF_DRAWCULL, e0, prof_mark(ctx, (hipStream_t)stream));
	return rc;
}

int nv_reset_count(nv_context* ctx, void* stream, uint32_t* d_count4a, uint32_t* d_count4b)
{
	if (!ctx || (!d_count4a && !d_count4b))
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	return nv::launch_reset_count((hipStream_t)stream, d_count4a, d_count4b);
}

int nv_tasksubmit(nv_context* ctx, void* stream, uint32_t* d_count4, NvMeshTaskCommand* d_commands)
{
	if (!ctx || !d_count4 || !d_commands)
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	return nv::launch_tasksubmit((hipStream_t)stream, d_count4, d_commands);
}

static int fill_cluster_args(nv_context* ctx, nv::ClusterArgs& a, const NvCullData* cull, int late,
                             const NvMeshTaskCommand* d_commands, const uint32_t* d_count4, const NvMeshDraw* d_draws,
                             const NvMeshlet* d_meshlets, uint32_t* d_meshletVisibility, const NvPyramidDesc* pyramid)
{
	if (!ctx || !cull || !d_commands || !d_draws || !d_meshlets)
		return NV_EINVAL;
	if (cull->clusterOcclusionEnabled == 1 && !d_meshletVisibility)
		return NV_EINVAL;
	if (late && cull->clusterOcclusionEnabled == 1 && !(pyramid && pyramid->d_base))
		return NV_EINVAL;
	memset(&a, 0, sizeof(a));
	a.cd = *cull;
	a.pyr = pyramid ? *pyramid : null_pyramid();
	a.commands = d_commands;
	a.count4 = d_count4;
	a.draws = d_draws;
	a.meshlets = d_meshlets;
	const bool soa = ctx->scene->mirroredFrom == d_meshlets && ctx->scene->soaBounds && ctx->scene->poolWords; // (poolWords: the SoA kernels read the pool's bounds unconditionally — ADVICE r5)
	a.soaBounds = soa ? ctx->scene->soaBounds : nullptr;
	a.soaCones = soa ? ctx->scene->soaCones : nullptr;
	a.poolBounds = soa ? reinterpret_cast<const float*>(ctx->scene->poolWords + 2) : nullptr;
	a.mvb = d_meshletVisibility;
	a.masks = ctx->masks;
	a.candList = ctx->candList;
	a.listStride = ctx->listStride;
	a.listMinPer = ctx->listMinPer;
	a.tileCounts = ctx->tileCounts;
	a.scatterTiles = scatter_grid(ctx);
	a.generations = ctx->ccBlocksPerCU;
	a.dealScale = ctx->dealScale;
	{
		// divisions by launch constants, for the kernels' prologues (args.h, host.cpp nv_division_magic): exact for n < 2^39 / d — the
		// largest numerator is 64 x the chunk count < 2^26, d <= 8192
		const uint32_t cullGrid = persistent_grid(ctx, ctx->ccBlocksPerCU);
		a.cullWavesMagic = nv_division_magic(cullGrid * 4u);
		a.genBlocks = cullGrid / 6u ? cullGrid / 6u : 1u;
		a.genBlocksMagic = nv_division_magic(a.genBlocks);
		a.tilesMagic = nv_division_magic(a.scatterTiles);
	}
	// Margin scale of the conservative filter / certified test and the view-only terms of its per-draw derivation: filtermath.h
	// (filter_k: 4 K u S with K = 48, u = 2^-24, S = max(1, |f0| + |f1|, |f2| + |f3|) — the error analysis is written for unit-length plane
	// coefficients; other finite coefficients scale the margins, non-finite or absurd ones, or near / far planes that are not finite,
	// switch both off and every command runs the reference arithmetic.  fp32, no contraction: this file is compiled with
	// -ffp-contract=off like the kernels).
	a.filterK = nv::filter_k(cull->frustum, cull->znear, cull->zfar);
	nv::filter_view_norms(cull->view, &a.viewRowNorm, &a.viewTransNorm, &a.viewSum);
	a.hostHint = ctx->hintDevice;
	return NV_OK;
}

int nv_clustercull(nv_context* ctx, void* stream, const NvCullData* cull, int late, const NvMeshTaskCommand* d_commands,
                   const uint32_t* d_count4, const NvMeshDraw* d_draws, const NvMeshlet* d_meshlets,
                   uint32_t* d_meshletVisibility, const NvPyramidDesc* pyramid, uint32_t* d_clusterIndices,
                   uint32_t* d_clusterCount4)
{
	if (!d_count4 || !d_clusterIndices || !d_clusterCount4)
		return NV_EINVAL;
	nv::ClusterArgs a;
	int rc = fill_cluster_args(ctx, a, cull, late, d_commands, d_count4, d_draws, d_meshlets, d_meshletVisibility, pyramid);
	if (rc)
		return rc;
	DeviceGuard guard(ctx->device);
	a.clusterIndices = d_clusterIndices;
	a.clusterCount4 = d_clusterCount4;
	a.fusedReset = ctx->fusedReset;
	a.fusedSubmit = ctx->fusedSubmit;
	a.packDirect = ctx->forceDirect != 3 ? 1u : 0u; // NV_OPT_CULL_FORM 4: one command per wave iteration also where the packed walk applies
	a.packBits = ctx->forceDirect == 4 ? 1u : 0u;   // NV_OPT_CULL_FORM 5: the early pass with visibility bits as a packed walk too
	a.countsSink = reinterpret_cast<unsigned long long*>(ctx->countsSink);
#ifdef NV_EXPERIMENTS
	a.debugMode = ctx->debugMode;
	if (ctx->debugMode & (8u | 268435456u)) // bit 3: the cull launch's wave stamps; bit 28: the occlusion stage's phase sums
	{
		// room for the cull launch at any NV_OPT_CULL_WORKGROUPS_PER_CU (4 waves per workgroup) AND for the occlusion stage's grid (bit 28:
		// listSharers x CC_LISTS blocks of 4 waves — more than the cull launch's on a small part or with NV_HIZ_SHARERS raised)
		const size_t timingWaves = std::max((size_t)persistent_grid(ctx, 8) * 4, (size_t)ctx->listSharers * nv::CC_LISTS * 4);
		if (ctx->timing && ctx->timingWaves < timingWaves)
		{
			scratch_free(ctx->timing);
			ctx->timing = nullptr;
		}
		if (!ctx->timing && scratch_alloc(&ctx->timing, timingWaves * 8 * sizeof(unsigned long long)) == hipSuccess)
			ctx->timingWaves = timingWaves;
		a.probeOut = ctx->timing;
	}
#endif
	// two pure maps: the cull kernel also accumulates survivors per scatter tile, so no workgroup waits on another
	hipStream_t s = (hipStream_t)stream;
	hipEvent_t e0 = prof_mark(ctx, s);
	bool shallow = ctx->hintHost && nv::clustercull_prefers_shallow(*ctx->hintHost) && !(ctx->debugMode & 65536u); // bit 16 (experiments): always deep
	if (ctx->forceShallow >= 0)
		shallow = ctx->forceShallow != 0;
	// mapped host words the previous launches left: [0] command count, [1] commands their filter did not (or would not
	// have) finished — possibly a launch or two behind, which only matters for speed
	bool direct = ctx->hintHost && nv::clustercull_prefers_direct(ctx->hintHost[0], ctx->hintHost[1], ctx->directPercent);
	if (!(ctx->hintHost && ctx->hintHost[0] != 0) && d_commands == ctx->taskCommandsFrom) // no statistic yet: by where the commands come from
		direct = true;
	// (the provenance is good for ONE cluster launch: a later list at the same address — a freed and reused buffer, a caller-built list — is not this
	// context's drawcull output unless another nv_drawcull(task) has written there since)
	const bool ownTaskCommands = d_commands == ctx->taskCommandsFrom;
	ctx->taskCommandsFrom = nullptr;
	// A pass of partial commands — what drawcull's LOD select leaves — takes the direct form's packed walk whatever the filter statistic says, where that form
	// exists (early form without visibility bits, over the mirror): its fill is estimated from the emitting draws and commands the task pass that wrote the
	// list left in hint words 2 and 3 (clustercull.hip clustercull_prefers_packed); a caller's own list has no such words and is taken as full.
	const bool bits = cull->clusterOcclusionEnabled == 1 && cull->postPass == 0;
	const bool twoStage = late && cull->clusterOcclusionEnabled == 1;
	if (!direct && ownTaskCommands && ctx->hintHost && a.soaBounds && a.filterK > 0.0f && (twoStage || (!late && !bits)))
	{
		const bool poolInCache = (uint64_t)ctx->scene->mirroredCount * 12u <= (48ull << 20);
		direct = nv::clustercull_prefers_packed(ctx->hintHost[3], ctx->hintHost[2], poolInCache ? 85u : 60u);
	}
	if (ctx->forceDirect >= 0)
		direct = ctx->forceDirect != 0;
	// Late pass with HiZ = three launches: the cull kernel in its early form (frustum + cone ballots), the occlusion probe
	// with one lane per survivor (clustercull.hip cluster_hiz_kernel: visibility bits, skip, tile counts), the scatter.
	a.deferHiz = twoStage ? 1u : 0u;
	// Where the filter would not pay (direct), an EARLY pass with visibility bits tests one LANE per cluster that can be visible at all — per set bit —
	// instead of one wave per command (clustercull.hip cluster_bits_kernel: 26 against 38-40 us at frame scale).  Without bits the direct form walks
	// packed windows of 64 valid meshlets (cluster_mask_kernel PACK, round 6: the cluster pass behind drawcull's LOD select 24 us against 34 for the
	// lane-per-valid-cluster form rounds 4-5 chose for a cache-resident pool, and 44 for one command per wave iteration); so does the late pass's first
	// stage.  NV_OPT_CULL_FORM 3 keeps one wave per command with visibility bits, 4 one command per wave iteration throughout.
	const bool laneForm = !late && direct && ctx->forceDirect < 2 && bits;
	if (laneForm)
		rc = nv::launch_cluster_bits(s, a, a.soaBounds != nullptr, persistent_grid(ctx, ctx->bitsBlocksPerCU));
	else
		rc = nv::launch_cluster_mask(s, a, twoStage ? 0 : late, a.soaBounds != nullptr, persistent_grid(ctx, ctx->ccBlocksPerCU), shallow, direct,
		                             a.commandCountOverride ? a.commandCountOverride : (ctx->hintHost ? ctx->hintHost[0] : 0u));
	count_cull_variant(ctx, a, laneForm, bits, late && !twoStage, shallow, direct); // (the two-stage late pass launches the early form)
	ctx->variants[NV_VARIANT_HIZ_STAGE] += twoStage ? 1u : 0u;
	hipEvent_t e1 = prof_mark(ctx, s);
	hipEvent_t eh = e1;
	if (rc == 0 && twoStage)
	{
		const uint32_t blocks = ctx->listSharers * nv::CC_LISTS;
		rc = nv::launch_cluster_hiz(s, a, a.soaBounds != nullptr, blocks);
		eh = prof_mark(ctx, s);
	}
	if (rc == 0 && !(ctx->debugMode & 16u)) // bit 4 (experiments): ballots only
		rc = nv::launch_cluster_scatter(s, a, a.scatterTiles, ctx->scatterWaves);
	hipEvent_t e2 = prof_mark(ctx, s);
	prof_push(ctx, NV_PROF_CLUSTER_CULL, e0, e1);
	if (twoStage)
		prof_push(ctx, NV_PROF_CLUSTER_HIZ, e1 ? e1 : nullptr, eh);
	prof_push(ctx, NV_PROF_CLUSTER_SCATTER, eh ? eh : nullptr, e2);
	return rc;
}

int nv_clustersubmit(nv_context* ctx, void* stream, uint32_t* d_clusterCount4, uint32_t* d_clusterIndices)
{
	if (!ctx || !d_clusterCount4 || !d_clusterIndices)
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	return nv::launch_clustersubmit((hipStream_t)stream, d_clusterCount4, d_clusterIndices);
}

int nv_taskcull(nv_context* ctx, void* stream, const NvCullData* cull, int late, const NvMeshTaskCommand* d_commands,
                const uint32_t* d_count4, const NvMeshDraw* d_draws, const NvMeshlet* d_meshlets,
                uint32_t* d_meshletVisibility, const NvPyramidDesc* pyramid, uint32_t* d_payloads, uint32_t* d_payloadCounts)
{
	if (!d_count4 || !d_payloads || !d_payloadCounts)
		return NV_EINVAL;
	nv::ClusterArgs a;
	int rc = fill_cluster_args(ctx, a, cull, late, d_commands, d_count4, d_draws, d_meshlets, d_meshletVisibility, pyramid);
	if (rc)
		return rc;
	DeviceGuard guard(ctx->device);
	a.clusterIndices = d_payloads;
	a.payloadCounts = d_payloadCounts;
	if (!late && !ctx->taskcullOneLaunch)
	{
		// Early pass: nv_clustercull's cull launch (filter / direct / lane-per-set-bit form by the same statistics), which writes the
		// payloads itself when ClusterArgs::payloadCounts is set (clustercull.hip: the end of a segment) — the filter form streams a
		// sparse pass at 8 bytes per meshlet and finishes most commands in 23 instructions, where the one-command-per-wave kernel
		// reads 12 and runs the reference's ~190 for every command (10 M meshlets: 38 us).  The cull kernel only snapshots the count
		// word when the reset is not fused; there is no count word here.
		hipStream_t s = (hipStream_t)stream;
		a.fusedReset = 1u;
		a.clusterCount4 = nullptr;
		// mapped hint words: [4] = the command count of the previous nv_taskcull (ring depth); [0], [1] = command count and filter
		// statistic of the previous nv_clustercull, a consistent pair (the payload form writes neither: no scatter launch follows it
		// that would publish its statistic).  A context that only ever calls nv_taskcull stays on the filter form.
		bool shallow = ctx->hintHost && nv::clustercull_prefers_shallow(ctx->hintHost[4]);
		if (ctx->forceShallow >= 0)
			shallow = ctx->forceShallow != 0;
		bool direct = ctx->hintHost && nv::clustercull_prefers_direct(ctx->hintHost[0], ctx->hintHost[1], ctx->directPercent);
		if (!(ctx->hintHost && ctx->hintHost[0] != 0) && d_commands == ctx->taskCommandsFrom) // (as in nv_clustercull)
			direct = true;
		ctx->taskCommandsFrom = nullptr;
		if (ctx->forceDirect >= 0)
			direct = ctx->forceDirect != 0;
		const bool bitsForm = direct && ctx->forceDirect < 2 && cull->clusterOcclusionEnabled == 1 && cull->postPass == 0;
		if (bitsForm)
			rc = nv::launch_cluster_bits(s, a, a.soaBounds != nullptr, persistent_grid(ctx, ctx->bitsBlocksPerCU));
		else
			rc = nv::launch_cluster_mask(s, a, 0, a.soaBounds != nullptr, persistent_grid(ctx, ctx->ccBlocksPerCU), shallow, direct,
			                             a.commandCountOverride ? a.commandCountOverride : (ctx->hintHost ? ctx->hintHost[4] : 0u));
		count_cull_variant(ctx, a, bitsForm, cull->clusterOcclusionEnabled == 1 && cull->postPass == 0, false, shallow, direct);
		return rc;
	}
	return nv::launch_taskcull((hipStream_t)stream, a, late, a.soaBounds != nullptr, (uint32_t)ctx->numCUs * 8);
}

int nv_cluster_expand(nv_context* ctx, void* stream, const NvMeshTaskCommand* d_commands, const NvMeshlet* d_meshlets,
                      const uint32_t* d_clusterIndices, const uint32_t* d_clusterCount4, NvClusterRecord* d_records,
                      uint32_t recordCapacity, uint64_t* d_totals3)
{
	if (!ctx || !d_commands || !d_meshlets || !d_clusterIndices || !d_clusterCount4 || !d_totals3 || (!d_records && recordCapacity))
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	const uint32_t grid = persistent_grid(ctx, 8);
	return nv::launch_cluster_expand((hipStream_t)stream, d_commands, d_meshlets, d_clusterIndices, d_clusterCount4, d_records, recordCapacity,
	                                 d_totals3, ctx->totalsPartials, grid);
}

int nv_trianglecull(nv_context* ctx, void* stream, const NvGlobals* globals, const NvMeshTaskCommand* d_commands, const NvMeshDraw* d_draws,
                    const NvMeshlet* d_meshlets, const uint32_t* d_meshletData, const NvVertex* d_vertices, const uint32_t* d_clusterIndices,
                    const uint32_t* d_clusterCount4, NvTriangleMask* d_masks, uint32_t maskCapacity, uint64_t* d_totals3)
{
	if (!ctx || !globals || !d_commands || !d_draws || !d_meshlets || !d_meshletData || !d_vertices || !d_clusterIndices || !d_clusterCount4 ||
	    !d_totals3 || (!d_masks && maskCapacity))
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	nv::TriangleArgs a;
	a.globals = *globals;
	a.commands = d_commands;
	a.draws = d_draws;
	a.meshlets = d_meshlets;
	a.meshletData = d_meshletData;
	a.vertices = d_vertices;
	a.clusterIndices = d_clusterIndices;
	a.cc4 = d_clusterCount4;
	a.masks = d_masks;
	a.capacity = maskCapacity;
	a.totals = reinterpret_cast<unsigned long long*>(d_totals3);
	const uint32_t grid = persistent_grid(ctx, 8);
	a.partials = ctx->totalsPartials;
	return nv::launch_trianglecull((hipStream_t)stream, a, grid);
}

int nv_meshlet_bounds(nv_context* ctx, void* stream, const NvVertex* d_vertices, const uint32_t* d_meshletData, NvMeshlet* d_meshlets, uint32_t meshletCount,
                      float* d_bounds8)
{
	if (!ctx || (meshletCount && (!d_vertices || !d_meshletData || !d_meshlets)))
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	const uint32_t blocks = (meshletCount + 3) / 4;
	const uint32_t cap = persistent_grid(ctx, 16);
	int rc = nv::launch_meshlet_bounds((hipStream_t)stream, d_vertices, d_meshletData, d_meshlets, meshletCount, d_bounds8, blocks < cap ? blocks : cap);
	// a mirror built from these records is stale now (its registration is by pointer, its contents a snapshot)
	if (rc == 0 && ctx->scene->mirroredFrom == d_meshlets)
		ctx->scene->mirroredFrom = nullptr;
	return rc;
}

int nv_depthreduce(nv_context* ctx, void* stream, const float* d_depth, uint32_t width, uint32_t height,
                   const NvPyramidDesc* pyramid)
{
	if (!ctx || !d_depth || !pyramid || !pyramid->d_base || !width || !height || !pyramid->levels || pyramid->levels > NV_MAX_MIPS)
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	hipEvent_t e0 = prof_mark(ctx, (hipStream_t)stream);
	int rc = nv::launch_depthreduce((hipStream_t)stream, d_depth, width, height, *pyramid);
	prof_push(ctx, NV_PROF_DEPTHREDUCE, e0, prof_mark(ctx, (hipStream_t)stream));
	return rc;
}

#ifdef NV_EXPERIMENTS
// experiments build only: number of library-owned blocks (of every context of this process) whose canary zones no longer hold
// their pattern, after synchronising the device; tests/conftest.py calls it after every GPU test
int nv_debug_check_scratch(void)
{
	if (hipDeviceSynchronize() != hipSuccess)
		return -1;
	GuardLock lock;
	int damaged = 0;
	std::vector<unsigned char> host(2 * NV_GUARD_BYTES);
	for (const GuardedBlock& b : guarded_blocks())
	{
		if (hipMemcpy(host.data(), b.base, NV_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess ||
		    hipMemcpy(host.data() + NV_GUARD_BYTES, b.base + NV_GUARD_BYTES + b.bytes, NV_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess)
			return -1;
		bool ok = true;
		for (unsigned char c : host)
			ok = ok && c == 0xC5;
		damaged += ok ? 0 : 1;
	}
	return damaged;
}
#endif

#ifdef NV_EXPERIMENTS
// experiments build only (not part of the public header): copies the NV_DEBUG_MODE bit-3 wave stamps to the host
int nv_debug_read_timing(nv_context* ctx, unsigned long long* out, uint32_t maxWaves)
{
	if (!ctx || !ctx->timing || !out)
		return NV_EINVAL;
	uint32_t waves = (uint32_t)ctx->timingWaves; // (the caller says how many it wants: the cull launch's, or the occlusion stage's)
	if (waves > maxWaves)
		waves = maxWaves;
	return (int)hipMemcpy(out, ctx->timing, (size_t)waves * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif

int nv_set_counts_sink(nv_context* ctx, uint64_t* d_out3)
{
	if (!ctx)
		return NV_EINVAL;
	ctx->countsSink = d_out3;
	return NV_OK;
}

int nv_pack_counts(nv_context* ctx, void* stream, const uint32_t* d_countA, const uint32_t* d_countB,
                   const uint32_t* d_countC, uint64_t* d_out3)
{
	if (!ctx || !d_out3)
		return NV_EINVAL;
	DeviceGuard guard(ctx->device);
	return nv::launch_pack_counts((hipStream_t)stream, d_countA, d_countB, d_countC, d_out3);
}

int nv_probe_cluster_scalars(nv_context* ctx, void* stream, const NvCullData* cull, const NvMeshTaskCommand* d_commands,
                             uint32_t commandCount, const NvMeshDraw* d_draws, const NvMeshlet* d_meshlets,
                             const NvPyramidDesc* pyramid, float* d_out16)
{
	if (!d_out16 || !commandCount)
		return NV_EINVAL;
	nv::ClusterArgs a;
	int rc = fill_cluster_args(ctx, a, cull, 0, d_commands, nullptr, d_draws, d_meshlets, nullptr, pyramid);
	if (rc == NV_EINVAL && cull && cull->clusterOcclusionEnabled == 1)
	{
		// the probe never touches visibility bits
		NvCullData c2 = *cull;
		c2.clusterOcclusionEnabled = 0;
		rc = fill_cluster_args(ctx, a, &c2, 0, d_commands, nullptr, d_draws, d_meshlets, nullptr, pyramid);
		if (rc == NV_OK)
			a.cd = *cull;
	}
	if (rc)
		return rc;
	DeviceGuard guard(ctx->device);
	a.commandCountOverride = commandCount;
	a.probeOut = d_out16;
	uint32_t blocks = (commandCount + 3) / 4;
	uint32_t cap = (uint32_t)ctx->numCUs * 8;
	return nv::launch_probe((hipStream_t)stream, a, a.soaBounds != nullptr, blocks < cap ? blocks : cap);
}

} // extern "C"
