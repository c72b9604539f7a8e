// args.h — kernel argument blocks shared by the launchers (context.hip) and the kernels, and the ordered-append scheme
// both cull passes use.
//
// Ordered append — deterministic (invocation-ordered) append for gfx950: the scheme shared by clustercull.hip and drawcull.hip.
//
// niagara appends survivors with one global atomicAdd per invocation (drawcull.comp.glsl:123,143;
// clustercull.comp.glsl:135 — "TODO: potentially slow global atomic"), so its output order is whatever the hardware
// serialises.  Here the append index of an item is the exclusive prefix sum of the emit counts of all items before it
// in invocation order: one valid serialisation of the reference's atomics, and bit-reproducible.
//
// Both passes compute it with TWO launches on the stream and no inter-workgroup wait at all:
//   1. the cull / decide kernel is a pure map.  It writes a compact per-item result (a 64-bit ballot per task command,
//      a byte per draw) and adds each wave's emit count to the count of the scatter tile the wave's items fall in
//      (one fire-and-forget atomicAdd per wave that emits anything; tiles = contiguous item ranges, one per CU);
//   2. the scatter kernel runs one workgroup per tile.  Its append base = count word + counts of the tiles before it
//      (<= 512 values, one load per lane), then one scan over the tile's results and the ordered stores.
// Measured on MI355X this beat both alternatives that keep a single launch: ticket-ordered tiles (one hot atomic word
// serialises at ~10 ns per returning atomic) and a chained decoupled look-back over co-resident tiles (needs a grid
// that is co-resident by construction, 2-4 dependent look-back rounds of ~1.5 us each, and bounded spins).  The launch
// boundary costs ~2 us and buys: no co-residency requirement, no spinning, nothing to re-arm after a fault.
//
// The per-tile counts live in two banks (ClusterCounts below): a pass adds into counts[parity] and the scatter
// kernel clears counts[parity ^ 1] and flips the parity, so nothing is memset between passes and a captured hipGraph
// replays correctly.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/niagara_vis.h"
#include "dealing.h"

// Tuning experiments (wave stamps, partial passes, alternative dealing ...) exist only in a separate build
// (-DNV_EXPERIMENTS -> libniagara_vis_exp.so, tools/): the product library carries no switch that could change a result
// or cost an instruction in a hot loop.
#ifdef NV_EXPERIMENTS
#define NV_DBG(args, bits) (((args).debugMode & (bits)) != 0)
#else
#define NV_DBG(args, bits) false
#endif

namespace nv
{

// Wave-wide inclusive prefix sum in six DPP adds (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast 15 and 31 across
// them: the sequence LLVM's own atomic optimizer emits for gfx9 wave64).  hipcc lowers __shfl_up / __shfl_xor to ds_bpermute_b32 — an
// LDS round trip of ~100 cycles per step, six steps per scan, on the dependent chain of every scatter launch and of every iteration of the
// lane kernels (round 4).  A lane whose DPP source is outside its row, masked by row_mask or inactive keeps `old` = 0, i.e. adds
// nothing.  Every call site runs with all 64 lanes active (a scan over a partial wave would also have been wrong with the shuffles).
//  row_bcast:15 / row_bcast:31 / wave_shr:1 exist on GFX9-family wave64 parts only (gfx950 is one): another ARCH must not compile
// these silently into something else.  And the helpers need ALL 64 lanes active (wave_sum_u32 reads lane 63; an inactive DPP source adds
// nothing): the experiments build traps on a partial exec mask.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "args.h: the DPP scans are written for GFX9-family wave64 targets (gfx950)"
#endif
#ifdef NV_EXPERIMENTS
#define NV_ASSERT_FULL_EXEC() do { if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap(); } while (0)
#else
#define NV_ASSERT_FULL_EXEC() do { } while (0)
#endif

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_pull_u32(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_scan_inclusive_u32(uint32_t v)
{
	NV_ASSERT_FULL_EXEC();
	v += dpp_pull_u32<0x111, 0xf>(v); // row_shr:1
	v += dpp_pull_u32<0x112, 0xf>(v); // row_shr:2
	v += dpp_pull_u32<0x114, 0xf>(v); // row_shr:4
	v += dpp_pull_u32<0x118, 0xf>(v); // row_shr:8
	v += dpp_pull_u32<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
	v += dpp_pull_u32<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3
	return v;
}

// lane permutations that stay inside the VALU (DPP) where hipcc's __shfl* would go through LDS (ds_bpermute_b32)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_move_u32(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_shift_up1_u32(uint32_t v) { return dpp_move_u32<0x138>(v); } // lane l <- lane l - 1 (wave_shr:1; lane 0 keeps its own)
__device__ __forceinline__ uint32_t quad_first_u32(uint32_t v) { return dpp_move_u32<0x00>(v); }      // quad_perm:[0,0,0,0]
__device__ __forceinline__ uint32_t quad_xor1_u32(uint32_t v) { return dpp_move_u32<0xB1>(v); }       // quad_perm:[1,0,3,2]
__device__ __forceinline__ uint32_t quad_xor2_u32(uint32_t v) { return dpp_move_u32<0x4E>(v); }       // quad_perm:[2,3,0,1]

// the sum over the wave, in every lane (wave-uniform: lane 63 of the scan)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_inclusive_u32(v), 63);
}

// Per-tile survivor counts handed from the cull kernel to the scatter kernel (clustercull.hip).  Two banks: a pass adds
// into counts[parity] and clears counts[parity ^ 1] for the next one, so nothing is memset between passes and a
// captured hipGraph replays correctly.
// Each counter sits in its own 64-byte line: atomics to one line serialise in its L2 channel (measured: 14 k adds into
// 8 lines cost 13 us, 78 k into 16 lines 20 us), spread over hundreds of lines they are free.
constexpr uint32_t CC_MAX_SCATTER_TILES = 512;
constexpr uint32_t CC_COUNT_STRIDE = 16; // words between two tile counters
constexpr uint32_t CC_LISTS = 256;       // sub-lists of the late pass's survivor-command list (cluster_hiz_kernel)
struct ClusterCounts
{
	uint32_t parity;   // read by the cull kernel; flipped by one thread of the scatter kernel
	uint32_t k2parity; // the parity of the running pass, written by the cull kernel for the scatter kernel
	uint32_t base;     // the count word as the pass found it (0 with the fused reset), snapshot by the cull kernel: the scatter
	                   // kernel's last tile overwrites the live word while other tiles may not have started yet
	uint32_t pad[29];
	uint32_t listOverflow[2]; // late pass with HiZ, banked like `counts`: a sub-list of ClusterArgs::candList ran out of room
	uint32_t pad2[30];
	uint32_t listCount[2][CC_LISTS * CC_COUNT_STRIDE]; // entries per sub-list, one counter per 64-byte line
	uint32_t counts[2][CC_MAX_SCATTER_TILES * CC_COUNT_STRIDE];
};

// Wave-uniform words every wave of a launch needs (counts, parity): read through the constant address space, i.e. with a
// scalar load served by the scalar cache of the CU — as vector (or agent-scope atomic) loads they are thousands of
// requests for ONE line of one L2 channel.  Only for words written by an EARLIER launch (the scalar cache is invalidated
// per dispatch, not within one).
typedef __attribute__((address_space(4))) const uint32_t* nv_uniform_u32p;
__device__ __forceinline__ uint32_t load_uniform_u32(const uint32_t* p)
{
	return *(nv_uniform_u32p)(uintptr_t)p;
}

struct ClusterArgs
{
	NvCullData cd;
	NvPyramidDesc pyr;
	const NvMeshTaskCommand* __restrict__ commands;
	const uint32_t* __restrict__ count4; // dccb: {commandCount, groupCountX, 64, 1}
	const NvMeshDraw* __restrict__ draws;
	const NvMeshlet* __restrict__ meshlets; // AoS (used when soaBounds == nullptr)
	const uint2* __restrict__ soaBounds;
	const uint32_t* __restrict__ soaCones;
	const float* __restrict__ poolBounds; // {3 x the largest |centre component|, the largest |radius|} of the mirrored pool (nv_upload_meshlets); set whenever soaBounds is
	uint32_t* __restrict__ mvb;
	uint32_t* __restrict__ clusterIndices;
	uint32_t* __restrict__ clusterCount4;
	uint32_t* __restrict__ payloadCounts; // taskcull only
	uint64_t* __restrict__ masks; // scratch: one 64-bit ballot per task command
	uint4* __restrict__ candList;    // scratch (late pass with HiZ): the commands that have frustum / cone survivors, two uint4 each, CC_LISTS sub-lists in no particular order
	uint32_t listStride;             // entries of room per sub-list
	uint32_t listMinPer;             // listed commands per block of the occlusion stage, at least (>= 4)
	ClusterCounts* __restrict__ tileCounts;
	uint32_t scatterTiles; // grid of the scatter kernel (<= CC_MAX_SCATTER_TILES)
	uint32_t generations;  // workgroups of the cull kernel per CU (its grid = generations x CUs)
	uint32_t dealScale;    // percent of the nominal start-delay compensation of the dealing (tuning; 100)
	// Divisions by launch constants, prepared by the host (context.hip magic_for): q = mulhi(n, magic) >> 7, exact for n < 2^39 / d;
	// 0 = not available (d < 256 or > 8192: the kernels divide).  The cull launch's prologue is on every wave's start-up path and
	// an instruction there costs what one per command costs in the filter loop (filler_sensitivity.sh).
	uint32_t cullWavesMagic;  // d = waves of the cull launch (grid x 4)
	uint32_t genBlocks, genBlocksMagic; // d = workgroups per generation of the cull launch (grid / generations)
	uint32_t tilesMagic;      // d = scatterTiles
	DealPlan plan;            // the cull launch's dealing for the command count the host expects (dealing.h); used iff the count word agrees
	float filterK;         // 4 K u S of the conservative filter / certified test (clustercull.hip make_filter); 0 = both off
	// what make_filter derives from the view matrix alone, done once on the host in the same fp32 operations (context.hip view_norms):
	// every wave computed these ~40 instructions in its prologue, and an instruction there costs what 1 / 25 per command costs (§4.1)
	float viewRowNorm;     // max over rows r of |V(r,0)| + |V(r,1)| + |V(r,2)|
	float viewTransNorm;   // max over rows r of |V(r,3)|
	float viewSum;         // the sum of the twelve entries, in make_filter's order: 0 x it is 0, or NaN for a non-finite view
	uint32_t* hostHint;    // mapped host word: the cull kernel leaves its command count here for the next launch's tuning
	uint32_t commandCountOverride; // probe/taskcull: explicit command count (0 = use count4[1]*64)
	float* __restrict__ probeOut;
#ifdef NV_EXPERIMENTS
	uint32_t debugMode; // NV_DEBUG_MODE of the experiments build
#endif
	uint32_t fusedReset; // NV_OPT_FUSED_COUNT_RESET
	uint32_t fusedSubmit; // NV_OPT_FUSED_SUBMIT
	uint32_t deferHiz;   // late pass with HiZ: the cull kernel leaves frustum / cone ballots and no tile counts, cluster_hiz_kernel finishes
	uint32_t packBits;   // the early pass WITH visibility bits as a packed walk too (entries = the valid meshlets of the commands that have a set bit) instead of one lane per set bit / one wave per command
	uint32_t packDirect; // the direct form walks windows of 64 valid meshlets (clustercull.hip PACK) where it applies; 0: one command per wave iteration (NV_OPT_CULL_FORM 4)
	unsigned long long* countsSink; // nv_set_counts_sink (nullptr: off)
};

struct DrawArgs
{
	NvCullData cd;
	NvPyramidDesc pyr;
	const NvMeshDraw* draws;
	// mirror of the decision's inputs (nv_upload_draws; nullptr: read the AoS records)
	const float4* __restrict__ soaWorld;      // world-space sphere: rotateQuat(center, q) * scale + position, radius * scale
	const uint2* __restrict__ soaScaleMesh;   // scale (fp32 bits), meshIndex
	const uint32_t* __restrict__ soaPostPass; // postPass
	const NvMesh* meshes;
	void* commands;
	uint32_t* count4;
	uint32_t* dvb;
	uint8_t* results;          // scratch: one result byte per draw, decide kernel -> scatter kernel
	ClusterCounts* tileCounts; // per-scatter-tile command counts (own instance, same two-bank scheme as clustercull)
	uint32_t scatterTiles;     // grid of the scatter kernel (<= CC_MAX_SCATTER_TILES)
	uint32_t fusedReset; // NV_OPT_FUSED_COUNT_RESET
	uint32_t fusedSubmit; // NV_OPT_FUSED_SUBMIT
	uint32_t meshCount;  // > 0 when nv_upload_meshes registered `meshes`: the table may be staged in LDS
	uint32_t* hostHint;  // mapped host words (context.hip): [2], [3] = emitting draws / commands of the last TASK pass, written by its scatter launch
	uint32_t taskList;   // TASK scatter in the list form (one lane per output command) instead of the per-draw form
	// TASK passes in the list form: the decide launch leaves one 16-byte record per EMITTING draw — {LOD's meshletOffset, meshletCount, meshletVisibilityOffset, draw | previous
	// visibility << 31} — compacted per wave at the front of the wave's own draw range, and the number of them per wave; the scatter launch reads those instead of the result bytes,
	// the draw records and the Mesh table.  tileDraws is then a whole number of waves' ranges.
	uint4* records;
	uint32_t* recordCounts;
	uint32_t recordsOn;
	uint32_t tileDraws;    // draws per scatter tile (scatter_tile_draws(), rounded up to whole waves of the decide launch when records are on)
	uint32_t unitsPerWave; // 64-draw units per wave of the decide launch
	uint32_t visFirst;   // early pass: request the visibility words ahead of the records and only the records of last frame's visible draws (a hint: results do not depend on it)
#ifdef NV_EXPERIMENTS
	uint32_t debugMode; // NV_DEBUG_MODE of the experiments build
#endif
};

// trianglecull.hip
struct TriangleArgs
{
	NvGlobals globals;
	const NvMeshTaskCommand* __restrict__ commands;
	const NvMeshDraw* __restrict__ draws;
	const NvMeshlet* __restrict__ meshlets;
	const uint32_t* __restrict__ meshletData;
	const NvVertex* __restrict__ vertices;
	const uint32_t* __restrict__ clusterIndices;
	const uint32_t* __restrict__ cc4;
	NvTriangleMask* __restrict__ masks;
	uint32_t capacity;
	unsigned long long* __restrict__ totals;
	unsigned long long* __restrict__ partials; // library scratch: 3 counters per workgroup of the launch
};

} // namespace nv
