// dealing.h — the static work assignment of cluster_mask_kernel (clustercull.hip) as a PLAN: everything about it that depends only on
// the pass's command count and the launch shape, in one function that compiles for the device and for the host (round 5).
//
// Why a plan: the dealing was ~150 scalar instructions on every wave's start-up path, between the count word's arrival and the wave's first
// command load — identical in all 6144 waves, and an instruction there costs the launch what 1 / 25 of an instruction per command costs in
// the stream (tools/experiments/filler_sensitivity.sh: +200 scalar instructions per wave = +1.0 us).  The host cannot
// know the indirect command count, but it knows the PREVIOUS launch's (the kernel leaves it in a mapped host word) and frames repeat: it derives the
// plan for that count and passes it with the kernel arguments; a wave uses it iff the count word it reads equals DealPlan::cmds (and
// the flags match what the kernel instance wants), and derives the plan itself with this same function otherwise.  Results never depend on the
// guess; a wrong one costs what every launch cost before.
//
// The weighted dealing itself (why later generations take fewer rounds) is described above deal_wave in clustercull.hip.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define NV_DP __host__ __device__ __forceinline__
#else
#define NV_DP static inline
#endif

namespace nv
{

struct DealPlan
{
	uint32_t cmds;          // the command count the plan was derived for
	uint32_t flags;         // DEAL_WEIGHTED_WANTED | chunk << 8: what the kernel instance asks for (a plan for another shape is not used)
	uint32_t weighted;      // 1: the weighted rounds apply (rounds / weightedTotal / restPerWave / restRem), 0: plain round-robin (perWaveChunks / evenRem)
	uint32_t rounds[6];     // weighted rounds per generation
	uint32_t weightedTotal; // chunks dealt in the weighted rounds
	uint32_t restPerWave, restRem; // what is left, dealt evenly: every wave restPerWave chunks, the first restRem waves one more
	uint32_t perWaveChunks, evenRem; // plain round-robin: likewise
	uint32_t tileCmds;      // commands per scatter tile (scatter_tile_commands)
	uint32_t tileMul31;     // a command's tile = mulhi((index >> 8) << 1, tileMul31): tileCmds is a multiple of 256, so index / tileCmds = (index >> 8) / d with
	                        // d = tileCmds >> 8, and m = floor(2^31 / d) + 1 gives floor(n / d) = floor(n m / 2^31) for n < 2^31 / d (n < 2^24 here, d <= 2^16)
	uint32_t numTiles;      // scatter tiles that hold commands: ceil(cmds / tileCmds)
	uint32_t waves;         // W = waves of the launch (the grid is a scalar load away — the dispatch packet — for a wave that has the plan in its arguments)
};

constexpr uint32_t DEAL_WEIGHTED_WANTED = 1u;
constexpr uint32_t DEAL_PACKED_TABLE = 1000u; // deal_plan's scalePercent from this value up: the packed walk's delay table
constexpr uint32_t DEAL_TILE_THREADS = 256u; // = CC_THREADS: a scatter tile is a multiple of the cull workgroup's commands per step

// q = n / d through the host's magic (host.cpp nv_division_magic: exact for n < 2^39 / d, 0 = not available)
NV_DP uint32_t deal_div(uint32_t n, uint32_t d, uint32_t magic)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return magic ? __umulhi(n, magic) >> 7 : n / d;
#else
	return magic ? (uint32_t)(((uint64_t)n * magic) >> 39) : n / d;
#endif
}

NV_DP uint32_t deal_magic(uint32_t d) // host.cpp nv_division_magic
{
	return d >= 256u && d <= 8192u ? (uint32_t)((1ull << 39) / d) + 1u : 0u;
}

NV_DP uint32_t deal_tile_commands(uint32_t numCmds, uint32_t tiles, uint32_t tilesMagic)
{
	const uint32_t T = (deal_div(numCmds + tiles - 1u, tiles, tilesMagic) + DEAL_TILE_THREADS - 1u) / DEAL_TILE_THREADS * DEAL_TILE_THREADS;
	return T ? T : DEAL_TILE_THREADS;
}

// a command's scatter tile (DealPlan::tileMul31)
NV_DP uint32_t deal_tile_of(uint32_t index, uint32_t tileMul31)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __umulhi(index >> 8 << 1, tileMul31);
#else
	return (uint32_t)(((uint64_t)(index >> 8 << 1) * tileMul31) >> 32);
#endif
}

// W = waves of the launch (grid x 4), genBlocks = workgroups per generation, gridBlocks = the grid; scalePercent = ClusterArgs::dealScale.
NV_DP DealPlan deal_plan(uint32_t numCmds, uint32_t chunk, bool weightedWanted, uint32_t W, uint32_t wavesMagic, uint32_t generations, uint32_t genBlocks,
                         uint32_t gridBlocks, uint32_t scalePercent, uint32_t tiles, uint32_t tilesMagic)
{
	DealPlan p;
	p.cmds = numCmds;
	p.flags = (weightedWanted ? DEAL_WEIGHTED_WANTED : 0u) | chunk << 8;
	p.weighted = 0;
	p.waves = W;
	const uint32_t numChunks = (numCmds + chunk - 1u) / chunk;
	p.perWaveChunks = deal_div(numChunks, W, wavesMagic);
	p.evenRem = numChunks - p.perWaveChunks * W;
	p.tileCmds = deal_tile_commands(numCmds, tiles, tilesMagic);
	p.tileMul31 = 0x80000000u / (p.tileCmds >> 8) + 1u; // (tileCmds >= 256)
	p.numTiles = (numCmds + p.tileCmds - 1u) / p.tileCmds; // <= tiles
	p.weightedTotal = p.restPerWave = p.restRem = 0;
#pragma unroll
	for (int k = 0; k < 6; ++k)
		p.rounds[k] = 0;
	if (!weightedWanted || generations != 6u || genBlocks * 6u != gridBlocks || p.perWaveChunks < 4u || p.perWaveChunks >= 60u)
		return p;
	const uint32_t genWaves = genBlocks * 4u;
	// delays in 1/16 command: { 0, 0.5, 2.4, 4.2, 7.1, 13.1 }, mean 4.55 — calibrated on the filter form (round 3).
	// scalePercent >= DEAL_PACKED_TABLE: the packed walk's table at (scalePercent - DEAL_PACKED_TABLE) % (round 6: a wave of the walk lives 18-23 us and is bound
	// by vector issue — the younger waves of a SIMD fall behind by more than their start delay.  2.5 x the filter form's delays balanced generations 0-4
	// (ends 17.7-18.5 us) and starved the last one (12.4 us): its entry is 1.8 x instead — tools/wave_timeline.py with NV_DIRECT=1)
	const bool packedTable = scalePercent >= DEAL_PACKED_TABLE;
	if (packedTable)
		scalePercent -= DEAL_PACKED_TABLE;
	const int delayFilter16[6] = { 0, 8, 38, 67, 114, 210 }, delayPacked16[6] = { 0, 20, 95, 168, 285, 380 };
	const int* delay16 = packedTable ? delayPacked16 : delayFilter16;
	const int mean16 = packedTable ? 158 : 73;
	// The dividend through the magic must stay below 2^39 / W (deal_div).  HERE perWaveChunks < 60, so numChunks < 61 W and the dividend is below
	// 61 W x 16 chunk = 3904 W for the chunk of 4 both passes use; 3904 W < 2^39 / W for W < 11 866 — and deal_magic offers a magic up to W = 8192 only
	// (tests/test_dealing.py holds host plan == wave-derived plan at that largest grid, at the largest counts that reach this line).
	const int perWave16 = (int)deal_div(numChunks * (chunk * 16u), W, wavesMagic);
	uint32_t rounds[6], weightedTotal = 0;
#pragma unroll
	for (int k = 0; k < 6; ++k)
	{
		// (the nominal scale is a constant per generation; any other one divides)
		const int adjust16 = scalePercent == 100u ? mean16 - delay16[k] : ((int)scalePercent * (mean16 - delay16[k])) / 100;
		const int target16 = perWave16 + adjust16 - (int)(chunk * 16u); // keep one even round for the remainder
		rounds[k] = target16 > 0 ? (uint32_t)target16 / (chunk * 16u) : 0u;
		weightedTotal += rounds[k] * genWaves;
	}
	if (weightedTotal > numChunks) // cannot happen (floors of targets that sum to less than the total); stay safe
		return p;
	const uint32_t rest = numChunks - weightedTotal;
	const uint32_t restPerWave = deal_div(rest, W, wavesMagic);
	// a wave's chunk table is one VGPR, lane j = its j-th chunk: 64 chunks; decided for the whole grid at once (every wave must take the same branch)
	if (rounds[0] + restPerWave + 1u > 64u)
		return p;
	p.weighted = 1;
#pragma unroll
	for (int k = 0; k < 6; ++k)
		p.rounds[k] = rounds[k];
	p.weightedTotal = weightedTotal;
	p.restPerWave = restPerWave;
	p.restRem = rest - restPerWave * W;
	return p;
}

// ---- the per-wave part (clustercull.hip deal_wave; host-compilable so that tests/test_dealing.py can hold the whole dealing — every chunk to
// exactly one wave — on the CPU).  Wave w of generation gen (workgroup index / genBlocks; genWaves = genBlocks x 4):
// its number of chunks, and the chunk its j-th table entry names (j < 64 when the plan is weighted; plain round-robin otherwise).
NV_DP uint32_t deal_wave_rounds(const DealPlan& p, uint32_t gen)
{
	const uint32_t g = gen < 6u ? gen : 5u;
	return p.rounds[0] * (g == 0) + p.rounds[1] * (g == 1) + p.rounds[2] * (g == 2) + p.rounds[3] * (g == 3) + p.rounds[4] * (g == 4) + p.rounds[5] * (g == 5);
}

NV_DP uint32_t deal_wave_chunks(const DealPlan& p, uint32_t w, uint32_t gen)
{
	return p.weighted ? deal_wave_rounds(p, gen) + p.restPerWave + (w < p.restRem ? 1u : 0u) : p.perWaveChunks + (w < p.evenRem ? 1u : 0u);
}

// entry j of wave w's chunk table: round j of the weighted part (chunks of earlier rounds = genWaves x the sum over generations of min(j, rounds)),
// then the even remainder
NV_DP uint32_t deal_wave_entry(const DealPlan& p, uint32_t w, uint32_t rounds, uint32_t genWaves, uint32_t j)
{
	if (!p.weighted)
		return j * p.waves + w;
	uint32_t before = 0;
#pragma unroll
	for (int k = 0; k < 6; ++k)
		before += j < p.rounds[k] ? j : p.rounds[k];
	return j < rounds ? before * genWaves + w : p.weightedTotal + w + (j - rounds) * p.waves;
}

} // namespace nv
