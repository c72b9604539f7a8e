// dealing_shim.cpp — host-side window onto niagara_amd/csrc/dealing.h for tests/test_dealing.py: the SAME source the cull kernel compiles (the plan,
// the per-wave chunk table, the tile of a command), built with g++ at test time.  Test infrastructure.
#include <cstdint>
#include <cstring>

#include "../niagara_amd/csrc/dealing.h"

extern "C" {

// plan as 18 words in DealPlan's field order
void shim_deal_plan(uint32_t numCmds, uint32_t chunk, int weightedWanted, uint32_t gridBlocks, uint32_t generations, uint32_t scalePercent, uint32_t tiles, int useMagic,
                    uint32_t* out)
{
	const uint32_t W = gridBlocks * 4u, genBlocks = gridBlocks / 6u ? gridBlocks / 6u : 1u;
	const nv::DealPlan p = nv::deal_plan(numCmds, chunk, weightedWanted != 0, W, useMagic ? nv::deal_magic(W) : 0u, generations, genBlocks, gridBlocks, scalePercent, tiles,
	                                     useMagic ? nv::deal_magic(tiles) : 0u);
	static_assert(sizeof(p) == 18 * sizeof(uint32_t), "DealPlan");
	memcpy(out, &p, sizeof(p));
}

// every wave's chunks, as the kernel walks them: owner[c] = wave that takes chunk c (~0: nobody), returns the number of chunks handed out twice;
// perWave[w] = chunks of wave w.  Waves are numbered generation-major (workgroup index = w / 4, generation = workgroup / genBlocks).
uint32_t shim_deal_all(const uint32_t* plan18, uint32_t gridBlocks, uint32_t numChunks, uint32_t* owner, uint32_t* perWave)
{
	nv::DealPlan p;
	memcpy(&p, plan18, sizeof(p));
	const uint32_t genBlocks = gridBlocks / 6u ? gridBlocks / 6u : 1u, genWaves = genBlocks * 4u;
	uint32_t bad = 0;
	for (uint32_t c = 0; c < numChunks; ++c)
		owner[c] = ~0u;
	for (uint32_t w = 0; w < gridBlocks * 4u; ++w)
	{
		const uint32_t gen = (w / 4u) / genBlocks;
		const uint32_t n = nv::deal_wave_chunks(p, w, gen), rounds = nv::deal_wave_rounds(p, gen);
		perWave[w] = n;
		for (uint32_t j = 0; j < n; ++j)
		{
			const uint32_t c = nv::deal_wave_entry(p, w, rounds, genWaves, j);
			if (c >= numChunks)
				continue; // (the last round may run past the pass's end: the kernel guards by index)
			if (owner[c] != ~0u)
				++bad;
			owner[c] = w;
		}
	}
	return bad;
}

uint32_t shim_tile_of(uint32_t index, uint32_t tileMul31) { return nv::deal_tile_of(index, tileMul31); }
}
