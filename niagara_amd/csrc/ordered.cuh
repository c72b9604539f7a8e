// ordered.cuh — deterministic (invocation-ordered) append for gfx950: the scheme shared by clustercull.hip and drawcull.hip.
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
// The per-tile counts live in two banks (ClusterCounts, args.cuh): a pass adds into counts[parity] and the scatter
// kernel clears counts[parity ^ 1] and flips the parity, so nothing is memset between passes and a captured hipGraph
// replays correctly.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nv
{

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_xor(v, o, 64);
	return v;
}

} // namespace nv
