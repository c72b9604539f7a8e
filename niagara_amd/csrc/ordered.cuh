// ordered.cuh — deterministic (invocation-ordered) append for gfx950.
//
// niagara appends survivors with one global atomicAdd per invocation (drawcull.comp.glsl:123,143;
// clustercull.comp.glsl:135 — "TODO: potentially slow global atomic"), so its output order is whatever the
// hardware serialises.  Here the append index of an item is the exclusive prefix sum of the emit counts of all
// items before it in invocation order, computed in ONE pass with a chained scan across workgroups
// (decoupled look-back).  That is one valid serialisation of the reference's atomics, it is bit-reproducible,
// and it replaces 1 atomic per survivor with 2 eight-byte publishes per TILE (one tile per workgroup per pass).
//
// MI355X specifics (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   * tiles are assigned statically (tile = blockIdx.x + round * gridDim.x) and a tile waits only on lower tiles.
//     Measured on MI355X, ticket-ordered tiles cost more than the cull itself (one hot atomic word serialises at
//     ~10 ns per returning atomic even when sharded over 32 lines), so the kernels instead launch a grid that is
//     co-resident by construction — context.hip launches 4 workgroups of 256 threads per CU, 16 of the CU's 32 wave
//     slots, with <= 32 KiB LDS each — which is the condition under which a statically ordered chain always makes
//     progress.  If other work shares the GPU the unscheduled workgroups start as soon as it drains;
//   * per-XCD L2s are not coherent: every shared word is an 8-byte {epoch, status, value} granule written by one
//     agent-scope relaxed atomic store and polled with agent-scope relaxed loads (the "data is the flag" form,
//     no fences needed because no other payload is handed over);
//   * state is self-cleaning and replay-safe: granules carry the launch epoch, which lives in device memory and
//     is advanced by the owner of the last tile, so nothing has to be memset between launches and a captured
//     hipGraph replays correctly;
//   * every spin is bounded; a timeout sets ctl->error (reported by nv_status) instead of hanging the GPU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define NV_SPIN_LIMIT (1u << 22)

namespace nv
{

struct OrderCtl
{
	uint32_t epoch; // >= 1; granules of other epochs read as "invalid"
	uint32_t error; // sticky: 1 = look-back spin bound hit
	uint32_t pad[30];
};

enum : uint32_t
{
	ST_INVALID = 0,
	ST_AGGREGATE = 1,
	ST_PREFIX = 2
};

__device__ __forceinline__ uint64_t pack_state(uint32_t epoch, uint32_t status, uint32_t value)
{
	return ((uint64_t)((epoch << 2) | status) << 32) | value;
}

__device__ __forceinline__ uint32_t load_epoch(const OrderCtl* ctl)
{
	return __hip_atomic_load(&ctl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_xor(v, o, 64);
	return v;
}

// Called by ALL threads of a 256-thread workgroup (4 waves) with workgroup-uniform arguments; `scratch` is 16 words
// of LDS.  Publishes this tile's aggregate, looks back for the exclusive prefix over a 256-tile window per round
// (wave w inspects tiles tile-1-64w-lane), publishes the inclusive prefix and returns the exclusive prefix to every
// thread.  With one tile per workgroup and ~1024 tiles all finishing together, the chain resolves in <= 4 rounds.
// base0 = value of the count word before the pass (what the first atomicAdd of the reference would return).
__device__ __forceinline__ uint32_t lookback_exclusive(uint64_t* __restrict__ state, OrderCtl* __restrict__ ctl, uint32_t tile,
                                                       uint32_t epoch, uint32_t aggregate, uint32_t base0, uint32_t* scratch)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;

	if (tile == 0)
	{
		if (threadIdx.x == 0)
			__hip_atomic_store(&state[0], pack_state(epoch, ST_PREFIX, base0 + aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return base0;
	}

	if (threadIdx.x == 0)
		__hip_atomic_store(&state[tile], pack_state(epoch, ST_AGGREGATE, aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

	enum : uint32_t
	{
		W_CONTINUE = 0, // 64 aggregates, no prefix: add them and keep looking further back
		W_DONE = 1,     // a prefix was found with nothing unpublished before it
		W_BLOCKED = 2   // an unpublished tile stands before the first prefix (or there is no prefix): poll again
	};

	uint32_t exclusive = 0;
	int64_t look = (int64_t)tile - 1; // nearest predecessor = lane 0 of wave 0
	uint32_t spins = 0;

	for (;;)
	{
		const int64_t idx = look - (int64_t)(wave * 64u + lane);
		uint32_t status = ST_PREFIX, value = 0; // positions before tile 0 contribute nothing
		if (idx >= 0)
		{
			uint64_t w = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			uint32_t tag = (uint32_t)(w >> 32);
			status = (tag >> 2) == (epoch & 0x3fffffffu) ? (tag & 3u) : (uint32_t)ST_INVALID;
			value = (uint32_t)w;
		}

		const uint64_t prefixMask = __ballot(status == ST_PREFIX);
		const uint64_t invalidMask = __ballot(status == ST_INVALID);
		uint32_t verdict, partial;
		if (prefixMask != 0)
		{
			const uint32_t p = (uint32_t)__builtin_ctzll(prefixMask);
			const uint64_t upto = p == 63 ? ~0ull : ((1ull << (p + 1)) - 1);
			verdict = (invalidMask & upto) == 0 ? (uint32_t)W_DONE : (uint32_t)W_BLOCKED;
			partial = wave_sum_u32(lane <= p ? value : 0u);
		}
		else
		{
			verdict = invalidMask == 0 ? (uint32_t)W_CONTINUE : (uint32_t)W_BLOCKED;
			partial = wave_sum_u32(value);
		}

		__syncthreads(); // scratch free (previous round's readers are done)
		if (lane == 0)
		{
			scratch[wave] = verdict;
			scratch[4 + wave] = partial;
		}
		__syncthreads();

		// every thread folds the four sub-windows nearest-first
		uint32_t sum = 0, outcome = W_CONTINUE;
#pragma unroll
		for (int w = 0; w < 4; ++w)
		{
			if (outcome == W_CONTINUE)
			{
				const uint32_t v = scratch[w];
				if (v != W_BLOCKED)
					sum += scratch[4 + w];
				outcome = v;
			}
		}

		if (outcome == W_DONE)
		{
			exclusive += sum;
			break;
		}
		if (outcome == W_CONTINUE)
		{
			exclusive += sum;
			look -= 256;
			continue;
		}

		if (++spins > NV_SPIN_LIMIT)
		{
			if (threadIdx.x == 0)
				__hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			break;
		}
		__builtin_amdgcn_s_sleep(2);
	}

	if (threadIdx.x == 0)
		__hip_atomic_store(&state[tile], pack_state(epoch, ST_PREFIX, exclusive + aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return exclusive;
}

// Called by ONE thread of the workgroup that owns the LAST tile, after its look-back: at that point every tile has
// published its inclusive prefix, so no workgroup polls the state array any more and every workgroup that owns a
// tile has long read the epoch.  The next launch (stream order) sees the advanced epoch; stale granules of this
// launch then read as "invalid".  On wrap-around the caller zeroes the state array (tags 2^30 launches old).
__device__ __forceinline__ void advance_epoch(OrderCtl* ctl, uint32_t epoch)
{
	uint32_t next = (epoch + 1) & 0x3fffffffu;
	__hip_atomic_store(&ctl->epoch, next == 0 ? 1u : next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

} // namespace nv
