// ordered.cuh — deterministic (invocation-ordered) append for gfx950.
//
// niagara appends survivors with one global atomicAdd per invocation (drawcull.comp.glsl:123,143;
// clustercull.comp.glsl:135 — "TODO: potentially slow global atomic"), so its output order is whatever the
// hardware serialises.  Here the append index of an item is the exclusive prefix sum of the emit counts of all
// items before it in invocation order, computed in ONE pass with a chained scan across workgroups
// (decoupled look-back).  That is one valid serialisation of the reference's atomics, it is bit-reproducible,
// and it replaces 1 atomic per survivor with 1 ticket + 2 eight-byte publishes per TILE.
//
// MI355X specifics (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   * nothing is assumed about dispatch order or residency: tiles are handed out by tickets, so a tile only ever
//     waits on tiles that some running workgroup already owns;
//   * one hot atomic word saturates at ~88 returning atomics/us, so tickets are sharded over NV_SHARDS words on
//     separate 128-B lines; tile id = n * NV_SHARDS + shard.  Deadlock freedom: the lowest unfinished tile T of
//     shard s is either owned (and then never waits on an unowned tile, by induction on T) or all earlier tiles
//     of s are finished, so a workgroup of s is free to draw T.  Every shard has workgroups because
//     gridDim.x >= NV_SHARDS and shard = blockIdx.x % NV_SHARDS;
//   * per-XCD L2s are not coherent: every shared word is an 8-byte {epoch, status, value} granule written by one
//     agent-scope relaxed atomic store and polled with agent-scope relaxed loads (the "data is the flag" form,
//     no fences needed because no other payload is handed over);
//   * state is self-cleaning and replay-safe: granules carry the launch epoch, which lives in device memory and
//     is advanced by the last workgroup to leave, so nothing has to be memset between launches and a captured
//     hipGraph replays correctly;
//   * every spin is bounded; a timeout sets ctl->error (reported by nv_status) instead of hanging the GPU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define NV_SHARDS 32u
#define NV_SPIN_LIMIT (1u << 22)

namespace nv
{

struct OrderCtl
{
	uint32_t epoch;  // >= 1; granules of other epochs read as "invalid"
	uint32_t exited; // workgroups that have left the tile loop in the current launch
	uint32_t error;  // sticky: 1 = look-back spin bound hit
	uint32_t pad[29];
	uint32_t ticket[NV_SHARDS][32]; // one counter per 128-B line
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

// one lane draws the next tile of its shard (returning agent-scope atomic)
__device__ __forceinline__ uint32_t draw_ticket(OrderCtl* ctl, uint32_t shard)
{
	uint32_t n = __hip_atomic_fetch_add(&ctl->ticket[shard][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return n * NV_SHARDS + shard;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_xor(v, o, 64);
	return v;
}

// Called by all 64 lanes of ONE wave with wave-uniform arguments.  Publishes this tile's aggregate, looks back
// for the exclusive prefix, publishes the inclusive prefix and returns the exclusive prefix.
// base0 = value of the count word before the pass (what the first atomicAdd of the reference would return).
__device__ __forceinline__ uint32_t lookback_exclusive(uint64_t* __restrict__ state, OrderCtl* __restrict__ ctl, uint32_t tile,
                                                       uint32_t epoch, uint32_t aggregate, uint32_t base0)
{
	const uint32_t lane = __lane_id();

	if (tile == 0)
	{
		if (lane == 0)
			__hip_atomic_store(&state[0], pack_state(epoch, ST_PREFIX, base0 + aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return base0;
	}

	if (lane == 0)
		__hip_atomic_store(&state[tile], pack_state(epoch, ST_AGGREGATE, aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

	uint32_t exclusive = 0;
	int64_t look = (int64_t)tile - 1; // nearest predecessor handled by lane 0
	uint32_t spins = 0;

	for (;;)
	{
		int64_t idx = look - (int64_t)lane;
		uint32_t status = ST_PREFIX, value = 0; // lanes before tile 0 contribute nothing
		if (idx >= 0)
		{
			uint64_t w = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			uint32_t tag = (uint32_t)(w >> 32);
			status = (tag >> 2) == (epoch & 0x3fffffffu) ? (tag & 3u) : (uint32_t)ST_INVALID;
			value = (uint32_t)w;
		}

		uint64_t prefixMask = __ballot(status == ST_PREFIX);
		uint64_t invalidMask = __ballot(status == ST_INVALID);

		if (prefixMask != 0)
		{
			uint32_t p = (uint32_t)__builtin_ctzll(prefixMask);
			uint64_t upto = p == 63 ? ~0ull : ((1ull << (p + 1)) - 1);
			if ((invalidMask & upto) == 0)
			{
				exclusive += wave_sum_u32(lane <= p ? value : 0u);
				break;
			}
		}
		else if (invalidMask == 0)
		{
			exclusive += wave_sum_u32(value);
			look -= 64;
			continue;
		}

		if (++spins > NV_SPIN_LIMIT)
		{
			if (lane == 0)
				__hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			break;
		}
		__builtin_amdgcn_s_sleep(2);
	}

	if (lane == 0)
		__hip_atomic_store(&state[tile], pack_state(epoch, ST_PREFIX, exclusive + aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return exclusive;
}

// Called by ONE thread of a workgroup once it has drawn a ticket past the end.  The last workgroup to leave
// resets the tickets and advances the epoch for the next launch (stream order makes it visible).
// Returns true for the last workgroup when the epoch wrapped and the caller must zero the state array.
__device__ __forceinline__ bool leave_and_maybe_reset(OrderCtl* ctl, uint32_t epoch)
{
	uint32_t old = __hip_atomic_fetch_add(&ctl->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (old != gridDim.x - 1)
		return false;
	for (uint32_t s = 0; s < NV_SHARDS; ++s)
		__hip_atomic_store(&ctl->ticket[s][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(&ctl->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	uint32_t next = (epoch + 1) & 0x3fffffffu;
	bool wrapped = next == 0;
	__hip_atomic_store(&ctl->epoch, wrapped ? 1u : next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return wrapped;
}

} // namespace nv
