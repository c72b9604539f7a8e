// Development check: LDS-DMA (global_load_lds_dword) issued from inline asm with M0 saved/restored, completion via
// a counted s_waitcnt vmcnt, data read back with ordinary LDS loads.  Build: hipcc --offload-arch=gfx950 -o glds_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int slot)
{
	__shared__ uint32_t s_stage[4][6][256];
	const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint32_t* p = &s_stage[wave][slot][128];
	const uint32_t ldsAddr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)p);
	const uint32_t off = (63u - lane) * 4u + (blockIdx.x * 4 + wave) * 256u; // reversed lanes: LDS slot = lane, data = src[.. 63-lane]
	uint32_t saved;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
	             : "=&s"(saved) : "s"(ldsAddr), "v"(off), "s"(src) : "memory");
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	dst[blockIdx.x * 256 + threadIdx.x] = s_stage[wave][slot][128 + lane];
}
int main()
{
	const int blocks = 64, n = blocks * 256;
	std::vector<uint32_t> h(n), o(n);
	for (int i = 0; i < n; ++i) h[i] = i * 2654435761u;
	uint32_t *s, *d;
	hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
	hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
	int bad = 0;
	for (int slot = 0; slot < 6; ++slot)
	{
		hipMemset(d, 0, n * 4);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, s, d, slot);
		hipMemcpy(o.data(), d, n * 4, hipMemcpyDeviceToHost);
		for (int i = 0; i < n; ++i)
		{
			int wave = i / 64, lane = i % 64;
			if (o[i] != h[wave * 64 + (63 - lane)]) ++bad;
		}
	}
	printf("glds asm test: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
	return bad != 0;
}
