// ubench_valu.hip — VALU issue-rate microbenchmark for gfx950 (development tool, not part of the product).
// Measures cycles per wave-instruction for the instruction kinds the cull kernels are made of, at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
template <int KIND>
__global__ void bench(float* out, uint64_t* cycles, int iters)
{
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	float b0 = 1.0001f, b1 = 0.9999f;
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, q = { b0, b1 };
	f2 p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
	uint32_t u0 = threadIdx.x, s0 = 0;
	uint64_t t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i)
	{
#pragma unroll
		for (int r = 0; r < REP / 8; ++r)
		{
			if (KIND == 0)
				asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %9\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %9"
				             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
			if (KIND == 1)
				asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
				             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
			if (KIND == 2)
				asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
				             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
			if (KIND == 3)
				asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
				             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
			if (KIND == 4)
				asm volatile("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %1, 4\n v_readlane_b32 %0, %1, 5\n v_readlane_b32 %0, %1, 6\n v_readlane_b32 %0, %1, 7\n v_readlane_b32 %0, %1, 8\n v_readlane_b32 %0, %1, 9\n v_readlane_b32 %0, %1, 10"
				             : "+s"(s0) : "v"(u0));
			if (KIND == 5)
				asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %9"
				             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
			if (KIND == 6)
				asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
				             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
			if (KIND == 7)
				asm volatile("s_and_b64 %0, %0, %0\n s_or_b64 %0, %0, %0\n s_and_b64 %0, %0, %0\n s_or_b64 %0, %0, %0\n s_and_b64 %0, %0, %0\n s_or_b64 %0, %0, %0\n s_and_b64 %0, %0, %0\n s_or_b64 %0, %0, %0"
				             : "+s"(t0) : : "scc");
			if (KIND == 8)
				asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %9\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %9\n v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %9\n v_cvt_f32_f16 %6, %8\n v_cvt_f32_f16 %7, %9"
				             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
		}
	}
	uint64_t t1 = __builtin_readcyclecounter();
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + (float)s0;
	if (threadIdx.x == 0)
		cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, uint64_t* cyc)
{
	const int iters = 2000;
	for (int wavesPerSimd : { 1, 2, 4, 8 })
	{
		int threads = 64 * 4 * wavesPerSimd; // one block per CU: 4 SIMDs x wavesPerSimd waves
		if (threads > 1024) { threads = 1024; }
		int blocks = 256 * (wavesPerSimd == 8 ? 2 : 1);
		hipEvent_t e0, e1;
		hipEventCreate(&e0); hipEventCreate(&e1);
		hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
		hipEventRecord(e0);
		hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
		hipEventRecord(e1);
		hipDeviceSynchronize();
		float ms; hipEventElapsedTime(&ms, e0, e1);
		uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
		double insts = (double)iters * REP;           // per wave
		double wavesOnSimd = wavesPerSimd;
		// wall-clock based: cycles per wave-instruction per SIMD at 2.4 GHz
		double cyc_per_inst = ms * 1e-3 * 2.4e9 / (insts * wavesOnSimd);
		printf("%-14s waves/SIMD=%d  %.3f ms  s_memtime cycles/inst(one wave)=%.2f  wall: %.2f cyc per wave-inst per SIMD\n", name, wavesPerSimd, ms,
		       (double)c / insts, cyc_per_inst);
	}
}

int main()
{
	float* out; uint64_t* cyc;
	hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 1 << 16);
	run<0>("v_mul_f32", out, cyc);
	run<5>("v_add_f32", out, cyc);
	run<2>("v_fma_f32", out, cyc);
	run<1>("v_pk_mul_f32", out, cyc);
	run<6>("v_pk_add_f32", out, cyc);
	run<3>("v_pk_fma_f32", out, cyc);
	run<4>("v_readlane", out, cyc);
	run<8>("v_cvt_f32_f16", out, cyc);
	run<7>("s_and/or_b64", out, cyc);
	return 0;
}
