// pk_rate.hip — microbenchmark: issue rate of v_fma_f32 / v_mul_f32 / v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950.
// Question (round 3): would evaluating TWO occlusion probes per lane with packed fp32 arithmetic halve the VALU time of the
// VALU-bound stages?  Only if a packed instruction issues in the time of a plain one.
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate tools/experiments/pk_rate.hip && ./pk_rate
// Result (MI355X, 2-4 waves per SIMD, cycles per wave-instruction and SIMD at the nominal 2.4 GHz): v_fma_f32 2.9, v_mul / v_add 2.7,
// v_pk_fma_f32 5.1, v_pk_mul / v_pk_add 4.9, v_rcp_f32 8.3, a cndmask / cvt / floor / max / and / lshl mix 4.1; one wave alone 5.1-6.1
// whatever the instruction.  Packing two probes into one lane buys 1.16x on the multiply-adds and nothing elsewhere: not built.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed)
{
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
	f2 m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
	for (int i = 0; i < iters; ++i)
	{
		if (KIND == 0)
		{
			REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
			                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
			                  : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
		}
		else if (KIND == 1)
		{
			REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
			                  "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
			                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
		}
		else if (KIND == 2)
		{
			REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n"
			                  "v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
			                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
		}
		else if (KIND == 3)
		{
			REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
			                  "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
			                  : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
		}
		else if (KIND == 4)
		{
			REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
			                  "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
			                  : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
		}
		else if (KIND == 5)
		{
			REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cvt_f32_u32 %1, %1\n v_floor_f32 %2, %2\n v_max_f32 %3, %3, %9\n"
			                  "v_cndmask_b32 %4, %4, %8, vcc\n v_cvt_u32_f32 %5, %5\n v_and_b32 %6, %6, %8\n v_lshlrev_b32 %7, 1, %7\n"
			                  : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x) : "vcc");)
		}
	}
	f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if (s.x + s.y == 12345.678f)
		out[threadIdx.x] = s.x;
}

template <int KIND>
static void run(const char* name, int wavesPerSimd, float* out)
{
	const int iters = 2000;
	const int blocks = 256 * wavesPerSimd; // 256 threads = 4 waves = one per SIMD of a CU
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	rate_kernel<KIND><<<blocks, 256>>>(out, 10, 1.f);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	rate_kernel<KIND><<<blocks, 256>>>(out, iters, 1.f);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double instPerWave = (double)iters * 64;
	const double cycles = ms * 1e-3 * 2.4e9;
	printf("%-34s waves/SIMD %d: %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wavesPerSimd, ms, cycles / (instPerWave * wavesPerSimd));
}

int main()
{
	float* out;
	hipMalloc(&out, 4096);
	for (int w = 1; w <= 4; w *= 2)
	{
		run<0>("v_fma_f32", w, out);
		run<1>("v_pk_fma_f32", w, out);
		run<3>("v_mul_f32 / v_add_f32", w, out);
		run<2>("v_pk_mul_f32 / v_pk_add_f32", w, out);
		run<4>("v_rcp_f32", w, out);
		run<5>("cndmask/cvt/floor/max/and/lshl mix", w, out);
	}
	return 0;
}
