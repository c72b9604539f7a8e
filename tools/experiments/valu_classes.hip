// valu_classes.hip — round 4: what each instruction CLASS of the occlusion probe costs a SIMD on gfx950, and what the IEEE division /
// square root cost as hipcc expands them against their no-scale forms (development tool, not part of the product).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-value -o valu_classes tools/experiments/valu_classes.hip && ./valu_classes
//
// Part 1: eight independent chains of ONE instruction kind per wave, 1 / 2 / 4 waves per SIMD (one 256 / 512 / 1024-lane workgroup per CU);
//         cycles per wave-instruction and SIMD from the wall time at the nominal 2.4 GHz AND from s_memtime of wave 0 (the real shader clock).
// Part 2: eight independent divisions (square roots) per lane and iteration: `n / d` (`__builtin_sqrtf`) as hipcc expands it under
//         -ffp-contract=off, the same Newton steps without v_div_scale / v_div_fmas / v_div_fixup (no-scale form: identical when the scale is 1
//         and the operands are ordinary), and the 6-instruction Markstein form; cycles per division and SIMD.
// Part 3: bit-equality of the three forms over 2^30 random operand pairs with exponents inside the guard range.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define X8(s0) s0(0) s0(1) s0(2) s0(3) s0(4) s0(5) s0(6) s0(7)

// unary: op %i, %i ; binary: op %i, %i, %8 ; ternary: op %i, %i, %8, %9
#define UN(op) "" op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define BI(op) "" op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define BIR(op) "" op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n" op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7\n"
#define TE(op) "" op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
#define REGS8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

enum Kind
{
	K_FMA, K_MUL, K_ADD, K_FMAC, K_CNDMASK_VCC, K_CNDMASK_SGPR, K_CMP, K_CMP_E64, K_GLMIN, K_MIN, K_MAX3, K_MED3, K_FLOOR, K_CVT_I32_F32, K_CVT_F32_I32,
	K_DIV_SCALE, K_DIV_FMAS, K_DIV_FIXUP, K_RCP, K_SQRT, K_RSQ, K_FREXP_EXP, K_FREXP_MANT, K_LDEXP, K_ADD_U32, K_AND, K_LSHL, K_MAD_U24, K_LSHL_ADD,
	K_MIN_I32, K_MOV, K_MOV_B64, K_LSHL_ADD_U64, K_FMA_MIX, K_CVT_F16, K_CMP_CLASS, K_FMA_NOP0, K_FMA_NOP1, K_FMA_NOP3, K_MUL_DEP, K_READLANE, K_MOV_DPP,
	K_SUBBREV, K_COUNT
};
static const char* kKindName[K_COUNT] = {
	"v_fma_f32", "v_mul_f32", "v_add_f32", "v_fmac_f32", "v_cndmask_b32 (vcc)", "v_cndmask_b32 (sgpr pair)", "v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32 -> sgpr pair",
	"v_cmp_lt + v_cndmask (gl_min; per pair)", "v_min_f32", "v_max3_f32", "v_med3_f32", "v_floor_f32", "v_cvt_i32_f32", "v_cvt_f32_i32", "v_div_scale_f32",
	"v_div_fmas_f32", "v_div_fixup_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_frexp_exp_i32_f32", "v_frexp_mant_f32", "v_ldexp_f32", "v_add_u32",
	"v_and_b32", "v_lshlrev_b32", "v_mad_u32_u24", "v_lshl_add_u32", "v_min_i32", "v_mov_b32", "v_mov_b64 (per instruction)", "v_lshl_add_u64", "v_fma_mix_f32",
	"v_cvt_f32_f16", "v_cmp_class_f32", "v_fma_f32 + s_nop 0 (per pair)", "v_fma_f32 + s_nop 1 (per pair)", "v_fma_f32 + s_nop 3 (per pair)",
	"v_mul_f32 dependent chain", "v_readlane_b32", "v_mov_b32 dpp quad_perm", "v_subbrev_co_u32"
};

template <int KIND>
__global__ __launch_bounds__(1024) void class_kernel(float* out, uint64_t* cycles, int iters, float seed)
{
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	float b = 1.0000001f, c = 1e-9f;
	uint64_t sm = 0x5555555555555555ull, so = 0;
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
	uint32_t s0 = 0;
	uint64_t w0 = threadIdx.x, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3, w4 = w0 + 4, w5 = w0 + 5, w6 = w0 + 6, w7 = w0 + 7;
	const uint64_t t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i)
	{
#pragma unroll
		for (int r = 0; r < 8; ++r)
		{
			switch (KIND)
			{
			case K_FMA: asm volatile(TE("v_fma_f32") : REGS8 : "v"(b), "v"(c)); break;
			case K_MUL: asm volatile(BI("v_mul_f32") : REGS8 : "v"(b)); break;
			case K_ADD: asm volatile(BI("v_add_f32") : REGS8 : "v"(c)); break;
			case K_FMAC: asm volatile(BIR("v_fmac_f32") : REGS8 : "v"(c), "v"(b)); break; // a += c * a
			case K_CNDMASK_VCC: asm volatile(BI("v_cndmask_b32") : REGS8 : "v"(b) : "vcc"); break; // (vcc implied by the e32 form)
			case K_CNDMASK_SGPR: asm volatile(TE("v_cndmask_b32") : REGS8 : "v"(b), "s"(sm)); break;
			case K_CMP: asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
				                     "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n" : REGS8 : "v"(b) : "vcc"); break;
			case K_CMP_E64: asm volatile("v_cmp_lt_f32 %9, %0, %8\n v_cmp_lt_f32 %9, %1, %8\n v_cmp_lt_f32 %9, %2, %8\n v_cmp_lt_f32 %9, %3, %8\n"
				                         "v_cmp_lt_f32 %9, %4, %8\n v_cmp_lt_f32 %9, %5, %8\n v_cmp_lt_f32 %9, %6, %8\n v_cmp_lt_f32 %9, %7, %8\n" : REGS8, "+v"(b), "+s"(so)); break;
			case K_GLMIN: asm volatile("v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %8, %1\n v_cndmask_b32 %1, %1, %8, vcc\n"
				                       "v_cmp_lt_f32 vcc, %8, %2\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %8, %3\n v_cndmask_b32 %3, %3, %8, vcc\n"
				                       "v_cmp_lt_f32 vcc, %8, %4\n v_cndmask_b32 %4, %4, %8, vcc\n v_cmp_lt_f32 vcc, %8, %5\n v_cndmask_b32 %5, %5, %8, vcc\n"
				                       "v_cmp_lt_f32 vcc, %8, %6\n v_cndmask_b32 %6, %6, %8, vcc\n v_cmp_lt_f32 vcc, %8, %7\n v_cndmask_b32 %7, %7, %8, vcc\n" : REGS8 : "v"(b) : "vcc"); break;
			case K_MIN: asm volatile(BI("v_min_f32") : REGS8 : "v"(b)); break;
			case K_MAX3: asm volatile(TE("v_max3_f32") : REGS8 : "v"(b), "v"(c)); break;
			case K_MED3: asm volatile(TE("v_med3_f32") : REGS8 : "v"(b), "v"(c)); break;
			case K_FLOOR: asm volatile(UN("v_floor_f32") : REGS8); break;
			case K_CVT_I32_F32: asm volatile(UN("v_cvt_i32_f32") : REGS8); break;
			case K_CVT_F32_I32: asm volatile(UN("v_cvt_f32_i32") : REGS8); break;
			case K_DIV_SCALE: asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_scale_f32 %1, vcc, %1, %8, %1\n v_div_scale_f32 %2, vcc, %2, %8, %2\n"
				                           "v_div_scale_f32 %3, vcc, %3, %8, %3\n v_div_scale_f32 %4, vcc, %4, %8, %4\n v_div_scale_f32 %5, vcc, %5, %8, %5\n"
				                           "v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_scale_f32 %7, vcc, %7, %8, %7\n" : REGS8 : "v"(b) : "vcc"); break;
			case K_DIV_FMAS: asm volatile(TE("v_div_fmas_f32") : REGS8 : "v"(b), "v"(c) : "vcc"); break;
			case K_DIV_FIXUP: asm volatile(TE("v_div_fixup_f32") : REGS8 : "v"(b), "v"(c)); break;
			case K_RCP: asm volatile(UN("v_rcp_f32") : REGS8); break;
			case K_SQRT: asm volatile(UN("v_sqrt_f32") : REGS8); break;
			case K_RSQ: asm volatile(UN("v_rsq_f32") : REGS8); break;
			case K_FREXP_EXP: asm volatile(UN("v_frexp_exp_i32_f32") : REGS8); break;
			case K_FREXP_MANT: asm volatile(UN("v_frexp_mant_f32") : REGS8); break;
			case K_LDEXP: asm volatile(BI("v_ldexp_f32") : REGS8 : "v"(1)); break;
			case K_ADD_U32: asm volatile(BI("v_add_u32") : REGS8 : "v"(b)); break;
			case K_AND: asm volatile(BI("v_and_b32") : REGS8 : "v"(b)); break;
			case K_LSHL: asm volatile(BIR("v_lshlrev_b32") : REGS8 : "v"(1)); break;
			case K_MAD_U24: asm volatile(TE("v_mad_u32_u24") : REGS8 : "v"(b), "v"(c)); break;
			case K_LSHL_ADD: asm volatile(TE("v_lshl_add_u32") : REGS8 : "v"(1), "v"(c)); break;
			case K_MIN_I32: asm volatile(BI("v_min_i32") : REGS8 : "v"(b)); break;
			case K_MOV: asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n"
				                     : REGS8 : "v"(b)); break;
			case K_MOV_B64: asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0\n"
				                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)); break;
			case K_LSHL_ADD_U64: asm volatile("v_lshl_add_u64 %0, %0, 2, %1\n v_lshl_add_u64 %1, %1, 2, %2\n v_lshl_add_u64 %2, %2, 2, %3\n v_lshl_add_u64 %3, %3, 2, %4\n"
				                              "v_lshl_add_u64 %4, %4, 2, %5\n v_lshl_add_u64 %5, %5, 2, %6\n v_lshl_add_u64 %6, %6, 2, %7\n v_lshl_add_u64 %7, %7, 2, %0\n"
				                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)); break;
			case K_FMA_MIX: asm volatile(TE("v_fma_mix_f32") : REGS8 : "v"(b), "v"(c)); break;
			case K_CVT_F16: asm volatile(UN("v_cvt_f32_f16") : REGS8); break;
			case K_CMP_CLASS: asm volatile("v_cmp_class_f32 vcc, %0, %8\n v_cmp_class_f32 vcc, %1, %8\n v_cmp_class_f32 vcc, %2, %8\n v_cmp_class_f32 vcc, %3, %8\n"
				                           "v_cmp_class_f32 vcc, %4, %8\n v_cmp_class_f32 vcc, %5, %8\n v_cmp_class_f32 vcc, %6, %8\n v_cmp_class_f32 vcc, %7, %8\n" : REGS8 : "v"(0x204) : "vcc"); break;
			case K_FMA_NOP0: asm volatile("v_fma_f32 %0, %0, %8, %9\n s_nop 0\n v_fma_f32 %1, %1, %8, %9\n s_nop 0\n v_fma_f32 %2, %2, %8, %9\n s_nop 0\n v_fma_f32 %3, %3, %8, %9\n s_nop 0\n"
				                          "v_fma_f32 %4, %4, %8, %9\n s_nop 0\n v_fma_f32 %5, %5, %8, %9\n s_nop 0\n v_fma_f32 %6, %6, %8, %9\n s_nop 0\n v_fma_f32 %7, %7, %8, %9\n s_nop 0\n" : REGS8 : "v"(b), "v"(c)); break;
			case K_FMA_NOP1: asm volatile("v_fma_f32 %0, %0, %8, %9\n s_nop 1\n v_fma_f32 %1, %1, %8, %9\n s_nop 1\n v_fma_f32 %2, %2, %8, %9\n s_nop 1\n v_fma_f32 %3, %3, %8, %9\n s_nop 1\n"
				                          "v_fma_f32 %4, %4, %8, %9\n s_nop 1\n v_fma_f32 %5, %5, %8, %9\n s_nop 1\n v_fma_f32 %6, %6, %8, %9\n s_nop 1\n v_fma_f32 %7, %7, %8, %9\n s_nop 1\n" : REGS8 : "v"(b), "v"(c)); break;
			case K_FMA_NOP3: asm volatile("v_fma_f32 %0, %0, %8, %9\n s_nop 3\n v_fma_f32 %1, %1, %8, %9\n s_nop 3\n v_fma_f32 %2, %2, %8, %9\n s_nop 3\n v_fma_f32 %3, %3, %8, %9\n s_nop 3\n"
				                          "v_fma_f32 %4, %4, %8, %9\n s_nop 3\n v_fma_f32 %5, %5, %8, %9\n s_nop 3\n v_fma_f32 %6, %6, %8, %9\n s_nop 3\n v_fma_f32 %7, %7, %8, %9\n s_nop 3\n" : REGS8 : "v"(b), "v"(c)); break;
			case K_MUL_DEP: asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n"
				                         : REGS8 : "v"(b)); break;
			case K_READLANE: asm volatile("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %1, 4\n v_readlane_b32 %0, %1, 5\n v_readlane_b32 %0, %1, 6\n v_readlane_b32 %0, %1, 7\n v_readlane_b32 %0, %1, 8\n"
				                          "v_readlane_b32 %0, %1, 9\n v_readlane_b32 %0, %1, 10\n" : "+s"(s0) : "v"(a0)); break;
			case K_MOV_DPP: asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
				                         "v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
				                         "v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
				                         "v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" : REGS8); break;
			case K_SUBBREV: asm volatile("v_subbrev_co_u32 %0, vcc, %0, %8, vcc\n v_subbrev_co_u32 %1, vcc, %1, %8, vcc\n v_subbrev_co_u32 %2, vcc, %2, %8, vcc\n v_subbrev_co_u32 %3, vcc, %3, %8, vcc\n"
				                         "v_subbrev_co_u32 %4, vcc, %4, %8, vcc\n v_subbrev_co_u32 %5, vcc, %5, %8, vcc\n v_subbrev_co_u32 %6, vcc, %6, %8, vcc\n v_subbrev_co_u32 %7, vcc, %7, %8, vcc\n" : REGS8 : "v"(b) : "vcc"); break;
			}
		}
	}
	const uint64_t t1 = __builtin_readcyclecounter();
	float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + (float)s0 + (float)(uint32_t)so + b + (float)(uint32_t)(w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7);
	if (s == 12345.678f)
		out[threadIdx.x] = s;
	if (threadIdx.x == 0)
		cycles[blockIdx.x] = t1 - t0;
}

// ---- part 2 / 3: the division and the square root
__device__ __forceinline__ float div_ieee(float n, float d) { return n / d; }
// hipcc's expansion without v_div_scale (scale 1), v_div_fmas (a plain fma then) and v_div_fixup (ordinary operands)
__device__ __forceinline__ float div_noscale(float n, float d)
{
	const float r0 = __builtin_amdgcn_rcpf(d);
	const float e = __builtin_fmaf(-d, r0, 1.0f);
	const float r1 = __builtin_fmaf(e, r0, r0);
	const float q0 = n * r1;
	const float m0 = __builtin_fmaf(-d, q0, n);
	const float q1 = __builtin_fmaf(m0, r1, q0);
	const float m1 = __builtin_fmaf(-d, q1, n);
	return __builtin_fmaf(m1, r1, q1);
}
__device__ __forceinline__ float div_markstein(float n, float d)
{
	const float r0 = __builtin_amdgcn_rcpf(d);
	const float e = __builtin_fmaf(-d, r0, 1.0f);
	const float r1 = __builtin_fmaf(e, r0, r0);
	const float q0 = n * r1;
	const float m0 = __builtin_fmaf(-d, q0, n);
	return __builtin_fmaf(m0, r1, q0);
}
__device__ __forceinline__ float sqrt_ieee(float x) { return __builtin_sqrtf(x); }
// hipcc's IEEE expansion (v_sqrt_f32, then the two neighbours tried with exact residuals) without the denormal rescaling and the class fix-up
__device__ __forceinline__ float sqrt_noscale(float x)
{
	const float s = __builtin_amdgcn_sqrtf(x);
	const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
	const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
	float r = rm <= 0.0f ? sm : s;
	r = rp > 0.0f ? sp : r;
	return r;
}

// the sequence LLVM emits for a correctly rounded fp32 square root that need not handle denormals (cullmath.h sqrt_ordinary)
__device__ __forceinline__ float sqrt_rsq(float x)
{
	const float y = __builtin_amdgcn_rsqf(x);
	float g = x * y;
	float h = 0.5f * y;
	const float e = __builtin_fmaf(-h, g, 0.5f);
	h = __builtin_fmaf(h, e, h);
	g = __builtin_fmaf(g, e, g);
	const float d = __builtin_fmaf(-g, g, x);
	return __builtin_fmaf(d, h, g);
}

// EVERY fp32 in [2^-40, 2^40]: both square-root forms against __builtin_sqrtf, and the refined reciprocal of the division's first two
// Newton steps against the correctly rounded 1 / d (what the six-instruction Markstein division would rest on)
__global__ __launch_bounds__(256) void exhaustive_kernel(unsigned long long* counts, uint32_t firstBits, uint32_t lastBits)
{
	unsigned long long badRsq = 0, badNb = 0, badRcp = 0;
	for (uint64_t u = (uint64_t)firstBits + blockIdx.x * 256u + threadIdx.x; u <= lastBits; u += (uint64_t)gridDim.x * 256u)
	{
		const float x = __uint_as_float((uint32_t)u);
		const uint32_t ref = __float_as_uint(sqrt_ieee(x));
		badRsq += __float_as_uint(sqrt_rsq(x)) != ref;
		badNb += __float_as_uint(sqrt_noscale(x)) != ref;
		const float r0 = __builtin_amdgcn_rcpf(x);
		const float r1 = __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
		badRcp += __float_as_uint(r1) != __float_as_uint(1.0f / x);
	}
	atomicAdd(&counts[0], badRsq);
	atomicAdd(&counts[1], badNb);
	atomicAdd(&counts[2], badRcp);
}

// v_cndmask_b32 reading VCC: alone it measured 23.5 cycles per instruction, right behind a v_cmp 3.8 — what about a second and third use of one compare,
// and a VCC written by the scalar unit?
template <int FORM>
__global__ __launch_bounds__(1024) void vcc_kernel(float* out, uint64_t* cycles, int iters, float seed)
{
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	float b = 1.0000001f;
	uint64_t sm = 0x5555555555555555ull, sm2 = 0;
	float sf = 1.0000001f, sf2 = 0;
	asm volatile("" : "+s"(sm2), "+s"(sf2), "+s"(sf));
	const uint64_t t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i)
	{
#pragma unroll
		for (int r = 0; r < 8; ++r)
		{
			if (FORM == 0) // one compare, two selects: 4 x (cmp, cndmask, cndmask) = 12 instructions
				asm volatile("v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n"
				             "v_cmp_lt_f32 vcc, %8, %2\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
				             "v_cmp_lt_f32 vcc, %8, %4\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
				             "v_cmp_lt_f32 vcc, %8, %6\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : REGS8 : "v"(b) : "vcc");
			else if (FORM == 1) // one compare, four selects: 2 x (cmp, 4 cndmask) = 10 instructions
				asm volatile("v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
				             "v_cmp_lt_f32 vcc, %8, %4\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : REGS8 : "v"(b) : "vcc");
			else if (FORM == 2) // VCC from the scalar unit, one select: 8 x (s_mov, cndmask) = 8 vector instructions
				asm volatile("s_mov_b64 vcc, %9\n v_cndmask_b32 %0, %0, %8, vcc\n s_mov_b64 vcc, %9\n v_cndmask_b32 %1, %1, %8, vcc\n s_mov_b64 vcc, %9\n v_cndmask_b32 %2, %2, %8, vcc\n"
				             "s_mov_b64 vcc, %9\n v_cndmask_b32 %3, %3, %8, vcc\n s_mov_b64 vcc, %9\n v_cndmask_b32 %4, %4, %8, vcc\n s_mov_b64 vcc, %9\n v_cndmask_b32 %5, %5, %8, vcc\n"
				             "s_mov_b64 vcc, %9\n v_cndmask_b32 %6, %6, %8, vcc\n s_mov_b64 vcc, %9\n v_cndmask_b32 %7, %7, %8, vcc\n" : REGS8 : "v"(b), "s"(sm) : "vcc");
			else if (FORM == 4) // a scalar-written SGPR pair (not VCC) as the select's mask: 8 x (s_mov, cndmask)
				asm volatile("s_mov_b64 %10, %9\n v_cndmask_b32 %0, %0, %8, %10\n s_mov_b64 %10, %9\n v_cndmask_b32 %1, %1, %8, %10\n s_mov_b64 %10, %9\n v_cndmask_b32 %2, %2, %8, %10\n"
				             "s_mov_b64 %10, %9\n v_cndmask_b32 %3, %3, %8, %10\n s_mov_b64 %10, %9\n v_cndmask_b32 %4, %4, %8, %10\n s_mov_b64 %10, %9\n v_cndmask_b32 %5, %5, %8, %10\n"
				             "s_mov_b64 %10, %9\n v_cndmask_b32 %6, %6, %8, %10\n s_mov_b64 %10, %9\n v_cndmask_b32 %7, %7, %8, %10\n" : REGS8 : "v"(b), "s"(sm), "s"(sm2));
			else if (FORM == 5) // a scalar-written SGPR as an arithmetic operand: 8 x (s_mov_b32, v_mul)
				asm volatile("s_mov_b32 %10, %9\n v_mul_f32 %0, %10, %0\n s_mov_b32 %10, %9\n v_mul_f32 %1, %10, %1\n s_mov_b32 %10, %9\n v_mul_f32 %2, %10, %2\n"
				             "s_mov_b32 %10, %9\n v_mul_f32 %3, %10, %3\n s_mov_b32 %10, %9\n v_mul_f32 %4, %10, %4\n s_mov_b32 %10, %9\n v_mul_f32 %5, %10, %5\n"
				             "s_mov_b32 %10, %9\n v_mul_f32 %6, %10, %6\n s_mov_b32 %10, %9\n v_mul_f32 %7, %10, %7\n" : REGS8 : "v"(b), "s"(sf), "s"(sf2));
			else if (FORM == 6) // VCC from the scalar unit, a multiply, then the select: 8 x (s_mov, mul, cndmask) — does one instruction of distance hide it?
				asm volatile("s_mov_b64 vcc, %9\n v_mul_f32 %1, %1, %8\n v_cndmask_b32 %0, %0, %8, vcc\n s_mov_b64 vcc, %9\n v_mul_f32 %2, %2, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
				             "s_mov_b64 vcc, %9\n v_mul_f32 %3, %3, %8\n v_cndmask_b32 %2, %2, %8, vcc\n s_mov_b64 vcc, %9\n v_mul_f32 %4, %4, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
				             "s_mov_b64 vcc, %9\n v_mul_f32 %5, %5, %8\n v_cndmask_b32 %4, %4, %8, vcc\n s_mov_b64 vcc, %9\n v_mul_f32 %6, %6, %8\n v_cndmask_b32 %5, %5, %8, vcc\n"
				             "s_mov_b64 vcc, %9\n v_mul_f32 %7, %7, %8\n v_cndmask_b32 %6, %6, %8, vcc\n s_mov_b64 vcc, %9\n v_mul_f32 %0, %0, %8\n v_cndmask_b32 %7, %7, %8, vcc\n" : REGS8 : "v"(b), "s"(sm) : "vcc");
			else if (FORM == 7) // a compare into an SGPR pair, the scalar unit ANDs it into another pair, the select reads that: 8 x (cmp, s_and, cndmask)
				asm volatile("v_cmp_lt_f32 %10, %8, %0\n s_and_b64 %10, %10, %9\n v_cndmask_b32 %0, %0, %8, %10\n v_cmp_lt_f32 %10, %8, %1\n s_and_b64 %10, %10, %9\n v_cndmask_b32 %1, %1, %8, %10\n"
				             "v_cmp_lt_f32 %10, %8, %2\n s_and_b64 %10, %10, %9\n v_cndmask_b32 %2, %2, %8, %10\n v_cmp_lt_f32 %10, %8, %3\n s_and_b64 %10, %10, %9\n v_cndmask_b32 %3, %3, %8, %10\n"
				             "v_cmp_lt_f32 %10, %8, %4\n s_and_b64 %10, %10, %9\n v_cndmask_b32 %4, %4, %8, %10\n v_cmp_lt_f32 %10, %8, %5\n s_and_b64 %10, %10, %9\n v_cndmask_b32 %5, %5, %8, %10\n"
				             "v_cmp_lt_f32 %10, %8, %6\n s_and_b64 %10, %10, %9\n v_cndmask_b32 %6, %6, %8, %10\n v_cmp_lt_f32 %10, %8, %7\n s_and_b64 %10, %10, %9\n v_cndmask_b32 %7, %7, %8, %10\n"
				             : REGS8 : "v"(b), "s"(sm), "s"(sm2) : "scc");
			else // selects on VCC with a multiply between them: 8 x (cndmask, mul) = 16 instructions
				asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_mul_f32 %1, %1, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_mul_f32 %3, %3, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_mul_f32 %5, %5, %8\n"
				             "v_cndmask_b32 %6, %6, %8, vcc\n v_mul_f32 %7, %7, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_mul_f32 %0, %0, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_mul_f32 %2, %2, %8\n"
				             "v_cndmask_b32 %5, %5, %8, vcc\n v_mul_f32 %4, %4, %8\n v_cndmask_b32 %7, %7, %8, vcc\n v_mul_f32 %6, %6, %8\n" : REGS8 : "v"(b) : "vcc");
		}
	}
	const uint64_t t1 = __builtin_readcyclecounter();
	float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if (s == 12345.678f)
		out[threadIdx.x] = s;
	if (threadIdx.x == 0)
		cycles[blockIdx.x] = t1 - t0;
}

template <int FORM>
__global__ __launch_bounds__(1024) void seq_kernel(float* out, uint64_t* cycles, int iters, float seed)
{
	float n[8], d[8];
#pragma unroll
	for (int k = 0; k < 8; ++k)
	{
		n[k] = seed + 1.5f + threadIdx.x * 0.37f + k;
		d[k] = seed + 2.25f + threadIdx.x * 0.11f + k * 0.5f;
	}
	const uint64_t t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i)
	{
#pragma unroll
		for (int k = 0; k < 8; ++k)
		{
			float q;
			if (FORM == 0) q = div_ieee(n[k], d[k]);
			else if (FORM == 1) q = div_noscale(n[k], d[k]);
			else if (FORM == 2) q = div_markstein(n[k], d[k]);
			else if (FORM == 3) q = sqrt_ieee(n[k]);
			else if (FORM == 4) q = sqrt_noscale(n[k]);
			else q = sqrt_rsq(n[k]);
			// feed the result back so that nothing is hoisted; keeps the operands ordinary (1 < x < 1e3)
			n[k] = q + d[k];
			asm volatile("" : "+v"(n[k]));
		}
	}
	const uint64_t t1 = __builtin_readcyclecounter();
	float s = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k)
		s += n[k];
	if (s == 12345.678f)
		out[threadIdx.x] = s;
	if (threadIdx.x == 0)
		cycles[blockIdx.x] = t1 - t0;
}

__device__ __forceinline__ uint32_t pcg(uint32_t& st)
{
	st = st * 747796405u + 2891336453u;
	uint32_t w = ((st >> ((st >> 28) + 4u)) ^ st) * 277803737u;
	return (w >> 22) ^ w;
}

// random operands with biased exponents in [loExp, hiExp], random signs and mantissas; mismatches counted per form
__global__ __launch_bounds__(256) void equal_kernel(unsigned long long* counts, uint32_t perThread, uint32_t loExp, uint32_t hiExp, int edge)
{
	uint32_t st = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u + (uint32_t)edge * 977u;
	unsigned long long badNo = 0, badMk = 0, badSq = 0;
	const uint32_t span = hiExp - loExp + 1u;
	for (uint32_t i = 0; i < perThread; ++i)
	{
		uint32_t a = pcg(st), b = pcg(st), e = pcg(st);
		uint32_t ma = a & 0x7fffffu, mb = b & 0x7fffffu;
		if (edge == 1) // mantissas near all-ones / all-zeros, where the rounding of the reciprocal is tightest
		{
			ma = (a & 1u) ? 0x7fffffu - (a >> 24 & 15u) : (a >> 24 & 15u);
			mb = (b & 1u) ? 0x7fffffu - (b >> 24 & 15u) : (b >> 24 & 15u);
		}
		const uint32_t ea = loExp + (e & 0xffffu) % span, eb = loExp + (e >> 16) % span;
		const float n = __uint_as_float((a & 0x80000000u) | ea << 23 | ma);
		const float d = __uint_as_float((b & 0x80000000u) | eb << 23 | mb);
		const uint32_t q = __float_as_uint(div_ieee(n, d));
		badNo += __float_as_uint(div_noscale(n, d)) != q;
		badMk += __float_as_uint(div_markstein(n, d)) != q;
		const float x = __uint_as_float(ea << 23 | ma);
		badSq += __float_as_uint(sqrt_noscale(x)) != __float_as_uint(sqrt_ieee(x));
	}
	atomicAdd(&counts[0], badNo);
	atomicAdd(&counts[1], badMk);
	atomicAdd(&counts[2], badSq);
}

template <typename K>
static void time_kernel(K kernel, const char* name, double instPerIter, float* out, uint64_t* cyc)
{
	const int iters = 1500;
	for (int w = 1; w <= 4; w *= 2)
	{
		const int threads = 256 * w; // one workgroup per CU: w waves on each of its four SIMDs
		hipEvent_t e0, e1;
		hipEventCreate(&e0);
		hipEventCreate(&e1);
		kernel<<<256, threads>>>(out, cyc, 10, 1.f);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		kernel<<<256, threads>>>(out, cyc, iters, 1.f);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		uint64_t c[256];
		hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
		double mean = 0;
		for (int i = 0; i < 256; ++i)
			mean += (double)c[i] / 256;
		const double units = (double)iters * instPerIter; // per wave
		printf("%-44s waves/SIMD %d: wall %.2f cyc (2.4 GHz)  s_memtime %.2f cyc  per unit and SIMD   [%.3f ms, %.2f GHz implied]\n", name, w,
		       ms * 1e-3 * 2.4e9 / (units * w), mean / (units * w), ms, mean / (ms * 1e-3) * 1e-9);
		hipEventDestroy(e0);
		hipEventDestroy(e1);
	}
}

template <int K>
struct RunAll
{
	static void go(float* out, uint64_t* cyc)
	{
		const bool pair = K == K_GLMIN || K == K_FMA_NOP0 || K == K_FMA_NOP1 || K == K_FMA_NOP3;
		(void)pair;
		time_kernel(class_kernel<K>, kKindName[K], 64.0, out, cyc); // 8 x 8 instructions (or pairs) per iteration
		RunAll<K + 1>::go(out, cyc);
	}
};
template <>
struct RunAll<K_COUNT>
{
	static void go(float*, uint64_t*) {}
};

int main(int argc, char** argv)
{
	float* out;
	uint64_t* cyc;
	hipMalloc(&out, 1 << 20);
	hipMalloc(&cyc, 1 << 16);
	if (argc < 2 || !strcmp(argv[1], "classes"))
		RunAll<0>::go(out, cyc);
	if (argc < 2 || !strcmp(argv[1], "seq"))
	{
		time_kernel(seq_kernel<0>, "division, hipcc's IEEE expansion (+1 add)", 8.0, out, cyc);
		time_kernel(seq_kernel<1>, "division, no-scale form (+1 add)", 8.0, out, cyc);
		time_kernel(seq_kernel<2>, "division, Markstein 6 (+1 add)", 8.0, out, cyc);
		time_kernel(seq_kernel<3>, "square root, hipcc's IEEE expansion (+1 add)", 8.0, out, cyc);
		time_kernel(seq_kernel<4>, "square root, no-scale form (+1 add)", 8.0, out, cyc);
		time_kernel(seq_kernel<5>, "square root, rsq + Newton (+1 add)", 8.0, out, cyc);
	}
	if (argc < 2 || !strcmp(argv[1], "vcc"))
	{
		time_kernel(vcc_kernel<0>, "v_cmp + 2 x v_cndmask vcc (per group of 3)", 32.0, out, cyc);
		time_kernel(vcc_kernel<1>, "v_cmp + 4 x v_cndmask vcc (per group of 5)", 16.0, out, cyc);
		time_kernel(vcc_kernel<2>, "s_mov vcc + v_cndmask vcc (per pair)", 64.0, out, cyc);
		time_kernel(vcc_kernel<3>, "v_cndmask vcc + v_mul (per pair)", 64.0, out, cyc);
		time_kernel(vcc_kernel<4>, "s_mov sgpr pair + v_cndmask on it (per pair)", 64.0, out, cyc);
		time_kernel(vcc_kernel<5>, "s_mov sgpr + v_mul reading it (per pair)", 64.0, out, cyc);
		time_kernel(vcc_kernel<6>, "s_mov vcc + v_mul + v_cndmask vcc (per group of 3)", 64.0, out, cyc);
		time_kernel(vcc_kernel<7>, "v_cmp -> sgpr, s_and, v_cndmask on it (per group of 3)", 64.0, out, cyc);
	}
	if (argc < 2 || !strcmp(argv[1], "exhaustive"))
	{
		unsigned long long* counts;
		hipMalloc(&counts, 64);
		hipMemset(counts, 0, 64);
		const uint32_t first = 0x2B800000u, last = 0x53800000u; // 2^-40 .. 2^40
		exhaustive_kernel<<<4096, 256>>>(counts, first, last);
		unsigned long long h[3];
		hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost);
		printf("every fp32 in [2^-40, 2^40] (%llu values): rsq + Newton sqrt differs from __builtin_sqrtf on %llu, the neighbour form on %llu; "
		       "the twice-refined v_rcp_f32 differs from the correctly rounded 1 / d on %llu\n", (unsigned long long)last - first + 1, h[0], h[1], h[2]);
	}
	if (argc < 2 || !strcmp(argv[1], "equal"))
	{
		unsigned long long* counts;
		hipMalloc(&counts, 64);
		for (int edge = 0; edge < 2; ++edge)
		{
			hipMemset(counts, 0, 64);
			const uint32_t blocks = 4096, per = 1024; // 2^30 pairs
			equal_kernel<<<blocks, 256>>>(counts, per, 80u, 174u, edge);
			unsigned long long h[3];
			hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost);
			printf("%s operands, exponents 2^-47..2^47, %llu pairs: no-scale division differs from n / d on %llu, Markstein 6 on %llu; no-scale sqrt on %llu\n",
			       edge ? "edge-mantissa" : "random", (unsigned long long)blocks * 256 * per, h[0], h[1], h[2]);
		}
	}
	return 0;
}
