// valu_rates.hip - issue rate of the integer VALU instructions the leaf kernels
// are made of, on gfx950: one wave64 per SIMD (and 4 per SIMD), a long unrolled
// chain of independent copies of one instruction, cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
#define ITER 256

#define KERNEL(name, decl, body) \
__global__ void name(int *out, long long *cyc, int seed) { \
	int a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
	int a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
	int b = seed * 77 + 5, c = seed + 9; decl \
	long long t0 = __builtin_readcyclecounter(); \
	for(int i = 0; i < ITER; ++i) { \
		_Pragma("unroll") for(int r = 0; r < REP / 8; ++r) { body } \
	} \
	long long t1 = __builtin_readcyclecounter(); \
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
	if(threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; \
}

#define OP8(asmstr) \
	asm volatile(asmstr : "+v"(a0) : "v"(b), "v"(c)); asm volatile(asmstr : "+v"(a1) : "v"(b), "v"(c)); \
	asm volatile(asmstr : "+v"(a2) : "v"(b), "v"(c)); asm volatile(asmstr : "+v"(a3) : "v"(b), "v"(c)); \
	asm volatile(asmstr : "+v"(a4) : "v"(b), "v"(c)); asm volatile(asmstr : "+v"(a5) : "v"(b), "v"(c)); \
	asm volatile(asmstr : "+v"(a6) : "v"(b), "v"(c)); asm volatile(asmstr : "+v"(a7) : "v"(b), "v"(c));

KERNEL(k_add, , OP8("v_add_u32 %0, %0, %1"))
KERNEL(k_mul_lo, , OP8("v_mul_lo_u32 %0, %0, %1"))
KERNEL(k_mul_hi, , OP8("v_mul_hi_i32 %0, %0, %1"))
KERNEL(k_mul24, , OP8("v_mul_i32_i24 %0, %0, %1"))
KERNEL(k_mulhi24, , OP8("v_mul_hi_i32_i24 %0, %0, %1"))
KERNEL(k_mad24, , OP8("v_mad_i32_i24 %0, %0, %1, %2"))
KERNEL(k_bfe, , OP8("v_bfe_i32 %0, %0, 8, 17"))
KERNEL(k_alignbit, , OP8("v_alignbit_b32 %0, %0, %1, 24"))
KERNEL(k_dot2, , OP8("v_dot2_i32_i16 %0, %0, %1, %2"))
KERNEL(k_dot4, , OP8("v_dot4_i32_i8 %0, %0, %1, %2"))
KERNEL(k_lshl_add, , OP8("v_lshl_add_u32 %0, %0, 3, %1"))
KERNEL(k_add3, , OP8("v_add3_u32 %0, %0, %1, %2"))
KERNEL(k_pk_mul_lo, , OP8("v_pk_mul_lo_u16 %0, %0, %1"))
KERNEL(k_pk_mad, , OP8("v_pk_mad_i16 %0, %0, %1, %2"))
KERNEL(k_pk_add, , OP8("v_pk_add_i16 %0, %0, %1"))
KERNEL(k_cndmask, , OP8("v_cndmask_b32 %0, %0, %1, vcc"))
KERNEL(k_readlane, int s; , OP8("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0"))


KERNEL(k_sub, , OP8("v_sub_u32 %0, %0, %1"))
KERNEL(k_and, , OP8("v_and_b32 %0, %0, %1"))
KERNEL(k_xor, , OP8("v_xor_b32 %0, %0, %1"))
KERNEL(k_lshl, , OP8("v_lshlrev_b32 %0, 3, %0"))
KERNEL(k_ashr, , OP8("v_ashrrev_i32 %0, 3, %0"))
KERNEL(k_lshr, , OP8("v_lshrrev_b32 %0, 3, %0"))
KERNEL(k_mov, , OP8("v_mov_b32 %0, %1"))
KERNEL(k_max, , OP8("v_max_i32 %0, %0, %1"))
KERNEL(k_addco, , OP8("v_add_co_u32 %0, vcc, %0, %1"))
KERNEL(k_cnd3, , OP8("v_cndmask_b32 %0, %0, %1, s[10:11]"))
KERNEL(k_mulu24, , OP8("v_mul_u32_u24 %0, %0, %1"))
KERNEL(k_bfeu, , OP8("v_bfe_u32 %0, %0, 8, 8"))
KERNEL(k_bfi, , OP8("v_bfi_b32 %0, %1, %0, %2"))
KERNEL(k_perm, , OP8("v_perm_b32 %0, %0, %1, %2"))
KERNEL(k_and_or, , OP8("v_and_or_b32 %0, %0, %1, %2"))
KERNEL(k_sub_ashr, , OP8("v_sub_u32 %0, %0, %1\n v_ashrrev_i32 %0, 1, %0"))
KERNEL(k_add_sdwa, , OP8("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"))
KERNEL(k_ashr_sdwa, , OP8("v_mov_b32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0"))
// mixes: 1 add + 1 mul24 ; 2 add + 1 mul24 ; add + bfe ; add + mad_i64
KERNEL(k_mix_add_mul, , OP8("v_add_u32 %0, %0, %1\n v_mul_i32_i24 %0, %0, %2"))
KERNEL(k_mix_2add_mul, , OP8("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %2\n v_mul_i32_i24 %0, %0, %2"))
KERNEL(k_mix_add_bfe, , OP8("v_add_u32 %0, %0, %1\n v_bfe_i32 %0, %0, 8, 17"))
KERNEL(k_mix_mul_bfe, , OP8("v_mul_i32_i24 %0, %0, %1\n v_bfe_i32 %0, %0, 8, 17"))

// 64-bit mad: vdst pair
__global__ void k_mad64(int *out, long long *cyc, int seed)
{
	long long a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 1, a5 = a0 + 2, a6 = a0 + 3, a7 = a0 + 4;
	int b = seed * 77 + 5, c = seed + 9 + threadIdx.x;
	long long t0 = __builtin_readcyclecounter();
	for(int i = 0; i < ITER; ++i) {
#pragma unroll
		for(int r = 0; r < REP / 8; ++r) {
#define M(x) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c) : "vcc");
			M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
		}
	}
	long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
	if(threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template<class K> void run(const char *name, K k, int waves_per_simd)
{
	int *out; long long *cyc;
	const int blocks = 256 * 4, threads = 64 * waves_per_simd;   // one workgroup per SIMD-ish; occupancy via block size
	hipMalloc(&out, blocks * threads * sizeof(int));
	hipMalloc(&cyc, blocks * sizeof(long long));
	hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, 2);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
	double insts = (double)blocks * waves_per_simd * ITER * REP;       // wave-instructions
	// 100 MHz s_memtime-style counter vs shader clock: report both wall-derived and counter
	printf("%-12s waves/blk %d: %8.3f ms  %.2f T lane-ops/s  (%6.2f wave-insts/ns chip)  cyc[0]=%lld\n", name, waves_per_simd, ms,
			insts * 64 / (ms * 1e-3) / 1e12, insts / (ms * 1e6), h[0]);
	hipFree(out); hipFree(cyc);
}

int main()
{
	for(int w : {4, 8}) {
		run("add_u32", k_add, w); run("mul_lo_u32", k_mul_lo, w); run("mul_hi_i32", k_mul_hi, w);
		run("mul_i32_i24", k_mul24, w); run("mul_hi_i24", k_mulhi24, w); run("mad_i32_i24", k_mad24, w);
		run("mad_i64_i32", k_mad64, w); run("bfe_i32", k_bfe, w); run("alignbit", k_alignbit, w);
		run("dot2_i32_i16", k_dot2, w); run("dot4_i32_i8", k_dot4, w); run("lshl_add", k_lshl_add, w);
		run("add3", k_add3, w); run("pk_mul_lo_u16", k_pk_mul_lo, w); run("pk_mad_i16", k_pk_mad, w);
		run("pk_add_i16", k_pk_add, w); run("cndmask", k_cndmask, w); run("readlane+add", k_readlane, w);
		run("sub_u32", k_sub, w); run("and_b32", k_and, w); run("xor_b32", k_xor, w); run("lshlrev", k_lshl, w);
		run("ashrrev", k_ashr, w); run("lshrrev", k_lshr, w); run("mov", k_mov, w); run("max_i32", k_max, w);
		run("add_co", k_addco, w); run("cndmask_sgpr", k_cnd3, w); run("mul_u32_u24", k_mulu24, w); run("bfe_u32", k_bfeu, w);
		run("bfi", k_bfi, w); run("perm", k_perm, w); run("and_or", k_and_or, w); run("sub+ashr(2)", k_sub_ashr, w);
		run("add_sdwa", k_add_sdwa, w); run("mov_sdwa_sext", k_ashr_sdwa, w);
		run("add+mul24(2)", k_mix_add_mul, w); run("2add+mul24(3)", k_mix_2add_mul, w); run("add+bfe(2)", k_mix_add_bfe, w);
		run("mul24+bfe(2)", k_mix_mul_bfe, w);
		printf("\n");
	}
	return 0;
}
