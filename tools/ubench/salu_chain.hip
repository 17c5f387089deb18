// salu_chain.hip - what one wavefront that has its SIMD to itself pays per DEPENDENT
// instruction on the scalar unit (the filter12 recurrence of k_leaf_recs, a2amd_fast.hip
// filt_window_s), on gfx950: chains of one instruction feeding itself, and the frame of
// independent ones for the issue rate; wall time from HIP events over 2 M instructions, one
// wave64 in the launch (s_memtime around scalar-only loops read back nonsense here), in
// cycles of the 2.4 GHz shader clock.
//   hipcc --offload-arch=gfx950 -O3 -o salu_chain salu_chain.hip && ./salu_chain
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITER 65536
#define REP 32

#define KERNEL(name, body) \
__global__ void name(int *out, int seed) { \
	int a = __builtin_amdgcn_readfirstlane(seed * 3 + 1), b = __builtin_amdgcn_readfirstlane(seed + 7); \
	int v = threadIdx.x + seed; \
	for(int i = 0; i < ITER; ++i) { \
		_Pragma("unroll") for(int r = 0; r < REP; ++r) { body } \
	} \
	out[threadIdx.x] = a ^ v; \
}

KERNEL(k_s_add, asm volatile("s_add_i32 %0, %0, %1" : "+s"(a) : "s"(b) : "scc");)
KERNEL(k_s_ashr, asm volatile("s_ashr_i32 %0, %0, 1" : "+s"(a) :: "scc");)
KERNEL(k_s_mul, asm volatile("s_mul_i32 %0, %0, %1" : "+s"(a) : "s"(b));)
KERNEL(k_s_mul_ashr_add, asm volatile("s_mul_i32 %0, %0, %1\n s_ashr_i32 %0, %0, 8\n s_add_i32 %0, %0, %1" : "+s"(a) : "s"(b) : "scc");)
KERNEL(k_v_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(v));)
KERNEL(k_v_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(v));)
KERNEL(k_readlane_add, asm volatile("v_readlane_b32 %0, %1, 3\n s_add_i32 %0, %0, %2" : "+s"(a) : "v"(v), "s"(b) : "scc");)
KERNEL(k_writelane, asm volatile("s_add_i32 %1, %1, %2\n v_writelane_b32 %0, %1, 5" : "+v"(v), "+s"(a) : "s"(b) : "scc");)
// independent scalar adds: the issue rate of a lone wavefront
KERNEL(k_s_add_indep, asm volatile("s_add_i32 s20, %0, %1\n s_add_i32 s21, %0, %1\n s_add_i32 s22, %0, %1\n s_add_i32 s23, %0, %1" :: "s"(a), "s"(b) : "s20", "s21", "s22", "s23", "scc");)

#define RUN(name, per) do { \
	float ms = 0; \
	hipLaunchKernelGGL(name, dim3(1), dim3(64), 0, 0, d_out, 1); \
	(void)hipEventRecord(e0, 0); \
	hipLaunchKernelGGL(name, dim3(1), dim3(64), 0, 0, d_out, 2); \
	(void)hipEventRecord(e1, 0); \
	(void)hipEventSynchronize(e1); \
	(void)hipEventElapsedTime(&ms, e0, e1); \
	printf("{\"chain\": \"%s\", \"instructions\": %d, \"ms\": %.3f, \"cycles_per_instruction_at_2.4GHz\": %.2f}\n", #name, ITER * REP * (per), ms, \
			ms * 2.4e6 / ((double)ITER * REP * (per))); \
} while(0)

int main()
{
	int *d_out;
	hipEvent_t e0, e1;
	(void)hipMalloc(&d_out, 64 * sizeof(int));
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	RUN(k_s_add, 1);
	RUN(k_s_ashr, 1);
	RUN(k_s_mul, 1);
	RUN(k_s_mul_ashr_add, 3);
	RUN(k_v_add, 1);
	RUN(k_v_mul_lo, 1);
	RUN(k_readlane_add, 2);
	RUN(k_writelane, 2);
	RUN(k_s_add_indep, 4);
	return 0;
}
