// valu_mix.hip - the issue rate of a shipped kernel's OWN instruction mix on gfx950.
//
// tools/isa_mix.py --kernel <k> --emit mix.inc reads the hot loop of <k> off the shipped
// liba2amd.so and writes its vector-ALU instructions as two inline-asm bodies:
//   MIX_BODY_KERNEL  verbatim: the kernel's registers, and with them its dependencies
//                    (loads, LDS, scalar instructions and waits left out: this is the
//                    VALU issue floor of the loop, not the loop);
//   MIX_BODY_INDEP   the same mnemonics and modifiers, operands renamed so that no
//                    instruction waits for another - the rate the SIMDs can issue THAT mix at.
// This file times both, at 1 .. 4 wavefronts per SIMD (the leaf kernels run at 3 - 4), and
// prints one JSON line: wave-instructions per ns chip-wide and T lane-ops/s (x 64 lanes).
//   hipcc --offload-arch=gfx950 -O3 -DMIX_INC='"variants/mix_osc2pan.inc"' -o valu_mix valu_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include MIX_INC

#define ITER 512

__global__ __launch_bounds__(256) void k_kernel(int *out, int n)
{
	for(int i = 0; i < n; ++i)
		asm volatile(MIX_BODY_KERNEL ::: MIX_CLOBBERS);
	if(n < 0)
		out[threadIdx.x] = n;
}

__global__ __launch_bounds__(256) void k_indep(int *out, int n)
{
	for(int i = 0; i < n; ++i)
		asm volatile(MIX_BODY_INDEP ::: MIX_CLOBBERS);
	if(n < 0)
		out[threadIdx.x] = n;
}

static double run(void (*k)(int *, int), int blocks, int *d)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, ITER);
	hipDeviceSynchronize();
	float best = 1e30f;
	for(int r = 0; r < 5; ++r) {
		hipEventRecord(e0);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, ITER);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		if(ms < best)
			best = ms;
	}
	return best;
}

int main()
{
	int *d;
	hipMalloc(&d, 1 << 20);
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	printf("{\"mix\": \"%s\", \"valu_per_trip\": %d, \"cus\": %d, \"rates\": {", MIX_NAME, MIX_NVALU, cus);
	for(int wps = 1; wps <= 4; ++wps) {
		// blocks of 4 wavefronts = one per SIMD; wps blocks per CU, many rounds
		const int blocks = cus * wps * 8;
		const double insts = (double)blocks * 4 * ITER * MIX_NVALU;
		const double tk = run(k_kernel, blocks, d), ti = run(k_indep, blocks, d);
		printf("%s\"%d\": {\"as_in_kernel_T\": %.2f, \"independent_T\": %.2f, \"as_in_kernel_ms\": %.4f, \"independent_ms\": %.4f}",
				wps > 1 ? ", " : "", wps, insts * 64 / (tk * 1e-3) / 1e12, insts * 64 / (ti * 1e-3) / 1e12, tk, ti);
	}
	printf("}}\n");
	return 0;
}
