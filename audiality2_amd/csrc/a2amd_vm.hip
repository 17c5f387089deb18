// a2amd_vm.hip - the scripted voice's VM on the device (include/a2amd_vm.h).
//
// k_vm: one LANE per adopted voice.  A voice's VM is a serial program over its own 64 registers
// (a2_VoiceProcessVM, src/core.c:1166-1744) that runs a handful of instructions every few
// milliseconds: lane = voice is its natural shape (nothing is wave-uniform, everything is
// independent), and what the kernel moves is small - the voice's state (A2DVmVoice, 0.4 KB) in and
// out once per batch, the program text from the cache.  Two passes per batch over the same
// interpreter (a2amd_vmcore.h, the source the host's copy is compiled from):
//
//   count   runs every voice through the batch on a private copy of its state, counts the records
//           it would emit, and gives each voice a place in the batch's record array: an exclusive
//           prefix sum over the wavefront and one atomic add per wavefront on the running total
//   emit    (after the host has made sure the total fits) runs them again and writes the records -
//           R_WRITE / R_F1SET / R_F1RAMP / R_SEG exactly as the host's recorder makes them of the
//           engine's calls - stores the advanced state, and points runs[voice] at the records: the
//           leaf kernels launched next execute them like host records.
//
// HBM traffic per voice and batch: 2 x 0.4 KB state + 16 B per record, against 504-808 B per
// voice and FRAGMENT of the leaf kernels: the VM is noise next to the rendering it controls.
#include <hip/hip_runtime.h>
#include "a2amd_device.h"
#include "a2amd_vmcore.h"
#include "a2amd_vmdev.h"

using namespace a2vm;

#define VM_TPB 64		// one wavefront per workgroup: voices diverge, a short block retires early

__global__ __launch_bounds__(VM_TPB)
void k_vm_count(A2DVmParams vp)
{
	const int i = (int)(blockIdx.x * VM_TPB + threadIdx.x), lane = (int)(threadIdx.x & 63);
	__shared__ VmSlot s_v[VM_TPB];
	__shared__ uint32_t s_code[VM_CODEWORDS];
	int n = 0, fault = 0;
	A2DVmVoice &v = s_v[threadIdx.x].v;
	if(i < vp.n)
		v = vp.vmv[vp.list[i]];
	const uint32_t *code = vm_stage_code(vp.code, v, i < vp.n, s_code);
	if(i < vp.n) {
		const Consts K = { vp.msdur, vp.samplerate, vp.basepitch, vp.ptab, vp.f1tab, vp.envlut };
		CountE e = { {}, 0 };
		const uint8_t *ff = vp.fragframes, *fb = vp.fragbase;
		run_batch(v, code, K, e, vp.now, 0, vp.nfrags, [ff, fb](int f) { return (unsigned)ff[f] | ((unsigned)fb[f] << 8); },
				&s_v[threadIdx.x].rt);
		n = e.n;
		fault = v.fault != 0;
	}
	// exclusive prefix over the wavefront (DPP-free: six shuffle steps once per launch)
	int incl = n;
#pragma unroll
	for(int d = 1; d < 64; d <<= 1) {
		const int t = __shfl_up(incl, d, 64);
		if(lane >= d)
			incl += t;
	}
	const int wave_total = __shfl(incl, 63, 64);
	unsigned base = 0;
	if(lane == 63 && wave_total)
		base = atomicAdd(vp.total, (unsigned)wave_total);
	base = (unsigned)__shfl((int)base, 63, 64);
	if(i < vp.n) {
		A2DRun r = { (int)(base + (unsigned)(incl - n)), n };
		vp.vmrun[i] = r;
	}
	const unsigned long long fb = __ballot(fault);
	if(lane == 0 && fb)
		atomicAdd(vp.total + 1, (unsigned)__popcll(fb));
}

__global__ __launch_bounds__(VM_TPB)
void k_vm_emit(A2DVmParams vp)
{
	const int i = (int)(blockIdx.x * VM_TPB + threadIdx.x);
	__shared__ VmSlot s_v[VM_TPB];
	__shared__ uint32_t s_code[VM_CODEWORDS];
	const bool has = i < vp.n;
	const int slot = has ? vp.list[i] : 0;
	A2DVmVoice &v = s_v[threadIdx.x].v;
	if(has)
		v = vp.vmv[slot];
	const uint32_t *code = vm_stage_code(vp.code, v, has, s_code);
	if(!has)
		return;
	const A2DRun place = vp.vmrun[i];
	const Consts K = { vp.msdur, vp.samplerate, vp.basepitch, vp.ptab, vp.f1tab, vp.envlut };
	A2DRun run = { 0, 0 };
	if((unsigned)place.first + (unsigned)place.count <= vp.rec_cap) {
		StoreE e = { {}, vp.recs + vp.rec_base + place.first, 0 };
		const uint8_t *ff = vp.fragframes, *fb = vp.fragbase;
		run_batch(v, code, K, e, vp.now, 0, vp.nfrags, [ff, fb](int f) { return (unsigned)ff[f] | ((unsigned)fb[f] << 8); },
				&s_v[threadIdx.x].rt);
		if(e.n) {
			run.first = (int)(vp.rec_base + (unsigned)place.first);
			run.count = e.n;
		}
		vp.vmv[slot] = v;
	}
	// (no room - the host checked: never - leaves the voice where it was, silent on the VM's side)
	vp.runs[v.voice] = run;
}

// The pool room k_vm_win will need for the batch AFTER this one: every voice run, on a copy, through the stretch of
// time the host expects that batch to cover, in fragments of 64 frames - PoolE counts the windows that begin inside
// one.  A window begins where the voice's VM wakes up (run_batch), whatever the fragments are: the count is a bound
// for any batch over the same stretch whose fragments are these or pieces of these (the engine cuts a fragment
// where a group's own VM wakes up, core.c:1852-1878), which vm_issue checks (a2amd_vm.cpp).  out[0] += entries.
// (Launched beside the render pass, read when the next batch is issued.)
__global__ __launch_bounds__(VM_TPB)
void k_vm_pool(A2DVmParams vp, unsigned *out)
{
	const int i = (int)(blockIdx.x * VM_TPB + threadIdx.x), lane = (int)(threadIdx.x & 63);
	__shared__ VmSlot s_v[VM_TPB];
	__shared__ uint32_t s_code[VM_CODEWORDS];
	int n = 0;
	A2DVmVoice &v = s_v[threadIdx.x].v;
	if(i < vp.n)
		v = vp.vmv[vp.list[i]];
	const uint32_t *code = vm_stage_code(vp.code, v, i < vp.n, s_code);
	if(i < vp.n) {
		const Consts K = { vp.msdur, vp.samplerate, vp.basepitch, vp.ptab, vp.f1tab, vp.envlut };
		PoolE e = { {}, 0, 0, 0 };
		const uint8_t *ff = vp.fragframes, *fb = vp.fragbase;
		run_batch(v, code, K, e, vp.now, 0, vp.nfrags, [ff, fb](int f) { return (unsigned)ff[f] | ((unsigned)fb[f] << 8); },
				&s_v[threadIdx.x].rt);
		n = e.pool;
	}
#pragma unroll
	for(int d = 32; d > 0; d >>= 1)
		n += __shfl_xor(n, d, 64);
	if(lane == 0 && n)
		atomicAdd(out, (unsigned)n);
}

int a2d_launch_vm_pool(const A2DVmParams &vp, unsigned *out, void *stream)
{
	if(vp.n <= 0)
		return 0;
	hipLaunchKernelGGL(k_vm_pool, dim3((vp.n + VM_TPB - 1) / VM_TPB), dim3(VM_TPB), 0, (hipStream_t)stream, vp, out);
	return (int)hipGetLastError();
}

int a2d_launch_vm(const A2DVmParams &vp, int emit, void *stream)
{
	if(vp.n <= 0)
		return 0;
	const int nblocks = (vp.n + VM_TPB - 1) / VM_TPB;
	if(emit)
		hipLaunchKernelGGL(k_vm_emit, dim3(nblocks), dim3(VM_TPB), 0, (hipStream_t)stream, vp);
	else
		hipLaunchKernelGGL(k_vm_count, dim3(nblocks), dim3(VM_TPB), 0, (hipStream_t)stream, vp);
	return (int)hipGetLastError();
}
