// lane_mapping.hip - A/B of the two ways to map wavetable voices onto a wavefront
// (the north-star sketched lane = voice, the product uses lane = frame).
//
// The same work both ways: N settled wtosc->panmix voices (Hermite interpolation of two taps
// from a table of precomputed coefficients, amplitude, two pan gains - the arithmetic of
// k_leaf_oscpan's settled path), F fragments of 64 frames, summed into one stereo bus.
//   A  lane = frame:  a wavefront takes V voices one after the other; its 64 lanes are the 64
//      frames of a fragment; every per-voice quantity is wave-uniform (scalar unit); the gather
//      of one voice is 64 near-consecutive addresses; the mix-down is a register add per lane.
//   B  lane = voice:  a wavefront takes 64 voices, one per lane, and walks the frames; every
//      per-voice quantity is a vector register; a gather is 64 unrelated addresses; the
//      mix-down is a butterfly reduction over the lanes for every frame and channel.
// Prints the time of both and checks that they produce the same bus.
//   hipcc --offload-arch=gfx950 -O3 -o lane_mapping lane_mapping.hip && ./lane_mapping [voices] [fragments]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)
#define WSIZE 2048	// wave length in samples (a power of two, like the built-in waves)
#define FCH 8		// fragments per register chunk (kernel A)

struct Voice { uint32_t dph; uint32_t ph_lo, ph_hi; int amp, v0, v1; };	// dph: 16:16 samples per frame, ph: 40:24

typedef int Coef3 __attribute__((ext_vector_type(3), aligned(4)));

__device__ __forceinline__ int hermite_c(const Coef3 k, unsigned ph)	// a2_Hermite from {a, b, c:d0}
{
	const int x = (int)((ph & 0xffu) << 7);
	int t = (int)((unsigned)k.x * (unsigned)x) >> 15;
	t = (int)((unsigned)(t + k.y) * (unsigned)x) >> 15;
	t = (int)((unsigned)(t + (k.z >> 16)) * (unsigned)x) >> 15;
	return (int)(int16_t)(k.z & 0xffff) + t;
}
__device__ __forceinline__ int mul64s(int a, int b, int sh) { return (int)(((int64_t)a * (int64_t)b) >> sh); }

// one output sample of a voice at 40:24 phase ph (already wrapped)
__device__ __forceinline__ void voice_sample(const int *__restrict__ coef, uint64_t ph, uint32_t dph, int amp, int v0, int v1,
		int &o0, int &o1)
{
	const unsigned pa = (unsigned)(ph >> 16), pb = pa + (dph >> 17);	// 24:8 tap positions (wtosc_Inter)
	const Coef3 ka = *(const Coef3 *)(coef + 3 * (size_t)((pa >> 8) & (2 * WSIZE - 1)));
	const Coef3 kb = *(const Coef3 *)(coef + 3 * (size_t)((pb >> 8) & (2 * WSIZE - 1)));
	const int x = mul64s(hermite_c(ka, pa) + hermite_c(kb, pb), amp, 17);
	o0 = mul64s(x, v0, 24);
	o1 = mul64s(x, v1, 24);
}

// ---- A: lane = frame -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lane_frame(const Voice *__restrict__ voices, int nvoices, int vpw, int nfrags,
		const int *__restrict__ coef, int *__restrict__ bus)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int first = (blockIdx.x * 4 + wv) * vpw;
	if(first >= nvoices)
		return;
	const int nv = min(vpw, nvoices - first);
	const uint64_t mask = ((uint64_t)WSIZE << 24) - 1;
	// (time slices over blockIdx.y as in the product: each takes a range of chunks)
	const int nchunks = (nfrags + FCH - 1) / FCH, per = (nchunks + gridDim.y - 1) / gridDim.y;
	for(int c = blockIdx.y * per; c < min(nchunks, (int)(blockIdx.y + 1) * per); ++c) {
		int acc0[FCH], acc1[FCH];
#pragma unroll
		for(int j = 0; j < FCH; ++j)
			acc0[j] = acc1[j] = 0;
		for(int v = 0; v < nv; ++v) {
			const Voice vc = voices[first + v];	// wave-uniform: scalar loads
			const uint64_t ph0 = ((uint64_t)vc.ph_lo | ((uint64_t)vc.ph_hi << 32)) + (uint64_t)(c * FCH * 64 + lane) * vc.dph;
#pragma unroll
			for(int j = 0; j < FCH; ++j) {
				int o0, o1;
				voice_sample(coef, (ph0 + (uint64_t)(j * 64) * vc.dph) & mask, vc.dph, vc.amp, vc.v0, vc.v1, o0, o1);
				acc0[j] += o0;
				acc1[j] += o1;
			}
		}
#pragma unroll
		for(int j = 0; j < FCH; ++j)
			if(c * FCH + j < nfrags) {
				atomicAdd(&bus[((size_t)(c * FCH + j) * 2) * 64 + lane], acc0[j]);
				atomicAdd(&bus[((size_t)(c * FCH + j) * 2 + 1) * 64 + lane], acc1[j]);
			}
	}
}

// ---- B: lane = voice -------------------------------------------------------------------
__device__ __forceinline__ int wave_sum(int x)
{
#pragma unroll
	for(int m = 32; m >= 1; m >>= 1)
		x += __shfl_xor(x, m, 64);
	return x;
}

__global__ __launch_bounds__(256) void k_lane_voice(const Voice *__restrict__ voices, int nvoices, int nfrags,
		const int *__restrict__ coef, int *__restrict__ bus)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int vi = (blockIdx.x * 4 + wv) * 64 + lane;
	const bool mine = vi < nvoices;
	Voice vc = { 0, 0, 0, 0, 0, 0 };
	if(mine)
		vc = voices[vi];
	const uint64_t mask = ((uint64_t)WSIZE << 24) - 1;
	const int per = (nfrags + gridDim.y - 1) / gridDim.y;
	const int f_lo = blockIdx.y * per, f_hi = min(nfrags, f_lo + per);
	uint64_t ph = ((uint64_t)vc.ph_lo | ((uint64_t)vc.ph_hi << 32)) + (uint64_t)f_lo * 64 * vc.dph;
	for(int f = f_lo; f < f_hi; ++f) {
		int keep0 = 0, keep1 = 0;
		for(int s = 0; s < 64; ++s) {
			int o0 = 0, o1 = 0;
			if(mine)
				voice_sample(coef, ph & mask, vc.dph, vc.amp, vc.v0, vc.v1, o0, o1);
			ph += vc.dph;
			const int s0 = wave_sum(o0), s1 = wave_sum(o1);	// the mix-down: over the lanes, per frame
			if(lane == s) {
				keep0 = s0;
				keep1 = s1;
			}
		}
		atomicAdd(&bus[((size_t)f * 2) * 64 + lane], keep0);
		atomicAdd(&bus[((size_t)f * 2 + 1) * 64 + lane], keep1);
	}
}

int main(int argc, char **argv)
{
	const int nvoices = argc > 1 ? atoi(argv[1]) : 65536, nfrags = argc > 2 ? atoi(argv[2]) : 256;
	std::vector<int16_t> wave(2 * WSIZE + 8);
	uint32_t r = 12345;
	for(int i = 0; i < WSIZE; ++i) {	// a band-limited-ish random wave, stored twice (the product's loop pads)
		r = r * 1664525u + 1013904223u;
		wave[i] = wave[i + WSIZE] = (int16_t)((int)(r >> 16) - 32768) / 2;
	}
	std::vector<int> coef(3 * (2 * WSIZE + 4));
	for(int i = 0; i < 2 * WSIZE; ++i) {
		const int dm = wave[(i + 2 * WSIZE - 1) % (2 * WSIZE)], d0 = wave[i], d1 = wave[(i + 1) % (2 * WSIZE)], d2 = wave[(i + 2) % (2 * WSIZE)];
		const int c = (d1 - dm) >> 1, a = (3 * (d0 - d1) + d2 - dm) >> 1, b = dm - d0 + c - a;
		coef[3 * i] = a;
		coef[3 * i + 1] = b;
		coef[3 * i + 2] = (int)(((unsigned)c << 16) | (uint16_t)d0);
	}
	std::vector<Voice> voices(nvoices);
	for(int k = 0; k < nvoices; ++k) {
		r = r * 1664525u + 1013904223u;
		voices[k].dph = 0x4000u + (r >> 15);		// 0.25 .. 2.25 samples per frame
		voices[k].ph_lo = r * 2654435761u;
		voices[k].ph_hi = (r >> 9) & 0x7;
		voices[k].amp = 4 << 16;
		voices[k].v0 = (int)(65536 * 256 * (0.3 + 0.4 * ((k % 17) / 16.0)));
		voices[k].v1 = (int)(65536 * 256 * (0.7 - 0.4 * ((k % 17) / 16.0)));
	}
	Voice *dv; int *dc, *busA, *busB;
	const size_t nbus = (size_t)nfrags * 2 * 64;
	CHECK(hipMalloc((void **)&dv, voices.size() * sizeof(Voice)));
	CHECK(hipMalloc((void **)&dc, coef.size() * sizeof(int)));
	CHECK(hipMalloc((void **)&busA, nbus * sizeof(int)));
	CHECK(hipMalloc((void **)&busB, nbus * sizeof(int)));
	CHECK(hipMemcpy(dv, voices.data(), voices.size() * sizeof(Voice), hipMemcpyHostToDevice));
	CHECK(hipMemcpy(dc, coef.data(), coef.size() * sizeof(int), hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const int vpw = 32, slices = 32;
	float best[2] = { 1e9f, 1e9f };
	for(int rep = 0; rep < 5; ++rep) {
		CHECK(hipMemset(busA, 0, nbus * sizeof(int)));
		CHECK(hipMemset(busB, 0, nbus * sizeof(int)));
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k_lane_frame, dim3((nvoices + 4 * vpw - 1) / (4 * vpw), slices), dim3(256), 0, 0, dv, nvoices, vpw, nfrags, dc, busA);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		best[0] = ms < best[0] ? ms : best[0];
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k_lane_voice, dim3((nvoices + 255) / 256, slices), dim3(256), 0, 0, dv, nvoices, nfrags, dc, busB);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		best[1] = ms < best[1] ? ms : best[1];
	}
	std::vector<int> a(nbus), b(nbus);
	CHECK(hipMemcpy(a.data(), busA, nbus * sizeof(int), hipMemcpyDeviceToHost));
	CHECK(hipMemcpy(b.data(), busB, nbus * sizeof(int), hipMemcpyDeviceToHost));
	size_t diff = 0, nz = 0;
	for(size_t i = 0; i < nbus; ++i) {
		diff += a[i] != b[i];
		nz += a[i] != 0;
	}
	const double vs = (double)nvoices * nfrags * 64;
	printf("{\"voices\": %d, \"fragments\": %d, \"lane_frame_ms\": %.3f, \"lane_voice_ms\": %.3f, \"lane_frame_voice_samples_per_s\": %.3g, "
			"\"lane_voice_voice_samples_per_s\": %.3g, \"lane_voice_over_lane_frame\": %.2f, \"bus_words_differing\": %zu, \"bus_words_nonzero\": %zu}\n",
			nvoices, nfrags, best[0], best[1], vs / (best[0] * 1e-3), vs / (best[1] * 1e-3), best[1] / best[0], diff, nz);
	return diff != 0;
}
