// a2amd_wavecap.hip - SURVEY 8 f3: a wave that was RENDERED on the device stays there.
//
// a2_RenderWave (src/render.c:144-177) runs a program in an off-line one-channel substate and
// writes what the substate's driver delivers, 24 bit samples in int32, into a new wave through the
// wave's stream (a2_Write(A2_I24), render.c:91); closing the stream allocates the wave, converts
// (a2_do_write, src/waves.c:174-177: >> 8 into int16), fixes the pads (a2_fix_pad, waves.c:89-105)
// and derives the mip levels (a2_render_mipmaps, waves.c:107-130).  With the drop-in the substate's
// audio is rendered HERE: the kernels below keep channel 0 of every batch's master bus as it is
// produced (k_capture), and build the wave's pool region - level 0, pads, mip levels - from that
// copy (k_wave_level0 / k_wave_mip / k_wave_pad), so the samples never come back from the host.
#include <hip/hip_runtime.h>
#include "a2amd_device.h"

// channel 0 of a batch's master bus ([fragment][channel][64]), the fragments' frames closed up
__global__ __launch_bounds__(64)
void k_capture(const int32_t *__restrict__ bus, int32_t *__restrict__ dst, const uint32_t *__restrict__ fragpos, int nch)
{
	const int f = (int)blockIdx.x, lane = (int)threadIdx.x;
	const uint32_t p0 = fragpos[f], n = fragpos[f + 1] - p0;
	if((uint32_t)lane < n)
		dst[p0 + lane] = bus[((size_t)f * nch) * A2D_FRAG + lane];
}

// a2_do_write, waves.c:174-177 (A2_I24 -> int16: the shift, then C's conversion to int16_t)
__global__ __launch_bounds__(256)
void k_wave_level0(const int32_t *__restrict__ pcm, int16_t *__restrict__ d, unsigned size)
{
	const unsigned s = blockIdx.x * 256u + threadIdx.x;
	if(s < size)
		d[s] = (int16_t)(pcm[s] >> 8);
}

// a2_render_mipmaps, waves.c:121-127: d[s] = (2 sd[2s] + sd[2s - 1] + sd[2s + 1]) >> 2 (sd with its pads)
__global__ __launch_bounds__(256)
void k_wave_mip(const int16_t *__restrict__ sd, int16_t *__restrict__ d, unsigned size)
{
	const unsigned s = blockIdx.x * 256u + threadIdx.x;
	if(s < size) {
		const int16_t *q = sd + 2 * (size_t)s;
		d[s] = (int16_t)((((int)q[0] << 1) + q[-1] + q[1]) >> 2);
	}
}

// a2_fix_pad, waves.c:89-105.  d = the level's first pad sample; one workgroup.
__global__ __launch_bounds__(256)
void k_wave_pad(int16_t *__restrict__ d, unsigned size, int looped, int pre, int post)
{
	const int t = (int)threadIdx.x;
	if(looped && size) {
		// memcpy(d, d + size, PRE): the last PRE samples of the payload (or what follows them, for a payload
		// shorter than the pad: the reference copies whatever is there - first pass fills the post pad)
		for(int i = t; i < post; i += 256)
			d[pre + size + i] = d[pre + (unsigned)i % size];
		__syncthreads();
		for(int i = t; i < pre; i += 256)
			d[i] = d[size + i];
	} else {
		for(int i = t; i < pre; i += 256)
			d[i] = 0;
		for(int i = t; i < post; i += 256)
			d[pre + size + i] = 0;
	}
}

int a2d_launch_capture(const int32_t *bus, int32_t *dst, const uint32_t *fragpos, int nfrags, int nch, void *stream)
{
	if(nfrags <= 0)
		return 0;
	hipLaunchKernelGGL(k_capture, dim3(nfrags), dim3(64), 0, (hipStream_t)stream, bus, dst, fragpos, nch);
	return (int)hipGetLastError();
}

// pool: first pad sample of level 0; off[l]: offset of level l's first pad sample from there
int a2d_launch_wave_from_pcm(const int32_t *pcm, int16_t *pool, const uint32_t *off, const uint32_t *size, int levels, int looped,
		int pre, int post, void *stream)
{
	hipStream_t st = (hipStream_t)stream;
	if(size[0])
		hipLaunchKernelGGL(k_wave_level0, dim3((size[0] + 255) / 256), dim3(256), 0, st, pcm, pool + off[0] + pre, size[0]);
	hipLaunchKernelGGL(k_wave_pad, dim3(1), dim3(256), 0, st, pool + off[0], size[0], looped, pre, post);
	for(int l = 1; l < levels; ++l) {
		if(size[l])
			hipLaunchKernelGGL(k_wave_mip, dim3((size[l] + 255) / 256), dim3(256), 0, st, pool + off[l - 1] + pre,
					pool + off[l] + pre, size[l]);
		hipLaunchKernelGGL(k_wave_pad, dim3(1), dim3(256), 0, st, pool + off[l], size[l], looped, pre, post);
	}
	return (int)hipGetLastError();
}
