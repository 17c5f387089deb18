// a2amd_dsp.h - device-side fixed point primitives shared by the kernels.
//
// Bit-exact restatements of include/a2_dsp.h (rampers, Hermite interpolator,
// noise LCG) and src/pitch.c (a2_P2I) of the reference for gfx950: int32/int64
// fixed point, wrap-around via unsigned arithmetic, arithmetic >> on signed
// values, shift counts masked to 5 bits like the reference's x86 targets.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEV __device__ __forceinline__

// ---------------------------------------------------------------------------
// fixed point helpers
// ---------------------------------------------------------------------------
DEV int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
DEV int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
DEV int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
DEV int wshl(int a, int n) { return (int)((unsigned)a << n); }
DEV int mul64s(int a, int b, int sh) { return (int)(((int64_t)a * (int64_t)b) >> sh); }

struct Ramp { int value, target, delta, timer; };

DEV Ramp ramp_load(const int *w) { Ramp r = { w[0], w[1], w[2], w[3] }; return r; }
DEV void ramp_store(int *w, const Ramp &r) { w[0] = r.value; w[1] = r.target; w[2] = r.delta; w[3] = r.timer; }

// a2_InitRamper, a2_dsp.h:121-125
DEV void ramp_init(Ramp &r, int v) { r.value = r.target = wshl(v, 8); r.delta = r.timer = 0; }

// a2_PrepareRamper, a2_dsp.h:128-149
DEV void ramp_prepare(Ramp &r, int frames)
{
	if(!r.timer) {
		r.value = r.target;
		r.delta = 0;
	} else if(frames <= (r.timer >> 8)) {
		int64_t d = (int64_t)wsub(r.target, r.value);
		r.delta = (int)((d * 256) / r.timer);
		r.timer = wsub(r.timer, frames << 8);
	} else {
		r.delta = wsub(r.target, r.value) / frames;
		r.timer = 0;
	}
}

// a2_RunRamper, a2_dsp.h:152-155
DEV void ramp_run(Ramp &r, int frames) { r.value = wadd(r.value, wmul(r.delta, frames)); }

// a2_SetRamper, a2_dsp.h:161-170
DEV void ramp_set(Ramp &r, int target, int start, int duration)
{
	r.target = wshl(target, 8);
	r.timer = wadd(duration, start);
	if(r.timer < 256)
		r.value = r.target;
	else
		r.value = wadd(r.value, wmul(r.delta, start) >> 8);
}

// a2_P2I, pitch.c:57-67 (shift count mod 32 as on the reference's x86 targets)
DEV unsigned p2i(const uint32_t *tab, int pitch)
{
	int n = pitch & 0xffff;
	int oct = pitch >> 16;
	unsigned base = tab[2 * (n >> 10)], coeff = tab[2 * (n >> 10) + 1];
	unsigned dph = coeff * (unsigned)(n & 0x3ff);
	dph >>= 2;
	dph += base;
	return dph >> ((unsigned)(7 - oct) & 31u);
}

// a2_Noise, a2_dsp.h:37-42
DEV int noise_next(unsigned &st)
{
	st = st * 1566083941u + 1u;
	return (int)((st * (st >> 16)) >> 16);
}

DEV int mul24(int a, int b)
{
	int r;
	asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

// a2_Hermite, a2_dsp.h:64-74; d = first payload sample, ph 24:8
DEV int hermite(const int16_t *d, unsigned ph)
{
	int i = (int)(ph >> 8);
	const int frac = (int)(ph & 0xff);
	int dm = d[i - 1], d0 = d[i], d1 = d[i + 1], d2 = d[i + 2];
	int c = (d1 - dm) >> 1;
	int a = (3 * (d0 - d1) + d2 - dm) >> 1;
	int b = dm - d0 + c - a;
	// The reference multiplies by x = frac << 7 in 32 bit ints (wrap-around and all)
	// and shifts right by 15; with p = a * frac exact (|a| < 2^20, frac < 2^8) that is
	// bits 8..24 of p, sign extended: a full-rate 24 bit multiply and a bit-field
	// extract (spelled as instructions; the compiler does not prove the ranges)
	a = __builtin_amdgcn_sbfe(mul24(a, frac), 8, 17);
	a = __builtin_amdgcn_sbfe(mul24(wadd(a, b), frac), 8, 17);
	return d0 + __builtin_amdgcn_sbfe(mul24(wadd(a, c), frac), 8, 17);
}

// wtosc_Inter, A2_HIFI build (config.h:108), wtosc.c:28-33
DEV int inter(const int16_t *d, unsigned ph, unsigned dph)
{
	return hermite(d, ph) + hermite(d, ph + (dph >> 1));
}

