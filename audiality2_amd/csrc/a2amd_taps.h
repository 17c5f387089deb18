// a2amd_taps.h - what the lane = frame kernels share: wave-uniform <-> per-lane moves, the two
// Hermite tap forms (four fetched samples / a precomputed coefficient entry), the coefficient
// table's buffer descriptor, exact small divisions, the chunk's bus sums.
// (a2amd_fast.hip: the quiet kernels and k_leaf_recs; a2amd_win.hip: the window kernels)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "a2amd_device.h"
#include "a2amd_dsp.h"
#include "a2amd_fm.h"

#define FAST_WPB   4		// wavefronts per workgroup
#define FAST_FCH   A2D_FAST_FCH	// fragments whose bus sums stay in registers at a time

DEV int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }
DEV int rdl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// park a wave-uniform value back in lane 'sel' of a per-lane register
#define WRL(reg, val) reg = me ? (val) : reg

// a2_Hermite (a2_dsp.h:64-74) on four already fetched samples.  The operands of
// the three products are below 2^20 and 2^15 in magnitude, so the 24 bit
// multiplier returns the same low 32 bits as the reference's int multiply
// (including its wrap-around) at full rate.
DEV int hermite4(int dm, int d0, int d1, int d2, int frac)
{
	int c = (d1 - dm) >> 1;
	int a = (3 * (d0 - d1) + d2 - dm) >> 1;
	int b = dm - d0 + c - a;
	// The reference multiplies by x = frac << 7 in 32 bit ints (wrap-around and
	// all) and shifts right by 15.  With p = a * frac exact (|a| < 2^20, frac <
	// 2^8), the low 32 bits of p << 7 are p's bits 0..24 moved up, so that result
	// is bits 8..24 of p, sign extended: one full-rate 24 bit multiply and one
	// bit-field extract, no x.  (Spelled as instructions: left to itself the
	// compiler proves the 24 bit range for one of the three products only.)
	a = __builtin_amdgcn_sbfe(fm_mul24(a, frac), 8, 17);
	a = __builtin_amdgcn_sbfe(fm_mul24(a + b, frac), 8, 17);
	return d0 + __builtin_amdgcn_sbfe(fm_mul24(a + c, frac), 8, 17);
}

// Four consecutive int16 samples as two dwords from a 2-byte aligned address
// (gfx950 global loads take any byte alignment in the HSA configuration).
struct __attribute__((packed, aligned(2))) Quad16 { uint32_t lo, hi; };

// Both taps of wtosc_Inter (wtosc.c:28-33): window A = d[i-1 .. i+2] for the tap
// at ph, window B = d[i'-1 .. i'+2] for the tap at ph + (dph16 >> 1), i' being i
// or i + 1 (dph16 <= 512 << 8).  Each window is one 8 byte load.
DEV int inter_quads(const Quad16 qa, const Quad16 qb, unsigned ph, unsigned ph2)
{
	int h0 = hermite4((int16_t)(qa.lo & 0xffff), (int)qa.lo >> 16, (int16_t)(qa.hi & 0xffff), (int)qa.hi >> 16,
			(int)(ph & 0xff));
	int h1 = hermite4((int16_t)(qb.lo & 0xffff), (int)qb.lo >> 16, (int16_t)(qb.hi & 0xffff), (int)qb.hi >> 16,
			(int)(ph2 & 0xff));
	return h0 + h1;
}

// wtosc_Inter at 24:8 phase ph from wave data d (first payload sample)
DEV int inter_dwords(const int16_t *d, unsigned ph, unsigned dph16)
{
	const unsigned ph2 = ph + (dph16 >> 1);
	const Quad16 qa = *(const Quad16 *)(d + (int)(ph >> 8) - 1);
	const Quad16 qb = *(const Quad16 *)(d + (int)(ph2 >> 8) - 1);
	return inter_quads(qa, qb, ph, ph2);
}

// Hermite coefficients of one window, precomputed per wave sample when the wave is
// uploaded (k_build_coef): a2_Hermite's a, b and - packed - c (high half) and d[i]
// (low half), one 12 byte entry per wave sample (A2D_COEF_WORDS 4: c and d[i] apart).  The settled paths fetch one entry per tap instead of four samples and go
// straight into the three multiply-shift-add steps, which are the reference's own
// (a2_dsp.h:64-74: x = frac << 7, 32 bit wrap-around products, arithmetic >> 15) - no
// unpacking of samples, no deriving a, b, c for every output frame.
// The table is read through ONE buffer descriptor (stride = entry, indexed): the entry index
// phase >> 8 goes to the load as it is and the hardware scales it - no address
// arithmetic in the vector unit; the level's first payload sample is the load's scalar
// byte offset.  (Round 2 added a byte offset
// (phase >> 8) * 12 to a scalar base: one more vector instruction per tap.)
#if A2D_COEF_WORDS == 4
typedef int Coef4 __attribute__((ext_vector_type(4)));		// a, b, c, d0
#define A2D_COEF_LOAD "llvm.amdgcn.struct.buffer.load.v4i32"
#else
typedef int Coef4 __attribute__((ext_vector_type(3)));		// a, b, c:d0 (the halves come apart inside the adds: SDWA)
#define A2D_COEF_LOAD "llvm.amdgcn.struct.buffer.load.v3i32"
#endif
typedef int CoefRsrc __attribute__((ext_vector_type(4)));
extern "C" __device__ Coef4 a2d_coef_load(CoefRsrc rsrc, int vindex, int voffset, int soffset, int aux)
		__asm(A2D_COEF_LOAD);

// the descriptor of the coefficient table (gfx9 buffer resource: base, stride in word 1
// bits 16-29, no swizzle, num_records unlimited, DATA_FORMAT 32)
DEV CoefRsrc coef_rsrc(const int *wavecoef)
{
	const uint64_t b = (uint64_t)wavecoef;
	CoefRsrc r;
	r.x = rfl((int)(unsigned)b);
	r.y = rfl((int)(((unsigned)(b >> 32) & 0xffffu) | ((4u * A2D_COEF_WORDS) << 16)));
	r.z = -1;
	r.w = 0x00020000;
	return r;
}

// byte offset of the entry of pool sample doff (the table follows the pool: host, a2amd_host.cpp)
DEV int coef_base(unsigned doff) { return (int)(doff * (4u * A2D_COEF_WORDS)); }

// a2_Hermite's x = frac << 7 from a 24:8 phase, one instruction: the shift takes the low byte
// of its operand (SDWA) - the compiler spells it as a shift and a mask
DEV int frac_x(unsigned ph)
{
#ifdef A2D_NO_SDWA_X
	return (int)((ph & 0xffu) << 7);
#else
	int x;
	const int seven = 7;
	asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
			: "=v"(x) : "v"(seven), "v"(ph));
	return x;
#endif
}

DEV int hermite_c(const Coef4 k, unsigned ph)
{
	const int x = frac_x(ph);
	int t = wmul(k.x, x) >> 15;
	t = wmul(wadd(t, k.y), x) >> 15;
#if A2D_COEF_WORDS == 4
	t = wmul(wadd(t, k.z), x) >> 15;
	return wadd(k.w, t);
#else
	t = wmul(wadd(t, k.z >> 16), x) >> 15;
	return wadd((int)(int16_t)(k.z & 0xffff), t);
#endif
}

// wtosc_Inter (wtosc.c:28-33) from the coefficient table: cb = coef_base() of the
// level's first payload sample, ph = 24:8 phase of the tap
DEV Coef4 coef_at(const CoefRsrc rs, int cb, unsigned ph)
{
	return a2d_coef_load(rs, (int)(ph >> 8), 0, cb, 0);
}

// The 24:8 phase of a tap: (ph + lane * dph) >> 16 for a wave-uniform 64 bit phase ph and
// ldph = lane * dph (below 2^31: dph <= A2D_MAXPHINC << 16 on the settled paths).  Split at
// bit 16 the sum needs no 64 bit vector arithmetic: the halves of ph stay scalar, the lane
// part is two adds and a shift of the fast issue class (a 64 bit multiply-add and a funnel
// shift otherwise).
DEV unsigned tap_phase(uint64_t ph, unsigned ldph)
{
	const unsigned lo = (unsigned)ph & 0xffffu, hi = (unsigned)(ph >> 16);
	return hi + ((lo + ldph) >> 16);
}

// lane * dph once per voice and oscillator (opaque to the compiler, which otherwise folds it
// back into a multiply-add per fragment)
DEV unsigned lane_dph(int lane, unsigned dph)
{
	unsigned t = (unsigned)lane * dph;
	asm("" : "+v"(t));
	return t;
}

// trunc(n / d) for |n| < 2^52, 0 < d < 2^31, exactly: the quotient of the correctly rounded double
// division is at most one off, and the remainder says which way (the compiler's 64 bit integer
// division is a few hundred instructions; a scripted voice needs one per ramping control and window)
DEV int64_t div_trunc_exact(int64_t n, int d)
{
	int64_t q = (int64_t)((double)n / (double)d);
	const int64_t r = n - q * d;
	if(n >= 0) {
		if(r < 0) --q; else if(r >= d) ++q;
	} else {
		if(r > 0) ++q; else if(r <= -(int64_t)d) --q;
	}
	return q;
}


// add the register sums of a chunk of fragments into the bus and clear them
template<int N>
DEV void flush_acc(int *busmem, int off, int nch, int f0, int nf, int lane, int dbg,
		int (&acc0)[N], int (&acc1)[N])
{
#pragma unroll
	for(int j = 0; j < N; ++j) {
		if(j < nf && off >= 0 && !(dbg & 1)) {
			int *dst = busmem + off + (size_t)(f0 + j) * nch * A2D_FRAG;
			if(acc0[j])
				atomicAdd(&dst[lane], acc0[j]);
			if(acc1[j])
				atomicAdd(&dst[A2D_FRAG + lane], acc1[j]);
		}
		acc0[j] = 0;
		acc1[j] = 0;
	}
}

