// a2amd_fm.h - the FM oscillator units (src/units/fm.c of the reference) for
// gfx950.  Shared by the general kernel (one lane walks a voice's window) and
// the fm -> panmix leaf kernel (lane = voice).
//
// An FM unit is a recurrence in time as soon as any operator has feedback
// (fm_osc reads the operator's previous raw sine value, fm.c:111-123), so a voice
// is always evaluated sample after sample; the parallel axis is the voice.
//
// Arithmetic: bit for bit the reference's.  The two 64 bit products of fm_osc
//     fb  = (int64)last * fb.value >> 17        out = (int64)last * a.value >> 16
// have a 16 bit left operand (|last| <= 32767: a linear interpolation between
// two int16 table entries), so each is computed exactly from two 24 bit
// multiplies of 'last' with the halves of the gain:
//     last * g = (last * (g >> 16)) * 65536 + last * (g & 0xffff)
//     => last * g >> 16 = last * gh + (last * gl >> 16)          (gl >= 0)
// and the table lookup a2_Lerp(d, ph) = (d[i] * (256 - x) + d[i + 1] * x) >> 8
// = d[i] + ((d[i + 1] - d[i]) * x >> 8) reads one packed {d[i], d[i + 1] - d[i]}
// entry.  The sine table itself (libm sin()) is computed by the host exactly as
// fm_OpenState does (fm.c:493-501) and shipped as data.
#pragma once
#include "a2amd_dsp.h"
#include "a2amd_device.h"

struct FmOp {			// A2_fmosc, fm.c:84-93
	Ramp a, fb, p;
	int last_pitch;
	unsigned phase, dphase;
	int last;
};

DEV void fmop_load(FmOp &o, const int *w)
{
	o.a = ramp_load(w + FO_A); o.fb = ramp_load(w + FO_FB); o.p = ramp_load(w + FO_P);
	o.last_pitch = w[FO_LASTPITCH]; o.phase = (unsigned)w[FO_PHASE];
	o.dphase = (unsigned)w[FO_DPHASE]; o.last = w[FO_LAST];
}

DEV void fmop_store(int *w, const FmOp &o)
{
	ramp_store(w + FO_A, o.a); ramp_store(w + FO_FB, o.fb); ramp_store(w + FO_P, o.p);
	w[FO_LASTPITCH] = o.last_pitch; w[FO_PHASE] = (int)o.phase;
	w[FO_DPHASE] = (int)o.dphase; w[FO_LAST] = o.last;
}

// operators / oversampling bits / structure per unit kind.  fm.c never sees
// config.h, so A2_HIFI is not defined where it picks the oversampling
// (fm.c:36-52): 1x / 2x / 4x / 4x; fm3p, fm4p and fm4r run at fm3's rate and
// fm2r at fm2's (fm.c:280-322).
DEV int fm_nops(int kind)
{
	const int n = kind - A2D_FM1;	// fm1 fm2 fm3 fm4 fm3p fm4p fm2r fm4r
	return (0x42434321 >> (4 * n)) & 15;
}

// The divisions of a2_PrepareRamper are rare (a ramp in flight) and long
// (64 bit signed division): keep them out of line.
static __device__ __attribute__((noinline)) int fm_cold_delta64(int diff, int timer)
{
	return (int)(((int64_t)diff * 256) / timer);
}

static __device__ __attribute__((noinline)) int fm_cold_delta32(int diff, int frames)
{
	return diff / frames;
}

DEV void fm_ramp_prepare(Ramp &r, int frames)	// a2_PrepareRamper, a2_dsp.h:128-149
{
	if(!r.timer) {
		r.value = r.target;
		r.delta = 0;
	} else if(frames <= (r.timer >> 8)) {
		r.delta = fm_cold_delta64(wsub(r.target, r.value), r.timer);
		r.timer = wsub(r.timer, frames << 8);
	} else {
		r.delta = fm_cold_delta32(wsub(r.target, r.value), frames);
		r.timer = 0;
	}
}

// fm_run_pitch, fm.c:126-141: the pitch ramper only ever runs half a window
DEV void fm_run_pitch(const uint32_t *ptab, FmOp &o, int frames, int detune)
{
	fm_ramp_prepare(o.p, frames);
	ramp_run(o.p, frames >> 1);
	const int newpitch = wadd(o.p.value, detune) >> 8;
	if(newpitch != o.last_pitch) {
		o.dphase = p2i(ptab, newpitch);
		o.last_pitch = newpitch;
	}
}

// 24 bit multiply(-add) the compiler cannot strength-"reduce" back into the
// quarter-rate 32 bit v_mul_lo_u32
DEV int fm_mul24(int a, int b)
{
	int r;
	asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

DEV int fm_mad24(int a, int b, int c)
{
	int r;
	asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}

struct FmGain { int ah, al, fbh, fbl; };

DEV void fm_gain(FmGain &g, const FmOp &o)
{
	g.ah = o.a.value >> 16; g.al = o.a.value & 0xffff;
	g.fbh = o.fb.value >> 16; g.fbl = o.fb.value & 0xffff;
}

// fm_osc, fm.c:111-123.  sine[i] = (uint16)s[i] | (s[i + 1] - s[i]) << 16
DEV int fm_osc(FmOp &o, const FmGain &g, int mod, const uint32_t *sine)
{
	const int fb = (__mul24(o.last, g.fbh) + (__mul24(o.last, g.fbl) >> 16)) >> 1;
	const unsigned ph = o.phase + (unsigned)mod + (unsigned)fb;
	const uint32_t e = sine[(ph >> 13) & 2047u];
	const int x = (int)((ph >> 5) & 255u);
	o.last = (int)(int16_t)(e & 0xffffu) + (__mul24((int)e >> 16, x) >> 8);
	return fm_mad24(o.last, g.ah, fm_mul24(o.last, g.al) >> 16);
}

// fm_process (fm.c:194-233) in two pieces: the head of a window (rampers
// prepared for 'frames' frames, pitches run, per-subsample phase steps derived)
// and one output frame.  PAR: 0 chain, 1 parallel modulators, 2 ring modulator.
template<int NOPS, int OSBITS>
DEV void fm_prepare(FmOp (&op)[NOPS], const uint32_t *ptab, int frames, unsigned (&step)[NOPS], unsigned (&fix)[NOPS])
{
	int detune = 0;
#pragma unroll
	for(int i = 0; i < NOPS; ++i) {
		fm_ramp_prepare(op[i].a, frames);
		fm_ramp_prepare(op[i].fb, frames);
		fm_run_pitch(ptab, op[i], frames, detune);
		detune = op[0].p.value;
	}
#pragma unroll
	for(int i = 0; i < NOPS; ++i) {
		step[i] = op[i].dphase >> OSBITS;
		fix[i] = op[i].dphase & ((1u << OSBITS) - 1u);	// "Fix the rounding error buildup!"
	}
}

template<int NOPS, int OSBITS, int PAR>
DEV int fm_frame(FmOp (&op)[NOPS], const unsigned (&step)[NOPS], const unsigned (&fix)[NOPS], const uint32_t *sine)
{
	FmGain g[NOPS];
#pragma unroll
	for(int i = 0; i < NOPS; ++i)
		fm_gain(g[i], op[i]);
	int vsum = 0;
#pragma unroll 1
	for(int os = 0; os < (1 << OSBITS); ++os) {
		int v;
		if(PAR == 2) {			// fm_sample_rm, fm.c:172-192
			int v0, v1;
			if(NOPS == 2) {
				v0 = fm_osc(op[0], g[0], 0, sine);
				v1 = fm_osc(op[1], g[1], 0, sine);
			} else {
				v0 = fm_osc(op[0], g[0], fm_osc(op[NOPS > 2 ? 2 : 0], g[NOPS > 2 ? 2 : 0], 0, sine), sine);
				v1 = fm_osc(op[1], g[1], fm_osc(op[NOPS > 3 ? 3 : 0], g[NOPS > 3 ? 3 : 0], 0, sine), sine);
			}
			v = mul64s(v0, v1, 23);
		} else {			// fm_sample, fm.c:151-165
			v = 0;
#pragma unroll
			for(int i = NOPS - 1; i >= 0; --i) {
				if(i && PAR == 1)
					v = wadd(v, fm_osc(op[i], g[i], 0, sine));
				else
					v = fm_osc(op[i], g[i], v, sine);
			}
		}
#pragma unroll
		for(int i = 0; i < NOPS; ++i)
			op[i].phase += step[i];
		vsum = wadd(vsum, v);
	}
#pragma unroll
	for(int i = 0; i < NOPS; ++i) {
		op[i].a.value = wadd(op[i].a.value, op[i].a.delta);
		op[i].fb.value = wadd(op[i].fb.value, op[i].fb.delta);
		op[i].phase += fix[i];
	}
	return vsum >> OSBITS;
}

// One whole window for one voice; emit(s, value) receives frame s (0-based).
template<int NOPS, int OSBITS, int PAR, typename Emit>
DEV void fm_window(FmOp (&op)[NOPS], const uint32_t *ptab, const uint32_t *sine, int frames, Emit emit)
{
	unsigned step[NOPS], fix[NOPS];
	fm_prepare<NOPS, OSBITS>(op, ptab, frames, step, fix);
	for(int s = 0; s < frames; ++s)
		emit(s, fm_frame<NOPS, OSBITS, PAR>(op, step, fix, sine));
}

// fm_Initialize (fm.c:338-400) and the write callbacks (fm.c:403-483) on
// operators held in registers (the register index is a compile-time constant
// in every arm: no dynamic indexing of the operator array)
template<int NOPS>
DEV void fm_set_phase_ops(FmOp (&op)[NOPS], int ph, unsigned sst)
{
#pragma unroll
	for(int i = 0; i < NOPS; ++i) {
		const int ssph = (int)((unsigned)ph + ((sst * (op[i].dphase >> 8)) >> 8));
		op[i].phase = (unsigned)((int)((unsigned)ssph * 2048u) >> 8);
	}
}

template<int NOPS>
DEV void fm_init_ops(FmOp (&op)[NOPS], const uint32_t *ptab, int pitch, unsigned sst)
{
	Ramp p;
	ramp_init(p, pitch);
	const unsigned dphase = p2i(ptab, p.value >> 8);
#pragma unroll
	for(int i = 0; i < NOPS; ++i) {
		ramp_init(op[i].a, 0);
		ramp_init(op[i].fb, 0);
		op[i].p = p;
		op[i].last_pitch = 0;
		op[i].last = 0;
		op[i].dphase = dphase;
	}
	fm_set_phase_ops<NOPS>(op, 0, sst);
}

template<int NOPS>
DEV void fm_write_ops(FmOp (&op)[NOPS], int reg, int v, int start, int dur)
{
	if(reg == 0) {
		fm_set_phase_ops<NOPS>(op, v, (unsigned)start);
		return;
	}
#pragma unroll
	for(int i = 0; i < NOPS; ++i) {
		if(reg == 1 + 3 * i) ramp_set(op[i].p, v, start, dur);
		if(reg == 2 + 3 * i) ramp_set(op[i].a, v, start, dur);
		if(reg == 3 + 3 * i) ramp_set(op[i].fb, v, start, dur);
	}
}

// fm_set_phase, fm.c:327-335 on the state words of one unit
DEV void fm_set_phase_words(int *w, int nops, int ph, unsigned sst)
{
	for(int i = 0; i < nops; ++i) {
		int *o = w + i * FO_WORDS;
		const int ssph = (int)((unsigned)ph + ((sst * ((unsigned)o[FO_DPHASE] >> 8)) >> 8));
		o[FO_PHASE] = (int)((unsigned)ssph * 2048u) >> 8;
	}
}

// fm_Initialize, fm.c:338-400.  pitch = transpose + basepitch
DEV void fm_init_words(const uint32_t *ptab, int *w, int nops, int pitch, unsigned sst)
{
	for(int i = 0; i < A2D_FMSTATE; ++i)
		w[i] = 0;
	Ramp p;
	ramp_init(p, pitch);
	const unsigned dphase = p2i(ptab, p.value >> 8);
	for(int i = 0; i < nops; ++i) {
		ramp_store(w + i * FO_WORDS + FO_P, p);
		w[i * FO_WORDS + FO_DPHASE] = (int)dphase;
	}
	fm_set_phase_words(w, nops, 0, sst);
}

// the write callbacks, fm.c:403-483.  reg 0 = phase, then p a fb per operator;
// the host added transpose + basepitch to op 0's pitch
DEV void fm_write_words(int *w, int nops, int reg, int v, int start, int dur)
{
	if(reg == 0) {
		fm_set_phase_words(w, nops, v, (unsigned)start);
		return;
	}
	const int k = (reg - 1) % 3;
	int *rw = w + ((reg - 1) / 3) * FO_WORDS + (k == 0 ? FO_P : k == 1 ? FO_A : FO_FB);
	Ramp r = ramp_load(rw);
	ramp_set(r, v, start, dur);
	ramp_store(rw, r);
}
