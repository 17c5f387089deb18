// a2amd_kernels.hip - HIP kernels of the MI355X voice-render backend (gfx950).
//
// k_voices: the general kernel.  One wavefront renders whole voices; the 64
// lanes are the 64 sample frames of a fragment.  That orientation makes every
// per-voice parameter wave-uniform, turns the wavetable gather into 64
// near-consecutive reads, and makes the voice -> bus mix-down a plain per-lane
// add (wrap-around int32 adds commute, so any order is bit exact).  Units that
// are recurrences in time (filter12, fbdelay with short delays, noise) fall
// back to a few lanes running the sample loop.
//
// Arithmetic follows the reference bit for bit (see DESIGN.md "Arithmetic"):
// int32/int64 fixed point, wrap-around via unsigned ops, arithmetic >> on
// signed values, shift counts masked to 5 bits like x86.
#include <hip/hip_runtime.h>
#include "a2amd_device.h"

#include "a2amd_dsp.h"
#include "a2amd_fm.h"

// ---------------------------------------------------------------------------
// per-wavefront working set in LDS
// ---------------------------------------------------------------------------
struct WaveLDS {
	int scratch[A2D_MAXCH][A2D_FRAG];	// the voice's scratch bus (core.c:365-395)
	int otile[A2D_MAXCH][A2D_FRAG];		// pending adds into the output bus
	int ftmp[2][A2D_FRAG];			// filter12: the recurrence's results, before they are wired out
	int us[A2D_MAXCHAIN][A2D_USTATE];	// unit states of the current voice
	int cursor[A2D_MAXVPW];			// next unread record of each of our voices
};

struct Ctx {
	const A2DParams *p;
	WaveLDS *l;
	const uint32_t *sine;	// fm sine table in LDS (when the context has fm units)
	int lane;
	int frag;		// fragment index inside the batch
	int own_off, own_nch;
	unsigned omask;		// channels of otile that hold data
};

DEV void lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// write one output sample of unit 'desc' on channel ch, frame = lane
DEV void emit(Ctx &c, uint32_t desc, int ch, int val)
{
	if(A2D_WIRED(desc)) {
		c.l->otile[ch][c.lane] = wadd(c.l->otile[ch][c.lane], val);
	} else if(A2D_ADD(desc)) {
		c.l->scratch[ch][c.lane] = wadd(c.l->scratch[ch][c.lane], val);
	} else {
		c.l->scratch[ch][c.lane] = val;
	}
}

// ---------------------------------------------------------------------------
// wtosc
// ---------------------------------------------------------------------------
struct Osc {
	int mode, wave;
	unsigned dphase;
	uint64_t phase;
	int noise, p_ramping;
	Ramp p, a;
	unsigned seed;
};

DEV Osc osc_load(const int *w)
{
	Osc o;
	o.mode = w[OW_MODE]; o.wave = w[OW_WAVE]; o.dphase = (unsigned)w[OW_DPHASE];
	o.phase = (uint64_t)(unsigned)w[OW_PHASE_LO] | ((uint64_t)(unsigned)w[OW_PHASE_HI] << 32);
	o.noise = w[OW_NOISE]; o.p_ramping = w[OW_PRAMPING];
	o.p = ramp_load(w + OW_P); o.a = ramp_load(w + OW_A);
	o.seed = (unsigned)w[OW_SEED];
	return o;
}

DEV void osc_store(int *w, const Osc &o)
{
	w[OW_MODE] = o.mode; w[OW_WAVE] = o.wave; w[OW_DPHASE] = (int)o.dphase;
	w[OW_PHASE_LO] = (int)(unsigned)o.phase; w[OW_PHASE_HI] = (int)(unsigned)(o.phase >> 32);
	w[OW_NOISE] = o.noise; w[OW_PRAMPING] = o.p_ramping;
	ramp_store(w + OW_P, o.p); ramp_store(w + OW_A, o.a);
	w[OW_SEED] = (int)o.seed;
}

// wtosc_run_pitch, wtosc.c:89-105
DEV void osc_run_pitch(const A2DParams &p, Osc &o, int frames)
{
	ramp_prepare(o.p, frames);
	if(o.dphase && (!o.p.timer && !o.p_ramping))
		return;
	unsigned lastv = (unsigned)o.p.value;
	ramp_run(o.p, frames);
	o.p_ramping = o.p.delta;
	o.dphase = p2i(p.ptab, (int)((lastv + (unsigned)o.p.value) >> 9));
}

// wtosc_set_phase, wtosc.c:369-378
DEV void osc_set_phase(const A2DParams &p, Osc &o, int ph, unsigned sst)
{
	if(o.wave < 0) {
		o.phase = 0;
		return;
	}
	unsigned period = p.waves[o.wave].period;
	ph = (int)((unsigned)ph + ((sst * (o.dphase >> 8)) >> 8));
	o.phase = (uint64_t)(((int64_t)ph * (int64_t)period) * 256);
}

// wtosc_Wave, wtosc.c:433-483 ('value' is already a device wave id)
DEV void osc_set_wave(const A2DParams &p, Osc &o, int id)
{
	int wt = 0;
	o.wave = id;
	if(id >= 0) {
		wt = p.waves[id].type;
		if((wt == 2 || wt == 3) && p.waves[id].size[0] > (unsigned)A2D_WTOSC_MAXLENGTH)
			wt = 0;
	}
	switch(wt) {
	  default: o.wave = -1; o.mode = A2D_OSC_OFF; break;
	  case 1: o.mode = A2D_OSC_NOISE; break;
	  case 2: o.mode = A2D_OSC_WAVE; break;
	  case 3: o.mode = A2D_OSC_MIPWAVE; break;
	}
}

// The sample loop of wtosc_do_fragment (wtosc.c:200-236) with the optional
// per-sample loop/end test, evaluated for all frames of the window at once:
//   ph_k = ph0 + k*dph  (mod wsize<<24 when looped), a_k = a0 + k*delta.
// Returns the phase the sequential loop would have returned.
DEV uint64_t osc_window(Ctx &c, uint32_t desc, Osc &o, const int16_t *d,
		int offset, int frames, uint64_t ph, unsigned dph, int looped, unsigned wsize)
{
	int k = c.lane - offset;
	bool in = (k >= 0) && (k < frames);
	int kk = in ? k : 0;
	uint64_t phk = ph + (uint64_t)kk * dph;
	int ndone = frames;
	if(wsize) {
		if(looped) {
			phk %= (uint64_t)wsize << 24;
		} else {
			// first frame whose phase is past the end stops the loop
			bool over = in && ((phk >> 24) >= wsize);
			unsigned long long m = __ballot(over);
			if(m)
				ndone = (int)__ffsll((long long)m) - 1 - offset;
		}
	}
	if(in) {
		int val = 0;
		if(k < ndone) {
			int ak = wadd(o.a.value, wmul(o.a.delta, k));
			int v = inter(d, (unsigned)(phk >> 16), dph >> 16);
			val = mul64s(v, ak, 17);
			emit(c, desc, 0, val);
		} else if(!A2D_ADD(desc)) {
			emit(c, desc, 0, 0);
		}
	}
	// state after the loop
	ramp_run(o.a, ndone);
	if(wsize && looped) {
		if(frames > 0) {
			uint64_t last = (ph + (uint64_t)(frames - 1) * dph) % ((uint64_t)wsize << 24);
			return last + dph;
		}
		return ph;
	}
	return ph + (uint64_t)ndone * dph;
}

DEV void osc_zero(Ctx &c, uint32_t desc, int offset, int frames)
{
	int k = c.lane - offset;
	if(!A2D_ADD(desc) && k >= 0 && k < frames)
		emit(c, desc, 0, 0);
}

// wtosc_wavetable, wtosc.c:239-286
DEV void osc_mipwave(Ctx &c, uint32_t desc, Osc &o, int offset, int frames)
{
	const A2DParams &p = *c.p;
	const A2DWave &w = p.waves[o.wave];
	if(!w.size[0]) {
		// wtosc_check_unloaded, wtosc.c:168-183.  The reference leaves the
		// output buffer untouched in this window (the voice replays what
		// another voice left in the shared scratch bus); we render silence,
		// as in every following window.
		o.wave = -1;
		o.mode = A2D_OSC_OFF;
		osc_zero(c, desc, offset, frames);
		return;
	}
	osc_run_pitch(p, o, frames);
	unsigned dph = ((o.dphase + 255) >> 8) * w.period;
	ramp_prepare(o.a, frames);
	unsigned mm = 0;
	for(; (dph > (A2D_MAXPHINC << 8)) && (mm < A2D_MIPS - 1); ++mm)
		dph >>= 1;
	uint64_t ph = o.phase >> mm;
	dph = (unsigned)(((uint64_t)o.dphase * w.period) >> mm);
	if(w.flags & 0x100u) {
		ph %= (uint64_t)w.size[mm] << 24;
	} else if((ph >> 24) > (uint64_t)(w.size[mm] + 1)) {
		osc_zero(c, desc, offset, frames);
		return;
	}
	if(dph > (A2D_MAXPHINC << 16)) {
		osc_zero(c, desc, offset, frames);
		ph += (uint64_t)dph * (unsigned)frames;
		o.phase = ph << mm;
		ramp_run(o.a, frames);
	} else {
		o.phase = osc_window(c, desc, o, p.wavepool + w.off[mm], offset, frames,
				ph, dph, 0, 0) << mm;
	}
}

// wtosc_wavetable_no_mip, wtosc.c:301-358
DEV void osc_wave(Ctx &c, uint32_t desc, Osc &o, int offset, int frames)
{
	const A2DParams &p = *c.p;
	const A2DWave &w = p.waves[o.wave];
	if(!w.size[0]) {
		o.wave = -1;
		o.mode = A2D_OSC_OFF;
		osc_zero(c, desc, offset, frames);
		return;
	}
	const int16_t *d = p.wavepool + w.off[0];
	osc_run_pitch(p, o, frames);
	uint64_t dph = (uint64_t)o.dphase * w.period;
	ramp_prepare(o.a, frames);
	if(dph >> 32) {
		osc_zero(c, desc, offset, frames);
		o.phase += dph * (unsigned)frames;
		ramp_run(o.a, frames);
	} else if(dph > (A2D_MAXPHINC << 16)) {
		o.phase = osc_window(c, desc, o, d, offset, frames, o.phase, (unsigned)dph,
				(w.flags & 0x100u) ? 1 : 0, w.size[0]);
	} else {
		if(w.flags & 0x100u) {
			// wtosc.c:340 uses a 32 bit "size << 24"; a zero modulus
			// traps in the reference, we leave the phase alone.
			unsigned m = w.size[0] << 24;
			if(m)
				o.phase %= m;
		} else if((o.phase >> 24) > (uint64_t)(w.size[0] + 1)) {
			osc_zero(c, desc, offset, frames);
			return;
		}
		o.phase = osc_window(c, desc, o, d, offset, frames, o.phase, (unsigned)dph, 0, 0);
	}
}

// wtosc_noise, wtosc.c:129-152.  Frame k draws a new sample iff dphase >= 2^23
// or bit 23.. of the phase change between k and k+1; the draw index of a frame
// is a prefix count, the RNG a uniform loop of at most 'frames' steps.
DEV void osc_noise(Ctx &c, uint32_t desc, Osc &o, int offset, int frames)
{
	const A2DParams &p = *c.p;
	osc_run_pitch(p, o, frames);
	ramp_prepare(o.a, frames);
	int k = c.lane - offset;
	bool in = (k >= 0) && (k < frames);
	uint64_t phk = o.phase + (uint64_t)(in ? k : 0) * o.dphase;
	uint64_t nph = phk + o.dphase;
	bool draw = in && ((o.dphase >= (1u << 23)) || ((nph ^ phk) >> 23));
	unsigned long long m = __ballot(draw);
	// draws up to and including my frame
	unsigned long long below = (c.lane >= 63) ? ~0ull : ((2ull << c.lane) - 1ull);
	int mine = __popcll(m & below);
	int total = __popcll(m);
	int held = o.noise, myval = o.noise;
	unsigned st = o.seed;
	for(int j = 1; j <= total; ++j) {
		held = noise_next(st) - 32767;
		if(j == mine)
			myval = held;
	}
	o.seed = st;
	o.noise = held;
	if(in) {
		int ak = wadd(o.a.value, wmul(o.a.delta, k));
		emit(c, desc, 0, wmul(myval, ak >> 10) >> 6);
	}
	o.phase += (uint64_t)(unsigned)frames * o.dphase;
	ramp_run(o.a, frames);
}

DEV void osc_process(Ctx &c, uint32_t desc, int *w, int offset, int frames)
{
	Osc o = osc_load(w);
	switch(o.mode) {
	  case A2D_OSC_OFF:	// wtosc_Off[Add], wtosc.c:108-126
		ramp_prepare(o.p, frames);
		ramp_prepare(o.a, frames);
		ramp_run(o.p, frames);
		ramp_run(o.a, frames);
		osc_zero(c, desc, offset, frames);
		break;
	  case A2D_OSC_NOISE: osc_noise(c, desc, o, offset, frames); break;
	  case A2D_OSC_WAVE: osc_wave(c, desc, o, offset, frames); break;
	  case A2D_OSC_MIPWAVE: osc_mipwave(c, desc, o, offset, frames); break;
	}
	lds_sync();
	if(c.lane == 0)
		osc_store(w, o);
}

// ---------------------------------------------------------------------------
// panmix, panmix.c:49-249
// ---------------------------------------------------------------------------
DEV void panmix_process(Ctx &c, uint32_t desc, int *w, int offset, int frames)
{
	Ramp vol = ramp_load(w + PW_VOL), pan = ramp_load(w + PW_PAN);
	int nin = A2D_NIN(desc), nout = A2D_NOUT(desc);
	int k = c.lane - offset;
	bool in = (k >= 0) && (k < frames);
	if(nin == 1 && nout == 1) {		// panmix_process11
		ramp_prepare(vol, frames);
		if(in) {
			int vk = wadd(vol.value, wmul(vol.delta, k));
			emit(c, desc, 0, mul64s(c.l->scratch[0][c.lane], vk, 24));
		}
		ramp_run(vol, frames);
	} else {
		// clamp variant is chosen before the rampers are prepared,
		// panmix.c:117-135 / :171-189 / :231-249
		bool clamp = pan.target > 0xffffff || pan.target < -0xffffff ||
				pan.value > 0xffffff || pan.value < -0xffffff;
		ramp_prepare(vol, frames);
		ramp_prepare(pan, frames);
		if(in) {
			int vk = wadd(vol.value, wmul(vol.delta, k));
			int pk = wadd(pan.value, wmul(pan.delta, k));
			int vp = mul64s(pk, vk, 24);
			int v0 = wsub(vk, vp), v1 = wadd(vk, vp);
			if(clamp) {
				int lim = wshl(vk, 1);
				if(v0 > lim) v0 = lim;
				if(v1 > lim) v1 = lim;
			}
			int i0 = c.l->scratch[0][c.lane];
			if(nin == 1) {		// panmix_process12
				emit(c, desc, 0, mul64s(i0, v0, 24));
				emit(c, desc, 1, mul64s(i0, v1, 24));
			} else {
				int i1 = c.l->scratch[1][c.lane];
				if(nout == 1) {	// panmix_process21
					int64_t s = (int64_t)i0 * v0 + (int64_t)i1 * v1;
					emit(c, desc, 0, (int)(s >> 25));
				} else {	// panmix_process22
					emit(c, desc, 0, mul64s(i0, v0, 24));
					emit(c, desc, 1, mul64s(i1, v1, 24));
				}
			}
		}
		ramp_run(vol, frames);
		ramp_run(pan, frames);
	}
	lds_sync();
	if(c.lane == 0) {
		ramp_store(w + PW_VOL, vol);
		ramp_store(w + PW_PAN, pan);
	}
}

// ---------------------------------------------------------------------------
// filter12, filter12.c:74-119: a recurrence in time; lane ch runs channel ch
// ---------------------------------------------------------------------------
DEV void f12_process(Ctx &c, uint32_t desc, int *w, int offset, int frames)
{
	Ramp q = ramp_load(w + FW_Q);
	int lp = w[FW_LP], bp = w[FW_BP], hp = w[FW_HP];
	int f0 = w[FW_F1], df = 0, f1 = f0;
	int channels = A2D_NIN(desc);
	ramp_prepare(q, frames);
	if(w[FW_RAMP]) {		// host ran the cutoff ramper + f12_pitch2coeff
		f1 = w[FW_F1NEXT];
		df = wadd(wsub(f1, f0), frames >> 1) / frames;
	}
	if(c.lane < channels) {
		// the recurrence, one lane per channel.  Sixteen frames at a time: sixteen
		// inputs fetched together, sixteen steps in registers, sixteen results
		// stored - with one LDS read and a read-modify-write per step on the
		// chain, the LDS latency was most of a filtered voice's fragment.
		int ch = c.lane;
		int d1 = w[FW_D1A + ch], d2 = w[FW_D2A + ch];
		int qv = q.value;
		const int *in = c.l->scratch[ch];
		int *out = c.l->ftmp[ch];
		for(int s0 = offset; s0 < offset + frames; s0 += 16) {
			int x[16];
			const int nn = min(16, offset + frames - s0);
#pragma unroll
			for(int k = 0; k < 16; ++k)
				x[k] = in[min(s0 + k, A2D_FRAG - 1)];
#pragma unroll
			for(int k = 0; k < 16; ++k)
				if(k < nn) {
					int f = f0 >> 12;
					int qq = qv >> 12;
					int d1s = d1 >> 4;
					int l = wadd(d2, wmul(f, d1s) >> 8);
					int h = wsub(wsub(x[k] >> 5, l), wmul(qq, d1s) >> 8);
					int b = wadd(wmul(f, h >> 4) >> 8, d1);
					x[k] = wadd(wadd(wmul(l, lp), wmul(b, bp)), wmul(h, hp)) >> 3;
					d1 = b;
					d2 = l;
					f0 = wadd(f0, df);
					qv = wadd(qv, q.delta);
				}
#pragma unroll
			for(int k = 0; k < 16; ++k)
				if(k < nn)
					out[s0 + k] = x[k];
		}
		w[FW_D1A + ch] = d1;
		w[FW_D2A + ch] = d2;
	}
	lds_sync();
	// ... and every lane puts its frame's result where the unit's wiring says
	{
		const int k = c.lane - offset;
		if(k >= 0 && k < frames)
			for(int ch = 0; ch < channels; ++ch)
				emit(c, desc, ch, c.l->ftmp[ch][c.lane]);
	}
	ramp_run(q, frames);
	lds_sync();
	if(c.lane == 0) {
		ramp_store(w + FW_Q, q);
		w[FW_F1] = f1;
		w[FW_RAMP] = 0;
	}
}

// ---------------------------------------------------------------------------
// fbdelay, fbdelay.c:69-126
// ---------------------------------------------------------------------------
DEV void fbd_process(Ctx &c, uint32_t desc, int *w, int offset, int frames)
{
	const int M = A2D_FBD_BUFSIZE - 1;
	int fbdelay = w[DW_FBDELAY], ldelay = w[DW_LDELAY], rdelay = w[DW_RDELAY];
	int drygain = w[DW_DRYGAIN], fbgain = w[DW_FBGAIN], lgain = w[DW_LGAIN], rgain = w[DW_RGAIN];
	int bufpos = w[DW_BUFPOS];
	int *b0 = c.p->fbdmem + (size_t)w[DW_BUFIDX] * 2 * A2D_FBD_BUFSIZE;
	int *b1 = b0 + A2D_FBD_BUFSIZE;
	int stereoin = A2D_NIN(desc) == 2, stereoout = A2D_NOUT(desc) == 2;
	int in1ch = stereoin ? 1 : 0;
	// All taps reach back at least a whole window: no frame of this window
	// reads what another frame of it writes, so the frames are independent.
	bool par = fbdelay >= frames && ldelay >= frames && rdelay >= frames &&
			fbdelay < A2D_FBD_BUFSIZE - 64 && ldelay < A2D_FBD_BUFSIZE - 64 &&
			rdelay < A2D_FBD_BUFSIZE - 64;
	if(par) {
		int k = c.lane - offset;
		if(k >= 0 && k < frames) {
			int pos = wadd(bufpos, k);
			int i0 = c.l->scratch[0][c.lane], i1 = c.l->scratch[in1ch][c.lane];
			int o0 = mul64s(b1[(pos - fbdelay) & M], fbgain, 16);
			int o1 = mul64s(b0[(pos - fbdelay) & M], fbgain, 16);
			int t0 = b0[(pos - ldelay) & M], t1 = b1[(pos - rdelay) & M];
			b0[pos & M] = wadd(i0, o0);
			b1[pos & M] = wadd(i1, o1);
			o0 = wadd(o0, mul64s(t0, lgain, 16));
			o1 = wadd(o1, mul64s(t1, rgain, 16));
			o0 = wadd(o0, mul64s(i0, drygain, 16));
			o1 = wadd(o1, mul64s(i1, drygain, 16));
			if(stereoout) {
				emit(c, desc, 0, o0);
				emit(c, desc, 1, o1);
			} else
				emit(c, desc, 0, wadd(o0, o1) >> 1);
		}
	} else if(c.lane == 0) {
		for(int s = offset; s < offset + frames; ++s) {
			int pos = wadd(bufpos, s - offset);
			int i0 = c.l->scratch[0][s], i1 = c.l->scratch[in1ch][s];
			int o0 = mul64s(b1[(pos - fbdelay) & M], fbgain, 16);
			int o1 = mul64s(b0[(pos - fbdelay) & M], fbgain, 16);
			b0[pos & M] = wadd(i0, o0);
			b1[pos & M] = wadd(i1, o1);
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
			o0 = wadd(o0, mul64s(b0[(pos - ldelay) & M], lgain, 16));
			o1 = wadd(o1, mul64s(b1[(pos - rdelay) & M], rgain, 16));
			o0 = wadd(o0, mul64s(i0, drygain, 16));
			o1 = wadd(o1, mul64s(i1, drygain, 16));
			int *t0, *t1;
			if(A2D_WIRED(desc)) { t0 = &c.l->otile[0][s]; t1 = &c.l->otile[1][s]; }
			else { t0 = &c.l->scratch[0][s]; t1 = &c.l->scratch[1][s]; }
			bool acc = A2D_WIRED(desc) || A2D_ADD(desc);
			if(stereoout) {
				*t0 = acc ? wadd(*t0, o0) : o0;
				*t1 = acc ? wadd(*t1, o1) : o1;
			} else {
				int m = wadd(o0, o1) >> 1;
				*t0 = acc ? wadd(*t0, m) : m;
			}
		}
	}
	// delay-line stores of this window must be visible to the lanes that read
	// them in a later window
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	lds_sync();
	if(c.lane == 0)
		w[DW_BUFPOS] = wadd(bufpos, frames);
}

// ---------------------------------------------------------------------------
// inline (core.c:1763-1776) and xinsert bypass (xinsert.c:145-161)
// ---------------------------------------------------------------------------
DEV void inline_process(Ctx &c, uint32_t desc, int offset, int frames)
{
	// The children of this voice were rendered by earlier launches and
	// summed into our own bus; this is the point of the chain where their
	// signal enters.  Replacing mode = take it, adding mode = add it.  An
	// inline whose outputs are wired needs nothing: the children already
	// added straight into the bus beyond us.
	if(A2D_WIRED(desc) || c.own_off < 0)
		return;
	int k = c.lane - offset;
	if(k < 0 || k >= frames)
		return;
	const int *src = c.p->busmem + c.own_off + (size_t)c.frag * c.own_nch * A2D_FRAG;
	int nout = A2D_NOUT(desc);
	for(int ch = 0; ch < nout; ++ch) {
		int v = src[ch * A2D_FRAG + c.lane];
		if(A2D_ADD(desc))
			v = wadd(v, c.l->scratch[ch][c.lane]);
		c.l->scratch[ch][c.lane] = v;
	}
}

DEV void xinsert_process(Ctx &c, uint32_t desc, const int *w, int offset, int frames)
{
	int k = c.lane - offset;
	if(k < 0 || k >= frames)
		return;
	int n = A2D_NIN(desc);
	// clients (xinsert.c:60-142): READ-only ones are handed the input - it is
	// left in the slot's tap half for the host; WRITE-only ones were run by
	// the host, their summed output waits in the inject half
	const int slot = w[XW_SLOT], mode = w[XW_MODE];
	int *tap = nullptr;
	const int *inj = nullptr;
	if(slot) {
		int32_t *base = c.p->xio + (size_t)(slot - 1) * A2D_XIO_SLOT + (size_t)c.frag * 8 * A2D_FRAG + c.lane;
		if(mode & 1)
			tap = base;
		if(mode & 2)
			inj = base + A2D_XIO_HALF;
	}
	for(int ch = 0; ch < n; ++ch) {
		const int in = c.l->scratch[ch][c.lane];
		const int add = inj ? inj[ch * A2D_FRAG] : 0;
		if(tap)
			tap[ch * A2D_FRAG] = in;
		if(A2D_WIRED(desc))
			// (mode 4: there are insert clients - the input is not passed on,
			// xinsert.c:101-104,121-123; what they make of it arrives before the root
			// chain runs, a2amd_unit_insert)
			c.l->otile[ch][c.lane] = wadd(c.l->otile[ch][c.lane], (mode & 4) ? add : wadd(in, add));
		else if(A2D_ADD(desc))
			// in == out on the scratch bus: the client output lands in the
			// input before "out += in" doubles it (xinsert.c:113-124)
			c.l->scratch[ch][c.lane] = wshl(wadd(in, add), 1);
		else if(inj)
			c.l->scratch[ch][c.lane] = wadd(in, add);
	}
}

// xsink (xsink.c:27-46): every client is handed the inputs; no outputs
DEV void xsink_process(Ctx &c, uint32_t desc, const int *w, int offset, int frames)
{
	int k = c.lane - offset;
	if(k < 0 || k >= frames || !w[XW_SLOT] || !(w[XW_MODE] & 1))
		return;
	int32_t *tap = c.p->xio + (size_t)(w[XW_SLOT] - 1) * A2D_XIO_SLOT + (size_t)c.frag * 8 * A2D_FRAG + c.lane;
	for(int ch = 0, n = A2D_NIN(desc); ch < n; ++ch)
		tap[ch * A2D_FRAG] = c.l->scratch[ch][c.lane];
}

// xsource (xsource.c:43-137): the sum of what its clients produced, or silence
DEV void xsource_process(Ctx &c, uint32_t desc, const int *w, int offset, int frames)
{
	int k = c.lane - offset;
	if(k < 0 || k >= frames)
		return;
	const int32_t *inj = nullptr;
	if(w[XW_SLOT] && (w[XW_MODE] & 2))
		inj = c.p->xio + (size_t)(w[XW_SLOT] - 1) * A2D_XIO_SLOT + A2D_XIO_HALF +
				(size_t)c.frag * 8 * A2D_FRAG + c.lane;
	for(int ch = 0, n = A2D_NOUT(desc); ch < n; ++ch)
		emit(c, desc, ch, inj ? inj[ch * A2D_FRAG] : 0);
}

// ---------------------------------------------------------------------------
// dc, dc.c:56-134: a generator; every frame is a closed form of the window
// ---------------------------------------------------------------------------
DEV void dc_process(Ctx &c, uint32_t desc, int *w, int offset, int frames)
{
	Ramp v = ramp_load(w + CW_VALUE);
	const int nout = A2D_NOUT(desc);
	const int k = c.lane - offset;
	const bool in = (k >= 0) && (k < frames);
	int val;
	if(w[CW_MODE] == 0) {		// A2DCRM_STEP
		// [0, e2): value; e2: the half-way "transient" sample; then target
		int e2 = 0;
		const int value = v.value;
		bool trans;
		if(v.timer >= 256) {
			if((unsigned)(v.timer >> 8) >= (unsigned)frames) {
				e2 = frames;
				v.timer = wsub(v.timer, frames << 8);
			} else {
				e2 = v.timer >> 8;
				v.timer &= 0xff;
			}
		}
		trans = (v.timer < 256) && (e2 < frames);
		const int tv = wadd(wmul(v.value >> 4, v.timer), wmul(v.target >> 4, 256 - v.timer)) >> 4;
		if(trans) {
			v.timer = 0;
			v.value = v.target;
		}
		val = k < e2 ? value : (k == e2 && trans) ? tv : v.target;
	} else {			// A2DCRM_LINEAR
		ramp_prepare(v, frames);
		val = wadd(v.value, wmul(v.delta, k));
		ramp_run(v, frames);
	}
	if(in)
		for(int o = 0; o < nout; ++o)
			emit(c, desc, o, val);
	lds_sync();
	if(c.lane == 0)
		ramp_store(w + CW_VALUE, v);
}

// ---------------------------------------------------------------------------
// waveshaper, waveshaper.c:57-112 (fixed point branch): stateless per frame
// ---------------------------------------------------------------------------
DEV void waveshaper_process(Ctx &c, uint32_t desc, int *w, int offset, int frames)
{
	Ramp am = ramp_load(w + SW_AMOUNT);
	const int channels = A2D_NIN(desc);
	const int k = c.lane - offset;
	ramp_prepare(am, frames);
	if(k >= 0 && k < frames) {
		const int a = wadd(am.value, wmul(am.delta, k));
		const int a3p1 = wadd(wadd(wshl(a, 1), a), 1 << 24);
		const int asqr = (int)(((int64_t)(a >> 4) * (int64_t)(a >> 4)) >> 24);
		for(int ch = 0; ch < channels; ++ch) {
			const int v = c.l->scratch[ch][c.lane];
			const int vsqr = (int)(((int64_t)v * (int64_t)v) >> 22);
			int64_t vout = (int64_t)v * (int64_t)a3p1;
			const int64_t sqrsub = (int64_t)a * (int64_t)vsqr;
			vout = v >= 0 ? vout - sqrsub : vout + sqrsub;
			vout /= (((int64_t)asqr * (int64_t)vsqr) >> 16) + (1 << 24);
			emit(c, desc, ch, (int)vout);
		}
	}
	ramp_run(am, frames);
	lds_sync();
	if(c.lane == 0)
		ramp_store(w + SW_AMOUNT, am);
}

// ---------------------------------------------------------------------------
// dcblock, dcblock.c:66-95: a recurrence in time; lane ch runs channel ch
// ---------------------------------------------------------------------------
DEV void dcb_process(Ctx &c, uint32_t desc, int *w, int offset, int frames)
{
	const int channels = A2D_NIN(desc);
	if(c.lane < channels) {
		const int ch = c.lane;
		const int f = w[BW_F1] >> 12;
		int d1v = w[BW_D1A + ch], d2v = w[BW_D2A + ch];
		for(int s = offset; s < offset + frames; ++s) {
			const int d1 = d1v >> 4;
			const int l = wadd(d2v, wmul(f, d1) >> 8);
			const int h = wsub(wsub(c.l->scratch[ch][s] >> 5, l), wshl(d1, 4));
			const int b = wadd(wmul(f, h >> 4) >> 8, d1v);
			const int fout = wshl(h, 5);
			if(A2D_WIRED(desc))
				c.l->otile[ch][s] = wadd(c.l->otile[ch][s], fout);
			else if(A2D_ADD(desc))
				c.l->scratch[ch][s] = wadd(c.l->scratch[ch][s], fout);
			else
				c.l->scratch[ch][s] = fout;
			d1v = b;
			d2v = l;
		}
		w[BW_D1A + ch] = d1v;
		w[BW_D2A + ch] = d2v;
	}
}

// ---------------------------------------------------------------------------
// limiter, limiter.c:51-158: the peak follower is a recurrence over both
// channels; lane 0 runs it
// ---------------------------------------------------------------------------
DEV int iabs_w(int x) { return x < 0 ? (int)(0u - (unsigned)x) : x; }

DEV void limiter_process(Ctx &c, uint32_t desc, int *w, int offset, int frames)
{
	const int channels = A2D_NIN(desc);
	if(c.lane == 0) {
		const unsigned release = (unsigned)w[LW_RELEASE], threshold = (unsigned)w[LW_THRESHOLD];
		unsigned peak = (unsigned)w[LW_PEAK];
		const bool wired = A2D_WIRED(desc), acc = wired || A2D_ADD(desc);
		for(int s = offset; s < offset + frames; ++s) {
			const int i0 = c.l->scratch[0][s], i1 = channels == 2 ? c.l->scratch[1][s] : 0;
			unsigned p;
			if(channels == 1)
				p = (unsigned)iabs_w(i0);
			else {
				const int lp = iabs_w(i0), rp = iabs_w(i1);
				p = (unsigned)(lp > rp ? lp : rp);
				p = p + ((p - (unsigned)iabs_w(wsub(lp, rp))) >> 1);
			}
			if(p > peak)
				peak = p;
			else {
				peak -= release;
				if(peak < threshold)
					peak = threshold;
				p = peak;
			}
			const int gain = (int)((32767LL << 16) / (long long)((p + 511u) >> 9));
			for(int ch = 0; ch < channels; ++ch) {
				const int r = mul64s(ch ? i1 : i0, gain, 16);
				int *t = wired ? &c.l->otile[ch][s] : &c.l->scratch[ch][s];
				*t = acc ? wadd(*t, r) : r;
			}
		}
		w[LW_PEAK] = (int)peak;
	}
}

// ---------------------------------------------------------------------------
// fm1..fm4r, fm.c:194-322: a recurrence in time (operator feedback); lane 0
// runs the window with the operators in registers
// ---------------------------------------------------------------------------
template<int NOPS, int OSBITS, int PAR>
static __device__ __attribute__((noinline)) void fm_run(int *fw, const uint32_t *ptab, const uint32_t *sine,
		int *dst, int frames, bool acc, bool writer)
{
	FmOp op[NOPS];
#pragma unroll
	for(int i = 0; i < NOPS; ++i)
		fmop_load(op[i], fw + i * FO_WORDS);
	fm_window<NOPS, OSBITS, PAR>(op, ptab, sine, frames, [&](int s, int v) {
		const int nv = acc ? wadd(dst[s], v) : v;
		if(writer)
			dst[s] = nv;
	});
	if(writer) {
#pragma unroll
		for(int i = 0; i < NOPS; ++i)
			fmop_store(fw + i * FO_WORDS, op[i]);
	}
}

DEV void fm_process(Ctx &c, uint32_t desc, int *w, int offset, int frames)
{
	// lanes 1..15 shadow lane 0 (same data, no stores): a wavefront with fewer
	// than 16 busy lanes runs this chain at half speed (see fmpan_body)
	if(c.lane < 16) {
		const bool writer = c.lane == 0;
		int *fw = c.p->fmstate + (size_t)w[MW_SLOT] * A2D_FMSTATE;
		const bool wired = A2D_WIRED(desc);
		int *dst = (wired ? c.l->otile[0] : c.l->scratch[0]) + offset;
		const bool acc = wired || A2D_ADD(desc);
		const uint32_t *pt = c.p->ptab;
		switch(A2D_KIND(desc)) {
		  case A2D_FM1: fm_run<1, 0, 0>(fw, pt, c.sine, dst, frames, acc, writer); break;
		  case A2D_FM2: fm_run<2, 1, 0>(fw, pt, c.sine, dst, frames, acc, writer); break;
		  case A2D_FM3: fm_run<3, 2, 0>(fw, pt, c.sine, dst, frames, acc, writer); break;
		  case A2D_FM4: fm_run<4, 2, 0>(fw, pt, c.sine, dst, frames, acc, writer); break;
		  case A2D_FM3P: fm_run<3, 2, 1>(fw, pt, c.sine, dst, frames, acc, writer); break;
		  case A2D_FM4P: fm_run<4, 2, 1>(fw, pt, c.sine, dst, frames, acc, writer); break;
		  case A2D_FM2R: fm_run<2, 1, 2>(fw, pt, c.sine, dst, frames, acc, writer); break;
		  case A2D_FM4R: fm_run<4, 2, 2>(fw, pt, c.sine, dst, frames, acc, writer); break;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	}
}

// ---------------------------------------------------------------------------
// control writes and unit init (uniform; lane 0 commits)
// ---------------------------------------------------------------------------
DEV void unit_init(const A2DParams &p, uint32_t desc, int *w, const A2DRec &r)
{
	for(int i = 0; i < A2D_USTATE; ++i)
		w[i] = 0;
	switch(A2D_KIND(desc)) {
	  case A2D_WTOSC: {	// wtosc_Initialize, wtosc.c:390-423; value = transpose + basepitch
		Osc o = osc_load(w);
		o.wave = -1;
		o.mode = A2D_OSC_OFF;
		ramp_init(o.a, 0);
		ramp_init(o.p, r.value);
		o.dphase = p2i(p.ptab, o.p.value >> 8);
		osc_store(w, o);
		break;
	  }
	  case A2D_PANMIX: {	// panmix_Initialize, panmix.c:252-284
		Ramp vol, pan;
		ramp_init(vol, 65536);
		ramp_init(pan, 0);
		ramp_store(w + PW_VOL, vol);
		ramp_store(w + PW_PAN, pan);
		break;
	  }
	  case A2D_FILTER12: {	// f12_Initialize, filter12.c:180-221; value = f1 from the host
		Ramp q;
		ramp_init(q, 0);
		ramp_set(q, 32768, 0, 0);	// f12_Q(u, 0, 0, 0)
		ramp_store(w + FW_Q, q);
		w[FW_LP] = 65536 >> 8;
		w[FW_F1] = r.value;
		break;
	  }
	  case A2D_FBDELAY: {	// fbdelay_Initialize, fbdelay.c:170-220; value = buffer index
		int sr = p.samplerate;
		w[DW_FBDELAY] = (int)((int64_t)(400 << 16) * sr / 65536000);
		w[DW_LDELAY] = (int)((int64_t)(280 << 16) * sr / 65536000);
		w[DW_RDELAY] = (int)((int64_t)(320 << 16) * sr / 65536000);
		w[DW_DRYGAIN] = 65536;
		w[DW_FBGAIN] = 16384;
		w[DW_LGAIN] = 32768;
		w[DW_RGAIN] = 32768;
		w[DW_BUFIDX] = r.value;
		break;
	  }
	  case A2D_INLINE:
		break;
	  case A2D_XINSERT: case A2D_XSINK: case A2D_XSOURCE:
		w[XW_SLOT] = w[XW_MODE] = 0;
		break;
	  case A2D_DC: {	// dc_Initialize, dc.c:160-188
		Ramp v;
		ramp_init(v, 0);
		ramp_store(w + CW_VALUE, v);
		w[CW_MODE] = 1;		// A2DCRM_LINEAR
		break;
	  }
	  case A2D_WAVESHAPER: {	// waveshaper_Initialize, waveshaper.c:131-156
		Ramp v;
		ramp_init(v, 0);
		ramp_store(w + SW_AMOUNT, v);
		break;
	  }
	  case A2D_DCBLOCK:	// dcb_Initialize, dcblock.c:119-153; value = f1 from the host
		w[BW_F1] = r.value;
		break;
	  case A2D_LIMITER:	// limiter_Initialize, limiter.c:168-198; value = release from the host
		w[LW_RELEASE] = r.value;
		w[LW_THRESHOLD] = (1 << 16) << 8;
		w[LW_PEAK] = 32768 << 8;
		break;
	  default:	// fm: value = transpose + basepitch, start = wake fraction, dur = pool slot
		w[MW_SLOT] = (int)r.dur;
		fm_init_words(p.ptab, p.fmstate + (size_t)r.dur * A2D_FMSTATE, fm_nops(A2D_KIND(desc)),
				r.value, r.start & 0xffu);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
		break;
	}
}

DEV void unit_write(const A2DParams &p, uint32_t desc, int *w, const A2DRec &r)
{
	int reg = A2D_RREG(r.head), v = r.value;
	int start = (int)r.start, dur = (int)r.dur;
	switch(A2D_KIND(desc)) {
	  case A2D_WTOSC: {
		Osc o = osc_load(w);
		switch(reg) {
		  case 0: osc_set_wave(p, o, v); break;
		  case 1:	// wtosc_Pitch, wtosc.c:486-492 (host added transpose + basepitch)
			ramp_set(o.p, v, start, dur);
			if(!dur)
				o.p_ramping = 1;
			break;
		  case 2: ramp_set(o.a, v, start, dur); break;
		  case 3: osc_set_phase(p, o, v, (unsigned)start); break;
		}
		osc_store(w, o);
		break;
	  }
	  case A2D_PANMIX: {
		Ramp rr = ramp_load(w + (reg ? PW_PAN : PW_VOL));
		ramp_set(rr, v, start, dur);
		ramp_store(w + (reg ? PW_PAN : PW_VOL), rr);
		break;
	  }
	  case A2D_FILTER12:
		switch(reg) {
		  case 1: {	// f12_Q, filter12.c:149-162 (host did the 1/v)
			Ramp q = ramp_load(w + FW_Q);
			ramp_set(q, v, start, dur);
			ramp_store(w + FW_Q, q);
			break;
		  }
		  case 2: w[FW_LP] = v >> 8; break;
		  case 3: w[FW_BP] = v >> 8; break;
		  case 4: w[FW_HP] = v >> 8; break;
		}
		break;
	  case A2D_FBDELAY:	// fbdelay.c:231-267
		if(reg < 3)
			w[DW_FBDELAY + reg] = (int)((int64_t)v * p.samplerate / 65536000);
		else if(reg < 7)
			w[DW_FBDELAY + reg] = v;
		break;
	  case A2D_INLINE:
		break;
	  case A2D_XINSERT: case A2D_XSINK: case A2D_XSOURCE:	// a2amd_unit_clients: reg 0 = slot + 1, reg 1 = mode
		w[reg ? XW_MODE : XW_SLOT] = v;
		break;
	  case A2D_DC:
		if(reg == 0) {		// dc_Value, dc.c:191-215
			Ramp rr = ramp_load(w + CW_VALUE);
			if(w[CW_MODE] == 0) {
				rr.target = wshl(v, 8);
				rr.timer = (int)((unsigned)dur >> 1) - start;
				if(rr.timer <= 0) {
					rr.value = rr.target;
					rr.timer = 0;
				}
			} else
				ramp_set(rr, v, start, dur);
			ramp_store(w + CW_VALUE, rr);
		} else			// dc_Mode, dc.c:218-238
			w[CW_MODE] = (v >> 16) == 1 ? 1 : 0;
		break;
	  case A2D_WAVESHAPER: {	// waveshaper_Amount, waveshaper.c:159-162
		Ramp rr = ramp_load(w + SW_AMOUNT);
		ramp_set(rr, v, start, dur);
		ramp_store(w + SW_AMOUNT, rr);
		break;
	  }
	  case A2D_DCBLOCK:	// dcb_CutOff, dcblock.c:112-117 (the host did pitch -> coefficient)
		w[BW_F1] = v;
		break;
	  case A2D_LIMITER:	// limiter.c:201-213 (the host did the scaling)
		w[reg ? LW_THRESHOLD : LW_RELEASE] = v;
		break;
	  default:	// fm.c:403-483
		fm_write_words(p.fmstate + (size_t)w[MW_SLOT] * A2D_FMSTATE, fm_nops(A2D_KIND(desc)),
				reg, v, start, dur);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
		break;
	}
}

// add the pending output tile into the bus it belongs to and clear it
DEV void flush_otile(Ctx &c, int out_off, int out_nch)
{
	if(out_off < 0)
		return;
	lds_sync();
	int *dst = c.p->busmem + out_off + (size_t)c.frag * out_nch * A2D_FRAG;
	for(int ch = 0; ch < out_nch; ++ch) {
		int v = c.l->otile[ch][c.lane];
		if(v)
			atomicAdd(&dst[ch * A2D_FRAG + c.lane], v);
		c.l->otile[ch][c.lane] = 0;
	}
}

// a voice record with the whole chain in it (this kernel renders chains of up to A2D_MAXCHAIN units)
struct VoiceFull {
	int32_t nunits;
	int32_t unit[A2D_MAXCHAIN];
	int32_t out_off, out_nch, own_off, own_nch;
};
DEV VoiceFull load_voice(const A2DParams &p, int slot)
{
	const A2DVoice b = p.voices[slot];
	VoiceFull v;
	v.nunits = b.nunits;
#pragma unroll
	for(int k = 0; k < A2D_MAXCHAIN; ++k)
		v.unit[k] = k < A2D_CHAIN_INLINE ? b.unit[k] : 0;
	if(b.nunits > A2D_CHAIN_INLINE && p.vext) {
		const A2DVoiceExt e = p.vext[slot];
#pragma unroll
		for(int k = A2D_CHAIN_INLINE; k < A2D_MAXCHAIN; ++k)
			v.unit[k] = e.unit[k - A2D_CHAIN_INLINE];
	}
	v.out_off = b.out_off; v.out_nch = b.out_nch; v.own_off = b.own_off; v.own_nch = b.own_nch;
	return v;
}

DEV void process_window(Ctx &c, const VoiceFull &v, int offset, int frames)
{
	for(int u = 0; u < v.nunits; ++u) {
		uint32_t desc = c.p->udesc[v.unit[u]];
		int *w = c.l->us[u];
		switch(A2D_KIND(desc)) {
		  case A2D_WTOSC: osc_process(c, desc, w, offset, frames); break;
		  case A2D_PANMIX: panmix_process(c, desc, w, offset, frames); break;
		  case A2D_FILTER12: f12_process(c, desc, w, offset, frames); break;
		  case A2D_FBDELAY: fbd_process(c, desc, w, offset, frames); break;
		  case A2D_INLINE: inline_process(c, desc, offset, frames); break;
		  case A2D_XINSERT: xinsert_process(c, desc, w, offset, frames); break;
		  case A2D_XSINK: xsink_process(c, desc, w, offset, frames); break;
		  case A2D_XSOURCE: xsource_process(c, desc, w, offset, frames); break;
		  case A2D_DC: dc_process(c, desc, w, offset, frames); break;
		  case A2D_WAVESHAPER: waveshaper_process(c, desc, w, offset, frames); break;
		  case A2D_DCBLOCK: dcb_process(c, desc, w, offset, frames); break;
		  case A2D_LIMITER: limiter_process(c, desc, w, offset, frames); break;
		  default: fm_process(c, desc, w, offset, frames); break;
		}
		lds_sync();
	}
}

#define WAVES_PER_BLOCK 4

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void k_voices(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int vpw)
{
	__shared__ WaveLDS lds[WAVES_PER_BLOCK];
	__shared__ uint32_t fmsine[2048];
	const A2DParams &p = *pp;
	if(p.fmstate) {		// the context has fm units
		for(int i = threadIdx.x; i < 2048; i += 64 * WAVES_PER_BLOCK)
			fmsine[i] = p.fmsine[i];
		__syncthreads();
	}
	const int wv = threadIdx.x >> 6;
	const int lane = threadIdx.x & 63;
	const int gw = blockIdx.x * WAVES_PER_BLOCK + wv;
	const int first = gw * vpw;
	if(first >= nlist)
		return;
	const int last = min(first + vpw, nlist);
	Ctx c;
	c.p = pp;
	c.l = &lds[wv];
	c.lane = lane;
	c.sine = fmsine;
	c.omask = 0;
	for(int ch = 0; ch < A2D_MAXCH; ++ch) {
		c.l->otile[ch][lane] = 0;
		c.l->scratch[ch][lane] = 0;
	}
	if(lane < A2D_MAXVPW)
		c.l->cursor[lane] = 0;
	lds_sync();

	// One voice per wavefront (the usual shape below a few thousand voices): its unit
	// states, its record cursor and its liveness stay in LDS / registers for the whole
	// batch instead of travelling to memory and back for every fragment - on a
	// plumbing-sized scene (a song: a few dozen voices, four nesting levels) those
	// round trips were most of a fragment's time.
	const bool resident = last - first == 1;
	for(int f = 0; f < p.nfrags; ++f) {
		const int nframes = p.fragframes[f];
		int cur_off = -1, cur_nch = 0;
		c.frag = f;
		for(int vi = first; vi < last; ++vi) {
			const int slot = list[vi];
			const VoiceFull v = load_voice(p, slot);
			const A2DRun run = p.runs[slot];
			if(v.out_off != cur_off) {
				flush_otile(c, cur_off, cur_nch);
				cur_off = v.out_off;
				cur_nch = v.out_nch;
			}
			c.own_off = v.own_off;
			c.own_nch = v.own_nch;
			// unit states -> LDS
			if(!resident || f == 0) {
				for(int u = 0; u < v.nunits; ++u)
					if(lane < A2D_USTATE)
						c.l->us[u][lane] = p.ustate[(size_t)v.unit[u] * A2D_USTATE + lane];
				lds_sync();
			}
			int active = (!resident || f == 0) ? p.vactive[slot] : c.l->cursor[A2D_MAXVPW - 1];

			// records of this fragment: the run is sorted by fragment;
			// skip what earlier fragments consumed
			int r0 = run.first + c.l->cursor[vi - first], r1 = run.first + run.count;
			bool explicit_ = (r0 < r1) && ((int)A2D_RFRAG(p.recs[r0].head) == f);
			if(!explicit_) {
				// no records: the engine called Process(0, frames)
				// once on every unit (core.c:1875-1876)
				if(active)
					process_window(c, v, 0, nframes);
			} else {
				for(; r0 < r1 && (int)A2D_RFRAG(p.recs[r0].head) == f; ++r0) {
					const A2DRec r = p.recs[r0];
					const int u = A2D_RUNIT(r.head);
					switch(A2D_ROP(r.head)) {
					  case R_SEG:
						if(active)
							process_window(c, v, r.dur & 0xffff, r.dur >> 16);
						break;
					  case R_INIT:
						if(lane == 0)
							unit_init(p, p.udesc[v.unit[u]], c.l->us[u], r);
						active = 1;
						break;
					  case R_WRITE:
						if(lane == 0)
							unit_write(p, p.udesc[v.unit[u]], c.l->us[u], r);
						break;
					  case R_F1SET:
						if(lane == 0) {
							c.l->us[u][FW_F1] = r.value;
							c.l->us[u][FW_RAMP] = 0;
						}
						break;
					  case R_F1RAMP:
						if(lane == 0) {
							c.l->us[u][FW_F1NEXT] = r.value;
							c.l->us[u][FW_RAMP] = 1;
						}
						break;
					  case R_NOISESEED:
						if(lane == 0)
							c.l->us[u][OW_SEED] = r.value;
						break;
					  case R_KILL:
						active = 0;
						break;
					}
					lds_sync();
				}
			}
			// unit states -> memory
			lds_sync();
			if(!resident || f == p.nfrags - 1) {
				for(int u = 0; u < v.nunits; ++u)
					if(lane < A2D_USTATE)
						p.ustate[(size_t)v.unit[u] * A2D_USTATE + lane] = c.l->us[u][lane];
				if(lane == 0)
					p.vactive[slot] = active;
			}
			if(lane == 0) {
				c.l->cursor[vi - first] = r0 - run.first;
				if(resident)
					c.l->cursor[A2D_MAXVPW - 1] = active;	// (a one-voice wavefront uses cursor[0] only)
			}
			lds_sync();
		}
		flush_otile(c, cur_off, cur_nch);
	}
}

int a2d_launch_voices(const A2DParams *dparams, const int *dlist, int nlist, int vpw, void *stream)
{
	if(nlist <= 0)
		return 0;
	if(vpw < 1)
		vpw = 1;
	if(vpw > A2D_MAXVPW)
		vpw = A2D_MAXVPW;
	int nwaves = (nlist + vpw - 1) / vpw;
	int nblocks = (nwaves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
	hipLaunchKernelGGL(k_voices, dim3(nblocks), dim3(64 * WAVES_PER_BLOCK), 0,
			(hipStream_t)stream, dparams, dlist, nlist, vpw);
	return (int)hipGetLastError();
}
