// a2amd_winctl.h - the control side of the window kernels (a2amd_win.hip): the slot / entry layout, a voice's
// control state carried lane = voice (oscillators, filter12, panmix: what the records change and what a window
// of the chain advances), the staging area of a control wavefront and the wavefront that carries it out.
// Shared by k_win_ctl (records made by the host or by k_vm_emit) and k_vm_win (a2amd_vmwin.hip: the device VM
// driving the same state without records in between).
#ifndef A2AMD_WINCTL_H
#define A2AMD_WINCTL_H
#include "a2amd_device.h"
#include "a2amd_dsp.h"
#include "a2amd_taps.h"

// Layout: one SLOT per (fragment, list position) - wslot[((f - fa) * nlist + i) * SW] - holds the
// fragment's first window (the only one, for most voices in most fragments); further windows of the
// same fragment ("extras": a script's sub-fragment windows) go to the voice's run of the pool wext, in
// order (A2D_WIN_WORDS apart whatever the class: one pool for all), widx[(f - fa) * nlist + i] naming the first.  What 64 control lanes write per fragment is one
// contiguous block, and a render wavefront reads its voices' slots of a fragment the same way: LANE =
// VOICE, a few 16-byte loads each, asked for a fragment ahead - the words then come out of the vector
// registers with v_readlane when the voice's turn comes (what the quiet kernels do with voice state).
// (The first cuts fetched entries through the scalar cache: 1.5 million 96-byte entries per batch
// is not what that cache is for - the render pass ran at a third of this.)
//
// slot / entry words: head, panmix (4), oscillators (6 each), then - filter classes - filter12 (7);
// SW words per class, a multiple of 4
enum { WE_HEAD = 0,	// off | len << 6 | clamp << 13 | fresh << 14 | mode0 << 15 | mode1 << 17 | extras << 19
	WE_VOL, WE_DVOL, WE_PAN, WE_DPAN,			// panmix: values at the window's first frame, per-frame deltas
	WE_OSC = 5 };						// 6 words per oscillator (below)
enum { WF_F0 = 0, WF_DF, WF_QV, WF_QD, WF_LP, WF_BP, WF_HP };	// filter12, from word 5 + 6 * NOSC: coefficient + step, q + step, mix levels
#define WIN_NW(NOSC)       (5 + 6 * (NOSC))			/* words the oscillator / pan stages read */
#define WIN_SW(NOSC, FILT) ((FILT) ? ((NOSC) == 1 ? 20 : 24) : ((NOSC) == 1 ? 12 : 20))
enum { WO_A = 0, WO_B, WO_C, WO_DPH, WO_AK, WO_DA };		//   taps:  level offset in the pool, phase lo / hi, increment
								//   noise: seed, held sample, phase lo, increment; then amplitude + delta
enum { WM_SILENT = 0, WM_TAPS, WM_NOISE };
#define WH_OFF(h)    ((int)((h) & 63u))
#define WH_LEN(h)    ((int)(((h) >> 6) & 127u))
#define WH_CLAMP     (1u << 13)
#define WH_FRESH     (1u << 14)
#define WH_MODE(h, o) (((h) >> (15 + 2 * (o))) & 3u)
#define WH_EXTRAS(h) ((int)(((h) >> 19) & 127u))
// Round 6: the window every voice has in most fragments - every oscillator on taps with its amplitude at rest, volume
// and pan at rest (classes without filter12); whole fragments and cut windows alike - is marked, and WE_VOL / WE_PAN hold the two
// GAINS of panmix_process12 (panmix.c:84-95: v0 = vol - vp, v1 = vol + vp, vp = pan * vol >> 24, clamped where the
// head says so) instead of volume and pan: the render pass takes such an entry through a straight line - no lane
// predicate, no mode or ramp branches, no 64-bit products on the scalar unit (k_win_render: 96 vector + 114 scalar
// instructions per window in the general path, a2amd_win.hip)
#define WH_PLAIN     (1u << 26)
// ... and in the filter classes, whose pan stage is a step of its own and keeps volume and pan as they are: the
// OSCILLATOR part of such a window is plain (whole fragment, taps, amplitudes at rest) - k_win_render_f's workers
#define WH_PLAINOSC  (1u << 27)

// (a uniform address through the constant address space is a scalar load)
typedef int Int4 __attribute__((ext_vector_type(4)));
typedef int Int8 __attribute__((ext_vector_type(8)));
DEV int sload(const void *p)
{
	typedef const __attribute__((address_space(4))) int *CI;
	return *(CI)(uintptr_t)p;
}
DEV Int8 sload8(const void *p)
{
	typedef const __attribute__((address_space(4))) Int8 *CI8;
	return *(CI8)(uintptr_t)p;
}

// ---------------------------------------------------------------------------
// the control pass: lane = voice
// ---------------------------------------------------------------------------
struct OscV {		// A2_wtosc (wtosc.c:66-80), one per lane
	int mode, wave;
	unsigned dphase;
	uint64_t phase;
	int p_ramping;
	Ramp p, a;
	int noise;
	unsigned seed;
	// what the oscillator reads of its wave (A2_wave, a2_waves.h:88-103), kept while the wave stays: the
	// descriptor's period, flags and level 0 size (the host rewrites descriptors between batches only), and
	// the mip level last played (cmm: -1 none yet, -2 the descriptor itself has not been read)
	unsigned wperiod, wflags, wsize0;
	int cmm;
	unsigned csize, coff;
};

DEV void osc_wave_fields(const A2DWave *waves, OscV &o)
{
	o.cmm = -1;
	o.wperiod = o.wflags = o.wsize0 = o.csize = o.coff = 0;
	if(o.wave >= 0) {
		const A2DWave *w = waves + o.wave;
		o.wperiod = w->period;
		o.wflags = w->flags;
		o.wsize0 = w->size[0];
	}
}

// a2_PrepareRamper, a2_dsp.h:128-149 (the 64 bit division by an exact double division: the first
// branch is only taken with timer >= 256 * frames > 0)
DEV void ramp_prepare_v(Ramp &r, int frames)
{
	if(r.timer == 0) {
		r.value = r.target;
		r.delta = 0;
	} else if(frames <= (r.timer >> 8)) {
		r.delta = (int)div_trunc_exact((int64_t)wsub(r.target, r.value) * 256, r.timer);
		r.timer = wsub(r.timer, frames << 8);
	} else {
		r.delta = frames > 0 ? wsub(r.target, r.value) / frames : 0;
		r.timer = 0;
	}
}

// ph %= (uint64_t)size << 24 (wtosc.c:260-263)
DEV uint64_t wrap_phase_v(uint64_t ph, unsigned size)
{
	if(ph >> 56)
		return size ? ph % ((uint64_t)size << 24) : ph;
	unsigned hi = (unsigned)(ph >> 24);
	if(hi >= size && size) {
		if(!(size & (size - 1)))
			hi &= size - 1;
		else
			hi %= size;
		ph = ((uint64_t)hi << 24) | (ph & 0xffffffu);
	}
	return ph;
}

// a2_P2I, pitch.c:57-67, from the workgroup's copy of the table
typedef uint32_t PTab[128];
DEV unsigned p2i_s(const PTab &tab, int pitch)
{
	const int n = pitch & 0xffff, oct = pitch >> 16;
	unsigned dph = tab[2 * (n >> 10) + 1] * (unsigned)(n & 0x3ff);
	dph >>= 2;
	dph += tab[2 * (n >> 10)];
	return dph >> ((unsigned)(7 - oct) & 31u);
}

// wtosc_run_pitch, wtosc.c:89-105
DEV void run_pitch_v(const PTab &ptab, OscV &o, int frames)
{
	ramp_prepare_v(o.p, frames);
	if(o.dphase && (!o.p.timer && !o.p_ramping))
		return;
	const unsigned lastv = (unsigned)o.p.value;
	ramp_run(o.p, frames);
	o.p_ramping = o.p.delta;
	o.dphase = p2i_s(ptab, (int)((lastv + (unsigned)o.p.value) >> 9));
}

// One window of an oscillator's control state (wtosc_wavetable wtosc.c:239-286, wtosc_noise :129-152,
// wtosc_Off :108-126): steps the state over 'len' frames and says what the frames are made of.
DEV int osc_window_v(const A2DWave *waves, const PTab &ptab, OscV &o, int len, int (&w)[6])
{
	int mode = WM_SILENT;
	w[0] = w[1] = w[2] = w[3] = w[4] = w[5] = 0;
	if(o.mode == A2D_OSC_MIPWAVE) {
		if(o.cmm == -2)
			osc_wave_fields(waves, o);
		if(!o.wsize0) {		// wtosc_check_unloaded, wtosc.c:168-183
			o.wave = -1;
			o.mode = A2D_OSC_OFF;
			return mode;
		}
		const unsigned period = o.wperiod;
		run_pitch_v(ptab, o, len);
		unsigned dph = ((o.dphase + 255) >> 8) * period;
		ramp_prepare_v(o.a, len);
		int mm = 0;
		for(; (dph > (A2D_MAXPHINC << 8)) && (mm < A2D_MIPS - 1); ++mm)
			dph >>= 1;
		if(mm != o.cmm) {
			const A2DWave *wv = waves + o.wave;
			o.csize = wv->size[mm];
			o.coff = wv->off[mm];
			o.cmm = mm;
		}
		uint64_t ph = o.phase >> mm;
		dph = (unsigned)(((uint64_t)o.dphase * period) >> mm);
		if(o.wflags & 0x100u)
			ph = wrap_phase_v(ph, o.csize);
		else if((ph >> 24) > (uint64_t)(o.csize + 1))
			return mode;	// all played: silence, state untouched
		if(dph <= (A2D_MAXPHINC << 16)) {
			mode = WM_TAPS;
			w[WO_A] = (int)o.coff;
			w[WO_B] = (int)(unsigned)ph;
			w[WO_C] = (int)(unsigned)(ph >> 32);
			w[WO_DPH] = (int)dph;
			w[WO_AK] = o.a.value;
			w[WO_DA] = o.a.delta;
		}
		ph += (uint64_t)dph * (unsigned)len;
		o.phase = ph << mm;
		ramp_run(o.a, len);
	} else if(o.mode == A2D_OSC_NOISE) {
		run_pitch_v(ptab, o, len);
		ramp_prepare_v(o.a, len);
		mode = WM_NOISE;
		w[WO_A] = (int)o.seed;
		w[WO_B] = o.noise;
		w[WO_C] = (int)(unsigned)o.phase;
		w[WO_DPH] = (int)o.dphase;
		w[WO_AK] = o.a.value;
		w[WO_DA] = o.a.delta;
		// a frame draws from the engine's LCG when its step crosses a 2^23 boundary of the phase
		// (wtosc.c:140-144): how many do is a difference of two quotients
		const uint64_t end = o.phase + (uint64_t)(unsigned)len * o.dphase;
		const unsigned total = o.dphase >= (1u << 23) ? (unsigned)len : (unsigned)((end >> 23) - (o.phase >> 23));
		unsigned st = o.seed;
		int held = o.noise;
		for(unsigned j = 0; j < total; ++j)
			held = noise_next(st) - 32767;
		o.seed = st;
		o.noise = held;
		o.phase = end;
		ramp_run(o.a, len);
	} else {
		ramp_prepare_v(o.p, len);
		ramp_prepare_v(o.a, len);
		ramp_run(o.p, len);
		ramp_run(o.a, len);
	}
	return mode;
}

// wtosc_Initialize, wtosc.c:390-423 (value = transpose + basepitch)
DEV void osc_init_v(const PTab &ptab, OscV &o, int pitch)
{
	o.wave = -1;
	o.mode = A2D_OSC_OFF;
	o.phase = 0;
	o.p_ramping = 0;
	o.noise = 0;
	o.seed = 0;
	o.cmm = -1;
	o.wperiod = o.wflags = o.wsize0 = o.csize = o.coff = 0;
	ramp_init(o.a, 0);
	ramp_init(o.p, pitch);
	o.dphase = p2i_s(ptab, o.p.value >> 8);
}

DEV void osc_write_v(const A2DWave *waves, OscV &o, int reg, int v, int start, int dur)
{
	switch(reg) {
	  case 0: {	// wtosc_Wave, wtosc.c:433-483 (the host resolved the handle; mip-mapped waves, the
			// noise generator and "off" reach these kernels)
		int wt = 0;
		o.wave = v;
		if(v >= 0) {
			const A2DWave *w = waves + v;
			wt = w->type;
			if(wt == 3 && w->size[0] > (unsigned)A2D_WTOSC_MAXLENGTH)
				wt = 0;
		}
		if(wt == 3)
			o.mode = A2D_OSC_MIPWAVE;
		else if(wt == 1)
			o.mode = A2D_OSC_NOISE;
		else {
			o.wave = -1;
			o.mode = A2D_OSC_OFF;
		}
		osc_wave_fields(waves, o);
		break;
	  }
	  case 1:	// wtosc_Pitch, wtosc.c:486-492 (host added transpose + basepitch)
		ramp_set(o.p, v, start, dur);
		if(!dur)
			o.p_ramping = 1;
		break;
	  case 2:
		ramp_set(o.a, v, start, dur);
		break;
	  case 3:	// wtosc_Phase -> wtosc_set_phase, wtosc.c:369-378
		if(o.wave < 0)
			o.phase = 0;
		else {
			if(o.cmm == -2)
				osc_wave_fields(waves, o);
			const int ph = (int)((unsigned)v + ((((unsigned)start) * (o.dphase >> 8)) >> 8));
			o.phase = (uint64_t)(((int64_t)ph * (int64_t)o.wperiod) * 256);
		}
		break;
	}
}

struct FiltV { Ramp q; int lp, bp, hp, f1, f1next, ramp; };

// A voice's control state as one lane carries it through a batch: the chain osc [osc] [filter12] panmix
// (the classes of the window kernels), whether the voice is alive, and - filter classes - a birth whose
// "filter starts from rest" flag has not been attached to a window yet.
template<int NOSC, int FILT>
struct CtlVoice {
	OscV os[NOSC];
	FiltV fs;
	Ramp vol, pan;
	int active, pending_fresh;
	int uu[NOSC + FILT + 1];	// the chain's unit slots
};

// ... as nothing: a lane without a voice
template<int NOSC, int FILT>
DEV void ctl_clear(CtlVoice<NOSC, FILT> &s)
{
	s.fs.q.value = s.fs.q.target = s.fs.q.delta = s.fs.q.timer = 0;
	s.fs.lp = s.fs.bp = s.fs.hp = s.fs.f1 = s.fs.f1next = s.fs.ramp = 0;
#pragma unroll
	for(int o = 0; o <= NOSC + FILT; ++o)
		s.uu[o] = 0;
#pragma unroll
	for(int o = 0; o < NOSC; ++o) {
		s.os[o].mode = 0; s.os[o].wave = -1; s.os[o].dphase = 0; s.os[o].phase = 0; s.os[o].p_ramping = 0;
		s.os[o].p.value = s.os[o].p.target = s.os[o].p.delta = s.os[o].p.timer = 0;
		s.os[o].a = s.os[o].p;
		s.os[o].noise = 0; s.os[o].seed = 0;
		s.os[o].cmm = -1;
		s.os[o].wperiod = s.os[o].wflags = s.os[o].wsize0 = s.os[o].csize = s.os[o].coff = 0;
	}
	s.vol.value = s.vol.target = s.vol.delta = s.vol.timer = 0;
	s.pan = s.vol;
	s.active = s.pending_fresh = 0;
}

// ... from the unit state words of voice 'slot' (s.uu[] set)
template<int NOSC, int FILT>
DEV void ctl_load(CtlVoice<NOSC, FILT> &s, const int *ustate, const int *vactive, int slot)
{
#pragma unroll
	for(int o = 0; o < NOSC; ++o) {
		const int *w = ustate + (size_t)s.uu[o] * A2D_USTATE;
		s.os[o].mode = w[OW_MODE]; s.os[o].wave = w[OW_WAVE]; s.os[o].dphase = (unsigned)w[OW_DPHASE];
		s.os[o].phase = (uint64_t)(unsigned)w[OW_PHASE_LO] | ((uint64_t)(unsigned)w[OW_PHASE_HI] << 32);
		s.os[o].p_ramping = w[OW_PRAMPING];
		s.os[o].p = ramp_load(w + OW_P);
		s.os[o].a = ramp_load(w + OW_A);
		s.os[o].noise = w[OW_NOISE];
		s.os[o].seed = (unsigned)w[OW_SEED];
		// (the wave's descriptor is read when a window or a phase write first needs it: the state words of
		// a unit that has not been initialized yet - a recycled slot - may hold anything)
		s.os[o].cmm = -2;
		s.os[o].wperiod = s.os[o].wflags = s.os[o].wsize0 = s.os[o].csize = s.os[o].coff = 0;
	}
	if(FILT) {
		const int *wf = ustate + (size_t)s.uu[NOSC] * A2D_USTATE;
		s.fs.q = ramp_load(wf + FW_Q);
		s.fs.lp = wf[FW_LP]; s.fs.bp = wf[FW_BP]; s.fs.hp = wf[FW_HP]; s.fs.f1 = wf[FW_F1];
		s.fs.f1next = wf[FW_F1NEXT]; s.fs.ramp = wf[FW_RAMP];
	}
	const int *wp = ustate + (size_t)s.uu[NOSC + FILT] * A2D_USTATE;
	s.vol = ramp_load(wp + PW_VOL);
	s.pan = ramp_load(wp + PW_PAN);
	s.active = vactive[slot];
}

template<int NOSC, int FILT>
DEV void ctl_store(const CtlVoice<NOSC, FILT> &s, int *ustate, int *vactive, int slot)
{
#pragma unroll
	for(int o = 0; o < NOSC; ++o) {
		int *w = ustate + (size_t)s.uu[o] * A2D_USTATE;
		w[OW_MODE] = s.os[o].mode; w[OW_WAVE] = s.os[o].wave; w[OW_DPHASE] = (int)s.os[o].dphase;
		w[OW_PHASE_LO] = (int)(unsigned)s.os[o].phase; w[OW_PHASE_HI] = (int)(unsigned)(s.os[o].phase >> 32);
		w[OW_PRAMPING] = s.os[o].p_ramping;
		ramp_store(w + OW_P, s.os[o].p);
		ramp_store(w + OW_A, s.os[o].a);
		w[OW_NOISE] = s.os[o].noise;
		w[OW_SEED] = (int)s.os[o].seed;
	}
	if(FILT) {
		int *wf = ustate + (size_t)s.uu[NOSC] * A2D_USTATE;
		ramp_store(wf + FW_Q, s.fs.q);
		wf[FW_LP] = s.fs.lp; wf[FW_BP] = s.fs.bp; wf[FW_HP] = s.fs.hp; wf[FW_F1] = s.fs.f1;
		wf[FW_F1NEXT] = s.fs.f1next; wf[FW_RAMP] = s.fs.ramp;
	}
	int *wp = ustate + (size_t)s.uu[NOSC + FILT] * A2D_USTATE;
	ramp_store(wp + PW_VOL, s.vol);
	ramp_store(wp + PW_PAN, s.pan);
	vactive[slot] = s.active;
}

// One window of the chain - frames [off, off + len) of a fragment (a2_VoiceProcess's Process calls in chain
// order, core.c:1875-1876) - as a closed-form entry W[], the voice's state stepped over it.
template<int NOSC, int FILT>
DEV unsigned ctl_window(CtlVoice<NOSC, FILT> &s, const A2DWave *waves, const PTab &ptab, int off, int len,
		int (&W)[WIN_SW(NOSC, FILT)])
{
	constexpr int SW = WIN_SW(NOSC, FILT), FW = WIN_NW(NOSC);
	unsigned head = (unsigned)(off & 63) | ((unsigned)(len & 127) << 6);
#pragma unroll
	for(int k = 0; k < SW; ++k)
		W[k] = 0;
#pragma unroll
	for(int o = 0; o < NOSC; ++o) {
		int w6[6];
		const int m = osc_window_v(waves, ptab, s.os[o], len, w6);
		head |= (unsigned)m << (15 + 2 * o);
#pragma unroll
		for(int k = 0; k < 6; ++k)
			W[WE_OSC + 6 * o + k] = w6[k];
	}
	if(FILT) {
		// f12_process's head, filter12.c:86-96 (the host / the device VM ran the cutoff
		// ramper and f12_pitch2coeff: R_F1SET / R_F1RAMP)
		FiltV &fs = s.fs;
		ramp_prepare_v(fs.q, len);
		W[FW + WF_F0] = fs.f1;
		if(fs.ramp) {
			const int f0 = fs.f1;
			fs.f1 = fs.f1next;
			W[FW + WF_DF] = len > 0 ? wadd(wsub(fs.f1, f0), len >> 1) / len : 0;
			fs.ramp = 0;
		}
		W[FW + WF_QV] = fs.q.value;
		W[FW + WF_QD] = fs.q.delta;
		ramp_run(fs.q, len);
		W[FW + WF_LP] = fs.lp; W[FW + WF_BP] = fs.bp; W[FW + WF_HP] = fs.hp;
		if(s.pending_fresh) {
			head |= WH_FRESH;
			s.pending_fresh = 0;
		}
	}
	// panmix_process12's head, panmix.c:84-95
	Ramp &vol = s.vol, &pan = s.pan;
	if(pan.target > 0xffffff || pan.target < -0xffffff || pan.value > 0xffffff || pan.value < -0xffffff)
		head |= WH_CLAMP;
	ramp_prepare_v(vol, len);
	ramp_prepare_v(pan, len);
	W[WE_VOL] = vol.value; W[WE_DVOL] = vol.delta;
	W[WE_PAN] = pan.value; W[WE_DPAN] = pan.delta;
	ramp_run(vol, len);
	ramp_run(pan, len);
	if(FILT && len > 0) {	// (cut windows too: k_win_render_f's workers send the lanes outside to the row's padding cell)
		bool plain = true;
#pragma unroll
		for(int o = 0; o < NOSC; ++o)
			plain = plain && WH_MODE(head, o) == WM_TAPS && W[WE_OSC + 6 * o + WO_DA] == 0;
		if(plain)
			head |= WH_PLAINOSC;
	}
	// (any window of the kind, cut ones too - a script's sub-fragment windows: the render pass clamps the frame index of
	// the lanes outside and drops their share with a select, no branch; first cut: whole fragments only, and every second
	// voice-fragment of a scripted batch took the general path)
	if(!FILT && len > 0 && !(W[WE_DVOL] | W[WE_DPAN])) {
		bool plain = true;
#pragma unroll
		for(int o = 0; o < NOSC; ++o)
			plain = plain && WH_MODE(head, o) == WM_TAPS && W[WE_OSC + 6 * o + WO_DA] == 0;
		if(plain) {
			// (win_pan's branch for volume and pan at rest, a2amd_win.hip: the same integer expressions)
			const int v = W[WE_VOL], p = W[WE_PAN];
			const int vp = mul64s(p, v, 24);
			int v0 = wsub(v, vp), v1 = wadd(v, vp);
			if(head & WH_CLAMP) {
				const int lim = wshl(v, 1);
				if(v0 > lim) v0 = lim;
				if(v1 > lim) v1 = lim;
			}
			W[WE_VOL] = v0;
			W[WE_PAN] = v1;
			head |= WH_PLAIN;
		}
	}
	W[WE_HEAD] = (int)head;
	return head;
}

// What a record other than a window does to the voice (the unit callbacks the engine / its VM would have made:
// Initialize, a control write, the filter's coefficient, the voice's end, a noise oscillator's seed).
template<int NOSC, int FILT>
DEV void ctl_apply(CtlVoice<NOSC, FILT> &s, const A2DWave *waves, const PTab &ptab, int *ustate,
		int op, int u, int reg, int value, unsigned dur, unsigned start)
{
	FiltV &fs = s.fs;
	if(op == R_INIT) {
#pragma unroll
		for(int o = 0; o < NOSC; ++o)
			if(u == o)
				osc_init_v(ptab, s.os[o], value);
		if(FILT && u == NOSC) {	// f12_Initialize, filter12.c:180-221; value = f1 from the host
			ramp_init(fs.q, 0);
			ramp_set(fs.q, 32768, 0, 0);	// f12_Q(u, 0, 0, 0)
			fs.lp = 65536 >> 8;
			fs.bp = fs.hp = fs.f1next = fs.ramp = 0;
			fs.f1 = value;
			s.pending_fresh = 1;
			int *wf = ustate + (size_t)s.uu[NOSC] * A2D_USTATE;
			wf[FW_D1B] = 0;
			wf[FW_D2B] = 0;
		}
		if(u == NOSC + FILT) {	// panmix_Initialize, panmix.c:252-284
			ramp_init(s.vol, 65536);
			ramp_init(s.pan, 0);
		}
		s.active = 1;
	} else if(op == R_WRITE) {
#pragma unroll
		for(int o = 0; o < NOSC; ++o)
			if(u == o)
				osc_write_v(waves, s.os[o], reg, value, (int)start, (int)dur);
		if(FILT && u == NOSC) {	// filter12.c:149-177 (the host did the 1/q)
			if(reg == 1)
				ramp_set(fs.q, value, (int)start, (int)dur);
			else if(reg == 2)
				fs.lp = value >> 8;
			else if(reg == 3)
				fs.bp = value >> 8;
			else if(reg == 4)
				fs.hp = value >> 8;
		}
		if(u == NOSC + FILT) {
			if(reg == 0)
				ramp_set(s.vol, value, (int)start, (int)dur);
			else
				ramp_set(s.pan, value, (int)start, (int)dur);
		}
	} else if(op == R_F1SET) {	// f12_CutOff without a ramp: the host's coefficient
		fs.f1 = value;
		fs.ramp = 0;
	} else if(op == R_F1RAMP) {	// ... and one per window while the cutoff ramps
		fs.f1next = value;
		fs.ramp = 1;
	} else if(op == R_KILL) {
		s.active = 0;
	} else if(op == R_NOISESEED) {	// the engine's RNG word as this window of a noise oscillator finds it
#pragma unroll
		for(int o = 0; o < NOSC; ++o)
			if(u == o)
				s.os[o].seed = (unsigned)value;
	}
}

// What the control wavefront leaves per fragment, in LDS: its 64 voices' slots as they go to memory, up to
// WIN_EXL extras per voice, where each voice's extras begin in the pool and how many were staged.  A second
// wavefront of the workgroup carries it out (win_ctl_writer): on gfx9 stores count in vmcnt like loads, and a walk
// that has to wait for a record it asked for an iteration ago would wait for its own stores' round trips with it -
// two microseconds per trip through the loop, four fifths of the first cut's control pass.
// (round 6: sized by class - SW words per entry, and ONE staged extra for the filter classes: k_vm_win's stage was
// 113 KB whatever the class, which beside two 32.5 KB workgroups of k_win_render_f is 3 KB more than a CU has.  For
// THAT pair it is the register file that decides in the end - DESIGN 2c - but 83 / 100 / 90 KB leave the other
// kernels of a batch the room the widest class never needed.)
#define WIN_EXLN(FILT) ((FILT) ? 1 : A2D_WIN_STAGED)
template<int NOSC, int FILT>
struct WinStage {
	static constexpr int SW = WIN_SW(NOSC, FILT), EXL = WIN_EXLN(FILT);
	int slot[2][64 * SW];				// [buffer][lane * SW + word]
	int ext[2][64][EXL][SW];
	unsigned e0[2][64];
	int nst[2][64];
};

// the wavefronts of a workgroup meet: LDS writes done - and nothing else waited for (__syncthreads() also waits for
// the wavefront's outstanding global stores and atomics)
DEV void win_meet()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The control wavefront's companion: after every fragment it copies what that wavefront staged - the 64 slots (one
// contiguous block in memory), the staged extras, the index of each voice's first extra - to memory, while the
// control wavefront walks the next fragment into the other buffer.
template<int NOSC, int FILT>
DEV void win_ctl_writer(int nlist, int first, int fa, int fb, int *__restrict__ wslot, int *__restrict__ wext,
		unsigned *__restrict__ widx, const WinStage<NOSC, FILT> &st)
{
	constexpr int SW = WIN_SW(NOSC, FILT);
	const int lane = threadIdx.x & 63;
	const int nv = max(0, min(64, nlist - first));
	for(int f = fa; f < fb; ++f) {
		const int sb = (f - fa) & 1;
		win_meet();
		int *const dst = wslot + ((size_t)(f - fa) * nlist + first) * SW;
		// (16 bytes per lane and trip: SW / 4 of them per slot)
		for(int i = lane; i < nv * (SW / 4); i += 64)
			((Int4 *)dst)[i] = ((const Int4 *)st.slot[sb])[i];
		if(lane < nv) {
			const unsigned e0 = st.e0[sb][lane];
			const int n = st.nst[sb][lane];
			widx[(size_t)(f - fa) * nlist + first + lane] = e0;
			for(int k = 0; k < n; ++k) {
				Int4 *o = (Int4 *)(wext + ((size_t)e0 + k) * A2D_WIN_WORDS);
				const Int4 *in = (const Int4 *)st.ext[sb][lane][k];
#pragma unroll
				for(int q = 0; q < SW / 4; ++q)
					o[q] = in[q];
			}
		}
	}
}

#endif /* A2AMD_WINCTL_H */
