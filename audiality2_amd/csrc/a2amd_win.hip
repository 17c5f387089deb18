// a2amd_win.hip - voices that carry command records, rendered in two passes (round 5).
//
// A scripted, enveloped or device-VM voice is a record stream: windows that begin and end inside
// a fragment (a2_VoiceProcess, core.c:1852-1878), control writes between them (wtosc.c:433-504,
// panmix.c:219-249, filter12.c:149-177 through a2_SetRamper, a2_dsp.h:161-170), births, deaths.
// Rounds 2-4 interpreted that stream on the SCALAR unit of the wavefront that also rendered the
// voice (k_leaf_recs, a2amd_fast.hip): one voice at a time per wavefront, more scalar than vector
// instructions, 7-16x under the quiet kernels.  Nothing the stream does depends on the AUDIO:
// rampers are linear in time (a2_dsp.h:128-155), an oscillator's phase is phase0 + k * dphase
// (wtosc.c:239-286), which draw of the engine's LCG a noise frame holds is a count of phase
// boundaries (wtosc.c:129-152), the filter's coefficient steps are the host's.  So the stream is
// resolved apart from the rendering:
//
//   k_win_ctl<NOSC, FILT>   lane = VOICE.  Each lane walks its voice's records through the
//                           batch's fragments and carries the voice's whole control state -
//                           rampers, pitch -> increment (a2_P2I), mip level, phase, wave,
//                           mode, liveness - in vector registers; for every window of the
//                           chain it writes one closed-form ENTRY (12 to 24 words): where
//                           each oscillator's taps start and how they step, amplitude, volume
//                           and pan values and per-frame deltas, the filter's cutoff
//                           coefficient and q with their steps.  64 voices per wavefront
//                           instead of one; no per-voice state ever sits in a scalar register.
//   k_win_render<NOSC>      lane = FRAME.  A wavefront takes a group of voices and a chunk of
//                           fragments and evaluates their entries - two coefficient-table taps
//                           per oscillator, the amplitude, the pan gains - with the entry in
//                           scalar registers and no record kinds left to tell apart.  Because an
//                           entry needs nothing of the window before it, the chunks of a batch
//                           are independent: a song's few dozen voices fill thousands of
//                           wavefronts instead of one serial walk each.
//   k_win_render_f<NOSC>    the same with filter12 between the oscillators and the panmix: the
//                           recurrence (filter12.c:97-118) is the one thing serial in time, so a
//                           workgroup owns its voices for the whole batch and runs a pipeline over
//                           the fragments like k_leaf_oscfiltpan - oscillator windows of fragment
//                           f + 1 into LDS rows, ONE wavefront filtering fragment f with lane =
//                           voice, the pan stage on fragment f - 1 - one barrier per step.
//
// State ownership: k_win_ctl reads and writes every unit state word except filter12's d1 / d2,
// which are k_win_render_f's (a voice born in the batch resets them through a flag in the entry).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <algorithm>
#include "a2amd_device.h"
#include "a2amd_dsp.h"
#include "a2amd_taps.h"

#include "a2amd_winctl.h"


template<int NOSC, int FILT>
DEV void win_ctl_body(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int idx, int skip_empty,
		int fa, int fb, int *__restrict__ wslot, int *__restrict__ wext, unsigned *__restrict__ widx,
		unsigned *__restrict__ wtop, unsigned wcap, int *__restrict__ wrc,
		const A2DVoice *__restrict__ voices, int *ustate, int *vactive, const A2DWave *__restrict__ waves,
		const PTab &ptab, const uint8_t *ffr, WinStage<NOSC, FILT> &st)
{
	const A2DParams &p = *pp;
	const int lane = threadIdx.x & 63;
	const bool listed = idx < nlist;
	const A2DRec *__restrict__ recs = p.recs;
	CtlVoice<NOSC, FILT> cv;
	int slot = 0, rc = 0, re = 0;
	bool live = false;
	ctl_clear(cv);
	if(listed) {
		slot = list[idx];
		const A2DVoice &vc = voices[slot];
#pragma unroll
		for(int o = 0; o <= NOSC + FILT; ++o)
			cv.uu[o] = vc.unit[o];
		const A2DRun run = p.runs[slot];
		rc = run.first;
		re = run.first + run.count;
		// skip_empty: a list of voices whose records the device VM writes (a2amd_vm.hip) - the ones it
		// left without any this batch are the quiet kernels' (runs[].count == 0), not ours
		live = !(skip_empty && run.count == 0);
	}
	if(live) {
		ctl_load(cv, ustate, vactive, slot);
		// (a later slab of the batch: the records of the fragments before it have been carried out - the slab
		// before left its place in the voice's run)
		if(fa > 0)
			rc = min(re, max(rc, wrc[idx]));
	}

	// room for this voice's extras: at most one per record (the wavefront takes its voices' sum from the
	// launch's pool in one atomic)
	unsigned e, elim;
	{
		const unsigned cap = live ? (unsigned)(re - rc) : 0u;
		unsigned pre = cap;
#pragma unroll
		for(int d = 1; d < 64; d <<= 1) {
			const unsigned t = (unsigned)__shfl_up((int)pre, d, 64);
			if(lane >= d)
				pre += t;
		}
		const unsigned total = (unsigned)__shfl((int)pre, 63, 64);
		unsigned base = 0;
		if(lane == 0 && total)
			base = atomicAdd(wtop, total);
		base = (unsigned)__shfl((int)base, 0, 64);
		e = base + pre - cap;
		elim = e + cap;
		if(base + total > wcap) {	// (the host sizes the pool by the same bound: never; the voices of
			if(lane == 0)		// this wavefront then lose their extra windows and the flag says so)
				atomicOr(wtop + 1, 1u);
			elim = e = 0;
		}
	}

	// the next two records of the voice, asked for ahead of their turn
	const Int4 none = { 0, 0, 0, 0 };
	Int4 rq = none, rq1 = none;
	if(live) {
		if(rc < re)
			rq = *(const Int4 *)(recs + rc);
		if(rc + 1 < re)
			rq1 = *(const Int4 *)(recs + rc + 1);
	}
#ifdef WIN_PROF
	const long long cp_in = __builtin_readcyclecounter();
	long long cp_meet = 0;
	int cp_trips = 0;
#endif
	for(int f = fa; f < fb; ++f) {
		const int n = (int)ffr[f];
		const int sb = (f - fa) & 1;
		int *const myslot = st.slot[sb] + lane * WIN_SW(NOSC, FILT);		// (LDS: win_ctl_writer carries it out)
		int nstaged = 0;
		const unsigned e0 = e;
		unsigned head0 = 0;
		int nwin = 0;
		// this lane's next thing to do in fragment f: its records, or - none - the engine's default
		// Process(0, frames) on every unit (core.c:1875-1876)
		Int4 r = { 0, 0, n << 16, 0 };
		int op = 0;
		bool inrec = false;
		if(live) {
			if(rc < re && (int)A2D_RFRAG((unsigned)rq.x) == f) {
				r = rq;
				inrec = true;
			}
			op = inrec ? (int)A2D_ROP((unsigned)r.x) : (cv.active ? R_SEG : 0);
			if(inrec && !op)
				op = R_NOP;
		}
		while(__ballot(op != 0)) {
#ifdef WIN_PROF
			++cp_trips;
#endif
			const int value = r.y;
			const unsigned dur = (unsigned)r.z, start = (unsigned)r.w;
			const int u = (int)A2D_RUNIT((unsigned)r.x), reg = (int)A2D_RREG((unsigned)r.x);
			if(op == R_SEG) {
				if(cv.active) {
					constexpr int SW = WIN_SW(NOSC, FILT);
					int W[SW];
					const unsigned head = ctl_window(cv, waves, ptab, (int)(dur & 0xffffu), (int)(dur >> 16), W);
					// the fragment's first window into the voice's slot (its head word follows when the
					// fragment is done: it counts the extras), the others into the pool
					int *dst = nullptr;
					if(!nwin) {
						head0 = head;
						dst = myslot;
					} else if(e < elim) {
						// (the first WIN_EXLN extras of a fragment are staged too; a busier fragment stores the rest itself)
						dst = nstaged < WIN_EXLN(FILT) && nstaged == nwin - 1 ? st.ext[sb][lane][nstaged++] : wext + (size_t)e * A2D_WIN_WORDS;
						++e;
					}
					++nwin;
					if(dst) {
#pragma unroll
						for(int k = 0; k < SW / 4; ++k) {
							const Int4 q = { W[4 * k], W[4 * k + 1], W[4 * k + 2], W[4 * k + 3] };
							((Int4 *)dst)[k] = q;
						}
					}
				}
			} else
				ctl_apply(cv, waves, ptab, ustate, op, u, reg, value, dur, start);
			// the next one
			if(inrec) {
				++rc;
				rq = rq1;
				rq1 = none;
				if(rc + 1 < re)
					rq1 = *(const Int4 *)(recs + rc + 1);
				inrec = false;
				op = 0;
				if(rc < re && (int)A2D_RFRAG((unsigned)rq.x) == f) {
					r = rq;
					inrec = true;
					op = (int)A2D_ROP((unsigned)rq.x);
					if(!op)
						op = R_NOP;
				}
			} else
				op = 0;
		}
		{
			// a voice born without a window behind its birth in this slab: its filter still starts from
			// rest - the slab's last slot carries the flag, with or without frames
			if(FILT && cv.pending_fresh && f == fb - 1) {
				head0 |= WH_FRESH;
				cv.pending_fresh = 0;
			}
			myslot[WE_HEAD] = (int)(head0 | ((unsigned)(nwin > 1 ? min(nwin - 1, (int)(e - e0)) : 0) << 19));
			st.e0[sb][lane] = e0;
			st.nst[sb][lane] = nstaged;
		}
#ifdef WIN_PROF
		const long long cp_m0 = __builtin_readcyclecounter();
#endif
		win_meet();	// (the writer takes buffer sb from here; this wavefront goes on into the other one)
#ifdef WIN_PROF
		cp_meet += __builtin_readcyclecounter() - cp_m0;
#endif
	}
#ifdef WIN_PROF
	if(blockIdx.x % 61 == 5 && lane == 0)
		printf("k_win_ctl<%d,%d> block %d: %d fragments, %d trips through the op switch: %lld cycles (of which %lld at the meeting point)\n",
				NOSC, FILT, (int)blockIdx.x, fb - fa, cp_trips, (long long)__builtin_readcyclecounter() - cp_in, cp_meet);
#endif

	if(live) {
		ctl_store(cv, ustate, vactive, slot);
		wrc[idx] = rc;
	}
}

template<int NOSC, int FILT>
__global__ __launch_bounds__(128)
void k_win_ctl(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int skip_empty,
		int fa, int fb, int *__restrict__ wslot, int *__restrict__ wext, unsigned *__restrict__ widx,
		unsigned *__restrict__ wtop, unsigned wcap, int *__restrict__ wrc,
		const A2DVoice *__restrict__ voices, int *ustate, int *vactive, const A2DWave *__restrict__ waves,
		const uint32_t *__restrict__ ptab)
{
	// the pitch table and the batch's fragment lengths: read in every window, from LDS
	__shared__ PTab s_ptab;
	__shared__ uint8_t s_ffr[A2D_MAXBATCH];
	__shared__ WinStage<NOSC, FILT> s_stage;
	for(int k = (int)threadIdx.x; k < 128; k += 128)
		s_ptab[k] = ptab[k];
	for(int k = (int)threadIdx.x; k < A2D_MAXBATCH; k += 128)
		s_ffr[k] = pp->fragframes[k];
	__syncthreads();
	if(rfl((int)(threadIdx.x >> 6)) == 0)
		win_ctl_body<NOSC, FILT>(pp, list, nlist, (int)(blockIdx.x * 64 + threadIdx.x), skip_empty, fa, fb, wslot, wext, widx, wtop, wcap,
				wrc, voices, ustate, vactive, waves, s_ptab, s_ffr, s_stage);
	else
		win_ctl_writer<NOSC, FILT>(nlist, (int)blockIdx.x * 64, fa, fb, wslot, wext, widx, s_stage);
}

// ---------------------------------------------------------------------------
// the render passes: lane = frame
// ---------------------------------------------------------------------------
#define WIN_FCH 4		// fragments of a chunk
#define WIN_WPB 8		// wavefronts per workgroup of k_win_render: their bus sums meet in LDS before the atomics

// An entry's words as wave-uniform values.  LaneAcc: the entry sits in lane v of NW vector registers (the
// lane = voice preload) and a word comes out with v_readlane; ScalAcc: already in scalar registers.
template<int NW>
struct LaneAcc {
	const int (&S)[NW];
	int v;
	DEV int operator()(int w) const { return rdl(S[w], v); }
};
template<int NW>
struct ScalAcc {
	int w[NW];
	DEV int operator()(int k) const { return w[k]; }
};

// lane 'lane' (one voice each) takes the first NW words of its entry at p (16-byte loads)
template<int NW>
DEV void lane_load(const int *p, bool on, int (&S)[NW])
{
#pragma unroll
	for(int k = 0; k < NW; ++k)
		S[k] = 0;
	if(on) {
#pragma unroll
		for(int q = 0; q < (NW + 3) / 4; ++q) {
			const Int4 x = ((const Int4 *)p)[q];
#pragma unroll
			for(int k = 0; k < 4; ++k)
				if(4 * q + k < NW)
					S[4 * q + k] = x[k];
		}
	}
}

// an entry every lane reads (a fragment's third window and beyond: rare), into scalar registers
template<int NW>
DEV ScalAcc<NW> bcast_load(const int *p)
{
	ScalAcc<NW> E;
#pragma unroll
	for(int q = 0; q < (NW + 3) / 4; ++q) {
		const Int4 x = ((const Int4 *)p)[q];
#pragma unroll
		for(int k = 0; k < 4; ++k)
			if(4 * q + k < NW)
				E.w[4 * q + k] = rfl(x[k]);
	}
	return E;
}

// The oscillators of one window in two halves: win_taps_issue() works out where this lane's frame reads the
// wave and ISSUES the coefficient loads, win_oscs_finish() interpolates - the render loops issue for the next
// entry before they finish this one, so a wavefront always has a window's loads in flight behind its arithmetic.
// (fl = frame within the window; 'in' = the lane holds one)
template<int NOSC>
struct WinTaps { Coef4 k1[NOSC], k2[NOSC]; unsigned t1[NOSC], t2[NOSC]; unsigned head; int ak[NOSC], da[NOSC]; };

// PLAIN (-1: look at the head; 0 / 1 / 2: the caller knows - 1 a plain window that is the whole fragment, 2 a plain window that
// is cut: the frame index of the lanes outside clamped, their share dropped with a select - the pipelined loops of win_run take a fragment's plain entries
// and its others apart, so that neither loop branches on the kind between an entry's loads and the next one's: a
// branch there cost more than the plain path saved - the wait at the join is for ALL loads in flight)
template<int NOSC, class EA, int PLAIN = -1>
DEV void win_taps_issue(const EA &E, const CoefRsrc rs, int lane, WinTaps<NOSC> &T)
{
	const unsigned head = (unsigned)E(WE_HEAD);
	T.head = head;
	if(PLAIN < 0 ? (head & WH_PLAIN) != 0 : PLAIN != 0) {
		// taps, everything at rest (a2amd_winctl.h).  A lane outside the window (a cut one) reads where frame 0 reads -
		// a valid address - and win_finish_pan drops what it made of it: no branch, no exec mask
		const int flp = lane - WH_OFF(head);
		const unsigned flc = PLAIN == 1 ? (unsigned)lane : (unsigned)flp < (unsigned)WH_LEN(head) ? (unsigned)flp : 0u;
#pragma unroll
		for(int o = 0; o < NOSC; ++o) {
			const int b = WE_OSC + 6 * o;
			const unsigned phlo = (unsigned)E(b + WO_B), phhi = (unsigned)E(b + WO_C), dph = (unsigned)E(b + WO_DPH);
			const int cb = coef_base((unsigned)E(b + WO_A));
			T.ak[o] = E(b + WO_AK);
			T.da[o] = 0;
			T.t1[o] = tap_phase((uint64_t)phlo | ((uint64_t)phhi << 32), flc * dph);
			T.t2[o] = T.t1[o] + ((dph >> 16) >> 1);
			T.k1[o] = coef_at(rs, cb, T.t1[o]);
			T.k2[o] = coef_at(rs, cb, T.t2[o]);
		}
		return;
	}
	const int fl = lane - WH_OFF(head);
	const bool in = (unsigned)fl < (unsigned)WH_LEN(head);
#pragma unroll
	for(int o = 0; o < NOSC; ++o) {
		const int b = WE_OSC + 6 * o;
		T.t1[o] = T.t2[o] = 0;
		T.ak[o] = T.da[o] = 0;
		if(WH_MODE(head, o) == WM_TAPS) {
			const unsigned phlo = (unsigned)E(b + WO_B), phhi = (unsigned)E(b + WO_C), dph = (unsigned)E(b + WO_DPH);
			const int cb = coef_base((unsigned)E(b + WO_A));
			T.ak[o] = E(b + WO_AK);
			T.da[o] = E(b + WO_DA);
			if(in) {
				const uint64_t ph = (uint64_t)phlo | ((uint64_t)phhi << 32);
				T.t1[o] = tap_phase(ph, (unsigned)fl * dph);
				T.t2[o] = T.t1[o] + ((dph >> 16) >> 1);
				T.k1[o] = coef_at(rs, cb, T.t1[o]);
				T.k2[o] = coef_at(rs, cb, T.t2[o]);
			}
		}
	}
}

template<int NOSC, class EA>
DEV int win_oscs_finish(const EA &E, const WinTaps<NOSC> &T, int fl, bool in)
{
	const unsigned head = T.head;
	int x = 0;
#pragma unroll
	for(int o = 0; o < NOSC; ++o) {
		const unsigned m = WH_MODE(head, o);
		const int b = WE_OSC + 6 * o;
		if(m == WM_TAPS) {
			if(in) {
				const int h = hermite_c(T.k1[o], T.t1[o]) + hermite_c(T.k2[o], T.t2[o]);
				// (an amplitude at rest - most windows - is one scalar for the window)
				const int ak = T.da[o] ? wadd(T.ak[o], wmul(T.da[o], fl)) : T.ak[o];
				x = wadd(x, mul64s(h, ak, 17));
			}
		} else if(m == WM_NOISE) {
			// wtosc_noise, wtosc.c:129-152: which draw of the LCG a frame holds is a prefix count over
			// the window's frames, the draws themselves a uniform loop from the window's seed
			const unsigned dph = (unsigned)E(b + WO_DPH);
			const unsigned phk = (unsigned)E(b + WO_C) + (unsigned)(in ? fl : 0) * dph;
			const bool draw = in && ((dph >= (1u << 23)) || (((phk + dph) ^ phk) >> 23));
			const unsigned long long dm = __ballot(draw);
			const int me = (int)(threadIdx.x & 63);
			const unsigned long long below = (me >= 63) ? ~0ull : ((2ull << me) - 1ull);
			const int mine = __popcll(dm & below), total = __popcll(dm);
			int held = E(b + WO_B), myval = held;
			unsigned st = (unsigned)E(b + WO_A);
			for(int j = 1; j <= total; ++j) {
				held = noise_next(st) - 32767;
				if(j == mine)
					myval = held;
			}
			if(in) {
				const int ak = wadd(E(b + WO_AK), wmul(E(b + WO_DA), fl));
				x = wadd(x, wmul(myval, ak >> 10) >> 6);
			}
		}
	}
	return x;
}

// panmix_process12 (panmix.c:78-125) of one window for input y; the window's words as scalars
DEV void win_pan(unsigned head, int vol, int dvol, int pan, int dpan, int y, int fl, bool in, int &acc0, int &acc1)
{
	const bool clamp = (head & WH_CLAMP) != 0;
	if(!(dvol | dpan)) {
		// volume and pan at rest - most windows: the two gains are the window's, worked out on the scalar unit
		const int vp = mul64s(pan, vol, 24);
		int v0 = wsub(vol, vp), v1 = wadd(vol, vp);
		if(clamp) {
			const int lim = wshl(vol, 1);
			if(v0 > lim) v0 = lim;
			if(v1 > lim) v1 = lim;
		}
		if(in) {
			acc0 = wadd(acc0, mul64s(y, v0, 24));
			acc1 = wadd(acc1, mul64s(y, v1, 24));
		}
	} else if(in) {
		const int vk = wadd(vol, wmul(dvol, fl));
		const int pk = wadd(pan, wmul(dpan, fl));
		const int vp = mul64s(pk, vk, 24);
		int v0 = wsub(vk, vp), v1 = wadd(vk, vp);
		if(clamp) {
			const int lim = wshl(vk, 1);
			if(v0 > lim) v0 = lim;
			if(v1 > lim) v1 = lim;
		}
		acc0 = wadd(acc0, mul64s(y, v0, 24));
		acc1 = wadd(acc1, mul64s(y, v1, 24));
	}
}

DEV void bus_add(int *busmem, int off, int nch, int f, int lane, int dbg, int &a0, int &a1)
{
	if(off >= 0 && !(dbg & 1)) {
		int *dst = busmem + off + (size_t)f * nch * A2D_FRAG;
		if(a0)
			atomicAdd(&dst[lane], a0);
		if(a1)
			atomicAdd(&dst[A2D_FRAG + lane], a1);
	}
	a0 = a1 = 0;
}

// the second half of an entry whose taps have been issued: interpolation, amplitude, pan stage into the bus sums
template<int NOSC, class EA, int PLAIN = -1>
DEV void win_finish_pan(const EA &E, const WinTaps<NOSC> &T, int lane, int &a0, int &a1)
{
	if(PLAIN < 0 ? (T.head & WH_PLAIN) != 0 : PLAIN != 0) {
		int x = 0;
#pragma unroll
		for(int o = 0; o < NOSC; ++o)
			x = wadd(x, mul64s(hermite_c(T.k1[o], T.t1[o]) + hermite_c(T.k2[o], T.t2[o]), T.ak[o], 17));
		if(PLAIN != 1)
			x = (unsigned)(lane - WH_OFF(T.head)) < (unsigned)WH_LEN(T.head) ? x : 0;	// (lanes outside a cut window)
		a0 = wadd(a0, mul64s(x, E(WE_VOL), 24));	// (WE_VOL / WE_PAN: the two gains, WH_PLAIN)
		a1 = wadd(a1, mul64s(x, E(WE_PAN), 24));
		return;
	}
	const int fl = lane - WH_OFF(T.head);
	const bool in = (unsigned)fl < (unsigned)WH_LEN(T.head);
	const int x = win_oscs_finish<NOSC>(E, T, fl, in);
	win_pan(T.head, E(WE_VOL), E(WE_DVOL), E(WE_PAN), E(WE_DPAN), x, fl, in, a0, a1);
}

// one entry, start to finish (not pipelined), oscillators + pan stage
template<int NOSC, class EA>
DEV void win_entry_now(const EA &E, const CoefRsrc rs, int lane, int &a0, int &a1)
{
	WinTaps<NOSC> T;
	win_taps_issue<NOSC>(E, rs, lane, T);
	win_finish_pan<NOSC>(E, T, lane, a0, a1);
}

// The voices whose entries sit one per lane in S (lane k = voice v0 + k of the wavefront's run, n of them), one
// after the other: the coefficient loads of voice k + 1 are issued before voice k's arithmetic.  mask: which
// lanes hold an entry at all.  on_bus(k): the wavefront moves to voice k (bus bookkeeping).
template<int NOSC, int NW, int PLAIN, class BUS>
DEV void win_run_kind(const int (&S)[NW], unsigned long long mask, const CoefRsrc rs, int lane, int &a0, int &a1, BUS on_bus)
{
	if(!mask)
		return;
	int k = (int)__builtin_ctzll(mask);
	mask &= mask - 1;
	WinTaps<NOSC> T0;
	{
		const LaneAcc<NW> E = { S, k };
		win_taps_issue<NOSC, LaneAcc<NW>, PLAIN>(E, rs, lane, T0);
	}
	for(;;) {
		const int kn = mask ? (int)__builtin_ctzll(mask) : -1;
		WinTaps<NOSC> T1;
		if(kn >= 0) {
			const LaneAcc<NW> En = { S, kn };
			win_taps_issue<NOSC, LaneAcc<NW>, PLAIN>(En, rs, lane, T1);
		}
		on_bus(k);
		{
			const LaneAcc<NW> E = { S, k };
			win_finish_pan<NOSC, LaneAcc<NW>, PLAIN>(E, T0, lane, a0, a1);
		}
		if(kn < 0)
			break;
		mask &= mask - 1;
		k = kn;
		T0 = T1;
	}
}

// ... a fragment's plain entries (WH_PLAIN, a2amd_winctl.h) first, in a loop that knows they are, then the others
template<int NOSC, int NW, class BUS>
DEV void win_run(const int (&S)[NW], unsigned long long mask, const CoefRsrc rs, int lane, int &a0, int &a1, BUS on_bus)
{
	const unsigned long long plain = mask & __ballot(((unsigned)S[WE_HEAD] & WH_PLAIN) != 0);
	// (offset 0, length 64: the low 13 bits of the head)
	const unsigned long long whole = plain & __ballot(((unsigned)S[WE_HEAD] & 0x1fffu) == ((unsigned)A2D_FRAG << 6));
	win_run_kind<NOSC, NW, 1>(S, whole, rs, lane, a0, a1, on_bus);
	win_run_kind<NOSC, NW, 2>(S, plain & ~whole, rs, lane, a0, a1, on_bus);
	win_run_kind<NOSC, NW, 0>(S, mask & ~plain, rs, lane, a0, a1, on_bus);
}

template<int NOSC>
__global__ __launch_bounds__(64 * WIN_WPB)
void k_win_render(const int *__restrict__ list, int nlist, int vpw, int fa, int fb, const int *__restrict__ wslot,
		const int *__restrict__ wext, const unsigned *__restrict__ widx, const A2DVoice *__restrict__ voices,
		const int *__restrict__ wavecoef, int *__restrict__ busmem, int dbg)
{
	constexpr int NW = WIN_NW(NOSC), SW = WIN_SW(NOSC, 0);
	// what the workgroup's wavefronts are left with at the end of a fragment: summed here before it goes to the bus
	// (sixteen thousand voices playing straight into one bus are as many atomic adds on the same 512 bytes per
	// fragment; neighbours in the list share their bus: it is sorted) - two buffers, one barrier per fragment
	__shared__ int part[2][WIN_WPB][2][64];
	__shared__ int part_off[2][WIN_WPB], part_nch[2][WIN_WPB];
	const int wv = rfl((int)(threadIdx.x >> 6));
	const int lane = threadIdx.x & 63;
	// a workgroup = WIN_WPB neighbouring voice groups x one chunk of fragments
	const int ngroups = (nlist + vpw - 1) / vpw;
	const int gblocks = (ngroups + WIN_WPB - 1) / WIN_WPB;
	const int chunk = (int)blockIdx.x / gblocks, group = ((int)blockIdx.x % gblocks) * WIN_WPB + wv;
	const int f0 = fa + chunk * WIN_FCH;
	const int nf = max(0, min(WIN_FCH, fb - f0));
	const int first = group * vpw, nv = group < ngroups ? min(vpw, nlist - first) : 0;
	const CoefRsrc rs = coef_rsrc(wavecoef);
	const bool mine = lane < nv;
	// lane v keeps voice v: its output bus, and per fragment its slot and its first extra
	int l_off = -1, l_nch = 2;
	if(mine) {
		const A2DVoice &vc = voices[list[first + lane]];
		l_off = vc.out_off;
		l_nch = vc.out_nch;
	}
	int S[NW], Sn[NW], X[NW];
	unsigned wi = 0, win = 0;
	lane_load<NW>(wslot + ((size_t)(f0 - fa) * nlist + first + lane) * SW, mine && nf > 0, S);
	if(mine && nf > 0)
		wi = widx[(size_t)(f0 - fa) * nlist + first + lane];
	for(int j = 0; j < nf; ++j) {
		const int f = f0 + j;
		// the first extras of this fragment (behind the slots just arrived) and the slots of the next one:
		// on their way while the slots are rendered
		const int nx = mine ? WH_EXTRAS((unsigned)S[WE_HEAD]) : 0;
		lane_load<NW>(wext + (size_t)wi * A2D_WIN_WORDS, nx > 0, X);
		if(j + 1 < nf) {
			lane_load<NW>(wslot + ((size_t)(f + 1 - fa) * nlist + first + lane) * SW, mine, Sn);
			if(mine)
				win = widx[(size_t)(f + 1 - fa) * nlist + first + lane];
		}
		int a0 = 0, a1 = 0;
		int cur_off = rdl(l_off, 0), cur_nch = rdl(l_nch, 0);
		auto on_bus = [&](int k) {
			const int voff = rdl(l_off, k);
			if(voff != cur_off) {
				bus_add(busmem, cur_off, cur_nch, f, lane, dbg, a0, a1);
				cur_off = voff;
				cur_nch = rdl(l_nch, k);
			}
		};
		win_run<NOSC, NW>(S, __ballot(mine && WH_LEN((unsigned)S[WE_HEAD]) != 0), rs, lane, a0, a1, on_bus);
		const unsigned long long xm = __ballot(nx > 0);
		win_run<NOSC, NW>(X, xm, rs, lane, a0, a1, on_bus);
		// a fragment's third window and beyond
		unsigned long long more = __ballot(nx > 1);
		while(more) {
			const int k = (int)__builtin_ctzll(more);
			more &= more - 1;
			on_bus(k);
			const int n = rdl(nx, k);
			const unsigned e0 = (unsigned)rdl((int)wi, k);
			for(int q = 1; q < n; ++q) {
				const ScalAcc<NW> E = bcast_load<NW>(wext + ((size_t)e0 + q) * A2D_WIN_WORDS);
				win_entry_now<NOSC>(E, rs, lane, a0, a1);
			}
		}
		// the wavefronts' sums of this fragment meet: wavefront r adds up row r (channel r) of its neighbours on the same bus
		{
			const int pb = j & 1;
			part[pb][wv][0][lane] = a0;
			part[pb][wv][1][lane] = a1;
			if(lane == 0) {
				part_off[pb][wv] = nv ? cur_off : -1;
				part_nch[pb][wv] = cur_nch;
			}
			win_meet();
			if(wv < 2) {
				// (every wavefront's row asked for at once, then added up bus by bus)
				const int po = lane < WIN_WPB ? part_off[pb][lane] : -2, pn = lane < WIN_WPB ? part_nch[pb][lane] : 2;
				int pv[WIN_WPB];
#pragma unroll
				for(int w = 0; w < WIN_WPB; ++w)
					pv[w] = part[pb][w][wv][lane];
				int sum = 0, off = -1, nch = 2;
#pragma unroll
				for(int w = 0; w <= WIN_WPB; ++w) {
					const int woff = w < WIN_WPB ? rdl(po, w < WIN_WPB ? w : 0) : -2;
					if(woff != off) {
						if(off >= 0 && sum && !(dbg & 1))
							atomicAdd(busmem + off + ((size_t)f * nch + wv) * A2D_FRAG + lane, sum);
						sum = 0;
						off = woff;
						nch = w < WIN_WPB ? rdl(pn, w < WIN_WPB ? w : 0) : 2;
					}
					if(w < WIN_WPB && woff >= 0)
						sum = wadd(sum, pv[w]);
				}
			}
		}
		if(j + 1 < nf) {
#pragma unroll
			for(int k = 0; k < NW; ++k)
				S[k] = Sn[k];
			wi = win;
		}
	}
}

// ---- with filter12: a workgroup owns its voices for the whole slab -----------------------------
#define WINF_PITCH 65
#ifndef WINF_W1
#define WINF_W1 8	/* (16 for the one-oscillator kernel measured slower: 0.71 against 0.63 ms, 16 384 gliding voices x 64) */
#endif
#ifndef WINF_SPEC
#define WINF_SPEC 1
#endif
#define WINF_MAXW(NOSC) ((NOSC) == 1 ? WINF_W1 : 8)	// wavefronts per workgroup, one of which filters (two oscillators: the
							// kernel needs more than the 128 registers sixteen would leave it)

// f12_process, filter12.c:97-118, over the frames [off, off + len) of a voice's row, in place.  Two things most
// windows do not need are decided for the whole wavefront (its lanes are its voices): LP - plain low passes, the
// band / high pass products drop out of the output sum; REST - cutoff and q at rest, their per-frame steps and
// the shifts of the values drop out of the loop.
template<bool LP, bool REST>
DEV void winf_filter_t(int *row, int off, int len, int f0v, int df, int qv, int qd, int lp, int bp, int hp, int &d1, int &d2)
{
	const int fq0 = f0v >> 12, qq0 = qv >> 12;
	auto step = [&](int xin, int at) {
		const int fq = REST ? fq0 : f0v >> 12, qq = REST ? qq0 : qv >> 12;
		const int d1s = d1 >> 4;
		const int l = wadd(d2, wmul(fq, d1s) >> 8);
		const int h = wsub(wsub(xin >> 5, l), wmul(qq, d1s) >> 8);
		const int b = wadd(wmul(fq, h >> 4) >> 8, d1);
		row[at] = (LP ? wmul(l, lp) : wadd(wadd(wmul(l, lp), wmul(b, bp)), wmul(h, hp))) >> 3;
		d1 = b;
		d2 = l;
		if(!REST) {
			f0v = wadd(f0v, df);
			qv = wadd(qv, qd);
		}
	};
	int k = 0;
	for(; k + 4 <= len; k += 4) {
		// (four inputs read before the first one's chain starts)
		const int at = off + k;
		const int x0 = row[at], x1 = row[at + 1], x2 = row[at + 2], x3 = row[at + 3];
		step(x0, at);
		step(x1, at + 1);
		step(x2, at + 2);
		step(x3, at + 3);
	}
	for(; k < len; ++k)
		step(row[off + k], off + k);
}

DEV void winf_filter(int *row, int off, int len, int f0v, int df, int qv, int qd, int lp, int bp, int hp, int &d1, int &d2)
{
	const bool lponly = WINF_SPEC && __all((bp | hp) == 0), rest = WINF_SPEC && __all((df | qd) == 0);
	// (Round 6, tried and dropped: a blocked loop for the case that every lane's window is the whole fragment - 8 or 4 frames
	// at a time, the next block on its way from the LDS, as k_leaf_oscfiltpan's filt_row - measured SLOWER on one box
	// against this loop: 16 384 x 64, wtosc-filter12-panmix scripted 0.896 -> 0.948 ms, gliding 0.633 -> 0.686; two
	// oscillators 1.472 -> 1.624, 1.049 -> 1.141 (profiles/r06_window_kernels_ab.txt).  The kernel sits at its 128
	// registers; the block buffers spilled.)
	if(lponly && rest)
		winf_filter_t<true, true>(row, off, len, f0v, df, qv, qd, lp, bp, hp, d1, d2);
	else if(lponly)
		winf_filter_t<true, false>(row, off, len, f0v, df, qv, qd, lp, bp, hp, d1, d2);
	else if(rest)
		winf_filter_t<false, true>(row, off, len, f0v, df, qv, qd, lp, bp, hp, d1, d2);
	else
		winf_filter_t<false, false>(row, off, len, f0v, df, qv, qd, lp, bp, hp, d1, d2);
}

// the pan stage's words of an entry (head + panmix), kept from the step that rendered its oscillators
struct PanW { int w[5]; };

template<int NOSC>
__global__ __launch_bounds__(64 * WINF_MAXW(NOSC)) __attribute__((amdgpu_waves_per_eu(NOSC == 1 ? 4 : 2, NOSC == 1 ? 4 : 2)))
void k_win_render_f(const int *__restrict__ list, int nlist, int vpg, int fa, int fb, const int *__restrict__ wslot,
		const int *__restrict__ wext, const unsigned *__restrict__ widx, const A2DVoice *__restrict__ voices, int *ustate,
		const int *__restrict__ wavecoef, int *__restrict__ busmem, int dbg)
{
	constexpr int NW = WIN_NW(NOSC), SW = WIN_SW(NOSC, 1), FW = WIN_NW(NOSC);
	extern __shared__ int winf_lds[];
	const int nw = (int)(blockDim.x >> 6);		// >= 2
	const int wv = rfl((int)(threadIdx.x >> 6));
	const int lane = threadIdx.x & 63;
	const int first = (int)blockIdx.x * vpg, nv = min(vpg, nlist - first);
	const int nfr = fb - fa;
	int *const tiles = winf_lds;
	int *const part = winf_lds + 3 * vpg * WINF_PITCH;		// [2][nw][2][64]
	int *const part_off = part + 2 * nw * 2 * 64;			// [2][nw] bus offset
	int *const part_nch = part_off + 2 * nw;			// [2][nw]
	const CoefRsrc rs = coef_rsrc(wavecoef);

	// the filter wavefront: lane = voice, d1 / d2 in registers over the slab
	int d1 = 0, d2 = 0, ufilt = -1;
	int fhd = 0, fw[7];				// the voice's slot of the fragment to be filtered next: head, filter words
	unsigned fwi = 0;
#pragma unroll
	for(int k = 0; k < 7; ++k)
		fw[k] = 0;
	auto fload = [&](int g) {
		const int *S = wslot + ((size_t)g * nlist + first + lane) * SW;
		fhd = S[WE_HEAD];
#pragma unroll
		for(int k = 0; k < 7; ++k)
			fw[k] = S[FW + k];
		fwi = widx[(size_t)g * nlist + first + lane];
	};
	if(wv == 0 && lane < nv) {
		ufilt = voices[list[first + lane]].unit[NOSC];
		d1 = ustate[(size_t)ufilt * A2D_USTATE + FW_D1A];
		d2 = ustate[(size_t)ufilt * A2D_USTATE + FW_D2A];
		fload(0);
	}
	// the others: a contiguous share of the voices each (neighbours in the list share their bus), one per lane
	const int nworkers = nw - 1;
	const int per = (nv + nworkers - 1) / nworkers;
	const int lo = min(nv, (wv - 1) * per), hi = wv ? min(nv, lo + per) : 0;
	const bool mine = wv && lo + lane < hi;
	int l_off = -1, l_nch = 2;
	if(mine) {
		const A2DVoice &vc = voices[list[first + lo + lane]];
		l_off = vc.out_off;
		l_nch = vc.out_nch;
	}
	// a worker's voices, lane = voice: the slot (and first extra) of the fragment whose oscillators it renders in
	// this step, of the next one (on their way), and the pan words of the two fragments before
	int S[NW], Sn[NW], X[NW];
	unsigned wi = 0, win = 0;
	PanW P1, P2, PX1, PX2;		// slots / first extras of fragments s - 1 and s - 2
	unsigned wi1 = 0, wi2 = 0;
#pragma unroll
	for(int k = 0; k < 5; ++k)
		P1.w[k] = P2.w[k] = PX1.w[k] = PX2.w[k] = 0;
	if(wv) {
		lane_load<NW>(wslot + ((size_t)first + lo + lane) * SW, mine && nfr > 0, S);
		if(mine && nfr > 0)
			wi = widx[first + lo + lane];
	}

	for(int s = 0; s < nfr + 3; ++s) {
		if(wv == 0) {
			const int g = s - 1;		// fragment (of the slab) to filter
			if(g >= 0 && g < nfr && lane < nv) {
				int *const row = tiles + (g % 3) * vpg * WINF_PITCH + lane * WINF_PITCH;
				const unsigned head = (unsigned)fhd, e0 = fwi;
				int q[7];
#pragma unroll
				for(int k = 0; k < 7; ++k)
					q[k] = fw[k];
				if(g + 1 < nfr)		// (the next fragment's, on their way while this one is filtered)
					fload(g + 1);
				if(head & WH_FRESH)
					d1 = d2 = 0;
				winf_filter(row, WH_OFF(head), WH_LEN(head), q[WF_F0], q[WF_DF], q[WF_QV], q[WF_QD], q[WF_LP], q[WF_BP], q[WF_HP], d1, d2);
				const int nx = WH_EXTRAS(head);
				for(int k = 0; k < nx; ++k) {
					const int *E = wext + ((size_t)e0 + k) * A2D_WIN_WORDS;
					const unsigned h = (unsigned)E[WE_HEAD];
					if(h & WH_FRESH)
						d1 = d2 = 0;
					winf_filter(row, WH_OFF(h), WH_LEN(h), E[FW + WF_F0], E[FW + WF_DF], E[FW + WF_QV], E[FW + WF_QD],
							E[FW + WF_LP], E[FW + WF_BP], E[FW + WF_HP], d1, d2);
				}
			}
		} else {
			// (the last worker first: the bus sums of the fragment panned in the step before - every wavefront's row
			// asked for at once, then added up bus by bus: one after the other they were 5 400 cycles of LDS round trips a step)
			if(wv == nw - 1 && s >= 3) {
				const int pb = (s - 3) & 1, g = s - 3;
				const int po = lane < nw ? part_off[pb * nw + lane] : -2, pn = lane < nw ? part_nch[pb * nw + lane] : 2;
#pragma unroll
				for(int ch = 0; ch < 2; ++ch) {
					int pv[WINF_MAXW(NOSC)];
#pragma unroll
					for(int w = 1; w < WINF_MAXW(NOSC); ++w)
						pv[w] = w < nw ? part[((pb * nw + w) * 2 + ch) * 64 + lane] : 0;
					int sum = 0, off = -1, nch = 2;
#pragma unroll
					for(int w = 1; w <= WINF_MAXW(NOSC); ++w) {
						const int woff = w < nw ? rdl(po, w < WINF_MAXW(NOSC) ? w : 0) : -2;
						if(woff != off) {
							if(off >= 0 && sum && !(dbg & 1))
								atomicAdd(busmem + off + ((size_t)(fa + g) * nch + ch) * A2D_FRAG + lane, sum);
							sum = 0;
							off = woff;
							nch = w < nw ? rdl(pn, w < WINF_MAXW(NOSC) ? w : 0) : 2;
						}
						if(w < WINF_MAXW(NOSC) && w < nw && woff >= 0)
							sum = wadd(sum, pv[w]);
					}
				}
			}
			// what this step needs from memory: the first extras of fragment s, the slots of fragment s + 1
			const int nx = (mine && s < nfr) ? WH_EXTRAS((unsigned)S[WE_HEAD]) : 0;
			lane_load<NW>(wext + (size_t)wi * A2D_WIN_WORDS, nx > 0, X);
			if(s + 1 < nfr) {
				lane_load<NW>(wslot + ((size_t)(s + 1) * nlist + first + lo + lane) * SW, mine, Sn);
				if(mine)
					win = widx[(size_t)(s + 1) * nlist + first + lo + lane];
			}
			// pan stage of fragment s - 2 out of its tile (its words were kept two steps ago)
			if(s >= 2 && s - 2 < nfr) {
				const int g = s - 2, pb = g & 1;
				const int *const tile = tiles + (g % 3) * vpg * WINF_PITCH;
				int a0 = 0, a1 = 0, cur_off = rdl(l_off, 0), cur_nch = rdl(l_nch, 0);
				auto pan_of = [&](const PanW &P, int k) {
					const unsigned h = (unsigned)rdl(P.w[WE_HEAD], k);
					const int fl = lane - WH_OFF(h);
					const bool in = (unsigned)fl < (unsigned)WH_LEN(h);
					const int y = in ? tile[(lo + k) * WINF_PITCH + lane] : 0;
					win_pan(h, rdl(P.w[WE_VOL], k), rdl(P.w[WE_DVOL], k), rdl(P.w[WE_PAN], k), rdl(P.w[WE_DPAN], k), y, fl, in, a0, a1);
				};
				for(int k = 0; k < hi - lo; ++k) {
					const int voff = rdl(l_off, k);
					if(voff != cur_off) {
						bus_add(busmem, cur_off, cur_nch, fa + g, lane, dbg, a0, a1);
						cur_off = voff;
						cur_nch = rdl(l_nch, k);
					}
					const unsigned h = (unsigned)rdl(P2.w[WE_HEAD], k);
					if(WH_LEN(h))
						pan_of(P2, k);
					const int n = WH_EXTRAS(h);
					if(n) {
						pan_of(PX2, k);
						const unsigned e0 = (unsigned)rdl((int)wi2, k);
						for(int q = 1; q < n; ++q) {
							const ScalAcc<5> E = bcast_load<5>(wext + ((size_t)e0 + q) * A2D_WIN_WORDS);
							const unsigned hq = (unsigned)E(WE_HEAD);
							const int fl = lane - WH_OFF(hq);
							const bool in = (unsigned)fl < (unsigned)WH_LEN(hq);
							const int y = in ? tile[(lo + k) * WINF_PITCH + lane] : 0;
							win_pan(hq, E(WE_VOL), E(WE_DVOL), E(WE_PAN), E(WE_DPAN), y, fl, in, a0, a1);
						}
					}
				}
				part[((pb * nw + wv) * 2 + 0) * 64 + lane] = a0;
				part[((pb * nw + wv) * 2 + 1) * 64 + lane] = a1;
				if(lane == 0) {
					part_off[pb * nw + wv] = hi > lo ? cur_off : -1;
					part_nch[pb * nw + wv] = cur_nch;
				}
			}
			// oscillators of fragment s into its tile: the slots, then the first extras (whose loads have had
			// the pan stage's time to arrive), the rest one by one
			if(s < nfr) {
				int *const tile = tiles + (s % 3) * vpg * WINF_PITCH;
				// (as win_run: the entries whose oscillator part is plain - WH_PLAINOSC, a2amd_winctl.h - in a pipelined loop that
				// knows they are, then the others; no branch on the kind between an entry's loads and the next one's)
				auto oscs_kind = [&](const int (&R)[NW], unsigned long long mask, auto plain_c) {
					constexpr int PL = decltype(plain_c)::value;
					if(!mask)
						return;
					int k = (int)__builtin_ctzll(mask);
					mask &= mask - 1;
					WinTaps<NOSC> T0;
					{
						const LaneAcc<NW> E = { R, k };
						win_taps_issue<NOSC, LaneAcc<NW>, PL>(E, rs, lane, T0);
					}
					for(;;) {
						const int kn = mask ? (int)__builtin_ctzll(mask) : -1;
						WinTaps<NOSC> T1;
						if(kn >= 0) {
							const LaneAcc<NW> En = { R, kn };
							win_taps_issue<NOSC, LaneAcc<NW>, PL>(En, rs, lane, T1);
						}
						if(PL) {
							int x = 0;
#pragma unroll
							for(int o = 0; o < NOSC; ++o)
								x = wadd(x, mul64s(hermite_c(T0.k1[o], T0.t1[o]) + hermite_c(T0.k2[o], T0.t2[o]), T0.ak[o], 17));
							// (PL == 2, a cut window: the lanes outside it write the row's padding cell - WINF_PITCH is 65 -
							// instead of a store under an exec mask)
							const int col = PL == 1 || (unsigned)(lane - WH_OFF(T0.head)) < (unsigned)WH_LEN(T0.head) ? lane : A2D_FRAG;
							tile[(lo + k) * WINF_PITCH + col] = x;
						} else {
							const LaneAcc<NW> E = { R, k };
							const int fl = lane - WH_OFF(T0.head);
							const bool in = (unsigned)fl < (unsigned)WH_LEN(T0.head);
							const int x = win_oscs_finish<NOSC>(E, T0, fl, in);
							if(in)
								tile[(lo + k) * WINF_PITCH + lane] = x;
						}
						if(kn < 0)
							break;
						mask &= mask - 1;
						k = kn;
						T0 = T1;
					}
				};
				auto oscs = [&](const int (&R)[NW], unsigned long long mask) {
					const unsigned long long plain = mask & __ballot(((unsigned)R[WE_HEAD] & WH_PLAINOSC) != 0);
					const unsigned long long whole = plain & __ballot(((unsigned)R[WE_HEAD] & 0x1fffu) == ((unsigned)A2D_FRAG << 6));
					oscs_kind(R, whole, std::integral_constant<int, 1>{});
					oscs_kind(R, plain & ~whole, std::integral_constant<int, 2>{});
					oscs_kind(R, mask & ~plain, std::integral_constant<int, 0>{});
				};
				oscs(S, __ballot(mine && WH_LEN((unsigned)S[WE_HEAD]) != 0));
				oscs(X, __ballot(nx > 0));
				unsigned long long more = __ballot(nx > 1);
				while(more) {
					const int k = (int)__builtin_ctzll(more);
					more &= more - 1;
					const int n = rdl(nx, k);
					const unsigned e0 = (unsigned)rdl((int)wi, k);
					for(int q = 1; q < n; ++q) {
						const ScalAcc<NW> E = bcast_load<NW>(wext + ((size_t)e0 + q) * A2D_WIN_WORDS);
						WinTaps<NOSC> T;
						win_taps_issue<NOSC, ScalAcc<NW>, 0>(E, rs, lane, T);
						const int fl = lane - WH_OFF(T.head);
						const bool in = (unsigned)fl < (unsigned)WH_LEN(T.head);
						const int x = win_oscs_finish<NOSC>(E, T, fl, in);
						if(in)
							tile[(lo + k) * WINF_PITCH + lane] = x;
					}
				}
			}
			// the steps move on: this fragment's pan words are kept for the step after next
			P2 = P1;
			PX2 = PX1;
			wi2 = wi1;
#pragma unroll
			for(int k = 0; k < 5; ++k) {
				P1.w[k] = s < nfr ? S[k] : 0;
				PX1.w[k] = s < nfr ? X[k] : 0;
			}
			wi1 = wi;
			if(s + 1 < nfr) {
#pragma unroll
				for(int k = 0; k < NW; ++k)
					S[k] = Sn[k];
				wi = win;
			} else
				S[WE_HEAD] = 0;
		}
		win_meet();
	}
	if(wv == 0 && lane < nv) {
		ustate[(size_t)ufilt * A2D_USTATE + FW_D1A] = d1;
		ustate[(size_t)ufilt * A2D_USTATE + FW_D2A] = d2;
	}
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
int a2d_launch_win_ctl(const A2DParams *dparams, const A2DParams &hp, int nosc, int filt, const int *dlist, int nlist,
		int skip_empty, int fa, int fb, int *wslot, int *wext, unsigned *widx, unsigned *wtop, unsigned wcap, int *wrc, void *stream)
{
	if(nlist <= 0 || fb <= fa)
		return 0;
	const int nblocks = (nlist + 63) / 64;
#define CTL_LAUNCH(N, F) hipLaunchKernelGGL((k_win_ctl<N, F>), dim3(nblocks), dim3(128), 0, (hipStream_t)stream, dparams, dlist, \
		nlist, skip_empty, fa, fb, wslot, wext, widx, wtop, wcap, wrc, hp.voices, hp.ustate, hp.vactive, hp.waves, hp.ptab)
	if(nosc == 1 && !filt)
		CTL_LAUNCH(1, 0);
	else if(nosc == 2 && !filt)
		CTL_LAUNCH(2, 0);
	else if(nosc == 1)
		CTL_LAUNCH(1, 1);
	else
		CTL_LAUNCH(2, 1);
#undef CTL_LAUNCH
	return (int)hipGetLastError();
}

int a2d_launch_win_render(const A2DParams &hp, int nosc, int filt, const int *dlist, int nlist, int fa, int fb,
		const int *wslot, const int *wext, const unsigned *widx, void *stream)
{
	if(nlist <= 0 || fb <= fa)
		return 0;
	if(!filt) {
		// voices per wavefront (one per lane at most): about 8 192 wavefronts in the launch
		const int nchunks = (fb - fa + WIN_FCH - 1) / WIN_FCH;
		static const int force = getenv("A2AMD_WVPW") ? atoi(getenv("A2AMD_WVPW")) : 0;
		int vpw = force > 0 ? force : (int)std::min<long long>(std::max<long long>(((long long)nlist * nchunks + 8191) / 8192, 1), 64);
		vpw = std::min(std::max(vpw, 1), 64);
		const int ngroups = (nlist + vpw - 1) / vpw;
		const int nblocks = nchunks * ((ngroups + WIN_WPB - 1) / WIN_WPB);
		if(nosc == 1)
			hipLaunchKernelGGL((k_win_render<1>), dim3(nblocks), dim3(64 * WIN_WPB), 0, (hipStream_t)stream, dlist, nlist, vpw,
					fa, fb, wslot, wext, widx, hp.voices, hp.wavecoef, hp.busmem, hp.debug);
		else
			hipLaunchKernelGGL((k_win_render<2>), dim3(nblocks), dim3(64 * WIN_WPB), 0, (hipStream_t)stream, dlist, nlist, vpw,
					fa, fb, wslot, wext, widx, hp.voices, hp.wavecoef, hp.busmem, hp.debug);
	} else {
		// voices per workgroup = lanes of its filter wavefront: spread out until every CU has a couple of
		// workgroups (a workgroup takes as long as its filter chain whatever its voice count), then fill up
		static const int force = getenv("A2AMD_WFVPG") ? atoi(getenv("A2AMD_WFVPG")) : 0;
		const int maxw = WINF_MAXW(nosc);
		// (measured at 16 384 voices x 64 fragments: one oscillator 32 per workgroup 0.63 ms, 48 - a workgroup and a half
		// per CU - 0.77; two oscillators 64 per workgroup 0.94, 32 1.03, 60 - 274 workgroups on 256 CUs - 1.49)
		int vpg = force > 0 ? force : nosc == 1 ? std::min(std::max((nlist + 511) / 512, 1), 48) : std::min(std::max((nlist + 255) / 256, 1), 64);
		vpg = std::min(std::max(vpg, 1), 64);
		// wavefronts: the filter's + one per ~4 voices
		static const int forcew = getenv("A2AMD_WFWAVES") ? atoi(getenv("A2AMD_WFWAVES")) : 0;
		int nw = forcew > 1 ? forcew : 1 + std::min(std::max((vpg + 3) / 4, 1), 7);
		nw = std::min(std::max(nw, 2), maxw);
		// (three tiles of vpg rows + the wavefronts' bus sums: within 64 KB of LDS)
		while(vpg > 1 && (size_t)(3 * vpg * WINF_PITCH + 2 * nw * 2 * 64 + 2 * nw * 2) * sizeof(int) > 65536)
			--vpg;
		const int nblocks = (nlist + vpg - 1) / vpg;
		const size_t dyn = (size_t)(3 * vpg * WINF_PITCH + 2 * nw * 2 * 64 + 2 * nw * 2) * sizeof(int);
		if(nosc == 1)
			hipLaunchKernelGGL((k_win_render_f<1>), dim3(nblocks), dim3(64 * nw), dyn, (hipStream_t)stream, dlist, nlist, vpg,
					fa, fb, wslot, wext, widx, hp.voices, hp.ustate, hp.wavecoef, hp.busmem, hp.debug);
		else
			hipLaunchKernelGGL((k_win_render_f<2>), dim3(nblocks), dim3(64 * nw), dyn, (hipStream_t)stream, dlist, nlist, vpg,
					fa, fb, wslot, wext, widx, hp.voices, hp.ustate, hp.wavecoef, hp.busmem, hp.debug);
	}
	return (int)hipGetLastError();
}
