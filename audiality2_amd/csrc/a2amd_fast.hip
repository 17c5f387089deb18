// a2amd_fast.hip - specialised kernels for voices that are "quiet" in a batch
// (no command records: every fragment is the engine's default
// Process(0, frames) on each unit).  They read and write the same device state
// as the general kernel (a2amd_kernels.hip), so a voice can move between the
// two from one batch to the next; results are bit-identical by construction
// and by test (tests/test_gpu_parity.py runs both).
//
//   k_leaf_oscpan   wtosc (mip-mapped wave) -> panmix 1->2, the BASELINE
//                   config 2 voice.  A wavefront owns up to 64 voices.  Lane =
//                   sample frame while rendering, so everything per voice is
//                   wave-uniform: the voice state is parked one voice per lane
//                   in 22 VGPRs and pulled into SGPRs with v_readlane when its
//                   turn comes, and rampers, pitch table lookup, mip selection
//                   and phase wrap run on the scalar unit while the vector
//                   unit does the two Hermite taps, the amplitude and the pan
//                   multiplies.  Wave data is fetched as three aligned dwords
//                   per frame (five int16 cover both taps) from addresses that
//                   are consecutive across lanes.  The voices of a wavefront
//                   are summed in registers and reach the bus with one atomic
//                   per (wavefront, fragment, channel, frame).
//   k_bus_driver    inline -> panmix 2->2 -> xinsert: the engine's root and
//                   group voices (audiality2.c:271-302).  One workgroup per
//                   voice; the fragments of the batch are independent once the
//                   two rampers have been stepped through them, so thread 0
//                   steps the rampers and the wavefronts then split the
//                   fragments.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include <stdlib.h>
#include <type_traits>
#include "a2amd_device.h"
#include "a2amd_dsp.h"
#include "a2amd_fm.h"

#include "a2amd_taps.h"

__global__ void k_build_coef(const int16_t *__restrict__ pool, int *__restrict__ coef, unsigned lo, unsigned hi)
{
	const unsigned j = lo + blockIdx.x * blockDim.x + threadIdx.x;
	if(j >= hi)
		return;
	const int dm = pool[j - 1], d0 = pool[j], d1 = pool[j + 1], d2 = pool[j + 2];
	const int c = (d1 - dm) >> 1;
	const int a = (3 * (d0 - d1) + d2 - dm) >> 1;
	const int b = dm - d0 + c - a;
	int *e = coef + A2D_COEF_WORDS * (size_t)j;
	e[0] = a;
	e[1] = b;
#if A2D_COEF_WORDS == 4
	e[2] = c;
	e[3] = d0;
#else
	e[2] = (int)(((unsigned)c << 16) | ((unsigned)d0 & 0xffffu));
#endif
}

// coefficient entries for pool samples [lo, hi): needs pool[lo - 1 .. hi + 1]
int a2d_launch_build_coef(const int16_t *pool, int *coef, unsigned lo, unsigned hi, void *stream)
{
	if(hi <= lo)
		return 0;
	hipLaunchKernelGGL(k_build_coef, dim3((hi - lo + 255) / 256), dim3(256), 0, (hipStream_t)stream, pool, coef, lo, hi);
	return (int)hipGetLastError();
}

// ---- cold paths: kept out of line so the hot loop stays small -------------
// (all operands are wave-uniform; results go back to SGPRs via readfirstlane)
__device__ __attribute__((noinline)) int cold_ramp_delta(int target, int value, int timer, int frames,
		int *newtimer)
{
	// the two ramping branches of a2_PrepareRamper, a2_dsp.h:134-148
	if(frames <= (timer >> 8)) {
		*newtimer = wsub(timer, frames << 8);
		if(timer > 0)
			return (int)div_trunc_exact((int64_t)wsub(target, value) * 256, timer);
		return (int)((((int64_t)wsub(target, value)) * 256) / timer);
	}
	*newtimer = 0;
	if(frames > 0)
		return (int)div_trunc_exact((int64_t)wsub(target, value), frames);
	return wsub(target, value) / frames;
}

__device__ __attribute__((noinline)) uint64_t cold_umod64(uint64_t a, uint64_t m)
{
	return a % m;
}

// ph %= (uint64_t)size << 24 (wtosc.c:260-263) for wave-uniform operands.  The
// low 24 bits pass through, so this is a 32 bit modulo of the sample index; the
// built-in waves and their mip levels have power-of-two sizes (a mask).
DEV uint64_t wrap_phase(uint64_t ph, unsigned size)
{
	if(ph >> 56) {		// never in practice: index beyond 32 bits
		uint64_t r = cold_umod64(ph, (uint64_t)size << 24);
		return (uint64_t)(unsigned)rfl((int)(unsigned)r) | ((uint64_t)(unsigned)rfl((int)(unsigned)(r >> 32)) << 32);
	}
	unsigned hi = (unsigned)(ph >> 24);
	if(hi >= size) {
		if(!(size & (size - 1)))
			hi &= size - 1;
		else
			hi = (unsigned)rfl((int)(hi % size));
		ph = ((uint64_t)hi << 24) | (ph & 0xffffffu);
	}
	return ph;
}

// a2_PrepareRamper with a scalar fast path for the finished ramp
DEV void ramp_prepare_s(Ramp &r, int frames)
{
	if(r.timer == 0) {
		r.value = r.target;
		r.delta = 0;
	} else {
		int nt;
		int d = cold_ramp_delta(r.target, r.value, r.timer, frames, &nt);
		r.delta = rfl(d);
		r.timer = rfl(nt);
	}
}

struct OscS {		// A2_wtosc, wave-uniform copy
	int mode, wave;
	unsigned dphase;
	uint64_t phase;
	int p_ramping;
	Ramp p, a;
	int noise;		// (wtosc_noise only: the sample held between draws, and the engine RNG word
	unsigned seed;		// the host recorded for the window, R_NOISESEED - k_leaf_recs)
};

struct FastPtrs {
	const int16_t *wavepool;
	const A2DWave *waves;
	const uint32_t *ptab;
	int dbg;
};

// One default fragment of wtosc, scalar control + vector frames (frame = lane):
// wtosc_wavetable (wtosc.c:239-286) / wtosc_Off (:108-126).  Returns the sample
// the oscillator leaves in the scratch buffer for this lane.  Everything named
// o.* lives in SGPRs.
// (lane = index of the frame within the window: lanes before a window that starts
// inside the fragment come in negative, k_leaf_recs)
// One fragment (or window) of an oscillator in two halves: osc_fragment_begin() steps the state and
// ISSUES the wave data loads, osc_fragment_end() interpolates.  A voice with two oscillators begins
// both before it ends either - on a wavefront that has its SIMD to itself (a song's few voices) the
// second oscillator's table look-ups and sample loads no longer wait for the first one's to return.
struct OscPend { Quad16 qa, qb; unsigned ph, ph2; int ak, x; bool taps; };

DEV int osc_fragment_end(const OscPend &pd)
{
	if(pd.taps)
		return mul64s(inter_quads(pd.qa, pd.qb, pd.ph, pd.ph2), pd.ak, 17);
	return pd.x;
}

DEV OscPend osc_fragment_begin(const FastPtrs &g, OscS &o, int nframes, int lane)
{
	OscPend pd;
	pd.qa.lo = pd.qa.hi = pd.qb.lo = pd.qb.hi = 0;
	pd.ph = pd.ph2 = 0;
	pd.ak = 0;
	pd.taps = false;
	int x = 0;
	const bool in = (unsigned)lane < (unsigned)nframes;
	if(o.mode == A2D_OSC_MIPWAVE) {
		const A2DWave *w = g.waves + o.wave;
		const unsigned size0 = w->size[0], period = w->period, flags = w->flags;
		if(!size0) {		// wtosc_check_unloaded, wtosc.c:168-183
			o.wave = -1;
			o.mode = A2D_OSC_OFF;
		} else {
			// wtosc_run_pitch, wtosc.c:89-105
			ramp_prepare_s(o.p, nframes);
			if(!(o.dphase && (!o.p.timer && !o.p_ramping))) {
				unsigned lastv = (unsigned)o.p.value;
				ramp_run(o.p, nframes);
				o.p_ramping = o.p.delta;
				o.dphase = (unsigned)rfl((int)p2i(g.ptab, (int)((lastv + (unsigned)o.p.value) >> 9)));
			}
			unsigned dph = ((o.dphase + 255) >> 8) * period;
			ramp_prepare_s(o.a, nframes);
			unsigned mm = 0;
			for(; (dph > (A2D_MAXPHINC << 8)) && (mm < A2D_MIPS - 1); ++mm)
				dph >>= 1;
			uint64_t ph = o.phase >> mm;
			dph = (unsigned)(((uint64_t)o.dphase * period) >> mm);
			const unsigned sizem = w->size[mm];
			bool play = true;
			if(flags & 0x100u) {
				ph = wrap_phase(ph, sizem);
			} else if((ph >> 24) > (uint64_t)(sizem + 1))
				play = false;	// all played: silence, state untouched
			if(play) {
				if(dph <= (A2D_MAXPHINC << 16)) {
					const int16_t *d = g.wavepool + w->off[mm];
					if(in) {
						// (inter_dwords, its loads issued here and its arithmetic left to osc_fragment_end)
						uint64_t phk = ph + (uint64_t)(unsigned)lane * dph;
						pd.ak = wadd(o.a.value, wmul(o.a.delta, lane));
						pd.ph = (unsigned)(phk >> 16);
						pd.ph2 = pd.ph + ((dph >> 16) >> 1);
						pd.qa = *(const Quad16 *)(d + (int)(pd.ph >> 8) - 1);
						pd.qb = *(const Quad16 *)(d + (int)(pd.ph2 >> 8) - 1);
						pd.taps = true;
					}
				}
				ph += (uint64_t)dph * (unsigned)nframes;
				o.phase = ph << mm;
				ramp_run(o.a, nframes);
			}
		}
	} else if(o.mode == A2D_OSC_NOISE) {	// wtosc_noise, wtosc.c:129-152
		// wtosc_run_pitch
		ramp_prepare_s(o.p, nframes);
		if(!(o.dphase && (!o.p.timer && !o.p_ramping))) {
			unsigned lastv = (unsigned)o.p.value;
			ramp_run(o.p, nframes);
			o.p_ramping = o.p.delta;
			o.dphase = (unsigned)rfl((int)p2i(g.ptab, (int)((lastv + (unsigned)o.p.value) >> 9)));
		}
		ramp_prepare_s(o.a, nframes);
		// a frame draws from the engine's one LCG when its step crosses a 2^23 boundary of the
		// phase: which draw a lane holds is a prefix count, the draws themselves a uniform loop
		const uint64_t phk = o.phase + (uint64_t)(unsigned)(in ? lane : 0) * o.dphase;
		const uint64_t nph = phk + o.dphase;
		const bool draw = in && ((o.dphase >= (1u << 23)) || ((nph ^ phk) >> 23));
		const unsigned long long m = __ballot(draw);
		const int me = (int)(threadIdx.x & 63);
		const unsigned long long below = (me >= 63) ? ~0ull : ((2ull << me) - 1ull);
		const int mine = __popcll(m & below), total = __popcll(m);
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
			int ak = wadd(o.a.value, wmul(o.a.delta, lane));
			x = wmul(myval, ak >> 10) >> 6;
		}
		o.phase += (uint64_t)(unsigned)nframes * o.dphase;
		ramp_run(o.a, nframes);
	} else {		// wtosc_Off, wtosc.c:108-126
		ramp_prepare_s(o.p, nframes);
		ramp_prepare_s(o.a, nframes);
		ramp_run(o.p, nframes);
		ramp_run(o.a, nframes);
	}
	pd.x = x;
	return pd;
}

DEV int osc_fragment_s(const FastPtrs &g, OscS &o, int nframes, int lane)
{
	const OscPend pd = osc_fragment_begin(g, o, nframes, lane);
	return osc_fragment_end(pd);
}

// One default fragment of panmix 1->2 adding into the voice's output bus
// (panmix_Process12Add / panmix_process12, panmix.c:78-125) for input x.
DEV void pan_fragment_s(Ramp &vol, Ramp &pan, int x, int nframes, int lane, int &acc0, int &acc1)
{
	const bool clamp = pan.target > 0xffffff || pan.target < -0xffffff ||
			pan.value > 0xffffff || pan.value < -0xffffff;
	ramp_prepare_s(vol, nframes);
	ramp_prepare_s(pan, nframes);
	if((unsigned)lane < (unsigned)nframes) {
		int vk = wadd(vol.value, wmul(vol.delta, lane));
		int pk = wadd(pan.value, wmul(pan.delta, lane));
		int vp = mul64s(pk, vk, 24);
		int v0 = wsub(vk, vp), v1 = wadd(vk, vp);
		if(clamp) {
			int lim = wshl(vk, 1);
			if(v0 > lim) v0 = lim;
			if(v1 > lim) v1 = lim;
		}
		acc0 = wadd(acc0, mul64s(x, v0, 24));
		acc1 = wadd(acc1, mul64s(x, v1, 24));
	}
	ramp_run(vol, nframes);
	ramp_run(pan, nframes);
}

DEV void oscpan_fragment(const FastPtrs &g, OscS &o, Ramp &vol, Ramp &pan,
		int nframes, int lane, int &acc0, int &acc1)
{
	int x = osc_fragment_s(g, o, nframes, lane);
	pan_fragment_s(vol, pan, x, nframes, lane, acc0, acc1);
}

// frames of fragment f from the per-lane table (f is wave-uniform)
DEV int frames_of(const int (&ffr)[A2D_MAXBATCH / 64], int f)
{
	int r = 0;
#pragma unroll
	for(int k = 0; k < A2D_MAXBATCH / 64; ++k)
		if((f >> 6) == k)
			r = rdl(ffr[k], f & 63);
	return r;
}

// state words of one wtosc->panmix voice, kept one voice per lane in VGPRs
enum { SV_MODE = 0, SV_WAVE, SV_DPHASE, SV_PHLO, SV_PHHI, SV_PRAMP, SV_P = 6, SV_A = 10,
	SV_VOL = 14, SV_PAN = 18, SV_NWORDS = 22 };

// Extra per-lane words derived once per launch from a voice's state
enum { DV_SETTLED = 0, DV_MM, DV_DPH, DV_SIZEM, DV_DOFF, DV_V0, DV_V1, DV_NWORDS };

// an oscillator's state from / to the unit's state words in memory (wave-uniform address)
// (volatile: what this wavefront stored a chunk ago must come from memory, not from a scalar or
// vector cache line fetched before)
DEV void osc_from_mem(OscS &o, const volatile int *w)
{
	o.mode = rfl(w[OW_MODE]);
	o.wave = rfl(w[OW_WAVE]);
	o.dphase = (unsigned)rfl(w[OW_DPHASE]);
	o.phase = (uint64_t)(unsigned)rfl(w[OW_PHASE_LO]) | ((uint64_t)(unsigned)rfl(w[OW_PHASE_HI]) << 32);
	o.p_ramping = rfl(w[OW_PRAMPING]);
	o.p.value = rfl(w[OW_P]); o.p.target = rfl(w[OW_P + 1]);
	o.p.delta = rfl(w[OW_P + 2]); o.p.timer = rfl(w[OW_P + 3]);
	o.a.value = rfl(w[OW_A]); o.a.target = rfl(w[OW_A + 1]);
	o.a.delta = rfl(w[OW_A + 2]); o.a.timer = rfl(w[OW_A + 3]);
	o.noise = 0;
	o.seed = 0;
}

DEV void osc_to_mem(int *w, const OscS &o)
{
	w[OW_MODE] = o.mode;
	w[OW_WAVE] = o.wave;
	w[OW_DPHASE] = (int)o.dphase;
	w[OW_PHASE_LO] = (int)(unsigned)o.phase;
	w[OW_PHASE_HI] = (int)(unsigned)(o.phase >> 32);
	w[OW_PRAMPING] = o.p_ramping;
	w[OW_P] = o.p.value; w[OW_P + 1] = o.p.target; w[OW_P + 2] = o.p.delta; w[OW_P + 3] = o.p.timer;
	w[OW_A] = o.a.value; w[OW_A + 1] = o.a.target; w[OW_A + 2] = o.a.delta; w[OW_A + 3] = o.a.timer;
}

#ifndef OSC1_WPE
#define OSC1_WPE 4
#endif
__global__ __launch_bounds__(64 * FAST_WPB) __attribute__((amdgpu_waves_per_eu(OSC1_WPE, OSC1_WPE)))
void k_leaf_oscpan(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int vpw,
		int ysplit, const A2DVoice *__restrict__ voices, const int *ustate,
		int *ustage, const int16_t *__restrict__ wavepool,
		const A2DWave *__restrict__ waves, const uint32_t *__restrict__ ptab, int *__restrict__ busmem,
		const int *__restrict__ wavecoef)
{
	const A2DParams &p = *pp;
	const int wv = rfl((int)(threadIdx.x >> 6));	// (wave-uniform, and known to the compiler as such)
	const int lane = threadIdx.x & 63;
	const int first = (blockIdx.x * FAST_WPB + wv) * vpw;
	if(first >= nlist)
		return;
	const int nv = min(vpw, nlist - first);
	const int nfrags = p.nfrags;
	const int dbg = p.debug;
	FastPtrs g = { wavepool, waves, ptab, dbg };
	const CoefRsrc crs = coef_rsrc(wavecoef);

	// this wavefront's slice of the batch, in chunks of FAST_FCH fragments
	const int nchunks = (nfrags + FAST_FCH - 1) / FAST_FCH;
	const int per = (nchunks + ysplit - 1) / ysplit;
	const int slice = blockIdx.y;
	const int c_lo = slice * per, c_hi = min(nchunks, c_lo + per);
	const bool last_slice = c_hi >= nchunks;
	if(c_lo >= nchunks)
		return;
	// (the slices that exist; voices that still have something moving are dealt over
	// them - slice s walks the whole batch for every nslices-th of them)
	const int nslices = (nchunks + per - 1) / per;

	// fragment lengths and their prefix sums: lane l of ffr[k] / fst[k] holds
	// fragment 64*k + l (byte loads are vector memory operations: once, here)
	int ffr[A2D_MAXBATCH / 64], fst[A2D_MAXBATCH / 64];
#pragma unroll
	for(int k = 0; k < A2D_MAXBATCH / 64; ++k) {
		ffr[k] = (k * 64 + lane < nfrags) ? p.fragframes[k * 64 + lane] : 0;
		fst[k] = (k * 64 + lane < nfrags) ? p.fragstart[k * 64 + lane] : 0;
	}

	// lane v keeps voice v: its unit ids, output bus and what the settled loop needs - amplitude,
	// phase, and what follows from the state words for a settled voice (dv).  The other state words
	// are read here to decide whether the voice is settled, and again, from memory, by the loop for
	// voices that are not (k_leaf_osc2pan: the registers they took cost spills in the loop that matters).
	int dv[DV_NWORDS], amp_l = 0, phlo_l = 0, phhi_l = 0;
	int u0 = 0, u1 = 0, my_off = -1, my_nch = 2;
	bool mine = false;	// this lane's voice is ours to render (no records this batch)
#pragma unroll
	for(int k = 0; k < DV_NWORDS; ++k)
		dv[k] = 0;
	if(lane < nv)
		mine = p.runs[list[first + lane]].count == 0;
	if(mine) {
		const A2DVoice &vc = voices[list[first + lane]];
		u0 = vc.unit[0];
		u1 = vc.unit[1];
		my_off = vc.out_off;
		my_nch = vc.out_nch;
		const int *w0 = ustate + (size_t)u0 * A2D_USTATE;
		const int *w1 = ustate + (size_t)u1 * A2D_USTATE;
		int sv[SV_NWORDS];
		sv[SV_MODE] = w0[OW_MODE]; sv[SV_WAVE] = w0[OW_WAVE]; sv[SV_DPHASE] = w0[OW_DPHASE];
		sv[SV_PHLO] = w0[OW_PHASE_LO]; sv[SV_PHHI] = w0[OW_PHASE_HI]; sv[SV_PRAMP] = w0[OW_PRAMPING];
#pragma unroll
		for(int k = 0; k < 4; ++k) {
			sv[SV_P + k] = w0[OW_P + k];
			sv[SV_A + k] = w0[OW_A + k];
			sv[SV_VOL + k] = w1[PW_VOL + k];
			sv[SV_PAN + k] = w1[PW_PAN + k];
		}
		// Settled voice: every ramper has arrived (a2_PrepareRamper would
		// change nothing) and the pitch is constant, so over the whole batch
		// only the phase moves; mip level, increment and the two pan gains
		// are fixed.  Anything else takes the per-fragment path.
		bool settled = sv[SV_MODE] == A2D_OSC_MIPWAVE && sv[SV_DPHASE] && !sv[SV_PRAMP] &&
				!(sv[SV_P + 3] | sv[SV_P + 2] | sv[SV_A + 3] | sv[SV_A + 2] |
				  sv[SV_VOL + 3] | sv[SV_VOL + 2] | sv[SV_PAN + 3] | sv[SV_PAN + 2]) &&
				sv[SV_P] == sv[SV_P + 1] && sv[SV_A] == sv[SV_A + 1] &&
				sv[SV_VOL] == sv[SV_VOL + 1] && sv[SV_PAN] == sv[SV_PAN + 1];
		if(settled) {
			const A2DWave *w = waves + sv[SV_WAVE];
			const unsigned period = w->period, dphase = (unsigned)sv[SV_DPHASE];
			unsigned dph = ((dphase + 255) >> 8) * period, mm = 0;	// wtosc.c:250-258
			for(; (dph > (A2D_MAXPHINC << 8)) && (mm < A2D_MIPS - 1); ++mm)
				dph >>= 1;
			dph = (unsigned)(((uint64_t)dphase * period) >> mm);
			settled = w->size[0] && (w->flags & 0x100u) && dph <= (A2D_MAXPHINC << 16);
			dv[DV_MM] = (int)mm;
			dv[DV_DPH] = (int)dph;
			dv[DV_SIZEM] = (int)w->size[mm];
			dv[DV_DOFF] = (int)w->off[mm];
			// panmix_process12 gains (panmix.c:89-104)
			const int vol = sv[SV_VOL], pan = sv[SV_PAN];
			const int vp = mul64s(pan, vol, 24);
			int v0 = wsub(vol, vp), v1 = wadd(vol, vp);
			if(pan > 0xffffff || pan < -0xffffff) {
				int lim = wshl(vol, 1);
				if(v0 > lim) v0 = lim;
				if(v1 > lim) v1 = lim;
			}
			dv[DV_V0] = v0;
			dv[DV_V1] = v1;
		}
		// (a wave the host marked as too big to stay in the caches as coefficient entries - 12 bytes per
		// sample and tap against 2 - is read as samples: A2D_WF_RAWTAPS, a2amd_wave_upload)
		dv[DV_SETTLED] = settled ? ((sv[SV_MODE] == A2D_OSC_MIPWAVE && (waves[sv[SV_WAVE]].flags & A2D_WF_RAWTAPS)) ? 3 : 1) : 0;
		amp_l = sv[SV_A];
		phlo_l = sv[SV_PHLO];
		phhi_l = sv[SV_PHHI];
	}
	const unsigned long long unsettled_mask = __ballot(mine && !dv[DV_SETTLED]);
	const unsigned long long settled_mask = __ballot(mine && dv[DV_SETTLED]);
	const unsigned long long raw_mask = __ballot(mine && dv[DV_SETTLED] == 3);

	// ---- settled voices: this slice's chunks ---------------------------------
	for(int c = c_lo; c < c_hi; ++c) {
		const int f0 = c * FAST_FCH;
		const int nf = min((int)FAST_FCH, nfrags - f0);
		int acc0[FAST_FCH], acc1[FAST_FCH];
#pragma unroll
		for(int j = 0; j < FAST_FCH; ++j)
			acc0[j] = acc1[j] = 0;
		int nfr[FAST_FCH];	// frames of the chunk's fragments (SGPRs)
#pragma unroll
		for(int j = 0; j < FAST_FCH; ++j)
			nfr[j] = (j < nf) ? frames_of(ffr, f0 + j) : 0;
		const unsigned before = (unsigned)frames_of(fst, f0);
		// frames from the start of the chunk to each of its fragments (SGPRs)
		unsigned pre[FAST_FCH];
		pre[0] = 0;
#pragma unroll
		for(int j = 1; j < FAST_FCH; ++j)
			pre[j] = pre[j - 1] + (unsigned)nfr[j - 1];
		// where each lane's voice stands when the chunk begins: one vector
		// multiply-add per lane instead of a scalar one per voice
		const uint64_t lbase = ((((uint64_t)(unsigned)phlo_l) | ((uint64_t)(unsigned)phhi_l << 32)) >>
				(unsigned)dv[DV_MM]) + (uint64_t)before * (unsigned)dv[DV_DPH];
		const int base_lo = (int)(unsigned)lbase, base_hi = (int)(unsigned)(lbase >> 32);
		int cur_off = rdl(my_off, 0), cur_nch = rdl(my_nch, 0);
		for(int v = 0; v < nv; ++v) {
			if(!((settled_mask >> v) & 1ull))
				continue;
			const int voff = rdl(my_off, v);
			if(voff != cur_off) {
				flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
				cur_off = voff;
				cur_nch = rdl(my_nch, v);
			}
			const unsigned dph = (unsigned)rdl(dv[DV_DPH], v);
			const unsigned sizem = (unsigned)rdl(dv[DV_SIZEM], v), doff = (unsigned)rdl(dv[DV_DOFF], v);
			const int v0 = rdl(dv[DV_V0], v), v1 = rdl(dv[DV_V1], v), amp = rdl(amp_l, v);
			const unsigned dph16 = dph >> 16;
			// The phase fragment f starts from is ((phase >> mm) + frames_before(f) * dph)
			// mod (size << 24): the reference's "ph %= size << 24; ...; ph += frames * dph"
			// per fragment (wtosc.c:259-285) is addition mod size << 24.
			uint64_t ph = (uint64_t)(unsigned)rdl(base_lo, v) | ((uint64_t)(unsigned)rdl(base_hi, v) << 32);
			uint64_t phs[FAST_FCH];
			if(!(sizem & (sizem - 1)) && !(ph >> 48)) {
				// a power-of-two size (the built-in waves and their mip levels):
				// the modulus is a mask - every fragment of the chunk straight
				// from the base, no chain, no branches
				const uint64_t mask = ((uint64_t)sizem << 24) - 1;
#pragma unroll
				for(int j = 0; j < FAST_FCH; ++j)
					phs[j] = (ph + (uint64_t)dph * pre[j]) & mask;
				ph = phs[FAST_FCH - 1] + (uint64_t)dph * (unsigned)nfr[FAST_FCH - 1];
			} else {
#pragma unroll
				for(int j = 0; j < FAST_FCH; ++j) {
					ph = wrap_phase(ph, sizem);
					phs[j] = ph;
					ph += (uint64_t)dph * (unsigned)nfr[j];
				}
			}
			// vector pass, branch free: all wave data loads of the chunk can be
			// in flight together (lanes past a short fragment read inside the
			// A2_WAVEPOST pad and are masked at the sum)
			// (lane * dph once per voice; opaque to the compiler, which otherwise
			// re-does the half-rate 64 bit multiply-add for every fragment)
			const unsigned ldph = lane_dph(lane, dph);
			// (uniform base = the coefficient entry of the level's first payload
			// sample, + an unsigned 32 bit byte offset per lane: the loads take the
			// scalar-base addressing form, no 64 bit adds.  Four fragments' entries
			// in flight at a time: 24 registers.)
			const int cb = coef_base(doff);
			if((raw_mask >> v) & 1ull) {
				// The north-star's case: a private sample wave that no cache holds.  The window a fragment
				// reads is 64 * dph / 2^16 + 5 int16 samples (<= 266 bytes); as coefficient entries
				// it would be six times that from HBM.  Two 8-byte loads per frame (the four samples of each
				// tap: neighbouring lanes share their cache lines) and a2_Hermite (a2_dsp.h:64-74) in full.
				const int16_t *d = wavepool + doff;
#pragma unroll
				for(int h = 0; h < FAST_FCH; h += 4) {
					Quad16 qa[4], qb[4];
					unsigned pa[4], pb[4];
#pragma unroll
					for(int j = 0; j < 4; ++j) {
						pa[j] = tap_phase(phs[h + j], ldph);
						pb[j] = pa[j] + (dph16 >> 1);
						qa[j] = *(const Quad16 *)(d + (int)(pa[j] >> 8) - 1);
						qb[j] = *(const Quad16 *)(d + (int)(pb[j] >> 8) - 1);
					}
#pragma unroll
					for(int j = 0; j < 4; ++j) {
						int sm = inter_quads(qa[j], qb[j], pa[j], pb[j]);
						int x = mul64s(sm, amp, 17);
						x = (lane < nfr[h + j]) ? x : 0;
						acc0[h + j] = wadd(acc0[h + j], mul64s(x, v0, 24));
						acc1[h + j] = wadd(acc1[h + j], mul64s(x, v1, 24));
					}
				}
			} else
#pragma unroll
			for(int h = 0; h < FAST_FCH; h += 4) {
				Coef4 ka[4], kb[4];
				unsigned pa[4], pb[4];
#pragma unroll
				for(int j = 0; j < 4; ++j) {
					pa[j] = tap_phase(phs[h + j], ldph);
					asm("" : "+v"(pa[j]));	// (keeps the offset a 32 bit value in the compiler's eyes)
					pb[j] = pa[j] + (dph16 >> 1);
					ka[j] = coef_at(crs, cb, pa[j]);
					kb[j] = coef_at(crs, cb, pb[j]);
				}
#pragma unroll
				for(int j = 0; j < 4; ++j) {
					int sm = hermite_c(ka[j], pa[j]) + hermite_c(kb[j], pb[j]);
					int x = mul64s(sm, amp, 17);
					x = (lane < nfr[h + j]) ? x : 0;
					acc0[h + j] = wadd(acc0[h + j], mul64s(x, v0, 24));
					acc1[h + j] = wadd(acc1[h + j], mul64s(x, v1, 24));
				}
			}
			if(last_slice && c == c_hi - 1) {
				// what the oscillator is left with after the batch: the
				// unwrapped end of the last fragment (wtosc.c:284)
				const bool me = lane == v;
				const uint64_t endph = ph << (unsigned)rdl(dv[DV_MM], v);
				WRL(phlo_l, (int)(unsigned)endph);
				WRL(phhi_l, (int)(unsigned)(endph >> 32));
			}
		}
		flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
	}

	// ---- voices with something still moving: one slice each walks the whole batch ----
	const unsigned long long todo_mask = unsettled_mask & __ballot(lane % nslices == slice);
	if(todo_mask) {
		for(int f0 = 0; f0 < nfrags; f0 += FAST_FCH) {
			const int nf = min((int)FAST_FCH, nfrags - f0);
			int acc0[FAST_FCH], acc1[FAST_FCH];
#pragma unroll
			for(int j = 0; j < FAST_FCH; ++j)
				acc0[j] = acc1[j] = 0;
			int cur_off = rdl(my_off, 0), cur_nch = rdl(my_nch, 0);
			for(int v = 0; v < nv; ++v) {
				if(!((todo_mask >> v) & 1ull))
					continue;
				const int voff = rdl(my_off, v);
				if(voff != cur_off) {
					flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
					cur_off = voff;
					cur_nch = rdl(my_nch, v);
				}
				// (its state from memory and back: k_leaf_osc2pan)
				OscS o;
				Ramp vol, pan;
				const int *src = f0 ? ustage : ustate;
				osc_from_mem(o, src + (size_t)rdl(u0, v) * A2D_USTATE);
				{
					const volatile int *wpv = src + (size_t)rdl(u1, v) * A2D_USTATE;
					vol.value = rfl(wpv[0]); vol.target = rfl(wpv[1]); vol.delta = rfl(wpv[2]); vol.timer = rfl(wpv[3]);
					pan.value = rfl(wpv[4]); pan.target = rfl(wpv[5]); pan.delta = rfl(wpv[6]); pan.timer = rfl(wpv[7]);
				}
				for(int j = 0; j < nf; ++j) {
					int o0 = 0, o1 = 0;
					oscpan_fragment(g, o, vol, pan, frames_of(ffr, f0 + j), lane, o0, o1);
#pragma unroll
					for(int jj = 0; jj < FAST_FCH; ++jj)
						if(jj == j) {
							acc0[jj] = wadd(acc0[jj], o0);
							acc1[jj] = wadd(acc1[jj], o1);
						}
				}
				if(lane == 0) {
					osc_to_mem(ustage + (size_t)rdl(u0, v) * A2D_USTATE, o);
					int *wpv = ustage + (size_t)rdl(u1, v) * A2D_USTATE;
					wpv[0] = vol.value; wpv[1] = vol.target; wpv[2] = vol.delta; wpv[3] = vol.timer;
					wpv[4] = pan.value; wpv[5] = pan.target; wpv[6] = pan.delta; wpv[7] = pan.timer;
				}
			}
			flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");	// (the next chunk reads what lane 0 stored)
			__builtin_amdgcn_s_waitcnt(0);
		}
	}

	// State out.  With one slice it goes straight back; with several the other
	// slices are still reading the old state, so it is staged and committed by
	// k_commit_oscpan afterwards.  The last slice owns the settled voices (their
	// end phase), the slice that walked it each of the others.
	// (a settled voice: as it was, the phase moved on; the others were stored by the loop above)
	if(mine && dv[DV_SETTLED] && last_slice) {
		const int *wi0 = ustate + (size_t)u0 * A2D_USTATE, *wi1 = ustate + (size_t)u1 * A2D_USTATE;
		int *w0 = ustage + (size_t)u0 * A2D_USTATE;
		int *w1 = ustage + (size_t)u1 * A2D_USTATE;
		if(w0 != wi0) {
			w0[OW_MODE] = wi0[OW_MODE]; w0[OW_WAVE] = wi0[OW_WAVE]; w0[OW_DPHASE] = wi0[OW_DPHASE];
			w0[OW_PRAMPING] = wi0[OW_PRAMPING];
#pragma unroll
			for(int k = 0; k < 4; ++k) {
				w0[OW_P + k] = wi0[OW_P + k];
				w0[OW_A + k] = wi0[OW_A + k];
				w1[PW_VOL + k] = wi1[PW_VOL + k];
				w1[PW_PAN + k] = wi1[PW_PAN + k];
			}
		}
		w0[OW_PHASE_LO] = phlo_l;
		w0[OW_PHASE_HI] = phhi_l;
	}
}

// staged state of the fast leaf voices (nosc oscillators + panmix) -> the unit
// state array
DEV void commit_block(const A2DCommit &cm, int block, const A2DVoice *__restrict__ voices,
		const A2DRun *__restrict__ runs, int *__restrict__ ustate)
{
	const int per = (cm.nosc + 1) * 16;
	int i = block * 256 + threadIdx.x;
	if(i >= cm.nlist * per)
		return;
	const int slot = cm.list[i / per];
	if(runs[slot].count)
		return;		// not rendered by the fast kernel this batch: nothing staged
	const A2DVoice &vc = voices[slot];
	const int u = (i % per) >> 4, k = i & 15;
	// the words the kernels stage: wtosc 0..14 except the noise sample, panmix 0..7
	if(u < cm.nosc ? (k != OW_NOISE && k < 15) : (k < 8)) {
		size_t a = (size_t)vc.unit[u] * A2D_USTATE + k;
		ustate[a] = cm.ustage[a];
	}
}

DEV int commit_blocks(const A2DCommit &cm) { return (cm.nlist * (cm.nosc + 1) * 16 + 255) / 256; }

__global__ __launch_bounds__(256)
void k_commit_oscpan(A2DCommit cm, const A2DVoice *__restrict__ voices, const A2DRun *__restrict__ runs,
		int *__restrict__ ustate)
{
	commit_block(cm, blockIdx.x, voices, runs, ustate);
}

// ---------------------------------------------------------------------------
// wtosc + wtosc -> panmix 1->2: two oscillators summed in the scratch buffer
// (the second one adding, compiler.c:3125), the BASELINE config 4 leaf
// ---------------------------------------------------------------------------
// Same plan as k_leaf_oscpan with the oscillator state doubled; 4 fragments per
// register chunk keep the 16 window loads of a chunk within the register budget.
#define OSC2_FCH A2D_OSC2_FCH
enum { OV_MODE = 0, OV_WAVE, OV_DPHASE, OV_PHLO, OV_PHHI, OV_PRAMP, OV_P = 6, OV_A = 10, OV_NWORDS = 14 };
enum { OD_MM = 0, OD_DPH, OD_SIZEM, OD_DOFF, OD_NWORDS };

DEV void osc_from_lanes(OscS &o, const int (&so)[OV_NWORDS], int v)
{
	o.mode = rdl(so[OV_MODE], v);
	o.wave = rdl(so[OV_WAVE], v);
	o.dphase = (unsigned)rdl(so[OV_DPHASE], v);
	o.phase = (uint64_t)(unsigned)rdl(so[OV_PHLO], v) | ((uint64_t)(unsigned)rdl(so[OV_PHHI], v) << 32);
	o.p_ramping = rdl(so[OV_PRAMP], v);
	o.p.value = rdl(so[OV_P], v); o.p.target = rdl(so[OV_P + 1], v);
	o.p.delta = rdl(so[OV_P + 2], v); o.p.timer = rdl(so[OV_P + 3], v);
	o.a.value = rdl(so[OV_A], v); o.a.target = rdl(so[OV_A + 1], v);
	o.a.delta = rdl(so[OV_A + 2], v); o.a.timer = rdl(so[OV_A + 3], v);
	o.noise = 0;
	o.seed = 0;
}

DEV void osc_to_lanes(int (&so)[OV_NWORDS], const OscS &o, bool me)
{
	WRL(so[OV_MODE], o.mode);
	WRL(so[OV_WAVE], o.wave);
	WRL(so[OV_DPHASE], (int)o.dphase);
	WRL(so[OV_PHLO], (int)(unsigned)o.phase);
	WRL(so[OV_PHHI], (int)(unsigned)(o.phase >> 32));
	WRL(so[OV_PRAMP], o.p_ramping);
	WRL(so[OV_P], o.p.value); WRL(so[OV_P + 1], o.p.target);
	WRL(so[OV_P + 2], o.p.delta); WRL(so[OV_P + 3], o.p.timer);
	WRL(so[OV_A], o.a.value); WRL(so[OV_A + 1], o.a.target);
	WRL(so[OV_A + 2], o.a.delta); WRL(so[OV_A + 3], o.a.timer);
}

#ifndef OSC2_WPE
#define OSC2_WPE 4	// (round 2: 3 wavefronts per SIMD with 143 registers and no spills beat 4 with 128 and 39 spills by a
			// fifth.  Round 3: with only the settled loop's values parked in registers the kernel needs 118 -
			// 4 wavefronts without spills: 2.30 -> 2.24 ms at configs[3]; 3 wavefronts of the same code: 2.40)
#endif
__global__ __launch_bounds__(64 * FAST_WPB) __attribute__((amdgpu_waves_per_eu(OSC2_WPE, OSC2_WPE)))
void k_leaf_osc2pan(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int vpw,
		int ysplit, const A2DVoice *__restrict__ voices, const int *ustate, int *ustage,
		const int16_t *__restrict__ wavepool, const A2DWave *__restrict__ waves,
		const uint32_t *__restrict__ ptab, int *__restrict__ busmem, const int *__restrict__ wavecoef)
{
	const A2DParams &p = *pp;
	const int wv = rfl((int)(threadIdx.x >> 6));	// (wave-uniform, and known to the compiler as such)
	const int lane = threadIdx.x & 63;
	const int first = (blockIdx.x * FAST_WPB + wv) * vpw;
	if(first >= nlist)
		return;
	const int nv = min(vpw, nlist - first);
	const int nfrags = p.nfrags;
	const int dbg = p.debug;
	FastPtrs g = { wavepool, waves, ptab, dbg };
	const CoefRsrc crs = coef_rsrc(wavecoef);

	const int nchunks = (nfrags + OSC2_FCH - 1) / OSC2_FCH;
	const int per = (nchunks + ysplit - 1) / ysplit;
	const int slice = blockIdx.y;
	const int c_lo = slice * per, c_hi = min(nchunks, c_lo + per);
	const bool last_slice = c_hi >= nchunks;
	if(c_lo >= nchunks)
		return;
	// (the slices that exist; voices that still have something moving are dealt over
	// them - slice s walks the whole batch for every nslices-th of them)
	const int nslices = (nchunks + per - 1) / per;

	int ffr[A2D_MAXBATCH / 64], fst[A2D_MAXBATCH / 64];
#pragma unroll
	for(int k = 0; k < A2D_MAXBATCH / 64; ++k) {
		ffr[k] = (k * 64 + lane < nfrags) ? p.fragframes[k * 64 + lane] : 0;
		fst[k] = (k * 64 + lane < nfrags) ? p.fragstart[k * 64 + lane] : 0;
	}

	// What stays in registers over the launch, one voice per lane: what the settled loop needs (level,
	// increment, table, amplitude and phase per oscillator, the two pan gains, the bus).  Everything
	// else of a voice's state is read when it is decided whether the voice is settled and again, from
	// memory, by the loop for voices that are not - rare, and 30 registers a wavefront.
	int od[2][OD_NWORDS], amp_l[2] = { 0, 0 }, phlo_l[2] = { 0, 0 }, phhi_l[2] = { 0, 0 };
	int v0l = 0, v1l = 0, settled_l = 0;
	int uu[3] = { 0, 0, 0 }, my_off = -1, my_nch = 2;
	bool mine = false;
#pragma unroll
	for(int o = 0; o < 2; ++o)
#pragma unroll
		for(int k = 0; k < OD_NWORDS; ++k)
			od[o][k] = 0;
	if(lane < nv)
		mine = p.runs[list[first + lane]].count == 0;
	if(mine) {
		const A2DVoice &vc = voices[list[first + lane]];
		my_off = vc.out_off;
		my_nch = vc.out_nch;
		bool settled = true;
		int wave_l[2], dphase_l[2];
#pragma unroll
		for(int o = 0; o < 2; ++o) {
			uu[o] = vc.unit[o];
			const int *w = ustate + (size_t)uu[o] * A2D_USTATE;
			wave_l[o] = w[OW_WAVE];
			dphase_l[o] = w[OW_DPHASE];
			amp_l[o] = w[OW_A];
			phlo_l[o] = w[OW_PHASE_LO];
			phhi_l[o] = w[OW_PHASE_HI];
			settled = settled && w[OW_MODE] == A2D_OSC_MIPWAVE && dphase_l[o] && !w[OW_PRAMPING] &&
					!(w[OW_P + 3] | w[OW_P + 2] | w[OW_A + 3] | w[OW_A + 2]) &&
					w[OW_P] == w[OW_P + 1] && w[OW_A] == w[OW_A + 1];
		}
		uu[2] = vc.unit[2];
		const int *wp = ustate + (size_t)uu[2] * A2D_USTATE;
		int sp[8];
#pragma unroll
		for(int k = 0; k < 8; ++k)
			sp[k] = wp[k];		// vol ramper, pan ramper
		settled = settled && !(sp[3] | sp[2] | sp[7] | sp[6]) && sp[0] == sp[1] && sp[4] == sp[5];
		if(settled) {
#pragma unroll
			for(int o = 0; o < 2; ++o) {
				const A2DWave *w = waves + wave_l[o];
				const unsigned period = w->period, dphase = (unsigned)dphase_l[o];
				unsigned dph = ((dphase + 255) >> 8) * period, mm = 0;	// wtosc.c:250-258
				for(; (dph > (A2D_MAXPHINC << 8)) && (mm < A2D_MIPS - 1); ++mm)
					dph >>= 1;
				dph = (unsigned)(((uint64_t)dphase * period) >> mm);
				settled = settled && w->size[0] && (w->flags & 0x100u) && dph <= (A2D_MAXPHINC << 16);
				od[o][OD_MM] = (int)mm;
				od[o][OD_DPH] = (int)dph;
				od[o][OD_SIZEM] = (int)w->size[mm];
				od[o][OD_DOFF] = (int)w->off[mm];
			}
			const int vol = sp[0], pan = sp[4];	// panmix_process12 gains (panmix.c:89-104)
			const int vp = mul64s(pan, vol, 24);
			v0l = wsub(vol, vp);
			v1l = wadd(vol, vp);
			if(pan > 0xffffff || pan < -0xffffff) {
				int lim = wshl(vol, 1);
				if(v0l > lim) v0l = lim;
				if(v1l > lim) v1l = lim;
			}
		}
		settled_l = settled ? 1 : 0;
	}
	const unsigned long long unsettled_mask = __ballot(mine && !settled_l);

	// ---- settled voices: this slice's chunks ----
	for(int c = c_lo; c < c_hi; ++c) {
		const int f0 = c * OSC2_FCH;
		const int nf = min((int)OSC2_FCH, nfrags - f0);
		int acc0[OSC2_FCH], acc1[OSC2_FCH], nfr[OSC2_FCH];
#pragma unroll
		for(int j = 0; j < OSC2_FCH; ++j) {
			acc0[j] = acc1[j] = 0;
			nfr[j] = (j < nf) ? frames_of(ffr, f0 + j) : 0;
		}
		const unsigned before = (unsigned)frames_of(fst, f0);
		unsigned pre[OSC2_FCH];		// frames from the start of the chunk to each fragment
		pre[0] = 0;
#pragma unroll
		for(int j = 1; j < OSC2_FCH; ++j)
			pre[j] = pre[j - 1] + (unsigned)nfr[j - 1];
		int cur_off = rdl(my_off, 0), cur_nch = rdl(my_nch, 0);
		for(int v = 0; v < nv; ++v) {
			if(!rdl(settled_l, v))
				continue;
			const int voff = rdl(my_off, v);
			if(voff != cur_off) {
				flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
				cur_off = voff;
				cur_nch = rdl(my_nch, v);
			}
			const int v0 = rdl(v0l, v), v1 = rdl(v1l, v);
			// one oscillator after the other: the eight coefficient entries of a
			// chunk (4 fragments x 2 taps, 24 registers) in flight at a time
#ifndef OSC2_SEQ
			// the coefficient entries of BOTH oscillators (4 fragments x 2 taps x 2: 48 registers) are in
			// flight before either is used (one oscillator after the other: -DOSC2_SEQ, 4 % slower)
			int xs[OSC2_FCH];
			uint64_t endph[2];
			Coef4 ka[2][OSC2_FCH], kb[2][OSC2_FCH];
			unsigned pa[2][OSC2_FCH], pb[2][OSC2_FCH];
			int amps[2];
#pragma unroll
			for(int o = 0; o < 2; ++o) {
				const unsigned mm = (unsigned)rdl(od[o][OD_MM], v), dph = (unsigned)rdl(od[o][OD_DPH], v);
				const unsigned sizem = (unsigned)rdl(od[o][OD_SIZEM], v), doff = (unsigned)rdl(od[o][OD_DOFF], v);
				amps[o] = rdl(amp_l[o], v);
				const uint64_t phase = (uint64_t)(unsigned)rdl(phlo_l[o], v) |
						((uint64_t)(unsigned)rdl(phhi_l[o], v) << 32);
				uint64_t ph = (phase >> mm) + (uint64_t)before * dph;
				const unsigned ldph = lane_dph(lane, dph);
				const int cb = coef_base(doff);
				uint64_t phs[OSC2_FCH];
				if(!(sizem & (sizem - 1)) && !(ph >> 48)) {
					const uint64_t mask = ((uint64_t)sizem << 24) - 1;
#pragma unroll
					for(int j = 0; j < OSC2_FCH; ++j)
						phs[j] = (ph + (uint64_t)dph * pre[j]) & mask;
					ph = phs[OSC2_FCH - 1] + (uint64_t)dph * (unsigned)nfr[OSC2_FCH - 1];
				} else {
#pragma unroll
					for(int j = 0; j < OSC2_FCH; ++j) {
						ph = wrap_phase(ph, sizem);
						phs[j] = ph;
						ph += (uint64_t)dph * (unsigned)nfr[j];
					}
				}
#pragma unroll
				for(int j = 0; j < OSC2_FCH; ++j) {
					pa[o][j] = tap_phase(phs[j], ldph);
					asm("" : "+v"(pa[o][j]));
					pb[o][j] = pa[o][j] + (dph >> 17);
					ka[o][j] = coef_at(crs, cb, pa[o][j]);
					kb[o][j] = coef_at(crs, cb, pb[o][j]);
				}
				endph[o] = ph << mm;
			}
#pragma unroll
			for(int o = 0; o < 2; ++o)
#pragma unroll
				for(int j = 0; j < OSC2_FCH; ++j) {
					const int y = mul64s(hermite_c(ka[o][j], pa[o][j]) + hermite_c(kb[o][j], pb[o][j]), amps[o], 17);
					xs[j] = o ? wadd(xs[j], y) : y;
				}
#else
			int xs[OSC2_FCH];
			uint64_t endph[2];
#pragma unroll
			for(int o = 0; o < 2; ++o) {
				const unsigned mm = (unsigned)rdl(od[o][OD_MM], v), dph = (unsigned)rdl(od[o][OD_DPH], v);
				const unsigned sizem = (unsigned)rdl(od[o][OD_SIZEM], v), doff = (unsigned)rdl(od[o][OD_DOFF], v);
				const int amp = rdl(amp_l[o], v);
				const uint64_t phase = (uint64_t)(unsigned)rdl(phlo_l[o], v) |
						((uint64_t)(unsigned)rdl(phhi_l[o], v) << 32);
				uint64_t ph = (phase >> mm) + (uint64_t)before * dph;
				const unsigned ldph = lane_dph(lane, dph);
				const int cb = coef_base(doff);
				uint64_t phs[OSC2_FCH];
				if(!(sizem & (sizem - 1)) && !(ph >> 48)) {
					// power-of-two size: the modulus is a mask (as in k_leaf_oscpan)
					const uint64_t mask = ((uint64_t)sizem << 24) - 1;
#pragma unroll
					for(int j = 0; j < OSC2_FCH; ++j)
						phs[j] = (ph + (uint64_t)dph * pre[j]) & mask;
					ph = phs[OSC2_FCH - 1] + (uint64_t)dph * (unsigned)nfr[OSC2_FCH - 1];
				} else {
#pragma unroll
					for(int j = 0; j < OSC2_FCH; ++j) {
						ph = wrap_phase(ph, sizem);
						phs[j] = ph;
						ph += (uint64_t)dph * (unsigned)nfr[j];
					}
				}
				Coef4 ka[OSC2_FCH], kb[OSC2_FCH];
				unsigned pa[OSC2_FCH], pb[OSC2_FCH];
#pragma unroll
				for(int j = 0; j < OSC2_FCH; ++j) {
					pa[j] = tap_phase(phs[j], ldph);
					asm("" : "+v"(pa[j]));	// (32 bit offsets: scalar-base loads, as in k_leaf_oscpan)
					pb[j] = pa[j] + (dph >> 17);
					ka[j] = coef_at(crs, cb, pa[j]);
					kb[j] = coef_at(crs, cb, pb[j]);
				}
#pragma unroll
				for(int j = 0; j < OSC2_FCH; ++j) {
					const int y = mul64s(hermite_c(ka[j], pa[j]) + hermite_c(kb[j], pb[j]), amp, 17);
					// the second oscillator adds into the scratch buffer (wrap-around)
					xs[j] = o ? wadd(xs[j], y) : y;
				}
				endph[o] = ph << mm;
			}
#endif
#pragma unroll
			for(int j = 0; j < OSC2_FCH; ++j) {
				const int x = (lane < nfr[j]) ? xs[j] : 0;
				acc0[j] = wadd(acc0[j], mul64s(x, v0, 24));
				acc1[j] = wadd(acc1[j], mul64s(x, v1, 24));
			}
			if(last_slice && c == c_hi - 1) {
				const bool me = lane == v;
#pragma unroll
				for(int o = 0; o < 2; ++o) {
					WRL(phlo_l[o], (int)(unsigned)endph[o]);
					WRL(phhi_l[o], (int)(unsigned)(endph[o] >> 32));
				}
			}
		}
		flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
	}

	// ---- voices with something still moving: one slice each walks the whole batch ----
	const unsigned long long todo_mask = unsettled_mask & __ballot(lane % nslices == slice);
	if(todo_mask) {
		for(int f0 = 0; f0 < nfrags; f0 += OSC2_FCH) {
			const int nf = min((int)OSC2_FCH, nfrags - f0);
			int acc0[OSC2_FCH], acc1[OSC2_FCH];
#pragma unroll
			for(int j = 0; j < OSC2_FCH; ++j)
				acc0[j] = acc1[j] = 0;
			int cur_off = rdl(my_off, 0), cur_nch = rdl(my_nch, 0);
			for(int v = 0; v < nv; ++v) {
				if(!((todo_mask >> v) & 1ull))
					continue;
				const int voff = rdl(my_off, v);
				if(voff != cur_off) {
					flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
					cur_off = voff;
					cur_nch = rdl(my_nch, v);
				}
				// (its state from memory and back: what this slice left after the chunk before is in
				// the staging area - or, one slice, in place - from the second chunk on)
				OscS oa, ob;
				Ramp vol, pan;
				const int *src = f0 ? ustage : ustate;
				osc_from_mem(oa, src + (size_t)rdl(uu[0], v) * A2D_USTATE);
				osc_from_mem(ob, src + (size_t)rdl(uu[1], v) * A2D_USTATE);
				{
					const volatile int *wpv = src + (size_t)rdl(uu[2], v) * A2D_USTATE;
					vol.value = rfl(wpv[0]); vol.target = rfl(wpv[1]); vol.delta = rfl(wpv[2]); vol.timer = rfl(wpv[3]);
					pan.value = rfl(wpv[4]); pan.target = rfl(wpv[5]); pan.delta = rfl(wpv[6]); pan.timer = rfl(wpv[7]);
				}
				for(int j = 0; j < nf; ++j) {
					const int n = frames_of(ffr, f0 + j);
					int o0 = 0, o1 = 0;
					const OscPend pa_ = osc_fragment_begin(g, oa, n, lane), pb_ = osc_fragment_begin(g, ob, n, lane);
					const int x = wadd(osc_fragment_end(pa_), osc_fragment_end(pb_));
					pan_fragment_s(vol, pan, x, n, lane, o0, o1);
#pragma unroll
					for(int jj = 0; jj < OSC2_FCH; ++jj)
						if(jj == j) {
							acc0[jj] = wadd(acc0[jj], o0);
							acc1[jj] = wadd(acc1[jj], o1);
						}
				}
				if(lane == 0) {
					osc_to_mem(ustage + (size_t)rdl(uu[0], v) * A2D_USTATE, oa);
					osc_to_mem(ustage + (size_t)rdl(uu[1], v) * A2D_USTATE, ob);
					int *wpv = ustage + (size_t)rdl(uu[2], v) * A2D_USTATE;
					wpv[0] = vol.value; wpv[1] = vol.target; wpv[2] = vol.delta; wpv[3] = vol.timer;
					wpv[4] = pan.value; wpv[5] = pan.target; wpv[6] = pan.delta; wpv[7] = pan.timer;
				}
			}
			flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");	// (the next chunk reads what lane 0 stored)
			__builtin_amdgcn_s_waitcnt(0);
		}
	}

	// A settled voice's state after the batch: as it was, with the oscillators' phases moved on (the
	// last slice writes it; voices that were not settled were stored by the loop above).
	if(mine && settled_l && last_slice) {
#pragma unroll
		for(int o = 0; o < 2; ++o) {
			const int *wi = ustate + (size_t)uu[o] * A2D_USTATE;
			int *w = ustage + (size_t)uu[o] * A2D_USTATE;
			if(w != wi) {
				w[OW_MODE] = wi[OW_MODE]; w[OW_WAVE] = wi[OW_WAVE]; w[OW_DPHASE] = wi[OW_DPHASE];
				w[OW_PRAMPING] = wi[OW_PRAMPING];
#pragma unroll
				for(int k = 0; k < 4; ++k) {
					w[OW_P + k] = wi[OW_P + k];
					w[OW_A + k] = wi[OW_A + k];
				}
			}
			w[OW_PHASE_LO] = phlo_l[o];
			w[OW_PHASE_HI] = phhi_l[o];
		}
		const int *wpi = ustate + (size_t)uu[2] * A2D_USTATE;
		int *wp = ustage + (size_t)uu[2] * A2D_USTATE;
		if(wp != wpi)
#pragma unroll
			for(int k = 0; k < 8; ++k)
				wp[k] = wpi[k];
	}
}

// ---------------------------------------------------------------------------
// The same voices in a batch in which they carry command records
// ---------------------------------------------------------------------------
// A voice whose VM woke up during the batch - control writes between windows,
// windows that start or end inside a fragment (a2_VoiceProcess, core.c:1852-1878),
// its birth (Initialize + the first writes) and its death - is skipped by the two
// kernels above (runs[slot].count != 0).  In a scripted scene that is every voice
// in every batch.  k_leaf_recs<NOSC> renders those voices of the wtosc[+wtosc]->panmix
// classes: the per-fragment path of the kernels above (scalar control, lane = frame)
// with the record stream of k_voices - R_SEG windows, R_WRITE (wtosc.c:433-504,
// panmix.c:219-249 through a2_SetRamper, a2_dsp.h:161-170), R_INIT (wtosc.c:390-423,
// panmix.c:252-284), R_KILL - executed on the scalar unit between the windows.  State
// stays in registers (one voice per lane, read out with v_readlane) over the whole
// batch; the sums of RECS_FCH fragments of all the wavefront's voices on one bus go
// out in one atomic add per fragment and channel.
#define RECS_FCH 4
#ifndef RECS_WPB
#define RECS_WPB 8		// wavefronts per workgroup
#endif

DEV void osc_init_s(const FastPtrs &g, OscS &o, int pitch)
{
	// wtosc_Initialize, wtosc.c:390-423 (value = transpose + basepitch)
	o.wave = -1;
	o.mode = A2D_OSC_OFF;
	o.phase = 0;
	o.p_ramping = 0;
	o.noise = 0;
	o.seed = 0;
	ramp_init(o.a, 0);
	ramp_init(o.p, pitch);
	o.dphase = (unsigned)rfl((int)p2i(g.ptab, o.p.value >> 8));
}

DEV void osc_write_s(const FastPtrs &g, OscS &o, int reg, int v, int start, int dur)
{
	switch(reg) {
	  case 0: {	// wtosc_Wave, wtosc.c:433-483 (the host resolved the handle; mip-mapped
			// waves, the noise generator and "off" reach this kernel)
		int wt = 0;
		o.wave = v;
		if(v >= 0) {
			const A2DWave *w = g.waves + v;
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
			const unsigned period = g.waves[o.wave].period;
			const int ph = (int)((unsigned)v + ((((unsigned)start) * (o.dphase >> 8)) >> 8));
			o.phase = (uint64_t)(((int64_t)ph * (int64_t)period) * 256);
		}
		break;
	}
}

// filter12 state of one voice, wave-uniform (A2_filter12, filter12.c:36-56; one channel)
enum { FS_Q = 0, FS_LP = 4, FS_BP, FS_HP, FS_F1, FS_D1, FS_D2, FS_F1NEXT, FS_RAMP, FS_NWORDS };
struct FiltS { Ramp q; int lp, bp, hp, f1, d1, d2, f1next, ramp; };

// (v_writelane_b32: one lane of a vector register from a scalar one.  Through the compiler's own
// intrinsic - this clang has no __builtin for it, the name binds to llvm.amdgcn.writelane - the
// lane select is routed through M0 by the compiler (two scalar registers in one VOP3 are one too
// many for the constant bus), where round 2's inline asm named m0 as a clobber, which the
// compiler does not promise to honour)
extern "C" __device__ int a2d_writelane(int src, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
DEV int writelane_s(int y, int val, int sel)
{
	return a2d_writelane(val, sel, y);
}

// One window of f12_process (filter12.c:74-119) over the frames a wavefront holds one per
// lane: the recurrence runs on the SCALAR unit - per frame a v_readlane, 17 to 26 scalar
// operations (nine of them the dependent chain d1 -> d1) and a v_writelane.  Measured on a
// wavefront that has its SIMD to itself (a song: -DRECS_PROF): ~7 cycles per instruction.
DEV int filt_window_s(FiltS &fs, int x, int off, int len, int lane)
{
	int f0 = fs.f1, df = 0, f1 = fs.f1;
	ramp_prepare_s(fs.q, len);
	if(fs.ramp) {		// the host ran the cutoff ramper and f12_pitch2coeff (R_F1RAMP)
		f1 = fs.f1next;
		df = rfl(wadd(wsub(f1, f0), len >> 1) / len);
	}
	int qv = fs.q.value, d1 = fs.d1, d2 = fs.d2;
	const int qd = fs.q.delta, lp = fs.lp, bp = fs.bp, hp = fs.hp;
	// (input and output in registers of their own: reading frame s + 1 does not wait
	// for frame s to be written)
	int y = x;
	// Three things most windows do not need, each a twelfth to a seventh of the frame's
	// instructions (the variant is chosen per window, uniformly):
	//   bit 0  bp = hp = 0, the plain low-pass most voices are: their products drop out of the sum
	//   bit 1  q is not ramping
	//   bit 2  the cutoff is not ramping
	auto frame = [&](int s, auto variant) {
		constexpr int V = decltype(variant)::value;
		const int xin = rdl(x, off + s);
		const int f = f0 >> 12, qq = qv >> 12;
		const int d1s = d1 >> 4;
		const int l = wadd(d2, wmul(f, d1s) >> 8);
		const int h = wsub(wsub(xin >> 5, l), wmul(qq, d1s) >> 8);
		const int b = wadd(wmul(f, h >> 4) >> 8, d1);
		const int out = ((V & 1) ? wmul(l, lp) : wadd(wadd(wmul(l, lp), wmul(b, bp)), wmul(h, hp))) >> 3;
		d1 = b;
		d2 = l;
		if(!(V & 4))
			f0 = wadd(f0, df);
		if(!(V & 2))
			qv = wadd(qv, qd);
		y = writelane_s(y, out, off + s);
	};
	// (four frames per trip: the output sums of one frame fill the waits of the next
	// one's recurrence, and a taken branch costs a wavefront on its own more than an
	// instruction)
	auto frames = [&](auto variant) {
		int s = 0;
		for(; s + 4 <= len; s += 4) {
			frame(s, variant);
			frame(s + 1, variant);
			frame(s + 2, variant);
			frame(s + 3, variant);
		}
		for(; s < len; ++s)
			frame(s, variant);
	};
	switch(((bp | hp) == 0 ? 1 : 0) | (qd == 0 ? 2 : 0) | (df == 0 ? 4 : 0)) {
	  case 0: frames(std::integral_constant<int, 0>()); break;
	  case 1: frames(std::integral_constant<int, 1>()); break;
	  case 2: frames(std::integral_constant<int, 2>()); break;
	  case 3: frames(std::integral_constant<int, 3>()); break;
	  case 4: frames(std::integral_constant<int, 4>()); break;
	  case 5: frames(std::integral_constant<int, 5>()); break;
	  case 6: frames(std::integral_constant<int, 6>()); break;
	  default: frames(std::integral_constant<int, 7>()); break;
	}
	fs.d1 = d1;
	fs.d2 = d2;
	fs.q.value = qv;	// (= a2_RunRamper(&q, 1) per frame)
	fs.f1 = f1;
	fs.ramp = 0;
	return y;
}

// Round 4, last: the same window as a relaxation ALONG THE LANES (RECS_JFILT).  Frame s of the window
// sits in lane s with its own input, cutoff and q; what it needs of frame s - 1 - d1 and d2, i.e. that
// frame's b and l - it reads from lane s - 1 through the DPP wave shift, on the vector unit, every
// lane at once.  One such step makes one more lane right: lane 0 takes the voice's state (the shift
// has no source for it and leaves it alone), after step k lanes 0..k hold what f12_process
// (filter12.c:97-118) computes for frames 0..k, and a lane that is right stays right, because its
// left neighbour no longer changes.  len - 1 steps of 13 vector instructions (no v_readlane, no
// v_writelane, no M0, the cutoff and q ramps in closed form per lane, the output sums once per
// window with lane = frame) against 19 - 28 scalar ones per frame above; what the other lanes
// compute in the meantime is thrown away, which costs a lone wavefront nothing.  A window that does
// not begin at frame 0 is rotated to lane 0 and back (ds_bpermute).
#ifndef RECS_JFILT
#define RECS_JFILT 1
#endif
#define F12_JSTEP \
	"s_nop 1\n\t" \
	"v_mov_b32_dpp %[bsh], %[B] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
	"v_ashrrev_i32 %[ds], 4, %[bsh]\n\t" \
	"v_mul_lo_u32 %[t1], %[F], %[ds]\n\t" \
	"v_ashrrev_i32 %[t1], 8, %[t1]\n\t" \
	"v_mul_lo_u32 %[t2], %[Q], %[ds]\n\t" \
	"v_ashrrev_i32 %[t2], 8, %[t2]\n\t" \
	"v_add_u32_dpp %[L], %[L], %[t1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
	"v_sub_u32 %[H], %[X], %[L]\n\t" \
	"v_sub_u32 %[H], %[H], %[t2]\n\t" \
	"v_ashrrev_i32 %[t1], 4, %[H]\n\t" \
	"v_mul_lo_u32 %[t1], %[F], %[t1]\n\t" \
	"v_ashrrev_i32 %[t1], 8, %[t1]\n\t" \
	"v_add_u32 %[B], %[t1], %[bsh]\n\t"
DEV int filt_window_j(FiltS &fs, int x, int off, int len, int lane)
{
	const int f0 = fs.f1;
	int df = 0, f1 = fs.f1;
	ramp_prepare_s(fs.q, len);
	if(fs.ramp) {		// the host ran the cutoff ramper and f12_pitch2coeff (R_F1RAMP)
		f1 = fs.f1next;
		df = rfl(wadd(wsub(f1, f0), len >> 1) / len);
	}
	const int qv = fs.q.value, qd = fs.q.delta;
	const int d1i = fs.d1, d2i = fs.d2;
	// lane = frame of the WINDOW
	int xw = x;
	if(off)
		xw = __builtin_amdgcn_ds_bpermute(((lane + off) & 63) << 2, x);
	const int X = xw >> 5;
	const int F = wadd(f0, wmul(df, lane)) >> 12;
	const int Q = wadd(qv, wmul(qd, lane)) >> 12;
	// step 0: every lane from the voice's state - right for lane 0
	const int ds0 = d1i >> 4;
	int L = wadd(d2i, wmul(F, ds0) >> 8);
	int H = wsub(wsub(X, L), wmul(Q, ds0) >> 8);
	int B = wadd(wmul(F, H >> 4) >> 8, d1i);
	int bsh = d1i;		// (lane 0 keeps this: d1 of the window's first frame)
	int ds, t1, t2;
	int n = len - 1;
	for(; n > 4; n -= 8)
		asm volatile(F12_JSTEP F12_JSTEP F12_JSTEP F12_JSTEP F12_JSTEP F12_JSTEP F12_JSTEP F12_JSTEP
				: [bsh] "+v"(bsh), [L] "+v"(L), [B] "+v"(B), [H] "+v"(H), [ds] "=&v"(ds), [t1] "=&v"(t1), [t2] "=&v"(t2)
				: [F] "v"(F), [Q] "v"(Q), [X] "v"(X));
	if(n > 0)	// (steps past len - 1 change nothing)
		asm volatile(F12_JSTEP F12_JSTEP F12_JSTEP F12_JSTEP
				: [bsh] "+v"(bsh), [L] "+v"(L), [B] "+v"(B), [H] "+v"(H), [ds] "=&v"(ds), [t1] "=&v"(t1), [t2] "=&v"(t2)
				: [F] "v"(F), [Q] "v"(Q), [X] "v"(X));
	int yw = wmul(L, fs.lp);
	if(fs.bp | fs.hp)
		yw = wadd(wadd(yw, wmul(B, fs.bp)), wmul(H, fs.hp));
	yw >>= 3;
	fs.d1 = rdl(B, len - 1);
	fs.d2 = rdl(L, len - 1);
	fs.q.value = wadd(qv, wmul(qd, len));	// (= a2_RunRamper(&q, 1) per frame)
	fs.f1 = f1;
	fs.ramp = 0;
	if(off)
		yw = __builtin_amdgcn_ds_bpermute(((lane - off) & 63) << 2, yw);
	return (unsigned)(lane - off) < (unsigned)len ? yw : x;
}

// Round 4: the filter of the records kernels with lane = VOICE.  filt_window_s above runs filter12's
// recurrence on the scalar unit, a window at a time, in the middle of the voice's walk: ~20 scalar
// instructions per frame and voice at the 5-7 cycles a lone scalar stream issues at - two thirds of
// what a scripted filter voice costs.  With RECS_VFILT the walk (lane = frame, scalar control) only
// renders the oscillators of a window into the voice's row of an LDS tile and works out what the
// window needs of the filter and of the pan stage - f0 and its step (the host's / the device VM's
// coefficient records), the q ramp, the mix levels; the volume and pan rampers' values and steps -
// in closed form (none of it depends on the audio), as one 12-word entry of a small pool.  When the
// chunk is walked (or the pool is full) the wavefront turns round: every lane takes ONE VOICE and
// runs the recurrence over its pending windows in place - 12 vector instructions per frame for up
// to 64 voices at once - then, lane = frame again, the pan stage reads the rows back window by
// window.  Filter state (d1, d2) never leaves the lane it is parked in.
#ifndef RECS_VFILT
#define RECS_VFILT 1
#endif
#define RECS_ENTRY 12		// words per pending window: [j | off << 4 | len << 12 | clamp << 20, f0, df, qv, qd, lp, bp, hp, vol, dvol, pan, dpan]
#define RECS_VF_FCH 2		// fragments per chunk on that path (rows of the chunk live in LDS)
DEV int recs_pool_cap(int vpw) { return vpw * RECS_VF_FCH * 2 + 16; }
DEV int recs_wave_words(int vpw) { return RECS_VF_FCH * vpw * 65 + recs_pool_cap(vpw) * RECS_ENTRY; }
static int recs_wave_words_host(int vpw) { return RECS_VF_FCH * vpw * 65 + (vpw * RECS_VF_FCH * 2 + 16) * RECS_ENTRY; }
extern __shared__ int recs_dyn[];

// (the body of the kernels below: gw = this wavefront's index among those of its class)
typedef int RecsPart[RECS_WPB][RECS_FCH * 2][64];
// A voice's next command record, fetched through the scalar cache and ahead of its use: the load is
// issued when the record before it is taken up (a window's worth of work earlier) and does not queue
// behind the wavefront's bus atomics as a vector load does (gfx9 counts both in vmcnt).  A lone
// wavefront - a song's voice - used to sit out a memory round trip per fragment just to learn that the
// fragment holds no record.
typedef int RecQ __attribute__((ext_vector_type(4)));	// head, value, dur, start
// (through the CONSTANT address space: a uniform address there is a scalar load the compiler issues
// where it stands and waits for where the value is used - records are written by the host only)
DEV RecQ rec_issue(const A2DRec *recs, int idx)
{
	typedef const __attribute__((address_space(4))) RecQ *CRecQ;
	return *(CRecQ)(uintptr_t)(recs + idx);
}
DEV RecQ rec_ready(RecQ r) { return r; }

template<int NOSC, int FILT, int VFT = 0>
DEV void recs_body(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int vpw, int gw,
		const A2DVoice *__restrict__ voices, int *ustate, int *vactive,
		const int16_t *__restrict__ wavepool, const A2DWave *__restrict__ waves,
		const uint32_t *__restrict__ ptab, int *__restrict__ busmem,
		RecsPart *part, int (*part_off)[RECS_WPB], int (*part_nch)[RECS_WPB], int skip_empty = 0)
{
#ifdef RECS_PROF
	const long long t_in = __builtin_readcyclecounter();
	const unsigned long long rt_in = __builtin_amdgcn_s_memrealtime();
	long long t_win = 0, t_rec = 0, t_osc = 0, t_flt = 0;
	int n_win = 0, n_rec = 0;
#endif
	const A2DParams &p = *pp;
	const int wv = rfl((int)(threadIdx.x >> 6));	// (wave-uniform, and known to the compiler as such)
	const int lane = threadIdx.x & 63;
	// (skip_empty bit 1: the launcher gave this launch the lane = voice filter, RECS_VFILT - worth it from a
	// few voices per wavefront up; a song's one-voice wavefronts keep the scalar recurrence)
	// (VFT, a kernel of its own: with both filters in one kernel the window filter's registers took
	// k_leaf_recs<1, 1> from 127 to 132 vector registers - four wavefronts per SIMD to three - and a launch
	// uses only one of the two)
	constexpr bool VF = FILT && RECS_VFILT && VFT;
	skip_empty &= 1;
	const int first = gw * vpw;
	// (a wavefront past the end of the list still meets the others at the barriers)
	const int nv = max(0, min(vpw, nlist - first));
	// what the workgroup's wavefronts are left with at the end of a chunk: summed
	// here before it goes to the bus (two buffers: one barrier per chunk)
	const int nfrags = p.nfrags;
	const int dbg = p.debug;
	const A2DRec *__restrict__ recs = p.recs;
	FastPtrs g = { wavepool, waves, ptab, dbg };

	int ffr[A2D_MAXBATCH / 64];
#pragma unroll
	for(int k = 0; k < A2D_MAXBATCH / 64; ++k)
		ffr[k] = (k * 64 + lane < nfrags) ? p.fragframes[k * 64 + lane] : 0;

	// lane v keeps voice v
	int so[NOSC][OV_NWORDS], sp[8], uu[NOSC + FILT + 1];
	int sf[FS_NWORDS];		// filter12 (FILT): q ramper, lp bp hp, f1, d1 d2, f1next, ramp flag
	int sn[NOSC][2];		// wtosc_noise: held sample, RNG word
#pragma unroll
	for(int o = 0; o < NOSC; ++o)
		sn[o][0] = sn[o][1] = 0;
	int my_off = -1, my_nch = 2, rcur = 0, rend = 0, act = 0, slot = -1;
	// skip_empty: a list of voices whose records the device VM writes (a2amd_vm.hip) - the ones it
	// left without any this batch are the quiet kernels' (runs[].count == 0), not ours
	bool skip = false;
#pragma unroll
	for(int k = 0; k < FS_NWORDS; ++k)
		sf[k] = 0;
#pragma unroll
	for(int o = 0; o < NOSC; ++o)
#pragma unroll
		for(int k = 0; k < OV_NWORDS; ++k)
			so[o][k] = 0;
#pragma unroll
	for(int k = 0; k < 8; ++k)
		sp[k] = 0;
#pragma unroll
	for(int o = 0; o <= NOSC + FILT; ++o)
		uu[o] = 0;
	if(lane < nv) {
		slot = list[first + lane];
		const A2DVoice &vc = voices[slot];
		my_off = vc.out_off;
		my_nch = vc.out_nch;
#pragma unroll
		for(int o = 0; o < NOSC; ++o) {
			uu[o] = vc.unit[o];
			const int *w = ustate + (size_t)uu[o] * A2D_USTATE;
			so[o][OV_MODE] = w[OW_MODE]; so[o][OV_WAVE] = w[OW_WAVE]; so[o][OV_DPHASE] = w[OW_DPHASE];
			so[o][OV_PHLO] = w[OW_PHASE_LO]; so[o][OV_PHHI] = w[OW_PHASE_HI]; so[o][OV_PRAMP] = w[OW_PRAMPING];
#pragma unroll
			for(int k = 0; k < 4; ++k) {
				so[o][OV_P + k] = w[OW_P + k];
				so[o][OV_A + k] = w[OW_A + k];
			}
			sn[o][0] = w[OW_NOISE];
			sn[o][1] = w[OW_SEED];
		}
		if(FILT) {
			uu[NOSC] = vc.unit[NOSC];
			const int *wf = ustate + (size_t)uu[NOSC] * A2D_USTATE;
#pragma unroll
			for(int k = 0; k < 4; ++k)
				sf[FS_Q + k] = wf[FW_Q + k];
			sf[FS_LP] = wf[FW_LP]; sf[FS_BP] = wf[FW_BP]; sf[FS_HP] = wf[FW_HP]; sf[FS_F1] = wf[FW_F1];
			sf[FS_D1] = wf[FW_D1A]; sf[FS_D2] = wf[FW_D2A]; sf[FS_F1NEXT] = wf[FW_F1NEXT]; sf[FS_RAMP] = wf[FW_RAMP];
		}
		uu[NOSC + FILT] = vc.unit[NOSC + FILT];
		const int *wp = ustate + (size_t)uu[NOSC + FILT] * A2D_USTATE;
#pragma unroll
		for(int k = 0; k < 8; ++k)
			sp[k] = wp[k];
		const A2DRun run = p.runs[slot];
		rcur = run.first;
		rend = run.first + run.count;
		skip = skip_empty && run.count == 0;
		act = skip ? 0 : vactive[slot];
	}

#ifdef RECS_PROF
	const long long t_pro = __builtin_readcyclecounter();
#endif
	const int fch = VF ? RECS_VF_FCH : RECS_FCH;
	for(int f0 = 0; f0 < nfrags; f0 += fch) {
		const int nf = min(fch, nfrags - f0);
		int acc0[RECS_FCH], acc1[RECS_FCH];
#pragma unroll
		for(int j = 0; j < RECS_FCH; ++j)
			acc0[j] = acc1[j] = 0;
		int cur_off = nv ? rdl(my_off, 0) : -1, cur_nch = rdl(my_nch, 0);
		// RECS_VFILT: the windows walked but not yet filtered and panned - per voice (lane = voice) a run of
		// pool entries [wbeg, wend) - and the turn-round that works them off
		int wbeg = 0, wend = 0, pool_n = 0;
		int *const tile = recs_dyn + wv * recs_wave_words(vpw);
		int *const pool = tile + RECS_VF_FCH * vpw * 65;
		const int pool_cap = recs_pool_cap(vpw);
		auto turn = [&]() {
			if(!pool_n)
				return;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			{	// lane = voice: f12_process's recurrence (filter12.c:97-118) over this voice's pending windows, in place
				int e = wbeg, s = 0, len = 0, base = 0;
				int f0v = 0, df = 0, qv = 0, qd = 0, lp = 0, bp = 0, hp = 0;
				int d1 = sf[FS_D1], d2 = sf[FS_D2];
				for(;;) {
					if(s >= len && e < wend) {
						const int *en = pool + e * RECS_ENTRY;
						const int w0 = en[0];
						f0v = en[1]; df = en[2]; qv = en[3]; qd = en[4]; lp = en[5]; bp = en[6]; hp = en[7];
						base = ((w0 & 15) * vpw + lane) * 65 + ((w0 >> 4) & 255);
						len = (w0 >> 12) & 255;
						s = 0;
						++e;
					}
					const bool act = s < len;
					const unsigned long long am = __ballot(act);
					if(!am)
						break;
					auto step = [&](int xin, int at) {
						const int f = f0v >> 12, qq = qv >> 12;
						const int d1s = d1 >> 4;
						const int l = wadd(d2, wmul(f, d1s) >> 8);
						const int h = wsub(wsub(xin >> 5, l), wmul(qq, d1s) >> 8);
						const int b = wadd(wmul(f, h >> 4) >> 8, d1);
						tile[at] = wadd(wadd(wmul(l, lp), wmul(b, bp)), wmul(h, hp)) >> 3;
						d1 = b;
						d2 = l;
						f0v = wadd(f0v, df);
						qv = wadd(qv, qd);
					};
					if(!__ballot(act && len - s < 4)) {
						// every busy lane has four frames of its window left: four steps, their inputs read
						// before the first one's chain starts, nothing predicated but the lanes themselves
						if(act) {
							const int at = base + s;
							const int x0 = tile[at], x1 = tile[at + 1], x2 = tile[at + 2], x3 = tile[at + 3];
							step(x0, at);
							step(x1, at + 1);
							step(x2, at + 2);
							step(x3, at + 3);
							s += 4;
						}
					} else if(act) {
						step(tile[base + s], base + s);
						++s;
					}
				}
				sf[FS_D1] = d1;
				sf[FS_D2] = d2;
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			// lane = frame: panmix_process12 (panmix.c:78-125) window by window, voice by voice
			for(int v = 0; v < nv; ++v) {
				const int eb = rdl(wbeg, v), ee = rdl(wend, v);
				if(eb == ee)
					continue;
				const int voff = rdl(my_off, v);
				if(voff != cur_off) {
					flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
					cur_off = voff;
					cur_nch = rdl(my_nch, v);
				}
				for(int e = eb; e < ee; ++e) {
					const int ev = pool[e * RECS_ENTRY + min(lane, RECS_ENTRY - 1)];	// (the entry, a word per lane)
					const int w0 = rdl(ev, 0), volv = rdl(ev, 8), vold = rdl(ev, 9), panv = rdl(ev, 10), pand = rdl(ev, 11);
					const int j = w0 & 15, off = (w0 >> 4) & 255, len = (w0 >> 12) & 255;
					const bool clamp = ((w0 >> 20) & 1) != 0;
					const int fl = lane - off;
					int o0 = 0, o1 = 0;
					if((unsigned)fl < (unsigned)len) {
						const int y = tile[(j * vpw + v) * 65 + lane];
						int vk = wadd(volv, wmul(vold, fl));
						int pk = wadd(panv, wmul(pand, fl));
						int vp = mul64s(pk, vk, 24);
						int v0 = wsub(vk, vp), v1 = wadd(vk, vp);
						if(clamp) {
							int lim = wshl(vk, 1);
							if(v0 > lim) v0 = lim;
							if(v1 > lim) v1 = lim;
						}
						o0 = mul64s(y, v0, 24);
						o1 = mul64s(y, v1, 24);
					}
#pragma unroll
					for(int jj = 0; jj < RECS_FCH; ++jj)
						if(jj == j) {
							acc0[jj] = wadd(acc0[jj], o0);
							acc1[jj] = wadd(acc1[jj], o1);
						}
				}
			}
			wbeg = wend = 0;
			pool_n = 0;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
		};
		for(int v = 0; v < nv; ++v) {
			const int voff = rdl(my_off, v);
			if(!VF && voff != cur_off) {
				flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
				cur_off = voff;
				cur_nch = rdl(my_nch, v);
			}
			OscS os[NOSC];
			Ramp vol, pan;
#pragma unroll
			for(int o = 0; o < NOSC; ++o) {
				osc_from_lanes(os[o], so[o], v);
				os[o].noise = rdl(sn[o][0], v);
				os[o].seed = (unsigned)rdl(sn[o][1], v);
			}
			vol.value = rdl(sp[0], v); vol.target = rdl(sp[1], v); vol.delta = rdl(sp[2], v); vol.timer = rdl(sp[3], v);
			pan.value = rdl(sp[4], v); pan.target = rdl(sp[5], v); pan.delta = rdl(sp[6], v); pan.timer = rdl(sp[7], v);
			FiltS fs;
			if(FILT) {
				fs.q.value = rdl(sf[FS_Q], v); fs.q.target = rdl(sf[FS_Q + 1], v);
				fs.q.delta = rdl(sf[FS_Q + 2], v); fs.q.timer = rdl(sf[FS_Q + 3], v);
				fs.lp = rdl(sf[FS_LP], v); fs.bp = rdl(sf[FS_BP], v); fs.hp = rdl(sf[FS_HP], v);
				fs.f1 = rdl(sf[FS_F1], v); fs.d1 = rdl(sf[FS_D1], v); fs.d2 = rdl(sf[FS_D2], v);
				fs.f1next = rdl(sf[FS_F1NEXT], v); fs.ramp = rdl(sf[FS_RAMP], v);
			}
			int rc = rdl(rcur, v), active = rdl(act, v);
			const int re = rdl(rend, v);
			const bool me = lane == v;
			RecQ nx = rec_issue(recs, rc < re ? rc : 0);
			for(int j = 0; j < nf; ++j) {
				const int f = f0 + j;
				const int n = frames_of(ffr, f);
				int o0 = 0, o1 = 0;
				// one window of the chain: frames [off, off + len) of the fragment
				auto window = [&](int off, int len) {
#ifdef RECS_PROF
					const long long w0 = __builtin_readcyclecounter();
#endif
					const int fl = lane - off;
					int x;
					if(NOSC > 1) {
						const OscPend pa_ = osc_fragment_begin(g, os[0], len, fl);
						const OscPend pb_ = osc_fragment_begin(g, os[NOSC - 1], len, fl);
						x = wadd(osc_fragment_end(pa_), osc_fragment_end(pb_));
					} else
						x = osc_fragment_s(g, os[0], len, fl);
					if(VF) {
						// the window goes on the list: its oscillator samples into the voice's row, what the
						// filter and the pan stage will need of it - all closed forms of the control state -
						// into a pool entry (f12_process's head, filter12.c:86-96; panmix_process12's, panmix.c:84-95)
						if(pool_n == pool_cap)
							turn();
						ramp_prepare_s(fs.q, len);
						const int wf0 = fs.f1;
						int wdf = 0;
						if(fs.ramp) {
							fs.f1 = fs.f1next;
							wdf = rfl(wadd(wsub(fs.f1, wf0), len >> 1) / len);
							fs.ramp = 0;
						}
						const int wqv = fs.q.value, wqd = fs.q.delta;
						ramp_run(fs.q, len);
						const bool clamp = pan.target > 0xffffff || pan.target < -0xffffff ||
								pan.value > 0xffffff || pan.value < -0xffffff;
						ramp_prepare_s(vol, len);
						ramp_prepare_s(pan, len);
						if((unsigned)fl < (unsigned)len)
							tile[(j * vpw + v) * 65 + lane] = x;
						{
							int ev = 0;
							ev = writelane_s(ev, j | (off << 4) | (len << 12) | ((int)clamp << 20), 0);
							ev = writelane_s(ev, wf0, 1); ev = writelane_s(ev, wdf, 2);
							ev = writelane_s(ev, wqv, 3); ev = writelane_s(ev, wqd, 4);
							ev = writelane_s(ev, fs.lp, 5); ev = writelane_s(ev, fs.bp, 6); ev = writelane_s(ev, fs.hp, 7);
							ev = writelane_s(ev, vol.value, 8); ev = writelane_s(ev, vol.delta, 9);
							ev = writelane_s(ev, pan.value, 10); ev = writelane_s(ev, pan.delta, 11);
							if(lane < RECS_ENTRY)
								pool[pool_n * RECS_ENTRY + lane] = ev;
						}
						ramp_run(vol, len);
						ramp_run(pan, len);
						if(rdl(wbeg, v) == rdl(wend, v))
							WRL(wbeg, pool_n);
						++pool_n;
						WRL(wend, pool_n);
					} else {
#ifdef RECS_PROF
					asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x));
					const long long w1 = __builtin_readcyclecounter();
					t_osc += w1 - w0;
#endif
					if(FILT)
						x = RECS_JFILT ? filt_window_j(fs, x, off, len, lane) : filt_window_s(fs, x, off, len, lane);
#ifdef RECS_PROF
					asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x));
					t_flt += __builtin_readcyclecounter() - w1;
#endif
					pan_fragment_s(vol, pan, x, len, fl, o0, o1);
					}
#ifdef RECS_PROF
					asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
					t_win += __builtin_readcyclecounter() - w0 + (o0 & 0);
					++n_win;
#endif
				};
				RecQ cur = rec_ready(nx);
				uint32_t head = rc < re ? (uint32_t)cur.x : 0xffffffffu;
				if((int)A2D_RFRAG(head) != f || rc >= re) {
					// no records in this fragment: the engine called Process(0, frames)
					// once on every unit (core.c:1875-1876)
					if(active)
						window(0, n);
				} else {
					do {
						const int value = cur.y;
						const unsigned dur = (unsigned)cur.z;
						const unsigned start = (unsigned)cur.w;
						const int u = (int)A2D_RUNIT(head), reg = (int)A2D_RREG(head);
						++rc;
						nx = rec_issue(recs, rc < re ? rc : 0);	// (on its way while this one is carried out)
						switch(A2D_ROP(head)) {
						  case R_SEG:
							if(active)
								window((int)(dur & 0xffffu), (int)(dur >> 16));
							break;
						  case R_INIT:
							// (the words this kernel does not keep: wtosc's noise
							// sample and seed)
#pragma unroll
							for(int o = 0; o <= NOSC + FILT; ++o)
								if(u == o && lane < A2D_USTATE)
									ustate[(size_t)rdl(uu[o], v) * A2D_USTATE + lane] = 0;
#pragma unroll
							for(int o = 0; o < NOSC; ++o)
								if(u == o)
									osc_init_s(g, os[o], value);
							if(FILT && u == NOSC) {	// f12_Initialize, filter12.c:180-221; value = f1 from the host
								ramp_init(fs.q, 0);
								ramp_set(fs.q, 32768, 0, 0);	// f12_Q(u, 0, 0, 0)
								fs.lp = 65536 >> 8;
								fs.bp = fs.hp = fs.d1 = fs.d2 = fs.f1next = fs.ramp = 0;
								fs.f1 = value;
								if(VF) {	// (the recurrence's state lives in the voice's lane: nothing of this
									WRL(sf[FS_D1], 0);	// voice is pending - it was not alive before)
									WRL(sf[FS_D2], 0);
								}
							}
							if(u == NOSC + FILT) {	// panmix_Initialize, panmix.c:252-284
								ramp_init(vol, 65536);
								ramp_init(pan, 0);
							}
							active = 1;
							break;
						  case R_WRITE:
#pragma unroll
							for(int o = 0; o < NOSC; ++o)
								if(u == o)
									osc_write_s(g, os[o], reg, value, (int)start, (int)dur);
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
									ramp_set(vol, value, (int)start, (int)dur);
								else
									ramp_set(pan, value, (int)start, (int)dur);
							}
							break;
						  case R_F1SET:		// f12_CutOff without a ramp: the host's coefficient
							if(FILT) {
								fs.f1 = value;
								fs.ramp = 0;
							}
							break;
						  case R_F1RAMP:	// ... and one per window while the cutoff ramps
							if(FILT) {
								fs.f1next = value;
								fs.ramp = 1;
							}
							break;
						  case R_KILL:
							active = 0;
							break;
						  case R_NOISESEED:	// the engine's RNG word as this window of a noise oscillator finds it
#pragma unroll
							for(int o = 0; o < NOSC; ++o)
								if(u == o)
									os[o].seed = (unsigned)value;
							break;
						  default:
							break;
						}
#ifdef RECS_PROF
						++n_rec;
#endif
						cur = rec_ready(nx);
						head = rc < re ? (uint32_t)cur.x : 0xffffffffu;
					} while(rc < re && (int)A2D_RFRAG(head) == f);
				}
#pragma unroll
				for(int jj = 0; jj < RECS_FCH; ++jj)
					if(jj == j) {
						acc0[jj] = wadd(acc0[jj], o0);
						acc1[jj] = wadd(acc1[jj], o1);
					}
			}
#pragma unroll
			for(int o = 0; o < NOSC; ++o) {
				osc_to_lanes(so[o], os[o], me);
				WRL(sn[o][0], os[o].noise);
				WRL(sn[o][1], (int)os[o].seed);
			}
			if(FILT) {
				WRL(sf[FS_Q], fs.q.value); WRL(sf[FS_Q + 1], fs.q.target); WRL(sf[FS_Q + 2], fs.q.delta);
				WRL(sf[FS_Q + 3], fs.q.timer);
				WRL(sf[FS_LP], fs.lp); WRL(sf[FS_BP], fs.bp); WRL(sf[FS_HP], fs.hp); WRL(sf[FS_F1], fs.f1);
				if(!VF) {
					WRL(sf[FS_D1], fs.d1); WRL(sf[FS_D2], fs.d2);
				}
				WRL(sf[FS_F1NEXT], fs.f1next); WRL(sf[FS_RAMP], fs.ramp);
			}
			WRL(sp[0], vol.value); WRL(sp[1], vol.target); WRL(sp[2], vol.delta); WRL(sp[3], vol.timer);
			WRL(sp[4], pan.value); WRL(sp[5], pan.target); WRL(sp[6], pan.delta); WRL(sp[7], pan.timer);
			WRL(rcur, rc);
			WRL(act, active);
		}
		if(VF)
			turn();
		// Sixteen thousand voices playing straight into one bus are as many atomic
		// adds on the same 512 bytes per fragment; the workgroup sums its wavefronts'
		// chunks in LDS first (neighbours in the list share their bus: it is sorted).
		{
#ifdef RECS_PROF
			const long long r0 = __builtin_readcyclecounter();
#endif
			const int pb = (f0 / fch) & 1;
			if((int)(blockDim.x >> 6) == 1) {
				// (a launch of a few dozen voices - a song - comes with ONE wavefront per workgroup:
				// nobody to sum with, no barrier, no trip through LDS)
				flush_acc(busmem, cur_off, cur_nch, f0, nf, lane, dbg, acc0, acc1);
			} else {
#pragma unroll
			for(int j = 0; j < RECS_FCH; ++j) {
				part[pb][wv][2 * j][lane] = acc0[j];
				part[pb][wv][2 * j + 1][lane] = acc1[j];
			}
			if(lane == 0) {
				part_off[pb][wv] = cur_off;
				part_nch[pb][wv] = cur_nch;
			}
			__syncthreads();
			// wavefront r owns row r of the chunk: fragment r / 2, channel r % 2
			// (a launch of a few dozen voices - a song - comes with ONE wavefront per
			// workgroup: nobody to sum with, and nobody's slow chunk to wait for)
			const int wpb = (int)(blockDim.x >> 6);
			for(int row = wv; row < 2 * nf; row += wpb) {
				int sum = 0, off = -1, nch = 2;
				for(int w = 0; w <= wpb; ++w) {
					const int woff = w < wpb ? part_off[pb][w] : -2;
					if(woff != off) {
						if(off >= 0 && sum && !(dbg & 1))
							atomicAdd(busmem + off + ((size_t)(f0 + (row >> 1)) * nch + (row & 1)) * A2D_FRAG + lane, sum);
						sum = 0;
						off = woff;
						nch = w < wpb ? part_nch[pb][w] : 2;
					}
					if(w < wpb && woff >= 0)
						sum = wadd(sum, part[pb][w][row][lane]);
				}
			}
			}
#ifdef RECS_PROF
			asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
			t_rec += __builtin_readcyclecounter() - r0;
#endif
		}
	}

#ifdef RECS_PROF
	const long long t_loop = __builtin_readcyclecounter();
#endif
	// state out (these voices are nobody else's this batch: straight to the state array)
	if(lane < nv && !skip) {
#pragma unroll
		for(int o = 0; o < NOSC; ++o) {
			int *w = ustate + (size_t)uu[o] * A2D_USTATE;
			w[OW_MODE] = so[o][OV_MODE]; w[OW_WAVE] = so[o][OV_WAVE]; w[OW_DPHASE] = so[o][OV_DPHASE];
			w[OW_PHASE_LO] = so[o][OV_PHLO]; w[OW_PHASE_HI] = so[o][OV_PHHI]; w[OW_PRAMPING] = so[o][OV_PRAMP];
#pragma unroll
			for(int k = 0; k < 4; ++k) {
				w[OW_P + k] = so[o][OV_P + k];
				w[OW_A + k] = so[o][OV_A + k];
			}
			w[OW_NOISE] = sn[o][0];
			w[OW_SEED] = sn[o][1];
		}
		if(FILT) {
			int *wf = ustate + (size_t)uu[NOSC] * A2D_USTATE;
#pragma unroll
			for(int k = 0; k < 4; ++k)
				wf[FW_Q + k] = sf[FS_Q + k];
			wf[FW_LP] = sf[FS_LP]; wf[FW_BP] = sf[FS_BP]; wf[FW_HP] = sf[FS_HP]; wf[FW_F1] = sf[FS_F1];
			wf[FW_D1A] = sf[FS_D1]; wf[FW_D2A] = sf[FS_D2]; wf[FW_F1NEXT] = sf[FS_F1NEXT]; wf[FW_RAMP] = sf[FS_RAMP];
		}
		int *wp = ustate + (size_t)uu[NOSC + FILT] * A2D_USTATE;
#pragma unroll
		for(int k = 0; k < 8; ++k)
			wp[k] = sp[k];
		vactive[slot] = act;
	}
#ifdef RECS_PROF
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
	if(nv > 0 && (FILT ? (blockIdx.x % 64) < 24 : (blockIdx.x % 175) < 2) && lane == 0)
		printf("k_leaf_recs<%d,%d> block %d wave %d: %d voices x %d fragments: prologue %lld, loop %lld (of which %d windows %lld - oscillators %lld, filter %lld -, %d records, block sums %lld), epilogue %lld cycles; %llu ticks of 10 ns\n",
				NOSC, FILT, (int)blockIdx.x, wv, nv, nfrags, t_pro - t_in, t_loop - t_pro, n_win, t_win, t_osc, t_flt, n_rec, t_rec,
				(long long)__builtin_readcyclecounter() - t_loop, (unsigned long long)__builtin_amdgcn_s_memrealtime() - rt_in);
#endif
}


#ifdef RECS_WPE
#define RECS_ATTR __attribute__((amdgpu_waves_per_eu(RECS_WPE, 8)))
#else
#define RECS_ATTR
#endif
template<int NOSC, int FILT, int VFT = 0>
__global__ __launch_bounds__(64 * RECS_WPB) RECS_ATTR
void k_leaf_recs(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int vpw,
		const A2DVoice *__restrict__ voices, int *ustate, int *vactive,
		const int16_t *__restrict__ wavepool, const A2DWave *__restrict__ waves,
		const uint32_t *__restrict__ ptab, int *__restrict__ busmem, int skip_empty)
{
	__shared__ RecsPart part[2];
	__shared__ int part_off[2][RECS_WPB], part_nch[2][RECS_WPB];
	recs_body<NOSC, FILT, VFT>(pp, list, nlist, vpw, (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), voices, ustate, vactive,
			wavepool, waves, ptab, busmem, part, part_off, part_nch, skip_empty);
}

// All four kinds in one launch, for plumbing-sized scenes (a song: a few dozen voices of each kind):
// on one stream the per-kind launches run one after the other although they are independent, and each
// takes as long as ONE voice's serial walk through the batch.  A workgroup looks its kind up in the
// segment table (whole workgroups per kind: they sum their wavefronts' output in LDS).
struct RecsSegs { const int *list[4]; int count[4]; };	// (1,0) (2,0) (1,1) (2,1)

__global__ __launch_bounds__(64 * RECS_WPB)
void k_leaf_recs_all(const A2DParams *__restrict__ pp, RecsSegs segs, int vpw,
		const A2DVoice *__restrict__ voices, int *ustate, int *vactive,
		const int16_t *__restrict__ wavepool, const A2DWave *__restrict__ waves,
		const uint32_t *__restrict__ ptab, int *__restrict__ busmem, int skip_mask)
{
	__shared__ RecsPart part[2];
	__shared__ int part_off[2][RECS_WPB], part_nch[2][RECS_WPB];
	int b = (int)blockIdx.x, kind = -1;
	const int wpb = (int)(blockDim.x >> 6);
#pragma unroll
	for(int k = 0; k < 4; ++k) {
		const int nb = ((segs.count[k] + vpw - 1) / vpw + wpb - 1) / wpb;
		if(kind < 0) {
			if(b < nb)
				kind = k;
			else
				b -= nb;
		}
	}
	const int gw = b * wpb + (int)(threadIdx.x >> 6);
#define RECS_BODY(K, N, F) case K: recs_body<N, F>(pp, segs.list[K], segs.count[K], vpw, gw, voices, ustate, vactive, \
		wavepool, waves, ptab, busmem, part, part_off, part_nch, (skip_mask >> K) & 1); break
	switch(rfl(kind)) {
	  RECS_BODY(0, 1, 0);
	  RECS_BODY(1, 2, 0);
	  RECS_BODY(2, 1, 1);
	  RECS_BODY(3, 2, 1);
	}
#undef RECS_BODY
}

// Wavefronts per workgroup of the records kernels: RECS_WPB to sum in LDS before the bus
// atomics, ONE while all of the launch's wavefronts are resident at once anyway (32 KB of LDS
// per workgroup: five per CU) - then each voice runs at its own pace instead of meeting the
// seven others at a barrier every four fragments (a song: the slowest voice changes from
// chunk to chunk, and the sum of the chunks' maxima was 15-20 % more than the slowest voice's
// own time; 64 fragments of 512 / 1 024 / 2 048 scripted filter voices: 0.49 / 0.52 / 1.21 ms
// with one wavefront per workgroup, 0.61 / 0.62 / 0.63 with eight).
static int recs_wpb(int nwaves)
{
	static const int force = getenv("A2AMD_RECS_WPB") ? atoi(getenv("A2AMD_RECS_WPB")) : 0;
	if(force >= 1 && force <= RECS_WPB)
		return force;
	return nwaves <= 1024 ? 1 : RECS_WPB;
}

int a2d_launch_leaf_recs_all(const A2DParams *dparams, const A2DParams &hp, const int *const *lists, const int *counts,
		int vpw, void *stream, int skip_mask)
{
	RecsSegs segs;
	int nblocks = 0, nwaves = 0;
	vpw = min(max(vpw, 1), 64);
	for(int k = 0; k < 4; ++k)
		nwaves += (counts[k] + vpw - 1) / vpw;
	const int wpb = recs_wpb(nwaves);
	for(int k = 0; k < 4; ++k) {
		segs.list[k] = lists[k];
		segs.count[k] = counts[k];
		nblocks += ((counts[k] + vpw - 1) / vpw + wpb - 1) / wpb;
	}
	if(!nblocks)
		return 0;
	// (dynamic LDS: the filter kinds' window rows and pool, RECS_VFILT)
	// (plumbing-sized launches keep the scalar recurrence: no rows, no pool)
	hipLaunchKernelGGL(k_leaf_recs_all, dim3(nblocks), dim3(64 * wpb), 0, (hipStream_t)stream, dparams, segs, vpw,
			hp.voices, hp.ustate, hp.vactive, hp.wavepool, hp.waves, hp.ptab, hp.busmem, skip_mask);
	return (int)hipGetLastError();
}

int a2d_launch_leaf_recs(const A2DParams *dparams, const A2DParams &hp, int nosc, int filt, const int *dlist,
		int nlist, int vpw, void *stream, int skip_empty)
{
	if(nlist <= 0)
		return 0;
	vpw = min(max(vpw, 1), 64);
	// The lane = voice filter (RECS_VFILT) pays from a few voices per wavefront up: at 4 096 voices and
	// more the launch gets 4-8 voices per wavefront (fewer, fatter wavefronts: the scalar walk of a voice
	// is latency, the recurrence 12 vector instructions per frame for all of a wavefront's voices).
	// (measured, 64 fragments of voices with a split window and a pitch ramp in every second fragment: 16 384
	// voices 3.46 -> 2.24 ms, 65 536: 13.5 -> 10.0; 4 096: 0.99 -> 1.36.  Against the window filter along the
	// lanes (RECS_JFILT, the end of round 4) the margin is thinner - 12 288 voices 2.02 (window filter) / 2.17 ms,
	// 16 384: 2.58 / 2.21, 32 768: 4.97 / 5.00, 65 536: 9.7 with it; two oscillators 16 384: 4.29 / 3.88,
	// 32 768: 8.24 / 6.58 (profiles/r04_jfilt_ab.txt) - hence the threshold.  A2AMD_VFILT=0 / 1
	// forces it off / on: A/B measurements, and the tests run the path at sizes an oracle can follow.)
	const char *fv = getenv("A2AMD_VFILT");
	const int force_vf = fv ? atoi(fv) : -1;
	const bool vf = filt && RECS_VFILT && (force_vf >= 0 ? force_vf != 0 : nlist >= 16384);
	if(vf && !getenv("A2AMD_RVPW"))
		vpw = min(max((nlist + 4095) / 4096, 4), 8);
	if(vf)
		vpw = min(vpw, 16);
	const int nwaves = (nlist + vpw - 1) / vpw;
	int wpb = recs_wpb(nwaves);
	size_t dyn = 0;
	if(vf) {
		// the workgroup's static 32 KB (bus sums) + its wavefronts' rows and pools within 64 KB
		const size_t per_wave = (size_t)recs_wave_words_host(vpw) * sizeof(int);
		wpb = (int)max((size_t)1, min((size_t)wpb, (size_t)(30 * 1024) / per_wave));
		dyn = wpb * per_wave;
	}
	const int nblocks = (nwaves + wpb - 1) / wpb;
#define RECS_LAUNCH(N, F) hipLaunchKernelGGL((k_leaf_recs<N, F, 0>), dim3(nblocks), dim3(64 * wpb), dyn, \
		(hipStream_t)stream, dparams, dlist, nlist, vpw, hp.voices, hp.ustate, hp.vactive, hp.wavepool, \
		hp.waves, hp.ptab, hp.busmem, skip_empty)
	if(nosc == 1 && !filt)
		RECS_LAUNCH(1, 0);
	else if(nosc == 2 && !filt)
		RECS_LAUNCH(2, 0);
	else if(nosc == 1 && !vf)
		RECS_LAUNCH(1, 1);
	else if(!vf)
		RECS_LAUNCH(2, 1);
#undef RECS_LAUNCH
#define RECS_LAUNCH(N, F) hipLaunchKernelGGL((k_leaf_recs<N, F, 1>), dim3(nblocks), dim3(64 * wpb), dyn, \
		(hipStream_t)stream, dparams, dlist, nlist, vpw, hp.voices, hp.ustate, hp.vactive, hp.wavepool, \
		hp.waves, hp.ptab, hp.busmem, skip_empty)
	else if(nosc == 1)
		RECS_LAUNCH(1, 1);
	else
		RECS_LAUNCH(2, 1);
#undef RECS_LAUNCH
	return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// wtosc -> filter12 (1 ch) -> panmix 1->2: the BASELINE config 3 voice
// ---------------------------------------------------------------------------
// The oscillator and the pan stage are independent per frame (lane = frame,
// throughput work); the filter (f12_process, filter12.c:74-119) is a recurrence in
// time that no lane can help another with (lane = voice, a dependent chain of 64
// steps per fragment).  Round 1 ran the three stages one after the other on ONE
// wavefront owning 8 voices: the filter stage then issues a wave-wide instruction
// for 8 busy lanes, 176 of the 253 vector instructions per voice-fragment.  Here a
// workgroup of FILT_WAVES wavefronts owns up to 64 voices and runs the stages as a
// pipeline over the fragments, a barrier per step:
//
//   one wavefront    B(f)    every lane = one voice: filter fragment f in place
//   the others       A(f+1)  oscillators of fragment f+1 -> rows of the next tile
//                    C(f-1)  rows of fragment f-1 x pan gains -> bus
//
// through a ring of three [voices][64+1] LDS tiles (+1: row and column accesses
// both bank-conflict free).  A step is as long as the filter wavefront's chain of
// dependent instructions - 64 frames x 12 instructions, about 4 600 cycles - so
// everything that is not the recurrence has been moved off that wavefront (filt_step)
// and the other stages are kept well below that: wavefronts whose voices are all
// settled run a loop of their own with the per-voice values in scalar registers (37
// vector instructions per voice and fragment), and the workgroup's bus sums meet in
// LDS before one wavefront adds them to the bus in device memory.
// Round 3 measurements (16 384 voices x 256 fragments): 1.15 ms with 8 wavefronts x 32
// voices and the general loop only; 0.85 with the all-settled loop; 0.72 with 16
// wavefronts x 64 voices; 0.68 with the 12-instruction recurrence; 0.58 with the LDS
// bus sums (configs[4]'s share, 32 768 voices on 128 buses: 2.42 -> 1.12 ms).
#define FILT_MAXV   64
#define FILT_PITCH  65
#ifndef FILT_ROWAHEAD
#define FILT_ROWAHEAD 1	// filt_row reads the next sixteen frames while it filters these sixteen
#endif
#ifndef FILT_WAVES
#define FILT_WAVES  16	// wavefronts per workgroup: one filters, the others run the oscillators / pans.  (One
			// workgroup of 16 per CU with 64 voices - every lane of the filter wavefront busy - since the
			// oscillator wavefronts have their all-settled loop; 8 and two workgroups per CU before.)
#endif
#ifndef FILT_BATCH
#define FILT_BATCH  5	// settled voices whose coefficient loads are in flight together
#endif
#ifndef FILT_PARTNER
#define FILT_PARTNER 0	// voices of an oscillator wavefront that shares the filter wavefront's SIMD
#endif
// Two shapes measured in round 3 and left switched off (tools/filt_sweep_*.sh, profiles/r03_filt_sweep.jsonl:
// 1.25 - 1.38 ms against 1.26 ms per 256 fragments x 16 384 voices - the oscillator wavefronts that
// sit next to a filter wavefront just wait longer at the barrier):
#ifndef FILT_LIGHT
#define FILT_LIGHT -1	// voices of an oscillator wavefront next to the OTHER workgroup's filter wavefront (-1: even share)
#endif
#ifndef FILT_FASTV
#define FILT_FASTV 6	// most voices an oscillator wavefront takes through its all-settled loop (0: the general loop only)
#endif
#ifndef FILT_ROT
#define FILT_ROT 0	// 1: workgroups that share a CU put their filter wavefronts on different SIMDs
#endif
enum { FV_Q = 0, FV_LP = 4, FV_BP, FV_HP, FV_F1, FV_D1, FV_D2, FV_NWORDS };

// one filter step (f12_process, filter12.c:98-117).  The filter wavefront's chain of dependent
// instructions is what a pipeline step waits for, so two things that are not part of the recurrence
// are done by the stages around it, all lanes busy: the oscillator stage stores the input already
// shifted (x5 = in >> 5), and where the whole workgroup runs pure low pass filters (LPRAW) the row
// keeps l and the pan stage scales it, (l * lp) >> 3.  12 instructions per frame instead of 15.
template<bool LPRAW>
DEV int filt_step(int x5, int qq, int ff, int lp, int bp, int hp, int &d1, int &d2)
{
	const int d1s = d1 >> 4;
	const int l = wadd(d2, wmul(ff, d1s) >> 8);
	const int h = wsub(wsub(x5, l), wmul(qq, d1s) >> 8);
	const int b = wadd(wmul(ff, h >> 4) >> 8, d1);
	d1 = b;
	d2 = l;
	return LPRAW ? l : (wadd(wadd(wmul(l, lp), wmul(b, bp)), wmul(h, hp)) >> 3);
}

// the filter along one voice's row: n frames in place.  The LDS round trip (~130
// cycles) must not sit on the recurrence: a full fragment is taken sixteen frames at
// a time - sixteen reads in flight, sixteen steps in registers, sixteen writes.
template<bool LPRAW, bool QREST>
DEV void filt_row(int *row, int n, int ff, int lp, int bp, int hp, int &d1, int &d2, int &qv, int qdelta)
{
	if(n == A2D_FRAG) {
#if FILT_ROWAHEAD
		// (round 4: the next sixteen frames are on their way from the LDS while these sixteen are filtered -
		// two register sets; before, each of a fragment's four groups waited out its own LDS round trip)
		int xb[2][16];
#pragma unroll
		for(int k = 0; k < 16; ++k)
			xb[0][k] = row[k];
#pragma unroll
		for(int g = 0; g < A2D_FRAG / 16; ++g) {
			if(g + 1 < A2D_FRAG / 16) {
#pragma unroll
				for(int k = 0; k < 16; ++k)
					xb[(g + 1) & 1][k] = row[(g + 1) * 16 + k];
			}
#pragma unroll
			for(int k = 0; k < 16; ++k) {
				xb[g & 1][k] = filt_step<LPRAW>(xb[g & 1][k], qv >> 12, ff, lp, bp, hp, d1, d2);
				if(!QREST)
					qv = wadd(qv, qdelta);
			}
#pragma unroll
			for(int k = 0; k < 16; ++k)
				row[g * 16 + k] = xb[g & 1][k];
		}
#else
#pragma unroll 1
		for(int s0 = 0; s0 < A2D_FRAG; s0 += 16) {
			int x[16];
#pragma unroll
			for(int k = 0; k < 16; ++k)
				x[k] = row[s0 + k];
#pragma unroll
			for(int k = 0; k < 16; ++k) {
				x[k] = filt_step<LPRAW>(x[k], qv >> 12, ff, lp, bp, hp, d1, d2);
				if(!QREST)
					qv = wadd(qv, qdelta);
			}
#pragma unroll
			for(int k = 0; k < 16; ++k)
				row[s0 + k] = x[k];
		}
#endif
		return;
	}
	for(int s = 0; s < n; ++s) {
		row[s] = filt_step<LPRAW>(row[s], qv >> 12, ff, lp, bp, hp, d1, d2);
		if(!QREST)
			qv = wadd(qv, qdelta);
	}
}

#ifndef FILT_WPE
#define FILT_WPE 4
#endif
#ifndef FILT_AHEAD
#define FILT_AHEAD 1	// the all-settled loop asks for a fragment's coefficient entries a step ahead
#endif
// a workgroup barrier that waits for this wavefront's LDS traffic only: loads from device memory
// stay in flight across it (__syncthreads() is a fence: it waits for them too)
DEV void filt_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ unsigned g_filt_turn[4096];	// FILT_ROT == 2: workgroups arriving on a CU take turns (k_leaf_oscfiltpan)
// Round 6: the body is shared by two kernels - NOSC = 1, k_leaf_oscfiltpan (wtosc -> filter12 -> panmix, configs[2]),
// and NOSC = 2, k_leaf_osc2filtpan: wtosc; wtosc (adding); filter12; panmix, the usual subtractive note (every lead of the
// reference's benchmark/k2*.a2s), which rounds 2-5 rendered through the records / window kernels whether or not it
// carried records.  The second oscillator adds into the voice's scratch before the filter reads it
// (wtosc.c:200-236 with A2_PROCADD: out[s] += v * a >> 17, a wrapping int32 add; filter12.c:98: in >> 5 of the SUM), so a
// row of the tile holds (xa + xb) >> 5: twice the taps per row, the filter and pan stages as they were.  An oscillator
// wavefront's all-settled loop holds 4 tap pairs x 3 registers per voice instead of 2, hence fewer voices per wavefront
// (FILT2_FASTV) and per workgroup (a2d_launch_leaf_osc2filtpan).  Everything added for NOSC = 2 is behind
// if constexpr(NOSC == 2).  (With a plain if the discarded branches still disturbed the register allocation of the
// one-oscillator kernel: 98 scratch accesses instead of 73.  As it stands k_leaf_oscfiltpan is round 5's code up to
// register numbering and the grouping of a few prologue loads - not bit-identical (tools/isa_mix.py hash
// 39e9eb94cc881251 -> 97b9066429681ca1), the same 21 858 instructions, the same 73 scratch accesses in the same
// loops, 154 of 160 mnemonic counts equal and the other six off by one: profiles/r06_oscfiltpan_spill_sites.txt.
// What configs[2] takes on it is measured)
// Voices an oscillator wavefront of the two-oscillator kernel takes through its all-settled loop: the LAUNCHER deals at
// most FILT2_LAUNCHV = 3 (a2d_osc2filtpan_max_vpg) - four tap pairs x 3 registers per voice in flight across the barrier
// (FILT_AHEAD): the 4-voice loop holds 27 - 29 scratch reloads and as many vmcnt(0) stalls per trip
// (profiles/r06_oscfiltpan_spill_sites.txt) and MEASURED 3.5x slower (16 384 voices x 256 fragments: 48 voices per
// workgroup = 4 per wavefront 3.83 ms, 32 = 2 - 3 per wavefront 1.09 ms, 64 = 5 - 6 per wavefront, general loop, 1.50 ms;
// profiles/r06_osc2filtpan_shapes.txt).  The KERNEL is nevertheless built with the 4-voice loop in it (FILT2_FASTV 4;
// a shape forced by A2AMD_F2VPW reaches it, the parity tests do): built WITHOUT it (FILT2_FASTV 3) the same 32-voice
// shape takes 1.34 ms instead of 1.08 - same box, interleaved, three runs each, and once more on another box
// (profiles/r06_fastv_build_ab.txt) - although the 2- and 3-voice loops and the filter wavefront's loop it executes
// are the same code in both builds (same instruction counts and event order: tools/r06 notes in DESIGN 6).  Where the
// code sits is what is left; NOT understood, and kept because it is measured.
#ifndef FILT2_FASTV
#define FILT2_FASTV 4
#endif
#ifndef FILT2_LAUNCHV
#define FILT2_LAUNCHV 3
#endif
template<int NOSC>
DEV void oscfiltpan_body(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int vpg,
		const A2DVoice *__restrict__ voices, int *ustate, const int16_t *__restrict__ wavepool,
		const A2DWave *__restrict__ waves, const uint32_t *__restrict__ ptab, int *__restrict__ busmem,
		const int *__restrict__ wavecoef)
{
	constexpr int FASTV = NOSC == 1 ? FILT_FASTV : FILT2_FASTV;
	constexpr int UF = NOSC, UP = NOSC + 1;		// chain positions of filter12 and panmix
	extern __shared__ __attribute__((aligned(16))) int tiles[];	// 3 x [vpg][FILT_PITCH]
	const A2DParams &p = *pp;
	const int wv = rfl((int)(threadIdx.x >> 6));	// (wave-uniform, and known to the compiler as such)
	const int lane = threadIdx.x & 63;
	const int first = blockIdx.x * vpg;
	const int nv = min(vpg, nlist - first);		// (the grid has no empty workgroups)
	const int nfrags = p.nfrags;
	const int dbg = p.debug;
	FastPtrs g = { wavepool, waves, ptab, dbg };
	const CoefRsrc crs = coef_rsrc(wavecoef);
	const int tsize = vpg * FILT_PITCH;

	// (fragment lengths straight from the parameter block, once per stage: no per-lane tables)
#define FILT_FRAMES(f) ((int)rfl((int)p.fragframes[f]))
#define FILT_START(f)  ((int)rfl((int)p.fragstart[f]))

	// Which wavefront filters, and which oscillator wavefronts sit next to a filter wavefront.
	// Where a wavefront runs is the hardware's choice (observed: wavefronts w and w + 4 of a
	// workgroup share a SIMD, simd = {0, 2, 1, 3}[(w + s) & 3] with s changing from workgroup to
	// workgroup), so with FILT_ROT == 2 the roles follow the PHYSICAL SIMD ids (s_getreg HW_ID):
	// the workgroups that arrive on a CU take turns (a counter per CU in device memory) putting
	// their filter wavefront - a stream of dependent instructions that wants a SIMD's issue slots -
	// on SIMD 0 and on SIMD 1; the other wavefront of that SIMD stays idle ("partner"), and the
	// oscillator wavefronts that sit on the OTHER workgroup's filter SIMD take fewer voices ("light").
	// The workgroup's share of a bus, fragment by fragment (a ring of three like the tiles): where all
	// its voices mix into ONE stereo bus the oscillator / pan wavefronts add their sums here (LDS
	// atomics) and an idle wavefront next to the filter ("flusher") adds the total to the bus in device
	// memory one step later - one pair of global atomics per workgroup and fragment instead of one per
	// wavefront (16 384 voices straight into the root bus: every wavefront of the chip on the same 128 words).
	__shared__ int s_acc[3][2][A2D_FRAG];
	__shared__ int s_simd[FILT_WAVES];
	__shared__ int s_target;
	int fw = 0, my_simd = wv & 3, tgt = 0;
	if(FILT_ROT == 2) {
		my_simd = (int)__builtin_amdgcn_s_getreg(2308) & 3;		// HW_ID.simd_id
		if(lane == 0)
			s_simd[wv] = my_simd;
		if(threadIdx.x == 0) {
			const unsigned hw = __builtin_amdgcn_s_getreg(63492);	// HW_ID: cu_id 11:8, sh_id 12, se_id 15:13
			const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 15u;	// XCC_ID
			const unsigned key = (xcc << 8) | ((hw >> 8) & 255u);
			s_target = (int)(atomicAdd(&g_filt_turn[key & 4095u], 1u) & 1u);
		}
		__syncthreads();
		tgt = s_target;
		fw = -1;
		for(int k = FILT_WAVES - 1; k >= 0; --k)
			if(s_simd[k] == tgt)
				fw = k;		// the first wavefront on that SIMD
		if(fw < 0) {		// (nobody there: the roles by wavefront number, as without FILT_ROT)
			fw = 0;
			tgt = s_simd[0];
		}
	} else if(FILT_ROT)
		fw = (int)((blockIdx.x >> 8) & 3u);
	if(wv == fw) {
		// ================= the filter wavefront: lane = voice =================
		// (its dependent chain is the workgroup's critical path: first in line for
		// the issue slots of the SIMD it shares with oscillator wavefronts)
		__builtin_amdgcn_s_setprio(3);
		int fv[FV_NWORDS], u1 = 0;
#pragma unroll
		for(int k = 0; k < FV_NWORDS; ++k)
			fv[k] = 0;
		bool mine = false;
		if(lane < nv)
			mine = p.runs[list[first + lane]].count == 0;
		if(mine) {
			u1 = voices[list[first + lane]].unit[UF];
			const int *w1 = ustate + (size_t)u1 * A2D_USTATE;
#pragma unroll
			for(int k = 0; k < 4; ++k)
				fv[FV_Q + k] = w1[FW_Q + k];
			fv[FV_LP] = w1[FW_LP]; fv[FV_BP] = w1[FW_BP]; fv[FV_HP] = w1[FW_HP];
			fv[FV_F1] = w1[FW_F1]; fv[FV_D1] = w1[FW_D1A]; fv[FV_D2] = w1[FW_D2A];
		}
		// the common shapes, decided for the whole wavefront: low pass only (for the batch: the
		// oscillator wavefronts come to the same answer from the same words), q at rest (per fragment)
		const bool lponly = __all(!mine || (fv[FV_BP] == 0 && fv[FV_HP] == 0));
#ifdef FILT_PROF
		long long tb = 0, tw = 0;
#endif
		for(int st = -1; st <= nfrags; ++st) {
			const int f = st;
#ifdef FILT_PROF
			const long long c0 = __builtin_readcyclecounter();
#endif
			if(f >= 0 && f < nfrags && mine) {
				const int n = FILT_FRAMES(f);
				int *row = tiles + (f % 3) * tsize + lane * FILT_PITCH;
				Ramp q = { fv[FV_Q], fv[FV_Q + 1], fv[FV_Q + 2], fv[FV_Q + 3] };
				ramp_prepare(q, n);	// (f12_process, filter12.c:86-96; no cutoff ramp in a quiet voice)
				const int ff = fv[FV_F1] >> 12, lp = fv[FV_LP], bp = fv[FV_BP], hp = fv[FV_HP];
				int d1 = fv[FV_D1], d2 = fv[FV_D2], qv = q.value;
				const bool qrest = __all(q.delta == 0);
				if(lponly && qrest)
					filt_row<true, true>(row, n, ff, lp, bp, hp, d1, d2, qv, q.delta);
				else if(lponly)
					filt_row<true, false>(row, n, ff, lp, bp, hp, d1, d2, qv, q.delta);
				else if(qrest)
					filt_row<false, true>(row, n, ff, lp, bp, hp, d1, d2, qv, q.delta);
				else
					filt_row<false, false>(row, n, ff, lp, bp, hp, d1, d2, qv, q.delta);
				ramp_run(q, n);
				fv[FV_Q] = q.value; fv[FV_Q + 1] = q.target; fv[FV_Q + 2] = q.delta; fv[FV_Q + 3] = q.timer;
				fv[FV_D1] = d1;
				fv[FV_D2] = d2;
			}
#ifdef FILT_PROF
			const long long c1 = __builtin_readcyclecounter();
#endif
			__syncthreads();
#ifdef FILT_PROF
			tb += c1 - c0;
			tw += __builtin_readcyclecounter() - c1;
#endif
		}
#ifdef FILT_PROF
		if((blockIdx.x == 7 || blockIdx.x == 263) && lane == 0 && nfrags > 100)
			printf("block %d wave %d simd %d cu %d (filter, %d voices): %lld cycles filtering, %lld waiting (%d fragments)\n",
					(int)blockIdx.x, wv, (int)__builtin_amdgcn_s_getreg(2308), (int)__builtin_amdgcn_s_getreg(6660), nv, tb, tw, nfrags);
#endif
		if(mine) {
			int *w1 = ustate + (size_t)u1 * A2D_USTATE;
#pragma unroll
			for(int k = 0; k < 4; ++k)
				w1[FW_Q + k] = fv[FV_Q + k];
			w1[FW_D1A] = fv[FV_D1];
			w1[FW_D2A] = fv[FV_D2];
		}
		return;
	}

	// ============ oscillator / pan wavefronts: lane = frame, voices [vb, ve) ============
	// Every fourth wavefront of a workgroup lands on the SIMD of wavefront 0, the filter, whose
	// stream of dependent instructions is the workgroup's critical path: those "partner"
	// wavefronts take FILT_PARTNER voices each (0: they only meet the others at the barriers) -
	// what the filter's SIMD has left over when the other three carry the rest of the
	// oscillator / pan work - and the other wavefronts share the remaining voices evenly.
	// With FILT_ROT two workgroups share a CU, their filter wavefronts on SIMDs fw and fw ^ 1: the
	// oscillator wavefronts of THIS workgroup that run on SIMD fw ^ 1 sit next to the other
	// workgroup's filter wavefront and take fewer voices ("light": FILT_LIGHT each, -1 = an even
	// share), so that the four SIMDs of the CU carry about the same number of instructions.
	int npart = FILT_WAVES / 4 - 1;			// the other wavefronts on the filter's SIMD
	int nlight = (FILT_ROT && FILT_LIGHT >= 0) ? FILT_WAVES / 4 : 0;
	int role = 0, ridx = 0;				// 0 full, 1 light, 2 partner; my index among those of my role
	if(FILT_ROT == 2) {
		npart = nlight = 0;
		for(int k = 0; k < FILT_WAVES; ++k) {
			if(k == fw)
				continue;
			const int r = s_simd[k] == tgt ? 2 : (FILT_LIGHT >= 0 && s_simd[k] == (tgt ^ 1)) ? 1 : 0;
			npart += r == 2;
			nlight += r == 1;
			if(k == wv)
				role = r;
		}
		for(int k = 0; k < wv; ++k)
			if(k != fw)
				ridx += (s_simd[k] == tgt ? 2 : (FILT_LIGHT >= 0 && s_simd[k] == (tgt ^ 1)) ? 1 : 0) == role;
	} else {
		const int sm = wv & 3;
		role = sm == fw ? 2 : (nlight && sm == (fw ^ 1)) ? 1 : 0;
		if(role == 2)
			ridx = (wv >> 2) - 1;			// (wv >> 2 == 0 is the filter wavefront)
		else if(role == 1)
			ridx = wv >> 2;
		else {
			for(int k = 0; k < sm; ++k)
				ridx += (k != fw && !(nlight && k == (fw ^ 1)));
			ridx += (wv >> 2) * (nlight ? 2 : 3);
		}
	}
	const bool flusher = role == 2 && ridx == 0;
	bool wgbus = false;
	int wg_off = -1;
	{
		int off_l = -1, nch_l = 0;
		bool m_l = false;
		if(lane < nv && p.runs[list[first + lane]].count == 0) {
			m_l = true;
			off_l = voices[list[first + lane]].out_off;
			nch_l = voices[list[first + lane]].out_nch;
		}
		const unsigned long long any = __ballot(m_l);
		if(any && npart > 0 && !(dbg & 8)) {	// (A2AMD_DEBUG bit 3: every wavefront straight to the bus, as before)
			wg_off = rdl(off_l, (int)__builtin_ctzll(any));
			wgbus = wg_off >= 0 && __all(!m_l || (off_l == wg_off && nch_l == 2));
		}
	}
	if(flusher) {
		for(int k = lane; k < 3 * 2 * A2D_FRAG; k += 64)
			(&s_acc[0][0][0])[k] = 0;
	}
	// one fragment's sums of this wavefront onto its bus
	auto bus_add = [&](int off, int nch, int fc, int acc0, int acc1) __attribute__((always_inline)) {
		if(dbg & 1)
			return;
		if(wgbus) {
			if(acc0) atomicAdd(&s_acc[fc % 3][0][lane], acc0);
			if(acc1) atomicAdd(&s_acc[fc % 3][1][lane], acc1);
		} else {
			int *dst = busmem + off + (size_t)fc * nch * A2D_FRAG;
			if(acc0) atomicAdd(&dst[lane], acc0);
			if(acc1) atomicAdd(&dst[A2D_FRAG + lane], acc1);
		}
	};
	// the flusher: the workgroup's sum of fragment f (complete since the last barrier) to the bus
	auto bus_flush = [&](int f) __attribute__((always_inline)) {
		const int a0 = s_acc[f % 3][0][lane], a1 = s_acc[f % 3][1][lane];
		s_acc[f % 3][0][lane] = 0;
		s_acc[f % 3][1][lane] = 0;
		int *dst = busmem + wg_off + (size_t)f * 2 * A2D_FRAG;
		if(a0) atomicAdd(&dst[lane], a0);
		if(a1) atomicAdd(&dst[A2D_FRAG + lane], a1);
	};
	const int nfull = max(1, FILT_WAVES - 1 - npart - nlight);
	const int pshare = min(FILT_PARTNER, nv / (FILT_WAVES - 1));	// (never more than an even share)
	const int lshare = nlight ? min(FILT_LIGHT, nv / (FILT_WAVES - 1)) : 0;
	const int rem = nv - npart * pshare - nlight * lshare;
	// (an even deal: 32 voices over six wavefronts are 6 6 5 5 5 5 - with ceil(32 / 6) each the
	// last one got 2 and the two SIMDs that hold the first four carried 12 voices to the third's 8)
	const int per = rem / nfull, extra = rem % nfull;
	int vb, ve;
	if(role == 2) {
		vb = rem + nlight * lshare + ridx * pshare;
		ve = min(nv, vb + pshare);
	} else if(role == 1) {
		vb = rem + ridx * lshare;
		ve = min(nv, vb + lshare);
	} else {
		vb = ridx * per + min(ridx, extra);
		ve = min(rem, vb + per + (ridx < extra ? 1 : 0));
	}
	const int mv = rfl(max(0, ve - vb));		// my voices: lane l parks voice vb + l

	int sv[SV_NWORDS], dv[DV_NWORDS];
	int u0 = 0, u2 = 0, my_off = -1, my_nch = 2, my_lp = 0;
	// The row format of the batch (filt_step): pure low pass filters in the whole workgroup leave
	// l in the rows and the pan stage applies lp.  Every wavefront works it out from the same words.
	bool lpraw;
	{
		bool plain = true;
		if(lane < nv && p.runs[list[first + lane]].count == 0) {
			const int *w1 = ustate + (size_t)voices[list[first + lane]].unit[UF] * A2D_USTATE;
			plain = w1[FW_BP] == 0 && w1[FW_HP] == 0;
		}
		lpraw = __all(plain);
	}
	bool mine = false;	// this lane's voice is ours to render (no records this batch)
#pragma unroll
	for(int k = 0; k < SV_NWORDS; ++k)
		sv[k] = 0;
#pragma unroll
	for(int k = 0; k < DV_NWORDS; ++k)
		dv[k] = 0;
	// NOSC == 2: the adding oscillator's state words (the SV_ words up to the amplitude ramper) and what is
	// derived from them once per launch (the DV_ words DV_MM .. DV_DOFF); DV_SETTLED speaks for both
	int sw[NOSC == 2 ? SV_VOL : 1], dw[NOSC == 2 ? DV_V0 : 1], u1b = 0;
	if constexpr(NOSC == 2) {
#pragma unroll
		for(int k = 0; k < SV_VOL; ++k)
			sw[k] = 0;
#pragma unroll
		for(int k = 0; k < DV_V0; ++k)
			dw[k] = 0;
	}
	if(lane < mv)
		mine = p.runs[list[first + vb + lane]].count == 0;
	const unsigned long long mine_mask = __ballot(mine);
	if(mine) {
		const A2DVoice &vc = voices[list[first + vb + lane]];
		u0 = vc.unit[0];
		u2 = vc.unit[UP];
		my_lp = ustate[(size_t)vc.unit[UF] * A2D_USTATE + FW_LP];
		my_off = vc.out_off;
		my_nch = vc.out_nch;
		const int *w0 = ustate + (size_t)u0 * A2D_USTATE;
		const int *w2 = ustate + (size_t)u2 * A2D_USTATE;
		sv[SV_MODE] = w0[OW_MODE]; sv[SV_WAVE] = w0[OW_WAVE]; sv[SV_DPHASE] = w0[OW_DPHASE];
		sv[SV_PHLO] = w0[OW_PHASE_LO]; sv[SV_PHHI] = w0[OW_PHASE_HI]; sv[SV_PRAMP] = w0[OW_PRAMPING];
#pragma unroll
		for(int k = 0; k < 4; ++k) {
			sv[SV_P + k] = w0[OW_P + k];
			sv[SV_A + k] = w0[OW_A + k];
			sv[SV_VOL + k] = w2[PW_VOL + k];
			sv[SV_PAN + k] = w2[PW_PAN + k];
		}
		bool settled = sv[SV_MODE] == A2D_OSC_MIPWAVE && sv[SV_DPHASE] && !sv[SV_PRAMP] &&
				!(sv[SV_P + 3] | sv[SV_P + 2] | sv[SV_A + 3] | sv[SV_A + 2] |
				  sv[SV_VOL + 3] | sv[SV_VOL + 2] | sv[SV_PAN + 3] | sv[SV_PAN + 2]) &&
				sv[SV_P] == sv[SV_P + 1] && sv[SV_A] == sv[SV_A + 1] &&
				sv[SV_VOL] == sv[SV_VOL + 1] && sv[SV_PAN] == sv[SV_PAN + 1];
		if constexpr(NOSC == 2) {
			u1b = vc.unit[1];
			const int *wb = ustate + (size_t)u1b * A2D_USTATE;
			sw[SV_MODE] = wb[OW_MODE]; sw[SV_WAVE] = wb[OW_WAVE]; sw[SV_DPHASE] = wb[OW_DPHASE];
			sw[SV_PHLO] = wb[OW_PHASE_LO]; sw[SV_PHHI] = wb[OW_PHASE_HI]; sw[SV_PRAMP] = wb[OW_PRAMPING];
#pragma unroll
			for(int k = 0; k < 4; ++k) {
				sw[SV_P + k] = wb[OW_P + k];
				sw[SV_A + k] = wb[OW_A + k];
			}
			settled = settled && sw[SV_MODE] == A2D_OSC_MIPWAVE && sw[SV_DPHASE] && !sw[SV_PRAMP] &&
					!(sw[SV_P + 3] | sw[SV_P + 2] | sw[SV_A + 3] | sw[SV_A + 2]) &&
					sw[SV_P] == sw[SV_P + 1] && sw[SV_A] == sw[SV_A + 1];
		}
		if(settled) {
			const A2DWave *w = waves + sv[SV_WAVE];
			const unsigned period = w->period, dphase = (unsigned)sv[SV_DPHASE];
			unsigned dph = ((dphase + 255) >> 8) * period, mm = 0;
			for(; (dph > (A2D_MAXPHINC << 8)) && (mm < A2D_MIPS - 1); ++mm)
				dph >>= 1;
			dph = (unsigned)(((uint64_t)dphase * period) >> mm);
			settled = w->size[0] && (w->flags & 0x100u) && dph <= (A2D_MAXPHINC << 16);
			dv[DV_MM] = (int)mm;
			dv[DV_DPH] = (int)dph;
			dv[DV_SIZEM] = (int)w->size[mm];
			dv[DV_DOFF] = (int)w->off[mm];
			if constexpr(NOSC == 2) {
				const A2DWave *w2 = waves + sw[SV_WAVE];
				const unsigned period2 = w2->period, dphase2 = (unsigned)sw[SV_DPHASE];
				unsigned dph2 = ((dphase2 + 255) >> 8) * period2, mm2 = 0;
				for(; (dph2 > (A2D_MAXPHINC << 8)) && (mm2 < A2D_MIPS - 1); ++mm2)
					dph2 >>= 1;
				dph2 = (unsigned)(((uint64_t)dphase2 * period2) >> mm2);
				settled = settled && w2->size[0] && (w2->flags & 0x100u) && dph2 <= (A2D_MAXPHINC << 16);
				dw[DV_MM] = (int)mm2;
				dw[DV_DPH] = (int)dph2;
				dw[DV_SIZEM] = (int)w2->size[mm2];
				dw[DV_DOFF] = (int)w2->off[mm2];
			}
			const int vol = sv[SV_VOL], pan = sv[SV_PAN];
			const int vp = mul64s(pan, vol, 24);
			int v0 = wsub(vol, vp), v1 = wadd(vol, vp);
			if(pan > 0xffffff || pan < -0xffffff) {
				int lim = wshl(vol, 1);
				if(v0 > lim) v0 = lim;
				if(v1 > lim) v1 = lim;
			}
			dv[DV_V0] = v0;
			dv[DV_V1] = v1;
		}
		dv[DV_SETTLED] = settled ? 1 : 0;
	}
	// Every lane keeps its settled voice's phase at the start of the next fragment to
	// render, in units of the voice's mip level and already reduced mod size << 24
	// (wtosc.c:259-285: "ph %= size << 24; ...; ph += frames * dph" per fragment is
	// this recurrence) - one vector update per stage for all voices, instead of a
	// scalar 64 bit multiply and modulo per voice and stage.
	uint64_t curph = 0;
	if(mine && dv[DV_SETTLED]) {
		const uint64_t phase = (uint64_t)(unsigned)sv[SV_PHLO] | ((uint64_t)(unsigned)sv[SV_PHHI] << 32);
		curph = (phase >> (unsigned)dv[DV_MM]) % ((uint64_t)(unsigned)dv[DV_SIZEM] << 24);
	}
	uint64_t lastph = 0;	// ... and at the start of the last fragment rendered (for the end state)
	uint64_t endph = 0;	// (the all-settled loop: the unwrapped phase after the last fragment)
	uint64_t curph2 = 0, lastph2 = 0, endph2 = 0;	// NOSC == 2: the same three for the adding oscillator
	if(NOSC == 2 && mine && dv[DV_SETTLED]) {
		const uint64_t phase = (uint64_t)(unsigned)sw[SV_PHLO] | ((uint64_t)(unsigned)sw[SV_PHHI] << 32);
		curph2 = (phase >> (unsigned)dw[DV_MM]) % ((uint64_t)(unsigned)dw[DV_SIZEM] << 24);
	}

	// ---- The common case, a path of its own: every voice of this wavefront is ours, settled and
	// mixes into the same stereo bus.  What the stages need per voice is then wave-uniform and constant
	// over the batch - phase increment, table offset, amplitude, pan gains, bus - and lives in
	// scalar registers for the whole fragment loop (the general loop below fetches it with a
	// v_readlane per value, voice and fragment, behind lane masks it has to evaluate first); the
	// phases advance in the scalar unit; lane * dph is kept per voice.  37 vector instructions per
	// voice and fragment instead of 75.  (One barrier per step like the general loop and the
	// filter wavefront: the wavefronts of a workgroup choose their loop independently.)
	const bool fast = FASTV > 0 && !flusher && mv > 0 && mv <= FASTV && mine_mask == ((1ull << mv) - 1ull) &&
			__all(lane >= mv || (dv[DV_SETTLED] && my_nch == 2 && my_off >= 0 && my_off == rdl(my_off, 0)));
	// (one copy of the loop per number of voices, 1 .. FILT_FASTV: no guards inside)
	auto fast_loop = [&](auto nvc, auto wgc) __attribute__((always_inline)) {
		constexpr int NV = decltype(nvc)::value;
		constexpr bool WG = decltype(wgc)::value;	// (the workgroup's sums meet in LDS: s_acc)
		unsigned s_dph[NV], s_size[NV], ldph[NV];
		int s_cb[NV], g_amp[NV], g_v0[NV], g_v1[NV], g_lp[NV];
		uint64_t s_ph[NV];
		constexpr int NB = NOSC == 2 ? NV : 1;		// the adding oscillator's copies of the same
		unsigned t_dph[NB], t_size[NB], ldph2[NB];
		int t_cb[NB], g_amp2[NB];
		uint64_t t_ph[NB];
		const int bus = rdl(my_off, 0);
#pragma unroll
		for(int k = 0; k < NV; ++k) {
			s_dph[k] = (unsigned)rdl(dv[DV_DPH], k);
			s_size[k] = (unsigned)rdl(dv[DV_SIZEM], k);
			s_cb[k] = coef_base((unsigned)rdl(dv[DV_DOFF], k));
			// (the gains of a voice as vector registers, the same value in every lane:
			// the scalar registers do not hold nine values for each of six voices)
			g_amp[k] = rdl(sv[SV_A], k);
			g_v0[k] = rdl(dv[DV_V0], k);
			g_v1[k] = rdl(dv[DV_V1], k);
			g_lp[k] = rdl(my_lp, k);
			asm("" : "+v"(g_amp[k]), "+v"(g_v0[k]), "+v"(g_v1[k]), "+v"(g_lp[k]));
			s_ph[k] = (uint64_t)(unsigned)rdl((int)(unsigned)curph, k) |
					((uint64_t)(unsigned)rdl((int)(unsigned)(curph >> 32), k) << 32);
			ldph[k] = lane_dph(lane, s_dph[k]);
			if constexpr(NOSC == 2) {
				t_dph[k] = (unsigned)rdl(dw[DV_DPH], k);
				t_size[k] = (unsigned)rdl(dw[DV_SIZEM], k);
				t_cb[k] = coef_base((unsigned)rdl(dw[DV_DOFF], k));
				g_amp2[k] = rdl(sw[SV_A], k);
				asm("" : "+v"(g_amp2[k]));
				t_ph[k] = (uint64_t)(unsigned)rdl((int)(unsigned)curph2, k) |
						((uint64_t)(unsigned)rdl((int)(unsigned)(curph2 >> 32), k) << 32);
				ldph2[k] = lane_dph(lane, t_dph[k]);
			}
		}
		// The coefficient entries of the fragment the phases stand at, all voices' loads in flight together.
		// FILT_AHEAD (round 4): asked for at the END of the step before.  The wavefronts of a workgroup march
		// in step - a barrier per fragment - so loads issued at the top of a step found all four wavefronts
		// of a SIMD waiting for them at the same time, with only the pan stage to hide behind; issued before
		// the barrier they have the barrier and the pan stage to arrive (the barrier itself waits for the
		// LDS only: filt_barrier).  configs[2]: 0.581 -> 0.546 ms per 256 fragments, configs[4] 8.74 -> 8.19.
		Coef4 ka[NV], kb[NV];
		unsigned pa[NV], pb[NV];
		Coef4 kc[NB], kd[NB];
		unsigned pc[NB], pd[NB];
		auto ask = [&]() __attribute__((always_inline)) {
#pragma unroll
			for(int k = 0; k < NV; ++k) {
				pa[k] = tap_phase(s_ph[k], ldph[k]);
				pb[k] = pa[k] + (s_dph[k] >> 17);
				ka[k] = coef_at(crs, s_cb[k], pa[k]);
				kb[k] = coef_at(crs, s_cb[k], pb[k]);
				if constexpr(NOSC == 2) {
					pc[k] = tap_phase(t_ph[k], ldph2[k]);
					pd[k] = pc[k] + (t_dph[k] >> 17);
					kc[k] = coef_at(crs, t_cb[k], pc[k]);
					kd[k] = coef_at(crs, t_cb[k], pd[k]);
				}
			}
		};
#if FILT_AHEAD
		if(nfrags > 0)
			ask();
#endif
#ifdef FILT_PROF
		long long ta = 0, tc = 0, tw = 0;
#endif
		for(int st = -1; st <= nfrags; ++st) {
#ifdef FILT_PROF
			const long long c0 = __builtin_readcyclecounter();
#endif
			// ---- A: the coefficient entries of fragment st + 1 were asked for a step ago (FILT_AHEAD) ----
			const int fa = st + 1;
#if !FILT_AHEAD
			if(fa < nfrags)
				ask();
#endif
			// ---- C: pan + mix-down of fragment st - 1 (rows hold zeros past a short fragment's end) ----
			const int fc = st - 1;
			if(fc >= 0) {
				const int *tile = tiles + (fc % 3) * tsize + vb * FILT_PITCH;
				int y[NV];
#pragma unroll
				for(int k = 0; k < NV; ++k)
					y[k] = tile[k * FILT_PITCH + lane];
				int acc0 = 0, acc1 = 0;
				if(lpraw) {
#pragma unroll
					for(int k = 0; k < NV; ++k)
						y[k] = wmul(y[k], g_lp[k]) >> 3;
				}
#pragma unroll
				for(int k = 0; k < NV; ++k) {
					acc0 = wadd(acc0, mul64s(y[k], g_v0[k], 24));
					acc1 = wadd(acc1, mul64s(y[k], g_v1[k], 24));
				}
				if(!(dbg & 1)) {
					if(WG) {
						if(acc0) atomicAdd(&s_acc[fc % 3][0][lane], acc0);
						if(acc1) atomicAdd(&s_acc[fc % 3][1][lane], acc1);
					} else {
						int *dst = busmem + bus + (size_t)fc * 2 * A2D_FRAG;
						if(acc0) atomicAdd(&dst[lane], acc0);
						if(acc1) atomicAdd(&dst[A2D_FRAG + lane], acc1);
					}
				}
			}
#ifdef FILT_PROF
			const long long c1 = __builtin_readcyclecounter();
#endif
			// ---- A, second half: Hermite, amplitude, rows of the next tile; the phases move on ----
			if(fa < nfrags) {
				const int n = FILT_FRAMES(fa);
				int *tile = tiles + (fa % 3) * tsize + vb * FILT_PITCH;
				int x[NV];
#pragma unroll
				for(int k = 0; k < NV; ++k) {
					if constexpr(NOSC == 2)	// (the adding oscillator: a wrapping add in the voice's scratch, then in >> 5)
						x[k] = wadd(mul64s(hermite_c(ka[k], pa[k]) + hermite_c(kb[k], pb[k]), g_amp[k], 17),
								mul64s(hermite_c(kc[k], pc[k]) + hermite_c(kd[k], pd[k]), g_amp2[k], 17)) >> 5;
					else
					x[k] = mul64s(hermite_c(ka[k], pa[k]) + hermite_c(kb[k], pb[k]), g_amp[k], 17) >> 5;	// (x5: filt_step)
				}
				if(n == A2D_FRAG) {
#pragma unroll
					for(int k = 0; k < NV; ++k)
						tile[k * FILT_PITCH + lane] = x[k];
				} else {
#pragma unroll
					for(int k = 0; k < NV; ++k)
						tile[k * FILT_PITCH + lane] = (lane < n) ? x[k] : 0;
				}
				// (after the last fragment the phases stay unwrapped: what the epilogue stores, wtosc.c:284)
				const bool wrap = fa != nfrags - 1;
#pragma unroll
				for(int k = 0; k < NV; ++k) {
					uint64_t ph = s_ph[k] + (uint64_t)s_dph[k] * (unsigned)n;
					unsigned hi = (unsigned)(ph >> 24);
					if(hi >= s_size[k] && wrap) {
						hi -= s_size[k];
						if(hi >= s_size[k])
							hi %= s_size[k];
						ph = ((uint64_t)hi << 24) | (ph & 0xffffffu);
					}
					s_ph[k] = ph;
					if constexpr(NOSC == 2) {
						uint64_t ph2 = t_ph[k] + (uint64_t)t_dph[k] * (unsigned)n;
						unsigned hi2 = (unsigned)(ph2 >> 24);
						if(hi2 >= t_size[k] && wrap) {
							hi2 -= t_size[k];
							if(hi2 >= t_size[k])
								hi2 %= t_size[k];
							ph2 = ((uint64_t)hi2 << 24) | (ph2 & 0xffffffu);
						}
						t_ph[k] = ph2;
					}
				}
			}
#if FILT_AHEAD
			if(fa + 1 < nfrags)
				ask();		// (the phases have moved on to fragment fa + 1)
#endif
#ifdef FILT_PROF
			const long long c2 = __builtin_readcyclecounter();
#endif
#if FILT_AHEAD
			filt_barrier();
#else
			__syncthreads();
#endif
#ifdef FILT_PROF
			ta += c1 - c0;
			tc += c2 - c1;
			tw += __builtin_readcyclecounter() - c2;
#endif
		}
#ifdef FILT_PROF
		if((blockIdx.x == 7 || blockIdx.x == 263) && lane == 0 && nfrags > 100)
			printf("block %d wave %d simd %d cu %d (osc/pan, all-settled loop, %d voices, sums %s): %lld cycles loads+pan, %lld hermite+rows, %lld waiting\n",
					(int)blockIdx.x, wv, (int)__builtin_amdgcn_s_getreg(2308), (int)__builtin_amdgcn_s_getreg(6660), NV,
					WG ? "in LDS" : "to the bus", ta, tc, tw);
#endif
#pragma unroll
		for(int k = 0; k < NV; ++k)
			if(lane == k) {
				endph = s_ph[k];
				if constexpr(NOSC == 2)
					endph2 = t_ph[k];
			}
	};
	if(fast) {
		switch(mv) {
#define FAST_CASE(N) case N: if(N <= FASTV) { if(wgbus) fast_loop(std::integral_constant<int, N>{}, std::true_type{}); \
			else fast_loop(std::integral_constant<int, N>{}, std::false_type{}); } break;
		FAST_CASE(1) FAST_CASE(2) FAST_CASE(3) FAST_CASE(4) FAST_CASE(5) FAST_CASE(6) FAST_CASE(7) FAST_CASE(8)
#undef FAST_CASE
		}
		// nothing but the oscillators' phases has moved (the general loop's epilogue below writes
		// every word back; here the other state words are not even kept in registers)
		if(mine) {
			const uint64_t ph = endph << (unsigned)dv[DV_MM];
			int *w0 = ustate + (size_t)u0 * A2D_USTATE;
			w0[OW_PHASE_LO] = (int)(unsigned)ph;
			w0[OW_PHASE_HI] = (int)(unsigned)(ph >> 32);
			if constexpr(NOSC == 2) {
				const uint64_t ph2 = endph2 << (unsigned)dw[DV_MM];
				int *wb = ustate + (size_t)u1b * A2D_USTATE;
				wb[OW_PHASE_LO] = (int)(unsigned)ph2;
				wb[OW_PHASE_HI] = (int)(unsigned)(ph2 >> 32);
			}
		}
		return;
	}

#ifdef FILT_PROF
	long long ta = 0, tc = 0, tw = 0;
#endif
	for(int st = -1; st <= nfrags; ++st) {
#ifdef FILT_PROF
		const long long c0 = __builtin_readcyclecounter();
#endif
		// ---- A: oscillators of fragment st + 1, frame = lane.  The coefficient loads of
		// the wavefront's first FILT_BATCH settled voices are issued here, the pan stage
		// of fragment st - 1 runs while they are in flight, the Hermite arithmetic after.
		const int fa = st + 1;
		Coef4 pqa[FILT_BATCH], pqb[FILT_BATCH];
		unsigned pqph[FILT_BATCH], pqph2[FILT_BATCH];
		int pnb = 0;
		if(fa < nfrags) {
			// (NOSC == 2: no batched prefix - this loop is the two-oscillator kernel's slow path, voice by voice)
			while(NOSC == 1 && pnb < FILT_BATCH && pnb < mv && ((mine_mask >> pnb) & 1ull) && rdl(dv[DV_SETTLED], pnb))
				++pnb;
			const unsigned cur_lo = (unsigned)curph, cur_hi = (unsigned)(curph >> 32);
			// (branch free: an unused slot fetches voice 0's entries again - with branches
			// around the slots the compiler waits for each pair of loads before the next)
			if(pnb) {
#pragma unroll
				for(int k = 0; k < FILT_BATCH; ++k) {
					const int vk = k < pnb ? k : 0;
					const unsigned dph = (unsigned)rdl(dv[DV_DPH], vk), doff = (unsigned)rdl(dv[DV_DOFF], vk);
					const uint64_t ph = (uint64_t)(unsigned)rdl((int)cur_lo, vk) |
							((uint64_t)(unsigned)rdl((int)cur_hi, vk) << 32);
					pqph[k] = tap_phase(ph, (unsigned)lane * dph);
					pqph2[k] = pqph[k] + (dph >> 17);
					const int cb = coef_base(doff);
					pqa[k] = coef_at(crs, cb, pqph[k]);
					pqb[k] = coef_at(crs, cb, pqph2[k]);
				}
			}
		}
		// ---- C: pan + mix-down of fragment st - 1, frame = lane ----
		const int fc = st - 1;
		if(fc >= 0) {
			const int n = FILT_FRAMES(fc);
			const int *tile = tiles + (fc % 3) * tsize + vb * FILT_PITCH;
			int acc0 = 0, acc1 = 0;
			int cur_off = rdl(my_off, 0), cur_nch = rdl(my_nch, 0);
			// (the rows of the first FILT_BATCH voices are fetched together: one LDS
			// round trip instead of one per voice)
			int yrow[FILT_BATCH];
#pragma unroll
			for(int k = 0; k < FILT_BATCH; ++k)
				yrow[k] = (k < mv) ? tile[k * FILT_PITCH + lane] : 0;
			for(int v = 0; v < mv; ++v) {
				if(!((mine_mask >> v) & 1ull))
					continue;
				const int voff = rdl(my_off, v);
				if(voff != cur_off) {
					if(cur_off >= 0)
						bus_add(cur_off, cur_nch, fc, acc0, acc1);
					acc0 = acc1 = 0;
					cur_off = voff;
					cur_nch = rdl(my_nch, v);
				}
				int y;
				if(v < FILT_BATCH) {
					y = yrow[0];
#pragma unroll
					for(int k = 1; k < FILT_BATCH; ++k)
						y = (v == k) ? yrow[k] : y;
				} else
					y = tile[v * FILT_PITCH + lane];
				if(lpraw)
					y = wmul(y, rdl(my_lp, v)) >> 3;	// (filt_step: the row holds l)
				if(rdl(dv[DV_SETTLED], v)) {
					const int v0 = rdl(dv[DV_V0], v), v1 = rdl(dv[DV_V1], v);
					if(lane < n) {
						acc0 = wadd(acc0, mul64s(y, v0, 24));
						acc1 = wadd(acc1, mul64s(y, v1, 24));
					}
				} else {
					Ramp vol, pan;
					vol.value = rdl(sv[SV_VOL], v); vol.target = rdl(sv[SV_VOL + 1], v);
					vol.delta = rdl(sv[SV_VOL + 2], v); vol.timer = rdl(sv[SV_VOL + 3], v);
					pan.value = rdl(sv[SV_PAN], v); pan.target = rdl(sv[SV_PAN + 1], v);
					pan.delta = rdl(sv[SV_PAN + 2], v); pan.timer = rdl(sv[SV_PAN + 3], v);
					pan_fragment_s(vol, pan, y, n, lane, acc0, acc1);
					const bool me = lane == v;
					WRL(sv[SV_VOL], vol.value); WRL(sv[SV_VOL + 1], vol.target);
					WRL(sv[SV_VOL + 2], vol.delta); WRL(sv[SV_VOL + 3], vol.timer);
					WRL(sv[SV_PAN], pan.value); WRL(sv[SV_PAN + 1], pan.target);
					WRL(sv[SV_PAN + 2], pan.delta); WRL(sv[SV_PAN + 3], pan.timer);
				}
			}
			if(cur_off >= 0)
				bus_add(cur_off, cur_nch, fc, acc0, acc1);
		}
		if(flusher && wgbus && st >= 2 && !(dbg & 1))
			bus_flush(st - 2);
#ifdef FILT_PROF
		const long long c1 = __builtin_readcyclecounter();
#endif
		// ---- A, second half ----
		if(fa < nfrags) {
			const int n = FILT_FRAMES(fa);
			int *tile = tiles + (fa % 3) * tsize + vb * FILT_PITCH;
			const unsigned cur_lo = (unsigned)curph, cur_hi = (unsigned)(curph >> 32);
			if(pnb) {
				int xs[FILT_BATCH];
#pragma unroll
				for(int k = 0; k < FILT_BATCH; ++k) {
					int sm = hermite_c(pqa[k], pqph[k]) + hermite_c(pqb[k], pqph2[k]);
					xs[k] = mul64s(sm, rdl(sv[SV_A], k < pnb ? k : 0), 17);
				}
#pragma unroll
				for(int k = 0; k < FILT_BATCH; ++k)
					if(k < pnb)
						tile[k * FILT_PITCH + lane] = (lane < n) ? xs[k] >> 5 : 0;	// (x5: filt_step)
			}
			for(int v = pnb; v < mv; ++v) {
				int x;
				if(!((mine_mask >> v) & 1ull)) {
					tile[v * FILT_PITCH + lane] = 0;
					continue;
				}
				if(rdl(dv[DV_SETTLED], v)) {
					const unsigned dph = (unsigned)rdl(dv[DV_DPH], v), doff = (unsigned)rdl(dv[DV_DOFF], v);
					const int amp = rdl(sv[SV_A], v);
					const uint64_t ph = (uint64_t)(unsigned)rdl((int)cur_lo, v) | ((uint64_t)(unsigned)rdl((int)cur_hi, v) << 32);
					unsigned ph16 = tap_phase(ph, (unsigned)lane * dph);
					const unsigned ph16b = ph16 + (dph >> 17);
					const int cb = coef_base(doff);
					int sm = hermite_c(coef_at(crs, cb, ph16), ph16) + hermite_c(coef_at(crs, cb, ph16b), ph16b);
					x = mul64s(sm, amp, 17);
				} else {
					OscS o;
					o.mode = rdl(sv[SV_MODE], v);
					o.wave = rdl(sv[SV_WAVE], v);
					o.dphase = (unsigned)rdl(sv[SV_DPHASE], v);
					o.phase = (uint64_t)(unsigned)rdl(sv[SV_PHLO], v) |
							((uint64_t)(unsigned)rdl(sv[SV_PHHI], v) << 32);
					o.p_ramping = rdl(sv[SV_PRAMP], v);
					o.p.value = rdl(sv[SV_P], v); o.p.target = rdl(sv[SV_P + 1], v);
					o.p.delta = rdl(sv[SV_P + 2], v); o.p.timer = rdl(sv[SV_P + 3], v);
					o.a.value = rdl(sv[SV_A], v); o.a.target = rdl(sv[SV_A + 1], v);
					o.a.delta = rdl(sv[SV_A + 2], v); o.a.timer = rdl(sv[SV_A + 3], v);
					x = osc_fragment_s(g, o, n, lane);
					const bool me = lane == v;
					WRL(sv[SV_MODE], o.mode);
					WRL(sv[SV_WAVE], o.wave);
					WRL(sv[SV_DPHASE], (int)o.dphase);
					WRL(sv[SV_PHLO], (int)(unsigned)o.phase);
					WRL(sv[SV_PHHI], (int)(unsigned)(o.phase >> 32));
					WRL(sv[SV_PRAMP], o.p_ramping);
					WRL(sv[SV_P], o.p.value); WRL(sv[SV_P + 1], o.p.target);
					WRL(sv[SV_P + 2], o.p.delta); WRL(sv[SV_P + 3], o.p.timer);
					WRL(sv[SV_A], o.a.value); WRL(sv[SV_A + 1], o.a.target);
					WRL(sv[SV_A + 2], o.a.delta); WRL(sv[SV_A + 3], o.a.timer);
				}
				if constexpr(NOSC == 2) {
					// the adding oscillator, the same two ways (wtosc.c:200-236 with A2_PROCADD: out[s] += ...)
					int xb;
					if(rdl(dv[DV_SETTLED], v)) {
						const unsigned dph = (unsigned)rdl(dw[DV_DPH], v), doff = (unsigned)rdl(dw[DV_DOFF], v);
						const int amp = rdl(sw[SV_A], v);
						const uint64_t ph = (uint64_t)(unsigned)rdl((int)(unsigned)curph2, v) |
								((uint64_t)(unsigned)rdl((int)(unsigned)(curph2 >> 32), v) << 32);
						unsigned ph16 = tap_phase(ph, (unsigned)lane * dph);
						const unsigned ph16b = ph16 + (dph >> 17);
						const int cb = coef_base(doff);
						int sm = hermite_c(coef_at(crs, cb, ph16), ph16) + hermite_c(coef_at(crs, cb, ph16b), ph16b);
						xb = mul64s(sm, amp, 17);
					} else {
						OscS o;
						o.mode = rdl(sw[SV_MODE], v);
						o.wave = rdl(sw[SV_WAVE], v);
						o.dphase = (unsigned)rdl(sw[SV_DPHASE], v);
						o.phase = (uint64_t)(unsigned)rdl(sw[SV_PHLO], v) |
								((uint64_t)(unsigned)rdl(sw[SV_PHHI], v) << 32);
						o.p_ramping = rdl(sw[SV_PRAMP], v);
						o.p.value = rdl(sw[SV_P], v); o.p.target = rdl(sw[SV_P + 1], v);
						o.p.delta = rdl(sw[SV_P + 2], v); o.p.timer = rdl(sw[SV_P + 3], v);
						o.a.value = rdl(sw[SV_A], v); o.a.target = rdl(sw[SV_A + 1], v);
						o.a.delta = rdl(sw[SV_A + 2], v); o.a.timer = rdl(sw[SV_A + 3], v);
						o.noise = 0;	// (a noise oscillator's every window carries a record: never here)
						o.seed = 0;
						xb = osc_fragment_s(g, o, n, lane);
						const bool me = lane == v;
						WRL(sw[SV_MODE], o.mode);
						WRL(sw[SV_WAVE], o.wave);
						WRL(sw[SV_DPHASE], (int)o.dphase);
						WRL(sw[SV_PHLO], (int)(unsigned)o.phase);
						WRL(sw[SV_PHHI], (int)(unsigned)(o.phase >> 32));
						WRL(sw[SV_PRAMP], o.p_ramping);
						WRL(sw[SV_P], o.p.value); WRL(sw[SV_P + 1], o.p.target);
						WRL(sw[SV_P + 2], o.p.delta); WRL(sw[SV_P + 3], o.p.timer);
						WRL(sw[SV_A], o.a.value); WRL(sw[SV_A + 1], o.a.target);
						WRL(sw[SV_A + 2], o.a.delta); WRL(sw[SV_A + 3], o.a.timer);
					}
					x = wadd(x, xb);
				}
				tile[v * FILT_PITCH + lane] = (lane < n) ? x >> 5 : 0;
			}
			// every settled voice moves on by n frames (all lanes at once)
			if(mine && dv[DV_SETTLED]) {
				lastph = curph;
				uint64_t ph = curph + (uint64_t)(unsigned)dv[DV_DPH] * (unsigned)n;
				const unsigned size = (unsigned)dv[DV_SIZEM];
				unsigned hi = (unsigned)(ph >> 24);
				if(hi >= size) {
					hi -= size;		// (one period further, usually; short mip levels wrap more often)
					if(hi >= size)
						hi = (size & (size - 1)) ? hi % size : (hi & (size - 1));
					ph = ((uint64_t)hi << 24) | (ph & 0xffffffu);
				}
				curph = ph;
				if constexpr(NOSC == 2) {
					lastph2 = curph2;
					uint64_t ph2 = curph2 + (uint64_t)(unsigned)dw[DV_DPH] * (unsigned)n;
					const unsigned size2 = (unsigned)dw[DV_SIZEM];
					unsigned hi2 = (unsigned)(ph2 >> 24);
					if(hi2 >= size2) {
						hi2 -= size2;
						if(hi2 >= size2)
							hi2 = (size2 & (size2 - 1)) ? hi2 % size2 : (hi2 & (size2 - 1));
						ph2 = ((uint64_t)hi2 << 24) | (ph2 & 0xffffffu);
					}
					curph2 = ph2;
				}
			}
		}
#ifdef FILT_PROF
		const long long c2 = __builtin_readcyclecounter();
#endif
		__syncthreads();
#ifdef FILT_PROF
		ta += c1 - c0;
		tc += c2 - c1;
		tw += __builtin_readcyclecounter() - c2;
#endif
	}
	// (step st flushed fragment st - 2; the last fragment's sums were complete at the last barrier)
	if(flusher && wgbus && nfrags >= 1 && !(dbg & 1))
		bus_flush(nfrags - 1);
#ifdef FILT_PROF
	if((blockIdx.x == 7 || blockIdx.x == 263) && lane == 0 && nfrags > 100)
		printf("block %d wave %d simd %d cu %d (osc/pan, %d voices): %lld cycles issue+pan, %lld oscillators, %lld waiting\n",
				(int)blockIdx.x, wv, (int)__builtin_amdgcn_s_getreg(2308), (int)__builtin_amdgcn_s_getreg(6660), mv, ta, tc, tw);
#endif

	if(mine) {
		if(dv[DV_SETTLED]) {
			// the unwrapped end of the last fragment (wtosc.c:284)
			const uint64_t ph = (lastph + (uint64_t)(unsigned)dv[DV_DPH] * (unsigned)FILT_FRAMES(nfrags - 1)) <<
					(unsigned)dv[DV_MM];
			sv[SV_PHLO] = (int)(unsigned)ph;
			sv[SV_PHHI] = (int)(unsigned)(ph >> 32);
		}
		int *w0 = ustate + (size_t)u0 * A2D_USTATE;
		int *w2 = ustate + (size_t)u2 * A2D_USTATE;
		w0[OW_MODE] = sv[SV_MODE]; w0[OW_WAVE] = sv[SV_WAVE]; w0[OW_DPHASE] = sv[SV_DPHASE];
		w0[OW_PHASE_LO] = sv[SV_PHLO]; w0[OW_PHASE_HI] = sv[SV_PHHI]; w0[OW_PRAMPING] = sv[SV_PRAMP];
#pragma unroll
		for(int k = 0; k < 4; ++k) {
			w0[OW_P + k] = sv[SV_P + k];
			w0[OW_A + k] = sv[SV_A + k];
			w2[PW_VOL + k] = sv[SV_VOL + k];
			w2[PW_PAN + k] = sv[SV_PAN + k];
		}
		if constexpr(NOSC == 2) {
			if(dv[DV_SETTLED]) {
				const uint64_t ph = (lastph2 + (uint64_t)(unsigned)dw[DV_DPH] * (unsigned)FILT_FRAMES(nfrags - 1)) <<
						(unsigned)dw[DV_MM];
				sw[SV_PHLO] = (int)(unsigned)ph;
				sw[SV_PHHI] = (int)(unsigned)(ph >> 32);
			}
			int *wb = ustate + (size_t)u1b * A2D_USTATE;
			wb[OW_MODE] = sw[SV_MODE]; wb[OW_WAVE] = sw[SV_WAVE]; wb[OW_DPHASE] = sw[SV_DPHASE];
			wb[OW_PHASE_LO] = sw[SV_PHLO]; wb[OW_PHASE_HI] = sw[SV_PHHI]; wb[OW_PRAMPING] = sw[SV_PRAMP];
#pragma unroll
			for(int k = 0; k < 4; ++k) {
				wb[OW_P + k] = sw[SV_P + k];
				wb[OW_A + k] = sw[SV_A + k];
			}
		}
	}
}

__global__ __launch_bounds__(64 * FILT_WAVES) __attribute__((amdgpu_waves_per_eu(FILT_WPE, FILT_WPE)))
void k_leaf_oscfiltpan(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int vpg,
		const A2DVoice *__restrict__ voices, int *ustate, const int16_t *__restrict__ wavepool,
		const A2DWave *__restrict__ waves, const uint32_t *__restrict__ ptab, int *__restrict__ busmem,
		const int *__restrict__ wavecoef)
{
	oscfiltpan_body<1>(pp, list, nlist, vpg, voices, ustate, wavepool, waves, ptab, busmem, wavecoef);
}

// wtosc; wtosc (adding); filter12; panmix 1->2: the subtractive note without records (round 6)
__global__ __launch_bounds__(64 * FILT_WAVES) __attribute__((amdgpu_waves_per_eu(FILT_WPE, FILT_WPE)))
void k_leaf_osc2filtpan(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int vpg,
		const A2DVoice *__restrict__ voices, int *ustate, const int16_t *__restrict__ wavepool,
		const A2DWave *__restrict__ waves, const uint32_t *__restrict__ ptab, int *__restrict__ busmem,
		const int *__restrict__ wavecoef)
{
	oscfiltpan_body<2>(pp, list, nlist, vpg, voices, ustate, wavepool, waves, ptab, busmem, wavecoef);
}

// ---------------------------------------------------------------------------
// inline -> panmix 2->2 -> xinsert (add, wired): root / group driver voices
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_bus_driver(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int consume,
		A2DCommitSet commits, int *__restrict__ master_host)
{
	__shared__ int fr[A2D_MAXBATCH][5];	// per fragment: vol, dvol, pan, dpan, clamp
	const A2DParams &p = *pp;
	if((int)blockIdx.x >= nlist) {
		// workgroups past the voices: state commits the leaf kernels left behind
		// (different voices than the ones rendered here)
		if(blockIdx.y)
			return;
		int b = (int)blockIdx.x - nlist;
		for(int k = 0; k < commits.n; ++k) {
			const int nb = commit_blocks(commits.c[k]);
			if(b < nb) {
				commit_block(commits.c[k], b, p.voices, p.runs, p.ustate);
				return;
			}
			b -= nb;
		}
		return;
	}
	if(p.runs[list[blockIdx.x]].count)
		return;		// carries records this batch: the general kernel renders it
	const A2DVoice &vc = p.voices[list[blockIdx.x]];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	int *w = p.ustate + (size_t)vc.unit[1] * A2D_USTATE;
	Ramp vol = ramp_load(w + PW_VOL), pan = ramp_load(w + PW_PAN);
	// Both rampers at rest (the usual case): the gains are the same for every
	// fragment, the fragments are independent and the workgroups of this voice
	// (blockIdx.y) split them.  Otherwise workgroup 0 steps the rampers through
	// the batch first and renders it alone.
	const bool settled = !(vol.timer | vol.delta | pan.timer | pan.delta) &&
			vol.value == vol.target && pan.value == pan.target;
	int fbeg = blockIdx.y * 4 + wv, fstep = gridDim.y * 4;
	if(!settled) {
		if(blockIdx.y)
			return;
		fbeg = wv;
		fstep = 4;
		if(threadIdx.x == 0) {
			// panmix_process22's rampers (panmix.c:192-249)
			for(int f = 0; f < p.nfrags; ++f) {
				const int n = p.fragframes[f];
				fr[f][4] = pan.target > 0xffffff || pan.target < -0xffffff ||
						pan.value > 0xffffff || pan.value < -0xffffff;
				ramp_prepare(vol, n);
				ramp_prepare(pan, n);
				fr[f][0] = vol.value; fr[f][1] = vol.delta;
				fr[f][2] = pan.value; fr[f][3] = pan.delta;
				ramp_run(vol, n);
				ramp_run(pan, n);
			}
			ramp_store(w + PW_VOL, vol);
			ramp_store(w + PW_PAN, pan);
		}
		__syncthreads();
	}
	const bool sclamp = pan.value > 0xffffff || pan.value < -0xffffff;
	for(int f = fbeg; f < p.nfrags; f += fstep) {
		if(lane >= p.fragframes[f])
			continue;
		int *src = p.busmem + vc.own_off + (size_t)f * vc.own_nch * A2D_FRAG;
		int *dst = p.busmem + vc.out_off + (size_t)f * vc.out_nch * A2D_FRAG;
		int i0 = src[lane], i1 = src[A2D_FRAG + lane];
		if(consume & 1) {
			// every bus of this batch is read here and nowhere else: leave it
			// zeroed for the next batch (no memset between batches)
			src[lane] = 0;
			src[A2D_FRAG + lane] = 0;
		}
		int vk = settled ? vol.value : wadd(fr[f][0], wmul(fr[f][1], lane));
		int pk = settled ? pan.value : wadd(fr[f][2], wmul(fr[f][3], lane));
		int vp = mul64s(pk, vk, 24);
		int v0 = wsub(vk, vp), v1 = wadd(vk, vp);
		if(settled ? sclamp : (fr[f][4] != 0)) {
			int lim = wshl(vk, 1);
			if(v0 > lim) v0 = lim;
			if(v1 > lim) v1 = lim;
		}
		int o0 = mul64s(i0, v0, 24), o1 = mul64s(i1, v1, 24);
		if((consume & 2) && vc.out_off == 0) {
			// ... and the master bus has one writer, the root: a plain store - straight into the
			// host's (pinned, mapped) readback buffer where the caller wants the audio there: no copy
			// command behind the batch (a2amd_render: 13 us of a 68 us step at configs[1])
			int *m = master_host ? master_host + (size_t)f * vc.out_nch * A2D_FRAG : dst;
			m[lane] = o0;
			m[A2D_FRAG + lane] = o1;
			continue;
		}
		if(o0)
			atomicAdd(&dst[lane], o0);
		if(o1)
			atomicAdd(&dst[A2D_FRAG + lane], o1);
	}
}

// ---------------------------------------------------------------------------
// inline -> fbdelay 2->2 [-> fbdelay 2->2 ...] (last one wired, adding): the group
// voices of BASELINE config 4 / benchmark/fmtest4.a2s:83-95
// ---------------------------------------------------------------------------
// fbdelay (fbdelay.c:69-126) is a recurrence in time only through its delay lines:
// frame t reads what frames t - fbdelay / t - ldelay / t - rdelay wrote.  With every
// tap at least R frames long (and at most 131072 - R), R consecutive frames read
// nothing any of them writes: they are independent, and the units of the chain run
// back to back on each frame.  So the batch is cut into rounds of
// floor(min(tap) / 64) whole fragments - with fmtest4's 631 ... 1379 ms taps the
// whole 256-fragment batch is ONE round - and the 1024 threads of the voice's
// workgroup take the frames of a round in parallel (consecutive threads =
// consecutive delay-line addresses), with a barrier between rounds.  The general
// kernel walks the same chain one fragment after the other on one wavefront.
#define FBC_THREADS 1024
#define FBC_MAXD 4
struct FbdU { int fb, l, r, dry, fbg, lg, rg, pos, add; int *b0, *b1; };

__global__ __launch_bounds__(FBC_THREADS)
void k_bus_fbdchain(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int consume)
{
	const A2DParams &p = *pp;
	const int slot = list[blockIdx.x];
	if(p.runs[slot].count)
		return;		// carries records this batch: the general kernel renders it
	if(!p.vactive[slot])
		return;
	const A2DVoice &vc = p.voices[slot];
	const int nd = vc.nunits - 1;
	const int M = A2D_FBD_BUFSIZE - 1;
	FbdU d[FBC_MAXD];
	int dmin = A2D_FBD_BUFSIZE, dmax = 0;
#pragma unroll
	for(int k = 0; k < FBC_MAXD; ++k)
		if(k < nd) {
			const int *w = p.ustate + (size_t)vc.unit[k + 1] * A2D_USTATE;
			d[k].fb = w[DW_FBDELAY]; d[k].l = w[DW_LDELAY]; d[k].r = w[DW_RDELAY];
			d[k].dry = w[DW_DRYGAIN]; d[k].fbg = w[DW_FBGAIN]; d[k].lg = w[DW_LGAIN]; d[k].rg = w[DW_RGAIN];
			d[k].pos = w[DW_BUFPOS];
			d[k].b0 = p.fbdmem + (size_t)w[DW_BUFIDX] * 2 * A2D_FBD_BUFSIZE;
			d[k].b1 = d[k].b0 + A2D_FBD_BUFSIZE;
			d[k].add = (int)A2D_ADD(p.udesc[vc.unit[k + 1]]);
			dmin = min(dmin, min(d[k].fb, min(d[k].l, d[k].r)));
			dmax = max(dmax, max(d[k].fb, max(d[k].l, d[k].r)));
		}
	// fragments per round (the host only lists voices whose taps allow >= 1)
	const int rf = max(1, min(dmin, A2D_FBD_BUFSIZE - dmax) / A2D_FRAG);
	const int nfrags = p.nfrags;
	int *own = p.busmem + vc.own_off;
	int *out = p.busmem + vc.out_off;
	const int out_nch = vc.out_nch;
	for(int f0 = 0; f0 < nfrags; f0 += rf) {
		const int fend = min(nfrags, f0 + rf);
		for(int idx = f0 * A2D_FRAG + (int)threadIdx.x; idx < fend * A2D_FRAG; idx += FBC_THREADS) {
			const int f = idx >> 6, lane = idx & 63;
			if(lane >= p.fragframes[f])
				continue;
			const int t = (int)p.fragstart[f] + lane;
			int *src = own + (size_t)f * 2 * A2D_FRAG;
			int x0 = src[lane], x1 = src[A2D_FRAG + lane];
			if(consume) {
				// every bus of this batch is read by its owner's fast kernel and
				// nowhere else: leave it zeroed for the next batch
				src[lane] = 0;
				src[A2D_FRAG + lane] = 0;
			}
#pragma unroll
			for(int k = 0; k < FBC_MAXD; ++k)
				if(k < nd) {
					const FbdU &u = d[k];
					const int pos = wadd(u.pos, t);
					int o0 = mul64s(u.b1[(pos - u.fb) & M], u.fbg, 16);
					int o1 = mul64s(u.b0[(pos - u.fb) & M], u.fbg, 16);
					const int t0 = u.b0[(pos - u.l) & M], t1 = u.b1[(pos - u.r) & M];
					u.b0[pos & M] = wadd(x0, o0);
					u.b1[pos & M] = wadd(x1, o1);
					o0 = wadd(o0, mul64s(t0, u.lg, 16));
					o1 = wadd(o1, mul64s(t1, u.rg, 16));
					o0 = wadd(o0, mul64s(x0, u.dry, 16));
					o1 = wadd(o1, mul64s(x1, u.dry, 16));
					if(k == nd - 1) {	// wired, adding: into the output bus
						int *dst = out + (size_t)f * out_nch * A2D_FRAG;
						if(o0)
							atomicAdd(&dst[lane], o0);
						if(o1)
							atomicAdd(&dst[A2D_FRAG + lane], o1);
					} else if(u.add) {
						x0 = wadd(x0, o0);
						x1 = wadd(x1, o1);
					} else {
						x0 = o0;
						x1 = o1;
					}
				}
		}
		if(fend < nfrags)
			__syncthreads();	// (delay-line stores of this round visible to the next)
	}
	if(threadIdx.x == 0) {
		int total = 0;
		for(int f = 0; f < nfrags; ++f)
			total += p.fragframes[f];
		for(int k = 0; k < nd; ++k)
			p.ustate[(size_t)vc.unit[k + 1] * A2D_USTATE + DW_BUFPOS] = wadd(d[k].pos, total);
	}
}

int a2d_launch_bus_fbdchain(const A2DParams *dparams, const int *dlist, int nlist, int consume, void *stream)
{
	if(nlist <= 0)
		return 0;
	hipLaunchKernelGGL(k_bus_fbdchain, dim3(nlist), dim3(FBC_THREADS), 0, (hipStream_t)stream,
			dparams, dlist, nlist, consume);
	return (int)hipGetLastError();
}

int a2d_launch_leaf_oscpan(const A2DParams *dparams, const A2DParams &hp, const int *dlist, int nlist,
		int vpw, int ysplit, int *ustage, void *stream, void *event_after_main, A2DCommit *defer)
{
	if(nlist <= 0)
		return 0;
	vpw = vpw < 1 ? 1 : (vpw > 64 ? 64 : vpw);
	if(ysplit < 1 || !ustage)
		ysplit = 1;
	int nwaves = (nlist + vpw - 1) / vpw;
	int nblocks = (nwaves + FAST_WPB - 1) / FAST_WPB;
	hipLaunchKernelGGL(k_leaf_oscpan, dim3(nblocks, ysplit), dim3(64 * FAST_WPB), 0, (hipStream_t)stream,
			dparams, dlist, nlist, vpw, ysplit, hp.voices, (const int *)hp.ustate,
			ysplit > 1 ? ustage : hp.ustate, hp.wavepool, hp.waves, hp.ptab, hp.busmem, hp.wavecoef);
	if(event_after_main)
		hipEventRecord((hipEvent_t)event_after_main, (hipStream_t)stream);
	if(ysplit > 1) {
		A2DCommit cm = { dlist, nlist, 1, ustage };
		if(defer)
			*defer = cm;
		else
			a2d_launch_commit(hp, cm, stream);
	} else if(defer)
		defer->nlist = 0;
	return (int)hipGetLastError();
}

int a2d_launch_leaf_osc2pan(const A2DParams *dparams, const A2DParams &hp, const int *dlist, int nlist,
		int vpw, int ysplit, int *ustage, void *stream, A2DCommit *defer)
{
	if(nlist <= 0)
		return 0;
	vpw = vpw < 1 ? 1 : (vpw > 64 ? 64 : vpw);
	if(ysplit < 1 || !ustage)
		ysplit = 1;
	int nwaves = (nlist + vpw - 1) / vpw;
	int nblocks = (nwaves + FAST_WPB - 1) / FAST_WPB;
	hipLaunchKernelGGL(k_leaf_osc2pan, dim3(nblocks, ysplit), dim3(64 * FAST_WPB), 0, (hipStream_t)stream,
			dparams, dlist, nlist, vpw, ysplit, hp.voices, (const int *)hp.ustate,
			ysplit > 1 ? ustage : hp.ustate, hp.wavepool, hp.waves, hp.ptab, hp.busmem, hp.wavecoef);
	if(ysplit > 1) {
		A2DCommit cm = { dlist, nlist, 2, ustage };
		if(defer)
			*defer = cm;
		else
			a2d_launch_commit(hp, cm, stream);
	} else if(defer)
		defer->nlist = 0;
	return (int)hipGetLastError();
}

__global__ void k_scatter_runs(const int *__restrict__ idx, const A2DRun *__restrict__ val, int n,
		A2DRun *__restrict__ runs)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n)
		runs[idx[i]] = val[i];
}

int a2d_launch_scatter_runs(const int *didx, const A2DRun *dval, int n, A2DRun *druns, void *stream)
{
	if(n <= 0)
		return 0;
	hipLaunchKernelGGL(k_scatter_runs, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
			didx, dval, n, druns);
	return (int)hipGetLastError();
}

int a2d_launch_leaf_oscfiltpan(const A2DParams *dparams, const A2DParams &hp, const int *dlist, int nlist,
		int vpg, void *stream)
{
	if(nlist <= 0)
		return 0;
	// voices per workgroup (one filter wavefront's lanes)
	vpg = vpg < 1 ? 1 : (vpg > FILT_MAXV ? FILT_MAXV : vpg);
	int nblocks = (nlist + vpg - 1) / vpg;
	size_t lds = (size_t)3 * vpg * FILT_PITCH * sizeof(int);
	hipLaunchKernelGGL(k_leaf_oscfiltpan, dim3(nblocks), dim3(64 * FILT_WAVES), lds, (hipStream_t)stream,
			dparams, dlist, nlist, vpg, hp.voices, hp.ustate, hp.wavepool, hp.waves, hp.ptab, hp.busmem, hp.wavecoef);
	return (int)hipGetLastError();
}

// voices per workgroup at which every oscillator wavefront of k_leaf_osc2filtpan stays in its all-settled loop:
// FILT_WAVES - 1 filter wavefront - the FILT_WAVES / 4 - 1 wavefronts that share its SIMD and take no voices
int a2d_osc2filtpan_max_vpg(void)
{
	const int nfull = FILT_WAVES - 1 - (FILT_WAVES / 4 - 1);
	const int v = nfull * (FILT2_LAUNCHV < FILT2_FASTV ? FILT2_LAUNCHV : FILT2_FASTV);
	return v > FILT_MAXV ? FILT_MAXV : v;
}

int a2d_launch_leaf_osc2filtpan(const A2DParams *dparams, const A2DParams &hp, const int *dlist, int nlist,
		int vpg, void *stream)
{
	if(nlist <= 0)
		return 0;
	vpg = vpg < 1 ? 1 : (vpg > FILT_MAXV ? FILT_MAXV : vpg);
	int nblocks = (nlist + vpg - 1) / vpg;
	size_t lds = (size_t)3 * vpg * FILT_PITCH * sizeof(int);
	hipLaunchKernelGGL(k_leaf_osc2filtpan, dim3(nblocks), dim3(64 * FILT_WAVES), lds, (hipStream_t)stream,
			dparams, dlist, nlist, vpg, hp.voices, hp.ustate, hp.wavepool, hp.waves, hp.ptab, hp.busmem, hp.wavecoef);
	return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// fmN -> panmix 1->2 leaf voices (fm.c + panmix.c)
// ---------------------------------------------------------------------------
// An FM voice is a recurrence in time (operator feedback), so here a lane is a
// voice for the oscillator part: every lane runs fm_process (fm.c:194-233) over
// the fragment with its operators in registers, the sine table in LDS, and
// writes its 64 samples into its row of a [voices][64+1] LDS tile.  The mix-down
// then runs frame = lane as in the other leaf kernels: row v * pan gains, summed
// over the voices of the wavefront in registers, one atomic per (fragment,
// channel, frame) into the bus.  One launch per unit kind present (the host
// groups the list by kind), so the operator count / oversampling / structure
// are compile-time constants of the body.
enum { PV_VOL = 0, PV_PAN = 4, PV_NWORDS = 8 };

// frames [off, off + len) of panmix 1->2 adding into the voice's output bus
// (panmix_Process12Add, panmix.c:78-125) for input x, frame = lane
DEV void pan_window_s(Ramp &vol, Ramp &pan, int x, int off, int len, int lane, int &acc0, int &acc1)
{
	const bool clamp = pan.target > 0xffffff || pan.target < -0xffffff ||
			pan.value > 0xffffff || pan.value < -0xffffff;
	ramp_prepare_s(vol, len);
	ramp_prepare_s(pan, len);
	const int k = lane - off;
	if(k >= 0 && k < len) {
		int vk = wadd(vol.value, wmul(vol.delta, k));
		int pk = wadd(pan.value, wmul(pan.delta, k));
		int vp = mul64s(pk, vk, 24);
		int v0 = wsub(vk, vp), v1 = wadd(vk, vp);
		if(clamp) {
			int lim = wshl(vk, 1);
			if(v0 > lim) v0 = lim;
			if(v1 > lim) v1 = lim;
		}
		acc0 = wadd(acc0, mul64s(x, v0, 24));
		acc1 = wadd(acc1, mul64s(x, v1, 24));
	}
	ramp_run(vol, len);
	ramp_run(pan, len);
}

// Unlike the wavetable leaf kernels this one also executes the voices' command
// records (control writes, sub-fragment windows, births and deaths): FM voices
// in real scripts are enveloped by the VM every few milliseconds, and sending
// each such voice-fragment through the general kernel (one lane per voice)
// would cost more than the rest of the batch.  The oscillator stage is frame
// synchronous - all lanes step through the fragment's frames together - and a
// lane whose window ended consumes its own records up to the next window
// (divergent, rare); the mix-down stage walks the same records for the panmix
// unit, voice by voice, on the scalar unit.
template<int NOPS, int OSBITS, int PAR>
DEV void fmpan_body(const A2DParams &p, const int *__restrict__ list, int first, int nv, int lane,
		int *tile, const uint32_t *sine, const A2DVoice *__restrict__ voices, int *ustate,
		int *fmstate, const uint32_t *__restrict__ ptab, int *__restrict__ busmem)
{
	const int nfrags = p.nfrags;
	const int dbg = p.debug;
	const A2DRec *__restrict__ recs = p.recs;
	int ffr[A2D_MAXBATCH / 64];
#pragma unroll
	for(int k = 0; k < A2D_MAXBATCH / 64; ++k)
		ffr[k] = (k * 64 + lane < nfrags) ? p.fragframes[k * 64 + lane] : 0;

	int pv[PV_NWORDS];
#pragma unroll
	for(int k = 0; k < PV_NWORDS; ++k)
		pv[k] = 0;
	FmOp op[NOPS];
#pragma unroll
	for(int i = 0; i < NOPS; ++i) {
		op[i].a = op[i].fb = op[i].p = Ramp{ 0, 0, 0, 0 };
		op[i].last_pitch = op[i].last = 0;
		op[i].phase = op[i].dphase = 0;
	}
	unsigned step[NOPS], fix[NOPS];
#pragma unroll
	for(int i = 0; i < NOPS; ++i)
		step[i] = fix[i] = 0;
	int slot = -1, u0 = 0, u1 = 0, my_off = -1, my_nch = 2, v0 = 0, v1 = 0, settled = 0;
	int run_first = 0, run_count = 0, fmslot = -1;
	int actB = 0;		// the voice is alive (vactive), as seen by the oscillator stage
	// A wavefront with fewer than 16 busy lanes runs a lane's serial chain only
	// about half as fast on MI355X (measured: fm4, 64 voices in 64 one-lane
	// wavefronts 34.0 ms per 16 384 frames, the same with 15 idle lanes put to
	// work on copies 14.5 ms; profiles/r01_fm_vpw_sweep.txt).  So lanes nv..15
	// shadow the wavefront's voices: same inputs, same arithmetic, no stores.
	const bool writer = lane < nv;
	if(lane < nv)
		slot = list[first + lane];
	else if(lane < 16 && !(p.debug & 4))
		slot = list[first + lane % nv];
	const bool mine = slot >= 0;
	const unsigned long long mine_mask = __ballot(mine);
	if(mine) {
		const A2DVoice &vc = voices[slot];
		u0 = vc.unit[0];
		u1 = vc.unit[1];
		my_off = vc.out_off;
		my_nch = vc.out_nch;
		const A2DRun run = p.runs[slot];
		run_first = run.first;
		run_count = run.count;
		actB = p.vactive[slot];
	}
	bool touched = mine && actB;	// the state in our registers is the voice's and goes back to memory
	if(touched) {
		fmslot = ustate[(size_t)u0 * A2D_USTATE + MW_SLOT];
		const int *fw = fmstate + (size_t)fmslot * A2D_FMSTATE;
#pragma unroll
		for(int i = 0; i < NOPS; ++i)
			fmop_load(op[i], fw + i * FO_WORDS);
		const int *w1 = ustate + (size_t)u1 * A2D_USTATE;
#pragma unroll
		for(int k = 0; k < 4; ++k) {
			pv[PV_VOL + k] = w1[PW_VOL + k];
			pv[PV_PAN + k] = w1[PW_PAN + k];
		}
		settled = !run_count && !(pv[PV_VOL + 3] | pv[PV_VOL + 2] | pv[PV_PAN + 3] | pv[PV_PAN + 2]) &&
				pv[PV_VOL] == pv[PV_VOL + 1] && pv[PV_PAN] == pv[PV_PAN + 1];
		if(settled) {
			const int vol = pv[PV_VOL], pan = pv[PV_PAN];
			const int vp = mul64s(pan, vol, 24);
			v0 = wsub(vol, vp);
			v1 = wadd(vol, vp);
			if(pan > 0xffffff || pan < -0xffffff) {
				int lim = wshl(vol, 1);
				if(v0 > lim) v0 = lim;
				if(v1 > lim) v1 = lim;
			}
		}
	}
	int curB = 0, curC = 0, actC = actB;	// record cursors / liveness of the two stages
	int *row = tile + lane * FILT_PITCH;
	// wave-uniform facts about the voices as bit masks: tests on the scalar unit
	const unsigned long long settled_mask = __ballot(settled != 0);
	const int prev_off = __shfl_up(my_off, 1);
	const unsigned long long newbus_mask = __ballot(mine && lane > 0 && my_off != prev_off);

	for(int f = 0; f < nfrags; ++f) {
		const int n = frames_of(ffr, f);
		// ---- oscillators, voice = lane, all lanes in step ----
		int wstart = A2D_FRAG, wend = 0;	// no window open
		bool pending = mine && curB < run_count && (int)A2D_RFRAG(recs[run_first + curB].head) == f;
		if(mine && !pending && actB) {
			// no records: the engine called Process(0, frames) (core.c:1875-1876)
			wstart = 0;
			wend = n;
			fm_prepare<NOPS, OSBITS>(op, ptab, n, step, fix);
		}
		const unsigned long long recs_mask = __ballot(pending);	// voices with records in this fragment
		if(!recs_mask) {
			// nobody in this wavefront has records in this fragment: plain loop
			if(wend) {
				if(nv >= 16) {		// no shadow lanes in this wavefront
					for(int s = 0; s < n; ++s)
						row[s] = fm_frame<NOPS, OSBITS, PAR>(op, step, fix, sine);
				} else {
					for(int s = 0; s < n; ++s) {
						const int y_ = fm_frame<NOPS, OSBITS, PAR>(op, step, fix, sine);
						if(writer)
							row[s] = y_;
					}
				}
			}
		} else
		for(int s = 0; s <= n; ++s) {
			if(pending && s >= wend) {
				// records up to and including the next window (at s == n:
				// whatever follows the last window of the fragment)
				for(;;) {
					if(curB >= run_count) {
						pending = false;
						break;
					}
					const A2DRec r = recs[run_first + curB];
					if((int)A2D_RFRAG(r.head) != f) {
						pending = false;
						break;
					}
					++curB;
					const int rop = (int)A2D_ROP(r.head), unit = (int)A2D_RUNIT(r.head);
					if(rop == R_SEG) {
						if(actB) {
							wstart = (int)(r.dur & 0xffffu);
							wend = wstart + (int)(r.dur >> 16);
							fm_prepare<NOPS, OSBITS>(op, ptab, (int)(r.dur >> 16), step, fix);
							break;
						}
					} else if(rop == R_INIT) {
						if(unit == 0) {	// value = transpose + basepitch, start = wake fraction, dur = pool slot
							fm_init_ops<NOPS>(op, ptab, r.value, r.start & 0xffu);
							fmslot = (int)r.dur;
							touched = true;
						}
						actB = 1;
					} else if(rop == R_WRITE) {
						if(unit == 0)
							fm_write_ops<NOPS>(op, (int)A2D_RREG(r.head), r.value, (int)r.start, (int)r.dur);
					} else if(rop == R_KILL)
						actB = 0;
				}
			}
			if(s >= wstart && s < wend) {
				const int y_ = fm_frame<NOPS, OSBITS, PAR>(op, step, fix, sine);
				if(writer)
					row[s] = y_;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// ---- pan + mix-down, frame = lane, voice by voice ----
		int acc0 = 0, acc1 = 0;
		int cur_off = rdl(my_off, 0), cur_nch = rdl(my_nch, 0);
		for(int v = 0; v < nv; ++v) {
			if(!((mine_mask >> v) & 1ull))
				continue;
			if((newbus_mask >> v) & 1ull) {
				if(cur_off >= 0 && !(dbg & 1)) {
					int *dst = busmem + cur_off + (size_t)f * cur_nch * A2D_FRAG;
					if(acc0) atomicAdd(&dst[lane], acc0);
					if(acc1) atomicAdd(&dst[A2D_FRAG + lane], acc1);
				}
				acc0 = acc1 = 0;
				cur_off = rdl(my_off, v);
				cur_nch = rdl(my_nch, v);
			}
			const int y = tile[v * FILT_PITCH + lane];
			const bool recs_now = (recs_mask >> v) & 1ull;
			if(!recs_now && ((settled_mask >> v) & 1ull)) {
				const int g0 = rdl(v0, v), g1 = rdl(v1, v);
				if(lane < n) {
					acc0 = wadd(acc0, mul64s(y, g0, 24));
					acc1 = wadd(acc1, mul64s(y, g1, 24));
				}
				continue;
			}
			const int rc = rdl(run_count, v), rf = rdl(run_first, v);
			int cc = rdl(curC, v);
			int act = rdl(actC, v);
			if(!recs_now && !act)
				continue;
			Ramp vol, pan;
			vol.value = rdl(pv[PV_VOL], v); vol.target = rdl(pv[PV_VOL + 1], v);
			vol.delta = rdl(pv[PV_VOL + 2], v); vol.timer = rdl(pv[PV_VOL + 3], v);
			pan.value = rdl(pv[PV_PAN], v); pan.target = rdl(pv[PV_PAN + 1], v);
			pan.delta = rdl(pv[PV_PAN + 2], v); pan.timer = rdl(pv[PV_PAN + 3], v);
			if(!recs_now)
				pan_window_s(vol, pan, y, 0, n, lane, acc0, acc1);
			else
				for(; cc < rc; ++cc) {
					const A2DRec r = recs[rf + cc];
					if((int)A2D_RFRAG(r.head) != f)
						break;
					const int rop = (int)A2D_ROP(r.head), unit = (int)A2D_RUNIT(r.head);
					if(rop == R_SEG) {
						if(act)
							pan_window_s(vol, pan, y, (int)(r.dur & 0xffffu), (int)(r.dur >> 16),
									lane, acc0, acc1);
					} else if(rop == R_INIT) {
						if(unit == 1) {	// panmix_Initialize, panmix.c:252-284
							ramp_init(vol, 65536);
							ramp_init(pan, 0);
						}
						act = 1;
					} else if(rop == R_WRITE) {
						if(unit == 1) {
							if(A2D_RREG(r.head))
								ramp_set(pan, r.value, (int)r.start, (int)r.dur);
							else
								ramp_set(vol, r.value, (int)r.start, (int)r.dur);
						}
					} else if(rop == R_KILL)
						act = 0;
				}
			const bool me = lane == v;
			WRL(pv[PV_VOL], vol.value); WRL(pv[PV_VOL + 1], vol.target);
			WRL(pv[PV_VOL + 2], vol.delta); WRL(pv[PV_VOL + 3], vol.timer);
			WRL(pv[PV_PAN], pan.value); WRL(pv[PV_PAN + 1], pan.target);
			WRL(pv[PV_PAN + 2], pan.delta); WRL(pv[PV_PAN + 3], pan.timer);
			WRL(curC, cc);
			WRL(actC, act);
		}
		if(cur_off >= 0 && !(dbg & 1)) {
			int *dst = busmem + cur_off + (size_t)f * cur_nch * A2D_FRAG;
			if(acc0) atomicAdd(&dst[lane], acc0);
			if(acc1) atomicAdd(&dst[A2D_FRAG + lane], acc1);
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}

	if(mine && writer)
		p.vactive[slot] = actB;
	if(touched && writer) {
		int *fw = fmstate + (size_t)fmslot * A2D_FMSTATE;
#pragma unroll
		for(int i = 0; i < NOPS; ++i)
			fmop_store(fw + i * FO_WORDS, op[i]);
		ustate[(size_t)u0 * A2D_USTATE + MW_SLOT] = fmslot;
		int *w1 = ustate + (size_t)u1 * A2D_USTATE;
#pragma unroll
		for(int k = 0; k < 4; ++k) {
			w1[PW_VOL + k] = pv[PV_VOL + k];
			w1[PW_PAN + k] = pv[PV_PAN + k];
		}
	}
}

// one instantiation per unit kind: each gets the registers it needs (fm1 a
// third of fm4's), and with them its own occupancy
template<int NOPS, int OSBITS, int PAR>
__global__ __launch_bounds__(64 * FAST_WPB)
void k_leaf_fmpan(const A2DParams *__restrict__ pp, const int *__restrict__ list, int nlist, int vpw,
		const A2DVoice *__restrict__ voices, int *ustate,
		int *fmstate, const uint32_t *__restrict__ fmsine, const uint32_t *__restrict__ ptab,
		int *__restrict__ busmem)
{
	extern __shared__ __attribute__((aligned(16))) int tiles[];
	__shared__ uint32_t sine[2048];		// static: a constant LDS address folds into ds_read's offset
	for(int i = threadIdx.x; i < 2048; i += 64 * FAST_WPB)
		sine[i] = fmsine[i];
	__syncthreads();
	const int wv = rfl((int)(threadIdx.x >> 6));	// (wave-uniform, and known to the compiler as such)
	const int lane = threadIdx.x & 63;
	const int first = (blockIdx.x * FAST_WPB + wv) * vpw;
	if(first >= nlist)
		return;
	fmpan_body<NOPS, OSBITS, PAR>(*pp, list, first, min(vpw, nlist - first), lane,
			tiles + wv * vpw * FILT_PITCH, sine, voices, ustate, fmstate, ptab, busmem);
}

template<int NOPS, int OSBITS, int PAR>
static int launch_fmpan(const A2DParams *dparams, const A2DParams &hp, const int *dlist, int nlist,
		int vpw, hipStream_t stream)
{
	const int nwaves = (nlist + vpw - 1) / vpw;
	const int nblocks = (nwaves + FAST_WPB - 1) / FAST_WPB;
	const size_t lds = (size_t)FAST_WPB * vpw * FILT_PITCH * sizeof(int);
	// (the attribute belongs to the function on ONE device: contexts of one
	// process may live on different GPUs)
	static bool attr_set[64];
	int dev = 0;
	(void)hipGetDevice(&dev);
	if(!attr_set[dev & 63]) {
		(void)hipFuncSetAttribute((const void *)k_leaf_fmpan<NOPS, OSBITS, PAR>,
				hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
		attr_set[dev & 63] = true;
	}
	hipLaunchKernelGGL((k_leaf_fmpan<NOPS, OSBITS, PAR>), dim3(nblocks), dim3(64 * FAST_WPB), lds, stream,
			dparams, dlist, nlist, vpw, hp.voices, hp.ustate, hp.fmstate, hp.fmsine, hp.ptab, hp.busmem);
	return (int)hipGetLastError();
}

// All kinds in one launch, for small voice counts: eight per-kind launches on
// one stream run one after the other although they are independent, and with a
// few dozen voices each of them is a single dependent chain (tens of
// microseconds per fragment).  Wavefronts look their kind up in the segment
// table; the register budget is that of the largest body.
struct FmSegs { int count[8]; };	// voices per kind, fm1 fm2 fm3 fm4 fm3p fm4p fm2r fm4r; list grouped in that order

__global__ __launch_bounds__(64 * FAST_WPB)
void k_leaf_fmpan_all(const A2DParams *__restrict__ pp, const int *__restrict__ list, FmSegs segs, int vpw,
		const A2DVoice *__restrict__ voices, int *ustate,
		int *fmstate, const uint32_t *__restrict__ fmsine, const uint32_t *__restrict__ ptab,
		int *__restrict__ busmem)
{
	extern __shared__ __attribute__((aligned(16))) int tiles[];
	__shared__ uint32_t sine[2048];
	for(int i = threadIdx.x; i < 2048; i += 64 * FAST_WPB)
		sine[i] = fmsine[i];
	__syncthreads();
	const int wv = rfl((int)(threadIdx.x >> 6));	// (wave-uniform, and known to the compiler as such)
	const int lane = threadIdx.x & 63;
	int gw = blockIdx.x * FAST_WPB + wv;	// wavefront index; which kind's range is it in?
	int kind = -1, first = 0, nv = 0, at = 0;
#pragma unroll
	for(int k = 0; k < 8; ++k) {
		const int nw = (segs.count[k] + vpw - 1) / vpw;
		if(kind < 0 && gw < nw) {
			kind = k;
			first = at + gw * vpw;
			nv = min(vpw, segs.count[k] - gw * vpw);
		}
		gw -= nw;
		at += segs.count[k];
	}
	if(kind < 0)
		return;
	int *tile = tiles + wv * vpw * FILT_PITCH;
#define FMPAN(N, OS, PAR) fmpan_body<N, OS, PAR>(*pp, list, first, nv, lane, tile, sine, voices, ustate, \
		fmstate, ptab, busmem)
	switch(rfl(kind)) {
	  case 0: FMPAN(1, 0, 0); break;
	  case 1: FMPAN(2, 1, 0); break;
	  case 2: FMPAN(3, 2, 0); break;
	  case 3: FMPAN(4, 2, 0); break;
	  case 4: FMPAN(3, 2, 1); break;
	  case 5: FMPAN(4, 2, 1); break;
	  case 6: FMPAN(2, 1, 2); break;
	  case 7: FMPAN(4, 2, 2); break;
	}
#undef FMPAN
}

int a2d_launch_leaf_fmpan_all(const A2DParams *dparams, const A2DParams &hp, const int *dlist,
		const int *count8, int vpw, void *stream)
{
	FmSegs segs;
	int nwaves = 0;
	vpw = min(max(vpw, 1), FILT_MAXV);
	for(int k = 0; k < 8; ++k) {
		segs.count[k] = count8[k];
		nwaves += (count8[k] + vpw - 1) / vpw;
	}
	if(!nwaves)
		return 0;
	const int nblocks = (nwaves + FAST_WPB - 1) / FAST_WPB;
	const size_t lds = (size_t)FAST_WPB * vpw * FILT_PITCH * sizeof(int);
	static bool attr_set[64];
	int dev = 0;
	(void)hipGetDevice(&dev);
	if(!attr_set[dev & 63]) {
		(void)hipFuncSetAttribute((const void *)k_leaf_fmpan_all, hipFuncAttributeMaxDynamicSharedMemorySize,
				150 * 1024);
		attr_set[dev & 63] = true;
	}
	hipLaunchKernelGGL(k_leaf_fmpan_all, dim3(nblocks), dim3(64 * FAST_WPB), lds, (hipStream_t)stream,
			dparams, dlist, segs, vpw, hp.voices, hp.ustate, hp.fmstate, hp.fmsine, hp.ptab, hp.busmem);
	return (int)hipGetLastError();
}

int a2d_launch_leaf_fmpan(const A2DParams *dparams, const A2DParams &hp, int kind, const int *dlist, int nlist,
		int vpw, void *stream)
{
	if(nlist <= 0)
		return 0;
	vpw = min(max(vpw, 1), FILT_MAXV);
	hipStream_t st = (hipStream_t)stream;
	switch(kind) {
	  case A2D_FM1: return launch_fmpan<1, 0, 0>(dparams, hp, dlist, nlist, vpw, st);
	  case A2D_FM2: return launch_fmpan<2, 1, 0>(dparams, hp, dlist, nlist, vpw, st);
	  case A2D_FM3: return launch_fmpan<3, 2, 0>(dparams, hp, dlist, nlist, vpw, st);
	  case A2D_FM4: return launch_fmpan<4, 2, 0>(dparams, hp, dlist, nlist, vpw, st);
	  case A2D_FM3P: return launch_fmpan<3, 2, 1>(dparams, hp, dlist, nlist, vpw, st);
	  case A2D_FM4P: return launch_fmpan<4, 2, 1>(dparams, hp, dlist, nlist, vpw, st);
	  case A2D_FM2R: return launch_fmpan<2, 1, 2>(dparams, hp, dlist, nlist, vpw, st);
	  case A2D_FM4R: return launch_fmpan<4, 2, 2>(dparams, hp, dlist, nlist, vpw, st);
	}
	return -1;
}

// root-bus partials -> staging buffer, leaving the bus zeroed (a2amd_rootbus_copy)
__global__ void k_park(int32_t *__restrict__ stage, int32_t *__restrict__ bus, unsigned words)
{
	for(unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) {
		stage[i] = bus[i];
		bus[i] = 0;
	}
}

// dst += src (wrap-around), src cleared: the same-device stand-in for the RCCL reduce
// of a2amd_render_group (several contexts of one process on ONE GPU)
__global__ void k_add_bus(int32_t *__restrict__ dst, int32_t *__restrict__ src, unsigned words)
{
	for(unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) {
		dst[i] = (int32_t)((uint32_t)dst[i] + (uint32_t)src[i]);
		src[i] = 0;
	}
}

__global__ void k_add_inject(const int32_t *__restrict__ inj, int32_t *__restrict__ bus, int nch, int n)
{
	const int f = blockIdx.x, lane = threadIdx.x & 63, ch = threadIdx.x >> 6;
	if(ch < n)
		bus[((size_t)f * nch + ch) * A2D_FRAG + lane] = wadd(bus[((size_t)f * nch + ch) * A2D_FRAG + lane],
				inj[((size_t)f * 8 + ch) * A2D_FRAG + lane]);
}

int a2d_launch_add_inject(const int32_t *inj, int32_t *bus, int nch, int n, int nfrags, void *stream)
{
	if(nfrags <= 0 || n <= 0)
		return 0;
	hipLaunchKernelGGL(k_add_inject, dim3(nfrags), dim3(64 * 8), 0, (hipStream_t)stream, inj, bus, nch, n);
	return (int)hipGetLastError();
}

int a2d_launch_add_bus(int32_t *dst, int32_t *src, unsigned words, void *stream)
{
	if(!words)
		return 0;
	hipLaunchKernelGGL(k_add_bus, dim3((words + 255) / 256 < 1024 ? (words + 255) / 256 : 1024), dim3(256), 0,
			(hipStream_t)stream, dst, src, words);
	return (int)hipGetLastError();
}

int a2d_launch_park(int32_t *stage, int32_t *bus, unsigned words, void *stream)
{
	if(!words)
		return 0;
	hipLaunchKernelGGL(k_park, dim3((words + 255) / 256 < 1024 ? (words + 255) / 256 : 1024), dim3(256), 0,
			(hipStream_t)stream, stage, bus, words);
	return (int)hipGetLastError();
}

int a2d_launch_commit(const A2DParams &hp, const A2DCommit &cm, void *stream)
{
	if(cm.nlist <= 0)
		return 0;
	hipLaunchKernelGGL(k_commit_oscpan, dim3((cm.nlist * (cm.nosc + 1) * 16 + 255) / 256), dim3(256), 0,
			(hipStream_t)stream, cm, hp.voices, hp.runs, hp.ustate);
	return (int)hipGetLastError();
}

int a2d_launch_bus_driver(const A2DParams *dparams, const int *dlist, int nlist, int nfrags, int consume,
		const A2DCommitSet *commits, void *stream, int *master_host)
{
	if(nlist <= 0)
		return 0;
	A2DCommitSet cs;
	cs.n = 0;
	int extra = 0;
	if(commits)
		for(int k = 0; k < commits->n; ++k)
			if(commits->c[k].nlist > 0) {
				cs.c[cs.n++] = commits->c[k];
				extra += (commits->c[k].nlist * (commits->c[k].nosc + 1) * 16 + 255) / 256;
			}
	// grid.y: workgroups per voice, 4 fragments in flight each
	hipLaunchKernelGGL(k_bus_driver, dim3(nlist + extra, nfrags >= 16 ? 16 : (nfrags + 3) / 4), dim3(256), 0,
			(hipStream_t)stream, dparams, dlist, nlist, consume, cs, master_host);
	return (int)hipGetLastError();
}
