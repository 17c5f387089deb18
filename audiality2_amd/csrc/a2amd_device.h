// a2amd_device.h - data layout shared by the host recorder and the HIP kernels.
//
// Everything the kernels touch lives in a handful of flat device arrays; the
// host owns the *structure* (which units form a voice, where a voice's output
// lands), the device owns the *signal state* (phases, rampers, filter and
// delay memories) and only ever changes it by executing command records.
#pragma once
#include <stdint.h>

#define A2D_MAXCHAIN   16     // units per voice (A2AMD_MAXCHAIN).  Round 6: 8 -> 16, what a record's 4-bit chain position
			      // (A2D_RUNIT) addresses; the reference's unit list has no cap (core.c:163-300)
#define A2D_CHAIN_INLINE 8    // ... of which the voice record (A2DVoice: one 64-byte line that every kernel reads per voice) holds
			      // the first 8; units 8 - 15 of a longer chain stand in a side array only the general kernel reads
			      // (A2DVoiceExt).  (First cut: unit[16] in the record itself, 96 bytes - k_leaf_oscfiltpan 0.5539 against
			      // 0.5505 ms, k_leaf_osc2pan 2.198 against 2.186 ms, same box, interleaved x 3: profiles/r06_maxchain_ab.txt)
#define A2D_MAXBATCH 256
#define A2D_USTATE    24      // int32 words of device state per unit
#define A2D_MAXCH      8
#define A2D_FRAG      64
#define A2D_MIPS      10
#define A2D_FBD_BUFSIZE 131072          // fbdelay.c:27
#define A2D_MAXPHINC  512               // a2_waves.h:57
#ifndef A2D_COEF_WORDS
#define A2D_COEF_WORDS 3                 // words per entry of the Hermite coefficient table (a2amd_fast.hip)
#endif
#define A2D_WTOSC_MAXLENGTH (0x01000000 - 1 - 131)   // wtosc.c:55

// unit kinds: numerically equal to a2amd_unitkind
enum { A2D_WTOSC = 0, A2D_PANMIX, A2D_FILTER12, A2D_FBDELAY, A2D_INLINE, A2D_XINSERT,
	A2D_FM1, A2D_FM2, A2D_FM3, A2D_FM4, A2D_FM3P, A2D_FM4P, A2D_FM2R, A2D_FM4R,
	A2D_DC, A2D_WAVESHAPER, A2D_DCBLOCK, A2D_LIMITER, A2D_XSINK, A2D_XSOURCE };
#define A2D_IS_FM(k) ((k) >= A2D_FM1 && (k) <= A2D_FM4R)

// wtosc Process variants (the reference swaps u->Process, wtosc.c:433-483)
enum { A2D_OSC_OFF = 0, A2D_OSC_NOISE, A2D_OSC_WAVE, A2D_OSC_MIPWAVE };

// packed static description of a unit instance (host owned)
//   bits 4 add (A2_PROCADD), 8-11 ninputs, 12-15 noutputs, 16 wired out, 24-31 kind
#define A2D_DESC(kind, add, nin, nout, wired) \
	(((uint32_t)(kind) << 24) | ((uint32_t)(add) << 4) | ((uint32_t)(nin) << 8) | \
	 ((uint32_t)(nout) << 12) | ((uint32_t)(wired) << 16))
#define A2D_KIND(d)  ((d) >> 24)
#define A2D_ADD(d)   (((d) >> 4) & 1u)
#define A2D_NIN(d)   (((d) >> 8) & 15u)
#define A2D_NOUT(d)  (((d) >> 12) & 15u)
#define A2D_WIRED(d) (((d) >> 16) & 1u)

// device unit state word indices ------------------------------------------
// ramper = 4 consecutive words {value, target, delta, timer} (a2_dsp.h:105)
enum {	// wtosc (A2_wtosc, wtosc.c:66-80)
	OW_MODE = 0, OW_WAVE, OW_DPHASE, OW_PHASE_LO, OW_PHASE_HI, OW_NOISE,
	OW_PRAMPING, OW_P = 7, OW_A = 11, OW_SEED = 15 };
enum {	// panmix (A2_panmix, panmix.c:35-40)
	PW_VOL = 0, PW_PAN = 4 };
enum {	// filter12 (A2_filter12, filter12.c:36-56); the cutoff ramper and the
	// float coefficient maths stay on the host, which ships f1 values
	FW_Q = 0, FW_LP = 4, FW_BP, FW_HP, FW_F1, FW_D1A, FW_D1B, FW_D2A, FW_D2B,
	FW_F1NEXT, FW_RAMP };
enum {	// fbdelay (A2_fbdelay, fbdelay.c:41-60)
	DW_FBDELAY = 0, DW_LDELAY, DW_RDELAY, DW_DRYGAIN, DW_FBGAIN, DW_LGAIN,
	DW_RGAIN, DW_BUFPOS, DW_BUFIDX };

enum {	// dc (A2_dc, dc.c:42-47)
	CW_VALUE = 0, CW_MODE = 4 };
enum {	// waveshaper (A2_waveshaper, waveshaper.c:44-48)
	SW_AMOUNT = 0 };
enum {	// dcblock (A2_dcblock, dcblock.c:33-49); the host ships f1
	BW_F1 = 0, BW_D1A, BW_D1B, BW_D2A, BW_D2B };
enum {	// limiter (A2_limiter, limiter.c:35-42); release / threshold computed by the host
	LW_RELEASE = 0, LW_THRESHOLD, LW_PEAK };
enum {	// fm (A2_fm, fm.c:95-105): the operators live in a pool of their own
	// (fmstate[slot][A2D_FMSTATE]); the unit's state words only name the slot
	MW_SLOT = 0 };
enum {	// one operator (A2_fmosc, fm.c:84-93), A2D_FMSTATE / 4 words
	FO_A = 0, FO_FB = 4, FO_P = 8, FO_LASTPITCH = 12, FO_PHASE, FO_DPHASE, FO_LAST, FO_WORDS };
#define A2D_FMSTATE 64

// A2DWave::flags, beside the engine's own (A2_LOOPED 0x100 ...): the settled wtosc -> panmix kernel reads this
// wave's samples themselves instead of its Hermite coefficient entries (a2amd_wave_upload decides)
#define A2D_WF_RAWTAPS 0x40000000u

// mirror of A2_wave for the device: offsets index the int16 wave pool and point
// at the first PAYLOAD sample of a level (i.e. data[level] + A2_WAVEPRE)
struct A2DWave {
	int32_t  type;
	uint32_t flags;
	uint32_t period;
	uint32_t size[A2D_MIPS];
	uint32_t off[A2D_MIPS];
	uint32_t pad;
};

// one voice = one unit chain (host owned)
struct A2DVoice {
	int32_t nunits;
	int32_t unit[A2D_CHAIN_INLINE];	// indices into udesc[] / ustate[]; the chain's further units: A2DVoiceExt
	int32_t out_off, out_nch;	// bus the wired outputs add into (int32 offset into busmem)
	int32_t own_off, own_nch;	// bus our 'inline' unit collects children in, or -1
	int32_t pad[3];
};

static_assert(sizeof(A2DVoice) == 64, "the voice record is one 64-byte line");
struct A2DVoiceExt { int32_t unit[A2D_MAXCHAIN - A2D_CHAIN_INLINE]; };	// [voice slot], where a context has a chain > 8 units

// command record, 16 bytes
struct A2DRec {
	uint32_t head;		// frag:16 | op:8 | unit:4 | reg:4
	int32_t  value;
	uint32_t dur;		// SEG: offset | frames << 16
	uint32_t start;
};
enum { R_SEG = 1, R_WRITE, R_INIT, R_KILL, R_F1SET, R_F1RAMP, R_NOISESEED, R_NOP };
#define A2D_HEAD(frag, op, unit, reg) \
	((uint32_t)(frag) | ((uint32_t)(op) << 16) | ((uint32_t)(unit) << 24) | ((uint32_t)(reg) << 28))
#define A2D_RFRAG(h) ((h) & 0xffffu)
#define A2D_ROP(h)   (((h) >> 16) & 0xffu)
#define A2D_RUNIT(h) (((h) >> 24) & 15u)
#define A2D_RREG(h)  (((h) >> 28) & 15u)

struct A2DRun { int32_t first, count; };

// ---- the scripted voice's VM on the device (include/a2amd_vm.h) ----------------
// One per adopted voice, device resident (the device is the authority on everything in it
// between adoption and recall): A2_vmstate (include/a2_vm.h:62-69), where a2_VoiceControl
// (core.c:143-149) sends each VM register, and the cutoff rampers of the voice's filter12 units,
// which for host-driven voices never leave the host (a2amd_host.h: HUnit::cutoff).
#define A2D_VM_NOWRITE 0xffu
#define A2D_VM_TRAPWRITE 0xfeu		// cmap: wired to something the device VM cannot write - the voice leaves before it would
#define A2D_VM_MAXCUT  2
#define A2D_VM_MAXENV  2
#define A2D_VM_MAXPOS  8		// chain positions the device VM's register map can name (cmap, A2DVmEnv::target: a nibble, of which 14 and
				// 15 mean something else): a voice with a longer chain stays the engine thread's (a2amd_vm_adopt)
#define A2D_VM_ENVPOS  14		// cmap chain position that stands for "the target register of env unit <register nibble>"
#define A2D_ENV_LUTSHIFT 6		// A2ENV_LUTSHIFT, env.c:27
#define A2D_ENV_LUTSIZE  (1 << A2D_ENV_LUTSHIFT)
#define A2D_ENV_LUTS     8		// A2ENVLUT_SPLINE, EXP1 .. EXP7 (env.c:36-47)
// src/units/env.c's A2_env (:88-99) for a voice the device VM runs: a control-rate unit without audio
// ports whose output is a register write on another unit through a control wire (env.c:135)
struct A2DVmEnv {
	int32_t ramper[4];		// A2_env.ramper
	int32_t lut;			// which table (A2ENVLUT_*)
	int32_t scale, offset, out;
	int32_t active;			// its Process is env_ProcessLUT (else env_ProcessOff)
	uint8_t regbase;		// VM register of its 'target' (then mode, down, time: env.c:50-56, registers[] = &vms->r[regbase])
	uint8_t k;			// units of the (backend) chain in front of it: a write to one of them takes effect behind the window
	uint8_t target;			// where its control output is wired: chain position << 4 | register, A2D_VM_NOWRITE = nowhere
	uint8_t pad;
};
struct A2DVmVoice {
	uint32_t waketime;		// A2_vmstate.waketime: 24:8 frames, engine time
	uint32_t code, ncode;		// the function's text in the code pool (32 bit words)
	uint16_t pc;
	uint8_t  state, ncut;
	int32_t  voice;			// backend voice slot (runs[] index)
	int32_t  fault;			// a trap the analysis should have ruled out (0 = none): the VM stopped there
	uint8_t  cmap[64];		// VM register -> chain position << 4 | unit register, A2D_VM_NOWRITE = none
	uint8_t  kind[A2D_MAXCHAIN];	// unit kinds of the chain
	uint8_t  cutpos[A2D_VM_MAXCUT];	// chain positions of the filter12 units whose cutoff the VM may write
	uint8_t  pad[2];
	int32_t  cut[A2D_VM_MAXCUT][4];	// ... and their cutoff rampers (A2_filter12.cutoff, filter12.c:38)
	A2DVmEnv env[A2D_VM_MAXENV];	// the voice's env units, nenv of them
	int32_t  nenv;
	uint32_t exit_when;		// has_exit: the VM run that starts at this wake time is the engine's (a2amd_vm_exit_time):
	int32_t  has_exit;		// the device VM stops in front of it (the walk has recalled the voice by then)
	int32_t  r[64];			// A2_vmstate.r
};

// what the VM kernel needs beside the voices (host copy: a2amd_ctx)
struct A2DVmParams {
	A2DVmVoice     *vmv;		// [vm slot]
	const int      *list;		// vm slots the kernel runs this batch
	int32_t         n;
	const uint32_t *code;		// code pool
	const uint32_t *ptab;
	const int32_t  *f1tab;		// [32][65536]: f12_pitch2coeff by (shift, fraction), or null
	const uint16_t *envlut;		// [A2D_ENV_LUTS][A2D_ENV_LUTSIZE + 2] (env.c:218-257), or null
	A2DRun         *runs;		// [voice slot]
	A2DRun         *vmrun;		// [list index]: where each voice's records go (count pass -> emit pass)
	A2DRec         *recs;		// the batch's record array (A2DParams::recs)
	uint32_t        rec_base, rec_cap;	// the VM's region of it, in records
	uint32_t       *total;		// records the count pass found; [1] = voices that faulted
	uint32_t        now;		// engine time of the batch's first frame
	uint32_t        msdur;		// A2_state.msdur
	int32_t         samplerate, basepitch;
	int32_t         nfrags;
	uint8_t         fragframes[A2D_MAXBATCH];
	uint8_t         fragbase[A2D_MAXBATCH];	// offset of each inside the engine's own fragment (core.c:1968)
};

// xinsert state words: client slot + 1 (0 = no clients) and A2AMD_XIO_* mode bits
enum { XW_SLOT = 0, XW_MODE = 1 };
#define A2D_XIO_HALF ((size_t)A2D_MAXBATCH * 8 * A2D_FRAG)	// words per direction of a slot
#define A2D_XIO_SLOT (2 * A2D_XIO_HALF)

#ifndef A2D_FAST_FCH
#define A2D_FAST_FCH 8	// k_leaf_oscpan: fragments per chunk (the host sizes the time slices by it)
#endif
#ifndef A2D_OSC2_FCH
#define A2D_OSC2_FCH 4	// k_leaf_osc2pan: fragments per chunk
#endif
#define A2D_MAXVPW   32       // voices one wavefront may walk per fragment

struct A2DParams {
	const A2DVoice *voices;
	const A2DVoiceExt *vext;	// units 8 - 15 of the chains that have them, or null (no such chain in the context)
	const uint32_t *udesc;
	int32_t        *ustate;		// [unit][A2D_USTATE]
	int32_t        *vactive;	// [voice slot]
	const A2DRun   *runs;		// [voice slot], this batch
	const A2DRec   *recs;
	const A2DWave  *waves;
	const int16_t  *wavepool;
	const int32_t  *wavecoef;	// [wave pool index][A2D_COEF_WORDS]: Hermite a, b, c:d0 of the window at that sample
	int32_t        *busmem;
	int32_t        *fbdmem;		// [bufidx][2][A2D_FBD_BUFSIZE]
	const uint32_t *ptab;		// 64 x {base, coeff}, pitch.c:70-96
	int32_t        *fmstate;	// [fm slot][A2D_FMSTATE]
	const uint32_t *fmsine;		// 2048 x {s[i], s[i+1]-s[i]} packed 16:16, fm.c:493-501
	int32_t *xio;			// xinsert client slots: [slot]{ tap[batch][ch][64], inject[batch][ch][64] }
	int32_t         nfrags;
	int32_t         samplerate;
	int32_t         debug;		// A2AMD_DEBUG ablation bits (perf experiments only)
	uint8_t         fragframes[A2D_MAXBATCH];
	uint16_t        fragstart[A2D_MAXBATCH];	// frames before each fragment
};

// a2amd_vm.hip: the count pass (emit = 0) or the emit pass of the VM kernel
int a2d_launch_vm(const A2DVmParams &vp, int emit, void *stream);
// k_vm_pool: the window pool entries k_vm_win would take for the batch described by vp (vp.list: VM slots, any
// classes) - one per window that begins inside a fragment - added to *out
int a2d_launch_vm_pool(const A2DVmParams &vp, unsigned *out, void *stream);
#define A2D_WIN_STAGED 2	/* further windows of a fragment a control lane keeps in LDS (WIN_EXL, a2amd_winctl.h) */
// Where k_vm_win leaves the state it has stepped - the voices' VM state, their units' control words, which of them are
// the quiet kernels' this batch (runs[]), the fault count.  Normally the live arrays; a SPECULATIVE pass (round 6,
// vm_speculate: the next batch's VM + control work done behind this batch, while the engine thread walks) leaves it in
// shadow arrays of the same shape, and k_vm_commit moves it over when the batch turns out to be the one predicted.
struct A2DVmwOut {
	A2DVmVoice *vmv;	// [vm slot]
	int        *ustate;	// [unit][A2D_USTATE]
	int        *vactive;	// [voice slot]
	A2DRun     *runs;	// [voice slot]
	uint32_t   *total;	// [1] += voices that faulted
	uint32_t   *idle;	// += voices of the list the VM leaves alone this batch (the quiet kernels'), or null
};
// a speculative pass's results into the live state: the voices of window class (nosc, filt) in vp.list (vm slots)
int a2d_launch_vm_commit(const A2DVmParams &vp, const A2DParams &hp, int nosc, int filt, const A2DVmwOut &from, void *stream);
// k_vm_win (a2amd_vmwin.hip): the VM voices of one window class (vp.list: their VM slots in the order of the
// class's voice list, vp.n) run through fragments [fa, fb) and write the window entries themselves - no records.
// now_fa: engine time of fragment fa's first frame, batch_end: of the batch's end; runs[voice].count says
// afterwards whose the voice is this batch (0: the quiet kernels').  Returns -1 for a class without a kernel.
int a2d_launch_vm_win(const A2DVmParams &vp, const A2DParams &hp, int nosc, int filt, int fa, int fb, uint32_t now_fa,
		uint32_t batch_end, int *wslot, int *wext, int *wscr, unsigned *widx, unsigned *wtop, unsigned wcap, void *stream,
		const A2DVmwOut *out = nullptr);	// (out: null = the live arrays)
#define A2D_VMW_ROW (64 - 1 - 1)	/* entries of wscr per voice of the list (A2D_WIN_WORDS each): a fragment's windows beyond
				 * the first and the staged ones - one at least (WIN_EXLN, a2amd_winctl.h) */

// launchers implemented in a2amd_kernels.hip (stream = hipStream_t)
// dparams / dlist are device pointers; 'vpw' voices of the list per wavefront
int a2d_launch_voices(const A2DParams *dparams, const int *dlist, int nlist, int vpw, void *stream);
// hp = host copy of *dparams (device pointers passed as direct kernel arguments)
// ysplit > 1 cuts the batch into time slices rendered by different wavefronts
// (needs the staging copy 'ustage' of the unit state array)
// a state commit a time-sliced leaf kernel left to be done (ustage -> ustate for its
// voices): rides along with the next driver-chain launch instead of one of its own
struct A2DCommit { const int *list; int nlist, nosc; const int *ustage; };
struct A2DCommitSet { A2DCommit c[2]; int n; };
int a2d_launch_leaf_oscpan(const A2DParams *dparams, const A2DParams &hp, const int *dlist, int nlist,
		int vpw, int ysplit, int *ustage, void *stream, void *event_after_main, A2DCommit *defer);
int a2d_launch_bus_driver(const A2DParams *dparams, const int *dlist, int nlist, int nfrags, int consume,
		const A2DCommitSet *commits, void *stream, int *master_host = nullptr);	// master_host: where the root stores the master bus (pinned host memory) instead of the bus memory
int a2d_launch_commit(const A2DParams &hp, const A2DCommit &cm, void *stream);
// quiet "inline; fbdelay 2->2 ... ; fbdelay 2->2 >" voices (one workgroup each)
int a2d_launch_bus_fbdchain(const A2DParams *dparams, const int *dlist, int nlist, int consume, void *stream);
int a2d_launch_park(int32_t *stage, int32_t *bus, unsigned words, void *stream);
int a2d_launch_add_bus(int32_t *dst, int32_t *src, unsigned words, void *stream);
// bus[f][ch][64] (nch channels per fragment) += inj[f][ch][64] (8 channels per fragment), ch < n
int a2d_launch_add_inject(const int32_t *inj, int32_t *bus, int nch, int n, int nfrags, void *stream);
// Hermite coefficient entries for wave pool samples [lo, hi) (reads pool[lo-1 .. hi+1])
int a2d_launch_build_coef(const int16_t *pool, int *coef, unsigned lo, unsigned hi, void *stream);
// a2amd_wavecap.hip
int a2d_launch_capture(const int32_t *bus, int32_t *dst, const uint32_t *fragpos, int nfrags, int nch, void *stream);
int a2d_launch_wave_from_pcm(const int32_t *pcm, int16_t *pool, const uint32_t *off, const uint32_t *size, int levels, int looped,
		int pre, int post, void *stream);
int a2d_osc2filtpan_max_vpg(void);
int a2d_launch_leaf_osc2filtpan(const A2DParams *dparams, const A2DParams &hp, const int *dlist, int nlist,
		int vpg, void *stream);
int a2d_launch_leaf_oscfiltpan(const A2DParams *dparams, const A2DParams &hp, const int *dlist, int nlist,
		int vpw, void *stream);
// runs[idx[i]] = val[i] for the few voices whose record run changed this batch
int a2d_launch_scatter_runs(const int *didx, const A2DRun *dval, int n, A2DRun *druns, void *stream);
int a2d_launch_leaf_osc2pan(const A2DParams *dparams, const A2DParams &hp, const int *dlist, int nlist,
		int vpw, int ysplit, int *ustage, void *stream, A2DCommit *defer);
// wtosc[+wtosc] [-> filter12] -> panmix voices that carry records this batch (nosc = 1 | 2, filt = 0 | 1)
// skip_empty: the list's voices get their records from the device VM (a2amd_vm.hip); those it left
// without any this batch are rendered by their class's quiet kernel
int a2d_launch_leaf_recs(const A2DParams *dparams, const A2DParams &hp, int nosc, int filt, const int *dlist,
		int nlist, int vpw, void *stream, int skip_empty = 0);
// Round 5 (a2amd_win.hip): the same voices in two passes.  The control pass (lane = voice) walks the
// records of fragments [fa, fb) and leaves closed-form window entries: the first window of fragment f for
// list position i in its slot wslot[((f - fa) * nlist + i) * A2D_WIN_SLOTWORDS(nosc, filt)], further windows
// of that fragment in the pool wext (A2D_WIN_WORDS apart) from index widx[(f - fa) * nlist + i] on (pool
// indices from the counter wtop[0], at most wcap; wtop[1] != 0: the pool was too small); wrc[i]: where a
// voice's walk through its records stands between two slabs of a batch.  The render pass (lane = frame)
// evaluates them.
#define A2D_WIN_WORDS 24
#define A2D_WIN_SLOTWORDS(nosc, filt) ((filt) ? ((nosc) == 1 ? 20 : 24) : ((nosc) == 1 ? 12 : 20))
int a2d_launch_win_ctl(const A2DParams *dparams, const A2DParams &hp, int nosc, int filt, const int *dlist, int nlist,
		int skip_empty, int fa, int fb, int *wslot, int *wext, unsigned *widx, unsigned *wtop, unsigned wcap, int *wrc,
		void *stream);
int a2d_launch_win_render(const A2DParams &hp, int nosc, int filt, const int *dlist, int nlist, int fa, int fb,
		const int *wslot, const int *wext, const unsigned *widx, void *stream);
// ... all four kinds in one launch: lists[k] / counts[k] for (nosc, filt) = (1,0) (2,0) (1,1) (2,1)
// skip_mask bit k: list k may hold voices without records this batch - their quiet kernel renders those (skip_empty)
int a2d_launch_leaf_recs_all(const A2DParams *dparams, const A2DParams &hp, const int *const *lists, const int *counts,
		int vpw, void *stream, int skip_mask = 0);
// fm -> panmix leaf voices of ONE unit kind (a2amd_unitkind A2AMD_FM1..FM4R)
int a2d_launch_leaf_fmpan(const A2DParams *dparams, const A2DParams &hp, int kind, const int *dlist, int nlist,
		int vpw, void *stream);
// ... or of all eight kinds in one launch (small voice counts); the list is
// grouped by kind in a2amd_unitkind order, count8[k] voices each
int a2d_launch_leaf_fmpan_all(const A2DParams *dparams, const A2DParams &hp, const int *dlist,
		const int *count8, int vpw, void *stream);
