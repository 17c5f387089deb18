// a2amd_host.h - what the translation units of liba2amd.so's host half share: the recorded
// scene (units, voices, waves, client slots), the context, the host copies of the few
// reference formulas the host evaluates, and the functions that cross file boundaries.
//   a2amd_host.cpp    the C ABI: state, waves, fragment clock, unit callbacks (the recorder)
//   a2amd_sched.cpp   what a batch becomes on the device: upload, launch order, graphs
//   a2amd_render.cpp  a2amd_render / collect / replay, statistics
//   a2amd_dist.cpp    multi-GPU: RCCL binding, the root-bus reduce, render groups
#ifndef A2AMD_HOST_H
#define A2AMD_HOST_H
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>		// (types only: the library is bound at run time, a2amd_dist_init)
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <vector>
#include <future>

#include "../../include/a2amd.h"
#include "../../include/a2amd_vm.h"
#include "a2amd_device.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace a2h {
struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;

};
extern Rccl g_rccl;
bool rccl_bind();

extern thread_local char g_err[256];


// ---- host copies of the few reference formulas the host must evaluate ----
struct Ramp { int value, target, delta, timer; };

inline int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
inline int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
inline int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// a2_InitRamper / a2_PrepareRamper / a2_RunRamper / a2_SetRamper, a2_dsp.h:121-170
inline void ramp_init(Ramp &r, int v) { r.value = r.target = (int)((unsigned)v << 8); r.delta = r.timer = 0; }
inline void ramp_prepare(Ramp &r, int frames)
{
	if(!r.timer) {
		r.value = r.target;
		r.delta = 0;
	} else if(frames <= (r.timer >> 8)) {
		r.delta = (int)((((int64_t)wsub(r.target, r.value)) * 256) / r.timer);
		r.timer = wsub(r.timer, frames << 8);
	} else {
		r.delta = wsub(r.target, r.value) / frames;
		r.timer = 0;
	}
}
inline void ramp_run(Ramp &r, int frames) { r.value = wadd(r.value, wmul(r.delta, frames)); }
inline void ramp_set(Ramp &r, int target, int start, int duration)
{
	r.target = (int)((unsigned)target << 8);
	r.timer = wadd(duration, start);
	if(r.timer < 256)
		r.value = r.target;
	else
		r.value = wadd(r.value, wmul(r.delta, start) >> 8);
}

// a2_pitch_open, pitch.c:70-96
inline void build_pitch_table(uint32_t *tab)
{
	unsigned b = 0x80000000u;
	for(unsigned i = 0; i < 64; ++i) {
		unsigned b2 = (unsigned)((double)0x80000000u * powf(2.0f, (i + 1) * (1.0f / 64)) + 0.5f);
		tab[2 * i] = b;
		tab[2 * i + 1] = (b2 - b + 128) >> 8;
		b = b2;
	}
}

// a2_P2I, pitch.c:57-67
inline unsigned p2i(const uint32_t *tab, int pitch)
{
	int n = pitch & 0xffff, oct = pitch >> 16;
	unsigned dph = tab[2 * (n >> 10) + 1] * (unsigned)(n & 0x3ff);
	dph >>= 2;
	dph += tab[2 * (n >> 10)];
	return dph >> ((unsigned)(7 - oct) & 31u);
}

// f12_pitch2coeff, filter12.c:65-72 -- float/double libm maths: host only
inline int f12_coeff(const uint32_t *tab, int cutoff_value, int samplerate)
{
	float f = p2i(tab, cutoff_value >> 8) * (261.626f / 16777216.0f);
	if(f > (samplerate >> 2))
		return 362 << 16;
	return (int)(512.0f * 65536.0f * sin(M_PI * f / samplerate));
}

// dcb_pitch2coeff, dcblock.c:57-64: the same maths from a 16:16 pitch
inline int dcb_coeff(const uint32_t *tab, int cutoff, int samplerate)
{
	float f = p2i(tab, cutoff) * (261.626f / 16777216.0f);
	if(f > (samplerate >> 2))
		return 362 << 16;
	return (int)(512.0f * 65536.0f * sin(M_PI * f / samplerate));
}

#define HIPCHK(c, call) do { hipError_t e_ = (call); if(e_ != hipSuccess) \
	return (c)->fail(A2AMD_EHIP, "%s: %s", #call, hipGetErrorString(e_)); } while(0)

template<class T> struct DevBuf {
	T *d = nullptr;
	size_t cap = 0;
};

struct HUnit {
	// --- what a Process call of a wtosc / any unit reads and writes: one cache line ---
	bool live = false;
	int kind = 0;
	int voice = -1, chainpos = 0;
	// wtosc: what the launch classes and the drop-in need to know of A2_wtosc at all
	// times - which Process variant is installed (wtosc.c:433-483) ...
	int mode = A2D_OSC_OFF, wave = -1;
	// ... and, ONLY while that is wtosc_Noise, enough of the rest to count the draws a
	// window takes from the engine's one RNG (wtosc.c:129-152): phase, increment, pitch
	// ramper.  In every other mode these fields are stale - the device's unit state is
	// the authority - and are rebuilt when the oscillator is switched to noise
	// (shadow_rebuild: the state the device was left with by the last batch + this
	// batch's records of the voice so far).
	unsigned dphase = 0;
	int p_ramping = 0;
	uint64_t phase = 0;
	Ramp p = {0, 0, 0, 0};
	// --- second line ---
	// filter12 shadow: the cutoff ramper never leaves the host
	Ramp cutoff = {0, 0, 0, 0};
	unsigned flags = 0;
	int nin = 0, nout = 0, wired = 0;
	// fbdelay: delay line pair index, and the three tap lengths in frames (what the
	// device holds, fbdelay.c:194-196,231-247): they decide the voice's launch class
	int fbdbuf = -1;
	int fbd_taps[3] = { 0, 0, 0 };
	// fm: slot in the operator state pool
	int fmslot = -1;
	// xinsert: client slot (tap / inject buffers) and A2AMD_XIO_* mode
	int xio = -1;
	unsigned xio_mode = 0;
};

struct HVoice {
	// --- touched by every Process call of the voice: kept within one cache line ---
	bool live = false, dying = false;
	bool resolved = false, started = false;
	bool listed_recs = false;	// already in a2amd_ctx::with_recs
	uint8_t plain = 0;		// 0 not known, 1 no unit's Process leaves anything to do on the host but note
					// the window (a2amd_voice_process takes its short path), 2 not so
	bool fancy_recs = false;	// this batch's records hold something k_leaf_recs does not execute
	bool mode_mix = false;		// an oscillator played something else than a mip-mapped wave at some
	bool cls_stale = true;	// what the launch class is made of has changed since upload() last classified the voice (or it never has)
					// point of the batch being recorded (its records go to the general kernel)
	int nunits = 0;
	int win_off = -1, win_frames = 0;
	int win_done = 0;		// units of the chain that have processed the window in progress
	long long touched = -1;		// serial of the fragment of the last touch
	long long walked = -1;		// serial of the last fragment the engine made a Process call in
	long long default_seg = -1;	// serial of the fragment whose only event so far is the default
					// window (one Process(0, frames) per unit): no record is made
					// for it unless something else follows in that fragment
	size_t frag_mark = 0;		// recs.size() when that fragment was first touched
	// --- second line: the records ---
	std::vector<A2DRec> recs;	// this batch, fragment order
	std::vector<A2DRec> deferred;	// writes waiting behind the open window's SEG record
	// --- structure (set up once) ---
	uint64_t key = 0;
	int nlive = 0;
	int unit[A2D_MAXCHAIN];
	int depth = 0;
	int inline_pos = -1;		// chain position of the inline unit, if any
	int out_off = 0, out_nch = 0;
	int own_off = -1, own_nch = 0;
	int cls = 0;			// launch class (CLS_*), set when the lists are rebuilt
	int vm = -1;			// slot of the device VM that runs this voice's program (a2amd_vm_adopt), or -1
	// a control of the voice is gliding (a write with a duration: pitch / amplitude / q / volume / pan): until
	// this walk_time the voice is rendered by the window kernels even in batches without records - the quiet
	// kernels take a voice that is not settled fragment by fragment on the scalar unit (4x a settled one)
	uint64_t moving_until = 0;
	bool listed_moving = false;	// in a2amd_ctx::moving
	long long moving_run = -1;	// serial_base of the batch in which upload() gave it the stand-in record run
	long long dynf2_run = -1;	// ... in which it was put on the records kernels' list of 2 x wtosc-filter12-panmix voices
};

struct DepthRange { int fast_first = 0, fast_count = 0, fbd_first = 0, fbd_count = 0, gen_first = 0, gen_count = 0,
		dyn_first = 0, dyn_count = 0; };
enum { CLS_GENERIC = 0, CLS_OSCPAN, CLS_OSCFILTPAN, CLS_BUSDRIVER, CLS_BUSGENERIC, CLS_OSC2PAN, CLS_FMPAN, CLS_FBDCHAIN,
	CLS_OSC2FILTPAN };

// host side of an xinsert client slot
struct XioSlot {
	int unit = -1;
	int last_unit = -1;		// whose taps 'tap' holds (that unit may be gone by now)
	std::vector<int32_t> tap, inj;	// [fragment][A2AMD_MAXCHANNELS][64]
	bool inj_used = false;
	bool tapped = false;		// had READ clients at some point of the batch being recorded
	std::vector<int32_t> late;	// a2amd_unit_insert: what insert clients made of the taps, same layout
	bool late_used = false;
};

struct HWave {
	bool live = false;
	uint64_t key = 0;
	A2DWave dw;
	size_t pool_off = 0, pool_len = 0;	// its region of the device wave pool (int16 units)
};


// ---- the scripted voice's VM on the device (include/a2amd_vm.h, a2amd_vm.cpp) ----
struct HVmProg {
	uint64_t key = 0;		// the host's name for the function (its code pointer)
	uint32_t off = 0, n = 0;	// its text in the code pool
	uint64_t sum = 0;		// of the text: a key may come back with another program behind it
};
struct HVm {
	bool live = false;
	// adopted in the batch being recorded: the device takes over with the NEXT batch, the host's copy
	// of the interpreter covers the rest of this one (fragments adopt_frag + 1 ...) at upload time
	bool pending = false;
	int adopt_frag = -1;
	bool fresh = false;		// carried to the end of its first batch and sent up: the kernel runs it from the next batch on
	int voice = -1, prog = -1;
	uint8_t func = 0;
	bool has_exit = false;		// taken for a stretch only (vm_lookahead): the engine wants it back ...
	uint32_t exit_when = 0;		// ... in the fragment that holds this engine time
	A2DVmVoice st;			// pending: the state; else stale (the device's is the authority)
};
struct VmHost {
	std::vector<HVm> vms;
	std::vector<int> free_slots;
	std::vector<HVmProg> progs;
	struct Proof { int prog; unsigned pc; int32_t tick; uint32_t msdur; a2amd_vm_info info; };
	std::vector<Proof> proofs;	// a2amd_vm_analyze() results, by (program, pc, tick)
	std::vector<uint32_t> code;	// host copy of the code pool
	size_t code_uploaded = 0;	// words of it the device has
	std::vector<int> pending;	// slots adopted in the open batch
	int n_fresh = 0;		// ... of which upload() has sent this many up (they join the list when the batch ends)
	std::vector<int> list;		// slots the kernel runs (active on the device), and ...
	std::vector<int> cls_lists;	// ... their voices by launch class, for the records kernels: [osc1 | osc2 | filt1]
	std::vector<int> o2f_voices;	// ... and the 2 x wtosc-filter12-panmix voices among the others (their records come from
					// k_vm_emit: every batch they stand on the records kernels' list of that class, skip_empty)
	int n_cls[3] = { 0, 0, 0 };
	// d_list: [list | cls_lists | the classes' VM slots in cls_lists' order (k_vm_win) | the VM slots of no class]
	int n_other = 0;
	bool fused = false;		// this batch: the class voices run in k_vm_win (vm_issue decides, issue_windows launches)
	uint32_t batch_now = 0;		// engine time of this batch's first frame
	bool total_pending = false;	// faults of a fused batch: read a batch later (h_total[1], total_ev)
	hipEvent_t total_ev = nullptr;
	size_t pool_used = 0;		// window pool entries the last fused batch took (sizes the next one's pool)
	uint64_t fused_batches = 0, vm_batches = 0;
	bool fused_off = false;		// the pool overflowed under k_vm_win (a prediction that did not hold): records from then on
	// The pool room of a fused batch is PREDICTED: behind every batch k_vm_pool runs the class voices through the
	// batch the host expects next - same voices, same fragments, starting where this one ends - on a stream of its
	// own; a batch that is what was predicted runs fused with exactly that room, any other through records.
	uint64_t list_serial = 0;	// counts the rebuilds of the lists
	hipStream_t pred_stream = nullptr;
	hipEvent_t pred_after = nullptr, pred_ev = nullptr;
	hipEvent_t spec_go = nullptr;		// on pred_stream, right in front of a speculative pass: the render pass waits for it
	bool spec_go_pending = false;
	uint32_t spec_idle[3] = { 0, 0, 0 };	// of the pass that was taken: voices of each class it left to the quiet kernels
	uint64_t quiet_skipped = 0;		// quiet-kernel launches not made because of that (issue_kernels)
	unsigned *d_pred = nullptr, *h_pred = nullptr;
	bool pred_valid = false;
	uint64_t pred_serial = 0;
	uint32_t pred_now = 0;
	uint32_t pred_span = 0;		// frames
	size_t pred_entries = 0;	// ... of the batch being issued (fused)
	uint64_t pred_why[7] = { 0, 0, 0, 0, 0, 0, 0 };	// batches by what kept them from being fused (0: nothing)
	// Round 6: the prediction does the WORK.  k_vm_win is one wavefront's serial trip per 64 voices - 0.5 - 0.64 ms per
	// 64 fragments whatever the voice count - and stood in front of every scripted batch's render pass while the
	// engine thread waited.  vm_speculate (in vm_predict's place when the batch is whole 64-frame fragments) runs the
	// class voices' VMs AND their control state through the batch expected next, behind this batch's leaf kernels,
	// on pred_stream: window entries into a slot set and pool of its own (d_swin ...), the stepped state into shadow
	// arrays (A2DVmwOut).  A batch that is exactly the one predicted (vm_issue: same lists, same time, same fragments;
	// no overflow, no fault) takes it: k_vm_commit moves the shadows over, the render pass reads the speculative
	// slots, k_vm_win is not launched.  Any other batch ignores it - nothing live was touched - and still has the
	// pool bound k_vm_pool would have given it (the pass's own counter, h_spec[0]).  A2AMD_VMSPEC=0: off.
	// (two sets of slots and pool: the pass for batch N + 1 runs beside the render pass of batch N, which may be reading
	// the set the pass for N wrote; spec_set = the one the last pass wrote)
	DevBuf<int> d_swin[2], d_swext[2], d_swscr;
	DevBuf<unsigned> d_swidx[2];
	int spec_set = 0;
	bool spec_launched_now = false;		// vm_speculate has run for the batch being issued (issue_kernels asks once)
	unsigned *d_swtop = nullptr;		// [2]: entries taken, overflow flag (the first words of one block of 8: d_stotal follows)
	DevBuf<A2DVmVoice> d_vmv_sh;
	DevBuf<int> d_ustate_sh, d_vactive_sh;	// (d_ustate_sh.cap in units, like d_ustate's)
	DevBuf<A2DRun> d_runs_sh;
	uint32_t *d_stotal = nullptr;		// = d_swtop + 2, [6]: (unused), voices that faulted, [2 + class] voices the pass left alone
	unsigned *h_spec = nullptr;		// pinned: the block of 8 as the pass left it
	bool spec_valid = false;		// a pass was launched for the prediction pred_* describe
	bool spec_use = false;			// this batch: taken (vm_issue decides, issue_windows commits + renders)
	int spec_nfrags = 0, spec_cls[3] = { 0, 0, 0 };
	uint8_t spec_ff[A2D_MAXBATCH], spec_fb[A2D_MAXBATCH];	// the fragments the pass was run for (frames, offset in the engine's)
	uint32_t cut_hint[16];			// a2amd_vm_expect_cuts: engine times at which the root's VM wakes next
	int n_cut_hint = 0;
	size_t spec_cap = 0;			// pool entries the pass had room for
	size_t spec_demand = 0;			// ... and what the last pass that was looked at asked for (true also when it overflowed)
	uint64_t spec_launched = 0, spec_taken = 0, spec_overflows = 0;
	// why a batch was not followed by a pass (0 switched off / no class voices / capturing, 1 a fragment that is not a
	// whole one, 2 the slots would not fit the budget) and why a pass was not taken (0 the batch was not the one
	// predicted - vm_issue's own reasons, 1 other fragments or class counts, 2 the pool overflowed, 3 a voice faulted)
	uint64_t spec_skip[4] = { 0, 0, 0, 0 }, spec_miss[4] = { 0, 0, 0, 0 };	// (spec_skip[3]: the lists had only just changed)
	// A pass in flight reads its slot list (a pointer into d_list, fixed at launch), the VM voices, the code pool: whoever
	// REWRITES those in place - vm_build_lists when the membership changed, vm_prepare_batch when adopted voices go up -
	// waits for it first (vm_wait_spec).  A stale pass whose later kernels started behind such a rewrite read voice slots
	// where VM slots had been and indexed d_vmv past its end: a memory access fault, intermittent, in every scene whose
	// voices come and go (vmloops.a2s).  So that the wait stays rare, no pass is launched until the lists have stood
	// for two batches (rebuilt_at): a song whose notes change every buffer launches none.
	uint64_t spec_waits = 0;
	uint64_t rebuilt_at = 0;	// vm_batches when the lists were last rebuilt (a steady batch takes upload()'s quiet path and
				// never comes by vm_build_lists: stability is counted in batches issued, not in calls there)
	bool list_dirty = false;
	std::vector<std::pair<int, A2DVmVoice>> to_upload;	// (slot, state) going up with this batch
	uint32_t t0 = 0;		// engine time of the context's frame 0 (walk_time = 0)
	bool t0_valid = false;
	uint32_t msdur = 0;
	uint64_t batch_time = 0;	// walk_time when the batch being recorded began
	uint64_t replayed = 0;		// frames the kept batch has been re-run over (a2amd_replay / KEEP)
	std::vector<int32_t> f1tab;	// [32][65536]: made on a thread of its own from the moment the context opens (two
					// million libm sines, tens of ms: not on the engine thread in the middle of a fragment)
	std::future<std::vector<int32_t>> f1_future;
	uint32_t f1_future_sum = 0;	// (of the pitch table that thread works from)
	uint32_t f1tab_sum = 0;		// (of the pitch table it was made from)
	bool f1tab_up = false;
	// the device's states as the last batch left them, fetched when the engine wants voices back:
	// one voice at a time until it is clear that many are wanted, then all at once
	std::vector<A2DVmVoice> snap;
	std::vector<uint8_t> snap_have;
	int snap_fetches = 0;
	uint32_t rec_cap = 0;		// the VM's region of the blob, in records
	size_t rec_off = 0;		// ... its byte offset in the blob
	a2amd_vm_stats stats = {};
	int unwalked = -1;		// a voice under the device VM that the host's walk left out of a fragment
	DevBuf<A2DVmVoice> d_vmv;
	DevBuf<uint32_t> d_code;
	DevBuf<int> d_list;		// [active slots | class lists (voice slots)]
	DevBuf<A2DRun> d_vmrun;
	int32_t *d_f1tab = nullptr;
	std::vector<uint16_t> envlut;	// [8][66] from the host's env unit (a2amd_vm_envluts)
	uint16_t *d_envlut = nullptr;
	bool envlut_up = false;
	uint32_t *d_total = nullptr, *h_total = nullptr;	// {records, faults}; pinned
	uint32_t last_total = 0;	// records the count pass of this batch found
	A2DVmVoice *h_stage = nullptr;	// pinned staging for recalls
	size_t h_stage_cap = 0;
};
} // namespace a2h
using namespace a2h;

struct a2amd_ctx {
	a2amd_config cfg;
	char err[256];
	hipStream_t stream = nullptr;
	bool own_stream = false;
	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
	bool profiling = false;
	std::vector<hipEvent_t> ev_pool;	// triples: before leaf, after leaf, after root
	size_t ev_used = 0;
	uint32_t ptab[128];

	std::vector<HUnit> units;
	std::vector<int> free_units, deferred_free_units;
	std::vector<HVoice> voices;
	std::vector<int> free_voices, deferred_free_voices;
	std::vector<HWave> waves;
	int building = -1;
	std::vector<int> stack;			// open inline windows (unit ids)
	std::vector<A2DRec> up_recs;		// upload()'s scratch, kept for their capacity
	std::vector<int> up_idx, up_now;
	std::vector<A2DRun> up_val;
	std::vector<int> with_recs, prev_with_recs;	// voices carrying records this / last batch
	std::vector<int> dirty_voices;		// voice mirror entries to re-upload
	long long serial_base = 0;		// fragments rendered before this batch
	int n_leaf_dyn = 0, static_len = 0;
	int n_dyn_osc1 = 0, n_dyn_osc2 = 0, n_dyn_filt = 0;	// ... of n_leaf_dyn, first in the list: k_leaf_recs renders them
	bool o2f_quiet = true;			// this batch: the class's voices without records go to k_leaf_osc2filtpan (upload() decides)
	int n_dyn_filt2 = 0, n_dyn_rest = 0;	// ... 2 x wtosc-filter12-panmix (round 6: a quiet kernel of its own) / the general kernel's
	int n_o2f_leaf = 0;			// 2 x wtosc-filter12-panmix leaves (list_all, behind the general leaves)
	int n_started_live = 0;			// voices the engine is walking
	int walked_started = 0;			// ... of which it has walked this many in the open fragment
	int n_noise = 0, n_cutoff_ramps = 0;
	// Self-cleaning buses: when every bus is read by k_bus_driver (its owner is a plain
	// driver chain without records this batch), that kernel zeroes what it read and the
	// root stores the master bus instead of adding to it - the batch needs no memset.
	bool owners_all_driver = false;		// (static: set when the lists are rebuilt)
	bool consume_ok = false;		// ... and none of them carries records this batch
	bool others_clean = false;		// every bus but the root's is known to be zero
	bool root_clean = false;		// ... and the root's own bus
	bool capturing = false;			// issue_kernels is being captured into a graph
	int n_clients = 0;			// units whose clients are served (a2amd_unit_clients mode != 0)
	int sub_resume = -1;			// SUBTREES phase paused for insert clients: the depth it goes on with
	int paused_at = 0;			// ... and the depth whose insert clients are to be served now (0: none)
	std::vector<int32_t> snap_ustate, snap_vactive;	// unit states / voice liveness as the last batch left them
	bool snap_valid = false;			// (fetched when an oscillator is switched to noise)
	// The default map: one byte per voice slot, set by the HOST for a voice that
	// received exactly the engine's default window (Process(0, all frames) on every
	// unit, nothing else) in the open fragment - the one-store-per-voice fast path of
	// the voice walk (a2amd_default_map).  Cleared when a fragment opens.
	std::vector<uint8_t> defmap;
	bool defmap_used = false;		// the host asked for the map in the open fragment
	bool defmap_dirty = false;		// ... in some fragment since it was last zeroed
	// voices whose default window is reported "until further notice" (a2amd_default_hold): as if
	// their byte in the map were stored in every fragment
	std::vector<uint8_t> held;
	size_t n_held = 0;
	uint64_t walk_time = 0;			// frames of all fragments before the open one
	unsigned prev_frames = 0;		// length of the fragment before the open one

	// fragment clock
	bool frag_open = false;
	int cur_frag = 0, nfrags = 0;
	unsigned fragframes[A2D_MAXBATCH];
	uint8_t fragbase[A2D_MAXBATCH];	// where each starts inside the engine's own fragment (a2amd_fragment_offset)
	bool uploaded = false;

	// host mirrors of host-owned device tables
	std::vector<A2DVoice> mvoices;
	std::vector<A2DVoiceExt> mvext;		// units 8 - 15, once a chain of the context has them (long_chains)
	bool long_chains = false;
	std::vector<uint32_t> mudesc;
	std::vector<A2DWave> mwaves;
	bool voices_dirty = true, udesc_dirty = true, waves_dirty = true, lists_dirty = true, ptab_dirty = true;
	std::vector<int> list_all;		// leaf list followed by per-depth lists
	int n_leaf = 0;
	int n_fast_leaf = 0, n_osc2_leaf = 0, n_filt_leaf = 0;	// list_all = [wtosc-panmix | 2 x wtosc-panmix | wtosc-filter12-panmix | fm-panmix | general leaves | per depth ...]
	int n_fm_leaf = 0, fm_kind_count[8] = { 0 };		// fm-panmix: grouped by unit kind (fm1..fm4r), one launch each
	int n_list_pads = 0;
	std::vector<DepthRange> depth_ranges;	// index = depth
	bool hosttiming = false;		// A2AMD_HOSTTIMING
	int no_fast = 0;			// A2AMD_NO_FAST bit mask: 1 wtosc-panmix, 2 wtosc-filter12-panmix, 4 driver chains -> general kernel (debugging / A-B tests)

	// bus memory allocator (units of int32)
	size_t bus_stride_frames;
	size_t bus_used = 0;
	std::map<int, std::vector<int>> bus_free;	// nch -> offsets
	std::vector<std::pair<int,int>> deferred_bus_free;

	// fbdelay buffers
	int fbd_count = 0;
	std::vector<int> fbd_free, fbd_deferred_free, fbd_to_zero;
	// fm operator state pool
	int fm_count = 0;
	std::vector<int> fm_free, fm_deferred_free;

	// xinsert client slots
	std::vector<XioSlot> xio;
	std::vector<int> xio_free, xio_deferred_free;

	// wave pool (int16 samples)
	size_t wavepool_used = 0;
	std::vector<std::pair<size_t, size_t>> wavepool_free;	// (offset, length) of dropped waves' regions, sorted, coalesced
	// a dropped wave's pool region and table slot serve the batch being recorded to its
	// end (oscillators still name it until they have rendered a window and noticed)
	std::vector<std::pair<size_t, size_t>> deferred_wavepool_free;
	std::vector<int> free_wave_slots, deferred_wave_slots;

	DevBuf<A2DVoice> d_voices;
	DevBuf<A2DVoiceExt> d_vext;
	DevBuf<uint32_t> d_udesc;
	DevBuf<int32_t> d_ustate;	// cap in units
	DevBuf<int32_t> d_ustage;	// staging copy for time-sliced kernels
	DevBuf<int32_t> d_vactive;
	DevBuf<A2DRun> d_runs;
	DevBuf<A2DRec> d_recs;
	DevBuf<A2DWave> d_waves;
	DevBuf<int16_t> d_wavepool;
	DevBuf<int32_t> d_wavecoef;	// cap in pool samples, A2D_COEF_WORDS words each (a2amd_fast.hip: Coef4)
	DevBuf<int32_t> d_busmem;
	DevBuf<int32_t> d_fbdmem;	// cap in buffer pairs
	DevBuf<int32_t> d_fmstate;	// cap in slots of A2D_FMSTATE words
	DevBuf<int32_t> d_xio;		// cap in slots of A2D_XIO_SLOT words
	uint32_t *d_fmsine = nullptr;
	DevBuf<int> d_list;
	DevBuf<int> d_scatter;	// idx[k] then A2DRun[k] for k_scatter_runs
	uint32_t *d_ptab = nullptr;
	A2DParams *d_params = nullptr;	// = start of the device blob of the uploaded batch
	// Everything a batch ships - parameter block, records, run-table updates,
	// this batch's exception lists - is assembled in ONE pinned staging buffer
	// and sent with one asynchronous copy (no host sync at upload; two staging
	// buffers alternate, each guarded by an event).
	DevBuf<char> d_blob;
	char *h_blob[2] = { nullptr, nullptr };
	size_t h_blob_cap[2] = { 0, 0 };
	hipEvent_t blob_ev[2] = { nullptr, nullptr };
	bool blob_busy[2] = { false, false };
	int blob_i = 0;
	const int *d_dyn = nullptr;	// this batch's exception lists inside the blob
	A2DParams hparams;
	// [0] = GRAPH_STEPS whole runs of the batch, [1] = one run, [2] = its SUBTREES
	// phase alone, [3] = its ROOT phase alone (multi-GPU steps)
	// captured batches: [slot + 4 * v], slot = which phases (a2amd_render: run_phases, a2amd_replay),
	// v = where the root stores the master bus - 0 the bus memory, 1 / 2 one of the host's readback buffers
	hipGraph_t graph[12] = {};
	hipGraphExec_t gexec[12] = {};
	const int32_t *gdst[12] = {};
	bool gdirect[12] = {};		// (the captured root launch stored there)
	// channel 0 of what this context renders, kept on the device (a2amd_capture_begin)
	struct Capture { int32_t *d = nullptr; size_t cap = 0, n = 0; bool on = false; uint32_t *d_fragpos = nullptr; } capture;
	uint64_t wave_h2d_bytes = 0;
	uint32_t waves_uploaded = 0, waves_resident = 0;
	int32_t *master_dst = nullptr;	// where the root of the batch being issued is to store the master bus (nullptr: bus memory)
	bool master_direct = false;	// ... and its launch did
	int32_t *h_master = nullptr;	// pinned
	size_t h_master_cap = 0;
	// Identical-batch fast path: a batch without records, with the same fragment
	// lengths as the one uploaded before it and nothing changed in between (no
	// births, deaths, waves, clients) finds everything it needs on the device
	// already - no blob, no copy - and its launch sequence in a graph.
	bool blob_quiet = false;		// the uploaded blob describes a record-free batch
	int blob_nfrags = 0;
	unsigned blob_frames[A2D_MAXBATCH];
	int quiet_streak = 0;			// consecutive batches that took the fast path
	// A2AMD_RENDER_ASYNC: master-bus readbacks in flight (a2amd_collect delivers them)
	struct Readback {
		int32_t *h = nullptr;		// pinned
		size_t cap = 0;
		hipEvent_t ev = nullptr;
		int nfrags = 0;
		unsigned total = 0;
		uint8_t frames[A2D_MAXBATCH];
	} rb[2];
	int rb_head = 0, rb_count = 0;

	// multi-GPU: this context renders the voice subtrees it was given; the root
	// voice's inline bus is summed over the ranks' contexts by one RCCL reduce per
	// batch, the root chain runs on rank 0 (a2amd_dist_init)
	ncclComm_t comm = nullptr;
	int dist_rank = 0, dist_ranks = 1;
	bool dist_local = false;	// one of several contexts of this process (a2amd_dist_init_local)
	hipEvent_t grp_ev = nullptr;	// ... its SUBTREES phase is done / its partial has been taken

	// the window kernels (a2amd_win.hip): a slab's slots (one per fragment and voice), the pool of further
	// windows, where each voice's begin per fragment, the pool counter + overflow flag
	std::vector<int> moving;	// voices with moving_until set
	size_t n_moving_listed = 0;	// ... of which this batch's upload() gave the stand-in record run
	DevBuf<int> d_win, d_wext, d_wrc, d_wscr;	// (d_wscr: k_vm_win's rows of parked windows, A2D_VMW_ROW per voice)
	DevBuf<unsigned> d_widx;
	unsigned *d_wtop = nullptr;	// [2 sets]{ pool counter, overflow flag }
	unsigned *h_wtop = nullptr;	// ... copied back behind every batch (pinned), looked at before the next
	hipEvent_t wtop_ev = nullptr;
	bool wtop_pending = false;
	hipStream_t win_fork[7] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };	// a stream per list of a batch with several (issue_windows)
	hipEvent_t win_fev[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };	// [list] done, [7] fork
	hipStream_t win_stream = nullptr;	// the control passes' stream (slabs: issue_windows, a2amd_sched.cpp)
	hipEvent_t win_ev[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };	// control pass of set 0 / 1 done, render pass of set 0 / 1 done, fork

	a2amd_stats stats;
	VmHost vm;

	int fail(int code, const char *fmt, ...)
	{
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(err, sizeof(err), fmt, ap);
		va_end(ap);
		snprintf(g_err, sizeof(g_err), "%s", err);
		return code;
	}
};

// The current HIP device is per host thread; a context may be driven from a
// thread other than the one that opened it, and contexts of one process may
// live on different GPUs (one engine state per GPU).
static inline void use_device(const a2amd_ctx *c) { (void)hipSetDevice(c->cfg.device); }

namespace a2h {
// grow a device array; keep = preserve old contents (device-owned data)
template<class T>
int grow(a2amd_ctx *c, DevBuf<T> &b, size_t need, size_t elem_mult, bool keep)
{
	if(need <= b.cap)
		return 0;
	// (big elements - the 1 MB client slots - start small)
	size_t ncap = std::max(need, b.cap ? b.cap * 2 : (elem_mult >= 65536 ? (size_t)2 : (size_t)1024));
	T *nd = nullptr;
	HIPCHK(c, hipMalloc((void **)&nd, ncap * elem_mult * sizeof(T)));
	if(keep && b.d && b.cap) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		HIPCHK(c, hipMemcpyAsync(nd, b.d, b.cap * elem_mult * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
	}
	// (on the context's stream: a memset on the null stream is not ordered with
	// what this stream does to the new buffer next)
	if(keep)
		HIPCHK(c, hipMemsetAsync((char *)nd + b.cap * elem_mult * sizeof(T), 0,
				(ncap - b.cap) * elem_mult * sizeof(T), c->stream));
	if(keep)	// ... nor with the synchronous copies some callers make into it right away
		HIPCHK(c, hipStreamSynchronize(c->stream));
	if(b.d) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		// (round 6) ... and the device VM's speculative pass, on a non-blocking stream of its own, may still be READING
		// the old array through a pointer it was launched with - voices, unit state, VM voices, code, lists - long after
		// the batch it was launched behind is done: a pass that is not taken is not waited for anywhere else.  (Found as a
		// memory access fault in vmloops.a2s at a2_Run(4096) - notes being born while a stale pass ran - that went away
		// when every launch was waited for, A2AMD_WIN_SYNC=1.)  Arrays grow rarely: the wait costs nothing in a steady scene.
		if(c->vm.pred_stream)
			HIPCHK(c, hipStreamSynchronize(c->vm.pred_stream));
		HIPCHK(c, hipFree(b.d));
	}
	b.d = nd;
	b.cap = ncap;
	return 0;
}

inline int rec_tag(const a2amd_ctx *c) { return c->frag_open ? c->cur_frag : c->nfrags; }
static inline bool is_held(const a2amd_ctx *c, int vi)
{
	return (size_t)vi < c->held.size() && c->held[vi];
}

static inline void unhold(a2amd_ctx *c, int vi)
{
	if(is_held(c, vi)) {
		c->held[vi] = 0;
		--c->n_held;
	}
}

// filter12's cutoff ramper, which lives on the host: does it still need the engine's Process calls?
// Not only while it ramps: the call AFTER a ramp's last window is the one in which a2_PrepareRamper
// snaps the value to the target (a2_dsp.h:131-135) - a ramp's tail can end a frame's worth short of it -
// and the next a2_SetRamper starts from that value.  Only a ramper that has snapped may be left alone
// (the voice "plain", its default windows reported through the map).
inline bool cutoff_moving(const HUnit &u) { return u.cutoff.timer || u.cutoff.delta || u.cutoff.value != u.cutoff.target; }
// (the wavetable leaf kernels only know mip-mapped waves, "off" and noise)
inline bool leaf_mode(int mode) { return mode == A2D_OSC_MIPWAVE || mode == A2D_OSC_OFF || mode == A2D_OSC_NOISE; }
// a tap the frame-parallel delay kernel can take: at least one fragment long, and
// short enough not to wrap onto the frames being written
inline bool fbd_tap_ok(int frames) { return frames >= A2D_FRAG && frames <= A2D_FBD_BUFSIZE - A2D_FRAG; }
// a2amd_sched.cpp
void drop_graphs(a2amd_ctx *c);
void touch(a2amd_ctx *c, int vi);
void spell_out_pending(a2amd_ctx *c, int vi);
void push_rec(a2amd_ctx *c, int vi, int op, int unit, int reg, int value, unsigned dur, unsigned start);
int bus_alloc(a2amd_ctx *c, int nch);
int close_fragment(a2amd_ctx *c);
void resolve_out(a2amd_ctx *c, HVoice &v);
void sync_voice_mirror(a2amd_ctx *c, int vi);
int upload(a2amd_ctx *c);
void wavepool_release(a2amd_ctx *c, size_t off, size_t len);
void end_batch(a2amd_ctx *c);
int issue_kernels(a2amd_ctx *c, unsigned phases, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2);
int ensure_clean(a2amd_ctx *c);
int build_graph(a2amd_ctx *c, int slot, int steps, unsigned phases = A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT);
long long now_serial(const a2amd_ctx *c);
// a2amd_host.cpp
void wave_tap_policy(a2amd_ctx *c);
int capture_append(a2amd_ctx *c);	// a2amd_render(): the batch just rendered joins the capture
// a2amd_render.cpp
double *dbg_counters();	// (A2AMD_HOSTTIMING counters)
double *dbg_why();
int fetch_taps(a2amd_ctx *c, bool final);
// a2amd_dist.cpp
int dist_reduce_root(a2amd_ctx *c);
// a2amd_vm.cpp
int vm_prepare_batch(a2amd_ctx *c);		// upload(): pending adoptions, program text, states
int vm_build_lists(a2amd_ctx *c);		// ... and the kernel's list, the records kernels' class lists
int vm_issue(a2amd_ctx *c, bool fused);		// issue_kernels(): the VM kernel's two passes (fused: of the voices k_vm_win does not run)
void vm_class_params(a2amd_ctx *c, int k, A2DVmParams *vp);	// k_vm_win's parameters for a window class
int vm_fused_done(a2amd_ctx *c);		// ... and its fault count on its way back
int vm_predict(a2amd_ctx *c);			// k_vm_pool for the batch after this one (issue_kernels, behind the window kernels)
bool vm_spec_wanted(a2amd_ctx *c);		// round 6: may a speculative pass follow the batch being issued?
// ... and over which fragments: the span of this batch again, cut where the root's VM is expected to wake (cut_hint)
bool vm_spec_plan(a2amd_ctx *c, int *nfrags, uint8_t *ff, uint8_t *fb);
int vm_speculate(a2amd_ctx *c);			// ... k_vm_win for the batch after this one, into shadows (behind the leaf kernels)
int vm_take_back(a2amd_ctx *c, int vi, bool inclusive, a2amd_vm_state *out, a2amd_vm_env *envs_out = nullptr);	// the voice is the host's again
void vm_end_batch(a2amd_ctx *c);
void vm_close(a2amd_ctx *c);
void vm_start_f1tab(a2amd_ctx *c);		// the cutoff -> coefficient table, made on a thread of its own
int vm_blob_room(a2amd_ctx *c);			// records the VM's region of the blob should hold

} // namespace a2h

#endif // A2AMD_HOST_H
