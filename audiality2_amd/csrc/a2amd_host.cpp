// a2amd_host.cpp - host half of liba2amd.so: the C ABI of include/a2amd.h.
//
// Records what the Audiality 2 engine asks of its units during a batch of
// fragments (per voice, in call order), keeps the little host-side state the
// reference itself computes with libm or draws from the engine-global RNG
// (filter12's cutoff -> coefficient, the noise oscillators' draw counts), and
// at a2amd_render() ships the records and launches the kernels:
//
//   leaf voices (no 'inline' unit)            one launch, all in parallel
//   voices with an 'inline' unit, by nesting  one launch per depth, deepest first
//
// which is the reference's depth-first voice walk (src/core.c:1883-1896) turned
// inside out: children only ever ADD into the bus their parent's inline unit
// collects (core.c:479-480, inline.c:32-33), so all children of all parents can
// run first and each parent picks the sum up where its inline unit sits.
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

#include "../../include/a2amd.h"
#include "a2amd_device.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {
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
} g_rccl;

// RCCL is bound on first use: the copy the process already has (a host application
// that brought its own, e.g. PyTorch's) if there is one - two copies in one process
// would each run their own proxy threads and topology detection - else ROCm's
bool rccl_bind()
{
	if(g_rccl.Reduce)
		return true;
	const char *names[] = { "librccl.so.1", "librccl.so" };
	void *h = nullptr;
	for(const char *n : names)
		if(!h)
			h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
	for(const char *n : names)
		if(!h)
			h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
	if(!h)
		return false;
	g_rccl.lib = h;
	*(void **)&g_rccl.GetUniqueId = dlsym(h, "ncclGetUniqueId");
	*(void **)&g_rccl.CommInitRank = dlsym(h, "ncclCommInitRank");
	*(void **)&g_rccl.CommDestroy = dlsym(h, "ncclCommDestroy");
	*(void **)&g_rccl.GetErrorString = dlsym(h, "ncclGetErrorString");
	*(void **)&g_rccl.Reduce = dlsym(h, "ncclReduce");
	*(void **)&g_rccl.CommInitAll = dlsym(h, "ncclCommInitAll");
	*(void **)&g_rccl.GroupStart = dlsym(h, "ncclGroupStart");
	*(void **)&g_rccl.GroupEnd = dlsym(h, "ncclGroupEnd");
	return g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.Reduce;
}
} // namespace

namespace {

thread_local char g_err[256] = "";

// ---- host copies of the few reference formulas the host must evaluate ----
struct Ramp { int value, target, delta, timer; };

inline int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
inline int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
inline int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// a2_InitRamper / a2_PrepareRamper / a2_RunRamper / a2_SetRamper, a2_dsp.h:121-170
void ramp_init(Ramp &r, int v) { r.value = r.target = (int)((unsigned)v << 8); r.delta = r.timer = 0; }
void ramp_prepare(Ramp &r, int frames)
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
void ramp_run(Ramp &r, int frames) { r.value = wadd(r.value, wmul(r.delta, frames)); }
void ramp_set(Ramp &r, int target, int start, int duration)
{
	r.target = (int)((unsigned)target << 8);
	r.timer = wadd(duration, start);
	if(r.timer < 256)
		r.value = r.target;
	else
		r.value = wadd(r.value, wmul(r.delta, start) >> 8);
}

// a2_pitch_open, pitch.c:70-96
void build_pitch_table(uint32_t *tab)
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
unsigned p2i(const uint32_t *tab, int pitch)
{
	int n = pitch & 0xffff, oct = pitch >> 16;
	unsigned dph = tab[2 * (n >> 10) + 1] * (unsigned)(n & 0x3ff);
	dph >>= 2;
	dph += tab[2 * (n >> 10)];
	return dph >> ((unsigned)(7 - oct) & 31u);
}

// f12_pitch2coeff, filter12.c:65-72 -- float/double libm maths: host only
int f12_coeff(const uint32_t *tab, int cutoff_value, int samplerate)
{
	float f = p2i(tab, cutoff_value >> 8) * (261.626f / 16777216.0f);
	if(f > (samplerate >> 2))
		return 362 << 16;
	return (int)(512.0f * 65536.0f * sin(M_PI * f / samplerate));
}

// dcb_pitch2coeff, dcblock.c:57-64: the same maths from a 16:16 pitch
int dcb_coeff(const uint32_t *tab, int cutoff, int samplerate)
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

} // namespace

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
	bool uploaded = false;

	// host mirrors of host-owned device tables
	std::vector<A2DVoice> mvoices;
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
	DevBuf<uint32_t> d_udesc;
	DevBuf<int32_t> d_ustate;	// cap in units
	DevBuf<int32_t> d_ustage;	// staging copy for time-sliced kernels
	DevBuf<int32_t> d_vactive;
	DevBuf<A2DRun> d_runs;
	DevBuf<A2DRec> d_recs;
	DevBuf<A2DWave> d_waves;
	DevBuf<int16_t> d_wavepool;
	DevBuf<int32_t> d_wavecoef;	// cap in pool samples, 3 words each (a2amd_fast.hip: Coef3)
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
	hipGraph_t graph[4] = {nullptr, nullptr, nullptr, nullptr};
	hipGraphExec_t gexec[4] = {nullptr, nullptr, nullptr, nullptr};
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

	a2amd_stats stats;

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

namespace {

void drop_graphs(a2amd_ctx *c);

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
		HIPCHK(c, hipFree(b.d));
	}
	b.d = nd;
	b.cap = ncap;
	return 0;
}

double *dbg_counters();	// (A2AMD_HOSTTIMING counters, defined with the timing dump below)
double *dbg_why();
int rec_tag(const a2amd_ctx *c) { return c->frag_open ? c->cur_frag : c->nfrags; }

void touch(a2amd_ctx *c, int vi)
{
	HVoice &v = c->voices[vi];
	const long long serial = c->serial_base + rec_tag(c);
	if(v.touched != serial) {
		v.touched = serial;
		v.frag_mark = v.recs.size();
	}
}

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

// The windows of the open fragment that were left unrecorded so far - the voice's
// default window, noted in HVoice::default_seg or by the host in the default map -
// become records: something else follows in the same fragment after all.
void spell_out_pending(a2amd_ctx *c, int vi)
{
	HVoice &dv = c->voices[vi];
	const bool marked = c->frag_open && c->defmap_used && (size_t)vi < c->defmap.size() && c->defmap[vi];
	if(marked || (c->frag_open && is_held(c, vi))) {
		// (a held voice that gets a record after all - its chain taken down from outside the
		// walk, say - had its default window in this fragment like a marked one; the hold ends)
		if(marked)
			c->defmap[vi] = 0;
		unhold(c, vi);
		dv.default_seg = c->serial_base + c->cur_frag;
		if(dv.walked != c->serial_base + c->cur_frag) {
			dv.walked = c->serial_base + c->cur_frag;
			++c->walked_started;
		}
	}
	if(dv.default_seg == c->serial_base + rec_tag(c)) {
		dv.default_seg = -1;
		A2DRec seg = { A2D_HEAD(rec_tag(c), R_SEG, 0, 0), 0, (unsigned)c->fragframes[rec_tag(c)] << 16, 0 };
		if(!dv.listed_recs) {
			dv.listed_recs = true;
			c->with_recs.push_back(vi);
		}
		dv.recs.push_back(seg);
	}
}

void push_rec(a2amd_ctx *c, int vi, int op, int unit, int reg, int value, unsigned dur, unsigned start)
{
	touch(c, vi);
	spell_out_pending(c, vi);
	A2DRec r;
	r.head = A2D_HEAD(rec_tag(c), op, unit, reg);
	r.value = value;
	r.dur = dur;
	r.start = start;
	HVoice &v = c->voices[vi];
	if(!v.listed_recs) {
		v.listed_recs = true;
		c->with_recs.push_back(vi);
	}
	// A write that reaches a unit which has already rendered the window the
	// voice is in (a control wire from a unit further down the chain, e.g. the
	// engine's env, env.c:135) takes effect after that window: the device
	// executes a window as one SEG record, so the write waits behind it.
	if(op != R_SEG && op != R_INIT && op != R_KILL && unit < v.win_done) {
		v.deferred.push_back(r);
		return;
	}
	v.recs.push_back(r);
	if(op == R_SEG) {
		v.win_done = 0;
		for(A2DRec &d : v.deferred)
			v.recs.push_back(d);
		v.deferred.clear();
	}
}

int bus_alloc(a2amd_ctx *c, int nch)
{
	auto &fl = c->bus_free[nch];
	if(!fl.empty()) {
		int off = fl.back();
		fl.pop_back();
		return off;
	}
	size_t off = c->bus_used;
	c->bus_used += c->bus_stride_frames * (size_t)nch;
	return (int)off;
}

long long now_serial(const a2amd_ctx *c) { return c->serial_base + rec_tag(c); }

// the engine-visible walk found no work for the VMs: close the fragment
int close_fragment(a2amd_ctx *c)
{
	if(!c->frag_open)
		return 0;
	const int f = c->cur_frag;
	const unsigned nframes = c->fragframes[f];
	if(c->defmap_used) {
		size_t n = 0;
		const size_t nv = std::min(c->defmap.size(), c->voices.size());
		for(size_t k = 0; k < nv; ++k)
			n += c->defmap[k];
		c->walked_started += (int)n;
	}
	c->walked_started += (int)c->n_held;
	if(c->walked_started != c->n_started_live) {
		if(c->hosttiming)
			dbg_counters()[2] += 1;
		// a live voice got no Process call this fragment: say so, or
		// the kernel would apply the default
		for(size_t vi = 0; vi < c->voices.size(); ++vi) {
			HVoice &v = c->voices[vi];
			if((c->defmap_used && vi < c->defmap.size() && c->defmap[vi]) || is_held(c, (int)vi))
				continue;	// (walked: the host marked its default window)
			if(v.live && v.started && !v.dying && v.walked != c->serial_base + f &&
					v.touched != c->serial_base + f) {
				A2DRec r = { A2D_HEAD(f, R_NOP, 0, 0), 0, 0, 0 };
				if(!v.listed_recs) {
					v.listed_recs = true;
					c->with_recs.push_back((int)vi);
				}
				v.recs.push_back(r);
				v.touched = c->serial_base + f;
				if(c->hosttiming)
					dbg_counters()[3] += 1;
			}
		}
	}
	c->frag_open = false;
	c->walk_time += nframes;
	c->prev_frames = nframes;
	c->defmap_used = false;
	return 0;
}

void resolve_out(a2amd_ctx *c, HVoice &v)
{
	if(v.resolved)
		return;
	v.depth = (int)c->stack.size();
	if(c->stack.empty()) {
		v.out_off = 0;
		v.out_nch = c->cfg.channels;
	} else {
		const HUnit &il = c->units[c->stack.back()];
		const HVoice &pv = c->voices[il.voice];
		if(il.wired) {
			v.out_off = pv.out_off;
			v.out_nch = pv.out_nch;
		} else {
			v.out_off = pv.own_off;
			v.out_nch = pv.own_nch;
		}
	}
	v.resolved = true;
	c->voices_dirty = true;
	c->dirty_voices.push_back((int)(&v - c->voices.data()));
	c->lists_dirty = true;
}

void sync_voice_mirror(a2amd_ctx *c, int vi)
{
	const HVoice &v = c->voices[vi];
	if(c->mvoices.size() <= (size_t)vi)
		c->mvoices.resize(vi + 1);
	A2DVoice &m = c->mvoices[vi];
	memset(&m, 0, sizeof(m));
	m.nunits = v.nunits;
	for(int i = 0; i < v.nunits; ++i)
		m.unit[i] = v.unit[i];
	m.out_off = v.out_off;
	m.out_nch = v.out_nch;
	m.own_off = v.own_off;
	m.own_nch = v.own_nch;
}

// what the wavetable leaf kernels play: mip-mapped waves, nothing, and - k_leaf_recs only, but a noise
// oscillator's every window carries an R_NOISESEED record, so the quiet kernels never see one - noise
inline bool leaf_mode(int mode) { return mode == A2D_OSC_MIPWAVE || mode == A2D_OSC_OFF || mode == A2D_OSC_NOISE; }

// wtosc (mip-mapped wave playing) -> panmix 1->2 adding into the output bus
bool is_oscpan_chain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits != 2 || v.out_nch < 2)
		return false;
	const HUnit &o = c->units[v.unit[0]], &pm = c->units[v.unit[1]];
	return o.kind == A2AMD_WTOSC && !(o.flags & A2AMD_PROCADD) && !o.wired &&
			leaf_mode(o.mode) &&
			pm.kind == A2AMD_PANMIX && pm.nin == 1 && pm.nout == 2 && pm.wired &&
			(pm.flags & A2AMD_PROCADD);
}

// wtosc (replacing) + wtosc (adding) -> panmix 1->2 adding into the output bus
bool is_osc2pan_chain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits != 3 || v.out_nch < 2)
		return false;
	const HUnit &a = c->units[v.unit[0]], &b = c->units[v.unit[1]], &pm = c->units[v.unit[2]];
	return a.kind == A2AMD_WTOSC && !(a.flags & A2AMD_PROCADD) && !a.wired &&
			leaf_mode(a.mode) &&
			b.kind == A2AMD_WTOSC && (b.flags & A2AMD_PROCADD) && !b.wired &&
			leaf_mode(b.mode) &&
			pm.kind == A2AMD_PANMIX && pm.nin == 1 && pm.nout == 2 && pm.wired &&
			(pm.flags & A2AMD_PROCADD);
}

// wtosc -> filter12 (1 channel, replacing) -> panmix 1->2 adding into the output bus
bool is_oscfiltpan_chain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits != 3 || v.out_nch < 2)
		return false;
	const HUnit &o = c->units[v.unit[0]], &f = c->units[v.unit[1]], &pm = c->units[v.unit[2]];
	return o.kind == A2AMD_WTOSC && !(o.flags & A2AMD_PROCADD) && !o.wired &&
			leaf_mode(o.mode) &&
			f.kind == A2AMD_FILTER12 && f.nin == 1 && !f.wired && !(f.flags & A2AMD_PROCADD) &&
			// (a ramping cutoff means an R_F1RAMP record per window: such a voice is never
			// without records, and the quiet kernel skips voices with records)
			pm.kind == A2AMD_PANMIX && pm.nin == 1 && pm.nout == 2 && pm.wired &&
			(pm.flags & A2AMD_PROCADD);
}

// wtosc (replacing) + wtosc (adding) -> filter12 (1 channel, replacing) -> panmix 1->2 adding into the
// output bus: the usual subtractive-synth note
bool is_osc2filtpan_chain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits != 4 || v.out_nch < 2)
		return false;
	const HUnit &a = c->units[v.unit[0]], &b = c->units[v.unit[1]], &f = c->units[v.unit[2]], &pm = c->units[v.unit[3]];
	return a.kind == A2AMD_WTOSC && !(a.flags & A2AMD_PROCADD) && !a.wired &&
			leaf_mode(a.mode) &&
			b.kind == A2AMD_WTOSC && (b.flags & A2AMD_PROCADD) && !b.wired &&
			leaf_mode(b.mode) &&
			f.kind == A2AMD_FILTER12 && f.nin == 1 && !f.wired && !(f.flags & A2AMD_PROCADD) &&
			pm.kind == A2AMD_PANMIX && pm.nin == 1 && pm.nout == 2 && pm.wired &&
			(pm.flags & A2AMD_PROCADD);
}

// fmN -> panmix 1->2 adding into the output bus
bool is_fmpan_chain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits != 2 || v.out_nch < 2)
		return false;
	const HUnit &o = c->units[v.unit[0]], &pm = c->units[v.unit[1]];
	return o.kind >= A2AMD_FM1 && o.kind <= A2AMD_FM4R && !(o.flags & A2AMD_PROCADD) && !o.wired &&
			pm.kind == A2AMD_PANMIX && pm.nin == 1 && pm.nout == 2 && pm.wired &&
			(pm.flags & A2AMD_PROCADD);
}

// inline 0 2; panmix 2 2; xinsert 2 >  (a2_rootdriver / a2_groupdriver)
bool is_driver_chain(const a2amd_ctx *c, const HVoice &v)
{
	if((v.nunits != 3 && v.nunits != 2) || v.own_nch != 2 || v.out_nch < 2 || v.own_off < 0)
		return false;
	const HUnit &il = c->units[v.unit[0]], &pm = c->units[v.unit[1]];
	if(!(il.kind == A2AMD_INLINE && !(il.flags & A2AMD_PROCADD) && !il.wired && il.nout == 2 &&
			pm.kind == A2AMD_PANMIX && pm.nin == 2 && pm.nout == 2))
		return false;
	if(v.nunits == 2)
		// inline 0 2; panmix 2 >  - what the drop-in's root voice looks like from here (the root's
		// xinsert stays the engine's): the panmix itself adds into the output bus
		return pm.wired && (pm.flags & A2AMD_PROCADD);
	const HUnit &xi = c->units[v.unit[2]];
	return !pm.wired && !(pm.flags & A2AMD_PROCADD) &&
			xi.kind == A2AMD_XINSERT && xi.nin == 2 && xi.wired && (xi.flags & A2AMD_PROCADD) &&
			!xi.xio_mode;	// (clients: the general kernel serves them)
}

// a tap the frame-parallel delay kernel can take: at least one fragment long, and
// short enough not to wrap onto the frames being written
inline bool fbd_tap_ok(int frames) { return frames >= A2D_FRAG && frames <= A2D_FBD_BUFSIZE - A2D_FRAG; }

// inline 0 2; fbdelay 2 2 [; fbdelay 2 2 ...]; the last one wired and adding
// (the group voices of benchmark/fmtest4.a2s:83-95), every tap >= one fragment
bool is_fbdchain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits < 2 || v.nunits > 5 || v.own_nch != 2 || v.out_nch < 2 || v.own_off < 0 || v.inline_pos != 0)
		return false;
	const HUnit &il = c->units[v.unit[0]];
	if(il.kind != A2AMD_INLINE || (il.flags & A2AMD_PROCADD) || il.wired || il.nout != 2)
		return false;
	for(int k = 1; k < v.nunits; ++k) {
		const HUnit &d = c->units[v.unit[k]];
		const bool last = k == v.nunits - 1;
		if(d.kind != A2AMD_FBDELAY || d.nin != 2 || d.nout != 2 || (d.wired != 0) != last ||
				(last && !(d.flags & A2AMD_PROCADD)))
			return false;
		for(int t = 0; t < 3; ++t)
			if(!fbd_tap_ok(d.fbd_taps[t]))
				return false;
	}
	return true;
}

int upload(a2amd_ctx *c)
{
	if(c->hosttiming) {
		// (A2AMD_HOSTTIMING: why a batch did not take the quiet path - first reason that applies)
		const int why = !c->blob_quiet ? 0 : !c->with_recs.empty() ? 1 : !c->prev_with_recs.empty() ? 2 :
				c->voices_dirty ? 3 : c->udesc_dirty ? 4 : c->waves_dirty ? 5 : c->lists_dirty ? 6 : c->ptab_dirty ? 7 :
				!c->dirty_voices.empty() ? 8 : !c->fbd_to_zero.empty() ? 9 : c->nfrags != c->blob_nfrags ? 10 :
				c->bus_used > c->d_busmem.cap ? 11 :
				memcmp(c->fragframes, c->blob_frames, (size_t)c->nfrags * sizeof(unsigned)) ? 10 : 12;
		dbg_why()[why] += 1;
	}
	if(c->blob_quiet && c->with_recs.empty() && c->prev_with_recs.empty() && !c->voices_dirty &&
			!c->udesc_dirty && !c->waves_dirty && !c->lists_dirty && !c->ptab_dirty &&
			c->dirty_voices.empty() && c->fbd_to_zero.empty() && c->bus_used <= c->d_busmem.cap) {
		bool inject = false;
		for(const XioSlot &x : c->xio)
			if(x.unit >= 0 && (x.inj_used || (c->units[x.unit].xio_mode & A2AMD_XIO_INJECT)))
				inject = true;
		const bool same = c->nfrags == c->blob_nfrags &&
				!memcmp(c->fragframes, c->blob_frames, (size_t)c->nfrags * sizeof(unsigned));
		if(!inject && same) {
			// the same quiet batch again: the device has it all (graphs stay valid)
			if(c->hosttiming)
				dbg_counters()[0] += 1;
			++c->quiet_streak;
			c->uploaded = true;
			return 0;
		}
		if(!inject && c->d_params && c->stream) {
			// A quiet batch again, cut into fragments differently - the engine's root voice woke
			// up in the middle of a fragment, which it does every 3 906 frames while it idles at
			// 'end' (core.c:1195), i.e. once per a2play buffer: nothing per voice has changed, only
			// the fragment table of the parameter block.  That block alone goes up again (a few
			// hundred bytes instead of a pass over every voice); graphs read it on the device and
			// stay valid while the NUMBER of fragments - their launch shapes - is the same.
			A2DParams p = c->hparams;
			p.nfrags = c->nfrags;
			memset(p.fragframes, 0, sizeof(p.fragframes));
			memset(p.fragstart, 0, sizeof(p.fragstart));
			for(int f = 0, acc = 0; f < c->nfrags; ++f) {
				p.fragframes[f] = (uint8_t)c->fragframes[f];
				p.fragstart[f] = (uint16_t)acc;
				acc += (int)c->fragframes[f];
			}
			const int bi = c->blob_i;
			c->blob_i ^= 1;
			if(c->blob_busy[bi]) {
				HIPCHK(c, hipEventSynchronize(c->blob_ev[bi]));
				c->blob_busy[bi] = false;
			}
			if(c->h_blob[bi] && c->h_blob_cap[bi] >= sizeof(A2DParams) && c->blob_ev[bi]) {
				memcpy(c->h_blob[bi], &p, sizeof(p));
				HIPCHK(c, hipMemcpyAsync(c->d_blob.d, c->h_blob[bi], sizeof(p), hipMemcpyHostToDevice, c->stream));
				HIPCHK(c, hipEventRecord(c->blob_ev[bi], c->stream));
				c->blob_busy[bi] = true;
				c->hparams = p;
				if(c->nfrags != c->blob_nfrags) {
					drop_graphs(c);
					c->quiet_streak = 0;
				} else
					++c->quiet_streak;
				c->blob_nfrags = c->nfrags;
				memcpy(c->blob_frames, c->fragframes, (size_t)c->nfrags * sizeof(unsigned));
				if(c->hosttiming)
					dbg_counters()[0] += 1;
				c->uploaded = true;
				return 0;
			}
		}
	}
	c->blob_quiet = false;
	c->quiet_streak = 0;
	drop_graphs(c);
	const size_t nv = c->voices.size(), nu = c->units.size();
	// capacities
	// (kept: only the entries that changed are re-sent below)
	if(int r = grow(c, c->d_voices, nv, 1, true)) return r;
	if(int r = grow(c, c->d_udesc, nu, 1, false)) return r;
	if(int r = grow(c, c->d_ustate, nu, A2D_USTATE, true)) return r;
	if(int r = grow(c, c->d_ustage, c->d_ustate.cap, A2D_USTATE, false)) return r;
	if(int r = grow(c, c->d_vactive, nv, 1, true)) return r;
	if(int r = grow(c, c->d_runs, nv, 1, true)) return r;
	{
		const int32_t *before = c->d_busmem.d;
		if(int r = grow(c, c->d_busmem, c->bus_used, 1, false)) return r;
		if(c->d_busmem.d != before)
			c->others_clean = c->root_clean = false;
	}
	if(c->fbd_count)
		if(int r = grow(c, c->d_fbdmem, c->fbd_count, 2 * (size_t)A2D_FBD_BUFSIZE, true)) return r;
	if(!c->xio.empty()) {
		if(int r = grow(c, c->d_xio, c->xio.size(), A2D_XIO_SLOT, true)) return r;
		// what the WRITE clients produced for this batch's fragments
		const size_t n = (size_t)c->nfrags * A2AMD_MAXCHANNELS * A2D_FRAG;
		for(size_t k = 0; k < c->xio.size(); ++k) {
			XioSlot &x = c->xio[k];
			// (also when the clients left in the middle of the batch, and zeros
			// over last batch's when they produced nothing)
			if(x.unit < 0 || !(x.inj_used || (c->units[x.unit].xio_mode & A2AMD_XIO_INJECT)))
				continue;
			HIPCHK(c, hipMemcpyAsync(c->d_xio.d + k * A2D_XIO_SLOT + A2D_XIO_HALF, x.inj.data(),
					n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			if(x.inj_used)
				std::fill(x.inj.begin(), x.inj.begin() + n, 0);
			x.inj_used = false;
		}
	}
	if(c->fm_count) {
		if(int r = grow(c, c->d_fmstate, c->fm_count, A2D_FMSTATE, true)) return r;
		if(!c->d_fmsine) {
			// fm_OpenState, fm.c:493-501: one period of sine and one pad
			// sample, computed here with the reference's own expression
			// (libm on the host) and shipped as {s[i], s[i+1] - s[i]} pairs
			int16_t sine[2049];
			uint32_t pairs[2048];
			for(int k = 0; k < 2049; ++k)
				sine[k] = (int16_t)(sin(k * 2.0f * M_PI / 2048) * 32767.0f);
			for(int k = 0; k < 2048; ++k)
				pairs[k] = (uint32_t)(uint16_t)sine[k] | ((uint32_t)(sine[k + 1] - sine[k]) << 16);
			HIPCHK(c, hipMalloc((void **)&c->d_fmsine, sizeof(pairs)));
			HIPCHK(c, hipMemcpy(c->d_fmsine, pairs, sizeof(pairs), hipMemcpyHostToDevice));
		}
	}

	if(c->voices_dirty && nv) {
		// re-upload the span of voice table entries that changed
		if(c->mvoices.size() < nv)
			c->mvoices.resize(nv);
		int lo = (int)nv, hi = -1;
		for(int vi : c->dirty_voices)
			if(vi < (int)nv) {
				sync_voice_mirror(c, vi);
				lo = std::min(lo, vi);
				hi = std::max(hi, vi);
			}
		c->dirty_voices.clear();
		if(hi >= lo)
			HIPCHK(c, hipMemcpyAsync(c->d_voices.d + lo, c->mvoices.data() + lo,
					(size_t)(hi - lo + 1) * sizeof(A2DVoice), hipMemcpyHostToDevice, c->stream));
		c->voices_dirty = false;
	}
	if(c->udesc_dirty && nu) {
		HIPCHK(c, hipMemcpyAsync(c->d_udesc.d, c->mudesc.data(), nu * sizeof(uint32_t),
				hipMemcpyHostToDevice, c->stream));
		c->udesc_dirty = false;
	}
	if(c->waves_dirty && !c->mwaves.empty()) {
		if(int r = grow(c, c->d_waves, c->mwaves.size(), 1, false)) return r;
		HIPCHK(c, hipMemcpyAsync(c->d_waves.d, c->mwaves.data(), c->mwaves.size() * sizeof(A2DWave),
				hipMemcpyHostToDevice, c->stream));
		c->waves_dirty = false;
	}
	if(c->ptab_dirty) {
		HIPCHK(c, hipMemcpyAsync(c->d_ptab, c->ptab, sizeof(c->ptab), hipMemcpyHostToDevice, c->stream));
		c->ptab_dirty = false;
	}
	for(int b : c->fbd_to_zero)
		HIPCHK(c, hipMemsetAsync(c->d_fbdmem.d + (size_t)b * 2 * A2D_FBD_BUFSIZE, 0,
				2 * (size_t)A2D_FBD_BUFSIZE * sizeof(int32_t), c->stream));
	c->fbd_to_zero.clear();

	// Records: one contiguous run per voice that has any.  The device keeps a
	// dense runs[slot] table (zero = quiet voice); only the entries that change
	// are written, by a scatter kernel: this batch's runs, and zeros for the
	// voices that carried records last batch but not now.
	std::vector<A2DRec> &recs = c->up_recs;
	std::vector<int> &sc_idx = c->up_idx, &now = c->up_now;
	std::vector<A2DRun> &sc_val = c->up_val;
	recs.clear();
	now.clear();
	{
		const size_t nsc = c->with_recs.size() + c->prev_with_recs.size();
		sc_idx.resize(nsc);
		sc_val.resize(nsc);
		now.resize(c->with_recs.size());
	}
	size_t nsc_used = 0, nnow = 0;
	for(int vi : c->with_recs) {
		HVoice &v = c->voices[vi];
		if(v.recs.empty()) {
			v.listed_recs = false;
			continue;
		}
		A2DRun r = { (int)recs.size(), (int)v.recs.size() };
		recs.insert(recs.end(), v.recs.begin(), v.recs.end());
		sc_idx[nsc_used] = vi;
		sc_val[nsc_used++] = r;
		now[nnow++] = vi;
	}
	now.resize(nnow);
	c->with_recs.swap(now);
	for(int vi : c->prev_with_recs)
		if(vi < (int)nv && c->voices[vi].recs.empty()) {
			A2DRun z = { 0, 0 };
			sc_idx[nsc_used] = vi;
			sc_val[nsc_used++] = z;
		}
	sc_idx.resize(nsc_used);
	sc_val.resize(nsc_used);
	c->prev_with_recs.clear();
	c->stats.records += recs.size();
	if(c->hosttiming) {
		dbg_counters()[1] += (double)recs.size();
		dbg_counters()[4] += (double)c->with_recs.size();
	}

	// Launch lists.  Static part, rebuilt when the voice tree changes: every
	// listed voice by class -
	//   leaves: [wtosc-panmix | wtosc-filter12-panmix | general]  (each sorted by
	//           output bus so a wavefront can sum several voices before touching it)
	//   voices with an inline unit, per nesting depth: [driver chain | general]
	// The fast kernels skip a voice whose runs[] entry is non-zero; those voices
	// form the dynamic part (this batch's exceptions) and go to the general kernel.
	if(c->lists_dirty) {
		bool owners_ok = !getenv("A2AMD_NO_SELFCLEAN"), root_driver = false;
		std::vector<int> fast_leaf, osc2_leaf, filt_leaf, fm_leaf, gen_leaf, o2f_leaf;
		std::map<int, std::pair<std::vector<int>, std::vector<int>>> bydepth;
		std::map<int, std::vector<int>> fbd_bydepth;
		int maxdepth = -1;
		for(size_t vi = 0; vi < nv; ++vi) {
			HVoice &v = c->voices[vi];
			// voices that died during this batch still render up to their
			// R_KILL record
			if(!(v.live || v.dying) || !v.resolved) {
				v.cls = -1;
				continue;
			}
			if(v.inline_pos >= 0) {
				auto &d = bydepth[v.depth];
				v.cls = !(c->no_fast & 4) && is_driver_chain(c, v) ? CLS_BUSDRIVER :
						!(c->no_fast & 32) && v.depth > 0 && is_fbdchain(c, v) ? CLS_FBDCHAIN : CLS_BUSGENERIC;
				// (the master bus at offset 0 is the root's alone)
				if(v.cls == CLS_BUSGENERIC || (v.out_off == 0) != (v.depth == 0))
					owners_ok = false;
				else if(v.depth == 0)
					root_driver = true;
				(v.cls == CLS_BUSDRIVER ? d.first : v.cls == CLS_FBDCHAIN ? fbd_bydepth[v.depth] : d.second).push_back((int)vi);
				maxdepth = std::max(maxdepth, v.depth);
			} else {
				v.cls = !(c->no_fast & 1) && is_oscpan_chain(c, v) ? CLS_OSCPAN :
						!(c->no_fast & 8) && is_osc2pan_chain(c, v) ? CLS_OSC2PAN :
						!(c->no_fast & 2) && is_oscfiltpan_chain(c, v) ? CLS_OSCFILTPAN :
						// (no quiet kernel of its own: k_leaf_recs renders it, records or not - unless an
						// oscillator leaves the mip-mapped waves somewhere in this batch)
						!(c->no_fast & 128) && !v.mode_mix && is_osc2filtpan_chain(c, v) ? CLS_OSC2FILTPAN :
						!(c->no_fast & 16) && is_fmpan_chain(c, v) ? CLS_FMPAN : CLS_GENERIC;
				if(v.out_off == 0)
					owners_ok = false;	// adds straight into the master bus
				(v.cls == CLS_OSCPAN ? fast_leaf : v.cls == CLS_OSC2PAN ? osc2_leaf :
				 v.cls == CLS_OSCFILTPAN ? filt_leaf : v.cls == CLS_FMPAN ? fm_leaf :
				 v.cls == CLS_OSC2FILTPAN ? o2f_leaf : gen_leaf).push_back((int)vi);
			}
		}
		auto by_bus = [&](int a, int b) { return c->voices[a].out_off < c->voices[b].out_off; };
		std::stable_sort(fast_leaf.begin(), fast_leaf.end(), by_bus);
		std::stable_sort(gen_leaf.begin(), gen_leaf.end(), by_bus);
		std::stable_sort(filt_leaf.begin(), filt_leaf.end(), by_bus);
		std::stable_sort(osc2_leaf.begin(), osc2_leaf.end(), by_bus);
		c->list_all = fast_leaf;
		c->n_fast_leaf = (int)fast_leaf.size();
		c->list_all.insert(c->list_all.end(), osc2_leaf.begin(), osc2_leaf.end());
		c->n_osc2_leaf = (int)osc2_leaf.size();
		c->list_all.insert(c->list_all.end(), filt_leaf.begin(), filt_leaf.end());
		c->n_filt_leaf = (int)filt_leaf.size();
		{
			// fm voices, grouped by unit kind: one launch per kind present
			std::stable_sort(fm_leaf.begin(), fm_leaf.end(), [&](int a, int b) {
				const int ka = c->units[c->voices[a].unit[0]].kind, kb = c->units[c->voices[b].unit[0]].kind;
				return ka != kb ? ka < kb : c->voices[a].out_off < c->voices[b].out_off;
			});
			for(int k = 0; k < 8; ++k)
				c->fm_kind_count[k] = 0;
			for(int vi : fm_leaf)
				++c->fm_kind_count[c->units[c->voices[vi].unit[0]].kind - A2AMD_FM1];
			c->list_all.insert(c->list_all.end(), fm_leaf.begin(), fm_leaf.end());
			c->n_fm_leaf = (int)fm_leaf.size();
		}
		c->list_all.insert(c->list_all.end(), gen_leaf.begin(), gen_leaf.end());
		c->n_leaf = (int)gen_leaf.size();
		std::stable_sort(o2f_leaf.begin(), o2f_leaf.end(), by_bus);
		c->list_all.insert(c->list_all.end(), o2f_leaf.begin(), o2f_leaf.end());
		c->n_o2f_leaf = (int)o2f_leaf.size();
		c->depth_ranges.assign(maxdepth + 1, DepthRange());
		for(int d = 0; d <= maxdepth; ++d) {
			auto &l = bydepth[d];
			DepthRange &r = c->depth_ranges[d];
			r.fast_first = (int)c->list_all.size();
			r.fast_count = (int)l.first.size();
			c->list_all.insert(c->list_all.end(), l.first.begin(), l.first.end());
			r.fbd_first = (int)c->list_all.size();
			r.fbd_count = (int)fbd_bydepth[d].size();
			c->list_all.insert(c->list_all.end(), fbd_bydepth[d].begin(), fbd_bydepth[d].end());
			r.gen_first = (int)c->list_all.size();
			r.gen_count = (int)l.second.size();
			c->list_all.insert(c->list_all.end(), l.second.begin(), l.second.end());
		}
		c->static_len = (int)c->list_all.size();
		if(int r = grow(c, c->d_list, c->list_all.size() + 64, 1, false)) return r;
		if(!c->list_all.empty())
			HIPCHK(c, hipMemcpyAsync(c->d_list.d, c->list_all.data(), c->list_all.size() * sizeof(int),
					hipMemcpyHostToDevice, c->stream));
		c->lists_dirty = false;
		c->owners_all_driver = owners_ok && root_driver;
	}
	std::vector<int> dyn_all;
	{
		// this batch's exceptions (shipped in the blob)
		std::vector<int> dyn_leaf;
		std::vector<std::vector<int>> dyn_bus(c->depth_ranges.size());
		for(int vi : c->with_recs) {
			const HVoice &v = c->voices[vi];
			// (fm-panmix voices execute their own records in k_leaf_fmpan)
			if(v.cls == CLS_OSCPAN || v.cls == CLS_OSCFILTPAN || v.cls == CLS_OSC2PAN)
				dyn_leaf.push_back(vi);
			else if((v.cls == CLS_BUSDRIVER || v.cls == CLS_FBDCHAIN) && v.depth < (int)dyn_bus.size()) {
				dyn_bus[v.depth].push_back(vi);
				static const int trace = getenv("A2AMD_HOSTTIMING") ? atoi(getenv("A2AMD_HOSTTIMING")) : 0;
				if(trace >= 3 && !v.recs.empty())
					fprintf(stderr, "a2amd: bus voice %d (depth %d, class %d) carries %zu records, first: frag %u op %u unit %u reg %u value %d\n",
							vi, v.depth, v.cls, v.recs.size(), A2D_RFRAG(v.recs[0].head), A2D_ROP(v.recs[0].head),
							A2D_RUNIT(v.recs[0].head), A2D_RREG(v.recs[0].head), v.recs[0].value);
			}
		}
		// Voices of the wtosc[+wtosc]->panmix classes whose records are what
		// k_leaf_recs executes (windows, writes, births, deaths; oscillators on
		// mip-mapped waves throughout the batch) go first, by class; the rest -
		// filter voices, a wave of another kind somewhere in the batch - to the
		// general kernel.
		std::vector<int> dyn_o1, dyn_o2, dyn_f1, dyn_rest;
		const bool no_recs_kernel = (c->no_fast & 64) != 0;
		for(int vi : dyn_leaf) {
			const HVoice &v = c->voices[vi];
			// (close_fragment's R_NOP is the one other record k_leaf_recs takes - as nothing)
			const bool ok = !no_recs_kernel && !v.mode_mix && !v.fancy_recs;
			(!ok ? dyn_rest : v.cls == CLS_OSCPAN ? dyn_o1 : v.cls == CLS_OSC2PAN ? dyn_o2 : dyn_f1).push_back(vi);
		}
		// (the walk order usually has them grouped by bus already)
		auto by_bus_dyn = [&](int a, int b) { return c->voices[a].out_off < c->voices[b].out_off; };
		for(std::vector<int> *l : { &dyn_o1, &dyn_o2, &dyn_f1, &dyn_rest })
			if(!std::is_sorted(l->begin(), l->end(), by_bus_dyn))
				std::stable_sort(l->begin(), l->end(), by_bus_dyn);
		std::vector<int> dyn = dyn_o1;
		dyn.insert(dyn.end(), dyn_o2.begin(), dyn_o2.end());
		dyn.insert(dyn.end(), dyn_f1.begin(), dyn_f1.end());
		dyn.insert(dyn.end(), dyn_rest.begin(), dyn_rest.end());
		c->n_dyn_osc1 = (int)dyn_o1.size();
		c->n_dyn_osc2 = (int)dyn_o2.size();
		c->n_dyn_filt = (int)dyn_f1.size();
		c->n_leaf_dyn = (int)dyn_leaf.size();
		for(size_t d = 0; d < dyn_bus.size(); ++d) {
			c->depth_ranges[d].dyn_first = (int)dyn.size();
			c->depth_ranges[d].dyn_count = (int)dyn_bus[d].size();
			dyn.insert(dyn.end(), dyn_bus[d].begin(), dyn_bus[d].end());
		}
		dyn_all.swap(dyn);
		c->consume_ok = c->owners_all_driver;
		for(const std::vector<int> &d : dyn_bus)
			if(!d.empty())
				c->consume_ok = false;	// a bus owner carries records: the general kernel renders it
	}

	A2DParams p;
	memset(&p, 0, sizeof(p));
	p.voices = c->d_voices.d;
	p.udesc = c->d_udesc.d;
	p.ustate = c->d_ustate.d;
	p.vactive = c->d_vactive.d;
	p.runs = c->d_runs.d;
	p.waves = c->d_waves.d;
	p.wavepool = c->d_wavepool.d;
	p.wavecoef = c->d_wavecoef.d;
	p.busmem = c->d_busmem.d;
	p.fbdmem = c->d_fbdmem.d;
	p.ptab = c->d_ptab;
	p.fmstate = c->d_fmstate.d;
	p.xio = c->d_xio.d;
	p.fmsine = c->d_fmsine;
	p.nfrags = c->nfrags;
	p.samplerate = c->cfg.samplerate;
	p.debug = getenv("A2AMD_DEBUG") ? atoi(getenv("A2AMD_DEBUG")) : 0;
	for(int f = 0, acc = 0; f < c->nfrags; ++f) {
		p.fragframes[f] = (uint8_t)c->fragframes[f];
		p.fragstart[f] = (uint16_t)acc;
		acc += (int)c->fragframes[f];
	}

	// the blob: [params | records | scatter indices | scatter runs | exception lists]
	auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t nsc = sc_idx.size();
	const size_t o_recs = up256(sizeof(A2DParams));
	const size_t o_idx = o_recs + up256((recs.size() + 1) * sizeof(A2DRec));
	const size_t o_val = o_idx + up256(nsc * sizeof(int));
	const size_t o_dyn = o_val + up256(nsc * sizeof(A2DRun));
	const size_t total = o_dyn + up256(dyn_all.size() * sizeof(int));
	if(int r = grow(c, c->d_blob, total, 1, false)) return r;
	const int bi = c->blob_i;
	c->blob_i ^= 1;
	if(c->blob_busy[bi]) {		// the copy that last read this staging buffer must have run
		HIPCHK(c, hipEventSynchronize(c->blob_ev[bi]));
		c->blob_busy[bi] = false;
	}
	if(total > c->h_blob_cap[bi]) {
		if(c->h_blob[bi])
			HIPCHK(c, hipHostFree(c->h_blob[bi]));
		c->h_blob_cap[bi] = std::max(total * 2, (size_t)65536);
		HIPCHK(c, hipHostMalloc((void **)&c->h_blob[bi], c->h_blob_cap[bi], hipHostMallocDefault));
	}
	if(!c->blob_ev[bi])
		HIPCHK(c, hipEventCreateWithFlags(&c->blob_ev[bi], hipEventDisableTiming));
	char *hb = c->h_blob[bi];
	p.recs = (const A2DRec *)(c->d_blob.d + o_recs);
	c->hparams = p;
	memcpy(hb, &p, sizeof(p));
	if(!recs.empty())
		memcpy(hb + o_recs, recs.data(), recs.size() * sizeof(A2DRec));
	if(nsc) {
		memcpy(hb + o_idx, sc_idx.data(), nsc * sizeof(int));
		memcpy(hb + o_val, sc_val.data(), nsc * sizeof(A2DRun));
	}
	if(!dyn_all.empty())
		memcpy(hb + o_dyn, dyn_all.data(), dyn_all.size() * sizeof(int));
	HIPCHK(c, hipMemcpyAsync(c->d_blob.d, hb, total, hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipEventRecord(c->blob_ev[bi], c->stream));
	c->blob_busy[bi] = true;
	c->d_params = (A2DParams *)c->d_blob.d;
	c->d_dyn = (const int *)(c->d_blob.d + o_dyn);
	if(nsc)
		if(a2d_launch_scatter_runs((const int *)(c->d_blob.d + o_idx), (const A2DRun *)(c->d_blob.d + o_val),
				(int)nsc, c->d_runs.d, c->stream))
			return c->fail(A2AMD_EHIP, "scatter launch failed");
	c->uploaded = true;
	c->blob_quiet = recs.empty() && dyn_all.empty();
	c->blob_nfrags = c->nfrags;
	memcpy(c->blob_frames, c->fragframes, (size_t)c->nfrags * sizeof(unsigned));
	return 0;
}

// Shape of the fast leaf launch, from sweeps on MI355X (DESIGN.md "Launch
// shape"): cut the batch into as many time slices as it has 8-fragment chunks
// (up to 8), then give a wavefront enough voices that about 4096 wavefronts
// (one resident round of 256 CUs x 16) share the work; at least 4 voices, so
// that their sum reaches the bus in one atomic instead of four, at most 32.
void pick_fast_shape(int n, int nfrags, int *vpw, int *ysplit)
{
	const int nchunks = (nfrags + A2D_FAST_FCH - 1) / A2D_FAST_FCH;
	int y = getenv("A2AMD_YSPLIT") ? atoi(getenv("A2AMD_YSPLIT")) : 32;
	y = std::min(std::max(y, 1), nchunks);
	int v = getenv("A2AMD_VPW") ? atoi(getenv("A2AMD_VPW")) :
			std::min(std::max((int)(((long long)n * y + 4095) / 4096), 4), 32);
	*vpw = std::min(std::max(v, 1), 64);
	*ysplit = y;
}

int pick_fast_vpw(int n)
{
	// enough wavefronts to fill 256 CUs x 4 SIMDs several times over, then
	// more voices per wavefront (fewer, fatter bus updates)
	if(getenv("A2AMD_VPW"))
		return std::min(std::max(atoi(getenv("A2AMD_VPW")), 1), 64);
	int v = (n + 4095) / 4096;
	return std::min(std::max(v, 1), 64);
}

int launch_depth(a2amd_ctx *c, int d, int consume, A2DCommitSet *pend)	// consume: 1 zero what is read, 2 root stores the master bus
{
	const DepthRange &r = c->depth_ranges[d];
	if(r.fast_count) {
		// (state commits the time-sliced leaf kernels left behind ride along)
		if(a2d_launch_bus_driver(c->d_params, c->d_list.d + r.fast_first, r.fast_count, c->nfrags, consume,
				pend, c->stream))
			return c->fail(A2AMD_EHIP, "bus driver launch failed: %s", hipGetErrorString(hipGetLastError()));
		pend->n = 0;
		++c->stats.launches;
	}
	if(r.fbd_count) {
		if(a2d_launch_bus_fbdchain(c->d_params, c->d_list.d + r.fbd_first, r.fbd_count, consume & 1, c->stream))
			return c->fail(A2AMD_EHIP, "delay chain launch failed: %s", hipGetErrorString(hipGetLastError()));
		++c->stats.launches;
	}
	if(r.gen_count) {
		if(a2d_launch_voices(c->d_params, c->d_list.d + r.gen_first, r.gen_count, 1, c->stream))
			return c->fail(A2AMD_EHIP, "bus launch failed: %s", hipGetErrorString(hipGetLastError()));
		++c->stats.launches;
	}
	if(r.dyn_count) {
		if(a2d_launch_voices(c->d_params, c->d_dyn + r.dyn_first, r.dyn_count, 1, c->stream))
			return c->fail(A2AMD_EHIP, "bus launch failed: %s", hipGetErrorString(hipGetLastError()));
		++c->stats.launches;
	}
	return 0;
}

int pick_vpw(int n)
{
	int v = n / 4096;
	return std::min(std::max(v, 1), (int)A2D_MAXVPW);
}

// give a region of the wave pool back: kept sorted by offset, neighbours merged
void wavepool_release(a2amd_ctx *c, size_t off, size_t len)
{
	if(!len)
		return;
	auto &fl = c->wavepool_free;
	auto it = std::lower_bound(fl.begin(), fl.end(), std::make_pair(off, (size_t)0));
	it = fl.insert(it, std::make_pair(off, len));
	if(it + 1 != fl.end() && it->first + it->second == (it + 1)->first) {
		it->second += (it + 1)->second;
		it = fl.erase(it + 1) - 1;
	}
	if(it != fl.begin() && (it - 1)->first + (it - 1)->second == it->first) {
		(it - 1)->second += it->second;
		it = fl.erase(it) - 1;
	}
	// the tail of the pool grows back into unused space
	if(it->first + it->second == c->wavepool_used) {
		c->wavepool_used = it->first;
		fl.erase(it);
	}
}

void end_batch(a2amd_ctx *c)
{
	// (graphs survive: upload() drops them unless the next batch is the same quiet one)
	// Records made after the last fragment of the batch was closed belong to
	// the first fragment of the next batch: carry them over.
	const int done = c->nfrags;
	c->serial_base += done;
	std::vector<int> carry;
	c->prev_with_recs.clear();
	for(int vi : c->with_recs) {
		HVoice &v = c->voices[vi];
		c->prev_with_recs.push_back(vi);
		size_t keep = 0;
		// (a voice that was set up but not walked yet keeps everything)
		const bool unborn = v.live && !v.resolved;
		// (the records are in fragment order: nothing to carry over unless the last one is)
		if(unborn || (!v.recs.empty() && (int)A2D_RFRAG(v.recs.back().head) >= done))
			for(size_t i = 0; i < v.recs.size(); ++i)
				if(unborn || (int)A2D_RFRAG(v.recs[i].head) >= done) {
					A2DRec r = v.recs[i];
					int f = (int)A2D_RFRAG(r.head) - done;
					r.head = (r.head & 0xffff0000u) | (uint32_t)(f < 0 ? 0 : f);
					v.recs[keep++] = r;
				}
		v.recs.resize(keep);
		v.frag_mark = 0;
		if(keep) {
			v.touched = c->serial_base;
			carry.push_back(vi);
		} else {
			v.touched = -1;
			v.listed_recs = false;
			if(v.mode_mix)
				c->lists_dirty = true;	// (it may have its leaf class back)
			v.mode_mix = false;
			v.fancy_recs = false;
		}
	}
	c->with_recs = carry;
	for(int vi : c->deferred_free_voices) {
		c->voices[vi] = HVoice();
		c->free_voices.push_back(vi);
		c->lists_dirty = true;
	}
	c->deferred_free_voices.clear();
	for(int ui : c->deferred_free_units)
		c->free_units.push_back(ui);
	c->deferred_free_units.clear();
	for(auto &b : c->deferred_bus_free)
		c->bus_free[b.second].push_back(b.first);
	c->deferred_bus_free.clear();
	for(int b : c->fbd_deferred_free)
		c->fbd_free.push_back(b);
	c->fbd_deferred_free.clear();
	for(int b : c->fm_deferred_free)
		c->fm_free.push_back(b);
	c->fm_deferred_free.clear();
	for(int b : c->xio_deferred_free) {
		c->xio[b].unit = -1;
		c->xio_free.push_back(b);
	}
	c->xio_deferred_free.clear();
	for(auto &r : c->deferred_wavepool_free)
		wavepool_release(c, r.first, r.second);
	c->deferred_wavepool_free.clear();
	for(int w : c->deferred_wave_slots)
		c->free_wave_slots.push_back(w);
	c->deferred_wave_slots.clear();
	c->nfrags = 0;
	c->cur_frag = 0;
	c->frag_open = false;
	c->uploaded = false;
	c->sub_resume = -1;
	c->paused_at = 0;
}

// does a voice at nesting depth d hold an xinsert in A2AMD_XIO_MUTE mode (insert clients)?
bool depth_has_mutes(const a2amd_ctx *c, int d)
{
	for(const XioSlot &x : c->xio)
		if(x.last_unit >= 0 && x.last_unit < (int)c->units.size() && (c->units[x.last_unit].xio_mode & A2AMD_XIO_MUTE) &&
				c->units[x.last_unit].voice >= 0 && c->voices[c->units[x.last_unit].voice].depth == d)
			return true;
	return false;
}

// the kernels of one batch, in stream order; e* may be null
int issue_kernels(a2amd_ctx *c, unsigned phases, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2)
{
	// (self-cleaning buses need both phases in one go: the root's bus is read in ROOT)
	const bool has_sub = (phases & A2AMD_RENDER_SUBTREES) != 0, has_root = (phases & A2AMD_RENDER_ROOT) != 0;
	const bool consume = c->consume_ok && has_sub && has_root;
	// Split phases (multi-GPU steps): the group buses are read - and zeroed - in the
	// SUBTREES phase; the root's bus by the ROOT phase on the rank that runs it, or
	// by a2amd_rootbus_copy() when the partials are parked (shard.GroupedRootReduce).
	const bool consume_sub = c->consume_ok && has_sub && !has_root;
	A2DCommitSet pend;
	pend.n = 0;
	pend.c[0].nlist = pend.c[1].nlist = 0;
	auto flush_commits = [&]() {
		for(int k = 0; k < pend.n; ++k)
			a2d_launch_commit(c->hparams, pend.c[k], c->stream);
		pend.n = 0;
	};
	if(phases & A2AMD_RENDER_SUBTREES) {
		if(c->sub_resume < 0) {
		// (a graph of a self-cleaning batch holds no memset: whoever launches it
		// clears the buses first if they are not known to be clean, ensure_clean())
		const bool selfclean = consume || consume_sub;
		if(c->capturing ? !selfclean : !(selfclean && c->others_clean && c->root_clean))
			HIPCHK(c, hipMemsetAsync(c->d_busmem.d, 0, c->bus_used * sizeof(int32_t), c->stream));
		c->others_clean = selfclean;
		c->root_clean = consume;
		if(e0)
			HIPCHK(c, hipEventRecord(e0, c->stream));
		if(c->n_fast_leaf) {
			int vpw, ysplit;
			pick_fast_shape(c->n_fast_leaf, c->nfrags, &vpw, &ysplit);
			// e1 right behind the main kernel when it is the only leaf kernel
			// of the batch: "leaf" time is then that kernel alone
			const bool solo = !c->n_osc2_leaf && !c->n_filt_leaf && !c->n_fm_leaf && !c->n_leaf && !c->n_leaf_dyn && !c->n_o2f_leaf;
			if(a2d_launch_leaf_oscpan(c->d_params, c->hparams, c->d_list.d, c->n_fast_leaf,
					vpw, ysplit, c->d_ustage.d, c->stream, solo ? (void *)e1 : nullptr, &pend.c[pend.n]))
				return c->fail(A2AMD_EHIP, "fast leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
			if(pend.c[pend.n].nlist)
				++pend.n;
			++c->stats.launches;
		}
		if(c->n_osc2_leaf) {
			int vpw, ysplit;
			pick_fast_shape(c->n_osc2_leaf, c->nfrags * A2D_FAST_FCH / A2D_OSC2_FCH, &vpw, &ysplit);	// (its own chunk length)
			if(a2d_launch_leaf_osc2pan(c->d_params, c->hparams, c->d_list.d + c->n_fast_leaf, c->n_osc2_leaf,
					vpw, ysplit, c->d_ustage.d, c->stream, &pend.c[pend.n]))
				return c->fail(A2AMD_EHIP, "2-osc leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
			if(pend.c[pend.n].nlist)
				++pend.n;
			++c->stats.launches;
		}
		if(c->n_filt_leaf) {
			// voices per workgroup = lanes of its filter wavefront: all 64 once there
			// are enough voices for a workgroup on every CU, else spread out (a
			// workgroup takes as long as its filter chain, whatever its voice count)
			const int nf = c->n_filt_leaf;
			int vpw = getenv("A2AMD_FVPW") ? atoi(getenv("A2AMD_FVPW")) :
					std::min(std::max((nf + 511) / 512, 1), 32);
			if(a2d_launch_leaf_oscfiltpan(c->d_params, c->hparams, c->d_list.d + c->n_fast_leaf + c->n_osc2_leaf,
					c->n_filt_leaf, vpw, c->stream))
				return c->fail(A2AMD_EHIP, "filter leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
			++c->stats.launches;
		}
		int fm_kinds = 0;
		for(int k = 0; k < 8; ++k)
			fm_kinds += c->fm_kind_count[k] != 0;
		if(fm_kinds > 1 && c->n_fm_leaf <= 16384 && !getenv("A2AMD_FMVPW")) {
			// several kinds, few voices: one launch for all of them (the
			// per-kind launches below would run back to back, each as long
			// as a voice's serial chain)
			if(a2d_launch_leaf_fmpan_all(c->d_params, c->hparams, c->d_list.d + c->n_fast_leaf + c->n_osc2_leaf +
					c->n_filt_leaf, c->fm_kind_count, (c->n_fm_leaf + 1023) / 1024, c->stream))
				return c->fail(A2AMD_EHIP, "fm leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
			++c->stats.launches;
		} else
		for(int k = 0, at = c->n_fast_leaf + c->n_osc2_leaf + c->n_filt_leaf; k < 8; at += c->fm_kind_count[k++]) {
			const int n = c->fm_kind_count[k];
			if(!n)
				continue;
			// A voice is a serial recurrence: a launch takes as long as its longest
			// lane, so few voices are spread over many wavefronts (idle lanes of a
			// wavefront shadow its voices, see fmpan_body) until there is one
			// wavefront per SIMD (1 024), then the lanes fill up
			// (profiles/r01_fm_vpw_sweep.txt).
			int vpw = getenv("A2AMD_FMVPW") ? atoi(getenv("A2AMD_FMVPW")) : (n + 1023) / 1024;
			vpw = std::min(std::max(vpw, 1), 64);
			if(a2d_launch_leaf_fmpan(c->d_params, c->hparams, A2AMD_FM1 + k, c->d_list.d + at, n, vpw, c->stream))
				return c->fail(A2AMD_EHIP, "fm leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
			++c->stats.launches;
		}
		if(c->n_leaf) {
			if(a2d_launch_voices(c->d_params, c->d_list.d + c->n_fast_leaf + c->n_osc2_leaf + c->n_filt_leaf +
					c->n_fm_leaf, c->n_leaf,
					pick_vpw(c->n_leaf), c->stream))
				return c->fail(A2AMD_EHIP, "leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
			++c->stats.launches;
		}
		{
			// (a wavefront walks its voices one after the other, fragment by fragment: as many
			// wavefronts as the chip holds before a wavefront gets a second voice)
			auto recs = [&](int nosc, int filt, const int *list, int n) -> int {
				if(!n)
					return 0;
				int vpw = getenv("A2AMD_RVPW") ? atoi(getenv("A2AMD_RVPW")) : (n + 8191) / 8192;
				if(a2d_launch_leaf_recs(c->d_params, c->hparams, nosc, filt, list, n, vpw, c->stream))
					return c->fail(A2AMD_EHIP, "leaf records launch failed: %s", hipGetErrorString(hipGetLastError()));
				++c->stats.launches;
				return 0;
			};
			// the 2 x wtosc-filter12-panmix leaves, with and without records, and - of the classes that
			// have quiet kernels of their own - this batch's voices with records
			const int *lists[4] = { c->d_dyn, c->d_dyn + c->n_dyn_osc1, c->d_dyn + c->n_dyn_osc1 + c->n_dyn_osc2,
					c->d_list.d + c->n_fast_leaf + c->n_osc2_leaf + c->n_filt_leaf + c->n_fm_leaf + c->n_leaf };
			const int counts[4] = { c->n_dyn_osc1, c->n_dyn_osc2, c->n_dyn_filt, c->n_o2f_leaf };
			const int total = counts[0] + counts[1] + counts[2] + counts[3];
			const int kinds = (counts[0] != 0) + (counts[1] != 0) + (counts[2] != 0) + (counts[3] != 0);
			if(kinds > 1 && total <= 4096 && !getenv("A2AMD_RVPW")) {
				// few voices of several kinds (a song): one launch - on one stream the per-kind
				// launches would run back to back, each as long as one voice's walk through the batch
				if(a2d_launch_leaf_recs_all(c->d_params, c->hparams, lists, counts, 1, c->stream))
					return c->fail(A2AMD_EHIP, "leaf records launch failed: %s", hipGetErrorString(hipGetLastError()));
				++c->stats.launches;
			} else {
				static const int nosc[4] = { 1, 2, 1, 2 }, filt[4] = { 0, 0, 1, 1 };
				for(int k = 0; k < 4; ++k)
					if(int r = recs(nosc[k], filt[k], lists[k], counts[k]))
						return r;
			}
		}
		if(c->n_leaf_dyn - c->n_dyn_osc1 - c->n_dyn_osc2 - c->n_dyn_filt > 0) {
			const int n = c->n_leaf_dyn - c->n_dyn_osc1 - c->n_dyn_osc2 - c->n_dyn_filt;
			if(a2d_launch_voices(c->d_params, c->d_dyn + c->n_dyn_osc1 + c->n_dyn_osc2 + c->n_dyn_filt, n, pick_vpw(n), c->stream))
				return c->fail(A2AMD_EHIP, "leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
			++c->stats.launches;
		}
		if(e1 && !(c->n_fast_leaf && !c->n_osc2_leaf && !c->n_filt_leaf && !c->n_fm_leaf && !c->n_leaf && !c->n_leaf_dyn && !c->n_o2f_leaf))
			HIPCHK(c, hipEventRecord(e1, c->stream));
		}	// (fresh start)
		// the voices that own a bus, deepest first.  With A2AMD_RENDER_TAPS the walk stops behind a
		// depth that holds a muted xinsert (insert clients: a2amd_unit_insertable): the host serves
		// them and calls again.
		const int dstart = c->sub_resume >= 0 ? c->sub_resume : (int)c->depth_ranges.size() - 1;
		c->sub_resume = -1;
		c->paused_at = 0;
		for(int d = dstart; d >= 1; --d) {
			if(int r = launch_depth(c, d, consume ? 3 : consume_sub ? 1 : 0, &pend))
				return r;
			if((phases & A2AMD_RENDER_TAPS) && depth_has_mutes(c, d)) {
				c->paused_at = d;
				c->sub_resume = d > 1 ? d - 1 : -1;
				break;
			}
		}
		// (the ROOT phase may run elsewhere, or later: nothing stays pending across calls)
		if(!(phases & A2AMD_RENDER_ROOT))
			flush_commits();
	}
	if(phases & A2AMD_RENDER_ROOT) {
		// The root chain adds into the master bus.  When its phase runs on its own
		// (multi-GPU steps: several SUBTREES phases may have gone by since the
		// last one, audiality2_amd/shard.py) the master bus is cleared here.
		// (... unless the root is a plain driver chain: then it stores the master bus)
		const bool root_stores = consume || c->consume_ok;
		if(!has_sub && !root_stores)
			HIPCHK(c, hipMemsetAsync(c->d_busmem.d, 0,
					(size_t)c->nfrags * c->cfg.channels * A2D_FRAG * sizeof(int32_t), c->stream));
		// (on its own the phase also leaves the root's bus zeroed behind it)
		if(!c->depth_ranges.empty())
			if(int r = launch_depth(c, 0, root_stores ? 3 : 0, &pend))
				return r;
		if(!has_sub && root_stores)
			c->root_clean = true;
		flush_commits();
		if(e2)
			HIPCHK(c, hipEventRecord(e2, c->stream));
		c->stats.fragments += c->nfrags;
		c->stats.voice_fragments += (uint64_t)c->nfrags * (uint64_t)(c->list_all.size() - c->n_list_pads);
	}
	return 0;
}

void drop_graphs(a2amd_ctx *c)
{
	for(int i = 0; i < 4; ++i) {
		if(c->gexec[i]) {
			hipGraphExecDestroy(c->gexec[i]);
			c->gexec[i] = nullptr;
		}
		if(c->graph[i]) {
			hipGraphDestroy(c->graph[i]);
			c->graph[i] = nullptr;
		}
	}
}

// before a graph of a self-cleaning batch (it holds no memset) is launched
int ensure_clean(a2amd_ctx *c)
{
	if(c->consume_ok && !(c->others_clean && c->root_clean)) {
		HIPCHK(c, hipMemsetAsync(c->d_busmem.d, 0, c->bus_used * sizeof(int32_t), c->stream));
		c->others_clean = c->root_clean = true;
	}
	return 0;
}

// capture 'steps' consecutive runs of the uploaded batch into one graph
int build_graph(a2amd_ctx *c, int slot, int steps, unsigned phases = A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT)
{
	hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
	if(e != hipSuccess)
		return c->fail(A2AMD_EHIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
	int r = 0;
	// (a captured run does not happen now: every graph starts from buses of
	// unknown state, and what it leaves behind is noted when it is launched)
	const bool oc = c->others_clean, rc = c->root_clean;
	c->capturing = true;
	for(int i = 0; i < steps && !r; ++i)
		r = issue_kernels(c, phases, nullptr, nullptr, nullptr);
	c->capturing = false;
	c->others_clean = oc;
	c->root_clean = rc;
	e = hipStreamEndCapture(c->stream, &c->graph[slot]);
	if(r)
		return r;
	if(e != hipSuccess)
		return c->fail(A2AMD_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
	e = hipGraphInstantiate(&c->gexec[slot], c->graph[slot], nullptr, nullptr, 0);
	if(e != hipSuccess)
		return c->fail(A2AMD_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
	return 0;
}

} // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char *a2amd_version(void) { return "a2amd 0.1 (gfx950)"; }

const char *a2amd_last_error(const a2amd_ctx *c) { return c ? c->err : g_err; }

int a2amd_device_count(void)
{
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// The current HIP device is per host thread; a context may be driven from a
// thread other than the one that opened it, and contexts of one process may
// live on different GPUs (one engine state per GPU).
static inline void use_device(const a2amd_ctx *c) { (void)hipSetDevice(c->cfg.device); }

int a2amd_open(const a2amd_config *cfg, a2amd_ctx **out)
{
	if(!cfg || !out || cfg->channels < 1 || cfg->channels > A2D_MAXCH || cfg->samplerate <= 0) {
		snprintf(g_err, sizeof(g_err), "a2amd_open: bad configuration");
		return A2AMD_EINVAL;
	}
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if(e != hipSuccess || ndev <= 0 || cfg->device >= ndev) {
		snprintf(g_err, sizeof(g_err), "a2amd_open: no usable HIP device (%s, %d devices, want #%d); "
				"this library has no CPU fallback", hipGetErrorString(e), ndev, cfg->device);
		return A2AMD_ENODEVICE;
	}
	a2amd_ctx *c = new a2amd_ctx();
	c->cfg = *cfg;
	c->err[0] = 0;
	if(!c->cfg.max_batch)
		c->cfg.max_batch = 1;
	if(c->cfg.max_batch > A2D_MAXBATCH)
		c->cfg.max_batch = A2D_MAXBATCH;
	memset(&c->stats, 0, sizeof(c->stats));
	build_pitch_table(c->ptab);
	c->no_fast = getenv("A2AMD_NO_FAST") ? atoi(getenv("A2AMD_NO_FAST")) : 0;
	c->hosttiming = getenv("A2AMD_HOSTTIMING") != nullptr;
	c->bus_stride_frames = (size_t)c->cfg.max_batch * A2D_FRAG;
	c->bus_used = c->bus_stride_frames * (size_t)c->cfg.channels;	// master bus at offset 0
#define OPENCHK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) { \
	snprintf(g_err, sizeof(g_err), "a2amd_open: %s: %s", #call, hipGetErrorString(e_)); \
	delete c; return A2AMD_EHIP; } } while(0)
	OPENCHK(hipSetDevice(cfg->device));
	if(cfg->stream)
		c->stream = (hipStream_t)cfg->stream;
	else {
		OPENCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		c->own_stream = true;
	}
	OPENCHK(hipEventCreate(&c->ev0));
	OPENCHK(hipEventCreate(&c->ev1));
	OPENCHK(hipEventCreate(&c->ev2));
	c->ev_pool.push_back(c->ev0);
	c->ev_pool.push_back(c->ev1);
	c->ev_pool.push_back(c->ev2);
	OPENCHK(hipMalloc((void **)&c->d_ptab, sizeof(c->ptab)));
	OPENCHK(hipMalloc((void **)&c->d_wavepool.d, (size_t)(8u << 20) * sizeof(int16_t)));
	c->d_wavepool.cap = 8u << 20;
	OPENCHK(hipMalloc((void **)&c->d_wavecoef.d, (size_t)(8u << 20) * 3 * sizeof(int32_t)));
	c->d_wavecoef.cap = 8u << 20;
#undef OPENCHK
	*out = c;
	return A2AMD_OK;
}

void a2amd_close(a2amd_ctx *c)
{
	if(!c)
		return;
	use_device(c);
	hipStreamSynchronize(c->stream);
	drop_graphs(c);
	if(c->comm && g_rccl.CommDestroy)
		g_rccl.CommDestroy(c->comm);
	if(c->grp_ev)
		hipEventDestroy(c->grp_ev);
	hipFree(c->d_voices.d); hipFree(c->d_udesc.d); hipFree(c->d_ustate.d); hipFree(c->d_ustage.d);
	hipFree(c->d_vactive.d); hipFree(c->d_runs.d); hipFree(c->d_recs.d);
	hipFree(c->d_waves.d); hipFree(c->d_wavepool.d); hipFree(c->d_wavecoef.d); hipFree(c->d_busmem.d);
	hipFree(c->d_fbdmem.d); hipFree(c->d_fmstate.d); hipFree(c->d_xio.d); hipFree(c->d_fmsine); hipFree(c->d_list.d); hipFree(c->d_scatter.d); hipFree(c->d_ptab); hipFree(c->d_blob.d);
	for(int k = 0; k < 2; ++k) { if(c->h_blob[k]) hipHostFree(c->h_blob[k]); if(c->blob_ev[k]) hipEventDestroy(c->blob_ev[k]); }
	if(c->h_master)
		hipHostFree(c->h_master);
	for(int k = 0; k < 2; ++k) { if(c->rb[k].h) hipHostFree(c->rb[k].h); if(c->rb[k].ev) hipEventDestroy(c->rb[k].ev); }
	for(hipEvent_t e : c->ev_pool)
		hipEventDestroy(e);
	if(c->own_stream)
		hipStreamDestroy(c->stream);
	delete c;
}

int a2amd_set_pitch_table(a2amd_ctx *c, const uint32_t *t)
{
	memcpy(c->ptab, t, sizeof(c->ptab));
	c->ptab_dirty = true;
	return A2AMD_OK;
}

int a2amd_get_pitch_table(const a2amd_ctx *c, uint32_t *t)
{
	memcpy(t, c->ptab, sizeof(c->ptab));
	return A2AMD_OK;
}

// ---- waves ------------------------------------------------------------------
int a2amd_wave_upload(a2amd_ctx *c, uint64_t key, const a2amd_wavedesc *w)
{
	use_device(c);
	if(!w)
		return c->fail(A2AMD_EINVAL, "wave_upload: null descriptor");
	int id = -1;
	for(size_t i = 0; i < c->waves.size(); ++i)
		if(c->waves[i].live && c->waves[i].key == key)
			id = (int)i;
	if(id < 0) {
		if(!c->free_wave_slots.empty()) {
			id = c->free_wave_slots.back();
			c->free_wave_slots.pop_back();
		} else {
			id = (int)c->waves.size();
			c->waves.push_back(HWave());
			c->mwaves.push_back(A2DWave());
		}
	} else {
		// same key again: the old data is replaced (what is recorded plays the old)
		if(c->waves[id].pool_len)
			c->deferred_wavepool_free.push_back(std::make_pair(c->waves[id].pool_off, c->waves[id].pool_len));
		--c->stats.live_waves;
	}
	HWave &hw = c->waves[id];
	hw.live = true;
	hw.key = key;
	memset(&hw.dw, 0, sizeof(hw.dw));
	hw.dw.type = w->type;
	hw.dw.flags = w->flags;
	hw.dw.period = w->period;
	int levels = w->type == A2AMD_WMIPWAVE ? A2D_MIPS : w->type == A2AMD_WWAVE ? 1 : 0;
	size_t total = 0;
	for(int l = 0; l < levels; ++l)
		total += A2AMD_WAVEPRE + (size_t)w->size[l] + A2AMD_WAVEPOST;
	total = (total + 7) & ~(size_t)7;
	// a region a dropped wave left behind (first fit), else the end of the pool
	size_t pos = (size_t)-1;
	for(size_t k = 0; k < c->wavepool_free.size() && total; ++k)
		if(c->wavepool_free[k].second >= total) {
			pos = c->wavepool_free[k].first;
			c->wavepool_free[k].first += total;
			c->wavepool_free[k].second -= total;
			if(!c->wavepool_free[k].second)
				c->wavepool_free.erase(c->wavepool_free.begin() + (long)k);
			// work in flight may still read the old contents
			HIPCHK(c, hipStreamSynchronize(c->stream));
			break;
		}
	if(pos == (size_t)-1) {
		if(c->wavepool_used + total > c->d_wavepool.cap) {
			if(int r = grow(c, c->d_wavepool, c->wavepool_used + total, 1, true)) return r;
			// the coefficient table follows the pool: rebuilt for what is in it
			if(int r = grow(c, c->d_wavecoef, c->d_wavepool.cap, 3, false)) return r;
			if(c->wavepool_used > 3 && a2d_launch_build_coef(c->d_wavepool.d, c->d_wavecoef.d, 1,
					(unsigned)c->wavepool_used - 2, c->stream))
				return c->fail(A2AMD_EHIP, "coefficient build failed");
			c->waves_dirty = true;
			drop_graphs(c);		// (the kernels take the table's address as an argument)
			c->blob_quiet = false;
		}
		pos = c->wavepool_used;
		c->wavepool_used += total;
	}
	hw.pool_off = pos;
	hw.pool_len = total;
	for(int l = 0; l < levels; ++l) {
		size_t n = A2AMD_WAVEPRE + (size_t)w->size[l] + A2AMD_WAVEPOST;
		HIPCHK(c, hipMemcpy(c->d_wavepool.d + pos, w->data[l], n * sizeof(int16_t), hipMemcpyHostToDevice));
		hw.dw.size[l] = w->size[l];
		hw.dw.off[l] = (uint32_t)(pos + A2AMD_WAVEPRE);
		pos += n;
	}
	// Hermite coefficients of every window of the region (a2amd_fast.hip: k_build_coef)
	if(total > 3 && a2d_launch_build_coef(c->d_wavepool.d, c->d_wavecoef.d, (unsigned)hw.pool_off + 1,
			(unsigned)(hw.pool_off + total) - 2, c->stream))
		return c->fail(A2AMD_EHIP, "coefficient build failed");
	c->mwaves[id] = hw.dw;
	c->waves_dirty = true;
	++c->stats.live_waves;
	return id;
}

int a2amd_wave_drop(a2amd_ctx *c, uint64_t key)
{
	for(size_t i = 0; i < c->waves.size(); ++i)
		if(c->waves[i].live && c->waves[i].key == key) {
			c->waves[i].live = false;
			c->waves[i].dw.size[0] = 0;	// "unloaded", waves.c:717-723
			if(c->waves[i].pool_len)
				c->deferred_wavepool_free.push_back(std::make_pair(c->waves[i].pool_off, c->waves[i].pool_len));
			c->waves[i].pool_len = 0;
			c->waves[i].key = 0;
			c->deferred_wave_slots.push_back((int)i);
			if(!c->nfrags && !c->frag_open) {
				// nothing recorded that could play it: free at once
				for(auto &r : c->deferred_wavepool_free)
					wavepool_release(c, r.first, r.second);
				c->deferred_wavepool_free.clear();
			}
			c->mwaves[i] = c->waves[i].dw;
			c->waves_dirty = true;
			--c->stats.live_waves;
			return A2AMD_OK;
		}
	return c->fail(A2AMD_EINVAL, "wave_drop: unknown key");
}

// ---- fragment clock -----------------------------------------------------------
int a2amd_fragment(a2amd_ctx *c, unsigned frames)
{
	if(!frames || frames > A2D_FRAG)
		return c->fail(A2AMD_EINVAL, "fragment of %u frames", frames);
	if(!c->stack.empty())
		return c->fail(A2AMD_ESTATE, "fragment inside an inline window");
	if(c->uploaded)
		return c->fail(A2AMD_ESTATE, "batch already uploaded; finish the render first");
	close_fragment(c);
	if(c->nfrags >= (int)c->cfg.max_batch)
		return c->fail(A2AMD_ESTATE, "more than max_batch=%u fragments without a render", c->cfg.max_batch);
	c->cur_frag = c->nfrags++;
	c->fragframes[c->cur_frag] = frames;
	c->frag_open = true;
	c->walked_started = 0;
	c->building = -1;
	if(c->defmap_dirty || c->defmap.size() < c->voices.size() + 4096) {
		// (the map only moves here, at a fragment boundary: the host holds its
		// address for the length of a fragment)
		if(c->defmap.size() < c->voices.size() + 4096)
			c->defmap.assign(c->voices.size() * 2 + 65536, 0);
		else
			std::fill(c->defmap.begin(), c->defmap.begin() + std::min(c->defmap.size(), c->voices.size()), 0);
		c->defmap_dirty = false;
	}
	c->defmap_used = false;
	return A2AMD_OK;
}

int a2amd_fragment_repeat(a2amd_ctx *c, unsigned frames, unsigned count)
{
	if(c->n_noise || c->n_cutoff_ramps)
		return c->fail(A2AMD_EUNSUPPORTED, "fragment_repeat with %d noise oscillators / %d cutoff ramps "
				"in flight", c->n_noise, c->n_cutoff_ramps);
	if(c->n_clients)
		return c->fail(A2AMD_EUNSUPPORTED, "fragment_repeat with clients on %d xinsert / xsink / xsource "
				"unit(s): their callbacks need every window", c->n_clients);
	for(unsigned i = 0; i < count; ++i) {
		if(int r = a2amd_fragment(c, frames))
			return r;
		// every live voice gets the default window: nothing to record, but
		// tell close_fragment() that nobody was skipped
		c->frag_open = false;
		c->walk_time += frames;
		c->prev_frames = frames;
	}
	return A2AMD_OK;
}

// ---- units ----------------------------------------------------------------------
int a2amd_unit_init(a2amd_ctx *c, uint64_t key, int kind, unsigned flags, int nin, int nout,
		int wired, int transpose, unsigned wakefrac)
{
	if(kind < 0 || kind >= A2AMD_NKINDS)
		return c->fail(A2AMD_EINVAL, "unit kind %d", kind);
	if(nin < 0 || nin > A2D_MAXCH || nout < 0 || nout > A2D_MAXCH)
		return c->fail(A2AMD_EINVAL, "bad channel counts %d->%d", nin, nout);
	const bool add = (flags & A2AMD_PROCADD) != 0;
	switch(kind) {
	  case A2AMD_WTOSC:
		if(nout != 1) return c->fail(A2AMD_EINVAL, "wtosc has 1 output");
		break;
	  case A2AMD_PANMIX:
	  case A2AMD_FBDELAY:
		if(nin < 1 || nin > 2 || nout < 1 || nout > 2)
			return c->fail(A2AMD_EINVAL, "unit kind %d with %d->%d channels", kind, nin, nout);
		break;
	  case A2AMD_FILTER12:
		if(nin != nout || nin < 1 || nin > 2)
			return c->fail(A2AMD_EINVAL, "filter12 %d->%d", nin, nout);
		break;
	  case A2AMD_XINSERT:
		if(nin != nout || nin < 1)
			return c->fail(A2AMD_EINVAL, "xinsert %d->%d", nin, nout);
		break;
	  case A2AMD_XSINK:
		if(nin < 1 || nout != 0)
			return c->fail(A2AMD_EINVAL, "xsink %d->%d", nin, nout);
		break;
	  case A2AMD_XSOURCE:
		if(nin != 0 || nout < 1)
			return c->fail(A2AMD_EINVAL, "xsource %d->%d", nin, nout);
		break;
	  case A2AMD_INLINE:
		if(nout < 1)
			return c->fail(A2AMD_EINVAL, "inline needs outputs");
		break;
	  case A2AMD_DC:	// dc.c:262-281: a source with one or two outputs
		if(nin != 0 || nout < 1 || nout > 2)
			return c->fail(A2AMD_EINVAL, "dc %d->%d", nin, nout);
		break;
	  case A2AMD_WAVESHAPER:
	  case A2AMD_DCBLOCK:
	  case A2AMD_LIMITER:	// A2_MATCHIO, one or two channels
		if(nin != nout || nin < 1 || nin > 2)
			return c->fail(A2AMD_EINVAL, "unit kind %d with %d->%d channels", kind, nin, nout);
		break;
	  default:	// fm1..fm4r, fm.c:532-834: no inputs, one output
		if(nin != 0 || nout != 1)
			return c->fail(A2AMD_EINVAL, "fm unit with %d->%d channels", nin, nout);
		break;
	}
	if(wired && !add && kind != A2AMD_INLINE)
		return c->fail(A2AMD_EUNSUPPORTED, "replacing-mode unit wired to the voice output bus");

	// voice under construction? (a2_PopulateVoice, core.c:350-420)
	int vi = c->building;
	if(vi < 0 || c->voices[vi].key != key) {
		if(!c->free_voices.empty()) {
			vi = c->free_voices.back();
			c->free_voices.pop_back();
		} else {
			vi = (int)c->voices.size();
			c->voices.push_back(HVoice());
		}
		HVoice &v = c->voices[vi];
		v = HVoice();
		v.live = true;
		v.key = key;
		c->building = vi;
		c->voices_dirty = true;
		++c->stats.live_voices;
	}
	HVoice &v = c->voices[vi];
	if(v.nunits >= A2D_MAXCHAIN)
		return c->fail(A2AMD_EUNSUPPORTED, "voice chain longer than %d units", A2D_MAXCHAIN);
	if(kind == A2AMD_INLINE && v.inline_pos >= 0)
		return c->fail(A2AMD_EUNSUPPORTED, "two inline units in one voice");

	int ui;
	if(!c->free_units.empty()) {
		ui = c->free_units.back();
		c->free_units.pop_back();
	} else {
		ui = (int)c->units.size();
		c->units.push_back(HUnit());
		c->mudesc.push_back(0);
	}
	HUnit &u = c->units[ui];
	u = HUnit();
	u.live = true;
	u.kind = kind;
	u.flags = flags;
	u.nin = nin;
	u.nout = nout;
	u.wired = wired ? 1 : 0;
	u.voice = vi;
	u.chainpos = v.nunits;
	v.unit[v.nunits++] = ui;
	v.plain = 0;
	++v.nlive;
	c->mudesc[ui] = A2D_DESC(kind, add ? 1 : 0, nin, nout, wired ? 1 : 0);
	c->udesc_dirty = true;
	c->voices_dirty = true;
	c->dirty_voices.push_back(vi);
	++c->stats.live_units;

	int initval = 0;
	switch(kind) {
	  case A2AMD_WTOSC:	// wtosc_Initialize, wtosc.c:390-423
		initval = transpose + c->cfg.basepitch;
		ramp_init(u.p, initval);
		u.dphase = p2i(c->ptab, u.p.value >> 8);
		u.phase = 0;
		u.p_ramping = 0;
		u.mode = A2D_OSC_OFF;
		u.wave = -1;
		break;
	  case A2AMD_FILTER12:	// f12_Initialize -> f12_CutOff(u, 0, 0, 0), filter12.c:141-147,203
		ramp_init(u.cutoff, 0);
		ramp_set(u.cutoff, transpose, 0, 0);
		initval = f12_coeff(c->ptab, u.cutoff.value, c->cfg.samplerate);
		break;
	  case A2AMD_FBDELAY:	// two zeroed delay lines, fbdelay.c:180-181
		if(!c->fbd_free.empty()) {
			u.fbdbuf = c->fbd_free.back();
			c->fbd_free.pop_back();
		} else
			u.fbdbuf = c->fbd_count++;
		c->fbd_to_zero.push_back(u.fbdbuf);
		initval = u.fbdbuf;
		{	// fbdelay_Initialize, fbdelay.c:183-196: 400 / 280 / 320 ms
			static const int ms[3] = { 400, 280, 320 };
			for(int t = 0; t < 3; ++t)
				u.fbd_taps[t] = (int)((int64_t)(ms[t] << 16) * c->cfg.samplerate / 65536000);
		}
		break;
	  case A2AMD_INLINE:
		v.inline_pos = u.chainpos;
		if(!wired) {
			v.own_nch = nout;
			v.own_off = bus_alloc(c, nout);
		}
		break;
	  default:
		break;
	}
	unsigned initdur = 0, initstart = 0;
	if(kind == A2AMD_DCBLOCK)	// dcb_Initialize, dcblock.c:131-132: cutoff -5.0 (8.18 Hz)
		initval = dcb_coeff(c->ptab, (int)((unsigned)(-5 * 65536) + (unsigned)transpose), c->cfg.samplerate);
	if(kind == A2AMD_LIMITER)	// limiter_Initialize, limiter.c:176-181
		initval = ((64 << 16) << 8) / c->cfg.samplerate;
	if(kind >= A2AMD_FM1 && kind <= A2AMD_FM4R) {	// fm_Initialize, fm.c:338-400
		if(!c->fm_free.empty()) {
			u.fmslot = c->fm_free.back();
			c->fm_free.pop_back();
		} else
			u.fmslot = c->fm_count++;
		initval = transpose + c->cfg.basepitch;
		initdur = (unsigned)u.fmslot;
		initstart = wakefrac & 0xffu;	// vms->waketime & 0xff: sub-sample start time
	}
	push_rec(c, vi, R_INIT, u.chainpos, 0, initval, initdur, initstart);
	return ui;
}

int a2amd_unit_deinit(a2amd_ctx *c, int ui)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "deinit of dead unit %d", ui);
	HUnit &u = c->units[ui];
	HVoice &v = c->voices[u.voice];
	if(c->building == u.voice)
		c->building = -1;
	if(u.kind == A2AMD_WTOSC && u.mode == A2D_OSC_NOISE)
		--c->n_noise;
	if(u.kind == A2AMD_FILTER12 && u.cutoff.timer)
		--c->n_cutoff_ramps;
	if(u.fbdbuf >= 0)
		c->fbd_deferred_free.push_back(u.fbdbuf);
	if(u.fmslot >= 0)
		c->fm_deferred_free.push_back(u.fmslot);
	if(u.xio >= 0)		// (the slot serves the batch being recorded to its end)
		c->xio_deferred_free.push_back(u.xio);
	if(u.xio_mode)
		--c->n_clients;
	u.live = false;
	c->deferred_free_units.push_back(ui);
	--c->stats.live_units;
	if(--v.nlive == 0) {
		// a2_VoiceFree (core.c:532-591) took the whole chain down
		push_rec(c, u.voice, R_KILL, 0, 0, 0, 0, 0);
		unhold(c, u.voice);
		if(v.started) {
			--c->n_started_live;
			if(c->frag_open && v.walked == c->serial_base + c->cur_frag)
				--c->walked_started;
		}
		v.dying = true;
		v.live = false;
		v.plain = 0;
		if(v.own_off >= 0)
			c->deferred_bus_free.push_back(std::make_pair(v.own_off, v.own_nch));
		c->deferred_free_voices.push_back(u.voice);
		c->lists_dirty = true;
		--c->stats.live_voices;
	}
	return A2AMD_OK;
}

static int shadow_rebuild(a2amd_ctx *c, int ui);

int a2amd_unit_write(a2amd_ctx *c, int ui, int reg, int value, unsigned start, unsigned dur, int transpose)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "write to dead unit %d", ui);
	HUnit &u = c->units[ui];
	c->building = -1;
	start &= 255;		// a2_VoiceControl, core.c:148
	switch(u.kind) {
	  case A2AMD_WTOSC:
		switch(reg) {
		  case 0: {	// wtosc_Wave, wtosc.c:433-483
			int wt = A2AMD_WOFF, id = -1;
			if(value >= 0 && value < (int)c->waves.size() && c->waves[value].live) {
				id = value;
				wt = c->waves[id].dw.type;
				if((wt == A2AMD_WWAVE || wt == A2AMD_WMIPWAVE) &&
						c->waves[id].dw.size[0] > (unsigned)A2D_WTOSC_MAXLENGTH)
					wt = A2AMD_WOFF;
			}
			int nmode = wt == A2AMD_WNOISE ? A2D_OSC_NOISE : wt == A2AMD_WWAVE ? A2D_OSC_WAVE :
					wt == A2AMD_WMIPWAVE ? A2D_OSC_MIPWAVE : A2D_OSC_OFF;
			if(nmode == A2D_OSC_NOISE && u.mode != A2D_OSC_NOISE)
				if(int r = shadow_rebuild(c, ui))	// from here on the host counts its draws
					return r;
			{
				// the wavetable leaf kernels only know mip-mapped waves (and "off"): a
				// voice that moves between the two kinds changes its launch class
				const bool was = leaf_mode(u.mode), is = leaf_mode(nmode);
				if(was != is)
					c->lists_dirty = true;
				if(!was || !is)
					c->voices[u.voice].mode_mix = true;
				c->voices[u.voice].plain = 0;
			}
			if(u.mode == A2D_OSC_NOISE && nmode != A2D_OSC_NOISE)
				--c->n_noise;
			if(u.mode != A2D_OSC_NOISE && nmode == A2D_OSC_NOISE)
				++c->n_noise;
			u.mode = nmode;
			u.wave = nmode == A2D_OSC_OFF ? -1 : id;
			value = id;
			break;
		  }
		  case 1:	// wtosc_Pitch, wtosc.c:486-492
			value = value + transpose + c->cfg.basepitch;
			ramp_set(u.p, value, (int)start, (int)dur);
			if(!dur)
				u.p_ramping = 1;
			break;
		  case 2:
			break;
		  case 3:	// wtosc_Phase -> wtosc_set_phase, wtosc.c:369-378
			if(u.wave < 0)
				u.phase = 0;
			else {
				int ph = (int)((unsigned)value + ((start * (u.dphase >> 8)) >> 8));
				u.phase = (uint64_t)(((int64_t)ph * (int64_t)c->waves[u.wave].dw.period) * 256);
			}
			break;
		  default:
			return c->fail(A2AMD_EINVAL, "wtosc register %d", reg);
		}
		break;
	  case A2AMD_PANMIX:
		if(reg < 0 || reg > 1)
			return c->fail(A2AMD_EINVAL, "panmix register %d", reg);
		break;
	  case A2AMD_FILTER12:
		switch(reg) {
		  case 0: {	// f12_CutOff, filter12.c:141-147: stays on the host
			bool was = u.cutoff.timer != 0;
			ramp_set(u.cutoff, value + transpose, (int)start, (int)dur);
			bool is = u.cutoff.timer != 0;
			c->n_cutoff_ramps += (int)is - (int)was;
			c->voices[u.voice].plain = 0;
			if(dur < 256)
				push_rec(c, u.voice, R_F1SET, u.chainpos, 0,
						f12_coeff(c->ptab, u.cutoff.value, c->cfg.samplerate), 0, 0);
			return A2AMD_OK;
		  }
		  case 1:	// f12_Q, filter12.c:149-162
			value = value < 512 ? 32768 : (65536 << 8) / value;
			break;
		  case 2: case 3: case 4:
			break;
		  default:
			return c->fail(A2AMD_EINVAL, "filter12 register %d", reg);
		}
		break;
	  case A2AMD_FBDELAY:
		if(reg < 0 || reg > 6)
			return c->fail(A2AMD_EINVAL, "fbdelay register %d", reg);
		if(reg < 3) {	// fbdelay_FBDelay / LDelay / RDelay, fbdelay.c:231-247
			const int frames = (int)((int64_t)value * c->cfg.samplerate / 65536000);
			// (a tap that stops or starts being at least a fragment long moves the
			// voice between the frame-parallel delay kernel and the general one)
			if(fbd_tap_ok(frames) != fbd_tap_ok(u.fbd_taps[reg]))
				c->lists_dirty = true;
			u.fbd_taps[reg] = frames;
		}
		break;
	  case A2AMD_INLINE:
	  case A2AMD_XINSERT:
	  case A2AMD_XSINK:
	  case A2AMD_XSOURCE:
		return c->fail(A2AMD_EINVAL, "unit kind %d has no registers", u.kind);
	  case A2AMD_DC:
		if(reg < 0 || reg > 1)
			return c->fail(A2AMD_EINVAL, "dc register %d", reg);
		break;
	  case A2AMD_WAVESHAPER:
		if(reg != 0)
			return c->fail(A2AMD_EINVAL, "waveshaper register %d", reg);
		break;
	  case A2AMD_DCBLOCK:	// dcb_CutOff, dcblock.c:112-117: float/libm, host only
		if(reg != 0)
			return c->fail(A2AMD_EINVAL, "dcblock register %d", reg);
		value = dcb_coeff(c->ptab, (int)((unsigned)value + (unsigned)transpose), c->cfg.samplerate);
		break;
	  case A2AMD_LIMITER:	// limiter_Release / limiter_Threshold, limiter.c:201-213
		if(reg == 0)
			value = (int)((unsigned)value << 8) / c->cfg.samplerate;
		else if(reg == 1) {
			unsigned t = (unsigned)value << 8;
			value = (int)(t < 256 ? 256u : t);
		} else
			return c->fail(A2AMD_EINVAL, "limiter register %d", reg);
		break;
	  default: {	// fm.c:403-483: phase | p a fb | p1 a1 fb1 | ...
		static const int nops[8] = { 1, 2, 3, 4, 3, 4, 2, 4 };
		if(reg < 0 || reg > 3 * nops[u.kind - A2AMD_FM1])
			return c->fail(A2AMD_EINVAL, "fm register %d", reg);
		if(reg == 1)	// fm_Pitch: operator 0 is the absolute one
			value = value + transpose + c->cfg.basepitch;
		break;
	  }
	}
	push_rec(c, u.voice, R_WRITE, u.chainpos, reg, value, dur, start);
	return A2AMD_OK;
}


// ---- wtosc phase shadow ------------------------------------------------------------
// The engine-global noise generator is shared by every noise oscillator and by
// the VM's RAND instructions (wtosc.c:136, a2_Noise()), so the host has to hand
// it back advanced by the right number of draws after each Process call of a
// noise oscillator; that count depends on the oscillator's phase.  Scripts do
// switch an oscillator from a wavetable to noise (benchmark/k2epilogue.a2s), so
// the phase is shadowed in wavetable mode too: the closed form of what the
// kernels compute sample by sample.  Pitch and phase only - no audio.
static void shadow_run_pitch(a2amd_ctx *c, HUnit &u, unsigned frames)
{
	ramp_prepare(u.p, (int)frames);		// wtosc_run_pitch, wtosc.c:88-106
	if(u.dphase && (!u.p.timer && !u.p_ramping))
		return;
	unsigned lastv = (unsigned)u.p.value;
	ramp_run(u.p, (int)frames);
	u.p_ramping = u.p.delta;
	u.dphase = p2i(c->ptab, (int)((lastv + (unsigned)u.p.value) >> 9));
}

static void shadow_wave(a2amd_ctx *c, HUnit &u, unsigned frames)
{
	const A2DWave &w = c->waves[u.wave].dw;
	const bool looped = (w.flags & 0x100u) != 0;	// A2_LOOPED
	if(!w.size[0]) {	// wtosc_check_unloaded, wtosc.c:168-183
		u.wave = -1;
		u.mode = A2D_OSC_OFF;
		return;
	}
	shadow_run_pitch(c, u, frames);
	if(u.mode == A2D_OSC_MIPWAVE) {		// wtosc_wavetable, wtosc.c:239-286
		unsigned dph = ((u.dphase + 255) >> 8) * w.period;
		unsigned mm = 0;
		for(; (dph > (A2D_MAXPHINC << 8)) && (mm < A2D_MIPS - 1); ++mm)
			dph >>= 1;
		uint64_t ph = u.phase >> mm;
		dph = (unsigned)(((uint64_t)u.dphase * w.period) >> mm);
		if(looped) {
			const uint64_t m = (uint64_t)w.size[mm] << 24;
			if(!(m & (m - 1)))
				ph &= m - 1;	// built-in waves: power-of-two levels
			else
				ph %= m;
		} else if((ph >> 24) > (uint64_t)(w.size[mm] + 1))
			return;		// all played
		u.phase = (ph + (uint64_t)dph * frames) << mm;
		return;
	}
	// wtosc_wavetable_no_mip, wtosc.c:301-358
	const uint64_t dph = (uint64_t)u.dphase * w.period;
	if(dph >> 32) {
		u.phase += dph * frames;
	} else if(dph > (A2D_MAXPHINC << 16)) {
		// per-sample loop/end test of wtosc_do_fragment, wtosc.c:207-226
		const uint64_t m = (uint64_t)w.size[0] << 24;
		if(looped) {
			u.phase = (u.phase + (uint64_t)(frames - 1) * dph) % m + dph;
		} else if(u.phase < m) {
			uint64_t steps = (m - u.phase + dph - 1) / dph;
			u.phase += (steps < frames ? steps : frames) * dph;
		}
	} else {
		if(looped) {
			unsigned m = w.size[0] << 24;	// 32 bit in the reference, wtosc.c:340
			if(m)
				u.phase %= m;
		} else if((u.phase >> 24) > (uint64_t)(w.size[0] + 1))
			return;
		u.phase += dph * frames;
	}
}

// one window of a wtosc as far as phase and pitch go (any Process variant)
static void shadow_window(a2amd_ctx *c, HUnit &u, unsigned frames)
{
	switch(u.mode) {
	  case A2D_OSC_OFF:	// wtosc_Off, wtosc.c:108-126
		ramp_prepare(u.p, (int)frames);
		ramp_run(u.p, (int)frames);
		break;
	  case A2D_OSC_NOISE:	// wtosc_noise, wtosc.c:129-152
		shadow_run_pitch(c, u, frames);
		u.phase += (uint64_t)frames * u.dphase;
		break;
	  default:
		if(u.wave >= 0)
			shadow_wave(c, u, frames);
		break;
	}
}

// which Process variant wtosc_Wave (wtosc.c:433-483) installs for wave slot 'value'
static int wave_mode(const a2amd_ctx *c, int value, int *id)
{
	int wt = A2AMD_WOFF;
	*id = -1;
	if(value >= 0 && value < (int)c->waves.size() && c->waves[value].live) {
		*id = value;
		wt = c->waves[value].dw.type;
		if((wt == A2AMD_WWAVE || wt == A2AMD_WMIPWAVE) && c->waves[value].dw.size[0] > (unsigned)A2D_WTOSC_MAXLENGTH)
			wt = A2AMD_WOFF;
	}
	return wt == A2AMD_WNOISE ? A2D_OSC_NOISE : wt == A2AMD_WWAVE ? A2D_OSC_WAVE :
			wt == A2AMD_WMIPWAVE ? A2D_OSC_MIPWAVE : A2D_OSC_OFF;
}

// ---- the oscillator state when it is needed on the host ----------------------------------
// The engine-global noise generator is shared by every noise oscillator and the VM's RAND
// instructions, so a window of a noise oscillator has to hand it back advanced by the right
// number of draws (a2amd_unit_process), and that number depends on the oscillator's phase
// and pitch - which, for an oscillator that has been playing a wave, only the device knows
// (scripts do switch oscillators from a wave to noise: benchmark/k2epilogue.a2s).  Round 1
// and the first half of round 2 shadowed every oscillator's phase on the host, window by
// window; that was a fifth of the engine thread's time with scripted voices.  Now nothing is
// shadowed while a wave plays.  When an oscillator is switched to noise, its state is
// rebuilt: the unit state words the device was left with by the last batch (one device to
// host copy of the state array per batch in which that happens), advanced over this batch's
// records of the voice up to now - the same arithmetic the kernels will run on them.
static int shadow_rebuild(a2amd_ctx *c, int ui)
{
	HUnit &u = c->units[ui];
	const int vi = u.voice;
	spell_out_pending(c, vi);
	if(!c->snap_valid) {
		use_device(c);
		const size_t nu = std::min(c->units.size(), c->d_ustate.cap), nv = std::min(c->voices.size(), c->d_vactive.cap);
		c->snap_ustate.assign(nu * A2D_USTATE, 0);
		c->snap_vactive.assign(nv, 0);
		HIPCHK(c, hipStreamSynchronize(c->stream));
		if(nu)
			HIPCHK(c, hipMemcpy(c->snap_ustate.data(), c->d_ustate.d, nu * A2D_USTATE * sizeof(int32_t),
					hipMemcpyDeviceToHost));
		if(nv)
			HIPCHK(c, hipMemcpy(c->snap_vactive.data(), c->d_vactive.d, nv * sizeof(int32_t), hipMemcpyDeviceToHost));
		c->snap_valid = true;
	}
	HUnit t = u;		// (its own copy: mode and wave as they were when the batch began)
	t.mode = A2D_OSC_OFF;
	t.wave = -1;
	t.dphase = 0;
	t.p_ramping = 0;
	t.phase = 0;
	t.p = Ramp{ 0, 0, 0, 0 };
	bool active = false;
	if((size_t)(ui + 1) * A2D_USTATE <= c->snap_ustate.size()) {
		const int32_t *w = &c->snap_ustate[(size_t)ui * A2D_USTATE];
		t.mode = w[OW_MODE];
		t.wave = w[OW_WAVE];
		t.dphase = (unsigned)w[OW_DPHASE];
		t.phase = (uint64_t)(uint32_t)w[OW_PHASE_LO] | ((uint64_t)(uint32_t)w[OW_PHASE_HI] << 32);
		t.p_ramping = w[OW_PRAMPING];
		t.p = Ramp{ w[OW_P], w[OW_P + 1], w[OW_P + 2], w[OW_P + 3] };
		if(t.wave >= (int)c->waves.size())
			t.wave = -1;
	}
	if((size_t)vi < c->snap_vactive.size())
		active = c->snap_vactive[vi] != 0;
	// this batch's fragments up to the open one, the way k_voices executes them
	const HVoice &v = c->voices[vi];
	const int upto = rec_tag(c);
	size_t r = 0;
	for(int f = 0; f <= upto && f < A2D_MAXBATCH; ++f) {
		if(!(r < v.recs.size() && (int)A2D_RFRAG(v.recs[r].head) == f)) {
			// no records: the default window - but not of the open fragment (the voice's
			// turn has not come, or its window so far would have been spelled out above)
			if(active && f < upto)
				shadow_window(c, t, c->fragframes[f]);
			continue;
		}
		for(; r < v.recs.size() && (int)A2D_RFRAG(v.recs[r].head) == f; ++r) {
			const A2DRec &rec = v.recs[r];
			const bool mine = (int)A2D_RUNIT(rec.head) == u.chainpos;
			switch(A2D_ROP(rec.head)) {
			  case R_SEG:
				if(active)
					shadow_window(c, t, rec.dur >> 16);
				break;
			  case R_INIT:
				active = true;
				if(mine) {	// wtosc_Initialize, wtosc.c:390-423
					ramp_init(t.p, rec.value);
					t.dphase = p2i(c->ptab, t.p.value >> 8);
					t.phase = 0;
					t.p_ramping = 0;
					t.mode = A2D_OSC_OFF;
					t.wave = -1;
				}
				break;
			  case R_KILL:
				active = false;
				break;
			  case R_WRITE:
				if(!mine)
					break;
				switch(A2D_RREG(rec.head)) {
				  case 0:
					t.mode = wave_mode(c, rec.value, &t.wave);
					if(t.mode == A2D_OSC_OFF)
						t.wave = -1;
					break;
				  case 1:	// wtosc_Pitch, wtosc.c:486-492 (transpose and base pitch are in the record)
					ramp_set(t.p, rec.value, (int)rec.start, (int)rec.dur);
					if(!rec.dur)
						t.p_ramping = 1;
					break;
				  case 3:	// wtosc_set_phase, wtosc.c:369-378
					if(t.wave < 0)
						t.phase = 0;
					else {
						int ph = (int)((unsigned)rec.value + ((rec.start * (t.dphase >> 8)) >> 8));
						t.phase = (uint64_t)(((int64_t)ph * (int64_t)c->waves[t.wave].dw.period) * 256);
					}
					break;
				}
				break;
			  default:
				break;
			}
		}
	}
	u.dphase = t.dphase;
	u.p_ramping = t.p_ramping;
	u.phase = t.phase;
	u.p = t.p;
	return A2AMD_OK;
}

int a2amd_unit_process(a2amd_ctx *c, int ui, unsigned offset, unsigned frames, uint32_t *noisestate)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "process of dead unit %d", ui);
	if(!c->frag_open || !frames || offset + frames > c->fragframes[c->cur_frag])
		return c->fail(A2AMD_ESTATE, "process [%u,+%u) outside the open fragment", offset, frames);
	HUnit &u = c->units[ui];
	const int vi = u.voice;
	HVoice &v = c->voices[vi];
	c->building = -1;
	unhold(c, vi);		// (a call speaks for itself)
	if(!v.resolved) {
		resolve_out(c, v);
		v.plain = 0;
	}
	if(!v.started) {
		v.started = true;
		v.plain = 0;
		++c->n_started_live;
	}
	touch(c, vi);
	if(v.walked != c->serial_base + c->cur_frag) {
		v.walked = c->serial_base + c->cur_frag;
		++c->walked_started;
	}
	if(u.chainpos == 0) {
		v.win_off = (int)offset;
		v.win_frames = (int)frames;
	} else if(v.win_off != (int)offset || v.win_frames != (int)frames)
		return c->fail(A2AMD_ESTATE, "units of one voice processed over different windows");

	switch(u.kind) {
	  case A2AMD_WTOSC:
		if(u.mode == A2D_OSC_NOISE) {
			// Count the draws wtosc_noise (wtosc.c:129-152) makes over this
			// window and hand the engine-global RNG back advanced, so VM
			// RAND instructions and other voices interleave as they do in
			// the reference; the kernel regenerates the same draws from
			// the seed recorded here.
			if(!noisestate)
				return c->fail(A2AMD_EINVAL, "noise oscillator needs the engine's noise state");
			push_rec(c, vi, R_NOISESEED, u.chainpos, 0, (int)*noisestate, 0, 0);
			shadow_run_pitch(c, u, frames);
			uint64_t end = u.phase + (uint64_t)frames * u.dphase;
			uint64_t draws = u.dphase >= (1u << 23) ? frames : (end >> 23) - (u.phase >> 23);
			static const bool trace = getenv("A2AMD_DEBUG_NOISE") != nullptr;	// (debugging aid: the test
			if(trace)								// suite's checker prints the same line)
				fprintf(stderr, "NOISE phase %llx dphase %x frames %u\n", (unsigned long long)u.phase, u.dphase, frames);
			uint32_t st = *noisestate;
			for(uint64_t i = 0; i < draws; ++i)
				st = st * 1566083941u + 1u;
			*noisestate = st;
			u.phase = end;
		}
		// (any other mode: the device does it all, and keeps the state - shadow_rebuild)
		break;
	  case A2AMD_FILTER12: {	// the head of f12_process, filter12.c:86-96
		bool was = u.cutoff.timer != 0;
		ramp_prepare(u.cutoff, (int)frames);
		if(u.cutoff.delta) {
			ramp_run(u.cutoff, (int)frames);
			push_rec(c, vi, R_F1RAMP, u.chainpos, 0,
					f12_coeff(c->ptab, u.cutoff.value, c->cfg.samplerate), 0, 0);
		}
		c->n_cutoff_ramps += (int)(u.cutoff.timer != 0) - (int)was;
		if(was && !u.cutoff.timer)
			v.plain = 0;	// (the ramp has arrived: the voice may be plain again)
		break;
	  }
	  case A2AMD_INLINE:
		c->stack.push_back(ui);
		break;
	  default:
		break;
	}
	v.win_done = u.chainpos + 1;
	if(u.chainpos == v.nunits - 1) {
		// The common case at scale - the voice's VM slept through the fragment,
		// the engine made one full-window Process call per unit, nothing else
		// happened to the voice - costs no record: the kernels apply that
		// default to every voice without records (push_rec spells it out if
		// something does follow in this fragment).
		if(offset == 0 && frames == c->fragframes[c->cur_frag] && v.recs.size() == v.frag_mark &&
				v.deferred.empty()) {
			v.default_seg = c->serial_base + c->cur_frag;
			v.win_done = 0;
		} else
			push_rec(c, vi, R_SEG, 0, 0, 0, offset | (frames << 16), 0);
	}
	return A2AMD_OK;
}

namespace {
// is every unit of the voice one whose Process leaves nothing to do on the host
// but - for a wtosc - the phase shadow?
void classify_plain(a2amd_ctx *c, HVoice &v)
{
	v.plain = 2;
	if(!v.live || v.dying || !v.resolved || !v.started || v.inline_pos >= 0)
		return;
	for(int k = 0; k < v.nunits; ++k) {
		const HUnit &u = c->units[v.unit[k]];
		switch(u.kind) {
		  case A2AMD_WTOSC:
			if(u.mode == A2D_OSC_NOISE)
				return;
			break;
		  case A2AMD_FILTER12:
			if(u.cutoff.timer || u.cutoff.delta)
				return;
			break;
		  case A2AMD_INLINE: case A2AMD_XINSERT: case A2AMD_XSINK: case A2AMD_XSOURCE:
			return;
		  default:
			break;
		}
	}
	v.plain = 1;
}
} // namespace

int a2amd_voice_process(a2amd_ctx *c, int head, unsigned offset, unsigned frames, uint32_t *noisestate)
{
	if(head < 0 || head >= (int)c->units.size() || !c->units[head].live)
		return c->fail(A2AMD_EINVAL, "process of dead unit %d", head);
	const int vi = c->units[head].voice;
	HVoice &pv = c->voices[vi];
	unhold(c, vi);
	if(!pv.plain)
		classify_plain(c, pv);
	static const int dbgw = getenv("A2AMD_DBG_WALK") ? atoi(getenv("A2AMD_DBG_WALK")) : 0;
	if(pv.plain == 1 && !(dbgw & 1)) {
		// The short path: one pass of bookkeeping for the whole chain (what
		// a2amd_unit_process does unit by unit) and the window - unrecorded if it is
		// the default one.
		if(!c->frag_open || !frames || offset + frames > c->fragframes[c->cur_frag])
			return c->fail(A2AMD_ESTATE, "process [%u,+%u) outside the open fragment", offset, frames);
		const long long serial = c->serial_base + c->cur_frag;
		c->building = -1;
		if(pv.touched != serial) {
			pv.touched = serial;
			pv.frag_mark = pv.recs.size();
		}
		if(pv.walked != serial) {
			pv.walked = serial;
			++c->walked_started;
		}
		pv.win_off = (int)offset;
		pv.win_frames = (int)frames;
		if(offset == 0 && frames == c->fragframes[c->cur_frag] && pv.recs.size() == pv.frag_mark &&
				pv.deferred.empty()) {
			pv.default_seg = serial;
			pv.win_done = 0;
		} else
			push_rec(c, vi, R_SEG, 0, 0, 0, offset | (frames << 16), 0);
		return (size_t)vi < c->defmap.size() && !(dbgw & 2) ? 1 : 0;
	}
	const int n = c->voices[vi].nunits;
	for(int k = 0; k < n; ++k) {
		const int ui = c->voices[vi].unit[k];
		if(c->units[ui].kind == A2AMD_INLINE)
			return c->fail(A2AMD_EINVAL, "voice_process on a voice with an inline unit");
		if(int r = a2amd_unit_process(c, ui, offset, frames, noisestate))
			return r;
	}
	// may the host mark this voice in the default map from the next fragment on?
	const HVoice &v = c->voices[vi];
	if((size_t)vi >= c->defmap.size())
		return 0;
	for(int k = 0; k < n; ++k) {
		const HUnit &u = c->units[v.unit[k]];
		if((u.kind == A2AMD_WTOSC && u.mode == A2D_OSC_NOISE) || u.xio_mode ||
				(u.kind == A2AMD_FILTER12 && (u.cutoff.timer || u.cutoff.delta)))
			return 0;
	}
	return (dbgw & 2) ? 0 : 1;
}

int a2amd_voice_markable(a2amd_ctx *c, int ui)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "markable: dead unit %d", ui);
	const int vi = c->units[ui].voice;
	const HVoice &v = c->voices[vi];
	// (walked at least once: its output bus is resolved, its place in the launch order known)
	if(!v.live || v.dying || !v.resolved || !v.started || (size_t)vi >= c->defmap.size())
		return 0;
	for(int k = 0; k < v.nunits; ++k) {
		const HUnit &u = c->units[v.unit[k]];
		if((u.kind == A2AMD_WTOSC && u.mode == A2D_OSC_NOISE) || u.xio_mode ||
				(u.kind == A2AMD_FILTER12 && (u.cutoff.timer || u.cutoff.delta)))
			return 0;
	}
	return 1;
}

int a2amd_voice_slot(a2amd_ctx *c, int ui)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "slot of dead unit %d", ui);
	return c->units[ui].voice;
}

int a2amd_default_hold(a2amd_ctx *c, const uint32_t *slots, unsigned n, unsigned lo, int on)
{
	if(on && c->held.size() < c->voices.size())
		c->held.resize(c->voices.size() * 2 + 65536, 0);
	for(unsigned k = 0; k < n; ++k) {
		const size_t vi = slots ? slots[k] : (size_t)lo + k;
		if(!on) {
			unhold(c, (int)vi);
			continue;
		}
		if(vi >= c->voices.size())
			return c->fail(A2AMD_EINVAL, "default_hold: slot %zu", vi);
		const HVoice &v = c->voices[vi];
		if(!v.live || v.dying || !v.resolved || !v.started)
			return c->fail(A2AMD_ESTATE, "default_hold: voice %zu has not been processed yet", vi);
		if(!c->held[vi]) {
			c->held[vi] = 1;
			++c->n_held;
		}
	}
	return A2AMD_OK;
}

int a2amd_default_release_all(a2amd_ctx *c)
{
	if(c->n_held)
		std::fill(c->held.begin(), c->held.end(), 0);
	c->n_held = 0;
	return A2AMD_OK;
}

uint8_t *a2amd_default_map(a2amd_ctx *c, unsigned *nslots)
{
	if(!c->frag_open)
		return nullptr;
	c->defmap_used = c->defmap_dirty = true;
	if(nslots)
		*nslots = (unsigned)c->defmap.size();
	return c->defmap.data();
}

int a2amd_unit_clients(a2amd_ctx *c, int ui, unsigned mode)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live ||
			(c->units[ui].kind != A2AMD_XINSERT && c->units[ui].kind != A2AMD_XSINK &&
			 c->units[ui].kind != A2AMD_XSOURCE))
		return c->fail(A2AMD_EINVAL, "unit %d is not a live xinsert / xsink / xsource", ui);
	if((mode & ~(unsigned)(A2AMD_XIO_TAP | A2AMD_XIO_INJECT | A2AMD_XIO_MUTE)) ||
			((mode & A2AMD_XIO_MUTE) && (!(mode & A2AMD_XIO_TAP) || a2amd_unit_insertable(c, ui) < 1)) ||
			(c->units[ui].kind == A2AMD_XSINK && (mode & A2AMD_XIO_INJECT)) ||
			(c->units[ui].kind == A2AMD_XSOURCE && (mode & A2AMD_XIO_TAP)))
		return c->fail(A2AMD_EINVAL, "client mode %#x on unit kind %d", mode, c->units[ui].kind);
	HUnit &u = c->units[ui];
	if(mode == u.xio_mode)
		return A2AMD_OK;
	if(mode && u.xio < 0) {
		if(!c->xio_free.empty()) {
			u.xio = c->xio_free.back();
			c->xio_free.pop_back();
		} else {
			u.xio = (int)c->xio.size();
			c->xio.emplace_back();
		}
		XioSlot &x = c->xio[u.xio];
		x.unit = x.last_unit = ui;
		x.tap.assign(A2D_XIO_HALF, 0);
		x.inj.assign(A2D_XIO_HALF, 0);
		x.inj_used = x.tapped = false;
	}
	if(mode & A2AMD_XIO_TAP)
		c->xio[u.xio].tapped = true;
	// (the device reads slot and mode from the unit's state words: two writes,
	// in order with the voice's windows)
	c->building = -1;
	push_rec(c, u.voice, R_WRITE, u.chainpos, 0, mode ? u.xio + 1 : 0, 0, 0);
	push_rec(c, u.voice, R_WRITE, u.chainpos, 1, (int)mode, 0, 0);
	c->n_clients += (int)(mode != 0) - (int)(u.xio_mode != 0);
	u.xio_mode = mode;
	c->lists_dirty = true;		// a driver chain with clients is served by the general kernel
	return A2AMD_OK;
}

int a2amd_unit_inject(a2amd_ctx *c, int ui, unsigned offset, unsigned frames, const int32_t *const *bufs)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live || !(c->units[ui].xio_mode & A2AMD_XIO_INJECT))
		return c->fail(A2AMD_EINVAL, "unit %d takes no client output", ui);
	if(!c->frag_open || !frames || offset + frames > c->fragframes[c->cur_frag])
		return c->fail(A2AMD_ESTATE, "inject [%u,+%u) outside the open fragment", offset, frames);
	HUnit &u = c->units[ui];
	XioSlot &x = c->xio[u.xio];
	for(int ch = 0; ch < u.nout; ++ch) {
		int32_t *d = x.inj.data() + ((size_t)c->cur_frag * A2AMD_MAXCHANNELS + ch) * A2D_FRAG + offset;
		for(unsigned k = 0; k < frames; ++k)
			d[k] = wadd(d[k], bufs[ch][k]);
	}
	x.inj_used = true;
	return A2AMD_OK;
}

int a2amd_unit_insertable(a2amd_ctx *c, int ui)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "insertable: dead unit %d", ui);
	const HUnit &u = c->units[ui];
	const HVoice &v = c->voices[u.voice];
	// The render goes nesting depth by nesting depth, deepest first, and can pause between two
	// of them: a unit that is the last of its voice and adds into the voice's output bus (the
	// parent's inline bus) can have its output replaced there before the parent's chain runs.
	// (depth and bus are known once the voice has been processed for the first time: resolve_out)
	return u.kind == A2AMD_XINSERT && v.resolved && v.depth >= 1 && v.out_off != 0 && u.chainpos == v.nunits - 1 &&
			u.wired && (u.flags & A2AMD_PROCADD) && !c->comm ? v.depth : 0;
}

int a2amd_render_paused(a2amd_ctx *c)
{
	return c->paused_at;
}

int a2amd_unit_insert(a2amd_ctx *c, int ui, unsigned fragment, unsigned offset, unsigned frames,
		const int32_t *const *bufs)
{
	if(ui < 0 || ui >= (int)c->units.size() || c->units[ui].xio < 0 || c->xio[c->units[ui].xio].last_unit != ui)
		return c->fail(A2AMD_EINVAL, "unit %d has had no clients", ui);
	if(!c->uploaded || (int)fragment >= c->nfrags || !frames || offset + frames > c->fragframes[fragment])
		return c->fail(A2AMD_ESTATE, "insert [%u,+%u) of fragment %u: not between the SUBTREES and the ROOT phase "
				"of a batch that has it", offset, frames, fragment);
	const HUnit &u = c->units[ui];
	XioSlot &x = c->xio[u.xio];
	if(x.late.empty())
		x.late.assign(A2D_XIO_HALF, 0);
	for(int ch = 0; ch < u.nin; ++ch) {
		int32_t *d = x.late.data() + ((size_t)fragment * A2AMD_MAXCHANNELS + ch) * A2D_FRAG + offset;
		for(unsigned k = 0; k < frames; ++k)
			d[k] = wadd(d[k], bufs[ch][k]);
	}
	x.late_used = true;
	return A2AMD_OK;
}

int a2amd_unit_tapped(a2amd_ctx *c, int ui, unsigned fragment, const int32_t **bufs)
{
	// (also for a unit that was deinitialised in the course of that batch)
	if(ui < 0 || ui >= (int)c->units.size() || c->units[ui].xio < 0 || c->xio[c->units[ui].xio].last_unit != ui)
		return c->fail(A2AMD_EINVAL, "unit %d has had no clients", ui);
	if(fragment >= A2D_MAXBATCH)
		return c->fail(A2AMD_EINVAL, "fragment %u", fragment);
	const HUnit &u = c->units[ui];
	for(int ch = 0; ch < u.nin; ++ch)
		bufs[ch] = c->xio[u.xio].tap.data() + ((size_t)fragment * A2AMD_MAXCHANNELS + ch) * A2D_FRAG;
	return u.nin;
}

int a2amd_inline_end(a2amd_ctx *c, int ui)
{
	if(c->stack.empty() || c->stack.back() != ui)
		return c->fail(A2AMD_ESTATE, "inline_end(%d) does not match the open window", ui);
	c->stack.pop_back();
	return A2AMD_OK;
}

// ---- render -------------------------------------------------------------------------
static double now_us()
{
	timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static double g_t[4], g_n;
static double g_why[13];
static double g_cnt[6];		// quiet uploads, records shipped, walk scans, R_NOPs, voices with records, graph launches
struct TimingDump { ~TimingDump() { if(getenv("A2AMD_HOSTTIMING") && g_n) fprintf(stderr,
	"a2amd host timing per render: upload %.1f us, issue %.1f us, readback %.1f us (%g renders; %g quiet uploads, "
	"%g graph launches, %g records, %g voices with records, %g walk scans, %g R_NOPs)\n",
	g_t[0] / g_n, g_t[1] / g_n, g_t[2] / g_n, g_n, g_cnt[0], g_cnt[5], g_cnt[1], g_cnt[4], g_cnt[2], g_cnt[3]);
	if(getenv("A2AMD_HOSTTIMING") && g_n) { fprintf(stderr, "a2amd uploads by first reason against the quiet path "
	"(blob, recs, prev recs, voices, udesc, waves, lists, ptab, dirty voices, fbd, nfrags, bus, none):");
	for(int k = 0; k < 13; ++k) fprintf(stderr, " %g", g_why[k]); fprintf(stderr, "\n"); } } } g_timing_dump;

namespace { double *dbg_counters() { return g_cnt; } double *dbg_why() { return g_why; } }
static int dist_reduce_root(a2amd_ctx *c);

// What the READ clients of a context's x-units are to be handed: the tapped windows of the
// batch, device -> host (a2amd_unit_tapped reads them).  final: the batch is complete - a
// slot stays tapped into the next batch only while its unit still has READ clients.
static int fetch_taps(a2amd_ctx *c, bool final)
{
	bool any = false;
	for(size_t k = 0; k < c->xio.size(); ++k) {
		XioSlot &x = c->xio[k];
		if(x.unit >= 0 && x.tapped) {
			if(final)
				x.tapped = (c->units[x.unit].xio_mode & A2AMD_XIO_TAP) != 0;
			HIPCHK(c, hipMemcpyAsync(x.tap.data(), c->d_xio.d + k * A2D_XIO_SLOT,
					(size_t)c->nfrags * A2AMD_MAXCHANNELS * A2D_FRAG * sizeof(int32_t),
					hipMemcpyDeviceToHost, c->stream));
			any = true;
		}
	}
	if(any)
		HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

int a2amd_render(a2amd_ctx *c, unsigned phases, int32_t *const *out, unsigned cap)
{
	use_device(c);
	const bool timing = c->hosttiming;	// (debug aid; the accumulators are process-wide and not thread safe)
	double t0 = timing ? now_us() : 0;
	if(!c->stack.empty())
		return c->fail(A2AMD_ESTATE, "render inside an inline window");
	close_fragment(c);
	unsigned total = 0;
	for(int f = 0; f < c->nfrags; ++f)
		total += c->fragframes[f];
	if(!c->nfrags) {
		if(!(phases & A2AMD_RENDER_KEEP))
			;	// nothing recorded: records made outside any fragment wait for the next batch
		return 0;
	}
	c->snap_valid = false;		// (the device's unit states move on)
	if(phases & A2AMD_RENDER_UPLOAD)
		if(int r = upload(c))
			return r;
	double t1 = timing ? now_us() : 0;
	if(timing)
		g_t[0] += t1 - t0;
	if((phases & (A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT)) && !c->uploaded)
		return c->fail(A2AMD_ESTATE, "render phases out of order: upload first");

	if((phases & A2AMD_RENDER_SUBTREES) && c->profiling) {
		if(c->ev_used + 3 > c->ev_pool.size())
			for(int i = 0; i < 3; ++i) {
				hipEvent_t e;
				HIPCHK(c, hipEventCreate(&e));
				c->ev_pool.push_back(e);
			}
		c->ev0 = c->ev_pool[c->ev_used];
		c->ev1 = c->ev_pool[c->ev_used + 1];
		c->ev2 = c->ev_pool[c->ev_used + 2];
		c->ev_used += 3;
	}
	// A record-free batch that has been seen before runs from a graph - one launch
	// instead of 3-5 separate commands: a kept batch re-run phase by phase
	// (multi-GPU steps), or the engine recording the same quiet batch again.
	auto run_phases = [&](unsigned kphases) -> int {
		if(!kphases)
			return 0;
		if(c->uploaded && !c->profiling && c->stream && c->with_recs.empty() && !getenv("A2AMD_NO_GRAPH") &&
				!(phases & A2AMD_RENDER_TAPS) && c->sub_resume < 0 && !c->paused_at &&
				((phases & A2AMD_RENDER_KEEP) ? (phases & ~A2AMD_RENDER_KEEP) ==
				 (phases & (A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT)) :
				 // (not for a realtime driver's one-fragment batches: measured, hipGraphLaunch
				 // costs more there than the three or four launches it replaces - 170 us
				 // against 25 us of host time per fragment at 65 536 voices)
				 c->quiet_streak >= 1 && c->nfrags >= 8)) {
			const int slot = kphases == A2AMD_RENDER_SUBTREES ? 2 : kphases == A2AMD_RENDER_ROOT ? 3 : 1;
			if(c->gexec[slot] || !build_graph(c, slot, 1, kphases)) {
				if(slot != 3)
					if(int r = ensure_clean(c))
						return r;
				HIPCHK(c, hipGraphLaunch(c->gexec[slot], c->stream));
				if(c->hosttiming)
					dbg_counters()[5] += 1;
				if(slot == 1)
					c->others_clean = c->root_clean = c->consume_ok;
				else if(slot == 2) {
					c->others_clean = c->consume_ok;
					c->root_clean = false;
				} else if(c->consume_ok)
					c->root_clean = true;
				if(kphases & A2AMD_RENDER_ROOT) {
					c->stats.fragments += c->nfrags;
					c->stats.voice_fragments += (uint64_t)c->nfrags * (c->list_all.size() - c->n_list_pads);
				}
				return 0;
			}
		}
		// Events only when profiling (each one from the pool, used once until read):
		// re-recording an event the GPU has not reached yet makes the runtime wait.
		const bool sub = (kphases & A2AMD_RENDER_SUBTREES) != 0, root = (kphases & A2AMD_RENDER_ROOT) != 0;
		return c->profiling ? issue_kernels(c, kphases | (phases & (A2AMD_RENDER_KEEP | A2AMD_RENDER_TAPS)), sub ? c->ev0 : nullptr,
				sub ? c->ev1 : nullptr, root ? c->ev2 : nullptr) :
				issue_kernels(c, kphases | (phases & (A2AMD_RENDER_KEEP | A2AMD_RENDER_TAPS)), nullptr, nullptr, nullptr);
	};
	const unsigned kphases = phases & (A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT);
	if(c->comm && !c->dist_local && kphases == (A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT)) {
		// multi-GPU batch: every rank its subtrees, ONE reduce of the root voice's
		// inline bus over xGMI, the root chain on rank 0 (its panmix must see the
		// sum: the multiply truncates)
		if(int r = run_phases(A2AMD_RENDER_SUBTREES))
			return r;
		if(int r = dist_reduce_root(c))
			return r;
		if(c->dist_rank == 0)
			if(int r = run_phases(A2AMD_RENDER_ROOT))
				return r;
	} else {
		if(kphases && !(phases & A2AMD_RENDER_UPLOAD)) {
			// what insert clients made of the voices' taps since the render paused
			// (a2amd_unit_insert) joins the voices' output bus before their parents' chains run
			for(size_t k = 0; k < c->xio.size(); ++k) {
				XioSlot &x = c->xio[k];
				if(!x.late_used)
					continue;
				const size_t n = (size_t)c->nfrags * A2AMD_MAXCHANNELS * A2D_FRAG;
				// (also for a voice that died in the course of the batch: its unit and
				// voice entries stay until the batch ends)
				if(x.last_unit >= 0 && x.last_unit < (int)c->units.size() && c->units[x.last_unit].voice >= 0) {
					const HVoice &v = c->voices[c->units[x.last_unit].voice];
					HIPCHK(c, hipMemcpyAsync(c->d_xio.d + k * A2D_XIO_SLOT + A2D_XIO_HALF, x.late.data(),
							n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
					HIPCHK(c, hipStreamSynchronize(c->stream));	// (x.late is pageable and cleared next)
					if(a2d_launch_add_inject(c->d_xio.d + k * A2D_XIO_SLOT + A2D_XIO_HALF, c->d_busmem.d + v.out_off,
							v.out_nch, std::min(c->units[x.last_unit].nin, v.out_nch), c->nfrags, c->stream))
						return c->fail(A2AMD_EHIP, "insert launch failed");
				}
				std::fill(x.late.begin(), x.late.begin() + n, 0);
				x.late_used = false;
			}
		}
		if(int r = run_phases(kphases))
			return r;
	}
	if((phases & A2AMD_RENDER_TAPS) && !(phases & A2AMD_RENDER_READBACK))
		// the seam for insert clients: the batch's taps so far, on the host
		if(int r = fetch_taps(c, false))
			return r;
	double t2 = timing ? now_us() : 0;
	if(timing) {
		g_t[1] += t2 - t1;
		g_n += 1;
	}
	if((phases & A2AMD_RENDER_READBACK) && c->comm && !c->dist_local && c->dist_rank != 0) {
		// (the master bus exists on rank 0 only)
		if(!(phases & A2AMD_RENDER_KEEP))
			end_batch(c);
		return (int)total;
	}
	if(phases & A2AMD_RENDER_READBACK) {
		const int nch = c->cfg.channels;
		size_t n = (size_t)c->nfrags * nch * A2D_FRAG;
		if(phases & A2AMD_RENDER_ASYNC) {
			// enqueue the copy and return: a2amd_collect() waits for it and fills
			// the caller's buffers, up to two batches later
			for(const XioSlot &x : c->xio)
				if(x.unit >= 0 && x.tapped)
					return c->fail(A2AMD_EUNSUPPORTED, "asynchronous readback with READ clients attached");
			if(c->rb_count == 2)
				return c->fail(A2AMD_ESTATE, "two readbacks in flight: a2amd_collect() first");
			a2amd_ctx::Readback &rb = c->rb[(c->rb_head + c->rb_count) & 1];
			if(n > rb.cap) {
				if(rb.h)
					HIPCHK(c, hipHostFree(rb.h));
				rb.h = nullptr;
				rb.cap = 0;
				HIPCHK(c, hipHostMalloc((void **)&rb.h, n * sizeof(int32_t), hipHostMallocDefault));
				rb.cap = n;
			}
			if(!rb.ev)
				HIPCHK(c, hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming));
			HIPCHK(c, hipMemcpyAsync(rb.h, c->d_busmem.d, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipEventRecord(rb.ev, c->stream));
			rb.nfrags = c->nfrags;
			rb.total = total;
			for(int f = 0; f < c->nfrags; ++f)
				rb.frames[f] = (uint8_t)c->fragframes[f];
			++c->rb_count;
			if(!(phases & A2AMD_RENDER_KEEP))
				end_batch(c);
			return (int)total;
		}
		if(!out)
			return c->fail(A2AMD_EINVAL, "readback without output buffers");
		if(total > cap)
			return c->fail(A2AMD_EINVAL, "output capacity %u < %u frames", cap, total);
		if(n > c->h_master_cap) {
			if(c->h_master)
				HIPCHK(c, hipHostFree(c->h_master));
			HIPCHK(c, hipHostMalloc((void **)&c->h_master, n * sizeof(int32_t), hipHostMallocDefault));
			c->h_master_cap = n;
		}
		HIPCHK(c, hipMemcpyAsync(c->h_master, c->d_busmem.d, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
		for(size_t k = 0; k < c->xio.size(); ++k) {
			// what the READ clients of xinsert units are to be handed
			XioSlot &x = c->xio[k];
			if(x.unit >= 0 && x.tapped) {
				x.tapped = (c->units[x.unit].xio_mode & A2AMD_XIO_TAP) != 0;
				HIPCHK(c, hipMemcpyAsync(x.tap.data(), c->d_xio.d + k * A2D_XIO_SLOT,
						(size_t)c->nfrags * A2AMD_MAXCHANNELS * A2D_FRAG * sizeof(int32_t),
						hipMemcpyDeviceToHost, c->stream));
			}
		}
		HIPCHK(c, hipStreamSynchronize(c->stream));
		if(timing) {
			g_t[2] += now_us() - t2;
			static const int trace = getenv("A2AMD_HOSTTIMING") ? atoi(getenv("A2AMD_HOSTTIMING")) : 0;
			if(trace >= 2)
				fprintf(stderr, "a2amd render: %d fragments, upload %.1f us, issue %.1f us, readback %.1f us\n",
						c->nfrags, t1 - t0, t2 - t1, now_us() - t2);
		}
		unsigned pos = 0;
		for(int f = 0; f < c->nfrags; ++f) {
			for(int ch = 0; ch < nch; ++ch)
				memcpy(out[ch] + pos, c->h_master + ((size_t)f * nch + ch) * A2D_FRAG,
						c->fragframes[f] * sizeof(int32_t));
			pos += c->fragframes[f];
		}
		float ms = 0;
		if(c->profiling && hipEventElapsedTime(&ms, c->ev0, c->ev2) == hipSuccess)
			c->stats.last_kernel_ms = ms;
		if(c->profiling && hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess)
			c->stats.last_leaf_ms = ms;
	}
	if(!(phases & A2AMD_RENDER_KEEP) && (phases & (A2AMD_RENDER_READBACK | A2AMD_RENDER_ROOT)))
		end_batch(c);
	return (int)total;
}

int a2amd_collect(a2amd_ctx *c, int32_t *const *out, unsigned cap)
{
	use_device(c);
	if(!c->rb_count)
		return 0;
	a2amd_ctx::Readback &rb = c->rb[c->rb_head];
	if(!out)
		return c->fail(A2AMD_EINVAL, "collect without output buffers");
	if(rb.total > cap)
		return c->fail(A2AMD_EINVAL, "output capacity %u < %u frames", cap, rb.total);
	HIPCHK(c, hipEventSynchronize(rb.ev));
	const int nch = c->cfg.channels;
	unsigned pos = 0;
	for(int f = 0; f < rb.nfrags; ++f) {
		for(int ch = 0; ch < nch; ++ch)
			memcpy(out[ch] + pos, rb.h + ((size_t)f * nch + ch) * A2D_FRAG, rb.frames[f] * sizeof(int32_t));
		pos += rb.frames[f];
	}
	c->rb_head ^= 1;
	--c->rb_count;
	return (int)rb.total;
}

int a2amd_replay(a2amd_ctx *c, unsigned steps)
{
	use_device(c);
	c->snap_valid = false;
	const int GRAPH_STEPS = 8;
	if(!c->uploaded || !c->nfrags)
		return c->fail(A2AMD_ESTATE, "replay without an uploaded batch");
	for(int vi = 0; vi < (int)c->voices.size(); ++vi)
		if(!c->voices[vi].recs.empty())
			return c->fail(A2AMD_ESTATE, "replay of a batch that carries command records");
	bool graphs = c->stream != nullptr && !c->profiling && !getenv("A2AMD_NO_GRAPH");
	if(graphs && !c->gexec[0]) {
		if(build_graph(c, 0, GRAPH_STEPS) || build_graph(c, 1, 1)) {
			drop_graphs(c);
			graphs = false;
		}
	}
	while(steps) {
		if(graphs && steps >= (unsigned)GRAPH_STEPS) {
			if(int r = ensure_clean(c))
				return r;
			HIPCHK(c, hipGraphLaunch(c->gexec[0], c->stream));
			c->others_clean = c->root_clean = c->consume_ok;
			steps -= GRAPH_STEPS;
			c->stats.fragments += (uint64_t)c->nfrags * GRAPH_STEPS;
			c->stats.voice_fragments += (uint64_t)c->nfrags * (c->list_all.size() - c->n_list_pads) * GRAPH_STEPS;
		} else if(graphs) {
			if(int r = ensure_clean(c))
				return r;
			HIPCHK(c, hipGraphLaunch(c->gexec[1], c->stream));
			c->others_clean = c->root_clean = c->consume_ok;
			--steps;
			c->stats.fragments += c->nfrags;
			c->stats.voice_fragments += (uint64_t)c->nfrags * (c->list_all.size() - c->n_list_pads);
		} else {
			if(c->profiling) {
				if(c->ev_used + 3 > c->ev_pool.size())
					for(int i = 0; i < 3; ++i) {
						hipEvent_t e;
						HIPCHK(c, hipEventCreate(&e));
						c->ev_pool.push_back(e);
					}
				c->ev0 = c->ev_pool[c->ev_used];
				c->ev1 = c->ev_pool[c->ev_used + 1];
				c->ev2 = c->ev_pool[c->ev_used + 2];
				c->ev_used += 3;
			}
			if(int r = c->profiling ?
					issue_kernels(c, A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT, c->ev0, c->ev1, c->ev2) :
					issue_kernels(c, A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT, nullptr, nullptr, nullptr))
				return r;
			--steps;
		}
	}
	return A2AMD_OK;
}

// ---- multi-GPU: RCCL over xGMI, called from here (no framework in the data path) -------
int a2amd_dist_unique_id(void *id128)
{
	if(!id128 || !rccl_bind()) {
		snprintf(g_err, sizeof(g_err), "a2amd_dist_unique_id: RCCL (librccl.so) is not available");
		return A2AMD_ENODEVICE;
	}
	static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
	ncclResult_t r = g_rccl.GetUniqueId((ncclUniqueId *)id128);
	if(r != ncclSuccess) {
		snprintf(g_err, sizeof(g_err), "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
		return A2AMD_EHIP;
	}
	return A2AMD_OK;
}

int a2amd_dist_init(a2amd_ctx *c, const void *id128, int rank, int nranks)
{
	use_device(c);
	if(!id128 || rank < 0 || rank >= nranks)
		return c->fail(A2AMD_EINVAL, "dist_init: rank %d of %d", rank, nranks);
	if(c->comm)
		return c->fail(A2AMD_ESTATE, "dist_init: already initialised");
	if(!rccl_bind())
		return c->fail(A2AMD_ENODEVICE, "dist_init: RCCL (librccl.so) is not available");
	ncclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	// (RCCL checks the runtime's last-error slot as it goes: an error some earlier,
	// unrelated call of this process left there must not become its "unhandled cuda error")
	(void)hipGetLastError();
	ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
	if(r != ncclSuccess) {
		c->comm = nullptr;
		return c->fail(A2AMD_EHIP, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
	}
	c->dist_rank = rank;
	c->dist_ranks = nranks;
	drop_graphs(c);
	c->blob_quiet = false;
	return A2AMD_OK;
}

int a2amd_dist_init_local(a2amd_ctx *const *ctxs, int n)
{
	if(!ctxs || n < 1 || n > 64)
		return A2AMD_EINVAL;
	a2amd_ctx *c0 = ctxs[0];
	bool distinct = true;
	for(int i = 0; i < n; ++i) {
		if(ctxs[i]->comm || ctxs[i]->dist_local)
			return c0->fail(A2AMD_ESTATE, "dist_init_local: context %d is already part of a group", i);
		for(int k = 0; k < i; ++k)
			if(ctxs[k]->cfg.device == ctxs[i]->cfg.device)
				distinct = false;
	}
	if(n > 1 && distinct) {
		// one communicator per GPU, all in this process (ncclCommInitAll)
		if(!rccl_bind() || !g_rccl.CommInitAll || !g_rccl.GroupStart || !g_rccl.GroupEnd)
			return c0->fail(A2AMD_ENODEVICE, "dist_init_local: RCCL (librccl.so) is not available");
		std::vector<ncclComm_t> comms(n);
		std::vector<int> devs(n);
		for(int i = 0; i < n; ++i)
			devs[i] = ctxs[i]->cfg.device;
		(void)hipGetLastError();	// (see a2amd_dist_init)
		ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, devs.data());
		if(r != ncclSuccess)
			return c0->fail(A2AMD_EHIP, "ncclCommInitAll: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
		for(int i = 0; i < n; ++i)
			ctxs[i]->comm = comms[i];
	}
	// (contexts sharing a GPU - a test box - exchange with a device-local add instead:
	// RCCL refuses two ranks on one device)
	for(int i = 0; i < n; ++i) {
		a2amd_ctx *c = ctxs[i];
		use_device(c);
		c->dist_local = true;
		c->dist_rank = i;
		c->dist_ranks = n;
		if(!c->grp_ev)
			HIPCHK(c, hipEventCreateWithFlags(&c->grp_ev, hipEventDisableTiming));
		drop_graphs(c);
		c->blob_quiet = false;
	}
	return A2AMD_OK;
}

int a2amd_rootbus(a2amd_ctx *c, void **devptr, uint64_t *bytes);

int a2amd_render_group(a2amd_ctx *const *ctxs, int n, unsigned phases, int32_t *const *out, unsigned cap)
{
	if(!ctxs || n < 1)
		return A2AMD_EINVAL;
	a2amd_ctx *c0 = ctxs[0];
	if(n == 1 && !c0->dist_local)
		return a2amd_render(c0, phases, out, cap);
	for(int i = 0; i < n; ++i)
		if(!ctxs[i]->dist_local || ctxs[i]->dist_rank != i || ctxs[i]->dist_ranks != n)
			return c0->fail(A2AMD_ESTATE, "render_group: not the group of a2amd_dist_init_local()");
	const unsigned keep = phases & A2AMD_RENDER_KEEP;
	// every context: its commands up, its subtrees rendered (the kernels of the
	// different GPUs run side by side: nothing here waits)
	int frames = 0;
	for(int i = 0; i < n; ++i) {
		int r = a2amd_render(ctxs[i], (phases & (A2AMD_RENDER_UPLOAD | A2AMD_RENDER_SUBTREES)) | keep, nullptr, 0);
		if(r < 0) {
			if(i)
				c0->fail(r, "%s", ctxs[i]->err);
			return r;
		}
		if(i == 0)
			frames = r;
		else if(r != frames)
			return c0->fail(A2AMD_ESTATE, "render_group: context %d recorded %d frames, context 0 %d", i, r, frames);
	}
	if(!frames)
		return 0;
	if(phases & A2AMD_RENDER_SUBTREES) {
		// the exchange: the partials of the root voice's inline bus -> context 0
		std::vector<void *> bus(n);
		uint64_t bytes = 0;
		for(int i = 0; i < n; ++i) {
			uint64_t b;
			if(int r = a2amd_rootbus(ctxs[i], &bus[i], &b))
				return i ? c0->fail(r, "%s", ctxs[i]->err) : r;
			if(i && b != bytes)
				return c0->fail(A2AMD_ESTATE, "render_group: root buses of different shapes");
			bytes = b;
		}
		if(c0->comm) {
			// ONE ncclReduce(int32, sum) over xGMI: a group call, one thread drives all ranks
			ncclResult_t r = g_rccl.GroupStart();
			for(int i = 0; i < n && r == ncclSuccess; ++i) {
				use_device(ctxs[i]);
				r = g_rccl.Reduce(bus[i], bus[i], bytes / 4, ncclInt32, ncclSum, 0, ctxs[i]->comm, ctxs[i]->stream);
			}
			ncclResult_t r2 = g_rccl.GroupEnd();
			if(r != ncclSuccess || r2 != ncclSuccess)
				return c0->fail(A2AMD_EHIP, "ncclReduce: %s", g_rccl.GetErrorString ?
						g_rccl.GetErrorString(r != ncclSuccess ? r : r2) : "failed");
			for(int i = 1; i < n; ++i) {
				a2amd_ctx *c = ctxs[i];
				use_device(c);
				HIPCHK(c, hipMemsetAsync(bus[i], 0, bytes, c->stream));
				c->root_clean = true;
			}
		} else {
			// contexts sharing one GPU: context 0's stream adds the others' partials
			// (and clears them) once their subtrees are done
			use_device(c0);
			for(int i = 1; i < n; ++i) {
				a2amd_ctx *c = ctxs[i];
				HIPCHK(c, hipEventRecord(c->grp_ev, c->stream));
				HIPCHK(c0, hipStreamWaitEvent(c0->stream, c->grp_ev, 0));
				if(a2d_launch_add_bus((int32_t *)bus[0], (int32_t *)bus[i], (unsigned)(bytes / 4), c0->stream))
					return c0->fail(A2AMD_EHIP, "bus add launch failed");
				c->root_clean = true;
			}
			// (their next batch must not start before the partials were taken)
			HIPCHK(c0, hipEventRecord(c0->grp_ev, c0->stream));
			for(int i = 1; i < n; ++i)
				HIPCHK(ctxs[i], hipStreamWaitEvent(ctxs[i]->stream, c0->grp_ev, 0));
		}
	}
	// the root chain and the audio: context 0
	use_device(c0);
	int r = a2amd_render(c0, (phases & (A2AMD_RENDER_ROOT | A2AMD_RENDER_READBACK | A2AMD_RENDER_ASYNC)) | keep, out, cap);
	if(r < 0)
		return r;
	// sink clients on voices of the other contexts: their taps come back with the audio
	if((phases & A2AMD_RENDER_READBACK) && !(phases & A2AMD_RENDER_ASYNC))
		for(int i = 1; i < n; ++i) {
			use_device(ctxs[i]);
			if(int r2 = fetch_taps(ctxs[i], true))
				return c0->fail(r2, "%s", ctxs[i]->err);
		}
	if(!keep && (phases & (A2AMD_RENDER_READBACK | A2AMD_RENDER_ROOT)))
		for(int i = 1; i < n; ++i) {
			use_device(ctxs[i]);
			end_batch(ctxs[i]);
		}
	return frames;
}

// the exchange step of a batch: sum the ranks' partials of the root voice's inline bus
// into rank 0's (int32 wrap-around sum: any order gives the same bits)
static int dist_reduce_root(a2amd_ctx *c)
{
	void *bus;
	uint64_t bytes;
	if(int r = a2amd_rootbus(c, &bus, &bytes))
		return r;
	ncclResult_t r = g_rccl.Reduce(bus, bus, bytes / 4, ncclInt32, ncclSum, 0, c->comm, c->stream);
	if(r != ncclSuccess)
		return c->fail(A2AMD_EHIP, "ncclReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
	if(c->dist_rank != 0) {
		// our partial has been delivered: the bus starts the next batch empty
		HIPCHK(c, hipMemsetAsync(bus, 0, bytes, c->stream));
		c->root_clean = true;
	}
	return A2AMD_OK;
}

int a2amd_rootbus(a2amd_ctx *c, void **devptr, uint64_t *bytes)
{
	// the root voice is the depth-0 voice with an inline unit (a2_rootdriver,
	// audiality2.c:271-291)
	for(size_t vi = 0; vi < c->voices.size(); ++vi) {
		const HVoice &v = c->voices[vi];
		if(v.live && v.resolved && v.depth == 0 && v.own_off >= 0) {
			if(!c->d_busmem.d)
				return c->fail(A2AMD_ESTATE, "no batch uploaded yet");
			*devptr = c->d_busmem.d + v.own_off;
			*bytes = (uint64_t)c->nfrags * v.own_nch * A2D_FRAG * sizeof(int32_t);
			return A2AMD_OK;
		}
	}
	return c->fail(A2AMD_ESTATE, "no root voice with an inline bus");
}

int a2amd_rootbus_copy(a2amd_ctx *c, void *stage, int to_stage)
{
	void *bus;
	uint64_t bytes;
	use_device(c);
	if(int r = a2amd_rootbus(c, &bus, &bytes))
		return r;
	if(!stage)
		return c->fail(A2AMD_EINVAL, "no staging buffer");
	if(to_stage) {
		// park: copy out and leave the root's bus zeroed for the next SUBTREES phase
		if(a2d_launch_park((int32_t *)stage, (int32_t *)bus, (unsigned)(bytes / 4), c->stream))
			return c->fail(A2AMD_EHIP, "park launch failed: %s", hipGetErrorString(hipGetLastError()));
		c->root_clean = true;
	} else {
		HIPCHK(c, hipMemcpyAsync(bus, stage, bytes, hipMemcpyDeviceToDevice, c->stream));
		c->root_clean = false;
	}
	return A2AMD_OK;
}

static int drain_events(a2amd_ctx *c)
{
	if(!c->ev_used)
		return 0;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	for(size_t i = 0; i + 2 < c->ev_used + 0 && i + 2 < c->ev_pool.size(); i += 3) {
		float a = 0, b = 0;
		if(hipEventElapsedTime(&a, c->ev_pool[i], c->ev_pool[i + 1]) == hipSuccess &&
				hipEventElapsedTime(&b, c->ev_pool[i], c->ev_pool[i + 2]) == hipSuccess) {
			c->stats.timed_leaf_ms += a;
			c->stats.timed_all_ms += b;
			++c->stats.timed_batches;
		}
	}
	c->ev_used = 0;
	return 0;
}

int a2amd_get_stats(a2amd_ctx *c, a2amd_stats *st)
{
	if(c->profiling)
		if(int r = drain_events(c))
			return r;
	*st = c->stats;
	return A2AMD_OK;
}

int a2amd_set_profiling(a2amd_ctx *c, int on)
{
	if(int r = drain_events(c))
		return r;
	if(on) {
		c->stats.timed_leaf_ms = c->stats.timed_all_ms = 0;
		c->stats.timed_batches = 0;
	}
	c->profiling = on != 0;
	return A2AMD_OK;
}

} // extern "C"
