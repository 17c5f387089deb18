// a2amd_vmwin.hip - the device VM driving the window kernels' control state directly (round 5).
//
// k_vm_count / k_vm_emit (a2amd_vm.hip) make RECORDS of what a scripted voice's VM does - some eighty
// per voice and 64-fragment batch for a script that writes every 3 ms - which k_win_ctl (a2amd_win.hip)
// then reads back one by one: the interpreter runs twice (the count pass sizes the record array, the
// host waits for its total), 22 MB of records per batch of 16 384 voices are written and read, and two
// lane = voice walks of the batch follow each other, each as long as one wavefront's serial trip.
// For the voices the window kernels render anyway - the chains wtosc [wtosc] [filter12] panmix - this
// kernel is both at once:
//
//   k_vm_win<NOSC, FILT>   lane = VOICE.  The lane runs its voice's VM through the batch's fragments
//                          (run_batch, a2amd_vmcore.h - a2_VoiceProcess, core.c:1847-1880) and applies
//                          what the VM does - control writes, cutoff coefficients, windows - to the
//                          voice's control state (CtlVoice, a2amd_winctl.h) in the same lane, writing
//                          the closed-form window entries k_win_render reads.  What the emitter makes is
//                          still "records", in the order the engine makes its calls - but they live in
//                          a queue of a few per lane in LDS and are carried out at the end of every
//                          window (one copy of the code that carries them out, shared with k_win_ctl).
//
// No count pass, no host round trip in the middle of the batch, no record array.  A voice the VM leaves
// alone for the whole batch (asleep beyond its end, stopped in front of the engine's run) is not taken:
// runs[voice].count says which (1: ours, 0: the quiet kernels'), written before those are launched.
//
// Further windows of a fragment ("extras") need room in the launch's pool: a control pass takes one per
// record up front (it knows the count); here the writer wavefront takes what a fragment's staged extras
// need, one atomic per wavefront and fragment that has any, off the control wavefront's path.  A lane
// with more than WIN_EXL extras in one fragment (a VM that wakes more often than every 20 frames) parks
// the others in a row of its own in memory (wscr: 62 entries per lane, a fragment has no more windows
// than frames) and moves them to the pool when the fragment is done and their number known.  So the
// pool holds exactly one entry per further window - at most one per VM run that begins inside a
// fragment - which is what the host sized it by (k_vm_pool, a2amd_vm.hip: counted ahead of the batch).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <algorithm>
#include "a2amd_device.h"
#include "a2amd_vmcore.h"
#include "a2amd_vmdev.h"
#include "a2amd_winctl.h"

using namespace a2vm;

// records a lane keeps between two drains, and from how many on a VM run gives way to a drain: ONE instruction
// makes at most one record per control register of the chain and two for a cutoff (SETALL / RAMPALL over
// 2 x 4 wtosc + 5 filter12 + 2 panmix registers: 16 for the widest class), so the ring - VMW_RING, below - never
// overflows; the queue says so if it ever does (TRAP_RECORDS: the voice stops and the host reports the fault)
#define VMW_GIVEWAY 6
// (round 6, advisor) ... and behind a run, before the drain, run_batch adds the env units' writes, a cutoff's coefficient
// step per filter, the window itself and the late env writes.  Worst case of the largest class (2 x wtosc, filter12,
// panmix): VMW_GIVEWAY + (2 x 4 + 5 + 2 registers = 15, + 1 for a cutoff's second record) + 2 x A2D_VM_MAXENV +
// A2D_VM_MAXCUT + 1 (R_SEG) = 29.  The ring is that sum for the class, two to spare: 22 records for one oscillator
// without a filter, 26 for two, 27 with a filter (it was 32 for all: LDS the other kernels of a batch can use, WinStage)
#define VMW_REGS(NOSC, FILT) (4 * (NOSC) + 5 * (FILT) + 2)
#define VMW_RING(NOSC, FILT) (VMW_GIVEWAY + VMW_REGS(NOSC, FILT) + 1 + 2 * A2D_VM_MAXENV + A2D_VM_MAXCUT + 1 + 2)
#define VMW_ROW(FILT) (64 - 1 - WIN_EXLN(FILT))	/* a fragment's windows beyond the staged ones: in the lane's row of wscr */
static_assert(VMW_ROW(1) <= A2D_VMW_ROW && VMW_ROW(0) <= A2D_VMW_ROW, "the host sizes the rows of wscr by A2D_VMW_ROW");
#define VMW_CODEWORDS 1024		/* program text staged in LDS (a2amd_vmdev.h): 4 KB here */

template<int NOSC, int FILT>
struct VmwStage {
	static constexpr int RING = VMW_RING(NOSC, FILT);
	WinStage<NOSC, FILT> w;
	int own[2][64];		// the lane took the fragment's room in the pool itself (e0 says where)
	Int4 ring[RING][64];
};

template<int NOSC, int FILT>
struct WinE {
	static constexpr bool fused = true;
	CtlVoice<NOSC, FILT> &cv;
	A2DVmVoice &v;
	VmwStage<NOSC, FILT> &st;
	const A2DWave *waves;
	const PTab &ptab;
	int *ustate;
	int *wext;
	int *wrow;		// this lane's row of wscr
	unsigned *wtop;
	unsigned wcap;
	int lane, fa, fb;
	int n, total;		// records queued / made so far
	bool again;
	unsigned left;
	// the fragment under way
	int nwin, nstaged, own;
	unsigned head0, block;
#ifdef WIN_PROF
	long long t_drain = 0, t_meet = 0, t_end = 0, t_mark = 0, t_seg[3] = { 0, 0, 0 };
	int n_drain = 0, n_recs = 0;
	DEV void mark(int k)
	{
		const long long t = __builtin_readcyclecounter();
		if(k)
			t_seg[k] += t - t_mark;
		t_mark = t;
	}
#else
	VMFN void mark(int) {}
#endif

	VMFN void rec(int frag, int op, int unit, int reg, int value, unsigned dur, unsigned start)
	{
		const Int4 q = { (int)A2D_HEAD(frag, op, unit, reg), value, (int)dur, (int)start };
		if(n < VmwStage<NOSC, FILT>::RING)
			st.ring[n][lane] = q;
		else
			v.fault = TRAP_RECORDS;
		++n;
		++total;
	}
	VMFN int count() const { return total; }
	VMFN bool resuming() const { return again; }
	VMFN unsigned resume() { again = false; return left; }
	VMFN bool crowded() const { return n > VMW_GIVEWAY; }
	VMFN void yield(unsigned inscount) { left = inscount; again = true; }

	DEV void begin_fragment()
	{
		nwin = nstaged = own = 0;
		head0 = block = 0;
	}

	// carries out what the VM has done since the last time, in order
	DEV void drain(int f)
	{
		constexpr int SW = WIN_SW(NOSC, FILT);
		const int sb = (f - fa) & 1;
		const int m = n < VmwStage<NOSC, FILT>::RING ? n : VmwStage<NOSC, FILT>::RING;
#ifdef WIN_PROF
		const long long t0 = __builtin_readcyclecounter();
		++n_drain;
		n_recs += m;
#endif
		for(int k = 0; k < m; ++k) {
			const Int4 r = st.ring[k][lane];
			const int op = (int)A2D_ROP((unsigned)r.x);
			if(op == R_SEG) {
				if(!cv.active)
					continue;
				const unsigned dur = (unsigned)r.z;
				int W[SW];
				const unsigned head = ctl_window(cv, waves, ptab, (int)(dur & 0xffffu), (int)(dur >> 16), W);
				int *dst = nullptr;
				if(!nwin) {
					head0 = head;
					dst = st.w.slot[sb] + lane * SW;
				} else if(nstaged < WIN_EXLN(FILT))
					dst = st.w.ext[sb][lane][nstaged++];
				else if(nwin - 1 - WIN_EXLN(FILT) < VMW_ROW(FILT)) {
					own = 1;	// (parked: end_fragment moves them)
					dst = wrow + (size_t)(nwin - 1 - WIN_EXLN(FILT)) * A2D_WIN_WORDS;
				}
				++nwin;
				if(dst) {
#pragma unroll
					for(int j = 0; j < SW / 4; ++j) {
						const Int4 q = { W[4 * j], W[4 * j + 1], W[4 * j + 2], W[4 * j + 3] };
						((Int4 *)dst)[j] = q;
					}
				}
			} else
				ctl_apply(cv, waves, ptab, ustate, op, (int)A2D_RUNIT((unsigned)r.x), (int)A2D_RREG((unsigned)r.x), r.y,
						(unsigned)r.z, (unsigned)r.w);
		}
		n = 0;
#ifdef WIN_PROF
		t_drain += __builtin_readcyclecounter() - t0;
#endif
	}

	// the fragment's slot gets its head word, the writer wavefront the fragment
	DEV void end_fragment(int f)
	{
		const int sb = (f - fa) & 1;
#ifdef WIN_PROF
		const long long t0 = __builtin_readcyclecounter();
#endif
		if(FILT && cv.pending_fresh && f == fb - 1) {
			head0 |= WH_FRESH;
			cv.pending_fresh = 0;
		}
		int extras = nwin > 1 ? min(nwin - 1, WIN_EXLN(FILT) + VMW_ROW(FILT)) : 0;
		if(own) {
			// the fragment's further windows, now that their number is known: room in the pool, the staged ones and
			// the parked ones moved there
			constexpr int SW = WIN_SW(NOSC, FILT);
			block = atomicAdd(wtop, (unsigned)extras);
			if(block + (unsigned)extras > wcap) {
				atomicOr(wtop + 1, 1u);	// (no room - the host counted them ahead: never; they are lost, the flag says so)
				extras = 0;
			} else
				for(int q = 0; q < extras; ++q) {
					const Int4 *in = q < WIN_EXLN(FILT) ? (const Int4 *)st.w.ext[sb][lane][q] :
							(const Int4 *)(wrow + (size_t)(q - WIN_EXLN(FILT)) * A2D_WIN_WORDS);
					Int4 *o = (Int4 *)(wext + ((size_t)block + q) * A2D_WIN_WORDS);
#pragma unroll
					for(int j = 0; j < SW / 4; ++j)
						o[j] = in[j];
				}
		}
		st.w.slot[sb][lane * WIN_SW(NOSC, FILT) + WE_HEAD] = (int)(head0 | ((unsigned)extras << 19));
		st.w.e0[sb][lane] = block;
		st.w.nst[sb][lane] = own ? 0 : nstaged;
		st.own[sb][lane] = own;
#ifdef WIN_PROF
		const long long t1 = __builtin_readcyclecounter();
#endif
		win_meet();
#ifdef WIN_PROF
		t_meet += __builtin_readcyclecounter() - t1;
		t_end += t1 - t0;
#endif
		begin_fragment();
	}
};

// The writer: what win_ctl_writer does, and the pool room of the fragment's staged extras - a prefix sum over the
// lanes and one atomic - before they go out.  A wavefront that finds the pool full (the host sized it: never)
// takes the extras out of its voices' head words again, and the flag says so.
template<int NOSC, int FILT>
DEV void vmw_writer(int nlist, int first, int fa, int fb, int *__restrict__ wslot, int *__restrict__ wext,
		unsigned *__restrict__ widx, unsigned *__restrict__ wtop, unsigned wcap, VmwStage<NOSC, FILT> &st)
{
	constexpr int SW = WIN_SW(NOSC, FILT);
	const int lane = threadIdx.x & 63;
	const int nv = max(0, min(64, nlist - first));
	for(int f = fa; f < fb; ++f) {
		const int sb = (f - fa) & 1;
		win_meet();
		const int own = lane < nv ? st.own[sb][lane] : 0;
		int n = lane < nv ? st.w.nst[sb][lane] : 0;
		unsigned e0 = st.w.e0[sb][lane];
		if(__ballot(n != 0)) {
			int pre = n;
#pragma unroll
			for(int d = 1; d < 64; d <<= 1) {
				const int t = __shfl_up(pre, d, 64);
				if(lane >= d)
					pre += t;
			}
			const int total = __shfl(pre, 63, 64);
			unsigned base = 0;
			if(lane == 0)
				base = atomicAdd(wtop, (unsigned)total);
			base = (unsigned)__shfl((int)base, 0, 64);
			if(base + (unsigned)total > wcap) {
				if(lane == 0)
					atomicOr(wtop + 1, 1u);
				if(n) {
					int *h = st.w.slot[sb] + lane * SW + WE_HEAD;
					*h = (int)((unsigned)*h & ~(127u << 19));
				}
				n = 0;
			} else if(!own)
				e0 = base + (unsigned)(pre - n);
		}
		int *const dst = wslot + ((size_t)(f - fa) * nlist + first) * SW;
		for(int i = lane; i < nv * (SW / 4); i += 64)
			((Int4 *)dst)[i] = ((const Int4 *)st.w.slot[sb])[i];
		if(lane < nv) {
			widx[(size_t)(f - fa) * nlist + first + lane] = e0;
			for(int k = 0; k < n; ++k) {
				Int4 *o = (Int4 *)(wext + ((size_t)e0 + k) * A2D_WIN_WORDS);
				const Int4 *in = (const Int4 *)st.w.ext[sb][lane][k];
#pragma unroll
				for(int q = 0; q < SW / 4; ++q)
					o[q] = in[q];
			}
		}
	}
}

// Is the voice the VM's this batch?  Not if nothing will happen to it: the VM asleep beyond the batch's end (no
// run: a run needs waketime - t <= 255 at a window start t < end, run_batch), stopped by a fault or in front of
// the run that is the engine's, and no env unit or cutoff ramp under way that makes records without the VM.
DEV bool vmw_idle(const A2DVmVoice &v, uint32_t batch_end)
{
	if(v.nenv | v.ncut) {
		for(int k = 0; k < v.nenv && k < A2D_VM_MAXENV; ++k)
			if(v.env[k].active)
				return false;
		for(int k = 0; k < (int)v.ncut && k < A2D_VM_MAXCUT; ++k)
			// (the ramper's timer: still on its way - or just arrived: the window after a ramp's last one sets value =
			// target and delta = 0, rp_prepare, without a record; a write that found the old delta would start from
			// somewhere else, rp_set)
			if(v.cut[k][3] | v.cut[k][2])
				return false;
	}
	if(v.fault || (v.has_exit && v.waketime == v.exit_when))
		return true;
	return (int32_t)(v.waketime - batch_end) >= 0;
}

template<int NOSC, int FILT>
__global__ __launch_bounds__(128)
void k_vm_win(A2DVmParams vp, int fa, int fb, uint32_t now_fa, uint32_t batch_end,
		int *__restrict__ wslot, int *__restrict__ wext, int *__restrict__ wscr, unsigned *__restrict__ widx, unsigned *__restrict__ wtop, unsigned wcap,
		const A2DVoice *__restrict__ voices, int *ustate, int *vactive, const A2DWave *__restrict__ waves,
		const uint32_t *__restrict__ ptab, A2DVmwOut out)
{
	// (out: where the stepped state goes - the arrays it was read from, or a speculative pass's shadows; everything
	// between the loads at the top and the stores at the bottom lives in registers and LDS.  ctl_apply's one direct
	// write to ustate is a filter12's R_INIT, and the VM makes no births.)
	__shared__ PTab s_ptab;
	__shared__ VmSlot s_v[64];
	__shared__ VmwStage<NOSC, FILT> s_stage;
	__shared__ uint32_t s_code[VMW_CODEWORDS];
	__shared__ int s_anylive;
	for(int k = (int)threadIdx.x; k < 128; k += 128)
		s_ptab[k] = ptab[k];
	const int wave = rfl((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
	const int first = (int)blockIdx.x * 64, idx = first + lane;
	constexpr int SW = WIN_SW(NOSC, FILT);
	bool live = false;
	int slot = 0;
	const uint32_t *code = nullptr;
	if(wave == 0) {
		// (a lane without a voice of ours leaves an empty slot every fragment: its head words never change)
		s_stage.w.slot[0][lane * SW + WE_HEAD] = 0;
		s_stage.w.slot[1][lane * SW + WE_HEAD] = 0;
		s_stage.w.e0[0][lane] = s_stage.w.e0[1][lane] = 0;
		s_stage.w.nst[0][lane] = s_stage.w.nst[1][lane] = 0;
		s_stage.own[0][lane] = s_stage.own[1][lane] = 0;
		if(idx < vp.n) {
			slot = vp.list[idx];
			A2DVmVoice &v = s_v[lane].v;
			v = vp.vmv[slot];
			// (a later slab of the batch: the first one decided)
			live = fa > 0 ? vp.runs[v.voice].count != 0 : !vmw_idle(v, batch_end);
			if(fa == 0) {
				const A2DRun run = { 0, live ? 1 : 0 };
				out.runs[v.voice] = run;
			}
		}
		const unsigned long long any = __ballot(live);
		if(lane == 0)
			s_anylive = any != 0;
		if(out.idle && fa == 0) {
			// (a speculative pass tells the host how many of the class it left to the quiet kernels: none - every voice of a
			// scripted scene, every batch - and the quiet kernel of the class need not be launched to find that out)
			const unsigned long long idl = __ballot(idx < vp.n && !live);
			if(lane == 0 && idl)
				atomicAdd(out.idle, (uint32_t)__popcll(idl));
		}
		code = vm_stage_code(vp.code, s_v[lane].v, live, s_code, VMW_CODEWORDS);
	}
	__syncthreads();
	if(wave != 0) {
		vmw_writer<NOSC, FILT>(vp.n, first, fa, fb, wslot, wext, widx, wtop, wcap, s_stage);
		return;
	}
	if(!s_anylive) {
		// (none of the 64 is ours this batch: the writer still carries out their - empty - slots, the render pass
		// reads every slot of the list)
		for(int f = fa; f < fb; ++f)
			win_meet();
		return;
	}
	if(live) {
		A2DVmVoice &v = s_v[lane].v;
		CtlVoice<NOSC, FILT> cv;
		ctl_clear(cv);
		const A2DVoice &vc = voices[v.voice];
#pragma unroll
		for(int o = 0; o <= NOSC + FILT; ++o)
			cv.uu[o] = vc.unit[o];
		ctl_load(cv, ustate, vactive, v.voice);
#ifdef WIN_PROF
		const long long t_in = __builtin_readcyclecounter();
#endif
		const Consts K = { vp.msdur, vp.samplerate, vp.basepitch, vp.ptab, vp.f1tab, vp.envlut };
		WinE<NOSC, FILT> e = { cv, v, s_stage, waves, s_ptab, ustate, wext, wscr + (size_t)idx * A2D_VMW_ROW * A2D_WIN_WORDS, wtop, wcap,
				lane, fa, fb, 0, 0, false, 0u, 0, 0, 0, 0u, 0u };
		const uint8_t *ff = vp.fragframes, *fbs = vp.fragbase;
		run_batch(v, code, K, e, now_fa, fa, fb, [ff, fbs](int f) { return (unsigned)ff[f] | ((unsigned)fbs[f] << 8); },
				&s_v[lane].rt);
#ifdef WIN_PROF
		if(blockIdx.x % 61 == 5 && lane == 7)
			printf("k_vm_win<%d,%d> block %d lane 7: %d fragments, %lld cycles: %lld in %d drains of %d records, %lld at the end of "
					"fragments + %lld meeting the writer\n", NOSC, FILT, (int)blockIdx.x, fb - fa,
					(long long)__builtin_readcyclecounter() - t_in, e.t_drain, e.n_drain, e.n_recs, e.t_end, e.t_meet);
		if(blockIdx.x % 61 == 5 && lane == 7)
			printf("    ... %lld in the VM's runs, %lld in the windows' env / cutoff / SEG part\n", e.t_seg[1], e.t_seg[2]);
#endif
		ctl_store(cv, out.ustate, out.vactive, v.voice);
		out.vmv[slot] = v;
		if(v.fault)
			atomicAdd(out.total + 1, 1u);
	}
}

// A speculative pass's results become the state (round 6, vm_speculate / vm_issue): for every voice of the class,
// whose it is this batch (runs[]); for the ones the pass ran, the VM voice and exactly the unit words ctl_store
// writes - NOT whole unit blocks: filter12's d1 / d2 live in the same block and belong to the render pass, which
// has moved them since the shadow was written.
template<int NOSC, int FILT>
__global__ __launch_bounds__(64)
void k_vm_commit(A2DVmParams vp, A2DVmwOut from, const A2DVoice *__restrict__ voices, int *ustate, int *vactive)
{
	const int idx = (int)(blockIdx.x * 64 + threadIdx.x);
	if(idx >= vp.n)
		return;
	const int slot = vp.list[idx];
	const int voice = vp.vmv[slot].voice;
	const A2DRun run = from.runs[voice];
	vp.runs[voice] = run;
	if(!run.count)
		return;		// (idle in that batch: the pass did not touch it)
	vp.vmv[slot] = from.vmv[slot];
	CtlVoice<NOSC, FILT> cv;
	ctl_clear(cv);
	const A2DVoice &vc = voices[voice];
#pragma unroll
	for(int o = 0; o <= NOSC + FILT; ++o)
		cv.uu[o] = vc.unit[o];
	ctl_load(cv, from.ustate, from.vactive, voice);
	ctl_store(cv, ustate, vactive, voice);
}

int a2d_launch_vm_commit(const A2DVmParams &vp, const A2DParams &hp, int nosc, int filt, const A2DVmwOut &from, void *stream)
{
	if(vp.n <= 0)
		return 0;
	const int nblocks = (vp.n + 63) / 64;
#define VMC_LAUNCH(N, F) hipLaunchKernelGGL((k_vm_commit<N, F>), dim3(nblocks), dim3(64), 0, (hipStream_t)stream, vp, from, \
		hp.voices, hp.ustate, hp.vactive)
	if(nosc == 1 && !filt)
		VMC_LAUNCH(1, 0);
	else if(nosc == 2 && !filt)
		VMC_LAUNCH(2, 0);
	else if(nosc == 1 && filt)
		VMC_LAUNCH(1, 1);
	else
		return -1;
#undef VMC_LAUNCH
	return (int)hipGetLastError();
}

int a2d_launch_vm_win(const A2DVmParams &vp, const A2DParams &hp, int nosc, int filt, int fa, int fb, uint32_t now_fa,
		uint32_t batch_end, int *wslot, int *wext, int *wscr, unsigned *widx, unsigned *wtop, unsigned wcap, void *stream,
		const A2DVmwOut *outp)
{
	if(vp.n <= 0 || fb <= fa)
		return 0;
	const int nblocks = (vp.n + 63) / 64;
	const A2DVmwOut live = { vp.vmv, hp.ustate, hp.vactive, vp.runs, vp.total, nullptr };
	const A2DVmwOut out = outp ? *outp : live;
#define VMW_LAUNCH(N, F) hipLaunchKernelGGL((k_vm_win<N, F>), dim3(nblocks), dim3(128), 0, (hipStream_t)stream, vp, fa, fb, now_fa, \
		batch_end, wslot, wext, wscr, widx, wtop, wcap, hp.voices, hp.ustate, hp.vactive, hp.waves, hp.ptab, out)
	if(nosc == 1 && !filt)
		VMW_LAUNCH(1, 0);
	else if(nosc == 2 && !filt)
		VMW_LAUNCH(2, 0);
	else if(nosc == 1 && filt)
		VMW_LAUNCH(1, 1);
	else
		return -1;
#undef VMW_LAUNCH
	return (int)hipGetLastError();
}
