// a2amd_vm.cpp - host side of the scripted voice's VM on the device (include/a2amd_vm.h):
// the proof that a program stays inside the subset (a2amd_vm_analyze), the program texts and voice
// states that live on the device, adoption and recall, and the two passes of the VM kernel per
// batch.  The interpreter itself is a2amd_vmcore.h - compiled here for the host (a voice adopted
// or taken back in the middle of a batch is carried over the rest / the start of that batch by the
// host's copy, which writes the same records into the voice's host record list) and in
// a2amd_vm.hip for the device.
//
// Life of an adopted voice:
//   adopt    (fragment f of the batch being recorded; the engine has processed the voice in f)
//            -> "pending": counts as a sleeping voice whose default window the host reports
//               (a2amd_default_hold / the default map) from f + 1 on
//   upload   the host interpreter runs the voice over fragments f + 1 .. end of the batch (records
//            into HVoice::recs), the resulting state goes to the device, the voice joins the list
//   batches  k_vm (a2amd_vm.hip): count pass, host reads the total, emit pass -> records in the
//            blob's VM region + runs[voice]; the leaf kernels execute them like host records
//   recall   the device's state as the last batch left it comes back (one copy per voice, or the
//            whole array once several are wanted), the host interpreter runs it over the fragments
//            of the open batch recorded so far, the engine gets its A2_vmstate; this batch the
//            voice is rendered from host records
#include "a2amd_host.h"
#include "a2amd_vmcore.h"

using namespace a2vm;

namespace {

uint64_t text_sum(const uint32_t *code, unsigned n)
{
	uint64_t h = 0xcbf29ce484222325ull;
	for(unsigned i = 0; i < n; ++i)
		h = (h ^ code[i]) * 0x100000001b3ull;
	return h;
}

unsigned ins_size(unsigned op)		// a2_InsSize, src/compiler.c:111-131
{
	switch(op) {
	  case A2AMD_OP_DELAY: case A2AMD_OP_TDELAY: case A2AMD_OP_LOAD: case A2AMD_OP_ADD: case A2AMD_OP_MUL:
	  case A2AMD_OP_MOD: case A2AMD_OP_QUANT: case A2AMD_OP_RAND: case A2AMD_OP_PUSH: case A2AMD_OP_DEBUG:
	  case A2AMD_OP_RAMP: case A2AMD_OP_RAMPALL:
		return 2;
	  default:
		return 1;
	}
}

// records of the host interpreter go where the recorder's go: the voice's list, fragment order
struct HostE : a2vm::PlainE {
	a2amd_ctx *c;
	int vi;
	void rec(int frag, int op, int unit, int reg, int value, unsigned dur, unsigned start)
	{
		HVoice &v = c->voices[vi];
		A2DRec r;
		r.head = A2D_HEAD(frag, op, unit, reg);
		r.value = value;
		r.dur = dur;
		r.start = start;
		if(!v.listed_recs) {
			v.listed_recs = true;
			c->with_recs.push_back(vi);
		}
		v.recs.push_back(r);
	}
	int count() const { return (int)c->voices[vi].recs.size(); }
};

Consts consts_of(a2amd_ctx *c)
{
	Consts K;
	K.msdur = c->vm.msdur;
	K.samplerate = c->cfg.samplerate;
	K.basepitch = c->cfg.basepitch;
	K.ptab = c->ptab;
	K.f1tab = c->vm.f1tab.empty() ? nullptr : c->vm.f1tab.data();
	K.envlut = c->vm.envlut.empty() ? nullptr : c->vm.envlut.data();
	return K;
}

// f12_pitch2coeff (filter12.c:65-72) for everything a2_P2I (pitch.c:57-67) can return: a2_P2I's
// result is X(n) >> ((7 - oct) & 31) with X a function of the 16 fraction bits n alone, so 32 x
// 65536 entries cover every pitch - made here with the reference's own float / double / libm
// expression (a2amd_host.h: f12_coeff), which is why the device never evaluates a sine
static uint32_t ptab_sum(const uint32_t *ptab)
{
	uint32_t sum = 0;
	for(int i = 0; i < 128; ++i)
		sum = sum * 31u + ptab[i];
	return sum;
}

static std::vector<int32_t> make_f1tab(const uint32_t *ptab, int samplerate)
{
	std::vector<int32_t> t((size_t)32 * 65536);
	for(unsigned sh = 0; sh < 32; ++sh) {
		// a pitch whose a2_P2I shift count is sh: oct = 7 - sh (any oct with (7 - oct) & 31 == sh)
		const int oct = 7 - (int)sh;
		for(unsigned n = 0; n < 65536; ++n) {
			const int pitch = (int)(((unsigned)oct << 16) | n);
			// (f12_coeff takes the ramper value: the pitch is value >> 8)
			float f = a2h::p2i(ptab, pitch) * (261.626f / 16777216.0f);
			int v;
			if(f > (samplerate >> 2))
				v = 362 << 16;
			else
				v = (int)(512.0f * 65536.0f * sin(M_PI * f / samplerate));
			t[(size_t)sh * 65536 + n] = v;
		}
	}
	return t;
}

// (a2amd_open / a2amd_set_pitch_table: the table is on its way long before the first filter voice is adopted)
void start_f1tab(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	static const bool off = getenv("A2AMD_NO_VM") != nullptr;
	if(off)
		return;
	struct Args { uint32_t ptab[128]; int sr; };
	Args a;
	memcpy(a.ptab, c->ptab, sizeof(a.ptab));
	a.sr = c->cfg.samplerate;
	m.f1_future_sum = ptab_sum(c->ptab);
	m.f1_future = std::async(std::launch::async, [a]() { return make_f1tab(a.ptab, a.sr); });
}

void build_f1tab(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	const uint32_t sum = ptab_sum(c->ptab);
	if(!m.f1tab.empty() && m.f1tab_sum == sum)
		return;
	if(m.f1_future.valid() && m.f1_future_sum == sum)
		m.f1tab = m.f1_future.get();
	else
		m.f1tab = make_f1tab(c->ptab, c->cfg.samplerate);
	m.f1tab_sum = sum;
	m.f1tab_up = false;
}

int analyze(const uint32_t *code, unsigned nwords, unsigned pc0, int32_t tick, uint32_t msdur, a2amd_vm_info *info)
{
	memset(info, 0, sizeof(*info));
	info->opcode = info->at = -1;
	if(!code || pc0 >= nwords || nwords > 65535) {
		info->at = (int)pc0;
		return info->reason = A2AMD_VM_BADCODE;
	}
	// pass 1: what is reachable, and is it all inside the subset
	std::vector<uint8_t> seen(nwords, 0);
	std::vector<unsigned> work(1, pc0), order;
	seen[pc0] = 1;
	uint64_t written = 0, marked = 0, explicit_ctl = 0;
	bool any_all = false, any_timing = false;
	auto fail = [&](int why, unsigned op, unsigned at) {
		info->opcode = (int)op;
		info->at = (int)at;
		return info->reason = why;
	};
	while(!work.empty()) {
		const unsigned pc = work.back();
		work.pop_back();
		order.push_back(pc);
		const uint32_t w = code[pc];
		const unsigned op = A2AMD_VM_OPCODE(w), a1 = A2AMD_VM_A1(w), a2 = A2AMD_VM_A2(w), sz = ins_size(op);
		if(pc + sz > nwords)
			return fail(A2AMD_VM_BADCODE, op, pc);
		const int32_t a3 = sz == 2 ? (int32_t)code[pc + 1] : 0;
		unsigned succ[2], ns = 0;
		bool regs_a1 = false, regs_a2 = false;
		switch(op) {
		  case A2AMD_OP_JUMP:
			succ[ns++] = a2;
			break;
		  case A2AMD_OP_LOOP:
			written |= 1ull << (a1 & 63);
			regs_a1 = true;
			succ[ns++] = a2;
			succ[ns++] = pc + sz;
			break;
		  case A2AMD_OP_JZ: case A2AMD_OP_JNZ: case A2AMD_OP_JG: case A2AMD_OP_JL: case A2AMD_OP_JGE: case A2AMD_OP_JLE:
			regs_a1 = true;
			succ[ns++] = a2;
			succ[ns++] = pc + sz;
			break;
		  case A2AMD_OP_DELAYR: case A2AMD_OP_TDELAYR:
			regs_a1 = true;
			// fall through
		  case A2AMD_OP_DELAY: case A2AMD_OP_TDELAY:
			any_timing = true;
			succ[ns++] = pc + sz;
			break;
		  case A2AMD_OP_SUBR: case A2AMD_OP_P2DR: case A2AMD_OP_NEGR: case A2AMD_OP_LOADR: case A2AMD_OP_ADDR:
		  case A2AMD_OP_MULR: case A2AMD_OP_GR: case A2AMD_OP_LR: case A2AMD_OP_GER: case A2AMD_OP_LER:
		  case A2AMD_OP_EQR: case A2AMD_OP_NER: case A2AMD_OP_ANDR: case A2AMD_OP_ORR: case A2AMD_OP_XORR:
		  case A2AMD_OP_NOTR:
			regs_a2 = true;
			// fall through
		  case A2AMD_OP_LOAD: case A2AMD_OP_ADD: case A2AMD_OP_MUL:
			regs_a1 = true;
			written |= 1ull << (a1 & 63);
			marked |= 1ull << (a1 & 63);
			succ[ns++] = pc + sz;
			break;
		  case A2AMD_OP_MOD: case A2AMD_OP_QUANT:
			if(a3 == 0 || a3 == -1)
				return fail(A2AMD_VM_DIVISOR, op, pc);
			regs_a1 = true;
			written |= 1ull << (a1 & 63);
			marked |= 1ull << (a1 & 63);
			succ[ns++] = pc + sz;
			break;
		  case A2AMD_OP_SET: case A2AMD_OP_RAMP:
			regs_a1 = true;
			explicit_ctl |= 1ull << (a1 & 63);
			succ[ns++] = pc + sz;
			break;
		  case A2AMD_OP_RAMPR:
			regs_a1 = regs_a2 = true;
			explicit_ctl |= 1ull << (a1 & 63);
			succ[ns++] = pc + sz;
			break;
		  case A2AMD_OP_RAMPALLR:
			regs_a1 = true;
			// fall through
		  case A2AMD_OP_SETALL: case A2AMD_OP_RAMPALL:
			any_all = true;
			succ[ns++] = pc + sz;
			break;
		  default:
			return fail(A2AMD_VM_OPCODE_OUT, op, pc);
		}
		if((regs_a1 && a1 >= A2AMD_VM_REGISTERS) || (regs_a2 && a2 >= A2AMD_VM_REGISTERS))
			return fail(A2AMD_VM_BADCODE, op, pc);
		for(unsigned k = 0; k < ns; ++k) {
			if(succ[k] >= nwords)
				return fail(A2AMD_VM_BADCODE, op, pc);
			if(!seen[succ[k]]) {
				seen[succ[k]] = 1;
				work.push_back(succ[k]);
			}
		}
	}
	info->reachable = (uint32_t)order.size();
	info->written = written;
	// what a2_VoiceControl may be called for: the explicit targets, and - at every timing
	// instruction and every SETALL / RAMPALL - whatever the register tracker holds: any register an
	// arithmetic instruction stores to.  (The tracker's mask is 32 bits wide: a register's twin 32
	// places up or down can stand in for it, core.c:1076-1083.)
	info->controlled = explicit_ctl;
	if(any_timing || any_all) {
		info->controlled |= marked;
	}
	// pass 2: no cycle without a certain yield, no 1000 instructions between two of them
	// (A2_OVERLOAD, core.c:1187-1190: the 1000th instruction of a VM run aborts the voice).
	// A certain yield ends a VM run: DELAY with a non-zero duration, TDELAY with one while
	// nothing reachable stores to R_TICK.
	const bool tick_const = !(written & (1ull << A2AMD_VM_R_TICK));
	auto certain_yield = [&](unsigned pc) {
		const unsigned op = A2AMD_VM_OPCODE(code[pc]);
		if(op == A2AMD_OP_DELAY)
			return ms2t(msdur, (int32_t)code[pc + 1]) != 0;
		if(op == A2AMD_OP_TDELAY)
			return tick_const && ticks2t(msdur, tick, (int32_t)code[pc + 1]) != 0;
		return false;
	};
	// longest[pc] = most instructions a VM run can execute from pc on, pc included (a yield counts
	// and ends the run).  Depth first with an explicit stack; a grey node reached again = a cycle.
	std::vector<int> longest(nwords, -1);
	std::vector<uint8_t> colour(nwords, 0);
	struct Frame { unsigned pc; unsigned k; };
	uint32_t worst = 0;
	for(unsigned start : order) {
		// (every reachable instruction may begin a run: the successor of a yield does, and pc0)
		if(longest[start] >= 0)
			continue;
		std::vector<Frame> st(1, Frame{ start, 0 });
		colour[start] = 1;
		while(!st.empty()) {
			Frame &fr = st.back();
			const unsigned pc = fr.pc;
			const uint32_t w = code[pc];
			const unsigned op = A2AMD_VM_OPCODE(w), a2 = A2AMD_VM_A2(w), sz = ins_size(op);
			unsigned succ[2], ns = 0;
			if(!certain_yield(pc)) {
				if(op == A2AMD_OP_JUMP)
					succ[ns++] = a2;
				else if(op == A2AMD_OP_LOOP || (op >= A2AMD_OP_JZ && op <= A2AMD_OP_JLE)) {
					succ[ns++] = a2;
					succ[ns++] = pc + sz;
				} else
					succ[ns++] = pc + sz;
			}
			if(fr.k < ns) {
				const unsigned nx = succ[fr.k++];
				if(colour[nx] == 1)
					return fail(A2AMD_VM_NOYIELD, op, pc);
				if(colour[nx] == 0) {
					colour[nx] = 1;
					st.push_back(Frame{ nx, 0 });
				}
				continue;
			}
			int best = 0;
			for(unsigned k = 0; k < ns; ++k)
				best = std::max(best, longest[succ[k]]);
			longest[pc] = best + 1;
			colour[pc] = 2;
			worst = std::max(worst, (uint32_t)longest[pc]);
			st.pop_back();
		}
	}
	info->longest = worst;
	if(worst >= A2AMD_VM_INSLIMIT)
		return fail(A2AMD_VM_NOYIELD, 0, pc0);
	return info->reason = A2AMD_VM_OK;
}

const char *reason_text(int r)
{
	static const char *const t[] = { "ok", "bad code", "instruction outside the subset", "MOD / QUANT by 0 or -1",
		"a loop without a certain delay (A2_OVERLOAD cannot be ruled out)", "writes a unit register the device VM does not",
		"never writes a unit register" };
	return r >= 0 && r <= A2AMD_VM_IDLE ? t[r] : "?";
}

// Where the device VM's stay ends for a voice the static analysis cannot vouch for: the host's copy of the
// interpreter runs the voice's FUTURE VM runs, one after the other, on a copy of its state - nothing but
// its own registers feeds them (no events: an event recalls the voice; no RAND; the env units and cutoff
// rampers do not read back), so they are exactly the runs the device will make - until one of them
// meets what the device cannot do: an instruction outside the subset (END, SLEEP, RETURN, CALL, WAKE, FORCE,
// spawning, messages, RAND, DEBUG ...), a zero divisor, A2_OVERLOAD, a write through a register wired to
// something the device VM does not write.  That run is the ENGINE'S: *exit_when = the wake time it starts
// at, and the walk (a2amd_walk.c) takes the voice back in the fragment that holds it - so a note that ends
// is the device's up to its last wake-up, and END itself (core.c:1191-1235: the attached / finalizing
// logic, a2_VoiceFree) is done by the engine, at the frame the engine would have done it.  Without such a
// run within VM_HORIZON ticks (or VM_MAXRUNS runs) the stay ends there; the voice may be taken again.
// *records = what the runs emit (a stay without any is not worth a hand-over).
#define VM_HORIZON	(1 << 29)	// 24:8 ticks: 43 s at 48 kHz
#define VM_MAXRUNS	8192
#define VM_MIN_STAY	2048		// frames: a shorter stay is not worth a hand-over and a recall
static void vm_lookahead(const A2DVmVoice &d0, const uint32_t *code, const Consts &K, uint32_t now, uint32_t *exit_when,
		int *records, int *runs_out)
{
	A2DVmVoice d = d0;
	CountE e = { {}, 0 };
	int runs = 0;
	*exit_when = d.waketime;
	for(; runs < VM_MAXRUNS; ++runs) {
		if((int)(d.waketime - now) > VM_HORIZON)
			break;
		const uint32_t at = d.waketime;
		const int n0 = e.n;
		if(run(d, code, K, e, 0)) {
			e.n = n0;		// (that run's writes are not the device's)
			*exit_when = at;
			*records = e.n;
			*runs_out = runs;
			return;
		}
	}
	*exit_when = d.waketime;
	*records = e.n;
	*runs_out = runs;
}

int grow_stage(a2amd_ctx *c, size_t n)
{
	VmHost &m = c->vm;
	if(n <= m.h_stage_cap)
		return 0;
	if(m.h_stage)
		HIPCHK(c, hipHostFree(m.h_stage));
	m.h_stage = nullptr;
	m.h_stage_cap = 0;
	const size_t cap = std::max(n, (size_t)64);
	HIPCHK(c, hipHostMalloc((void **)&m.h_stage, cap * sizeof(A2DVmVoice), hipHostMallocDefault));
	m.h_stage_cap = cap;
	return 0;
}

// the state of an ACTIVE voice (one the device runs) as the last rendered batch left it
int fetch_state(a2amd_ctx *c, int slot, A2DVmVoice *out)
{
	VmHost &m = c->vm;
	use_device(c);
	if(m.snap_have.size() < m.vms.size()) {
		m.snap_have.resize(m.vms.size(), 0);
		m.snap.resize(m.vms.size());
	}
	if(!m.snap_have[slot]) {
		if(++m.snap_fetches > 8 && !m.list.empty()) {
			// several are wanted (a chord released, a group killed): all of them, once
			const size_t n = std::min(m.vms.size(), m.d_vmv.cap);
			if(int r = grow_stage(c, n))
				return r;
			HIPCHK(c, hipMemcpyAsync(m.h_stage, m.d_vmv.d, n * sizeof(A2DVmVoice), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			for(size_t k = 0; k < n; ++k)
				if(!m.snap_have[k]) {
					m.snap[k] = m.h_stage[k];
					m.snap_have[k] = 1;
				}
		} else {
			if(int r = grow_stage(c, 1))
				return r;
			HIPCHK(c, hipMemcpyAsync(m.h_stage, m.d_vmv.d + slot, sizeof(A2DVmVoice), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			m.snap[slot] = m.h_stage[0];
			m.snap_have[slot] = 1;
		}
	}
	*out = m.snap[slot];
	return 0;
}

} // namespace

namespace a2h {

// The voice vi goes back to the host in the open fragment g (or, outside any fragment, before the
// first fragment of the next batch): its VM state is brought to the start of g (inclusive: to the
// END of g - the voice has had its window in g, which the host reported by the map or a hold), the
// records of the fragments the device VM would have covered in this batch go into the voice's
// host record list, the filter cutoffs return to their host shadows.  out (may be null) = the
// engine's A2_vmstate.
int vm_take_back(a2amd_ctx *c, int vi, bool inclusive, a2amd_vm_state *out, a2amd_vm_env *envs_out)
{
	VmHost &m = c->vm;
	HVoice &v = c->voices[vi];
	const int slot = v.vm;
	if(slot < 0 || slot >= (int)m.vms.size() || !m.vms[slot].live)
		return c->fail(A2AMD_ESTATE, "voice %d is not run by the device VM", vi);
	HVm &h = m.vms[slot];
	A2DVmVoice st;
	int f0;
	if(h.pending) {
		st = h.st;
		f0 = h.adopt_frag + 1;
	} else {
		if(int r = fetch_state(c, slot, &st))
			return r;
		f0 = 0;
	}
	// fragments of the open batch the VM has to be carried over
	int f1 = c->frag_open ? c->cur_frag + (inclusive ? 1 : 0) : c->nfrags;
	if(f1 > c->nfrags)
		f1 = c->nfrags;
	if(f0 < f1) {
		const HVmProg &p = m.progs[h.prog];
		const Consts K = consts_of(c);
		HostE e = { {}, c, vi };
		uint64_t frames_before = 0;
		for(int f = 0; f < f0; ++f)
			frames_before += c->fragframes[f];
		const uint32_t now = m.t0 + (uint32_t)((m.batch_time + frames_before) << 8);
		const unsigned *ff = c->fragframes;
		const uint8_t *fb = c->fragbase;
		const size_t mark = c->voices[vi].recs.size();
		run_batch(st, m.code.data() + p.off, K, e, now, f0, f1, [ff, fb](int f) { return ff[f] | ((unsigned)fb[f] << 8); });
		if(st.fault)
			return c->fail(A2AMD_ESTATE, "device VM: voice %d faulted (trap %d at pc %u): the analysis let a program through "
					"that it should not have", vi, st.fault, (unsigned)st.pc);
		HVoice &vv = c->voices[vi];
		if(inclusive && c->frag_open) {
			// the open fragment's windows are on record now: nothing is left to be spelled out for it
			// (sched: spell_out_pending) - or, if the VM did nothing in it, exactly the default window
			bool any = false;
			for(size_t k = mark; k < vv.recs.size(); ++k)
				any |= (int)A2D_RFRAG(vv.recs[k].head) == c->cur_frag;
			const long long serial = c->serial_base + c->cur_frag;
			if(c->defmap_used && (size_t)vi < c->defmap.size() && c->defmap[vi])
				c->defmap[vi] = 0;
			vv.touched = serial;
			vv.frag_mark = any ? mark : vv.recs.size();
			vv.default_seg = any ? -1 : serial;
			if(vv.walked != serial) {
				vv.walked = serial;
				++c->walked_started;
			}
		}
	}
	unhold(c, vi);
	HVoice &vv = c->voices[vi];
	// the cutoff rampers go back to where host-driven voices keep them
	for(int k = 0; k < (int)st.ncut && k < A2D_VM_MAXCUT; ++k) {
		HUnit &u = c->units[vv.unit[st.cutpos[k]]];
		const bool was = u.cutoff.timer != 0;
		u.cutoff.value = st.cut[k][0];
		u.cutoff.target = st.cut[k][1];
		u.cutoff.delta = st.cut[k][2];
		u.cutoff.timer = st.cut[k][3];
		c->n_cutoff_ramps += (int)(u.cutoff.timer != 0) - (int)was;
	}
	vv.plain = 0;
	if(envs_out)
		for(int k = 0; k < A2AMD_VM_MAXENV; ++k) {
			a2amd_vm_env &o = envs_out[k];
			memset(&o, 0, sizeof(o));
			if(k >= st.nenv)
				continue;
			memcpy(o.ramper, st.env[k].ramper, sizeof(o.ramper));
			o.lut = st.env[k].lut;
			o.scale = st.env[k].scale;
			o.offset = st.env[k].offset;
			o.out = st.env[k].out;
			o.active = st.env[k].active;
			o.regbase = st.env[k].regbase;
			o.before = st.env[k].k;
		}
	if(out) {
		out->waketime = st.waketime;
		out->state = st.state;
		out->func = h.func;
		out->pc = st.pc;
		memcpy(out->r, st.r, sizeof(out->r));
	}
	// (its runs[] entry holds last batch's VM run: cleared with the next upload unless the voice
	// carries host records by then)
	if(!h.pending)
		c->prev_with_recs.push_back(vi);
	if(h.pending)
		m.pending.erase(std::remove(m.pending.begin(), m.pending.end(), slot), m.pending.end());
	else
		m.list_dirty = true;
	h.live = false;
	h.pending = false;
	m.free_slots.push_back(slot);
	if((size_t)slot < m.snap_have.size())
		m.snap_have[slot] = 0;
	vv.vm = -1;
	--m.stats.live;
	++m.stats.recalled;
	return A2AMD_OK;
}

// the first filter12 / dcblock unit of a context, a new pitch table: the coefficient table the device VM will
// want (a2vm::f1_of_pitch) starts being made, on a thread of its own
void vm_start_f1tab(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	if((m.f1_future.valid() && m.f1_future_sum == ptab_sum(c->ptab)) || (!m.f1tab.empty() && m.f1tab_sum == ptab_sum(c->ptab)))
		return;
	start_f1tab(c);
}

int vm_blob_room(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	if(m.list.empty() && m.pending.empty())
		return 0;
	// (a guess; the count pass says what is needed, vm_issue grows the region when it is more)
	const size_t n = m.list.size() + m.pending.size();
	const size_t want = n * (size_t)std::max(c->nfrags, 1) * 2 + 4096;
	return (int)std::min(std::max(want, (size_t)m.rec_cap), (size_t)1 << 28);
}

// upload(): the voices adopted during this batch are carried to its end by the host interpreter
// and sent up; the kernel's list and the records kernels' class lists follow the membership
// a speculative pass that may still be running reads what the caller is about to rewrite (VmHost::spec_waits)
static int vm_wait_spec(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	if(m.pred_stream && (m.spec_valid || m.pred_valid)) {
		use_device(c);
		HIPCHK(c, hipStreamSynchronize(m.pred_stream));
		++m.spec_waits;
	}
	return 0;
}

int vm_prepare_batch(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	if(m.pending.empty() && m.to_upload.empty() && m.code_uploaded == m.code.size() && (m.f1tab_up || m.f1tab.empty()) &&
			(m.envlut_up || m.envlut.empty()))
		return 0;
	use_device(c);
	if(int r = vm_wait_spec(c))
		return r;
	const std::vector<int> pend = m.pending;
	for(int slot : pend) {
		HVm &h = m.vms[slot];
		if(!h.live || !h.pending)
			continue;
		const HVmProg &p = m.progs[h.prog];
		const Consts K = consts_of(c);
		HostE e = { {}, c, h.voice };
		uint64_t frames_before = 0;
		for(int f = 0; f <= h.adopt_frag && f < c->nfrags; ++f)
			frames_before += c->fragframes[f];
		const uint32_t now = m.t0 + (uint32_t)((m.batch_time + frames_before) << 8);
		const unsigned *ff = c->fragframes;
		const uint8_t *fb = c->fragbase;
		run_batch(h.st, m.code.data() + p.off, K, e, now, h.adopt_frag + 1, c->nfrags, [ff, fb](int f) { return ff[f] | ((unsigned)fb[f] << 8); });
		if(h.st.fault)
			return c->fail(A2AMD_ESTATE, "device VM: voice %d faulted (trap %d at pc %u)", h.voice, h.st.fault, (unsigned)h.st.pc);
		h.pending = false;
		h.adopt_frag = -1;
		h.fresh = true;		// (this batch it is rendered from the host's records: the kernel's turn comes with the next)
		++m.n_fresh;
		m.to_upload.push_back(std::make_pair(slot, h.st));
	}
	m.pending.clear();
	if(m.code_uploaded != m.code.size()) {
		if(int r = grow(c, m.d_code, m.code.size(), 1, true))
			return r;
		HIPCHK(c, hipMemcpyAsync(m.d_code.d + m.code_uploaded, m.code.data() + m.code_uploaded,
				(m.code.size() - m.code_uploaded) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));	// (pageable source that may move)
		m.code_uploaded = m.code.size();
	}
	if(!m.f1tab.empty() && !m.f1tab_up) {
		if(!m.d_f1tab)
			HIPCHK(c, hipMalloc((void **)&m.d_f1tab, m.f1tab.size() * sizeof(int32_t)));
		HIPCHK(c, hipMemcpyAsync(m.d_f1tab, m.f1tab.data(), m.f1tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		m.f1tab_up = true;
	}
	if(!m.envlut.empty() && !m.envlut_up) {
		if(!m.d_envlut)
			HIPCHK(c, hipMalloc((void **)&m.d_envlut, m.envlut.size() * sizeof(uint16_t)));
		HIPCHK(c, hipMemcpyAsync(m.d_envlut, m.envlut.data(), m.envlut.size() * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		m.envlut_up = true;
	}
	if(!m.to_upload.empty()) {
		if(int r = grow(c, m.d_vmv, m.vms.size(), 1, true))
			return r;
		if(int r = grow_stage(c, m.to_upload.size()))
			return r;
		// (contiguous runs of slots go up in one copy each)
		std::sort(m.to_upload.begin(), m.to_upload.end(), [](const std::pair<int, A2DVmVoice> &a,
				const std::pair<int, A2DVmVoice> &b) { return a.first < b.first; });
		for(size_t k = 0; k < m.to_upload.size(); ++k)
			m.h_stage[k] = m.to_upload[k].second;
		for(size_t k = 0; k < m.to_upload.size();) {
			size_t e = k + 1;
			while(e < m.to_upload.size() && m.to_upload[e].first == m.to_upload[e - 1].first + 1)
				++e;
			HIPCHK(c, hipMemcpyAsync(m.d_vmv.d + m.to_upload[k].first, m.h_stage + k, (e - k) * sizeof(A2DVmVoice),
					hipMemcpyHostToDevice, c->stream));
			k = e;
		}
		HIPCHK(c, hipStreamSynchronize(c->stream));	// (the staging buffer is reused)
		m.to_upload.clear();
	}
	return 0;
}

int vm_build_lists(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	if(m.list_dirty) {
		use_device(c);
		if(int r = vm_wait_spec(c))
			return r;
		m.rebuilt_at = m.vm_batches;
		m.list.clear();
		++m.list_serial;
		std::vector<std::pair<int, int>> cls[3];	// (voice slot, VM slot)
		std::vector<int> other;
		m.o2f_voices.clear();
		for(size_t s = 0; s < m.vms.size(); ++s) {
			const HVm &h = m.vms[s];
			if(!h.live || h.pending || h.fresh)
				continue;
			m.list.push_back((int)s);
		}
		// the voices by launch class, each class sorted by output bus like the static lists
		// (the classes are set when upload() rebuilds its lists: call after that)
		for(int s : m.list) {
			const HVoice &v = c->voices[m.vms[s].voice];
			const int k = v.cls == CLS_OSCPAN ? 0 : v.cls == CLS_OSC2PAN ? 1 : v.cls == CLS_OSCFILTPAN ? 2 : -1;
			if(k >= 0)
				cls[k].push_back(std::make_pair(m.vms[s].voice, s));
			else {
				other.push_back(s);
				if(v.cls == CLS_OSC2FILTPAN)
					m.o2f_voices.push_back(m.vms[s].voice);
			}
		}
		m.cls_lists.clear();
		std::vector<int> cls_vm;
		for(int k = 0; k < 3; ++k) {
			auto by_bus = [&](const std::pair<int, int> &a, const std::pair<int, int> &b) {
				return c->voices[a.first].out_off < c->voices[b.first].out_off; };
			if(!std::is_sorted(cls[k].begin(), cls[k].end(), by_bus))	// (slots follow the walk: usually grouped already)
				std::stable_sort(cls[k].begin(), cls[k].end(), by_bus);
			m.n_cls[k] = (int)cls[k].size();
			for(const auto &pr : cls[k]) {
				m.cls_lists.push_back(pr.first);
				cls_vm.push_back(pr.second);
			}
		}
		m.n_other = (int)other.size();
		const size_t n = m.list.size() + 2 * m.cls_lists.size() + other.size();
		if(int r = grow(c, m.d_list, n + 64, 1, false))
			return r;
		if(int r = grow(c, m.d_vmrun, m.list.size() + 64, 1, false))
			return r;
		if(n) {
			std::vector<int> all = m.list;
			all.insert(all.end(), m.cls_lists.begin(), m.cls_lists.end());
			all.insert(all.end(), cls_vm.begin(), cls_vm.end());
			all.insert(all.end(), other.begin(), other.end());
			HIPCHK(c, hipMemcpyAsync(m.d_list.d, all.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
		}
		m.list_dirty = false;
	}
	return 0;
}

// issue_kernels(): count pass, the host reads the total (and grows the blob's VM region when the
// records do not fit), emit pass.  The leaf kernels that follow find runs[voice] and the records
// where host records would be.
// what the VM kernels are told beside their voice list: the whole list, this batch's time and fragments
static void fill_params(a2amd_ctx *c, A2DVmParams &vp)
{
	VmHost &m = c->vm;
	memset(&vp, 0, sizeof(vp));
	vp.vmv = m.d_vmv.d;
	vp.list = m.d_list.d;
	vp.n = (int32_t)m.list.size();
	vp.code = m.d_code.d;
	vp.ptab = c->d_ptab;
	vp.f1tab = m.d_f1tab;
	vp.envlut = m.d_envlut;
	vp.runs = c->d_runs.d;
	vp.vmrun = m.d_vmrun.d;
	vp.recs = (A2DRec *)c->hparams.recs;
	vp.total = m.d_total;
	vp.now = m.t0 + (uint32_t)((m.batch_time + m.replayed) << 8);
	vp.msdur = m.msdur;
	vp.samplerate = c->cfg.samplerate;
	vp.basepitch = c->cfg.basepitch;
	vp.nfrags = c->nfrags;
	for(int f = 0; f < c->nfrags; ++f) {
		vp.fragframes[f] = (uint8_t)c->fragframes[f];
		vp.fragbase[f] = c->fragbase[f];
	}
}

// k_vm_win's parameters for window class k (0: wtosc-panmix, 1: 2 x wtosc-panmix, 2: wtosc-filter12-panmix) of a
// fused batch (vm_issue has run: the batch's time is the one it used)
void vm_class_params(a2amd_ctx *c, int k, A2DVmParams *vp)
{
	VmHost &m = c->vm;
	fill_params(c, *vp);
	vp->now = m.batch_now;
	const int *l = m.d_list.d + m.list.size() + m.cls_lists.size();
	for(int q = 0; q < k; ++q)
		l += m.n_cls[q];
	vp->list = l;
	vp->n = m.n_cls[k];
}

// Behind a batch's window kernels: what k_vm_win would take from the pool for the NEXT batch if that one has this
// one's fragments, begins where this one ends and finds the lists as they are - all of which vm_issue checks before
// it believes the number.  On a stream of its own: nothing of this batch waits for it.
int vm_predict(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	// (round 6: vm_speculate predicts by doing - and it may already have, from inside issue_windows: its flags are not
	// this function's to clear)
	if(m.spec_launched_now || vm_spec_wanted(c))
		return 0;
	m.pred_valid = m.spec_valid = false;
	const int ncls = m.n_cls[0] + m.n_cls[1] + m.n_cls[2];
	if(m.list.empty() || !ncls || c->capturing || m.fused_off)
		return 0;
	if(!m.pred_stream) {
		HIPCHK(c, hipStreamCreateWithFlags(&m.pred_stream, hipStreamNonBlocking));
		HIPCHK(c, hipEventCreateWithFlags(&m.pred_after, hipEventDisableTiming));
		HIPCHK(c, hipEventCreateWithFlags(&m.pred_ev, hipEventDisableTiming));
		HIPCHK(c, hipMalloc((void **)&m.d_pred, 2 * sizeof(unsigned)));
		HIPCHK(c, hipHostMalloc((void **)&m.h_pred, 2 * sizeof(unsigned), hipHostMallocDefault));
	}
	A2DVmParams vp;
	fill_params(c, vp);
	uint32_t frames = 0;
	for(int f = 0; f < c->nfrags; ++f)
		frames += c->fragframes[f];
	if((frames + A2D_FRAG - 1) / A2D_FRAG > A2D_MAXBATCH)
		return 0;
	vp.now = m.batch_now + (frames << 8);
	vp.nfrags = 0;
	for(uint32_t at = 0; at < frames; at += A2D_FRAG) {	// (the stretch this batch covered, uncut)
		vp.fragframes[vp.nfrags] = (uint8_t)std::min<uint32_t>(A2D_FRAG, frames - at);
		vp.fragbase[vp.nfrags++] = 0;
	}
	vp.list = m.d_list.d + m.list.size() + m.cls_lists.size();	// the three classes' VM slots, one after the other
	vp.n = ncls;
	HIPCHK(c, hipEventRecord(m.pred_after, c->stream));
	HIPCHK(c, hipStreamWaitEvent(m.pred_stream, m.pred_after, 0));
	HIPCHK(c, hipMemsetAsync(m.d_pred, 0, 2 * sizeof(unsigned), m.pred_stream));
	if(a2d_launch_vm_pool(vp, m.d_pred, m.pred_stream))
		return c->fail(A2AMD_EHIP, "VM pool launch failed: %s", hipGetErrorString(hipGetLastError()));
	HIPCHK(c, hipMemcpyAsync(m.h_pred, m.d_pred, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, m.pred_stream));
	HIPCHK(c, hipEventRecord(m.pred_ev, m.pred_stream));
	m.pred_valid = true;
	m.pred_serial = m.list_serial;
	m.pred_now = vp.now;
	m.pred_span = frames;
	++c->stats.launches;
	return 0;
}

// The fragments of the batch expected next: this batch's span again (it has to be whole engine fragments or pieces of
// them - the test vm_issue makes of a prediction), beginning where this one ends, each 64-frame fragment cut where
// the root's VM is expected to wake inside it (a2amd_vm_expect_cuts; a2_VoiceProcess, core.c:1852-1878: from a fragment
// start F the engine processes (W - F) >> 8 frames up to the wake-up at W if W - F > 255 ticks, else it runs the VM
// first and the fragment stays whole).  The pass writes a slot per (fragment, voice), and "the next batch has these
// fragments" is something vm_issue checks word for word.
bool vm_spec_plan(a2amd_ctx *c, int *nfrags, uint8_t *ff, uint8_t *fb)
{
	VmHost &m = c->vm;
	uint32_t frames = 0;
	for(int f = 0; f < c->nfrags; ++f) {
		if(!c->fragframes[f] || frames / A2D_FRAG != (frames + c->fragframes[f] - 1) / A2D_FRAG)
			return false;
		frames += c->fragframes[f];
	}
	if(!frames || frames % A2D_FRAG)
		return false;
	const uint32_t next_now = m.batch_now + (frames << 8);
	int n = 0;
	for(uint32_t k = 0; k < frames / A2D_FRAG; ++k) {
		const uint32_t F = next_now + ((k * A2D_FRAG) << 8);
		int cut = 0, ncuts = 0;
		for(int h = 0; h < m.n_cut_hint; ++h) {
			const int32_t d = (int32_t)(m.cut_hint[h] - F);
			if(d > 255 && d < (A2D_FRAG << 8)) {
				cut = d >> 8;
				++ncuts;
			}
		}
		if(ncuts > 1 || n + 2 > A2D_MAXBATCH)
			return false;	// (two wake-ups inside one fragment, a list that does not fit: no prediction)
		if(ncuts) {
			ff[n] = (uint8_t)cut; fb[n++] = 0;
			ff[n] = (uint8_t)(A2D_FRAG - cut); fb[n++] = (uint8_t)cut;
		} else {
			ff[n] = A2D_FRAG; fb[n++] = 0;
		}
	}
	*nfrags = n;
	return true;
}

// May the batch being issued be followed by a speculative pass (VmHost, round 6)?
static int vm_spec_why_not(a2amd_ctx *c, int *nfrags, uint8_t *ff, uint8_t *fb)
{
	VmHost &m = c->vm;
	static const bool on = !(getenv("A2AMD_VMSPEC") && !atoi(getenv("A2AMD_VMSPEC")));
	const int ncls = m.n_cls[0] + m.n_cls[1] + m.n_cls[2];
	if(!on || m.list.empty() || !ncls || c->capturing || m.fused_off || c->nfrags < 1 || c->nfrags > A2D_MAXBATCH)
		return 1;
	// Short batches are left alone: k_vm_win over ONE fragment is a few tens of microseconds, and the commit plus a
	// pass that shares the chip with the batch's own kernels cost more than that (measured, 16 384 scripted voices,
	// a2_Run(64): 140 -> 164 us per call with the pass, 130 -> 137 with filter12; a2_Run(1024) = 16 fragments: 335 -> 295,
	// 496 -> 421; a2_Run(4096): 1 108 -> 904, 1 713 -> 1 360 - profiles/r06_vm_speculation_ab.txt).  A2AMD_VMSPEC_MIN moves it.
	static const int min_frags = getenv("A2AMD_VMSPEC_MIN") ? atoi(getenv("A2AMD_VMSPEC_MIN")) : 8;
	if(c->nfrags < min_frags)
		return 1;
	if(m.vm_batches - m.rebuilt_at < 2)	// (the lists have only just changed: VmHost::rebuilt_at)
		return 4;
	if(!vm_spec_plan(c, nfrags, ff, fb))
		return 2;
	// (the slots of the three classes for the whole batch in one piece: within the window kernels' own budget)
	static const size_t budget = (size_t)(getenv("A2AMD_WIN_MB") ? atoi(getenv("A2AMD_WIN_MB")) : 1024) * (1u << 20) / sizeof(int);
	static const int nosc[3] = { 1, 2, 1 }, filt[3] = { 0, 0, 1 };
	size_t words = 0;
	for(int k = 0; k < 3; ++k)
		words += (size_t)m.n_cls[k] * (size_t)*nfrags * A2D_WIN_SLOTWORDS(nosc[k], filt[k]);
	return words <= budget ? 0 : 3;
}

bool vm_spec_wanted(a2amd_ctx *c)
{
	int n;
	uint8_t ff[A2D_MAXBATCH], fb[A2D_MAXBATCH];
	return !vm_spec_why_not(c, &n, ff, fb);
}


// ... launched behind the batch's leaf kernels (issue_kernels): the quiet kernels have moved the phases of the class
// voices that were idle in this batch, k_vm_win / the window control pass everything else the pass reads.
int vm_speculate(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	// (not wanted: vm_predict has made this batch's prediction - k_vm_pool's count - and it stands.  The first cut of
	// this function cleared pred_valid before it looked: no batch was fused any more when the pass was off or the
	// fragments were not whole, and the audio - through records - was still right; the statistics line gave it away)
	if(m.spec_launched_now)		// (once per batch: issue_windows may have called already, issue_kernels asks again)
		return 0;
	m.spec_launched_now = true;
	int nfrags = 0;
	if(const int why = vm_spec_why_not(c, &nfrags, m.spec_ff, m.spec_fb)) {
		const int ncls0 = m.n_cls[0] + m.n_cls[1] + m.n_cls[2];
		if(!m.list.empty() && ncls0)
			++m.spec_skip[why - 1];
		static const int trace = getenv("A2AMD_HOSTTIMING") ? atoi(getenv("A2AMD_HOSTTIMING")) : 0;
		if(trace >= 2 && why == 2 && !m.list.empty() && ncls0)
			fprintf(stderr, "a2amd device VM: no speculative pass behind a batch of %d fragments: its span is not whole engine "
					"fragments, or two wake-ups of the root are expected inside one\n", c->nfrags);
		return 0;
	}
	m.pred_valid = m.spec_valid = false;
	use_device(c);
	static const int nosc[3] = { 1, 2, 1 }, filt[3] = { 0, 0, 1 };
	const int ncls = m.n_cls[0] + m.n_cls[1] + m.n_cls[2];
	if(!m.pred_stream) {
		HIPCHK(c, hipStreamCreateWithFlags(&m.pred_stream, hipStreamNonBlocking));
		HIPCHK(c, hipEventCreateWithFlags(&m.pred_after, hipEventDisableTiming));
		HIPCHK(c, hipEventCreateWithFlags(&m.pred_ev, hipEventDisableTiming));
		HIPCHK(c, hipMalloc((void **)&m.d_pred, 2 * sizeof(unsigned)));
		HIPCHK(c, hipHostMalloc((void **)&m.h_pred, 2 * sizeof(unsigned), hipHostMallocDefault));
	}
	if(!m.d_swtop) {
		// (one block - one memset in front of a pass, one copy behind it: every call on the pass's stream is some 8 us
		// of the engine thread between the commit and the pass, and the pass is what the next buffer waits for)
		HIPCHK(c, hipMalloc((void **)&m.d_swtop, 8 * sizeof(unsigned)));
		m.d_stotal = m.d_swtop + 2;		// ([0..1] as d_total, [2 + class] idle voices)
		HIPCHK(c, hipHostMalloc((void **)&m.h_spec, 8 * sizeof(unsigned), hipHostMallocDefault));
	}
	size_t slotwords = 0;
	for(int k = 0; k < 3; ++k)
		slotwords += (size_t)m.n_cls[k] * (size_t)nfrags * A2D_WIN_SLOTWORDS(nosc[k], filt[k]);
	// pool room: what the last fused / speculative batch took, doubled, and never less than a window per voice and
	// eight fragments; a pass that finds it too small says so (h_spec[1]), is not taken, and the next gets twice that
	size_t want_cap = std::max<size_t>(2 * std::max(m.spec_demand, m.pool_used) + 4096, (size_t)ncls * (size_t)((nfrags + 7) / 8) + 4096);
	if(m.spec_overflows)
		want_cap = std::max(want_cap, 2 * m.spec_cap);
	want_cap = std::min<size_t>(want_cap, 0xfffffff0u);
	m.spec_overflows = 0;
	const int set = m.spec_set ^ 1;		// (the other one may be under the render pass of the batch being issued)
	if(slotwords > m.d_swin[set].cap || want_cap > m.d_swext[set].cap || (size_t)ncls * nfrags > m.d_swidx[set].cap ||
			(size_t)ncls * A2D_VMW_ROW > m.d_swscr.cap || m.d_vmv.cap > m.d_vmv_sh.cap || c->d_ustate.cap > m.d_ustate_sh.cap ||
			c->d_vactive.cap > m.d_vactive_sh.cap || c->d_runs.cap > m.d_runs_sh.cap) {
		// (the last pass may still be reading / writing them)
		HIPCHK(c, hipStreamSynchronize(m.pred_stream));
		if(int r = grow(c, m.d_swin[set], slotwords, 1, false)) return r;
		if(int r = grow(c, m.d_swext[set], want_cap, A2D_WIN_WORDS, false)) return r;
		if(int r = grow(c, m.d_swidx[set], (size_t)ncls * nfrags, 1, false)) return r;
		if(int r = grow(c, m.d_swscr, (size_t)ncls * A2D_VMW_ROW, A2D_WIN_WORDS, false)) return r;
		if(int r = grow(c, m.d_vmv_sh, m.d_vmv.cap, 1, false)) return r;
		if(int r = grow(c, m.d_ustate_sh, c->d_ustate.cap, A2D_USTATE, false)) return r;	// (cap in units, like d_ustate's)
		if(int r = grow(c, m.d_vactive_sh, c->d_vactive.cap, 1, false)) return r;
		if(int r = grow(c, m.d_runs_sh, c->d_runs.cap, 1, false)) return r;
	}
	m.spec_cap = std::min<size_t>(m.d_swext[set].cap, 0xfffffff0u);
	A2DVmParams vp;
	fill_params(c, vp);
	uint32_t frames = 0;
	for(int f = 0; f < c->nfrags; ++f)
		frames += c->fragframes[f];
	vp.now = m.batch_now + (frames << 8);		// (the same span again, beginning where this batch ends ...)
	vp.nfrags = nfrags;				// (... in the fragments vm_spec_plan expects)
	for(int f = 0; f < nfrags; ++f) {
		vp.fragframes[f] = m.spec_ff[f];
		vp.fragbase[f] = m.spec_fb[f];
	}
	// (the counters first: they are this stream's own - the last pass's copies to the host are in front of them - and
	// the pass is then the very next thing behind the wait)
	HIPCHK(c, hipMemsetAsync(m.d_swtop, 0, 8 * sizeof(unsigned), m.pred_stream));
	HIPCHK(c, hipEventRecord(m.pred_after, c->stream));
	HIPCHK(c, hipStreamWaitEvent(m.pred_stream, m.pred_after, 0));
	// The pass goes onto the chip BEFORE the render pass of the batch being issued (issue_windows waits for spec_go):
	// it is one long wavefront per 64 voices that wants a corner of every CU for half a millisecond, the render pass
	// fills whatever is free.  The other way round - the two become ready at the same moment, behind the leaf
	// kernels, and the render pass's queue won by microseconds - k_win_render_f's workgroups held every CU until
	// they were done and the pass ran BEHIND them - which, for the filter class, is as good as it gets: two workgroups
	// of k_win_render_f (8 wavefronts of 128 registers) ARE a CU's register file, with the pass's wavefront there only
	// one fits, and "pass and half the render pass, then the other half" measured 1 475 us per buffer against 1 360 for
	// "render pass, then pass".  Without filter voices the pass shares the CUs with k_win_render, and whether it got
	// onto them first was a race that went one way in one process and the other way in the next: 575 or 745 us per
	// buffer of OscPanScripted, bimodal over runs (profiles/r06_timeline_before.txt, r06_go_ab.txt).
	static const bool go = !(getenv("A2AMD_VMSPEC_GO") && !atoi(getenv("A2AMD_VMSPEC_GO")));
	if(go && !m.n_cls[2]) {
		if(!m.spec_go)
			HIPCHK(c, hipEventCreateWithFlags(&m.spec_go, hipEventDisableTiming));
		HIPCHK(c, hipEventRecord(m.spec_go, m.pred_stream));
		m.spec_go_pending = true;
	}
	const int *l = m.d_list.d + m.list.size() + m.cls_lists.size();
	size_t at = 0, atw = 0, atv = 0;
	for(int k = 0; k < 3; l += m.n_cls[k], ++k) {
		m.spec_cls[k] = m.n_cls[k];
		if(!m.n_cls[k])
			continue;
		vp.list = l;
		vp.n = m.n_cls[k];
		const A2DVmwOut out = { m.d_vmv_sh.d, m.d_ustate_sh.d, m.d_vactive_sh.d, m.d_runs_sh.d, m.d_stotal, m.d_stotal + 2 + k };
		if(a2d_launch_vm_win(vp, c->hparams, nosc[k], filt[k], 0, nfrags, vp.now, vp.now + (frames << 8), m.d_swin[set].d + atw,
				m.d_swext[set].d, m.d_swscr.d + atv * A2D_VMW_ROW * A2D_WIN_WORDS, m.d_swidx[set].d + at, m.d_swtop,
				(unsigned)m.spec_cap, m.pred_stream, &out))
			return c->fail(A2AMD_EHIP, "speculative VM pass launch failed: %s", hipGetErrorString(hipGetLastError()));
		// (A2AMD_WIN_SYNC=1, debugging: wait for the pass and say so - the last line names the kernel that faulted)
		static const bool dbgsync = getenv("A2AMD_WIN_SYNC") != nullptr;
		if(dbgsync) {
			const hipError_t e = hipStreamSynchronize(m.pred_stream);
			fprintf(stderr, "a2amd device VM: speculative pass <%d,%d> of %d voices over %d fragments into set %d (slots at %zu words, "
					"index at %zu, rows at %zu; pool room %zu): %s\n", nosc[k], filt[k], m.n_cls[k], nfrags, set, atw, at, atv,
					m.spec_cap, hipGetErrorString(e));
		}
		at += (size_t)m.n_cls[k] * (size_t)nfrags;
		atw += (size_t)m.n_cls[k] * (size_t)nfrags * A2D_WIN_SLOTWORDS(nosc[k], filt[k]);
		atv += (size_t)m.n_cls[k];
		++c->stats.launches;
	}
	HIPCHK(c, hipMemcpyAsync(m.h_spec, m.d_swtop, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, m.pred_stream));
	HIPCHK(c, hipEventRecord(m.pred_ev, m.pred_stream));
	m.pred_valid = m.spec_valid = true;
	m.pred_serial = m.list_serial;
	m.pred_now = vp.now;
	m.pred_span = frames;
	m.spec_nfrags = nfrags;
	m.spec_set = set;
	++m.spec_launched;
	return 0;
}

// faults of a fused batch's voices (k_vm_win counts them in d_total[1]): copied back behind the batch, looked at
// before the next one (vm_issue)
int vm_fused_done(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	if(!m.total_ev)
		HIPCHK(c, hipEventCreateWithFlags(&m.total_ev, hipEventDisableTiming));
	HIPCHK(c, hipMemcpyAsync(m.h_total, m.d_total, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipEventRecord(m.total_ev, c->stream));
	m.total_pending = true;
	return 0;
}

// fused (the window kernels render this batch's record voices, issue_kernels decides): the voices of the three
// window classes are not run here - k_vm_win, launched by issue_windows() with vm_params(), runs their VMs and
// writes their window entries itself; count / emit then cover only the voices of other chains.
int vm_issue(a2amd_ctx *c, bool fused)
{
	VmHost &m = c->vm;
	m.fused = false;
	m.spec_use = false;
	m.spec_launched_now = false;
	m.spec_go_pending = false;
	// (the faults of the last fused batch are looked at whether or not a voice is left: all of them may have been
	// recalled since)
	if(m.total_pending) {
		use_device(c);
		HIPCHK(c, hipEventSynchronize(m.total_ev));	// (behind the batch before this one: long done)
		m.total_pending = false;
		if(m.h_total[1])
			return c->fail(A2AMD_ESTATE, "device VM: %u voice(s) faulted in an earlier batch (the analysis let a program through "
					"that it should not have)", m.h_total[1]);
	}
	if(m.list.empty()) {
		m.last_total = 0;
		m.pred_valid = m.spec_valid = false;
		return 0;
	}
	if(c->capturing)
		return c->fail(A2AMD_ESTATE, "device VM voices in a captured batch");
	use_device(c);
	if(!m.d_total) {
		HIPCHK(c, hipMalloc((void **)&m.d_total, 2 * sizeof(uint32_t)));
		HIPCHK(c, hipHostMalloc((void **)&m.h_total, 2 * sizeof(uint32_t), hipHostMallocDefault));
	}
	A2DVmParams vp;
	fill_params(c, vp);
	m.batch_now = vp.now;
	++m.vm_batches;
	// fused only if this batch is the one k_vm_pool was run for behind the last one: its count is the pool's bound
	m.fused = false;
	if(fused && m.n_cls[0] + m.n_cls[1] + m.n_cls[2] > 0) {
		// ... the same voices, beginning at the time predicted, covering the stretch predicted, in the fragments predicted
		// (64 frames each from the first) or pieces of them
		int why = !m.pred_valid ? 1 : m.pred_serial != m.list_serial ? 2 : m.pred_now != vp.now ? 3 : 0;
		if(!why) {
			uint32_t at = 0;
			for(int f = 0; f < c->nfrags && !why; ++f) {
				if(at / A2D_FRAG != (at + c->fragframes[f] - 1) / A2D_FRAG)
					why = 5;	// (a fragment across a predicted boundary)
				at += c->fragframes[f];
			}
			if(!why && at != m.pred_span)
				why = 4;
		}
		++m.pred_why[why];
		static const int trace = getenv("A2AMD_HOSTTIMING") ? atoi(getenv("A2AMD_HOSTTIMING")) : 0;
		if(why && m.spec_valid)
			++m.spec_miss[0];
		if(trace >= 2 && why >= 4)
			fprintf(stderr, "a2amd device VM: batch of %d fragments not fused: %s (predicted: %u frames)\n", c->nfrags,
					why == 4 ? "another length" : "a fragment across a 64-frame boundary", m.pred_span);
		if(!why) {
			HIPCHK(c, hipEventSynchronize(m.pred_ev));	// (launched beside the last batch's render pass: done)
			m.pred_entries = m.spec_valid ? m.h_spec[0] : m.h_pred[0];
			m.fused = true;
			if(m.spec_valid) {
				// the speculative pass is taken if this batch is EXACTLY the one it was run for - the same class
				// lists (the serial says so), the same number of whole fragments - and it neither ran out of pool
				// room nor met a fault (the fused pass that runs instead then reports it)
				bool exact = c->nfrags == m.spec_nfrags;
				for(int f = 0; f < c->nfrags && exact; ++f)
					exact = c->fragframes[f] == m.spec_ff[f] && c->fragbase[f] == m.spec_fb[f];
				for(int k = 0; k < 3; ++k)
					exact = exact && m.spec_cls[k] == m.n_cls[k];
				m.spec_demand = m.h_spec[0];
				if(m.h_spec[1])
					++m.spec_overflows;
				m.spec_use = exact && !m.h_spec[1] && !m.h_spec[3];
				if(m.spec_use) {
					++m.spec_taken;
					for(int k = 0; k < 3; ++k)
						m.spec_idle[k] = m.h_spec[4 + k];
				}
				else
					++m.spec_miss[!exact ? 1 : m.h_spec[1] ? 2 : 3];
			}
		}
	}
	m.pred_valid = m.spec_valid = false;
	if(m.fused) {
		vp.list = m.d_list.d + m.list.size() + 2 * m.cls_lists.size();
		vp.n = m.n_other;
	}
	// (the count of records and faults: k_vm_count / k_vm_emit and a k_vm_win of this batch's own add to it - a batch
	// whose class voices took a speculative pass and that has no others runs none of them)
	if(!(m.spec_use && vp.n == 0))
		HIPCHK(c, hipMemsetAsync(m.d_total, 0, 2 * sizeof(uint32_t), c->stream));
	uint64_t batch_frames = 0;
	for(int f = 0; f < c->nfrags; ++f)
		batch_frames += c->fragframes[f];
	if(vp.n == 0) {		// (every voice is k_vm_win's)
		m.last_total = 0;
		m.stats.vm_voice_batches += m.list.size();
		m.replayed += batch_frames;
		return 0;
	}
	if(a2d_launch_vm(vp, 0, c->stream))
		return c->fail(A2AMD_EHIP, "VM count launch failed: %s", hipGetErrorString(hipGetLastError()));
	HIPCHK(c, hipMemcpyAsync(m.h_total, m.d_total, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	const uint32_t total = m.h_total[0];
	m.last_total = total;
	if(m.h_total[1])
		return c->fail(A2AMD_ESTATE, "device VM: %u voice(s) faulted (the analysis let a program through that it should "
				"not have)", m.h_total[1]);
	if(total > m.rec_cap) {
		// the blob grows behind its host-made part (which stays as it is)
		const size_t ncap = std::max((size_t)total * 2, (size_t)65536);
		const size_t nbytes = m.rec_off + ncap * sizeof(A2DRec);
		char *nd = nullptr;
		HIPCHK(c, hipMalloc((void **)&nd, nbytes));
		HIPCHK(c, hipMemcpy(nd, c->d_blob.d, m.rec_off, hipMemcpyDeviceToDevice));
		const ptrdiff_t o_recs = (const char *)c->hparams.recs - c->d_blob.d;
		const ptrdiff_t o_dyn = (const char *)c->d_dyn - c->d_blob.d;
		HIPCHK(c, hipFree(c->d_blob.d));
		c->d_blob.d = nd;
		c->d_blob.cap = nbytes;
		c->hparams.recs = (const A2DRec *)(nd + o_recs);
		c->d_params = (A2DParams *)nd;
		c->d_dyn = (const int *)(nd + o_dyn);
		HIPCHK(c, hipMemcpy(nd, &c->hparams, sizeof(A2DParams), hipMemcpyHostToDevice));
		m.rec_cap = (uint32_t)ncap;
		drop_graphs(c);
		vp.recs = (A2DRec *)c->hparams.recs;
	}
	vp.rec_base = (uint32_t)((m.rec_off - (size_t)((const char *)c->hparams.recs - c->d_blob.d)) / sizeof(A2DRec));
	vp.rec_cap = m.rec_cap;
	if(a2d_launch_vm(vp, 1, c->stream))
		return c->fail(A2AMD_EHIP, "VM emit launch failed: %s", hipGetErrorString(hipGetLastError()));
	static const bool dump = getenv("A2AMD_VM_DUMP") != nullptr;	// (... and device-made)
	if(dump) {
		std::vector<A2DRun> runs(m.list.size());
		std::vector<A2DRec> recs(total);
		HIPCHK(c, hipStreamSynchronize(c->stream));
		HIPCHK(c, hipMemcpy(runs.data(), m.d_vmrun.d, runs.size() * sizeof(A2DRun), hipMemcpyDeviceToHost));
		if(total)
			HIPCHK(c, hipMemcpy(recs.data(), vp.recs + vp.rec_base, total * sizeof(A2DRec), hipMemcpyDeviceToHost));
		for(size_t k = 0; k < runs.size(); ++k)
			for(int q = 0; q < runs[k].count && (size_t)(runs[k].first + q) < recs.size(); ++q) {
				const A2DRec &r = recs[runs[k].first + q];
				fprintf(stderr, "REC %lld v%d f%u op%u u%u r%u val %d dur %u start %u (vm)\n", c->serial_base, m.vms[m.list[k]].voice,
						A2D_RFRAG(r.head), A2D_ROP(r.head), A2D_RUNIT(r.head), A2D_RREG(r.head), r.value, r.dur, r.start);
			}
	}
	c->stats.launches += 2;
	c->stats.records += total;
	m.stats.vm_voice_batches += m.list.size();
	m.stats.vm_records += total;
	// (a kept batch that is run again renders the next stretch of time)
	uint64_t frames = 0;
	for(int f = 0; f < c->nfrags; ++f)
		frames += c->fragframes[f];
	m.replayed += frames;
	return 0;
}

void vm_end_batch(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	m.replayed = 0;
	m.snap_fetches = 0;
	if(m.n_fresh) {
		for(HVm &h : m.vms)
			h.fresh = false;
		m.n_fresh = 0;
		m.list_dirty = true;
	}
	if(!m.snap_have.empty())
		std::fill(m.snap_have.begin(), m.snap_have.end(), 0);
}

void vm_close(a2amd_ctx *c)
{
	VmHost &m = c->vm;
	if(c->hosttiming && m.vm_batches)
		fprintf(stderr, "a2amd device VM: %llu batches with device VM voices, %llu of them through k_vm_win (the others: %llu not "
				"predicted, %llu other voices, %llu another time, %llu / %llu / %llu other fragments)\n",
				(unsigned long long)m.vm_batches, (unsigned long long)m.fused_batches, (unsigned long long)m.pred_why[1],
				(unsigned long long)m.pred_why[2], (unsigned long long)m.pred_why[3], (unsigned long long)m.pred_why[4],
				(unsigned long long)m.pred_why[5], (unsigned long long)m.pred_why[6]);
	if(c->hosttiming && m.spec_launched)
		fprintf(stderr, "a2amd device VM: %llu speculative passes (k_vm_win for the batch expected next, into shadows), %llu of them taken "
				"(k_vm_commit + render pass in k_vm_win's place); pool room %zu entries, last demand %zu; not taken: %llu not the "
				"batch predicted, %llu other fragments / classes, %llu pool overflow, %llu faults; batches without a pass: %llu off / "
				"short, %llu no plan (span not whole fragments), %llu over the slot budget, %llu lists just changed; %llu waits for a "
				"pass in flight before a rewrite; %llu quiet-kernel launches not made (every voice of the class the VM's)\n",
				(unsigned long long)m.spec_launched, (unsigned long long)m.spec_taken, m.spec_cap, m.spec_demand,
				(unsigned long long)m.spec_miss[0], (unsigned long long)m.spec_miss[1], (unsigned long long)m.spec_miss[2],
				(unsigned long long)m.spec_miss[3], (unsigned long long)m.spec_skip[0], (unsigned long long)m.spec_skip[1],
				(unsigned long long)m.spec_skip[2], (unsigned long long)m.spec_skip[3], (unsigned long long)m.spec_waits,
				(unsigned long long)m.quiet_skipped);
	if(m.pred_stream)
		hipStreamSynchronize(m.pred_stream);	// (a pass still running reads the arrays freed below)
	for(int k = 0; k < 2; ++k) {
		hipFree(m.d_swin[k].d);
		hipFree(m.d_swext[k].d);
		hipFree(m.d_swidx[k].d);
	}
	hipFree(m.d_swscr.d);
	hipFree(m.d_swtop);
	hipFree(m.d_vmv_sh.d);
	hipFree(m.d_ustate_sh.d);
	hipFree(m.d_vactive_sh.d);
	hipFree(m.d_runs_sh.d);
	if(m.h_spec)
		hipHostFree(m.h_spec);
	hipFree(m.d_vmv.d);
	hipFree(m.d_code.d);
	hipFree(m.d_list.d);
	hipFree(m.d_vmrun.d);
	hipFree(m.d_f1tab);
	hipFree(m.d_envlut);
	hipFree(m.d_total);
	if(m.total_ev)
		hipEventDestroy(m.total_ev);
	if(m.pred_stream) {
		hipStreamSynchronize(m.pred_stream);
		hipStreamDestroy(m.pred_stream);
		hipEventDestroy(m.pred_after);
		hipEventDestroy(m.pred_ev);
		if(m.spec_go)
			hipEventDestroy(m.spec_go);
		hipFree(m.d_pred);
		hipHostFree(m.h_pred);
	}
	if(m.h_total)
		hipHostFree(m.h_total);
	if(m.h_stage)
		hipHostFree(m.h_stage);
}

} // namespace a2h

extern "C" {

int a2amd_vm_analyze(const uint32_t *code, unsigned nwords, unsigned pc, int32_t tick, uint32_t msdur, a2amd_vm_info *info)
{
	a2amd_vm_info tmp;
	return analyze(code, nwords, pc, tick, msdur, info ? info : &tmp);
}

int a2amd_vm_program(a2amd_ctx *c, uint64_t key, const uint32_t *code, unsigned nwords)
{
	VmHost &m = c->vm;
	if(!code || !nwords || nwords > 65535)
		return c->fail(A2AMD_EINVAL, "vm_program: %u words", nwords);
	const uint64_t sum = text_sum(code, nwords);
	for(size_t k = 0; k < m.progs.size(); ++k)
		if(m.progs[k].key == key && m.progs[k].n == nwords && m.progs[k].sum == sum)
			return (int)k;
	HVmProg p;
	p.key = key;
	p.off = (uint32_t)m.code.size();
	p.n = nwords;
	p.sum = sum;
	m.code.insert(m.code.end(), code, code + nwords);
	m.progs.push_back(p);
	m.stats.programs = (uint32_t)m.progs.size();
	return (int)m.progs.size() - 1;
}

int a2amd_vm_envluts(a2amd_ctx *c, const uint16_t *luts)
{
	if(!luts)
		return c->fail(A2AMD_EINVAL, "vm_envluts: null");
	c->vm.envlut.assign(luts, luts + A2D_ENV_LUTS * (A2D_ENV_LUTSIZE + 2));
	c->vm.envlut_up = false;
	return A2AMD_OK;
}

int a2amd_vm_adopt(a2amd_ctx *c, int head, int prog, const a2amd_vm_state *st, const int32_t *wr_unit,
		const uint8_t *wr_reg, uint32_t now, uint32_t msdur, const a2amd_vm_env *envs, int nenv)
{
	VmHost &m = c->vm;
	if(head < 0 || head >= (int)c->units.size() || !c->units[head].live || !st || !wr_unit || !wr_reg)
		return c->fail(A2AMD_EINVAL, "vm_adopt: bad unit %d", head);
	if(prog < 0 || prog >= (int)m.progs.size())
		return c->fail(A2AMD_EINVAL, "vm_adopt: program %d", prog);
	const int vi = c->units[head].voice;
	HVoice &v = c->voices[vi];
	if(v.vm >= 0)
		return c->fail(A2AMD_ESTATE, "vm_adopt: voice %d is adopted already", vi);
	if(!c->frag_open || c->uploaded)
		return c->fail(A2AMD_ESTATE, "vm_adopt outside a fragment");
	if(v.inline_pos >= 0)
		return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: voice %d owns a bus", vi);
	// (round 6 lifted the backend's chains to A2D_MAXCHAIN = 16 units; the VM's register map still names a chain position
	// in a nibble that has 14 and 15 taken - A2D_VM_ENVPOS, A2D_VM_NOWRITE - and run_batch looks kinds up by pos & 7)
	if(v.nunits > A2D_VM_MAXPOS)
		return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: voice %d has a chain of %d units (the device VM takes up to %d)", vi,
				v.nunits, A2D_VM_MAXPOS);
	// (not now - which says nothing about the program: A2AMD_ESTATE)
	const bool marked = c->defmap_used && (size_t)vi < c->defmap.size() && c->defmap[vi];
	if(!v.live || v.dying || !v.resolved || !v.started || (v.walked != c->serial_base + c->cur_frag && !marked))
		return c->fail(A2AMD_ESTATE, "vm_adopt: voice %d has not been processed in the open fragment (live %d dying %d "
				"resolved %d started %d walked %lld, fragment %lld)", vi, (int)v.live, (int)v.dying, (int)v.resolved,
				(int)v.started, v.walked, c->serial_base + c->cur_frag);
	if(st->state != A2AMD_VM_WAITING)
		return c->fail(A2AMD_ESTATE, "vm_adopt: VM state %d (only a voice that waits in a delay is taken)", st->state);
	if(m.t0_valid && m.msdur != msdur && (m.stats.live || !m.pending.empty()))
		return c->fail(A2AMD_EINVAL, "vm_adopt: msdur %u, the context's voices run on %u", msdur, m.msdur);
	const HVmProg &p = m.progs[prog];
	A2DVmVoice d;
	memset(&d, 0, sizeof(d));
	memset(d.cmap, A2D_VM_NOWRITE, sizeof(d.cmap));
	d.waketime = st->waketime;
	d.code = p.off;
	d.ncode = p.n;
	d.pc = st->pc;
	d.state = st->state;
	d.voice = vi;
	memcpy(d.r, st->r, sizeof(d.r));
	bool need_f1 = false;
	for(int k = 0; k < v.nunits; ++k) {
		const HUnit &u = c->units[v.unit[k]];
		d.kind[k] = (uint8_t)u.kind;
		if(u.kind == A2AMD_WTOSC && u.mode == A2D_OSC_NOISE)
			return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: a noise oscillator draws from the engine's RNG");
		if(u.xio_mode || u.kind == A2AMD_XINSERT || u.kind == A2AMD_XSINK || u.kind == A2AMD_XSOURCE || u.kind == A2AMD_INLINE)
			return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: unit kind %d in the chain", u.kind);
		if(u.kind == A2AMD_FILTER12) {
			// every filter12 of the chain brings its cutoff ramper along: one that is still ramping
			// needs a coefficient per window whoever writes it
			if(d.ncut >= A2D_VM_MAXCUT)
				return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: more than %d filter12 units", A2D_VM_MAXCUT);
			d.cutpos[d.ncut] = (uint8_t)k;
			d.cut[d.ncut][0] = u.cutoff.value;
			d.cut[d.ncut][1] = u.cutoff.target;
			d.cut[d.ncut][2] = u.cutoff.delta;
			d.cut[d.ncut][3] = u.cutoff.timer;
			++d.ncut;
			need_f1 = true;
		}
	}
	// (thousands of voices of one program are adopted at the same few pcs: the proof is kept)
	a2amd_vm_info info;
	{
		const int32_t tick = st->r[A2AMD_VM_R_TICK];
		bool hit = false;
		for(const VmHost::Proof &q : m.proofs)
			if(q.prog == prog && q.pc == st->pc && q.tick == tick && q.msdur == msdur) {
				info = q.info;
				hit = true;
				break;
			}
		if(!hit) {
			analyze(m.code.data() + p.off, p.n, st->pc, tick, msdur, &info);
			if(m.proofs.size() >= 256)
				m.proofs.erase(m.proofs.begin());
			VmHost::Proof q = { prog, st->pc, tick, msdur, info };
			m.proofs.push_back(q);
		}
	}
	// What the analysis cannot prove for ALL of the program's future - it reaches an instruction outside the
	// subset (every program that ends does), a loop it cannot bound - the look-ahead below settles for the
	// stretch of it that comes first: the voice is taken with an exit time.
	const bool dynamic = info.reason == A2AMD_VM_OPCODE_OUT || info.reason == A2AMD_VM_NOYIELD || info.reason == A2AMD_VM_DIVISOR;
	if(info.reason && !dynamic)
		return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: %s (opcode %d at %d)", reason_text(info.reason), info.opcode, info.at);
	bool any = false;
	// the voice's env units: where their control outputs go must be a register the device VM writes
	if(nenv < 0 || nenv > A2AMD_VM_MAXENV || (nenv && !envs))
		return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: %d env units", nenv);
	if(nenv && m.envlut.empty())
		return c->fail(A2AMD_ESTATE, "vm_adopt: env units, but no tables (a2amd_vm_envluts)");
	d.nenv = nenv;
	for(int k = 0; k < nenv; ++k) {
		A2DVmEnv &en = d.env[k];
		memcpy(en.ramper, envs[k].ramper, sizeof(en.ramper));
		en.lut = envs[k].lut;
		en.scale = envs[k].scale;
		en.offset = envs[k].offset;
		en.out = envs[k].out;
		en.active = envs[k].active;
		if(envs[k].regbase < 0 || envs[k].regbase + 3 >= A2AMD_VM_REGISTERS || envs[k].before < 0 || envs[k].before > v.nunits ||
				envs[k].lut < 0 || envs[k].lut >= A2D_ENV_LUTS)
			return c->fail(A2AMD_EINVAL, "vm_adopt: env unit %d", k);
		en.regbase = (uint8_t)envs[k].regbase;
		en.k = (uint8_t)envs[k].before;
		en.target = A2D_VM_NOWRITE;
		if(envs[k].out_unit != -1) {
			int pos = -1;
			for(int q = 0; q < v.nunits; ++q)
				if(v.unit[q] == envs[k].out_unit)
					pos = q;
			if(pos < 0 || !write_supported(c->units[v.unit[pos]].kind, envs[k].out_reg))
				return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: %s (env unit %d -> unit %d register %d)",
						reason_text(A2AMD_VM_TARGET), k, envs[k].out_unit, envs[k].out_reg);
			en.target = (uint8_t)((pos << 4) | envs[k].out_reg);
			need_f1 |= write_needs_f1tab(c->units[v.unit[pos]].kind, envs[k].out_reg);
			if(en.active)
				any = true;	// (a running segment is work the engine no longer has to do)
		}
	}
	for(int r = 0; r < A2AMD_VM_REGISTERS; ++r) {
		if(wr_unit[r] == -1)
			continue;
		// (a register the VM can never pass to a2_VoiceControl may be wired to anything)
		if(!dynamic && !(info.controlled & (1ull << r)))
			continue;
		if(wr_unit[r] <= -3 && -3 - wr_unit[r] < nenv) {
			d.cmap[r] = (uint8_t)((A2D_VM_ENVPOS << 4) | (-3 - wr_unit[r]));
			any |= d.env[-3 - wr_unit[r]].target != A2D_VM_NOWRITE;
			continue;
		}
		int pos = -1;
		for(int k = 0; k < v.nunits; ++k)
			if(v.unit[k] == wr_unit[r])
				pos = k;
		if(pos < 0 || !write_supported(c->units[v.unit[pos]].kind, wr_reg[r])) {
			if(dynamic) {	// (the stay ends in front of the first VM run that writes through it)
				d.cmap[r] = A2D_VM_TRAPWRITE;
				continue;
			}
			return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: %s (VM register %d -> unit %d register %d)",
					reason_text(A2AMD_VM_TARGET), r, wr_unit[r], (int)wr_reg[r]);
		}
		d.cmap[r] = (uint8_t)((pos << 4) | wr_reg[r]);
		need_f1 |= write_needs_f1tab(c->units[v.unit[pos]].kind, wr_reg[r]);
		any = true;
	}
	// (the tracker's 32 bit mask lets register r + 32 ride on r's bit: a program that uses both
	// halves is rare and its aliasing is reproduced by the interpreter - but the set of registers
	// that may be written must then cover the twins too, which 'controlled' does not track)
	if(!dynamic && (info.written >> 32) && (uint32_t)info.written & (uint32_t)(info.written >> 32))
		return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: registers r and r + 32 both written (tracker aliasing)");
	if(!any)
		return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: %s", reason_text(A2AMD_VM_IDLE));
	if(need_f1)
		build_f1tab(c);
	bool has_exit = false;
	uint32_t exit_when = 0;
	if(dynamic) {
		int nrec = 0, runs = 0;
		const uint32_t msdur0 = m.msdur;
		m.msdur = msdur;		// (consts_of)
		vm_lookahead(d, m.code.data() + p.off, consts_of(c), now, &exit_when, &nrec, &runs);
		m.msdur = msdur0;
		has_exit = true;
		if(!runs)	// the voice's very next VM run is the engine's: never, from this pc
			return c->fail(A2AMD_EUNSUPPORTED, "vm_adopt: %s (opcode %d at %d) in the voice's next VM run",
					reason_text(info.reason), info.opcode, info.at);
		// a stay that is over within the buffer being walked, or that writes nothing, is not worth the hand-over
		bool env_running = false;
		for(int k = 0; k < nenv; ++k)
			env_running |= d.env[k].active && d.env[k].target != A2D_VM_NOWRITE;
		if((int)(exit_when - now) < (VM_MIN_STAY << 8) || (!nrec && !env_running))
			return c->fail(A2AMD_ESTATE, "vm_adopt: the device VM's stay would be %d frames and %d writes long",
					(int)(exit_when - now) >> 8, nrec);
	}
	// the context's clock: engine time of walk_time 0
	{
		uint64_t before = 0;	// frames of the batch's fragments before the open one
		for(int f = 0; f < c->cur_frag; ++f)
			before += c->fragframes[f];
		const uint32_t t0 = now - (uint32_t)((m.batch_time + before) << 8);
		if(m.t0_valid && t0 != m.t0 && (m.stats.live || !m.pending.empty()))
			return c->fail(A2AMD_ESTATE, "vm_adopt: the engine's clock (%u) has left the context's (%u)", t0, m.t0);
		m.t0 = t0;
		m.t0_valid = true;
		m.msdur = msdur;
	}
	int slot;
	if(!m.free_slots.empty()) {
		slot = m.free_slots.back();
		m.free_slots.pop_back();
	} else {
		slot = (int)m.vms.size();
		m.vms.push_back(HVm());
	}
	HVm &h = m.vms[slot];
	h = HVm();
	h.live = true;
	h.pending = true;
	h.adopt_frag = c->cur_frag;
	h.voice = vi;
	h.prog = prog;
	h.func = st->func;
	d.has_exit = has_exit;
	d.exit_when = exit_when;
	h.st = d;
	h.has_exit = has_exit;
	h.exit_when = exit_when;
	m.pending.push_back(slot);
	v.vm = slot;
	v.plain = 0;
	++m.stats.live;
	++m.stats.adopted;
	return A2AMD_OK;
}

// (include/a2amd_vm.h: where the next batch's fragments will be cut - for vm_spec_plan)
int a2amd_vm_expect_cuts(a2amd_ctx *c, const uint32_t *when, int n)
{
	if(!c || n < 0 || (n && !when))
		return A2AMD_EINVAL;
	VmHost &m = c->vm;
	m.n_cut_hint = std::min(n, 16);
	for(int k = 0; k < m.n_cut_hint; ++k)
		m.cut_hint[k] = when[k];
	return 0;
}

int a2amd_vm_exit_time(a2amd_ctx *c, int head, uint32_t *when)
{
	if(head < 0 || head >= (int)c->units.size() || !c->units[head].live)
		return 0;
	const int slot = c->voices[c->units[head].voice].vm;
	if(slot < 0 || slot >= (int)c->vm.vms.size() || !c->vm.vms[slot].live || !c->vm.vms[slot].has_exit)
		return 0;
	if(when)
		*when = c->vm.vms[slot].exit_when;
	return 1;
}

int a2amd_vm_adopted(a2amd_ctx *c, int head)
{
	if(head < 0 || head >= (int)c->units.size() || !c->units[head].live)
		return 0;
	return c->voices[c->units[head].voice].vm >= 0;
}

int a2amd_vm_recall(a2amd_ctx *c, const int32_t *heads, unsigned n, a2amd_vm_state *out, a2amd_vm_env *envs_out)
{
	if(!heads || !out)
		return c->fail(A2AMD_EINVAL, "vm_recall: null argument");
	if(n > 8)
		c->vm.snap_fetches = 8;		// (several at once: one copy of all states)
	for(unsigned k = 0; k < n; ++k) {
		if(heads[k] < 0 || heads[k] >= (int)c->units.size() || !c->units[heads[k]].live)
			return c->fail(A2AMD_EINVAL, "vm_recall: bad unit %d", heads[k]);
		if(int r = vm_take_back(c, c->units[heads[k]].voice, false, &out[k], envs_out ? envs_out + (size_t)k * A2AMD_VM_MAXENV : nullptr))
			return r;
	}
	return A2AMD_OK;
}

int a2amd_vm_trace_host(const uint32_t *code, unsigned nwords, a2amd_vm_state *st, const int32_t *wr_unit,
		const uint8_t *wr_reg, const int32_t *kinds, int nkinds, uint32_t now, uint32_t msdur, int32_t samplerate,
		int32_t basepitch, const uint8_t *fragframes, unsigned nfrags, uint32_t *recs, unsigned cap)
{
	return a2amd_vm_trace_host_env(code, nwords, st, wr_unit, wr_reg, kinds, nkinds, now, msdur, samplerate, basepitch,
			fragframes, nullptr, nfrags, nullptr, 0, nullptr, recs, cap);
}

static int trace_impl(const uint32_t *code, unsigned nwords, a2amd_vm_state *st, const int32_t *wr_unit,
		const uint8_t *wr_reg, const int32_t *kinds, int nkinds, uint32_t now, uint32_t msdur, int32_t samplerate,
		int32_t basepitch, const uint8_t *fragframes, const uint8_t *fragbase, unsigned nfrags, a2amd_vm_env *envs, int nenv,
		const uint16_t *envluts, uint32_t *recs, unsigned cap, int32_t *has_exit, uint32_t *exit_when, int32_t *stay_records);

int a2amd_vm_trace_host_env(const uint32_t *code, unsigned nwords, a2amd_vm_state *st, const int32_t *wr_unit,
		const uint8_t *wr_reg, const int32_t *kinds, int nkinds, uint32_t now, uint32_t msdur, int32_t samplerate,
		int32_t basepitch, const uint8_t *fragframes, const uint8_t *fragbase, unsigned nfrags, a2amd_vm_env *envs, int nenv,
		const uint16_t *envluts, uint32_t *recs, unsigned cap)
{
	return trace_impl(code, nwords, st, wr_unit, wr_reg, kinds, nkinds, now, msdur, samplerate, basepitch, fragframes, fragbase,
			nfrags, envs, nenv, envluts, recs, cap, nullptr, nullptr, nullptr);
}

int a2amd_vm_trace_host_exit(const uint32_t *code, unsigned nwords, a2amd_vm_state *st, const int32_t *wr_unit,
		const uint8_t *wr_reg, const int32_t *kinds, int nkinds, uint32_t now, uint32_t msdur, int32_t samplerate,
		int32_t basepitch, const uint8_t *fragframes, const uint8_t *fragbase, unsigned nfrags, a2amd_vm_env *envs, int nenv,
		const uint16_t *envluts, uint32_t *recs, unsigned cap, int32_t *has_exit, uint32_t *exit_when, int32_t *stay_records)
{
	if(!has_exit || !exit_when)
		return A2AMD_EINVAL;
	return trace_impl(code, nwords, st, wr_unit, wr_reg, kinds, nkinds, now, msdur, samplerate, basepitch, fragframes, fragbase,
			nfrags, envs, nenv, envluts, recs, cap, has_exit, exit_when, stay_records);
}

static int trace_impl(const uint32_t *code, unsigned nwords, a2amd_vm_state *st, const int32_t *wr_unit,
		const uint8_t *wr_reg, const int32_t *kinds, int nkinds, uint32_t now, uint32_t msdur, int32_t samplerate,
		int32_t basepitch, const uint8_t *fragframes, const uint8_t *fragbase, unsigned nfrags, a2amd_vm_env *envs, int nenv,
		const uint16_t *envluts, uint32_t *recs, unsigned cap, int32_t *has_exit, uint32_t *exit_when, int32_t *stay_records)
{
	if(!code || !st || !wr_unit || !wr_reg || !kinds || !fragframes || !recs || nkinds < 1 || nkinds > A2D_VM_MAXPOS)
		return A2AMD_EINVAL;
	if(nenv < 0 || nenv > A2AMD_VM_MAXENV || (nenv && (!envs || !envluts)))
		return A2AMD_EINVAL;
	static thread_local uint32_t ptab[128];
	static thread_local std::vector<int32_t> f1;
	static thread_local int f1_sr = 0;
	build_pitch_table(ptab);
	A2DVmVoice d;
	memset(&d, 0, sizeof(d));
	memset(d.cmap, A2D_VM_NOWRITE, sizeof(d.cmap));
	d.waketime = st->waketime;
	d.ncode = nwords;
	d.pc = st->pc;
	d.state = st->state;
	memcpy(d.r, st->r, sizeof(d.r));
	bool need_f1 = false;
	for(int k = 0; k < nkinds; ++k) {
		d.kind[k] = (uint8_t)kinds[k];
		if(kinds[k] == A2AMD_FILTER12 && d.ncut < A2D_VM_MAXCUT) {
			d.cutpos[d.ncut] = (uint8_t)k;
			// (f12_Initialize, filter12.c:203: cutoff 0 + transpose, no ramp)
			d.cut[d.ncut][0] = d.cut[d.ncut][1] = (int)((unsigned)st->r[A2AMD_VM_R_TRANSPOSE] << 8);
			++d.ncut;
			need_f1 = true;
		}
	}
	for(int r = 0; r < A2AMD_VM_REGISTERS; ++r)
		if(wr_unit[r] >= 0 && wr_unit[r] < nkinds) {
			if(has_exit && !write_supported(kinds[wr_unit[r]], wr_reg[r])) {
				d.cmap[r] = A2D_VM_TRAPWRITE;	// (as a2amd_vm_adopt under a look-ahead)
				continue;
			}
			d.cmap[r] = (uint8_t)((wr_unit[r] << 4) | wr_reg[r]);
			need_f1 |= write_needs_f1tab(kinds[wr_unit[r]], wr_reg[r]);
		} else if(wr_unit[r] <= -3 && -3 - wr_unit[r] < nenv)
			d.cmap[r] = (uint8_t)((A2D_VM_ENVPOS << 4) | (-3 - wr_unit[r]));
		else if(has_exit && wr_unit[r] == -2)
			d.cmap[r] = A2D_VM_TRAPWRITE;
	d.nenv = nenv;
	for(int k = 0; k < nenv; ++k) {
		// (as a2amd_vm_adopt, with chain positions where that has backend unit ids)
		A2DVmEnv &en = d.env[k];
		memcpy(en.ramper, envs[k].ramper, sizeof(en.ramper));
		en.lut = envs[k].lut;
		en.scale = envs[k].scale;
		en.offset = envs[k].offset;
		en.out = envs[k].out;
		en.active = envs[k].active;
		if(envs[k].regbase < 0 || envs[k].regbase + 3 >= A2AMD_VM_REGISTERS || envs[k].before < 0 || envs[k].before > nkinds ||
				envs[k].lut < 0 || envs[k].lut >= A2D_ENV_LUTS)
			return A2AMD_EINVAL;
		en.regbase = (uint8_t)envs[k].regbase;
		en.k = (uint8_t)envs[k].before;
		en.target = A2D_VM_NOWRITE;
		if(envs[k].out_unit >= 0) {
			if(envs[k].out_unit >= nkinds || !write_supported(kinds[envs[k].out_unit], envs[k].out_reg))
				return A2AMD_EUNSUPPORTED;
			en.target = (uint8_t)((envs[k].out_unit << 4) | envs[k].out_reg);
			need_f1 |= write_needs_f1tab(kinds[envs[k].out_unit], envs[k].out_reg);
		}
	}
	Consts K;
	K.msdur = msdur;
	K.samplerate = samplerate;
	K.basepitch = basepitch;
	K.ptab = ptab;
	K.f1tab = nullptr;
	K.envlut = envluts;
	if(need_f1) {
		if(f1.empty() || f1_sr != samplerate) {
			f1.resize((size_t)32 * 65536);
			for(unsigned sh = 0; sh < 32; ++sh)
				for(unsigned n = 0; n < 65536; ++n) {
					const int pitch = (int)(((unsigned)(7 - (int)sh) << 16) | n);
					float f = a2h::p2i(ptab, pitch) * (261.626f / 16777216.0f);
					f1[(size_t)sh * 65536 + n] = f > (samplerate >> 2) ? 362 << 16 :
							(int)(512.0f * 65536.0f * sin(M_PI * f / samplerate));
				}
			f1_sr = samplerate;
		}
		K.f1tab = f1.data();
	}
	std::vector<A2DRec> buf((size_t)cap + 8);
	struct BoundedE : a2vm::PlainE {
		A2DRec *out;
		int n, cap;
		void rec(int frag, int op, int unit, int reg, int value, unsigned dur, unsigned start)
		{
			if(n < cap) {
				out[n].head = A2D_HEAD(frag, op, unit, reg);
				out[n].value = value;
				out[n].dur = dur;
				out[n].start = start;
			}
			++n;
		}
		int count() const { return n; }
	} e = { {}, buf.data(), 0, (int)cap };
	if(has_exit) {
		// the stretch the device VM would be given (a2amd_vm_adopt, vm_lookahead): the interpreter stops in front of
		// the first VM run that is the engine's
		int nrec = 0, runs = 0;
		uint32_t when = 0;
		vm_lookahead(d, code, K, now, &when, &nrec, &runs);
		d.has_exit = 1;
		d.exit_when = when;
		*has_exit = 1;
		*exit_when = when;
		if(stay_records)
			*stay_records = nrec;
	}
	run_batch(d, code, K, e, now, 0, (int)nfrags, [fragframes, fragbase](int f) {
		return (unsigned)fragframes[f] | (fragbase ? (unsigned)fragbase[f] << 8 : 0u); });
	if(d.fault)
		return A2AMD_ESTATE;
	st->waketime = d.waketime;
	st->state = d.state;
	st->pc = d.pc;
	memcpy(st->r, d.r, sizeof(st->r));
	for(int k = 0; k < nenv; ++k) {		// (where the segments stand at the end)
		memcpy(envs[k].ramper, d.env[k].ramper, sizeof(envs[k].ramper));
		envs[k].lut = d.env[k].lut;
		envs[k].scale = d.env[k].scale;
		envs[k].offset = d.env[k].offset;
		envs[k].out = d.env[k].out;
		envs[k].active = d.env[k].active;
	}
	const int n = std::min(e.n, (int)cap);
	memcpy(recs, buf.data(), (size_t)n * sizeof(A2DRec));
	return e.n;
}

int a2amd_vm_get_stats(a2amd_ctx *c, a2amd_vm_stats *out)
{
	if(!out)
		return A2AMD_EINVAL;
	*out = c->vm.stats;
	return A2AMD_OK;
}

} // extern "C"
