// a2amd_vmcore.h - the interpreter of the scripted voice's VM, ONE source for the
// HIP kernel (a2amd_vm.hip: one lane per voice) and for the host (a2amd_vm.cpp: catching a voice
// up when the engine wants it back, the parity tests' a2amd_vm_trace_host).
//
// What it restates, for the instruction subset of include/a2amd_vm.h:
//   a2_VoiceProcessVM      src/core.c:1166-1744   run()
//   a2_VoiceControl        src/core.c:143-149     control()  (+ what the drop-in's write callbacks add:
//                                                 a2amd_unit_write, a2amd_host.cpp)
//   A2_regtracker          src/core.c:1064-1116   Tracker
//   a2_ms2t / a2_ticks2t   src/core.c:1120-1131
//   a2_VoiceProcessVMEv    src/core.c:1784-1839   (the "VM only" loop: an adopted voice has no events)
//   a2_VoiceProcess        src/core.c:1847-1880   run_batch(): VM runs alternate with unit windows
// and what it emits are the command records the host's recorder makes of the same calls
// (a2amd_unit_write / a2amd_unit_process, a2amd_host.cpp): R_WRITE, R_F1SET, R_F1RAMP, R_SEG.
// Arithmetic is the engine's on x86-64 / gcc: wrap-around ints, arithmetic >>, shift counts mod 32.
#pragma once
#include <stdint.h>
#include "a2amd_device.h"
#include "../../include/a2amd_vm.h"

#if defined(__HIPCC__)
#define VMFN __host__ __device__ inline
#else
#define VMFN inline
#endif

namespace a2vm {

struct Consts {
	uint32_t msdur;
	int32_t samplerate, basepitch;
	const uint32_t *ptab;		// 64 x {base, coeff}, pitch.c:70-96
	const int32_t *f1tab;		// [32][65536] or null
	const uint16_t *envlut;		// [A2D_ENV_LUTS][A2D_ENV_LUTSIZE + 2] or null
};

enum { TRAP_NONE = 0, TRAP_OVERLOAD, TRAP_OPCODE, TRAP_DIVISOR, TRAP_PC,
	TRAP_TARGET,	// a write through a VM register wired to something the device VM does not write (A2D_VM_TRAPWRITE)
	TRAP_RECORDS };	// more records between two drains than a lane of k_vm_win holds (a2amd_vmwin.hip)
// (not a trap: a run that gives way to its caller - an emitter with a bounded queue, E::fused - and is entered again)
enum { RUN_YIELD = -1 };
// (env keeps its own copy of the state's msdur, env.c:236: the same expression)
VMFN uint32_t en_msdur(const Consts &K) { return K.msdur; }

VMFN int vadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
VMFN int vsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
VMFN int vmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// a2_ms2t, core.c:1128-1131
VMFN unsigned ms2t(uint32_t msdur, int d)
{
	return (unsigned)(((int64_t)d * (int64_t)msdur + 0x7fffff) >> 24);
}

// a2_ticks2t, core.c:1120-1124 (the products are unsigned 64 bit there)
VMFN unsigned ticks2t(uint32_t msdur, int tick, int d)
{
	const uint64_t t = ((uint64_t)(int64_t)d * (uint64_t)(int64_t)tick + 127) >> 8;
	return (unsigned)((t * (uint64_t)msdur + 0x7fffffffu) >> 32);
}

// a2_P2I, pitch.c:57-67
VMFN unsigned p2i(const uint32_t *tab, int pitch)
{
	const int n = pitch & 0xffff, oct = pitch >> 16;
	unsigned dph = tab[2 * (n >> 10) + 1] * (unsigned)(n & 0x3ff);
	dph >>= 2;
	dph += tab[2 * (n >> 10)];
	return dph >> ((unsigned)(7 - oct) & 31u);
}

// f12_pitch2coeff (filter12.c:65-72) / dcb_pitch2coeff (dcblock.c:57-64) of a 16:16 pitch, from the
// table the host made with the reference's own float / libm expression for every value a2_P2I can
// return: indexed by the shift count and the 16 fraction bits a2_P2I derives its result from
VMFN int f1_of_pitch(const Consts &K, int pitch)
{
	const unsigned n = (unsigned)pitch & 0xffffu, sh = (unsigned)(7 - (pitch >> 16)) & 31u;
	return K.f1tab[(size_t)sh * 65536u + n];
}

// a2_InitRamper ... a2_SetRamper, a2_dsp.h:121-170, on four words {value, target, delta, timer}
VMFN void rp_prepare(int32_t *r, int frames)
{
	if(!r[3]) {
		r[0] = r[1];
		r[2] = 0;
	} else if(frames <= (r[3] >> 8)) {
		r[2] = (int)((((int64_t)vsub(r[1], r[0])) * 256) / r[3]);
		r[3] = vsub(r[3], frames << 8);
	} else {
		r[2] = vsub(r[1], r[0]) / frames;
		r[3] = 0;
	}
}
VMFN void rp_run(int32_t *r, int frames) { r[0] = vadd(r[0], vmul(r[2], frames)); }
VMFN void rp_set(int32_t *r, int target, int start, int duration)
{
	r[1] = (int)((unsigned)target << 8);
	r[3] = vadd(duration, start);
	if(r[3] < 256)
		r[0] = r[1];
	else
		r[0] = vadd(r[0], vmul(r[2], start) >> 8);
}

// A2_regtracker, core.c:1064-1099.  (The mask is 32 bits wide and the engine shifts a 1 by the
// register number: on its x86 targets registers r and r + 32 share a bit.  Kept.)
struct Tracker {
	uint32_t mask, position;
	uint8_t regs[A2AMD_VM_REGISTERS];
};
VMFN void rt_mark(Tracker &rt, unsigned r)
{
	const uint32_t b = 1u << (r & 31u);
	if(b & rt.mask)
		return;
	rt.mask |= b;
	rt.regs[rt.position++ & 63u] = (uint8_t)r;
}
VMFN void rt_unmark(Tracker &rt, unsigned r)
{
	const uint32_t b = 1u << (r & 31u);
	if(b & rt.mask) {
		rt.mask &= ~b;
		for(uint32_t i = 0; i < rt.position; ++i)
			if(rt.regs[i] == r) {
				rt.regs[i] = rt.regs[--rt.position];
				break;
			}
	}
}

// Can the device VM make this register write?  (kind = a2amd_unitkind, reg = the unit's register
// index.)  Not: wtosc 'w' (a wave handle: the host resolves and uploads waves, and the voice's
// launch class hangs on it), fbdelay's tap lengths (launch class), anything without registers.
VMFN bool write_supported(int kind, int reg)
{
	switch(kind) {
	  case A2D_WTOSC:      return reg >= 1 && reg <= 3;
	  case A2D_PANMIX:     return reg >= 0 && reg <= 1;
	  case A2D_FILTER12:   return reg >= 0 && reg <= 4;
	  case A2D_FBDELAY:    return reg >= 3 && reg <= 6;
	  case A2D_DC:         return reg >= 0 && reg <= 1;
	  case A2D_WAVESHAPER: return reg == 0;
	  case A2D_DCBLOCK:    return reg == 0;
	  case A2D_LIMITER:    return reg >= 0 && reg <= 1;
	  default:
		if(A2D_IS_FM(kind)) {
			const int nops = kind == A2D_FM1 ? 1 : (kind == A2D_FM2 || kind == A2D_FM2R) ? 2 :
					(kind == A2D_FM3 || kind == A2D_FM3P) ? 3 : 4;
			return reg >= 0 && reg <= 3 * nops;
		}
		return false;
	}
}
VMFN bool write_needs_f1tab(int kind, int reg)
{
	return (kind == A2D_FILTER12 || kind == A2D_DCBLOCK) && reg == 0;
}

template<class E>
VMFN void write_unit(A2DVmVoice &v, const Consts &K, E &e, int frag, unsigned m, int value, unsigned start, unsigned dur);

// env_Target, env.c:137-215: a write to an env unit's 'target' register.  Linear / zero-length segments go
// straight through the control wire; the others set up a unity ramp over a table that env_lut() below
// walks window by window.  ci = the unit's registers in the VM's register file (target, mode, down, time).
template<class E>
VMFN void env_target(A2DVmVoice &v, const Consts &K, E &e, int frag, int slot, int val, unsigned start, unsigned dur)
{
	A2DVmEnv &en = v.env[slot & (A2D_VM_MAXENV - 1)];
	const int32_t *ci = v.r + en.regbase;
	if(en.target == A2D_VM_NOWRITE)		// (output not connected)
		return;
	if(ci[3])				// ramp duration override, env.c:148-149
		dur = ms2t(en_msdur(K), ci[3]);
	int mode;
	if(dur >= 256u - start) {
		mode = ci[2] >> 16;
		if(val >= en.out || mode == 0)
			mode = ci[1] >> 16;
	} else
		mode = 1;
	int lut;
	if(mode == -1) {			// A2ENVRM_SPLINE
		lut = 0;
		mode = 1;
	} else if(mode >= 2 && mode <= 8)	// EXP1 .. EXP7
		lut = 1 + mode - 2;
	else if(mode >= -8 && mode <= -2)	// IEXP1 .. IEXP7
		lut = 1 - mode - 2;
	else {					// LINK, LINEAR, anything else
		en.out = val;
		en.active = 0;
		write_unit(v, K, e, frag, en.target, val, start, dur);
		return;
	}
	en.lut = lut;
	int rstart, rend;
	if(mode >= 0) {
		rstart = 0;
		rend = 1 << 16;
		en.scale = vsub(val, en.out);
		en.offset = en.out;
	} else {
		rstart = 1 << 16;
		rend = 0;
		en.scale = vsub(en.out, val);
		en.offset = vsub(en.out, en.scale);
	}
	en.ramper[0] = (int)((unsigned)rstart << 8);
	rp_set(en.ramper, rend, (int)start, (int)dur);
	en.active = 1;
}

// env_ProcessLUT, env.c:116-134: one window of a running segment; the write goes to the wired register with
// the window's offset as start and its length as duration
VMFN void env_lut(const Consts &K, A2DVmEnv &en, int frames)
{
	const uint16_t *t = K.envlut + (size_t)en.lut * (A2D_ENV_LUTSIZE + 2);
	rp_prepare(en.ramper, frames);
	rp_run(en.ramper, frames);
	uint32_t i = (uint32_t)(en.ramper[0] >> (24 - A2D_ENV_LUTSHIFT));
	// A re-targeted unity ramp can overshoot 1.0 by a window's worth: the engine then reads on past the table's
	// two pad entries - into the NEXT table of its one malloc'ed array of eight (env.c:32-35, 259), which is
	// how ours is laid out too, so the same words come back.  Past the last table it reads the heap (or
	// crashes: a negative ramp value is a huge index); that nobody can follow, and the index stays in the array.
	const uint32_t imax = (uint32_t)(A2D_ENV_LUTS - en.lut) * (A2D_ENV_LUTSIZE + 2) - 2;
	if(i > imax)
		i = imax;
	const uint32_t f = (uint32_t)(en.ramper[0] >> (24 - 16 - A2D_ENV_LUTSHIFT)) & 65535u;
	// (as the engine computes it: unsigned 32 bit products, then int)
	en.out = (int)((f * (uint32_t)t[i + 1] + (65536u - f) * (uint32_t)t[i]) >> 7);
	en.out = vadd((int)(((int64_t)en.out * (int64_t)en.scale) >> 24), en.offset);
	if(!en.ramper[2])
		en.active = 0;
}

// a2_VoiceControl (core.c:143-149) -> the unit's write callback -> the record a2amd_unit_write
// (a2amd_host.cpp) makes of it.  E: rec(frag, op, unit, reg, value, dur, start).
template<class E>
VMFN void control(A2DVmVoice &v, const Consts &K, E &e, int frag, unsigned reg, unsigned start, unsigned dur)
{
	const unsigned m = v.cmap[reg & 63u];
	if(m == A2D_VM_NOWRITE)
		return;
	if(m == A2D_VM_TRAPWRITE) {	// (only ever met by the host's look-ahead, which ends the voice's stay in front of this VM run)
		v.fault = TRAP_TARGET;
		return;
	}
	if((m >> 4) == A2D_VM_ENVPOS) {		// an env unit's 'target' register
		env_target(v, K, e, frag, (int)(m & 15u), v.r[reg & 63u], start & 255u, dur);
		return;
	}
	write_unit(v, K, e, frag, m, v.r[reg & 63u], start, dur);
}

// the unit's write callback -> the record a2amd_unit_write makes of it (m = chain position << 4 | register)
template<class E>
VMFN void write_unit(A2DVmVoice &v, const Consts &K, E &e, int frag, unsigned m, int value, unsigned start, unsigned dur)
{
	const int pos = (int)(m >> 4), ureg = (int)(m & 15u), kind = v.kind[pos & (A2D_VM_MAXPOS - 1)];
	const int transpose = v.r[A2AMD_VM_R_TRANSPOSE];
	start &= 255u;
	switch(kind) {
	  case A2D_WTOSC:
		if(ureg == 1)		// wtosc_Pitch, wtosc.c:486-492
			value = vadd(vadd(value, transpose), K.basepitch);
		break;
	  case A2D_FILTER12:
		if(ureg == 0) {		// f12_CutOff, filter12.c:141-147: the ramper lives here
			for(int k = 0; k < (int)v.ncut && k < A2D_VM_MAXCUT; ++k)
				if(v.cutpos[k] == pos) {
					rp_set(v.cut[k], vadd(value, transpose), (int)start, (int)dur);
					if(dur < 256)
						e.rec(frag, R_F1SET, pos, 0, f1_of_pitch(K, v.cut[k][0] >> 8), 0, 0);
				}
			return;
		}
		if(ureg == 1)		// f12_Q, filter12.c:149-162
			value = value < 512 ? 32768 : (65536 << 8) / value;
		break;
	  case A2D_DCBLOCK:		// dcb_CutOff, dcblock.c:112-117
		value = f1_of_pitch(K, vadd(value, transpose));
		break;
	  case A2D_LIMITER:		// limiter_Release / limiter_Threshold, limiter.c:201-213
		if(ureg == 0)
			value = (int)((unsigned)value << 8) / K.samplerate;
		else {
			const unsigned t = (unsigned)value << 8;
			value = (int)(t < 256u ? 256u : t);
		}
		break;
	  default:
		if(A2D_IS_FM(kind) && ureg == 1)	// fm_Pitch, fm.c:403-483: operator 0 is the absolute one
			value = vadd(vadd(value, transpose), K.basepitch);
		break;
	}
	e.rec(frag, R_WRITE, pos, ureg, value, dur, start);
}

template<class E>
VMFN void rt_apply(const Tracker &rt, A2DVmVoice &v, const Consts &K, E &e, int frag, unsigned start, unsigned dur)
{
	for(uint32_t i = 0; i < rt.position; ++i)
		control(v, K, e, frag, rt.regs[i], start, dur);
}

// a2_VoiceProcessVM, core.c:1166-1744: instructions until a timing instruction with dt > 0.
// Returns TRAP_NONE when the VM has rescheduled itself, a TRAP_* where the engine's would have
// aborted (which a2amd_vm_analyze() has ruled out for every adopted voice).
template<class E>
VMFN int run(A2DVmVoice &v, const uint32_t *code, const Consts &K, E &e, int frag, Tracker *rtmem = nullptr)
{
	int32_t *r = v.r;
	unsigned inscount = A2AMD_VM_INSLIMIT;
	// (the tracker's register list is indexed at run time: the kernel hands in a place in LDS for it)
	Tracker rtlocal;
	Tracker &rt = rtmem ? *rtmem : rtlocal;
	if(e.resuming())	// (a run that gave way, below: its tracker - in rtmem - and its instruction count go on)
		inscount = e.resume();
	else
		rt.mask = rt.position = 0;
	if(v.state == A2AMD_VM_WAITING)
		v.state = A2AMD_VM_RUNNING;
	for(;;) {
		if(v.fault)
			return v.fault;
		if(e.crowded()) {
			e.yield(inscount);
			return RUN_YIELD;
		}
		if(v.pc >= v.ncode)
			return TRAP_PC;
		const uint32_t w = code[v.pc];
		const unsigned op = A2AMD_VM_OPCODE(w), a1 = A2AMD_VM_A1(w) & 63u, a2 = A2AMD_VM_A2(w);
		// (two-word instructions keep a3 in the next word; read where it exists)
		const int32_t a3 = (v.pc + 1u < v.ncode) ? (int32_t)code[v.pc + 1] : 0;
		unsigned dt = 0, cdur = 0;
		bool timing = false;
		int ctl = 0;		// 1: a write through register a1's wire, 2: through all the tracker holds
		if(!--inscount)
			return TRAP_OVERLOAD;
		switch(op) {
		  // local flow control, core.c:1282-1320
		  case A2AMD_OP_JUMP:
			v.pc = (uint16_t)a2;
			continue;
		  case A2AMD_OP_LOOP:
			r[a1] = vsub(r[a1], 65536);
			if(r[a1] <= 0)
				break;
			v.pc = (uint16_t)a2;
			continue;
		  case A2AMD_OP_JZ:
			if(r[a1])
				break;
			v.pc = (uint16_t)a2;
			continue;
		  case A2AMD_OP_JNZ:
			if(!r[a1])
				break;
			v.pc = (uint16_t)a2;
			continue;
		  case A2AMD_OP_JG:
			if(r[a1] <= 0)
				break;
			v.pc = (uint16_t)a2;
			continue;
		  case A2AMD_OP_JL:
			if(r[a1] >= 0)
				break;
			v.pc = (uint16_t)a2;
			continue;
		  case A2AMD_OP_JGE:
			if(r[a1] < 0)
				break;
			v.pc = (uint16_t)a2;
			continue;
		  case A2AMD_OP_JLE:
			if(r[a1] > 0)
				break;
			v.pc = (uint16_t)a2;
			continue;

		  // timing, core.c:1323-1336
		  case A2AMD_OP_DELAY:
			dt = ms2t(K.msdur, a3);
			++v.pc;
			timing = true;
			break;
		  case A2AMD_OP_DELAYR:
			dt = ms2t(K.msdur, r[a1]);
			timing = true;
			break;
		  case A2AMD_OP_TDELAY:
			dt = ticks2t(K.msdur, r[A2AMD_VM_R_TICK], a3);
			++v.pc;
			timing = true;
			break;
		  case A2AMD_OP_TDELAYR:
			dt = ticks2t(K.msdur, r[A2AMD_VM_R_TICK], r[a1]);
			timing = true;
			break;

		  // arithmetics, core.c:1339-1411
		  case A2AMD_OP_SUBR:
			r[a1] = vsub(r[a1], r[a2 & 63u]);
			rt_mark(rt, a1);
			break;
		  // (register divisors: never proven by the static analysis - a voice whose program has them is
		  // taken for a stretch the host's look-ahead has run through, vm_lookahead in a2amd_vm.cpp)
		  case A2AMD_OP_DIVR:
			if(!r[a2 & 63u])
				return TRAP_DIVISOR;
			r[a1] = (int)(((int64_t)r[a1] << 16) / (int64_t)r[a2 & 63u]);
			rt_mark(rt, a1);
			break;
		  case A2AMD_OP_MODR:
			if(r[a2 & 63u] == 0 || r[a2 & 63u] == -1)
				return TRAP_DIVISOR;
			r[a1] %= r[a2 & 63u];
			rt_mark(rt, a1);
			break;
		  case A2AMD_OP_QUANTR:
			if(r[a2 & 63u] == 0 || r[a2 & 63u] == -1)
				return TRAP_DIVISOR;
			r[a1] = vmul(r[a1] / r[a2 & 63u], r[a2 & 63u]);
			rt_mark(rt, a1);
			break;
		  case A2AMD_OP_P2DR: {
			// A2_1K_DIV_MIDDLEC / a2_P2I(...), a2_pitch.h:42 (a zero increment divides by zero
			// in the engine; there is nothing to match)
			const unsigned d = p2i(K.ptab, r[a2 & 63u]);
			r[a1] = d ? (int)(4202608409623LL / (int64_t)d) : 0;
			rt_mark(rt, a1);
			break;
		  }
		  case A2AMD_OP_NEGR:
			r[a1] = vsub(0, r[a2 & 63u]);
			rt_mark(rt, a1);
			break;
		  case A2AMD_OP_LOAD:
			r[a1] = a3;
			rt_mark(rt, a1);
			++v.pc;
			break;
		  case A2AMD_OP_LOADR:
			r[a1] = r[a2 & 63u];
			rt_mark(rt, a1);
			break;
		  case A2AMD_OP_ADD:
			r[a1] = vadd(r[a1], a3);
			rt_mark(rt, a1);
			++v.pc;
			break;
		  case A2AMD_OP_ADDR:
			r[a1] = vadd(r[a1], r[a2 & 63u]);
			rt_mark(rt, a1);
			break;
		  case A2AMD_OP_MUL:
			r[a1] = (int)(((int64_t)r[a1] * (int64_t)a3) >> 16);
			rt_mark(rt, a1);
			++v.pc;
			break;
		  case A2AMD_OP_MULR:
			r[a1] = (int)(((int64_t)r[a1] * (int64_t)r[a2 & 63u]) >> 16);
			rt_mark(rt, a1);
			break;
		  case A2AMD_OP_MOD:
			if(a3 == 0 || a3 == -1)
				return TRAP_DIVISOR;
			r[a1] %= a3;
			rt_mark(rt, a1);
			++v.pc;
			break;
		  case A2AMD_OP_QUANT:
			if(a3 == 0 || a3 == -1)
				return TRAP_DIVISOR;
			r[a1] = vmul(r[a1] / a3, a3);
			rt_mark(rt, a1);
			++v.pc;
			break;

		  // comparison and boolean operators, core.c:1413-1456
		  case A2AMD_OP_GR:  r[a1] = (r[a1] > r[a2 & 63u]) << 16;  rt_mark(rt, a1); break;
		  case A2AMD_OP_LR:  r[a1] = (r[a1] < r[a2 & 63u]) << 16;  rt_mark(rt, a1); break;
		  case A2AMD_OP_GER: r[a1] = (r[a1] >= r[a2 & 63u]) << 16; rt_mark(rt, a1); break;
		  case A2AMD_OP_LER: r[a1] = (r[a1] <= r[a2 & 63u]) << 16; rt_mark(rt, a1); break;
		  case A2AMD_OP_EQR: r[a1] = (r[a1] == r[a2 & 63u]) << 16; rt_mark(rt, a1); break;
		  case A2AMD_OP_NER: r[a1] = (r[a1] != r[a2 & 63u]) << 16; rt_mark(rt, a1); break;
		  case A2AMD_OP_ANDR: r[a1] = (r[a1] && r[a2 & 63u]) << 16; rt_mark(rt, a1); break;
		  case A2AMD_OP_ORR:  r[a1] = (r[a1] || r[a2 & 63u]) << 16; rt_mark(rt, a1); break;
		  case A2AMD_OP_XORR: r[a1] = (!r[a1] != !r[a2 & 63u]) << 16; rt_mark(rt, a1); break;
		  case A2AMD_OP_NOTR: r[a1] = (!r[a2 & 63u]) << 16; rt_mark(rt, a1); break;

		  // unit control, core.c:1459-1489 (carried out behind the switch)
		  case A2AMD_OP_SET:
			ctl = 1;
			break;
		  case A2AMD_OP_SETALL:		// a2_RTSetAll, core.c:1109-1116
			ctl = 2;
			break;
		  case A2AMD_OP_RAMP:
			ctl = 1;
			cdur = ms2t(K.msdur, a3);
			++v.pc;
			break;
		  case A2AMD_OP_RAMPR:
			ctl = 1;
			cdur = ms2t(K.msdur, r[a2 & 63u]);
			break;
		  case A2AMD_OP_RAMPALL:
			ctl = 2;
			cdur = ms2t(K.msdur, a3);
			++v.pc;
			break;
		  case A2AMD_OP_RAMPALLR:
			ctl = 2;
			cdur = ms2t(K.msdur, r[a1]);
			break;

		  default:
			return TRAP_OPCODE;
		}
		++v.pc;
		// The writes through the control wires - one register (SET / RAMP*), all the tracker holds (SETALL / RAMPALL*:
		// a2_RTApply + reset, core.c:1101-1116), or what a timing instruction finds in the tracker ("timing:",
		// core.c:1719-1733) - in ONE place: control() with everything behind it (the units' write callbacks, the
		// recorder) is most of the interpreter's text, and seven copies of it were more than the instruction cache holds.
		if(ctl | (int)timing) {
			const bool one = ctl == 1;
			const unsigned nreg = one ? 1u : rt.position, dur = timing ? dt : cdur;
			for(unsigned i = 0; i < nreg; ++i)
				control(v, K, e, frag, one ? a1 : (unsigned)rt.regs[i], v.waketime, dur);
			if(one)
				rt_unmark(rt, a1);
			else if(!timing)
				rt.mask = rt.position = 0;
		}
		if(!timing)
			continue;
		if(v.fault)
			return v.fault;
		if(!dt)
			continue;
		v.state = A2AMD_VM_WAITING;
		v.waketime += dt;
		return TRAP_NONE;
	}
}

// a2_VoiceProcess (core.c:1847-1880) over fragments [f0, f1) of a batch whose fragment f0 starts at
// engine time 'now': at every window start the VM runs while it is due within that frame
// (a2_VoiceProcessVMEv's "VM only" loop, core.c:1822-1838), then the units get their window - of
// which the recorder keeps: a cutoff ramp's coefficient for the window (the head of f12_process,
// filter12.c:86-96, as a2amd_unit_process makes it), and the window itself unless it is the
// fragment's only event (the default window, a2amd_unit_process).  FF: frames of fragment f.
// Returns the engine time at the end of fragment f1 - 1; v.fault is set where a trap stopped the VM.
template<class E, class FF>
VMFN uint32_t run_batch(A2DVmVoice &v, const uint32_t *code, const Consts &K, E &e, uint32_t now, int f0, int f1, FF frames_of,
		Tracker *rtmem = nullptr)
{
	uint32_t fs = now;
	for(int f = f0; f < f1; ++f) {
		// (frames | offset of the fragment inside the engine's own fragment << 8: what the engine hands
		// a2_ProcessVoices as 'offset', core.c:1968 - not 0 where the root voice's program cut the fragment)
		const unsigned ff = frames_of(f);
		const int frames = (int)(ff & 255u), base = (int)(ff >> 8);
		const int before = e.count();
		int s = 0;
		while(s < frames) {
			const uint32_t t = fs + ((uint32_t)s << 8);
			int res = 0;
			bool gave_way = false;
			e.mark(0);
			for(;;) {
				const int nextvm = (int)(v.waketime - t);	// a2_TSDiff
				if(nextvm > 255) {
					res = nextvm >> 8;
					break;
				}
				if(v.fault || (v.has_exit && v.waketime == v.exit_when)) {	// (stopped for good; or at the run that is the engine's)
					res = frames;
					break;
				}
				const int trap = run(v, code, K, e, f, rtmem);
				if(trap == RUN_YIELD) {		// (k_vm_win: the records so far are carried out, then the run goes on)
					gave_way = true;
					break;
				}
				if(trap) {
					v.fault = trap;
					res = frames;
					break;
				}
			}
			e.mark(1);
			// (a run that gave way - k_vm_win: the records so far are carried out below, then the run goes on - has not
			// got to its window yet)
			if(!gave_way) {
				if(res > frames - s)
					res = frames - s;
				// The units get the window in chain order (core.c:1875-1876).  What the recorder keeps of that:
				// a running env segment's write through its control wire (env.c:116-134) - into the records
				// before the window if the target unit comes behind the env in the chain, behind it if the
				// target has already rendered the window (push_rec, a2amd_sched.cpp) - and a ramping cutoff's
				// coefficient for the window (the head of f12_process, filter12.c:86-96).
				int nlate = 0, late_slot[A2D_VM_MAXENV];
				// (round 6: only the chain positions that HOLD an env unit or a cutoff ramper are visited, in order - the
				// same visits in the same order as the walk over all 0 .. A2D_MAXCHAIN positions it replaces, which with chains of
				// up to 16 units was seventeen trips through these two loops per window, each re-reading the voice from LDS:
				// 3 800 cycles per window, a third of k_vm_win for the filter class, whose every voice has a cutoff
				// ramper - moving or not (-DWIN_PROF: profiles/r06_vm_win_cycles.txt).  Activity is looked at when a
				// position is visited, as before: an env's write may start or stop a later one.)
				unsigned pmask = 0;
				if(v.nenv | v.ncut) {
					for(int k = 0; k < v.nenv && k < A2D_VM_MAXENV; ++k)
						if((unsigned)v.env[k].k <= A2D_VM_MAXPOS)
							pmask |= 1u << v.env[k].k;
					for(int k = 0; k < (int)v.ncut && k < A2D_VM_MAXCUT; ++k)
						if(v.cutpos[k] <= A2D_VM_MAXPOS)
							pmask |= 1u << v.cutpos[k];
				}
				while(pmask) {
					const int p = __builtin_ctz(pmask);
					pmask &= pmask - 1;
					for(int k = 0; k < v.nenv && k < A2D_VM_MAXENV; ++k) {
						A2DVmEnv &en = v.env[k];
						if(en.k != p || !en.active)
							continue;
						env_lut(K, en, res);
						const bool cutoff = v.kind[(en.target >> 4) & (A2D_VM_MAXPOS - 1)] == A2D_FILTER12 && (en.target & 15u) == 0;
						if((int)(en.target >> 4) >= p || cutoff)	// (a cutoff write is host state, not a record)
							write_unit(v, K, e, f, en.target, en.out, (unsigned)(base + s), (unsigned)res << 8);
						else
							late_slot[nlate++] = k;
					}
					for(int k = 0; k < (int)v.ncut && k < A2D_VM_MAXCUT; ++k)
						if(v.cutpos[k] == p) {
							rp_prepare(v.cut[k], res);
							if(v.cut[k][2]) {
								rp_run(v.cut[k], res);
								e.rec(f, R_F1RAMP, v.cutpos[k], 0, f1_of_pitch(K, v.cut[k][0] >> 8), 0, 0);
							}
						}
				}
				// (the fragment's default window - nothing else in it - is no record; an emitter that is the
				// records' reader as well gets every window)
				if(E::fused || !(s == 0 && res == frames && e.count() == before && !nlate))
					e.rec(f, R_SEG, 0, 0, 0, (unsigned)s | ((unsigned)res << 16), 0);
				for(int k = 0; k < nlate; ++k) {
					const A2DVmEnv &en = v.env[late_slot[k]];
					write_unit(v, K, e, f, en.target, en.out, (unsigned)(base + s), (unsigned)res << 8);
				}
				s += res;
			}
			e.mark(2);
			e.drain(f);
		}
		e.end_fragment(f);
		fs += (uint32_t)frames << 8;
	}
	return fs;
}

// (what an emitter without a queue of its own says to run() / run_batch())
struct PlainE {
	static constexpr bool fused = false;
	VMFN bool resuming() const { return false; }
	VMFN unsigned resume() { return 0; }
	VMFN bool crowded() const { return false; }
	VMFN void yield(unsigned) {}
	VMFN void drain(int) {}
	VMFN void end_fragment(int) {}
	VMFN void mark(int) {}		// (measurement builds: where run_batch is)
};

// What k_vm_win (a2amd_vmwin.hip) takes from the window pool for a voice: one entry per window of a fragment beyond
// its first.  Run ahead of the batch it is about (k_vm_pool, a2amd_vm.hip), on a copy of the voice.
struct PoolE : PlainE {
	static constexpr bool fused = true;	// (every window is a record: run_batch)
	int nseg, total, pool;
	VMFN void rec(int, int op, int, int, int, unsigned, unsigned) { ++total; nseg += op == R_SEG; }
	VMFN int count() const { return total; }
	VMFN void end_fragment(int)
	{
		if(nseg > 1)
			pool += nseg - 1;
		nseg = 0;
	}
};

struct CountE : PlainE {
	int n;
	VMFN void rec(int, int, int, int, int, unsigned, unsigned) { ++n; }
	VMFN int count() const { return n; }
};

struct StoreE : PlainE {
	A2DRec *out;
	int n;
	VMFN void rec(int frag, int op, int unit, int reg, int value, unsigned dur, unsigned start)
	{
		A2DRec r;
		r.head = A2D_HEAD(frag, op, unit, reg);
		r.value = value;
		r.dur = dur;
		r.start = start;
		out[n++] = r;
	}
	VMFN int count() const { return n; }
};

} // namespace a2vm
