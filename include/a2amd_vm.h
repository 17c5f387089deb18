/*
 * a2amd_vm.h - the scripted voice's VM on the device: C ABI of the part of
 * liba2amd.so that stands in for a2_VoiceProcessVM / a2_VoiceProcessVMEv / a2_VoiceProcess
 * (src/core.c:1166-1744, :1784-1839, :1847-1880) for voices whose program the host has PROVEN to
 * stay inside a subset of the instruction set:
 *
 *   flow      JUMP LOOP JZ JNZ JG JL JGE JLE                     core.c:1282-1320
 *   timing    DELAY DELAYR TDELAY TDELAYR                        core.c:1323-1336, :1721-1733
 *   maths     SUBR P2DR NEGR LOAD LOADR ADD ADDR MUL MULR        core.c:1339-1388
 *             MOD QUANT (immediate divisor not 0 / -1)
 *   compare   GR LR GER LER EQR NER ANDR ORR XORR NOTR           core.c:1413-1456
 *   control   SET SETALL RAMP RAMPR RAMPALL RAMPALLR             core.c:1459-1489
 *             (a2_VoiceControl, core.c:143-149, through the register tracker, core.c:1064-1116)
 *
 *             DIVR MODR QUANTR (register divisors, since round 5: a zero divisor ends the stay, below)
 *
 * That subset is what the device EXECUTES.  Which voices it is given (rounds 5-6):
 *
 *  - a voice whose program a2amd_vm_analyze() can vouch for as a whole from its current pc - it never reaches
 *    END / SLEEP / RETURN / CALL, spawning, messages, RAND (the engine-global RNG), a zero divisor, or a loop
 *    without a delay of at least one 256th of a frame in it (A2_OVERLOAD, core.c:1190) - is the device's for
 *    good (round 4's rule);
 *  - any OTHER voice that waits in a delay is taken FOR A STRETCH: a2amd_vm_adopt() runs the host's copy of
 *    the interpreter ahead on a copy of the state (its future runs depend on nothing but its own registers:
 *    an event recalls it, env units and cutoff rampers do not read back) up to the first VM run that would
 *    meet an instruction outside the subset.  That run is the ENGINE'S: its wake time is the voice's exit time
 *    (a2amd_vm_exit_time), the walk recalls the voice in the fragment that holds it, and the engine's own VM
 *    executes END (with a2_VoiceFree, the parent's wake-up, the detach logic), the CALL, the RAND draw, the
 *    SLEEP, the message - at the engine's own frame.  The device never executes those; it is told when to stop.
 *    A stay shorter than 2 048 frames or without a single write is not worth the hand-over (A2AMD_ESTATE: "not
 *    now" - the units back off for 8 ... 4 096 fragments before they ask again, round 6); a voice whose very
 *    next run is the engine's is refused from that pc (A2AMD_EUNSUPPORTED).
 *
 * Never taken: voices with a noise oscillator (the draws of the engine's one LCG interleave with every other noise
 * voice's and with RAND in WALK order, wtosc.c:135, core.c:1400), with subvoices, with a call stack (inside a
 * function or a message handler), with an API handle, with events queued; voices that own a bus (an inline unit) or
 * whose chain is longer than 8 units (the register map names a chain position in three bits; the backend itself
 * renders chains of up to A2AMD_MAXCHAIN = 16).
 *
 * Either way nothing the engine does depends on WHEN an adopted voice's VM runs.  The engine hands the voice
 * over (a2amd_vm_adopt: program text, A2_vmstate, which VM register feeds which unit register),
 * stops visiting it (INTEGRATION.md option C: a2amd_walk.c treats it as asleep), and the device
 * runs the VM between the unit windows: a kernel with one lane per voice interprets the ENGINE'S
 * OWN bytecode.  For the three window-kernel classes (wtosc [wtosc] [filter12] panmix) it carries the writes out on
 * the voice's control state in the same lane and writes the window entries the render pass reads (k_vm_win,
 * a2amd_vmwin.hip); for every other chain it writes, per batch, the same command records the host would have
 * recorded from the engine's write / Process calls (R_WRITE, R_SEG, R_F1SET, R_F1RAMP; a2amd_device.h), which
 * the leaf kernels execute as before.  When the engine needs the voice back - an event in its
 * queue, a window that is not the backend's fragment, a kill, its exit time - a2amd_vm_recall() returns the
 * A2_vmstate the engine's own VM would have reached by the start of the open fragment.
 *
 * Plain C types only.  The instruction encoding and the opcode numbers are the engine's
 * (src/internals.h:152-231); a2amd_walk.c, compiled against that header, static-asserts every
 * value below, tests/test_plugin_abi.py does the same from the test suite.
 */
#ifndef A2AMD_VM_H
#define A2AMD_VM_H

#include <stdint.h>
#include "a2amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define A2AMD_VM_REGISTERS	64	/* A2_REGISTERS, include/a2_vm.h:31 */
#define A2AMD_VM_INSLIMIT	1000	/* A2_INSLIMIT, src/config.h:119 */
#define A2AMD_VM_R_TICK		0	/* R_TICK, include/a2_vm.h:52 */
#define A2AMD_VM_R_TRANSPOSE	1	/* R_TRANSPOSE */

/* A2_vstates, include/a2_vm.h:41-48 */
enum { A2AMD_VM_RUNNING = 0, A2AMD_VM_WAITING, A2AMD_VM_INTERRUPT, A2AMD_VM_ENDING, A2AMD_VM_FINALIZING };

/* A2_opcodes, src/internals.h:152-207 (OP_END = 0, then A2_ALLINSTRUCTIONS in order) */
#define A2AMD_VM_ALLOPS \
	A2V(END) A2V(RETURN) A2V(CALL) \
	A2V(JUMP) A2V(LOOP) A2V(JZ) A2V(JNZ) A2V(JG) A2V(JL) A2V(JGE) A2V(JLE) \
	A2V(DELAY) A2V(DELAYR) A2V(TDELAY) A2V(TDELAYR) \
	A2V(SLEEP) A2V(WAKE) A2V(FORCE) \
	A2V(SUBR) A2V(DIVR) A2V(P2DR) A2V(NEGR) A2V(LOAD) A2V(LOADR) A2V(ADD) A2V(ADDR) \
	A2V(MUL) A2V(MULR) A2V(MOD) A2V(MODR) A2V(QUANT) A2V(QUANTR) A2V(RAND) A2V(RANDR) \
	A2V(GR) A2V(LR) A2V(GER) A2V(LER) A2V(EQR) A2V(NER) \
	A2V(ANDR) A2V(ORR) A2V(XORR) A2V(NOTR) \
	A2V(SET) A2V(SETALL) A2V(RAMP) A2V(RAMPR) A2V(RAMPALL) A2V(RAMPALLR) \
	A2V(PUSH) A2V(PUSHR) \
	A2V(SPAWN) A2V(SPAWNR) A2V(SPAWND) A2V(SPAWNDR) A2V(SPAWNV) A2V(SPAWNVR) A2V(SPAWNA) A2V(SPAWNAR) \
	A2V(SEND) A2V(SENDR) A2V(SENDA) A2V(SENDS) A2V(WAIT) A2V(KILL) A2V(KILLR) A2V(KILLA) \
	A2V(DETACH) A2V(DETACHR) A2V(DETACHA) \
	A2V(DEBUG) A2V(DEBUGR) \
	A2V(INITV) A2V(SIZEOF) A2V(SIZEOFR)
#define A2V(x) A2AMD_OP_##x,
typedef enum a2amd_vm_opcodes { A2AMD_VM_ALLOPS A2AMD_VM_OPCODES } a2amd_vm_opcodes;
#undef A2V

/* A2_instruction, src/internals.h:218-224: 32 bit words; opcode:8 a1:8 a2:16 [a3:32 in the next
 * word for the two-word instructions, a2_InsSize, src/compiler.c:111-131] (little endian) */
#define A2AMD_VM_OPCODE(w)	((unsigned)(w) & 0xffu)
#define A2AMD_VM_A1(w)		(((unsigned)(w) >> 8) & 0xffu)
#define A2AMD_VM_A2(w)		(((unsigned)(w) >> 16) & 0xffffu)

/* A2_vmstate, include/a2_vm.h:62-69, field for field */
typedef struct a2amd_vm_state {
	uint32_t waketime;	/* 24:8 frames, engine time (A2_state.now_fragstart's clock) */
	uint8_t  state;		/* A2AMD_VM_RUNNING ... */
	uint8_t  func;
	uint16_t pc;
	int32_t  r[A2AMD_VM_REGISTERS];
} a2amd_vm_state;

/* Why a program is not taken (a2amd_vm_info.reason) */
enum {
	A2AMD_VM_OK = 0,
	A2AMD_VM_BADCODE,	/* pc / jump target outside the function, truncated instruction */
	A2AMD_VM_OPCODE_OUT,	/* reaches an instruction outside the subset (info.opcode, info.at) */
	A2AMD_VM_DIVISOR,	/* MOD / QUANT by 0 or -1 */
	A2AMD_VM_NOYIELD,	/* a cycle without a delay that is certain to be > 0, or a straight run
				 * of A2_INSLIMIT instructions: A2_OVERLOAD cannot be ruled out */
	A2AMD_VM_TARGET,	/* may write a VM register wired to a unit register the device VM does not
				 * write (wtosc 'w', fbdelay's tap lengths, a unit that is not ours ...) */
	A2AMD_VM_IDLE		/* never writes a unit register: nothing to gain (the voice sleeps anyway) */
};

typedef struct a2amd_vm_info {
	int32_t  reason;	/* A2AMD_VM_OK: adoptable */
	int32_t  opcode, at;	/* the offending instruction (A2AMD_VM_OPCODE_OUT, _DIVISOR, _NOYIELD) */
	uint64_t written;	/* VM registers some reachable instruction stores to */
	uint64_t controlled;	/* VM registers a reachable SET / RAMP / timing instruction may pass to
				 * a2_VoiceControl (explicitly, or through the register tracker) */
	uint32_t reachable;	/* instructions reachable from pc */
	uint32_t longest;	/* most instructions between two certain yields */
} a2amd_vm_info;

/* Static analysis of the function 'code' (nwords 32 bit words) from 'pc': everything the voice can
 * still execute.  tick = the voice's r[R_TICK] (TDELAY immediates are certain yields only while
 * no reachable instruction writes R_TICK), msdur = A2_state.msdur (src/audiality2.c:499).  Pure
 * host code: callable without a GPU.  Returns info->reason. */
int a2amd_vm_analyze(const uint32_t *code, unsigned nwords, unsigned pc, int32_t tick, uint32_t msdur,
		a2amd_vm_info *info);

/* The text of one function (A2_function.code / .size, src/internals.h:417-425), uploaded once per
 * context; 'key' identifies it on the host (the code pointer).  Returns a program id >= 0. */
int a2amd_vm_program(a2amd_ctx *ctx, uint64_t key, const uint32_t *code, unsigned nwords);

/* An env unit in the voice's chain (src/units/env.c): a control-rate unit without audio ports
 * whose output is a register write on another unit through a control wire (env.c:135).  For a voice the
 * device VM runs, its state travels with the VM state and its segments are evaluated on the device - window
 * by window, in chain order - by the same interpreter. */
#define A2AMD_VM_MAXENV 2
typedef struct a2amd_vm_env {
	int32_t ramper[4];	/* A2_env.ramper {value, target, delta, timer}, env.c:92 */
	int32_t lut;		/* A2ENVLUT_* (env.c:36-47) of the running segment */
	int32_t scale, offset, out;	/* env.c:96-98 */
	int32_t active;		/* its Process is env_ProcessLUT (env.c:116-134), not env_ProcessOff */
	int32_t regbase;	/* the VM register its 'target' register is (then mode, down, time) */
	int32_t before;		/* backend units of the voice's chain in front of it */
	int32_t out_unit;	/* backend unit its control output is wired to (a2_ControlWire, core.c:330-345), -1 none */
	int32_t out_reg;	/* ... and that unit's register */
} a2amd_vm_env;
/* The eight tables of env.c:218-257 (spline, EXP1 .. EXP7; 64 + 2 entries each), made by the host with the
 * reference's expressions (float / double libm): once per context, before the first voice with an env unit
 * is adopted. */
int a2amd_vm_envluts(a2amd_ctx *ctx, const uint16_t *luts8x66);

/* Hand the voice 'head_unit' belongs to over to the device VM, from the OPEN fragment on (in which
 * the engine has already processed it: the device takes over with the next one).
 *   st        the voice's A2_vmstate as the engine leaves it now (state A2AMD_VM_WAITING)
 *   wr_unit   [A2AMD_VM_REGISTERS] backend unit id VM register i is wired to (A2_voice.cregs[i],
 *             src/internals.h:573: the unit whose write callback a2_VoiceControl calls), -1 none,
 *             -2 somebody else's callback, -3 - k the 'target' register of envs[k]
 *   envs      the voice's env units, nenv <= A2AMD_VM_MAXENV of them (may be NULL / 0)
 *   wr_reg    [A2AMD_VM_REGISTERS] ... and that unit's register index (a2amd_unit_write's 'reg')
 *   now       engine time (A2_state.now_fragstart + (offset << 8), core.c:1855) of the START of the
 *             open fragment
 *   msdur     A2_state.msdur
 * Fails with A2AMD_EUNSUPPORTED (and a reason in a2amd_last_error) when a2amd_vm_analyze() or the
 * voice's chain says no; the voice then simply stays with the engine. */
int a2amd_vm_adopt(a2amd_ctx *ctx, int head_unit, int prog, const a2amd_vm_state *st,
		const int32_t *wr_unit, const uint8_t *wr_reg, uint32_t now, uint32_t msdur,
		const a2amd_vm_env *envs, int nenv);

/* A voice whose program a2amd_vm_analyze() cannot vouch for as a whole - it reaches END, SLEEP, CALL /
 * RETURN, WAKE / FORCE, spawning, messages, RAND, a register divisor that may be zero, a loop it cannot
 * bound: every note of every song - is taken for the stretch of its future that needs none of that: the
 * host runs the voice's coming VM runs ahead on a copy of its registers (they depend on nothing else)
 * up to the first one that meets such an instruction.  That run, and what follows, is the engine's:
 * returns 1 and *when = the engine time (24:8) it starts at - the voice has to be recalled in the
 * fragment that holds that time, before the engine processes it there - or 0 for a voice taken for
 * good (or not taken). */
int a2amd_vm_exit_time(a2amd_ctx *ctx, int head_unit, uint32_t *when);

/* Round 6: where the NEXT batch's fragments will be cut.  The engine cuts the root voice's 64-frame window where the
 * root's own VM wakes up (a2_VoiceProcess, core.c:1852-1878: at a window start 'now' it processes (waketime - now) >> 8
 * frames if that difference exceeds 255 ticks) - and an attached root that has reached END wakes every 1 000 000 ticks
 * for ever (OP_END, core.c:1191-1217: about one cut per 4 096-frame buffer at 48 kHz).  Whoever sees the root's
 * A2_vmstate (the drop-in units) says before every a2amd_render() when the root will wake next: 'when' = up to 16
 * engine times (24:8 ticks), ascending.  The backend uses them for nothing but its SPECULATIVE pass (the device VM's
 * work for the batch expected next, done ahead - a2amd_vm.cpp vm_speculate): a batch whose fragments are not the ones
 * predicted simply does not take it.  Never required; n = 0 forgets the last announcement. */
int a2amd_vm_expect_cuts(a2amd_ctx *ctx, const uint32_t *when, int n);

/* 1 while the voice 'head_unit' belongs to is run by the device VM. */
int a2amd_vm_adopted(a2amd_ctx *ctx, int head_unit);

/* The engine wants n voices back, BEFORE it processes them in the open fragment: out[k] = the
 * A2_vmstate the engine's own a2_VoiceProcess calls would have left in voice k by now (every VM run
 * due before the start of the open fragment executed, none of those due in it).  From the open
 * fragment on the voices are the host's again (their unit windows and writes arrive as calls).
 * One device round trip for the lot.  envs_out (may be NULL): [n][A2AMD_VM_MAXENV], the state of voice k's
 * env units as of the same moment (ramper, lut, scale, offset, out, active). */
int a2amd_vm_recall(a2amd_ctx *ctx, const int32_t *head_units, unsigned n, a2amd_vm_state *out,
		a2amd_vm_env *envs_out);

/* ---- test / measurement access (no GPU needed) -----------------------------------------*/
/* Run the HOST copy of the interpreter - the same source the kernel is compiled from
 * (csrc/a2amd_vmcore.h) - over 'nfrags' fragments of fragframes[] frames, the first starting at
 * engine time 'now', and return the command records it emits (a2amd_device.h: A2DRec, 4 words
 * each) in recs[0 .. return value), at most cap of them; *st is advanced.  kinds[chainpos] = the
 * unit kinds of the voice's chain, wr_unit here = chain position.  f1tab may be NULL (no cutoff
 * writes).  The parity tests drive this against traces of the compiled reference. */
int a2amd_vm_trace_host(const uint32_t *code, unsigned nwords, a2amd_vm_state *st,
		const int32_t *wr_unit, const uint8_t *wr_reg, const int32_t *kinds, int nkinds,
		uint32_t now, uint32_t msdur, int32_t samplerate, int32_t basepitch,
		const uint8_t *fragframes, unsigned nfrags, uint32_t *recs, unsigned cap);

/* The same with the voice's env units (envs[k].out_unit = chain position of the wired unit, -1 none; wr_unit[r] = -3 - k
 * for the VM register that is envs[k]'s 'target'; envluts = the eight tables of a2amd_vm_envluts(); *envs is advanced)
 * and with the offset of every fragment inside the engine's own fragment (a2amd_fragment_offset; NULL: all 0). */
int a2amd_vm_trace_host_env(const uint32_t *code, unsigned nwords, a2amd_vm_state *st,
		const int32_t *wr_unit, const uint8_t *wr_reg, const int32_t *kinds, int nkinds,
		uint32_t now, uint32_t msdur, int32_t samplerate, int32_t basepitch,
		const uint8_t *fragframes, const uint8_t *fragbase, unsigned nfrags,
		a2amd_vm_env *envs, int nenv, const uint16_t *envluts, uint32_t *recs, unsigned cap);

/* The same for a voice taken for a stretch (a2amd_vm_exit_time): the look-ahead a2amd_vm_adopt makes for a program
 * the analysis cannot vouch for as a whole runs first - *has_exit = 1, *exit_when = the wake time of the first VM run
 * that is the engine's (END, SLEEP, CALL, RAND ..., a write the device VM cannot make), *stay_records (may be NULL) =
 * the writes the stay holds - and the interpreter stops in front of that run: the records returned are what the
 * device VM produces up to there (default windows afterwards), *st is the state the engine gets back. */
int a2amd_vm_trace_host_exit(const uint32_t *code, unsigned nwords, a2amd_vm_state *st,
		const int32_t *wr_unit, const uint8_t *wr_reg, const int32_t *kinds, int nkinds,
		uint32_t now, uint32_t msdur, int32_t samplerate, int32_t basepitch,
		const uint8_t *fragframes, const uint8_t *fragbase, unsigned nfrags,
		a2amd_vm_env *envs, int nenv, const uint16_t *envluts, uint32_t *recs, unsigned cap,
		int32_t *has_exit, uint32_t *exit_when, int32_t *stay_records);

typedef struct a2amd_vm_stats {
	uint64_t adopted, recalled, released;	/* voices, so far */
	uint64_t vm_voice_batches;		/* sum over batches of voices the VM kernel ran */
	uint64_t vm_records;			/* records it emitted */
	uint32_t live;				/* voices under the device VM now */
	uint32_t programs;
} a2amd_vm_stats;
int a2amd_vm_get_stats(a2amd_ctx *ctx, a2amd_vm_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* A2AMD_VM_H */
