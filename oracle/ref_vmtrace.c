/*
 * ref_vmtrace.c - what the COMPILED REFERENCE's VM does to one scripted voice, for pinning the device VM's
 * interpreter (audiality2_amd/csrc/a2amd_vmcore.h) on the CPU.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile against the reference's headers - src/internals.h too:
 * the program text, the VM state and the voice's register write table are engine internals - and linked with
 * oracle/_ref/libaudiality2.so.  Never shipped, never on the product path.
 *
 *   ref_vmtrace <file.a2s> <program> <warm_fragments> <fragments> [args...]
 *
 * starts <program> as a voice under the root voice, lets the engine run it for <warm_fragments> fragments of 64
 * frames (until its VM waits in a delay), then writes, as one JSON object on stdout:
 *   - what a2amd_vm_trace_host() needs to take over at that moment: the function's code words, the A2_vmstate,
 *     which VM register is wired to which register of which unit of the chain, the engine's clock and constants;
 *   - what the engine's own VM did with the voice over the next <fragments> fragments: every call of a unit's
 *     write callback (a2_VoiceControl, core.c:143-149: register, value, start, duration, and the voice's
 *     transpose register at that moment) and every window the voice's units were given (a2_VoiceProcess,
 *     core.c:1847-1880: offset, frames), in order ("p"; logged at the voice's first unit that is not an env, so that an env
 *     unit's write through its control wire, env.c:131, lands on the side of the window it belongs to), and every
 *     window of the ROOT voice ("r": where the engine cut its own fragments - the backend's fragments).
 *     A voice that ENDS inside the traced stretch (its program runs out: OP_END, core.c:1191-1235, then a2_VoiceFree)
 *     is traced up to the fragment it goes away in ("gone"; "fragments" = that fragment + 1, no "end_state").
 *   - for env units (src/units/env.c): where they sit in the chain, which VM registers are theirs, where the control
 *     output is wired.  Their internal state is private to env.c; a trace is therefore started before the program
 *     has written any env's 'target' (warm_fragments 0 and a delay at the head of the program), when it is the
 *     state env_Initialize leaves (env.c:225-250).
 * tests/golden/make_vm_traces.py keeps the output as fixtures; tests/test_device_vm.py replays them.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "internals.h"

#define MAXEV 200000
static struct { char kind; int frag, a, b; unsigned c, d; int e; } ev[MAXEV];
static int nev, cur_frag;
static A2_voice *voice;
static A2_write_cb orig_write[A2_REGISTERS];
static A2_process_cb orig_process;

static void log_write(int reg, int value, unsigned start, unsigned dur)
{
	if(nev < MAXEV)
	{
		ev[nev].kind = 'w'; ev[nev].frag = cur_frag; ev[nev].a = reg; ev[nev].b = value;
		ev[nev].c = start; ev[nev].d = dur; ev[nev].e = voice->s.r[R_TRANSPOSE];
		++nev;
	}
}

#define T(n) static void tw##n(A2_unit *u, int v, unsigned s, unsigned d) { log_write(n, v, s, d); orig_write[n](u, v, s, d); }
T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(8) T(9) T(10) T(11) T(12) T(13) T(14) T(15)
T(16) T(17) T(18) T(19) T(20) T(21) T(22) T(23) T(24) T(25) T(26) T(27) T(28) T(29) T(30) T(31)
T(32) T(33) T(34) T(35) T(36) T(37) T(38) T(39) T(40) T(41) T(42) T(43) T(44) T(45) T(46) T(47)
T(48) T(49) T(50) T(51) T(52) T(53) T(54) T(55) T(56) T(57) T(58) T(59) T(60) T(61) T(62) T(63)
#undef T
#define T(n) tw##n,
static const A2_write_cb tramp[A2_REGISTERS] = {
T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(8) T(9) T(10) T(11) T(12) T(13) T(14) T(15)
T(16) T(17) T(18) T(19) T(20) T(21) T(22) T(23) T(24) T(25) T(26) T(27) T(28) T(29) T(30) T(31)
T(32) T(33) T(34) T(35) T(36) T(37) T(38) T(39) T(40) T(41) T(42) T(43) T(44) T(45) T(46) T(47)
T(48) T(49) T(50) T(51) T(52) T(53) T(54) T(55) T(56) T(57) T(58) T(59) T(60) T(61) T(62) T(63) };

static A2_process_cb orig_root_process;
static void hook_root(A2_unit *u, unsigned offset, unsigned frames)
{
	if(nev < MAXEV)
	{
		ev[nev].kind = 'r'; ev[nev].frag = cur_frag; ev[nev].a = (int)offset; ev[nev].b = (int)frames;
		++nev;
	}
	orig_root_process(u, offset, frames);
}

/* the window, as the voice's first audio unit is given it */
static void hook_process(A2_unit *u, unsigned offset, unsigned frames)
{
	if(nev < MAXEV)
	{
		ev[nev].kind = 'p'; ev[nev].frag = cur_frag; ev[nev].a = (int)offset; ev[nev].b = (int)frames;
		++nev;
	}
	orig_process(u, offset, frames);
}

int main(int argc, const char *argv[])
{
	A2_driver *drv;
	A2_config *cfg;
	A2_interface *i;
	A2_state *st;
	A2_handle bank, prog;
	A2_voice *root;
	A2_unit *u, *last = NULL, *rootlast = NULL;
	int nenvs = 0;
	const A2_function *fn;
	int args[8], nargs = 0, warm, nfrags, k, r, tries = 0, gone = -1;
	RCHM_handleinfo *hi;
	if(argc < 5)
	{
		fprintf(stderr, "usage: ref_vmtrace <a2s> <program> <warm_fragments> <fragments> [args...]\n");
		return 1;
	}
	warm = atoi(argv[3]);
	nfrags = atoi(argv[4]);
	for(k = 5; k < argc && nargs < 8; ++k)
		args[nargs++] = (int)(atof(argv[k]) * 65536.0);
	if(!(drv = a2_NewDriver(A2_AUDIODRIVER, "buffer")) || !(cfg = a2_OpenConfig(48000, 64, 2, A2_AUTOCLOSE)))
		return 2;
	a2_AddDriver(cfg, drv);
	if(!(i = a2_Open(cfg)) || (bank = a2_Load(i, argv[1], 0)) < 0 || (prog = a2_Get(i, bank, argv[2])) < 0)
	{
		fprintf(stderr, "ref_vmtrace: cannot load %s / %s\n", argv[1], argv[2]);
		return 3;
	}
	st = ((A2_interface_i *)i)->state;
	a2_TimestampReset(i);
	/* (A2_VMTRACE_DETACHED=1: started the way a script's "Note P V" starts a voice - detached, no handle - so that a
	 * program that runs out really ends, core.c:1208-1221; a2_Starta's voice hangs around at END for its handle) */
	if(getenv("A2_VMTRACE_DETACHED") ? a2_Playa(i, a2_RootVoice(i), prog, nargs, args) != A2_OK :
			a2_Starta(i, a2_RootVoice(i), prog, nargs, args) < 0)
		return 4;
	for(k = 0; k < warm; ++k)
		a2_Run(i, 64);
	hi = rchm_Get(&st->ss->hm, st->rootvoice);
	root = hi ? (A2_voice *)hi->d.data : NULL;
	while(root && (!(voice = root->sub) || voice->s.state != A2_WAITING) && tries++ < 64)
		a2_Run(i, 64);
	if(!voice || !voice->program || voice->s.state != A2_WAITING || voice->sub || voice->events || voice->stack)
	{
		fprintf(stderr, "ref_vmtrace: no leaf voice waiting in a delay\n");
		return 5;
	}
	fn = &voice->program->funcs[voice->s.func];
	printf("{\"script\": \"%s\", \"program\": \"%s\", \"samplerate\": %d, \"basepitch\": %d, \"msdur\": %u, \"now\": %u,\n",
			strrchr(argv[1], '/') ? strrchr(argv[1], '/') + 1 : argv[1], argv[2], cfg->samplerate, cfg->basepitch,
			st->msdur, st->now_fragstart);
	printf(" \"args\": [");
	for(k = 0; k < nargs; ++k)
		printf("%s%d", k ? ", " : "", args[k]);
	printf("],\n \"code\": [");
	for(k = 0; k < fn->size; ++k)
		printf("%s%u", k ? ", " : "", fn->code[k]);
	printf("],\n \"state\": {\"waketime\": %u, \"state\": %d, \"func\": %d, \"pc\": %d, \"r\": [", voice->s.waketime,
			voice->s.state, voice->s.func, voice->s.pc);
	for(k = 0; k < A2_REGISTERS; ++k)
	{
		/* (registers the function never touches hold whatever the voice's block held before: left out) */
		int used = k <= fn->topreg || k < voice->ncregs || k < fn->argv + fn->argc;
		printf("%s%d", k ? ", " : "", used ? voice->s.r[k] : 0);
	}
	printf("]},\n \"units\": [");
	for(u = voice->units, k = 0; u; u = u->next, ++k)
	{
		printf("%s\"%s\"", k ? ", " : "", u->descriptor->name);
		if(!strcmp(u->descriptor->name, "env"))
			++nenvs;
		else if(!last)
			last = u;	/* (the first audio unit: where the voice's windows are logged) */
	}
	printf("],\n \"envs\": [");
	for(u = voice->units, k = 0, r = 0; u; u = u->next, ++k)
		if(!strcmp(u->descriptor->name, "env"))
		{
			/* which register of which unit is the control output wired to?  (a2_ControlWire, core.c:330-345,
			 * copied unit and callback from the voice's register write table) */
			int q, outreg = -1;
			for(q = 0; q < voice->ncregs; ++q)
				if(u->coutputs[0].write && voice->cregs[q].unit == u->coutputs[0].unit &&
						voice->cregs[q].write == u->coutputs[0].write)
					outreg = q;
			printf("%s{\"unit\": %d, \"regbase\": %d, \"out_vmreg\": %d}", r++ ? ", " : "", k,
					(int)(u->registers - voice->s.r), outreg);
		}
	printf("],\n \"cregs\": [");
	for(r = 0; r < A2_REGISTERS; ++r)
	{
		int pos = -1, reg = -1;
		if(r < voice->ncregs && voice->cregs[r].write)
			for(u = voice->units, k = 0; u; u = u->next, ++k)
				if(u == voice->cregs[r].unit)
				{
					pos = k;
					reg = (int)(&voice->s.r[r] - u->registers);
				}
		printf("%s[%d, %d]", r ? ", " : "", pos, reg);
	}
	printf("],\n");
	/* from here on the voice's writes and windows are logged */
	for(r = 0; r < voice->ncregs; ++r)
		if(voice->cregs[r].write)
		{
			orig_write[r] = voice->cregs[r].write;
			voice->cregs[r].write = tramp[r];
		}
	/* (an env's control output calls the wired unit's callback directly: through the same trampoline, then) */
	for(u = voice->units; u; u = u->next)
		if(!strcmp(u->descriptor->name, "env") && u->coutputs[0].write)
			for(r = 0; r < voice->ncregs; ++r)
				if(voice->cregs[r].unit == u->coutputs[0].unit && orig_write[r] == u->coutputs[0].write)
					u->coutputs[0].write = tramp[r];
	for(u = root->units; u; u = u->next)
		rootlast = u;
	for(cur_frag = 0; cur_frag < nfrags; ++cur_frag)
	{
		if(rootlast && rootlast->Process != hook_root)
		{
			orig_root_process = rootlast->Process;
			rootlast->Process = hook_root;
		}
		if(last && last->Process != hook_process)
		{
			orig_process = last->Process;
			last->Process = hook_process;
		}
		a2_Run(i, 64);
		if(root->sub != voice)
		{
			/* the voice ended (a2_VoiceFree, core.c:1892) in this fragment: the trace stops here, "gone" says
			 * where - the windows logged for this fragment end at the frame of its last VM run */
			gone = cur_frag;
			++cur_frag;
			break;
		}
	}
	if(gone >= 0)
		printf(" \"fragments\": %d, \"gone\": %d,\n \"events\": [", cur_frag, gone);
	else
		printf(" \"fragments\": %d, \"end_state\": {\"waketime\": %u, \"state\": %d, \"pc\": %d},\n \"events\": [", nfrags,
				voice->s.waketime, voice->s.state, voice->s.pc);
	for(k = 0; k < nev; ++k)
		if(ev[k].kind == 'w')
			printf("%s[\"w\", %d, %d, %d, %u, %u, %d]", k ? ", " : "", ev[k].frag, ev[k].a, ev[k].b, ev[k].c, ev[k].d, ev[k].e);
		else
			printf("%s[\"%c\", %d, %d, %d]", k ? ", " : "", ev[k].kind, ev[k].frag, ev[k].a, ev[k].b);
	printf("]}\n");
	/* (the trampolines stay in place until the voice dies with the state) */
	a2_Close(i);
	return nev < MAXEV ? 0 : 7;
}
