/*
 * a2amd_walk.c - INTEGRATION.md option C: Audiality 2's voice walk with a short cut for
 * sleeping voices.  Builds into liba2amd_walk.so, loaded in front of libaudiality2 together
 * with the drop-in units (LD_PRELOAD="liba2amd_walk.so liba2amd_units.so", or link order).
 *
 * What it replaces: a2_ProcessVoices (src/core.c:1883-1896), the loop over a list of sibling
 * voices that a2_AudioCallback (core.c:1968) and the inline unit (a2_ProcessSubvoices,
 * core.c:1749-1759) call once per window.  The engine's loop visits EVERY voice in EVERY
 * fragment: a2_VoiceProcess (core.c:1847-1880) asks the VM / event queue how long the voice
 * sleeps and calls Process on each of its units.  For a voice whose VM sleeps through the
 * window, whose event queue is empty and whose units are all the drop-in's, that visit has
 * exactly one effect - "this voice got the default window" - which the drop-in units record as
 * one byte store (a2amd_default_map, include/a2amd.h).  With tens of thousands of voices the
 * visit itself (the 1 424 byte A2_voice, its unit blocks: DRAM latency on a pointer chase) is
 * what one engine thread spends its time on (SURVEY.md 8e caveat 2: 1.7 ms per fragment for
 * 65 536 idle voices, 8 ms for 262 144).
 *
 * What it does instead: per list of siblings it remembers, position by position, which voice
 * stood there in the last walk and where its byte in the default map is.  A voice that still
 * stands there, has no events and does not wake before the window ends (the engine's own
 * test, a2_VoiceProcessVMEv, core.c:1816-1823, on the engine's own fields) gets the byte
 * store right here - one cache line of the A2_voice read, prefetched a few positions ahead
 * from the remembered pointers - and nothing else.  Every other voice is handed, alone, to the
 * ENGINE'S OWN a2_ProcessVoices (found with dlsym(RTLD_NEXT)): VM, events, unit calls, voice
 * death are the engine's code, not a copy of it.  The audio is bit-identical by construction:
 * the short cut stores the byte the unit's own Process would have stored.
 *
 * This is the wake-queue idea of SURVEY.md 8e (ii) / 8f-4 in the form that needs no hook on
 * the engine's event sends: the test reads A2_voice.events and A2_vmstate.waketime directly.
 *
 * Compiled against the engine's INTERNAL headers (src/internals.h: A2_voice, A2_state), which
 * is why it is a separate library with its own recipe (build.py: build_walk): it is
 * locked to the engine version it was built for, and the compiler checks every field it
 * touches.  The engine binary itself is unmodified; all it has to be is what a default -fPIC
 * build is - a library that calls its own a2_ProcessVoices through the PLT.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "internals.h"			/* the engine's: A2_state, A2_voice (src/internals.h:559-586, :600-700) */
#include "../../include/a2amd_walk.h"	/* a2amd_walkview */
#include "../../include/a2amd_vm.h"	/* the opcode numbers the device VM is built with */

#define AHEAD		40		/* positions the prefetch runs ahead (a DRAM access at ~3 ns per position) */
#define STAMP_NOUNITS	0xffffffffu	/* a voice without units and without subvoices: nothing to do while it sleeps */
#define E_NOEVENTS	1u		/* ENT.flags: its event queue was empty when it was last looked at */
#define E_APIHANDLE	2u		/* ... it has an API handle: the application can send it events at any time */
#define E_GROUP		4u		/* ... it has subvoices: it sleeps unseen only if their whole list does (ENT.sub) */
#define E_VM		8u		/* ... its program runs on the device (a2amd_units_vm_adopt): asleep whatever its
					 * (stale) wake time says, and handed to the engine only after a2amd_units_vm_recall */
#define E_VMEXIT	16u		/* ... for a stretch only: the engine gets it back in the fragment that holds ENT.vm_exit
					 * (a2amd_vm_exit_time, include/a2amd_vm.h: the VM run that ends the voice, sleeps, calls ...) */

/* the device VM interprets the engine's bytecode: the numbers it was built with are the engine's */
#define A2V(x) _Static_assert((int)OP_##x == (int)A2AMD_OP_##x, "opcode " #x);
A2AMD_VM_ALLOPS
#undef A2V
_Static_assert((int)A2_OPCODES == (int)A2AMD_VM_OPCODES, "number of opcodes");
_Static_assert(A2_REGISTERS == A2AMD_VM_REGISTERS && A2_INSLIMIT == A2AMD_VM_INSLIMIT, "VM limits");
_Static_assert(R_TICK == A2AMD_VM_R_TICK && R_TRANSPOSE == A2AMD_VM_R_TRANSPOSE, "fixed registers");
_Static_assert((int)A2_RUNNING == (int)A2AMD_VM_RUNNING && (int)A2_WAITING == (int)A2AMD_VM_WAITING &&
		(int)A2_INTERRUPT == (int)A2AMD_VM_INTERRUPT && (int)A2_ENDING == (int)A2AMD_VM_ENDING &&
		(int)A2_FINALIZING == (int)A2AMD_VM_FINALIZING, "VM states");
_Static_assert(sizeof(A2_vmstate) == sizeof(a2amd_vm_state) && offsetof(A2_vmstate, r) == offsetof(a2amd_vm_state, r) &&
		offsetof(A2_vmstate, pc) == offsetof(a2amd_vm_state, pc) && offsetof(A2_vmstate, state) == offsetof(a2amd_vm_state, state),
		"A2_vmstate");
_Static_assert(sizeof(A2_instruction) == 8, "A2_instruction");

typedef struct ENT
{
	A2_voice	*v;		/* the voice that stood at this position of the list in the last walk */
	uint32_t	slotdev;	/* its byte in the default map: slot (bits 0..27), context (28..31) */
	uint32_t	stamp;		/* a2amd_units_standing() when it was last visited; 0 = always visit */
	uint32_t	wake;		/* A2_vmstate.waketime when it was last looked at */
	uint32_t	flags;		/* E_* */
	uint32_t	vm_exit;	/* E_VMEXIT: engine time of the first VM run that is the engine's again */
	struct LIST	*sub;		/* E_GROUP: the list of its subvoices */
} ENT;

/* ... and what the engine will touch when it visits that voice: its unit blocks (prefetch hints
 * for the voices that are handed to the engine's loop, never dereferenced here) */
typedef struct UPTR
{
	A2_unit		*u[3];
} UPTR;

#define BLKN 256
#define SUM_STREAK 8
#define HOLD_MIN 32	/* fragments */
typedef struct BLK
{
	uint32_t	wake;		/* the earliest wake time in the block (timed) */
	uint32_t	lo, hi;		/* its voices' bytes in the map of context dev: within [lo, hi]; all of that range if
					 * hi - lo == BLKN - 1, else the bytes LIST.blk_slots names */
	uint8_t		ok, timed, dev;
} BLK;

typedef struct LIST
{
	A2_voice	**head;		/* the list: &parent->sub (key) */
	struct LIST	*above;		/* the list its parent voice stands in (NULL: the root's), as of the last walk */
	ENT		*e;
	UPTR		*up;		/* (parallel to e) */
	unsigned	n, cap;
	unsigned long long epoch;	/* e[0..n) was the whole list, in order, when the state's epoch had this value */
	unsigned long long quiet_visit;	/* the parent was found asleep, without events, for the window of this visit */
	/* the whole list in one test: every voice of it sleeps, reports through the map (or has nothing
	 * to report), and nothing about that can have changed while the epoch holds and no voice of
	 * the list has been visited */
	int		sum_ok;
	unsigned long long sum_epoch;
	uint32_t	sum_wake;	/* the earliest wake time in the list - and in the lists below it */
	int		sum_timed;	/* ... if any voice there has one (voices the device VM runs do not) */
	unsigned	*gidx, ng;	/* the entries with subvoices (E_GROUP), as of the summary */
	unsigned long long sum_voices;	/* voices in the list and below */
	unsigned long long held_gen;	/* == WSTATE.hold_gen: the list's voices are held (a2amd_units_hold) */
	unsigned	reached;	/* the fragment (a2amd_walkview.frag_serial) in which the walk last passed it asleep */
	uint32_t	sum_dev, sum_lo, sum_cnt;	/* their bytes in the default map: one range of one context ... */
	int		sum_range;	/* ... or not (then entry by entry) */
	/* A list that never sleeps as a whole - a pad's 16 384 voices and ONE sequencer voice in the same group -
	 * in blocks of BLKN entries: a block whose voices all slept unseen, none of them a group, their bytes one
	 * range of one context's map, is one test and one memset the next time (while the list is taken from memory
	 * and none of the block's voices has been handed to the engine). */
	struct BLK	*blk;
	unsigned	blk_cap;
	uint32_t	*blk_slots;	/* [entry]: its byte in the map, for the blocks (blk_cap * BLKN of them) */
	unsigned	streak;		/* walks in a row in which every voice of the list slept unseen */
} LIST;

typedef struct WSTATE
{
	A2_state	*st;
	a2amd_walkview	view;
	int		served;		/* the drop-in serves this state */
	LIST		**lists;	/* open addressing on 'head'; the LISTs stay where they are (nested calls hold them) */
	unsigned	nlists, cap_lists;
	unsigned long long epoch;	/* what a remembered list's epoch must equal to count: a birth or a death makes the ONE list
					 * it happened in stale (list_changed) - the others keep what they know; only a new engine
					 * state, or a change that cannot be placed, moves this */
	struct LIST	*cur_list;	/* the list the innermost a2_ProcessVoices in progress is walking (deaths happen there) */
	unsigned long long visits, cur_visit;	/* the engine call in progress one level up (quiet_visit) */
	int		hooks_broken;	/* a list changed without the epoch moving: never trust remembered lists */
	unsigned long long hold_gen;	/* holds made under an older value have been released wholesale */
	unsigned	n_held;		/* lists held */
	uint32_t	*scratch;	/* slots of a list, per context */
	unsigned	scratch_cap;
	unsigned long long skipped, unread, visited, in_blocks, summaries, holds, slow_walks, blocks_made;
	/* (code, pc) pairs the device VM has turned down for good: not offered again.  (The text's size and a sum of its
	 * words go with the address: a program compiled later may stand where a released one stood.) */
	struct { const unsigned *code; unsigned pc, size, sum; } vm_no[64];
	unsigned	vm_no_pos;
	unsigned long long adopted, recalled;
} WSTATE;

static void (*engine_walk)(A2_state *st, A2_voice **head, unsigned offset, unsigned frames);
static A2_voice *(*engine_voicenew)(A2_state *st, A2_voice *parent, unsigned when);
static void (*engine_voicefree)(A2_state *st, A2_voice **head);
/* One record per engine state that has been walked: a table that grows (engine states - a2_Render's
 * substates among them - come and go and nobody tells us).  A record is given to another state only
 * when the one it was made for is provably closed (the drop-in's serial for it has moved on: no
 * thread can be inside that state's calls any more); records of states the drop-in does not serve
 * hold nothing (their lists are only ever filled under the A2AMD_WALK_CUT test hook). */
static WSTATE **wstates;
static unsigned n_wstates, cap_wstates;
static pthread_mutex_t wmtx = PTHREAD_MUTEX_INITIALIZER;
static __thread WSTATE *last_ws;
static int walk_off = -1, walk_stats, walk_cut, walk_nocache, walk_nohold, walk_novm, walk_globalepoch, walk_noblocks, walk_nohold_min;

static inline int wstate_closed(const WSTATE *w)
{
	return w->served && *w->view.serial != w->view.serial_value;
}

static WSTATE *wstate_of(A2_state *st)
{
	WSTATE *w = last_ws;
	unsigned i;
	if(w && w->st == st && w->view.cfg == st->config && (!w->served || *w->view.serial == w->view.serial_value))
		return w;
	pthread_mutex_lock(&wmtx);
	{
		WSTATE *dead = NULL;
		w = NULL;
		for(i = 0; i < n_wstates; ++i)
			if(wstates[i]->st == st)
			{
				w = wstates[i];
				break;
			}
			else if(!dead && wstate_closed(wstates[i]))
				dead = wstates[i];
		if(!w && dead)
		{
			/* (a closed state's record: what it remembers is dropped below, like that of a new
			 * state at the address of a closed one) */
			w = dead;
			w->st = st;
			w->view.cfg = NULL;
			w->served = 0;
		}
		if(!w && n_wstates < (1u << 20))
		{
			if(n_wstates == cap_wstates)
			{
				unsigned nc = cap_wstates ? cap_wstates * 2 : 64;
				WSTATE **nw = (WSTATE **)realloc(wstates, nc * sizeof(WSTATE *));
				if(nw)
				{
					wstates = nw;
					cap_wstates = nc;
				}
			}
			if(n_wstates < cap_wstates && (w = (WSTATE *)calloc(1, sizeof(WSTATE))))
			{
				w->st = st;
				wstates[n_wstates++] = w;
			}
		}
		/* (no record to be had: the caller hands the whole list to the engine's own loop) */
	}
	pthread_mutex_unlock(&wmtx);
	if(w && (w->view.cfg != st->config || (w->served && *w->view.serial != w->view.serial_value)))
	{
		/* a new engine state (at the address of a closed one, possibly): forget everything */
		unsigned k;
		for(k = 0; k < w->cap_lists; ++k)
			if(w->lists[k])
			{
				free(w->lists[k]->e);
				free(w->lists[k]->up);
				free(w->lists[k]->gidx);
				free(w->lists[k]->blk);
				free(w->lists[k]->blk_slots);
				free(w->lists[k]);
			}
		free(w->lists);
		w->lists = NULL;
		w->nlists = w->cap_lists = 0;
		++w->epoch;
		++w->hold_gen;		/* (the closed state's contexts, and their holds, are gone) */
		w->n_held = 0;
		memset(w->vm_no, 0, sizeof(w->vm_no));
		w->served = a2amd_units_walkview(st->config, &w->view) == 0;
		if(!w->served)
			w->view.cfg = st->config;
		else
			*w->view.walker = 1;
	}
	last_ws = w;
	return w;
}

static LIST *list_of(WSTATE *w, A2_voice **head)
{
	unsigned k, mask;
	if(w->nlists * 2 >= w->cap_lists)
	{
		unsigned nc = w->cap_lists ? w->cap_lists * 2 : 64, j;
		LIST **nl = (LIST **)calloc(nc, sizeof(LIST *));
		if(!nl)
			return NULL;
		for(j = 0; j < w->cap_lists; ++j)
			if(w->lists[j])
			{
				k = (unsigned)(((uintptr_t)w->lists[j]->head >> 4) * 2654435761u) & (nc - 1);
				while(nl[k])
					k = (k + 1) & (nc - 1);
				nl[k] = w->lists[j];
			}
		free(w->lists);
		w->lists = nl;
		w->cap_lists = nc;
	}
	mask = w->cap_lists - 1;
	k = (unsigned)(((uintptr_t)head >> 4) * 2654435761u) & mask;
	while(w->lists[k] && w->lists[k]->head != head)
		k = (k + 1) & mask;
	if(!w->lists[k])
	{
		if(!(w->lists[k] = (LIST *)calloc(1, sizeof(LIST))))
			return NULL;
		w->lists[k]->head = head;
		++w->nlists;
	}
	return w->lists[k];
}

static void report(void)
{
	unsigned i;
	for(i = 0; i < n_wstates; ++i)
		if(wstates[i]->skipped || wstates[i]->visited)
			fprintf(stderr, "a2amd walk: state %p: %llu voice visits skipped, %llu made (%llu of the skipped without "
					"reading the voice%s); %llu voices handed to the device VM, %llu taken back\n", (void *)wstates[i]->st,
					wstates[i]->skipped, wstates[i]->visited, wstates[i]->unread,
					wstates[i]->hooks_broken ? "; HOOKS BROKEN" : "", wstates[i]->adopted, wstates[i]->recalled);
	for(i = 0; i < n_wstates; ++i)
		if(wstates[i]->in_blocks || wstates[i]->summaries)
			fprintf(stderr, "a2amd walk: state %p: %llu of the skipped in blocks of %d, %llu list summaries made, %llu lists held, "
					"%llu walks voice by voice, %llu block summaries made\n", (void *)wstates[i]->st, wstates[i]->in_blocks, BLKN, wstates[i]->summaries,
					wstates[i]->holds, wstates[i]->slow_walks, wstates[i]->blocks_made);
}

static void bind_engine(void)
{
	*(void **)&engine_walk = dlsym(RTLD_NEXT, "a2_ProcessVoices");
	*(void **)&engine_voicenew = dlsym(RTLD_NEXT, "a2_VoiceNew");
	*(void **)&engine_voicefree = dlsym(RTLD_NEXT, "a2_VoiceFree");
	{
		/* This file reads and relinks the engine's OWN structures (A2_voice, A2_state) as the headers
		 * it was compiled with lay them out: in front of another engine version it would corrupt
		 * them.  The engine says which one it is. */
		unsigned (*linked)(void) = NULL;
		*(void **)&linked = dlsym(RTLD_NEXT, "a2_LinkedVersion");
		if(!linked || linked() != (unsigned)A2_VERSION)
		{
			fprintf(stderr, "a2amd walk: built for Audiality 2 %d.%d.%d.%d, the engine behind it is %s: every voice list "
					"goes to the engine's own loop\n", A2_MAJOR(A2_VERSION), A2_MINOR(A2_VERSION), A2_MICRO(A2_VERSION),
					A2_BUILD(A2_VERSION), linked ? "another version" : "not there");
			walk_off = 1;
		}
	}
	if(!engine_walk || !engine_voicenew || !engine_voicefree)
	{
		fprintf(stderr, "a2amd walk: no a2_ProcessVoices / a2_VoiceNew / a2_VoiceFree behind these - load "
				"liba2amd_walk.so IN FRONT of libaudiality2\n");
		engine_walk = NULL;
	}
	if(walk_off < 1)
		walk_off = getenv("A2AMD_WALK_OFF") != NULL;	/* A/B: every voice is handed to the engine's loop */
	/* A/B: lists are never trusted from memory - every sleeping voice's A2_voice is read */
	walk_nocache = getenv("A2AMD_WALK_NOCACHE") != NULL;
	walk_globalepoch = getenv("A2AMD_WALK_GLOBALEPOCH") != NULL;	/* (A/B: rounds 3-4's one epoch per state) */
	/* A/B: sleeping lists are marked fragment by fragment instead of being put on hold */
	walk_nohold = getenv("A2AMD_WALK_NOHOLD") != NULL;
	walk_noblocks = getenv("A2AMD_WALK_NOBLOCKS") != NULL;
	walk_nohold_min = walk_noblocks;	/* (A/B: every voice of a list that does not sleep as a whole, one by one) */
	/* test hook: voices are handed to the engine's loop run by run even in a state the drop-in
	 * does not serve (the engine's own CPU units): exercises the cut / relink / voice death
	 * logic without a GPU; no visit is ever skipped there */
	walk_cut = getenv("A2AMD_WALK_CUT") != NULL;
	/* A/B: no voice is offered to the device VM */
	walk_novm = getenv("A2AMD_NO_VM") != NULL;
	if((walk_stats = getenv("A2AMD_WALK_STATS") != NULL))
		atexit(report);
}

/* a voice is about to be made or freed: what is remembered about the state's lists is void, and
 * no voice stays on hold (its slot may be somebody else's a moment later) */
static LIST *list_lookup(const WSTATE *w, A2_voice **head)
{
	unsigned k, mask;
	if(!w->cap_lists)
		return NULL;
	mask = w->cap_lists - 1;
	k = (unsigned)(((uintptr_t)head >> 4) * 2654435761u) & mask;
	while(w->lists[k] && w->lists[k]->head != head)
		k = (k + 1) & mask;
	return w->lists[k];
}

static int hold_list(WSTATE *w, LIST *sl, int on);

/* A voice was born into, or died in, THIS list (round 5: rounds 3-4 moved the state's one epoch, which
 * voided every remembered list and every summary of the state - a note played anywhere made the next
 * walk read all 16 384 sleeping voices of a pad again).  The list itself is stale until it has been
 * walked link by link; the summaries above it - which count its voices and its earliest wake time -
 * are void; its hold (its voices' default windows, a2amd_units_hold) goes, because a slot of it may be
 * somebody else's a moment later.  Every other list keeps what it knows. */
static void list_changed(WSTATE *w, LIST *L)
{
	LIST *a;
	unsigned depth = 0, k;
	L->epoch = w->epoch - 1;
	for(a = L; a; a = a->above)
	{
		a->sum_ok = 0;
		if(++depth > 64)
		{
			/* (a stale 'above' chain - LISTs are reused with the addresses of their heads: give up on placing it) */
			++w->epoch;
			break;
		}
	}
	if(L->held_gen == w->hold_gen && w->hold_gen)
	{
		if(w->served && L->reached == *w->view.frag_serial)
			/* the walk has passed it in this fragment: its voices HAVE had their default window in it -
			 * stored byte by byte before the hold goes (as structure_changes does for all) */
			for(k = 0; k < L->n; ++k)
			{
				const ENT *e = &L->e[k];
				if(e->stamp != STAMP_NOUNITS && (e->slotdev & 0x0fffffffu) < w->view.map_cap[e->slotdev >> 28])
					w->view.map[e->slotdev >> 28][e->slotdev & 0x0fffffffu] = 1;
			}
		hold_list(w, L, 0);
	}
}

static void structure_changes(WSTATE *w)
{
	++w->epoch;
	if(w->n_held)
	{
		if(w->served)
		{
			/* In the middle of a walk (a voice's program ended, a note was spawned): the held
			 * lists the walk has already passed in this fragment HAVE had their default window in
			 * it - that is stored, byte by byte, before the holds go; the lists it has not
			 * reached yet get whatever the rest of the walk finds for them (a list whose parent
			 * has just died gets nothing, as in the engine: core.c:1856-1868, :1892). */
			const unsigned cur = *w->view.frag_serial;
			unsigned t, k;
			for(t = 0; t < w->cap_lists; ++t)
			{
				const LIST *sl = w->lists[t];
				if(!sl || sl->held_gen != w->hold_gen || sl->reached != cur)
					continue;
				for(k = 0; k < sl->n; ++k)
				{
					const ENT *e = &sl->e[k];
					if(e->stamp != STAMP_NOUNITS && (e->slotdev & 0x0fffffffu) < w->view.map_cap[e->slotdev >> 28])
						w->view.map[e->slotdev >> 28][e->slotdev & 0x0fffffffu] = 1;
				}
			}
			a2amd_units_hold(w->view.state, 0, NULL, 0, 0);
		}
		++w->hold_gen;
		w->n_held = 0;
	}
}

/* The two places where the engine's voice lists change (src/core.c:456-482, :532-581), interposed
 * like the walk itself: what is remembered about the lists of a state holds while its epoch does. */
A2_voice *a2_VoiceNew(A2_state *st, A2_voice *parent, unsigned when)
{
	WSTATE *w;
	if(walk_off < 0)
		bind_engine();
	if(!engine_walk)
		return NULL;
	if((w = wstate_of(st)))
	{
		/* born at the head of its parent's list (core.c:474-475) */
		LIST *L = (parent && !walk_globalepoch) ? list_lookup(w, &parent->sub) : NULL;
		if(L)
			list_changed(w, L);
		else if(!parent || walk_globalepoch || w->cap_lists == 0)
			structure_changes(w);
		/* (a parent whose list was never walked: nothing is remembered about it) */
	}
	return engine_voicenew(st, parent, when);
}

void a2_VoiceFree(A2_state *st, A2_voice **head)
{
	WSTATE *w;
	if(walk_off < 0)
		bind_engine();
	if(!engine_walk)
		return;
	if((w = wstate_of(st)))
	{
		/* which list?  head is the link that points at the voice: the list's own head (a parent that
		 * ends frees its subvoices through &v->sub, core.c:540-560) or, from the engine's loop
		 * (core.c:1892), a link inside the list that loop was given - the one our walk is on */
		LIST *L = walk_globalepoch ? NULL : list_lookup(w, head);
		if(!L && !walk_globalepoch)
			L = w->cur_list;
		if(L)
		{
			/* (the voice's own subvoices go with it, core.c:548-551: through this hook, list by list) */
			list_changed(w, L);
		}
		else
			structure_changes(w);
	}
	engine_voicefree(st, head);
}

/* Asleep for the whole window, and nothing to do for it but the byte store?  a2_VoiceProcessVMEv
 * (core.c:1816-1823) with an empty event queue returns (waketime - now) >> 8 frames;
 * a2_VoiceProcess then makes ONE Process call per unit for the window if that is at least its
 * length.  'wake' / 'noevents': the voice's own fields, or what the entry remembers of them. */
/* A whole list asleep: its summary holds (nothing was born or died, no voice of it was visited
 * since), its earliest wake time - the lists below included - lies beyond the window, and the same
 * goes for the subvoice lists of its group voices, whose own standing with the units is checked
 * too.  O(lists), not O(voices). */
static int list_sleeps(const WSTATE *w, const LIST *sl, unsigned now, unsigned frames)
{
	const a2amd_walkview *vw = &w->view;
	unsigned g;
	if(!sl || !sl->n || !sl->sum_ok || sl->sum_epoch != w->epoch || sl->epoch != w->epoch ||
			(sl->sum_timed && (a2_TSDiff(sl->sum_wake, now) >> 8) < (int)frames) ||
			(sl->sum_cnt && sl->sum_lo + sl->sum_cnt > vw->map_cap[sl->sum_dev]))
		return 0;
	for(g = 0; g < sl->ng; ++g)
	{
		const ENT *e = &sl->e[sl->gidx[g]];
		if(e->stamp != STAMP_NOUNITS)
		{
			const unsigned dev = e->slotdev >> 28, slot = e->slotdev & 0x0fffffffu;
			if(!(slot < vw->map_cap[dev] && slot < vw->qstamp_cap[dev] && vw->qstamp[dev][slot] == e->stamp))
				return 0;
		}
		if(!list_sleeps(w, e->sub, now, frames))
			return 0;
	}
	return 1;
}

static inline int entry_sleeps(const WSTATE *w, const ENT *e, unsigned wake, int noevents, unsigned now,
		unsigned frames, int deflt)
{
	const a2amd_walkview *vw = &w->view;
	unsigned dev, slot;
	if(!e->stamp || !noevents || (!(e->flags & E_VM) && (a2_TSDiff(wake, now) >> 8) < (int)frames) ||
			((e->flags & E_VMEXIT) && (a2_TSDiff(e->vm_exit, now) >> 8) < (int)frames))
		return 0;
	if(e->stamp != STAMP_NOUNITS)
	{
		/* only the open root window - one backend fragment - can be reported through the map
		 * (amd_quick_process, a2amd_units.c, makes the same test) */
		if(!deflt)
			return 0;
		dev = e->slotdev >> 28;
		slot = e->slotdev & 0x0fffffffu;
		if(!(slot < vw->map_cap[dev] && slot < vw->qstamp_cap[dev] && vw->qstamp[dev][slot] == e->stamp))
			return 0;
	}
	/* a voice with subvoices (a group): they would be walked for this window (core.c:1749-1759,
	 * :1888-1889) - unless their whole list sleeps, and then they get this very window */
	if(e->flags & E_GROUP)
		return deflt && list_sleeps(w, e->sub, now, frames);
	return 1;
}

/* ... looking at the voice itself (one cache line of it) */
static inline int voice_sleeps(const WSTATE *w, LIST *l, unsigned k, const A2_voice *v, unsigned now,
		unsigned frames, int deflt)
{
	ENT *e;
	if(k >= l->n || !w->served)
		return 0;
	e = &l->e[k];
	if(e->v != v)
		return 0;
	e->wake = v->s.waketime;
	e->flags = (e->flags & ~E_NOEVENTS) | (v->events ? 0 : E_NOEVENTS);
	return entry_sleeps(w, e, e->wake, e->flags & E_NOEVENTS, now, frames, deflt);
}

/* ... from memory: the list has not changed since it was last walked, its parent slept through
 * this window without events - so nobody can have sent this voice an event or moved its wake
 * time since they were last read (only a voice's own VM, its parent's VM and event handling,
 * and - for a voice with an API handle - the application ever do) */
static inline int entry_sleeps_unread(const WSTATE *w, const ENT *e, unsigned now, unsigned frames, int deflt)
{
	return (e->flags & (E_NOEVENTS | E_APIHANDLE)) == E_NOEVENTS &&
			entry_sleeps(w, e, e->wake, 1, now, frames, deflt);
}

/* the engine is about to visit this voice: the lines it will touch - the A2_voice's first line
 * (events, wake time), the one with its flags, the one with 'units' and 'sub', and its unit
 * blocks (A2_unit: next first, Process last - 64 bytes that usually straddle two lines; the
 * drop-in's own data behind them) - are asked for now, a run of voices ahead of the calls */
static inline void hint_visit(const A2_voice *v, const UPTR *up)
{
	int q;
	__builtin_prefetch(v);
	__builtin_prefetch(&v->flags);
	__builtin_prefetch(&v->units);
	for(q = 0; q < 3 && up->u[q]; ++q)
	{
		__builtin_prefetch(up->u[q]);
		__builtin_prefetch((const char *)up->u[q] + 56);
		if(!q)
			__builtin_prefetch((const char *)up->u[q] + 128);
	}
}

static inline void mark_default(const WSTATE *w, const ENT *e)
{
	/* (a backend that has failed hands out no map: map_cap 0, and the state renders silence) */
	if(e->stamp != STAMP_NOUNITS && (e->slotdev & 0x0fffffffu) < w->view.map_cap[e->slotdev >> 28])
		w->view.map[e->slotdev >> 28][e->slotdev & 0x0fffffffu] = 1;	/* = amd_quick_process() */
}

/* The voices of a sleeping list (list_sleeps) are put ON HOLD with the backend - "default window
 * in every fragment until further notice" (a2amd_default_hold) - instead of being marked fragment
 * by fragment: a list asleep costs nothing at all until its earliest wake time comes, a voice is
 * born or dies, or its parent wakes.  on = 0 releases them (before any of them is handed to the
 * engine's loop). */
static int hold_list(WSTATE *w, LIST *sl, int on)
{
	unsigned k, dev, n;
	if(on && walk_nohold)
		return -1;
	if(sl->n > w->scratch_cap)
	{
		uint32_t *ns = (uint32_t *)realloc(w->scratch, sl->n * 2 * sizeof(uint32_t));
		if(!ns)
			return -1;
		w->scratch = ns;
		w->scratch_cap = sl->n * 2;
	}
	for(dev = 0; dev < A2AMD_WALK_MAXDEV; ++dev)
	{
		for(n = k = 0; k < sl->n; ++k)
			if(sl->e[k].stamp != STAMP_NOUNITS && (sl->e[k].slotdev >> 28) == dev)
				w->scratch[n++] = sl->e[k].slotdev & 0x0fffffffu;
		if(n && a2amd_units_hold(w->view.state, dev, w->scratch, n, on) && on)
		{
			hold_list(w, sl, 0);
			return -1;
		}
	}
	if(on)
	{
		sl->held_gen = w->hold_gen;
		++w->n_held;
		++w->holds;
	}
	else if(sl->held_gen == w->hold_gen)
	{
		sl->held_gen = 0;
		--w->n_held;
	}
	return 0;
}

/* the default window for every voice of a sleeping list (list_sleeps) and of the lists below it */
static void mark_list(WSTATE *w, LIST *sl, unsigned now)
{
	unsigned k, g;
	sl->reached = *w->view.frag_serial;
	/* (a hold is made and released voice by voice: for a list that will wake within HOLD_MIN fragments - a pad
	 * with a sequencer voice in its group - the map's bytes, one memset per fragment, are the cheaper way) */
	if(sl->held_gen != w->hold_gen && ((sl->sum_timed && sl->sum_range && !walk_nohold_min &&
			(a2_TSDiff(sl->sum_wake, now) >> 8) < HOLD_MIN * A2_MAXFRAG) || hold_list(w, sl, 1)))
	{
		/* (no hold to be had: fragment by fragment, then) */
		if(sl->sum_range)
		{
			if(sl->sum_cnt)
				memset(w->view.map[sl->sum_dev] + sl->sum_lo, 1, sl->sum_cnt);
		}
		else
			for(k = 0; k < sl->n; ++k)
				if(!(sl->e[k].flags & E_GROUP))
					mark_default(w, &sl->e[k]);
		for(g = 0; g < sl->ng; ++g)
			mark_default(w, &sl->e[sl->gidx[g]]);
	}
	for(g = 0; g < sl->ng; ++g)
		mark_list(w, sl->e[sl->gidx[g]].sub, now);
}

/* ... for one voice that sleeps unseen (entry_sleeps), its subvoices included */
static inline void mark_entry(WSTATE *w, const ENT *e, unsigned now)
{
	mark_default(w, e);
	if(e->flags & E_GROUP)
	{
		mark_list(w, e->sub, now);
		w->skipped += e->sub->sum_voices;
		w->unread += e->sub->sum_voices;
	}
}


/* The list as remembered against the list as it is, after voices were born or died (the epoch
 * moved): voices are born at the HEAD of their parent's list (a2_VoiceNew, core.c:474-475) and die
 * anywhere.  Matching position by position, one birth would make every voice behind it a stranger
 * - handed to the engine's loop although it sleeps, a voice the device VM runs taken back for
 * nothing.  At a mismatch at position k: did the voices remembered at k, k + 1, ... die (the voice
 * standing here is remembered a few places on), or were voices born in front of the one remembered
 * here (it stands a few links on)?  The table is shifted accordingly; new places start blank. */
static void realign(LIST *l, unsigned k, A2_voice *v)
{
	unsigned d, j;
	A2_voice *p;
	for(d = 1; d <= 16 && k + d < l->n; ++d)
		if(l->e[k + d].v == v)
		{
			memmove(&l->e[k], &l->e[k + d], (l->n - k - d) * sizeof(ENT));
			memmove(&l->up[k], &l->up[k + d], (l->n - k - d) * sizeof(UPTR));
			l->n -= d;
			return;
		}
	for(p = v, j = 1; j <= 4096 && p->next; ++j)
	{
		p = p->next;
		if(p != l->e[k].v)
			continue;
		if(l->n + j > l->cap)
		{
			unsigned nc = (l->n + j) * 2;
			ENT *ne = (ENT *)realloc(l->e, nc * sizeof(ENT));
			UPTR *nu = ne ? (UPTR *)realloc(l->up, nc * sizeof(UPTR)) : NULL;
			if(ne)
				l->e = ne;
			if(nu)
				l->up = nu;
			if(!ne || !nu)
				return;
			l->cap = nc;
		}
		memmove(&l->e[k + j], &l->e[k], (l->n - k) * sizeof(ENT));
		memmove(&l->up[k + j], &l->up[k], (l->n - k) * sizeof(UPTR));
		memset(&l->e[k], 0, j * sizeof(ENT));
		memset(&l->up[k], 0, j * sizeof(UPTR));
		for(p = v, d = 0; d < j; p = p->next, ++d)
			l->e[k + d].v = p;	/* (stamp 0: visited once before anything is taken from memory) */
		l->n += j;
		return;
	}
}

/* ---- voices whose VM runs on the device (include/a2amd_vm.h) ------------------------
 * The engine is about to process the 'run' voices from 'v' on (linked by ->next): those the device
 * VM runs get their A2_vmstate back first - what the engine's own VM would have left there by the
 * start of this fragment (a2amd_units_vm_recall) - and are the engine's again. */
static void recall_run(WSTATE *w, LIST *l, unsigned k, A2_voice *v, unsigned run)
{
	const void *heads[64];
	void *states[64];
	unsigned n = 0, kk;
	A2_voice *p;
	if(!w->served || !*w->view.vm_live)
		return;
	for(p = v, kk = 0; kk < run && p; p = p->next, ++kk)
	{
		int is_vm;
		if(k + kk < l->n && l->e[k + kk].v == p && !(l->e[k + kk].flags & E_VM))
			is_vm = 0;	/* (the common case: one remembered flag, the voice is not read) */
		else
			/* a remembered E_VM is a hint only - the entry may be older than the voice at this
			 * address (a freed subtree, a reused A2_voice) - so the units are asked */
			is_vm = p->units && a2amd_units_vm_is(p->units);
		if(!is_vm)
		{
			if(k + kk < l->n && l->e[k + kk].v == p)
				l->e[k + kk].flags &= ~(E_VM | E_VMEXIT);
			continue;
		}
		if(k + kk < l->n && l->e[k + kk].v == p)
			l->e[k + kk].flags &= ~(E_VM | E_VMEXIT);
		heads[n] = p->units;
		states[n++] = &p->s;
		if(n == 64)
		{
			a2amd_units_vm_recall(heads, n, states);
			w->recalled += n;
			n = 0;
		}
	}
	if(n)
	{
		a2amd_units_vm_recall(heads, n, states);
		w->recalled += n;
	}
}

/* The engine has just processed 'p' in the open root window: can the device run its program from
 * here on?  A leaf voice waiting in a delay, no events, no call stack, nobody outside the tree who
 * could send it anything (no API handle) - and a program that a2amd_vm_analyze() can prove to stay
 * inside the subset from p->s.pc on.  Returns 1 when the voice was handed over. */
static unsigned text_sum(const unsigned *code, unsigned size)
{
	unsigned k, h = 2166136261u;
	for(k = 0; k < size; ++k)
		h = (h ^ code[k]) * 16777619u;
	return h;
}

static int offer_to_vm(WSTATE *w, A2_state *st, A2_voice *p, int *has_exit, uint32_t *exit_when)
{
	void *wr_unit[A2_REGISTERS], *wr_fn[A2_REGISTERS];
	const A2_function *fn;
	unsigned r;
	int rc;
	*has_exit = 0;
	if(walk_novm || !w->served || p->sub || p->events || p->stack || (p->flags & A2_APIHANDLE) || !p->units ||
			!p->program || p->s.state != A2_WAITING)
		return 0;
	fn = &p->program->funcs[p->s.func];
	for(r = 0; r < 64; ++r)
		if(w->vm_no[r].code == fn->code && w->vm_no[r].pc == p->s.pc && w->vm_no[r].size == fn->size)
		{
			if(w->vm_no[r].sum == text_sum(fn->code, fn->size))
				return 0;
			w->vm_no[r].code = NULL;	/* (another program at that address) */
		}
	for(r = 0; r < A2_REGISTERS; ++r)
	{
		wr_unit[r] = r < p->ncregs ? (void *)p->cregs[r].unit : NULL;
		wr_fn[r] = r < p->ncregs ? (void *)p->cregs[r].write : NULL;
	}
	rc = a2amd_units_vm_adopt(p->units, fn->code, fn->size, &p->s, wr_unit, wr_fn,
			st->now_fragstart + (*w->view.base << 8), st->msdur, has_exit, exit_when);
	if(rc == -1)
	{
		w->vm_no[w->vm_no_pos & 63].code = fn->code;
		w->vm_no[w->vm_no_pos & 63].size = fn->size;
		w->vm_no[w->vm_no_pos & 63].sum = text_sum(fn->code, fn->size);
		w->vm_no[w->vm_no_pos++ & 63].pc = p->s.pc;
	}
	if(!rc)
		++w->adopted;
	return !rc;
}

/* The replacement.  Same contract as the engine's (internals.h:968-973). */
void a2_ProcessVoices(A2_state *st, A2_voice **head, unsigned offset, unsigned frames)
{
	WSTATE *w;
	LIST *l;
	unsigned k = 0, now;
	int deflt, cached, all_unread;
	unsigned acc_at = 0;		/* the block summary under way (BLK) */
	int acc_ok = 0, acc_timed = 0, acc_best = 0;
	uint32_t acc_lo = 0, acc_hi = 0, acc_dev = 0, acc_wake = 0;
	if(walk_off < 0)
		bind_engine();
	if(!engine_walk)
		return;		/* (reported once; the state renders silence) */
	if(walk_off || !(w = wstate_of(st)) || !(w->served || walk_cut) || !(l = list_of(w, head)))
	{
		engine_walk(st, head, offset, frames);
		return;
	}
	deflt = w->served && offset == *w->view.base && frames == *w->view.win_frames;
	now = st->now_fragstart + (offset << 8);	/* a2_VoiceProcess, core.c:1856 */
	/* may the list be taken from memory?  (see entry_sleeps_unread) */
	cached = w->served && !walk_nocache && !w->hooks_broken && l->n && l->epoch == w->epoch &&
			l->quiet_visit && l->quiet_visit == w->cur_visit;
	l->quiet_visit = 0;
	if(cached && deflt && list_sleeps(w, l, now, frames))
	{
		mark_list(w, l, now);
		w->skipped += l->sum_voices;
		w->unread += l->sum_voices;
		return;
	}
	if(!cached)
		++w->slow_walks;
	/* voice by voice, then: none of them stays on hold */
	if(l->held_gen == w->hold_gen && w->hold_gen)
		hold_list(w, l, 0);
	all_unread = cached && deflt;
	/* the blocks' summaries hold while the list is taken from memory */
	if(!(cached && deflt) && l->blk)
		memset(l->blk, 0, l->blk_cap * sizeof(BLK));
	if(cached && deflt && !walk_noblocks && l->n >= 2 * BLKN && l->blk_cap < l->n / BLKN)
	{
		BLK *nb = (BLK *)realloc(l->blk, (l->n / BLKN + 16) * sizeof(BLK));
		uint32_t *ns = nb ? (uint32_t *)realloc(l->blk_slots, (size_t)(l->n / BLKN + 16) * BLKN * sizeof(uint32_t)) : NULL;
		if(nb)
			l->blk = nb;
		if(ns)
			l->blk_slots = ns;
		if(nb && ns)
		{
			memset(nb + l->blk_cap, 0, (l->n / BLKN + 16 - l->blk_cap) * sizeof(BLK));
			l->blk = nb;
			l->blk_cap = l->n / BLKN + 16;
		}
	}
	for(;;)
	{
		A2_voice *v, *last, *rest, *p;
		unsigned run = 1, kk;
		unsigned long long epoch0, visit0;
		if(cached)
		{
			if(k >= l->n)
				break;
			if(deflt && !(k % BLKN) && k / BLKN < l->blk_cap && k + BLKN <= l->n)
			{
				const BLK *b = &l->blk[k / BLKN];
				if(b->ok && !(b->timed && (a2_TSDiff(b->wake, now) >> 8) < (int)frames) &&
						b->hi < w->view.map_cap[b->dev])
				{
					uint8_t *const map = w->view.map[b->dev];
					if(b->hi - b->lo == BLKN - 1)
						memset(map + b->lo, 1, BLKN);
					else
					{
						const uint32_t *sl = l->blk_slots + k;
						unsigned q;
						for(q = 0; q < BLKN; ++q)
							map[sl[q]] = 1;
					}
					k += BLKN;
					w->skipped += BLKN;
					w->unread += BLKN;
					w->in_blocks += BLKN;
					continue;
				}
				/* (a new summary of this block is under way: every voice of it that sleeps unseen adds to it) */
				acc_at = k;
				acc_ok = 1;
				acc_timed = 0;
				acc_best = 0x7fffffff;
				acc_lo = 0xffffffffu;
				acc_hi = 0;
				acc_dev = 0xffffffffu;
			}
			if(entry_sleeps_unread(w, &l->e[k], now, frames, deflt))
			{
				const ENT *e = &l->e[k];
				mark_entry(w, e, now);
				if(acc_ok && k - acc_at < BLKN)
				{
					const uint32_t slot = e->slotdev & 0x0fffffffu, dev = e->slotdev >> 28;
					if((e->flags & E_GROUP) || e->stamp == STAMP_NOUNITS || (acc_dev != 0xffffffffu && dev != acc_dev))
						acc_ok = 0;
					else
					{
						int d;
						acc_dev = dev;
						l->blk_slots[k] = slot;
						if(slot < acc_lo)
							acc_lo = slot;
						if(slot > acc_hi)
							acc_hi = slot;
						if(!(e->flags & E_VM) || (e->flags & E_VMEXIT))
						{
							const uint32_t t = (e->flags & E_VM) ? e->vm_exit : e->wake;
							d = a2_TSDiff(t, now);
							acc_timed = 1;
							if(d < acc_best)
							{
								acc_best = d;
								acc_wake = t;
							}
						}
						if(k - acc_at == BLKN - 1 && acc_at / BLKN < l->blk_cap)
						{
							++w->blocks_made;
							BLK *b = &l->blk[acc_at / BLKN];
							b->wake = acc_wake;
							b->timed = (uint8_t)acc_timed;
							b->lo = acc_lo;
							b->hi = acc_hi;
							b->dev = (uint8_t)acc_dev;
							b->ok = 1;
						}
					}
				}
				++k;
				++w->skipped;
				++w->unread;
				continue;
			}
			acc_ok = 0;
			/* (head: the link that points at this voice - an address, not a load) */
			v = l->e[k].v;
			if(k)
				head = &l->e[k - 1].v->next;
		}
		else
		{
			if(!(v = *head))
				break;
			if(k + AHEAD < l->n)
				__builtin_prefetch(l->e[k + AHEAD].v);	/* (a hint: never dereferenced here) */
			if(k < l->n && l->e[k].v != v && l->epoch != w->epoch)
				realign(l, k, v);
			if(voice_sleeps(w, l, k, v, now, frames, deflt))
			{
				mark_entry(w, &l->e[k], now);
				head = &v->next;
				++k;
				++w->skipped;
				continue;
			}
			if(k < l->n && l->e[k].v != v && l->epoch == w->epoch && w->served && !w->hooks_broken)
			{
				/* the list changed and neither a2_VoiceNew nor a2_VoiceFree came by here: an engine
				 * build that binds them locally - remembered lists cannot be trusted */
				w->hooks_broken = 1;
				fprintf(stderr, "a2amd walk: a2_VoiceNew / a2_VoiceFree are not interposable in this engine "
						"build; every voice's wake time will be read in every fragment\n");
			}
		}
		/* everything else is the engine's business: its own loop, on this voice - and on the
		 * voices behind it that need a visit as well - cut out of the list for the call */
		all_unread = 0;
		l->sum_ok = 0;
		acc_ok = 0;
		last = v;
		if(k < l->n && l->e[k].v == v)
			hint_visit(v, &l->up[k]);
		if(cached)
			while(run < 4096 && k + run < l->n && !entry_sleeps_unread(w, &l->e[k + run], now, frames, deflt))
			{
				last = l->e[k + run].v;
				hint_visit(last, &l->up[k + run]);
				++run;
			}
		else
			while(run < 4096 && last->next && !voice_sleeps(w, l, k + run, last->next, now, frames, deflt))
			{
				last = last->next;
				if(k + run < l->n && l->e[k + run].v == last)
					hint_visit(last, &l->up[k + run]);
				++run;
			}
		/* a voice of the run that sleeps through the window with no events (a group voice: its
		 * subvoices need the walk) cannot send its subvoices anything in this window: their list
		 * may be taken from memory by the call its inline unit will make */
		for(p = v, kk = 0; kk < run; p = p->next, ++kk)
			if(p->sub && !p->events && (a2_TSDiff(p->s.waketime, now) >> 8) >= (int)frames)
			{
				LIST *sl = list_of(w, &p->sub);
				if(sl)
					sl->quiet_visit = w->visits + 1;
			}
		for(kk = k / BLKN; kk < l->blk_cap && kk <= (k + run) / BLKN; ++kk)
			l->blk[kk].ok = 0;	/* (what is remembered of these voices is about to change) */
		recall_run(w, l, k, v, run);
		rest = last->next;
		last->next = NULL;
		epoch0 = w->epoch;
		visit0 = w->cur_visit;
		w->cur_visit = ++w->visits;
		{
			LIST *outer = w->cur_list;
			w->cur_list = l;
			engine_walk(st, head, offset, frames);
			w->cur_list = outer;
		}
		w->cur_visit = visit0;
		w->visited += run;
		if(w->epoch != epoch0 || l->epoch != w->epoch)
		{
			cached = 0;		/* voices were born or died in THIS list: the rest of it is read */
			if(l->blk)		/* (... and what stands where in it may move) */
				memset(l->blk, 0, l->blk_cap * sizeof(BLK));
		}
		/* what is left of them (voices that ended were freed: a2_VoiceFree, core.c:1892) goes back in
		 * front of the rest; and what to do with each while it sleeps */
		for(p = *head; p; p = p->next)
		{
			kk = k++;
			if(kk >= l->cap)
			{
				unsigned nc = l->cap ? l->cap * 2 : 16;
				ENT *ne = (ENT *)realloc(l->e, nc * sizeof(ENT));
				UPTR *nu = ne ? (UPTR *)realloc(l->up, nc * sizeof(UPTR)) : NULL;
				if(ne)
					l->e = ne;
				if(nu)
				{
					l->up = nu;
					l->cap = nc;
				}
			}
			if(kk < l->cap)
			{
				ENT *e = &l->e[kk];
				if(kk >= l->n)
					l->n = kk + 1;
				A2_unit *pu = p->units;
				int q;
				for(q = 0; q < 3; ++q)
				{
					l->up[kk].u[q] = pu;
					pu = pu ? pu->next : NULL;
				}
				e->v = p;
				e->slotdev = 0;
				e->wake = p->s.waketime;
				e->flags = (p->events ? 0 : E_NOEVENTS) | ((p->flags & A2_APIHANDLE) ? E_APIHANDLE : 0) |
						(p->sub ? E_GROUP : 0);
				e->sub = p->sub ? list_of(w, &p->sub) : NULL;
				if(e->sub)
					e->sub->above = l;
				if(p->sub && !e->sub)
					e->stamp = 0;		/* (out of memory: its subvoices keep being walked) */
				else if(!p->units)
					e->stamp = STAMP_NOUNITS;
				else
				{
					/* (a voice whose program the device can run from here on is handed over now)
					 * (only from the root window itself: a voice processed window by window - its
					 * parent woke in mid-fragment - would be wanted back for the next one) */
					int has_exit;
					uint32_t exit_when;
					if(deflt && offer_to_vm(w, st, p, &has_exit, &exit_when))
					{
						e->flags |= E_VM;
						if(has_exit)
						{
							e->flags |= E_VMEXIT;
							e->vm_exit = exit_when;
						}
					}
					e->stamp = w->served ? a2amd_units_standing(p->units, &e->slotdev) : 0;
				}
			}
			else
				cached = 0;
			head = &p->next;
		}
		*head = rest;
	}
	/* (a long list taken in blocks costs a few dozen tests per walk; its summary - every entry read - is worth making
	 * once it has slept through SUM_STREAK walks in a row, not every time a sequencer voice in it dozes off) */
	if(!(cached && all_unread))
		l->streak = 0;
	else if(l->blk && l->n >= 2 * BLKN && ++l->streak < SUM_STREAK)
		return;
	if(cached && all_unread && l->n)
	{
		/* every voice of the list slept, unseen: next time one test will do (until a voice of the
		 * list is visited again, the epoch moves, or the earliest wake time - in this list or below -
		 * comes) */
		uint32_t lo = 0xffffffffu, hi = 0, cnt = 0, dev = 0, ng = 0;
		int one_dev = 1, best = 0x7fffffff, fits = 1;
		unsigned long long voices = l->n;
		l->sum_ok = 0;
		for(k = 0; k < l->n; ++k)
			ng += (l->e[k].flags & E_GROUP) != 0;
		if(ng > l->ng || !l->gidx)
		{
			unsigned *gi = (unsigned *)realloc(l->gidx, (ng ? ng : 1) * sizeof(unsigned));
			if(!gi)
				return;
			l->gidx = gi;
		}
		l->ng = 0;
		l->sum_timed = 0;
		for(k = 0; k < l->n; ++k)
		{
			const ENT *e = &l->e[k];
			int d = a2_TSDiff(e->wake, now);
			if(!(e->flags & E_VM))		/* (a voice the device VM runs never wakes here ... */
			{
				l->sum_timed = 1;
				if(d < best)
				{
					best = d;
					l->sum_wake = e->wake;
				}
			}
			else if(e->flags & E_VMEXIT)	/* ... before the VM run that is the engine's again) */
			{
				d = a2_TSDiff(e->vm_exit, now);
				l->sum_timed = 1;
				if(d < best)
				{
					best = d;
					l->sum_wake = e->vm_exit;
				}
			}
			if(e->stamp != STAMP_NOUNITS &&
					(e->slotdev & 0x0fffffffu) >= w->view.map_cap[e->slotdev >> 28])
				fits = 0;
			if(e->flags & E_GROUP)
			{
				l->gidx[l->ng++] = k;
				d = a2_TSDiff(e->sub->sum_wake, now);
				if(e->sub->sum_timed)
				{
					l->sum_timed = 1;
					if(d < best)
					{
						best = d;
						l->sum_wake = e->sub->sum_wake;
					}
				}
				voices += e->sub->sum_voices;
				continue;
			}
			if(e->stamp == STAMP_NOUNITS)
				continue;
			if(!cnt)
				dev = e->slotdev >> 28;
			one_dev &= (e->slotdev >> 28) == dev;
			if((e->slotdev & 0x0fffffffu) < lo)
				lo = e->slotdev & 0x0fffffffu;
			if((e->slotdev & 0x0fffffffu) > hi)
				hi = e->slotdev & 0x0fffffffu;
			++cnt;
		}
		if(!fits)
			return;
		l->sum_voices = voices;
		l->sum_dev = dev;
		/* (slots are unique: cnt of them between lo and hi = lo + cnt - 1 are exactly that range) */
		l->sum_range = !cnt || (one_dev && hi - lo + 1 == cnt);
		l->sum_lo = (cnt && l->sum_range) ? lo : 0;
		l->sum_cnt = l->sum_range ? cnt : 0;
		l->sum_epoch = w->epoch;
		l->sum_ok = 1;
		++w->summaries;
	}
	if(!cached)
	{
		/* walked link by link to its end: e[0..k) is the list */
		if(k < l->n)
			l->n = k;
		l->epoch = l->n == k ? w->epoch : w->epoch - 1;
	}
}

#ifndef A2AMD_SRCHASH
#define A2AMD_SRCHASH "unstamped"
#endif
const char *a2amd_walk_source_stamp(void) { return "A2AMD_SRCHASH:" A2AMD_SRCHASH; }
/* ... and of the ENGINE headers it was compiled against (this file reads A2_voice / A2_state members: a library made
 * for another engine version has another layout).  Checked wherever that engine's tree is (build.py). */
#ifndef A2AMD_ENGHASH
#define A2AMD_ENGHASH "unstamped"
#endif
const char *a2amd_walk_engine_stamp(void) { return "A2AMD_ENGHASH:" A2AMD_ENGHASH; }
