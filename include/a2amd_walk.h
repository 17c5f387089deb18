/*
 * a2amd_walk.h - the interface between the drop-in units (liba2amd_units.so) and the
 * replacement of the engine's voice walk (liba2amd_walk.so, INTEGRATION.md option C).
 * No engine types in it: a2amd_walk.c includes the engine's own internal headers next to
 * this file, a2amd_units.c the mirror declarations of a2amd_plugin.h.
 */
#ifndef A2AMD_WALK_H
#define A2AMD_WALK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- INTEGRATION.md option C: the engine's voice walk with a short cut for sleeping voices ----
 * a2_ProcessVoices (src/core.c:1883-1896) visits every voice in every fragment, whatever its
 * units do; audiality2_amd/csrc/a2amd_walk.c (compiled against the ENGINE's internal headers
 * into liba2amd_walk.so, loaded in front of libaudiality2) interposes that one function and
 * skips the visit for a voice that would get nothing but the default window.  What it needs
 * from the drop-in units: */
#define A2AMD_WALK_MAXDEV 8
typedef struct a2amd_walkview
{
	const void		*cfg;		/* the A2_config this view belongs to (a2amd_units_walkview) */
	uint8_t *const		*map;		/* [dev]: the backends' default maps of the open fragment (a2amd_default_map) */
	const unsigned		*map_cap;	/* [dev] */
	const unsigned		*base;		/* engine offset and length of the open root window: only the window */
	const unsigned		*win_frames;	/* (*base, *win_frames) may be reported through the map */
	uint32_t *const		*qstamp;	/* [dev][slot]: non-zero while that voice's head unit reports its default
						 * windows through the map (A2_unit.Process == the drop-in's byte store);
						 * a new value every time it starts to */
	const unsigned		*qstamp_cap;	/* [dev] */
	int			*walker;	/* set by the walk: the units' own prefetch hints are switched off */
	const unsigned		*serial;	/* *serial == serial_value while the engine state this view was made for is open */
	unsigned		serial_value;
	void			*state;		/* for a2amd_units_hold() */
	const unsigned		*frag_serial;	/* counts the root windows (= backend fragments) the state has opened */
	const unsigned		*vm_live;	/* voices of the state the device VM runs right now (a2amd_units_vm_adopt) */
} a2amd_walkview;
/* The view of the engine state with this A2_config; 0, or -1 when the drop-in does not serve it. */
int a2amd_units_walkview(const void *cfg, a2amd_walkview *out);
/* 'head' = the first unit of a voice (A2_voice.units).  Returns the voice's current stamp -
 * 0 unless the unit is the head of a chain of the drop-in's own units in byte-store mode -
 * and its slot (bits 0..27) and context (bits 28..31) in *slotdev. */
uint32_t a2amd_units_standing(const void *head, uint32_t *slotdev);
/* a2amd_default_hold() / a2amd_default_release_all() (include/a2amd.h) on context 'dev' of the engine
 * state 'state' (a2amd_walkview.state): the voices in slots[0..n) report their default window in
 * every fragment until released.  n == 0 and on == 0: every hold of the state, in all contexts. */
int a2amd_units_hold(void *state, unsigned dev, const uint32_t *slots, unsigned n, int on);

/* ---- the scripted voice's VM on the device (include/a2amd_vm.h) --------------------
 * The walk is where a voice's VM state (A2_voice.s), its program (A2_voice.program) and its register
 * wiring (A2_voice.cregs) are visible; the units know which backend unit each of them is.
 *
 * a2amd_units_vm_adopt: the engine has just processed the voice whose first unit is 'head' in the
 * open root window; hand it to the device VM from the next fragment on.
 *   code / nwords   the voice's current function (A2_function.code / .size, src/internals.h:417-425)
 *   vmstate         &A2_voice.s (A2_vmstate, include/a2_vm.h:62-69)
 *   wr_unit, wr_fn  A2_voice.cregs[i].unit / .write for the 64 VM registers (src/internals.h:573)
 *   now             st->now_fragstart + (root window offset << 8): engine time of the open fragment's start
 *   msdur           A2_state.msdur
 * Returns 0 (adopted: the voice now counts as asleep for good - report its default windows like a
 * sleeping voice's, never hand it to the engine without a2amd_units_vm_recall()), -1 (never, for this
 * program from this pc: outside the subset), -2 (not now).
 * *has_exit = 1: the voice was taken for a stretch only (a2amd_vm_exit_time(), a2amd_vm.h) - it has to be
 * recalled and handed to the engine in the fragment that holds the engine time *exit_when. */
int a2amd_units_vm_adopt(const void *head, const uint32_t *code, unsigned nwords, const void *vmstate,
		void *const *wr_unit, void *const *wr_fn, uint32_t now, uint32_t msdur, int *has_exit, uint32_t *exit_when);
/* build stamps ("A2AMD_SRCHASH:<32 hex>", see a2amd_source_stamp() in a2amd.h) */
const char *a2amd_units_source_stamp(void);
const char *a2amd_walk_source_stamp(void);
/* 1 when the voice is the device VM's. */
int a2amd_units_vm_is(const void *head);
/* The engine is about to process these n voices again: vmstates[k] (A2_vmstate *) receives what the
 * engine's own VM would have left there by the start of the open fragment. */
int a2amd_units_vm_recall(const void *const *heads, unsigned n, void *const *vmstates);


#ifdef __cplusplus
}
#endif
#endif /* A2AMD_WALK_H */
