/*
 * a2amd_units.c - drop-in voice units for Audiality 2 backed by liba2amd.so.
 *
 * Builds into liba2amd_units.so.  Loaded ahead of libaudiality2 (LD_PRELOAD,
 * or simply earlier on the link line) its unit descriptors take the place of
 * the engine's built-in wtosc / panmix / filter12 / fbdelay (and wrap inline /
 * xinsert): the engine's unit table (src/audiality2.c:183-207) refers to the
 * descriptors by symbol, and ELF symbol interposition binds it to ours.  The
 * engine itself - A2S compiler, VM, event scheduler, voice tree, API - runs
 * unchanged on the CPU and keeps calling Initialize / write / Process /
 * Deinitialize exactly as before (include/a2_units.h:115-176); here those calls
 * are forwarded to the GPU backend (include/a2amd.h), which renders all voices
 * of a fragment together.
 *
 * Where the audio re-enters the engine: the root voice is
 * "inline; panmix; xinsert >" (a2_rootdriver, audiality2.c:271-291).  Every
 * time the engine processes a window of the root voice, its inline unit walks
 * the whole voice tree (our units record), then its panmix is called: at that
 * point everything that feeds this window has been recorded, so the GPU renders
 * it and the result is written where the engine expects the root panmix to have
 * left it - the input buffers of the root xinsert.  The engine's own xinsert
 * then moves it to the master bus and feeds any sink/insert clients (a2play's
 * silence detector, a2_Render's stream) with real audio.
 *
 * One engine window of the root voice = one fragment of the backend; offsets of
 * all other voices are rebased to the start of that window.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <math.h>
#include "../../include/a2amd.h"
#include "../../include/a2amd_plugin.h"
#include "../../include/a2amd_vm.h"

/* engine API we call back into (public: a2_waves.h:183, a2_properties.h:106) */
extern A2P_wave *a2_GetWave(void *iface, int handle);
extern int a2_GetStateProperty(void *iface, int prop, int *v);
extern int a2_SetStateProperty(void *iface, int prop, int v);

/* engine-internal, present when the engine's symbols are visible (xinsert.c:57,
 * xinsertapi.c:114) */
extern int a2r_Error(void *st, int e, const char *info) __attribute__((weak));
extern int a2_XinsertRemoveClient(A2P_xinsert_client *xic) __attribute__((weak));

#define WALK_AHEAD 32		/* most chain heads the walk's prefetch hints run ahead of the engine (look_ahead()) */
#define MAXDEV    8		/* GPUs one engine state may be spread over (A2AMD_DEVICES) */

/* A2AMD_DEVICES > 1: what the engine asked of a voice before it was first processed -
 * i.e. before the drop-in knows which bus, and therefore which GPU, it belongs to */
typedef struct BIRTHOP
{
	A2P_unit	*u;
	A2P_vmstate	*vms;		/* the voice */
	int		is_write;
	int		kind, nin, nout, wired;		/* Initialize */
	unsigned	flags, wakefrac;
	int		reg, v;				/* write */
	unsigned	start, dur;
	int		transpose;
} BIRTHOP;

/* a READ client's window, waiting for the audio of its fragment */
typedef struct PENDING
{
	A2P_xinsert		*xi;
	A2P_xinsert_client	*xic;
	int			uid;		/* backend unit, or -1: the root xinsert (its input is the master bus) */
	int			dev;
	int			frag;		/* fragment of the batch the window lies in */
	unsigned		pos;		/* root xinsert: frames into the buffer */
	const char		*what;		/* for error reports */
	unsigned		offset, frames;
	int			insert;		/* a READ and WRITE client: served in the middle of the render, */
	int			depth;		/* ... when it pauses behind this nesting depth */
} PENDING;

typedef struct HOSTSTATE
{
	A2P_config	*cfg;
	a2amd_ctx	*ctx;		/* = ctxs[0]: owns the root chain, delivers the audio */
	int		ndev;		/* contexts (GPUs) this state's voice subtrees are dealt over */
	a2amd_ctx	*ctxs[MAXDEV];
	int		root_uid[2][MAXDEV];	/* the root voice's inline and panmix in every context */
	int		rr;		/* next context for a voice that mixes straight into the root's bus */
	int		load[MAXDEV];	/* units alive in each context: a new subtree goes where the fewest are */
	int		dev_stack[72];	/* context of each open inline window, by depth */
	BIRTHOP		*births;
	int		nbirths, cap_births;
	int		refs;
	int		failed;		/* the backend reported an error: this state renders silence from here on */
	int		depth;		/* open inline windows */
	unsigned	base;		/* engine offset of the current root window */
	unsigned	win_frames;	/* ... and its length = the open backend fragment */
	uint8_t		*map[MAXDEV];	/* the backends' default maps for that fragment (a2amd_default_map) */
	unsigned	map_cap[MAXDEV];
	int		noise_oscs;	/* oscillators playing the noise wave: their voices need the engine's RNG */
	int		ninserts;	/* pending windows of insert clients in the batch being recorded */
	int		no_quick;	/* A2AMD_NO_QUICK=1: every Process call is forwarded (A/B measurements) */
	/* a2amd_walkview (INTEGRATION.md option C): which voices report their default windows
	 * through the map right now, as a stamp per (context, slot) */
	uint32_t	*qstamp[MAXDEV];
	unsigned	qstamp_cap[MAXDEV];
	uint32_t	stamp_ctr;
	int		walker;		/* liba2amd_walk.so walks this state: no prefetch hints of our own */
	unsigned	serial;		/* a number of its own for every engine state ever opened (a2amd_walkview) */
	unsigned	frag_serial;	/* root windows opened so far (a2amd_walkview) */
	/* Round 6: the root voice's last wake time (root_vms, below, is its VM state inside its A2_voice) and the distance
	 * of its last two wake-ups: what hint_cuts() announces to the backend - where the engine will cut the next buffer's
	 * fragments (a2amd_vm_expect_cuts, include/a2amd_vm.h) */
	unsigned	root_wake, root_period;
	int		root_wake_valid;
	unsigned	vm_live;	/* voices of this state the device VM runs (a2amd_units_vm_adopt) */
	int		no_vm;		/* A2AMD_NO_VM=1: no voice is handed to the device VM (A/B measurements) */
	/* Engine states (master states and a2_Render's substates) come and go; their records are
	 * allocated as needed, chained, and reused when a state has closed - never freed: the voice
	 * walk of INTEGRATION.md option C holds pointers into them (a2amd_walkview). */
	struct HOSTSTATE *next_state;
	int		index;		/* position in that chain (A2AMD_DEVICE=all deals states over the GPUs by it) */
	char		errmsg[160];	/* (a2r_Error keeps the pointer: realtime states post it to the API side) */
	char		chainmsg[200];
	/* the last WALK_AHEAD chain heads the engine called, oldest first from walk_pos
	 * (look_ahead()); emptied whenever a unit goes away */
	struct XTRA	*walk_ring[WALK_AHEAD];
	unsigned	walk_pos;
	int		walk_ahead;	/* how many heads ahead; A2AMD_WALK_AHEAD=0 switches the hints off (A/B measurements) */
	A2P_vmstate	*root_vms;	/* the root voice (first voice of a state) */
	A2P_vmstate	*chain_vms;	/* voice whose chain is being populated */
	A2P_unit	*chain_last;
	A2P_unit	*chain_head;	/* ... its first forwarded unit, forwarded units so far, env units in front of them */
	int		chain_nfwd;
	A2P_unit	*chain_env[2];
	int		chain_nenv;
	int		capturing;	/* a2_RenderWave's substate: what it renders is kept on the device */
	/* ... and in the state that asked for the wave: captures waiting for the engine thread to build the wave's
	 * device copy from them, at the wave's first use (a2_RenderWave may be called from the API thread of a
	 * realtime state: the backend context belongs to the engine thread) */
	struct PENDCAP { A2P_wave *w; int handle; a2amd_capture *cap; } *pendcaps;	/* (handle: the wave's, by which w is looked up again before it is trusted) */
	int		npendcaps, cap_pendcaps;
	pthread_mutex_t	pendcaps_mtx;
	int		pendcaps_mtx_ok;
	int		envluts_sent[MAXDEV];	/* the context has env's tables (a2amd_vm_envluts) */
	A2P_xinsert	*root_xi;	/* the root voice's xinsert (the engine's own instance) */
	void		*engine_state;	/* A2_state, for a2r_Error */
	/* wave registry: engine object -> device wave id */
	A2P_wave	**wave_ptr;
	int		(*wave_id)[MAXDEV];	/* per context, -1 = not uploaded there */
	int		nwaves, cap_waves;
	int		swept;		/* the registry has been checked for released waves in this buffer */
	int32_t		out[A2AMD_MAXCHANNELS][A2AMD_MAXFRAG];
	PENDING		*pend;
	int		npend, cap_pend;
	A2P_xinsert	**zombies;	/* client lists of xinserts that died owing windows */
	int		nzombies, cap_zombies;
	/* One GPU round trip per driver buffer (a2_Run() / a realtime driver's callback)
	 * instead of one per root window: */
	unsigned	max_batch;	/* fragments the backend records between renders */
	A2P_audiodriver	*drv;		/* the state's audio driver, once our Process sits in front of the engine's */
	void		(*drv_process)(A2P_audiodriver *driver, unsigned frames);	/* the engine's: a2_AudioCallback */
	int		in_buffer;	/* inside drv_process() */
	int		decided;	/* ... and 'batching' has been chosen for this buffer */
	int		batching;	/* the buffer is recorded whole and rendered at its end */
	unsigned	batch_frags;	/* backend fragments recorded since the last render */
	unsigned	rec_pos;	/* frames of the buffer recorded so far */
	unsigned	win_pos;	/* ... up to the start of the open root window */
	unsigned	acc_pos;	/* frames of the buffer rendered so far (in acc) */
	unsigned	acc_cap;
	/* round 5: the first half of a long buffer is on the GPU while the engine walks the second (flush_part) */
	unsigned	split_frames;	/* this buffer: hand the batch over once this many frames are recorded (0: whole) */
	unsigned	part_at, part_frames;	/* the part in flight: where its audio goes in acc, how many frames */
	int32_t		*acc[A2AMD_MAXCHANNELS];	/* the buffer's master bus, as rendered */
	int32_t		*rinj[A2AMD_MAXCHANNELS];	/* what WRITE clients of the root xinsert produced */
	int		rinj_used;
} HOSTSTATE;

static HOSTSTATE *states;	/* chain of every record ever allocated; refs == 0: closed, reusable */
static pthread_mutex_t states_mtx = PTHREAD_MUTEX_INITIALIZER;	/* independent master states may open from different threads */

/* an a2_RenderWave() call in progress on this thread (the interposed entry point at the end of
 * this file): the next engine state that opens is its off-line substate, and what that state renders is
 * kept in device memory for the wave */
typedef struct RENDERCAP
{
	struct HOSTSTATE	*sub;
	a2amd_capture		*cap;
	int			armed;
} RENDERCAP;
static __thread RENDERCAP *rendering;

/* Our per-instance data lives in the engine's 384 byte instance block: right
 * behind the A2_unit header (the cache line after the one the engine's dispatch
 * loop has just touched) for the units that are entirely ours, at the tail of
 * the block for the two wrapped engine units, whose own structs (A2_inline,
 * A2_xinsert) start the block. */
typedef struct XTRA
{
	HOSTSTATE	*hs;
	int		slot;		/* head: the voice's slot in the backend's default map */
	int		dev;		/* which of the state's contexts holds the voice */
	/* head of a chain of our own: what the engine will touch WALK_AHEAD voices from
	 * here, as the last walk found it (prefetch hints only, never dereferenced) */
	A2P_unit	*ahead;		/* that voice's head unit ... */
	A2P_unit	*ahead_tail[2];	/* ... the units behind it ... */
	A2P_vmstate	*ahead_vms;	/* ... and its VM state, inside its A2_voice */
	/* (what a register write and the first unit's Process read ends with the first 64 bytes) */
	A2P_vmstate	*vms;
	int		uid;		/* backend unit id, -1 = not forwarded */
	int16_t		kind;
	int8_t		pending;	/* its Initialize / writes wait in HOSTSTATE.births for the voice's first window */
	int8_t		is_noise;	/* wtosc: it plays the noise wave (set by its 'w' write) */
	A2P_unit	*head;		/* the voice's head unit, once its chain was found to be all ours */
	A2P_unit	*tail[2];	/* head: the units behind this one */
	unsigned	client_mode;	/* xinsert / xsink / xsource: A2AMD_XIO_*, what its clients need */
	int		is_root;
	A2P_process_cb	orig_process;
	int		is_head;	/* first forwarded unit of its voice */
	int		chain_checked;	/* the units behind us in the voice have been looked at */
	int		refused;	/* an unsupported client was reported once */
	void		(*orig_setprocess)(A2P_unit *u);	/* root xinsert: the engine's xi_SetProcess */
	/* head: the device VM runs the voice's program (a2amd_units_vm_adopt) / when the engine last had to
	 * take it back, or looked at it and found it awake (frag_serial: a voice is offered to the VM when it
	 * has been awake twice within a second or two - one that sleeps for minutes gains nothing) */
	int		vm;
	unsigned	vm_seen;
	int		vm_seen_valid;
	/* head: the backend looked ahead through the voice's program and found the stay not worth it (too short, no
	 * writes): not asked again before frag_serial vm_retry - the look-ahead is up to thousands of VM runs on the
	 * engine thread, and a voice that loops on delays would otherwise pay it at every wake-up.  Doubles per refusal. */
	unsigned	vm_retry;
	uint8_t		vm_backoff;
	/* head: the voice's env units (a2_env_unitdesc below) and how many forwarded units stand in front of each;
	 * nenv > 2: too many for the device VM */
	A2P_unit	*env[2];
	uint8_t		nenv, env_before[2];
} XTRA;

_Static_assert(MAXDEV == A2AMD_WALK_MAXDEV, "a2amd_walkview");
_Static_assert(sizeof(A2P_unit) <= 64 && 64 + sizeof(XTRA) <= A2P_BLOCK_SIZE && offsetof(XTRA, head) == 64, "XTRA placement");

/* the voice of head unit 'x' starts (on) / stops reporting its default windows through the map */
static inline void set_stamp(HOSTSTATE *hs, XTRA *x, int on)
{
	if((unsigned)x->slot < hs->qstamp_cap[x->dev])
	{
		if(on && !++hs->stamp_ctr)
			hs->stamp_ctr = 1;
		hs->qstamp[x->dev][x->slot] = on ? hs->stamp_ctr : 0;
	}
}

/* room for the stamp of (context, slot); without it that voice simply keeps being visited */
static void stamp_room(HOSTSTATE *hs, int dev, int slot)
{
	if((unsigned)slot >= hs->qstamp_cap[dev])
	{
		unsigned nc = hs->qstamp_cap[dev] ? hs->qstamp_cap[dev] : 1024;
		uint32_t *nq;
		while(nc <= (unsigned)slot)
			nc *= 2;
		if((nq = (uint32_t *)realloc(hs->qstamp[dev], nc * sizeof(uint32_t))))
		{
			memset(nq + hs->qstamp_cap[dev], 0, (nc - hs->qstamp_cap[dev]) * sizeof(uint32_t));
			hs->qstamp[dev] = nq;
			hs->qstamp_cap[dev] = nc;
		}
	}
}

static inline XTRA *xtra(A2P_unit *u)
{
	if(u->descriptor == &a2_inline_unitdesc || u->descriptor == &a2_xinsert_unitdesc ||
			u->descriptor == &a2_xsink_unitdesc || u->descriptor == &a2_xsource_unitdesc)
		return (XTRA *)((char *)u + A2P_BLOCK_SIZE - sizeof(XTRA));
	return (XTRA *)((char *)u + 64);
}

/* A backend call failed.  There is no CPU path to continue on, and a library
 * inside somebody's audio application does not get to abort(): the error goes
 * to the engine's own channel (a2r_Error, src/interface.c:401-424 - the
 * application sees it through a2_LastRTError() / its log) and this engine state
 * renders silence from here on. */
static void fail(HOSTSTATE *hs, const char *what, int rc)
{
	static char nostate[160];
	char *m;
	if(hs && hs->failed)
		return;
	m = hs ? hs->errmsg : nostate;
	snprintf(m, sizeof(nostate), "a2amd: %s failed (%d): %s", what, rc, a2amd_last_error(NULL));
	if(hs)
		hs->failed = 1;
	if(a2r_Error && hs && hs->engine_state)
		a2r_Error(hs->engine_state, A2P_INTERNAL, m);
	else
		fprintf(stderr, "a2amd units: %s\n", m);
}

static const A2P_unitdesc *orig_desc(const char *sym)
{
	/* the engine's own descriptor of a wrapped unit: the next definition of the
	 * symbol in load order (interposition), or - in an engine
	 * built with the drop-in in its unit table - the same
	 * object under the name <sym>_cpu */
	const A2P_unitdesc *d = (const A2P_unitdesc *)dlsym(RTLD_NEXT, sym);
	if(!d)
	{
		char alt[96];
		snprintf(alt, sizeof(alt), "%s_cpu", sym);
		d = (const A2P_unitdesc *)dlsym(RTLD_DEFAULT, alt);
	}
	if(!d)
		fprintf(stderr, "a2amd units: the engine has no %s to wrap\n", sym);
	return d;	/* NULL: the caller fails its OpenState / Initialize, the engine reports it */
}

/* ---- state open / close (A2_unitdesc.OpenState / CloseState) -------------*/
static int amd_open(A2P_config *cfg, void **statedata)
{
	int rc = 0;
	/* No GPU: fail where the engine can still fail cleanly - a2_RegisterUnit()
	 * returns the error and a2_Open() hands it to the application
	 * (src/units.c:142-146, src/audiality2.c:256-260).  (The backend itself is opened
	 * later: A2_config.basepitch is not valid yet.) */
	if(a2amd_device_count() <= 0)
	{
		fprintf(stderr, "a2amd units: no usable HIP device; this library has no CPU fallback\n");
		return A2P_DEVICEOPEN;
	}
	pthread_mutex_lock(&states_mtx);
	{
		static unsigned serials;
		HOSTSTATE *hs, *freeone = NULL, **tail = &states;
		int n = 0;
		for(hs = states; hs; tail = &hs->next_state, hs = hs->next_state, ++n)
			if(hs->refs && hs->cfg == cfg)
			{
				++hs->refs;
				*statedata = hs;
				pthread_mutex_unlock(&states_mtx);
				return 0;
			}
			else if(!hs->refs && !freeone)
				freeone = hs;
		if(!freeone && (freeone = (HOSTSTATE *)calloc(1, sizeof(HOSTSTATE))))
		{
			freeone->index = n;
			*tail = freeone;
		}
		if(!freeone)
			rc = A2P_OOMEMORY;
		else
		{
			HOSTSTATE *nx = freeone->next_state;
			const int ix = freeone->index;
			memset(freeone, 0, sizeof(HOSTSTATE));
			freeone->next_state = nx;
			freeone->index = ix;
			freeone->cfg = cfg;
			freeone->refs = 1;
			if(!++serials)
				++serials;
			freeone->serial = serials;
			freeone->pendcaps_mtx_ok = !pthread_mutex_init(&freeone->pendcaps_mtx, NULL);
			if(rendering && rendering->armed)
			{
				rendering->armed = 0;
				rendering->sub = freeone;
				freeone->capturing = 1;
			}
			*statedata = freeone;
		}
	}
	pthread_mutex_unlock(&states_mtx);
	return rc;
}

static void amd_close(void *statedata)
{
	HOSTSTATE *hs = (HOSTSTATE *)statedata;
	int c;
	if(!hs)
		return;
	pthread_mutex_lock(&states_mtx);
	if(!--hs->refs)
	{
		if(hs->capturing && rendering && rendering->sub == hs && hs->ctxs[0] && !hs->failed)
			a2amd_capture_end(hs->ctxs[0], &rendering->cap);
		if(getenv("A2AMD_WAVE_STATS") && hs->ctxs[0])
		{
			uint64_t bytes = 0;
			uint32_t up = 0, res = 0;
			a2amd_wave_stats(hs->ctxs[0], &bytes, &up, &res);
			fprintf(stderr, "a2amd units: state %d%s: %u waves copied from the host (%llu bytes), %u built on the device from "
					"what it rendered\n", hs->index, hs->capturing ? " (a2_RenderWave substate)" : "", up,
					(unsigned long long)bytes, res);
		}
		for(c = 0; c < MAXDEV; ++c)
			if(hs->ctxs[c])
				a2amd_close(hs->ctxs[c]);
		hs->ctx = NULL;
		for(c = 0; c < hs->npendcaps; ++c)
			a2amd_capture_free(hs->pendcaps[c].cap);
		free(hs->pendcaps);
		if(hs->pendcaps_mtx_ok)
			pthread_mutex_destroy(&hs->pendcaps_mtx);
		free(hs->births);
		/* (the engine clears A2_audiodriver.Process itself when the state closes,
		 * src/audiality2.c:733) */
		free(hs->wave_ptr);
		free(hs->wave_id);
		free(hs->pend);
		free(hs->zombies);
		for(c = 0; c < MAXDEV; ++c)
			free(hs->qstamp[c]);
		for(c = 0; c < A2AMD_MAXCHANNELS; ++c)
		{
			free(hs->acc[c]);
			free(hs->rinj[c]);
		}
		{
			HOSTSTATE *nx = hs->next_state;
			const int ix = hs->index;
			memset(hs, 0, sizeof(*hs));	/* (refs = 0: free for the next state; serial = 0: a2amd_walkview) */
			hs->next_state = nx;
			hs->index = ix;
		}
	}
	pthread_mutex_unlock(&states_mtx);
}

/* the backend is opened on first use: A2_config.basepitch is only valid once
 * a2_Open() has returned (audiality2.c:398) */
static a2amd_ctx *ctx_of(HOSTSTATE *hs)
{
	if(!hs->ctx && !hs->failed)
	{
		a2amd_config c;
		int rc;
		memset(&c, 0, sizeof(c));
		c.struct_size = sizeof(c);
		c.samplerate = hs->cfg->samplerate;
		c.basepitch = hs->cfg->basepitch;
		c.channels = hs->cfg->channels;
		/* A2AMD_DEVICE=<n>: that HIP device; A2AMD_DEVICE=all: the engine
		 * states of this process are dealt round robin over all devices
		 * (independent states exchange nothing: no collective) */
		c.device = getenv("A2AMD_DEVICE") ? atoi(getenv("A2AMD_DEVICE")) : 0;
		if(getenv("A2AMD_DEVICE") && !strcmp(getenv("A2AMD_DEVICE"), "all"))
		{
			int n = a2amd_device_count();
			c.device = n > 0 ? hs->index % n : 0;
		}
		/* fragments recorded per GPU round trip: a whole driver buffer, up to 256
		 * (A2AMD_BATCH=1: one round trip per root window, as in round 1) */
		c.max_batch = getenv("A2AMD_BATCH") ? (unsigned)atoi(getenv("A2AMD_BATCH")) : 256;
		if(c.max_batch < 1)
			c.max_batch = 1;
		if(c.max_batch > 256)
			c.max_batch = 256;
		hs->max_batch = c.max_batch;
		hs->no_quick = getenv("A2AMD_NO_QUICK") != NULL;
		hs->no_vm = getenv("A2AMD_NO_VM") != NULL;
		hs->walk_ahead = getenv("A2AMD_WALK_AHEAD") ? atoi(getenv("A2AMD_WALK_AHEAD")) : 12;
		if(hs->walk_ahead < 0 || hs->walk_ahead > WALK_AHEAD)
			hs->walk_ahead = WALK_AHEAD;
		/* A2AMD_DEVICES=<n>: this ONE engine state is spread over n GPUs - every voice
		 * subtree below the root (an a2_NewGroup group, src/interface.c:888, or a voice
		 * playing straight into the root's bus) lives on one of them, dealt round robin;
		 * the root chain runs on the first.  Devices c.device, c.device + 1, ... (modulo
		 * the number present: on a single-GPU box the contexts share it). */
		hs->ndev = getenv("A2AMD_DEVICES") ? atoi(getenv("A2AMD_DEVICES")) : 1;
		if(hs->ndev < 1)
			hs->ndev = 1;
		if(hs->ndev > MAXDEV)
			hs->ndev = MAXDEV;
		if(hs->cfg->channels != 2)
			hs->ndev = 1;	/* (mono substates - a2_Render's - stay on one GPU) */
		{
			int d, nd = a2amd_device_count(), base = c.device;
			for(d = 0; d < hs->ndev && !hs->failed; ++d)
			{
				c.device = nd > 0 ? (base + d) % nd : 0;
				if((rc = a2amd_open(&c, &hs->ctxs[d])))
				{
					hs->ctxs[d] = NULL;
					fail(hs, "a2amd_open", rc);
				}
			}
			if(!hs->failed && hs->ndev > 1 && (rc = a2amd_dist_init_local(hs->ctxs, hs->ndev)))
				fail(hs, "a2amd_dist_init_local", rc);
			if(hs->failed)
				for(d = 0; d < MAXDEV; ++d)
					if(hs->ctxs[d])
					{
						a2amd_close(hs->ctxs[d]);
						hs->ctxs[d] = NULL;
					}
			hs->ctx = hs->ctxs[0];
			if(hs->ctx && hs->capturing && (rc = a2amd_capture_begin(hs->ctx)))
				hs->capturing = 0;	/* (the wave is uploaded from the engine's copy, then) */
		}
	}
	return hs->ctx;
}

#define XCTX(x) ((x)->hs->ctxs[(x)->dev])

static int wave_id_of(HOSTSTATE *hs, int dev, A2P_wave *w);
static inline XTRA *xtra(A2P_unit *u);
static int is_ours(const A2P_unitdesc *d);

static BIRTHOP *new_birthop(HOSTSTATE *hs)
{
	if(hs->nbirths == hs->cap_births)
	{
		int nc = hs->cap_births ? hs->cap_births * 2 : 256;
		BIRTHOP *nb = (BIRTHOP *)realloc(hs->births, nc * sizeof(BIRTHOP));
		if(!nb)
			return NULL;
		hs->births = nb;
		hs->cap_births = nc;
	}
	memset(&hs->births[hs->nbirths], 0, sizeof(BIRTHOP));
	return &hs->births[hs->nbirths++];
}

/* A2AMD_DEVICES > 1: the voice of unit 'x' is being processed for the first time.
 * Where its output goes is known now - into the bus of the innermost open inline
 * window - and with that its GPU: the one that holds that bus, or, if that is the
 * root's (every context has a partial of it), the next one in turn.  Its units are
 * created there and the register writes it has received so far follow. */
static void route_voice(XTRA *x)
{
	HOSTSTATE *hs = x->hs;
	A2P_vmstate *vms = x->vms;
	int dev;
	if(hs->depth <= 1)
	{
		/* a subtree of its own: the context with the fewest units so far, in turn among equals
		 * (groups made in a row before any of their voices exist are dealt round robin) */
		int d, best = hs->rr % hs->ndev;
		for(d = 1; d < hs->ndev; ++d)
			if(hs->load[(hs->rr + d) % hs->ndev] < hs->load[best])
				best = (hs->rr + d) % hs->ndev;
		dev = best;
		hs->rr = best + 1;
	}
	else
		dev = hs->dev_stack[hs->depth - 1];
	int k, n = 0, rc;
	for(k = 0; k < hs->nbirths; ++k)
	{
		BIRTHOP *b = &hs->births[k];
		XTRA *bx;
		if(b->vms != vms)
		{
			hs->births[n++] = *b;
			continue;
		}
		bx = xtra(b->u);
		if(hs->failed)
			continue;
		if(!b->is_write)
		{
			bx->dev = dev;
			bx->pending = 0;
			++hs->load[dev];
			bx->uid = a2amd_unit_init(hs->ctxs[dev], (uint64_t)(uintptr_t)vms, b->kind, b->flags, b->nin,
					b->nout, b->wired, b->transpose, b->wakefrac);
			if(bx->uid < 0)
				fail(hs, "a2amd_unit_init", bx->uid);
		}
		else
		{
			int v = b->v;
			if(bx->kind == A2AMD_WTOSC && b->reg == 0)
				v = wave_id_of(hs, dev, a2_GetWave(hs->cfg->interface, v >> 16));
			if(bx->uid >= 0 && (rc = a2amd_unit_write(hs->ctxs[dev], bx->uid, b->reg, v, b->start, b->dur, b->transpose)))
				fail(hs, "a2amd_unit_write", rc);
		}
	}
	hs->nbirths = n;
}

/* ---- Initialize / Deinitialize ---------------------------------------------*/
static int amd_init(int kind, A2P_unit *u, A2P_vmstate *vms, void *sd, unsigned flags, int forward)
{

	HOSTSTATE *hs = (HOSTSTATE *)sd;
	XTRA *x = xtra(u);
	a2amd_ctx *ctx = ctx_of(hs);
	int wired = u->outputs != u->inputs;
	unsigned lflags = flags;
	memset(x, 0, sizeof(*x));
	x->hs = hs;
	x->vms = vms;
	x->kind = kind;
	x->uid = -1;
	if(!ctx)
		return A2P_DEVICEOPEN;	/* no GPU backend: the engine reports A2_VOICEINIT and drops the voice */
	if(!hs->root_vms)
		hs->root_vms = vms;
	x->is_root = vms == hs->root_vms;
	if(hs->chain_vms != vms)
	{
		hs->chain_vms = vms;
		hs->chain_last = NULL;
		hs->chain_head = NULL;
		hs->chain_nfwd = hs->chain_nenv = 0;
	}
	if(x->is_root && kind == A2AMD_PANMIX)
	{
		/* the root panmix is where the audio returns to the engine: for
		 * the backend it feeds the master bus directly */
		wired = 1;
		lflags |= A2AMD_PROCADD;
	}
	if(x->is_root && kind == A2AMD_XINSERT)
		forward = 0;	/* stays with the engine (its input is the master bus we hand back) */
	if(forward && !hs->chain_last && kind != A2AMD_INLINE &&
			(u->ninputs || ((flags & A2AMD_PROCADD) && !wired)))
	{
		/* the first replaced unit of a voice consumes (or adds to) a signal:
		 * something in front of it produced that on the CPU */
		fprintf(stderr, "a2amd units: '%s' is the first GPU-rendered unit of its voice but takes its "
				"input from a unit that is not replaced: mixed CPU/GPU chains are not "
				"supported (no CPU fallback)\n", u->descriptor->name);
		return A2P_NOTIMPLEMENTED;	/* A2_VOICEINIT for this voice */
	}
	if(forward && hs->ndev > 1 && x->is_root && kind != A2AMD_INLINE && kind != A2AMD_PANMIX)
	{
		/* Over several contexts the root voice's chain is mirrored in each of them, and the mirrors
		 * are addressed by kind (root_uid): that covers the engine's own root driver, inline + panmix
		 * (+ xinsert, which stays with the engine).  A custom root program with anything else in it
		 * would need a unit id per context: refused, not approximated. */
		fprintf(stderr, "a2amd units: the root voice runs '%s': with A2AMD_DEVICES > 1 only the stock root "
				"driver (inline, panmix, xinsert) is supported\n", u->descriptor->name);
		return A2P_NOTIMPLEMENTED;
	}
	if(forward && hs->ndev > 1 && !x->is_root)
	{
		/* which GPU?  known when the voice is first processed (route_voice) */
		BIRTHOP *b = new_birthop(hs);
		if(!b)
			return A2P_OOMEMORY;
		x->is_head = !hs->chain_last;
		x->pending = 1;
		b->u = u;
		b->vms = vms;
		b->kind = kind;
		b->flags = lflags;
		b->nin = u->ninputs;
		b->nout = u->noutputs;
		b->wired = wired;
		b->transpose = vms->r[A2P_R_TRANSPOSE];
		b->wakefrac = vms->waketime & 0xff;
	}
	else if(forward)
	{
		int d;
		x->is_head = !hs->chain_last && !x->is_root;
		/* (the root voice's units exist in every context: each renders its subtrees
		 * into its own partial of the root's bus) */
		for(d = x->is_root ? hs->ndev - 1 : 0; d >= 0; --d)
		{
			x->uid = a2amd_unit_init(hs->ctxs[d], (uint64_t)(uintptr_t)vms, kind, lflags, u->ninputs,
					u->noutputs, wired, vms->r[A2P_R_TRANSPOSE], vms->waketime & 0xff);
			if(x->uid < 0)
			{
				fprintf(stderr, "a2amd units: cannot instantiate unit kind %d: %s\n", kind,
						a2amd_last_error(hs->ctxs[d]));
				return A2P_NOTIMPLEMENTED;	/* the engine reports A2_VOICEINIT and drops the voice */
			}
			if(x->is_root && (kind == A2AMD_INLINE || kind == A2AMD_PANMIX))
				hs->root_uid[kind == A2AMD_PANMIX][d] = x->uid;
		}
	}
	if(forward)
	{
		if(!hs->chain_last)
		{
			/* (the env units the chain started with are this head's) */
			int k;
			hs->chain_head = u;
			x->nenv = hs->chain_nenv;
			for(k = 0; k < hs->chain_nenv && k < 2; ++k)
			{
				x->env[k] = hs->chain_env[k];
				x->env_before[k] = 0;
			}
		}
		++hs->chain_nfwd;
	}
	hs->chain_last = u;
	return 0;
}

static void amd_deinit(A2P_unit *u)
{
	{
		/* a unit of the voice whose chain was populated last goes away: the next voice may be handed the
		 * same A2_voice - what is known about "the chain being populated" (chain_last, the env units seen,
		 * the head) must not survive */
		XTRA *xc = xtra(u);
		if(xc->hs && xc->hs->chain_vms == xc->vms)
		{
			xc->hs->chain_vms = NULL;
			xc->hs->chain_last = NULL;
			xc->hs->chain_head = NULL;
			xc->hs->chain_nfwd = xc->hs->chain_nenv = 0;
		}
	}
	if(u->descriptor == &a2_env_unitdesc)
		return;
	if(is_ours(u->descriptor) && u->descriptor != &a2_inline_unitdesc && u->descriptor != &a2_xinsert_unitdesc &&
			u->descriptor != &a2_xsink_unitdesc && u->descriptor != &a2_xsource_unitdesc)
	{
		/* (the head of a voice the device VM runs: the backend takes the voice back itself) */
		XTRA *xv = (XTRA *)((char *)u + 64);
		if(xv->vm && xv->head == u && xv->hs)
		{
			xv->vm = 0;
			if(xv->hs->vm_live)
				--xv->hs->vm_live;
		}
	}
	XTRA *x = xtra(u);
	int rc;
	if(x->hs)
		memset(x->hs->walk_ring, 0, sizeof(x->hs->walk_ring));
	if(!x->hs)
		return;
	if(x->hs->chain_last == u)
		x->hs->chain_last = NULL;
	if(x->head == u)
		set_stamp(x->hs, x, 0);		/* (head of a chain of our own: its slot goes back to the backend) */
	x->hs->noise_oscs -= x->is_noise;
	x->is_noise = 0;
	if(x->pending)
	{
		/* never processed: nothing of it exists on any GPU */
		HOSTSTATE *hs = x->hs;
		int k, n = 0;
		for(k = 0; k < hs->nbirths; ++k)
			if(hs->births[k].u != u)
				hs->births[n++] = hs->births[k];
		hs->nbirths = n;
		x->pending = 0;
		return;
	}
	if(x->is_root && x->hs->ndev > 1 && (x->kind == A2AMD_INLINE || x->kind == A2AMD_PANMIX))
	{
		int d;
		for(d = 0; d < x->hs->ndev; ++d)
			if(!x->hs->failed && (rc = a2amd_unit_deinit(x->hs->ctxs[d], x->hs->root_uid[x->kind == A2AMD_PANMIX][d])))
				fail(x->hs, "a2amd_unit_deinit", rc);
		return;
	}
	if(x->hs->ndev > 1 && x->hs->load[x->dev] > 0)
		--x->hs->load[x->dev];
	if(x->uid >= 0 && !x->hs->failed && (rc = a2amd_unit_deinit(XCTX(x), x->uid)))
		fail(x->hs, "a2amd_unit_deinit", rc);
}

/* ---- Process ---------------------------------------------------------------------*/
/* A wave released by the application keeps its A2_wave for one more engine
 * cycle with size[0] = 0 ("unloaded", a2_discard_wave, waves.c:711-718: set
 * under the engine lock, i.e. between two buffers; freed by an end-of-cycle
 * event once every state has processed a buffer).  The engine's wtosc looks at
 * that at the top of every Process call (wtosc.c:168-183), which only matters
 * for oscillators playing the wave; the registry below must forget the object in
 * that one cycle whether anybody plays it or not - the allocator hands the same
 * address to the next wave - so it is swept once per buffer, before the first
 * voice is walked. */
static void sweep_waves(HOSTSTATE *hs)
{
	int i, rc;
	if(hs->npendcaps)
	{
		/* a rendered wave that was released before anybody played it: its capture goes with it */
		pthread_mutex_lock(&hs->pendcaps_mtx);
		for(i = 0; i < hs->npendcaps; )
			/* (by handle, not through the remembered pointer: a state that sat out the cycle in which the released
			 * wave lingered would read freed memory - and find a later wave at the same address) */
			if(a2_GetWave(hs->cfg->interface, hs->pendcaps[i].handle) != hs->pendcaps[i].w || !hs->pendcaps[i].w->size[0])
			{
				a2amd_capture_free(hs->pendcaps[i].cap);
				hs->pendcaps[i] = hs->pendcaps[--hs->npendcaps];
			}
			else
				++i;
		pthread_mutex_unlock(&hs->pendcaps_mtx);
	}
	for(i = 0; i < hs->nwaves; )
	{
		A2P_wave *w = hs->wave_ptr[i];
		if((w->type == A2AMD_WWAVE || w->type == A2AMD_WMIPWAVE) && !w->size[0])
		{
			int d;
			for(d = 0; d < hs->ndev; ++d)
				if(hs->wave_id[i][d] >= 0 && !hs->failed &&
						(rc = a2amd_wave_drop(hs->ctxs[d], (uint64_t)(uintptr_t)w)))
					fail(hs, "a2amd_wave_drop", rc);
			hs->wave_ptr[i] = hs->wave_ptr[hs->nwaves - 1];
			memcpy(hs->wave_id[i], hs->wave_id[hs->nwaves - 1], sizeof(hs->wave_id[i]));
			--hs->nwaves;
		}
		else
			++i;
	}
}

/* Units of the engine (or of the application, a2_RegisterUnit) that are NOT
 * replaced render on the CPU into buffers the GPU never sees.  One with audio
 * ports in the middle of a replaced chain would silently process silence, so
 * it is refused, loudly, the first time the chain runs.  (Units without audio
 * ports - the engine's env - are fine: they only write control registers.) */
static int is_ours(const A2P_unitdesc *d)
{
	static const A2P_unitdesc *const ours[] = {
		&a2_wtosc_unitdesc, &a2_panmix_unitdesc, &a2_filter12_unitdesc, &a2_fbdelay_unitdesc,
		&a2_inline_unitdesc, &a2_xinsert_unitdesc, &a2_fm1_unitdesc, &a2_fm2_unitdesc,
		&a2_fm3_unitdesc, &a2_fm4_unitdesc, &a2_fm3p_unitdesc, &a2_fm4p_unitdesc,
		&a2_fm2r_unitdesc, &a2_fm4r_unitdesc, &a2_dc_unitdesc, &a2_waveshaper_unitdesc,
		&a2_dcblock_unitdesc, &a2_limiter_unitdesc, &a2_xsink_unitdesc, &a2_xsource_unitdesc, &a2_env_unitdesc };
	unsigned i;
	for(i = 0; i < sizeof(ours) / sizeof(ours[0]); ++i)
		if(d == ours[i])
			return 1;
	return 0;
}

static void check_chain_behind(A2P_unit *u)
{
	const A2P_unit *n;
	XTRA *x = xtra(u);

	for(n = u->next; n && (!is_ours(n->descriptor) || n->descriptor == &a2_env_unitdesc); n = n->next)
		if(n->descriptor->maxinputs || n->descriptor->maxoutputs)	/* (the unit's own counts are
				not meaningful for a port-less unit: the engine still hands env one) */
		{
			/* reported once per voice through the engine's error channel; that
			 * unit keeps processing the silence in its CPU buffers */
			char *msg = x->hs->chainmsg;	/* (per state: a2r_Error keeps the pointer) */
			snprintf(msg, sizeof(x->hs->chainmsg), "a2amd: unit '%s' sits behind a GPU-rendered '%s' in one voice "
					"but is not replaced: it processes silence (mixed CPU/GPU chains are "
					"not supported, no CPU fallback)", n->descriptor->name, u->descriptor->name);
			if(a2r_Error && x->hs->engine_state)
				a2r_Error(x->hs->engine_state, A2P_NOTIMPLEMENTED, msg);
			else
				fprintf(stderr, "a2amd units: %s\n", msg);
			return;
		}
}

static void client_error(A2P_xinsert *xi, int res, const char *info)
{
	if(a2r_Error)
		a2r_Error(xi->state, res, info);
	else
		fprintf(stderr, "a2amd units: %s: error %d\n", info, res);
}

static PENDING *new_pending(HOSTSTATE *hs)
{
	if(hs->npend == hs->cap_pend)
	{
		int nc = hs->cap_pend ? hs->cap_pend * 2 : 256;
		PENDING *np = (PENDING *)realloc(hs->pend, nc * sizeof(PENDING));
		if(!np)
			return NULL;
		hs->pend = np;
		hs->cap_pend = nc;
	}
	memset(&hs->pend[hs->npend], 0, sizeof(PENDING));
	return &hs->pend[hs->npend++];
}

/* xi_process (src/units/xinsert.c:60-142) for an xinsert whose audio is on the
 * GPU.  WRITE-only clients (a2_SourceCallback, a2_OpenSource) produce audio
 * without looking at any: they run here, in the walk, and their sum travels
 * with the batch.  READ-only clients (a2_SinkCallback, a2_OpenSink) are handed
 * the unit's input, which exists once the fragment has been rendered: their
 * windows are noted and served, in walk order, when the audio comes back
 * (deliver_pending).  An insert client (READ and WRITE) would need its voice's
 * audio on the host in the middle of the GPU batch: reported through the
 * engine's error channel, once, and not served (it sees nothing, the voice's
 * signal passes through as if it were not there). */
static void serve_clients(XTRA *x, A2P_xinsert *xi, unsigned offset, unsigned frames)
{
	HOSTSTATE *hs = x->hs;
	A2P_xinsert_client *xic;
	A2P_unit *xu = &xi->header;
	const int nch = x->kind == A2AMD_XSOURCE ? xu->noutputs : xu->ninputs;
	const char *what = x->kind == A2AMD_XSOURCE ? "xsource client callback" :
			x->kind == A2AMD_XSINK ? "xsink client callback" : "xinsert client callback";
	unsigned mode = 0;
	int rc, i, idepth = 0;
	unsigned s;
	if(hs->failed)
		return;
	/* xsink hands every client its inputs (xsink.c:41-44), xsource adds up every
	 * client's output (xsource.c:69-77); xinsert looks at the client's flags */
	for(xic = xi->clients; xic; xic = xic->next)
		if(x->kind == A2AMD_XSINK)
			mode |= A2AMD_XIO_TAP;
		else if(x->kind == A2AMD_XSOURCE)
			mode |= A2AMD_XIO_INJECT;
		else if((xic->flags & A2P_XI_READ) && (xic->flags & A2P_XI_WRITE))
		{
			/* an insert client: where the render can pause for it (the last unit of a voice
			 * below the root: a2_NewGroup's xinsert), its input is tapped and not passed on,
			 * and render_batch() runs it before the parent voice's chain is rendered */
			if((idepth = a2amd_unit_insertable(XCTX(x), x->uid)) >= 1)
				mode |= A2AMD_XIO_TAP | A2AMD_XIO_MUTE;
			else if(!x->refused)
			{
				x->refused = 1;
				client_error(xi, A2P_NOTIMPLEMENTED, "a2amd: insert client (a2_InsertCallback) on an xinsert "
						"that is not the last unit of its voice: "
						"its audio is on the GPU; not served (sink and source clients are)");
			}
		}
		else
			mode |= (xic->flags & A2P_XI_WRITE) ? A2AMD_XIO_INJECT : A2AMD_XIO_TAP;
	if(mode != x->client_mode)
	{
		if((rc = a2amd_unit_clients(XCTX(x), x->uid, mode)))
		{
			fail(hs, "a2amd_unit_clients", rc);
			return;
		}
		x->client_mode = mode;
	}
	if(mode & A2AMD_XIO_INJECT)
	{
		/* (the reference hands WRITE-only clients uninitialised stack arrays,
		 * xinsert.c:66, xsource.c:50, and mixes all of them into the output,
		 * xinsert.c:113-118; ours are cleared, so a client that fills only some
		 * channels - the source streams do - adds silence to the others, not
		 * stack contents) */
		int32_t sum[A2AMD_MAXCHANNELS][A2AMD_MAXFRAG], tmp[A2AMD_MAXCHANNELS][A2AMD_MAXFRAG];
		int32_t *bufp[A2AMD_MAXCHANNELS];
		const int32_t *sump[A2AMD_MAXCHANNELS];
		memset(sum, 0, sizeof(sum));
		for(xic = xi->clients; xic; xic = xic->next)
		{
			if(x->kind == A2AMD_XINSERT && ((xic->flags & (A2P_XI_READ | A2P_XI_WRITE)) != A2P_XI_WRITE))
				continue;
			memset(tmp, 0, sizeof(tmp));
			for(i = 0; i < nch; ++i)
				bufp[i] = tmp[i];
			if((rc = xic->callback(bufp, nch, frames, xic->userdata)))
				client_error(xi, rc, what);
			for(i = 0; i < nch; ++i)
				for(s = 0; s < frames; ++s)
					sum[i][s] = (int32_t)((uint32_t)sum[i][s] + (uint32_t)tmp[i][s]);
		}
		for(i = 0; i < nch; ++i)
			sump[i] = sum[i];
		if((rc = a2amd_unit_inject(XCTX(x), x->uid, offset - hs->base, frames, sump)))
			fail(hs, "a2amd_unit_inject", rc);
	}
	if(mode & A2AMD_XIO_TAP)
		for(xic = xi->clients; xic; xic = xic->next)
		{
			PENDING *p;
			const int ins = x->kind == A2AMD_XINSERT && (xic->flags & A2P_XI_READ) && (xic->flags & A2P_XI_WRITE);
			if(x->kind == A2AMD_XINSERT && (xic->flags & A2P_XI_WRITE) && !(ins && (mode & A2AMD_XIO_MUTE)))
				continue;
			if(!(p = new_pending(hs)))
			{
				fail(hs, "out of memory for sink client windows", -1);
				return;
			}
			p->xi = xi;
			p->xic = xic;
			p->uid = x->uid;
			p->dev = x->dev;
			p->frag = hs->batch_frags ? (int)hs->batch_frags - 1 : 0;
			p->pos = 0;
			p->what = what;
			p->offset = offset - hs->base;
			p->frames = frames;
			p->insert = ins;
			p->depth = idepth;
			hs->ninserts += ins;
		}
}

/* Audio is back: hand the READ clients their windows, in walk order. */
static void deliver_pending(HOSTSTATE *hs)
{
	int k, i, n, rc;
	for(k = 0; k < hs->npend; ++k)
	{
		PENDING *p = &hs->pend[k];
		const int32_t *bufs[A2AMD_MAXCHANNELS];
		int32_t *bufp[A2AMD_MAXCHANNELS];
		A2P_xinsert_client *c;
		/* (the engine may have removed the client since; its unit, or the
		 * stand-in xi_deinit left for it, is alive) */
		for(c = p->xi->clients; c && c != p->xic; c = c->next)
			;
		if(!c || hs->failed || p->insert)
			continue;
		if(p->uid < 0)
		{
			/* the root xinsert's input is the master bus as rendered */
			n = p->xi->header.ninputs;
			for(i = 0; i < n; ++i)
				bufp[i] = hs->acc[i] + p->pos;
		}
		else
		{
			if((n = a2amd_unit_tapped(hs->ctxs[p->dev], p->uid, (unsigned)p->frag, bufs)) < 0)
			{
				fail(hs, "a2amd_unit_tapped", n);
				continue;
			}
			for(i = 0; i < n; ++i)
				bufp[i] = (int32_t *)bufs[i] + p->offset;
		}
		if((rc = c->callback(bufp, n, p->frames, c->userdata)))
			client_error(p->xi, rc, p->what);
	}
	hs->npend = 0;
	for(k = 0; k < hs->nzombies; ++k)
	{
		/* what xi_Deinitialize (xinsert.c:204-211) would have done */
		A2P_xinsert *z = hs->zombies[k];
		while(z->clients)
			a2_XinsertRemoveClient(z->clients);
		free(z);
	}
	hs->nzombies = 0;
}

/* ---- one GPU round trip per driver buffer ----------------------------------------
 * The engine's per-buffer entry point (a2_AudioCallback, src/core.c:1927-2001)
 * cuts a buffer into fragments of at most 64 frames, walks the voice tree once
 * per fragment and copies the master bus into A2_audiodriver.buffers.  Nothing it
 * or the VM reads during that walk is produced by audio rendering (the noise
 * oscillators' draw counts and filter coefficients are mirrored on the host), so
 * the walk of the whole buffer can be RECORDED first and rendered afterwards: our
 * Process, installed in front of the engine's, lets the engine walk, then renders
 * the recorded batch (up to 256 fragments per launch sequence, more than one
 * sequence for longer buffers) and writes the result where a2_ProcessMaster
 * (core.c:1900-1907) would have.  Sink clients get their windows after that, in
 * walk order (they cannot tell: they run inside the same a2_Run()).  The one
 * thing that cannot be deferred is an INSERT client on the root xinsert - it
 * transforms the master bus on the CPU window by window - so a buffer that starts
 * with one attached is rendered the old way, one round trip per root window. */
static int root_has_inserts(HOSTSTATE *hs)
{
	A2P_xinsert_client *c;
	if(!hs->root_xi)
		return 0;
	for(c = hs->root_xi->clients; c; c = c->next)
		if((c->flags & A2P_XI_READ) && (c->flags & A2P_XI_WRITE))
			return 1;
	return 0;
}

static int grow_acc(HOSTSTATE *hs, unsigned frames)
{
	int c;
	if(frames <= hs->acc_cap)
		return 1;
	for(c = 0; c < hs->cfg->channels && c < A2AMD_MAXCHANNELS; ++c)
	{
		int32_t *a = (int32_t *)realloc(hs->acc[c], frames * sizeof(int32_t));
		int32_t *r = a ? (int32_t *)realloc(hs->rinj[c], frames * sizeof(int32_t)) : NULL;
		if(a)
			hs->acc[c] = a;
		if(r)
			hs->rinj[c] = r;
		if(!a || !r)
			return 0;
	}
	hs->acc_cap = frames;
	return 1;
}

/* The insert clients' turn, between the two halves of the render: each is handed a copy
 * of its window of the unit's input and what it makes of it is summed up as the unit's
 * output (xi_process, xinsert.c:95-118), in walk order. */
static void deliver_inserts(HOSTSTATE *hs, int depth, int dev)
{
	int k, i, n, rc;
	unsigned s;
	for(k = 0; k < hs->npend; ++k)
	{
		PENDING *p = &hs->pend[k];
		const int32_t *bufs[A2AMD_MAXCHANNELS];
		int32_t tmp[A2AMD_MAXCHANNELS][A2AMD_MAXFRAG];
		int32_t *bufp[A2AMD_MAXCHANNELS];
		const int32_t *outp[A2AMD_MAXCHANNELS];
		A2P_xinsert_client *c;
		if(!p->insert || p->depth != depth || p->dev != dev || hs->failed)
			continue;
		for(c = p->xi->clients; c && c != p->xic; c = c->next)
			;
		if(!c)
			continue;
		if((n = a2amd_unit_tapped(hs->ctxs[p->dev], p->uid, (unsigned)p->frag, bufs)) < 0)
		{
			fail(hs, "a2amd_unit_tapped", n);
			continue;
		}
		for(i = 0; i < n; ++i)
		{
			for(s = 0; s < p->frames; ++s)
				tmp[i][s] = bufs[i][p->offset + s];
			bufp[i] = tmp[i];
			outp[i] = tmp[i];
		}
		if((rc = c->callback(bufp, n, p->frames, c->userdata)))
			client_error(p->xi, rc, p->what);
		if((rc = a2amd_unit_insert(hs->ctxs[p->dev], p->uid, (unsigned)p->frag, p->offset, p->frames, outp)))
			fail(hs, "a2amd_unit_insert", rc);
	}
}

/* Before a render: when the root's VM wakes next - where the engine will cut the NEXT buffer's fragments
 * (a2_VoiceProcess, core.c:1852-1878).  An attached root that has reached END (A2_ENDING) wakes every 1 000 000 ticks
 * (OP_END, core.c:1191-1217); one that waits in a delay of its own loop, at the distance of its last two wake-ups - a
 * guess, and a wrong one costs the backend nothing but a speculative pass it does not take. */
static void hint_cuts(HOSTSTATE *hs)
{
	uint32_t when[16];
	int n = 0, d;
	const A2P_vmstate *r = hs->root_vms;
	if(r && (r->state == 3 /* A2_ENDING */ || r->state == 1 /* A2_WAITING */))
	{
		const unsigned period = r->state == 3 ? 1000000u : hs->root_period;
		unsigned w = r->waketime;
		when[n++] = w;
		while(period >= (64u << 8) && n < 16)
			when[n++] = (w += period);
	}
	for(d = 0; d < hs->ndev; ++d)
		if(!hs->failed)
			a2amd_vm_expect_cuts(hs->ctxs[d], when, n);
}

/* the batch recorded so far -> audio (outp[channel], at most cap frames) */
static int render_batch(HOSTSTATE *hs, int32_t **outp, unsigned cap)
{
	int n;
	hint_cuts(hs);
	if(!hs->ninserts)
		return a2amd_render_group(hs->ctxs, hs->ndev, A2AMD_RENDER_ALL, outp, cap);
	hs->ninserts = 0;
	{
		/* every context renders its subtrees; the render pauses behind every nesting depth that holds
		 * insert clients, deepest first */
		int d;
		for(d = 0; d < hs->ndev; ++d)
		{
			if((n = a2amd_render(hs->ctxs[d], A2AMD_RENDER_UPLOAD | A2AMD_RENDER_SUBTREES | A2AMD_RENDER_TAPS, NULL, 0)) < 0)
				return n;
			while((n = a2amd_render_paused(hs->ctxs[d])) > 0)
			{
				deliver_inserts(hs, n, d);
				if(n == 1)
					break;
				if((n = a2amd_render(hs->ctxs[d], A2AMD_RENDER_SUBTREES | A2AMD_RENDER_TAPS, NULL, 0)) < 0)
					return n;
			}
		}
	}
	if(hs->ndev > 1)	/* (what the clients of voices right below the root wrote, the exchange, the root chain) */
		return a2amd_render_group(hs->ctxs, hs->ndev, A2AMD_RENDER_EXCHANGE | A2AMD_RENDER_ROOT | A2AMD_RENDER_READBACK, outp, cap);
	return a2amd_render(hs->ctx, A2AMD_RENDER_ROOT | A2AMD_RENDER_READBACK, outp, cap);
}

/* the part of the buffer that was handed over early (flush_part): its audio into its place */
static void collect_part(HOSTSTATE *hs)
{
	int32_t *outp[A2AMD_MAXCHANNELS];
	int c, n;
	if(!hs->part_frames)
		return;
	for(c = 0; c < A2AMD_MAXCHANNELS; ++c)
		outp[c] = hs->acc[c] ? hs->acc[c] + hs->part_at : NULL;
	if(!hs->failed && (n = a2amd_collect(hs->ctx, outp, hs->part_frames)) != (int)hs->part_frames)
		fail(hs, "a2amd_collect", n);
	if(hs->failed)
		for(c = 0; c < hs->cfg->channels; ++c)
			memset(hs->acc[c] + hs->part_at, 0, hs->part_frames * sizeof(int32_t));
	hs->part_frames = 0;
}

static void flush_batch(HOSTSTATE *hs);

/* Round 5: a long buffer (a2play's 4 096 frames) is not rendered in ONE round trip behind the engine's walk
 * of all of it - the GPU idle while the engine walks, the engine idle while the GPU renders - but in two: what
 * has been recorded by the middle of the buffer is launched without waiting (A2AMD_RENDER_ASYNC), the engine
 * walks the second half beside it, and the buffer's end renders the rest and collects both.  Only where nothing
 * has to come back in between: one context, no sink / insert clients waiting for their windows. */
static void flush_part(HOSTSTATE *hs)
{
	int n;
	if(!hs->batch_frags)
		return;
	if(hs->failed || hs->ndev != 1 || hs->ninserts || hs->npend || hs->nzombies || hs->part_frames)
	{
		flush_batch(hs);
		return;
	}
	hint_cuts(hs);
	n = a2amd_render(hs->ctx, A2AMD_RENDER_ALL | A2AMD_RENDER_ASYNC, NULL, 0);
	if(n == A2AMD_EUNSUPPORTED)
	{
		/* (READ clients attached somewhere: their taps have to come back with the audio) */
		hs->split_frames = 0;
		flush_batch(hs);
		return;
	}
	if(n != (int)(hs->rec_pos - hs->acc_pos))
		fail(hs, "a2amd_render", n);
	hs->part_at = hs->acc_pos;
	hs->part_frames = hs->failed ? 0 : hs->rec_pos - hs->acc_pos;
	if(hs->failed)
	{
		int c;
		for(c = 0; c < hs->cfg->channels; ++c)
			memset(hs->acc[c] + hs->acc_pos, 0, (hs->rec_pos - hs->acc_pos) * sizeof(int32_t));
	}
	hs->acc_pos = hs->rec_pos;
	hs->batch_frags = 0;
}

/* render what has been recorded of the buffer so far */
static void flush_batch(HOSTSTATE *hs)
{
	int32_t *outp[A2AMD_MAXCHANNELS];
	int c, n;
	if(!hs->batch_frags)
	{
		collect_part(hs);
		return;
	}
	for(c = 0; c < A2AMD_MAXCHANNELS; ++c)
		outp[c] = hs->acc[c] ? hs->acc[c] + hs->acc_pos : NULL;
	if(!hs->failed)
	{
		n = render_batch(hs, outp, hs->acc_cap - hs->acc_pos);
		if(n != (int)(hs->rec_pos - hs->acc_pos))
			fail(hs, "a2amd_render", n);
	}
	if(hs->failed)
		for(c = 0; c < hs->cfg->channels; ++c)
			memset(hs->acc[c] + hs->acc_pos, 0, (hs->rec_pos - hs->acc_pos) * sizeof(int32_t));
	hs->acc_pos = hs->rec_pos;
	hs->batch_frags = 0;
	collect_part(hs);	/* (rendered before this batch: its copy is done) */
	if(hs->npend || hs->nzombies)
		deliver_pending(hs);
}

static HOSTSTATE *state_of_config(A2P_config *cfg)
{
	HOSTSTATE *hs;
	for(hs = states; hs; hs = hs->next_state)
		if(hs->refs && hs->cfg == cfg)
			return hs;
	return NULL;
}

/* A2AMD_HOSTTIMING=1: where a buffer's time goes (the engine's walk incl. our
 * recording callbacks / the GPU round trip), printed when the process exits */
static double t_walk, t_flush, n_buffers;
static int timing_on = -1;
static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}
static void timing_dump(void)
{
	if(n_buffers)
		fprintf(stderr, "a2amd units: per driver buffer: engine walk + recording %.1f us, render + delivery %.1f us "
				"(%g buffers)\n", t_walk / n_buffers * 1e6, t_flush / n_buffers * 1e6, n_buffers);
}

static void amd_drv_process(A2P_audiodriver *drv, unsigned frames)
{
	HOSTSTATE *hs = state_of_config(drv->driver.config);
	int c;
	unsigned s;
	double t0 = 0, t1 = 0;
	if(timing_on < 0)
	{
		timing_on = getenv("A2AMD_HOSTTIMING") ? atoi(getenv("A2AMD_HOSTTIMING")) : 0;
		if(timing_on)
			atexit(timing_dump);
	}
	if(!hs || !hs->drv_process)
		return;
	hs->in_buffer = 1;
	hs->decided = hs->batching = hs->swept = 0;
	hs->rec_pos = hs->win_pos = hs->acc_pos = 0;
	hs->batch_frags = 0;
	hs->rinj_used = 0;
	hs->part_frames = 0;
	{
		/* (A2AMD_SPLIT=0: the whole buffer in one round trip, as rounds 2-4 did; n: from n frames on) */
		static int split_min = -1;
		if(split_min < 0)
			split_min = getenv("A2AMD_SPLIT") ? atoi(getenv("A2AMD_SPLIT")) : 0;
		hs->split_frames = split_min > 0 && frames >= (unsigned)split_min && hs->ndev == 1 ? (frames / 2 + 63) / 64 * 64 : 0;
	}
	if(!grow_acc(hs, frames))
		hs->decided = 1;	/* (out of memory: this buffer the old way) */
	if(timing_on)
		t0 = now_s();
	hs->drv_process(drv, frames);	/* a2_AudioCallback: the engine walks, our units record */
	if(timing_on)
		t1 = now_s();
	if(hs->batching)
	{
		flush_batch(hs);
		for(c = 0; c < hs->cfg->channels; ++c)
		{
			if(hs->rinj_used)
				for(s = 0; s < hs->acc_pos; ++s)
					hs->acc[c][s] = (int32_t)((uint32_t)hs->acc[c][s] + (uint32_t)hs->rinj[c][s]);
			memcpy(drv->buffers[c], hs->acc[c], hs->acc_pos * sizeof(int32_t));
		}
	}
	hs->in_buffer = hs->batching = 0;
	if(timing_on)
	{
		const double t2 = now_s();
		t_walk += t1 - t0;
		t_flush += t2 - t1;
		n_buffers += 1;
		if(timing_on >= 2)
			fprintf(stderr, "a2amd buffer: %u frames, walk %.1f us, render + delivery %.1f us\n", frames,
					(t1 - t0) * 1e6, (t2 - t1) * 1e6);
	}
}

/* put our Process in front of the engine's (which a2_Open installs after the
 * units' states are open, src/audiality2.c:507-511): done from inside the first
 * buffer the engine processes, which is itself rendered window by window */
static void hook_driver(HOSTSTATE *hs)
{
	A2P_driver *d;
	if(hs->drv || hs->max_batch <= 1)
		return;
	for(d = (A2P_driver *)hs->cfg->drivers; d; d = d->next)
		if(d->type == A2P_AUDIODRIVER)
		{
			A2P_audiodriver *ad = (A2P_audiodriver *)d;
			if(!ad->Process || ad->Process == amd_drv_process)
				return;
			hs->drv = ad;
			hs->drv_process = ad->Process;
			ad->Process = amd_drv_process;
			return;
		}
}

static void forward_process(XTRA *x, unsigned offset, unsigned frames)
{
	HOSTSTATE *hs = x->hs;
	uint32_t noise = 0, before = 0;
	int rc, v = 0, is_noise = 0;
	if(hs->failed)
		return;
	if(x->kind == A2AMD_WTOSC)
		is_noise = x->is_noise;
	if(is_noise)
	{
		/* the engine-global RNG the noise oscillators share with the VM's
		 * RAND instructions (internals.h:682), through the public property
		 * (only oscillators that play the noise wave draw from it) */
		a2_GetStateProperty(hs->cfg->interface, A2P_PNOISESEED, &v);
		noise = before = (uint32_t)v;
	}
	if(x->is_root && hs->ndev > 1)
	{
		/* the root voice's units are walked in every context */
		int d;
		for(d = 0; d < hs->ndev && !hs->failed; ++d)
			if((rc = a2amd_unit_process(hs->ctxs[d], hs->root_uid[x->kind == A2AMD_PANMIX][d], offset - hs->base,
					frames, NULL)))
				fail(hs, "a2amd_unit_process", rc);
	}
	else if((rc = a2amd_unit_process(XCTX(x), x->uid, offset - hs->base, frames, is_noise ? &noise : NULL)))
		fail(hs, "a2amd_unit_process", rc);
	if(noise != before)
		a2_SetStateProperty(hs->cfg->interface, A2P_PNOISESEED, (int)noise);
}

/* ---- the walk's hot path ---------------------------------------------------------------
 * a2_VoiceProcess calls every unit of every voice once per window (core.c:1875-1876);
 * with tens of thousands of voices whose VMs sleep, that call - and the memory it
 * touches - is all the engine thread does.  For a voice whose chain consists of
 * replaced units only ("simple": no inline, no x-units, no unit of the engine's or the
 * application's in between), the first unit's Process speaks for the whole chain
 * (a2amd_voice_process) and the others' are empty; once the backend agrees that the
 * voice is at rest (no noise, no ramping pitch or cutoff), the first unit's Process
 * shrinks to one byte store into the backend's default map per fragment
 * (amd_quick_process) until a register of the voice is written again. */
static void amd_head_process(A2P_unit *u, unsigned offset, unsigned frames);
static int env_busy(const XTRA *head);
static int env_chain_ok(A2P_unit *head);
static void head_process(A2P_unit *u, XTRA *x, HOSTSTATE *hs, unsigned offset, unsigned frames);

static void amd_noop(A2P_unit *u, unsigned offset, unsigned frames)
{
}

/* With tens of thousands of voices the walk is DRAM latency: the engine chases v->next,
 * v->units and u->next through blocks it allocated one by one (a sampling profiler finds
 * 40 % of the thread at the first instruction of amd_noop: the u->Process load through
 * u->next, core.c:1875-1876).  The order of the walk is the same fragment after fragment
 * while no voice comes or goes, so every unit of ours that does anything in its Process -
 * the heads of our own chains, the units of group voices - remembers which of them the
 * engine called walk_ahead calls later last time, and asks for that one's lines now.
 * Hints only: a stale pointer is never dereferenced, and the ring is emptied whenever a
 * unit is destroyed (the XTRA it points at would be a freed block). */
static inline void look_ahead(HOSTSTATE *hs, XTRA *x, A2P_unit *u)
{
	XTRA *t = hs->walk_ring[hs->walk_pos];
	/* (bit 0: the unit keeps its XTRA at the end of its block, xtra()) */
	A2P_unit *me = (A2P_unit *)((uintptr_t)u | ((char *)x != (char *)u + 64));
	if(t && t->ahead != me && t != x)
	{
		t->ahead = me;
		t->ahead_tail[0] = x->tail[0];
		t->ahead_tail[1] = x->tail[1];
		t->ahead_vms = x->vms;
	}
	hs->walk_ring[hs->walk_pos] = x;
	if(++hs->walk_pos >= (unsigned)hs->walk_ahead)	/* (no division on this path) */
		hs->walk_pos = 0;
	if(x->ahead)
	{
		/* (blocks and voices come from malloc, 16 byte aligned: the 64 bytes of an
		 * A2_unit - next first, Process last - usually lie on two cache lines)
		 * A2_voice as the engine lays it out (internals.h:559-586): next / events
		 * in the 32 bytes before the VM state, whose first word is the wake time;
		 * flags behind the registers; units / sub behind the register write table */
		const char *h = (const char *)((uintptr_t)x->ahead & ~(uintptr_t)1);
		const char *hx = ((uintptr_t)x->ahead & 1) ? h + A2P_BLOCK_SIZE - sizeof(XTRA) : h + 64;
		const char *v = (const char *)x->ahead_vms - 4 * sizeof(void *);
		const size_t flags = 4 * sizeof(void *) + sizeof(A2P_vmstate) + 4;
		const size_t units = flags + 4 + 64 * 2 * sizeof(void *);
		int k;
		__builtin_prefetch(h);
		__builtin_prefetch(h + 56);
		__builtin_prefetch(hx);			/* ... and what we keep of it */
		__builtin_prefetch(hx + offsetof(XTRA, head) - 1);
		for(k = 0; k < 2; ++k)
		{
			__builtin_prefetch(x->ahead_tail[k]);
			__builtin_prefetch((const char *)x->ahead_tail[k] + 56);
		}
		__builtin_prefetch(v);
		__builtin_prefetch(x->ahead_vms);
		__builtin_prefetch(v + flags);
		__builtin_prefetch(v + units);
		__builtin_prefetch(v + units + 15);
	}
}

static void amd_quick_process(A2P_unit *u, unsigned offset, unsigned frames)
{
	XTRA *x = (XTRA *)((char *)u + 64);	/* (own units only: no descriptor look-up) */
	HOSTSTATE *hs = x->hs;
	if(hs->walk_ahead && !hs->walker)
		look_ahead(hs, x, u);
	if(offset == hs->base && frames == hs->win_frames && (unsigned)x->slot < hs->map_cap[x->dev])
	{
		hs->map[x->dev][x->slot] = 1;	/* Process(0, all frames) on each unit: the default */
		return;
	}
	u->Process = amd_head_process;
	set_stamp(hs, x, 0);
	head_process(u, x, hs, offset, frames);
}

static void amd_head_process(A2P_unit *u, unsigned offset, unsigned frames)
{
	XTRA *x = (XTRA *)((char *)u + 64);
	HOSTSTATE *hs = x->hs;
	if(hs->walk_ahead && !hs->walker)
		look_ahead(hs, x, u);
	head_process(u, x, hs, offset, frames);
}

static void head_process(A2P_unit *u, XTRA *x, HOSTSTATE *hs, unsigned offset, unsigned frames)
{
	uint32_t noise = 0, before = 0;
	int rc, v = 0;
	if(hs->failed)
		return;
	if(hs->noise_oscs)
	{
		/* (the engine-global RNG of the noise oscillators, internals.h:682; only
		 * while some oscillator of this state plays the noise wave) */
		a2_GetStateProperty(hs->cfg->interface, A2P_PNOISESEED, &v);
		noise = before = (uint32_t)v;
	}
	rc = a2amd_voice_process(XCTX(x), x->uid, offset - hs->base, frames, hs->noise_oscs ? &noise : NULL);
	if(rc < 0)
		fail(hs, "a2amd_voice_process", rc);
	else if(rc == 1 && !hs->no_quick && !(x->nenv && env_busy(x)))
	{
		u->Process = amd_quick_process;
		set_stamp(hs, x, 1);
	}
	if(noise != before)
		a2_SetStateProperty(hs->cfg->interface, A2P_PNOISESEED, (int)noise);
}

/* is the voice 'u' heads made of our own plain units only?  then wire it up */
static int setup_simple_chain(A2P_unit *u)
{
	static const A2P_unitdesc *const plain[] = {
		&a2_wtosc_unitdesc, &a2_panmix_unitdesc, &a2_filter12_unitdesc, &a2_fbdelay_unitdesc,
		&a2_fm1_unitdesc, &a2_fm2_unitdesc, &a2_fm3_unitdesc, &a2_fm4_unitdesc, &a2_fm3p_unitdesc,
		&a2_fm4p_unitdesc, &a2_fm2r_unitdesc, &a2_fm4r_unitdesc, &a2_dc_unitdesc,
		&a2_waveshaper_unitdesc, &a2_dcblock_unitdesc, &a2_limiter_unitdesc };
	XTRA *x = xtra(u);
	A2P_unit *n;
	unsigned i;
	int slot;
	for(n = u; n; n = n->next)
	{
		if(n->descriptor == &a2_env_unitdesc)
			continue;
		for(i = 0; i < sizeof(plain) / sizeof(plain[0]); ++i)
			if(n->descriptor == plain[i])
				break;
		if(i == sizeof(plain) / sizeof(plain[0]) || xtra(n)->uid < 0)
			return 0;
	}
	if(x->nenv && !env_chain_ok(u))
		return 0;
	if((slot = a2amd_voice_slot(XCTX(x), x->uid)) < 0)
		return 0;
	x->slot = slot;
	stamp_room(x->hs, x->dev, slot);
	set_stamp(x->hs, x, 0);
	x->tail[0] = u->next;
	x->tail[1] = u->next ? u->next->next : NULL;
	for(n = u; n; n = n->next)
	{
		xtra(n)->head = u;
		if(n->descriptor == &a2_env_unitdesc)
			continue;	/* (keeps its own Process) */
		xtra(n)->chain_checked = 1;
		n->Process = n == u ? amd_head_process : amd_noop;
	}
	for(i = 0; i < x->nenv; ++i)
		xtra(x->env[i])->head = u;	/* (those in front of the head too) */
	return 1;
}

static int null_walk = -1;	/* A2AMD_NULLWALK=1 (measurement only): leaf units record nothing - the engine's bare walk */

static void amd_process(A2P_unit *u, unsigned offset, unsigned frames)
{
	XTRA *x;
	HOSTSTATE *hs;
	if(null_walk < 0)
		null_walk = getenv("A2AMD_NULLWALK") ? atoi(getenv("A2AMD_NULLWALK")) : 0;
	if(null_walk == 2)
		return;
	x = xtra(u);
	hs = x->hs;
	if(null_walk == 1 && !x->is_root)
		return;
	if(x->pending)
		route_voice(x);

	if(!x->chain_checked)
	{
		x->chain_checked = 1;
		check_chain_behind(u);
		if(x->is_head && setup_simple_chain(u))
		{
			amd_head_process(u, offset, frames);
			return;
		}
	}
	if(hs->walk_ahead)
		look_ahead(hs, x, u);
	forward_process(x, offset, frames);
	if(x->is_root && x->kind == A2AMD_PANMIX && !hs->batching)
	{
		/* end of a root window: render it and hand the result to the engine */
		int32_t *outp[A2AMD_MAXCHANNELS];
		int c, n;
		for(c = 0; c < A2AMD_MAXCHANNELS; ++c)
			outp[c] = hs->out[c];
		if(!hs->failed)
		{
			n = render_batch(hs, outp, A2AMD_MAXFRAG);
			if(n != (int)frames)
				fail(hs, "a2amd_render", n);
		}
		if(hs->failed)
			memset(hs->out, 0, sizeof(hs->out));
		hs->batch_frags = 0;
		for(c = 0; c < u->noutputs; ++c)
			memcpy(u->outputs[c] + offset, hs->out[c], frames * sizeof(int32_t));
		if(hs->npend || hs->nzombies)
			deliver_pending(hs);
	}
}

static void amd_inline_process(A2P_unit *u, unsigned offset, unsigned frames)
{

	XTRA *x = xtra(u);
	HOSTSTATE *hs = x->hs;
	int rc;
	if(!hs->depth)
	{
		/* a window of the root voice opens a backend fragment */
		hook_driver(hs);
		if(!hs->swept)
		{
			sweep_waves(hs);
			hs->swept = hs->in_buffer;	/* (outside our driver hook: every root window) */
		}
		if(hs->in_buffer && !hs->decided)
		{
			hs->decided = 1;
			hs->batching = !root_has_inserts(hs);
		}
		if(hs->batching && hs->batch_frags == hs->max_batch)
			flush_batch(hs);
		else if(hs->batching && hs->split_frames && !hs->part_frames && hs->acc_pos == 0 && hs->rec_pos >= hs->split_frames)
			flush_part(hs);
		{
			int d;
			for(d = 0; d < hs->ndev; ++d)
			{
				if(!hs->failed && (rc = a2amd_fragment(hs->ctxs[d], frames)))
					fail(hs, "a2amd_fragment", rc);
				if(offset && !hs->failed && (rc = a2amd_fragment_offset(hs->ctxs[d], offset)))
					fail(hs, "a2amd_fragment_offset", rc);
				hs->map[d] = hs->failed ? NULL : a2amd_default_map(hs->ctxs[d], &hs->map_cap[d]);
				if(!hs->map[d])
					hs->map_cap[d] = 0;
			}
		}
		if(x->is_root && x->vms)
		{
			if(!hs->root_wake_valid || x->vms->waketime != hs->root_wake)
			{
				hs->root_period = hs->root_wake_valid ? x->vms->waketime - hs->root_wake : 0;
				hs->root_wake = x->vms->waketime;
				hs->root_wake_valid = 1;
			}
		}
		{
			/* A2AMD_ROOTTRACE=1 (debugging aid): a window of the root voice that is not a whole fragment - the engine cut
			 * it (a2_VoiceProcess, core.c:1852-1878) where the root's own VM wakes up or an event for it falls due */
			static int rt = -1;
			if(rt < 0)
				rt = getenv("A2AMD_ROOTTRACE") != NULL;
			if(rt && (frames != 64 || offset))
				fprintf(stderr, "a2amd units: root window (%u, %u) at frame %u of the buffer: root VM state %d pc %d waketime %u\n",
						offset, frames, hs->rec_pos, x->vms ? (int)x->vms->state : -1, x->vms ? (int)x->vms->pc : -1,
						x->vms ? x->vms->waketime : 0u);
		}
		++hs->batch_frags;
		++hs->frag_serial;
		hs->base = offset;
		hs->win_frames = frames;
		hs->win_pos = hs->rec_pos;
		hs->rec_pos += frames;
	}
	if(x->pending)
		route_voice(x);
	if(hs->walk_ahead && !hs->walker)
		look_ahead(hs, x, u);
	if(x->head && !(offset == hs->base && frames == hs->win_frames))
		set_stamp(hs, x, 0);	/* (a window that is not the default one: group_standing()) */
	forward_process(x, offset, frames);
	if(hs->depth < 70)
		hs->dev_stack[hs->depth] = x->dev;	/* (our subvoices' buses live where we do) */
	++hs->depth;
	x->orig_process(u, offset, frames);	/* the engine walks the subvoices */
	--hs->depth;
	if(x->is_root && hs->ndev > 1)
	{
		int d;
		for(d = 0; d < hs->ndev && !hs->failed; ++d)
			if((rc = a2amd_inline_end(hs->ctxs[d], hs->root_uid[0][d])))
				fail(hs, "a2amd_inline_end", rc);
	}
	else if(!hs->failed && (rc = a2amd_inline_end(XCTX(x), x->uid)))
		fail(hs, "a2amd_inline_end", rc);
}

/* The root voice's xinsert stays the engine's (it feeds the master bus and the
 * application's clients - a2play's sink, an insert effect - with the audio we
 * hand back).  In a batched buffer that audio does not exist yet when it runs:
 * READ clients get their windows from the rendered master bus afterwards,
 * WRITE-only clients run now and their output is added to the buffer at the end
 * (xi_process, xinsert.c:95-119: output = input + what the clients wrote). */
static void amd_rootx_process(A2P_unit *u, unsigned offset, unsigned frames)
{
	XTRA *x = xtra(u);
	HOSTSTATE *hs = x->hs;
	A2P_xinsert *xi = (A2P_xinsert *)u;
	A2P_xinsert_client *xic;
	int i, rc;
	unsigned s;
	if(!hs->batching)
	{
		x->orig_process(u, offset, frames);
		return;
	}
	for(xic = xi->clients; xic; xic = xic->next)
	{
		if(!(xic->flags & A2P_XI_WRITE))
		{
			PENDING *p = new_pending(hs);
			if(!p)
			{
				fail(hs, "out of memory for sink client windows", -1);
				return;
			}
			p->xi = xi;
			p->xic = xic;
			p->uid = -1;
			p->frag = 0;
			p->pos = hs->win_pos;
			p->what = "xinsert client callback";
			p->offset = 0;
			p->frames = frames;
		}
		else if(!(xic->flags & A2P_XI_READ))
		{
			int32_t tmp[A2AMD_MAXCHANNELS][A2AMD_MAXFRAG];
			int32_t *bufp[A2AMD_MAXCHANNELS];
			if(!hs->rinj_used)
			{
				for(i = 0; i < hs->cfg->channels; ++i)
					memset(hs->rinj[i], 0, hs->acc_cap * sizeof(int32_t));
				hs->rinj_used = 1;
			}
			memset(tmp, 0, sizeof(tmp));
			for(i = 0; i < u->ninputs; ++i)
				bufp[i] = tmp[i];
			if((rc = xic->callback(bufp, u->ninputs, frames, xic->userdata)))
				client_error(xi, rc, "xinsert client callback");
			for(i = 0; i < u->ninputs && i < hs->cfg->channels; ++i)
				for(s = 0; s < frames; ++s)
					hs->rinj[i][hs->win_pos + s] = (int32_t)((uint32_t)hs->rinj[i][hs->win_pos + s] +
							(uint32_t)tmp[i][s]);
		}
		else if(!x->refused)
		{
			/* attached after the buffer began (a timestamped a2_InsertCallback landing in the
			 * middle of a batched buffer): served from the next buffer on - and said so, once,
			 * since the reference would have replaced the master bus from this window on */
			x->refused = 1;
			client_error(hs->root_xi, A2P_NOTIMPLEMENTED, "a2amd: insert client attached to the root voice in the "
					"middle of a buffer: it takes effect with the next a2_Run() buffer");
		}
	}
}

static void amd_rootx_setprocess(A2P_unit *u)
{
	XTRA *x = xtra(u);
	x->orig_setprocess(u);		/* xi_SetProcess picks the engine's variant ... */
	x->orig_process = u->Process;
	u->Process = amd_rootx_process;	/* ... which runs unless the buffer is batched */
}

/* ---- control register writes --------------------------------------------------*/
static int wave_id_of(HOSTSTATE *hs, int dev, A2P_wave *w)
{
	a2amd_wavedesc d;
	int i, k, id, levels, slot = -1;
	if(!w)
		return -1;
	for(i = 0; i < hs->nwaves; ++i)
		if(hs->wave_ptr[i] == w)
		{
			if(hs->wave_id[i][dev] >= 0)
				return hs->wave_id[i][dev];
			slot = i;	/* (known, but not on this GPU yet) */
			break;
		}
	if(hs->failed || !ctx_of(hs))
		return -1;
	if(slot < 0 && hs->nwaves == hs->cap_waves)
	{
		int nc = hs->cap_waves ? hs->cap_waves * 2 : 256;
		A2P_wave **np = (A2P_wave **)realloc(hs->wave_ptr, nc * sizeof(A2P_wave *));
		int (*ni)[MAXDEV] = np ? (int (*)[MAXDEV])realloc(hs->wave_id, nc * sizeof(hs->wave_id[0])) : NULL;
		if(np)
			hs->wave_ptr = np;
		if(ni)
			hs->wave_id = ni;
		if(!np || !ni)
		{
			fail(hs, "out of memory for the wave registry", -1);
			return -1;
		}
		hs->cap_waves = nc;
	}
	memset(&d, 0, sizeof(d));
	d.type = w->type;
	d.flags = w->flags;
	d.period = w->period;
	levels = w->type == A2AMD_WMIPWAVE ? A2AMD_MIPLEVELS : w->type == A2AMD_WWAVE ? 1 : 0;
	for(i = 0; i < levels; ++i)
	{
		d.size[i] = w->size[i];
		d.data[i] = w->data[i];
	}
	id = -1;
	if(hs->npendcaps)
	{
		/* the device rendered this wave (a2_RenderWave below): its copy is built from what the device kept - on
		 * the context of this state that gets to play it first; others take the engine's copy */
		a2amd_capture *cap = NULL;
		pthread_mutex_lock(&hs->pendcaps_mtx);
		for(i = 0; i < hs->npendcaps; ++i)
			if(hs->pendcaps[i].w == w && a2_GetWave(hs->cfg->interface, hs->pendcaps[i].handle) == w)
			{
				cap = hs->pendcaps[i].cap;
				hs->pendcaps[i] = hs->pendcaps[--hs->npendcaps];
				break;
			}
		pthread_mutex_unlock(&hs->pendcaps_mtx);
		if(cap)
		{
			if(a2amd_capture_frames(cap) == w->size[0])
				id = a2amd_wave_upload_captured(hs->ctxs[dev], (uint64_t)(uintptr_t)w, &d, cap);
			if(getenv("A2AMD_WAVE_STATS"))
				fprintf(stderr, "a2amd units: wave %p, %u frames rendered on the device: %s\n", (void *)w, a2amd_capture_frames(cap),
						id >= 0 ? "device copy built from the capture" : "uploaded from the engine's copy");
			a2amd_capture_free(cap);
		}
	}
	if(id < 0)
		id = a2amd_wave_upload(hs->ctxs[dev], (uint64_t)(uintptr_t)w, &d);
	if(id < 0)
	{
		fail(hs, "a2amd_wave_upload", id);
		return -1;
	}
	if(slot < 0)
	{
		slot = hs->nwaves++;
		hs->wave_ptr[slot] = w;
		for(k = 0; k < MAXDEV; ++k)
			hs->wave_id[slot][k] = -1;
	}
	hs->wave_id[slot][dev] = id;
	return id;
}

static void amd_write(A2P_unit *u, int reg, int v, unsigned start, unsigned dur)
{
	XTRA *x = xtra(u);
	int rc;
	if(x->kind == A2AMD_WTOSC && reg == 0)		/* wtosc_Wave, wtosc.c:433-440 */
	{
		A2P_wave *w = a2_GetWave(x->hs->cfg->interface, v >> 16);
		const int noise = w && w->type == A2AMD_WNOISE;
		x->hs->noise_oscs += noise - x->is_noise;
		x->is_noise = noise;
		if(!x->pending)
			v = wave_id_of(x->hs, x->dev, w);
	}
	if(x->pending)
	{
		/* (its voice has no GPU yet: route_voice() replays this) */
		BIRTHOP *b = new_birthop(x->hs);
		if(!b)
		{
			fail(x->hs, "out of memory", -1);
			return;
		}
		b->u = u;
		b->vms = x->vms;
		b->is_write = 1;
		b->reg = reg;
		b->v = v;
		b->start = start;
		b->dur = dur;
		b->transpose = x->vms->r[A2P_R_TRANSPOSE];
		return;
	}
	/* (a voice that was reporting its default windows through the map calls in again) */
	if(x->head)
	{
		/* (also a group voice: a2amd_units_standing() grants it again after the next visit) */
		if(x->head->Process == amd_quick_process)
			x->head->Process = amd_head_process;
		set_stamp(x->hs, xtra(x->head), 0);
	}
	if(x->hs->failed)
		return;
	if((rc = a2amd_unit_write(XCTX(x), x->uid, reg, v, start, dur, x->vms->r[A2P_R_TRANSPOSE])))
		fail(x->hs, "a2amd_unit_write", rc);
}

#define WR(n) static void wr##n(A2P_unit *u, int v, unsigned s, unsigned d) { amd_write(u, n, v, s, d); }
WR(0) WR(1) WR(2) WR(3) WR(4) WR(5) WR(6) WR(7) WR(8) WR(9) WR(10) WR(11) WR(12)
static const A2P_write_cb wr_table[13] = { wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7, wr8, wr9, wr10, wr11, wr12 };

/* ---- the replaced units -----------------------------------------------------------*/
#define OWN_UNIT(K, name, regvals) \
static int name##_init(A2P_unit *u, A2P_vmstate *vms, void *sd, unsigned f) \
{ \
	static const int defaults[] = regvals; \
	unsigned i; \
	int rc = amd_init(K, u, vms, sd, f, 1); \
	if(rc) \
		return rc; \
	for(i = 0; i < sizeof(defaults) / sizeof(defaults[0]); ++i) \
		u->registers[i] = defaults[i];	/* the VM reads these back */ \
	u->Process = amd_process; \
	return 0; \
}
#define ARR(...) { __VA_ARGS__ }
/* register defaults: wtosc.c:409-413, panmix.c:262-263, filter12.c:191-195,
 * fbdelay.c:183-194 */
OWN_UNIT(A2AMD_WTOSC, wtosc, ARR(0, 0, 0, 0))
OWN_UNIT(A2AMD_PANMIX, panmix, ARR(65536, 0))
OWN_UNIT(A2AMD_FILTER12, filter12, ARR(0, 0, 65536, 0, 0))
OWN_UNIT(A2AMD_FBDELAY, fbdelay, ARR(400 << 16, 280 << 16, 320 << 16, 65536, 16384, 32768, 32768))

static const A2P_crdesc wtosc_regs[] = { { "w", wr0 }, { "p", wr1 }, { "a", wr2 }, { "phase", wr3 },
		{ NULL, NULL } };
static const A2P_crdesc panmix_regs[] = { { "vol", wr0 }, { "pan", wr1 }, { NULL, NULL } };
static const A2P_constdesc panmix_consts[] = { { "CENTER", 0 }, { "LEFT", -65536 }, { "RIGHT", 65536 },
		{ NULL, 0 } };
static const A2P_crdesc filter12_regs[] = { { "cutoff", wr0 }, { "q", wr1 }, { "lp", wr2 }, { "bp", wr3 },
		{ "hp", wr4 }, { NULL, NULL } };
static const A2P_crdesc fbdelay_regs[] = { { "fbdelay", wr0 }, { "ldelay", wr1 }, { "rdelay", wr2 },
		{ "drygain", wr3 }, { "fbgain", wr4 }, { "lgain", wr5 }, { "rgain", wr6 }, { NULL, NULL } };

const A2P_unitdesc a2_wtosc_unitdesc = { "wtosc", 0, wtosc_regs, NULL, NULL, 0, 0, 1, 1,
	A2P_BLOCK_SIZE, wtosc_init, amd_deinit, amd_open, amd_close };
const A2P_unitdesc a2_panmix_unitdesc = { "panmix", 0, panmix_regs, NULL, panmix_consts, 1, 2, 1, 2,
	A2P_BLOCK_SIZE, panmix_init, amd_deinit, amd_open, amd_close };
const A2P_unitdesc a2_filter12_unitdesc = { "filter12", A2P_MATCHIO, filter12_regs, NULL, NULL, 1, 2, 1, 2,
	A2P_BLOCK_SIZE, filter12_init, amd_deinit, amd_open, amd_close };
const A2P_unitdesc a2_fbdelay_unitdesc = { "fbdelay", 0, fbdelay_regs, NULL, NULL, 1, 2, 1, 2,
	A2P_BLOCK_SIZE, fbdelay_init, amd_deinit, amd_open, amd_close };

/* ---- env, src/units/env.c ---------------------------------------------------
 * A control-rate unit: no audio ports, four registers (target mode down time), one control output
 * that the program wires to a register of another unit of the voice (a2_ControlWire, core.c:330-345).
 * A write to 'target' starts a segment; linear ones are passed on as a ramp of the wired register
 * (env.c:172-181), the others walk one of eight tables, one step per window, and write the wired
 * register at the head of every window (env.c:116-134).
 * It is ours - rather than the engine's, which would do as well while the engine runs the voice -
 * so that the segment in flight can travel with the voice: when the device VM takes the voice
 * (a2amd_units_vm_adopt) the state below goes along, the device steps the segment (a2amd_vmcore.h:
 * env_target / env_lut, the same arithmetic) and writes it back on recall.  Between those, on a
 * voice the ENGINE runs, this is the reference's unit restated: it renders nothing, its output is
 * the wired unit's write callback - the same record a VM instruction's write would leave. */
typedef struct ENVSTATE
{
	int32_t		ramper[4];	/* A2_ramper {value, target, delta, timer}, a2_dsp.h */
	int32_t		lut;		/* table of the running segment */
	int32_t		scale, offset, out;	/* env.c:96-98 */
	uint32_t	msdur;
} ENVSTATE;
#define ENVS(u)	((ENVSTATE *)((char *)(u) + 64 + sizeof(XTRA)))
_Static_assert(64 + sizeof(XTRA) + sizeof(ENVSTATE) <= A2P_BLOCK_SIZE, "ENVSTATE placement");
#define ENV_LUTSIZE 64	/* A2ENV_LUTSIZE, env.c:27-28 */

static uint16_t env_luts[8][ENV_LUTSIZE + 2];	/* spline, EXP1 .. EXP7 (env.c:36-47) */
static pthread_once_t env_luts_once = PTHREAD_ONCE_INIT;

/* env_InitLUTs, env.c:218-257: the expressions - and the types they are evaluated in - are the
 * reference's, so that every entry comes out the same (float constants widened to double next to a
 * double, libm's cos / pow) */
static void env_make_luts(void)
{
	static const char degree[7] = { 1, 2, 3, 4, 6, 9, 13 };
	int i, j;
	for(i = 0; i < ENV_LUTSIZE; ++i)
		env_luts[0][i] = (1.0f - cos(i * M_PI / (ENV_LUTSIZE - 1))) * 16384.0f + 0.5f;
	for(j = 0; j < 7; ++j)
	{
		const float d = degree[j];
		const double c = pow(0.1f, d);
		const double rc = 0.002f + 0.1f * pow(0.8f, d);
		for(i = 0; i < ENV_LUTSIZE; ++i)
		{
			const double x = 1.0f - (double)i / ENV_LUTSIZE;
			const double r = (1.0f - x) * rc;
			env_luts[1 + j][i] = (pow(c, x) * (1.0f - r) + r - c * x) * 32768.0f + 0.5f;
		}
	}
	for(j = 0; j < 8; ++j)
		env_luts[j][ENV_LUTSIZE] = env_luts[j][ENV_LUTSIZE + 1] = 32768;
}

/* a2_PrepareRamper / a2_RunRamper / a2_SetRamper, a2_dsp.h (wrapping 32 bit arithmetic, as
 * a2amd_vmcore.h: rp_prepare, rp_run, rp_set) */
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static void ramper_prepare(int32_t *r, int frames)
{
	if(!r[3])
	{
		r[0] = r[1];
		r[2] = 0;
	}
	else if(frames <= (r[3] >> 8))
	{
		r[2] = (int32_t)(((int64_t)wsub(r[1], r[0]) * 256) / r[3]);
		r[3] = wsub(r[3], frames << 8);
	}
	else
	{
		r[2] = wsub(r[1], r[0]) / frames;
		r[3] = 0;
	}
}
static void ramper_set(int32_t *r, int target, int start, int duration)
{
	r[1] = (int32_t)((uint32_t)target << 8);
	r[3] = wadd(duration, start);
	if(r[3] < 256)
		r[0] = r[1];
	else
		r[0] = wadd(r[0], wmul(r[2], start) >> 8);
}

static void env_off(A2P_unit *u, unsigned offset, unsigned frames)
{
}

/* one window of a running segment: env_ProcessLUT, env.c:116-134 */
static void env_run(A2P_unit *u, unsigned offset, unsigned frames)
{
	ENVSTATE *e = ENVS(u);
	A2P_cport *co = (A2P_cport *)u->coutputs;
	const uint16_t *t = env_luts[e->lut & 7];
	uint32_t i, f;
	ramper_prepare(e->ramper, (int)frames);
	e->ramper[0] = wadd(e->ramper[0], wmul(e->ramper[2], (int)frames));
	i = (uint32_t)(e->ramper[0] >> (24 - 6));
	/* A re-targeted unity ramp can overshoot 1.0 by a window's worth: the engine then reads on past the
	 * table's two pad entries - into the NEXT table of its one malloc'ed array of eight (env.c:32-35, 259),
	 * which is how env_luts is laid out too, so the same words come back (found by the differential fuzzer,
	 * seed 3779).  Past the last table the engine reads its heap (or crashes: a negative ramp value is a
	 * huge index); that nobody can follow, and the index stays in the array. */
	{
		const uint32_t imax = (uint32_t)(8 - (e->lut & 7)) * (ENV_LUTSIZE + 2) - 2;
		if(i > imax)
			i = imax;
	}
	f = (uint32_t)(e->ramper[0] >> (24 - 16 - 6)) & 65535u;
	e->out = (int32_t)((f * (uint32_t)t[i + 1] + (65536u - f) * (uint32_t)t[i]) >> 7);
	e->out = wadd((int32_t)(((int64_t)e->out * e->scale) >> 24), e->offset);
	co->write(co->unit, e->out, offset, frames << 8);
	if(!e->ramper[2])
		u->Process = env_off;
}

/* a write to 'target': env_Target, env.c:137-215 */
static void env_target(A2P_unit *u, int v, unsigned start, unsigned dur)
{
	ENVSTATE *e = ENVS(u);
	XTRA *x = xtra(u);
	const int *ci = u->registers;
	A2P_cport *co = (A2P_cport *)u->coutputs;
	int mode, lut;
	if(!co->write)
		return;
	if(ci[3])	/* the 'time' register overrides the ramp's duration */
		dur = (unsigned)(((int64_t)ci[3] * e->msdur + 0x7fffff) >> 24);
	if(dur >= 256 - start)
	{
		mode = ci[2] >> 16;	/* 'down', unless going up or LINKed to 'mode' */
		if(v >= e->out || !mode)
			mode = ci[1] >> 16;
	}
	else
		mode = 1;	/* (no time to bend anything) */
	if(mode == -1)		/* SPLINE */
	{
		lut = 0;
		mode = 1;
	}
	else if(mode >= 2 && mode <= 8)		/* EXP1 .. EXP7 */
		lut = mode - 1;
	else if(mode >= -8 && mode <= -2)	/* IEXP1 .. IEXP7: the same tables, walked backwards */
		lut = -mode - 1;
	else
	{
		e->out = v;
		co->write(co->unit, v, start, dur);
		u->Process = env_off;
		return;
	}
	e->lut = lut;
	if(mode >= 0)
	{
		e->scale = wsub(v, e->out);
		e->offset = e->out;
		e->ramper[0] = 0;
		ramper_set(e->ramper, 1 << 16, (int)start, (int)dur);
	}
	else
	{
		e->scale = wsub(e->out, v);
		e->offset = wsub(e->out, e->scale);
		e->ramper[0] = (1 << 16) << 8;
		ramper_set(e->ramper, 0, (int)start, (int)dur);
	}
	u->Process = env_run;
	/* (the voice's units want their windows one by one while a segment runs) */
	if(x->head && x->hs)
	{
		if(x->head->Process == amd_quick_process)
			x->head->Process = amd_head_process;
		set_stamp(x->hs, xtra(x->head), 0);
	}
}

static int env_init(A2P_unit *u, A2P_vmstate *vms, void *sd, unsigned flags)
{
	HOSTSTATE *hs = (HOSTSTATE *)sd;
	XTRA *x = xtra(u);
	ENVSTATE *e = ENVS(u);
	memset(x, 0, sizeof(*x));
	memset(e, 0, sizeof(*e));
	x->hs = hs;
	x->vms = vms;
	x->kind = -1;
	x->uid = -1;
	if(hs->chain_vms != vms)
	{
		hs->chain_vms = vms;
		hs->chain_last = NULL;
		hs->chain_head = NULL;
		hs->chain_nfwd = hs->chain_nenv = 0;
	}
	if(hs->chain_head)
	{
		XTRA *hx = xtra(hs->chain_head);
		if(hx->nenv < 2)
		{
			hx->env[hx->nenv] = u;
			hx->env_before[hx->nenv] = (uint8_t)hs->chain_nfwd;
		}
		if(hx->nenv < 3)
			++hx->nenv;
	}
	else
	{
		if(hs->chain_nenv < 2)
			hs->chain_env[hs->chain_nenv] = u;
		if(hs->chain_nenv < 3)
			++hs->chain_nenv;
	}
	e->msdur = hs->cfg->samplerate * 65.536f + .5f;	/* env.c:236 */
	u->registers[0] = 0;		/* env.c:243-246 */
	u->registers[1] = 1;		/* A2ENVRM_LINEAR - as the reference has it: not shifted */
	u->registers[2] = 0;		/* A2ENVRM_LINK */
	u->registers[3] = 0;
	u->Process = env_off;
	return 0;
}

static int env_open(A2P_config *cfg, void **statedata)
{
	pthread_once(&env_luts_once, env_make_luts);
	return amd_open(cfg, statedata);
}

/* a segment is running on one of the head's env units */
static int env_busy(const XTRA *head)
{
	int k;
	for(k = 0; k < head->nenv && k < 2; ++k)
		if(head->env[k]->Process == env_run)
			return 1;
	return head->nenv > 2;
}

/* Which forwarded unit of head's chain - by position - and which of its registers is this write
 * callback?  -1: none of ours. */
static int wired_to(A2P_unit *head, const A2P_cport *co, int *reg)
{
	A2P_unit *n;
	int k, pos = 0;
	for(k = 0; k < 13; ++k)
		if(co->write == wr_table[k])
			break;
	if(k == 13)
		return -1;
	for(n = head; n; n = n->next)
	{
		if(n->descriptor == &a2_env_unitdesc)
			continue;
		if(n == co->unit)
		{
			*reg = k;
			return pos;
		}
		++pos;
	}
	return -1;
}

/* With the head's Process speaking for the whole chain (setup_simple_chain) the chain's window is
 * ONE event; the engine gives the window to unit after unit (core.c:1875-1876), with an env's write
 * between two of them.  The two agree when every env either runs before all the others, or is wired
 * to a unit in front of it - which has rendered the window either way. */
static int env_chain_ok(A2P_unit *head)
{
	XTRA *x = xtra(head);
	int k;
	if(x->nenv > 2)
		return 0;
	for(k = 0; k < x->nenv; ++k)
	{
		const A2P_cport *co = (const A2P_cport *)x->env[k]->coutputs;
		int reg, pos;
		if(!co->write)
			continue;
		if((pos = wired_to(head, co, &reg)) < 0)
			return 0;	/* (wired to another env, or to a register of somebody else's) */
		if(x->env_before[k] && pos >= x->env_before[k])
			return 0;
	}
	return 1;
}

static const A2P_crdesc env_regs[] = { { "target", env_target }, { "mode", NULL }, { "down", NULL }, { "time", NULL },
		{ NULL, NULL } };
static const A2P_codesc env_couts[] = { { "out" }, { NULL } };
/* env.c:301-338: ramp modes as 16:16 values */
static const A2P_constdesc env_consts[] = {
	{ "IEXP7", -8 * 65536 }, { "IEXP6", -7 * 65536 }, { "IEXP5", -6 * 65536 }, { "IEXP4", -5 * 65536 }, { "IEXP3", -4 * 65536 },
	{ "IEXP2", -3 * 65536 }, { "IEXP1", -2 * 65536 }, { "SPLINE", -1 * 65536 }, { "LINK", 0 }, { "LINEAR", 1 << 16 },
	{ "EXP1", 2 << 16 }, { "EXP2", 3 << 16 }, { "EXP3", 4 << 16 }, { "EXP4", 5 << 16 }, { "EXP5", 6 << 16 },
	{ "EXP6", 7 << 16 }, { "EXP7", 8 << 16 }, { NULL, 0 } };
const A2P_unitdesc a2_env_unitdesc = { "env", 0, env_regs, env_couts, env_consts, 0, 0, 0, 0,
	A2P_BLOCK_SIZE, env_init, amd_deinit, env_open, amd_close };

/* ---- the FM oscillators, fm.c:509-834 ----------------------------------------------
 * Eight descriptors over one register list (phase, then p a fb per operator,
 * fm.c:54-79), all registers initially 0 (fm.c:373-376). */
OWN_UNIT(A2AMD_FM1, fm1, ARR(0, 0, 0, 0))
OWN_UNIT(A2AMD_FM2, fm2, ARR(0, 0, 0, 0, 0, 0, 0))
OWN_UNIT(A2AMD_FM3, fm3, ARR(0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
OWN_UNIT(A2AMD_FM4, fm4, ARR(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
OWN_UNIT(A2AMD_FM3P, fm3p, ARR(0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
OWN_UNIT(A2AMD_FM4P, fm4p, ARR(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
OWN_UNIT(A2AMD_FM2R, fm2r, ARR(0, 0, 0, 0, 0, 0, 0))
OWN_UNIT(A2AMD_FM4R, fm4r, ARR(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))

#define FM_REGS(n) static const A2P_crdesc fm##n##_regs[] = { { "phase", wr0 }, \
	{ "p", wr1 }, { "a", wr2 }, { "fb", wr3 }, \
	{ n > 1 ? "p1" : NULL, n > 1 ? wr4 : NULL }, { "a1", wr5 }, { "fb1", wr6 }, \
	{ n > 2 ? "p2" : NULL, n > 2 ? wr7 : NULL }, { "a2", wr8 }, { "fb2", wr9 }, \
	{ n > 3 ? "p3" : NULL, n > 3 ? wr10 : NULL }, { "a3", wr11 }, { "fb3", wr12 }, { NULL, NULL } };
FM_REGS(1) FM_REGS(2) FM_REGS(3) FM_REGS(4)

#define FM_DESC(sym, nm, regs) const A2P_unitdesc a2_##sym##_unitdesc = { nm, 0, regs, NULL, NULL, \
	0, 0, 1, 1, A2P_BLOCK_SIZE, sym##_init, amd_deinit, amd_open, amd_close };
FM_DESC(fm1, "fm1", fm1_regs) FM_DESC(fm2, "fm2", fm2_regs) FM_DESC(fm3, "fm3", fm3_regs)
FM_DESC(fm4, "fm4", fm4_regs) FM_DESC(fm3p, "fm3p", fm3_regs) FM_DESC(fm4p, "fm4p", fm4_regs)
FM_DESC(fm2r, "fm2r", fm2_regs) FM_DESC(fm4r, "fm4r", fm4_regs)

/* ---- the small units: dc.c:241-281, waveshaper.c:165-193, dcblock.c:162-190,
 * limiter.c:222-250; register defaults dc.c:170-171, waveshaper.c:139,
 * dcblock.c:131, limiter.c:176-177 ----------------------------------------------------*/
OWN_UNIT(A2AMD_DC, dc, ARR(0, 1 << 16))
OWN_UNIT(A2AMD_WAVESHAPER, wshaper, ARR(0))
OWN_UNIT(A2AMD_DCBLOCK, dcblock, ARR(-5 * 65536))
OWN_UNIT(A2AMD_LIMITER, limiter, ARR(64 << 16, 1 << 16))
static const A2P_crdesc dc_regs[] = { { "value", wr0 }, { "mode", wr1 }, { NULL, NULL } };
static const A2P_constdesc dc_consts[] = { { "STEP", 0 }, { "LINEAR", 1 << 16 }, { NULL, 0 } };
static const A2P_crdesc ws_regs[] = { { "amount", wr0 }, { NULL, NULL } };
static const A2P_crdesc dcb_regs[] = { { "cutoff", wr0 }, { NULL, NULL } };
static const A2P_crdesc lim_regs[] = { { "release", wr0 }, { "threshold", wr1 }, { NULL, NULL } };
const A2P_unitdesc a2_dc_unitdesc = { "dc", 0, dc_regs, NULL, dc_consts, 0, 0, 1, 2,
	A2P_BLOCK_SIZE, dc_init, amd_deinit, amd_open, amd_close };
const A2P_unitdesc a2_waveshaper_unitdesc = { "waveshaper", A2P_MATCHIO, ws_regs, NULL, NULL, 1, 2, 1, 2,
	A2P_BLOCK_SIZE, wshaper_init, amd_deinit, amd_open, amd_close };
const A2P_unitdesc a2_dcblock_unitdesc = { "dcblock", A2P_MATCHIO, dcb_regs, NULL, NULL, 1, 2, 1, 2,
	A2P_BLOCK_SIZE, dcblock_init, amd_deinit, amd_open, amd_close };
const A2P_unitdesc a2_limiter_unitdesc = { "limiter", A2P_MATCHIO, lim_regs, NULL, NULL, 1, 2, 1, 2,
	A2P_BLOCK_SIZE, limiter_init, amd_deinit, amd_open, amd_close };

/* ---- the wrapped engine-internal units ------------------------------------------------
 * inline and xinsert need engine internals (the voice behind a vmstate, the
 * xinsert client list), so the engine's own instances keep doing that part:
 * their Initialize runs first inside the same instance block, ours adds the
 * backend bookkeeping.  Both use two state pointers: ours for the backend, the
 * original's for itself. */
typedef struct WRAPSTATE { HOSTSTATE *hs; void *orig_sd; } WRAPSTATE;

static int wrap_open(const char *sym, A2P_config *cfg, void **statedata)
{
	const A2P_unitdesc *od = orig_desc(sym);
	WRAPSTATE *ws;
	void *hs = NULL;
	int rc;
	if(!od)
		return A2P_NOTIMPLEMENTED;
	if(!(ws = (WRAPSTATE *)calloc(1, sizeof(WRAPSTATE))))
		return A2P_OOMEMORY;
	if((rc = amd_open(cfg, &hs)))
	{
		free(ws);
		return rc;
	}
	ws->hs = (HOSTSTATE *)hs;
	if(od->OpenState && (rc = od->OpenState(cfg, &ws->orig_sd)))
	{
		amd_close(hs);
		free(ws);
		return rc;
	}
	*statedata = ws;
	return 0;
}

static void wrap_close(const char *sym, void *statedata)
{
	const A2P_unitdesc *od = orig_desc(sym);
	WRAPSTATE *ws = (WRAPSTATE *)statedata;
	if(od && od->CloseState)
		od->CloseState(ws->orig_sd);
	amd_close(ws->hs);
	free(ws);
}

static int inl_open(A2P_config *cfg, void **sd) { return wrap_open("a2_inline_unitdesc", cfg, sd); }
static void inl_close(void *sd) { wrap_close("a2_inline_unitdesc", sd); }

static int inl_init(A2P_unit *u, A2P_vmstate *vms, void *sd, unsigned flags)
{
	WRAPSTATE *ws = (WRAPSTATE *)sd;
	const A2P_unitdesc *od = orig_desc("a2_inline_unitdesc");
	int rc;
	if(!ctx_of(ws->hs))
		return A2P_DEVICEOPEN;	/* (before the engine's half has touched the voice) */
	rc = od ? od->Initialize(u, vms, ws->orig_sd, flags) : A2P_NOTIMPLEMENTED;
	if(rc)
		return rc;
	if((rc = amd_init(A2AMD_INLINE, u, vms, ws->hs, flags, 1)))
		return rc;
	xtra(u)->orig_process = u->Process;
	u->Process = amd_inline_process;
	return 0;
}

static void inl_deinit(A2P_unit *u)
{
	const A2P_unitdesc *od = orig_desc("a2_inline_unitdesc");
	amd_deinit(u);
	if(od && od->Deinitialize)
		od->Deinitialize(u);
}

/* ---- xinsert, xsink, xsource (src/units/xinsert.c, xsink.c, xsource.c) ----------
 * The engine's instances keep the A2_xinsert struct the client API works on
 * (a2_XinsertAddClient / a2_XinsertRemoveClient, xinsertapi.c:72-157) and, on
 * the root voice, their own Process (the root's audio is back on the CPU when
 * it runs).  Elsewhere their Process would shuffle the engine's unused CPU
 * buffers and hand THOSE to the clients, so ours is installed instead - also
 * when the client API re-installs one (A2_xinsert.SetProcess). */
static void amd_x_process(A2P_unit *u, unsigned offset, unsigned frames)
{
	XTRA *x = xtra(u);
	HOSTSTATE *hs = x->hs;
	int rc;
	if(!x->chain_checked)
	{
		x->chain_checked = 1;
		check_chain_behind(u);
	}
	if(hs->failed)
		return;
	if(x->pending)
		route_voice(x);
	if(((A2P_xinsert *)u)->clients || x->client_mode)
		serve_clients(x, (A2P_xinsert *)u, offset, frames);
	if(!hs->failed && (rc = a2amd_unit_process(XCTX(x), x->uid, offset - hs->base, frames, NULL)))
		fail(hs, "a2amd_unit_process", rc);
}

static void amd_x_setprocess(A2P_unit *u)
{
	u->Process = amd_x_process;
}

static int x_init(const char *sym, int kind, A2P_unit *u, A2P_vmstate *vms, void *sd, unsigned flags)
{
	WRAPSTATE *ws = (WRAPSTATE *)sd;
	const A2P_unitdesc *od = orig_desc(sym);
	int rc;
	if(!ctx_of(ws->hs))
		return A2P_DEVICEOPEN;
	rc = od ? od->Initialize(u, vms, ws->orig_sd, flags) : A2P_NOTIMPLEMENTED;
	if(rc)
		return rc;
	if((rc = amd_init(kind, u, vms, ws->hs, flags, 1)))
		return rc;
	if(xtra(u)->uid >= 0 || xtra(u)->pending)
	{
		((A2P_xinsert *)u)->SetProcess = amd_x_setprocess;
		amd_x_setprocess(u);
	}
	else if(xtra(u)->is_root && kind == A2AMD_XINSERT)
	{
		/* the root voice's xinsert: the engine's Process, except in batched buffers */
		A2P_xinsert *xi = (A2P_xinsert *)u;
		xtra(u)->orig_setprocess = xi->SetProcess;
		xi->SetProcess = amd_rootx_setprocess;
		amd_rootx_setprocess(u);
		ws->hs->root_xi = xi;
		ws->hs->engine_state = xi->state;
	}
	return 0;
}

static void x_deinit(const char *sym, A2P_unit *u)
{
	const A2P_unitdesc *od = orig_desc(sym);
	HOSTSTATE *hs = xtra(u)->hs;
	A2P_xinsert *z = NULL;
	int k, n = 0;
	if(hs->root_xi == (A2P_xinsert *)u)
		hs->root_xi = NULL;
	/* Windows its READ clients are still owed: the voice dies in the middle of a
	 * fragment that is not rendered yet.  xi_Deinitialize (xinsert.c:204-211)
	 * would tell the clients they are removed now; instead the client list
	 * moves to a stand-in A2_xinsert that lives until the fragment's audio is
	 * back: the clients get those windows and are removed then
	 * (deliver_pending), the way the engine does it. */
	for(k = 0; k < hs->npend; ++k)
		if((A2P_unit *)hs->pend[k].xi == u)
			++n;
	if(n && a2_XinsertRemoveClient && hs->nzombies == hs->cap_zombies)
	{
		int nc = hs->cap_zombies ? hs->cap_zombies * 2 : 16;
		A2P_xinsert **nz = (A2P_xinsert **)realloc(hs->zombies, nc * sizeof(A2P_xinsert *));
		if(nz)
		{
			hs->zombies = nz;
			hs->cap_zombies = nc;
		}
	}
	if(n && a2_XinsertRemoveClient && hs->nzombies < hs->cap_zombies &&
			(z = (A2P_xinsert *)malloc(sizeof(A2P_xinsert))))
	{
		A2P_xinsert *xi = (A2P_xinsert *)u;
		A2P_xinsert_client *c;
		*z = *xi;
		z->SetProcess = amd_x_setprocess;
		for(c = z->clients; c; c = c->next)
			c->unit = z;
		xi->clients = NULL;
		for(k = 0; k < hs->npend; ++k)
			if((A2P_unit *)hs->pend[k].xi == u)
				hs->pend[k].xi = z;
		hs->zombies[hs->nzombies++] = z;
	}
	else if(n)
	{
		/* (out of memory, or an engine that hides a2_XinsertRemoveClient: those windows are lost) */
		for(n = k = 0; k < hs->npend; ++k)
			if((A2P_unit *)hs->pend[k].xi != u)
				hs->pend[n++] = hs->pend[k];
			else if(hs->pend[k].insert && hs->ninserts > 0)
				--hs->ninserts;		/* (render_batch pauses for insert windows only while there are some) */
		hs->npend = n;
	}
	amd_deinit(u);
	if(od && od->Deinitialize)
		od->Deinitialize(u);
}

#define X_UNIT(KIND, name) \
	static int name##_open(A2P_config *cfg, void **sd) { return wrap_open("a2_" #name "_unitdesc", cfg, sd); } \
	static void name##_close(void *sd) { wrap_close("a2_" #name "_unitdesc", sd); } \
	static int name##_init(A2P_unit *u, A2P_vmstate *vms, void *sd, unsigned flags) \
		{ return x_init("a2_" #name "_unitdesc", KIND, u, vms, sd, flags); } \
	static void name##_deinit(A2P_unit *u) { x_deinit("a2_" #name "_unitdesc", u); }
X_UNIT(A2AMD_XINSERT, xinsert)
X_UNIT(A2AMD_XSINK, xsink)
X_UNIT(A2AMD_XSOURCE, xsource)

const A2P_unitdesc a2_inline_unitdesc = { "inline", 0, NULL, NULL, NULL, 0, 0, 1, A2AMD_MAXCHANNELS,
	A2P_BLOCK_SIZE, inl_init, inl_deinit, inl_open, inl_close };
const A2P_unitdesc a2_xinsert_unitdesc = { "xinsert", A2P_MATCHIO | A2P_XINSERT, NULL, NULL, NULL,
	1, A2AMD_MAXCHANNELS, 1, A2AMD_MAXCHANNELS, A2P_BLOCK_SIZE, xinsert_init, xinsert_deinit, xinsert_open, xinsert_close };
const A2P_unitdesc a2_xsink_unitdesc = { "xsink", A2P_XINSERT, NULL, NULL, NULL,
	1, A2AMD_MAXCHANNELS, 0, 0, A2P_BLOCK_SIZE, xsink_init, xsink_deinit, xsink_open, xsink_close };
const A2P_unitdesc a2_xsource_unitdesc = { "xsource", A2P_XINSERT, NULL, NULL, NULL,
	0, 0, 1, A2AMD_MAXCHANNELS, A2P_BLOCK_SIZE, xsource_init, xsource_deinit, xsource_open, xsource_close };


/* ---- INTEGRATION.md option C: what liba2amd_walk.so (a2amd_walk.c) asks of the units --------*/
int a2amd_units_walkview(const void *cfg, a2amd_walkview *out)
{
	HOSTSTATE *hs;
	int rc = -1;
	pthread_mutex_lock(&states_mtx);
	for(hs = states; hs; hs = hs->next_state)
		if(hs->refs && hs->cfg == cfg)
		{
			out->cfg = cfg;
			out->map = hs->map;
			out->map_cap = hs->map_cap;
			out->base = &hs->base;
			out->win_frames = &hs->win_frames;
			out->qstamp = hs->qstamp;
			out->qstamp_cap = hs->qstamp_cap;
			out->walker = &hs->walker;
			out->serial = &hs->serial;
			out->serial_value = hs->serial;
			out->state = hs;
			out->frag_serial = &hs->frag_serial;
			out->vm_live = &hs->vm_live;
			rc = 0;
			break;
		}
	pthread_mutex_unlock(&states_mtx);
	return rc;
}

int a2amd_units_hold(void *state, unsigned dev, const uint32_t *slots, unsigned n, int on)
{
	HOSTSTATE *hs = (HOSTSTATE *)state;
	int rc = 0, d;
	if(!hs || hs->failed)
		return -1;
	if(!n && !on)
	{
		for(d = 0; d < hs->ndev; ++d)
			if(hs->ctxs[d])
				a2amd_default_release_all(hs->ctxs[d]);
		return 0;
	}
	if(dev >= (unsigned)hs->ndev || !hs->ctxs[dev])
		return -1;
	if((rc = a2amd_default_hold(hs->ctxs[dev], slots, n, 0, on)) && on)
		a2amd_default_hold(hs->ctxs[dev], slots, n, 0, 0);	/* (all or nothing) */
	return rc;
}

/* A voice whose chain starts with an inline unit - a group (a2_NewGroup's driver), a delay bus:
 * its default window is Process(offset, frames) on each of its units and, in between, the walk of
 * its subvoices.  When the walk finds the whole subtree asleep there is nothing in that but the
 * backend's "this voice got its default window" - the same byte in the default map. */
static uint32_t group_standing(A2P_unit *head, uint32_t *slotdev)
{
	XTRA *x = xtra(head);
	HOSTSTATE *hs = x->hs;
	A2P_unit *n;
	if(!hs || x->is_root || x->pending || x->uid < 0 || hs->failed || hs->no_quick)
		return 0;
	if(!x->head)
	{
		/* first time: every unit of the chain ours and in the voice's context; the voice's slot */
		int slot;
		for(n = head; n; n = n->next)
			if(!is_ours(n->descriptor) || xtra(n)->uid < 0 || xtra(n)->pending || xtra(n)->dev != x->dev)
				return 0;
		if((slot = a2amd_voice_slot(XCTX(x), x->uid)) < 0 || slot >= (1 << 28))
			return 0;
		x->slot = slot;
		stamp_room(hs, x->dev, slot);
		if((unsigned)slot >= hs->qstamp_cap[x->dev])
			return 0;
		for(n = head; n; n = n->next)
			xtra(n)->head = head;
	}
	for(n = head; n; n = n->next)
	{
		const int k = xtra(n)->kind;
		if((k == A2AMD_XINSERT || k == A2AMD_XSINK || k == A2AMD_XSOURCE) &&
				(((A2P_xinsert *)n)->clients || xtra(n)->client_mode))
		{
			set_stamp(hs, x, 0);	/* clients want every window */
			return 0;
		}
	}
	if(!hs->qstamp[x->dev][x->slot])
	{
		if(a2amd_voice_markable(XCTX(x), x->uid) != 1)
			return 0;
		set_stamp(hs, x, 1);
	}
	*slotdev = (uint32_t)x->slot | ((uint32_t)x->dev << 28);
	return hs->qstamp[x->dev][x->slot];
}

/* the walk hands over A2_voice.units: the chain's head is its first unit that is not an env */
static inline const A2P_unit *past_envs(const A2P_unit *u)
{
	while(u && u->descriptor == &a2_env_unitdesc)
		u = u->next;
	return u;
}

uint32_t a2amd_units_standing(const void *head_unit, uint32_t *slotdev)
{
	const A2P_unit *head = past_envs((const A2P_unit *)head_unit);
	const XTRA *x;
	if(!head)
		return 0;
	if(head->descriptor == &a2_inline_unitdesc)
		return group_standing((A2P_unit *)head, slotdev);
	if(head->Process != amd_quick_process)
		return 0;
	x = (const XTRA *)((const char *)head + 64);
	if((unsigned)x->slot >= x->hs->qstamp_cap[x->dev] || (unsigned)x->slot >= (1u << 28))
		return 0;
	*slotdev = (uint32_t)x->slot | ((uint32_t)x->dev << 28);
	return x->hs->qstamp[x->dev][x->slot];
}


/* ---- scripted voices whose VM runs on the device (include/a2amd_vm.h) -------------
 * The walk (a2amd_walk.c) sees the engine's side of a voice - VM state, program text, which VM
 * register is wired to which unit's write callback - and calls here; this side knows which backend
 * unit an A2_unit is and which register a write callback stands for. */

int a2amd_units_vm_is(const void *head_unit)
{
	const A2P_unit *head = past_envs((const A2P_unit *)head_unit);
	const XTRA *x;
	if(!head || head->descriptor == &a2_inline_unitdesc || head->descriptor == &a2_xinsert_unitdesc ||
			head->descriptor == &a2_xsink_unitdesc || head->descriptor == &a2_xsource_unitdesc || !is_ours(head->descriptor))
		return 0;
	x = (const XTRA *)((const char *)head + 64);
	return x->head == head && x->vm;
}

int a2amd_units_vm_adopt(const void *head_unit, const uint32_t *code, unsigned nwords, const void *vmstate,
		void *const *wr_unit, void *const *wr_fn, uint32_t now, uint32_t msdur, int *has_exit, uint32_t *exit_when)
{
	A2P_unit *head = (A2P_unit *)past_envs((const A2P_unit *)head_unit), *n;
	XTRA *x;
	HOSTSTATE *hs;
	int32_t wu[A2AMD_VM_REGISTERS];
	uint8_t wreg[A2AMD_VM_REGISTERS];
	a2amd_vm_env envs[A2AMD_VM_MAXENV];
	int r, k, prog, rc;
	if(!head || !is_ours(head->descriptor) || head->descriptor == &a2_inline_unitdesc ||
			head->descriptor == &a2_xinsert_unitdesc || head->descriptor == &a2_xsink_unitdesc ||
			head->descriptor == &a2_xsource_unitdesc)
		return -1;
	x = (XTRA *)((char *)head + 64);
	hs = x->hs;
	/* a chain of our own plain units, wired up (setup_simple_chain), alive on one GPU */
	if(!hs || hs->failed || hs->no_vm || x->head != head || x->pending || x->uid < 0 || x->vm)
		return -2;
	/* about to wake again within 2 048 fragments (2.7 s)?  (a voice that wakes once a minute is better off
	 * with the quiet kernels between its wake-ups; rounds 3-4 waited for it to be SEEN awake twice, which
	 * cost a long-lived voice its first period and a note - awake twice in its life - everything) */
	if((int32_t)(((const a2amd_vm_state *)vmstate)->waketime - now) > (2048 * 64) << 8)
		return -2;
	/* (a voice just given back is not offered again in the same fragment) */
	if(x->vm_seen_valid && x->vm_seen == hs->frag_serial)
		return -2;
	if(x->vm_backoff && (int)(hs->frag_serial - x->vm_retry) < 0)
		return -2;
	if(x->nenv > A2AMD_VM_MAXENV)
		return -1;
	for(k = 0; k < x->nenv; ++k)
	{
		/* the voice's env units: state, place in the chain, where the control output goes */
		A2P_unit *eu = x->env[k];
		const ENVSTATE *e = ENVS(eu);
		const A2P_cport *co = (const A2P_cport *)eu->coutputs;
		a2amd_vm_env *o = &envs[k];
		int reg = 0, pos;
		memset(o, 0, sizeof(*o));
		memcpy(o->ramper, e->ramper, sizeof(o->ramper));
		o->lut = e->lut;
		o->scale = e->scale;
		o->offset = e->offset;
		o->out = e->out;
		o->active = eu->Process == env_run;
		o->regbase = (int32_t)(eu->registers - x->vms->r);
		o->before = x->env_before[k];
		o->out_unit = -1;
		if(co->write)
		{
			if((pos = wired_to(head, co, &reg)) < 0)
				return -1;
			o->out_unit = xtra(co->unit)->uid;
			o->out_reg = reg;
		}
	}
	if(x->nenv && !hs->envluts_sent[x->dev])
	{
		if((rc = a2amd_vm_envluts(XCTX(x), &env_luts[0][0])))
			return -1;
		hs->envluts_sent[x->dev] = 1;
	}
	for(r = 0; r < A2AMD_VM_REGISTERS; ++r)
	{
		wu[r] = -1;
		wreg[r] = 0;
		if(!wr_fn[r])
			continue;
		wu[r] = -2;	/* (somebody else's callback: fine as long as the program cannot reach it) */
		if((A2P_write_cb)wr_fn[r] == env_target)
		{
			for(k = 0; k < x->nenv; ++k)
				if((void *)x->env[k] == wr_unit[r])
					wu[r] = -3 - k;
			continue;
		}
		for(k = 0; k < 13; ++k)
			if((A2P_write_cb)wr_fn[r] == wr_table[k])
				break;
		if(k == 13)
			continue;
		for(n = head; n; n = n->next)
			if((void *)n == wr_unit[r])
				break;
		if(!n || !is_ours(n->descriptor) || xtra(n)->uid < 0 || xtra(n)->dev != x->dev)
			continue;
		wu[r] = xtra(n)->uid;
		wreg[r] = (uint8_t)k;
	}
	if((prog = a2amd_vm_program(XCTX(x), (uint64_t)(uintptr_t)code, code, nwords)) < 0)
		return -1;
	rc = a2amd_vm_adopt(XCTX(x), x->uid, prog, (const a2amd_vm_state *)vmstate, wu, wreg, now, msdur, envs, x->nenv);
	if(rc)
	{
		static int trace = -1;
		if(trace < 0)
			trace = getenv("A2AMD_VM_TRACE") != NULL;
		if(trace)
			fprintf(stderr, "a2amd units: voice not taken by the device VM: %s\n", a2amd_last_error(XCTX(x)));
		if(rc != A2AMD_EUNSUPPORTED)
		{
			/* "not now": 8, 16 ... 4 096 fragments until the next look */
			if(x->vm_backoff < 10)
				++x->vm_backoff;
			x->vm_retry = hs->frag_serial + (4u << x->vm_backoff);
		}
		return rc == A2AMD_EUNSUPPORTED ? -1 : -2;
	}
	x->vm_backoff = 0;
	/* from here on it stands like a sleeping voice: default windows through the map / holds */
	head->Process = amd_quick_process;
	set_stamp(hs, x, 1);
	x->vm = 1;
	++hs->vm_live;
	if(has_exit)
	{
		uint32_t when = 0;
		*has_exit = a2amd_vm_exit_time(XCTX(x), x->uid, &when);
		if(exit_when)
			*exit_when = when;
	}
	return 0;
}

int a2amd_units_vm_recall(const void *const *heads, unsigned n, void *const *vmstates)
{
	unsigned k, j;
	int rc = 0, d;
	for(d = 0; d < MAXDEV; ++d)
	{
		/* (one backend call - one device round trip - per context) */
		int32_t uids[64];
		a2amd_vm_state sts[64];
		a2amd_vm_env envs[64][A2AMD_VM_MAXENV];
		unsigned idx[64], m = 0;
		HOSTSTATE *hs = NULL;
		for(k = 0; k <= n; ++k)
		{
			/* a head is trusted only after the checks a2amd_units_vm_is() makes: the caller's note
			 * that a voice is the device VM's may be older than the voice (addresses are reused) */
			XTRA *x = k < n && a2amd_units_vm_is(heads[k]) ?
					(XTRA *)((char *)past_envs((const A2P_unit *)heads[k]) + 64) : NULL;
			if(k < n && (!x || x->dev != d))
				continue;
			if(x)
			{
				hs = x->hs;
				uids[m] = x->uid;
				idx[m++] = k;
			}
			if(m && (m == 64 || k == n))
			{
				XTRA *x0 = (XTRA *)((char *)past_envs((const A2P_unit *)heads[idx[0]]) + 64);
				int r2 = hs->failed ? -1 : a2amd_vm_recall(XCTX(x0), uids, m, sts, &envs[0][0]);
				if(r2 && !hs->failed)
					fail(hs, "a2amd_vm_recall", r2);
				for(j = 0; j < m; ++j)
				{
					A2P_unit *head = (A2P_unit *)past_envs((const A2P_unit *)heads[idx[j]]);
					XTRA *xj = (XTRA *)((char *)head + 64);
					int q;
					if(!r2)
						memcpy(vmstates[idx[j]], &sts[j], sizeof(a2amd_vm_state));
					for(q = 0; !r2 && q < xj->nenv && q < A2AMD_VM_MAXENV; ++q)
					{
						/* its env units are where the device left them */
						ENVSTATE *e = ENVS(xj->env[q]);
						const a2amd_vm_env *o = &envs[j][q];
						memcpy(e->ramper, o->ramper, sizeof(e->ramper));
						e->lut = o->lut;
						e->scale = o->scale;
						e->offset = o->offset;
						e->out = o->out;
						xj->env[q]->Process = o->active ? env_run : env_off;
					}
					xj->vm = 0;
					xj->vm_seen = hs->frag_serial;
					xj->vm_seen_valid = 1;	/* (not offered again in this fragment) */
					head->Process = amd_head_process;
					set_stamp(hs, xj, 0);
					if(hs->vm_live)
						--hs->vm_live;
				}
				if(r2)
					rc = -1;
				m = 0;
			}
		}
	}
	return rc;
}


/* ---- a2_RenderWave(), src/render.c:144-177 ----------------------------------------
 * The engine's own function does all the work - opens the off-line substate, runs the program, writes the
 * substate's output into the new wave, closes the stream (conversion, pads, mip levels on the host: the
 * engine's A2_wave is complete and is what a2_GetWave() hands anybody).  This entry point stands in front
 * of it (the same interposition as the unit descriptors: the compiler's call, src/compiler.c:3359, and the
 * application's both bind here) only to note which state is the substate: that state's drop-in context keeps
 * what it renders in device memory (a2amd_capture_begin), and when the engine has made its wave, the device
 * builds ITS copy - samples, pads, mip levels, coefficient entries - from there (a2amd_wave_upload_captured; on the
 * engine thread, when the wave is first played: wave_id_of) and the wave registry takes it as uploaded.  The rendered samples travel device -> host once, for the
 * engine's copy, and never back.  Waves with A2_NORMALIZE / A2_XFADE / A2_REVMIX, a substate on another GPU,
 * a state spread over several GPUs (A2AMD_DEVICES > 1): uploaded from the engine's copy on first use as
 * before (wave_id_of).  A2AMD_NO_RESIDENT=1 switches this off (A/B). */
int a2_RenderWave(void *iface, int wt, unsigned period, int flags, unsigned samplerate, unsigned length, void *props,
		int program, unsigned argc, int *argv)
{
	static int (*engine_render)(void *, int, unsigned, int, unsigned, unsigned, void *, int, unsigned, int *);
	RENDERCAP rc, *outer = rendering;
	int wh;
	if(!engine_render)
		*(void **)&engine_render = dlsym(RTLD_NEXT, "a2_RenderWave");
	if(!engine_render)
	{
		fprintf(stderr, "a2amd units: the engine's a2_RenderWave is not visible\n");
		return -A2P_INTERNAL;
	}
	memset(&rc, 0, sizeof(rc));
	rc.armed = !getenv("A2AMD_NO_RESIDENT");
	rendering = &rc;
	wh = engine_render(iface, wt, period, flags, samplerate, length, props, program, argc, argv);
	rendering = outer;
	if(rc.cap)
	{
		/* the capture waits, with the wave it belongs to, for the state's engine thread: the wave's device copy
		 * is built at its first use (wave_id_of) */
		A2P_wave *w = wh >= 0 ? a2_GetWave(iface, wh) : NULL;
		HOSTSTATE *hs;
		int kept = 0;
		pthread_mutex_lock(&states_mtx);
		for(hs = states; hs; hs = hs->next_state)
			if(hs->refs && hs->cfg && hs->cfg->interface == iface)
				break;
		pthread_mutex_unlock(&states_mtx);
		if(w && hs && hs->pendcaps_mtx_ok && (w->type == A2AMD_WWAVE || w->type == A2AMD_WMIPWAVE) &&
				w->size[0] == a2amd_capture_frames(rc.cap))
		{
			pthread_mutex_lock(&hs->pendcaps_mtx);
			if(hs->npendcaps == hs->cap_pendcaps)
			{
				int nc = hs->cap_pendcaps ? hs->cap_pendcaps * 2 : 16;
				struct PENDCAP *np = (struct PENDCAP *)realloc(hs->pendcaps, nc * sizeof(*np));
				if(np)
				{
					hs->pendcaps = np;
					hs->cap_pendcaps = nc;
				}
			}
			if(hs->npendcaps < hs->cap_pendcaps)
			{
				hs->pendcaps[hs->npendcaps].w = w;
				hs->pendcaps[hs->npendcaps].handle = wh;
				hs->pendcaps[hs->npendcaps++].cap = rc.cap;
				kept = 1;
			}
			pthread_mutex_unlock(&hs->pendcaps_mtx);
		}
		if(!kept)
		{
			if(getenv("A2AMD_WAVE_STATS"))
				fprintf(stderr, "a2amd units: a2_RenderWave: wave %d (%p), capture of %u frames not used (state %p, %u samples)\n", wh,
						(void *)w, a2amd_capture_frames(rc.cap), (void *)hs, w ? w->size[0] : 0);
			a2amd_capture_free(rc.cap);
		}
	}
	else if(getenv("A2AMD_WAVE_STATS"))
		fprintf(stderr, "a2amd units: a2_RenderWave: wave %d, nothing captured (substate %p, armed %d)\n", wh, (void *)rc.sub, rc.armed);
	return wh;
}

#ifndef A2AMD_SRCHASH
#define A2AMD_SRCHASH "unstamped"
#endif
const char *a2amd_units_source_stamp(void) { return "A2AMD_SRCHASH:" A2AMD_SRCHASH; }
