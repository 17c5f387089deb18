/*
 * a2amd.h - C ABI of the MI355X voice-render backend ("liba2amd.so")
 *
 * This is the drop-in boundary for ONE path of Audiality 2: the per-voice
 * unit-generator render loop.  The host (the unmodified Audiality 2 engine:
 * A2S VM, event scheduler, voice tree) keeps running on the CPU and forwards
 * the four callbacks of its unit plugin surface to this library, which records
 * them per fragment and evaluates all recorded unit chains on the GPU.
 *
 * Every entry point below names the reference interface it stands in for
 * (path:line under the reference tree).  Plain C types only: no torch, no HIP
 * types (a hipStream_t / device pointer travels as void*).
 *
 *   reference callback / call site                       entry point here
 *   --------------------------------------------------   -----------------------
 *   A2_unitdesc.OpenState   include/a2_units.h:159       a2amd_open
 *   A2_unitdesc.CloseState  include/a2_units.h:160       a2amd_close
 *   A2_unitdesc.Initialize  include/a2_units.h:132       a2amd_unit_init
 *        (called from a2_AddUnit, src/core.c:290)
 *   A2_unitdesc.Deinitialize include/a2_units.h:142      a2amd_unit_deinit
 *        (called from a2_DestroyUnit, src/core.c:318)
 *   A2_crdesc.write         include/a2_units.h:115       a2amd_unit_write
 *        (called from a2_VoiceControl, src/core.c:143)
 *   A2_unit.Process         include/a2_units.h:176       a2amd_unit_process
 *        (called from a2_VoiceProcess, src/core.c:1875)
 *   a2_inline_Process[Add]  src/core.c:1763-1776         a2amd_unit_process + a2amd_inline_end
 *   a2_XinsertAddClient / a2_XinsertRemoveClient         a2amd_unit_clients
 *        src/xinsertapi.c:72-157
 *   xi_process, client loop src/units/xinsert.c:95-119   a2amd_unit_inject (WRITE clients' output)
 *        xsrc_process       src/units/xsource.c:69-77    a2amd_unit_tapped (READ clients' input)
 *        xsink_Process      src/units/xsink.c:41-44
 *   a2_GetWave/A2_wave      include/a2_waves.h:88-103    a2amd_wave_upload / a2amd_wave_drop
 *   a2_AudioCallback fragment loop src/core.c:1964-1973  a2amd_fragment
 *   a2_ProcessMaster        src/core.c:1900-1907         a2amd_render (master bus -> caller)
 *
 * Numeric formats are the reference's: audio 8:24 int32, control values 16:16,
 * ramp durations 24:8 frames, sub-sample start 0..255.
 *
 * Threading: like the reference's unit callbacks, all calls for one context
 * come from one thread (the engine context of one A2 state).
 */
#ifndef A2AMD_H
#define A2AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A2AMD_MAXFRAG      64   /* A2_MAXFRAG, include/audiality2.h.cmake:50 */
#define A2AMD_MAXCHANNELS   8   /* A2_MAXCHANNELS, include/audiality2.h.cmake:56 */
#define A2AMD_MIPLEVELS    10   /* A2_MIPLEVELS, include/a2_waves.h:33 */
#define A2AMD_WAVEPRE       1   /* A2_WAVEPRE,  include/a2_waves.h:60 */
#define A2AMD_WAVEPOST    131   /* A2_WAVEPOST, include/a2_waves.h:63-64 */
#define A2AMD_MAXCHAIN     16   /* max units in one voice's chain (ours; 8 until round 6.  The reference's list has no cap) */

typedef struct a2amd_ctx a2amd_ctx;

/* Error codes (negative returns).  0 = OK. */
typedef enum a2amd_errors {
	A2AMD_OK = 0,
	A2AMD_ENODEVICE   = -1,  /* no HIP device / HIP runtime failure       */
	A2AMD_EINVAL      = -2,  /* bad argument                               */
	A2AMD_ENOMEM      = -3,
	A2AMD_EUNSUPPORTED= -4,  /* chain / mode outside the hot-path scope    */
	A2AMD_ESTATE      = -5,  /* call out of sequence                       */
	A2AMD_EHIP        = -6   /* a HIP call failed; see a2amd_last_error()  */
} a2amd_errors;

/* Unit kinds on the hot path. */
typedef enum a2amd_unitkind {
	A2AMD_WTOSC = 0,     /* src/units/wtosc.c    regs: w p a phase            */
	A2AMD_PANMIX,        /* src/units/panmix.c   regs: vol pan                */
	A2AMD_FILTER12,      /* src/units/filter12.c regs: cutoff q lp bp hp      */
	A2AMD_FBDELAY,       /* src/units/fbdelay.c  regs: fbdelay ldelay rdelay drygain fbgain lgain rgain */
	A2AMD_INLINE,        /* src/units/inline.c + src/core.c:1763-1776         */
	A2AMD_XINSERT,       /* src/units/xinsert.c (bypass; clients: a2amd_unit_clients) */
	/* the FM oscillators of src/units/fm.c.  regs, in
	 * order: phase, then p a fb / p1 a1 fb1 / p2 a2 fb2 / p3 a3 fb3 for as
	 * many operators as the unit has (fm.c:54-79). */
	A2AMD_FM1,           /* fm.c:532-552  o0 ->                               */
	A2AMD_FM2,           /* fm.c:579-599  o1 -> o0 ->                         */
	A2AMD_FM3,           /* fm.c:631-651  o2 -> o1 -> o0 ->                   */
	A2AMD_FM4,           /* fm.c:688-708  o3 -> o2 -> o1 -> o0 ->             */
	A2AMD_FM3P,          /* fm.c:719-739  (o1 + o2) -> o0 ->                  */
	A2AMD_FM4P,          /* fm.c:752-772  (o1 + o2 + o3) -> o0 ->             */
	A2AMD_FM2R,          /* fm.c:783-803  o0 * o1 (ring modulator)            */
	A2AMD_FM4R,          /* fm.c:814-834  (o2 -> o0) * (o3 -> o1)             */
	/* the small units (env is not built) */
	A2AMD_DC,            /* src/units/dc.c         regs: value mode  (STEP=0, LINEAR=1<<16) */
	A2AMD_WAVESHAPER,    /* src/units/waveshaper.c regs: amount                        */
	A2AMD_DCBLOCK,       /* src/units/dcblock.c    regs: cutoff                        */
	A2AMD_LIMITER,       /* src/units/limiter.c    regs: release threshold             */
	/* the other two hosts of stream / callback clients
	 * (a2amd_unit_clients): inputs only, outputs only */
	A2AMD_XSINK,         /* src/units/xsink.c:27-46                                    */
	A2AMD_XSOURCE,       /* src/units/xsource.c:27-137                                 */
	A2AMD_NKINDS
} a2amd_unitkind;

/* A2_unitflags bit the units care about (include/a2_units.h:72). */
#define A2AMD_PROCADD 0x00000001u

/* A2_wavetypes (include/a2_waves.h:79-85) and A2_LOOPED (:112). */
#define A2AMD_WOFF     0
#define A2AMD_WNOISE   1
#define A2AMD_WWAVE    2
#define A2AMD_WMIPWAVE 3
#define A2AMD_LOOPED   0x00000100u

typedef struct a2amd_config {
	uint32_t struct_size;   /* sizeof(a2amd_config), for ABI growth          */
	int32_t  samplerate;    /* A2_config.samplerate                           */
	int32_t  basepitch;     /* A2_config.basepitch (16:16), audiality2.c:398  */
	int32_t  channels;      /* master bus channels, 1..8                      */
	int32_t  device;        /* HIP device ordinal                             */
	uint32_t max_batch;     /* max fragments recorded between renders (>=1)   */
	void    *stream;        /* hipStream_t to launch on; NULL = own stream    */
} a2amd_config;

/* Mirror of A2_wave (include/a2_waves.h:88-103); data[] point at the first pad
 * sample (i.e. the reference's w->d.wave.data[i], which has A2_WAVEPRE samples
 * before and A2_WAVEPOST after the size[i] payload samples). */
typedef struct a2amd_wavedesc {
	int32_t  type;                      /* A2AMD_W*                           */
	uint32_t flags;                     /* A2AMD_LOOPED ...                   */
	uint32_t period;
	uint32_t size[A2AMD_MIPLEVELS];     /* payload samples per level          */
	const int16_t *data[A2AMD_MIPLEVELS];
} a2amd_wavedesc;

/* ---- state ---------------------------------------------------------------*/
int  a2amd_open(const a2amd_config *cfg, a2amd_ctx **out);
void a2amd_close(a2amd_ctx *ctx);
const char *a2amd_last_error(const a2amd_ctx *ctx);   /* ctx may be NULL */
const char *a2amd_version(void);
/* "A2AMD_SRCHASH:<32 hex>": hash of the sources this library was built from (the build recipe
 * compares it with the tree, audiality2_amd/build.py; liba2amd_units.so and liba2amd_walk.so
 * carry the same kind of stamp). */
const char *a2amd_source_stamp(void);
/* Number of HIP devices a2amd_config.device may name (0 without a GPU). */
int  a2amd_device_count(void);

/* The 64x{base,coeff} table of a2_P2I (src/pitch.c:57-96).  a2amd_open builds
 * it with the host's powf exactly as the reference does; a host that wants the
 * engine's own table bit-for-bit may overwrite it (128 uint32). */
int  a2amd_set_pitch_table(a2amd_ctx *ctx, const uint32_t *tab128);
int  a2amd_get_pitch_table(const a2amd_ctx *ctx, uint32_t *tab128);

/* ---- waves ---------------------------------------------------------------*/
/* Upload all mip levels incl. pads.  'key' is whatever identifies the wave on
 * the host (the A2_wave pointer); returns a wave id >= 0 to be used as the
 * value of a wtosc 'w' write.  Re-uploading an existing key refreshes it. */
int  a2amd_wave_upload(a2amd_ctx *ctx, uint64_t key, const a2amd_wavedesc *w);
/* Host wave was unloaded (size[0] == 0, src/waves.c:717-723): oscillators
 * still pointing at it fall silent like wtosc_check_unloaded (wtosc.c:168). */
int  a2amd_wave_drop(a2amd_ctx *ctx, uint64_t key);

/* ---- a wave the device rendered stays on the device ------------------------------
 * a2_RenderWave (src/render.c:144-177) runs a program in an off-line one-channel substate and writes what
 * the substate's driver delivers into a new wave (a2_Write(A2_I24), render.c:91; conversion, pads and
 * mip levels when the stream closes: src/waves.c:155-237, :89-130, :513-527).  When the substate's units
 * are the drop-in's, the samples exist in device memory first:
 *   a2amd_capture_begin(sub)        from here on channel 0 of every batch the context renders is also kept,
 *                                   device to device, in a capture buffer (frames closed up)
 *   a2amd_capture_end(sub, &cap)    detaches the buffer (it outlives the context; *cap = NULL: nothing rendered)
 *   a2amd_wave_upload_captured(ctx, key, w, cap)
 *                                   as a2amd_wave_upload(), with level 0 converted from the capture on the device
 *                                   (>> 8 into int16, waves.c:174-177), pads (a2_fix_pad) and mip levels
 *                                   (a2_render_mipmaps) built there: w->data[] is not read, nothing is copied
 *                                   from the host.  w->size[0] must be the capture's frame count and the flags
 *                                   free of A2_NORMALIZE / A2_XFADE / A2_REVMIX (the host-side post-processing
 *                                   of waves.c:300-390 is not restated): A2AMD_EUNSUPPORTED otherwise, and for a
 *                                   capture on another GPU - the caller uploads the engine's copy instead. */
typedef struct a2amd_capture a2amd_capture;
int  a2amd_capture_begin(a2amd_ctx *ctx);
int  a2amd_capture_end(a2amd_ctx *ctx, a2amd_capture **cap);
unsigned a2amd_capture_frames(const a2amd_capture *cap);
void a2amd_capture_free(a2amd_capture *cap);
int  a2amd_wave_upload_captured(a2amd_ctx *ctx, uint64_t key, const a2amd_wavedesc *w, const a2amd_capture *cap);
/* wave data copied from the host so far (bytes, waves) and waves built from captures */
int  a2amd_wave_stats(a2amd_ctx *ctx, uint64_t *h2d_bytes, uint32_t *uploaded, uint32_t *resident);

/* ---- fragment clock --------------------------------------------------------*/
/* Begin the next fragment of 'frames' (1..64) sample frames; all following
 * init/write/process calls belong to it (src/core.c:1964-1973). */
int  a2amd_fragment(a2amd_ctx *ctx, unsigned frames);
/* Where the open fragment starts inside the ENGINE's own fragment: the 'offset' a2_VoiceProcess hands the
 * units (src/core.c:1856-1876) - 0, unless the root voice's program woke in mid-fragment and the engine's
 * fragment became two of the backend's.  Only the env units of voices the device VM runs need it (they pass
 * it on as the start of their writes, src/units/env.c:131); default 0. */
int  a2amd_fragment_offset(a2amd_ctx *ctx, unsigned offset);

/* ---- unit callbacks --------------------------------------------------------*/
/*
 * Initialize.  'voice_key' identifies the owning voice (the A2_vmstate pointer
 * handed to Initialize); consecutive inits with the same key form that voice's
 * chain in order (a2_PopulateVoice, src/core.c:350-420).  'wired_out' != 0 when
 * the unit's outputs are the voice's output bus (A2_IO_WIREOUT, core.c:240-243)
 * rather than the scratch bus.  'transpose' = vms->r[R_TRANSPOSE],
 * 'wakefrac' = vms->waketime & 0xff (wtosc.c:400-407).
 * Returns a unit id >= 0.
 */
int  a2amd_unit_init(a2amd_ctx *ctx, uint64_t voice_key, int kind,
		unsigned flags, int ninputs, int noutputs, int wired_out,
		int transpose, unsigned wakefrac);
int  a2amd_unit_deinit(a2amd_ctx *ctx, int unit);

/* Control register write; 'reg' indexes the unit's register table in the
 * reference's order.  For wtosc reg 0 ('w') 'value' is a wave id from
 * a2amd_wave_upload() (or -1 for no wave) instead of a 16:16 handle.
 * 'transpose' = current vms->r[R_TRANSPOSE] (read by wtosc 'p' and filter12
 * 'cutoff', wtosc.c:489, filter12.c:144). */
int  a2amd_unit_write(a2amd_ctx *ctx, int unit, int reg, int value,
		unsigned start, unsigned duration, int transpose);

/* Process [offset, offset+frames) of the current fragment.  'noisestate'
 * points at the engine-global RNG word (A2_state.noisestate, internals.h:682)
 * or NULL; a wtosc in noise mode consumes draws from it in call order exactly
 * like wtosc_noise (wtosc.c:129-152), so VM RAND instructions interleave
 * correctly. */
int  a2amd_unit_process(a2amd_ctx *ctx, int unit, unsigned offset,
		unsigned frames, uint32_t *noisestate);
/* The voice walk's hot path (a2_VoiceProcess, src/core.c:1875-1876: one Process call
 * per unit and window).  a2amd_voice_process() = a2amd_unit_process() on every unit
 * of the voice 'head_unit' belongs to, in chain order, for one window (voices
 * without an inline unit).  It returns 1 when, from the next fragment on, the host
 * may report this voice's DEFAULT window - Process(0, all frames of the fragment)
 * on each unit and nothing else: what a voice whose VM sleeps gets - by storing 1 at
 * a2amd_default_map()[a2amd_voice_slot()] instead of calling anything: one byte
 * store per voice and fragment, which is what lets one engine thread walk tens of
 * thousands of sleeping voices.  (0: keep calling - a noise oscillator, a ramping
 * pitch or cutoff, clients.)  Any unit_write / unit_deinit on the voice ends the
 * permission until voice_process() grants it again; a mark followed by such a call
 * in the same fragment is handled (the window is recorded first).  The map's
 * address is valid for the open fragment only: ask once per fragment. */
int  a2amd_voice_process(a2amd_ctx *ctx, int head_unit, unsigned offset, unsigned frames,
		uint32_t *noisestate);
int  a2amd_voice_slot(a2amd_ctx *ctx, int unit);
/* The same for ANY voice, bus owners included (a voice with an inline unit: a2_NewGroup's driver, a
 * delay bus): 1 when the default window of the voice 'unit' belongs to - Process(0, all frames) on
 * each unit, a2amd_inline_end() included - may be reported by the byte store alone.  That is the case
 * once the voice has been processed at least once and while none of its units needs per-call host
 * work (noise, a ramping cutoff, clients).  A host that walks the voice tree itself and finds a
 * whole subtree asleep (src/core.c:1883-1896 would visit every voice of it) marks the subtree's
 * voices and makes no call at all.  Voices that get records in the same fragment must be
 * processed by calls, parents before children (their output bus is resolved from the open inline
 * windows). */
int  a2amd_voice_markable(a2amd_ctx *ctx, int unit);
/* ... and for as long as the host says: the n voices slots[0..n) (or, with slots == NULL, the range
 * [lo, lo + n) of slots) count as having stored their byte in EVERY fragment from the open one on
 * (on != 0) until the hold is released (on == 0), the voice is processed by a call, gets a record,
 * or dies.  A host that has found a subtree asleep until some known time does nothing at all for it
 * per fragment.  a2amd_default_release_all() ends every hold of the context - from the OPEN fragment
 * on: a host that releases in the middle of its walk stores the bytes of the voices whose turn had
 * already come in this fragment itself. */
int  a2amd_default_hold(a2amd_ctx *ctx, const uint32_t *slots, unsigned n, unsigned lo, int on);
int  a2amd_default_release_all(a2amd_ctx *ctx);
uint8_t *a2amd_default_map(a2amd_ctx *ctx, unsigned *nslots);

/* Clients of an A2AMD_XINSERT / A2AMD_XSINK / A2AMD_XSOURCE unit
 * (a2_XinsertAddClient, src/xinsertapi.c:72-111; served by xi_process,
 * src/units/xinsert.c:60-142, xsink_Process, xsink.c:27-46, xsrc_process,
 * xsource.c:43-78).  The client callbacks stay
 * with the host; 'mode' says what the unit does for them from its next window
 * on: A2AMD_XIO_TAP leaves every window's input where a2amd_unit_tapped() finds
 * it after a READBACK render (what READ-only clients are handed),
 * A2AMD_XIO_INJECT adds what a2amd_unit_inject() collected to the unit's output
 * (what WRITE-only clients produced).
 * Insert clients (READ and WRITE: a2_InsertCallback; xi_process hands each of them
 * a copy of the input and emits the sum of what they made of it INSTEAD of the
 * input, xinsert.c:95-123) need the voice's audio on the host in the middle of
 * the render.  That is served where the render has a seam: on a unit for which
 * a2amd_unit_insertable() says 1 - the last unit of a voice directly below the root
 * voice (the xinsert of a2_NewGroup's driver, audiality2.c:283-302), wired and
 * adding.  Mode A2AMD_XIO_TAP | A2AMD_XIO_MUTE: the unit's input is tapped and NOT
 * passed on; the host renders a2amd_render(UPLOAD | SUBTREES | TAPS), reads the
 * windows with a2amd_unit_tapped(), runs the clients and hands the sum of their
 * outputs back with a2amd_unit_insert(), then a2amd_render(ROOT | READBACK) adds
 * it to the voice's output before the root chain runs. */
#define A2AMD_XIO_TAP     1u
#define A2AMD_XIO_INJECT  2u
#define A2AMD_XIO_MUTE    4u
int  a2amd_unit_clients(a2amd_ctx *ctx, int unit, unsigned mode);
int  a2amd_unit_insertable(a2amd_ctx *ctx, int unit);
int  a2amd_render_paused(a2amd_ctx *ctx);
/* ... window [offset, offset+frames) of fragment 'fragment' of the batch being
 * rendered (between its SUBTREES | TAPS and its ROOT phase); adds up over calls */
int  a2amd_unit_insert(a2amd_ctx *ctx, int unit, unsigned fragment, unsigned offset, unsigned frames,
		const int32_t *const *bufs);
/* Add 'frames' frames per input channel (bufs[ch][0..frames)) to what the unit
 * emits over [offset, offset+frames) of the open fragment - call it before the
 * a2amd_unit_process() of that window. */
int  a2amd_unit_inject(a2amd_ctx *ctx, int unit, unsigned offset, unsigned frames,
		const int32_t *const *bufs);
/* After a2amd_render(... READBACK): bufs[ch] = the 64 frames the unit's inputs
 * carried in fragment 'fragment' of that batch, where it had READ clients
 * (valid until the next render).  Returns the number of channels. */
int  a2amd_unit_tapped(a2amd_ctx *ctx, int unit, unsigned fragment, const int32_t **bufs);

/* Closes the window opened by a2amd_unit_process() on an A2AMD_INLINE unit,
 * i.e. the return of a2_ProcessSubvoices (src/core.c:1769,1775). */
int  a2amd_inline_end(a2amd_ctx *ctx, int unit);

/* ---- render ----------------------------------------------------------------*/
#define A2AMD_RENDER_SUBTREES 1u  /* kernels: everything below the root voice */
#define A2AMD_RENDER_ROOT     2u  /* kernels: root voice chain -> master bus  */
#define A2AMD_RENDER_UPLOAD   4u  /* ship the recorded commands to the GPU    */
#define A2AMD_RENDER_READBACK 8u  /* master bus -> out[], wait for the GPU    */
#define A2AMD_RENDER_TAPS     64u /* with SUBTREES, without ROOT: wait for the GPU and fetch the batch's taps (a2amd_unit_tapped) */
#define A2AMD_RENDER_KEEP    16u  /* keep the recording (re-run it next call) */
#define A2AMD_RENDER_ASYNC   32u  /* READBACK does not wait: a2amd_collect() delivers */
#define A2AMD_RENDER_EXCHANGE 128u /* a batch whose subtrees were rendered in steps (paused for insert clients): a2amd_render()
				    * adds what the clients wrote since the last pause to the voices' buses and launches nothing;
				    * a2amd_render_group() does that in every context, then the exchange of the root-bus partials */
#define A2AMD_RENDER_ALL     15u
/*
 * Evaluate the recorded fragments on the GPU.  With A2AMD_RENDER_ALL the
 * master bus of every recorded fragment is written, planar, to
 * out[c][0..total_frames) (host pointers; a2_ProcessMaster) and the recording
 * restarts.  The phases may be issued by separate calls, in the order UPLOAD,
 * SUBTREES, ROOT, READBACK; out may be NULL unless READBACK is requested.
 * Multi-GPU: run UPLOAD|SUBTREES, reduce a2amd_rootbus() across ranks (int32
 * sum is order independent), then ROOT|READBACK on the rank that owns the root
 * chain.  With KEEP the recording and its uploaded form survive the call, so
 * the next SUBTREES|ROOT renders the NEXT batch of audio from the same command
 * stream (what an engine whose VMs all sleep would record again).
 * Returns the number of frames in the batch or a negative error.
 */
int  a2amd_render(a2amd_ctx *ctx, unsigned phases, int32_t *const *out,
		unsigned out_capacity_frames);
/* Deliver the oldest batch rendered with A2AMD_RENDER_READBACK | A2AMD_RENDER_ASYNC
 * (a2_ProcessMaster, src/core.c:1900-1907, one batch late): waits for its copy,
 * writes out[c][0..frames) and returns the frame count, 0 when none is in
 * flight.  At most two may be: render, render, collect, render, collect ...
 * keeps the GPU busy while the host consumes the previous batch (an offline
 * a2_Run() loop whose caller reads buffer k while buffer k+1 renders). */
int  a2amd_collect(a2amd_ctx *ctx, int32_t *const *out, unsigned out_capacity_frames);
/* Re-run the uploaded, record-free batch 'steps' more times (each run renders
 * the next batch of audio; see A2AMD_RENDER_KEEP).  The launch sequence is
 * captured once into a hipGraph and replayed, 8 runs per graph launch, unless
 * profiling is on (then every run is bracketed with events instead). */
int  a2amd_replay(a2amd_ctx *ctx, unsigned steps);
/* Device pointer + size (bytes) of the root voice's inline bus partials for
 * the fragments of the current batch: int32 [batch][channels][64]. */
int  a2amd_rootbus(a2amd_ctx *ctx, void **devptr, uint64_t *bytes);
/* Copy the root-bus partials of the current batch to (to_stage != 0) or from a
 * device staging buffer of that size, on the context's stream: lets a host park
 * several batches' partials and sum them across ranks with one collective
 * (audiality2_amd/shard.py: GroupedRootReduce). */
int  a2amd_rootbus_copy(a2amd_ctx *ctx, void *stage, int to_stage);

/* ---- multi-GPU ----------------------------------------*/
/* One context per GPU (one process per GPU, or several contexts in one process),
 * each recording and rendering the voice SUBTREES it is given - the children of the
 * root voice, e.g. the a2_NewGroup groups (src/interface.c:888), dealt over the
 * contexts by the host - under its own copy of the root voice.  The only exchange is
 * the root voice's inline bus: a2amd_render(... SUBTREES | ROOT ...) on a context
 * that went through a2amd_dist_init() renders its subtrees, sums the bus over all
 * ranks with ONE ncclReduce(int32, sum) per batch on the context's stream (RCCL over
 * xGMI, called from this library; wrap-around integer sums give the same bits in
 * any order), and runs the root chain - whose panmix truncates, so it must see the
 * sum - on rank 0, where READBACK / a2amd_collect deliver the audio; on the other
 * ranks READBACK delivers nothing.  The 128 byte id comes from a2amd_dist_unique_id()
 * on one rank and reaches the others through whatever channel the host application
 * has (MPI, a socket, torch.distributed's store). */
int  a2amd_dist_unique_id(void *id128);
int  a2amd_dist_init(a2amd_ctx *ctx, const void *id128, int rank, int nranks);
/* The same inside ONE process - one engine state spread over several GPUs: the n
 * contexts (one per GPU, ctxs[0] owning the root chain) form a group
 * (ncclCommInitAll), and a2amd_render_group() renders a batch all of them recorded:
 * every context's subtrees, the one reduce (an RCCL group call from the calling
 * thread), the root chain + READBACK on ctxs[0].  Each context must have been given
 * the same fragments (a2amd_fragment) and its own copy of the root voice.  (Contexts
 * that share a GPU - a single-GPU test box - exchange with a device-local add: RCCL
 * does not take two ranks on one device.) */
int  a2amd_dist_init_local(a2amd_ctx *const *ctxs, int n);
int  a2amd_render_group(a2amd_ctx *const *ctxs, int n, unsigned phases, int32_t *const *out,
		unsigned out_capacity_frames);

/* Begin 'count' further fragments of 'frames' frames in which the engine's
 * voice walk finds every VM asleep: each live voice gets exactly one
 * Process(0, frames) per unit and nothing else happens (src/core.c:1852-1878
 * with no wake-ups).  Fails with A2AMD_EUNSUPPORTED while a noise oscillator,
 * a ramping filter cutoff or a unit with clients (a2amd_unit_clients) needs
 * per-call host work.  (An oscillator may be switched to noise after such a
 * stretch: the host keeps no copy of a wavetable oscillator's phase, it fetches
 * the device's when the switch comes.) */
int  a2amd_fragment_repeat(a2amd_ctx *ctx, unsigned frames, unsigned count);

/* ---- introspection (tests, bench) --------------------------------------*/
typedef struct a2amd_stats {
	uint64_t fragments;        /* fragments rendered so far                 */
	uint64_t voice_fragments;  /* sum over fragments of voices processed    */
	uint64_t records;          /* command records shipped                   */
	uint64_t launches;         /* kernel launches                           */
	double   last_kernel_ms;   /* HIP-event time of the last batch's kernels */
	double   last_leaf_ms;     /* ... of the leaf-voice kernel alone        */
	uint32_t live_units, live_voices, live_waves, reserved;
	/* with a2amd_set_profiling(ctx, 1): HIP-event times, summed over every
	 * batch rendered since profiling was switched on */
	double   timed_leaf_ms;    /* the leaf-voice kernel                     */
	double   timed_all_ms;     /* all kernels of a batch                    */
	uint64_t timed_batches;
} a2amd_stats;
/* Waits for the stream when profiling is on (event times are read back). */
int  a2amd_get_stats(a2amd_ctx *ctx, a2amd_stats *st);
/* Bracket every batch's kernels with HIP events on the launch stream; the sums
 * appear in a2amd_stats.  Switching it on resets the sums. */
int  a2amd_set_profiling(a2amd_ctx *ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* A2AMD_H */
