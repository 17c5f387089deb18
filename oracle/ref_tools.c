/*
 * ref_tools.c - harness around the COMPILED REFERENCE (oracle/_ref).
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile against the reference's
 * public headers and linked with oracle/_ref/libaudiality2.so; never shipped,
 * never on the product path.  It does two things:
 *
 *   ref_tools dump  <outdir>
 *       Probes the reference's own a2_P2I() and dumps its built-in wave
 *       objects (all mip levels incl. pads) so the restatement in a2o.c can be
 *       checked against them.
 *
 *   ref_tools trace <file.a2s> <program> <frames> <buffer> <rate> <channels>
 *                   <out.trace> <out.pcm> [args...]
 *       Runs <program> through the unmodified engine, offline, and captures
 *       (a) the engine's real output and (b) every call the engine makes
 *       through the unit plugin surface for the hot-path units, in order.
 *       Capture works by symbol interposition: this executable defines
 *       a2_wtosc_unitdesc & co, so the engine's unit table (audiality2.c:183)
 *       binds to the wrappers below, which log the call and then forward to
 *       the reference's own unit (found with dlsym(RTLD_NEXT)).  The engine
 *       therefore still renders with its own DSP; the log is a by-product.
 *
 * The trace is the common input of the CPU restatement (a2o.c) and of the
 * GPU backend: replaying it must reproduce <out.pcm> bit for bit.
 *
 * Trace format: little-endian int32 records of 8 words {op, a, b, c, d, e, f,
 * g}; a WAVE record is followed by its int16 payload padded to 4 bytes.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "audiality2.h"
#include "a2_units.h"
#include "a2_waves.h"
#include "a2_properties.h"
#include "a2_drivers.h"

enum { T_FRAGMENT = 1, T_INIT, T_DEINIT, T_WRITE, T_PROCESS, T_INLINE_END,
		T_WAVE, T_CONFIG, T_WAVEDROP };
enum { K_WTOSC = 0, K_PANMIX, K_FILTER12, K_FBDELAY, K_INLINE, K_XINSERT,
		K_FM1, K_FM2, K_FM3, K_FM4, K_FM3P, K_FM4P, K_FM2R, K_FM4R,
		K_DC, K_WAVESHAPER, K_DCBLOCK, K_LIMITER,
		K_COUNT };	/* = a2amd_unitkind, include/a2amd.h */

static FILE *tracef;
static int trace_muted;		/* set while a2_Load() renders waves in a substate */
static A2_interface *g_iface;
static int next_uid, next_voice;
static A2_vmstate *last_init_vms;
static int chain_open;

static void rec(int op, int a, int b, int c, int d, int e, int f, int g)
{
	int32_t r[8] = { op, a, b, c, d, e, f, g };
	if(tracef && !trace_muted)
		fwrite(r, sizeof(r), 1, tracef);
}

/* ---- wave registry: A2_wave* -> trace wave id ---------------------------*/
#define MAXWAVES 1024
static A2_wave *wave_ptr[MAXWAVES];
static int nwaves;

static int wave_id(A2_wave *w)
{
	int i, levels;
	if(!w || trace_muted)
		return -1;
	for(i = 0; i < nwaves; ++i)
		if(wave_ptr[i] == w)
			return i;
	if(nwaves >= MAXWAVES)
		return -1;
	wave_ptr[nwaves] = w;
	levels = w->type == A2_WMIPWAVE ? A2_MIPLEVELS :
			w->type == A2_WWAVE ? 1 : 0;
	rec(T_WAVE, nwaves, w->type, w->flags, w->period, levels, 0, 0);
	if(tracef)
	{
		uint32_t sizes[A2_MIPLEVELS];
		memset(sizes, 0, sizeof(sizes));
		for(i = 0; i < levels; ++i)
			sizes[i] = w->d.wave.size[i];
		fwrite(sizes, sizeof(sizes), 1, tracef);
		for(i = 0; i < levels; ++i)
		{
			size_t n = A2_WAVEPRE + sizes[i] + A2_WAVEPOST;
			int16_t z = 0;
			fwrite(w->d.wave.data[i], 2, n, tracef);
			if(n & 1)
				fwrite(&z, 2, 1, tracef);
		}
	}
	return nwaves++;
}

/* ---- per-instance bookkeeping, kept at the tail of the 384 byte block ---*/
typedef struct XTRA
{
	int		uid;
	int		kind;
	A2_vmstate	*vms;
	A2_process_cb	orig_process;
} XTRA;

#define BLOCKSIZE 384	/* A2_BLOCK_SIZE */
static inline XTRA *xtra(A2_unit *u)
{
	return (XTRA *)((char *)u + BLOCKSIZE - sizeof(XTRA));
}

static const A2_unitdesc *orig[K_COUNT];
static const char *orig_sym[K_COUNT] = {
	"a2_wtosc_unitdesc", "a2_panmix_unitdesc", "a2_filter12_unitdesc",
	"a2_fbdelay_unitdesc", "a2_inline_unitdesc", "a2_xinsert_unitdesc",
	"a2_fm1_unitdesc", "a2_fm2_unitdesc", "a2_fm3_unitdesc",
	"a2_fm4_unitdesc", "a2_fm3p_unitdesc", "a2_fm4p_unitdesc",
	"a2_fm2r_unitdesc", "a2_fm4r_unitdesc",
	"a2_dc_unitdesc", "a2_waveshaper_unitdesc", "a2_dcblock_unitdesc",
	"a2_limiter_unitdesc"
};

static const A2_unitdesc *get_orig(int k)
{
	if(!orig[k])
	{
		orig[k] = (const A2_unitdesc *)dlsym(RTLD_NEXT, orig_sym[k]);
		if(!orig[k])
		{
			fprintf(stderr, "ref_tools: no %s in the reference!\n",
					orig_sym[k]);
			exit(2);
		}
		if(orig[k]->instancesize + sizeof(XTRA) > BLOCKSIZE)
		{
			fprintf(stderr, "ref_tools: no room in %s\n", orig_sym[k]);
			exit(2);
		}
	}
	return orig[k];
}

static unsigned noise_state(void)
{
	int v = 0;
	a2_GetStateProperty(g_iface, A2_PNOISESEED, &v);
	return (unsigned)v;
}

static void tr_process(A2_unit *u, unsigned offset, unsigned frames)
{
	XTRA *x = xtra(u);
	unsigned before = x->kind == K_WTOSC ? noise_state() : 0;
	chain_open = 0;
	if(x->kind == K_WTOSC)
	{
		/* log after the call so the record can carry the state after */
		u->Process = x->orig_process;
		x->orig_process(u, offset, frames);
		rec(T_PROCESS, x->uid, offset, frames, 1, before, noise_state(), 0);
	}
	else
	{
		rec(T_PROCESS, x->uid, offset, frames, 0, 0, 0, 0);
		u->Process = x->orig_process;
		x->orig_process(u, offset, frames);
		if(x->kind == K_INLINE)
			rec(T_INLINE_END, x->uid, 0, 0, 0, 0, 0, 0);
	}
	x->orig_process = u->Process;	/* the unit may have swapped it */
	u->Process = tr_process;
}

static void tr_write(A2_unit *u, int reg, int v, unsigned start, unsigned dur)
{
	XTRA *x = xtra(u);
	const A2_unitdesc *od = get_orig(x->kind);
	int lv = v;
	chain_open = 0;
	if(x->kind == K_WTOSC && reg == 0)
		lv = wave_id(a2_GetWave(g_iface, v >> 16));
	rec(T_WRITE, x->uid, reg, lv, start, dur, x->vms->r[R_TRANSPOSE], 0);
	u->Process = x->orig_process;
	od->registers[reg].write(u, v, start, dur);
	x->orig_process = u->Process;
	u->Process = tr_process;
}

#define WR(n) static void tr_write##n(A2_unit *u, int v, unsigned s, unsigned d) \
	{ tr_write(u, n, v, s, d); }
WR(0) WR(1) WR(2) WR(3) WR(4) WR(5) WR(6) WR(7) WR(8) WR(9) WR(10) WR(11) WR(12)

static A2_errors tr_init(int kind, A2_unit *u, A2_vmstate *vms, void *sd,
		unsigned flags)
{
	const A2_unitdesc *od = get_orig(kind);
	XTRA *x = xtra(u);
	A2_errors res;
	int wired = u->outputs != u->inputs;
	if(!chain_open || vms != last_init_vms)
		++next_voice;
	last_init_vms = vms;
	chain_open = 1;
	x->uid = next_uid++;
	x->kind = kind;
	x->vms = vms;
	rec(T_INIT, x->uid, next_voice, kind, flags,
			u->ninputs | (u->noutputs << 8) | (wired << 16),
			vms->r[R_TRANSPOSE], vms->waketime & 0xff);
	if((res = od->Initialize(u, vms, sd, flags)))
		return res;
	x->orig_process = u->Process;
	u->Process = tr_process;
	return A2_OK;
}

static void tr_deinit_k(int kind, A2_unit *u)
{
	const A2_unitdesc *od = get_orig(kind);
	chain_open = 0;
	rec(T_DEINIT, xtra(u)->uid, 0, 0, 0, 0, 0, 0);
	if(od->Deinitialize)
		od->Deinitialize(u);
}

static A2_errors tr_open_k(int kind, A2_config *cfg, void **sd)
{
	const A2_unitdesc *od = get_orig(kind);
	if(od->OpenState)
		return od->OpenState(cfg, sd);
	*sd = NULL;
	return A2_OK;
}

static void tr_close_k(int kind, void *sd)
{
	const A2_unitdesc *od = get_orig(kind);
	if(od->CloseState)
		od->CloseState(sd);
}

#define KIND_FUNCS(K, name) \
static A2_errors name##_init(A2_unit *u, A2_vmstate *vms, void *sd, unsigned f) \
	{ return tr_init(K, u, vms, sd, f); } \
static void name##_deinit(A2_unit *u) { tr_deinit_k(K, u); } \
static A2_errors name##_open(A2_config *cfg, void **sd) { return tr_open_k(K, cfg, sd); } \
static void name##_close(void *sd) { tr_close_k(K, sd); }

KIND_FUNCS(K_WTOSC, wtosc)
KIND_FUNCS(K_PANMIX, panmix)
KIND_FUNCS(K_FILTER12, filter12)
KIND_FUNCS(K_FBDELAY, fbdelay)
KIND_FUNCS(K_INLINE, inl)
KIND_FUNCS(K_XINSERT, xins)
KIND_FUNCS(K_FM1, fm1)
KIND_FUNCS(K_FM2, fm2)
KIND_FUNCS(K_FM3, fm3)
KIND_FUNCS(K_FM4, fm4)
KIND_FUNCS(K_FM3P, fm3p)
KIND_FUNCS(K_FM4P, fm4p)
KIND_FUNCS(K_FM2R, fm2r)
KIND_FUNCS(K_FM4R, fm4r)
KIND_FUNCS(K_DC, dc)
KIND_FUNCS(K_WAVESHAPER, wshaper)
KIND_FUNCS(K_DCBLOCK, dcblock)
KIND_FUNCS(K_LIMITER, limiter)

static const A2_crdesc wtosc_regs[] = {
	{ "w", tr_write0 }, { "p", tr_write1 }, { "a", tr_write2 },
	{ "phase", tr_write3 }, { NULL, NULL } };
static const A2_crdesc panmix_regs[] = {
	{ "vol", tr_write0 }, { "pan", tr_write1 }, { NULL, NULL } };
static const A2_constdesc panmix_consts[] = {
	{ "CENTER", 0 }, { "LEFT", (-1) << 16 }, { "RIGHT", 1 << 16 },
	{ NULL, 0 } };
static const A2_crdesc filter12_regs[] = {
	{ "cutoff", tr_write0 }, { "q", tr_write1 }, { "lp", tr_write2 },
	{ "bp", tr_write3 }, { "hp", tr_write4 }, { NULL, NULL } };
static const A2_crdesc fbdelay_regs[] = {
	{ "fbdelay", tr_write0 }, { "ldelay", tr_write1 }, { "rdelay", tr_write2 },
	{ "drygain", tr_write3 }, { "fbgain", tr_write4 }, { "lgain", tr_write5 },
	{ "rgain", tr_write6 }, { NULL, NULL } };

/* fm.c:509-530,562-577,603-629,654-686: a prefix of one list */
#define FM_REGS(n) static const A2_crdesc fm##n##_regs[] = { \
	{ "phase", tr_write0 }, { "p", tr_write1 }, { "a", tr_write2 }, \
	{ "fb", tr_write3 }, { n > 1 ? "p1" : NULL, n > 1 ? tr_write4 : NULL }, \
	{ "a1", tr_write5 }, { "fb1", tr_write6 }, \
	{ n > 2 ? "p2" : NULL, n > 2 ? tr_write7 : NULL }, { "a2", tr_write8 }, \
	{ "fb2", tr_write9 }, { n > 3 ? "p3" : NULL, n > 3 ? tr_write10 : NULL }, \
	{ "a3", tr_write11 }, { "fb3", tr_write12 }, { NULL, NULL } };
FM_REGS(1) FM_REGS(2) FM_REGS(3) FM_REGS(4)

/* The interposing descriptors.  Same names / limits as the originals
 * (wtosc.c:516, panmix.c:313, filter12.c:241, fbdelay.c:289, inline.c:50,
 * xinsert.c:232); instancesize claims the whole block so XTRA fits. */
const A2_unitdesc a2_wtosc_unitdesc = { "wtosc", 0, wtosc_regs, NULL, NULL,
	0, 0, 1, 1, BLOCKSIZE, wtosc_init, wtosc_deinit, wtosc_open, wtosc_close };
const A2_unitdesc a2_panmix_unitdesc = { "panmix", 0, panmix_regs, NULL,
	panmix_consts, 1, 2, 1, 2, BLOCKSIZE, panmix_init, panmix_deinit,
	panmix_open, panmix_close };
const A2_unitdesc a2_filter12_unitdesc = { "filter12", A2_MATCHIO,
	filter12_regs, NULL, NULL, 1, 2, 1, 2, BLOCKSIZE, filter12_init,
	filter12_deinit, filter12_open, filter12_close };
const A2_unitdesc a2_fbdelay_unitdesc = { "fbdelay", 0, fbdelay_regs, NULL,
	NULL, 1, 2, 1, 2, BLOCKSIZE, fbdelay_init, fbdelay_deinit, fbdelay_open,
	fbdelay_close };
const A2_unitdesc a2_inline_unitdesc = { "inline", 0, NULL, NULL, NULL,
	0, 0, 1, A2_MAXCHANNELS, BLOCKSIZE, inl_init, inl_deinit, inl_open,
	inl_close };
const A2_unitdesc a2_xinsert_unitdesc = { "xinsert", A2_MATCHIO | A2_XINSERT,
	NULL, NULL, NULL, 1, A2_MAXCHANNELS, 1, A2_MAXCHANNELS, BLOCKSIZE,
	xins_init, xins_deinit, xins_open, xins_close };

/* fm.c:532,579,631,688,719,752,783,814; fm_Initialize reads the structure off
 * the descriptor's name (fm.c:346-350), so the names are load-bearing */
#define FM_DESC(sym, nm, regs) const A2_unitdesc a2_##sym##_unitdesc = { nm, 0, \
	regs, NULL, NULL, 0, 0, 1, 1, BLOCKSIZE, sym##_init, sym##_deinit, \
	sym##_open, sym##_close };
FM_DESC(fm1, "fm1", fm1_regs) FM_DESC(fm2, "fm2", fm2_regs)
FM_DESC(fm3, "fm3", fm3_regs) FM_DESC(fm4, "fm4", fm4_regs)
FM_DESC(fm3p, "fm3p", fm3_regs) FM_DESC(fm4p, "fm4p", fm4_regs)
FM_DESC(fm2r, "fm2r", fm2_regs) FM_DESC(fm4r, "fm4r", fm4_regs)

/* dc.c:241-281, waveshaper.c:165-193, dcblock.c:162-190, limiter.c:222-250 */
static const A2_crdesc dc_regs[] = {
	{ "value", tr_write0 }, { "mode", tr_write1 }, { NULL, NULL } };
static const A2_constdesc dc_consts[] = {
	{ "STEP", 0 << 16 }, { "LINEAR", 1 << 16 }, { NULL, 0 } };
static const A2_crdesc ws_regs[] = { { "amount", tr_write0 }, { NULL, NULL } };
static const A2_crdesc dcb_regs[] = { { "cutoff", tr_write0 }, { NULL, NULL } };
static const A2_crdesc lim_regs[] = {
	{ "release", tr_write0 }, { "threshold", tr_write1 }, { NULL, NULL } };
const A2_unitdesc a2_dc_unitdesc = { "dc", 0, dc_regs, NULL, dc_consts,
	0, 0, 1, 2, BLOCKSIZE, dc_init, dc_deinit, dc_open, dc_close };
const A2_unitdesc a2_waveshaper_unitdesc = { "waveshaper", A2_MATCHIO, ws_regs,
	NULL, NULL, 1, 2, 1, 2, BLOCKSIZE, wshaper_init, wshaper_deinit,
	wshaper_open, wshaper_close };
const A2_unitdesc a2_dcblock_unitdesc = { "dcblock", A2_MATCHIO, dcb_regs,
	NULL, NULL, 1, 2, 1, 2, BLOCKSIZE, dcblock_init, dcblock_deinit,
	dcblock_open, dcblock_close };
const A2_unitdesc a2_limiter_unitdesc = { "limiter", A2_MATCHIO, lim_regs,
	NULL, NULL, 1, 2, 1, 2, BLOCKSIZE, limiter_init, limiter_deinit,
	limiter_open, limiter_close };

/* ---------------------------------------------------------------------- */

static int do_dump(const char *dir)
{
	char fn[1024];
	FILE *f;
	A2_config *cfg;
	A2_driver *drv;
	A2_interface *i;
	static const char *names[24] = { "pulse1", "pulse2", "pulse3", "pulse4",
		"pulse5", "pulse6", "pulse7", "pulse8", "pulse9", "pulse10",
		"pulse15", "pulse20", "pulse25", "pulse30", "pulse35", "pulse40",
		"pulse45", "pulse50", "saw", "triangle", "sine", "asine", "hsine",
		"qsine" };
	int k;
	uint32_t x = 12345;
	if(!(drv = a2_NewDriver(A2_AUDIODRIVER, "buffer")))
		return 1;
	if(!(cfg = a2_OpenConfig(48000, 64, 2, A2_AUTOCLOSE)))
		return 1;
	a2_AddDriver(cfg, drv);
	if(!(i = a2_Open(cfg)))
		return 1;

	/* a2_P2I probes: int32 pairs {pitch, result} */
	snprintf(fn, sizeof(fn), "%s/p2i_probe.bin", dir);
	if(!(f = fopen(fn, "wb")))
		return 1;
	for(k = 0; k < 65536 + 20000; ++k)
	{
		int32_t pr[2];
		if(k < 65536)
			pr[0] = k;		/* one full octave, every step */
		else
		{
			x = x * 1664525u + 1013904223u;
			pr[0] = (int32_t)(x >> 8) - (1 << 23);	/* +-128 oct */
		}
		pr[1] = (int32_t)a2_P2I(pr[0]);
		fwrite(pr, sizeof(pr), 1, f);
	}
	fclose(f);

	/* built-in waves: per wave {type, flags, period, size[10]} + data */
	snprintf(fn, sizeof(fn), "%s/builtin_waves.bin", dir);
	if(!(f = fopen(fn, "wb")))
		return 1;
	for(k = 0; k < 24; ++k)
	{
		A2_wave *w = a2_GetWave(i, a2_Get(i, A2_ROOTBANK, names[k]));
		int32_t hdr[3];
		int l;
		if(!w)
		{
			fprintf(stderr, "no wave %s\n", names[k]);
			return 1;
		}
		hdr[0] = w->type;
		hdr[1] = w->flags;
		hdr[2] = w->period;
		fwrite(hdr, sizeof(hdr), 1, f);
		fwrite(w->d.wave.size, sizeof(unsigned), A2_MIPLEVELS, f);
		for(l = 0; l < A2_MIPLEVELS; ++l)
			fwrite(w->d.wave.data[l], 2, A2_WAVEPRE +
					w->d.wave.size[l] + A2_WAVEPOST, f);
	}
	fclose(f);

	snprintf(fn, sizeof(fn), "%s/config.txt", dir);
	if((f = fopen(fn, "w")))
	{
		fprintf(f, "samplerate %d\nbasepitch %d\n", cfg->samplerate,
				cfg->basepitch);
		fclose(f);
	}
	a2_Close(i);
	return 0;
}


/*
 * A2REF_UPLOAD=<frames>: upload a wave through the API before the program
 * starts (a2_UploadWave, waves.c:559), hand its handle to the program as one
 * more argument, and release it again after <frames> frames while voices are
 * still playing it - the "wave unloaded under a running oscillator" case of
 * wtosc_check_unloaded (wtosc.c:168-183, waves.c:717-723).  The trace gets a
 * T_WAVEDROP record at that point (between two a2_Run() calls).
 */
static A2_handle upload_test_wave(A2_interface *i)
{
	static int16_t data[3000];
	int k;
	for(k = 0; k < 3000; ++k)
		data[k] = (int16_t)(((k * 7919) % 4001 - 2000) * 6 + (k % 300 - 150) * 40);
	return a2_UploadWave(i, A2_WMIPWAVE, 0, A2_LOOPED, A2_I16, data, sizeof(data));
}

static int do_trace(int argc, const char *argv[])
{
	const char *script = argv[2], *program = argv[3];
	int frames = atoi(argv[4]), buffer = atoi(argv[5]);
	int rate = atoi(argv[6]), channels = atoi(argv[7]);
	const char *tracefn = argv[8], *pcmfn = argv[9];
	int nargs = argc - 10, pargs[A2_MAXARGS], k, done = 0, c;
	A2_config *cfg;
	A2_driver *drv;
	A2_handle bank, prog, vh, upwave = -1;
	A2_wave *upwave_ptr = NULL;
	int release_at = 0;
	FILE *pcm;

	if(nargs > A2_MAXARGS)
		nargs = A2_MAXARGS;
	for(k = 0; k < nargs; ++k)
		pargs[k] = (int)(atof(argv[10 + k]) * 65536.0);

	if(!(tracef = fopen(tracefn, "wb")) || !(pcm = fopen(pcmfn, "wb")))
	{
		fprintf(stderr, "cannot open output files\n");
		return 1;
	}
	if(!(drv = a2_NewDriver(A2_AUDIODRIVER, "buffer")))
		return 1;
	/*
	 * A2REF_REALTIME=1: open the master state with the A2_REALTIME flag, so
	 * that API calls travel through the message FIFOs (interface.c:932-975).
	 * Needed for scripts that render waves at load time: a2_RenderWave ends
	 * with a2_Release() on the master interface, which the direct-call
	 * variant used by offline states does not implement (interface.c:496-505).
	 */
	if(!(cfg = a2_OpenConfig(rate, buffer, channels, A2_AUTOCLOSE |
			(getenv("A2REF_REALTIME") ? A2_REALTIME : 0))))
		return 1;
	a2_AddDriver(cfg, drv);
	if(!(g_iface = a2_Open(cfg)))
	{
		fprintf(stderr, "a2_Open failed: %s\n",
				a2_ErrorString(a2_LastError()));
		return 1;
	}
	rec(T_CONFIG, cfg->samplerate, cfg->basepitch, cfg->channels, buffer,
			(int)noise_state(), 0, 0);
	/*
	 * A script may render waves while it is compiled (a2_RenderWave in an
	 * offline substate, compiler.c:3334-3370): those unit calls belong to
	 * another engine state with its own fragments and noise generator, and
	 * are not part of this state's trace.  The waves they produce are
	 * logged (T_WAVE) when an oscillator first plays them.
	 */
	trace_muted = 1;
	bank = a2_Load(g_iface, script, 0);
	trace_muted = 0;
	if(bank < 0)
	{
		fprintf(stderr, "cannot load %s: %s\n", script,
				a2_ErrorString(-bank));
		return 1;
	}
	if((prog = a2_Get(g_iface, bank, program)) < 0)
	{
		fprintf(stderr, "no program %s\n", program);
		return 1;
	}
	if(getenv("A2REF_UPLOAD"))
	{
		release_at = atoi(getenv("A2REF_UPLOAD"));
		if((upwave = upload_test_wave(g_iface)) < 0 || nargs >= A2_MAXARGS)
			return 1;
		upwave_ptr = a2_GetWave(g_iface, upwave);
		pargs[nargs++] = upwave << 16;
	}
	a2_TimestampReset(g_iface);
	vh = a2_Starta(g_iface, a2_RootVoice(g_iface), prog, nargs, pargs);
	if(vh < 0)
	{
		fprintf(stderr, "start failed: %s\n", a2_ErrorString(-vh));
		return 1;
	}
	while(done < frames)
	{
		int n = frames - done < buffer ? frames - done : buffer;
		int left = n;
		if(upwave >= 0 && done >= release_at)
		{
			int k;
			for(k = 0; k < nwaves; ++k)
				if(wave_ptr[k] == upwave_ptr)
				{
					rec(T_WAVEDROP, k, 0, 0, 0, 0, 0, 0);
					wave_ptr[k] = NULL;	/* the address may be reused */
				}
			if(a2_Release(g_iface, upwave))
			{
				/* offline states cannot release (interface.c:496-505) */
				fprintf(stderr, "a2_Release failed: use A2REF_REALTIME=1\n");
				return 1;
			}
			upwave = -1;
		}
		/* a2_AudioCallback cuts the buffer into fragments of <= 64
		 * frames (core.c:1964-1973); log the same cuts. */
		while(left)
		{
			int fr = left > A2_MAXFRAG ? A2_MAXFRAG : left;
			rec(T_FRAGMENT, fr, 0, 0, 0, 0, 0, 0);
			left -= fr;
			if(n > A2_MAXFRAG)
			{
				fprintf(stderr, "use buffer <= 64 when tracing\n");
				return 1;
			}
		}
		if(a2_Run(g_iface, n) < 0)
		{
			fprintf(stderr, "a2_Run failed\n");
			return 1;
		}
		a2_PumpMessages(g_iface);
		for(c = 0; c < channels; ++c)
			fwrite(((A2_audiodriver *)drv)->buffers[c], 4, n, pcm);
		done += n;
	}
	fclose(pcm);
	fclose(tracef);
	tracef = NULL;
	a2_Close(g_iface);
	return 0;
}

int main(int argc, const char *argv[])
{
	if(argc >= 3 && !strcmp(argv[1], "dump"))
		return do_dump(argv[2]);
	if(argc >= 10 && !strcmp(argv[1], "trace"))
		return do_trace(argc, argv);
	fprintf(stderr, "usage: ref_tools dump <dir> | trace <a2s> <program> "
			"<frames> <buffer> <rate> <channels> <out.trace> "
			"<out.pcm> [args...]\n");
	return 1;
}
