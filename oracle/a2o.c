/*
 * a2o.c - CPU oracle: restatement of the Audiality 2 voice-render path.
 *
 * TEST INFRASTRUCTURE ONLY (see a2o.h).  Written from the behaviour of the
 * reference (olofson/audiality2 v1.9.4); every function cites the reference
 * lines it follows.  Arithmetic notes:
 *   - the reference relies on x86-64 gcc behaviour for signed overflow (wraps),
 *     arithmetic >> of negatives and shift counts taken mod 32; this file is
 *     compiled with -fwrapv and spells the mod-32 shift out.
 *   - int means int32_t, unsigned means uint32_t throughout, as on the
 *     reference's targets.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include "a2o.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define MAXFRAG   A2AMD_MAXFRAG
#define MAXCH     A2AMD_MAXCHANNELS
#define MIPS      A2AMD_MIPLEVELS
#define WAVEPRE   A2AMD_WAVEPRE
#define WAVEPOST  A2AMD_WAVEPOST
#define MAXPHINC  512            /* A2_MAXPHINC, a2_waves.h:57 */
#define WTOSC_MAXLENGTH (0x01000000 - WAVEPRE - WAVEPOST)  /* wtosc.c:55 */
#define FBD_BUFSIZE 131072       /* fbdelay.c:27 */
#define MIDDLEC 261.626f         /* a2_pitch.h:41 */

/* ------------------------------------------------------------------------
 * a2_dsp.h
 * ----------------------------------------------------------------------*/

/* a2_Noise, a2_dsp.h:37-42 */
int a2o_noise(uint32_t *nstate)
{
	*nstate *= 1566083941u;
	(*nstate)++;
	return (int)(*nstate * (*nstate >> 16) >> 16);
}

/* a2_Hermite, a2_dsp.h:64-74.  ph is 24:8; reads d[i-1..i+2]. */
int a2o_hermite(const int16_t *d, unsigned ph)
{
	int i = (int)(ph >> 8);
	int x = (int)(ph & 0xff) << 7;
	int c = (d[i + 1] - d[i - 1]) >> 1;
	int a = (3 * (d[i] - d[i + 1]) + d[i + 2] - d[i - 1]) >> 1;
	int b = d[i - 1] - d[i] + c - a;
	a = a * x >> 15;
	a = (a + b) * x >> 15;
	return d[i] + ((a + c) * x >> 15);
}

/* wtosc_Inter, A2_HIFI variant (src/config.h:108), wtosc.c:28-33 */
static inline int inter(const int16_t *d, unsigned ph, unsigned dph)
{
	return a2o_hermite(d, ph) + a2o_hermite(d, ph + (dph >> 1));
}

typedef struct { int value, target, delta, timer; } ramper;

/* a2_InitRamper, a2_dsp.h:121-125 */
static void ramp_init(ramper *r, int v)
{
	r->value = r->target = v << 8;
	r->delta = r->timer = 0;
}

/* a2_PrepareRamper, a2_dsp.h:128-149 */
static void ramp_prepare(ramper *r, int frames)
{
	if(!r->timer)
	{
		r->value = r->target;
		r->delta = 0;
	}
	else if(frames <= (r->timer >> 8))
	{
		r->delta = (int)(((int64_t)(r->target - r->value) << 8) / r->timer);
		r->timer -= frames << 8;
	}
	else
	{
		r->delta = (r->target - r->value) / frames;
		r->timer = 0;
	}
}

/* a2_RunRamper, a2_dsp.h:152-155 */
static void ramp_run(ramper *r, int frames)
{
	r->value += r->delta * frames;
}

/* a2_SetRamper, a2_dsp.h:161-170 */
static void ramp_set(ramper *r, int target, int start, int duration)
{
	r->target = target << 8;
	r->timer = duration + start;
	if(r->timer < 256)
		r->value = r->target;
	else
		r->value += r->delta * start >> 8;
}

/* ------------------------------------------------------------------------
 * src/pitch.c
 * ----------------------------------------------------------------------*/

/* a2_pitch_open, pitch.c:70-96: 64 segments {base, coeff}. */
void a2o_build_pitch_table(uint32_t *tab)
{
	unsigned i, b = 0x80000000u;
	for(i = 0; i < 64; ++i)
	{
		unsigned b2 = (unsigned)((double)0x80000000u * powf(2.0f,
				(i + 1) * (1.0f / 64)) + 0.5f);
		tab[2 * i] = b;
		tab[2 * i + 1] = (b2 - b + 128) >> 8;
		b = b2;
	}
}

/* a2_P2I, pitch.c:57-67.  The reference's "dph >> (7 - oct)" runs on x86 with
 * the count taken mod 32; wtosc depends on that. */
unsigned a2o_p2i(const uint32_t *tab, int pitch)
{
	int n = pitch & 0xffff;
	int oct = pitch >> 16;
	const uint32_t *pe = tab + 2 * (n >> 10);
	unsigned dph = pe[1] * (unsigned)(n & 0x3ff);
	dph >>= 2;
	dph += pe[0];
	return dph >> ((unsigned)(7 - oct) & 31);
}

/* f12_pitch2coeff, filter12.c:65-72 */
int a2o_f12_coeff(const uint32_t *tab, int cutoff_value, int samplerate)
{
	float f = a2o_p2i(tab, cutoff_value >> 8) * (MIDDLEC / 16777216.0f);
	if(f > (samplerate >> 2))
		return 362 << 16;
	return (int)(512.0f * 65536.0f * sin(M_PI * f / samplerate));
}

/* dcb_pitch2coeff, dcblock.c:57-64: as filter12's, from a 16:16 pitch */
int a2o_dcb_coeff(const uint32_t *tab, int cutoff, int samplerate)
{
	float f = a2o_p2i(tab, cutoff) * (MIDDLEC / 16777216.0f);
	if(f > (samplerate >> 2))
		return 362 << 16;
	return (int)(512.0f * 65536.0f * sin(M_PI * f / samplerate));
}

/* ------------------------------------------------------------------------
 * src/waves.c: padding, mip maps, built-in waves
 * ----------------------------------------------------------------------*/

/* a2_fix_pad, waves.c:90-106; d = first pad sample of the level */
static void fix_pad(int16_t *d, unsigned size, int looped)
{
	if(looped && size)
	{
		int i;
		d[0] = d[size];		/* A2_WAVEPRE == 1 */
		for(i = 0; i < WAVEPOST; ++i)
			d[WAVEPRE + size + i] = d[WAVEPRE + i % size];
	}
	else
	{
		d[0] = 0;
		memset(d + WAVEPRE + size, 0, WAVEPOST * 2);
	}
}

/* a2_wave_alloc + a2_render_mipmaps, waves.c:59-130 */
unsigned a2o_wave_pyramid(const int16_t *src, unsigned length, int looped,
		int miplevels, int16_t *dst, uint32_t *sizes, uint32_t *offsets)
{
	unsigned pos = 0;
	int i;
	for(i = 0; i < miplevels; ++i)
	{
		unsigned size = (length + (1u << i) - 1) >> i;
		sizes[i] = size;
		offsets[i] = pos;
		pos += WAVEPRE + size + WAVEPOST;
	}
	memcpy(dst + offsets[0] + WAVEPRE, src, length * sizeof(int16_t));
	fix_pad(dst + offsets[0], sizes[0], looped);
	for(i = 1; i < miplevels; ++i)
	{
		unsigned s;
		const int16_t *sd = dst + offsets[i - 1] + WAVEPRE;
		int16_t *d = dst + offsets[i] + WAVEPRE;
		for(s = 0; s < sizes[i]; ++s)
			d[s] = (int16_t)((((int)sd[s * 2] << 1) + sd[(int)(s * 2) - 1] +
					sd[s * 2 + 1]) >> 2);
		fix_pad(dst + offsets[i], sizes[i], looped);
	}
	return pos;
}

/* a2_InitWaves, waves.c:629-708.  The reference reuses one stack buffer for
 * all waves, so each wave is built on top of the previous one's contents. */
int a2o_builtin_wave(int idx, int16_t *out, int pulse1_hole)
{
	static const int duty[18] = { 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15, 20,
			25, 30, 35, 40, 45, 50 };
	int16_t buf[2048];
	int j, s, w = 0;
	if(idx < 0 || idx >= 24)
		return -1;
	memset(buf, 0, sizeof(buf));
	buf[20] = (int16_t)pulse1_hole;
	for(j = 0; j < 18; ++j, ++w)
	{
		int s1 = (2048 * duty[j] + 50) / 100;
		for(s = 0; s < s1; ++s)
			buf[s] = 32767;
		for(++s; s < 2048; ++s)
			buf[s] = -32767;
		if(w == idx)
			goto done;
	}
	for(s = 0; s < 2048; ++s)
		buf[s] = (int16_t)(s * 65534 / 2048 - 32767);
	if(w++ == idx)
		goto done;
	for(s = 0; s < 1024; ++s)
	{
		int16_t v = (int16_t)(s * 65534 * 2 / 2048 - 32767);
		buf[s + 512] = v;
		buf[(5 * 2048 / 4 - s - 1) % 2048] = v;
	}
	if(w++ == idx)
		goto done;
	for(s = 0; s < 2048; ++s)
		buf[s] = (int16_t)(sin(s * 2.0f * M_PI / 2048) * 32767.0f);
	if(w++ == idx)
		goto done;
	for(s = 1024; s < 2048; ++s)
		buf[s] = (int16_t)-buf[s];
	if(w++ == idx)
		goto done;
	for(s = 1024; s < 2048; ++s)
		buf[s] = 0;
	if(w++ == idx)
		goto done;
	for(s = 0; s < 512; ++s)
		buf[s + 1024] = buf[s];
done:
	memcpy(out, buf, sizeof(buf));
	return 0;
}

/* ------------------------------------------------------------------------
 * Engine objects
 * ----------------------------------------------------------------------*/

typedef struct a2o_wave
{
	uint64_t	key;
	int		live;
	int		type;
	unsigned	flags, period;
	unsigned	size[MIPS];
	int16_t		*data[MIPS];	/* first pad sample of each level */
} a2o_wave;

enum { OSC_OFF = 0, OSC_NOISE, OSC_WAVE, OSC_MIPWAVE };

struct a2o_voice;

typedef struct a2o_unit
{
	int		live, kind;
	unsigned	flags;
	int		nin, nout, wired;
	struct a2o_voice *voice;
	/* wtosc (A2_wtosc, wtosc.c:66-80) */
	int		mode;
	int		wave;		/* index into ctx->waves or -1 */
	unsigned	dphase;
	uint64_t	phase;
	int		noise;
	int		p_ramping;
	ramper		p, a;
	/* panmix (A2_panmix, panmix.c:35-40) */
	ramper		vol, pan;
	/* filter12 (A2_filter12, filter12.c:36-56) */
	ramper		cutoff, q;
	int		lp, bp, hp, f1, d1[2], d2[2];
	/* fbdelay (A2_fbdelay, fbdelay.c:41-60) */
	int		fbdelay, ldelay, rdelay;
	int		drygain, fbgain, lgain, rgain;
	int32_t		*lbuf, *rbuf;
	int		bufpos;
	/* dc (A2_dc, dc.c:42-47): ramper 'a' is reused as the value ramper */
	int		dcmode;
	/* waveshaper (A2_waveshaper, waveshaper.c:44-48): 'a' = amount */
	/* dcblock (A2_dcblock, dcblock.c:33-49): f1, d1[], d2[] as filter12's */
	/* limiter (A2_limiter, limiter.c:35-42) */
	unsigned	threshold, peak;
	int		release;
	/* xinsert clients (A2_xinsert.clients, xinsert.h:64): the callbacks stay
	 * with the caller; what READ-only clients are handed is kept per fragment,
	 * what WRITE-only clients produced for the coming window waits in inj */
	unsigned	xio_mode;
	int32_t		*tap;		/* [256 fragments][MAXCH][MAXFRAG] */
	int32_t		inj[A2AMD_MAXCHANNELS][A2AMD_MAXFRAG];
	/* fm (A2_fm / A2_fmosc, fm.c:81-105) */
	int		nops;
	struct a2o_fmosc
	{
		ramper		a, fb, p;
		int		last_pitch;
		unsigned	phase, dphase;
		int		last;
	} op[4];
} a2o_unit;

typedef struct a2o_voice
{
	uint64_t	key;
	int		live;
	int		nunits, nlive;
	int		units[A2AMD_MAXCHAIN];
	int		resolved;
	int32_t		*out[MAXCH];	/* the voice's output bus */
	int32_t		scratch[MAXCH][MAXFRAG];
} a2o_voice;

struct a2o_ctx
{
	a2amd_config	cfg;
	uint32_t	ptab[128];
	char		err[256];

	a2o_wave	*waves;
	int		nwaves;
	a2o_unit	*units;
	int		nunits, cap_units, unit_hint;	/* no dead unit slot below unit_hint */
	a2o_voice	**voices;
	int		nvoices, cap_voices, voice_hint;
	a2o_voice	*building;	/* voice being populated */

	/* inline windows currently open (src/core.c:1769: recursion depth) */
	int		stack[256];
	int		sp;

	/* call pattern of the last explicitly walked fragment (for
	 * a2o_fragment_repeat): unit ids, or -1 - unit for an inline_end */
	int		*pattern;
	int		npattern, cap_pattern;
	int		replaying;

	/* fragment clock + master bus (A2_state.master) */
	int		frag_open;
	unsigned	frag_frames;
	int		batch_frags;	/* fragments begun since the last render */
	int32_t		master[MAXCH][MAXFRAG];
	int32_t		*outbuf[MAXCH];
	unsigned	out_frames, out_cap;
};

static int fail(a2o_ctx *c, int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	if(c)
		vsnprintf(c->err, sizeof(c->err), fmt, ap);
	va_end(ap);
	return code;
}

const char *a2o_last_error(const a2o_ctx *c)
{
	return c ? c->err : "no context";
}

int a2o_open(const a2amd_config *cfg, a2o_ctx **out)
{
	a2o_ctx *c;
	if(!cfg || !out || cfg->channels < 1 || cfg->channels > MAXCH)
		return A2AMD_EINVAL;
	c = (a2o_ctx *)calloc(1, sizeof(a2o_ctx));
	if(!c)
		return A2AMD_ENOMEM;
	c->cfg = *cfg;
	a2o_build_pitch_table(c->ptab);
	*out = c;
	return A2AMD_OK;
}

void a2o_close(a2o_ctx *c)
{
	int i, j;
	if(!c)
		return;
	for(i = 0; i < c->nwaves; ++i)
		for(j = 0; j < MIPS; ++j)
			free(c->waves[i].data[j]);
	free(c->waves);
	for(i = 0; i < c->nunits; ++i)
	{
		free(c->units[i].lbuf);
		free(c->units[i].rbuf);
		free(c->units[i].tap);
	}
	free(c->units);
	free(c->pattern);
	for(i = 0; i < c->nvoices; ++i)
		free(c->voices[i]);
	free(c->voices);
	for(i = 0; i < MAXCH; ++i)
		free(c->outbuf[i]);
	free(c);
}

int a2o_set_pitch_table(a2o_ctx *c, const uint32_t *t)
{
	memcpy(c->ptab, t, sizeof(c->ptab));
	return A2AMD_OK;
}

int a2o_get_pitch_table(const a2o_ctx *c, uint32_t *t)
{
	memcpy(t, c->ptab, sizeof(c->ptab));
	return A2AMD_OK;
}

/* ---- waves ---------------------------------------------------------------*/

int a2o_wave_upload(a2o_ctx *c, uint64_t key, const a2amd_wavedesc *w)
{
	int i, id = -1, levels;
	a2o_wave *ww;
	for(i = 0; i < c->nwaves; ++i)
		if(c->waves[i].live && c->waves[i].key == key)
			id = i;
	if(id < 0)
	{
		a2o_wave *nw = (a2o_wave *)realloc(c->waves,
				sizeof(a2o_wave) * (c->nwaves + 1));
		if(!nw)
			return A2AMD_ENOMEM;
		c->waves = nw;
		id = c->nwaves++;
		memset(&c->waves[id], 0, sizeof(a2o_wave));
	}
	ww = &c->waves[id];
	for(i = 0; i < MIPS; ++i)
	{
		free(ww->data[i]);
		ww->data[i] = NULL;
		ww->size[i] = 0;
	}
	ww->key = key;
	ww->live = 1;
	ww->type = w->type;
	ww->flags = w->flags;
	ww->period = w->period;
	levels = w->type == A2AMD_WMIPWAVE ? MIPS :
			w->type == A2AMD_WWAVE ? 1 : 0;
	for(i = 0; i < levels; ++i)
	{
		size_t n = WAVEPRE + (size_t)w->size[i] + WAVEPOST;
		ww->size[i] = w->size[i];
		ww->data[i] = (int16_t *)malloc(n * sizeof(int16_t));
		if(!ww->data[i])
			return A2AMD_ENOMEM;
		memcpy(ww->data[i], w->data[i], n * sizeof(int16_t));
	}
	return id;
}

/* a2_discard_wave, waves.c:717-723: size[0] = 0 marks "unloaded" */
int a2o_wave_drop(a2o_ctx *c, uint64_t key)
{
	int i;
	for(i = 0; i < c->nwaves; ++i)
		if(c->waves[i].live && c->waves[i].key == key)
		{
			c->waves[i].size[0] = 0;
			c->waves[i].live = 0;
			return A2AMD_OK;
		}
	return A2AMD_EINVAL;
}

/* ---- fragment clock --------------------------------------------------------*/

/* a2_ProcessMaster, core.c:1900-1907 (append instead of driver buffers) */
static int harvest(a2o_ctx *c)
{
	int ch;
	if(!c->frag_open)
		return 0;
	if(c->out_frames + c->frag_frames > c->out_cap)
	{
		unsigned ncap = c->out_cap ? c->out_cap * 2 : 4096;
		for(ch = 0; ch < c->cfg.channels; ++ch)
		{
			int32_t *nb = (int32_t *)realloc(c->outbuf[ch],
					ncap * sizeof(int32_t));
			if(!nb)
				return A2AMD_ENOMEM;
			c->outbuf[ch] = nb;
		}
		c->out_cap = ncap;
	}
	for(ch = 0; ch < c->cfg.channels; ++ch)
		memcpy(c->outbuf[ch] + c->out_frames, c->master[ch],
				c->frag_frames * sizeof(int32_t));
	c->out_frames += c->frag_frames;
	c->frag_open = 0;
	return 0;
}

/* one turn of the while(remain) loop in a2_AudioCallback, core.c:1964-1973 */
int a2o_fragment(a2o_ctx *c, unsigned frames)
{
	int r;
	if(!frames || frames > MAXFRAG)
		return fail(c, A2AMD_EINVAL, "fragment of %u frames", frames);
	if(c->sp)
		return fail(c, A2AMD_ESTATE, "fragment inside an inline window");
	if((r = harvest(c)))
		return r;
	memset(c->master, 0, sizeof(c->master));	/* a2_ClearBus */
	c->frag_frames = frames;
	c->frag_open = 1;
	++c->batch_frags;
	c->building = NULL;
	if(!c->replaying)
		c->npattern = 0;
	return A2AMD_OK;
}

static void pattern_add(a2o_ctx *c, int code)
{
	if(c->replaying)
		return;
	if(c->npattern == c->cap_pattern)
	{
		int nc = c->cap_pattern ? c->cap_pattern * 2 : 4096;
		int *np = (int *)realloc(c->pattern, nc * sizeof(int));
		if(!np)
			return;
		c->pattern = np;
		c->cap_pattern = nc;
	}
	c->pattern[c->npattern++] = code;
}

/*
 * 'count' more fragments in which the engine's voice walk finds every VM
 * asleep: the Process calls of the last walked fragment again, each over the
 * full fragment (core.c:1852-1878 with no wake-ups).  No noise oscillators.
 */
int a2o_fragment_repeat(a2o_ctx *c, unsigned frames, unsigned count)
{
	unsigned n;
	int i, r = A2AMD_OK;
	for(i = 0; i < c->nunits; ++i)
		if(c->units[i].live && c->units[i].xio_mode)
			return fail(c, A2AMD_EUNSUPPORTED, "fragment_repeat with clients "
					"on unit %d: their callbacks need every window", i);
	c->replaying = 1;
	for(n = 0; n < count && !r; ++n)
	{
		if((r = a2o_fragment(c, frames)))
			break;
		for(i = 0; i < c->npattern && !r; ++i)
			if(c->pattern[i] >= 0)
				r = a2o_unit_process(c, c->pattern[i], 0, frames, NULL);
			else
				r = a2o_inline_end(c, -1 - c->pattern[i]);
	}
	c->replaying = 0;
	return r;
}

int a2o_render(a2o_ctx *c, unsigned phases, int32_t *const *out, unsigned cap)
{
	int ch, n, r;
	(void)phases;
	if(c->sp)
		return fail(c, A2AMD_ESTATE, "render inside an inline window");
	if((r = harvest(c)))
		return r;
	if(c->out_frames > cap)
		return fail(c, A2AMD_EINVAL, "output capacity %u < %u", cap,
				c->out_frames);
	for(ch = 0; ch < c->cfg.channels; ++ch)
		memcpy(out[ch], c->outbuf[ch], c->out_frames * sizeof(int32_t));
	n = (int)c->out_frames;
	c->out_frames = 0;
	c->batch_frags = 0;
	c->building = NULL;
	return n;
}

/* ---- units: init / deinit ------------------------------------------------*/

static a2o_unit *get_unit(a2o_ctx *c, int id)
{
	if(id < 0 || id >= c->nunits || !c->units[id].live)
		return NULL;
	return &c->units[id];
}

static a2o_wave *osc_wave(a2o_ctx *c, a2o_unit *o)
{
	return o->wave >= 0 ? &c->waves[o->wave] : NULL;
}

/* wtosc_set_phase, wtosc.c:369-378 */
static void wtosc_set_phase(a2o_ctx *c, a2o_unit *o, int ph, unsigned sst)
{
	a2o_wave *w = osc_wave(c, o);
	if(!w)
	{
		o->phase = 0;
		return;
	}
	ph += sst * (o->dphase >> 8) >> 8;
	o->phase = (uint64_t)((int64_t)ph * w->period << 8);
}

/* ---- fm.c: structure tables, sine table, phase -----------------------------*/

/* operators / oversampling bits / structure (0 chain, 1 parallel modulators,
 * 2 ring modulator) per unit kind.  fm.c only includes fm.h -> a2_units.h, never
 * config.h, so A2_HIFI (config.h:108) is NOT defined where fm.c:36-52 picks the
 * oversampling: the unit is built with the "standard" factors 1/2/4/4, and
 * fm3p/fm4p/fm4r use fm3's, fm2r fm2's (fm.c:280-322). */
static const struct { int nops, osbits, parallel; } fm_shape[8] = {
	{ 1, 0, 0 }, { 2, 1, 0 }, { 3, 2, 0 }, { 4, 2, 0 },	/* fm1..fm4 */
	{ 3, 2, 1 }, { 4, 2, 1 },				/* fm3p fm4p */
	{ 2, 1, 2 }, { 4, 2, 2 }				/* fm2r fm4r */
};
#define IS_FM(k)	((k) >= A2AMD_FM1 && (k) <= A2AMD_FM4R)

/* fm_OpenState, fm.c:493-501: one period of sine + 1 pad sample */
static int16_t fm_sine[2049];
static int fm_sine_ready;
static void fm_build_sine(void)
{
	int s;
	if(fm_sine_ready)
		return;
	for(s = 0; s < 2049; ++s)
		fm_sine[s] = (int16_t)(sin(s * 2.0f * M_PI / 2048) * 32767.0f);
	fm_sine_ready = 1;
}

/* fm_set_phase, fm.c:327-335 */
static void fm_set_phase(a2o_unit *fm, int ph, unsigned sst)
{
	int i;
	for(i = 0; i < fm->nops; ++i)
	{
		int ssph = (int)((unsigned)ph + (sst * (fm->op[i].dphase >> 8) >> 8));
		fm->op[i].phase = (unsigned)((int)((unsigned)ssph * 2048u) >> 8);
	}
}

static void f12_set_q(a2o_unit *f, int v, int start, int dur);
static void f12_set_cutoff(a2o_ctx *c, a2o_unit *f, int v, int transpose,
		int start, int dur);

int a2o_unit_init(a2o_ctx *c, uint64_t key, int kind, unsigned flags,
		int nin, int nout, int wired, int transpose, unsigned wakefrac)
{
	a2o_unit *u;
	a2o_voice *v;
	int id, i;
	if(kind < 0 || kind >= A2AMD_NKINDS)
		return fail(c, A2AMD_EINVAL, "unit kind %d", kind);
	if(nin < 0 || nin > MAXCH || nout < 0 || nout > MAXCH)
		return fail(c, A2AMD_EINVAL, "bad channel counts");

	/* voice chain under construction? (a2_PopulateVoice, core.c:350) */
	v = c->building;
	if(!v || v->key != key)
	{
		v = (a2o_voice *)calloc(1, sizeof(a2o_voice));
		if(!v)
			return A2AMD_ENOMEM;
		v->key = key;
		v->live = 1;
		if(c->nvoices == c->cap_voices)
		{
			int nc = c->cap_voices ? c->cap_voices * 2 : 256;
			a2o_voice **nv = (a2o_voice **)realloc(c->voices,
					nc * sizeof(a2o_voice *));
			if(!nv)
				return A2AMD_ENOMEM;
			c->voices = nv;
			c->cap_voices = nc;
		}
		/* reuse a dead voice record if there is one */
		for(i = c->voice_hint; i < c->nvoices; ++i)
			if(!c->voices[i]->live)
				break;
		c->voice_hint = i + 1;
		if(i < c->nvoices)
		{
			free(c->voices[i]);
			c->voices[i] = v;
		}
		else
			c->voices[c->nvoices++] = v;
		c->building = v;
	}
	if(v->nunits >= A2AMD_MAXCHAIN)
		return fail(c, A2AMD_EUNSUPPORTED, "chain longer than %d",
				A2AMD_MAXCHAIN);

	for(id = c->unit_hint; id < c->nunits; ++id)
		if(!c->units[id].live)
			break;
	c->unit_hint = id + 1;
	if(id == c->nunits)
	{
		if(c->nunits == c->cap_units)
		{
			int nc = c->cap_units ? c->cap_units * 2 : 1024;
			a2o_unit *nu = (a2o_unit *)realloc(c->units,
					nc * sizeof(a2o_unit));
			if(!nu)
				return A2AMD_ENOMEM;
			c->units = nu;
			c->cap_units = nc;
		}
		c->units[c->nunits++].tap = NULL;
	}
	u = &c->units[id];
	free(u->tap);		/* (left readable by a2o_unit_deinit) */
	memset(u, 0, sizeof(*u));
	u->live = 1;
	u->kind = kind;
	u->flags = flags;
	u->nin = nin;
	u->nout = nout;
	u->wired = wired;
	u->voice = v;
	v->units[v->nunits++] = id;
	++v->nlive;

	switch(kind)
	{
	  case A2AMD_WTOSC:	/* wtosc_Initialize, wtosc.c:390-423 */
		u->noise = 0;
		u->wave = -1;
		u->mode = OSC_OFF;
		ramp_init(&u->a, 0);
		ramp_init(&u->p, transpose + c->cfg.basepitch);
		u->dphase = a2o_p2i(c->ptab, u->p.value >> 8);
		u->p_ramping = 0;
		wtosc_set_phase(c, u, 0, wakefrac & 0xff);
		break;
	  case A2AMD_PANMIX:	/* panmix_Initialize, panmix.c:252-284 */
		ramp_init(&u->vol, 65536);
		ramp_init(&u->pan, 0);
		if(nin < 1 || nin > 2 || nout < 1 || nout > 2)
			return fail(c, A2AMD_EINVAL, "panmix %d->%d", nin, nout);
		break;
	  case A2AMD_FILTER12:	/* f12_Initialize, filter12.c:180-221 */
		if(nin != nout || nin < 1 || nin > 2)
			return fail(c, A2AMD_EINVAL, "filter12 %d->%d", nin, nout);
		ramp_init(&u->cutoff, 0);
		ramp_init(&u->q, 0);
		f12_set_cutoff(c, u, 0, transpose, 0, 0);
		f12_set_q(u, 0, 0, 0);
		u->lp = 65536 >> 8;
		u->bp = 0;
		u->hp = 0;
		break;
	  case A2AMD_FBDELAY:	/* fbdelay_Initialize, fbdelay.c:170-220 */
		if(nin < 1 || nin > 2 || nout < 1 || nout > 2)
			return fail(c, A2AMD_EINVAL, "fbdelay %d->%d", nin, nout);
		u->lbuf = (int32_t *)calloc(FBD_BUFSIZE, sizeof(int32_t));
		u->rbuf = (int32_t *)calloc(FBD_BUFSIZE, sizeof(int32_t));
		if(!u->lbuf || !u->rbuf)
			return A2AMD_ENOMEM;
		u->bufpos = 0;
		u->fbdelay = (int)((int64_t)(400 << 16) * c->cfg.samplerate / 65536000);
		u->ldelay = (int)((int64_t)(280 << 16) * c->cfg.samplerate / 65536000);
		u->rdelay = (int)((int64_t)(320 << 16) * c->cfg.samplerate / 65536000);
		u->drygain = 65536;
		u->fbgain = 16384;
		u->lgain = 32768;
		u->rgain = 32768;
		break;
	  case A2AMD_INLINE:	/* a2i_Initialize, inline.c:26-39 */
	  case A2AMD_XINSERT:	/* xi_Initialize, xinsert.c:196-212 */
		break;
	  case A2AMD_XSINK:	/* xsink_Initialize, xsink.c:56-71; 1..8 inputs, no outputs (:91-112) */
		if(nin < 1 || nout != 0)
			return fail(c, A2AMD_EINVAL, "xsink %d->%d", nin, nout);
		break;
	  case A2AMD_XSOURCE:	/* xsrc_Initialize, xsource.c:140-156; no inputs (:171-192) */
		if(nin != 0 || nout < 1)
			return fail(c, A2AMD_EINVAL, "xsource %d->%d", nin, nout);
		break;
	  case A2AMD_DC:	/* dc_Initialize, dc.c:160-188 */
		if(nin != 0 || nout < 1 || nout > 2)
			return fail(c, A2AMD_EINVAL, "dc %d->%d", nin, nout);
		ramp_init(&u->a, 0);
		u->dcmode = 1;		/* A2DCRM_LINEAR */
		break;
	  case A2AMD_WAVESHAPER:	/* waveshaper_Initialize, waveshaper.c:131-156 */
		if(nin != nout || nin < 1 || nin > 2)
			return fail(c, A2AMD_EINVAL, "waveshaper %d->%d", nin, nout);
		ramp_init(&u->a, 0);
		break;
	  case A2AMD_DCBLOCK:	/* dcb_Initialize, dcblock.c:119-153: cutoff -5.0 */
		if(nin != nout || nin < 1 || nin > 2)
			return fail(c, A2AMD_EINVAL, "dcblock %d->%d", nin, nout);
		u->f1 = a2o_dcb_coeff(c->ptab, (int)((unsigned)(-5 << 16) + (unsigned)transpose),
				c->cfg.samplerate);
		break;
	  case A2AMD_LIMITER:	/* limiter_Initialize, limiter.c:168-198 */
		if(nin != nout || nin < 1 || nin > 2)
			return fail(c, A2AMD_EINVAL, "limiter %d->%d", nin, nout);
		u->release = ((64 << 16) << 8) / c->cfg.samplerate;
		u->threshold = (unsigned)((1 << 16) << 8);
		u->peak = 32768u << 8;
		break;
	  default:		/* fm_Initialize, fm.c:338-400 */
		if(nin != 0 || nout != 1)
			return fail(c, A2AMD_EINVAL, "fm %d->%d", nin, nout);
		fm_build_sine();
		u->nops = fm_shape[kind - A2AMD_FM1].nops;
		for(i = 0; i < u->nops; ++i)
		{
			ramp_init(&u->op[i].a, 0);
			ramp_init(&u->op[i].fb, 0);
			ramp_init(&u->op[i].p, transpose + c->cfg.basepitch);
			u->op[i].last_pitch = 0;
			u->op[i].last = 0;
		}
		u->op[0].dphase = a2o_p2i(c->ptab, u->op[0].p.value >> 8);
		for(i = 1; i < u->nops; ++i)
			u->op[i].dphase = u->op[0].dphase;
		fm_set_phase(u, 0, wakefrac & 0xff);
		break;
	}
	return id;
}

int a2o_unit_deinit(a2o_ctx *c, int id)
{
	a2o_unit *u = get_unit(c, id);
	a2o_voice *v;
	if(!u)
		return fail(c, A2AMD_EINVAL, "deinit of dead unit %d", id);
	v = u->voice;
	free(u->lbuf);		/* fbdelay_Deinitialize, fbdelay.c:222-228 */
	free(u->rbuf);
	u->lbuf = u->rbuf = NULL;	/* (u->tap stays readable until the slot is reused) */
	u->live = 0;
	if(id < c->unit_hint)
		c->unit_hint = id;
	if(c->building == v)
		c->building = NULL;
	if(--v->nlive == 0)
	{
		v->live = 0;
		c->voice_hint = 0;
	}
	return A2AMD_OK;
}

/* ---- units: control writes ------------------------------------------------*/

/* f12_CutOff, filter12.c:141-147 */
static void f12_set_cutoff(a2o_ctx *c, a2o_unit *f, int v, int transpose,
		int start, int dur)
{
	ramp_set(&f->cutoff, v + transpose, start, dur);
	if(dur < 256)
		f->f1 = a2o_f12_coeff(c->ptab, f->cutoff.value, c->cfg.samplerate);
}

/* f12_Q, filter12.c:149-162 */
static void f12_set_q(a2o_unit *f, int v, int start, int dur)
{
	if(v < 512)
		ramp_set(&f->q, 32768, start, dur);
	else
		ramp_set(&f->q, (65536 << 8) / v, start, dur);
}

int a2o_unit_write(a2o_ctx *c, int id, int reg, int value, unsigned start,
		unsigned dur, int transpose)
{
	a2o_unit *u = get_unit(c, id);
	if(!u)
		return fail(c, A2AMD_EINVAL, "write to dead unit %d", id);
	c->building = NULL;
	start &= 255;		/* a2_VoiceControl, core.c:148 */
	switch(u->kind)
	{
	  case A2AMD_WTOSC:
		switch(reg)
		{
		  case 0:	/* wtosc_Wave, wtosc.c:433-483 */
		  {
			a2o_wave *w = NULL;
			int wt = A2AMD_WOFF;
			if(value >= 0 && value < c->nwaves && c->waves[value].live)
			{
				w = &c->waves[value];
				wt = w->type;
			}
			u->wave = w ? value : -1;
			if((wt == A2AMD_WWAVE || wt == A2AMD_WMIPWAVE) &&
					w->size[0] > WTOSC_MAXLENGTH)
				wt = A2AMD_WOFF;
			switch(wt)
			{
			  default:
			  case A2AMD_WOFF:
				u->wave = -1;
				u->mode = OSC_OFF;
				break;
			  case A2AMD_WNOISE:
				u->mode = OSC_NOISE;
				break;
			  case A2AMD_WWAVE:
				u->mode = OSC_WAVE;
				break;
			  case A2AMD_WMIPWAVE:
				u->mode = OSC_MIPWAVE;
				break;
			}
			break;
		  }
		  case 1:	/* wtosc_Pitch, wtosc.c:486-492 */
			ramp_set(&u->p, value + transpose + c->cfg.basepitch,
					start, dur);
			if(!dur)
				u->p_ramping = 1;
			break;
		  case 2:	/* wtosc_Amplitude, wtosc.c:495-498 */
			ramp_set(&u->a, value, start, dur);
			break;
		  case 3:	/* wtosc_Phase, wtosc.c:501-504 */
			wtosc_set_phase(c, u, value, start);
			break;
		  default:
			return fail(c, A2AMD_EINVAL, "wtosc reg %d", reg);
		}
		break;
	  case A2AMD_PANMIX:	/* panmix_Vol / panmix_Pan, panmix.c:287-295 */
		if(reg == 0)
			ramp_set(&u->vol, value, start, dur);
		else if(reg == 1)
			ramp_set(&u->pan, value, start, dur);
		else
			return fail(c, A2AMD_EINVAL, "panmix reg %d", reg);
		break;
	  case A2AMD_FILTER12:	/* filter12.c:141-177 */
		switch(reg)
		{
		  case 0: f12_set_cutoff(c, u, value, transpose, start, dur); break;
		  case 1: f12_set_q(u, value, start, dur); break;
		  case 2: u->lp = value >> 8; break;
		  case 3: u->bp = value >> 8; break;
		  case 4: u->hp = value >> 8; break;
		  default:
			return fail(c, A2AMD_EINVAL, "filter12 reg %d", reg);
		}
		break;
	  case A2AMD_FBDELAY:	/* fbdelay.c:231-267 */
		switch(reg)
		{
		  case 0: u->fbdelay = (int)((int64_t)value * c->cfg.samplerate / 65536000); break;
		  case 1: u->ldelay = (int)((int64_t)value * c->cfg.samplerate / 65536000); break;
		  case 2: u->rdelay = (int)((int64_t)value * c->cfg.samplerate / 65536000); break;
		  case 3: u->drygain = value; break;
		  case 4: u->fbgain = value; break;
		  case 5: u->lgain = value; break;
		  case 6: u->rgain = value; break;
		  default:
			return fail(c, A2AMD_EINVAL, "fbdelay reg %d", reg);
		}
		break;
	  case A2AMD_DC:
		if(reg == 0)	/* dc_Value, dc.c:191-215 */
		{
			if(u->dcmode == 0)	/* A2DCRM_STEP */
			{
				u->a.target = (int)((unsigned)value << 8);
				u->a.timer = (int)(dur >> 1) - (int)start;
				if(u->a.timer <= 0)
				{
					u->a.value = u->a.target;
					u->a.timer = 0;
				}
			}
			else
				ramp_set(&u->a, value, start, dur);
		}
		else if(reg == 1)	/* dc_Mode, dc.c:218-238: anything but LINEAR is STEP */
			u->dcmode = (value >> 16) == 1 ? 1 : 0;
		else
			return fail(c, A2AMD_EINVAL, "dc reg %d", reg);
		break;
	  case A2AMD_WAVESHAPER:	/* waveshaper_Amount, waveshaper.c:159-162 */
		if(reg != 0)
			return fail(c, A2AMD_EINVAL, "waveshaper reg %d", reg);
		ramp_set(&u->a, value, start, dur);
		break;
	  case A2AMD_DCBLOCK:	/* dcb_CutOff, dcblock.c:112-117 */
		if(reg != 0)
			return fail(c, A2AMD_EINVAL, "dcblock reg %d", reg);
		u->f1 = a2o_dcb_coeff(c->ptab, (int)((unsigned)value + (unsigned)transpose),
				c->cfg.samplerate);
		break;
	  case A2AMD_LIMITER:	/* limiter.c:201-213 */
		if(reg == 0)
			u->release = (int)((unsigned)value << 8) / c->cfg.samplerate;
		else if(reg == 1)
		{
			u->threshold = (unsigned)value << 8;
			if(u->threshold < 256)
				u->threshold = 256;
		}
		else
			return fail(c, A2AMD_EINVAL, "limiter reg %d", reg);
		break;
	  default:
		if(!IS_FM(u->kind))
			return fail(c, A2AMD_EINVAL, "unit kind %d has no registers",
					u->kind);
		/* fm.c:403-483: phase | p a fb | p1 a1 fb1 | ... */
		if(reg < 0 || reg > 3 * u->nops)
			return fail(c, A2AMD_EINVAL, "fm reg %d", reg);
		if(reg == 0)
			fm_set_phase(u, value, start);
		else
		{
			struct a2o_fmosc *o = &u->op[(reg - 1) / 3];
			switch((reg - 1) % 3)
			{
			  case 0:	/* fm_Pitch: only op0 is absolute */
				if(reg == 1)
					value += transpose + c->cfg.basepitch;
				ramp_set(&o->p, value, start, dur);
				break;
			  case 1: ramp_set(&o->a, value, start, dur); break;
			  case 2: ramp_set(&o->fb, value, start, dur); break;
			}
		}
		break;
	}
	return A2AMD_OK;
}

/* ---- units: Process ---------------------------------------------------------*/

/* wtosc_run_pitch, wtosc.c:89-105 */
static void wtosc_run_pitch(a2o_ctx *c, a2o_unit *o, unsigned frames)
{
	unsigned lastv;
	ramp_prepare(&o->p, (int)frames);
	if(o->dphase && (!o->p.timer && !o->p_ramping))
		return;
	lastv = (unsigned)o->p.value;
	ramp_run(&o->p, (int)frames);
	o->p_ramping = o->p.delta;
	o->dphase = a2o_p2i(c->ptab, (int)((lastv + (unsigned)o->p.value) >> 9));
}

/* wtosc_Off / wtosc_OffAdd, wtosc.c:108-126 */
static void wtosc_off(a2o_unit *o, int32_t *out, unsigned offset,
		unsigned frames, int add)
{
	ramp_prepare(&o->p, (int)frames);
	ramp_prepare(&o->a, (int)frames);
	ramp_run(&o->p, (int)frames);
	ramp_run(&o->a, (int)frames);
	if(!add)
		memset(out + offset, 0, frames * sizeof(int32_t));
}

/* wtosc_noise, wtosc.c:129-152 */
static void wtosc_noise(a2o_ctx *c, a2o_unit *o, int32_t *out, unsigned offset,
		unsigned frames, int add, uint32_t *nstate)
{
	unsigned s, end = offset + frames;
	wtosc_run_pitch(c, o, frames);
	ramp_prepare(&o->a, (int)frames);
	if(getenv("A2AMD_DEBUG_NOISE"))		/* (to compare with the product's host-side phase shadow) */
		fprintf(stderr, "NOISE phase %llx dphase %x frames %u\n", (unsigned long long)o->phase, o->dphase, frames);
	for(s = offset; s < end; ++s)
	{
		uint64_t nph = o->phase + o->dphase;
		if((o->dphase >= (1u << 23)) || ((nph ^ o->phase) >> 23))
			o->noise = a2o_noise(nstate) - 32767;
		o->phase = nph;
		if(add)
			out[s] += o->noise * (o->a.value >> 10) >> 6;
		else
			out[s] = o->noise * (o->a.value >> 10) >> 6;
		ramp_run(&o->a, 1);
	}
}

/* wtosc_do_fragment, wtosc.c:200-236 */
static uint64_t wtosc_do_fragment(a2o_unit *o, const int16_t *d, int32_t *out,
		unsigned offset, unsigned frames, uint64_t ph, unsigned dph,
		int add, int looped, unsigned wsize)
{
	unsigned s, end = offset + frames;
	for(s = offset; s < end; ++s)
	{
		int v;
		if(wsize)
		{
			if(looped)
				ph %= (uint64_t)wsize << 24;
			else if((ph >> 24) >= wsize)
			{
				if(!add)
					memset(out + s, 0, (end - s) * sizeof(int32_t));
				break;
			}
		}
		v = inter(d, (unsigned)(ph >> 16), dph >> 16);
		if(add)
			out[s] += (int32_t)((int64_t)v * o->a.value >> 17);
		else
			out[s] = (int32_t)((int64_t)v * o->a.value >> 17);
		ph += dph;
		ramp_run(&o->a, 1);
	}
	return ph;
}

/* wtosc_check_unloaded, wtosc.c:168-183 */
/* wtosc_check_unloaded, wtosc.c:168-183.  DELIBERATE DEVIATION: the reference
 * returns from Process without touching its output buffer in the window in
 * which it notices, so the voice plays whatever the previously rendered voice
 * left in the scratch bus they share (core.c:365-395) for that one window.
 * That depends on the traversal order of unrelated voices; here (and on the
 * GPU) the window is silent, as every following one is (wtosc_Off). */
static int wtosc_check_unloaded(a2o_unit *o, a2o_wave *w, int32_t *out,
		unsigned offset, unsigned frames, int add)
{
	if(w->size[0])
		return 0;
	o->wave = -1;
	o->mode = OSC_OFF;
	if(!add)
		memset(out + offset, 0, frames * sizeof(int32_t));
	return 1;
}

/* wtosc_wavetable, wtosc.c:239-286 */
static void wtosc_wavetable(a2o_ctx *c, a2o_unit *o, int32_t *out,
		unsigned offset, unsigned frames, int add)
{
	unsigned mm, dph;
	uint64_t ph;
	a2o_wave *w = osc_wave(c, o);
	if(wtosc_check_unloaded(o, w, out, offset, frames, add))
		return;
	wtosc_run_pitch(c, o, frames);
	dph = ((o->dphase + 255) >> 8) * w->period;
	ramp_prepare(&o->a, (int)frames);
	for(mm = 0; (dph > (MAXPHINC << 8)) && (mm < MIPS - 1); ++mm)
		dph >>= 1;
	ph = o->phase >> mm;
	dph = (unsigned)((uint64_t)o->dphase * w->period >> mm);

	if(w->flags & A2AMD_LOOPED)
		ph %= (uint64_t)w->size[mm] << 24;
	else if((ph >> 24) > (w->size[mm] + WAVEPRE))
	{
		if(!add)
			memset(out + offset, 0, frames * sizeof(int32_t));
		return;
	}

	if(dph > (MAXPHINC << 16))
	{
		if(!add)
			memset(out + offset, 0, frames * sizeof(int32_t));
		ph += (uint64_t)dph * frames;
		o->phase = ph << mm;
		ramp_run(&o->a, (int)frames);
	}
	else
		o->phase = wtosc_do_fragment(o, w->data[mm] + WAVEPRE, out,
				offset, frames, ph, dph, add, 0, 0) << mm;
}

/* wtosc_wavetable_no_mip, wtosc.c:301-358 */
static void wtosc_wavetable_no_mip(a2o_ctx *c, a2o_unit *o, int32_t *out,
		unsigned offset, unsigned frames, int add)
{
	uint64_t dph;
	a2o_wave *w = osc_wave(c, o);
	const int16_t *d;
	if(wtosc_check_unloaded(o, w, out, offset, frames, add))
		return;
	d = w->data[0] + WAVEPRE;
	wtosc_run_pitch(c, o, frames);
	dph = (uint64_t)o->dphase * w->period;
	ramp_prepare(&o->a, (int)frames);

	if(dph >> 32)
	{
		if(!add)
			memset(out + offset, 0, frames * sizeof(int32_t));
		o->phase += dph * frames;
		ramp_run(&o->a, (int)frames);
	}
	else if(dph > (MAXPHINC << 16))
	{
		o->phase = wtosc_do_fragment(o, d, out, offset, frames,
				o->phase, (unsigned)dph, add,
				(w->flags & A2AMD_LOOPED) ? 1 : 0, w->size[0]);
	}
	else
	{
		if(w->flags & A2AMD_LOOPED)
		{
			/*
			 * wtosc.c:340 computes the modulus as a 32 bit
			 * "size << 24", which is 0 for sizes that are multiples
			 * of 256: the reference then divides by zero (traps).
			 * The oracle leaves the phase alone in that case.
			 */
			unsigned m = w->size[0] << 24;
			if(m)
				o->phase %= m;
		}
		else if((o->phase >> 24) > (w->size[0] + WAVEPRE))
		{
			if(!add)
				memset(out + offset, 0, frames * sizeof(int32_t));
			return;
		}
		o->phase = wtosc_do_fragment(o, d, out, offset, frames,
				o->phase, (unsigned)dph, add, 0, 0);
	}
}

/* panmix_process11, panmix.c:49-65 */
static void panmix11(a2o_unit *pm, const int32_t *in, int32_t *out,
		unsigned offset, unsigned frames, int add)
{
	unsigned s, end = offset + frames;
	ramp_prepare(&pm->vol, (int)frames);
	for(s = offset; s < end; ++s)
	{
		int32_t r = (int32_t)((int64_t)in[s] * pm->vol.value >> 24);
		if(add)
			out[s] += r;
		else
			out[s] = r;
		ramp_run(&pm->vol, 1);
	}
}

static inline void pan_gains(a2o_unit *pm, int clamp, int *v0, int *v1)
{
	int vp = (int)((int64_t)pm->pan.value * pm->vol.value >> 24);
	*v0 = pm->vol.value - vp;
	*v1 = pm->vol.value + vp;
	if(clamp)
	{
		if(*v0 > pm->vol.value << 1)
			*v0 = pm->vol.value << 1;
		if(*v1 > pm->vol.value << 1)
			*v1 = pm->vol.value << 1;
	}
}

/* clamp variant selection, panmix.c:117-135 etc. */
static int pan_needs_clamp(const a2o_unit *pm)
{
	return pm->pan.target > 0xffffff || pm->pan.target < -0xffffff ||
			pm->pan.value > 0xffffff || pm->pan.value < -0xffffff;
}

/* panmix_process12, panmix.c:78-115 */
static void panmix12(a2o_unit *pm, const int32_t *in, int32_t *out0,
		int32_t *out1, unsigned offset, unsigned frames, int add)
{
	unsigned s, end = offset + frames;
	int clamp = pan_needs_clamp(pm);
	ramp_prepare(&pm->vol, (int)frames);
	ramp_prepare(&pm->pan, (int)frames);
	for(s = offset; s < end; ++s)
	{
		int v0, v1, ins = in[s];
		pan_gains(pm, clamp, &v0, &v1);
		if(add)
		{
			out0[s] += (int32_t)((int64_t)ins * v0 >> 24);
			out1[s] += (int32_t)((int64_t)ins * v1 >> 24);
		}
		else
		{
			out0[s] = (int32_t)((int64_t)ins * v0 >> 24);
			out1[s] = (int32_t)((int64_t)ins * v1 >> 24);
		}
		ramp_run(&pm->vol, 1);
		ramp_run(&pm->pan, 1);
	}
}

/* panmix_process21, panmix.c:137-169 */
static void panmix21(a2o_unit *pm, const int32_t *in0, const int32_t *in1,
		int32_t *out, unsigned offset, unsigned frames, int add)
{
	unsigned s, end = offset + frames;
	int clamp = pan_needs_clamp(pm);
	ramp_prepare(&pm->vol, (int)frames);
	ramp_prepare(&pm->pan, (int)frames);
	for(s = offset; s < end; ++s)
	{
		int v0, v1;
		int32_t r;
		pan_gains(pm, clamp, &v0, &v1);
		r = (int32_t)(((int64_t)in0[s] * v0 + (int64_t)in1[s] * v1) >> 25);
		if(add)
			out[s] += r;
		else
			out[s] = r;
		ramp_run(&pm->vol, 1);
		ramp_run(&pm->pan, 1);
	}
}

/* panmix_process22, panmix.c:192-229 */
static void panmix22(a2o_unit *pm, const int32_t *in0, const int32_t *in1,
		int32_t *out0, int32_t *out1, unsigned offset, unsigned frames,
		int add)
{
	unsigned s, end = offset + frames;
	int clamp = pan_needs_clamp(pm);
	ramp_prepare(&pm->vol, (int)frames);
	ramp_prepare(&pm->pan, (int)frames);
	for(s = offset; s < end; ++s)
	{
		int v0, v1, in0s = in0[s], in1s = in1[s];
		pan_gains(pm, clamp, &v0, &v1);
		if(add)
		{
			out0[s] += (int32_t)((int64_t)in0s * v0 >> 24);
			out1[s] += (int32_t)((int64_t)in1s * v1 >> 24);
		}
		else
		{
			out0[s] = (int32_t)((int64_t)in0s * v0 >> 24);
			out1[s] = (int32_t)((int64_t)in1s * v1 >> 24);
		}
		ramp_run(&pm->vol, 1);
		ramp_run(&pm->pan, 1);
	}
}

/* f12_process, filter12.c:74-119 */
static void f12_process(a2o_ctx *c, a2o_unit *f, int32_t **in, int32_t **out,
		unsigned offset, unsigned frames, int add, int channels)
{
	unsigned s, end = offset + frames;
	int ch, df, f0 = f->f1;
	ramp_prepare(&f->q, (int)frames);
	ramp_prepare(&f->cutoff, (int)frames);
	if(f->cutoff.delta)
	{
		ramp_run(&f->cutoff, (int)frames);
		f->f1 = a2o_f12_coeff(c->ptab, f->cutoff.value, c->cfg.samplerate);
		df = (f->f1 - f0 + ((int)frames >> 1)) / (int)frames;
	}
	else
		df = 0;
	for(s = offset; s < end; ++s)
	{
		int ff = f0 >> 12;
		int q = f->q.value >> 12;
		for(ch = 0; ch < channels; ++ch)
		{
			int d1 = f->d1[ch] >> 4;
			int l = f->d2[ch] + (ff * d1 >> 8);
			int h = (in[ch][s] >> 5) - l - (q * d1 >> 8);
			int b = (ff * (h >> 4) >> 8) + f->d1[ch];
			int fout = (l * f->lp + b * f->bp + h * f->hp) >> 3;
			if(add)
				out[ch][s] += fout;
			else
				out[ch][s] = fout;
			f->d1[ch] = b;
			f->d2[ch] = l;
		}
		f0 += df;
		ramp_run(&f->q, 1);
	}
}

/* fbdelay_process, fbdelay.c:69-126 */
#define WI(x) ((unsigned)(fbd->bufpos - (x)) & (FBD_BUFSIZE - 1))
static void fbdelay_process(a2o_unit *fbd, int32_t **in, int32_t **out,
		unsigned offset, unsigned frames, int add, int stereoin,
		int stereoout)
{
	unsigned s, end = offset + frames;
	int32_t *b0 = fbd->lbuf, *b1 = fbd->rbuf;
	const int32_t *in0 = in[0], *in1 = in[stereoin ? 1 : 0];
	int32_t *out0 = out[0], *out1 = stereoout ? out[1] : NULL;
	for(s = offset; s < end; ++s)
	{
		int i0 = in0[s], i1 = in1[s];
		int o0 = (int)((int64_t)b1[WI(fbd->fbdelay)] * fbd->fbgain >> 16);
		int o1 = (int)((int64_t)b0[WI(fbd->fbdelay)] * fbd->fbgain >> 16);
		b0[WI(0)] = i0 + o0;
		b1[WI(0)] = i1 + o1;
		o0 += (int)((int64_t)b0[WI(fbd->ldelay)] * fbd->lgain >> 16);
		o1 += (int)((int64_t)b1[WI(fbd->rdelay)] * fbd->rgain >> 16);
		o0 += (int)((int64_t)i0 * fbd->drygain >> 16);
		o1 += (int)((int64_t)i1 * fbd->drygain >> 16);
		if(add)
		{
			if(stereoout)
			{
				out0[s] += o0;
				out1[s] += o1;
			}
			else
				out0[s] += (o0 + o1) >> 1;
		}
		else
		{
			if(stereoout)
			{
				out0[s] = o0;
				out1[s] = o1;
			}
			else
				out0[s] = (o0 + o1) >> 1;
		}
		++fbd->bufpos;
	}
}
#undef WI

/*
 * Which bus do a voice's wired outputs land in?  A new voice inherits its
 * parent's outputs (core.c:479-480), which an 'inline' unit re-points at its
 * own outputs (inline.c:32-33).  Seen from the call stream that is: the bus of
 * the innermost inline window open when the voice first processes, or the
 * master bus (A2_state.master) when none is.
 */

/* ---- dc.c, waveshaper.c, dcblock.c, limiter.c -------------------------------*/

/* dc_process, dc.c:56-134 */
static void dc_process(a2o_unit *dc, int32_t **out, unsigned offset,
		unsigned frames, int outputs, int add)
{
	ramper *v = &dc->a;
	unsigned s, end = offset + frames;
	int o;
#define DC_PUT(x) for(o = 0; o < outputs; ++o) { if(add) out[o][s] += (x); else out[o][s] = (x); }
	if(dc->dcmode == 0)	/* A2DCRM_STEP */
	{
		s = offset;
		if(v->timer >= 256)
		{
			unsigned e2;
			if((unsigned)(v->timer >> 8) >= frames)
			{
				e2 = end;
				v->timer -= (int)(frames << 8);
			}
			else
			{
				e2 = s + (unsigned)(v->timer >> 8);
				v->timer &= 0xff;
			}
			for( ; s < e2; ++s)
				DC_PUT(v->value)
		}
		if((v->timer < 256) && (s < end))
		{
			int tv = ((v->value >> 4) * v->timer +
					(v->target >> 4) * (256 - v->timer)) >> 4;
			DC_PUT(tv)
			++s;
			v->timer = 0;
			v->value = v->target;
		}
		for( ; s < end; ++s)
			DC_PUT(v->target)
	}
	else			/* A2DCRM_LINEAR */
	{
		ramp_prepare(v, (int)frames);
		for(s = offset; s < end; ++s)
		{
			DC_PUT(v->value)
			ramp_run(v, 1);
		}
	}
#undef DC_PUT
}

/* waveshaper_process, waveshaper.c:57-112 (the fixed point branch) */
static void waveshaper_process(a2o_unit *ws, int32_t **in, int32_t **out,
		unsigned offset, unsigned frames, int add, int channels)
{
	unsigned s, end = offset + frames;
	int c;
	ramp_prepare(&ws->a, (int)frames);
	for(s = offset; s < end; ++s)
	{
		int32_t a = ws->a.value;
		int32_t a3p1 = (int32_t)(((uint32_t)a << 1) + (uint32_t)a + (1u << 24));
		int32_t asqr = (int32_t)((int64_t)(a >> 4) * (a >> 4) >> 24);
		for(c = 0; c < channels; ++c)
		{
			int32_t v = in[c][s];
			int32_t vsqr = (int32_t)((int64_t)v * v >> 22);
			int64_t vout = (int64_t)v * a3p1;
			int64_t sqrsub = (int64_t)a * vsqr;
			if(v >= 0)
				vout -= sqrsub;
			else
				vout += sqrsub;
			vout /= ((int64_t)asqr * vsqr >> 16) + (1 << 24);
			if(add)
				out[c][s] += (int32_t)vout;
			else
				out[c][s] = (int32_t)vout;
		}
		ramp_run(&ws->a, 1);
	}
}

/* dcb_process, dcblock.c:66-95 */
static void dcb_process(a2o_unit *dcb, int32_t **in, int32_t **out,
		unsigned offset, unsigned frames, int add, int channels)
{
	unsigned s, end = offset + frames;
	int c, f = dcb->f1 >> 12;
	for(s = offset; s < end; ++s)
		for(c = 0; c < channels; ++c)
		{
			int d1 = dcb->d1[c] >> 4;
			int l = dcb->d2[c] + (f * d1 >> 8);
			int h = (in[c][s] >> 5) - l - (int)((unsigned)d1 << 4);
			int b = (f * (h >> 4) >> 8) + dcb->d1[c];
			int fout = (int)((unsigned)h << 5);
			if(add)
				out[c][s] += fout;
			else
				out[c][s] = fout;
			dcb->d1[c] = b;
			dcb->d2[c] = l;
		}
}

static int iabs(int x)
{
	return x < 0 ? (int)(0u - (unsigned)x) : x;	/* abs(), INT_MIN stays */
}

/* limiter_process11 / limiter_process22, limiter.c:51-158 */
static void limiter_process(a2o_unit *lim, int32_t **in, int32_t **out,
		unsigned offset, unsigned frames, int add, int channels)
{
	unsigned s, end = offset + frames;
	int c;
	for(s = offset; s < end; ++s)
	{
		int gain;
		unsigned p;
		if(channels == 1)
			p = (unsigned)iabs(in[0][s]);
		else
		{
			int lp = iabs(in[0][s]), rp = iabs(in[1][s]);
			p = (unsigned)(lp > rp ? lp : rp);
			p = p + ((p - (unsigned)iabs(lp - rp)) >> 1);
		}
		if(p > lim->peak)
			lim->peak = p;
		else
		{
			lim->peak -= (unsigned)lim->release;
			if(lim->peak < lim->threshold)
				lim->peak = lim->threshold;
			p = lim->peak;
		}
		gain = (int)((32767LL << 16) / ((p + 511) >> 9));
		for(c = 0; c < channels; ++c)
		{
			int32_t r = (int32_t)((int64_t)in[c][s] * gain >> 16);
			if(add)
				out[c][s] += r;
			else
				out[c][s] = r;
		}
	}
}

/* ---- fm.c ----------------------------------------------------------------*/

/* a2_Lerp, a2_dsp.h:50-55 */
static int lerp16(const int16_t *d, unsigned ph)
{
	int i = (int)(ph >> 8);
	int x = (int)(ph & 0xff);
	return (d[i] * (256 - x) + d[i + 1] * x) >> 8;
}

/* fm_osc, fm.c:111-123 */
static int32_t fm_osc(struct a2o_fmosc *o, int mod)
{
	int fb = (int)((int64_t)o->last * o->fb.value >> 17);
	unsigned ph = (o->phase + (unsigned)mod + (unsigned)fb) >> (24 - 8 - 11);
	o->last = lerp16(fm_sine, ph & ((2048u << 8) - 1));
	return (int32_t)((int64_t)o->last * o->a.value >> 16);
}

/* fm_run_pitch, fm.c:126-141: the ramper only ever runs half a window */
static void fm_run_pitch(a2o_ctx *c, struct a2o_fmosc *o, unsigned frames,
		int detune)
{
	int newpitch;
	ramp_prepare(&o->p, (int)frames);
	ramp_run(&o->p, (int)(frames >> 1));
	newpitch = (int)((unsigned)o->p.value + (unsigned)detune) >> 8;
	if(newpitch != o->last_pitch)
	{
		o->dphase = a2o_p2i(c->ptab, newpitch);
		o->last_pitch = newpitch;
	}
}

/* fm_sample, fm.c:151-165 */
static int fm_sample(a2o_unit *fm, int osbits, int parallel)
{
	int i, v = 0;
	for(i = fm->nops - 1; i >= 0; --i)
	{
		if(i && parallel)
			v = (int)((unsigned)v + (unsigned)fm_osc(&fm->op[i], 0));
		else
			v = fm_osc(&fm->op[i], v);
		fm->op[i].phase += fm->op[i].dphase >> osbits;
	}
	return v;
}

/* fm_sample_rm, fm.c:172-192 */
static int fm_sample_rm(a2o_unit *fm, int osbits)
{
	int i, v[2];
	if(fm->nops == 2)
		for(i = 0; i < 2; ++i)
		{
			v[i] = fm_osc(&fm->op[i], 0);
			fm->op[i].phase += fm->op[i].dphase >> osbits;
		}
	else
		for(i = 0; i < 2; ++i)
		{
			v[i] = fm_osc(&fm->op[i], fm_osc(&fm->op[i + 2], 0));
			fm->op[i].phase += fm->op[i].dphase >> osbits;
			fm->op[i + 2].phase += fm->op[i + 2].dphase >> osbits;
		}
	return (int)((int64_t)v[0] * v[1] >> 23);
}

/* fm_process, fm.c:194-233 */
static void fm_process(a2o_ctx *c, a2o_unit *fm, int32_t *out, unsigned offset,
		unsigned frames, int add)
{
	const int osbits = fm_shape[fm->kind - A2AMD_FM1].osbits;
	const int parallel = fm_shape[fm->kind - A2AMD_FM1].parallel;
	const unsigned oversample = 1u << osbits;
	unsigned s, os, end = offset + frames;
	int i, detune = 0;
	for(i = 0; i < fm->nops; ++i)
	{
		ramp_prepare(&fm->op[i].a, (int)frames);
		ramp_prepare(&fm->op[i].fb, (int)frames);
		fm_run_pitch(c, &fm->op[i], frames, detune);
		detune = fm->op[0].p.value;
	}
	for(s = offset; s < end; ++s)
	{
		unsigned vsum = 0;
		for(os = 0; os < oversample; ++os)
			if(parallel == 2)
				vsum += (unsigned)fm_sample_rm(fm, osbits);
			else
				vsum += (unsigned)fm_sample(fm, osbits, parallel);
		for(i = 0; i < fm->nops; ++i)
		{
			ramp_run(&fm->op[i].a, 1);
			ramp_run(&fm->op[i].fb, 1);
			/* "Fix the rounding error buildup!" */
			fm->op[i].phase += fm->op[i].dphase & (oversample - 1);
		}
		if(add)
			out[s] += (int)vsum >> osbits;
		else
			out[s] = (int)vsum >> osbits;
	}
}

static void resolve_out(a2o_ctx *c, a2o_voice *v)
{
	int ch;
	if(v->resolved)
		return;
	if(!c->sp)
		for(ch = 0; ch < MAXCH; ++ch)
			v->out[ch] = c->master[ch];
	else
	{
		a2o_unit *il = &c->units[c->stack[c->sp - 1]];
		a2o_voice *pv = il->voice;
		for(ch = 0; ch < MAXCH; ++ch)
			v->out[ch] = il->wired ? pv->out[ch] : pv->scratch[ch];
	}
	v->resolved = 1;
}

/*
 * xi_process, xinsert.c:60-142, for READ-only and WRITE-only clients.  The
 * client callbacks themselves are the application's: READ-only ones are
 * handed the unit's inputs (:97-101) - kept here per fragment for
 * a2o_unit_tapped(); the buffers the WRITE-only ones filled (:113-118) arrived
 * through a2o_unit_inject() as their sum.  There are no insert clients
 * (:104-111), so the input is bypassed into the output as well (:121-124).
 */
static int xi_process(a2o_ctx *c, a2o_unit *u, int32_t **in, int32_t **out,
		unsigned o, unsigned f, int add)
{
	int32_t obufs[MAXCH][MAXFRAG];
	int32_t *obufp[MAXCH];
	unsigned s;
	int i;
	if(c->batch_frags > 256)
		return fail(c, A2AMD_ESTATE, "more than 256 fragments in a batch "
				"with xinsert clients");
	for(i = 0; i < u->nin; ++i)
	{
		if(add || in[i] != out[i])
			obufp[i] = out[i];
		else
			obufp[i] = obufs[i];
		if(!add)
			memset(obufp[i], 0, sizeof(int32_t) * MAXFRAG);
	}
	if(u->xio_mode & A2AMD_XIO_TAP)
		for(i = 0; i < u->nin; ++i)
			memcpy(u->tap + ((size_t)(c->batch_frags - 1) * MAXCH + i) * MAXFRAG + o,
					in[i] + o, f * sizeof(int32_t));
	if(u->xio_mode & A2AMD_XIO_INJECT)
		for(i = 0; i < u->nin; ++i)
			for(s = o; s < o + f; ++s)
			{
				obufp[i][s] += u->inj[i][s];
				u->inj[i][s] = 0;
			}
	for(i = 0; i < u->nin; ++i)
		for(s = o; s < o + f; ++s)
			obufp[i][s] += in[i][s];
	if(!add)
		for(i = 0; i < u->nin; ++i)
			if(obufp[i] != out[i])
				for(s = o; s < o + f; ++s)
					out[i][s] = obufp[i][s];
	return A2AMD_OK;
}

/* a2_XinsertAddClient / a2_XinsertRemoveClient, xinsertapi.c:72-111, :114-157,
 * as far as the unit is concerned: which kinds of client it has. */
int a2o_unit_clients(a2o_ctx *c, int id, unsigned mode)
{
	a2o_unit *u = get_unit(c, id);
	if(!u || (u->kind != A2AMD_XINSERT && u->kind != A2AMD_XSINK && u->kind != A2AMD_XSOURCE))
		return fail(c, A2AMD_EINVAL, "unit %d is not a live xinsert / xsink / xsource", id);
	if((mode & ~(unsigned)(A2AMD_XIO_TAP | A2AMD_XIO_INJECT)) ||
			(u->kind == A2AMD_XSINK && (mode & A2AMD_XIO_INJECT)) ||
			(u->kind == A2AMD_XSOURCE && (mode & A2AMD_XIO_TAP)))
		return fail(c, A2AMD_EINVAL, "client mode %#x on unit kind %d", mode, u->kind);
	if(mode && !u->tap && !(u->tap = (int32_t *)calloc((size_t)256 * MAXCH * MAXFRAG, sizeof(int32_t))))
		return A2AMD_ENOMEM;
	if(!(mode & A2AMD_XIO_INJECT))
		memset(u->inj, 0, sizeof(u->inj));
	u->xio_mode = mode;
	c->building = NULL;
	return A2AMD_OK;
}

int a2o_unit_inject(a2o_ctx *c, int id, unsigned offset, unsigned frames,
		const int32_t *const *bufs)
{
	a2o_unit *u = get_unit(c, id);
	unsigned s;
	int i;
	if(!u || !(u->xio_mode & A2AMD_XIO_INJECT))
		return fail(c, A2AMD_EINVAL, "unit %d takes no client output", id);
	if(!c->frag_open || !frames || offset + frames > c->frag_frames)
		return fail(c, A2AMD_ESTATE, "inject [%u,+%u) outside fragment", offset, frames);
	for(i = 0; i < u->nout; ++i)
		for(s = 0; s < frames; ++s)
			u->inj[i][offset + s] += bufs[i][s];
	return A2AMD_OK;
}

int a2o_unit_tapped(a2o_ctx *c, int id, unsigned fragment, const int32_t **bufs)
{
	/* (also for a unit that was deinitialised in the course of that batch) */
	a2o_unit *u = id >= 0 && id < c->nunits ? &c->units[id] : NULL;
	int i;
	if(!u || !u->tap)
		return fail(c, A2AMD_EINVAL, "unit %d has had no clients", id);
	if(fragment >= 256)
		return fail(c, A2AMD_EINVAL, "fragment %u", fragment);
	for(i = 0; i < u->nin; ++i)
		bufs[i] = u->tap + ((size_t)fragment * MAXCH + i) * MAXFRAG;
	return u->nin;
}

int a2o_unit_process(a2o_ctx *c, int id, unsigned offset, unsigned frames,
		uint32_t *nstate)
{
	a2o_unit *u = get_unit(c, id);
	a2o_voice *v;
	int32_t *in[MAXCH], *out[MAXCH];
	int ch, add;
	if(!u)
		return fail(c, A2AMD_EINVAL, "process of dead unit %d", id);
	if(!c->frag_open || !frames || offset + frames > c->frag_frames)
		return fail(c, A2AMD_ESTATE, "process [%u,+%u) outside fragment",
				offset, frames);
	c->building = NULL;
	if(!offset)
		pattern_add(c, id);
	v = u->voice;
	resolve_out(c, v);
	for(ch = 0; ch < MAXCH; ++ch)
	{
		in[ch] = v->scratch[ch];
		out[ch] = u->wired ? v->out[ch] : v->scratch[ch];
	}
	add = (u->flags & A2AMD_PROCADD) != 0;

	switch(u->kind)
	{
	  case A2AMD_WTOSC:
		switch(u->mode)
		{
		  case OSC_OFF:
			wtosc_off(u, out[0], offset, frames, add);
			break;
		  case OSC_NOISE:
		  {
			uint32_t dummy = 0;
			if(!nstate)
				return fail(c, A2AMD_EINVAL, "noise oscillator "
						"needs a noise state");
			wtosc_noise(c, u, out[0], offset, frames, add,
					nstate ? nstate : &dummy);
			break;
		  }
		  case OSC_WAVE:
			wtosc_wavetable_no_mip(c, u, out[0], offset, frames, add);
			break;
		  case OSC_MIPWAVE:
			wtosc_wavetable(c, u, out[0], offset, frames, add);
			break;
		}
		break;
	  case A2AMD_PANMIX:
		switch(((u->nin - 1) << 1) + (u->nout - 1))
		{
		  case 0: panmix11(u, in[0], out[0], offset, frames, add); break;
		  case 1: panmix12(u, in[0], out[0], out[1], offset, frames, add); break;
		  case 2: panmix21(u, in[0], in[1], out[0], offset, frames, add); break;
		  case 3: panmix22(u, in[0], in[1], out[0], out[1], offset, frames, add); break;
		}
		break;
	  case A2AMD_FILTER12:
		f12_process(c, u, in, out, offset, frames, add, u->nin);
		break;
	  case A2AMD_FBDELAY:
		fbdelay_process(u, in, out, offset, frames, add, u->nin == 2,
				u->nout == 2);
		break;
	  case A2AMD_INLINE:	/* a2_inline_Process[Add], core.c:1763-1776 */
		if(c->sp >= 256)
			return fail(c, A2AMD_EUNSUPPORTED, "inline nesting");
		if(!add)
			for(ch = 0; ch < u->nout; ++ch)
				memset(out[ch] + offset, 0, frames * sizeof(int32_t));
		c->stack[c->sp++] = id;
		break;
	  case A2AMD_XINSERT:
		if(u->xio_mode)
		{
			int r = xi_process(c, u, in, out, offset, frames, add);
			if(r)
				return r;
			break;
		}
		/* xi_ProcessBypass[Add], xinsert.c:145-161 */
		for(ch = 0; ch < u->nin; ++ch)
		{
			unsigned s;
			if(add)
				for(s = offset; s < offset + frames; ++s)
					out[ch][s] += in[ch][s];
			else if(in[ch] != out[ch])
				for(s = offset; s < offset + frames; ++s)
					out[ch][s] = in[ch][s];
		}
		break;
	  case A2AMD_XSINK:	/* xsink_Process, xsink.c:27-46: every client gets the inputs */
		if(u->xio_mode & A2AMD_XIO_TAP)
		{
			if(c->batch_frags > 256)
				return fail(c, A2AMD_ESTATE, "more than 256 fragments in a "
						"batch with xsink clients");
			for(ch = 0; ch < u->nin; ++ch)
				memcpy(u->tap + ((size_t)(c->batch_frags - 1) * MAXCH + ch) * MAXFRAG + offset,
						in[ch] + offset, frames * sizeof(int32_t));
		}
		break;
	  case A2AMD_XSOURCE:
	  {
		/* xsrc_ProcessAdd, xsource.c:84-87 via :43-78: out += every client's
		 * buffers.  Replacing: one client writes the output itself
		 * (xsrc_ProcessSingle, :90-103); none - silence (xsrc_ProcessNil,
		 * :105-112).  (With several clients in replacing mode the reference
		 * adds an uninitialised array per client, :56-59 with :74-75; here
		 * they add up as in adding mode.) */
		unsigned s;
		for(ch = 0; ch < u->nout; ++ch)
			for(s = offset; s < offset + frames; ++s)
			{
				int32_t x = (u->xio_mode & A2AMD_XIO_INJECT) ? u->inj[ch][s] : 0;
				u->inj[ch][s] = 0;
				out[ch][s] = add ? (int32_t)((uint32_t)out[ch][s] + (uint32_t)x) : x;
			}
		break;
	  }
	  case A2AMD_DC:
		dc_process(u, out, offset, frames, u->nout, add);
		break;
	  case A2AMD_WAVESHAPER:
		waveshaper_process(u, in, out, offset, frames, add, u->nin);
		break;
	  case A2AMD_DCBLOCK:
		dcb_process(u, in, out, offset, frames, add, u->nin);
		break;
	  case A2AMD_LIMITER:
		limiter_process(u, in, out, offset, frames, add, u->nin);
		break;
	  default:		/* fm*_Process[Add], fm.c:235-322 */
		fm_process(c, u, out[0], offset, frames, add);
		break;
	}
	return A2AMD_OK;
}

int a2o_inline_end(a2o_ctx *c, int id)
{
	if(!c->sp || c->stack[c->sp - 1] != id)
		return fail(c, A2AMD_ESTATE, "inline_end(%d) does not match", id);
	--c->sp;
	pattern_add(c, -1 - id);
	return A2AMD_OK;
}
