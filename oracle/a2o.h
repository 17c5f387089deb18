/*
 * a2o.h - CPU oracle for the Audiality 2 voice-render path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing outside tests/, bench.py's cpu_baseline
 * leg and __graft_entry__.smoke() may include, link or call this.  It is a
 * plain-C restatement of the reference's unit DSP (src/units/{wtosc,panmix,
 * filter12,fbdelay}.c, include/a2_dsp.h, src/pitch.c, the bus semantics of
 * src/core.c:1749-1896 and the wave preparation of src/waves.c) that executes
 * immediately, call by call, like the reference's own Process() callbacks do.
 *
 * It exposes the same call protocol as include/a2amd.h (prefix a2o_ instead
 * of a2amd_), so one recorded call trace can be fed to both.
 *
 * Pinning: checked against outputs of the compiled reference itself
 * (oracle/_ref, built by oracle/Makefile) -- see oracle/README.md and
 * tests/golden/.  The reference ships no golden vectors of its own.
 */
#ifndef A2O_H
#define A2O_H

#include <stdint.h>
#include "../include/a2amd.h"   /* shared enums / structs of the call protocol */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct a2o_ctx a2o_ctx;

int  a2o_open(const a2amd_config *cfg, a2o_ctx **out);
void a2o_close(a2o_ctx *ctx);
const char *a2o_last_error(const a2o_ctx *ctx);
int  a2o_set_pitch_table(a2o_ctx *ctx, const uint32_t *tab128);
int  a2o_get_pitch_table(const a2o_ctx *ctx, uint32_t *tab128);
int  a2o_wave_upload(a2o_ctx *ctx, uint64_t key, const a2amd_wavedesc *w);
int  a2o_wave_drop(a2o_ctx *ctx, uint64_t key);
int  a2o_fragment(a2o_ctx *ctx, unsigned frames);
int  a2o_fragment_repeat(a2o_ctx *ctx, unsigned frames, unsigned count);
int  a2o_unit_init(a2o_ctx *ctx, uint64_t voice_key, int kind, unsigned flags,
		int ninputs, int noutputs, int wired_out, int transpose,
		unsigned wakefrac);
int  a2o_unit_deinit(a2o_ctx *ctx, int unit);
int  a2o_unit_write(a2o_ctx *ctx, int unit, int reg, int value,
		unsigned start, unsigned duration, int transpose);
int  a2o_unit_process(a2o_ctx *ctx, int unit, unsigned offset,
		unsigned frames, uint32_t *noisestate);
int  a2o_inline_end(a2o_ctx *ctx, int unit);
/* clients of xinsert / xsink / xsource units (xinsert.c:60-142, xsink.c:27-46,
 * xsource.c:43-137): as a2amd_unit_clients / _inject / _tapped */
int  a2o_unit_clients(a2o_ctx *ctx, int unit, unsigned mode);
int  a2o_unit_inject(a2o_ctx *ctx, int unit, unsigned offset, unsigned frames,
		const int32_t *const *bufs);
int  a2o_unit_tapped(a2o_ctx *ctx, int unit, unsigned fragment, const int32_t **bufs);
int  a2o_render(a2o_ctx *ctx, unsigned phases, int32_t *const *out,
		unsigned out_capacity_frames);

/* --- stand-alone pieces, exported for unit tests against reference dumps --- */
void     a2o_build_pitch_table(uint32_t *tab128);          /* pitch.c:70-96   */
unsigned a2o_p2i(const uint32_t *tab128, int pitch);       /* pitch.c:57-67   */
int      a2o_noise(uint32_t *nstate);                      /* a2_dsp.h:37-42  */
int      a2o_hermite(const int16_t *d, unsigned ph);       /* a2_dsp.h:64-74  */
int      a2o_f12_coeff(const uint32_t *tab128, int cutoff_value_8_24,
		int samplerate);                           /* filter12.c:65-72 */
/* Wave preparation (waves.c:59-130): given 'length' mip-0 samples, fill a
 * contiguous pyramid; returns total int16 written.  sizes[] receives payload
 * sizes, offsets[] the index of each level's first pad sample. */
unsigned a2o_wave_pyramid(const int16_t *src, unsigned length, int looped,
		int miplevels, int16_t *dst, uint32_t *sizes, uint32_t *offsets);
/* Built-in wave #idx of a2_InitWaves (waves.c:629-708) in table order
 * pulse1..9,10,15..50, saw, triangle, sine, asine, hsine, qsine (24 waves),
 * 2048 samples.  The one sample the reference leaves uninitialised in pulse1
 * (buf[20], waves.c:640-647) is set to 'pulse1_hole'. */
int      a2o_builtin_wave(int idx, int16_t *buf2048, int pulse1_hole);

#ifdef __cplusplus
}
#endif
#endif
