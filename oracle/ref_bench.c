/*
 * ref_bench.c - times the COMPILED REFERENCE (oracle/_ref/libaudiality2.so) on
 * a sustained-voice workload, one engine state per thread.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY: this is the "cpu_baseline" leg of
 * bench.py (kind "reference").  No wrappers, no tracing: the reference's own
 * units render.  The reference is single-threaded per engine state; T threads
 * means T independent master states with voices/T voices each, the only
 * thread-safe arrangement (include/audiality2.h.cmake:163-166).
 *
 * usage: ref_bench <script.a2s> <program> <voices> <fragments> <threads>
 * prints: one JSON line {voice_samples_per_s, seconds, voices, fragments, threads, active_voices,
 *         run_us_p50 / p99 / max (per a2_Run() call of thread 0), buffer_frames}
 * env:    A2REF_BUFFER=<frames> per a2_Run() call (default 64)
 *         A2REF_HASH=1   FNV-1a 64 of everything rendered while timed, per state ("hashes")
 *         A2REF_HASH=<n> (n > 1) ... of the first n 64-frame fragments' worth of frames only, so
 *                        that a short CPU run and a long drop-in run can be compared
 *         A2REF_HASH_AT=<k>  a second hash ("tail_hashes") of the n fragments from fragment k of the
 *                        timed run on: the far end of what both runs render
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "audiality2.h"
#include "a2_drivers.h"
#include "a2_properties.h"

typedef struct JOB
{
	const char	*script, *program;
	int		voices, fragments;
	double		seconds;
	int		active;
	int		ok;
	unsigned long long hash;	/* FNV-1a 64 of everything the state rendered while timed */
	unsigned long long hash2;	/* ... of the window A2REF_HASH_AT names */
	double		*run_s;		/* seconds per a2_Run() call */
	int		nruns, buffer;
} JOB;

static int cmp_double(const void *a, const void *b)
{
	double x = *(const double *)a, y = *(const double *)b;
	return (x > y) - (x < y);
}

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static void *run(void *arg)
{
	JOB *j = (JOB *)arg;
	A2_driver *drv;
	A2_config *cfg;
	A2_interface *i;
	A2_handle bank, prog;
	int args[2], f, av = 0;
	/* A2REF_BUFFER=<frames>: the a2_Run() buffer (default 64 = one fragment per call;
	 * a2play's default is 4096); 'fragments' stays a count of 64 frame units */
	int buffer = getenv("A2REF_BUFFER") ? atoi(getenv("A2REF_BUFFER")) : 64;
	int nbuf;
	long hash_frames = 0, hashed = 0, at_frames = -1, pos = 0;
	double t0, t1;
	if(!(drv = a2_NewDriver(A2_AUDIODRIVER, "buffer")))
		return NULL;
	if(buffer < 64)
		buffer = 64;
	if(!(cfg = a2_OpenConfig(48000, buffer, 2, A2_AUTOCLOSE)))
		return NULL;
	a2_AddDriver(cfg, drv);
	if(!(i = a2_Open(cfg)))
		return NULL;
	if((bank = a2_Load(i, j->script, 0)) < 0)
		return NULL;
	if((prog = a2_Get(i, bank, j->program)) < 0)
		return NULL;
	args[0] = (j->voices / 4) << 16;
	if(strstr(j->program, "Tree"))		/* FilterTree(G): top-level groups of 32 768 voices */
		args[0] = (j->voices / 32768) << 16;
	args[1] = (int)(65536.0 * 4.0 / (j->voices > 4 ? j->voices : 4));
	a2_TimestampReset(i);
	if(a2_Starta(i, a2_RootVoice(i), prog, 2, args) < 0)
		return NULL;
	/* warm-up: instantiates the voices */
	/* (grouped programs populate one group of 256 voices per 2.7 ms = 2.1 fragments) */
	{
		int wf = j->voices / 4 * 2 / 64 + 16;
		if(strstr(j->program, "Groups") && wf < j->voices / 256 * 3 + 64)
			wf = j->voices / 256 * 3 + 64;
		if(strstr(j->program, "Tree"))
			wf = 320;		/* (128 sub-groups x 2.7 ms, all top groups at once) */
		for(f = 0; f < (wf + buffer / 64 - 1) / (buffer / 64); ++f)
			a2_Run(i, buffer);
	}
	j->hash = 0xcbf29ce484222325ULL;
	nbuf = (j->fragments * 64 + buffer - 1) / buffer;
	j->fragments = nbuf * (buffer / 64);
	j->run_s = calloc(nbuf, sizeof(double));
	j->nruns = nbuf;
	j->buffer = buffer;
	if(getenv("A2REF_HASH"))
	{
		long n = atol(getenv("A2REF_HASH"));
		hash_frames = n > 1 ? n * 64 : (long)nbuf * buffer;
		if(getenv("A2REF_HASH_AT"))
			at_frames = atol(getenv("A2REF_HASH_AT")) * 64;
	}
	j->hash2 = 0xcbf29ce484222325ULL;
	t0 = t1 = now();
	for(f = 0; f < nbuf; ++f)
	{
		double t2;
		a2_Run(i, buffer);
		if(hashed < hash_frames)
		{
			/* per 64-frame fragment, channel 0 then channel 1: the same value whatever the buffer */
			int c, o;
			unsigned k;
			for(o = 0; o < buffer && hashed < hash_frames; o += 64, hashed += 64)
				for(c = 0; c < 2; ++c)
				{
					const unsigned char *b = (const unsigned char *)
							(((A2_audiodriver *)drv)->buffers[c] + o);
					for(k = 0; k < 64 * 4; ++k)
						j->hash = (j->hash ^ b[k]) * 0x100000001b3ULL;
				}
		}
		if(at_frames >= 0 && pos + buffer > at_frames && pos < at_frames + hash_frames)
		{
			int c, o;
			unsigned k;
			for(o = 0; o < buffer; o += 64)
				if(pos + o >= at_frames && pos + o < at_frames + hash_frames)
					for(c = 0; c < 2; ++c)
					{
						const unsigned char *b = (const unsigned char *)
								(((A2_audiodriver *)drv)->buffers[c] + o);
						for(k = 0; k < 64 * 4; ++k)
							j->hash2 = (j->hash2 ^ b[k]) * 0x100000001b3ULL;
					}
		}
		pos += buffer;
		t2 = now();
		j->run_s[f] = t2 - t1;
		t1 = t2;
	}
	j->seconds = now() - t0;
	a2_GetStateProperty(i, A2_PACTIVEVOICES, &av);
	j->active = av;
	j->ok = 1;
	a2_Close(i);
	return NULL;
}

int main(int argc, const char *argv[])
{
	int voices, fragments, threads, t, active = 0;
	double worst = 0;
	JOB *jobs;
	pthread_t *th;
	if(argc < 6)
	{
		fprintf(stderr, "usage: ref_bench <a2s> <program> <voices> "
				"<fragments> <threads>\n");
		return 1;
	}
	voices = atoi(argv[3]);
	fragments = atoi(argv[4]);
	threads = atoi(argv[5]);
	if(threads < 1)
		threads = 1;
	jobs = calloc(threads, sizeof(JOB));
	th = calloc(threads, sizeof(pthread_t));
	for(t = 0; t < threads; ++t)
	{
		jobs[t].script = argv[1];
		jobs[t].program = argv[2];
		jobs[t].voices = voices / threads;
		jobs[t].fragments = fragments;
		pthread_create(&th[t], NULL, run, &jobs[t]);
	}
	for(t = 0; t < threads; ++t)
	{
		pthread_join(th[t], NULL);
		if(!jobs[t].ok)
		{
			fprintf(stderr, "ref_bench: thread %d failed\n", t);
			return 1;
		}
		if(jobs[t].seconds > worst)
			worst = jobs[t].seconds;
		active += jobs[t].active;
	}
	printf("{\"voice_samples_per_s\": %.6g, \"seconds\": %.6f, \"voices\": %d, "
			"\"fragments\": %d, \"threads\": %d, \"active_voices\": %d",
			(double)(voices / threads) * threads * 64.0 * jobs[0].fragments / worst,
			worst, (voices / threads) * threads, jobs[0].fragments, threads, active);
	{
		/* the first quarter of the timed calls (a drop-in that hands voices over to the device does it there) and
		 * the second half (steady state) apart, before the whole is sorted */
		int n = jobs[0].nruns, q = n / 4, h = n / 2;
		double *tmp = malloc((n > 0 ? n : 1) * sizeof(double));
		if(tmp && q > 0 && n - h > 0)
		{
			memcpy(tmp, jobs[0].run_s, q * sizeof(double));
			qsort(tmp, q, sizeof(double), cmp_double);
			printf(", \"run_us_p99_startup\": %.1f", tmp[(int)((q - 1) * 0.99)] * 1e6);
			memcpy(tmp, jobs[0].run_s + h, (n - h) * sizeof(double));
			qsort(tmp, n - h, sizeof(double), cmp_double);
			printf(", \"run_us_p50_steady\": %.1f, \"run_us_p99_steady\": %.1f", tmp[(n - h) / 2] * 1e6,
					tmp[(int)((n - h - 1) * 0.99)] * 1e6);
		}
		free(tmp);
	}
	qsort(jobs[0].run_s, jobs[0].nruns, sizeof(double), cmp_double);
	printf(", \"buffer_frames\": %d, \"run_us_p50\": %.1f, \"run_us_p99\": %.1f, \"run_us_max\": %.1f",
			jobs[0].buffer, jobs[0].run_s[jobs[0].nruns / 2] * 1e6,
			jobs[0].run_s[(int)((jobs[0].nruns - 1) * 0.99)] * 1e6,
			jobs[0].run_s[jobs[0].nruns - 1] * 1e6);
	if(getenv("A2REF_HASH"))
	{
		printf(", \"hashes\": [");
		for(t = 0; t < threads; ++t)
			printf("%s\"%016llx\"", t ? ", " : "", jobs[t].hash);
		printf("]");
		if(getenv("A2REF_HASH_AT"))
		{
			printf(", \"tail_hashes\": [");
			for(t = 0; t < threads; ++t)
				printf("%s\"%016llx\"", t ? ", " : "", jobs[t].hash2);
			printf("]");
		}
	}
	printf("}\n");
	return 0;
}
