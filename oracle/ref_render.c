/*
 * ref_render.c - renders an A2S program offline through the COMPILED REFERENCE
 * engine (oracle/_ref/libaudiality2.so) and writes the driver buffers.
 *
 * TEST INFRASTRUCTURE.  No wrappers, nothing interposed by this program: run
 * as is, the reference's own units render (= what ref_tools captured in the
 * golden fixtures); run with LD_PRELOAD=audiality2_amd/liba2amd_units.so the
 * same engine renders through the GPU drop-in units.  tests/test_dropin.py
 * compares the two.
 *
 * usage: ref_render <file.a2s> <program> <frames> <buffer> <rate> <channels> <out.pcm> [args...]
 * output: per a2_Run() call, <channels> planar int32 blocks of <buffer> frames
 */
#include <stdio.h>
#include <stdlib.h>
#include "audiality2.h"
#include "a2_drivers.h"
#include "a2_vm.h"


/*
 * A2REF_UPLOAD=<frames>: upload a wave through the API before the program
 * starts (a2_UploadWave, waves.c:559), hand its handle to the program as one
 * more argument, and release it again after <frames> frames while voices are
 * still playing it - the "wave unloaded under a running oscillator" case of
 * wtosc_check_unloaded (wtosc.c:168-183, waves.c:717-723).
 */
static A2_handle upload_test_wave(A2_interface *i)
{
	static int16_t data[3000];
	int k;
	for(k = 0; k < 3000; ++k)
		data[k] = (int16_t)(((k * 7919) % 4001 - 2000) * 6 + (k % 300 - 150) * 40);
	return a2_UploadWave(i, A2_WMIPWAVE, 0, A2_LOOPED, A2_I16, data, sizeof(data));
}

/* A2REF_SINK=1: attach a sink callback (a2_SinkCallback, audiality2.h.cmake:512)
 * to the voice the program runs on - which then needs an xinsert or xsink unit -
 * and report the peak it saw. */
static int sink_peak;
static A2_errors sink_cb(int32_t **buffers, unsigned nbuffers, unsigned frames, void *userdata)
{
	unsigned c, s;
	for(c = 0; c < nbuffers; ++c)
		for(s = 0; s < frames; ++s)
		{
			int v = buffers[c][s] < 0 ? -buffers[c][s] : buffers[c][s];
			if(v > sink_peak)
				sink_peak = v;
		}
	return A2_OK;
}

int main(int argc, const char *argv[])
{
	int frames, buffer, rate, channels, nargs, pargs[A2_MAXARGS], k, done = 0, c;
	A2_config *cfg;
	A2_driver *drv;
	A2_interface *i;
	A2_handle bank, prog, upwave = -1, vh;
	int release_at = 0;
	FILE *pcm;
	if(argc < 8)
	{
		fprintf(stderr, "usage: ref_render <a2s> <program> <frames> <buffer> "
				"<rate> <channels> <out.pcm> [args...]\n");
		return 1;
	}
	frames = atoi(argv[3]);
	buffer = atoi(argv[4]);
	rate = atoi(argv[5]);
	channels = atoi(argv[6]);
	nargs = argc - 8 > A2_MAXARGS ? A2_MAXARGS : argc - 8;
	for(k = 0; k < nargs; ++k)
		pargs[k] = (int)(atof(argv[8 + k]) * 65536.0);
	if(!(pcm = fopen(argv[7], "wb")))
		return 1;
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
	if(!(i = a2_Open(cfg)))
	{
		fprintf(stderr, "a2_Open failed: %s\n", a2_ErrorString(a2_LastError()));
		return 1;
	}
	if((bank = a2_Load(i, argv[1], 0)) < 0 || (prog = a2_Get(i, bank, argv[2])) < 0)
	{
		fprintf(stderr, "cannot load %s / %s\n", argv[1], argv[2]);
		return 1;
	}
	if(getenv("A2REF_UPLOAD"))
	{
		release_at = atoi(getenv("A2REF_UPLOAD"));
		if((upwave = upload_test_wave(i)) < 0 || nargs >= A2_MAXARGS)
			return 1;
		pargs[nargs++] = upwave << 16;
	}
	a2_TimestampReset(i);
	if((vh = a2_Starta(i, a2_RootVoice(i), prog, nargs, pargs)) < 0)
		return 1;
	if(getenv("A2REF_SINK") && a2_SinkCallback(i, vh, sink_cb, NULL) < 0)
	{
		fprintf(stderr, "a2_SinkCallback failed: %s\n", a2_ErrorString(a2_LastError()));
		return 1;
	}
	while(done < frames)
	{
		int n = frames - done < buffer ? frames - done : buffer;
		if(upwave >= 0 && done >= release_at)
		{
			if(a2_Release(i, upwave))
			{
				/* offline states cannot release (interface.c:496-505) */
				fprintf(stderr, "a2_Release failed: use A2REF_REALTIME=1\n");
				return 1;
			}
			upwave = -1;
		}
		if(a2_Run(i, n) < 0)
			return 1;
		a2_PumpMessages(i);
		for(c = 0; c < channels; ++c)
			fwrite(((A2_audiodriver *)drv)->buffers[c], 4, n, pcm);
		done += n;
	}
	fclose(pcm);
	if(getenv("A2REF_SINK"))
		printf("sink peak %d\n", sink_peak);
	a2_Close(i);
	return 0;
}
