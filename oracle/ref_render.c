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
#include "a2_units.h"


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
static unsigned long long sink_hash = 0xCBF29CE484222325ull, sink_frames;
static A2_errors sink_cb(int32_t **buffers, unsigned nbuffers, unsigned frames, void *userdata)
{
	unsigned c, s, b;
	for(c = 0; c < nbuffers; ++c)
		for(s = 0; s < frames; ++s)
		{
			int v = buffers[c][s] < 0 ? -buffers[c][s] : buffers[c][s];
			if(v > sink_peak)
				sink_peak = v;
			/* FNV-1a over everything the sink was handed, in the order it came */
			for(b = 0; b < 4; ++b)
				sink_hash = (sink_hash ^ (((unsigned)buffers[c][s] >> (8 * b)) & 255)) * 0x100000001B3ull;
		}
	sink_frames += frames;
	return A2_OK;
}

/* A2REF_SOURCE=1: attach a source callback (a2_SourceCallback,
 * audiality2.h.cmake:525) to the same voice: a deterministic sawtooth pair that
 * does not care how the engine slices its windows. */
static unsigned source_pos;
static A2_errors source_cb(int32_t **buffers, unsigned nbuffers, unsigned frames, void *userdata)
{
	unsigned c, s;
	if(!buffers)
		return A2_OK;	/* removal notification */
	for(c = 0; c < nbuffers; ++c)
		for(s = 0; s < frames; ++s)
			buffers[c][s] = (int)(((source_pos + s) * (c ? 77003u : 52361u)) & 0x3fffff) - 0x200000;
	source_pos += frames;
	return A2_OK;
}

/* A2REF_INSERT=1: an insert callback (a2_InsertCallback, audiality2.h.cmake:538):
 * reads what it is handed and writes back a function of it and of the time. */
static unsigned insert_pos;
static A2_errors insert_cb(int32_t **buffers, unsigned nbuffers, unsigned frames, void *userdata)
{
	unsigned c, s;
	if(!buffers)
		return A2_OK;
	for(c = 0; c < nbuffers; ++c)
		for(s = 0; s < frames; ++s)
			buffers[c][s] = buffers[c][s] / 2 + (int)(((insert_pos + s) * 31u) & 0xffff) - 0x8000;
	insert_pos += frames;
	return A2_OK;
}

/* (a second one, for the outer group of A2REF_NEST / A2REF_INSERT_OUTER) */
static unsigned insert2_pos;
static A2_errors insert2_cb(int32_t **buffers, unsigned nbuffers, unsigned frames, void *userdata)
{
	unsigned c, s;
	if(!buffers)
		return A2_OK;
	for(c = 0; c < nbuffers; ++c)
		for(s = 0; s < frames; ++s)
			buffers[c][s] = -buffers[c][s] / 3 + (int)(((insert2_pos + s) * 7u) & 0x3fff);
	insert2_pos += frames;
	return A2_OK;
}

/* A2REF_FOREIGN=1: the application registers a unit of its own
 * (a2_RegisterUnit, a2_units.h:327) before the script is compiled: "thru", 1-2
 * channels, copies (or adds) its inputs to its outputs. */
static void thru_process(A2_unit *u, unsigned offset, unsigned frames)
{
	unsigned c, s;
	for(c = 0; c < u->noutputs; ++c)
		for(s = offset; s < offset + frames; ++s)
			u->outputs[c][s] = u->inputs[c][s];
}
static void thru_process_add(A2_unit *u, unsigned offset, unsigned frames)
{
	unsigned c, s;
	for(c = 0; c < u->noutputs; ++c)
		for(s = offset; s < offset + frames; ++s)
			u->outputs[c][s] += u->inputs[c][s];
}
static A2_errors thru_init(A2_unit *u, A2_vmstate *vms, void *statedata, unsigned flags)
{
	u->Process = (flags & A2_PROCADD) ? thru_process_add : thru_process;
	return A2_OK;
}
static const A2_unitdesc thru_unitdesc = { "thru", A2_MATCHIO, NULL, NULL, NULL, 1, 2, 1, 2,
		sizeof(A2_unit), thru_init, NULL, NULL, NULL };
/* ... and "tone", a source: no inputs, a fixed sawtooth on 1-2 outputs */
static void tone_process(A2_unit *u, unsigned offset, unsigned frames)
{
	unsigned c, s;
	for(c = 0; c < u->noutputs; ++c)
		for(s = offset; s < offset + frames; ++s)
			u->outputs[c][s] = (int)((s * 65536u) & 0x3fffff) - 0x200000;
}
static A2_errors tone_init(A2_unit *u, A2_vmstate *vms, void *statedata, unsigned flags)
{
	u->Process = tone_process;
	return A2_OK;
}
static const A2_unitdesc tone_unitdesc = { "tone", 0, NULL, NULL, NULL, 0, 0, 1, 2,
		sizeof(A2_unit), tone_init, NULL, NULL, NULL };

int main(int argc, const char *argv[])
{
	int frames, buffer, rate, channels, nargs, pargs[A2_MAXARGS], k, done = 0, c;
	A2_config *cfg;
	A2_driver *drv;
	A2_interface *i;
	A2_handle bank, prog, upwave = -1, vh, ch, srcstream = -1, sinkstream = -1;
	unsigned long long stream_hash = 0xCBF29CE484222325ull, stream_frames = 0;
	int release_at = 0, sink_dies = 0, sinkstream_done = 0;
	/* A2REF_KILL=<frame>: a2_Kill() the voice the program runs on, timestamped */
	int kill_at = getenv("A2REF_KILL") ? atoi(getenv("A2REF_KILL")) : -1;
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
			(getenv("A2REF_REALTIME") ? A2_REALTIME : 0) |
			(getenv("A2REF_KILL") ? A2_TIMESTAMP : 0))))
		return 1;
	a2_AddDriver(cfg, drv);
	if(!(i = a2_Open(cfg)))
	{
		fprintf(stderr, "a2_Open failed: %s\n", a2_ErrorString(a2_LastError()));
		return 1;
	}
	if(getenv("A2REF_FOREIGN") && ((k = a2_RegisterUnit(i, &thru_unitdesc)) < 0 ||
			a2_Export(i, A2_ROOTBANK, k, NULL) || (k = a2_RegisterUnit(i, &tone_unitdesc)) < 0 ||
			a2_Export(i, A2_ROOTBANK, k, NULL)))	/* (as a2_Open does for its own, audiality2.c:256-262) */
	{
		fprintf(stderr, "a2_RegisterUnit failed: %s\n", a2_ErrorString(a2_LastError()));
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
	{
		/* A2REF_NEST=n: the program is started under n nested groups (a2_NewGroup,
		 * audiality2.h.cmake: the a2_groupdriver voices, inline; panmix; xinsert) instead
		 * of straight under the root voice; A2REF_INSERT_OUTER=1 puts an insert client on
		 * the outermost of them as well */
		A2_handle parent = a2_RootVoice(i);
		int k, nest = getenv("A2REF_NEST") ? atoi(getenv("A2REF_NEST")) : 0;
		for(k = 0; k < nest; ++k)
		{
			if((parent = a2_NewGroup(i, parent)) < 0)
			{
				fprintf(stderr, "a2_NewGroup failed: %s\n", a2_ErrorString(a2_LastError()));
				return 1;
			}
			if(!k && getenv("A2REF_INSERT_OUTER") && a2_InsertCallback(i, parent, insert2_cb, NULL) < 0)
				return 1;
		}
		if((vh = a2_Starta(i, parent, prog, nargs, pargs)) < 0)
			return 1;
		/* A2REF_SIBLING=1: a second subtree under the root voice, made AFTER the program's
		 * voice (so the engine's newest-first walk reaches it first): a group
		 * (a2_NewGroup) playing the same program.  With the drop-in's A2AMD_DEVICES=2
		 * the program's own voice - the one the clients below attach to - then lives
		 * in the SECOND backend context. */
		if(getenv("A2REF_SIBLING"))
		{
			A2_handle g = a2_NewGroup(i, a2_RootVoice(i));
			if(g < 0 || a2_Starta(i, g, prog, nargs, pargs) < 0)
				return 1;
		}
	}
	/* A2REF_ROOTCLIENTS=1: the clients below go on the root voice (the root driver's
	 * xinsert, audiality2.c:271-291 - where a2play puts its sink) instead */
	if(getenv("A2REF_ROOTCLIENTS"))
		ch = a2_RootVoice(i);
	else
		ch = vh;
	if(getenv("A2REF_SOURCE") && a2_SourceCallback(i, ch, source_cb, NULL) < 0)
	{
		fprintf(stderr, "a2_SourceCallback failed: %s\n", a2_ErrorString(a2_LastError()));
		return 1;
	}
	if(getenv("A2REF_SINK") && a2_SinkCallback(i, ch, sink_cb, NULL) < 0)
	{
		fprintf(stderr, "a2_SinkCallback failed: %s\n", a2_ErrorString(a2_LastError()));
		return 1;
	}
	if(getenv("A2REF_INSERT") && a2_InsertCallback(i, ch, insert_cb, NULL) < 0)
		return 1;
	/* A2REF_SRCSTREAM=<channel> / A2REF_SINKSTREAM=<channel>: the buffered
	 * variants (a2_OpenSource / a2_OpenSink, audiality2.h.cmake:561-577): a
	 * source stream feeding that channel of the voice, written ahead of every
	 * a2_Run(), and a sink stream, drained after it. */
	{
		/* A2REF_STREAMS=1 = A2REF_SRCSTREAM=1 A2REF_SINKSTREAM=0 (the channels) */
		const char *both = getenv("A2REF_STREAMS");
		const char *src = getenv("A2REF_SRCSTREAM") ? getenv("A2REF_SRCSTREAM") : both ? "1" : NULL;
		const char *snk = getenv("A2REF_SINKSTREAM") ? getenv("A2REF_SINKSTREAM") : both ? "0" : NULL;
		if((src && (srcstream = a2_OpenSource(i, ch, atoi(src), 4 * buffer, 0)) < 0) ||
				(snk && (sinkstream = a2_OpenSink(i, ch, atoi(snk), 4 * buffer, 0)) < 0))
		{
			fprintf(stderr, "cannot open streams: %s\n", a2_ErrorString(a2_LastError()));
			return 1;
		}
	}
	while(done < frames)
	{
		int n = frames - done < buffer ? frames - done : buffer;
		if(kill_at >= 0 && done >= kill_at)
		{
			/* 0.7 ms into the coming buffer: the voice dies in the middle of a
			 * fragment, with whatever clients it has */
			a2_TimestampReset(i);
			a2_TimestampBump(i, a2_ms2Timestamp(i, 0.7));
			a2_Kill(i, vh);
			kill_at = -1;
			srcstream = -1;		/* (its reader is going away) */
			/* ... and so is the sink stream's writer: in an offline state the engine
			 * free()s the client object the moment the voice dies
			 * (a2_XinsertRemoveClient, xinsertapi.c:149-153) while the stream handle
			 * keeps pointing at it (xi_stream_available, xinsertapi.c:376-381), so
			 * the stream must not be touched after this a2_Run() */
			sink_dies = 1;
		}
		if(srcstream >= 0)
		{
			int32_t sb[4096];
			for(k = 0; k < n && k < 4096; ++k)
				sb[k] = (int)(((done + k) * 40503u) & 0x3fffff) - 0x200000;
			/* (A2_I24 = the engine's 8:24: the only format these streams take, xinsertapi.c:340) */
			if((k = a2_Write(i, srcstream, A2_I24, sb, n * 4)))
			{
				fprintf(stderr, "a2_Write: %s\n", a2_ErrorString(k));
				return 1;
			}
		}
		/* A2REF_REUPLOAD=1 (with A2REF_UPLOAD): half way through, kill the voices that
		 * play the uploaded wave, release it two buffers later - while NOTHING plays
		 * it -, and two buffers after that upload a different wave (which tends to get
		 * the released A2_wave's address) and start the program again on it */
		if(getenv("A2REF_REUPLOAD") && upwave >= 0)
		{
			static int phase, at;
			if(phase == 0 && done >= frames / 2)
			{
				a2_Kill(i, vh);
				phase = 1;
				at = done;
			}
			else if(phase == 1 && done >= at + 2 * buffer)
			{
				if(a2_Release(i, upwave))
				{
					fprintf(stderr, "a2_Release failed: use A2REF_REALTIME=1\n");
					return 1;
				}
				phase = 2;
			}
			else if(phase == 2 && done >= at + 4 * buffer)
			{
				static int16_t data2[3000];
				for(k = 0; k < 3000; ++k)
					data2[k] = (int16_t)(((k * 4001) % 7919 - 3959) * 4);
				if((upwave = a2_UploadWave(i, A2_WMIPWAVE, 0, A2_LOOPED, A2_I16, data2, sizeof(data2))) < 0)
					return 1;
				pargs[nargs - 1] = upwave << 16;
				if((vh = a2_Starta(i, a2_RootVoice(i), prog, nargs, pargs)) < 0)
					return 1;
				phase = 3;
			}
		}
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
		if(sink_dies)
			sinkstream_done = 1;
		if(sinkstream >= 0 && !sinkstream_done)
		{
			int32_t sb[4096];
			int avail = a2_Available(i, sinkstream);
			unsigned b;
			if(avail > 4096)
				avail = 4096;
			if(avail > 0 && !a2_Read(i, sinkstream, A2_I24, sb, avail * 4))
			{
				for(k = 0; k < avail; ++k)
					for(b = 0; b < 4; ++b)
						stream_hash = (stream_hash ^ (((unsigned)sb[k] >> (8 * b)) & 255)) * 0x100000001B3ull;
				stream_frames += avail;
			}
		}
		for(c = 0; c < channels; ++c)
			fwrite(((A2_audiodriver *)drv)->buffers[c], 4, n, pcm);
		done += n;
	}
	fclose(pcm);
	if(sinkstream >= 0)
		printf("sink stream frames %llu hash %016llx\n", stream_frames, stream_hash);
	if(getenv("A2REF_SINK"))
		printf("sink peak %d frames %llu hash %016llx\n", sink_peak, sink_frames, sink_hash);
	a2_Close(i);
	return 0;
}
