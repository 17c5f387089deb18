// a2amd_render.cpp - a2amd_render(): upload, the kernels of the requested phases (from a hipGraph
// where a quiet batch repeats), the master bus and the taps back to the host; a2amd_collect,
// a2amd_replay, statistics.  (Split out of a2amd_host.cpp in round 3.)
#include "a2amd_host.h"

extern "C" {
// ---- render -------------------------------------------------------------------------
static double now_us()
{
	timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static double g_t[4], g_n;
static double g_why[13];
static double g_cnt[6];		// quiet uploads, records shipped, walk scans, R_NOPs, voices with records, graph launches
struct TimingDump { ~TimingDump() { if(getenv("A2AMD_HOSTTIMING") && g_n) fprintf(stderr,
	"a2amd host timing per render: upload %.1f us, issue %.1f us, readback %.1f us (%g renders; %g quiet uploads, "
	"%g graph launches, %g records, %g voices with records, %g walk scans, %g R_NOPs)\n",
	g_t[0] / g_n, g_t[1] / g_n, g_t[2] / g_n, g_n, g_cnt[0], g_cnt[5], g_cnt[1], g_cnt[4], g_cnt[2], g_cnt[3]);
	if(getenv("A2AMD_HOSTTIMING") && g_n) { fprintf(stderr, "a2amd uploads by first reason against the quiet path "
	"(blob, recs, prev recs, voices, udesc, waves, lists, ptab, dirty voices, fbd, nfrags, bus, none):");
	for(int k = 0; k < 13; ++k) fprintf(stderr, " %g", g_why[k]); fprintf(stderr, "\n"); } } } g_timing_dump;

} // extern "C"
namespace a2h { double *dbg_counters() { return g_cnt; } double *dbg_why() { return g_why; } }
extern "C" {

// What the READ clients of a context's x-units are to be handed: the tapped windows of the
// batch, device -> host (a2amd_unit_tapped reads them).  final: the batch is complete - a
// slot stays tapped into the next batch only while its unit still has READ clients.
} // extern "C"
int a2h::fetch_taps(a2amd_ctx *c, bool final)
{
	bool any = false;
	for(size_t k = 0; k < c->xio.size(); ++k) {
		XioSlot &x = c->xio[k];
		if(x.unit >= 0 && x.tapped) {
			if(final)
				x.tapped = (c->units[x.unit].xio_mode & A2AMD_XIO_TAP) != 0;
			HIPCHK(c, hipMemcpyAsync(x.tap.data(), c->d_xio.d + k * A2D_XIO_SLOT,
					(size_t)c->nfrags * A2AMD_MAXCHANNELS * A2D_FRAG * sizeof(int32_t),
					hipMemcpyDeviceToHost, c->stream));
			any = true;
		}
	}
	if(any)
		HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

extern "C" {
int a2amd_render(a2amd_ctx *c, unsigned phases, int32_t *const *out, unsigned cap)
{
	use_device(c);
	const bool timing = c->hosttiming;	// (debug aid; the accumulators are process-wide and not thread safe)
	double t0 = timing ? now_us() : 0;
	if(!c->stack.empty())
		return c->fail(A2AMD_ESTATE, "render inside an inline window");
	close_fragment(c);
	unsigned total = 0;
	for(int f = 0; f < c->nfrags; ++f)
		total += c->fragframes[f];
	if(!c->nfrags) {
		if(!(phases & A2AMD_RENDER_KEEP))
			;	// nothing recorded: records made outside any fragment wait for the next batch
		return 0;
	}
	c->snap_valid = false;		// (the device's unit states move on)
	if(c->vm.unwalked >= 0)
		return c->fail(A2AMD_ESTATE, "voice %d is run by the device VM, but the host's walk left it out of a fragment "
				"(no default window reported, no call)", c->vm.unwalked);
	if(phases & A2AMD_RENDER_UPLOAD)
		if(int r = upload(c))
			return r;
	double t1 = timing ? now_us() : 0;
	if(timing)
		g_t[0] += t1 - t0;
	if((phases & (A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT)) && !c->uploaded)
		return c->fail(A2AMD_ESTATE, "render phases out of order: upload first");

	if((phases & A2AMD_RENDER_SUBTREES) && c->profiling) {
		if(c->ev_used + 3 > c->ev_pool.size())
			for(int i = 0; i < 3; ++i) {
				hipEvent_t e;
				HIPCHK(c, hipEventCreate(&e));
				c->ev_pool.push_back(e);
			}
		c->ev0 = c->ev_pool[c->ev_used];
		c->ev1 = c->ev_pool[c->ev_used + 1];
		c->ev2 = c->ev_pool[c->ev_used + 2];
		c->ev_used += 3;
	}
	// Where the master bus is to go.  A quiet batch whose root is a plain driver chain, rendered and
	// read back in one call: the root kernel stores the audio straight into the host's pinned (and
	// device-mapped) readback buffer - no copy command behind the kernels, which at configs[1] is
	// 13 of a step's 68 us (the copy and the gap in front of it).
	c->master_dst = nullptr;
	c->master_direct = false;
	int gv = 0;		// graph variant: the destination is an argument of the root's kernel
	{
		const unsigned both = A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT;
		static const bool off = getenv("A2AMD_NO_DIRECT") != nullptr;
		bool taps = false;
		for(const XioSlot &x : c->xio)
			if(x.unit >= 0 && x.tapped)
				taps = true;
		if(!off && (phases & A2AMD_RENDER_READBACK) && (phases & both) == both && !(phases & A2AMD_RENDER_KEEP) &&
				!(phases & A2AMD_RENDER_TAPS) && !c->comm && !c->dist_local && c->uploaded && c->with_recs.empty() &&
				c->consume_ok && c->sub_resume < 0 && !c->paused_at && !taps) {
			const size_t n = (size_t)c->nfrags * c->cfg.channels * A2D_FRAG;
			if(phases & A2AMD_RENDER_ASYNC) {
				if(c->rb_count < 2) {
					const int ri = (c->rb_head + c->rb_count) & 1;
					a2amd_ctx::Readback &rb = c->rb[ri];
					if(n > rb.cap) {
						if(rb.h)
							HIPCHK(c, hipHostFree(rb.h));
						rb.h = nullptr;
						rb.cap = 0;
						HIPCHK(c, hipHostMalloc((void **)&rb.h, n * sizeof(int32_t), hipHostMallocDefault));
						rb.cap = n;
					}
					c->master_dst = rb.h;
					gv = 1 + ri;
				}
			} else if(out) {
				if(n > c->h_master_cap) {
					if(c->h_master)
						HIPCHK(c, hipHostFree(c->h_master));
					c->h_master = nullptr;
					c->h_master_cap = 0;
					HIPCHK(c, hipHostMalloc((void **)&c->h_master, n * sizeof(int32_t), hipHostMallocDefault));
					c->h_master_cap = n;
				}
				c->master_dst = c->h_master;
				gv = 1;
			}
		}
	}
	// A record-free batch that has been seen before runs from a graph - one launch
	// instead of 3-5 separate commands: a kept batch re-run phase by phase
	// (multi-GPU steps), or the engine recording the same quiet batch again.
	auto run_phases = [&](unsigned kphases) -> int {
		if(!kphases)
			return 0;
		if(c->uploaded && !c->profiling && c->stream && c->with_recs.empty() && !getenv("A2AMD_NO_GRAPH") &&
				c->vm.list.empty() &&	// (the VM's passes meet the host in between: not for a graph)
				!(phases & A2AMD_RENDER_TAPS) && c->sub_resume < 0 && !c->paused_at &&
				((phases & A2AMD_RENDER_KEEP) ? (phases & ~A2AMD_RENDER_KEEP) ==
				 (phases & (A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT)) :
				 // (not for a realtime driver's one-fragment batches: measured, hipGraphLaunch
				 // costs more there than the three or four launches it replaces - 170 us
				 // against 25 us of host time per fragment at 65 536 voices)
				 c->quiet_streak >= 1 && c->nfrags >= 8)) {
			const int slot = kphases == A2AMD_RENDER_SUBTREES ? 2 : kphases == A2AMD_RENDER_ROOT ? 3 : 1;
			const int gi = slot + 4 * (slot == 2 ? 0 : gv);
			if((c->gexec[gi] && c->gdst[gi] == c->master_dst) || !build_graph(c, gi, 1, kphases)) {
				if(slot != 3)
					if(int r = ensure_clean(c))
						return r;
				HIPCHK(c, hipGraphLaunch(c->gexec[gi], c->stream));
				c->master_direct = c->gdirect[gi];	// (what the captured launch did)
				if(c->hosttiming)
					dbg_counters()[5] += 1;
				if(slot == 1)
					c->others_clean = c->root_clean = c->consume_ok;
				else if(slot == 2) {
					c->others_clean = c->consume_ok;
					c->root_clean = false;
				} else if(c->consume_ok)
					c->root_clean = true;
				if(kphases & A2AMD_RENDER_ROOT) {
					c->stats.fragments += c->nfrags;
					c->stats.voice_fragments += (uint64_t)c->nfrags * (c->list_all.size() - c->n_list_pads);
				}
				return 0;
			}
		}
		// Events only when profiling (each one from the pool, used once until read):
		// re-recording an event the GPU has not reached yet makes the runtime wait.
		const bool sub = (kphases & A2AMD_RENDER_SUBTREES) != 0, root = (kphases & A2AMD_RENDER_ROOT) != 0;
		return c->profiling ? issue_kernels(c, kphases | (phases & (A2AMD_RENDER_KEEP | A2AMD_RENDER_TAPS)), sub ? c->ev0 : nullptr,
				sub ? c->ev1 : nullptr, root ? c->ev2 : nullptr) :
				issue_kernels(c, kphases | (phases & (A2AMD_RENDER_KEEP | A2AMD_RENDER_TAPS)), nullptr, nullptr, nullptr);
	};
	const unsigned kphases = phases & (A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT);
	if(c->comm && !c->dist_local && kphases == (A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT)) {
		// multi-GPU batch: every rank its subtrees, ONE reduce of the root voice's
		// inline bus over xGMI, the root chain on rank 0 (its panmix must see the
		// sum: the multiply truncates)
		if(int r = run_phases(A2AMD_RENDER_SUBTREES))
			return r;
		if(int r = dist_reduce_root(c))
			return r;
		if(c->dist_rank == 0)
			if(int r = run_phases(A2AMD_RENDER_ROOT))
				return r;
	} else {
		if((kphases || (phases & A2AMD_RENDER_EXCHANGE)) && !(phases & A2AMD_RENDER_UPLOAD)) {
			// what insert clients made of the voices' taps since the render paused
			// (a2amd_unit_insert) joins the voices' output bus before their parents' chains run
			for(size_t k = 0; k < c->xio.size(); ++k) {
				XioSlot &x = c->xio[k];
				if(!x.late_used)
					continue;
				const size_t n = (size_t)c->nfrags * A2AMD_MAXCHANNELS * A2D_FRAG;
				// (also for a voice that died in the course of the batch: its unit and
				// voice entries stay until the batch ends)
				if(x.last_unit >= 0 && x.last_unit < (int)c->units.size() && c->units[x.last_unit].voice >= 0) {
					const HVoice &v = c->voices[c->units[x.last_unit].voice];
					HIPCHK(c, hipMemcpyAsync(c->d_xio.d + k * A2D_XIO_SLOT + A2D_XIO_HALF, x.late.data(),
							n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
					HIPCHK(c, hipStreamSynchronize(c->stream));	// (x.late is pageable and cleared next)
					if(a2d_launch_add_inject(c->d_xio.d + k * A2D_XIO_SLOT + A2D_XIO_HALF, c->d_busmem.d + v.out_off,
							v.out_nch, std::min(c->units[x.last_unit].nin, v.out_nch), c->nfrags, c->stream))
						return c->fail(A2AMD_EHIP, "insert launch failed");
				}
				std::fill(x.late.begin(), x.late.begin() + n, 0);
				x.late_used = false;
			}
		}
		if(int r = run_phases(kphases))
			return r;
	}
	if(c->capture.on && (kphases & A2AMD_RENDER_ROOT))
		if(int r = capture_append(c))
			return r;
	if((phases & A2AMD_RENDER_TAPS) && !(phases & A2AMD_RENDER_READBACK))
		// the seam for insert clients: the batch's taps so far, on the host
		if(int r = fetch_taps(c, false))
			return r;
	double t2 = timing ? now_us() : 0;
	if(timing) {
		g_t[1] += t2 - t1;
		g_n += 1;
	}
	if((phases & A2AMD_RENDER_READBACK) && c->comm && !c->dist_local && c->dist_rank != 0) {
		// (the master bus exists on rank 0 only)
		if(!(phases & A2AMD_RENDER_KEEP))
			end_batch(c);
		return (int)total;
	}
	if(phases & A2AMD_RENDER_READBACK) {
		const int nch = c->cfg.channels;
		size_t n = (size_t)c->nfrags * nch * A2D_FRAG;
		if(phases & A2AMD_RENDER_ASYNC) {
			// enqueue the copy and return: a2amd_collect() waits for it and fills
			// the caller's buffers, up to two batches later
			for(const XioSlot &x : c->xio)
				if(x.unit >= 0 && x.tapped)
					return c->fail(A2AMD_EUNSUPPORTED, "asynchronous readback with READ clients attached");
			if(c->rb_count == 2)
				return c->fail(A2AMD_ESTATE, "two readbacks in flight: a2amd_collect() first");
			a2amd_ctx::Readback &rb = c->rb[(c->rb_head + c->rb_count) & 1];
			if(n > rb.cap) {
				if(rb.h)
					HIPCHK(c, hipHostFree(rb.h));
				rb.h = nullptr;
				rb.cap = 0;
				HIPCHK(c, hipHostMalloc((void **)&rb.h, n * sizeof(int32_t), hipHostMallocDefault));
				rb.cap = n;
			}
			if(!rb.ev)
				HIPCHK(c, hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming));
			if(!(c->master_direct && c->master_dst == rb.h))	// (else the root kernel has stored it there)
				HIPCHK(c, hipMemcpyAsync(rb.h, c->d_busmem.d, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipEventRecord(rb.ev, c->stream));
			rb.nfrags = c->nfrags;
			rb.total = total;
			for(int f = 0; f < c->nfrags; ++f)
				rb.frames[f] = (uint8_t)c->fragframes[f];
			++c->rb_count;
			if(!(phases & A2AMD_RENDER_KEEP))
				end_batch(c);
			return (int)total;
		}
		if(!out)
			return c->fail(A2AMD_EINVAL, "readback without output buffers");
		if(total > cap)
			return c->fail(A2AMD_EINVAL, "output capacity %u < %u frames", cap, total);
		if(n > c->h_master_cap) {
			if(c->h_master)
				HIPCHK(c, hipHostFree(c->h_master));
			HIPCHK(c, hipHostMalloc((void **)&c->h_master, n * sizeof(int32_t), hipHostMallocDefault));
			c->h_master_cap = n;
		}
		if(!(c->master_direct && c->master_dst == c->h_master))	// (else the root kernel has stored it there)
			HIPCHK(c, hipMemcpyAsync(c->h_master, c->d_busmem.d, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
		for(size_t k = 0; k < c->xio.size(); ++k) {
			// what the READ clients of xinsert units are to be handed
			XioSlot &x = c->xio[k];
			if(x.unit >= 0 && x.tapped) {
				x.tapped = (c->units[x.unit].xio_mode & A2AMD_XIO_TAP) != 0;
				HIPCHK(c, hipMemcpyAsync(x.tap.data(), c->d_xio.d + k * A2D_XIO_SLOT,
						(size_t)c->nfrags * A2AMD_MAXCHANNELS * A2D_FRAG * sizeof(int32_t),
						hipMemcpyDeviceToHost, c->stream));
			}
		}
		HIPCHK(c, hipStreamSynchronize(c->stream));
		if(timing) {
			g_t[2] += now_us() - t2;
			static const int trace = getenv("A2AMD_HOSTTIMING") ? atoi(getenv("A2AMD_HOSTTIMING")) : 0;
			if(trace >= 2)
				fprintf(stderr, "a2amd render: %d fragments, upload %.1f us, issue %.1f us, readback %.1f us\n",
						c->nfrags, t1 - t0, t2 - t1, now_us() - t2);
		}
		unsigned pos = 0;
		for(int f = 0; f < c->nfrags; ++f) {
			for(int ch = 0; ch < nch; ++ch)
				memcpy(out[ch] + pos, c->h_master + ((size_t)f * nch + ch) * A2D_FRAG,
						c->fragframes[f] * sizeof(int32_t));
			pos += c->fragframes[f];
		}
		float ms = 0;
		if(c->profiling && hipEventElapsedTime(&ms, c->ev0, c->ev2) == hipSuccess)
			c->stats.last_kernel_ms = ms;
		if(c->profiling && hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess)
			c->stats.last_leaf_ms = ms;
	}
	if(!(phases & A2AMD_RENDER_KEEP) && (phases & (A2AMD_RENDER_READBACK | A2AMD_RENDER_ROOT)))
		end_batch(c);
	return (int)total;
}

int a2amd_collect(a2amd_ctx *c, int32_t *const *out, unsigned cap)
{
	use_device(c);
	if(!c->rb_count)
		return 0;
	a2amd_ctx::Readback &rb = c->rb[c->rb_head];
	if(!out)
		return c->fail(A2AMD_EINVAL, "collect without output buffers");
	if(rb.total > cap)
		return c->fail(A2AMD_EINVAL, "output capacity %u < %u frames", cap, rb.total);
	HIPCHK(c, hipEventSynchronize(rb.ev));
	const int nch = c->cfg.channels;
	unsigned pos = 0;
	for(int f = 0; f < rb.nfrags; ++f) {
		for(int ch = 0; ch < nch; ++ch)
			memcpy(out[ch] + pos, rb.h + ((size_t)f * nch + ch) * A2D_FRAG, rb.frames[f] * sizeof(int32_t));
		pos += rb.frames[f];
	}
	c->rb_head ^= 1;
	--c->rb_count;
	return (int)rb.total;
}

int a2amd_replay(a2amd_ctx *c, unsigned steps)
{
	use_device(c);
	c->snap_valid = false;
	const int GRAPH_STEPS = 8;
	if(!c->uploaded || !c->nfrags)
		return c->fail(A2AMD_ESTATE, "replay without an uploaded batch");
	for(int vi = 0; vi < (int)c->voices.size(); ++vi)
		if(!c->voices[vi].recs.empty())
			return c->fail(A2AMD_ESTATE, "replay of a batch that carries command records");
	bool graphs = c->stream != nullptr && !c->profiling && !getenv("A2AMD_NO_GRAPH") && c->vm.list.empty();
	c->master_dst = nullptr;	// (replayed steps are not read back: the master bus stays in the bus memory)
	c->master_direct = false;
	if(graphs && (!c->gexec[0] || !c->gexec[1])) {
		if(build_graph(c, 0, GRAPH_STEPS) || build_graph(c, 1, 1)) {
			drop_graphs(c);
			graphs = false;
		}
	}
	while(steps) {
		if(graphs && steps >= (unsigned)GRAPH_STEPS) {
			if(int r = ensure_clean(c))
				return r;
			HIPCHK(c, hipGraphLaunch(c->gexec[0], c->stream));
			c->others_clean = c->root_clean = c->consume_ok;
			steps -= GRAPH_STEPS;
			c->stats.fragments += (uint64_t)c->nfrags * GRAPH_STEPS;
			c->stats.voice_fragments += (uint64_t)c->nfrags * (c->list_all.size() - c->n_list_pads) * GRAPH_STEPS;
		} else if(graphs) {
			if(int r = ensure_clean(c))
				return r;
			HIPCHK(c, hipGraphLaunch(c->gexec[1], c->stream));
			c->others_clean = c->root_clean = c->consume_ok;
			--steps;
			c->stats.fragments += c->nfrags;
			c->stats.voice_fragments += (uint64_t)c->nfrags * (c->list_all.size() - c->n_list_pads);
		} else {
			if(c->profiling) {
				if(c->ev_used + 3 > c->ev_pool.size())
					for(int i = 0; i < 3; ++i) {
						hipEvent_t e;
						HIPCHK(c, hipEventCreate(&e));
						c->ev_pool.push_back(e);
					}
				c->ev0 = c->ev_pool[c->ev_used];
				c->ev1 = c->ev_pool[c->ev_used + 1];
				c->ev2 = c->ev_pool[c->ev_used + 2];
				c->ev_used += 3;
			}
			if(int r = c->profiling ?
					issue_kernels(c, A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT, c->ev0, c->ev1, c->ev2) :
					issue_kernels(c, A2AMD_RENDER_SUBTREES | A2AMD_RENDER_ROOT, nullptr, nullptr, nullptr))
				return r;
			--steps;
		}
	}
	return A2AMD_OK;
}

static int drain_events(a2amd_ctx *c)
{
	if(!c->ev_used)
		return 0;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	for(size_t i = 0; i + 2 < c->ev_used + 0 && i + 2 < c->ev_pool.size(); i += 3) {
		float a = 0, b = 0;
		if(hipEventElapsedTime(&a, c->ev_pool[i], c->ev_pool[i + 1]) == hipSuccess &&
				hipEventElapsedTime(&b, c->ev_pool[i], c->ev_pool[i + 2]) == hipSuccess) {
			c->stats.timed_leaf_ms += a;
			c->stats.timed_all_ms += b;
			++c->stats.timed_batches;
		}
	}
	c->ev_used = 0;
	return 0;
}

int a2amd_get_stats(a2amd_ctx *c, a2amd_stats *st)
{
	if(c->profiling)
		if(int r = drain_events(c))
			return r;
	*st = c->stats;
	return A2AMD_OK;
}

int a2amd_set_profiling(a2amd_ctx *c, int on)
{
	if(int r = drain_events(c))
		return r;
	if(on) {
		c->stats.timed_leaf_ms = c->stats.timed_all_ms = 0;
		c->stats.timed_batches = 0;
	}
	c->profiling = on != 0;
	return A2AMD_OK;
}

} // extern "C"
