// a2amd_host.cpp - host half of liba2amd.so: the C ABI of include/a2amd.h.
//
// Records what the Audiality 2 engine asks of its units during a batch of
// fragments (per voice, in call order), keeps the little host-side state the
// reference itself computes with libm or draws from the engine-global RNG
// (filter12's cutoff -> coefficient, the noise oscillators' draw counts), and
// at a2amd_render() ships the records and launches the kernels:
//
//   leaf voices (no 'inline' unit)            one launch, all in parallel
//   voices with an 'inline' unit, by nesting  one launch per depth, deepest first
//
// which is the reference's depth-first voice walk (src/core.c:1883-1896) turned
// inside out: children only ever ADD into the bus their parent's inline unit
// collects (core.c:479-480, inline.c:32-33), so all children of all parents can
// run first and each parent picks the sum up where its inline unit sits.
#include "a2amd_host.h"

namespace a2h { thread_local char g_err[256] = ""; }

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char *a2amd_version(void) { return "a2amd 0.1 (gfx950)"; }
#ifndef A2AMD_SRCHASH
#define A2AMD_SRCHASH "unstamped"
#endif
const char *a2amd_source_stamp(void) { return "A2AMD_SRCHASH:" A2AMD_SRCHASH; }

const char *a2amd_last_error(const a2amd_ctx *c) { return c ? c->err : g_err; }

int a2amd_device_count(void)
{
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}


int a2amd_open(const a2amd_config *cfg, a2amd_ctx **out)
{
	if(!cfg || !out || cfg->channels < 1 || cfg->channels > A2D_MAXCH || cfg->samplerate <= 0) {
		snprintf(g_err, sizeof(g_err), "a2amd_open: bad configuration");
		return A2AMD_EINVAL;
	}
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if(e != hipSuccess || ndev <= 0 || cfg->device >= ndev) {
		snprintf(g_err, sizeof(g_err), "a2amd_open: no usable HIP device (%s, %d devices, want #%d); "
				"this library has no CPU fallback", hipGetErrorString(e), ndev, cfg->device);
		return A2AMD_ENODEVICE;
	}
	a2amd_ctx *c = new a2amd_ctx();
	c->cfg = *cfg;
	c->err[0] = 0;
	if(!c->cfg.max_batch)
		c->cfg.max_batch = 1;
	if(c->cfg.max_batch > A2D_MAXBATCH)
		c->cfg.max_batch = A2D_MAXBATCH;
	memset(&c->stats, 0, sizeof(c->stats));
	build_pitch_table(c->ptab);
	c->no_fast = getenv("A2AMD_NO_FAST") ? atoi(getenv("A2AMD_NO_FAST")) : 0;
	c->hosttiming = getenv("A2AMD_HOSTTIMING") != nullptr;
	c->bus_stride_frames = (size_t)c->cfg.max_batch * A2D_FRAG;
	c->bus_used = c->bus_stride_frames * (size_t)c->cfg.channels;	// master bus at offset 0
#define OPENCHK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) { \
	snprintf(g_err, sizeof(g_err), "a2amd_open: %s: %s", #call, hipGetErrorString(e_)); \
	delete c; return A2AMD_EHIP; } } while(0)
	OPENCHK(hipSetDevice(cfg->device));
	if(cfg->stream)
		c->stream = (hipStream_t)cfg->stream;
	else {
		OPENCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		c->own_stream = true;
	}
	OPENCHK(hipEventCreate(&c->ev0));
	OPENCHK(hipEventCreate(&c->ev1));
	OPENCHK(hipEventCreate(&c->ev2));
	c->ev_pool.push_back(c->ev0);
	c->ev_pool.push_back(c->ev1);
	c->ev_pool.push_back(c->ev2);
	OPENCHK(hipMalloc((void **)&c->d_ptab, sizeof(c->ptab)));
	OPENCHK(hipMalloc((void **)&c->d_wavepool.d, (size_t)(8u << 20) * sizeof(int16_t)));
	c->d_wavepool.cap = 8u << 20;
	OPENCHK(hipMalloc((void **)&c->d_wavecoef.d, (size_t)(8u << 20) * A2D_COEF_WORDS * sizeof(int32_t)));
	c->d_wavecoef.cap = 8u << 20;
#undef OPENCHK
	*out = c;
	return A2AMD_OK;
}

void a2amd_close(a2amd_ctx *c)
{
	if(!c)
		return;
	use_device(c);
	hipStreamSynchronize(c->stream);
	// (round 6) ... and the device VM's speculative pass: launched behind the LAST batch on a non-blocking stream of its
	// own, waited for by nobody if no batch follows, and reading voices / unit state / runs - which are freed below,
	// before vm_close() gets to its stream (hipFree does not wait for it: a memory access fault at exit, intermittent,
	// in every short render with device VM voices)
	if(c->vm.pred_stream)
		hipStreamSynchronize(c->vm.pred_stream);
	drop_graphs(c);
	if(c->comm && g_rccl.CommDestroy)
		g_rccl.CommDestroy(c->comm);
	if(c->grp_ev)
		hipEventDestroy(c->grp_ev);
	hipFree(c->d_voices.d); hipFree(c->d_vext.d); hipFree(c->d_udesc.d); hipFree(c->d_ustate.d); hipFree(c->d_ustage.d);
	hipFree(c->d_vactive.d); hipFree(c->d_runs.d); hipFree(c->d_recs.d);
	hipFree(c->d_win.d); hipFree(c->d_wext.d); hipFree(c->d_wscr.d); hipFree(c->d_wrc.d); hipFree(c->d_widx.d); hipFree(c->d_wtop);
	if(c->h_wtop)
		hipHostFree(c->h_wtop);
	if(c->wtop_ev)
		hipEventDestroy(c->wtop_ev);
	if(c->win_stream)
		hipStreamDestroy(c->win_stream);
	for(int k = 0; k < 7; ++k)
		if(c->win_fork[k])
			hipStreamDestroy(c->win_fork[k]);
	for(int k = 0; k < 8; ++k)
		if(c->win_fev[k])
			hipEventDestroy(c->win_fev[k]);
	for(int k = 0; k < 5; ++k)
		if(c->win_ev[k])
			hipEventDestroy(c->win_ev[k]);
	hipFree(c->d_waves.d); hipFree(c->d_wavepool.d); hipFree(c->d_wavecoef.d); hipFree(c->d_busmem.d);
	vm_close(c);
	hipFree(c->capture.d); hipFree(c->capture.d_fragpos);
	hipFree(c->d_fbdmem.d); hipFree(c->d_fmstate.d); hipFree(c->d_xio.d); hipFree(c->d_fmsine); hipFree(c->d_list.d); hipFree(c->d_scatter.d); hipFree(c->d_ptab); hipFree(c->d_blob.d);
	for(int k = 0; k < 2; ++k) { if(c->h_blob[k]) hipHostFree(c->h_blob[k]); if(c->blob_ev[k]) hipEventDestroy(c->blob_ev[k]); }
	if(c->h_master)
		hipHostFree(c->h_master);
	for(int k = 0; k < 2; ++k) { if(c->rb[k].h) hipHostFree(c->rb[k].h); if(c->rb[k].ev) hipEventDestroy(c->rb[k].ev); }
	for(hipEvent_t e : c->ev_pool)
		hipEventDestroy(e);
	if(c->own_stream)
		hipStreamDestroy(c->stream);
	delete c;
}

int a2amd_set_pitch_table(a2amd_ctx *c, const uint32_t *t)
{
	memcpy(c->ptab, t, sizeof(c->ptab));
	c->ptab_dirty = true;
	return A2AMD_OK;
}

int a2amd_get_pitch_table(const a2amd_ctx *c, uint32_t *t)
{
	memcpy(t, c->ptab, sizeof(c->ptab));
	return A2AMD_OK;
}

// ---- waves ------------------------------------------------------------------
} // extern "C"
// Which waves the settled wtosc -> panmix kernel reads as SAMPLES (A2D_WF_RAWTAPS) rather than as Hermite
// coefficient entries.  The entries save a third of the interpolation's vector instructions and cost six
// times the bytes (12 per sample and tap against 2): right while the table sits in L2 / the Infinity
// Cache - the 24 built-in waves are 1.5 MB of entries - wrong for a bank of private sample waves that no
// cache holds (north_star: "wavetable / mipmap data ... behind coalesced HBM gathers").  So: by footprint.
// Once the pool's entries outgrow A2AMD_COEF_CACHE_MB (default 192, about what the caches keep of a
// streaming set), every wave of 4 096 samples or more is played from its samples.  A2AMD_RAW=0 / 1 forces
// the choice for all waves (A/B measurements, tests).
namespace a2h {
void wave_tap_policy(a2amd_ctx *c)
{
	const char *e = getenv("A2AMD_RAW"), *m = getenv("A2AMD_COEF_CACHE_MB");
	const int force = (e && *e) ? atoi(e) : -1;
	const size_t budget = (size_t)((m && *m) ? atoi(m) : 192) << 20;
	const bool big = c->wavepool_used * (4 * A2D_COEF_WORDS) > budget;
	for(size_t i = 0; i < c->waves.size(); ++i) {
		HWave &w = c->waves[i];
		if(!w.live || w.dw.type != A2AMD_WMIPWAVE)
			continue;
		const bool raw = force >= 0 ? force != 0 : (big && w.dw.size[0] >= 4096);
		const uint32_t f = (w.dw.flags & ~A2D_WF_RAWTAPS) | (raw ? A2D_WF_RAWTAPS : 0u);
		if(f != w.dw.flags) {
			w.dw.flags = f;
			c->mwaves[i].flags = f;
			c->waves_dirty = true;
		}
	}
}
}
extern "C" {
} // extern "C"

struct a2amd_capture {
	int32_t *d;
	size_t n;
	int device;
};

namespace a2h {
int capture_append(a2amd_ctx *c)
{
	a2amd_ctx::Capture &cp = c->capture;
	std::vector<uint32_t> pos((size_t)c->nfrags + 1);
	size_t n = cp.n;
	for(int f = 0; f < c->nfrags; ++f) {
		pos[f] = (uint32_t)n;
		n += c->fragframes[f];
	}
	pos[c->nfrags] = (uint32_t)n;
	if(n > 0x7fffffffu)
		return c->fail(A2AMD_ENOMEM, "capture beyond 2^31 frames");
	if(n > cp.cap) {
		// (the stream is in order: the copy follows the kernels that wrote the old buffer)
		const size_t ncap = std::max(n * 2, (size_t)1 << 16);
		int32_t *nd = nullptr;
		HIPCHK(c, hipMalloc((void **)&nd, ncap * sizeof(int32_t)));
		if(cp.n)
			HIPCHK(c, hipMemcpyAsync(nd, cp.d, cp.n * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		hipFree(cp.d);
		cp.d = nd;
		cp.cap = ncap;
	}
	if(!cp.d_fragpos)
		HIPCHK(c, hipMalloc((void **)&cp.d_fragpos, (A2D_MAXBATCH + 1) * sizeof(uint32_t)));
	HIPCHK(c, hipMemcpyAsync(cp.d_fragpos, pos.data(), pos.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));	// (pos is pageable and about to go)
	if(a2d_launch_capture(c->d_busmem.d, cp.d, cp.d_fragpos, c->nfrags, c->cfg.channels, c->stream))
		return c->fail(A2AMD_EHIP, "capture launch failed");
	cp.n = n;
	return 0;
}
} // namespace a2h

extern "C" {

int a2amd_capture_begin(a2amd_ctx *c)
{
	if(c->comm)
		return c->fail(A2AMD_EUNSUPPORTED, "capture in a distributed context");
	c->capture.on = true;
	c->capture.n = 0;
	drop_graphs(c);		// (a captured launch sequence may store the master bus in host memory)
	return A2AMD_OK;
}

int a2amd_capture_end(a2amd_ctx *c, a2amd_capture **out)
{
	if(!out)
		return c->fail(A2AMD_EINVAL, "capture_end: null");
	*out = nullptr;
	use_device(c);
	HIPCHK(c, hipStreamSynchronize(c->stream));
	if(c->capture.on && c->capture.n) {
		a2amd_capture *cap = new a2amd_capture;
		cap->d = c->capture.d;
		cap->n = c->capture.n;
		cap->device = c->cfg.device;
		c->capture.d = nullptr;
		c->capture.cap = 0;
		*out = cap;
	}
	c->capture.on = false;
	c->capture.n = 0;
	return A2AMD_OK;
}

unsigned a2amd_capture_frames(const a2amd_capture *cap) { return cap ? (unsigned)cap->n : 0; }

void a2amd_capture_free(a2amd_capture *cap)
{
	if(!cap)
		return;
	int dev = 0;
	hipGetDevice(&dev);
	hipSetDevice(cap->device);
	hipFree(cap->d);
	hipSetDevice(dev);
	delete cap;
}

int a2amd_wave_stats(a2amd_ctx *c, uint64_t *h2d, uint32_t *uploaded, uint32_t *resident)
{
	if(h2d) *h2d = c->wave_h2d_bytes;
	if(uploaded) *uploaded = c->waves_uploaded;
	if(resident) *resident = c->waves_resident;
	return A2AMD_OK;
}

static int wave_place(a2amd_ctx *c, uint64_t key, const a2amd_wavedesc *w, const a2amd_capture *cap);

int a2amd_wave_upload(a2amd_ctx *c, uint64_t key, const a2amd_wavedesc *w)
{
	if(!w)
		return c->fail(A2AMD_EINVAL, "wave_upload: null descriptor");
	return wave_place(c, key, w, nullptr);
}

int a2amd_wave_upload_captured(a2amd_ctx *c, uint64_t key, const a2amd_wavedesc *w, const a2amd_capture *cap)
{
	if(!w || !cap)
		return c->fail(A2AMD_EINVAL, "wave_upload_captured: null");
	if(cap->device != c->cfg.device)
		return c->fail(A2AMD_EUNSUPPORTED, "wave_upload_captured: the capture lives on GPU %d, the context on %d", cap->device, c->cfg.device);
	if(w->type != A2AMD_WWAVE && w->type != A2AMD_WMIPWAVE)
		return c->fail(A2AMD_EUNSUPPORTED, "wave_upload_captured: wave type %d", w->type);
	// A2_NORMALIZE 0x10000, A2_XFADE 0x40000, A2_REVMIX 0x80000 (include/a2_waves.h:113-115): a2_postprocess and the
	// normalising conversion of src/waves.c are host work on the whole wave
	if(w->flags & 0x000d0000u)
		return c->fail(A2AMD_EUNSUPPORTED, "wave_upload_captured: flags %#x ask for post-processing", w->flags);
	if((size_t)w->size[0] != cap->n)
		return c->fail(A2AMD_EUNSUPPORTED, "wave_upload_captured: the wave has %u samples, the capture %zu", w->size[0], cap->n);
	for(int l = 1; l < (w->type == A2AMD_WMIPWAVE ? A2D_MIPS : 1); ++l)
		if(w->size[l] != ((w->size[0] + (1u << l) - 1) >> l))	// a2_wave_alloc, waves.c:76
			return c->fail(A2AMD_EINVAL, "wave_upload_captured: level %d has %u samples", l, w->size[l]);
	return wave_place(c, key, w, cap);
}

static int wave_place(a2amd_ctx *c, uint64_t key, const a2amd_wavedesc *w, const a2amd_capture *cap)
{
	use_device(c);
	int id = -1;
	for(size_t i = 0; i < c->waves.size(); ++i)
		if(c->waves[i].live && c->waves[i].key == key)
			id = (int)i;
	if(id < 0) {
		if(!c->free_wave_slots.empty()) {
			id = c->free_wave_slots.back();
			c->free_wave_slots.pop_back();
		} else {
			id = (int)c->waves.size();
			c->waves.push_back(HWave());
			c->mwaves.push_back(A2DWave());
		}
	} else {
		// same key again: the old data is replaced (what is recorded plays the old)
		if(c->waves[id].pool_len)
			c->deferred_wavepool_free.push_back(std::make_pair(c->waves[id].pool_off, c->waves[id].pool_len));
		--c->stats.live_waves;
	}
	HWave &hw = c->waves[id];
	hw.live = true;
	hw.key = key;
	memset(&hw.dw, 0, sizeof(hw.dw));
	hw.dw.type = w->type;
	hw.dw.flags = w->flags;
	hw.dw.period = w->period;
	int levels = w->type == A2AMD_WMIPWAVE ? A2D_MIPS : w->type == A2AMD_WWAVE ? 1 : 0;
	size_t total = 0;
	for(int l = 0; l < levels; ++l)
		total += A2AMD_WAVEPRE + (size_t)w->size[l] + A2AMD_WAVEPOST;
	total = (total + 7) & ~(size_t)7;
	// a region a dropped wave left behind (first fit), else the end of the pool
	size_t pos = (size_t)-1;
	for(size_t k = 0; k < c->wavepool_free.size() && total; ++k)
		if(c->wavepool_free[k].second >= total) {
			pos = c->wavepool_free[k].first;
			c->wavepool_free[k].first += total;
			c->wavepool_free[k].second -= total;
			if(!c->wavepool_free[k].second)
				c->wavepool_free.erase(c->wavepool_free.begin() + (long)k);
			// work in flight may still read the old contents
			HIPCHK(c, hipStreamSynchronize(c->stream));
			break;
		}
	if(pos == (size_t)-1) {
		if(c->wavepool_used + total > c->d_wavepool.cap) {
			// (the kernels address the coefficient table by unsigned 32 bit byte offsets, A2D_COEF_WORDS words
			// per sample: 2^32 / 12 = 357 M samples = 0.7 GB of wave data, 4.3 GB of entries)
			if((c->wavepool_used + total) * (4 * A2D_COEF_WORDS) > (((size_t)1 << 32) - 4096))
				return c->fail(A2AMD_ENOMEM, "wave pool beyond %zu samples", (((size_t)1 << 32) - 4096) / (4 * A2D_COEF_WORDS));
			if(int r = grow(c, c->d_wavepool, c->wavepool_used + total, 1, true)) return r;
			// the coefficient table follows the pool: rebuilt for what is in it
			if(int r = grow(c, c->d_wavecoef, c->d_wavepool.cap, A2D_COEF_WORDS, false)) return r;
			if(c->wavepool_used > 3 && a2d_launch_build_coef(c->d_wavepool.d, c->d_wavecoef.d, 1,
					(unsigned)c->wavepool_used - 2, c->stream))
				return c->fail(A2AMD_EHIP, "coefficient build failed");
			c->waves_dirty = true;
			drop_graphs(c);		// (the kernels take the table's address as an argument)
			c->blob_quiet = false;
		}
		pos = c->wavepool_used;
		c->wavepool_used += total;
	}
	hw.pool_off = pos;
	hw.pool_len = total;
	uint32_t loff[A2D_MIPS] = { 0 }, lsize[A2D_MIPS] = { 0 };
	const size_t pos0 = pos;
	for(int l = 0; l < levels; ++l) {
		size_t n = A2AMD_WAVEPRE + (size_t)w->size[l] + A2AMD_WAVEPOST;
		if(!cap) {
			HIPCHK(c, hipMemcpy(c->d_wavepool.d + pos, w->data[l], n * sizeof(int16_t), hipMemcpyHostToDevice));
			c->wave_h2d_bytes += n * sizeof(int16_t);
		}
		loff[l] = (uint32_t)(pos - pos0);
		lsize[l] = w->size[l];
		hw.dw.size[l] = w->size[l];
		hw.dw.off[l] = (uint32_t)(pos + A2AMD_WAVEPRE);
		pos += n;
	}
	if(cap) {
		// level 0 from the samples the device rendered, pads and mip levels derived here
		if(levels && a2d_launch_wave_from_pcm(cap->d, c->d_wavepool.d + pos0, loff, lsize, levels, (w->flags & 0x100u) != 0,
				A2AMD_WAVEPRE, A2AMD_WAVEPOST, c->stream))
			return c->fail(A2AMD_EHIP, "wave build from capture failed");
		++c->waves_resident;
	} else
		++c->waves_uploaded;
	// Hermite coefficients of every window of the region (a2amd_fast.hip: k_build_coef)
	if(total > 3 && a2d_launch_build_coef(c->d_wavepool.d, c->d_wavecoef.d, (unsigned)hw.pool_off + 1,
			(unsigned)(hw.pool_off + total) - 2, c->stream))
		return c->fail(A2AMD_EHIP, "coefficient build failed");
	c->mwaves[id] = hw.dw;
	c->waves_dirty = true;
	++c->stats.live_waves;
	wave_tap_policy(c);
	return id;
}

int a2amd_wave_drop(a2amd_ctx *c, uint64_t key)
{
	for(size_t i = 0; i < c->waves.size(); ++i)
		if(c->waves[i].live && c->waves[i].key == key) {
			c->waves[i].live = false;
			c->waves[i].dw.size[0] = 0;	// "unloaded", waves.c:717-723
			if(c->waves[i].pool_len)
				c->deferred_wavepool_free.push_back(std::make_pair(c->waves[i].pool_off, c->waves[i].pool_len));
			c->waves[i].pool_len = 0;
			c->waves[i].key = 0;
			c->deferred_wave_slots.push_back((int)i);
			if(!c->nfrags && !c->frag_open) {
				// nothing recorded that could play it: free at once
				for(auto &r : c->deferred_wavepool_free)
					wavepool_release(c, r.first, r.second);
				c->deferred_wavepool_free.clear();
			}
			c->mwaves[i] = c->waves[i].dw;
			c->waves_dirty = true;
			--c->stats.live_waves;
			return A2AMD_OK;
		}
	return c->fail(A2AMD_EINVAL, "wave_drop: unknown key");
}

// ---- fragment clock -----------------------------------------------------------
int a2amd_fragment(a2amd_ctx *c, unsigned frames)
{
	if(!frames || frames > A2D_FRAG)
		return c->fail(A2AMD_EINVAL, "fragment of %u frames", frames);
	if(!c->stack.empty())
		return c->fail(A2AMD_ESTATE, "fragment inside an inline window");
	if(c->uploaded)
		return c->fail(A2AMD_ESTATE, "batch already uploaded; finish the render first");
	close_fragment(c);
	if(c->nfrags >= (int)c->cfg.max_batch)
		return c->fail(A2AMD_ESTATE, "more than max_batch=%u fragments without a render", c->cfg.max_batch);
	if(!c->nfrags)
		c->vm.batch_time = c->walk_time;	// (the device VM's clock, a2amd_vm.cpp)
	c->cur_frag = c->nfrags++;
	c->fragframes[c->cur_frag] = frames;
	c->fragbase[c->cur_frag] = 0;
	c->frag_open = true;
	c->walked_started = 0;
	c->building = -1;
	if(c->defmap_dirty || c->defmap.size() < c->voices.size() + 4096) {
		// (the map only moves here, at a fragment boundary: the host holds its
		// address for the length of a fragment)
		if(c->defmap.size() < c->voices.size() + 4096)
			c->defmap.assign(c->voices.size() * 2 + 65536, 0);
		else
			std::fill(c->defmap.begin(), c->defmap.begin() + std::min(c->defmap.size(), c->voices.size()), 0);
		c->defmap_dirty = false;
	}
	c->defmap_used = false;
	return A2AMD_OK;
}

int a2amd_fragment_offset(a2amd_ctx *c, unsigned offset)
{
	if(!c->frag_open || c->uploaded)
		return c->fail(A2AMD_ESTATE, "fragment_offset outside a fragment");
	if(offset + c->fragframes[c->cur_frag] > A2D_FRAG)
		return c->fail(A2AMD_EINVAL, "fragment_offset %u with %u frames", offset, c->fragframes[c->cur_frag]);
	c->fragbase[c->cur_frag] = (uint8_t)offset;
	return A2AMD_OK;
}

int a2amd_fragment_repeat(a2amd_ctx *c, unsigned frames, unsigned count)
{
	if(c->n_noise || c->n_cutoff_ramps)
		return c->fail(A2AMD_EUNSUPPORTED, "fragment_repeat with %d noise oscillators / %d cutoff ramps "
				"in flight", c->n_noise, c->n_cutoff_ramps);
	if(c->n_clients)
		return c->fail(A2AMD_EUNSUPPORTED, "fragment_repeat with clients on %d xinsert / xsink / xsource "
				"unit(s): their callbacks need every window", c->n_clients);
	for(unsigned i = 0; i < count; ++i) {
		if(int r = a2amd_fragment(c, frames))
			return r;
		// every live voice gets the default window: nothing to record, but
		// tell close_fragment() that nobody was skipped
		c->frag_open = false;
		c->walk_time += frames;
		c->prev_frames = frames;
	}
	return A2AMD_OK;
}

// ---- units ----------------------------------------------------------------------
int a2amd_unit_init(a2amd_ctx *c, uint64_t key, int kind, unsigned flags, int nin, int nout,
		int wired, int transpose, unsigned wakefrac)
{
	if(kind < 0 || kind >= A2AMD_NKINDS)
		return c->fail(A2AMD_EINVAL, "unit kind %d", kind);
	if(nin < 0 || nin > A2D_MAXCH || nout < 0 || nout > A2D_MAXCH)
		return c->fail(A2AMD_EINVAL, "bad channel counts %d->%d", nin, nout);
	const bool add = (flags & A2AMD_PROCADD) != 0;
	switch(kind) {
	  case A2AMD_WTOSC:
		if(nout != 1) return c->fail(A2AMD_EINVAL, "wtosc has 1 output");
		break;
	  case A2AMD_PANMIX:
	  case A2AMD_FBDELAY:
		if(nin < 1 || nin > 2 || nout < 1 || nout > 2)
			return c->fail(A2AMD_EINVAL, "unit kind %d with %d->%d channels", kind, nin, nout);
		break;
	  case A2AMD_FILTER12:
		if(nin != nout || nin < 1 || nin > 2)
			return c->fail(A2AMD_EINVAL, "filter12 %d->%d", nin, nout);
		vm_start_f1tab(c);	// (its cutoff may one day be written by the device VM)
		break;
	  case A2AMD_XINSERT:
		if(nin != nout || nin < 1)
			return c->fail(A2AMD_EINVAL, "xinsert %d->%d", nin, nout);
		break;
	  case A2AMD_XSINK:
		if(nin < 1 || nout != 0)
			return c->fail(A2AMD_EINVAL, "xsink %d->%d", nin, nout);
		break;
	  case A2AMD_XSOURCE:
		if(nin != 0 || nout < 1)
			return c->fail(A2AMD_EINVAL, "xsource %d->%d", nin, nout);
		break;
	  case A2AMD_INLINE:
		if(nout < 1)
			return c->fail(A2AMD_EINVAL, "inline needs outputs");
		break;
	  case A2AMD_DC:	// dc.c:262-281: a source with one or two outputs
		if(nin != 0 || nout < 1 || nout > 2)
			return c->fail(A2AMD_EINVAL, "dc %d->%d", nin, nout);
		break;
	  case A2AMD_WAVESHAPER:
	  case A2AMD_DCBLOCK:
	  case A2AMD_LIMITER:	// A2_MATCHIO, one or two channels
		if(nin != nout || nin < 1 || nin > 2)
			return c->fail(A2AMD_EINVAL, "unit kind %d with %d->%d channels", kind, nin, nout);
		break;
	  default:	// fm1..fm4r, fm.c:532-834: no inputs, one output
		if(nin != 0 || nout != 1)
			return c->fail(A2AMD_EINVAL, "fm unit with %d->%d channels", nin, nout);
		break;
	}
	if(wired && !add && kind != A2AMD_INLINE)
		return c->fail(A2AMD_EUNSUPPORTED, "replacing-mode unit wired to the voice output bus");

	// voice under construction? (a2_PopulateVoice, core.c:350-420)
	int vi = c->building;
	if(vi < 0 || c->voices[vi].key != key) {
		if(!c->free_voices.empty()) {
			vi = c->free_voices.back();
			c->free_voices.pop_back();
		} else {
			vi = (int)c->voices.size();
			c->voices.push_back(HVoice());
		}
		HVoice &v = c->voices[vi];
		v = HVoice();
		v.live = true;
		v.key = key;
		c->building = vi;
		c->voices_dirty = true;
		++c->stats.live_voices;
	}
	HVoice &v = c->voices[vi];
	if(v.nunits >= A2D_MAXCHAIN)
		return c->fail(A2AMD_EUNSUPPORTED, "voice chain longer than %d units", A2D_MAXCHAIN);
	if(kind == A2AMD_INLINE && v.inline_pos >= 0)
		return c->fail(A2AMD_EUNSUPPORTED, "two inline units in one voice");

	int ui;
	if(!c->free_units.empty()) {
		ui = c->free_units.back();
		c->free_units.pop_back();
	} else {
		ui = (int)c->units.size();
		c->units.push_back(HUnit());
		c->mudesc.push_back(0);
	}
	HUnit &u = c->units[ui];
	u = HUnit();
	u.live = true;
	u.kind = kind;
	u.flags = flags;
	u.nin = nin;
	u.nout = nout;
	u.wired = wired ? 1 : 0;
	u.voice = vi;
	u.chainpos = v.nunits;
	v.unit[v.nunits++] = ui;
	v.plain = 0;
	++v.nlive;
	c->mudesc[ui] = A2D_DESC(kind, add ? 1 : 0, nin, nout, wired ? 1 : 0);
	c->udesc_dirty = true;
	c->voices_dirty = true;
	c->dirty_voices.push_back(vi);
	++c->stats.live_units;

	int initval = 0;
	switch(kind) {
	  case A2AMD_WTOSC:	// wtosc_Initialize, wtosc.c:390-423
		initval = transpose + c->cfg.basepitch;
		ramp_init(u.p, initval);
		u.dphase = p2i(c->ptab, u.p.value >> 8);
		u.phase = 0;
		u.p_ramping = 0;
		u.mode = A2D_OSC_OFF;
		u.wave = -1;
		break;
	  case A2AMD_FILTER12:	// f12_Initialize -> f12_CutOff(u, 0, 0, 0), filter12.c:141-147,203
		ramp_init(u.cutoff, 0);
		ramp_set(u.cutoff, transpose, 0, 0);
		initval = f12_coeff(c->ptab, u.cutoff.value, c->cfg.samplerate);
		break;
	  case A2AMD_FBDELAY:	// two zeroed delay lines, fbdelay.c:180-181
		if(!c->fbd_free.empty()) {
			u.fbdbuf = c->fbd_free.back();
			c->fbd_free.pop_back();
		} else
			u.fbdbuf = c->fbd_count++;
		c->fbd_to_zero.push_back(u.fbdbuf);
		initval = u.fbdbuf;
		{	// fbdelay_Initialize, fbdelay.c:183-196: 400 / 280 / 320 ms
			static const int ms[3] = { 400, 280, 320 };
			for(int t = 0; t < 3; ++t)
				u.fbd_taps[t] = (int)((int64_t)(ms[t] << 16) * c->cfg.samplerate / 65536000);
		}
		break;
	  case A2AMD_INLINE:
		v.inline_pos = u.chainpos;
		if(!wired) {
			v.own_nch = nout;
			v.own_off = bus_alloc(c, nout);
		}
		break;
	  default:
		break;
	}
	unsigned initdur = 0, initstart = 0;
	if(kind == A2AMD_DCBLOCK)	// dcb_Initialize, dcblock.c:131-132: cutoff -5.0 (8.18 Hz)
		initval = dcb_coeff(c->ptab, (int)((unsigned)(-5 * 65536) + (unsigned)transpose), c->cfg.samplerate);
	if(kind == A2AMD_LIMITER)	// limiter_Initialize, limiter.c:176-181
		initval = ((64 << 16) << 8) / c->cfg.samplerate;
	if(kind >= A2AMD_FM1 && kind <= A2AMD_FM4R) {	// fm_Initialize, fm.c:338-400
		if(!c->fm_free.empty()) {
			u.fmslot = c->fm_free.back();
			c->fm_free.pop_back();
		} else
			u.fmslot = c->fm_count++;
		initval = transpose + c->cfg.basepitch;
		initdur = (unsigned)u.fmslot;
		initstart = wakefrac & 0xffu;	// vms->waketime & 0xff: sub-sample start time
	}
	push_rec(c, vi, R_INIT, u.chainpos, 0, initval, initdur, initstart);
	return ui;
}

int a2amd_unit_deinit(a2amd_ctx *c, int ui)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "deinit of dead unit %d", ui);
	if(c->voices[c->units[ui].voice].vm >= 0) {
		// a voice the device VM runs is taken down (a2_VoiceFree on it or on a voice above it): what
		// its VM did in this batch up to here goes on record first - the open fragment included when
		// the voice has had its window in it (the host reported it by the map or a hold)
		const int vi = c->units[ui].voice;
		const bool had = c->frag_open && ((c->defmap_used && (size_t)vi < c->defmap.size() && c->defmap[vi]) || is_held(c, vi));
		if(int r = vm_take_back(c, vi, had, nullptr))
			return r;
	}
	HUnit &u = c->units[ui];
	HVoice &v = c->voices[u.voice];
	if(c->building == u.voice)
		c->building = -1;
	if(u.kind == A2AMD_WTOSC && u.mode == A2D_OSC_NOISE)
		--c->n_noise;
	if(u.kind == A2AMD_FILTER12 && u.cutoff.timer)
		--c->n_cutoff_ramps;
	if(u.fbdbuf >= 0)
		c->fbd_deferred_free.push_back(u.fbdbuf);
	if(u.fmslot >= 0)
		c->fm_deferred_free.push_back(u.fmslot);
	if(u.xio >= 0)		// (the slot serves the batch being recorded to its end)
		c->xio_deferred_free.push_back(u.xio);
	if(u.xio_mode)
		--c->n_clients;
	u.live = false;
	c->deferred_free_units.push_back(ui);
	--c->stats.live_units;
	if(--v.nlive == 0) {
		// a2_VoiceFree (core.c:532-591) took the whole chain down
		push_rec(c, u.voice, R_KILL, 0, 0, 0, 0, 0);
		unhold(c, u.voice);
		if(v.started) {
			--c->n_started_live;
			if(c->frag_open && v.walked == c->serial_base + c->cur_frag)
				--c->walked_started;
		}
		v.dying = true;
		v.live = false;
		v.plain = 0;
		if(v.own_off >= 0)
			c->deferred_bus_free.push_back(std::make_pair(v.own_off, v.own_nch));
		c->deferred_free_voices.push_back(u.voice);
		c->lists_dirty = true;
		--c->stats.live_voices;
	}
	return A2AMD_OK;
}

static int shadow_rebuild(a2amd_ctx *c, int ui);

int a2amd_unit_write(a2amd_ctx *c, int ui, int reg, int value, unsigned start, unsigned dur, int transpose)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "write to dead unit %d", ui);
	HUnit &u = c->units[ui];
	if(c->voices[u.voice].vm >= 0)
		return c->fail(A2AMD_ESTATE, "write to unit %d of a voice the device VM runs: a2amd_vm_recall() first", ui);
	c->building = -1;
	start &= 255;		// a2_VoiceControl, core.c:148
	if(dur >= 256 && ((u.kind == A2AMD_WTOSC && (reg == 1 || reg == 2)) || u.kind == A2AMD_PANMIX ||
			(u.kind == A2AMD_FILTER12 && reg == 1))) {
		// a glide: the ramp lasts (start + dur) >> 8 frames from the frame the open window starts at (somewhere in
		// the open fragment), then one more window settles the ramper (a2_PrepareRamper's first branch,
		// a2_dsp.h:131-135; wtosc's extra pitch update, wtosc.c:99-100) - a margin of three fragments covers both
		HVoice &v = c->voices[u.voice];
		const uint64_t until = c->walk_time + 64 + ((start + dur) >> 8) + 192;
		if(until > v.moving_until)
			v.moving_until = until;
		if(!v.listed_moving) {
			v.listed_moving = true;
			c->moving.push_back(u.voice);
		}
	}
	switch(u.kind) {
	  case A2AMD_WTOSC:
		switch(reg) {
		  case 0: {	// wtosc_Wave, wtosc.c:433-483
			int wt = A2AMD_WOFF, id = -1;
			if(value >= 0 && value < (int)c->waves.size() && c->waves[value].live) {
				id = value;
				wt = c->waves[id].dw.type;
				if((wt == A2AMD_WWAVE || wt == A2AMD_WMIPWAVE) &&
						c->waves[id].dw.size[0] > (unsigned)A2D_WTOSC_MAXLENGTH)
					wt = A2AMD_WOFF;
			}
			int nmode = wt == A2AMD_WNOISE ? A2D_OSC_NOISE : wt == A2AMD_WWAVE ? A2D_OSC_WAVE :
					wt == A2AMD_WMIPWAVE ? A2D_OSC_MIPWAVE : A2D_OSC_OFF;
			if(nmode == A2D_OSC_NOISE && u.mode != A2D_OSC_NOISE)
				if(int r = shadow_rebuild(c, ui))	// from here on the host counts its draws
					return r;
			{
				// the wavetable leaf kernels only know mip-mapped waves (and "off"): a
				// voice that moves between the two kinds changes its launch class
				const bool was = leaf_mode(u.mode), is = leaf_mode(nmode);
				if(was != is)
					c->lists_dirty = true;
				if(!was || !is)
					c->voices[u.voice].mode_mix = true;
				c->voices[u.voice].cls_stale = true;
				c->voices[u.voice].plain = 0;
			}
			if(u.mode == A2D_OSC_NOISE && nmode != A2D_OSC_NOISE)
				--c->n_noise;
			if(u.mode != A2D_OSC_NOISE && nmode == A2D_OSC_NOISE)
				++c->n_noise;
			u.mode = nmode;
			u.wave = nmode == A2D_OSC_OFF ? -1 : id;
			value = id;
			break;
		  }
		  case 1:	// wtosc_Pitch, wtosc.c:486-492
			value = value + transpose + c->cfg.basepitch;
			ramp_set(u.p, value, (int)start, (int)dur);
			if(!dur)
				u.p_ramping = 1;
			break;
		  case 2:
			break;
		  case 3:	// wtosc_Phase -> wtosc_set_phase, wtosc.c:369-378
			if(u.wave < 0)
				u.phase = 0;
			else {
				int ph = (int)((unsigned)value + ((start * (u.dphase >> 8)) >> 8));
				u.phase = (uint64_t)(((int64_t)ph * (int64_t)c->waves[u.wave].dw.period) * 256);
			}
			break;
		  default:
			return c->fail(A2AMD_EINVAL, "wtosc register %d", reg);
		}
		break;
	  case A2AMD_PANMIX:
		if(reg < 0 || reg > 1)
			return c->fail(A2AMD_EINVAL, "panmix register %d", reg);
		break;
	  case A2AMD_FILTER12:
		switch(reg) {
		  case 0: {	// f12_CutOff, filter12.c:141-147: stays on the host
			bool was = u.cutoff.timer != 0;
			ramp_set(u.cutoff, value + transpose, (int)start, (int)dur);
			bool is = u.cutoff.timer != 0;
			c->n_cutoff_ramps += (int)is - (int)was;
			c->voices[u.voice].plain = 0;
			if(dur < 256)
				push_rec(c, u.voice, R_F1SET, u.chainpos, 0,
						f12_coeff(c->ptab, u.cutoff.value, c->cfg.samplerate), 0, 0);
			return A2AMD_OK;
		  }
		  case 1:	// f12_Q, filter12.c:149-162
			value = value < 512 ? 32768 : (65536 << 8) / value;
			break;
		  case 2: case 3: case 4:
			break;
		  default:
			return c->fail(A2AMD_EINVAL, "filter12 register %d", reg);
		}
		break;
	  case A2AMD_FBDELAY:
		if(reg < 0 || reg > 6)
			return c->fail(A2AMD_EINVAL, "fbdelay register %d", reg);
		if(reg < 3) {	// fbdelay_FBDelay / LDelay / RDelay, fbdelay.c:231-247
			const int frames = (int)((int64_t)value * c->cfg.samplerate / 65536000);
			// (a tap that stops or starts being at least a fragment long moves the
			// voice between the frame-parallel delay kernel and the general one)
			if(fbd_tap_ok(frames) != fbd_tap_ok(u.fbd_taps[reg])) {
				c->lists_dirty = true;
				c->voices[u.voice].cls_stale = true;
			}
			u.fbd_taps[reg] = frames;
		}
		break;
	  case A2AMD_INLINE:
	  case A2AMD_XINSERT:
	  case A2AMD_XSINK:
	  case A2AMD_XSOURCE:
		return c->fail(A2AMD_EINVAL, "unit kind %d has no registers", u.kind);
	  case A2AMD_DC:
		if(reg < 0 || reg > 1)
			return c->fail(A2AMD_EINVAL, "dc register %d", reg);
		break;
	  case A2AMD_WAVESHAPER:
		if(reg != 0)
			return c->fail(A2AMD_EINVAL, "waveshaper register %d", reg);
		break;
	  case A2AMD_DCBLOCK:	// dcb_CutOff, dcblock.c:112-117: float/libm, host only
		if(reg != 0)
			return c->fail(A2AMD_EINVAL, "dcblock register %d", reg);
		value = dcb_coeff(c->ptab, (int)((unsigned)value + (unsigned)transpose), c->cfg.samplerate);
		break;
	  case A2AMD_LIMITER:	// limiter_Release / limiter_Threshold, limiter.c:201-213
		if(reg == 0)
			value = (int)((unsigned)value << 8) / c->cfg.samplerate;
		else if(reg == 1) {
			unsigned t = (unsigned)value << 8;
			value = (int)(t < 256 ? 256u : t);
		} else
			return c->fail(A2AMD_EINVAL, "limiter register %d", reg);
		break;
	  default: {	// fm.c:403-483: phase | p a fb | p1 a1 fb1 | ...
		static const int nops[8] = { 1, 2, 3, 4, 3, 4, 2, 4 };
		if(reg < 0 || reg > 3 * nops[u.kind - A2AMD_FM1])
			return c->fail(A2AMD_EINVAL, "fm register %d", reg);
		if(reg == 1)	// fm_Pitch: operator 0 is the absolute one
			value = value + transpose + c->cfg.basepitch;
		break;
	  }
	}
	push_rec(c, u.voice, R_WRITE, u.chainpos, reg, value, dur, start);
	return A2AMD_OK;
}


// ---- wtosc phase shadow ------------------------------------------------------------
// The engine-global noise generator is shared by every noise oscillator and by
// the VM's RAND instructions (wtosc.c:136, a2_Noise()), so the host has to hand
// it back advanced by the right number of draws after each Process call of a
// noise oscillator; that count depends on the oscillator's phase.  Scripts do
// switch an oscillator from a wavetable to noise (benchmark/k2epilogue.a2s), so
// the phase is shadowed in wavetable mode too: the closed form of what the
// kernels compute sample by sample.  Pitch and phase only - no audio.
static void shadow_run_pitch(a2amd_ctx *c, HUnit &u, unsigned frames)
{
	ramp_prepare(u.p, (int)frames);		// wtosc_run_pitch, wtosc.c:88-106
	if(u.dphase && (!u.p.timer && !u.p_ramping))
		return;
	unsigned lastv = (unsigned)u.p.value;
	ramp_run(u.p, (int)frames);
	u.p_ramping = u.p.delta;
	u.dphase = p2i(c->ptab, (int)((lastv + (unsigned)u.p.value) >> 9));
}

static void shadow_wave(a2amd_ctx *c, HUnit &u, unsigned frames)
{
	const A2DWave &w = c->waves[u.wave].dw;
	const bool looped = (w.flags & 0x100u) != 0;	// A2_LOOPED
	if(!w.size[0]) {	// wtosc_check_unloaded, wtosc.c:168-183
		u.wave = -1;
		u.mode = A2D_OSC_OFF;
		return;
	}
	shadow_run_pitch(c, u, frames);
	if(u.mode == A2D_OSC_MIPWAVE) {		// wtosc_wavetable, wtosc.c:239-286
		unsigned dph = ((u.dphase + 255) >> 8) * w.period;
		unsigned mm = 0;
		for(; (dph > (A2D_MAXPHINC << 8)) && (mm < A2D_MIPS - 1); ++mm)
			dph >>= 1;
		uint64_t ph = u.phase >> mm;
		dph = (unsigned)(((uint64_t)u.dphase * w.period) >> mm);
		if(looped) {
			const uint64_t m = (uint64_t)w.size[mm] << 24;
			if(!(m & (m - 1)))
				ph &= m - 1;	// built-in waves: power-of-two levels
			else
				ph %= m;
		} else if((ph >> 24) > (uint64_t)(w.size[mm] + 1))
			return;		// all played
		u.phase = (ph + (uint64_t)dph * frames) << mm;
		return;
	}
	// wtosc_wavetable_no_mip, wtosc.c:301-358
	const uint64_t dph = (uint64_t)u.dphase * w.period;
	if(dph >> 32) {
		u.phase += dph * frames;
	} else if(dph > (A2D_MAXPHINC << 16)) {
		// per-sample loop/end test of wtosc_do_fragment, wtosc.c:207-226
		const uint64_t m = (uint64_t)w.size[0] << 24;
		if(looped) {
			u.phase = (u.phase + (uint64_t)(frames - 1) * dph) % m + dph;
		} else if(u.phase < m) {
			uint64_t steps = (m - u.phase + dph - 1) / dph;
			u.phase += (steps < frames ? steps : frames) * dph;
		}
	} else {
		if(looped) {
			unsigned m = w.size[0] << 24;	// 32 bit in the reference, wtosc.c:340
			if(m)
				u.phase %= m;
		} else if((u.phase >> 24) > (uint64_t)(w.size[0] + 1))
			return;
		u.phase += dph * frames;
	}
}

// one window of a wtosc as far as phase and pitch go (any Process variant)
static void shadow_window(a2amd_ctx *c, HUnit &u, unsigned frames)
{
	switch(u.mode) {
	  case A2D_OSC_OFF:	// wtosc_Off, wtosc.c:108-126
		ramp_prepare(u.p, (int)frames);
		ramp_run(u.p, (int)frames);
		break;
	  case A2D_OSC_NOISE:	// wtosc_noise, wtosc.c:129-152
		shadow_run_pitch(c, u, frames);
		u.phase += (uint64_t)frames * u.dphase;
		break;
	  default:
		if(u.wave >= 0)
			shadow_wave(c, u, frames);
		break;
	}
}

// which Process variant wtosc_Wave (wtosc.c:433-483) installs for wave slot 'value'
static int wave_mode(const a2amd_ctx *c, int value, int *id)
{
	int wt = A2AMD_WOFF;
	*id = -1;
	if(value >= 0 && value < (int)c->waves.size() && c->waves[value].live) {
		*id = value;
		wt = c->waves[value].dw.type;
		if((wt == A2AMD_WWAVE || wt == A2AMD_WMIPWAVE) && c->waves[value].dw.size[0] > (unsigned)A2D_WTOSC_MAXLENGTH)
			wt = A2AMD_WOFF;
	}
	return wt == A2AMD_WNOISE ? A2D_OSC_NOISE : wt == A2AMD_WWAVE ? A2D_OSC_WAVE :
			wt == A2AMD_WMIPWAVE ? A2D_OSC_MIPWAVE : A2D_OSC_OFF;
}

// ---- the oscillator state when it is needed on the host ----------------------------------
// The engine-global noise generator is shared by every noise oscillator and the VM's RAND
// instructions, so a window of a noise oscillator has to hand it back advanced by the right
// number of draws (a2amd_unit_process), and that number depends on the oscillator's phase
// and pitch - which, for an oscillator that has been playing a wave, only the device knows
// (scripts do switch oscillators from a wave to noise: benchmark/k2epilogue.a2s).  Round 1
// and the first half of round 2 shadowed every oscillator's phase on the host, window by
// window; that was a fifth of the engine thread's time with scripted voices.  Now nothing is
// shadowed while a wave plays.  When an oscillator is switched to noise, its state is
// rebuilt: the unit state words the device was left with by the last batch (one device to
// host copy of the state array per batch in which that happens), advanced over this batch's
// records of the voice up to now - the same arithmetic the kernels will run on them.
static int shadow_rebuild(a2amd_ctx *c, int ui)
{
	HUnit &u = c->units[ui];
	const int vi = u.voice;
	spell_out_pending(c, vi);
	if(!c->snap_valid) {
		use_device(c);
		const size_t nu = std::min(c->units.size(), c->d_ustate.cap), nv = std::min(c->voices.size(), c->d_vactive.cap);
		c->snap_ustate.assign(nu * A2D_USTATE, 0);
		c->snap_vactive.assign(nv, 0);
		HIPCHK(c, hipStreamSynchronize(c->stream));
		if(nu)
			HIPCHK(c, hipMemcpy(c->snap_ustate.data(), c->d_ustate.d, nu * A2D_USTATE * sizeof(int32_t),
					hipMemcpyDeviceToHost));
		if(nv)
			HIPCHK(c, hipMemcpy(c->snap_vactive.data(), c->d_vactive.d, nv * sizeof(int32_t), hipMemcpyDeviceToHost));
		c->snap_valid = true;
	}
	HUnit t = u;		// (its own copy: mode and wave as they were when the batch began)
	t.mode = A2D_OSC_OFF;
	t.wave = -1;
	t.dphase = 0;
	t.p_ramping = 0;
	t.phase = 0;
	t.p = Ramp{ 0, 0, 0, 0 };
	bool active = false;
	if((size_t)(ui + 1) * A2D_USTATE <= c->snap_ustate.size()) {
		const int32_t *w = &c->snap_ustate[(size_t)ui * A2D_USTATE];
		t.mode = w[OW_MODE];
		t.wave = w[OW_WAVE];
		t.dphase = (unsigned)w[OW_DPHASE];
		t.phase = (uint64_t)(uint32_t)w[OW_PHASE_LO] | ((uint64_t)(uint32_t)w[OW_PHASE_HI] << 32);
		t.p_ramping = w[OW_PRAMPING];
		t.p = Ramp{ w[OW_P], w[OW_P + 1], w[OW_P + 2], w[OW_P + 3] };
		if(t.wave >= (int)c->waves.size())
			t.wave = -1;
	}
	if((size_t)vi < c->snap_vactive.size())
		active = c->snap_vactive[vi] != 0;
	// this batch's fragments up to the open one, the way k_voices executes them
	const HVoice &v = c->voices[vi];
	const int upto = rec_tag(c);
	size_t r = 0;
	for(int f = 0; f <= upto && f < A2D_MAXBATCH; ++f) {
		if(!(r < v.recs.size() && (int)A2D_RFRAG(v.recs[r].head) == f)) {
			// no records: the default window - but not of the open fragment (the voice's
			// turn has not come, or its window so far would have been spelled out above)
			if(active && f < upto)
				shadow_window(c, t, c->fragframes[f]);
			continue;
		}
		for(; r < v.recs.size() && (int)A2D_RFRAG(v.recs[r].head) == f; ++r) {
			const A2DRec &rec = v.recs[r];
			const bool mine = (int)A2D_RUNIT(rec.head) == u.chainpos;
			switch(A2D_ROP(rec.head)) {
			  case R_SEG:
				if(active)
					shadow_window(c, t, rec.dur >> 16);
				break;
			  case R_INIT:
				active = true;
				if(mine) {	// wtosc_Initialize, wtosc.c:390-423
					ramp_init(t.p, rec.value);
					t.dphase = p2i(c->ptab, t.p.value >> 8);
					t.phase = 0;
					t.p_ramping = 0;
					t.mode = A2D_OSC_OFF;
					t.wave = -1;
				}
				break;
			  case R_KILL:
				active = false;
				break;
			  case R_WRITE:
				if(!mine)
					break;
				switch(A2D_RREG(rec.head)) {
				  case 0:
					t.mode = wave_mode(c, rec.value, &t.wave);
					if(t.mode == A2D_OSC_OFF)
						t.wave = -1;
					break;
				  case 1:	// wtosc_Pitch, wtosc.c:486-492 (transpose and base pitch are in the record)
					ramp_set(t.p, rec.value, (int)rec.start, (int)rec.dur);
					if(!rec.dur)
						t.p_ramping = 1;
					break;
				  case 3:	// wtosc_set_phase, wtosc.c:369-378
					if(t.wave < 0)
						t.phase = 0;
					else {
						int ph = (int)((unsigned)rec.value + ((rec.start * (t.dphase >> 8)) >> 8));
						t.phase = (uint64_t)(((int64_t)ph * (int64_t)c->waves[t.wave].dw.period) * 256);
					}
					break;
				}
				break;
			  default:
				break;
			}
		}
	}
	u.dphase = t.dphase;
	u.p_ramping = t.p_ramping;
	u.phase = t.phase;
	u.p = t.p;
	return A2AMD_OK;
}

int a2amd_unit_process(a2amd_ctx *c, int ui, unsigned offset, unsigned frames, uint32_t *noisestate)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "process of dead unit %d", ui);
	if(!c->frag_open || !frames || offset + frames > c->fragframes[c->cur_frag])
		return c->fail(A2AMD_ESTATE, "process [%u,+%u) outside the open fragment", offset, frames);
	HUnit &u = c->units[ui];
	const int vi = u.voice;
	HVoice &v = c->voices[vi];
	if(v.vm >= 0)
		return c->fail(A2AMD_ESTATE, "process call on unit %d of a voice the device VM runs: a2amd_vm_recall() first", ui);
	c->building = -1;
	unhold(c, vi);		// (a call speaks for itself)
	if(!v.resolved) {
		resolve_out(c, v);
		v.plain = 0;
	}
	if(!v.started) {
		v.started = true;
		v.plain = 0;
		++c->n_started_live;
	}
	touch(c, vi);
	if(v.walked != c->serial_base + c->cur_frag) {
		v.walked = c->serial_base + c->cur_frag;
		++c->walked_started;
	}
	if(u.chainpos == 0) {
		v.win_off = (int)offset;
		v.win_frames = (int)frames;
	} else if(v.win_off != (int)offset || v.win_frames != (int)frames)
		return c->fail(A2AMD_ESTATE, "units of one voice processed over different windows");

	switch(u.kind) {
	  case A2AMD_WTOSC:
		if(u.mode == A2D_OSC_NOISE) {
			// Count the draws wtosc_noise (wtosc.c:129-152) makes over this
			// window and hand the engine-global RNG back advanced, so VM
			// RAND instructions and other voices interleave as they do in
			// the reference; the kernel regenerates the same draws from
			// the seed recorded here.
			if(!noisestate)
				return c->fail(A2AMD_EINVAL, "noise oscillator needs the engine's noise state");
			push_rec(c, vi, R_NOISESEED, u.chainpos, 0, (int)*noisestate, 0, 0);
			shadow_run_pitch(c, u, frames);
			uint64_t end = u.phase + (uint64_t)frames * u.dphase;
			uint64_t draws = u.dphase >= (1u << 23) ? frames : (end >> 23) - (u.phase >> 23);
			static const bool trace = getenv("A2AMD_DEBUG_NOISE") != nullptr;	// (debugging aid: the test
			if(trace)								// suite's checker prints the same line)
				fprintf(stderr, "NOISE phase %llx dphase %x frames %u\n", (unsigned long long)u.phase, u.dphase, frames);
			uint32_t st = *noisestate;
			for(uint64_t i = 0; i < draws; ++i)
				st = st * 1566083941u + 1u;
			*noisestate = st;
			u.phase = end;
		}
		// (any other mode: the device does it all, and keeps the state - shadow_rebuild)
		break;
	  case A2AMD_FILTER12: {	// the head of f12_process, filter12.c:86-96
		bool was = u.cutoff.timer != 0;
		const bool moving = cutoff_moving(u);
		ramp_prepare(u.cutoff, (int)frames);
		if(u.cutoff.delta) {
			ramp_run(u.cutoff, (int)frames);
			push_rec(c, vi, R_F1RAMP, u.chainpos, 0,
					f12_coeff(c->ptab, u.cutoff.value, c->cfg.samplerate), 0, 0);
		}
		c->n_cutoff_ramps += (int)(u.cutoff.timer != 0) - (int)was;
		if(moving && !cutoff_moving(u))
			v.plain = 0;	// (the ramp has arrived and the ramper has snapped to its target: the voice may be plain again)
		break;
	  }
	  case A2AMD_INLINE:
		c->stack.push_back(ui);
		break;
	  default:
		break;
	}
	v.win_done = u.chainpos + 1;
	if(u.chainpos == v.nunits - 1) {
		// The common case at scale - the voice's VM slept through the fragment,
		// the engine made one full-window Process call per unit, nothing else
		// happened to the voice - costs no record: the kernels apply that
		// default to every voice without records (push_rec spells it out if
		// something does follow in this fragment).
		if(offset == 0 && frames == c->fragframes[c->cur_frag] && v.recs.size() == v.frag_mark &&
				v.deferred.empty()) {
			v.default_seg = c->serial_base + c->cur_frag;
			v.win_done = 0;
		} else
			push_rec(c, vi, R_SEG, 0, 0, 0, offset | (frames << 16), 0);
	}
	return A2AMD_OK;
}

namespace {
// is every unit of the voice one whose Process leaves nothing to do on the host
// but - for a wtosc - the phase shadow?
void classify_plain(a2amd_ctx *c, HVoice &v)
{
	v.plain = 2;
	if(!v.live || v.dying || !v.resolved || !v.started || v.inline_pos >= 0)
		return;
	for(int k = 0; k < v.nunits; ++k) {
		const HUnit &u = c->units[v.unit[k]];
		switch(u.kind) {
		  case A2AMD_WTOSC:
			if(u.mode == A2D_OSC_NOISE)
				return;
			break;
		  case A2AMD_FILTER12:
			if(cutoff_moving(u))
				return;
			break;
		  case A2AMD_INLINE: case A2AMD_XINSERT: case A2AMD_XSINK: case A2AMD_XSOURCE:
			return;
		  default:
			break;
		}
	}
	v.plain = 1;
}
} // namespace

int a2amd_voice_process(a2amd_ctx *c, int head, unsigned offset, unsigned frames, uint32_t *noisestate)
{
	if(head < 0 || head >= (int)c->units.size() || !c->units[head].live)
		return c->fail(A2AMD_EINVAL, "process of dead unit %d", head);
	const int vi = c->units[head].voice;
	HVoice &pv = c->voices[vi];
	if(pv.vm >= 0)
		return c->fail(A2AMD_ESTATE, "process call on a voice the device VM runs: a2amd_vm_recall() first");
	unhold(c, vi);
	if(!pv.plain)
		classify_plain(c, pv);
	static const int dbgw = getenv("A2AMD_DBG_WALK") ? atoi(getenv("A2AMD_DBG_WALK")) : 0;
	if(pv.plain == 1 && !(dbgw & 1)) {
		// The short path: one pass of bookkeeping for the whole chain (what
		// a2amd_unit_process does unit by unit) and the window - unrecorded if it is
		// the default one.
		if(!c->frag_open || !frames || offset + frames > c->fragframes[c->cur_frag])
			return c->fail(A2AMD_ESTATE, "process [%u,+%u) outside the open fragment", offset, frames);
		const long long serial = c->serial_base + c->cur_frag;
		c->building = -1;
		if(pv.touched != serial) {
			pv.touched = serial;
			pv.frag_mark = pv.recs.size();
		}
		if(pv.walked != serial) {
			pv.walked = serial;
			++c->walked_started;
		}
		pv.win_off = (int)offset;
		pv.win_frames = (int)frames;
		if(offset == 0 && frames == c->fragframes[c->cur_frag] && pv.recs.size() == pv.frag_mark &&
				pv.deferred.empty()) {
			pv.default_seg = serial;
			pv.win_done = 0;
		} else
			push_rec(c, vi, R_SEG, 0, 0, 0, offset | (frames << 16), 0);
		return (size_t)vi < c->defmap.size() && !(dbgw & 2) ? 1 : 0;
	}
	const int n = c->voices[vi].nunits;
	for(int k = 0; k < n; ++k) {
		const int ui = c->voices[vi].unit[k];
		if(c->units[ui].kind == A2AMD_INLINE)
			return c->fail(A2AMD_EINVAL, "voice_process on a voice with an inline unit");
		if(int r = a2amd_unit_process(c, ui, offset, frames, noisestate))
			return r;
	}
	// may the host mark this voice in the default map from the next fragment on?
	const HVoice &v = c->voices[vi];
	if((size_t)vi >= c->defmap.size())
		return 0;
	for(int k = 0; k < n; ++k) {
		const HUnit &u = c->units[v.unit[k]];
		if((u.kind == A2AMD_WTOSC && u.mode == A2D_OSC_NOISE) || u.xio_mode ||
				(u.kind == A2AMD_FILTER12 && cutoff_moving(u)))
			return 0;
	}
	return (dbgw & 2) ? 0 : 1;
}

int a2amd_voice_markable(a2amd_ctx *c, int ui)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "markable: dead unit %d", ui);
	const int vi = c->units[ui].voice;
	const HVoice &v = c->voices[vi];
	// (walked at least once: its output bus is resolved, its place in the launch order known)
	if(!v.live || v.dying || !v.resolved || !v.started || (size_t)vi >= c->defmap.size())
		return 0;
	for(int k = 0; k < v.nunits; ++k) {
		const HUnit &u = c->units[v.unit[k]];
		if((u.kind == A2AMD_WTOSC && u.mode == A2D_OSC_NOISE) || u.xio_mode ||
				(u.kind == A2AMD_FILTER12 && cutoff_moving(u)))
			return 0;
	}
	return 1;
}

int a2amd_voice_slot(a2amd_ctx *c, int ui)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "slot of dead unit %d", ui);
	return c->units[ui].voice;
}

int a2amd_default_hold(a2amd_ctx *c, const uint32_t *slots, unsigned n, unsigned lo, int on)
{
	if(on && c->held.size() < c->voices.size())
		c->held.resize(c->voices.size() * 2 + 65536, 0);
	for(unsigned k = 0; k < n; ++k) {
		const size_t vi = slots ? slots[k] : (size_t)lo + k;
		if(!on) {
			unhold(c, (int)vi);
			continue;
		}
		if(vi >= c->voices.size())
			return c->fail(A2AMD_EINVAL, "default_hold: slot %zu", vi);
		const HVoice &v = c->voices[vi];
		if(!v.live || v.dying || !v.resolved || !v.started)
			return c->fail(A2AMD_ESTATE, "default_hold: voice %zu has not been processed yet", vi);
		if(!c->held[vi]) {
			c->held[vi] = 1;
			++c->n_held;
		}
	}
	return A2AMD_OK;
}

int a2amd_default_release_all(a2amd_ctx *c)
{
	if(c->n_held)
		std::fill(c->held.begin(), c->held.end(), 0);
	c->n_held = 0;
	return A2AMD_OK;
}

uint8_t *a2amd_default_map(a2amd_ctx *c, unsigned *nslots)
{
	if(!c->frag_open)
		return nullptr;
	c->defmap_used = c->defmap_dirty = true;
	if(nslots)
		*nslots = (unsigned)c->defmap.size();
	return c->defmap.data();
}

int a2amd_unit_clients(a2amd_ctx *c, int ui, unsigned mode)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live ||
			(c->units[ui].kind != A2AMD_XINSERT && c->units[ui].kind != A2AMD_XSINK &&
			 c->units[ui].kind != A2AMD_XSOURCE))
		return c->fail(A2AMD_EINVAL, "unit %d is not a live xinsert / xsink / xsource", ui);
	if((mode & ~(unsigned)(A2AMD_XIO_TAP | A2AMD_XIO_INJECT | A2AMD_XIO_MUTE)) ||
			((mode & A2AMD_XIO_MUTE) && (!(mode & A2AMD_XIO_TAP) || a2amd_unit_insertable(c, ui) < 1)) ||
			(c->units[ui].kind == A2AMD_XSINK && (mode & A2AMD_XIO_INJECT)) ||
			(c->units[ui].kind == A2AMD_XSOURCE && (mode & A2AMD_XIO_TAP)))
		return c->fail(A2AMD_EINVAL, "client mode %#x on unit kind %d", mode, c->units[ui].kind);
	HUnit &u = c->units[ui];
	if(mode == u.xio_mode)
		return A2AMD_OK;
	if(mode && u.xio < 0) {
		if(!c->xio_free.empty()) {
			u.xio = c->xio_free.back();
			c->xio_free.pop_back();
		} else {
			u.xio = (int)c->xio.size();
			c->xio.emplace_back();
		}
		XioSlot &x = c->xio[u.xio];
		x.unit = x.last_unit = ui;
		x.tap.assign(A2D_XIO_HALF, 0);
		x.inj.assign(A2D_XIO_HALF, 0);
		x.inj_used = x.tapped = false;
	}
	if(mode & A2AMD_XIO_TAP)
		c->xio[u.xio].tapped = true;
	// (the device reads slot and mode from the unit's state words: two writes,
	// in order with the voice's windows)
	c->building = -1;
	push_rec(c, u.voice, R_WRITE, u.chainpos, 0, mode ? u.xio + 1 : 0, 0, 0);
	push_rec(c, u.voice, R_WRITE, u.chainpos, 1, (int)mode, 0, 0);
	c->n_clients += (int)(mode != 0) - (int)(u.xio_mode != 0);
	u.xio_mode = mode;
	c->lists_dirty = true;		// a driver chain with clients is served by the general kernel
	c->voices[u.voice].cls_stale = true;
	return A2AMD_OK;
}

int a2amd_unit_inject(a2amd_ctx *c, int ui, unsigned offset, unsigned frames, const int32_t *const *bufs)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live || !(c->units[ui].xio_mode & A2AMD_XIO_INJECT))
		return c->fail(A2AMD_EINVAL, "unit %d takes no client output", ui);
	if(!c->frag_open || !frames || offset + frames > c->fragframes[c->cur_frag])
		return c->fail(A2AMD_ESTATE, "inject [%u,+%u) outside the open fragment", offset, frames);
	HUnit &u = c->units[ui];
	XioSlot &x = c->xio[u.xio];
	for(int ch = 0; ch < u.nout; ++ch) {
		int32_t *d = x.inj.data() + ((size_t)c->cur_frag * A2AMD_MAXCHANNELS + ch) * A2D_FRAG + offset;
		for(unsigned k = 0; k < frames; ++k)
			d[k] = wadd(d[k], bufs[ch][k]);
	}
	x.inj_used = true;
	return A2AMD_OK;
}

int a2amd_unit_insertable(a2amd_ctx *c, int ui)
{
	if(ui < 0 || ui >= (int)c->units.size() || !c->units[ui].live)
		return c->fail(A2AMD_EINVAL, "insertable: dead unit %d", ui);
	const HUnit &u = c->units[ui];
	const HVoice &v = c->voices[u.voice];
	// The render goes nesting depth by nesting depth, deepest first, and can pause between two
	// of them: a unit that is the last of its voice and adds into the voice's output bus (the
	// parent's inline bus) can have its output replaced there before the parent's chain runs.
	// (depth and bus are known once the voice has been processed for the first time: resolve_out)
	return u.kind == A2AMD_XINSERT && v.resolved && v.depth >= 1 && v.out_off != 0 && u.chainpos == v.nunits - 1 &&
			u.wired && (u.flags & A2AMD_PROCADD) && !c->comm ? v.depth : 0;
}

int a2amd_render_paused(a2amd_ctx *c)
{
	return c->paused_at;
}

int a2amd_unit_insert(a2amd_ctx *c, int ui, unsigned fragment, unsigned offset, unsigned frames,
		const int32_t *const *bufs)
{
	if(ui < 0 || ui >= (int)c->units.size() || c->units[ui].xio < 0 || c->xio[c->units[ui].xio].last_unit != ui)
		return c->fail(A2AMD_EINVAL, "unit %d has had no clients", ui);
	if(!c->uploaded || (int)fragment >= c->nfrags || !frames || offset + frames > c->fragframes[fragment])
		return c->fail(A2AMD_ESTATE, "insert [%u,+%u) of fragment %u: not between the SUBTREES and the ROOT phase "
				"of a batch that has it", offset, frames, fragment);
	const HUnit &u = c->units[ui];
	XioSlot &x = c->xio[u.xio];
	if(x.late.empty())
		x.late.assign(A2D_XIO_HALF, 0);
	for(int ch = 0; ch < u.nin; ++ch) {
		int32_t *d = x.late.data() + ((size_t)fragment * A2AMD_MAXCHANNELS + ch) * A2D_FRAG + offset;
		for(unsigned k = 0; k < frames; ++k)
			d[k] = wadd(d[k], bufs[ch][k]);
	}
	x.late_used = true;
	return A2AMD_OK;
}

int a2amd_unit_tapped(a2amd_ctx *c, int ui, unsigned fragment, const int32_t **bufs)
{
	// (also for a unit that was deinitialised in the course of that batch)
	if(ui < 0 || ui >= (int)c->units.size() || c->units[ui].xio < 0 || c->xio[c->units[ui].xio].last_unit != ui)
		return c->fail(A2AMD_EINVAL, "unit %d has had no clients", ui);
	if(fragment >= A2D_MAXBATCH)
		return c->fail(A2AMD_EINVAL, "fragment %u", fragment);
	const HUnit &u = c->units[ui];
	for(int ch = 0; ch < u.nin; ++ch)
		bufs[ch] = c->xio[u.xio].tap.data() + ((size_t)fragment * A2AMD_MAXCHANNELS + ch) * A2D_FRAG;
	return u.nin;
}

int a2amd_inline_end(a2amd_ctx *c, int ui)
{
	if(c->stack.empty() || c->stack.back() != ui)
		return c->fail(A2AMD_ESTATE, "inline_end(%d) does not match the open window", ui);
	c->stack.pop_back();
	return A2AMD_OK;
}

} // extern "C"
