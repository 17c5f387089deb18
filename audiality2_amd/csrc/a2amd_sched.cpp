// a2amd_sched.cpp - what a recorded batch becomes on the device: the records and lists that go
// up (upload), the launch order - leaf voices first, bus owners by nesting depth, deepest first -
// the launch shapes, hipGraphs of quiet batches, and the end of a batch.  (Split out of
// a2amd_host.cpp in round 3; the design notes are at the top of that file.)
#include "a2amd_host.h"
#include <functional>

namespace a2h {




void touch(a2amd_ctx *c, int vi)
{
	HVoice &v = c->voices[vi];
	const long long serial = c->serial_base + rec_tag(c);
	if(v.touched != serial) {
		v.touched = serial;
		v.frag_mark = v.recs.size();
	}
}



// The windows of the open fragment that were left unrecorded so far - the voice's
// default window, noted in HVoice::default_seg or by the host in the default map -
// become records: something else follows in the same fragment after all.
void spell_out_pending(a2amd_ctx *c, int vi)
{
	HVoice &dv = c->voices[vi];
	const bool marked = c->frag_open && c->defmap_used && (size_t)vi < c->defmap.size() && c->defmap[vi];
	if(marked || (c->frag_open && is_held(c, vi))) {
		// (a held voice that gets a record after all - its chain taken down from outside the
		// walk, say - had its default window in this fragment like a marked one; the hold ends)
		if(marked)
			c->defmap[vi] = 0;
		unhold(c, vi);
		dv.default_seg = c->serial_base + c->cur_frag;
		if(dv.walked != c->serial_base + c->cur_frag) {
			dv.walked = c->serial_base + c->cur_frag;
			++c->walked_started;
		}
	}
	if(dv.default_seg == c->serial_base + rec_tag(c)) {
		dv.default_seg = -1;
		A2DRec seg = { A2D_HEAD(rec_tag(c), R_SEG, 0, 0), 0, (unsigned)c->fragframes[rec_tag(c)] << 16, 0 };
		if(!dv.listed_recs) {
			dv.listed_recs = true;
			c->with_recs.push_back(vi);
		}
		dv.recs.push_back(seg);
	}
}

void push_rec(a2amd_ctx *c, int vi, int op, int unit, int reg, int value, unsigned dur, unsigned start)
{
	touch(c, vi);
	spell_out_pending(c, vi);
	A2DRec r;
	r.head = A2D_HEAD(rec_tag(c), op, unit, reg);
	r.value = value;
	r.dur = dur;
	r.start = start;
	HVoice &v = c->voices[vi];
	if(!v.listed_recs) {
		v.listed_recs = true;
		c->with_recs.push_back(vi);
	}
	// A write that reaches a unit which has already rendered the window the
	// voice is in (a control wire from a unit further down the chain, e.g. the
	// engine's env, env.c:135) takes effect after that window: the device
	// executes a window as one SEG record, so the write waits behind it.
	if(op != R_SEG && op != R_INIT && op != R_KILL && unit < v.win_done) {
		v.deferred.push_back(r);
		return;
	}
	v.recs.push_back(r);
	if(op == R_SEG) {
		v.win_done = 0;
		for(A2DRec &d : v.deferred)
			v.recs.push_back(d);
		v.deferred.clear();
	}
}

int bus_alloc(a2amd_ctx *c, int nch)
{
	auto &fl = c->bus_free[nch];
	if(!fl.empty()) {
		int off = fl.back();
		fl.pop_back();
		return off;
	}
	size_t off = c->bus_used;
	c->bus_used += c->bus_stride_frames * (size_t)nch;
	return (int)off;
}

long long now_serial(const a2amd_ctx *c) { return c->serial_base + rec_tag(c); }

// the engine-visible walk found no work for the VMs: close the fragment
int close_fragment(a2amd_ctx *c)
{
	if(!c->frag_open)
		return 0;
	const int f = c->cur_frag;
	const unsigned nframes = c->fragframes[f];
	if(c->defmap_used) {
		size_t n = 0;
		const size_t nv = std::min(c->defmap.size(), c->voices.size());
		for(size_t k = 0; k < nv; ++k)
			n += c->defmap[k];
		c->walked_started += (int)n;
	}
	c->walked_started += (int)c->n_held;
	if(c->walked_started != c->n_started_live) {
		if(c->hosttiming)
			dbg_counters()[2] += 1;
		// a live voice got no Process call this fragment: say so, or
		// the kernel would apply the default
		for(size_t vi = 0; vi < c->voices.size(); ++vi) {
			HVoice &v = c->voices[vi];
			if((c->defmap_used && vi < c->defmap.size() && c->defmap[vi]) || is_held(c, (int)vi))
				continue;	// (walked: the host marked its default window)
			if(v.live && v.started && !v.dying && v.walked != c->serial_base + f &&
					v.touched != c->serial_base + f) {
				if(v.vm >= 0)
					c->vm.unwalked = (int)vi;	// (its VM cannot be told to skip a fragment: a2amd_render fails)
				A2DRec r = { A2D_HEAD(f, R_NOP, 0, 0), 0, 0, 0 };
				if(!v.listed_recs) {
					v.listed_recs = true;
					c->with_recs.push_back((int)vi);
				}
				v.recs.push_back(r);
				v.touched = c->serial_base + f;
				if(c->hosttiming)
					dbg_counters()[3] += 1;
			}
		}
	}
	c->frag_open = false;
	c->walk_time += nframes;
	c->prev_frames = nframes;
	c->defmap_used = false;
	return 0;
}

void resolve_out(a2amd_ctx *c, HVoice &v)
{
	if(v.resolved)
		return;
	v.depth = (int)c->stack.size();
	if(c->stack.empty()) {
		v.out_off = 0;
		v.out_nch = c->cfg.channels;
	} else {
		const HUnit &il = c->units[c->stack.back()];
		const HVoice &pv = c->voices[il.voice];
		if(il.wired) {
			v.out_off = pv.out_off;
			v.out_nch = pv.out_nch;
		} else {
			v.out_off = pv.own_off;
			v.out_nch = pv.own_nch;
		}
	}
	v.resolved = true;
	v.cls_stale = true;
	c->voices_dirty = true;
	c->dirty_voices.push_back((int)(&v - c->voices.data()));
	c->lists_dirty = true;
}

void sync_voice_mirror(a2amd_ctx *c, int vi)
{
	const HVoice &v = c->voices[vi];
	if(c->mvoices.size() <= (size_t)vi)
		c->mvoices.resize(vi + 1);
	A2DVoice &m = c->mvoices[vi];
	memset(&m, 0, sizeof(m));
	m.nunits = v.nunits;
	for(int i = 0; i < v.nunits && i < A2D_CHAIN_INLINE; ++i)
		m.unit[i] = v.unit[i];
	if(v.nunits > A2D_CHAIN_INLINE)
		c->long_chains = true;
	if(c->long_chains) {
		if(c->mvext.size() <= (size_t)vi)
			c->mvext.resize(vi + 1);
		A2DVoiceExt &e = c->mvext[vi];
		memset(&e, 0, sizeof(e));
		for(int i = A2D_CHAIN_INLINE; i < v.nunits; ++i)
			e.unit[i - A2D_CHAIN_INLINE] = v.unit[i];
	}
	m.out_off = v.out_off;
	m.out_nch = v.out_nch;
	m.own_off = v.own_off;
	m.own_nch = v.own_nch;
}

// what the wavetable leaf kernels play: mip-mapped waves, nothing, and - k_leaf_recs only, but a noise
// oscillator's every window carries an R_NOISESEED record, so the quiet kernels never see one - noise

// wtosc (mip-mapped wave playing) -> panmix 1->2 adding into the output bus
bool is_oscpan_chain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits != 2 || v.out_nch < 2)
		return false;
	const HUnit &o = c->units[v.unit[0]], &pm = c->units[v.unit[1]];
	return o.kind == A2AMD_WTOSC && !(o.flags & A2AMD_PROCADD) && !o.wired &&
			leaf_mode(o.mode) &&
			pm.kind == A2AMD_PANMIX && pm.nin == 1 && pm.nout == 2 && pm.wired &&
			(pm.flags & A2AMD_PROCADD);
}

// wtosc (replacing) + wtosc (adding) -> panmix 1->2 adding into the output bus
bool is_osc2pan_chain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits != 3 || v.out_nch < 2)
		return false;
	const HUnit &a = c->units[v.unit[0]], &b = c->units[v.unit[1]], &pm = c->units[v.unit[2]];
	return a.kind == A2AMD_WTOSC && !(a.flags & A2AMD_PROCADD) && !a.wired &&
			leaf_mode(a.mode) &&
			b.kind == A2AMD_WTOSC && (b.flags & A2AMD_PROCADD) && !b.wired &&
			leaf_mode(b.mode) &&
			pm.kind == A2AMD_PANMIX && pm.nin == 1 && pm.nout == 2 && pm.wired &&
			(pm.flags & A2AMD_PROCADD);
}

// wtosc -> filter12 (1 channel, replacing) -> panmix 1->2 adding into the output bus
bool is_oscfiltpan_chain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits != 3 || v.out_nch < 2)
		return false;
	const HUnit &o = c->units[v.unit[0]], &f = c->units[v.unit[1]], &pm = c->units[v.unit[2]];
	return o.kind == A2AMD_WTOSC && !(o.flags & A2AMD_PROCADD) && !o.wired &&
			leaf_mode(o.mode) &&
			f.kind == A2AMD_FILTER12 && f.nin == 1 && !f.wired && !(f.flags & A2AMD_PROCADD) &&
			// (a ramping cutoff means an R_F1RAMP record per window: such a voice is never
			// without records, and the quiet kernel skips voices with records)
			pm.kind == A2AMD_PANMIX && pm.nin == 1 && pm.nout == 2 && pm.wired &&
			(pm.flags & A2AMD_PROCADD);
}

// wtosc (replacing) + wtosc (adding) -> filter12 (1 channel, replacing) -> panmix 1->2 adding into the
// output bus: the usual subtractive-synth note
bool is_osc2filtpan_chain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits != 4 || v.out_nch < 2)
		return false;
	const HUnit &a = c->units[v.unit[0]], &b = c->units[v.unit[1]], &f = c->units[v.unit[2]], &pm = c->units[v.unit[3]];
	return a.kind == A2AMD_WTOSC && !(a.flags & A2AMD_PROCADD) && !a.wired &&
			leaf_mode(a.mode) &&
			b.kind == A2AMD_WTOSC && (b.flags & A2AMD_PROCADD) && !b.wired &&
			leaf_mode(b.mode) &&
			f.kind == A2AMD_FILTER12 && f.nin == 1 && !f.wired && !(f.flags & A2AMD_PROCADD) &&
			pm.kind == A2AMD_PANMIX && pm.nin == 1 && pm.nout == 2 && pm.wired &&
			(pm.flags & A2AMD_PROCADD);
}

// fmN -> panmix 1->2 adding into the output bus
bool is_fmpan_chain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits != 2 || v.out_nch < 2)
		return false;
	const HUnit &o = c->units[v.unit[0]], &pm = c->units[v.unit[1]];
	return o.kind >= A2AMD_FM1 && o.kind <= A2AMD_FM4R && !(o.flags & A2AMD_PROCADD) && !o.wired &&
			pm.kind == A2AMD_PANMIX && pm.nin == 1 && pm.nout == 2 && pm.wired &&
			(pm.flags & A2AMD_PROCADD);
}

// inline 0 2; panmix 2 2; xinsert 2 >  (a2_rootdriver / a2_groupdriver)
bool is_driver_chain(const a2amd_ctx *c, const HVoice &v)
{
	if((v.nunits != 3 && v.nunits != 2) || v.own_nch != 2 || v.out_nch < 2 || v.own_off < 0)
		return false;
	const HUnit &il = c->units[v.unit[0]], &pm = c->units[v.unit[1]];
	if(!(il.kind == A2AMD_INLINE && !(il.flags & A2AMD_PROCADD) && !il.wired && il.nout == 2 &&
			pm.kind == A2AMD_PANMIX && pm.nin == 2 && pm.nout == 2))
		return false;
	if(v.nunits == 2)
		// inline 0 2; panmix 2 >  - what the drop-in's root voice looks like from here (the root's
		// xinsert stays the engine's): the panmix itself adds into the output bus
		return pm.wired && (pm.flags & A2AMD_PROCADD);
	const HUnit &xi = c->units[v.unit[2]];
	return !pm.wired && !(pm.flags & A2AMD_PROCADD) &&
			xi.kind == A2AMD_XINSERT && xi.nin == 2 && xi.wired && (xi.flags & A2AMD_PROCADD) &&
			!xi.xio_mode;	// (clients: the general kernel serves them)
}

// a tap the frame-parallel delay kernel can take: at least one fragment long, and
// short enough not to wrap onto the frames being written

// inline 0 2; fbdelay 2 2 [; fbdelay 2 2 ...]; the last one wired and adding
// (the group voices of benchmark/fmtest4.a2s:83-95), every tap >= one fragment
bool is_fbdchain(const a2amd_ctx *c, const HVoice &v)
{
	if(v.nunits < 2 || v.nunits > 5 || v.own_nch != 2 || v.out_nch < 2 || v.own_off < 0 || v.inline_pos != 0)
		return false;
	const HUnit &il = c->units[v.unit[0]];
	if(il.kind != A2AMD_INLINE || (il.flags & A2AMD_PROCADD) || il.wired || il.nout != 2)
		return false;
	for(int k = 1; k < v.nunits; ++k) {
		const HUnit &d = c->units[v.unit[k]];
		const bool last = k == v.nunits - 1;
		if(d.kind != A2AMD_FBDELAY || d.nin != 2 || d.nout != 2 || (d.wired != 0) != last ||
				(last && !(d.flags & A2AMD_PROCADD)))
			return false;
		for(int t = 0; t < 3; ++t)
			if(!fbd_tap_ok(d.fbd_taps[t]))
				return false;
	}
	return true;
}

int upload(a2amd_ctx *c)
{
	// (voices the device VM took over during this batch: the host's interpreter carries them to its
	// end - records into their lists - and their states go up, a2amd_vm.cpp)
	if(int r = vm_prepare_batch(c))
		return r;
	if(c->hosttiming) {
		// (A2AMD_HOSTTIMING: why a batch did not take the quiet path - first reason that applies)
		const int why = !c->blob_quiet ? 0 : !c->with_recs.empty() ? 1 : !c->prev_with_recs.empty() ? 2 :
				c->voices_dirty ? 3 : c->udesc_dirty ? 4 : c->waves_dirty ? 5 : c->lists_dirty ? 6 : c->ptab_dirty ? 7 :
				!c->dirty_voices.empty() ? 8 : !c->fbd_to_zero.empty() ? 9 : c->nfrags != c->blob_nfrags ? 10 :
				c->bus_used > c->d_busmem.cap ? 11 :
				memcmp(c->fragframes, c->blob_frames, (size_t)c->nfrags * sizeof(unsigned)) ? 10 : 12;
		dbg_why()[why] += 1;
	}
	if(c->blob_quiet && c->with_recs.empty() && c->prev_with_recs.empty() && c->moving.empty() && !c->voices_dirty &&
			!c->udesc_dirty && !c->waves_dirty && !c->lists_dirty && !c->ptab_dirty && !c->vm.list_dirty &&
			c->dirty_voices.empty() && c->fbd_to_zero.empty() && c->bus_used <= c->d_busmem.cap) {
		bool inject = false;
		for(const XioSlot &x : c->xio)
			if(x.unit >= 0 && (x.inj_used || (c->units[x.unit].xio_mode & A2AMD_XIO_INJECT)))
				inject = true;
		const bool same = c->nfrags == c->blob_nfrags &&
				!memcmp(c->fragframes, c->blob_frames, (size_t)c->nfrags * sizeof(unsigned));
		if(!inject && same) {
			// the same quiet batch again: the device has it all (graphs stay valid)
			if(c->hosttiming)
				dbg_counters()[0] += 1;
			++c->quiet_streak;
			c->uploaded = true;
			return 0;
		}
		if(!inject && c->d_params && c->stream) {
			// A quiet batch again, cut into fragments differently - the engine's root voice woke
			// up in the middle of a fragment, which it does every 3 906 frames while it idles at
			// 'end' (core.c:1195), i.e. once per a2play buffer: nothing per voice has changed, only
			// the fragment table of the parameter block.  That block alone goes up again (a few
			// hundred bytes instead of a pass over every voice); graphs read it on the device and
			// stay valid while the NUMBER of fragments - their launch shapes - is the same.
			A2DParams p = c->hparams;
			p.nfrags = c->nfrags;
			memset(p.fragframes, 0, sizeof(p.fragframes));
			memset(p.fragstart, 0, sizeof(p.fragstart));
			for(int f = 0, acc = 0; f < c->nfrags; ++f) {
				p.fragframes[f] = (uint8_t)c->fragframes[f];
				p.fragstart[f] = (uint16_t)acc;
				acc += (int)c->fragframes[f];
			}
			const int bi = c->blob_i;
			c->blob_i ^= 1;
			if(c->blob_busy[bi]) {
				HIPCHK(c, hipEventSynchronize(c->blob_ev[bi]));
				c->blob_busy[bi] = false;
			}
			if(c->h_blob[bi] && c->h_blob_cap[bi] >= sizeof(A2DParams) && c->blob_ev[bi]) {
				memcpy(c->h_blob[bi], &p, sizeof(p));
				HIPCHK(c, hipMemcpyAsync(c->d_blob.d, c->h_blob[bi], sizeof(p), hipMemcpyHostToDevice, c->stream));
				HIPCHK(c, hipEventRecord(c->blob_ev[bi], c->stream));
				c->blob_busy[bi] = true;
				c->hparams = p;
				if(c->nfrags != c->blob_nfrags) {
					drop_graphs(c);
					c->quiet_streak = 0;
				} else
					++c->quiet_streak;
				c->blob_nfrags = c->nfrags;
				memcpy(c->blob_frames, c->fragframes, (size_t)c->nfrags * sizeof(unsigned));
				if(c->hosttiming)
					dbg_counters()[0] += 1;
				c->uploaded = true;
				return 0;
			}
		}
	}
	c->blob_quiet = false;
	c->quiet_streak = 0;
	drop_graphs(c);
	const size_t nv = c->voices.size(), nu = c->units.size();
	// capacities
	// (kept: only the entries that changed are re-sent below)
	if(int r = grow(c, c->d_voices, nv, 1, true)) return r;
	if(int r = grow(c, c->d_udesc, nu, 1, false)) return r;
	if(int r = grow(c, c->d_ustate, nu, A2D_USTATE, true)) return r;
	if(int r = grow(c, c->d_ustage, c->d_ustate.cap, A2D_USTATE, false)) return r;
	if(int r = grow(c, c->d_vactive, nv, 1, true)) return r;
	if(int r = grow(c, c->d_runs, nv, 1, true)) return r;
	{
		const int32_t *before = c->d_busmem.d;
		if(int r = grow(c, c->d_busmem, c->bus_used, 1, false)) return r;
		if(c->d_busmem.d != before)
			c->others_clean = c->root_clean = false;
	}
	if(c->fbd_count)
		if(int r = grow(c, c->d_fbdmem, c->fbd_count, 2 * (size_t)A2D_FBD_BUFSIZE, true)) return r;
	if(!c->xio.empty()) {
		if(int r = grow(c, c->d_xio, c->xio.size(), A2D_XIO_SLOT, true)) return r;
		// what the WRITE clients produced for this batch's fragments
		const size_t n = (size_t)c->nfrags * A2AMD_MAXCHANNELS * A2D_FRAG;
		for(size_t k = 0; k < c->xio.size(); ++k) {
			XioSlot &x = c->xio[k];
			// (also when the clients left in the middle of the batch, and zeros
			// over last batch's when they produced nothing)
			if(x.unit < 0 || !(x.inj_used || (c->units[x.unit].xio_mode & A2AMD_XIO_INJECT)))
				continue;
			HIPCHK(c, hipMemcpyAsync(c->d_xio.d + k * A2D_XIO_SLOT + A2D_XIO_HALF, x.inj.data(),
					n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			if(x.inj_used)
				std::fill(x.inj.begin(), x.inj.begin() + n, 0);
			x.inj_used = false;
		}
	}
	if(c->fm_count) {
		if(int r = grow(c, c->d_fmstate, c->fm_count, A2D_FMSTATE, true)) return r;
		if(!c->d_fmsine) {
			// fm_OpenState, fm.c:493-501: one period of sine and one pad
			// sample, computed here with the reference's own expression
			// (libm on the host) and shipped as {s[i], s[i+1] - s[i]} pairs
			int16_t sine[2049];
			uint32_t pairs[2048];
			for(int k = 0; k < 2049; ++k)
				sine[k] = (int16_t)(sin(k * 2.0f * M_PI / 2048) * 32767.0f);
			for(int k = 0; k < 2048; ++k)
				pairs[k] = (uint32_t)(uint16_t)sine[k] | ((uint32_t)(sine[k + 1] - sine[k]) << 16);
			HIPCHK(c, hipMalloc((void **)&c->d_fmsine, sizeof(pairs)));
			HIPCHK(c, hipMemcpy(c->d_fmsine, pairs, sizeof(pairs), hipMemcpyHostToDevice));
		}
	}

	if(c->voices_dirty && nv) {
		// re-upload the span of voice table entries that changed
		if(c->mvoices.size() < nv)
			c->mvoices.resize(nv);
		int lo = (int)nv, hi = -1;
		for(int vi : c->dirty_voices)
			if(vi < (int)nv) {
				sync_voice_mirror(c, vi);
				lo = std::min(lo, vi);
				hi = std::max(hi, vi);
			}
		c->dirty_voices.clear();
		if(hi >= lo)
			HIPCHK(c, hipMemcpyAsync(c->d_voices.d + lo, c->mvoices.data() + lo,
					(size_t)(hi - lo + 1) * sizeof(A2DVoice), hipMemcpyHostToDevice, c->stream));
		if(c->long_chains) {
			// (the first chain of more than 8 units: the side array comes into being, whole; from then on it follows the
			// voice table's dirty span)
			const bool fresh = c->d_vext.cap < c->d_voices.cap;
			if(int r = grow(c, c->d_vext, c->d_voices.cap, 1, true)) return r;
			if(c->mvext.size() < nv)
				c->mvext.resize(nv);
			const int elo = fresh ? 0 : lo, ehi = fresh ? (int)nv - 1 : hi;
			if(ehi >= elo)
				HIPCHK(c, hipMemcpyAsync(c->d_vext.d + elo, c->mvext.data() + elo,
						(size_t)(ehi - elo + 1) * sizeof(A2DVoiceExt), hipMemcpyHostToDevice, c->stream));
		}
		c->voices_dirty = false;
	}
	if(c->udesc_dirty && nu) {
		HIPCHK(c, hipMemcpyAsync(c->d_udesc.d, c->mudesc.data(), nu * sizeof(uint32_t),
				hipMemcpyHostToDevice, c->stream));
		c->udesc_dirty = false;
	}
	if(c->waves_dirty && !c->mwaves.empty()) {
		if(int r = grow(c, c->d_waves, c->mwaves.size(), 1, false)) return r;
		HIPCHK(c, hipMemcpyAsync(c->d_waves.d, c->mwaves.data(), c->mwaves.size() * sizeof(A2DWave),
				hipMemcpyHostToDevice, c->stream));
		c->waves_dirty = false;
	}
	if(c->ptab_dirty) {
		HIPCHK(c, hipMemcpyAsync(c->d_ptab, c->ptab, sizeof(c->ptab), hipMemcpyHostToDevice, c->stream));
		c->ptab_dirty = false;
	}
	for(int b : c->fbd_to_zero)
		HIPCHK(c, hipMemsetAsync(c->d_fbdmem.d + (size_t)b * 2 * A2D_FBD_BUFSIZE, 0,
				2 * (size_t)A2D_FBD_BUFSIZE * sizeof(int32_t), c->stream));
	c->fbd_to_zero.clear();

	// Records: one contiguous run per voice that has any.  The device keeps a
	// dense runs[slot] table (zero = quiet voice); only the entries that change
	// are written, by a scatter kernel: this batch's runs, and zeros for the
	// voices that carried records last batch but not now.
	std::vector<A2DRec> &recs = c->up_recs;
	std::vector<int> &sc_idx = c->up_idx, &now = c->up_now;
	std::vector<A2DRun> &sc_val = c->up_val;
	recs.clear();
	now.clear();
	{
		const size_t nsc = c->with_recs.size() + c->prev_with_recs.size();
		sc_idx.resize(nsc);
		sc_val.resize(nsc);
		now.resize(c->with_recs.size());
	}
	size_t nsc_used = 0, nnow = 0;
	static const bool keep_bus_windows = getenv("A2AMD_BUS_WINDOWS") != nullptr;	// (A/B: round 4's routing)
	for(int vi : c->with_recs) {
		HVoice &v = c->voices[vi];
		if(v.recs.empty()) {
			v.listed_recs = false;
			continue;
		}
		// A group voice with the stateless chain inline -> panmix -> xinsert whose records are windows only - its own VM
		// woke inside a fragment (a sequencer that spawns notes), or its parent's did - and whose volume and pan are at
		// rest: the windows change nothing (a2_PrepareRamper with no ramp under way sets the same gains for every window,
		// a2_dsp.h:128-133; inline, panmix and xinsert keep no state from frame to frame), so it stays with the bus
		// kernel of its depth instead of going to the general kernel - one wavefront that is as long as the batch.
		if(!keep_bus_windows && v.inline_pos >= 0 && v.vm < 0 && v.moving_until <= c->vm.batch_time && !(c->no_fast & 4) &&
				v.resolved && is_driver_chain(c, v)) {
			bool only_windows = true;
			for(const A2DRec &r : v.recs)
				if(A2D_ROP(r.head) != R_SEG) {
					only_windows = false;
					break;
				}
			if(only_windows) {
				v.recs.clear();
				v.listed_recs = false;
				continue;
			}
		}
		A2DRun r = { (int)recs.size(), (int)v.recs.size() };
		recs.insert(recs.end(), v.recs.begin(), v.recs.end());
		sc_idx[nsc_used] = vi;
		sc_val[nsc_used++] = r;
		now[nnow++] = vi;
	}
	// Voices without records whose controls are still gliding (HVoice::moving_until): the quiet kernels take an
	// unsettled voice fragment by fragment on the scalar unit for the whole batch; the window kernels resolve its
	// rampers in closed form.  They are given ONE shared stand-in record that belongs to no fragment - a run the
	// quiet kernels skip and the control pass never consumes: default windows throughout.
	static const bool no_moving = getenv("A2AMD_NO_MOVING") != nullptr;
	c->n_moving_listed = 0;
	if(!c->moving.empty()) {
		int nop_at = -1;
		const uint64_t t0 = c->vm.batch_time;	// walk_time when this batch began
		for(size_t i = 0; i < c->moving.size();) {
			const int vi = c->moving[i];
			HVoice &v = c->voices[vi];
			if(!(v.live || v.dying) || v.moving_until <= t0) {
				v.listed_moving = false;
				v.moving_until = 0;
				c->moving[i] = c->moving.back();
				c->moving.pop_back();
				continue;
			}
			++i;
			// (a voice slot that was recycled while it stood in the list stands in it twice)
			if(v.moving_run == c->serial_base)
				continue;
			if(no_moving || !v.recs.empty() || v.vm >= 0 || c->lists_dirty || v.mode_mix || !v.resolved ||
					!(v.cls == CLS_OSCPAN || v.cls == CLS_OSC2PAN || v.cls == CLS_OSCFILTPAN || v.cls == CLS_OSC2FILTPAN))
				continue;
			if(nop_at < 0) {
				nop_at = (int)recs.size();
				A2DRec nop = { A2D_HEAD(0xffff, R_NOP, 0, 0), 0, 0, 0 };
				recs.push_back(nop);
			}
			const A2DRun r = { nop_at, 1 };
			sc_idx.resize(nsc_used + 1 + c->prev_with_recs.size());
			sc_val.resize(sc_idx.size());
			sc_idx[nsc_used] = vi;
			sc_val[nsc_used++] = r;
			now.resize(nnow + 1);
			now[nnow++] = vi;
			v.moving_run = c->serial_base;
			v.listed_recs = true;
			++c->n_moving_listed;
		}
	}
	now.resize(nnow);
	c->with_recs.swap(now);
	if(sc_idx.size() < nsc_used + c->prev_with_recs.size()) {
		sc_idx.resize(nsc_used + c->prev_with_recs.size());
		sc_val.resize(sc_idx.size());
	}
	for(int vi : c->prev_with_recs)
		if(vi < (int)nv && c->voices[vi].recs.empty() && c->voices[vi].moving_run != c->serial_base) {
			A2DRun z = { 0, 0 };
			sc_idx[nsc_used] = vi;
			sc_val[nsc_used++] = z;
		}
	sc_idx.resize(nsc_used);
	sc_val.resize(nsc_used);
	c->prev_with_recs.clear();
	c->stats.records += recs.size();
	if(c->hosttiming) {
		dbg_counters()[1] += (double)recs.size();
		dbg_counters()[4] += (double)c->with_recs.size();
	}

	// Launch lists.  Static part, rebuilt when the voice tree changes: every
	// listed voice by class -
	//   leaves: [wtosc-panmix | wtosc-filter12-panmix | general]  (each sorted by
	//           output bus so a wavefront can sum several voices before touching it)
	//   voices with an inline unit, per nesting depth: [driver chain | general]
	// The fast kernels skip a voice whose runs[] entry is non-zero; those voices
	// form the dynamic part (this batch's exceptions) and go to the general kernel.
	if(c->lists_dirty) {
		static const bool cls_check = getenv("A2AMD_CLS_CHECK") != nullptr;
		bool owners_ok = !getenv("A2AMD_NO_SELFCLEAN"), root_driver = false;
		std::vector<int> fast_leaf, osc2_leaf, filt_leaf, fm_leaf, gen_leaf, o2f_leaf;
		std::map<int, std::pair<std::vector<int>, std::vector<int>>> bydepth;
		std::map<int, std::vector<int>> fbd_bydepth;
		int maxdepth = -1;
		for(size_t vi = 0; vi < nv; ++vi) {
			HVoice &v = c->voices[vi];
			// voices that died during this batch still render up to their
			// R_KILL record
			if(!(v.live || v.dying) || !v.resolved) {
				v.cls = -1;
				continue;
			}
			// (what a voice's class is made of - its units' kinds, wiring and modes, its place in the tree, clients on its
			// xinsert, delay taps, mode_mix - changes at a handful of places, each of which says so: cls_stale.  A rebuild -
			// one per batch in which a note is born or ends - classifies only those voices; A2AMD_CLS_CHECK=1, the test
			// suite's setting, classifies all and fails if a remembered class is not what comes out)
			const bool classify = v.cls_stale || v.cls < 0 || cls_check;
			const int was_cls = v.cls;
			if(v.inline_pos >= 0) {
				auto &d = bydepth[v.depth];
				if(classify)
					v.cls = !(c->no_fast & 4) && is_driver_chain(c, v) ? CLS_BUSDRIVER :
							!(c->no_fast & 32) && v.depth > 0 && is_fbdchain(c, v) ? CLS_FBDCHAIN : CLS_BUSGENERIC;
				// (the master bus at offset 0 is the root's alone)
				if(v.cls == CLS_BUSGENERIC || (v.out_off == 0) != (v.depth == 0))
					owners_ok = false;
				else if(v.depth == 0)
					root_driver = true;
				(v.cls == CLS_BUSDRIVER ? d.first : v.cls == CLS_FBDCHAIN ? fbd_bydepth[v.depth] : d.second).push_back((int)vi);
				maxdepth = std::max(maxdepth, v.depth);
			} else {
				if(classify)
					v.cls = !(c->no_fast & 1) && is_oscpan_chain(c, v) ? CLS_OSCPAN :
							!(c->no_fast & 8) && is_osc2pan_chain(c, v) ? CLS_OSC2PAN :
							!(c->no_fast & 2) && is_oscfiltpan_chain(c, v) ? CLS_OSCFILTPAN :
							// (no quiet kernel of its own: k_leaf_recs renders it, records or not - unless an
							// oscillator leaves the mip-mapped waves somewhere in this batch)
							!(c->no_fast & 128) && !v.mode_mix && is_osc2filtpan_chain(c, v) ? CLS_OSC2FILTPAN :
							!(c->no_fast & 16) && is_fmpan_chain(c, v) ? CLS_FMPAN : CLS_GENERIC;
				if(v.out_off == 0)
					owners_ok = false;	// adds straight into the master bus
				(v.cls == CLS_OSCPAN ? fast_leaf : v.cls == CLS_OSC2PAN ? osc2_leaf :
				 v.cls == CLS_OSCFILTPAN ? filt_leaf : v.cls == CLS_FMPAN ? fm_leaf :
				 v.cls == CLS_OSC2FILTPAN ? o2f_leaf : gen_leaf).push_back((int)vi);
			}
			if(cls_check && !v.cls_stale && was_cls >= 0 && was_cls != v.cls)
				return c->fail(A2AMD_ESTATE, "voice %zu: remembered launch class %d, classified %d (a change nobody reported)",
						vi, was_cls, v.cls);
			v.cls_stale = false;
		}
		auto by_bus = [&](int a, int b) { return c->voices[a].out_off < c->voices[b].out_off; };
		// (slots follow the order of birth, births the walk: a class is usually grouped by bus already - one pass
		// says so, where a sort of 16 384 voices by a key two loads away was most of a rebuild; a scene with a note
		// born or ended in most batches rebuilds in most batches)
		auto sort_by_bus = [&](std::vector<int> &l) {
			if(!std::is_sorted(l.begin(), l.end(), by_bus))
				std::stable_sort(l.begin(), l.end(), by_bus);
		};
		sort_by_bus(fast_leaf);
		sort_by_bus(gen_leaf);
		sort_by_bus(filt_leaf);
		sort_by_bus(osc2_leaf);
		c->list_all = fast_leaf;
		c->n_fast_leaf = (int)fast_leaf.size();
		c->list_all.insert(c->list_all.end(), osc2_leaf.begin(), osc2_leaf.end());
		c->n_osc2_leaf = (int)osc2_leaf.size();
		c->list_all.insert(c->list_all.end(), filt_leaf.begin(), filt_leaf.end());
		c->n_filt_leaf = (int)filt_leaf.size();
		{
			// fm voices, grouped by unit kind: one launch per kind present
			std::stable_sort(fm_leaf.begin(), fm_leaf.end(), [&](int a, int b) {
				const int ka = c->units[c->voices[a].unit[0]].kind, kb = c->units[c->voices[b].unit[0]].kind;
				return ka != kb ? ka < kb : c->voices[a].out_off < c->voices[b].out_off;
			});
			for(int k = 0; k < 8; ++k)
				c->fm_kind_count[k] = 0;
			for(int vi : fm_leaf)
				++c->fm_kind_count[c->units[c->voices[vi].unit[0]].kind - A2AMD_FM1];
			c->list_all.insert(c->list_all.end(), fm_leaf.begin(), fm_leaf.end());
			c->n_fm_leaf = (int)fm_leaf.size();
		}
		c->list_all.insert(c->list_all.end(), gen_leaf.begin(), gen_leaf.end());
		c->n_leaf = (int)gen_leaf.size();
		sort_by_bus(o2f_leaf);
		c->list_all.insert(c->list_all.end(), o2f_leaf.begin(), o2f_leaf.end());
		c->n_o2f_leaf = (int)o2f_leaf.size();
		c->depth_ranges.assign(maxdepth + 1, DepthRange());
		for(int d = 0; d <= maxdepth; ++d) {
			auto &l = bydepth[d];
			DepthRange &r = c->depth_ranges[d];
			r.fast_first = (int)c->list_all.size();
			r.fast_count = (int)l.first.size();
			c->list_all.insert(c->list_all.end(), l.first.begin(), l.first.end());
			r.fbd_first = (int)c->list_all.size();
			r.fbd_count = (int)fbd_bydepth[d].size();
			c->list_all.insert(c->list_all.end(), fbd_bydepth[d].begin(), fbd_bydepth[d].end());
			r.gen_first = (int)c->list_all.size();
			r.gen_count = (int)l.second.size();
			c->list_all.insert(c->list_all.end(), l.second.begin(), l.second.end());
		}
		c->static_len = (int)c->list_all.size();
		if(int r = grow(c, c->d_list, c->list_all.size() + 64, 1, false)) return r;
		if(!c->list_all.empty())
			HIPCHK(c, hipMemcpyAsync(c->d_list.d, c->list_all.data(), c->list_all.size() * sizeof(int),
					hipMemcpyHostToDevice, c->stream));
		c->lists_dirty = false;
		c->owners_all_driver = owners_ok && root_driver;
		if(!c->vm.list.empty() || c->vm.stats.live)
			c->vm.list_dirty = true;	// (launch classes may have moved)
	}
	if(int r = vm_build_lists(c))
		return r;
	std::vector<int> dyn_all;
	{
		// this batch's exceptions (shipped in the blob)
		std::vector<int> dyn_leaf;
		std::vector<std::vector<int>> dyn_bus(c->depth_ranges.size());
		static const bool dump = getenv("A2AMD_VM_DUMP") != nullptr;	// (debugging aid: the records of a batch, host-made ...)
		for(int vi : c->with_recs) {
			const HVoice &v = c->voices[vi];
			if(dump)
				for(const A2DRec &r : v.recs)
					fprintf(stderr, "REC %lld v%d f%u op%u u%u r%u val %d dur %u start %u\n", c->serial_base, vi,
							A2D_RFRAG(r.head), A2D_ROP(r.head), A2D_RUNIT(r.head), A2D_RREG(r.head), r.value, r.dur, r.start);
			// (fm-panmix voices execute their own records in k_leaf_fmpan)
			if(v.cls == CLS_OSCPAN || v.cls == CLS_OSCFILTPAN || v.cls == CLS_OSC2PAN || v.cls == CLS_OSC2FILTPAN)
				dyn_leaf.push_back(vi);
			else if((v.cls == CLS_BUSDRIVER || v.cls == CLS_FBDCHAIN) && v.depth < (int)dyn_bus.size()) {
				dyn_bus[v.depth].push_back(vi);
				static const int trace = getenv("A2AMD_HOSTTIMING") ? atoi(getenv("A2AMD_HOSTTIMING")) : 0;
				if(trace >= 3 && !v.recs.empty())
					fprintf(stderr, "a2amd: bus voice %d (depth %d, class %d) carries %zu records, first: frag %u op %u unit %u reg %u value %d\n",
							vi, v.depth, v.cls, v.recs.size(), A2D_RFRAG(v.recs[0].head), A2D_ROP(v.recs[0].head),
							A2D_RUNIT(v.recs[0].head), A2D_RREG(v.recs[0].head), v.recs[0].value);
			}
		}
		// Voices of the wtosc[+wtosc]->panmix classes whose records are what
		// k_leaf_recs executes (windows, writes, births, deaths; oscillators on
		// mip-mapped waves throughout the batch) go first, by class; the rest -
		// filter voices, a wave of another kind somewhere in the batch - to the
		// general kernel.
		std::vector<int> dyn_o1, dyn_o2, dyn_f1, dyn_f2, dyn_rest;
		const bool no_recs_kernel = (c->no_fast & 64) != 0;
		for(int vi : dyn_leaf) {
			HVoice &v = c->voices[vi];
			// (close_fragment's R_NOP is the one other record k_leaf_recs takes - as nothing)
			const bool ok = !no_recs_kernel && !v.mode_mix && !v.fancy_recs;
			if(ok && v.cls == CLS_OSC2FILTPAN)
				v.dynf2_run = c->serial_base;
			(!ok ? dyn_rest : v.cls == CLS_OSCPAN ? dyn_o1 : v.cls == CLS_OSC2PAN ? dyn_o2 :
			 v.cls == CLS_OSCFILTPAN ? dyn_f1 : dyn_f2).push_back(vi);
		}
		// Round 6: 2 x wtosc-filter12-panmix has a quiet kernel of its own (k_leaf_osc2filtpan) and rounds 2-5's "the records
		// kernels render every voice of the class" is over: like the other three classes, a voice is the quiet kernel's
		// in a batch in which its run is empty and the records / window kernels' otherwise.  Whose records the DEVICE VM
		// writes (k_vm_emit, after this list is made) the host cannot tell which it will be: the class's adopted voices
		// stand on the list in every batch, and the list is taken with skip_empty - the one rule "runs[v].count == 0 <=>
		// the quiet kernel's" decides for every voice, whoever wrote its records.
		for(int vi : c->vm.o2f_voices) {
			HVoice &v = c->voices[vi];
			if(v.dynf2_run == c->serial_base || v.cls != CLS_OSC2FILTPAN || !(v.live || v.dying) || v.vm < 0)
				continue;
			v.dynf2_run = c->serial_base;
			dyn_f2.push_back(vi);
		}
		// ... but not in a small scene.  A launch of k_leaf_osc2filtpan takes as long as its filter wavefront's chain
		// through the batch whatever its voice count - and on the context's one stream it runs BEHIND the records kernel,
		// which renders a song's few such voices inside the launch that renders the song's other record voices
		// (measured: the 60-voice song, 500 s, 2.94 s in round 5 -> 3.53 s with the quiet kernel for its handful of notes,
		// profiles/r06_song_timing.jsonl).  Below A2AMD_O2F_MIN voices of the class (default 512) every voice of it is the
		// records / window kernels', as in rounds 2 - 5.
		static const int o2f_min = getenv("A2AMD_O2F_MIN") ? atoi(getenv("A2AMD_O2F_MIN")) : 512;
		c->o2f_quiet = c->n_o2f_leaf >= o2f_min;
		if(!c->o2f_quiet && c->n_o2f_leaf) {
			const int at = c->n_fast_leaf + c->n_osc2_leaf + c->n_filt_leaf + c->n_fm_leaf + c->n_leaf;
			for(int k = 0; k < c->n_o2f_leaf; ++k) {
				const int vi = c->list_all[at + k];
				HVoice &v = c->voices[vi];
				if(v.dynf2_run == c->serial_base || v.mode_mix)
					continue;
				v.dynf2_run = c->serial_base;
				dyn_f2.push_back(vi);
			}
		}
		// (the walk order usually has them grouped by bus already)
		auto by_bus_dyn = [&](int a, int b) { return c->voices[a].out_off < c->voices[b].out_off; };
		for(std::vector<int> *l : { &dyn_o1, &dyn_o2, &dyn_f1, &dyn_f2, &dyn_rest })
			if(!std::is_sorted(l->begin(), l->end(), by_bus_dyn))
				std::stable_sort(l->begin(), l->end(), by_bus_dyn);
		std::vector<int> dyn = dyn_o1;
		dyn.insert(dyn.end(), dyn_o2.begin(), dyn_o2.end());
		dyn.insert(dyn.end(), dyn_f1.begin(), dyn_f1.end());
		dyn.insert(dyn.end(), dyn_f2.begin(), dyn_f2.end());
		dyn.insert(dyn.end(), dyn_rest.begin(), dyn_rest.end());
		c->n_dyn_osc1 = (int)dyn_o1.size();
		c->n_dyn_osc2 = (int)dyn_o2.size();
		c->n_dyn_filt = (int)dyn_f1.size();
		c->n_dyn_filt2 = (int)dyn_f2.size();
		c->n_dyn_rest = (int)dyn_rest.size();
		c->n_leaf_dyn = (int)dyn.size();	// (with the device VM's voices of the fourth class, which dyn_leaf does not hold)
		for(size_t d = 0; d < dyn_bus.size(); ++d) {
			c->depth_ranges[d].dyn_first = (int)dyn.size();
			c->depth_ranges[d].dyn_count = (int)dyn_bus[d].size();
			dyn.insert(dyn.end(), dyn_bus[d].begin(), dyn_bus[d].end());
		}
		dyn_all.swap(dyn);
		c->consume_ok = c->owners_all_driver;
		for(const std::vector<int> &d : dyn_bus)
			if(!d.empty())
				c->consume_ok = false;	// a bus owner carries records: the general kernel renders it
	}

	A2DParams p;
	memset(&p, 0, sizeof(p));
	p.voices = c->d_voices.d;
	p.vext = c->long_chains ? c->d_vext.d : nullptr;
	p.udesc = c->d_udesc.d;
	p.ustate = c->d_ustate.d;
	p.vactive = c->d_vactive.d;
	p.runs = c->d_runs.d;
	p.waves = c->d_waves.d;
	p.wavepool = c->d_wavepool.d;
	p.wavecoef = c->d_wavecoef.d;
	p.busmem = c->d_busmem.d;
	p.fbdmem = c->d_fbdmem.d;
	p.ptab = c->d_ptab;
	p.fmstate = c->d_fmstate.d;
	p.xio = c->d_xio.d;
	p.fmsine = c->d_fmsine;
	p.nfrags = c->nfrags;
	p.samplerate = c->cfg.samplerate;
	p.debug = getenv("A2AMD_DEBUG") ? atoi(getenv("A2AMD_DEBUG")) : 0;
	for(int f = 0, acc = 0; f < c->nfrags; ++f) {
		p.fragframes[f] = (uint8_t)c->fragframes[f];
		p.fragstart[f] = (uint16_t)acc;
		acc += (int)c->fragframes[f];
	}

	// the blob: [params | records | scatter indices | scatter runs | exception lists]
	auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t nsc = sc_idx.size();
	const size_t o_recs = up256(sizeof(A2DParams));
	const size_t o_idx = o_recs + up256((recs.size() + 1) * sizeof(A2DRec));
	const size_t o_val = o_idx + up256(nsc * sizeof(int));
	const size_t o_dyn = o_val + up256(nsc * sizeof(A2DRun));
	// ... and behind what the host makes, room for the records the device VM writes (a2amd_vm.cpp)
	const size_t host_total = o_dyn + up256(dyn_all.size() * sizeof(int));
	const size_t vm_room = (size_t)vm_blob_room(c);
	const size_t total = host_total;
	if(int r = grow(c, c->d_blob, host_total + vm_room * sizeof(A2DRec), 1, false)) return r;
	c->vm.rec_off = host_total;
	c->vm.rec_cap = (uint32_t)((c->d_blob.cap - host_total) / sizeof(A2DRec));
	const int bi = c->blob_i;
	c->blob_i ^= 1;
	if(c->blob_busy[bi]) {		// the copy that last read this staging buffer must have run
		HIPCHK(c, hipEventSynchronize(c->blob_ev[bi]));
		c->blob_busy[bi] = false;
	}
	if(total > c->h_blob_cap[bi]) {
		if(c->h_blob[bi])
			HIPCHK(c, hipHostFree(c->h_blob[bi]));
		c->h_blob_cap[bi] = std::max(total * 2, (size_t)65536);
		HIPCHK(c, hipHostMalloc((void **)&c->h_blob[bi], c->h_blob_cap[bi], hipHostMallocDefault));
	}
	if(!c->blob_ev[bi])
		HIPCHK(c, hipEventCreateWithFlags(&c->blob_ev[bi], hipEventDisableTiming));
	char *hb = c->h_blob[bi];
	p.recs = (const A2DRec *)(c->d_blob.d + o_recs);
	c->hparams = p;
	memcpy(hb, &p, sizeof(p));
	if(!recs.empty())
		memcpy(hb + o_recs, recs.data(), recs.size() * sizeof(A2DRec));
	if(nsc) {
		memcpy(hb + o_idx, sc_idx.data(), nsc * sizeof(int));
		memcpy(hb + o_val, sc_val.data(), nsc * sizeof(A2DRun));
	}
	if(!dyn_all.empty())
		memcpy(hb + o_dyn, dyn_all.data(), dyn_all.size() * sizeof(int));
	HIPCHK(c, hipMemcpyAsync(c->d_blob.d, hb, total, hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipEventRecord(c->blob_ev[bi], c->stream));
	c->blob_busy[bi] = true;
	c->d_params = (A2DParams *)c->d_blob.d;
	c->d_dyn = (const int *)(c->d_blob.d + o_dyn);
	if(nsc)
		if(a2d_launch_scatter_runs((const int *)(c->d_blob.d + o_idx), (const A2DRun *)(c->d_blob.d + o_val),
				(int)nsc, c->d_runs.d, c->stream))
			return c->fail(A2AMD_EHIP, "scatter launch failed");
	c->uploaded = true;
	c->blob_quiet = recs.empty() && dyn_all.empty();
	c->blob_nfrags = c->nfrags;
	memcpy(c->blob_frames, c->fragframes, (size_t)c->nfrags * sizeof(unsigned));
	return 0;
}

// Shape of the fast leaf launch, from sweeps on MI355X (DESIGN.md "Launch
// shape"): cut the batch into as many time slices as it has 8-fragment chunks
// (up to 8), then give a wavefront enough voices that about 4096 wavefronts
// (one resident round of 256 CUs x 16) share the work; at least 4 voices, so
// that their sum reaches the bus in one atomic instead of four, at most 32.
void pick_fast_shape(int n, int nfrags, int *vpw, int *ysplit, int ymax = 32)
{
	const int nchunks = (nfrags + A2D_FAST_FCH - 1) / A2D_FAST_FCH;
	int y = getenv("A2AMD_YSPLIT") ? atoi(getenv("A2AMD_YSPLIT")) : ymax;
	y = std::min(std::max(y, 1), nchunks);
	int v = getenv("A2AMD_VPW") ? atoi(getenv("A2AMD_VPW")) :
			std::min(std::max((int)(((long long)n * y + 4095) / 4096), 4), 32);
	*vpw = std::min(std::max(v, 1), 64);
	*ysplit = y;
}

int pick_fast_vpw(int n)
{
	// enough wavefronts to fill 256 CUs x 4 SIMDs several times over, then
	// more voices per wavefront (fewer, fatter bus updates)
	if(getenv("A2AMD_VPW"))
		return std::min(std::max(atoi(getenv("A2AMD_VPW")), 1), 64);
	int v = (n + 4095) / 4096;
	return std::min(std::max(v, 1), 64);
}

int launch_depth(a2amd_ctx *c, int d, int consume, A2DCommitSet *pend)	// consume: 1 zero what is read, 2 root stores the master bus
{
	const DepthRange &r = c->depth_ranges[d];
	if(r.fast_count) {
		// (state commits the time-sliced leaf kernels left behind ride along)
		// (the root, a plain driver chain that stores the master bus: into the host's buffer where asked)
		// (... and only if nothing else adds into the master bus at depth 0: no voice with records there this
		// batch, no leaf voice mixing straight into it - owners_all_driver says so for the static lists)
		const bool direct = d == 0 && (consume & 2) && c->master_dst && r.fast_count == 1 && !r.gen_count && !r.fbd_count &&
				!r.dyn_count && c->owners_all_driver && !c->capture.on;	// (a capture reads the bus memory)
		if(direct)
			c->master_direct = true;
		if(a2d_launch_bus_driver(c->d_params, c->d_list.d + r.fast_first, r.fast_count, c->nfrags, consume,
				pend, c->stream, direct ? c->master_dst : nullptr))
			return c->fail(A2AMD_EHIP, "bus driver launch failed: %s", hipGetErrorString(hipGetLastError()));
		pend->n = 0;
		++c->stats.launches;
	}
	if(r.fbd_count) {
		if(a2d_launch_bus_fbdchain(c->d_params, c->d_list.d + r.fbd_first, r.fbd_count, consume & 1, c->stream))
			return c->fail(A2AMD_EHIP, "delay chain launch failed: %s", hipGetErrorString(hipGetLastError()));
		++c->stats.launches;
	}
	if(r.gen_count) {
		if(a2d_launch_voices(c->d_params, c->d_list.d + r.gen_first, r.gen_count, 1, c->stream))
			return c->fail(A2AMD_EHIP, "bus launch failed: %s", hipGetErrorString(hipGetLastError()));
		++c->stats.launches;
	}
	if(r.dyn_count) {
		if(a2d_launch_voices(c->d_params, c->d_dyn + r.dyn_first, r.dyn_count, 1, c->stream))
			return c->fail(A2AMD_EHIP, "bus launch failed: %s", hipGetErrorString(hipGetLastError()));
		++c->stats.launches;
	}
	return 0;
}

int pick_vpw(int n)
{
	int v = n / 4096;
	return std::min(std::max(v, 1), (int)A2D_MAXVPW);
}

// give a region of the wave pool back: kept sorted by offset, neighbours merged
void wavepool_release(a2amd_ctx *c, size_t off, size_t len)
{
	if(!len)
		return;
	auto &fl = c->wavepool_free;
	auto it = std::lower_bound(fl.begin(), fl.end(), std::make_pair(off, (size_t)0));
	it = fl.insert(it, std::make_pair(off, len));
	if(it + 1 != fl.end() && it->first + it->second == (it + 1)->first) {
		it->second += (it + 1)->second;
		it = fl.erase(it + 1) - 1;
	}
	if(it != fl.begin() && (it - 1)->first + (it - 1)->second == it->first) {
		(it - 1)->second += it->second;
		it = fl.erase(it) - 1;
	}
	// the tail of the pool grows back into unused space
	if(it->first + it->second == c->wavepool_used) {
		c->wavepool_used = it->first;
		fl.erase(it);
	}
}

void end_batch(a2amd_ctx *c)
{
	// (graphs survive: upload() drops them unless the next batch is the same quiet one)
	// Records made after the last fragment of the batch was closed belong to
	// the first fragment of the next batch: carry them over.
	const int done = c->nfrags;
	c->serial_base += done;
	std::vector<int> carry;
	c->prev_with_recs.clear();
	for(int vi : c->with_recs) {
		HVoice &v = c->voices[vi];
		c->prev_with_recs.push_back(vi);
		size_t keep = 0;
		// (a voice that was set up but not walked yet keeps everything)
		const bool unborn = v.live && !v.resolved;
		// (the records are in fragment order: nothing to carry over unless the last one is)
		if(unborn || (!v.recs.empty() && (int)A2D_RFRAG(v.recs.back().head) >= done))
			for(size_t i = 0; i < v.recs.size(); ++i)
				if(unborn || (int)A2D_RFRAG(v.recs[i].head) >= done) {
					A2DRec r = v.recs[i];
					int f = (int)A2D_RFRAG(r.head) - done;
					r.head = (r.head & 0xffff0000u) | (uint32_t)(f < 0 ? 0 : f);
					v.recs[keep++] = r;
				}
		v.recs.resize(keep);
		v.frag_mark = 0;
		if(keep) {
			v.touched = c->serial_base;
			carry.push_back(vi);
		} else {
			v.touched = -1;
			v.listed_recs = false;
			if(v.mode_mix) {
				c->lists_dirty = true;	// (it may have its leaf class back)
				v.cls_stale = true;
			}
			v.mode_mix = false;
			v.fancy_recs = false;
		}
	}
	c->with_recs = carry;
	for(int vi : c->deferred_free_voices) {
		c->voices[vi] = HVoice();
		c->free_voices.push_back(vi);
		c->lists_dirty = true;
	}
	c->deferred_free_voices.clear();
	for(int ui : c->deferred_free_units)
		c->free_units.push_back(ui);
	c->deferred_free_units.clear();
	for(auto &b : c->deferred_bus_free)
		c->bus_free[b.second].push_back(b.first);
	c->deferred_bus_free.clear();
	for(int b : c->fbd_deferred_free)
		c->fbd_free.push_back(b);
	c->fbd_deferred_free.clear();
	for(int b : c->fm_deferred_free)
		c->fm_free.push_back(b);
	c->fm_deferred_free.clear();
	for(int b : c->xio_deferred_free) {
		c->xio[b].unit = -1;
		c->xio_free.push_back(b);
	}
	c->xio_deferred_free.clear();
	for(auto &r : c->deferred_wavepool_free)
		wavepool_release(c, r.first, r.second);
	c->deferred_wavepool_free.clear();
	for(int w : c->deferred_wave_slots)
		c->free_wave_slots.push_back(w);
	c->deferred_wave_slots.clear();
	c->nfrags = 0;
	c->cur_frag = 0;
	c->frag_open = false;
	c->uploaded = false;
	c->sub_resume = -1;
	c->paused_at = 0;
	vm_end_batch(c);
}

// does a voice at nesting depth d hold an xinsert in A2AMD_XIO_MUTE mode (insert clients)?
bool depth_has_mutes(const a2amd_ctx *c, int d)
{
	for(const XioSlot &x : c->xio)
		if(x.last_unit >= 0 && x.last_unit < (int)c->units.size() && (c->units[x.last_unit].xio_mode & A2AMD_XIO_MUTE) &&
				c->units[x.last_unit].voice >= 0 && c->voices[c->units[x.last_unit].voice].depth == d)
			return true;
	return false;
}

// the kernels of one batch, in stream order; e* may be null
// The wavetable leaf voices that carry records this batch - the four host-made class lists (wtosc |
// 2 x wtosc [-> filter12] -> panmix; the last one with its quiet voices) and the three lists of voices
// whose records the device VM has just written - through the window kernels (a2amd_win.hip): per slab
// of the batch the control pass of every list, then the render pass of every list.  A slot per fragment
// and voice, a pool of further windows sized by the bound the control pass allocates by (one per record);
// a batch whose slots would not fit A2AMD_WIN_MB (1 024) is cut into slabs of fragments.
static int issue_windows(a2amd_ctx *c, const int *const *lists, const int *counts, const std::function<int()> &mid)
{
	const int sset = c->vm.spec_set;	// (the speculative slot set a taken pass wrote: mid() may launch the next pass)
	// vmk >= 0: a class of VM voices run by k_vm_win; spec: ... that a speculative pass has already run for this batch
	// (vm_issue took it): its entries stand in the pass's own slot set and pool, at (sat, satw) - no control pass, no
	// room in this batch's slot memory, k_vm_commit in k_vm_win's place
	struct Job { int nosc, filt, n, skip; const int *list; int vmk; bool spec; size_t sat, satw; };
	static const int nosc[4] = { 1, 2, 1, 2 }, filt[4] = { 0, 0, 1, 1 };
	Job jobs[7];
	int nj = 0;
	size_t nvoices = 0;		// ... of the jobs that take slots here
	const bool spec = c->vm.fused && c->vm.spec_use;
	for(int k = 0; k < 4; ++k)
		if(counts[k]) {
			jobs[nj++] = Job{ nosc[k], filt[k], counts[k], k == 3 && c->o2f_quiet, lists[k], -1, false, 0, 0 };	// (k == 3: see upload(), dyn_f2)
			nvoices += (size_t)counts[k];
		}
	// (the control pass takes room in the pool by the length of a voice's record run: every gliding voice's run is
	// the ONE shared stand-in record)
	size_t nrec = c->up_recs.size() + c->n_moving_listed;
	size_t nfusedv = 0;
	if(!c->vm.list.empty()) {
		const int *l = c->vm.d_list.d + c->vm.list.size();
		size_t sat = 0, satw = 0;	// (the speculative pass's layout: the three classes one after the other, vm_speculate)
		for(int k = 0; k < 3; l += c->vm.n_cls[k++])
			if(c->vm.n_cls[k]) {
				jobs[nj++] = Job{ nosc[k], filt[k], c->vm.n_cls[k], 1, l, c->vm.fused ? k : -1, spec, sat, satw };
				if(!spec)
					nvoices += (size_t)c->vm.n_cls[k];
				if(c->vm.fused && !spec)
					nfusedv += (size_t)c->vm.n_cls[k];
				sat += (size_t)c->vm.n_cls[k] * (size_t)c->nfrags;
				satw += (size_t)c->vm.n_cls[k] * (size_t)c->nfrags * A2D_WIN_SLOTWORDS(nosc[k], filt[k]);
			}
		nrec += c->vm.last_total;
	}
	// (k_vm_win takes pool room as its voices' VMs make further windows: how much was counted ahead, by k_vm_pool
	// behind the last batch - vm_issue, a2amd_vm.cpp - for exactly this batch)
	if(nfusedv)
		nrec += c->vm.pred_entries + 64;
	if(nfusedv || spec)
		++c->vm.fused_batches;
	if(!nj)
		return 0;
	// (a batch whose every list took a speculative pass takes nothing from THIS pool and has no control pass of its own:
	// its counter is neither cleared nor read back - two calls in front of the commit, two behind the render pass, of a
	// buffer whose time is the engine thread's as much as the GPU's)
	bool plain_jobs = false;
	for(int j = 0; j < nj; ++j)
		plain_jobs = plain_jobs || !jobs[j].spec;
	if(c->wtop_pending) {
		// (the copy sits behind the batch before this one: done, or nearly - waited for rather than polled, because this
		// batch's memset and copy overwrite d_wtop / h_wtop and an overflow flag not read here would be lost)
		HIPCHK(c, hipEventSynchronize(c->wtop_ev));
		c->wtop_pending = false;
		c->vm.pool_used = std::max(c->h_wtop[0], c->h_wtop[2]);
		if(c->h_wtop[1] || c->h_wtop[3]) {
			// (a bound that did not hold - the control pass's record count, k_vm_pool's prediction: never)
			c->vm.fused_off = true;
			return c->fail(A2AMD_ESTATE, "window pool overflow in an earlier batch (%u / %u entries taken of %zu): voices lost windows",
					c->h_wtop[0], c->h_wtop[2], c->d_wext.cap);
		}
	}
	static const size_t budget = (size_t)(getenv("A2AMD_WIN_MB") ? atoi(getenv("A2AMD_WIN_MB")) : 1024) * (1u << 20) /
			(A2D_WIN_WORDS * sizeof(int)) / 2;
	// Slabs: the control pass (lane = voice: a few hundred wavefronts, each as long as its voices' walk) of slab
	// k + 1 runs on a stream of its own beside the render pass of slab k - two sets of slots and pools.  A batch
	// of 16 fragments or more is cut into four slabs (A2AMD_WIN_SLABS), a shorter one is one slab.
	// (measured, 16 384 scripted voices x 64 fragments: 0.50 ms as one slab, 0.76 as four - the control pass is as
	// long as ONE wavefront's walk, and a wavefront that shares its SIMD with seven render wavefronts walks at a
	// fraction of its pace - so slabs are what the memory bound asks for, not the default)
	static const int want_slabs = getenv("A2AMD_WIN_SLABS") ? std::max(1, atoi(getenv("A2AMD_WIN_SLABS"))) : 1;
	const int nfrags = c->nfrags;
	// (a taken speculative pass covers the batch in one piece: with it, the lists that do take slots here are not cut
	// into slabs for the asking - A2AMD_WIN_SLABS - only where the memory bound demands it)
	int per = nfrags >= 16 && !spec ? (nfrags + want_slabs - 1) / want_slabs : nfrags;
	if(nvoices && nvoices * (size_t)per > budget)
		per = (int)std::max<size_t>(1, budget / nvoices);
	const int nslabs = (nfrags + per - 1) / per;
	const bool two = nslabs > 1;
	const size_t nslots = nvoices * (size_t)per, cap = std::max<size_t>(nrec, 1);
	size_t slotwords = 0;		// of a slab: every list's slots at its class's size
	for(int j = 0; j < nj; ++j)
		if(!jobs[j].spec)
			slotwords += (size_t)jobs[j].n * (size_t)per * A2D_WIN_SLOTWORDS(jobs[j].nosc, jobs[j].filt);
	if(cap >= ((size_t)1 << 32))
		return c->fail(A2AMD_EUNSUPPORTED, "a batch of %zu records", cap);
	const size_t sets = two ? 2 : 1;
	if(sets * slotwords > c->d_win.cap || sets * cap > c->d_wext.cap || sets * nslots > c->d_widx.cap || nvoices > c->d_wrc.cap ||
			nfusedv * A2D_VMW_ROW > c->d_wscr.cap ||
			!c->d_wtop || (two && !c->win_stream)) {
		if(c->capturing)
			return c->fail(A2AMD_ESTATE, "window pool too small inside a graph capture");
		if(int r = grow(c, c->d_win, sets * slotwords, 1, false)) return r;
		if(int r = grow(c, c->d_wext, sets * cap, A2D_WIN_WORDS, false)) return r;
		if(int r = grow(c, c->d_widx, sets * nslots, 1, false)) return r;
		if(int r = grow(c, c->d_wrc, nvoices, 1, false)) return r;
		if(int r = grow(c, c->d_wscr, nfusedv * A2D_VMW_ROW, A2D_WIN_WORDS, false)) return r;
		if(!c->d_wtop) {
			HIPCHK(c, hipMalloc((void **)&c->d_wtop, 4 * sizeof(unsigned)));
			HIPCHK(c, hipMemsetAsync(c->d_wtop, 0, 4 * sizeof(unsigned), c->stream));
		}
		if(two && !c->win_stream) {
			HIPCHK(c, hipStreamCreateWithFlags(&c->win_stream, hipStreamNonBlocking));
			for(int k = 0; k < 5; ++k)
				HIPCHK(c, hipEventCreateWithFlags(&c->win_ev[k], hipEventDisableTiming));
		}
	}
	// (the sets are laid out by the CURRENT capacities: a graph captured earlier was dropped when they grew)
	const size_t half_win = c->d_win.cap / 2, half_ext = c->d_wext.cap / 2, half_idx = c->d_widx.cap / 2;
	hipStream_t sc = two ? c->win_stream : c->stream;		// the control passes' stream
	if(two) {
		// everything issued so far (uploads, the device VM's records, the quiet kernels) comes first
		HIPCHK(c, hipEventRecord(c->win_ev[4], c->stream));
		HIPCHK(c, hipStreamWaitEvent(sc, c->win_ev[4], 0));
	}
	// a list's control pass over fragments [fa, fb): k_win_ctl over its records, or - VM voices of a fused batch - k_vm_win
	uint32_t frames_to[A2D_MAXBATCH + 1];
	frames_to[0] = 0;
	for(int f = 0; f < nfrags; ++f)
		frames_to[f + 1] = frames_to[f] + (uint32_t)c->fragframes[f];
	auto control = [&](const Job &b, int fa, int fb, int *wslot, int *wext, unsigned *widx, unsigned *wtop, unsigned wcap, int *wrc,
			hipStream_t st) -> int {
		int r;
		if(b.spec) {
			// the speculative pass has done this job's control work: its results become the state (once: the first slab)
			if(fa != 0)
				return 0;
			A2DVmParams vp;
			vm_class_params(c, b.vmk, &vp);
			VmHost &m = c->vm;
			const A2DVmwOut from = { m.d_vmv_sh.d, m.d_ustate_sh.d, m.d_vactive_sh.d, m.d_runs_sh.d, m.d_stotal, nullptr };
			r = a2d_launch_vm_commit(vp, c->hparams, b.nosc, b.filt, from, st);
		} else if(b.vmk >= 0) {
			A2DVmParams vp;
			vm_class_params(c, b.vmk, &vp);
			size_t before = 0;	// (voices of the classes in front of this one: their rows of d_wscr)
			for(int q = 0; q < b.vmk; ++q)
				before += (size_t)c->vm.n_cls[q];
			int *const wscr = c->d_wscr.d + before * A2D_VMW_ROW * A2D_WIN_WORDS;
			r = a2d_launch_vm_win(vp, c->hparams, b.nosc, b.filt, fa, fb, vp.now + (frames_to[fa] << 8), vp.now + (frames_to[nfrags] << 8),
					wslot, wext, wscr, widx, wtop, wcap, st);
		} else
			r = a2d_launch_win_ctl(c->d_params, c->hparams, b.nosc, b.filt, b.list, b.n, b.skip, fa, fb, wslot, wext, widx, wtop, wcap, wrc, st);
		// (A2AMD_WIN_SYNC=1, debugging: wait for every launch and say so - the last line names the kernel that faulted)
		static const bool dbgsync = getenv("A2AMD_WIN_SYNC") != nullptr;
		if(dbgsync && !r && !c->capturing) {
			const hipError_t e = hipStreamSynchronize(st);
			fprintf(stderr, "a2amd windows: control pass <%d,%d> %s of %d voices, fragments [%d, %d): %s\n", b.nosc, b.filt,
					b.spec ? "k_vm_commit (speculative pass taken)" : b.vmk >= 0 ? "k_vm_win" : "k_win_ctl", b.n, fa, fb,
					hipGetErrorString(e));
		}
		return r;
	};
	static const bool wtiming = getenv("A2AMD_WIN_TIMING") != nullptr;
	static hipEvent_t tev[3] = { nullptr, nullptr, nullptr };
	// Several lists, one slab (a song: a few dozen voices of four classes): each list's control pass and render
	// pass are kernels of a few wavefronts that take as long as one voice's walk / one filter chain - on one stream
	// they would run back to back; here every list gets a stream of its own between a fork and a join (the render
	// passes only ever ADD to the buses).  A2AMD_WIN_FORK=0: one stream.
	static const bool fork_ok = !(getenv("A2AMD_WIN_FORK") && !atoi(getenv("A2AMD_WIN_FORK")));
	bool forked = false;
	if(nj > 1 && nslabs == 1 && fork_ok && !wtiming) {
		if(!c->win_fork[0]) {
			if(c->capturing)
				return c->fail(A2AMD_ESTATE, "window streams missing inside a graph capture");
			for(int j = 0; j < 7; ++j) {
				HIPCHK(c, hipStreamCreateWithFlags(&c->win_fork[j], hipStreamNonBlocking));
				HIPCHK(c, hipEventCreateWithFlags(&c->win_fev[j], hipEventDisableTiming));
			}
			HIPCHK(c, hipEventCreateWithFlags(&c->win_fev[7], hipEventDisableTiming));
		}
		if(plain_jobs)
			HIPCHK(c, hipMemsetAsync(c->d_wtop, 0, 2 * sizeof(unsigned), c->stream));
		HIPCHK(c, hipEventRecord(c->win_fev[7], c->stream));
		size_t at = 0, atw = 0, atv = 0;
		for(int j = 0; j < nj; ++j) {
			const Job &b = jobs[j];
			hipStream_t sj = c->win_fork[j];
			HIPCHK(c, hipStreamWaitEvent(sj, c->win_fev[7], 0));
			if(control(b, 0, nfrags, c->d_win.d + atw, c->d_wext.d, c->d_widx.d + at, c->d_wtop,
					(unsigned)std::min<size_t>(c->d_wext.cap, 0xffffffffu), c->d_wrc.d + atv, sj))
				return c->fail(A2AMD_EHIP, "window control launch failed: %s", hipGetErrorString(hipGetLastError()));
			if(b.spec ? a2d_launch_win_render(c->hparams, b.nosc, b.filt, b.list, b.n, 0, nfrags, c->vm.d_swin[sset].d + b.satw,
						c->vm.d_swext[sset].d, c->vm.d_swidx[sset].d + b.sat, sj) :
					a2d_launch_win_render(c->hparams, b.nosc, b.filt, b.list, b.n, 0, nfrags, c->d_win.d + atw,
						c->d_wext.d, c->d_widx.d + at, sj))
				return c->fail(A2AMD_EHIP, "window render launch failed: %s", hipGetErrorString(hipGetLastError()));
			HIPCHK(c, hipEventRecord(c->win_fev[j], sj));
			HIPCHK(c, hipStreamWaitEvent(c->stream, c->win_fev[j], 0));
			if(!b.spec) {
				at += (size_t)b.n * (size_t)nfrags;
				atw += (size_t)b.n * (size_t)nfrags * A2D_WIN_SLOTWORDS(b.nosc, b.filt);
				atv += (size_t)b.n;
			}
			c->stats.launches += 2;
		}
		forked = true;
	}
	int k = 0;
	for(int fa = 0; fa < nfrags && !forked; fa += per, ++k) {
		const int fb = std::min(nfrags, fa + per), set = two ? (k & 1) : 0;
		int *const wslot = c->d_win.d + (size_t)set * half_win;
		int *const wext = c->d_wext.d + (size_t)set * half_ext * A2D_WIN_WORDS;
		unsigned *const widx = c->d_widx.d + (size_t)set * half_idx;
		unsigned *const wtop = c->d_wtop + 2 * set;
		if(two && k >= 2)	// (the render pass that read this set has to be done with it)
			HIPCHK(c, hipStreamWaitEvent(sc, c->win_ev[2 + set], 0));
		if(wtiming && !two && !c->capturing) {
			for(int q = 0; q < 3; ++q)
				if(!tev[q])
					HIPCHK(c, hipEventCreate(&tev[q]));
			HIPCHK(c, hipEventRecord(tev[0], c->stream));
		}
		if(plain_jobs)
			HIPCHK(c, hipMemsetAsync(wtop, 0, 2 * sizeof(unsigned), sc));
		size_t at = 0, atw = 0, atv = 0;
		for(int j = 0; j < nj; ++j) {
			const Job &b = jobs[j];
			if(control(b, fa, fb, wslot + atw, wext, widx + at, wtop,
					(unsigned)std::min<size_t>(two ? half_ext : c->d_wext.cap, 0xffffffffu), c->d_wrc.d + atv, sc))
				return c->fail(A2AMD_EHIP, "window control launch failed: %s", hipGetErrorString(hipGetLastError()));
			if(b.spec)
				continue;	// (no slots of this batch's: the speculative pass's own)
			at += (size_t)b.n * (size_t)(fb - fa);
			atw += (size_t)b.n * (size_t)(fb - fa) * A2D_WIN_SLOTWORDS(b.nosc, b.filt);
			atv += (size_t)b.n;
			++c->stats.launches;
		}
		if(two) {
			HIPCHK(c, hipEventRecord(c->win_ev[set], sc));
			HIPCHK(c, hipStreamWaitEvent(c->stream, c->win_ev[set], 0));
		}
		if(wtiming && !two && !c->capturing)
			HIPCHK(c, hipEventRecord(tev[1], c->stream));
		// (one slab, one stream: the batch's other leaf kernels and the speculative pass for the next batch go here, the
		// render passes behind them - issue_kernels)
		if(nslabs == 1 && !wtiming) {
			if(int r = mid())
				return r;
			if(c->vm.spec_go_pending) {	// (the pass first, then the render pass: vm_speculate)
				HIPCHK(c, hipStreamWaitEvent(c->stream, c->vm.spec_go, 0));
				c->vm.spec_go_pending = false;
			}
		}
		at = atw = 0;
		for(int j = 0; j < nj; ++j) {
			const Job &b = jobs[j];
			if(b.spec) {
				// (the whole batch at once, behind the first slab's control passes - the commit among them)
				if(fa == 0 && a2d_launch_win_render(c->hparams, b.nosc, b.filt, b.list, b.n, 0, nfrags, c->vm.d_swin[sset].d + b.satw,
						c->vm.d_swext[sset].d, c->vm.d_swidx[sset].d + b.sat, c->stream))
					return c->fail(A2AMD_EHIP, "window render launch failed: %s", hipGetErrorString(hipGetLastError()));
				if(fa == 0)
					++c->stats.launches;
				continue;
			}
			if(a2d_launch_win_render(c->hparams, b.nosc, b.filt, b.list, b.n, fa, fb, wslot + atw,
					wext, widx + at, c->stream))
				return c->fail(A2AMD_EHIP, "window render launch failed: %s", hipGetErrorString(hipGetLastError()));
			at += (size_t)b.n * (size_t)(fb - fa);
			atw += (size_t)b.n * (size_t)(fb - fa) * A2D_WIN_SLOTWORDS(b.nosc, b.filt);
			++c->stats.launches;
		}
		if(two)
			HIPCHK(c, hipEventRecord(c->win_ev[2 + set], c->stream));
		if(wtiming && !two && !c->capturing) {
			// (A2AMD_WIN_TIMING=1: the two passes' times per slab on stderr - a measurement aid that waits for the GPU)
			float t_ctl = 0, t_ren = 0;
			HIPCHK(c, hipEventRecord(tev[2], c->stream));
			HIPCHK(c, hipEventSynchronize(tev[2]));
			hipEventElapsedTime(&t_ctl, tev[0], tev[1]);
			hipEventElapsedTime(&t_ren, tev[1], tev[2]);
			fprintf(stderr, "a2amd windows: fragments [%d, %d), %zu voices in %d list(s), %zu records: control pass %.1f us, render pass %.1f us\n",
					fa, fb, nvoices, nj, nrec, t_ctl * 1e3, t_ren * 1e3);
		}
	}
	// The pool's overflow flag (a bound the host got wrong: voices would lose windows) is never left unread: it is
	// copied back behind the batch and looked at before the next one's windows are issued - a batch late, but loud
	// (A2AMD_WIN_CHECK=1, the test suite's setting: at once, with a wait).
	if(!c->capturing && plain_jobs) {
		if(!c->h_wtop) {
			HIPCHK(c, hipHostMalloc((void **)&c->h_wtop, 4 * sizeof(unsigned), hipHostMallocDefault));
			memset(c->h_wtop, 0, 4 * sizeof(unsigned));
			HIPCHK(c, hipEventCreateWithFlags(&c->wtop_ev, hipEventDisableTiming));
		}
		HIPCHK(c, hipMemcpyAsync(c->h_wtop, c->d_wtop, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipEventRecord(c->wtop_ev, c->stream));
		c->wtop_pending = true;
	}
	if(c->vm.fused && !c->vm.spec_use)	// (a pass that met a fault is not taken: vm_issue)
		if(int r = vm_fused_done(c))
			return r;
	static const bool check = getenv("A2AMD_WIN_CHECK") != nullptr;
	if(check && !c->capturing && plain_jobs) {
		unsigned top[4] = { 0, 0, 0, 0 };
		HIPCHK(c, hipMemcpyAsync(top, c->d_wtop, sizeof(top), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		if(top[1] || (two && top[3])) {
			if(c->vm.fused)
				c->vm.fused_off = true;
			return c->fail(A2AMD_ESTATE, "window pool overflow (%u / %u of %zu entries)", top[0], top[2], c->d_wext.cap);
		}
	}
	return 0;
}

int issue_kernels(a2amd_ctx *c, unsigned phases, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2)
{
	// (self-cleaning buses need both phases in one go: the root's bus is read in ROOT)
	const bool has_sub = (phases & A2AMD_RENDER_SUBTREES) != 0, has_root = (phases & A2AMD_RENDER_ROOT) != 0;
	const bool consume = c->consume_ok && has_sub && has_root;
	// Split phases (multi-GPU steps): the group buses are read - and zeroed - in the
	// SUBTREES phase; the root's bus by the ROOT phase on the rank that runs it, or
	// by a2amd_rootbus_copy() when the partials are parked (shard.GroupedRootReduce).
	const bool consume_sub = c->consume_ok && has_sub && !has_root;
	A2DCommitSet pend;
	pend.n = 0;
	pend.c[0].nlist = pend.c[1].nlist = 0;
	auto flush_commits = [&]() {
		for(int k = 0; k < pend.n; ++k)
			a2d_launch_commit(c->hparams, pend.c[k], c->stream);
		pend.n = 0;
	};
	if(phases & A2AMD_RENDER_SUBTREES) {
		if(c->sub_resume < 0) {
		// (a graph of a self-cleaning batch holds no memset: whoever launches it
		// clears the buses first if they are not known to be clean, ensure_clean())
		const bool selfclean = consume || consume_sub;
		if(c->capturing ? !selfclean : !(selfclean && c->others_clean && c->root_clean))
			HIPCHK(c, hipMemsetAsync(c->d_busmem.d, 0, c->bus_used * sizeof(int32_t), c->stream));
		c->others_clean = selfclean;
		c->root_clean = consume;
		if(e0)
			HIPCHK(c, hipEventRecord(e0, c->stream));
		// the scripted voices the device runs itself: their VMs first - the records of this batch,
		// runs[] pointing at them (a2amd_vm.cpp / a2amd_vm.hip) - then the kernels as for host records
		// The voices that carry records this batch: the 2 x wtosc-filter12-panmix leaves, with and without
		// records, and - of the classes that have quiet kernels of their own - this batch's voices with records.
		// Round 5: the record stream resolved by a lane = voice control pass, the windows rendered
		// from closed-form entries (a2amd_win.hip).  A2AMD_WIN=0: k_leaf_recs, the kernels of rounds
		// 2-4 that interpret the records on the scalar unit of the rendering wavefront (A/B).
		// Which one: the window kernels' two passes are each as long as ONE voice's walk through the batch (the
		// control pass) / one filter chain (the render pass) whatever the voice count - 0.4 to 1 ms per 64 fragments -
		// where k_leaf_recs, one wavefront per voice, takes 0.3 ms for a song's few dozen voices and 0.4 - 0.7 ms for a
		// thousand; from a few thousand voices on it is k_leaf_recs that queues up (16 384: 1.1 - 3.9 ms against
		// 0.5 - 1.8).  A2AMD_WIN=0 / 1 forces (the parity tests run both), A2AMD_WIN_MIN moves the threshold.
		const int *rlists[4] = { c->d_dyn, c->d_dyn + c->n_dyn_osc1, c->d_dyn + c->n_dyn_osc1 + c->n_dyn_osc2,
				c->d_dyn + c->n_dyn_osc1 + c->n_dyn_osc2 + c->n_dyn_filt };
		// (the fourth list: this batch's record-carrying 2 x wtosc-filter12-panmix voices and the device VM's voices of that
		// class, taken with skip_empty - upload())
		const int rcounts[4] = { c->n_dyn_osc1, c->n_dyn_osc2, c->n_dyn_filt, c->n_dyn_filt2 };
		const int rtotal = rcounts[0] + rcounts[1] + rcounts[2] + rcounts[3];
		bool use_win;
		{
			const char *wenv = getenv("A2AMD_WIN");
			static const int win_min = getenv("A2AMD_WIN_MIN") ? atoi(getenv("A2AMD_WIN_MIN")) : 2048;
			int nwinv = rtotal;
			for(int k = 0; k < 3; ++k)
				nwinv += c->vm.list.empty() ? 0 : c->vm.n_cls[k];
			use_win = wenv ? atoi(wenv) != 0 : nwinv >= win_min;
		}
		// (A2AMD_VMWIN=0: the device VM's voices through records - k_vm_count / k_vm_emit - whatever renders them)
		static const bool vmwin_ok = !(getenv("A2AMD_VMWIN") && !atoi(getenv("A2AMD_VMWIN")));
		if(int r = vm_issue(c, use_win && vmwin_ok && !c->vm.fused_off))
			return r;
		// The batch's other leaf kernels - the quiet kernels of the classes, the records kernels where the window kernels
		// are not in use, the general kernel - as a block that runs once: normally behind the window kernels, and
		// (round 6) from INSIDE issue_windows, between its control passes and its render passes, when a speculative VM
		// pass is to follow: that pass may start as soon as the control passes (k_vm_win / k_vm_commit: the stepped state
		// of the VM voices that are live this batch) and the quiet kernels (the phases of those that are idle) are done,
		// and then has the render pass - the long one - beside it, not in front of it.  All of these kernels only ever
		// ADD to the buses, and runs[] is written by the control passes: their order among themselves is free.
		bool leaves_done = false;
		auto leaves = [&]() -> int {
			if(leaves_done)
				return 0;
			leaves_done = true;
			// A class whose quiet voices are ALL the device VM's, none of them left alone this batch - the speculative pass
			// that was taken for it counted (A2DVmwOut::idle) - has nothing for its quiet kernel to do, and a launch that
			// finds that out voice by voice is 10 us (k_leaf_oscpan + commit) to 80 us (k_leaf_oscfiltpan: sixteen
			// wavefronts per workgroup through the batch's fragments, barriers and all) in front of the next pass
			// (profiles/r06_timeline_before.txt).  The launch classes contain the VM's (every live voice of the class is in
			// the list, classify above): equal counts are equal sets.
			// Not for the filter class: there the 80 us happen to keep the order that is best for it - the render pass onto
			// the CUs first, the next pass behind it (vm_speculate) - and without them the two start together: 16 384 voices,
			// a2_Run(4096) 1 398 -> 1 416 us per buffer, a2_Run(1024) 399 -> 416; with the pass held back behind the render
			// pass instead 1 370 and 408 (profiles/r06_go_skip_ab.txt).  A2AMD_VMSKIP: bit 0 the classes without filter12,
			// bit 1 the one with.
			static const int vmskip = getenv("A2AMD_VMSKIP") ? atoi(getenv("A2AMD_VMSKIP")) : 1;
			auto none_quiet = [&](int k, int nleaf) {
				return (vmskip & (k == 2 ? 2 : 1)) && c->vm.spec_use && !c->vm.list.empty() && c->vm.n_cls[k] == nleaf && c->vm.spec_idle[k] == 0;
			};
			if(c->n_fast_leaf && none_quiet(0, c->n_fast_leaf)) {
				const bool solo = !c->n_osc2_leaf && !c->n_filt_leaf && !c->n_fm_leaf && !c->n_leaf && !c->n_leaf_dyn && !c->n_o2f_leaf;
				if(solo && e1)
					HIPCHK(c, hipEventRecord(e1, c->stream));
				++c->vm.quiet_skipped;
			} else if(c->n_fast_leaf) {
				int vpw, ysplit;
				pick_fast_shape(c->n_fast_leaf, c->nfrags, &vpw, &ysplit);
				// e1 right behind the main kernel when it is the only leaf kernel
				// of the batch: "leaf" time is then that kernel alone
				const bool solo = !c->n_osc2_leaf && !c->n_filt_leaf && !c->n_fm_leaf && !c->n_leaf && !c->n_leaf_dyn && !c->n_o2f_leaf;
				if(a2d_launch_leaf_oscpan(c->d_params, c->hparams, c->d_list.d, c->n_fast_leaf,
						vpw, ysplit, c->d_ustage.d, c->stream, solo ? (void *)e1 : nullptr, &pend.c[pend.n]))
					return c->fail(A2AMD_EHIP, "fast leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
				if(pend.c[pend.n].nlist)
					++pend.n;
				++c->stats.launches;
			}
			if(c->n_osc2_leaf && none_quiet(1, c->n_osc2_leaf))
				++c->vm.quiet_skipped;
			else if(c->n_osc2_leaf) {
				int vpw, ysplit;
				// (its own chunk length; 16 time slices: 2.19 ms against 2.25 with 32 at configs[3], round 3)
				pick_fast_shape(c->n_osc2_leaf, c->nfrags * A2D_FAST_FCH / A2D_OSC2_FCH, &vpw, &ysplit, 16);
				if(a2d_launch_leaf_osc2pan(c->d_params, c->hparams, c->d_list.d + c->n_fast_leaf, c->n_osc2_leaf,
						vpw, ysplit, c->d_ustage.d, c->stream, &pend.c[pend.n]))
					return c->fail(A2AMD_EHIP, "2-osc leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
				if(pend.c[pend.n].nlist)
					++pend.n;
				++c->stats.launches;
			}
			if(c->n_filt_leaf && none_quiet(2, c->n_filt_leaf))
				++c->vm.quiet_skipped;
			else if(c->n_filt_leaf) {
				// voices per workgroup = lanes of its filter wavefront: all 64 once there
				// are enough voices for a workgroup on every CU (one 16-wavefront workgroup
				// per CU), else spread out (a workgroup takes as long as its filter chain,
				// whatever its voice count)
				const int nf = c->n_filt_leaf;
				int vpw = getenv("A2AMD_FVPW") ? atoi(getenv("A2AMD_FVPW")) :
						std::min(std::max((nf + 255) / 256, 1), 64);
				if(a2d_launch_leaf_oscfiltpan(c->d_params, c->hparams, c->d_list.d + c->n_fast_leaf + c->n_osc2_leaf,
						c->n_filt_leaf, vpw, c->stream))
					return c->fail(A2AMD_EHIP, "filter leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
				++c->stats.launches;
			}
			if(c->n_o2f_leaf && c->o2f_quiet) {
				// 2 x wtosc-filter12-panmix without records (round 6).  A workgroup (16 wavefronts, 128 registers: one per CU)
				// takes as long as its filter wavefront's chain whatever its voice count, as long as every oscillator
				// wavefront stays in its all-settled loop (a2d_osc2filtpan_max_vpg voices): the voices are dealt over the
				// fewest whole rounds of 256 workgroups that allows (16 384 voices: 2 rounds of 32 - measured 1.09 ms per 256
				// fragments against 3.83 with 48 and 1.50 with 64 voices per workgroup)
				const int nf = c->n_o2f_leaf;
				static const int env_vpg = getenv("A2AMD_F2VPW") ? atoi(getenv("A2AMD_F2VPW")) : 0;
				const int maxv = a2d_osc2filtpan_max_vpg();
				const int rounds = std::max(1, (nf + 256 * maxv - 1) / (256 * maxv));
				const int vpg = env_vpg ? env_vpg : std::min(std::max((nf + 256 * rounds - 1) / (256 * rounds), 1), maxv);
				if(a2d_launch_leaf_osc2filtpan(c->d_params, c->hparams, c->d_list.d + c->n_fast_leaf + c->n_osc2_leaf +
						c->n_filt_leaf + c->n_fm_leaf + c->n_leaf, nf, vpg, c->stream))
					return c->fail(A2AMD_EHIP, "2-osc filter leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
				++c->stats.launches;
			}
			int fm_kinds = 0;
			for(int k = 0; k < 8; ++k)
				fm_kinds += c->fm_kind_count[k] != 0;
			if(fm_kinds > 1 && c->n_fm_leaf <= 16384 && !getenv("A2AMD_FMVPW")) {
				// several kinds, few voices: one launch for all of them (the
				// per-kind launches below would run back to back, each as long
				// as a voice's serial chain)
				if(a2d_launch_leaf_fmpan_all(c->d_params, c->hparams, c->d_list.d + c->n_fast_leaf + c->n_osc2_leaf +
						c->n_filt_leaf, c->fm_kind_count, (c->n_fm_leaf + 1023) / 1024, c->stream))
					return c->fail(A2AMD_EHIP, "fm leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
				++c->stats.launches;
			} else
			for(int k = 0, at = c->n_fast_leaf + c->n_osc2_leaf + c->n_filt_leaf; k < 8; at += c->fm_kind_count[k++]) {
				const int n = c->fm_kind_count[k];
				if(!n)
					continue;
				// A voice is a serial recurrence: a launch takes as long as its longest
				// lane, so few voices are spread over many wavefronts (idle lanes of a
				// wavefront shadow its voices, see fmpan_body) until there is one
				// wavefront per SIMD (1 024), then the lanes fill up
				// (profiles/r01_fm_vpw_sweep.txt).
				int vpw = getenv("A2AMD_FMVPW") ? atoi(getenv("A2AMD_FMVPW")) : (n + 1023) / 1024;
				vpw = std::min(std::max(vpw, 1), 64);
				if(a2d_launch_leaf_fmpan(c->d_params, c->hparams, A2AMD_FM1 + k, c->d_list.d + at, n, vpw, c->stream))
					return c->fail(A2AMD_EHIP, "fm leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
				++c->stats.launches;
			}
			if(c->n_leaf) {
				if(a2d_launch_voices(c->d_params, c->d_list.d + c->n_fast_leaf + c->n_osc2_leaf + c->n_filt_leaf +
						c->n_fm_leaf, c->n_leaf,
						pick_vpw(c->n_leaf), c->stream))
					return c->fail(A2AMD_EHIP, "leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
				++c->stats.launches;
			}
			{
				// (a wavefront walks its voices one after the other, fragment by fragment: as many
				// wavefronts as the chip holds before a wavefront gets a second voice)
				auto recs = [&](int nosc, int filt, const int *list, int n, int skip_empty) -> int {
					if(!n)
						return 0;
					int vpw = getenv("A2AMD_RVPW") ? atoi(getenv("A2AMD_RVPW")) : (n + 8191) / 8192;
					if(a2d_launch_leaf_recs(c->d_params, c->hparams, nosc, filt, list, n, vpw, c->stream, skip_empty))
						return c->fail(A2AMD_EHIP, "leaf records launch failed: %s", hipGetErrorString(hipGetLastError()));
					++c->stats.launches;
					return 0;
				};
				const int *const *lists = rlists;
				const int *counts = rcounts;
				const int total = rtotal;
				const int kinds = (counts[0] != 0) + (counts[1] != 0) + (counts[2] != 0) + (counts[3] != 0);
				if(!use_win) {
				if(kinds > 1 && total <= 4096 && !getenv("A2AMD_RVPW")) {
					// few voices of several kinds (a song): one launch - on one stream the per-kind
					// launches would run back to back, each as long as one voice's walk through the batch
					if(a2d_launch_leaf_recs_all(c->d_params, c->hparams, lists, counts, 1, c->stream, c->o2f_quiet ? 8 : 0))
						return c->fail(A2AMD_EHIP, "leaf records launch failed: %s", hipGetErrorString(hipGetLastError()));
					++c->stats.launches;
				} else {
					static const int nosc[4] = { 1, 2, 1, 2 }, filt[4] = { 0, 0, 1, 1 };
					for(int k = 0; k < 4; ++k)
						if(int r = recs(nosc[k], filt[k], lists[k], counts[k], k == 3 && c->o2f_quiet))
							return r;
				}
				// ... and the voices whose records the device VM has just written, by class (those it
				// left without records this batch were rendered by their quiet kernels above)
				if(!c->vm.list.empty()) {
					static const int nosc[3] = { 1, 2, 1 }, filt[3] = { 0, 0, 1 };
					const int *l = c->vm.d_list.d + c->vm.list.size();
					for(int k = 0; k < 3; l += c->vm.n_cls[k++]) {
						const int n = c->vm.n_cls[k];
						if(!n)
							continue;
						int vpw = getenv("A2AMD_RVPW") ? atoi(getenv("A2AMD_RVPW")) : (n + 8191) / 8192;
						if(a2d_launch_leaf_recs(c->d_params, c->hparams, nosc[k], filt[k], l, n, vpw, c->stream, 1))
							return c->fail(A2AMD_EHIP, "leaf records launch failed: %s", hipGetErrorString(hipGetLastError()));
						++c->stats.launches;
					}
				}
				}
			}
			if(c->n_dyn_rest > 0) {
				const int n = c->n_dyn_rest;
				if(a2d_launch_voices(c->d_params, c->d_dyn + c->n_dyn_osc1 + c->n_dyn_osc2 + c->n_dyn_filt + c->n_dyn_filt2, n,
						pick_vpw(n), c->stream))
					return c->fail(A2AMD_EHIP, "leaf launch failed: %s", hipGetErrorString(hipGetLastError()));
				++c->stats.launches;
			}
			return 0;
		};
		static const bool spec_early = !(getenv("A2AMD_VMSPEC_EARLY") && !atoi(getenv("A2AMD_VMSPEC_EARLY")));
		const std::function<int()> mid = [&]() -> int {
			if(!spec_early || !vmwin_ok || !vm_spec_wanted(c))
				return 0;	// (no pass to follow: the old order)
			if(int r = leaves())
				return r;
			// (a time-sliced quiet kernel leaves its voices' end state in the staging array until a commit that rides
			// along with a later launch: the pass reads the unit state of the class voices that were idle this batch)
			flush_commits();
			return vm_speculate(c);
		};
		// ... the window kernels first: k_vm_win says in runs[] which of the VM's voices are the quiet kernels' this batch
		if(use_win) {
			if(int r = issue_windows(c, rlists, rcounts, mid))
				return r;
			if(vmwin_ok)
				if(int r = vm_predict(c))
					return r;
		}
		if(int r = leaves())
			return r;
		if(e1 && !(c->n_fast_leaf && !c->n_osc2_leaf && !c->n_filt_leaf && !c->n_fm_leaf && !c->n_leaf && !c->n_leaf_dyn && !c->n_o2f_leaf))
			HIPCHK(c, hipEventRecord(e1, c->stream));
		// Round 6: behind the leaf kernels (the quiet ones have moved the phases of the VM voices that were idle this
		// batch), on a stream of its own beside the bus kernels, the readback and the engine thread's next walk:
		// the class voices' VM + control pass for the batch expected next (vm_speculate, a2amd_vm.cpp)
		if(use_win && vmwin_ok && !c->vm.spec_launched_now && vm_spec_wanted(c)) {
			flush_commits();	// (see mid(): the pass reads what the quiet kernels have staged)
			if(int r = vm_speculate(c))
				return r;
		}
		}	// (fresh start)
		// the voices that own a bus, deepest first.  With A2AMD_RENDER_TAPS the walk stops behind a
		// depth that holds a muted xinsert (insert clients: a2amd_unit_insertable): the host serves
		// them and calls again.
		const int dstart = c->sub_resume >= 0 ? c->sub_resume : (int)c->depth_ranges.size() - 1;
		c->sub_resume = -1;
		c->paused_at = 0;
		for(int d = dstart; d >= 1; --d) {
			if(int r = launch_depth(c, d, consume ? 3 : consume_sub ? 1 : 0, &pend))
				return r;
			if((phases & A2AMD_RENDER_TAPS) && depth_has_mutes(c, d)) {
				c->paused_at = d;
				c->sub_resume = d > 1 ? d - 1 : -1;
				break;
			}
		}
		// (the ROOT phase may run elsewhere, or later: nothing stays pending across calls)
		if(!(phases & A2AMD_RENDER_ROOT))
			flush_commits();
	}
	if(phases & A2AMD_RENDER_ROOT) {
		// The root chain adds into the master bus.  When its phase runs on its own
		// (multi-GPU steps: several SUBTREES phases may have gone by since the
		// last one, audiality2_amd/shard.py) the master bus is cleared here.
		// (... unless the root is a plain driver chain: then it stores the master bus)
		const bool root_stores = consume || c->consume_ok;
		if(!has_sub && !root_stores)
			HIPCHK(c, hipMemsetAsync(c->d_busmem.d, 0,
					(size_t)c->nfrags * c->cfg.channels * A2D_FRAG * sizeof(int32_t), c->stream));
		// (on its own the phase also leaves the root's bus zeroed behind it)
		if(!c->depth_ranges.empty())
			if(int r = launch_depth(c, 0, root_stores ? 3 : 0, &pend))
				return r;
		if(!has_sub && root_stores)
			c->root_clean = true;
		flush_commits();
		if(e2)
			HIPCHK(c, hipEventRecord(e2, c->stream));
		c->stats.fragments += c->nfrags;
		c->stats.voice_fragments += (uint64_t)c->nfrags * (uint64_t)(c->list_all.size() - c->n_list_pads);
	}
	return 0;
}

void drop_graphs(a2amd_ctx *c)
{
	for(int i = 0; i < 12; ++i) {
		if(c->gexec[i]) {
			hipGraphExecDestroy(c->gexec[i]);
			c->gexec[i] = nullptr;
		}
		if(c->graph[i]) {
			hipGraphDestroy(c->graph[i]);
			c->graph[i] = nullptr;
		}
	}
}

// before a graph of a self-cleaning batch (it holds no memset) is launched
int ensure_clean(a2amd_ctx *c)
{
	if(c->consume_ok && !(c->others_clean && c->root_clean)) {
		HIPCHK(c, hipMemsetAsync(c->d_busmem.d, 0, c->bus_used * sizeof(int32_t), c->stream));
		c->others_clean = c->root_clean = true;
	}
	return 0;
}

// capture 'steps' consecutive runs of the uploaded batch into one graph
int build_graph(a2amd_ctx *c, int slot, int steps, unsigned phases)
{
	if(c->gexec[slot]) {		// (built for another destination of the master bus)
		hipGraphExecDestroy(c->gexec[slot]);
		c->gexec[slot] = nullptr;
	}
	if(c->graph[slot]) {
		hipGraphDestroy(c->graph[slot]);
		c->graph[slot] = nullptr;
	}
	hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
	if(e != hipSuccess)
		return c->fail(A2AMD_EHIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
	int r = 0;
	// (a captured run does not happen now: every graph starts from buses of
	// unknown state, and what it leaves behind is noted when it is launched)
	const bool oc = c->others_clean, rc = c->root_clean;
	c->master_direct = false;
	c->capturing = true;
	for(int i = 0; i < steps && !r; ++i)
		r = issue_kernels(c, phases, nullptr, nullptr, nullptr);
	c->capturing = false;
	c->others_clean = oc;
	c->root_clean = rc;
	e = hipStreamEndCapture(c->stream, &c->graph[slot]);
	if(r)
		return r;
	if(e != hipSuccess)
		return c->fail(A2AMD_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
	e = hipGraphInstantiate(&c->gexec[slot], c->graph[slot], nullptr, nullptr, 0);
	if(e != hipSuccess)
		return c->fail(A2AMD_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
	c->gdst[slot] = c->master_dst;
	c->gdirect[slot] = c->master_direct;
	return 0;
}

} // namespace a2h
