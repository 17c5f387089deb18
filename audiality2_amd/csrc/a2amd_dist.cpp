// a2amd_dist.cpp - multi-GPU: librccl bound at run time, the one exchange of a batch (ncclReduce of
// the root voice's inline bus over xGMI, int32 sum), one process per GPU (a2amd_dist_init) or several
// GPUs in one process (a2amd_dist_init_local / a2amd_render_group).  (Split out of a2amd_host.cpp
// in round 3.)
#include "a2amd_host.h"

namespace a2h {
Rccl g_rccl;


// RCCL is bound on first use: the copy the process already has (a host application
// that brought its own, e.g. PyTorch's) if there is one - two copies in one process
// would each run their own proxy threads and topology detection - else ROCm's
bool rccl_bind()
{
	if(g_rccl.Reduce)
		return true;
	const char *names[] = { "librccl.so.1", "librccl.so" };
	void *h = nullptr;
	for(const char *n : names)
		if(!h)
			h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
	for(const char *n : names)
		if(!h)
			h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
	if(!h)
		return false;
	g_rccl.lib = h;
	*(void **)&g_rccl.GetUniqueId = dlsym(h, "ncclGetUniqueId");
	*(void **)&g_rccl.CommInitRank = dlsym(h, "ncclCommInitRank");
	*(void **)&g_rccl.CommDestroy = dlsym(h, "ncclCommDestroy");
	*(void **)&g_rccl.GetErrorString = dlsym(h, "ncclGetErrorString");
	*(void **)&g_rccl.Reduce = dlsym(h, "ncclReduce");
	*(void **)&g_rccl.CommInitAll = dlsym(h, "ncclCommInitAll");
	*(void **)&g_rccl.GroupStart = dlsym(h, "ncclGroupStart");
	*(void **)&g_rccl.GroupEnd = dlsym(h, "ncclGroupEnd");
	return g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.Reduce;
}
} // namespace a2h

extern "C" {
// ---- multi-GPU: RCCL over xGMI, called from here (no framework in the data path) -------
int a2amd_dist_unique_id(void *id128)
{
	if(!id128 || !rccl_bind()) {
		snprintf(g_err, sizeof(g_err), "a2amd_dist_unique_id: RCCL (librccl.so) is not available");
		return A2AMD_ENODEVICE;
	}
	static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
	ncclResult_t r = g_rccl.GetUniqueId((ncclUniqueId *)id128);
	if(r != ncclSuccess) {
		snprintf(g_err, sizeof(g_err), "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
		return A2AMD_EHIP;
	}
	return A2AMD_OK;
}

int a2amd_dist_init(a2amd_ctx *c, const void *id128, int rank, int nranks)
{
	use_device(c);
	if(!id128 || rank < 0 || rank >= nranks)
		return c->fail(A2AMD_EINVAL, "dist_init: rank %d of %d", rank, nranks);
	if(c->comm)
		return c->fail(A2AMD_ESTATE, "dist_init: already initialised");
	if(!rccl_bind())
		return c->fail(A2AMD_ENODEVICE, "dist_init: RCCL (librccl.so) is not available");
	ncclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	// (RCCL checks the runtime's last-error slot as it goes: an error some earlier,
	// unrelated call of this process left there must not become its "unhandled cuda error")
	(void)hipGetLastError();
	ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
	if(r != ncclSuccess) {
		c->comm = nullptr;
		return c->fail(A2AMD_EHIP, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
	}
	c->dist_rank = rank;
	c->dist_ranks = nranks;
	drop_graphs(c);
	c->blob_quiet = false;
	return A2AMD_OK;
}

int a2amd_dist_init_local(a2amd_ctx *const *ctxs, int n)
{
	if(!ctxs || n < 1 || n > 64)
		return A2AMD_EINVAL;
	a2amd_ctx *c0 = ctxs[0];
	bool distinct = true;
	for(int i = 0; i < n; ++i) {
		if(ctxs[i]->comm || ctxs[i]->dist_local)
			return c0->fail(A2AMD_ESTATE, "dist_init_local: context %d is already part of a group", i);
		for(int k = 0; k < i; ++k)
			if(ctxs[k]->cfg.device == ctxs[i]->cfg.device)
				distinct = false;
	}
	if(n > 1 && distinct) {
		// one communicator per GPU, all in this process (ncclCommInitAll)
		if(!rccl_bind() || !g_rccl.CommInitAll || !g_rccl.GroupStart || !g_rccl.GroupEnd)
			return c0->fail(A2AMD_ENODEVICE, "dist_init_local: RCCL (librccl.so) is not available");
		std::vector<ncclComm_t> comms(n);
		std::vector<int> devs(n);
		for(int i = 0; i < n; ++i)
			devs[i] = ctxs[i]->cfg.device;
		(void)hipGetLastError();	// (see a2amd_dist_init)
		ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, devs.data());
		if(r != ncclSuccess)
			return c0->fail(A2AMD_EHIP, "ncclCommInitAll: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
		for(int i = 0; i < n; ++i)
			ctxs[i]->comm = comms[i];
	}
	// (contexts sharing a GPU - a test box - exchange with a device-local add instead:
	// RCCL refuses two ranks on one device)
	for(int i = 0; i < n; ++i) {
		a2amd_ctx *c = ctxs[i];
		use_device(c);
		c->dist_local = true;
		c->dist_rank = i;
		c->dist_ranks = n;
		if(!c->grp_ev)
			HIPCHK(c, hipEventCreateWithFlags(&c->grp_ev, hipEventDisableTiming));
		drop_graphs(c);
		c->blob_quiet = false;
	}
	return A2AMD_OK;
}

int a2amd_rootbus(a2amd_ctx *c, void **devptr, uint64_t *bytes);

int a2amd_render_group(a2amd_ctx *const *ctxs, int n, unsigned phases, int32_t *const *out, unsigned cap)
{
	if(!ctxs || n < 1)
		return A2AMD_EINVAL;
	a2amd_ctx *c0 = ctxs[0];
	if(n == 1 && !c0->dist_local)
		return a2amd_render(c0, phases, out, cap);
	for(int i = 0; i < n; ++i)
		if(!ctxs[i]->dist_local || ctxs[i]->dist_rank != i || ctxs[i]->dist_ranks != n)
			return c0->fail(A2AMD_ESTATE, "render_group: not the group of a2amd_dist_init_local()");
	const unsigned keep = phases & A2AMD_RENDER_KEEP;
	// every context: its commands up, its subtrees rendered (the kernels of the
	// different GPUs run side by side: nothing here waits)
	int frames = 0;
	for(int i = 0; i < n; ++i) {
		int r = a2amd_render(ctxs[i], (phases & (A2AMD_RENDER_UPLOAD | A2AMD_RENDER_SUBTREES | A2AMD_RENDER_EXCHANGE)) | keep, nullptr, 0);
		if(r < 0) {
			if(i)
				c0->fail(r, "%s", ctxs[i]->err);
			return r;
		}
		if(i == 0)
			frames = r;
		else if(r != frames)
			return c0->fail(A2AMD_ESTATE, "render_group: context %d recorded %d frames, context 0 %d", i, r, frames);
	}
	if(!frames)
		return 0;
	if(phases & (A2AMD_RENDER_SUBTREES | A2AMD_RENDER_EXCHANGE)) {
		// the exchange: the partials of the root voice's inline bus -> context 0
		std::vector<void *> bus(n);
		uint64_t bytes = 0;
		for(int i = 0; i < n; ++i) {
			uint64_t b;
			if(int r = a2amd_rootbus(ctxs[i], &bus[i], &b))
				return i ? c0->fail(r, "%s", ctxs[i]->err) : r;
			if(i && b != bytes)
				return c0->fail(A2AMD_ESTATE, "render_group: root buses of different shapes");
			bytes = b;
		}
		if(c0->comm) {
			// ONE ncclReduce(int32, sum) over xGMI: a group call, one thread drives all ranks
			ncclResult_t r = g_rccl.GroupStart();
			for(int i = 0; i < n && r == ncclSuccess; ++i) {
				use_device(ctxs[i]);
				r = g_rccl.Reduce(bus[i], bus[i], bytes / 4, ncclInt32, ncclSum, 0, ctxs[i]->comm, ctxs[i]->stream);
			}
			ncclResult_t r2 = g_rccl.GroupEnd();
			if(r != ncclSuccess || r2 != ncclSuccess)
				return c0->fail(A2AMD_EHIP, "ncclReduce: %s", g_rccl.GetErrorString ?
						g_rccl.GetErrorString(r != ncclSuccess ? r : r2) : "failed");
			for(int i = 1; i < n; ++i) {
				a2amd_ctx *c = ctxs[i];
				use_device(c);
				HIPCHK(c, hipMemsetAsync(bus[i], 0, bytes, c->stream));
				c->root_clean = true;
			}
		} else {
			// contexts sharing one GPU: context 0's stream adds the others' partials
			// (and clears them) once their subtrees are done
			use_device(c0);
			for(int i = 1; i < n; ++i) {
				a2amd_ctx *c = ctxs[i];
				HIPCHK(c, hipEventRecord(c->grp_ev, c->stream));
				HIPCHK(c0, hipStreamWaitEvent(c0->stream, c->grp_ev, 0));
				if(a2d_launch_add_bus((int32_t *)bus[0], (int32_t *)bus[i], (unsigned)(bytes / 4), c0->stream))
					return c0->fail(A2AMD_EHIP, "bus add launch failed");
				c->root_clean = true;
			}
			// (their next batch must not start before the partials were taken)
			HIPCHK(c0, hipEventRecord(c0->grp_ev, c0->stream));
			for(int i = 1; i < n; ++i)
				HIPCHK(ctxs[i], hipStreamWaitEvent(ctxs[i]->stream, c0->grp_ev, 0));
		}
	}
	// the root chain and the audio: context 0
	use_device(c0);
	int r = a2amd_render(c0, (phases & (A2AMD_RENDER_ROOT | A2AMD_RENDER_READBACK | A2AMD_RENDER_ASYNC)) | keep, out, cap);
	if(r < 0)
		return r;
	// sink clients on voices of the other contexts: their taps come back with the audio
	if((phases & A2AMD_RENDER_READBACK) && !(phases & A2AMD_RENDER_ASYNC))
		for(int i = 1; i < n; ++i) {
			use_device(ctxs[i]);
			if(int r2 = fetch_taps(ctxs[i], true))
				return c0->fail(r2, "%s", ctxs[i]->err);
		}
	if(!keep && (phases & (A2AMD_RENDER_READBACK | A2AMD_RENDER_ROOT)))
		for(int i = 1; i < n; ++i) {
			use_device(ctxs[i]);
			end_batch(ctxs[i]);
		}
	return frames;
}

// the exchange step of a batch: sum the ranks' partials of the root voice's inline bus
// into rank 0's (int32 wrap-around sum: any order gives the same bits)
} // extern "C"
int a2h::dist_reduce_root(a2amd_ctx *c)
{
	void *bus;
	uint64_t bytes;
	if(int r = a2amd_rootbus(c, &bus, &bytes))
		return r;
	ncclResult_t r = g_rccl.Reduce(bus, bus, bytes / 4, ncclInt32, ncclSum, 0, c->comm, c->stream);
	if(r != ncclSuccess)
		return c->fail(A2AMD_EHIP, "ncclReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
	if(c->dist_rank != 0) {
		// our partial has been delivered: the bus starts the next batch empty
		HIPCHK(c, hipMemsetAsync(bus, 0, bytes, c->stream));
		c->root_clean = true;
	}
	return A2AMD_OK;
}

extern "C" {
int a2amd_rootbus(a2amd_ctx *c, void **devptr, uint64_t *bytes)
{
	// the root voice is the depth-0 voice with an inline unit (a2_rootdriver,
	// audiality2.c:271-291)
	for(size_t vi = 0; vi < c->voices.size(); ++vi) {
		const HVoice &v = c->voices[vi];
		if(v.live && v.resolved && v.depth == 0 && v.own_off >= 0) {
			if(!c->d_busmem.d)
				return c->fail(A2AMD_ESTATE, "no batch uploaded yet");
			*devptr = c->d_busmem.d + v.own_off;
			*bytes = (uint64_t)c->nfrags * v.own_nch * A2D_FRAG * sizeof(int32_t);
			return A2AMD_OK;
		}
	}
	return c->fail(A2AMD_ESTATE, "no root voice with an inline bus");
}

int a2amd_rootbus_copy(a2amd_ctx *c, void *stage, int to_stage)
{
	void *bus;
	uint64_t bytes;
	use_device(c);
	if(int r = a2amd_rootbus(c, &bus, &bytes))
		return r;
	if(!stage)
		return c->fail(A2AMD_EINVAL, "no staging buffer");
	if(to_stage) {
		// park: copy out and leave the root's bus zeroed for the next SUBTREES phase
		if(a2d_launch_park((int32_t *)stage, (int32_t *)bus, (unsigned)(bytes / 4), c->stream))
			return c->fail(A2AMD_EHIP, "park launch failed: %s", hipGetErrorString(hipGetLastError()));
		c->root_clean = true;
	} else {
		HIPCHK(c, hipMemcpyAsync(bus, stage, bytes, hipMemcpyDeviceToDevice, c->stream));
		c->root_clean = false;
	}
	return A2AMD_OK;
}

} // extern "C"
