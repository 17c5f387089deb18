"""Call-trace reader and replayer.

A trace is the ordered list of calls the Audiality 2 engine made through its
unit plugin surface (Initialize / write / Process / Deinitialize, see
include/a2amd.h) while rendering a script, captured from the unmodified
reference (see tests/golden/).  Replaying it against a library that exports the
call protocol of include/a2amd.h under some symbol prefix ("a2amd_" = the GPU
library) must reproduce the audio the reference rendered.

Format (little endian): int32 records of 8 words {op,a,b,c,d,e,f,g}; a WAVE
record is followed by uint32 size[10] and the int16 payload of every level
(pads included), each level padded to a multiple of 4 bytes.
"""
import ctypes as C
import gzip
import lzma

import numpy as np

T_FRAGMENT, T_INIT, T_DEINIT, T_WRITE, T_PROCESS, T_INLINE_END, T_WAVE, T_CONFIG, T_WAVEDROP = range(1, 10)
MIPLEVELS = 10
WAVEPRE = 1
WAVEPOST = 131


class Trace:
    """Parsed trace: .config dict, .records (list of tuples), .waves."""

    def __init__(self, path):
        opener = gzip.open if str(path).endswith(".gz") else lzma.open if str(path).endswith(".xz") else open
        with opener(path, "rb") as f:
            raw = f.read()
        words = np.frombuffer(raw[: len(raw) // 4 * 4], dtype="<i4")
        self.records = []
        self.waves = {}
        self.config = None
        i = 0
        n = len(words)
        while i + 8 <= n:
            r = tuple(int(x) for x in words[i : i + 8])
            i += 8
            op = r[0]
            if op == T_WAVE:
                wid, wtype, flags, period, levels = r[1:6]
                sizes = words[i : i + MIPLEVELS].astype(np.uint32)
                i += MIPLEVELS
                data = []
                for lv in range(levels):
                    cnt = WAVEPRE + int(sizes[lv]) + WAVEPOST
                    nw = (cnt + 1) // 2
                    d = words[i : i + nw].view("<i2")[:cnt].copy()
                    i += nw
                    data.append(d)
                self.waves[wid] = dict(type=wtype, flags=flags & 0xFFFFFFFF, period=period,
                                       sizes=[int(s) for s in sizes], data=data)
                self.records.append((T_WAVE, wid, 0, 0, 0, 0, 0, 0))
            elif op == T_CONFIG:
                self.config = dict(samplerate=r[1], basepitch=r[2], channels=r[3],
                                   buffer=r[4], noiseseed=r[5] & 0xFFFFFFFF)
            else:
                self.records.append(r)
        if self.config is None:
            raise ValueError("trace without CONFIG record")
        self.kinds = {r[1]: r[3] for r in self.records if r[0] == T_INIT}

    @property
    def total_frames(self):
        return sum(r[1] for r in self.records if r[0] == T_FRAGMENT)


class a2amd_config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("samplerate", C.c_int32),
                ("basepitch", C.c_int32), ("channels", C.c_int32),
                ("device", C.c_int32), ("max_batch", C.c_uint32),
                ("stream", C.c_void_p)]


class a2amd_wavedesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("flags", C.c_uint32), ("period", C.c_uint32),
                ("size", C.c_uint32 * MIPLEVELS),
                ("data", C.POINTER(C.c_int16) * MIPLEVELS)]


class Backend:
    """Thin ctypes veneer over one implementation of the call protocol."""

    def __init__(self, lib, prefix, samplerate, basepitch, channels=2, device=0,
                 max_batch=64, stream=None):
        self.lib, self.prefix, self.channels = lib, prefix, channels

        def fn(name, restype, *argtypes):
            f = getattr(lib, prefix + name)
            f.restype, f.argtypes = restype, list(argtypes)
            return f

        vp, i32, u32, u64 = C.c_void_p, C.c_int, C.c_uint, C.c_uint64
        self._open = fn("open", i32, C.POINTER(a2amd_config), C.POINTER(vp))
        self._close = fn("close", None, vp)
        self._err = fn("last_error", C.c_char_p, vp)
        self._wave_upload = fn("wave_upload", i32, vp, u64, C.POINTER(a2amd_wavedesc))
        self._wave_drop = fn("wave_drop", i32, vp, u64)
        self._fragment = fn("fragment", i32, vp, u32)
        self._unit_init = fn("unit_init", i32, vp, u64, i32, u32, i32, i32, i32, i32, u32)
        self._unit_deinit = fn("unit_deinit", i32, vp, i32)
        self._unit_write = fn("unit_write", i32, vp, i32, i32, i32, u32, u32, i32)
        self._unit_process = fn("unit_process", i32, vp, i32, u32, u32, C.POINTER(C.c_uint32))
        self._inline_end = fn("inline_end", i32, vp, i32)
        self._unit_clients = fn("unit_clients", i32, vp, i32, u32)
        self._unit_inject = fn("unit_inject", i32, vp, i32, u32, u32, C.POINTER(C.POINTER(C.c_int32)))
        self._unit_tapped = fn("unit_tapped", i32, vp, i32, u32, C.POINTER(C.POINTER(C.c_int32)))
        self._render = fn("render", i32, vp, u32, C.POINTER(C.POINTER(C.c_int32)), u32)
        self._set_pt = fn("set_pitch_table", i32, vp, C.POINTER(C.c_uint32))
        self._get_pt = fn("get_pitch_table", i32, vp, C.POINTER(C.c_uint32))
        # the host walk's short cuts (the product library only; the checker has none)
        self.has_walk = hasattr(lib, prefix + "voice_process")
        if self.has_walk:
            self._voice_process = fn("voice_process", i32, vp, i32, u32, u32, C.POINTER(C.c_uint32))
            self._voice_slot = fn("voice_slot", i32, vp, i32)
            self._default_map = fn("default_map", C.POINTER(C.c_uint8), vp, C.POINTER(u32))
        cfg = a2amd_config(C.sizeof(a2amd_config), samplerate, basepitch, channels,
                           device, max_batch, stream)
        self.ctx = vp()
        rc = self._open(C.byref(cfg), C.byref(self.ctx))
        if rc != 0:
            raise RuntimeError(f"{prefix}open failed: {rc} "
                               f"{self._err(None).decode(errors='replace')}")
        self.noise = C.c_uint32(0)
        self._keep = []

    def close(self):
        if self.ctx:
            self._close(self.ctx)
            self.ctx = None

    def _chk(self, rc, what):
        if rc < 0:
            msg = self._err(self.ctx)
            raise RuntimeError(f"{self.prefix}{what}: {rc} {msg.decode(errors='replace') if msg else ''}")
        return rc

    def wave_upload(self, key, wtype, flags, period, sizes, data):
        d = a2amd_wavedesc()
        d.type, d.flags, d.period = wtype, flags, period
        for lv, arr in enumerate(data):
            arr = np.ascontiguousarray(arr, dtype=np.int16)
            self._keep.append(arr)
            d.size[lv] = sizes[lv]
            d.data[lv] = arr.ctypes.data_as(C.POINTER(C.c_int16))
        return self._chk(self._wave_upload(self.ctx, key, C.byref(d)), "wave_upload")

    def wave_drop(self, key):
        return self._chk(self._wave_drop(self.ctx, key), "wave_drop")

    def fragment(self, frames):
        return self._chk(self._fragment(self.ctx, frames), "fragment")

    def unit_init(self, voice_key, kind, flags, nin, nout, wired, transpose=0, wakefrac=0):
        return self._chk(self._unit_init(self.ctx, voice_key, kind, flags, nin, nout, wired,
                                         transpose, wakefrac), "unit_init")

    def unit_deinit(self, unit):
        return self._chk(self._unit_deinit(self.ctx, unit), "unit_deinit")

    def unit_write(self, unit, reg, value, start=0, dur=0, transpose=0):
        return self._chk(self._unit_write(self.ctx, unit, reg, value, start, dur, transpose),
                         "unit_write")

    def unit_process(self, unit, offset, frames, use_noise=True):
        return self._chk(self._unit_process(self.ctx, unit, offset, frames,
                                            C.byref(self.noise) if use_noise else None),
                         "unit_process")

    def voice_process(self, units, offset, frames):
        """One window of a whole voice (units = its chain): a2amd_voice_process where the
        library has it, else unit by unit.  Returns 1 when the voice's default windows may
        be reported through mark_default() from the next fragment on."""
        if self.has_walk:
            return self._chk(self._voice_process(self.ctx, units[0], offset, frames, C.byref(self.noise)),
                             "voice_process")
        for u in units:
            self.unit_process(u, offset, frames)
        return 0

    def mark_default(self, units):
        """The voice got exactly the default window in the open fragment: one byte store
        into the default map (a2amd_default_map)."""
        n = C.c_uint(0)
        m = self._default_map(self.ctx, C.byref(n))
        slot = self._chk(self._voice_slot(self.ctx, units[0]), "voice_slot")
        assert m and slot < n.value
        m[slot] = 1

    def inline_end(self, unit):
        return self._chk(self._inline_end(self.ctx, unit), "inline_end")

    def unit_clients(self, unit, mode):
        """mode: 1 = tap the unit's input for READ clients, 2 = take WRITE clients' output."""
        return self._chk(self._unit_clients(self.ctx, unit, mode), "unit_clients")

    def unit_inject(self, unit, offset, audio):
        """audio: int32 [channels, frames] produced by the WRITE clients for this window."""
        audio = np.ascontiguousarray(audio, dtype=np.int32)
        ptrs = (C.POINTER(C.c_int32) * audio.shape[0])()
        for c in range(audio.shape[0]):
            ptrs[c] = audio[c].ctypes.data_as(C.POINTER(C.c_int32))
        return self._chk(self._unit_inject(self.ctx, unit, offset, audio.shape[1], ptrs), "unit_inject")

    def unit_tapped(self, unit, fragment):
        """int32 [channels, 64]: what the unit's inputs carried in `fragment` of the last batch."""
        ptrs = (C.POINTER(C.c_int32) * 8)()
        n = self._chk(self._unit_tapped(self.ctx, unit, fragment, ptrs), "unit_tapped")
        return np.stack([np.ctypeslib.as_array(ptrs[c], shape=(64,)).copy() for c in range(n)])

    def render(self, capacity_frames, phases=15):
        out = np.zeros((self.channels, max(capacity_frames, 1)), dtype=np.int32)
        ptrs = (C.POINTER(C.c_int32) * self.channels)()
        for c in range(self.channels):
            ptrs[c] = out[c].ctypes.data_as(C.POINTER(C.c_int32))
        n = self._chk(self._render(self.ctx, phases, ptrs, capacity_frames), "render")
        return out[:, :n]

    def set_pitch_table(self, tab):
        tab = np.ascontiguousarray(tab, dtype=np.uint32)
        return self._chk(self._set_pt(self.ctx, tab.ctypes.data_as(C.POINTER(C.c_uint32))), "set_pt")

    def get_pitch_table(self):
        tab = np.zeros(128, dtype=np.uint32)
        self._chk(self._get_pt(self.ctx, tab.ctypes.data_as(C.POINTER(C.c_uint32))), "get_pt")
        return tab


def replay(trace, backend, batch=64, check_noise=True, max_fragments=None):
    """Feed `trace` to `backend`; returns int32 [channels, frames].

    `batch` fragments are recorded between renders.  When `check_noise` is set
    the engine-global noise state after every wtosc Process call is compared
    with what the reference had at that point.
    """
    umap, wmap = {}, {}
    outs = []
    nfrag = 0
    pending = 0
    backend.noise.value = trace.config["noiseseed"]
    for r in trace.records:
        op = r[0]
        if op == T_FRAGMENT:
            if max_fragments is not None and nfrag >= max_fragments:
                break
            if pending and nfrag % batch == 0:
                outs.append(backend.render(pending))
                pending = 0
            backend.fragment(r[1])
            pending += r[1]
            nfrag += 1
        elif op == T_WAVE:
            w = trace.waves[r[1]]
            wmap[r[1]] = backend.wave_upload(0x1000 + r[1], w["type"], w["flags"], w["period"],
                                             w["sizes"], w["data"])
        elif op == T_WAVEDROP:
            # the wave was released between two a2_Run() calls: what was
            # recorded so far plays it to the end
            if pending:
                outs.append(backend.render(pending))
                pending = 0
            backend.wave_drop(0x1000 + r[1])
            wmap.pop(r[1], None)
        elif op == T_INIT:
            _, uid, voice, kind, flags, io, transpose, wakefrac = r
            umap[uid] = backend.unit_init(voice, kind, flags & 0xFFFFFFFF, io & 0xFF,
                                          (io >> 8) & 0xFF, (io >> 16) & 1, transpose, wakefrac)
        elif op == T_DEINIT:
            backend.unit_deinit(umap.pop(r[1]))
        elif op == T_WRITE:
            _, uid, reg, value, start, dur, transpose, _g = r
            if trace.kinds.get(uid) == 0 and reg == 0:      # wtosc 'w'
                value = wmap.get(value, -1)
            backend.unit_write(umap[uid], reg, value, start & 0xFFFFFFFF, dur & 0xFFFFFFFF, transpose)
        elif op == T_PROCESS:
            _, uid, offset, frames, is_osc, before, after, _g = r
            if is_osc:
                backend.noise.value = before & 0xFFFFFFFF
            backend.unit_process(umap[uid], offset, frames)
            if is_osc and check_noise and backend.noise.value != (after & 0xFFFFFFFF):
                raise AssertionError(f"noise state diverged at unit {uid}: "
                                     f"{backend.noise.value:#x} != {after & 0xFFFFFFFF:#x}")
        elif op == T_INLINE_END:
            backend.inline_end(umap[r[1]])
    if pending:
        outs.append(backend.render(pending))
    return np.concatenate(outs, axis=1) if outs else np.zeros((backend.channels, 0), np.int32)


def read_pcm(path, channels, buffer):
    """PCM written by ref_tools: per a2_Run() call, `channels` planar blocks."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = np.frombuffer(f.read(), dtype="<i4")
    nfull = len(raw) // (channels * buffer)
    body = raw[: nfull * channels * buffer].reshape(nfull, channels, buffer)
    out = body.transpose(1, 0, 2).reshape(channels, nfull * buffer)
    rest = raw[nfull * channels * buffer:]
    if len(rest):
        out = np.concatenate([out, rest.reshape(channels, -1)], axis=1)
    return out
