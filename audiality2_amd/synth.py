"""Synthetic workloads for the tests and bench.py.

Host-side helpers that play the role of the Audiality 2 engine for a fixed,
scripted voice tree: they prepare wave data the way the engine's wave loader
does (src/waves.c:59-130: mip levels + pad samples) and issue the unit
callbacks (Initialize / write / Process) in the order the engine's voice walk
would (src/core.c:1847-1896) against any backend that speaks the call protocol
of include/a2amd.h.  Workload shapes follow BASELINE.json / SURVEY.md 8(d).
"""
import numpy as np

from .replay import WAVEPOST, WAVEPRE, MIPLEVELS

K_WTOSC, K_PANMIX, K_FILTER12, K_FBDELAY, K_INLINE, K_XINSERT = range(6)
# the FM oscillators (a2amd_unitkind): name -> (kind, operators)
FM_KINDS = {"fm1": (6, 1), "fm2": (7, 2), "fm3": (8, 3), "fm4": (9, 4),
            "fm3p": (10, 3), "fm4p": (11, 4), "fm2r": (12, 2), "fm4r": (13, 4)}
PROCADD = 1
WOFF, WNOISE, WWAVE, WMIPWAVE = range(4)
LOOPED = 0x100
WAVEPERIOD = 2048


def basepitch_for(samplerate):
    """A2_config.basepitch, src/audiality2.c:398: log2(261.626/rate) in 16:16."""
    f = np.float32(261.626) / np.float32(samplerate)
    p = np.float32(np.log2(np.float64(f)))
    return int(np.float32(p * np.float32(65536.0) + np.float32(0.5)))


def fix(x):
    """float -> 16:16 like the engine's API conversion (truncation)."""
    return int(x * 65536.0)


def wave_pyramid(samples, looped=True, levels=MIPLEVELS):
    """Mip pyramid with pads as the engine's wave loader builds it
    (a2_wave_alloc / a2_fix_pad / a2_render_mipmaps, src/waves.c:59-130).
    Returns (sizes, [int16 array per level incl. pads])."""
    src = np.asarray(samples, dtype=np.int16)
    length = len(src)
    sizes, data = [], []
    prev = None
    for lv in range(levels):
        size = (length + (1 << lv) - 1) >> lv
        buf = np.zeros(WAVEPRE + size + WAVEPOST, dtype=np.int16)
        if lv == 0:
            buf[WAVEPRE:WAVEPRE + size] = src
        else:
            p = prev.astype(np.int32)
            k = np.arange(size)
            # d[s] = (2*sd[2s] + sd[2s-1] + sd[2s+1]) >> 2, sd = previous payload
            buf[WAVEPRE:WAVEPRE + size] = ((p[WAVEPRE + 2 * k] << 1) + p[WAVEPRE + 2 * k - 1]
                                           + p[WAVEPRE + 2 * k + 1]) >> 2
        if looped and size:
            buf[0] = buf[size]
            idx = np.arange(WAVEPOST) % size
            buf[WAVEPRE + size:] = buf[WAVEPRE + idx]
        sizes.append(size)
        data.append(buf)
        prev = buf
    return sizes, data


def test_waves(count=24):
    """`count` band-rich looped single-cycle waves, period 2048 (the size and
    kind of the engine's built-in geometric waves; our own shapes)."""
    n = np.arange(WAVEPERIOD)
    out = []
    for i in range(count):
        kind = i % 4
        if kind == 0:      # pulse with duty depending on i
            duty = (i // 4 + 1) * WAVEPERIOD // 16
            w = np.where(n < duty, 32767, -32767)
        elif kind == 1:    # saw
            w = n * 65534 // WAVEPERIOD - 32767
        elif kind == 2:    # triangle
            t = np.where(n < WAVEPERIOD // 2, n, WAVEPERIOD - 1 - n)
            w = t * 65534 * 2 // WAVEPERIOD - 32767
        else:              # sine with i-dependent harmonic
            h = i // 4 + 1
            w = (np.sin(n * 2.0 * np.pi * h / WAVEPERIOD) * 32767.0).astype(np.int64)
        out.append(np.asarray(w, dtype=np.int16))
    return out


class Scene:
    """A scripted voice tree driven through the unit callbacks."""

    def __init__(self, backend, nwaves=24):
        self.be = backend
        self.next_key = 1
        self.wave_ids = []
        for i, w in enumerate(test_waves(nwaves)):
            sizes, data = wave_pyramid(w)
            self.wave_ids.append(backend.wave_upload(0x100 + i, WMIPWAVE, LOOPED, WAVEPERIOD,
                                                     sizes + [0] * (MIPLEVELS - len(sizes)), data))
        self.noise_id = backend.wave_upload(0x50, WNOISE, LOOPED, 256, [0] * MIPLEVELS, [])
        self.rootv = None
        self.groups = []          # each: dict(units=[inline, d1, d2], leaves=[...])
        self.leaves = []          # leaves directly under the root
        self.nvoices = 0

    def _key(self):
        k = self.next_key
        self.next_key += 1
        return k

    def private_waves(self, count, length=65536, period=256, seed=7):
        """`count` looped, mip-mapped sample waves of `length` samples each that no two voices need share -
        SURVEY 8(d)'s "private sample wave" case: the waves of a sampler, not the 24 built-in cycles that sit
        in L2.  Deterministic content (a seeded random walk, band-limited a little), uploaded like
        a2_UploadWave's result (src/waves.c:155-237: all mip levels, pads).  Voices made by add_voices(...,
        private=True) play wave (voice number mod count)."""
        rng = np.random.default_rng(seed)
        base = rng.integers(-3000, 3001, size=length + 4096, dtype=np.int32)
        base = np.cumsum(base)
        base -= (np.arange(len(base)) * (base[-1] // len(base)))     # (no drift)
        base = (base * (28000.0 / max(1, np.abs(base).max()))).astype(np.int16)
        self.private_ids = []
        for i in range(count):
            # every wave a different stretch / polarity / level of the walk: distinct data, cheap to make
            off = (i * 977) % 4096
            w = base[off:off + length].astype(np.int32)
            if i & 1:
                w = -w
            w = (w * (0.5 + 0.5 * ((i * 37) % 64) / 63.0)).astype(np.int16)
            w[-64:] = (w[-64:].astype(np.int32) * np.arange(63, -1, -1) // 64 + w[:64].astype(np.int32) * np.arange(64) // 64).astype(np.int16)
            sizes, data = wave_pyramid(w)
            self.private_ids.append(self.be.wave_upload(0x10000 + i, WMIPWAVE, LOOPED, period,
                                                        sizes + [0] * (MIPLEVELS - len(sizes)), data))
        return self.private_ids

    # -- tree construction (what a2_PopulateVoice + the voice's first VM run do) --
    def root(self, channels=2):
        """a2_rootdriver (audiality2.c:271-280): inline 0 *; panmix * *; xinsert * >"""
        be, k = self.be, self._key()
        self.rootv = [be.unit_init(k, K_INLINE, 0, 0, channels, 0),
                      be.unit_init(k, K_PANMIX, 0, channels, channels, 0),
                      be.unit_init(k, K_XINSERT, PROCADD, channels, channels, 1)]
        return self.rootv

    # the two delays of benchmark/fmtest4.a2s:83-95 (tempo 120 4: a tick is 125 ms):
    # (fbdelay, ldelay, rdelay) in ms and (fbgain, lgain, rgain)
    FMTEST4 = (((631.25, 756.25, 1003.75), (0.03, 0.05, 0.05)),
               ((868.75, 1126.25, 1378.75), (0.03, 0.05, 0.05)))

    def add_group(self, fb=(63.1, 75.6, 100.4), gains=(0.3, 0.25, 0.25), preset=None):
        """inline 0 *; fbdelay * *; fbdelay * >   (benchmark/fmtest4.a2s shape);
        preset="fmtest4": that song's own delay times and gains (BASELINE configs[3])"""
        be, k = self.be, self._key()
        u = [be.unit_init(k, K_INLINE, 0, 0, 2, 0),
             be.unit_init(k, K_FBDELAY, 0, 2, 2, 0),
             be.unit_init(k, K_FBDELAY, PROCADD, 2, 2, 1)]
        for i, d in enumerate((u[1], u[2])):
            dfb, dg = self.FMTEST4[i] if preset == "fmtest4" else (fb, gains)
            for reg, ms in enumerate(dfb):
                be.unit_write(d, reg, fix(ms))
            be.unit_write(d, 4, fix(dg[0]))
            be.unit_write(d, 5, fix(dg[1]))
            be.unit_write(d, 6, fix(dg[2]))
        g = dict(units=u, leaves=[])
        self.groups.append(g)
        return g

    def add_bus_group(self, parent=None):
        """A group voice as a2_NewGroup makes it (src/interface.c:888: the program
        a2_groupdriver, audiality2.c:292-302): inline 0 *; panmix * *; xinsert * > -
        under the root voice or under another such group."""
        be, k = self.be, self._key()
        u = [be.unit_init(k, K_INLINE, 0, 0, 2, 0),
             be.unit_init(k, K_PANMIX, 0, 2, 2, 0),
             be.unit_init(k, K_XINSERT, PROCADD, 2, 2, 1)]
        g = dict(units=u, leaves=[], subs=[])
        (self.groups if parent is None else parent.setdefault("subs", [])).append(g)
        return g

    def add_voices(self, n, chain="osc-pan", group=None, total=None, private=False):
        """n sustained voices with the per-voice parameters of SURVEY.md 8(d):
        wave 7k mod 24, pitch ((k mod 61)-30)/12 oct, pan ((k mod 17)-8)/8 ..."""
        be = self.be
        total = total or n
        amp = max(1, (4 << 16) // max(total, 1))
        dst = self.leaves if group is None else group["leaves"]
        for _ in range(n):
            k = self.nvoices
            self.nvoices += 1
            key = self._key()
            p = fix(((k % 61) - 30) / 12.0)
            units = []
            if chain == "osc-pan":
                units = [be.unit_init(key, K_WTOSC, 0, 0, 1, 0),
                         be.unit_init(key, K_PANMIX, PROCADD, 1, 2, 1)]
                oscs, pan = [units[0]], units[1]
            elif chain == "osc-filter-pan":
                units = [be.unit_init(key, K_WTOSC, 0, 0, 1, 0),
                         be.unit_init(key, K_FILTER12, 0, 1, 1, 0),
                         be.unit_init(key, K_PANMIX, PROCADD, 1, 2, 1)]
                oscs, pan = [units[0]], units[2]
                be.unit_write(units[1], 0, p + fix(2.0))      # cutoff
                be.unit_write(units[1], 1, fix(5.0))          # q
            elif chain == "osc2-filter-pan":
                units = [be.unit_init(key, K_WTOSC, 0, 0, 1, 0),
                         be.unit_init(key, K_WTOSC, PROCADD, 0, 1, 0),
                         be.unit_init(key, K_FILTER12, 0, 1, 1, 0),
                         be.unit_init(key, K_PANMIX, PROCADD, 1, 2, 1)]
                oscs, pan = units[:2], units[3]
                be.unit_write(units[2], 0, p + fix(2.5))      # cutoff
                be.unit_write(units[2], 1, fix(3.0))          # q
            elif chain == "osc2-pan":
                units = [be.unit_init(key, K_WTOSC, 0, 0, 1, 0),
                         be.unit_init(key, K_WTOSC, PROCADD, 0, 1, 0),
                         be.unit_init(key, K_PANMIX, PROCADD, 1, 2, 1)]
                oscs, pan = units[:2], units[2]
            elif chain.endswith("-pan") and (chain[:-4] in FM_KINDS or chain == "fmmix-pan"):
                # fmN; panmix.  "fmmix" cycles through the eight units.
                name = sorted(FM_KINDS)[k % 8] if chain == "fmmix-pan" else chain[:-4]
                kind, nops = FM_KINDS[name]
                units = [be.unit_init(key, kind, 0, 0, 1, 0, wakefrac=(k * 37) & 255),
                         be.unit_init(key, K_PANMIX, PROCADD, 1, 2, 1)]
                oscs, pan = [], units[1]
                fm = units[0]
                ring = name.endswith("r")
                be.unit_write(fm, 1, p)                                   # p
                # a (ramping in every fourth voice), fb
                be.unit_write(fm, 2, fix(1.0) if ring else amp)
                if k % 4 == 0:
                    be.unit_write(fm, 2, (fix(1.0) if ring else amp) // 2, 0, 3000 << 8)
                be.unit_write(fm, 3, fix((k % 5) / 8.0))
                ratios = (1.0 + (k % 3) * 0.5, 2.01, (k % 7) * 0.5 + 0.5)
                depths = (0.5 + (k % 4) * 0.25, 0.7, 0.4)
                fbs = ((k % 3) / 4.0, 0.2, (k % 2) * 0.5)
                for j in range(1, nops):
                    be.unit_write(fm, 1 + 3 * j, fix(ratios[j - 1]))      # pN (relative)
                    be.unit_write(fm, 2 + 3 * j, amp if (ring and j == 1) else fix(depths[j - 1]))
                    be.unit_write(fm, 3 + 3 * j, fix(fbs[j - 1]))
                if k % 8 == 3:
                    be.unit_write(fm, 1, p + fix(0.5), 0, 1500 << 8)      # pitch glide
                be.unit_write(fm, 0, (k * 2654435761) % 65536)            # phase
            else:
                raise ValueError(chain)
            for j, o in enumerate(oscs):
                be.unit_write(o, 0, self.private_ids[(k + j) % len(self.private_ids)] if private else
                              self.wave_ids[(7 * k + j) % len(self.wave_ids)])
                be.unit_write(o, 1, p + (fix(0.01) if j == 0 else -fix(0.01)) * (len(oscs) > 1))
                be.unit_write(o, 2, amp)
                be.unit_write(o, 3, (k * 2654435761) % 65536)
            be.unit_write(pan, 1, fix(((k % 17) - 8) / 8.0))
            dst.append(units)

    # -- the engine's per-fragment voice walk (core.c:1847-1896), all VMs asleep --
    def walk(self, frames=64):
        be = self.be
        be.fragment(frames)
        if self.rootv:
            be.unit_process(self.rootv[0], 0, frames)
        def walk_group(g):
            be.unit_process(g["units"][0], 0, frames)
            for sub in g.get("subs", ()):
                walk_group(sub)
            for units in g["leaves"]:
                for u in units:
                    be.unit_process(u, 0, frames)
            be.inline_end(g["units"][0])
            for u in g["units"][1:]:
                be.unit_process(u, 0, frames)

        for g in self.groups:
            walk_group(g)
        for units in self.leaves:
            for u in units:
                be.unit_process(u, 0, frames)
        if self.rootv:
            be.inline_end(self.rootv[0])
            be.unit_process(self.rootv[1], 0, frames)
            be.unit_process(self.rootv[2], 0, frames)

    def run(self, fragments, batch=64, frames=64):
        """Walk `fragments` fragments, rendering every `batch`; returns int32
        [channels, fragments*frames]."""
        outs, pending = [], 0
        for _ in range(fragments):
            self.walk(frames)
            pending += 1
            if pending == batch:
                outs.append(self.be.render(pending * frames))
                pending = 0
        if pending:
            outs.append(self.be.render(pending * frames))
        return np.concatenate(outs, axis=1)
