"""Debugging aid: where does the host's wtosc phase shadow leave the oracle's phase?
   A2AMD_DEBUG_NOISE=1 python tools/dbg_walk.py gpu 2> g.txt; ... ora 2> o.txt; python tools/dbg_walk.py cmp g.txt o.txt"""
import sys, ctypes
sys.path.insert(0,'tests'); sys.path.insert(0,'.')

def run(be, t):
    uw, ui, vp, md = be.unit_write, be.unit_init, be.voice_process, be.mark_default
    ids = {}
    n = [0]
    def init(*a, **k):
        r = ui(*a, **k); ids[r] = n[0]; n[0] += 1; return r
    def v(units, off, cnt):
        sys.stderr.write(f"V {ids.get(units[0])} {off} {cnt}\n"); sys.stderr.flush(); return vp(units, off, cnt)
    def m(units):
        sys.stderr.write(f"M {ids.get(units[0])}\n"); sys.stderr.flush(); return md(units)
    def w(u, reg, val, start=0, dur=0, transpose=0):
        sys.stderr.write(f"W {ids.get(u)} {reg} {val} {start} {dur}\n"); sys.stderr.flush(); return uw(u, reg, val, start, dur, transpose)
    be.unit_init, be.unit_write, be.voice_process, be.mark_default = init, w, v, m
    return t._wt_script(be, "osc-pan", bfrags=16, batches=3, walk=True, noise=True)

if sys.argv[1] == "cmp":
    def load(p):
        out = []; cur = None
        for l in open(p):
            f = l.split()
            if f and f[0] in "VMW": cur = f
            elif f and f[0] == "NOISE": out.append((tuple(f[:7]), cur))
        return out
    a, b = load(sys.argv[2]), load(sys.argv[3])
    for i, (x, y) in enumerate(zip(a, b)):
        if x[0] != y[0]:
            print("first", i, x, y)
            u = x[1][1]
            hist = []
            for l in open(sys.argv[2]):
                f = l.split()
                if f and f[0] in "VMW" and f[1] in (u, str(int(u) + 1)): hist.append(l.strip())
                if f and f[0] == "NOISE" and tuple(f[:7]) == x[0]: break
            print("\n".join(hist[-50:]))
            break
else:
    import conftest
    import test_gpu_parity as t
    if sys.argv[1] == "gpu":
        g = t.make_gpu(max_batch=16); run(g, t); g.close()
    else:
        o = t.make_oracle(ctypes.CDLL(conftest.ORACLE_SO)); run(o, t); o.close()
