"""The arithmetic behind the records kernels' filter window (a2amd_fast.hip: filt_window_j, RECS_JFILT).

filter12's per-frame step (the reference's filter12.c:97-118, one channel, 32-bit wrapping arithmetic) is a
recurrence in time: frame s needs d1 and d2 as frame s - 1 left them.  The kernel keeps frame s in lane s and lets
every lane recompute its frame from what its LEFT neighbour currently holds (DPP wavefront shift), all lanes at
once; lane 0 always reads the voice's state.  This file states in numpy - no GPU, no library - why that is the same
thing: after k such sweeps lanes 0..k hold exactly the sequential results, a lane that is right stays right, and
sweeps past len - 1 change nothing (the kernel rounds the count up to its unroll factor).  The GPU tests compare
the kernel itself with the oracle; this one pins the method, with the cutoff and q ramps in their closed per-lane
form and values large enough to wrap.
"""
import numpy as np
import pytest

I32 = np.int32


def w(x):
    """wrap to int32 (the engine's arithmetic is C int on two's complement hardware)"""
    return np.asarray(x, dtype=np.int64).astype(np.uint32).astype(I32)


def wmul(a, b):
    return w(np.asarray(a, dtype=np.int64) * np.asarray(b, dtype=np.int64))


def sequential(x, f0, df, qv, qd, d1, d2):
    """frame by frame, as f12_process does: returns per-frame l, b, h and the final (d1, d2, f0, qv)"""
    n = len(x)
    L, B, H = np.zeros(n, I32), np.zeros(n, I32), np.zeros(n, I32)
    d1, d2, f0, qv = I32(d1), I32(d2), I32(f0), I32(qv)
    for s in range(n):
        f, q = f0 >> 12, qv >> 12
        d1s = d1 >> 4
        l = w(int(d2) + int(wmul(f, d1s) >> 8))
        h = w(int(x[s] >> 5) - int(l) - int(wmul(q, d1s) >> 8))
        b = w(int(wmul(f, h >> 4) >> 8) + int(d1))
        L[s], B[s], H[s] = l, b, h
        d1, d2 = b, l
        f0, qv = w(int(f0) + int(df)), w(int(qv) + int(qd))
    return L, B, H, (d1, d2, f0, qv)


def sweep(X, F, Q, L, B, d1i, d2i):
    """one step of the kernel's loop: every lane from its left neighbour's b / l, lane 0 from the state"""
    bsh = np.concatenate(([I32(d1i)], B[:-1]))
    lsh = np.concatenate(([I32(d2i)], L[:-1]))
    ds = bsh >> 4
    Ln = w(lsh.astype(np.int64) + (wmul(F, ds) >> 8))
    Hn = w(X.astype(np.int64) - Ln - (wmul(Q, ds) >> 8))
    Bn = w((wmul(F, Hn >> 4) >> 8).astype(np.int64) + bsh)
    return Ln, Bn, Hn


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("n", [1, 2, 5, 17, 63, 64])
def test_sweeps_along_the_lanes_reproduce_the_recurrence(seed, n):
    rng = np.random.default_rng(1000 * n + seed)
    big = seed >= 3         # (half of the cases with magnitudes whose products wrap)
    x = rng.integers(-2**(30 if big else 22), 2**(30 if big else 22), n).astype(I32)
    f0 = I32(rng.integers(0, 2**(31 if big else 28)))
    df = I32(rng.integers(-2**20, 2**20)) if seed % 2 else I32(0)
    qv = I32(rng.integers(2**12, 2**(30 if big else 24)))
    qd = I32(rng.integers(-2**14, 2**14)) if seed % 3 else I32(0)
    d1, d2 = (I32(v) for v in rng.integers(-2**(30 if big else 24), 2**(30 if big else 24), 2))
    Ls, Bs, Hs, (e1, e2, ef, eq) = sequential(x, f0, df, qv, qd, d1, d2)

    lanes = np.arange(64, dtype=np.int64)
    X = np.zeros(64, I32)
    X[:n] = x >> 5          # (lanes past the window hold whatever: their results are never read)
    X[n:] = rng.integers(-2**20, 2**20, 64 - n)
    F = w(int(f0) + int(df) * lanes) >> 12
    Q = w(int(qv) + int(qd) * lanes) >> 12
    # step 0: every lane from the voice's state - right for lane 0
    L = np.full(64, d2, I32)
    B = np.full(64, d1, I32)
    L, B, H = sweep(X, F, Q, L, B, d1, d2)
    L0 = w(int(d2) + int(wmul(F[0], d1 >> 4) >> 8))
    assert L[0] == L0 == Ls[0] and B[0] == Bs[0]
    steps = ((n - 1 + 3) // 4) * 4 if n - 1 <= 4 else ((n - 1 + 7) // 8) * 8   # the kernel's rounding (4- and 8-step blocks)
    for k in range(1, max(steps, n - 1) + 1):
        L, B, H = sweep(X, F, Q, L, B, d1, d2)
        upto = min(k, n - 1)
        assert np.array_equal(L[:upto + 1], Ls[:upto + 1]), (k, "l")
        assert np.array_equal(B[:upto + 1], Bs[:upto + 1]), (k, "b")
        assert np.array_equal(H[:upto + 1], Hs[:upto + 1]), (k, "h")
    # what the kernel takes back: the state after the window's last frame, the ramps after len frames
    assert B[n - 1] == e1 and L[n - 1] == e2
    assert w(int(f0) + int(df) * n) == ef and w(int(qv) + int(qd) * n) == eq
