"""Numerical experiment (CPU, numpy): error of a 32-bit fixed-point form of the matrix path — effective rows and samples as four signed
8-bit digits each, products summed exactly in integers, digit-pair classes i + j <= KEEP kept — against the fp64 dot product, next to
the reference's own float loop.  Usage: python tools/sim/int8_scheme.py [columns]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import flush_schemes as S
T, F, P, Q, KC, bank = S.T, S.F, S.P, S.Q, S.KC, S.bank

def rows64(st):
    rows, ips = [], []
    for i in range(32):
        n = st * 32 + i
        off = T / 2 + n * Q / P
        ip = int(np.floor(off)); fr = (off - ip) * F; fi = int(np.floor(fr)); frac = fr - fi
        rows.append(bank[fi].astype(np.float64) * (1 - frac) + bank[fi + 1].astype(np.float64) * frac); ips.append(ip)
    shift = np.array(ips) - ips[0]
    K = ((T + shift.max() + 2 + KC - 1) // KC) * KC
    G = np.zeros((32, K))
    for i in range(32): G[i, shift[i]:shift[i] + T] = rows[i]
    return G, shift

def digits(q):
    """signed base-256 digits d0 (most significant) .. d3 of int64 array q (|q| < 2^31 - 2^23): sum d_i 256^(3-i) == q"""
    u = (q + 0x80808080) & 0xffffffff
    return [(((u >> (8 * (3 - i))) & 0xff) - 128).astype(np.int64) for i in range(4)]

def fixed_point(G, X, keep, hbits=30, xbits=30):
    hq = np.rint(G * 2.0 ** hbits).astype(np.int64); xq = np.rint(X.astype(np.float64) * 2.0 ** xbits).astype(np.int64)
    a, b = digits(hq), digits(xq)
    assert all((sum(d * 256 ** (3 - i) for i, d in enumerate(dd)) == q).all() for dd, q in ((a, hq), (b, xq)))
    y = np.zeros((G.shape[0], X.shape[1]))
    for i in range(4):
        for j in range(4):
            if i + j <= keep:
                y += (a[i] @ b[j]).astype(np.float64) * 2.0 ** (8 * (6 - i - j) - hbits - xbits)
    return y.astype(np.float32)

def run(signal, keep, N, seed=1, **kw):
    rng = np.random.default_rng(seed); errs, refs, f32s = [], [], []
    for st in range(5):
        G, shift = rows64(st); K = G.shape[1]
        X = signal(rng, K, N)
        truth = G @ X.astype(np.float64)
        errs.append((fixed_point(G, X, keep, **kw).astype(np.float64) - truth).ravel())
        Am = G.astype(np.float32)
        f32s.append((S.sched_current(Am, X, T // 2 - 7, T // 2 + shift.max() + 8).astype(np.float64) - truth).ravel())
        yr = np.zeros((32, N), np.float32)
        for i in range(32):
            h = Am[i, shift[i]:shift[i] + T]; x = X[shift[i]:shift[i] + T]
            acc = np.zeros(N, np.float32)
            for lo in range(T // 2):
                hi = T - 1 - lo
                acc = (acc + ((h[lo] * x[lo]).astype(np.float32) + (h[hi] * x[hi]).astype(np.float32))).astype(np.float32)
            yr[i] = acc
        refs.append((yr.astype(np.float64) - truth).ravel())
    e, r, f = np.concatenate(errs), np.concatenate(refs), np.concatenate(f32s)
    st = lambda v: (np.sqrt(np.mean(v ** 2)), np.abs(v).max())
    return st(e), st(f), st(r)

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    for sname, sig in S.SIGNALS.items():
        print(f"== {sname}")
        for keep in (3, 4, 6):
            (rms, mx), (frms, fmx), (rrms, rmx) = run(sig, keep, N)
            print(f"   classes i+j <= {keep}: rms {rms:.3e} max {mx:.3e} | f32 matrix kernel rms {frms:.3e} max {fmx:.3e} | reference float loop rms {rrms:.3e} max {rmx:.3e} | bar {2.0**-23:.3e}")
