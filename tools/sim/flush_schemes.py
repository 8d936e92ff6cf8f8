"""Numerical experiment (CPU, numpy): accumulated rounding error of the matrix-core kernel's f32 MFMA chain under different
fp64-flush schedules, against the fp64 dot product, next to the reference's own float loop (outside-in pairs).
Emulates v_mfma_f32_32x32x2_f32 as a chain of correctly rounded f32 FMAs in K order.  Usage: python tools/sim/flush_schemes.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise

T = F = 988; P, Q = 160, 147; KC = 32
L = A.lib()
bank = np.zeros((F + 1, T), np.float32)
L.artamdBuildFilterBank(T, F, 1.0, A.BLACKMAN_HARRIS, bank.ctypes.data_as(C.POINTER(C.c_float)))

def fma32(acc, a, b):
    return (acc.astype(np.float64) + a.astype(np.float64) * b.astype(np.float64)).astype(np.float32)

def rows_for_tile(st):
    """effective rows of slot tile st, shifted to the tile's K origin: returns A[32, K], shifts"""
    rows, ips = [], []
    for i in range(32):
        n = st * 32 + i
        off = T / 2 + n * Q / P            # (any base; only the fractional structure matters)
        ip = int(np.floor(off)); fr = (off - ip) * F; fi = int(np.floor(fr)); frac = fr - fi
        g = (bank[fi].astype(np.float64) * (1 - frac) + bank[fi + 1].astype(np.float64) * frac).astype(np.float32)
        rows.append(g); ips.append(ip)
    shift = np.array(ips) - ips[0]
    K = ((T + shift.max() + 2 + KC - 1) // KC) * KC
    Am = np.zeros((32, K), np.float32)
    for i in range(32):
        Am[i, shift[i]:shift[i] + T] = rows[i]
    return Am, shift

def simulate(signal, schedule_fn, N=4096, seed=1):
    rng = np.random.default_rng(seed)
    errs, refs = [], []
    for st in range(5):
        Am, shift = rows_for_tile(st)
        K = Am.shape[1]
        X = signal(rng, K, N)                     # [K, N] float32: column = one output's window in tile coordinates
        truth = Am.astype(np.float64) @ X.astype(np.float64)          # [32, N]
        band_lo, band_hi = T // 2 - 1 - 6, T // 2 + shift.max() + 2 + 6
        y = schedule_fn(Am, X, band_lo, band_hi)
        errs.append((y.astype(np.float64) - truth).ravel())
        # the reference's float loop: outside-in pairs over each row's own T taps
        yr = np.zeros((32, N), np.float32)
        for i in range(32):
            h = Am[i, shift[i]:shift[i] + T]; x = X[shift[i]:shift[i] + T]
            acc = np.zeros(N, np.float32)
            for lo in range(T // 2):
                hi = T - 1 - lo
                pair = (h[lo] * x[lo]).astype(np.float32) + (h[hi] * x[hi]).astype(np.float32)
                acc = (acc + pair.astype(np.float32)).astype(np.float32)
            yr[i] = acc
        refs.append((yr.astype(np.float64) - truth).ravel())
    e, r = np.concatenate(errs), np.concatenate(refs)
    return np.sqrt(np.mean(e ** 2)), np.abs(e).max(), np.sqrt(np.mean(r ** 2)), np.abs(r).max()

def chain(Am, X, ks, acc=None):
    """f32 FMA chain over K columns ks (in that order) for all 32 rows x N columns"""
    if acc is None:
        acc = np.zeros((32, X.shape[1]), np.float32)
    for k in ks:
        acc = fma32(acc, Am[:, k:k + 1], X[k:k + 1, :])
    return acc

def kernel_order(c0):
    """K order inside a chunk as the kernel walks it: groups of 8, lanes 0-31 take k 0-3, lanes 32-63 k 4-7 (a fixed permutation:
    each MFMA sums k and k+4 ... emulated as the sequence 0,4,1,5,2,6,3,7 within the group)"""
    out = []
    for g in range(0, KC, 8):
        out += [c0 + g + q for q in (0, 4, 1, 5, 2, 6, 3, 7)]
    return out

def sched_current(Am, X, band_lo, band_hi, pair_taps=32, quad_taps=160):
    K = Am.shape[1]; nch = K // KC
    lo_b, hi_b = band_lo // KC, (band_hi + KC - 1) // KC
    total = np.zeros((32, X.shape[1]), np.float64)
    c = 0
    def share(c):
        d = min(abs(c - lo_b) if c < lo_b else 1e9, abs(c - (hi_b - 1)) if c >= hi_b else 1e9) if not (lo_b <= c < hi_b) else 0
        dist = (lo_b - c - 1) * KC if c < lo_b else (c - hi_b) * KC if c >= hi_b else -1
        return 4 if dist >= quad_taps else 2 if dist >= pair_taps else 1
    while c < nch:
        inband = c * KC < band_hi and c * KC + KC > band_lo
        if inband:
            ks = kernel_order(c * KC)
            for g in range(0, KC, 4):            # flush every 4 k (two MFMAs)
                total += chain(Am, X, ks[g:g + 4]).astype(np.float64)
            c += 1
        else:
            n = share(c)
            n = min(n, nch - c)
            while n > 1 and any((cc * KC < band_hi and cc * KC + KC > band_lo) for cc in range(c, c + n)):
                n //= 2
            acc = None
            for cc in range(c, c + n):
                acc = chain(Am, X, kernel_order(cc * KC), acc)
            total += acc.astype(np.float64)
            c += n
    return total.astype(np.float32)

def make_outside_in(near_chunks=1, tail_flush_every=None, split_sides=False, right_ascending=False):
    def sched(Am, X, band_lo, band_hi):
        K = Am.shape[1]; nch = K // KC
        lo_b, hi_b = band_lo // KC, (band_hi + KC - 1) // KC
        near = list(range(max(0, lo_b - near_chunks), min(nch, hi_b + near_chunks)))
        left = list(range(0, near[0])); right = list(range(nch - 1, near[-1], -1))
        if right_ascending: right = right[::-1]
        total = np.zeros((32, X.shape[1]), np.float64)
        if split_sides:
            for side in (left, right):
                acc = None
                for j, cc in enumerate(side):
                    acc = chain(Am, X, kernel_order(cc * KC), acc)
                    if tail_flush_every and (j + 1) % tail_flush_every == 0:
                        total += acc.astype(np.float64); acc = None
                if acc is not None: total += acc.astype(np.float64)
        else:
            order = []
            l, r = left[:], right[:]
            while l or r:                     # farthest first, alternating sides
                if l: order.append(l.pop(0))
                if r: order.append(r.pop(0))
            acc = None
            for j, cc in enumerate(order):
                acc = chain(Am, X, kernel_order(cc * KC), acc)
                if tail_flush_every and (j + 1) % tail_flush_every == 0:
                    total += acc.astype(np.float64); acc = None
            if acc is not None: total += acc.astype(np.float64)
        for cc in near:
            inband = cc * KC < band_hi and cc * KC + KC > band_lo
            ks = kernel_order(cc * KC)
            if inband:
                for g in range(0, KC, 4):
                    total += chain(Am, X, ks[g:g + 4]).astype(np.float64)
            else:
                total += chain(Am, X, ks).astype(np.float64)
        return total.astype(np.float32)
    return sched

SIGNALS = {
    "noise +-0.5 (tests, bench)": lambda rng, K, N: (rng.random((K, N)) - 0.5).astype(np.float32),
    "noise +-1": lambda rng, K, N: (2 * rng.random((K, N)) - 1).astype(np.float32),
    "full-scale sine f=0.013": lambda rng, K, N: np.sin(2 * np.pi * 0.013 * (np.arange(K)[:, None] + rng.integers(0, 10000, N)[None, :])).astype(np.float32),
    "full-scale square p=37": lambda rng, K, N: np.where(((np.arange(K)[:, None] + rng.integers(0, 10000, N)[None, :]) // 37) % 2 == 0, 1.0, -1.0).astype(np.float32),
    "DC 1.0": lambda rng, K, N: np.ones((K, N), np.float32),
}
SCHEMES = {
    "current (band/4k, singles, pairs, quads)": sched_current,
    "outside-in, ONE f32 acc for all tails, near=1": make_outside_in(1),
    "outside-in, ONE f32 acc, near=2": make_outside_in(2),
    "outside-in, flush every 8 tail chunks, near=1": make_outside_in(1, 8),
    "per side in->out... (left asc, right desc) one acc each, near=1": make_outside_in(1, None, True),
    "per side, flush every 6, near=1": make_outside_in(1, 6, True),
    "IN ORDER: left asc one acc, right asc (inside-out) one acc, near=1": make_outside_in(1, None, True, True),
    "IN ORDER, near=0 (only the band chunks flushed)": make_outside_in(0, None, True, True),
    "outside-in per side, near=0": make_outside_in(0, None, True, False),
    "IN ORDER, right side flushed every 4 chunks, near=1": None,
}
def _inorder_right4(Am, X, band_lo, band_hi):
    K = Am.shape[1]; nch = K // KC
    lo_b, hi_b = band_lo // KC, (band_hi + KC - 1) // KC
    total = np.zeros((32, X.shape[1]), np.float64)
    acc = None
    for cc in range(0, max(0, lo_b - 1)):
        acc = chain(Am, X, kernel_order(cc * KC), acc)
    if acc is not None: total += acc.astype(np.float64)
    for cc in range(max(0, lo_b - 1), min(nch, hi_b + 1)):
        inband = cc * KC < band_hi and cc * KC + KC > band_lo
        ks = kernel_order(cc * KC)
        if inband:
            for g in range(0, KC, 4): total += chain(Am, X, ks[g:g + 4]).astype(np.float64)
        else: total += chain(Am, X, ks).astype(np.float64)
    acc = None; j = 0
    for cc in range(min(nch, hi_b + 1), nch):
        acc = chain(Am, X, kernel_order(cc * KC), acc); j += 1
        if j % 4 == 0: total += acc.astype(np.float64); acc = None
    if acc is not None: total += acc.astype(np.float64)
    return total.astype(np.float32)
SCHEMES["IN ORDER, right side flushed every 4 chunks, near=1"] = _inorder_right4
for k in list(SCHEMES)[1:6]: del SCHEMES[k]
def make_inorder(band_step=4, trim=False, right_every=4, near=1):
    def sched(Am, X, band_lo, band_hi):
        K = Am.shape[1]; nch = K // KC
        lo_b, hi_b = band_lo // KC, (band_hi + KC - 1) // KC
        total = np.zeros((32, X.shape[1]), np.float64)
        acc = None
        for cc in range(0, max(0, lo_b - near)):
            acc = chain(Am, X, kernel_order(cc * KC), acc)
        if acc is not None: total += acc.astype(np.float64)
        for cc in range(max(0, lo_b - near), min(nch, hi_b + near)):
            inband = cc * KC < band_hi and cc * KC + KC > band_lo
            ks = kernel_order(cc * KC)
            if inband and not trim:
                for g in range(0, KC, band_step): total += chain(Am, X, ks[g:g + band_step]).astype(np.float64)
            elif inband:
                # only the 8-k groups that intersect the band are flushed finely; the rest of the chunk shares one accumulator
                acc = None
                for g8 in range(0, KC, 8):
                    k0 = cc * KC + g8
                    grp = ks[g8:g8 + 8]
                    if k0 < band_hi and k0 + 8 > band_lo:
                        for g in range(0, 8, band_step): total += chain(Am, X, grp[g:g + band_step]).astype(np.float64)
                    else:
                        acc = chain(Am, X, grp, acc)
                if acc is not None: total += acc.astype(np.float64)
            else: total += chain(Am, X, ks).astype(np.float64)
        acc = None; j = 0
        for cc in range(min(nch, hi_b + near), nch):
            acc = chain(Am, X, kernel_order(cc * KC), acc); j += 1
            if right_every and j % right_every == 0: total += acc.astype(np.float64); acc = None
        if acc is not None: total += acc.astype(np.float64)
        return total.astype(np.float32)
    return sched
SCHEMES = {"current (band/4k, singles, pairs, quads)": sched_current,
           "in order, band 4k, right/4": make_inorder(4),
           "in order, band 4k TRIMMED to the band's 8-k groups": make_inorder(4, True),
           "in order, band 8k": make_inorder(8),
           "in order, band 8k trimmed": make_inorder(8, True),
           "in order, band 2k": make_inorder(2),
           "in order, band 4k, near=2": make_inorder(4, False, 4, 2),
           }

def make_row_windows(margin_lo=6, margin_hi=7, lead=4, near=1, right_every=4, quad_rows=8):
    """round 4: ONE accumulator through the band; behind every MFMA pair (4 k of an 8-k group) only the register quads whose rows
    (8 consecutive slots: 4 per half-wave) have a central tap within the group's 8 k's — [centre - margin_lo - lead, centre + margin_hi) — are
    flushed and cleared; the other rows' sums ride on (they are tail sums there)"""
    def sched(Am, X, band_lo, band_hi):
        K = Am.shape[1]; nch = K // KC
        lo_b, hi_b = band_lo // KC, (band_hi + KC - 1) // KC
        # the rows' centres from the matrix itself (first non-zero column + T/2 - 1)
        shift = np.array([np.flatnonzero(Am[i])[0] for i in range(32)]); shift = shift - shift[0]
        total = np.zeros((32, X.shape[1]), np.float64)
        acc = None
        for cc in range(0, max(0, lo_b - near)):
            acc = chain(Am, X, kernel_order(cc * KC), acc)
        if acc is not None: total += acc.astype(np.float64)
        acc = np.zeros((32, X.shape[1]), np.float32)
        wins = []
        for q in range(32 // quad_rows):
            rows = list(range(q * quad_rows, (q + 1) * quad_rows))
            c_lo = T // 2 - 1 + shift[rows[0]]; c_hi = T // 2 - 1 + shift[rows[-1]]
            wins.append((rows, c_lo - margin_lo - lead, c_hi + margin_hi))
        for cc in range(max(0, lo_b - near), min(nch, hi_b + near)):
            ks = kernel_order(cc * KC)
            for g in range(0, KC, 4):
                acc = chain(Am, X, ks[g:g + 4], acc)
                k0 = cc * KC + (g // 8) * 8
                for rows, w_lo, w_hi in wins:
                    if k0 < w_hi and k0 + 8 > w_lo:
                        total[rows] += acc[rows].astype(np.float64); acc[rows] = 0
        total += acc.astype(np.float64)
        acc = None; j = 0
        for cc in range(min(nch, hi_b + near), nch):
            acc = chain(Am, X, kernel_order(cc * KC), acc); j += 1
            if right_every and j % right_every == 0: total += acc.astype(np.float64); acc = None
        if acc is not None: total += acc.astype(np.float64)
        return total.astype(np.float32)
    return sched
SCHEMES = {"current (band/4k, singles, pairs, quads)": sched_current,
           "in order, band 4k, right/4 (shipped)": make_inorder(4),
           "row-quad windows [-6-4, +7), 8-row quads": make_row_windows(),
           "row-quad windows, no lead": make_row_windows(lead=0),
           "row-quad windows, margins 4/5": make_row_windows(4, 5),
           "row-quad windows, margins 8/9": make_row_windows(8, 9),
           "in order, band 8k": make_inorder(8),
           }
if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    for sname, sig in SIGNALS.items():
        print(f"== {sname}")
        for name, fn in SCHEMES.items():
            rms, mx, rrms, rmx = simulate(sig, fn, N)
            print(f"   {name:62s} rms {rms:.3e} max {mx:.3e} | reference float loop rms {rrms:.3e} max {rmx:.3e} | ratio {rms / rrms:.2f}")
