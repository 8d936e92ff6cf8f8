"""Long randomised comparison of the matrix-core FIR path against the oracle on LARGE blocks (the pytest fuzz uses blocks of a
few taps, which mostly exercise the tiles at the history seam): random channel counts (every compiled column group and
generic ones), taps, rational ratios, block sizes up to a few hundred thousand frames, several calls per stream + flush.
Default sample width and the 8-byte build.  Usage: python tools/fuzz_long.py [--seconds S] [--seed N] [--kernel 7]  (GPU box; CPU oracle)."""
import argparse, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import audio_resampler_amd as A
import _oracle
from _hip import tolerance_ok

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=240); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--budget", type=float, default=4e8); ap.add_argument("--only", type=int, default=-1, help="replay: draw every session's parameters and data but run only session number N"); ap.add_argument("--kernel", type=int, default=2, help="kernel preference of the 4-byte contexts: 2 matrix path (automatic among its kernels), 7 fixed point wherever it can run")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
BH, INTERP, LOWPASS, PRECISE = _oracle.BH, _oracle.INTERP, _oracle.LOWPASS, _oracle.PRECISE
t_end = time.time() + args.seconds
n_ok = n_bad = 0
kinds = {}
session = -1
while time.time() < t_end:
    session += 1
    wide = bool(rng.integers(0, 4) == 0)
    ch = int(rng.choice([1, 2, 3, 4, 6, 8, 16, 32, 33]))
    T = int(rng.choice([48, 156, 380, 988, 1024]))
    src, dst = [(44100, 48000), (96000, 44100), (48000, 32000), (8000, 48000), (44100, 88200), (48000, 44100), (44100, 96000), (192000, 48000)][int(rng.integers(0, 8))]
    interp = bool(rng.integers(0, 3))
    fixed = bool(rng.integers(0, 2))
    flags = BH | (INTERP if interp else 0)
    F = int(rng.choice([T, 160, 380])) if not fixed else int(rng.choice([147, 160, 320, T]))
    budget = int(args.budget if not wide else args.budget / 2)                # oracle multiply-adds per session (a few seconds)
    max_frames = max(2000, min(300000, budget // (ch * T * (2 if interp else 1))))
    calls = [int(rng.integers(max_frames // 8, max_frames // 2)) for _ in range(int(rng.integers(2, 5)))]
    ratio = dst / src
    if wide:
        W = A.wide(); Wo = _oracle.wide(); dt = np.float64
        mk_h = lambda: W.Resampler(ch, T, F, 0.0, flags | (LOWPASS if fixed else 0), (float(src), float(dst), 0) if fixed else None)
        mk_o = lambda: Wo.OracleResampler(ch, T, F, 0.0, flags | (LOWPASS if fixed else 0) | PRECISE, fixed=(float(src), float(dst), 0) if fixed else None)
    else:
        dt = np.float32
        mk_h = lambda: A.Resampler(ch, T, F, 0.0, flags | (LOWPASS if fixed else 0), (float(src), float(dst), 0) if fixed else None)
        mk_o = lambda: _oracle.OracleResampler(ch, T, F, 0.0, flags | (LOWPASS if fixed else 0) | PRECISE, fixed=(float(src), float(dst), 0) if fixed else None)
    if args.only >= 0 and session != args.only:
        if not wide or True:
            level = float(10.0 ** rng.uniform(-5.0, 1.5)) if rng.integers(0, 2) else 1.0
            rng.random((sum(calls) + 8, ch)); rng.random(ch)       # (the draws the session would have made)
        if session > args.only: break
        continue
    try:
        h, o = mk_h(), mk_o()
    except Exception as e:
        continue
    h.set_kernel(2 if wide else args.kernel)
    adv = T / 2
    h.advance(adv); o.advance(adv)
    # (every session at its own level, every channel a little apart: the tolerance below is relative to the session's level, as
    # float arithmetic's — and the fixed-point kernel's block floating point's — is)
    level = float(10.0 ** rng.uniform(-5.0, 1.5)) if rng.integers(0, 2) else 1.0
    unit = 2.0 ** np.ceil(np.log2(level))                 # the power of two at or above the level: what "full scale" is for this session
    x = ((rng.random((sum(calls) + 8, ch)) - 0.5) * (level * (0.5 + 0.5 * rng.random(ch)))).astype(dt)
    pos, bad, used_kernels = 0, None, set()
    for k, n in enumerate(calls):
        cap = int(n * ratio * 1.01 + T + 16)
        u, g, y = h.process(x[pos:pos + n], cap, ratio)
        used_kernels.add(h.last_kernel())
        uo, go, yo = o.process(x[pos:pos + n], cap, ratio)
        if (u, g) != (uo, go): bad = ("counts", k, (u, g), (uo, go)); break
        y = np.array(y); yo = np.array(yo)
        if wide:
            # (relative to the session's unit, as for the 4-byte sessions: both results are rounded sums of ~1000 double products)
            d = np.abs(y - yo) / unit; tol = 2.0 ** -47 * np.maximum(1.0, np.abs(yo) / unit)
            if not np.all(d <= tol): bad = ("value64", k, float(d.max()), level); break
        else:
            ok, worst, rms = tolerance_ok(y.astype(np.float64) / unit, yo.astype(np.float64) / unit)
            if not ok: bad = ("value", k, worst, level, int(np.argmax(np.abs(y.astype(np.float64) - yo)) // ch), g); break
        pos += u
    if bad is None:
        u, g, y = h.process(None, 2 * T, ratio, flush=True); uo, go, yo = o.process(None, 2 * T, ratio, flush=True)
        if g != go: bad = ("flush counts", g, go)
        elif g and not wide and not tolerance_ok(np.array(y, np.float64) / unit, np.array(yo, np.float64) / unit)[0]: bad = ("flush value",)
    desc = f"#{session} wide={int(wide)} ch={ch} T={T} F={F} {src}->{dst} interp={int(interp)} fixed={int(fixed)} calls={calls} kernels={sorted(used_kernels)}"
    kinds[(wide, tuple(sorted(used_kernels)))] = kinds.get((wide, tuple(sorted(used_kernels))), 0) + 1
    if bad: n_bad += 1; print("FAIL", desc, bad, flush=True)
    else: n_ok += 1
print(f"sessions ok {n_ok} failed {n_bad}; by (wide, kernels used): {kinds}", flush=True)
sys.exit(1 if n_bad else 0)
