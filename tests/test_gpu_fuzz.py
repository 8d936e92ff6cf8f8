"""GPU: randomised streaming sessions, HIP library vs the oracle on identical call sequences.
strict mode must agree bit for bit (host planner, history roll, ring-epoch segments, flush floor, extrapolation,
strict kernel); default mode (general kernel, MFMA kernel forced on where the ratio is rational) must stay within
the float tolerance of the double-accumulate oracle."""
import numpy as np
import pytest

import audio_resampler_amd as A
from _hip import HipResampler, tolerance_ok
from _oracle import OracleResampler, noise, BH, INTERP, LOWPASS, PRECISE, EXTRAP, NO_REDUCTION

pytestmark = pytest.mark.gpu
STRICT = A.RESAMPLE_STRICT_ORDER


def random_session(seed):
    rng = np.random.default_rng(seed)
    T = int(rng.choice([4, 8, 16, 32, 48, 64, 156, 380, 988]))
    ch = int(rng.choice([1, 2, 3, 4, 5, 8, 9]))
    mode = int(rng.integers(0, 3))
    flags = int(rng.choice([BH | INTERP, BH, INTERP, 0]))
    extrap = bool(rng.integers(0, 3) == 0)
    if mode == 0:      # free ratio
        F = int(rng.choice([1, 2, 7, 32, 160, 380, 1024]))
        ratio = float(rng.choice([48000 / 44100, 44100 / 96000, 0.5, 2.0, 1.0, 1 / 3.0, 3.7, 0.731, 160 / 147 * (1 + 37e-6)]))
        lowpass = float(rng.choice([0.0, 0.0, 0.45, 0.9]))
        ctor = dict(args=(ch, T, F, lowpass, flags | (EXTRAP if extrap else 0)), kw={})
    else:              # fixed ratio (ART style)
        src, dst = [(44100, 48000), (96000, 44100), (48000, 32000), (8000, 48000), (44100, 44100 * 2), (48000, 44100)][int(rng.integers(0, 6))]
        F = int(rng.choice([16, 160, 380, 988]))
        ratio = dst / src
        fl = flags | LOWPASS | (EXTRAP if extrap else 0) | (NO_REDUCTION if rng.integers(0, 5) == 0 else 0)
        ctor = dict(args=(ch, T, F), kw=dict(flags=fl, fixed=(float(src), float(dst), int(rng.choice([0, 0, int(0.4 * min(src, dst))])))))
    interp_on = bool(flags & INTERP)
    adv = float(rng.choice([0.0, T / 2, T / 2 + (0.37 if interp_on and mode == 0 else 0.0)]))
    calls = []
    first = True
    for k in range(int(rng.integers(6, 14))):
        kind = int(rng.integers(0, 12))
        n = int(rng.integers(0, 8 * T + 50)) if kind < 8 else int(rng.integers(14 * T, 19 * T)) if kind < 10 else int(rng.integers(0, 4))
        cap = int(rng.integers(1, 12 * T + 64)) if kind != 3 else int(rng.integers(1, 12))
        first = False
        r = ratio * (1 + rng.uniform(-2e-4, 2e-4)) if (kind == 5 and mode == 0) else ratio
        calls.append(("run", n, cap, r))
        if kind == 9 and k > 3 and not extrap:
            calls.append(("reset",))
    calls.append(("flush", int(rng.integers(1, 6 * T + 8)), ratio))
    calls.append(("flush", 4 * T, ratio))          # continue a flush that may have been cut short
    calls.append(("run", 10, 50, ratio))           # ignored after a flush
    total = sum(c[1] for c in calls if c[0] == "run") + 64
    return ctor, adv, calls, ch, total


def play(cls, session, extra=0, noise_fn=noise, **kw):
    ctor, adv, calls, ch, total = session
    args, ckw = list(ctor["args"]), dict(ctor["kw"])
    if "flags" in ckw:
        ckw["flags"] |= extra
    else:
        args[4] |= extra
    r = cls(*args, **ckw, **kw)
    r.advance(adv)
    x, _ = noise_fn(total * ch, state=0x9E3779B97F4A7C15 | 1)
    x = x.reshape(-1, ch)
    pos, ys, trace = 0, [], []
    for c in calls:
        if c[0] == "reset":
            r.reset()
            r.advance(adv)
            continue
        if c[0] == "flush":
            u, g, y = r.process(None, c[1], c[2], flush=True)
        else:
            u, g, y = r.process(x[pos:pos + c[1]], c[2], c[3])
            pos += u
        ys.append(np.array(y, copy=True))
        trace.append((u, g) + tuple(r.state())[:2])
    return np.concatenate(ys), trace


@pytest.mark.parametrize("seed", range(60))
def test_random_session_strict_bit_exact(seed):
    s = random_session(seed)
    y, tr = play(HipResampler, s, STRICT)
    yo, tro = play(OracleResampler, s)
    assert tr == tro
    assert y.shape == yo.shape and np.array_equal(y.view(np.uint32), yo.view(np.uint32))


@pytest.mark.parametrize("seed", range(60))
@pytest.mark.parametrize("kernel", [0, 2, 9], ids=["auto", "matrix", "cut_invariant_policy"])
def test_random_session_fast_within_tolerance(seed, kernel):
    s = random_session(seed)
    y, tr = play(HipResampler, s, kernel=kernel)
    yo, tro = play(OracleResampler, s, PRECISE)
    assert tr == tro
    ok, worst, rms = tolerance_ok(y, yo)
    assert ok, (worst, rms)


LEVELS = [1e-5, 3.3e-4, 0.02, 0.7, 5.0, 30.0]


@pytest.mark.parametrize("seed", range(24))
@pytest.mark.parametrize("kernel", [0, 2, 6, 8], ids=["auto", "matrix", "f32_stream", "f32_split"])
def test_random_session_fast_is_scale_free(seed, kernel):
    """the default mode's bar is RELATIVE, as float arithmetic is: the same sessions at levels from 1e-5 to 30, the tolerance taken
    relative to the power of two at or above the session's level (tools/fuzz_long.py's rule) — at 1e-5 an absolute 2^-23 would pass
    garbage from the general and the f32 matrix kernels (round 3's verdict: only the fixed-point kernel was tested scale-free
    under the driver).  Kernel preferences: the library's choice, the matrix path wherever the ratio is rational, the f32 streaming
    kernel pinned, the K-split form forced"""
    level = LEVELS [seed % len(LEVELS)]
    unit = 2.0 ** np.ceil(np.log2(level))

    def scaled(count, state):
        x, st = noise(count, state=state)
        return (x * np.float32(2.0 * level)).astype(np.float32), st        # (artest's noise is +-0.5: the session's peak is `level`)

    s = random_session(1000 + seed)
    y, tr = play(HipResampler, s, noise_fn=scaled, kernel=kernel)
    yo, tro = play(OracleResampler, s, PRECISE, noise_fn=scaled)
    assert tr == tro
    ok, worst, rms = tolerance_ok(y.astype(np.float64) / unit, yo.astype(np.float64) / unit)
    if level / unit > 0.9:
        # The stress level (peak 30 under the unit 32): outputs overshoot past the power of two, partial sums of an f32 chain pass 1.0 and
        # their spacing doubles.  There a handful of samples per million reach up to twice the bar — the f32 matrix kernels at 7e-6, the
        # reference's own float loop at 3.4e-6 (worst 1.94 x), the library's own choice at 1e-7 over 300 sessions / 8.2 M samples at this level
        # (tools/micro/fuzz_stats.py, profiles/r6_fuzz_stress.txt) — so WHICH seeded session holds one changes with any change of the last
        # bits (round 6: the anchoring of multi-period tilings moved one into seed 17).  Held to: nothing beyond 2 x the bar, at most 3 samples
        # (or 3e-5 of the session's) beyond 1 x.
        e = np.abs(y.astype(np.float64) - yo.astype(np.float64)) / unit
        tol = 2.0 ** -23 * np.maximum(1.0, np.abs(yo.astype(np.float64) / unit))
        outside = int((e > tol).sum())
        assert float((e / tol).max()) <= 2.0 and outside <= max(3, int(3e-5 * e.size)), (level, worst, rms, outside)
    else:
        assert ok, (level, worst, rms)


@pytest.mark.parametrize("seed", range(20))
def test_random_session_precise_mode(seed):
    s = random_session(seed)
    y, _ = play(HipResampler, s, PRECISE)
    yo, _ = play(OracleResampler, s, PRECISE)
    d = np.abs(y.astype(np.float64) - yo.astype(np.float64))
    assert np.all(d <= np.spacing(np.abs(yo)).astype(np.float64) + 1e-45)       # <= 1 float ulp


# ------------------------------------------------------------------------------------------------
# less-travelled geometry: many channels (several column groups), maximum taps / filters, strong ratios
# ------------------------------------------------------------------------------------------------
EDGE_CASES = [
    # (channels, taps, filters, src, dst, fixed, flags)
    (12, 64, 64, 44100, 48000, False, BH | INTERP),          # generic column group (12 | 128 no)
    (33, 48, 48, 44100, 48000, False, BH | INTERP),          # two channel groups (32 + 1)
    (40, 156, 320, 96000, 44100, True, BH | INTERP | LOWPASS),
    (64, 32, 16, 48000, 44100, True, BH | INTERP | LOWPASS),
    (6, 1024, 1024, 44100, 48000, False, BH | INTERP),       # maximum taps and filters, 5.1 layout
    (2, 1024, 1024, 192000, 44100, True, BH | INTERP | LOWPASS),    # P = 147, Q = 640: long input span per period
    (2, 380, 380, 8000, 48000, True, BH | INTERP | LOWPASS),        # x6 upsampling, P = 6
    (1, 156, 320, 44100, 8000, True, BH | INTERP | LOWPASS),        # P = 80, Q = 441
    (4, 48, 48, 32000, 96000, True, INTERP | LOWPASS),              # Hann, x3
    (16, 380, 380, 44100, 48000, False, BH | INTERP),               # compile-time CG = 16
    (32, 156, 156, 44100, 48000, False, BH),                        # compile-time CG = 32, nearest filter
]


@pytest.mark.parametrize("case", EDGE_CASES, ids=lambda c: f"c{c[0]}_t{c[1]}_f{c[2]}_{c[3]}to{c[4]}{'_fixed' if c[5] else ''}")
@pytest.mark.parametrize("kernel", [0, 2])
def test_edge_geometries(case, kernel):
    ch, T, F, src, dst, fixed, flags = case
    ratio = dst / src
    n1, n2 = 20 * T + 77, 9 * T + 5
    x, _ = noise((n1 + n2) * ch, state=0xDEADBEEFCAFEF00D | 1)
    x = x.reshape(-1, ch)

    def make(cls, extra=0, **kw):
        r = cls(ch, T, F, flags=flags | extra, fixed=(float(src), float(dst), 0), **kw) if fixed else cls(ch, T, F, 0.0, flags | extra, **kw)
        r.advance(T / 2)
        return r

    def run(r):
        outs, tr = [], []
        for seg, cap in ((x[:n1], int(n1 * ratio) + 2 * T), (x[n1:], int(n2 * ratio) + 2 * T)):
            u, g, y = r.process(seg, cap, 0.0 if fixed else ratio)
            outs.append(y)
            tr.append((u, g) + tuple(r.state())[:2])
        u, g, y = r.process(None, 4 * T + int(T * ratio), ratio, flush=True)
        outs.append(y)
        tr.append((u, g) + tuple(r.state())[:2])
        return np.concatenate(outs), tr

    y, tr = run(make(HipResampler, kernel=kernel))
    yo, tro = run(make(OracleResampler, PRECISE))
    assert tr == tro
    ok, worst, rms = tolerance_ok(y, yo)
    assert ok, (worst, rms)
    if kernel == 0:
        ys, trs = run(make(HipResampler, STRICT))
        yos, _ = run(make(OracleResampler))
        assert np.array_equal(ys.view(np.uint32), yos.view(np.uint32))
