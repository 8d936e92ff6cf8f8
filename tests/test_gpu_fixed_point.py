"""GPU (-m gpu): the fixed-point form of the matrix-core path (fir_matrix_i8.hip: samples and effective rows as four signed
8-bit digits, exact integer accumulation on v_mfma_i32_32x32x32_i8, one float rounding per output) — the kernel of big
regular launches with long filters (kernel preference 7: of every regular launch).  Samples are BLOCK floating point: the
launch is cut into exponent blocks of ~10-40k input frames, and inside a block every channel is scaled by its own power of two,
taken from the channel's peak |x| over the frames the block's outputs read, so the arithmetic is scale-free like the
reference's float loop (any finite amplitude, quiet channels beside loud ones, quiet passages after loud ones).  Its
only errors are the 2^-31 quantisation of the effective rows (about 5e-9 x the signal's rms), of samples more than 2^-6 below
their channel's peak (2^-31 x that peak) and ONE float rounding, so against the double-accumulate oracle's float it must sit
within one float spacing + 2^-24 x the channel's peak everywhere, at about half the f32 kernels' rms error or less (they carry
~T roundings and only promise the parity bar) and no worse than 1.25 x the reference's own float loop AT EVERY AMPLITUDE;
infinities and NaNs — in the call's input or in the history — must hand the launch to the f32 streaming kernel, bit for bit."""
import os

import numpy as np
import pytest

import audio_resampler_amd as A
from _hip import HipResampler, tolerance_ok
from _oracle import OracleResampler, noise, BH, INTERP, LOWPASS, PRECISE

pytestmark = pytest.mark.gpu

CASES = [
    # (channels, taps, filters, src, dst, fixed-ratio form, flags, blocks)
    (8, 988, 988, 44100, 48000, False, BH | INTERP, (70000, 50000, 131072)),          # headline shape
    (4, 988, 988, 44100, 48000, False, BH | INTERP, (90000, 90000)),
    (2, 380, 380, 44100, 48000, False, BH | INTERP, (200000, 100001)),                # 13 chunks per tile (odd)
    (1, 380, 380, 44100, 48000, False, BH | INTERP, (300000,)),                       # mono: half the columns idle
    (16, 156, 156, 44100, 48000, False, BH | INTERP, (60000, 60000)),
    (32, 988, 988, 44100, 48000, False, BH | INTERP, (30000, 30000)),
    (2, 380, 380, 44100, 48000, True, BH | INTERP | LOWPASS, (150000, 150000)),       # ART form: nearest filter, SNAP, low-pass
    (8, 988, 988, 96000, 44100, True, BH | INTERP | LOWPASS, (140000, 140000)),       # downsampling: P = 147, Q = 320 (period stride 1)
    (2, 380, 320, 44100, 48000, False, BH, (120000, 120000)),                         # nearest filter, pass-through samples
    (2, 380, 32, 44100, 48000, False, BH, (120000, 120000)),                          # nearest filter, F < P: five pass-through slots per period, two of them in one tile
    (2, 64, 160, 48000, 44100, False, BH, (250000,)),                                 # P = 147, Q = 160, 3 chunks
    (2, 380, 380, 48000, 40000, False, BH | INTERP, (200000,)),                       # P = 5, Q = 6: period stride 2, 5 live rows per tile
    (8, 988, 988, 44100, 48000, False, BH | INTERP, (2000, 3000, 500, 9000)),         # small calls
]
IDS = [f"c{c[0]}_t{c[1]}_f{c[2]}_{c[3]}to{c[4]}{'_fixed' if c[5] else ''}_{len(c[7])}calls" for c in CASES]


def _make(case, kernel, oracle=False):
    ch, T, F, src, dst, fixed, flags, blocks = case
    if oracle:
        r = OracleResampler(ch, T, F, flags=flags | PRECISE, fixed=(float(src), float(dst), 0)) if fixed else OracleResampler(ch, T, F, 0.0, flags | PRECISE)
    else:
        r = HipResampler(ch, T, F, flags=flags, fixed=(float(src), float(dst), 0), kernel=kernel) if fixed else HipResampler(ch, T, F, 0.0, flags, kernel=kernel)
    r.advance(T / 2)
    return r


def _spacing(t64):
    """float32 spacing of the binade each value lies in"""
    return 2.0 ** (np.floor(np.log2(np.maximum(np.abs(t64), 2.0 ** -126))) - 23)


def _play(r, x, blocks, ratio, fixed, want_state=None):
    outs, pos = [], 0
    for n in blocks:
        u, g, y = r.process(x[pos:pos + n], int(n * ratio) + 4000, 0.0 if fixed else ratio)
        assert u == n
        if want_state is not None:
            assert r.last_kernel() == 2 and r.fixed_point() [0] == want_state, (r.last_kernel(), r.fixed_point())
        outs.append(np.array(y).copy())
        pos += n
    outs.append(np.array(r.process(None, 8000, ratio, flush=True) [2]).copy())
    return outs


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_fixed_point_kernel_is_correctly_rounded_against_the_double_accumulate_oracle(case):
    ch, T, F, src, dst, fixed, flags, blocks = case
    ratio, total = dst / src, sum(blocks)
    x, _ = noise(total * ch, state=0xFACADE5EED | 1)
    x = x.reshape(total, ch)
    # (calls of a few hundred frames go to the general kernel: only the results are checked there)
    y = np.concatenate(_play(_make(case, 7), x, blocks, ratio, fixed, want_state=1 if min(blocks) >= 20000 else None))
    t = np.concatenate(_play(_make(case, 0, oracle=True), x, blocks, ratio, fixed))
    assert y.shape == t.shape
    y64, t64 = y.astype(np.float64), t.astype(np.float64)
    # both round a value known to a few 1e-9 (rows on a 2^-30 grid: 2.7e-10 rms per tap x sqrt (T) x the signal's rms): the same
    # float, its neighbour across a rounding boundary, or — for small results — anything within that absolute error
    err = np.abs(y64 - t64)
    assert np.all(err <= _spacing(t64) + 2.0 ** -25), float((err - _spacing(t64)).max())
    assert tolerance_ok(y, t) [0]
    # about half the f32 kernels' rms error, or less
    yf = np.concatenate(_play(_make(case, 6), x, blocks, ratio, fixed))
    assert np.sqrt(np.mean((y64 - t64) ** 2)) <= 0.6 * np.sqrt(np.mean((yf.astype(np.float64) - t64) ** 2))


def _rms(v):
    return float(np.sqrt(np.mean(np.asarray(v, np.float64) ** 2)))


AMPLITUDES = [1.9, 40.0, 2.0 ** -8, 1e-3, 1e-6, 1e-12, 3e20]


@pytest.mark.parametrize("amplitude", AMPLITUDES, ids=[f"{a:g}" for a in AMPLITUDES])
def test_fixed_point_kernel_is_scale_free_on_the_headline_shape(amplitude):
    """8 ch x 988 taps interpolating, 44.1k -> 48k, noise of the given amplitude (from far above the old +-1.98 range of the
    digits to -240 dBFS): the launch runs in fixed point, every sample within one float spacing + 2^-24 x amplitude of the
    double-accumulate oracle, rms error <= 1.25 x the reference float loop's (measured ~0.5 x) and <= 0.6 x the f32 matrix
    kernel's — the same figures at every amplitude, as for float arithmetic"""
    ch, T, frames = 8, 988, 70000
    ratio = 48000 / 44100
    x, _ = noise(frames * ch, state=0xA11CE | 1)
    x = (x.reshape(frames, ch).astype(np.float64) * (amplitude / 0.5)).astype(np.float32)
    cap = int(frames * ratio) + 4000
    r = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=7); r.advance(T / 2)
    f = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=6); f.advance(T / 2)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE); o.advance(T / 2)
    l = OracleResampler(ch, T, T, 0.0, BH | INTERP); l.advance(T / 2)
    u, g, y = r.process(x, cap, ratio)
    assert r.last_kernel() == 2 and r.fixed_point() [0] == 1, (r.last_kernel(), r.fixed_point())
    uf, gf, yf = f.process(x, cap, ratio)
    uo, go, yo = o.process(x, cap, ratio, threads=8)
    ul, gl, yl = l.process(x, cap, ratio, threads=8)
    assert (u, g) == (uo, go) == (ul, gl) == (uf, gf)
    y64, t64 = np.array(y, np.float64), np.array(yo, np.float64)
    err = np.abs(y64 - t64)
    assert np.all(err <= _spacing(t64) + 2.0 ** -24 * amplitude), float((err - _spacing(t64)).max() / amplitude)
    e_fixed, e_loop, e_f32 = _rms(y64 - t64), _rms(np.array(yl, np.float64) - t64), _rms(np.array(yf, np.float64) - t64)
    assert e_fixed <= 1.25 * e_loop, (e_fixed, e_loop)
    assert e_fixed <= 0.6 * e_f32, (e_fixed, e_f32)


def test_fixed_point_exponents_are_per_channel():
    """one call, eight channels at levels from full scale down to -200 dBFS side by side (and one silent): every channel has its
    own block exponent, so every channel on its own meets the bar against the reference float loop"""
    ch, T, frames = 8, 988, 70000
    ratio = 48000 / 44100
    levels = np.array([1.0, 1e-4, 0.3, 1e-10, 0.0, 2.0 ** -12, 7.0, 1e-2])         # channel 1: -80 dBFS beside a loud channel 0
    x, _ = noise(frames * ch, state=0xC0FFEE | 1)
    x = (x.reshape(frames, ch).astype(np.float64) * levels).astype(np.float32)
    cap = int(frames * ratio) + 4000
    r = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=7); r.advance(T / 2)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE); o.advance(T / 2)
    l = OracleResampler(ch, T, T, 0.0, BH | INTERP); l.advance(T / 2)
    u, g, y = r.process(x, cap, ratio)
    assert r.fixed_point() [0] == 1
    uo, go, yo = o.process(x, cap, ratio, threads=8)
    ul, gl, yl = l.process(x, cap, ratio, threads=8)
    y64, t64, l64 = np.array(y, np.float64), np.array(yo, np.float64), np.array(yl, np.float64)
    for c in range(ch):
        peak = 0.5 * levels [c]
        err = np.abs(y64 [:, c] - t64 [:, c])
        assert np.all(err <= _spacing(t64 [:, c]) + 2.0 ** -24 * peak), (c, float(err.max()))
        if peak == 0.0:
            assert not np.any(y64 [:, c])
        else:
            assert _rms(y64 [:, c] - t64 [:, c]) <= 1.25 * _rms(l64 [:, c] - t64 [:, c]), c


def test_fixed_point_floor_is_relative_to_the_channels_peak_in_the_exponent_block():
    """a call that starts loud and drops by 80 dB after 20,000 frames: exponents belong to blocks of ~10k input frames (8 channels x
    988 taps), so only the outputs whose block still sees loud frames are computed on the loud grid — their error floor is
    2^-31 x the loud peak rms (about -196 dB below that peak), not relative to the quiet signal; one block later the quiet
    signal has its own exponent and meets the float-loop bar again, inside the same call"""
    ch, T, frames, loud = 8, 988, 90000, 20000
    ratio = 48000 / 44100
    x, _ = noise(2 * frames * ch, state=0xF00D | 1)
    x = x.reshape(2 * frames, ch).copy()
    x [loud:] *= np.float32(1e-4)
    cap = int(frames * ratio) + 4000
    r = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=7); r.advance(T / 2)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE); o.advance(T / 2)
    l = OracleResampler(ch, T, T, 0.0, BH | INTERP); l.advance(T / 2)
    errs, refs = [], []
    for k in range(2):
        blk = x [k * frames:(k + 1) * frames]
        u, g, y = r.process(blk, cap, ratio)
        assert r.fixed_point() [0] == 1
        uo, go, yo = o.process(blk, cap, ratio, threads=8)
        ul, gl, yl = l.process(blk, cap, ratio, threads=8)
        errs.append(np.array(y, np.float64) - np.array(yo, np.float64))
        refs.append(np.array(yl, np.float64) - np.array(yo, np.float64))
        assert tolerance_ok(np.array(y), np.array(yo)) [0]
    n_loud = int((loud - 2 * T) * ratio)                        # outputs that only see loud samples
    n_mixed = int((loud + 2 * T) * ratio)                       # from here on: quiet samples only, possibly still on the loud grid
    n_quiet = int((loud + 14000) * ratio)                       # one exponent block (9,408 frames + a window) later
    assert _rms(errs [0] [:n_loud]) <= 1.25 * _rms(refs [0] [:n_loud])
    assert _rms(errs [0] [n_mixed:n_quiet]) <= 2.0 ** -31 * 0.5                # the loud block's floor
    assert _rms(errs [0] [n_quiet:]) <= 1.25 * _rms(refs [0] [n_quiet:])       # the quiet signal on its own grid, same call
    assert _rms(errs [1]) <= 1.25 * _rms(refs [1])


EDGE_MANTISSAS = [0x7dffff, 0x7e0000, 0x7efefe, 0x7efeff, 0x7eff80, 0x7effff, 0x7f0000, 0x7fffff]


@pytest.mark.parametrize("sign", [1.0, -1.0], ids=["positive", "negative"])
def test_peaks_at_the_top_of_the_digit_range_do_not_carry_out_of_the_top_digit(sign):
    """four signed base-256 digits end at 0x7f7f7f7f, 0x8081 short of 2^31 - 2^23: a channel's peak with mantissa 0x7efeff ... 0x7effff
    scaled to exponent 30 would carry out of the dword (found by tools/fuzz_long.py --kernel 7 --seed 41, session 1279: one such
    sample among 6.7 M, an output wrong by half of full scale).  Every channel here has its peak on one of the mantissas around
    the boundaries, at different binary exponents, positive or negative, one exponent block each"""
    ch, T, frames = 8, 48, 60000
    ratio = 32000 / 48000
    x, _ = noise(frames * ch, state=0xED6E | 1)
    x = x.reshape(frames, ch).copy()
    for c, mant in enumerate(EDGE_MANTISSAS):
        peak = np.array([(126 + c % 3) << 23 | mant], np.uint32).view(np.float32) [0]         # 0.99.. / 1.98.. / 3.9..
        x [:, c] *= np.float32(peak)                                                            # |noise| <= 0.5: below the peak
        for f in range(5000 + 37 * c, frames, 9000):                                            # one peak sample per exponent block
            x [f, c] = np.float32(sign) * peak
    cap = int(frames * ratio) + 400
    r = HipResampler(ch, T, T, 0.0, BH | LOWPASS, fixed=(48000.0, 32000.0, 0), kernel=7); r.advance(T / 2)
    o = OracleResampler(ch, T, T, 0.0, BH | LOWPASS | PRECISE, fixed=(48000.0, 32000.0, 0)); o.advance(T / 2)
    u, g, y = r.process(x, cap, ratio)
    assert r.last_kernel() == 2 and r.fixed_point() [0] == 1, (r.last_kernel(), r.fixed_point())
    uo, go, yo = o.process(x, cap, ratio, threads=8)
    assert (u, g) == (uo, go)
    y64, t64 = np.array(y, np.float64), np.array(yo, np.float64)
    for c in range(ch):
        err = np.abs(y64 [:, c] - t64 [:, c])
        assert np.all(err <= _spacing(t64 [:, c]) + 2.0 ** -24 * 4.0), (c, hex(EDGE_MANTISSAS [c]), float(err.max()))


BAD = [("above the old range", 2.5, 1), ("below the old range", -1.99, 1), ("huge", -3e30, 1), ("infinity", np.inf, 2), ("NaN", np.nan, 2)]


@pytest.mark.parametrize("shape", [(8, 988, 988, BH | INTERP), (2, 380, 320, BH)], ids=["c8_t988_interp", "c2_t380_f320_nearest_passthrough"])
@pytest.mark.parametrize("what,value,state", BAD, ids=[b [0].replace(" ", "_") for b in BAD])
def test_samples_the_digits_cannot_hold_hand_the_launch_to_the_f32_kernel(what, value, state, shape):
    """an infinity or a NaN anywhere in the call, or in the history it still convolves with: the fixed-point kernel stands down on
    the device and the f32 streaming kernel's tile loop produces the call — the same bits as with that kernel pinned; calls that
    do not touch the sample run in fixed point again.  Any FINITE sample, however large, is held: the channel's exponent follows
    its peak (the launch stays in fixed point; the other samples of that channel in the same exponent block sit on the outlier's grid)"""
    ch, T, F, flags = shape
    frames = 60000
    ratio = 48000 / 44100
    x, _ = noise(4 * frames * ch, state=0xBAD5A | 1)
    x = x.reshape(4 * frames, ch).copy()
    x [frames + frames // 2, ch // 2] = value                 # in the second call
    x [3 * frames - 40, ch - 1] = value                           # at the end of the third: still in the fourth call's history
    states, outs = [], {}
    for kernel in (7, 6):
        r = HipResampler(ch, T, F, 0.0, flags, kernel=kernel); r.advance(T / 2)
        outs [kernel] = []
        for k in range(4):
            u, g, y = r.process(x [k * frames:(k + 1) * frames], int(frames * ratio) + 4000, ratio)
            assert u == frames and r.last_kernel() == 2
            if kernel == 7:
                states.append(r.fixed_point() [0])
            outs [kernel].append(np.array(y).copy())
    assert states == [1, state, state, state], states
    for k in (1, 2, 3):
        a, b = outs [7] [k], outs [6] [k]
        if state == 2:
            # the f32 tile loop on the fixed-point kernel's workgroups.  (Until round 5: the bits of the f32 kernel pinned.  Since the rows are kept
            # across calls the launch's tiles are anchored on the stream's canonical period, not on the launch's first output — other K origins for
            # the f32 chains, rows blended at phases ~1e-8 filter steps away: the f32 kernel's values to within the bar, NaNs and infinities where
            # it has them; ARTAMD_ROWS_CACHE=0 gives the old anchoring and the old bits, test_stand_by_without_cached_rows_is_the_f32_kernel)
            assert a.shape == b.shape
            if os.environ.get("ARTAMD_TEST_STANDBY_BITS") == "1":
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (what, k)
            fa, fb = np.isfinite(a), np.isfinite(b)
            # (a tile's zero-padded K columns reach a few dozen frames beyond a row's taps, and 0 x infinity is a NaN: which outputs at the
            # edge of the sample's reach are lost depends on the tile's K origin — both kernels lose the sample's own window, the edges differ)
            assert np.count_nonzero(fa != fb) <= 80 * ch and np.count_nonzero(~fa) > 0 and np.count_nonzero(~fb) > 0, (what, k, np.count_nonzero(fa != fb))
            both = fa & fb
            d = np.abs(a [both].astype(np.float64) - b [both].astype(np.float64))
            assert np.all(d <= 2.0 ** -23 * np.maximum(1.0, np.abs(b [both].astype(np.float64)))), (what, k, float(d.max()))
        else:
            # fixed point on the outlier's grid: within the parity bar's size of the f32 kernel, relative to the channel's peak
            peak = np.maximum(np.abs(x [max(k * frames - 2 * T, 0):(k + 1) * frames]).max(axis=0), 0.5).astype(np.float64)
            with np.errstate(over="ignore", invalid="ignore"):
                d = np.abs(a.astype(np.float64) - b.astype(np.float64))
            assert np.all(d <= 2.0 ** -22 * np.maximum(peak, np.abs(b.astype(np.float64)))), (what, k, float(d.max()))
    # the first call never saw the sample: fixed point, and within half an ulp of the f32 kernel's neighbourhood
    assert not np.array_equal(outs [7] [0].view(np.uint32), outs [6] [0].view(np.uint32))
    assert np.all(np.abs(outs [7] [0].astype(np.float64) - outs [6] [0].astype(np.float64)) <= 2.0 ** -22)


def test_stand_by_without_cached_rows_is_the_f32_kernel():
    """ARTAMD_ROWS_CACHE=0 (every launch builds its rows from its own positions, tiles anchored on its first output): the stand-by is the f32
    streaming kernel bit for bit — the infinity / NaN cases above in a process with the switch set"""
    import os, subprocess, sys
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.abspath(__file__),
                        "-k", "test_samples_the_digits_cannot_hold and (infinity or NaN)"],
                       env=dict(os.environ, ARTAMD_ROWS_CACHE="0", ARTAMD_TEST_STANDBY_BITS="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_fixed_point_kernel_skips_the_zero_digit_plane_and_is_chosen_where_it_pays():
    """diagnostics: of the 13 digit-pair products per chunk, the 4 with the rows' most significant digit are only issued in the
    chunks around the rows' centres (taps fall off as 1 / distance), the 4 with the second digit not in the window's tails (taps below
    2^-15) — on the headline filter, between 8 and 9 of 13 on average.
    The automatic choice takes the fixed-point kernel for long filters in big calls only (fir_matrix.hip, artfir_planes_bytes)."""
    ratio = 48000 / 44100
    for ch, T, frames, want in ((8, 988, 300000, 1), (8, 988, 60000, 0), (2, 380, 1000000, 0)):
        x, _ = noise(frames * ch)
        r = HipResampler(ch, T, T, 0.0, BH | INTERP); r.advance(T / 2)
        u, g, y = r.process(x.reshape(frames, ch), int(frames * ratio) + 4000, ratio)
        state, pairs = r.fixed_point()
        assert u == frames and r.last_kernel() == 2 and state == want, (ch, T, frames, state)
        if want:
            assert 8.0 < pairs < 9.0, pairs


def test_fixed_point_long_call_with_many_ring_epochs():
    """one call of 2.2 M frames x 2 channels x 988 taps: > 192 ring epochs, so several launches with n_begin > 0, each with its own
    staging pass — every launch in fixed point, the whole call inside the bar of the double-accumulate oracle and within a
    rounding or two of the f32 matrix kernel"""
    ch, T, frames = 2, 988, 2200000
    ratio = 48000 / 44100
    x, _ = noise(frames * ch)
    x = x.reshape(frames, ch)
    outs = {}
    for kernel in (7, 6):
        r = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=kernel); r.advance(T / 2)
        u, g, y = r.process(x, int(frames * ratio) + 4000, ratio)
        assert u == frames and r.last_kernel() == 2 and r.fixed_point() [0] == (1 if kernel == 7 else 0)
        outs [kernel] = np.array(y)
    assert outs [7].shape == outs [6].shape
    assert np.all(np.abs(outs [7].astype(np.float64) - outs [6].astype(np.float64)) <= 2.0 ** -22)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE); o.advance(T / 2)
    n = 300000                                               # (the oracle on the first 300k frames: the first launches' seams)
    uo, go, yo = o.process(x [:n], int(n * ratio) + 4000, ratio, threads=2)
    yo = np.array(yo)
    assert tolerance_ok(outs [7] [:len(yo) - 2000], yo [:len(yo) - 2000]) [0]
