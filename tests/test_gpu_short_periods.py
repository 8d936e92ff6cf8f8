"""GPU (-m gpu): ratios whose period is SHORT — 2x and 4x conversions, 48k <-> 32k, 44.1k -> 88.2k: 1 ... 3 outputs per period — or
fits its tiles badly (80 outputs: two and a half tiles).  The matrix-core kernels' tiles hold 32 consecutive slots of one period,
so these ratios are taken several periods at a time (fir_common.hip.h: artfir_period_multiple; any multiple of a period is a
period): every kernel of the path — one tile per workgroup (5), f32 streaming (6), fixed point (7), the library's choice (2) —
against the double-accumulate oracle, streaming across calls and the flush, interpolating and nearest-filter mode (the ART
form: as many filters as outputs per period, low-pass included), and the 8-byte build's fp64 kernels."""
import numpy as np
import pytest

import audio_resampler_amd as A
from _hip import HipResampler, tolerance_ok
from _oracle import OracleResampler, noise, BH, INTERP, LOWPASS, PRECISE
import _oracle

pytestmark = pytest.mark.gpu

RATIOS = [(44100, 88200), (48000, 32000), (192000, 48000), (32000, 48000), (48000, 96000), (48000, 192000), (44100, 24000), (8000, 48000)]


def _multiple(P, rows=32):
    """the rule of fir_common.hip.h restated (what the tests expect the library to choose)"""
    tiles = -(-P // rows)
    if tiles * rows * 100 <= P * 115:
        return 1
    from math import gcd
    if rows // gcd(P, rows) * P <= 16 * rows:
        return rows // gcd(P, rows)
    best, waste = 1, tiles * rows / P
    mu = 2
    while mu * P <= 16 * rows:
        w = (-(-mu * P // rows)) * rows / (mu * P)
        if w < waste - 1e-9:
            best, waste = mu, w
            if w <= 1.04:
                break
        mu += 1
    return best


def test_the_rule_fills_the_tiles():
    from math import gcd
    for src, dst in RATIOS + [(44100, 48000), (96000, 44100)]:
        g = gcd(src, dst)
        P = dst // g
        mu = _multiple(P)
        padded = -(-mu * P // 32) * 32
        assert padded / (mu * P) <= 1.15, (src, dst, P, mu)
    assert _multiple(160) == 1 and _multiple(147) == 1 and _multiple(2) == 16 and _multiple(1) == 32 and _multiple(80) == 2 and _multiple(50) == 5


@pytest.mark.parametrize("kernel", [2, 5, 6, 7])
@pytest.mark.parametrize("ratio", RATIOS, ids=[f"{s}to{d}" for s, d in RATIOS])
def test_short_period_ratios_meet_the_bar_on_every_matrix_kernel(ratio, kernel):
    src, dst = ratio
    ch, T = 4, 380
    blocks = (60000, 45001, 3000)
    r_ = dst / src
    total = sum(blocks)
    x, _ = noise(total * ch, state=0x5E0D7 | 1)
    x = x.reshape(total, ch)
    h = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=kernel); h.advance(T / 2)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE); o.advance(T / 2)
    pos = 0
    for n in blocks:
        cap = int(n * r_) + 4000
        u, g, y = h.process(x[pos:pos + n], cap, r_)
        uo, go, yo = o.process(x[pos:pos + n], cap, r_, threads=8)
        assert (u, g) == (uo, go)
        assert h.last_kernel() == 2, h.last_kernel()
        if kernel == 7:
            assert h.fixed_point()[0] == 1
        ok, worst, rms = tolerance_ok(np.array(y), np.array(yo))
        assert ok, (n, worst)
        pos += n
    u, g, y = h.process(None, 8000, r_, flush=True)
    uo, go, yo = o.process(None, 8000, r_, flush=True)
    assert g == go and tolerance_ok(np.array(y), np.array(yo))[0]


@pytest.mark.parametrize("kernel", [5, 6, 7])
@pytest.mark.parametrize("ratio", [(48000, 32000), (44100, 88200), (192000, 48000), (44100, 24000)], ids=lambda r: f"{r[0]}to{r[1]}")
def test_short_period_ratios_in_the_art_form(ratio, kernel):
    """resampleFixedRatioInit with as many filters as the period has outputs, nearest-filter mode, low-pass: 2, 2, 1 and 80 filters"""
    src, dst = ratio
    from math import gcd
    F = dst // gcd(src, dst)
    ch, T = 8, 988
    blocks = (90000, 70000)
    r_ = dst / src
    total = sum(blocks)
    x, _ = noise(total * ch, state=0xA27F0 | 1)
    x = x.reshape(total, ch)
    h = HipResampler(ch, T, F, 0.0, BH | LOWPASS, fixed=(float(src), float(dst), 0), kernel=kernel); h.advance(T / 2)
    o = OracleResampler(ch, T, F, 0.0, BH | LOWPASS | PRECISE, fixed=(float(src), float(dst), 0)); o.advance(T / 2)
    pos = 0
    for n in blocks:
        cap = int(n * r_) + 4000
        u, g, y = h.process(x[pos:pos + n], cap, 0.0)
        uo, go, yo = o.process(x[pos:pos + n], cap, 0.0, threads=8)
        assert (u, g) == (uo, go) and h.last_kernel() == 2
        assert tolerance_ok(np.array(y), np.array(yo))[0]
        pos += n


def test_streaming_and_tile_kernels_agree_bit_for_bit_on_short_periods():
    """the f32 streaming kernel and the one-tile-per-workgroup kernel share tiles and K order: same bits, also several periods at a time"""
    for src, dst in [(44100, 88200), (48000, 32000), (192000, 48000)]:
        ch, T, n = 2, 380, 150000
        r_ = dst / src
        x, _ = noise(n * ch, state=0xB17 | 1)
        x = x.reshape(n, ch)
        outs = []
        for kernel in (6, 5):
            h = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=kernel); h.advance(T / 2)
            u, g, y = h.process(x, int(n * r_) + 4000, r_)
            assert h.last_kernel() == 2
            outs.append(np.array(y).copy())
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), (src, dst)


@pytest.mark.parametrize("ratio", [(44100, 88200), (48000, 32000), (192000, 48000)], ids=lambda r: f"{r[0]}to{r[1]}")
def test_short_period_ratios_in_the_8_byte_build(ratio):
    src, dst = ratio
    W, Wo = A.wide(), _oracle.wide()
    ch, T, n = 4, 380, 120000
    r_ = dst / src
    rng = np.random.default_rng(7)
    x = (rng.random((n, ch)) - 0.5)
    h = W.Resampler(ch, T, T, 0.0, BH | INTERP); h.set_kernel(2); h.advance(T / 2)
    o = Wo.OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE); o.advance(T / 2)
    cap = int(n * r_) + 4000
    u, g, y = h.process(x, cap, r_)
    uo, go, yo = o.process(x, cap, r_)
    assert (u, g) == (uo, go) and h.last_kernel() == 2
    d = np.abs(np.array(y) - np.array(yo))
    assert np.all(d <= 2.0 ** -47 * np.maximum(1.0, np.abs(np.array(yo)))), float(d.max())


ODD_RATIOS = [(7, 3), (3, 7), (5, 4), (4, 5), (13, 11), (24, 25), (25, 24), (3, 1), (1, 3), (50, 49), (99, 100), (64, 63), (1, 1)]


@pytest.mark.parametrize("kernel", [2, 7])
@pytest.mark.parametrize("pq", ODD_RATIOS, ids=[f"{p}over{q}" for p, q in ODD_RATIOS])
def test_arbitrary_small_rational_ratios(pq, kernel):
    """outputs per period 1 ... 99: every one of them goes through the period rule (exact fill, best fit below 16 tiles, or as it is)
    — matrix path (library's choice of kernel, and fixed point forced) against the oracle, two calls and the flush"""
    p, q = pq
    ch, T = 2, 256
    blocks = (70001, 52000)
    r_ = p / q
    total = sum(blocks)
    x, _ = noise(total * ch, state=(0xD00D + 977 * p + q) | 1)
    x = x.reshape(total, ch)
    h = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=kernel); h.advance(T / 2)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE); o.advance(T / 2)
    pos = 0
    for n in blocks:
        cap = int(n * r_) + 4000
        u, g, y = h.process(x[pos:pos + n], cap, r_)
        uo, go, yo = o.process(x[pos:pos + n], cap, r_, threads=8)
        assert (u, g) == (uo, go)
        assert h.last_kernel() == 2, h.last_kernel()
        assert tolerance_ok(np.array(y), np.array(yo))[0], n
        pos += n
    u, g, y = h.process(None, 8000, r_, flush=True)
    uo, go, yo = o.process(None, 8000, r_, flush=True)
    assert g == go and (g == 0 or tolerance_ok(np.array(y), np.array(yo))[0])
