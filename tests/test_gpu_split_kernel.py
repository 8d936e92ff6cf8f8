"""GPU (-m gpu): fir_mfma_split_kernel — the f32 streaming kernel for launches of few tiles (calls of some ten thousand frames), a
tile's K range cut into 2 .. 8 work items whose fp64 partial sums the last-arriving wave adds in the order of the parts.
Kernel preference 8 forces it wherever the matrix path runs; the library's own choice (0 / 2) takes it where the parts of an XCD's tiles fit one round of its 32 CUs (long filters only).
Against the double-accumulate oracle (the parity bar), against the unsplit streaming kernel (a part starts its own f32 accumulators where
the unsplit walk carries one through the rows' tails: last-bit differences in some outputs, both inside the bar), and against itself (the
result must not depend on which part arrives last: repeated runs give the same bits)."""
import numpy as np
import pytest

import audio_resampler_amd as A
from _hip import HipResampler, tolerance_ok
from _oracle import OracleResampler, noise, BH, INTERP, LOWPASS, PRECISE

pytestmark = pytest.mark.gpu

CASES = [
    # (channels, taps, filters, src, dst, fixed, flags, blocks)
    (8, 988, 988, 44100, 48000, False, BH | INTERP, (65536, 65536, 16384, 32768)),     # the calls of round 2's cliff: 8 / 4 / 4 parts ...
    (2, 380, 380, 44100, 48000, False, BH | INTERP, (65536, 30000)),
    (1, 380, 380, 44100, 48000, False, BH | INTERP, (100000,)),
    (32, 988, 988, 44100, 48000, False, BH | INTERP, (20000, 9000)),
    (2, 380, 320, 44100, 48000, False, BH, (60000, 60000)),                            # nearest filter, pass-through samples
    (8, 988, 988, 96000, 44100, True, BH | INTERP | LOWPASS, (50000, 70000)),          # ART form, P = 147
    (4, 988, 988, 48000, 32000, False, BH | INTERP, (40000, 40000)),                   # short period, several at a time
    (16, 156, 156, 44100, 48000, False, BH | INTERP, (30000, 30000)),                  # 6 chunks: not split below four chunks a part
]
IDS = [f"c{c[0]}_t{c[1]}_f{c[2]}_{c[3]}to{c[4]}{'_fixed' if c[5] else ''}" for c in CASES]


def _play(case, kernel):
    ch, T, F, src, dst, fixed, flags, blocks = case
    ratio, total = dst / src, sum(blocks)
    x, _ = noise(total * ch, state=0x5B117 | 1)
    x = x.reshape(total, ch)
    if kernel == "oracle":
        r = OracleResampler(ch, T, F, flags=flags | PRECISE, fixed=(float(src), float(dst), 0)) if fixed else OracleResampler(ch, T, F, 0.0, flags | PRECISE)
    else:
        r = HipResampler(ch, T, F, flags=flags, fixed=(float(src), float(dst), 0), kernel=kernel) if fixed else HipResampler(ch, T, F, 0.0, flags, kernel=kernel)
    r.advance(T / 2)
    outs, pos = [], 0
    for n in blocks:
        u, g, y = r.process(x[pos:pos + n], int(n * ratio) + 4000, 0.0 if fixed else ratio)
        assert u == n
        if kernel != "oracle":
            assert r.last_kernel() == 2
        outs.append(np.array(y).copy())
        pos += n
    outs.append(np.array(r.process(None, 8000, ratio, flush=True)[2]).copy())
    return np.concatenate(outs)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_split_kernel_meets_the_bar_and_does_not_depend_on_the_order_of_arrival(case):
    y = _play(case, 8)
    t = _play(case, "oracle")
    s = _play(case, 6)
    assert y.shape == t.shape == s.shape
    assert tolerance_ok(y, t)[0]
    assert tolerance_ok(y, s)[0]
    # no worse than the unsplit kernel against the truth (rms)
    e_split, e_stream = np.sqrt(np.mean((y.astype(np.float64) - t) ** 2)), np.sqrt(np.mean((s.astype(np.float64) - t) ** 2))
    assert e_split <= 1.1 * e_stream + 1e-12, (e_split, e_stream)
    assert np.array_equal(y.view(np.uint32), _play(case, 8).view(np.uint32))           # order of arrival does not matter


def test_the_library_takes_the_split_kernel_where_it_was_measured_to_win_and_only_there():
    """8 ch x 988 taps: the 32,768-frame call (70 tiles: half the CUs idle through a K walk) runs in three parts — the forced split kernel's
    bits; the 65,536-frame call (140 tiles: two period groups per XCD, 20 tiles — their parts would need a second round) does not — the streaming kernel's bits"""
    ch, T = 8, 988
    ratio = 48000 / 44100
    for frames, same_as in ((32768, 8), (65536, 6)):
        x, _ = noise(frames * ch, state=0xC11FF | 1)
        x = x.reshape(frames, ch)
        outs = {}
        for kernel in (0, 8, 6):
            r = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=kernel); r.advance(T / 2)
            u, g, y = r.process(x, int(frames * ratio) + 4000, ratio)
            assert r.last_kernel() == 2
            outs[kernel] = np.array(y).copy()
        assert np.array_equal(outs[0].view(np.uint32), outs[same_as].view(np.uint32)), frames
        assert tolerance_ok(outs[0], outs[6])[0]
