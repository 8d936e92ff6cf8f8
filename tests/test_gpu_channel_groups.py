"""Channel counts the matrix-core kernels are not compiled for (3, 5, 6, 7, 12, 24, 33, 64 ...) run their matrix-path launches in groups of a
compiled width (fir_dispatch.hip, fir_in_groups): against the oracle like every other stream, and — a channel's arithmetic depending neither
on its group's width nor on its neighbours — bit for bit what the same channel gives inside a stream of a compiled width."""
import numpy as np
import pytest

import audio_resampler_amd as A
from _hip import HipResampler, tolerance_ok
from _oracle import noise, OracleResampler, BH, INTERP, PRECISE

pytestmark = pytest.mark.gpu
R = 48000 / 44100


def _run(r, x, sizes, ratio=R):
    outs = []; pos = 0; kinds = []
    for n in sizes:
        u, g, y = r.process(x[pos:pos + n], int(n * ratio) + 4000, ratio)
        assert u == n
        outs.append(np.array(y).copy()); pos += n; kinds.append((int(r.last_kernel()), int(r.fixed_point()[0])))
    return np.concatenate(outs), kinds


@pytest.mark.parametrize("ch,wide,T", [(6, 8, 988), (3, 4, 988), (12, 16, 512), (5, 8, 380), (33, 64, 988), (24, 32, 156)])
def test_a_channel_is_the_same_in_a_group_and_in_a_stream_of_a_compiled_width(ch, wide, T):
    sizes = [150000, 3000, 60000] if ch < 33 else [40000, 3000, 20000]
    x, _ = noise(sum(sizes) * wide, state=ch * 77 + 1); x = x.reshape(-1, wide)
    if wide > 32:                                               # (a 64-channel stream is two groups of 32 itself: compare with a 32-channel one)
        # (kernel preference 7 — the fixed-point kernel wherever it can run — in all three: the library's own choice between the f32 and the fixed-point kernels
        # follows the STREAM's size (a model of both kernels' times since round 5), which a 33-, a 32- and a 64-channel stream do not share at every call size)
        narrow = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=7); narrow.advance(T / 2)
        ref = HipResampler(32, T, T, 0.0, BH | INTERP, kernel=7); ref.advance(T / 2)
        ya, ka = _run(narrow, np.ascontiguousarray(x[:, :ch]), sizes)
        yb, kb = _run(ref, np.ascontiguousarray(x[:, :32]), sizes)
        assert np.array_equal(ya[:, :32].view(np.uint32), yb.view(np.uint32)), (ka, kb)
        assert all(k[0] == 2 for k in ka)
        # the channels of the LAST group (one channel wide here, copied into a 4-wide buffer): what the same channels give in the
        # second group of a 64-channel stream (the kernel family follows the STREAM's size, so a stream of a similar size is the reference)
        full = HipResampler(wide, T, T, 0.0, BH | INTERP, kernel=7); full.advance(T / 2)
        yc, kc = _run(full, x, sizes)
        # (like for like: the kernel family AND the f32 kernels' K split follow the stream's size, which a 33- and a 64-channel stream do
        # not share at every call size; the first call runs the fixed-point kernel in both — whose bits depend on no launch geometry)
        assert ka[0] == kc[0] == (2, 1), (ka, kc)
        frames = int(sizes[0] * R) - 8                                 # (the whole first call, less the few frames its end may shift)
        assert np.array_equal(ya[:frames, 32:ch].view(np.uint32), yc[:frames, 32:ch].view(np.uint32)), (ka, kc)
        return
    narrow = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=7 if T >= 512 else 2); narrow.advance(T / 2)
    ref = HipResampler(wide, T, T, 0.0, BH | INTERP, kernel=7 if T >= 512 else 2); ref.advance(T / 2)
    ya, ka = _run(narrow, np.ascontiguousarray(x[:, :ch]), sizes)
    yb, kb = _run(ref, x, sizes)
    assert ka == kb and all(k[0] == 2 for k in ka), (ka, kb)     # (the matrix path, the same kernel family, in both)
    assert np.array_equal(ya.view(np.uint32), yb[:, :ch].view(np.uint32)), (ch, ka)


@pytest.mark.parametrize("ch", [3, 6, 7, 12, 40])
def test_grouped_launches_against_the_oracle(ch):
    T = 380
    big = int(3.5e8 / (ch * T))                                 # (frames from which such a stream's calls take the matrix path, with room)
    sizes = [big, 2500, big // 2]
    x, _ = noise(sum(sizes) * ch, state=ch * 13 + 5); x = x.reshape(-1, ch)
    r = HipResampler(ch, T, T, 0.0, BH | INTERP); r.advance(T / 2)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE); o.advance(T / 2)
    y, kinds = _run(r, x, sizes)
    assert (2, 0) in kinds or (2, 1) in kinds, kinds             # (the big calls took the matrix path)
    yo = []; pos = 0
    for n in sizes:
        u, g, yy = o.process(x[pos:pos + n], int(n * R) + 4000, R); yo.append(np.array(yy).copy()); pos += n
    ok, worst, rms = tolerance_ok(y, np.concatenate(yo))
    assert ok, (ch, worst, rms)


def test_channel_groups_in_the_8_byte_build():
    """libartamd64: a 6-channel stream's big calls on the fp64 matrix kernel in a group of 8 — the same bits per channel as in an 8-channel stream, and the oracle's bar"""
    W = A.wide()
    import _oracle
    Wo = _oracle.wide()
    T, sizes = 380, [200000, 3000, 60000]
    x, _ = noise(sum(sizes) * 8, state=4242); x = x.reshape(-1, 8).astype(np.float64)
    a = W.Resampler(6, T, T, 0.0, BH | INTERP); a.advance(T / 2)
    b = W.Resampler(8, T, T, 0.0, BH | INTERP); b.advance(T / 2)
    o = Wo.OracleResampler(6, T, T, 0.0, BH | INTERP); o.advance(T / 2)
    pos = 0
    for n in sizes:
        cap = int(n * R) + 4000
        ua, ga, ya = a.process(np.ascontiguousarray(x[pos:pos + n, :6]), cap, R)
        ub, gb, yb = b.process(x[pos:pos + n], cap, R)
        uo, go, yo = o.process(np.ascontiguousarray(x[pos:pos + n, :6]), cap, R)
        pos += n
        assert (ua, ga) == (ub, gb) == (uo, go)
        if a.last_kernel() == b.last_kernel():
            assert np.array_equal(np.array(ya).view(np.uint64), np.array(yb)[:, :6].view(np.uint64)), n
        assert np.abs(np.array(ya) - np.array(yo)).max() <= 2.0 ** -44, n
    assert a.last_kernel() in (1, 2)
