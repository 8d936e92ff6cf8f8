"""resampleProcessPlanarDevice on big calls: through the context's interleaved staging (two transposing copies on the device) onto the
matrix-core path — bit for bit the interleaved call's output, with either side or both planar; small calls take the planes as they come
(the general kernel), as before."""
import numpy as np
import pytest
import torch

import audio_resampler_amd as A
from _oracle import noise, BH, INTERP

pytestmark = pytest.mark.gpu
R = 48000 / 44100


@pytest.mark.parametrize("ch,T,sizes", [(8, 988, (300000, 5000, 120000)), (2, 380, (700000, 300000)), (6, 512, (200000, 90000))])
def test_big_planar_device_calls_equal_interleaved_ones(ch, T, sizes):
    x, _ = noise(sum(sizes) * ch, state=ch * 31 + T); x = x.reshape(-1, ch)
    a = A.Resampler(ch, T, T, 0.0, BH | INTERP); a.advance(T / 2)
    b = A.Resampler(ch, T, T, 0.0, BH | INTERP); b.advance(T / 2)
    c = A.Resampler(ch, T, T, 0.0, BH | INTERP); c.advance(T / 2)
    pos = 0
    for n in sizes:
        cap = int(n * R) + 4000
        xi = torch.from_numpy(np.ascontiguousarray(x[pos:pos + n])).cuda()
        xp = torch.from_numpy(np.ascontiguousarray(x[pos:pos + n].T)).cuda()
        yi = torch.zeros(cap, ch, device="cuda"); yp = torch.zeros(ch, cap, device="cuda"); ym = torch.zeros(cap, ch, device="cuda")
        ua, ga = a.process_device(xi, n, yi, cap, R)
        ub, gb = b.process_planar_device(xp, n, n, yp, cap, cap, R)
        uc, gc = c.process_planar_device(xp, n, n, ym, 0, cap, R)          # (planar in, interleaved out)
        a.synchronize(); b.synchronize(); c.synchronize(); torch.cuda.synchronize()
        assert (ua, ga) == (ub, gb) == (uc, gc) and ua == n
        assert a.last_kernel() == b.last_kernel() == c.last_kernel(), (n, a.last_kernel(), b.last_kernel())
        ref = yi[:ga].cpu().numpy()
        assert np.array_equal(ref.view(np.uint32), np.ascontiguousarray(yp[:, :gb].cpu().numpy().T).view(np.uint32)), n
        assert np.array_equal(ref.view(np.uint32), ym[:gc].cpu().numpy().view(np.uint32)), n
        pos += n
    assert a.last_kernel() == 2


def test_big_planar_device_calls_on_a_sharded_context(monkeypatch):
    """the shards of a multi-device context take their runs of planes the same way"""
    monkeypatch.setenv("ARTAMD_SHARDS", "4")
    ch, T, n = 16, 988, 200000
    x, _ = noise(n * ch, state=99); x = x.reshape(-1, ch)
    a = A.Resampler(ch, T, T, 0.0, BH | INTERP); a.advance(T / 2)
    b = A.Resampler(ch, T, T, 0.0, BH | INTERP | A.RESAMPLE_MULTITHREADED); b.advance(T / 2)
    assert len(b.shards()) == 4
    cap = int(n * R) + 4000
    xi = torch.from_numpy(x).cuda(); xp = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    yi = torch.zeros(cap, ch, device="cuda"); yp = torch.zeros(ch, cap, device="cuda")
    ua, ga = a.process_device(xi, n, yi, cap, R)
    ub, gb = b.process_planar_device(xp, n, n, yp, cap, cap, R)
    a.synchronize(); b.synchronize(); torch.cuda.synchronize()
    assert (ua, ga) == (ub, gb) and a.last_kernel() == b.last_kernel() == 2
    assert np.array_equal(yi[:ga].cpu().numpy().view(np.uint32), np.ascontiguousarray(yp[:, :gb].cpu().numpy().T).view(np.uint32))


@pytest.mark.parametrize("n", [200000, 3000])
def test_mixed_layout_device_calls_on_a_sharded_context(monkeypatch, n):
    """one side planar, the other interleaved (a pitch of 0), on a context whose channels are spread over shards: the interleaved side is
    a strided slice of the stream-wide frames for every shard — big calls (matrix path) and small ones (the planes as they come)"""
    monkeypatch.setenv("ARTAMD_SHARDS", "4")
    ch, T = 16, 988
    x, _ = noise(n * ch, state=123 + n); x = x.reshape(-1, ch)
    a = A.Resampler(ch, T, T, 0.0, BH | INTERP); a.advance(T / 2)
    b = A.Resampler(ch, T, T, 0.0, BH | INTERP | A.RESAMPLE_MULTITHREADED); b.advance(T / 2)
    c = A.Resampler(ch, T, T, 0.0, BH | INTERP | A.RESAMPLE_MULTITHREADED); c.advance(T / 2)
    assert len(b.shards()) == 4 and len(c.shards()) == 4
    cap = int(n * R) + 4000
    xi = torch.from_numpy(x).cuda(); xp = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    yi = torch.zeros(cap, ch, device="cuda"); yb = torch.full((cap, ch), 7.0, device="cuda"); yc = torch.full((ch, cap), 7.0, device="cuda")
    ua, ga = a.process_device(xi, n, yi, cap, R)
    ub, gb = b.process_planar_device(xp, n, n, yb, 0, cap, R)              # planar in, interleaved out
    uc, gc = c.process_planar_device(xi, 0, n, yc, cap, cap, R)            # interleaved in, planar out
    a.synchronize(); b.synchronize(); c.synchronize(); torch.cuda.synchronize()
    assert (ua, ga) == (ub, gb) == (uc, gc) and ua == n and ga > 0
    ref = yi[:ga].cpu().numpy()
    if n >= 100000:          # same kernel, same bits; small calls of a shard run the general kernel on planes: inside the bar, other bits
        assert np.array_equal(ref.view(np.uint32), yb[:gb].cpu().numpy().view(np.uint32))
        assert np.array_equal(ref.view(np.uint32), np.ascontiguousarray(yc[:, :gc].cpu().numpy().T).view(np.uint32))
    else:
        assert np.abs(ref - yb[:gb].cpu().numpy()).max() <= 2.0 ** -22 and np.abs(ref - yc[:, :gc].cpu().numpy().T).max() <= 2.0 ** -22
    assert float(yb[gb:].min()) == 7.0 and float(yc[:, gc:].min()) == 7.0   # nothing written past the frames made
