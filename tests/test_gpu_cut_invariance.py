"""GPU (-m gpu): a fixed-ratio stream (resampleFixedRatioInit, the reference's `-e` mode) on the f32 streaming kernel (kernel preference 6) gives
the same bits however its input is cut into calls — the reference's property (resampler.c:323-335, 529-535; SURVEY appendix C: one checksum at
every -b) on a matrix-core kernel, not only under RESAMPLE_STRICT_ORDER.  What carries it: the rows kept across calls are built once, for the
stream's canonical period, and every launch's tiles are anchored on that period — an output lands in the same tile row, walks the same K chunks
and is flushed at the same points whichever call brought it.  Holds where every launch is the streaming kernel's: calls of at least one period
of outputs (1,000 frames here), host-pointer calls of any length from there, device-pointer calls whose input is aligned to a frame of one or
two channels / to 16 bytes from four channels on (random cuts of a stereo stream start at odd frames of the caller's buffer).  (The library's own choice, preference 0, takes other kernels for other call sizes: within the
parity bar, not the same bits.)"""
import hashlib
import numpy as np
import pytest

import audio_resampler_amd as A
from _hip import HipResampler
from _oracle import noise, BH, INTERP

pytestmark = pytest.mark.gpu

TOTAL = 300000
STREAMS = [
    # (channels, taps, max filters, flags): fixed 44.1k -> 48k
    (2, 380, 380, BH | INTERP),        # resolves to 320 filters = 2 per phase: nearest filter, slots ON input samples in every period
    (8, 988, 988, BH | INTERP),        # 988 filters, 160 phases: interpolating rows
    (1, 380, 380, BH | INTERP),
    (4, 256, 256, BH | INTERP),
]


def _cuts(kind, rng):
    if kind == "one":
        return [TOTAL]
    if kind.isdigit():
        k = int(kind); c = [k] * (TOTAL // k)
        return c + ([TOTAL - sum(c)] if TOTAL - sum(c) else [])
    c = []
    while sum(c) < TOTAL:
        c.append(int(min(rng.integers(1000, 90000), TOTAL - sum(c))))
    if c[-1] < 1000 and len(c) > 1:
        c[-2] += c[-1]; c.pop()
    return c


def _play(stream, cuts, x, device):
    ch, T, F, flags = stream
    r = HipResampler(ch, T, F, flags=flags, fixed=(44100.0, 48000.0, 0), kernel=6); r.advance(T / 2)
    outs, pos = [], 0
    if device:
        import torch
        d_x = torch.from_numpy(x).cuda()
    for n in cuts:
        cap = int(n * 48000 / 44100) + 4000
        if device:
            d_y = torch.zeros(cap, ch, device="cuda")
            u, g = r.process_device(d_x[pos:pos + n], n, d_y, cap, 0.0)
            y = d_y[:g].cpu().numpy()
        else:
            u, g, y = r.process(x[pos:pos + n], cap, 0.0)
        assert u == n and r.last_kernel() == 2
        outs.append(np.array(y).copy()); pos += n
    return np.concatenate(outs)


@pytest.mark.parametrize("stream", STREAMS, ids=[f"c{s[0]}_t{s[1]}" for s in STREAMS])
def test_fixed_ratio_output_does_not_depend_on_the_cuts(stream):
    ch = stream[0]
    x, _ = noise(TOTAL * ch, state=0xC075 | 1)
    x = x.reshape(TOTAL, ch)
    rng = np.random.default_rng(11)
    ref = _play(stream, [TOTAL], x, device=True)
    want = hashlib.sha256(ref.tobytes()).hexdigest()
    for kind, device in (("65536", True), ("16384", True), ("4096", True), ("1000", True), ("random", False), ("random", True), ("random", True), ("16384", False)):
        y = _play(stream, _cuts(kind, rng), x, device)
        assert y.shape == ref.shape, (kind, device, y.shape, ref.shape)
        assert hashlib.sha256(y.tobytes()).hexdigest() == want, (kind, device, int(np.count_nonzero(y.view(np.uint32) != ref.view(np.uint32))))
