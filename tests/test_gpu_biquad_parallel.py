"""GPU (-m gpu): the time-parallel biquad cascade (pcm_kernels.hip, biquad_spec_kernel) — speculative chunks with exact
verification — against the oracle's serial recurrence (reference biquad.c:106-163): every bit, every filter state, for the
filters ART designs (art.c:736-760), narrow ones (which take the serial kernels), higher orders and awkward signals."""
import ctypes as C

import numpy as np
import pytest
import torch

import audio_resampler_amd as A
from _oracle import load_oracle, noise, f32p, Biquad as OBiquad, BiquadCoeffs as OCoeffs

pytestmark = pytest.mark.gpu


def _design(kind, freq, gain=1.0):
    L, OL = A.lib(), load_oracle()
    co, oc = A.BiquadCoefficients(), OCoeffs()
    (L.biquad_lowpass if kind == "lp" else L.biquad_highpass)(C.byref(co), freq)
    (OL.ora_biquad_lowpass if kind == "lp" else OL.ora_biquad_highpass)(C.byref(oc), freq)
    return co, oc, gain


def _run_bank(x, designs, lengths):
    """x [frames, ch]; designs[k][s] = (coeffs, oracle coeffs, gain).  Returns (repairs, chunks) after asserting equality."""
    L, OL = A.lib(), load_oracle()
    frames, ch = x.shape
    nsec = len(designs[0])
    secs = (A.Biquad * (ch * nsec))()
    osecs = [[OBiquad() for _ in range(nsec)] for _ in range(ch)]
    for k in range(ch):
        for s in range(nsec):
            co, oc, g = designs[k][s]
            L.biquad_init(C.byref(secs[k * nsec + s]), C.byref(co), g)
            OL.ora_biquad_init(C.byref(osecs[k][s]), C.byref(oc), g)
    bank = A.BiquadBank(secs, ch, nsec)
    want = x.copy()
    d = torch.from_numpy(x.copy()).cuda()
    pos = 0
    for n in lengths:
        bank.apply_device(d[pos:pos + n], n)
        view = want[pos:pos + n]
        for k in range(ch):
            for s in range(nsec):
                OL.ora_biquad_buffer(C.byref(osecs[k][s]), C.cast(view.ctypes.data + 4 * k, f32p), n, ch)
        pos += n
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(np.argmax((got != want).any(axis=1)))
    state = bank.read()
    hist = lambda q, arr: [arr[(q.index - i) & 3] for i in range(4)]
    for k in range(ch):
        for s in range(nsec):
            a, b = state[k * nsec + s], osecs[k][s]
            assert hist(a, a.x) == hist(b, b.x) and hist(a, a.y) == hist(b, b.y), (k, s)
    return bank.repairs()


def test_art_prefilter_cascade_at_bench_size():
    """config C's pre-filter: 8 channels x 2 cascaded low-pass sections at 0.45 x 44.1k / 96k, a million frames in three calls"""
    ch, frames = 8, 1 << 20
    x, _ = noise(frames * ch)
    x = x.reshape(frames, ch)
    d = _design("lp", 44100 * 0.45 / 96000)
    repairs = _run_bank(x, [[d, d]] * ch, [400000, 1 << 19, frames - 400000 - (1 << 19)])
    print("chunks recomputed:", repairs)
    assert repairs < 200            # of ~37,000 chunks: the speculation essentially always rejoins


@pytest.mark.parametrize("kind,freq", [("lp", 0.45), ("lp", 0.1), ("lp", 0.03), ("lp", 0.004), ("hp", 0.02), ("hp", 0.3), ("hp", 0.0005)])
def test_cutoffs_from_wide_to_narrow(kind, freq):
    """wide filters forget in tens of frames, 0.03 needs a few hundred, 0.004 and below never within the cap (serial kernels)"""
    ch, frames = 5, 150000
    x, _ = noise(frames * ch, state=0x5EED5EED5EED | 1)
    x = x.reshape(frames, ch)
    d = _design(kind, freq, 0.9)
    _run_bank(x, [[d, d]] * ch, [70001, 79999])


def test_mixed_sections_orders_and_counts():
    """one to four sections per channel, orders 1-4 (hand-made stable coefficient sets), different filters per channel"""
    OL = load_oracle()
    sets = {1: dict(a0=0.2, a1=0.15, b1=-0.5),
            2: dict(a0=0.2, a1=0.15, a2=0.1, b1=-0.5, b2=0.2),
            3: dict(a0=0.2, a1=0.15, a2=0.1, a3=-0.05, b1=-0.5, b2=0.2, b3=-0.1),
            4: dict(a0=0.2, a1=0.15, a2=0.1, a3=-0.05, a4=0.02, b1=-0.5, b2=0.2, b3=-0.1, b4=0.03)}
    frames = 60000
    for nsec in (1, 2, 3, 4):
        ch = 6
        x, _ = noise(frames * ch, state=(0xABCDEF + nsec) | 1)
        x = x.reshape(frames, ch)
        designs = []
        for k in range(ch):
            row = []
            for s in range(nsec):
                kw = sets[1 + (k + s) % 4]
                row.append((A.BiquadCoefficients(**kw), OCoeffs(**kw), 0.8))
            designs.append(row)
        _run_bank(x, designs, [frames])


def test_awkward_signals():
    """full-scale square wave, silence (the state decays through denormals), DC steps, an impulse train"""
    ch, frames = 4, 120000
    t = np.arange(frames)
    x = np.zeros((frames, ch), np.float32)
    x[:, 0] = np.where((t // 37) % 2 == 0, 1.0, -1.0)
    x[20000:60000, 1] = 0.75
    x[::1001, 2] = 1.0
    x[:30000, 3] = noise(30000)[0] * 2e-20       # tiny values, then silence
    d1, d2 = _design("lp", 0.2), _design("hp", 0.05)
    _run_bank(x, [[d1, d2]] * ch, [frames])


def test_host_api_strided_calls_use_the_parallel_form():
    """ART's calling pattern at its block size (16,384 frames, one section of one channel per call, art.c:1011-1017)"""
    L, OL = A.lib(), load_oracle()
    ch, frames, blocks = 2, 16384, 3
    x, _ = noise(frames * blocks * ch, state=0x1234567 | 1)
    buf = x.reshape(frames * blocks, ch).copy()
    want = buf.copy()
    co, oc, _ = _design("lp", 44100 * 0.45 / 96000)
    filt = [[A.Biquad(), A.Biquad()] for _ in range(ch)]
    ofilt = [[OBiquad(), OBiquad()] for _ in range(ch)]
    for k in range(ch):
        for s in range(2):
            L.biquad_init(C.byref(filt[k][s]), C.byref(co), 1.0)
            OL.ora_biquad_init(C.byref(ofilt[k][s]), C.byref(oc), 1.0)
    before = L.artamdBiquadRepairs()
    for blk in range(blocks):
        view, oview = buf[blk * frames:(blk + 1) * frames], want[blk * frames:(blk + 1) * frames]
        for k in range(ch):
            for s in range(2):
                L.biquad_apply_buffer(C.byref(filt[k][s]), C.cast(view.ctypes.data + 4 * k, f32p), frames, ch)
                OL.ora_biquad_buffer(C.byref(ofilt[k][s]), C.cast(oview.ctypes.data + 4 * k, f32p), frames, ch)
    assert np.array_equal(buf.view(np.uint32), want.view(np.uint32))
    for k in range(ch):
        for s in range(2):
            assert bytes(filt[k][s])[:80] == bytes(ofilt[k][s])[:80]
    print("chunks recomputed (host calls):", L.artamdBiquadRepairs() - before)


@pytest.mark.parametrize("warmup", ["2", "20"])
def test_verification_and_repair_when_the_speculation_fails(monkeypatch, warmup):
    """ARTAMD_BIQUAD_WARMUP forces a warm-up far too short for the filters: nearly every chunk starts from a state that has not
    converged, the boundary check catches each one and the chunks are recomputed from the exact state — same bits, many repairs"""
    monkeypatch.setenv("ARTAMD_BIQUAD_WARMUP", warmup)
    sets = {1: dict(a0=0.2, a1=0.15, b1=-0.5), 2: dict(a0=0.2, a1=0.15, a2=0.1, b1=-0.5, b2=0.2),
            3: dict(a0=0.2, a1=0.15, a2=0.1, a3=-0.05, b1=-0.5, b2=0.2, b3=-0.1),
            4: dict(a0=0.2, a1=0.15, a2=0.1, a3=-0.05, a4=0.02, b1=-0.5, b2=0.2, b3=-0.1, b4=0.03)}
    frames, ch = 40000, 6
    total = 0
    for nsec in (1, 2, 3, 4):
        x, _ = noise(frames * ch, state=(0x51DE + nsec) | 1)
        x = x.reshape(frames, ch)
        designs = [[(A.BiquadCoefficients(**sets[1 + (k + s) % 4]), OCoeffs(**sets[1 + (k + s) % 4]), 0.8) for s in range(nsec)] for k in range(ch)]
        total += _run_bank(x, designs, [15000, 25000])
    d = _design("lp", 44100 * 0.45 / 96000)
    x, _ = noise(200000 * 8)
    total += _run_bank(x.reshape(200000, 8), [[d, d]] * 8, [200000])
    assert total > 1000, total
