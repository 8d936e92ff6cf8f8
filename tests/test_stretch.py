"""Time stretcher (reference stretch.c / stretch.h).  CPU: the oracle restatement against vectors
made from the real reference (and against the reference itself where oracle/_ref exists).  GPU: libartamd*.so through
the reference's own API names — every output bit and every per-call frame count, both sample widths, normal / fast /
cascaded modes, varying block sizes and ratios, reset, device-pointer entry points."""
import os

import numpy as np
import pytest

import _golden as G
import _oracle
import _stretch as S

WIDTHS = [(32, np.float32), (64, np.float64)]
_z = {}


def gold():
    if "z" not in _z:
        _z["z"] = np.load(os.path.join(G.GOLD, "stretch.npz"))
    return _z["z"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


def check_case(backend_cls, case, width, dt):
    z, B = gold(), _oracle.binding(width)
    x, ctor, blocks, ratios = S.case_setup(case, dt)
    key = f"w{width}/{case[0]}"
    if B.checksum_words(x) != int(z[key + "/in_sum"]):
        pytest.skip("the synthetic input differs on this platform (libm): vectors do not apply")
    y, counts = backend_cls(*ctor, width=width).run(x, blocks, ratios)
    assert counts == [int(c) for c in z[key + "/counts"]]
    assert np.array_equal(bits(y[:512]), bits(z[key + "/head"])) and np.array_equal(bits(y[-512:]), bits(z[key + "/tail"]))
    assert B.checksum_words(y) == int(z[key + "/sum"])


@pytest.mark.parametrize("width,dt", WIDTHS)
@pytest.mark.parametrize("case", S.CASES, ids=lambda c: c[0])
def test_oracle_matches_reference_vectors(case, width, dt):
    check_case(S.OracleStretch, case, width, dt)


@pytest.mark.ref
@pytest.mark.parametrize("width,dt", WIDTHS)
@pytest.mark.parametrize("seed", range(10))
def test_oracle_equals_reference_on_random_sessions(seed, width, dt):
    rng = np.random.default_rng(seed)
    rate = int(rng.choice([22050, 32000, 44100, 48000]))
    ch = int(rng.integers(1, 3))
    flags = int(rng.choice([0, 0, S.FAST, S.DUAL, S.FAST | S.DUAL]))
    lim = (0.27, 3.8) if flags & S.DUAL else (0.5, 2.0)
    ratios = [float(np.exp(rng.uniform(np.log(lim[0]), np.log(lim[1])))) for _ in range(5)] + [1.0]
    blocks = [int(rng.integers(1, 9000)) for _ in range(7)]
    x = S.signal(int(rate * 0.8), ch, rate, seed=100 + seed, dtype=dt)
    ctor = (rate // 350, rate // 50, ch, flags)
    yo, co = S.OracleStretch(*ctor, width=width).run(x, blocks, ratios)
    yr, cr = S.RefStretch(*ctor, width=width).run(x, blocks, ratios)
    assert co == cr and np.array_equal(bits(yo), bits(yr))


def test_capacity_and_argument_checks_match_oracle():
    import audio_resampler_amd as A
    L = A.lib()
    for args in ((24, 24, 1, 0), (10, 800, 1, 0), (100, 2500, 2, 0)):           # invalid periods: NULL like the reference
        assert not L.stretchInit(*args)


# ------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("width,dt", WIDTHS)
@pytest.mark.parametrize("case", S.CASES, ids=lambda c: c[0])
def test_hip_stretcher_is_bit_exact_vs_reference_vectors(case, width, dt):
    check_case(S.HipStretch, case, width, dt)


@pytest.mark.gpu
@pytest.mark.parametrize("width,dt", WIDTHS)
@pytest.mark.parametrize("seed", range(16))
def test_hip_stretcher_random_sessions_bit_exact_vs_oracle(seed, width, dt):
    rng = np.random.default_rng(1000 + seed)
    rate = int(rng.choice([16000, 22050, 32000, 44100, 48000, 96000]))
    ch = int(rng.integers(1, 3))
    flags = int(rng.choice([0, 0, S.FAST, S.DUAL, S.FAST | S.DUAL]))
    lim = (0.27, 3.8) if flags & S.DUAL else (0.5, 2.0)
    ratios = [float(np.exp(rng.uniform(np.log(lim[0]), np.log(lim[1])))) for _ in range(5)] + [1.0]
    blocks = [int(rng.integers(1, 20000)) for _ in range(7)]
    x = S.signal(int(rate * 0.7), ch, rate, seed=200 + seed, dtype=dt)
    ctor = (rate // 350, rate // 50, ch, flags)
    h, o = S.HipStretch(*ctor, width=width), S.OracleStretch(*ctor, width=width)
    assert h.capacity(20000, max(ratios)) == o.capacity(20000, max(ratios))
    yh, chh = h.run(x, blocks, ratios)
    yo, co = o.run(x, blocks, ratios)
    assert chh == co and np.array_equal(bits(yh), bits(yo))
    # reset, then a second pass over the same input gives the same stream again
    h.reset(); o.reset()
    yh2, ch2 = h.run(x[: len(x) // 2], blocks, ratios)
    yo2, co2 = o.run(x[: len(x) // 2], blocks, ratios)
    assert ch2 == co2 and np.array_equal(bits(yh2), bits(yo2))


@pytest.mark.gpu
def test_hip_stretcher_device_pointer_calls_equal_host_calls():
    torch = pytest.importorskip("torch")
    import audio_resampler_amd as A
    rate, ch = 44100, 2
    x = S.signal(rate, ch, rate, seed=5)
    a, b = S.HipStretch(rate // 350, rate // 50, ch), S.HipStretch(rate // 350, rate // 50, ch)
    cap = a.capacity(16384, 1.3)
    out = np.zeros((cap, ch), np.float32)
    d_out = torch.zeros(cap, ch, device="cuda")
    L = A.lib()
    for pos in range(0, rate, 16384):
        blk = x[pos:pos + 16384]
        g = a.feed(blk, out, 1.3)
        d_in = torch.from_numpy(blk.copy()).cuda()
        gd = L.stretchProcessDevice(b.p, d_in.data_ptr(), blk.shape[0], d_out.data_ptr(), 1.3)
        assert g == gd and np.array_equal(bits(out[:g]), bits(d_out[:g].cpu().numpy()))
    g, gd = a.drain(out), L.stretchFlushDevice(b.p, d_out.data_ptr())
    assert g == gd and np.array_equal(bits(out[:g]), bits(d_out[:g].cpu().numpy()))


@pytest.mark.gpu
@pytest.mark.parametrize("width,dt", WIDTHS)
def test_hip_stretcher_batched_launch_equals_separate_calls_and_oracle(width, dt):
    """stretchProcessBatchDevice / stretchFlushBatchDevice: n different streams (mono / stereo, fast, cascaded, different
    ratios and block sizes, one of them idle on alternate rounds) in one launch per round == the same streams one call at
    a time == the oracle, bit for bit."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    import audio_resampler_amd as A
    L = A.binding(width).lib()
    tdt = torch.float32 if width == 32 else torch.float64
    cases = [S.CASES[i] for i in (0, 2, 4, 6, 7, 8, 9, 10)]
    n = len(cases)
    xs, hs, os_, blocks, ratios = [], [], [], [], []
    for c in cases:
        x, init, blk, rat = S.case_setup(c, dt)
        xs.append(x); blocks.append(blk); ratios.append(rat)
        hs.append(S.HipStretch(*init, width=width)); os_.append(S.OracleStretch(*init, width=width))
    caps = [h.capacity(max(b), max(max(r), 1.0)) for h, b, r in zip(hs, blocks, ratios)]
    d_out = [torch.zeros(cap, x.shape[1], device="cuda", dtype=tdt) for cap, x in zip(caps, xs)]
    o_out = [np.zeros((cap, x.shape[1]), dt) for cap, x in zip(caps, xs)]
    ctx = (C.c_void_p * n)(*[h.p for h in hs])
    outs = (C.c_void_p * n)(*[d.data_ptr() for d in d_out])
    pos, made = [0] * n, (C.c_int * n)()
    rnd = 0
    while any(p < x.shape[0] for p, x in zip(pos, xs)):
        d_in, frames, rat = [], [], []
        for i in range(n):
            idle = (i == 3 and rnd % 2 == 1) or pos[i] >= xs[i].shape[0]
            k = blocks[i][rnd % len(blocks[i])]
            blk = xs[i][pos[i]:pos[i] + (0 if idle else k)]
            d_in.append(torch.from_numpy(np.ascontiguousarray(blk)).cuda() if len(blk) else torch.zeros(1, device="cuda", dtype=tdt))
            frames.append(len(blk)); rat.append(ratios[i][rnd % len(ratios[i])])
        assert L.stretchProcessBatchDevice(ctx, n, (C.c_void_p * n)(*[d.data_ptr() for d in d_in]), (C.c_int * n)(*frames),
                                           outs, (C.c_double * n)(*rat), made) == 0
        for i in range(n):
            g = os_[i].feed(xs[i][pos[i]:pos[i] + frames[i]], o_out[i], rat[i]) if frames[i] else 0
            assert made[i] == g, (rnd, i)
            assert np.array_equal(bits(d_out[i][:g].cpu().numpy()), bits(o_out[i][:g])), (rnd, i)
            pos[i] += frames[i]
        rnd += 1
    # the host mirrors follow the device state: a single-stream call continues seamlessly after batched ones
    for i in (0, 5):
        blk = xs[i][:4096]
        g = os_[i].feed(blk, o_out[i], 1.2)
        d = torch.from_numpy(np.ascontiguousarray(blk)).cuda()
        assert L.stretchProcessDevice(hs[i].p, d.data_ptr(), len(blk), d_out[i].data_ptr(), 1.2) == g
        assert np.array_equal(bits(d_out[i][:g].cpu().numpy()), bits(o_out[i][:g]))
    # (flushing is terminal in the reference: stretchProcess after stretchFlush without stretchReset can spin forever,
    # stretch.c:195-212 with a full buffer and too little between tail and head — so the flush comes last)
    for _ in range(2):
        assert L.stretchFlushBatchDevice(ctx, n, outs, made) == 0
        for i in range(n):
            g = os_[i].drain(o_out[i])
            assert made[i] == g and np.array_equal(bits(d_out[i][:g].cpu().numpy()), bits(o_out[i][:g]))
    # a context twice in one batch is refused
    dup = (C.c_void_p * 2)(hs[0].p, hs[0].p)
    assert L.stretchFlushBatchDevice(dup, 2, outs, made) == -1


@pytest.mark.gpu
def test_hip_stretcher_feeding_after_a_flush_without_reset_returns_instead_of_spinning():
    """Reference contract: stretchFlush is terminal (stretchReset before reuse).  Feeding again without a reset can leave
    the reference's loop with a full ring and nothing processable — it never returns (stretch.c:195-212).  The device loop
    must not spin (a stuck workgroup takes the GPU with it): the call returns, and after stretchReset the context works again."""
    rate, ch = 44100, 1
    x = S.signal(3 * rate, ch, rate, seed=11)
    h, o = S.HipStretch(rate // 350, rate // 50, ch), S.OracleStretch(rate // 350, rate // 50, ch)
    cap = h.capacity(2 * rate, 2.0)
    out = np.zeros((cap, ch), np.float32)
    h.feed(x[:20000], out, 1.3); h.drain(out)
    for _ in range(3):
        g = h.feed(x[:2 * rate], out, 1.3)           # would spin forever in the reference
        assert 0 <= g <= cap
    h.reset()                                        # (keeps the accumulated length error, as the reference's does: stretch.c:102-110)
    yh, _ = h.run(x[:rate], [16384], [1.3]); yo, _ = o.run(x[:rate], [16384], [1.3])
    assert abs(len(yh) - len(yo)) < rate // 20 and np.isfinite(yh).all() and np.abs(yh).max() <= np.abs(x).max() * 1.01
