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
