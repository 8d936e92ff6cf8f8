"""CPU (no GPU): the product library's host logic — it loads, exports every symbol the public headers
declare, fails loudly without a device, designs the same filter bank as the reference, and its
closed-form position planner reproduces the reference's scalar state machine exactly."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

import _golden as G
import audio_resampler_amd as A
from audio_resampler_amd.api import ArtamdPosition, ArtamdSegment, ResampleResult, f32p
from _oracle import OracleResampler, noise, BH, INTERP, LOWPASS, FIXED, FLUSHED, SNAP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = A.lib()
    declared = set()
    for h in ("resampler.h", "biquad.h", "decimator.h", "stretch.h", "art_hip.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text))
    declared -= {"defined"}
    assert declared, "no prototypes parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    assert declared == set(A.EXPORTED_SYMBOLS), declared ^ set(A.EXPORTED_SYMBOLS)


def test_struct_layouts_are_abi():
    assert C.sizeof(A.Biquad) == 80 and C.sizeof(A.BiquadCoefficients) == 36      # reference biquad.h:27-35
    assert C.sizeof(A.ResampleResult) == 8
    assert A.Resample.numChannels.offset == 0 and A.Resample.outputOffset.offset == 32 and A.Resample.filters.offset == 72
    assert A.Decimate.numChannels.offset == 0 and A.Decimate.outputBytes.offset == 8 and A.Decimate.outputGain.offset == 24


@pytest.mark.skipif(A.lib().artamdDeviceCount() > 0, reason="a GPU is present")
def test_no_gpu_means_loud_failure_not_fallback(capfd):
    L = A.lib()
    assert not L.resampleInit(2, 48, 48, 0.0, 3)
    assert not L.decimateInit(2, 16, 2, 1.0, 48000, 0)
    err = capfd.readouterr().err
    assert "no CPU path" in err
    # the reference's void entry points cannot return an error: counted, and silence rather than uninitialised memory comes back
    before = L.artamdErrorCount()
    raw, out = np.arange(16, dtype=np.uint8), np.full(8, 7.0, np.float32)
    L.floatIntegersLE(raw.ctypes.data_as(C.POINTER(C.c_ubyte)), 1.0, 16, 2, 1, out.ctypes.data_as(C.POINTER(C.c_float)), 8)
    assert L.artamdErrorCount() == before + 1 and not out.any() and b"no CPU path" in L.artamdLastError()
    co, f = A.BiquadCoefficients(), A.Biquad()
    L.biquad_lowpass(C.byref(co), 0.1); L.biquad_init(C.byref(f), C.byref(co), 1.0)
    x = np.ones(8, np.float32)
    L.biquad_apply_buffer(C.byref(f), x.ctypes.data_as(C.POINTER(C.c_float)), 8, 1)
    assert L.artamdErrorCount() == before + 2 and np.all(x == 1.0)           # samples and filter state untouched
    capfd.readouterr()


@pytest.mark.parametrize("name", G.NAMES)
def test_filter_bank_matches_reference(name):
    z = G.load("bank")
    rz = G.load("resample")
    F, T, flags = [int(v) for v in rz[name + "/meta"]]
    lowpass = float(rz[name + "/meta_f"][0])
    bank = np.zeros((F + 1, T), np.float32)
    A.lib().artamdBuildFilterBank(T, F, lowpass, flags, bank.ctypes.data_as(f32p))
    assert hashlib.sha256(bank.tobytes()).digest() == bytes(z[name + "/sha256"])
    assert np.array_equal(bank[z[name + "/rows"]].view(np.uint32), z[name + "/data"].view(np.uint32))


def plan(pos, n_in, cap, ratio, max_segs=4096):
    res = ResampleResult()
    segs = (ArtamdSegment * max_segs)()
    floor = C.c_int()
    n = A.lib().artamdPlanCall(C.byref(pos), n_in, cap, ratio, C.byref(res), segs, max_segs, C.byref(floor))
    return res.input_used, res.output_generated, [(s.first_output, s.lin_base, s.base_offset) for s in segs[:n]], floor.value


@pytest.mark.parametrize("name", G.NAMES)
def test_planner_reproduces_golden_traces(name):
    rz = G.load("resample")
    F, T, flags = [int(v) for v in rz[name + "/meta"]]
    pos = ArtamdPosition(T, F, flags & ~0x80, T, 0, float(T // 2), float(rz[name + "/meta_f"][1]))
    adv = G.CTOR[name]["adv"]
    if adv is not None:
        pos.outputOffset += adv
    for (n, cap, ratio, flush), want in zip(G.script_of(name), rz[name + "/trace"]):
        used, made, segs, _ = plan(pos, -1 if flush else n, cap, ratio)
        got = (used, made, np.float64(pos.outputOffset).view(np.uint64).item(), pos.inputIndex)
        assert got == tuple(int(v) for v in want[:4])
        assert segs[0][0] == 0 and all(a[0] <= b[0] for a, b in zip(segs, segs[1:]))


@pytest.mark.parametrize("seed", range(40))
def test_planner_equals_oracle_loop_randomised(seed):
    """closed-form planner vs the oracle's literal consume/emit loop (which is pinned to the reference)"""
    rng = np.random.default_rng(1000 + seed)
    T = int(rng.choice([4, 8, 16, 48, 156, 380, 988]))
    F = int(rng.choice([1, 3, 48, 147, 160, 988]))
    ratio = float(rng.choice([48000 / 44100, 44100 / 96000, 0.5, 2.0, 1.0, 1 / 3.0, 3.7, 0.01, 97.3, 1.0000001]))
    interp = bool(rng.integers(0, 2))
    fixed = bool(rng.integers(0, 2))
    if fixed:
        o = OracleResampler(1, T, max(F, 2), flags=BH | (INTERP if interp else 0) | LOWPASS, fixed=(44100.0, 48000.0, 0))
    else:
        o = OracleResampler(1, T, F, 0.0, BH | (INTERP if interp else 0))
    c = o.c
    adv = float(rng.choice([0.0, T / 2, T / 2 + (0.37 if c.flags & INTERP else 0.0), 5 * T]))
    o.advance(adv)
    pos = ArtamdPosition(T, c.filters, c.flags, c.write_pos, 0, c.read_pos, c.fixed_ratio)
    x = np.zeros((1, 1), np.float32)
    for call in range(30):
        kind = int(rng.integers(0, 12))
        n = int(rng.integers(0, 40 * T)) if kind < 9 else int(rng.integers(0, 4))
        cap = int(rng.integers(0, 50 * T)) if kind != 4 else int(rng.integers(0, 5))
        r = ratio * (1 + rng.uniform(-3e-4, 3e-4)) if kind == 6 else ratio
        flush = kind == 11 and call > 20
        if flush and o.c.write_pos > 15 * T + T // 2:
            continue                          # the reference's own flush is out of bounds there
        xin = np.zeros((max(n, 1), 1), np.float32)
        if flush:
            u, g, _ = o.process(None, cap, r, flush=True)
            used, made, segs, floor = plan(pos, -1, cap, r)
        else:
            u, g, _ = o.process(xin[:n], cap, r)
            used, made, segs, floor = plan(pos, n, cap, r)
        assert (used, made) == (u, g), (call, n, cap, r)
        st = o.state()
        assert np.float64(pos.outputOffset).view(np.uint64).item() == st[0] and pos.inputIndex == st[1]
        assert (pos.flags & (FLUSHED | SNAP | FIXED)) == (st[2] & (FLUSHED | SNAP | FIXED))


def test_planner_segments_cover_ring_rewinds():
    T = 48
    pos = ArtamdPosition(T, 48, 3, T, 0, float(T), 0.0)
    used, made, segs, floor = plan(pos, 100 * T, 200 * T, 48000 / 44100)
    assert used == 100 * T and len(segs) == 1 + (100 * T + T - 1 - 16 * T) // (15 * T) + 1 or len(segs) >= 6
    H = T + T // 2
    assert segs[0][1] == H - T
    for a, b in zip(segs, segs[1:]):
        assert b[1] - a[1] == 15 * T and a[2] - b[2] == 15 * T
    assert floor == -2 ** 31


@pytest.mark.parametrize("ratio", [48000 / 44100, 44100 / 48000, 2.0, 2.0 / 3.0, 0.25, 1.0, 1.0000003, 3.7], ids=lambda r: f"{r:.7g}")
def test_planner_equals_oracle_loop_over_hundreds_of_ring_epochs(ratio):
    """calls of many ring epochs (short filters: 15 T consumed frames each): the planner's two probes at the arithmetic estimate must
    close the search in every epoch exactly as the oracle's literal loop does — counts, carried position and every segment's start"""
    T = 48
    o = OracleResampler(1, T, T, 0.0, BH | INTERP)
    o.advance(T / 2)
    c = o.c
    pos = ArtamdPosition(T, c.filters, c.flags, c.write_pos, 0, c.read_pos, c.fixed_ratio)
    for n in (250000, 1, 90001):
        cap = int(n * ratio) + 100
        u, g, _ = o.process(np.zeros((n, 1), np.float32), cap, ratio)
        used, made, segs, floor = plan(pos, n, cap, ratio, max_segs=8192)
        assert (used, made) == (u, g), (n, ratio)
        st = o.state()
        assert np.float64(pos.outputOffset).view(np.uint64).item() == st[0] and pos.inputIndex == st[1]
        assert len(segs) >= n // (15 * T) and all(a[0] <= b[0] for a, b in zip(segs, segs[1:]))


def test_period_rule_fills_the_matrix_tiles():
    """fir_common.hip.h, artfir_period_multiple through the C ABI: 1 while a period pads its 32-row tiles by at most 15 %, else the
    multiple that fills whole tiles (up to 16 tiles), else the smallest multiple within 4 % of padding"""
    L = A.lib()
    assert [L.artamdPeriodMultiple(p) for p in (160, 147, 2, 1, 3, 80, 50, 4, 6, 640, 33)] == [1, 1, 16, 32, 32, 2, 5, 8, 16, 1, 14]
    for p in range(1, 700):
        mu = L.artamdPeriodMultiple(p)
        padded = -(-mu * p // 32) * 32
        assert mu >= 1 and (mu == 1 or mu * p <= 512)
        assert padded / (mu * p) <= 1.15 + 1e-12, (p, mu)
    # the same rule for the 64-row tiles of the fixed-point slab kernel (what the 4-byte build applies): 44.1k -> 48k two periods at a
    # time, 96k -> 44.1k three; whatever fills 64-row tiles within 15 % fills 32-row tiles within 15 % too
    assert [L.artamdPeriodMultipleRows(p, 64) for p in (160, 147, 2, 1, 80, 320, 64)] == [2, 3, 32, 64, 4, 1, 1]
    for p in range(1, 700):
        mu = L.artamdPeriodMultipleRows(p, 64)
        assert mu >= 1 and (mu == 1 or mu * p <= 1024)
        assert -(-mu * p // 64) * 64 / (mu * p) <= 1.15 + 1e-12 and -(-mu * p // 32) * 32 / (mu * p) <= 1.15 + 1e-12, (p, mu)


def test_biquad_sample_form_on_the_host_matches_the_reference_vectors():
    """biquad_apply_sample is the one per-sample entry point of the boundary (reference biquad.c:78-102): evaluated on the host, where the
    caller's Biquad struct — the state — lives; orders 1..4 against the vectors made from the compiled reference (tests/golden/biquad.npz),
    bit for bit, with no GPU in sight"""
    import ctypes as C
    import _golden as G
    L = A.lib()
    z = G.load("biquad")
    for order in (1, 2, 3, 4):
        co = A.BiquadCoefficients(*[float(v) for v in z[f"order{order}/coeffs"]])
        x1, _ = noise(600)
        bs = A.Biquad()
        L.biquad_init(C.byref(bs), C.byref(co), 0.8)
        assert bs.order == order
        ys = np.array([L.biquad_apply_sample(C.byref(bs), float(v)) for v in x1[:600]], np.float32)
        want = z[f"order{order}/sample"]
        n = min(len(want), 600)
        assert np.array_equal(ys[:n].view(np.uint32), want[:n].view(np.uint32))
