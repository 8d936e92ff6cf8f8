"""CPU: the oracle (oracle/art_oracle.c) against the golden vectors generated from the real reference.
This is what pins the oracle on machines where /root/reference does not exist."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import _golden as G
from _artest import run_artest, PRESETS
from _oracle import (OracleResampler, load_oracle, checksum_words, checksum_bytes, noise, BiquadCoeffs, Biquad, f32p, u8p,
                     BH, INTERP, LOWPASS, PRECISE, DITHER_HP, SHAPE_ATH)


@pytest.mark.parametrize("name", G.NAMES)
def test_bank_matches_reference(name):
    z = G.load("bank")
    r = G.make(OracleResampler, name)
    bank = r.bank()
    assert tuple(z[name + "/shape"]) == bank.shape
    assert hashlib.sha256(bank.tobytes()).digest() == bytes(z[name + "/sha256"])
    assert np.array_equal(bank[z[name + "/rows"]].view(np.uint32), z[name + "/data"].view(np.uint32))


@pytest.mark.parametrize("name", G.NAMES)
@pytest.mark.parametrize("tag,flag", [("strict", 0), ("precise", PRECISE)])
def test_resample_bit_exact(name, tag, flag):
    y, trace = G.replay(G.make(OracleResampler, name, flag), name)
    full, head, tail, csum = G.expected(name, tag)
    ref_trace = G.load("resample")[name + "/trace"]
    assert np.array_equal(trace[:, :4], ref_trace[:, :4])            # used, generated, outputOffset bits, inputIndex
    assert checksum_words(y) == csum
    assert np.array_equal(y[:256].view(np.uint32), head.view(np.uint32))
    assert np.array_equal(y[-256:].view(np.uint32), tail.view(np.uint32))
    if full is not None:
        assert np.array_equal(y.view(np.uint32), full.view(np.uint32))


def test_make_build_within_one_ulp_fullscale_of_precise():
    """The contract the GPU float path is held to holds for the reference's own shipping build."""
    z = G.load("resample")
    for name in G.NAMES:
        if name + "/y_make" not in z.files:
            continue
        ym, yp = z[name + "/y_make"].astype(np.float64), z[name + "/y_precise"].astype(np.float64)
        tol = 2.0 ** -23 * np.maximum(1.0, np.abs(yp))
        assert np.all(np.abs(ym - yp) <= tol), name


ARTEST_CASES = [
    ("-1 -c1 -n2 -s44100 -d48000", 1, 1, 44100, 48000, dict(), None),
    ("-3 -c2 -n2 -s44100 -d48000", 3, 2, 44100, 48000, dict(), None),
    ("-3 -c2 -n2 -z -s44100 -d48000", 3, 2, 44100, 48000, dict(hann=True), None),
    ("-3 -c2 -n2 -p -s44100 -d48000", 3, 2, 44100, 48000, dict(precise=True), None),
    ("-3 -e -c2 -n2 -s44100 -d48000", 3, 2, 44100, 48000, dict(exact=True), None),
    ("-4 -c2 -n2 -s44100 -d48000", 4, 2, 44100, 48000, dict(), None),
    ("-2 -c2 -n2 -s44100 -d48000", 2, 2, 44100, 48000, dict(), None),
    ("-3 -c2 -n2 -s44100 -d48000 -b1000", 3, 2, 44100, 48000, dict(block=1000), None),
    ("-3 -e -c2 -n2 -s48000 -d44100 -l", 3, 2, 48000, 44100, dict(exact=True, lowpass=True), None),
    ("-4 -e -l -c8 -n1 -s96000 -d44100", 4, 8, 96000, 44100, dict(exact=True, lowpass=True, seconds=1), None),
    ("-4 -e -l -c8 -n3 -s96000 -d44100", 4, 8, 96000, 44100, dict(exact=True, lowpass=True, seconds=3), None),
    # reference reads out of bounds in this run's flush (DESIGN.md "reference bugs"): counts only
    ("-4 -e -l -c8 -n2 -s96000 -d44100", 4, 8, 96000, 44100, dict(exact=True, lowpass=True, ub_tail=True), None),
    ("-4 -c8 -n2 -o16 -s44100 -d48000", 4, 8, 44100, 48000, dict(), 16),
]


def artest_backend(cls, preset, chans, src, dst, exact=False, lowpass=False, hann=False, precise=False, **kw):
    taps, filters = PRESETS[preset]
    flags = INTERP | (0 if hann else BH) | (LOWPASS if lowpass else 0) | (PRECISE if precise else 0)
    if exact:
        r = cls(chans, taps, filters, flags=flags, fixed=(float(src), float(dst), 0), **kw)
    else:
        r = cls(chans, taps, filters, 0.0, flags, **kw)
    r.advance(taps / 2.0)
    return r


@pytest.mark.parametrize("args,preset,chans,src,dst,opt,outbits", ARTEST_CASES)
def test_oracle_reproduces_reference_artest_checksums(args, preset, chans, src, dst, opt, outbits):
    """Whole-program known answers: the reference's own test program (strict build) vs the oracle
    driven through the restated artest loop — noise, fades, 4096-frame blocks, flush, decimation."""
    want = G.kat()["strict"][args]
    opt = dict(opt)
    block = opt.pop("block", 4096)
    seconds = opt.pop("seconds", 2)
    ub_tail = opt.pop("ub_tail", False)
    dec = None
    if outbits:
        L = load_oracle()
        d = L.ora_decimate_init(chans, outbits, 2, 1.0, dst, DITHER_HP | SHAPE_ATH)     # artest.c:119,440

        def dec(y):
            out = np.zeros(y.size * 2, np.uint8)
            clips = L.ora_decimate_interleaved(d, y.ctypes.data_as(f32p), y.shape[0], out.ctypes.data_as(u8p))
            return out, clips
    res = run_artest(lambda: artest_backend(OracleResampler, preset, chans, src, dst, **opt), chans, PRESETS[preset][0],
                     src, dst, seconds, block=block, ratio_arg=0.0 if opt.get("exact") else None, decimator=dec)
    assert res["in_checksum"] == want["input"]["checksum"]
    assert res["out_frames"] == want["output"]["count"]
    if ub_tail:
        return
    assert res["out_checksum"] == want["output"]["checksum"]
    if outbits:
        assert res["dec_checksum"] == want["decimate"]["checksum"]
        assert res["clips"] == want["decimate"]["clips"]


def test_biquad_design_and_cascade():
    L = load_oracle()
    z = G.load("biquad")
    for key in [k for k in z.files if k.startswith("design/")]:
        row = z[key]
        c = BiquadCoeffs()
        (L.ora_biquad_lowpass if "/lp" in key else L.ora_biquad_highpass)(C.byref(c), float(row[0]))
        got = np.array([getattr(c, n) for n, _ in BiquadCoeffs._fields_], np.float32)
        assert np.array_equal(got.view(np.uint32), row[1:].astype(np.float32).view(np.uint32)), key
    ch, frames = 8, 3000
    x, _ = noise(frames * ch)
    buf = x.reshape(frames, ch).copy()
    c = BiquadCoeffs()
    L.ora_biquad_lowpass(C.byref(c), 44100 * 0.45 / 96000)
    filt = [[Biquad(), Biquad()] for _ in range(ch)]
    for pair in filt:
        for b in pair:
            L.ora_biquad_init(C.byref(b), C.byref(c), 1.0)
    for blk in range(3):
        view = buf[blk * 1000:(blk + 1) * 1000]
        for k in range(ch):
            for b in filt[k]:
                L.ora_biquad_buffer(C.byref(b), C.cast(view.ctypes.data + 4 * k, f32p), 1000, ch)
    assert np.array_equal(buf.view(np.uint32), z["cascade/y"].view(np.uint32))
    for order in (1, 2, 3, 4):
        co = z[f"order{order}/coeffs"]
        c = BiquadCoeffs(*[float(v) for v in co])
        x1, _ = noise(600)
        bb, bs = Biquad(), Biquad()
        L.ora_biquad_init(C.byref(bb), C.byref(c), 0.8)
        L.ora_biquad_init(C.byref(bs), C.byref(c), 0.8)
        assert bb.order == order
        yb = x1.copy()
        L.ora_biquad_buffer(C.byref(bb), yb.ctypes.data_as(f32p), 600, 1)
        ys = np.array([L.ora_biquad_sample(C.byref(bs), float(v)) for v in x1], np.float32)
        assert np.array_equal(yb.view(np.uint32), z[f"order{order}/buffer"].view(np.uint32))
        assert np.array_equal(ys.view(np.uint32), z[f"order{order}/sample"].view(np.uint32))


def decimate_input():
    ch, frames = 2, 6000
    x, _ = noise(frames * ch)
    x = (x * 1.9).astype(np.float32)
    x[100:110] = 1.5
    x[200:210] = -1.5
    return ch, frames, x


def test_decimator_all_combos_bit_exact():
    L = load_oracle()
    z = G.load("decimate")
    ch, frames, x = decimate_input()
    for (bits, nbytes, dither, shape, rate, want_sum, want_clips) in z["table"]:
        bits, nbytes, dither, shape, rate = int(bits), int(nbytes), int(dither), int(shape), int(rate)
        d = L.ora_decimate_init(ch, bits, nbytes, 1.0, rate, dither | shape)
        buf = np.zeros(frames * ch * nbytes, np.uint8)
        clips = 0
        for blk in range(3):
            seg = x[blk * 2000 * ch:(blk + 1) * 2000 * ch]
            clips += L.ora_decimate_interleaved(d, seg.ctypes.data_as(f32p), 2000, C.cast(buf.ctypes.data + blk * 2000 * ch * nbytes, u8p))
        L.ora_decimate_free(d)
        assert checksum_bytes(buf) == int(want_sum), (bits, nbytes, dither, shape, rate)
        assert clips == int(want_clips)
        key = f"bytes/{bits}_{nbytes}_{dither}_{shape}_{rate}"
        if key in z.files:
            assert np.array_equal(buf, z[key])


def test_decimator_planar_and_ingest():
    L = load_oracle()
    z = G.load("decimate")
    ch, frames, x = decimate_input()
    d = L.ora_decimate_init(ch, 16, 2, 1.0, 48000, DITHER_HP | SHAPE_ATH)
    planes = [np.ascontiguousarray(x.reshape(frames, ch)[:, k]) for k in range(ch)]
    outs = [np.zeros(frames * 2, np.uint8) for _ in range(ch)]
    ip = (f32p * ch)(*[p.ctypes.data_as(f32p) for p in planes])
    op = (u8p * ch)(*[o.ctypes.data_as(u8p) for o in outs])
    clips = L.ora_decimate_planar(d, ip, frames, op)
    L.ora_decimate_free(d)
    assert clips == int(z["planar/clips"])
    assert np.array_equal(np.stack(outs), z["planar/bytes"])
    raw = z["ingest/raw"]
    for bits, nbytes in ((8, 1), (16, 2), (24, 3), (24, 4), (12, 2), (20, 3)):
        o = np.zeros(50, np.float32)
        L.ora_float_integers_le(raw.ctypes.data_as(u8p), 0.75, bits, nbytes, 2, o.ctypes.data_as(f32p), 50)
        assert np.array_equal(o.view(np.uint32), z[f"ingest/{bits}_{nbytes}"].view(np.uint32))
