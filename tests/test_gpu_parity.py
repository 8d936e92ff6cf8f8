"""GPU (-m gpu): the HIP path, called through the C ABI, against the golden vectors made from the real
reference and against the oracle on the same seeded inputs.

Contract (DESIGN.md "parity"):
  * frame counts and the carried position after every call: exact, always;
  * RESAMPLE_STRICT_ORDER: every output bit equals the reference built with -O2 -ffp-contract=off;
  * EXTEND_CONVOLUTION_MATH: equals the reference's double-accumulate mode except where the fp64
    summation order flips the final float rounding (<= 1 float ulp, < 0.1 % of samples);
  * default (fast) mode: |y - y_precise| <= 2^-23 * max(1, |y|)  (one float32 ulp at full scale),
    and RMS error no worse than the reference's own shipping build;
  * biquad, decimator, ingest: bit-exact.
"""
import ctypes as C

import numpy as np
import pytest

import _golden as G
import audio_resampler_amd as A
from _artest import run_artest, PRESETS
from _hip import HipResampler, tolerance_ok
from _oracle import (OracleResampler, load_oracle, noise, checksum_words, checksum_bytes, f32p, u8p,
                     BH, INTERP, LOWPASS, PRECISE, DITHER_HP, SHAPE_ATH)
from test_oracle_golden import artest_backend, ARTEST_CASES, decimate_input

pytestmark = pytest.mark.gpu

STRICT = A.RESAMPLE_STRICT_ORDER


def test_native_library_is_the_in_tree_build():
    import os
    assert os.path.samefile(os.path.dirname(A.api.LIB_PATH), os.path.dirname(A.__file__))
    assert A.lib().artamdDeviceCount() >= 1


@pytest.mark.parametrize("name", G.NAMES)
def test_strict_mode_is_bit_exact_vs_reference(name):
    y, trace = G.replay(G.make(HipResampler, name, STRICT), name)
    full, head, tail, csum = G.expected(name, "strict")
    assert np.array_equal(trace[:, :4], G.load("resample")[name + "/trace"][:, :4])
    assert np.array_equal(y[:256].view(np.uint32), head.view(np.uint32))
    assert np.array_equal(y[-256:].view(np.uint32), tail.view(np.uint32))
    assert checksum_words(y) == csum
    if full is not None:
        assert np.array_equal(y.view(np.uint32), full.view(np.uint32))


@pytest.mark.parametrize("name", ["P_mono_48x48", "B_fixed_160x380", "C_small_147x156_lp", "lp_frac", "E_asrc_380_nolerp"])
def test_strict_precise_mode_is_bit_exact_vs_reference(name):
    y, _ = G.replay(G.make(HipResampler, name, STRICT | PRECISE), name)
    assert checksum_words(y) == G.expected(name, "precise")[3]


@pytest.mark.parametrize("name", G.NAMES)
def test_fast_mode_within_one_ulp_fullscale_of_precise(name):
    y, trace = G.replay(G.make(HipResampler, name), name)
    truth, _ = G.replay(G.make(OracleResampler, name, PRECISE), name)
    assert np.array_equal(trace[:, :4], G.load("resample")[name + "/trace"][:, :4])
    assert y.shape == truth.shape
    ok, worst, rms = tolerance_ok(y, truth)
    assert ok, (worst, rms)
    # no worse than the reference's own float build (same inputs, strict source order)
    ref_float, _ = G.replay(G.make(OracleResampler, name), name)
    _, _, rms_ref = tolerance_ok(ref_float, truth)
    assert rms <= rms_ref * 1.25 + 1e-12, (rms, rms_ref)


@pytest.mark.parametrize("kernel", [2, 7], ids=["matrix", "matrix_fixed_point"])
@pytest.mark.parametrize("name", G.NAMES)
def test_mfma_kernel_within_one_ulp_fullscale_of_precise(name, kernel):
    """same contract with the MFMA periodic-phase kernel forced on wherever the ratio is rational
    (short calls included; non-rational calls fall back to the general kernel inside the library) — and with its
    fixed-point form forced on wherever that can run (kernel preference 7)"""
    r = G.make(HipResampler, name, kernel=kernel)
    y, trace = G.replay(r, name)
    truth, _ = G.replay(G.make(OracleResampler, name, PRECISE), name)
    assert np.array_equal(trace[:, :4], G.load("resample")[name + "/trace"][:, :4])
    ok, worst, rms = tolerance_ok(y, truth)
    assert ok, (worst, rms)
    ref_float, _ = G.replay(G.make(OracleResampler, name), name)
    # (with preference 7 regular launches run in fixed point: effective rows on a 2^-31 grid, i.e. a floor of ~2^-31 x the
    # signal's rms x sqrt (taps) — it only shows where the float arithmetic of a tiny filter happens to be exact, e.g. 4 taps)
    floor = 2.0 ** -30 * float(np.sqrt(np.mean(truth.astype(np.float64) ** 2))) if kernel == 7 else 0.0
    assert rms <= tolerance_ok(ref_float, truth)[2] * 1.25 + 1e-12 + floor


def test_mfma_kernel_is_the_one_running_the_headline_config():
    ch, T, blk = 8, 988, 60000          # 65k output frames per call: beyond the measured general/MFMA crossover (~47k)
    x, _ = noise(ch * 2 * blk)
    x = x.reshape(-1, ch)
    r = HipResampler(ch, T, T, 0.0, BH | INTERP)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE)
    for b in (r, o):
        b.advance(T / 2)
    for k in range(2):
        u, g, y = r.process(x[k * blk:(k + 1) * blk], 66000, 48000 / 44100)
        uo, go, yo = o.process(x[k * blk:(k + 1) * blk], 66000, 48000 / 44100, threads=8)
        assert (u, g) == (uo, go)
        assert r.last_kernel() == 2
        ok, worst, rms = tolerance_ok(y, yo)
        assert ok and rms < 2.0e-8, (worst, rms)


@pytest.mark.parametrize("name", G.NAMES)
def test_precise_mode_matches_double_accumulate_reference(name):
    y, _ = G.replay(G.make(HipResampler, name, PRECISE), name)
    truth, _ = G.replay(G.make(OracleResampler, name, PRECISE), name)
    diff = y.view(np.int32).astype(np.int64) - truth.view(np.int32).astype(np.int64)
    same_sign = np.signbit(y) == np.signbit(truth)
    assert np.all(np.abs(diff[same_sign]) <= 1)
    assert np.all(np.abs(y[~same_sign] - truth[~same_sign]) < 1e-30)
    assert np.mean(diff != 0) < 1e-3


@pytest.mark.parametrize("args,preset,chans,src,dst,opt,outbits", ARTEST_CASES)
def test_reference_artest_known_answers_in_strict_mode(args, preset, chans, src, dst, opt, outbits):
    """The reference's whole test program, restated, driving the HIP library: same checksums as the
    reference's own artest (strict build) — resampler, flush and (with -o) the decimator."""
    want = G.kat()["strict"][args]
    opt = dict(opt)
    block = opt.pop("block", 4096)
    seconds = opt.pop("seconds", 2)
    ub_tail = opt.pop("ub_tail", False)
    dec = None
    if outbits:
        d = A.Decimator(chans, outbits, 2, 1.0, dst, DITHER_HP | SHAPE_ATH)
        dec = d.process
    res = run_artest(lambda: artest_backend(HipResampler, preset, chans, src, dst, extra=STRICT, **opt), chans,
                     PRESETS[preset][0], src, dst, seconds, block=block, ratio_arg=0.0 if opt.get("exact") else None, decimator=dec)
    assert res["out_frames"] == want["output"]["count"]
    if ub_tail:
        return
    assert res["out_checksum"] == want["output"]["checksum"]
    if outbits:
        assert res["dec_checksum"] == want["decimate"]["checksum"]
        assert res["clips"] == want["decimate"]["clips"]


def test_headline_config_fast_mode_full_artest_run_vs_oracle():
    """8 ch, 44.1k->48k, preset -4 (988x988 interpolating), 1 s, 4096-frame blocks + flush: every sample
    within tolerance of the double-accumulate oracle; frame count equals the reference's."""
    mk = lambda cls, **kw: run_artest(lambda: artest_backend(cls, 4, 8, 44100, 48000, **kw), 8, 988, 44100, 48000, 1, collect=True)
    got = mk(HipResampler)
    truth = mk(OracleResampler, precise=True)
    assert got["out_frames"] == truth["out_frames"]
    ok, worst, rms = tolerance_ok(got["y"], truth["y"])
    assert ok, (worst, rms)
    assert rms < 2.5e-8


def test_planar_and_device_entry_points_equal_interleaved():
    torch = pytest.importorskip("torch")
    ch, T = 3, 64
    x, _ = noise(ch * 5000)
    x = x.reshape(-1, ch)
    ratio = 48000 / 44100
    # one kernel for all four so that the comparison is about the entry-point plumbing only
    a, b, c, d = (HipResampler(ch, T, 32, 0.7, BH | INTERP, kernel=1) for _ in range(4))
    for r in (a, b, c, d):
        r.advance(T / 2)
    outs = []
    for k in range(4):
        seg = x[k * 1200:(k + 1) * 1200]
        ua, ga, ya = a.process(seg, 2000, ratio, and_flush=(k == 3))
        ub, gb, yb = b.process_planar([seg[:, i] for i in range(ch)], 2000, ratio, and_flush=(k == 3))
        assert (ua, ga) == (ub, gb)
        assert np.array_equal(ya.view(np.uint32), np.stack(yb, axis=1).view(np.uint32))
        din = torch.from_numpy(seg.copy()).cuda()
        dout = torch.zeros(2000, ch, device="cuda")
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        uc, gc = c.process_device(din, 1200, dout, 2000, ratio, and_flush=(k == 3))
        torch.cuda.synchronize()
        assert (uc, gc) == (ua, ga)
        assert np.array_equal(dout[:gc].cpu().numpy().view(np.uint32), ya.view(np.uint32))
        if k < 3:
            dpl = torch.from_numpy(np.ascontiguousarray(seg.T)).cuda()
            dpo = torch.zeros(ch, 2048, device="cuda")
            ud, gd = d.process_planar_device(dpl, 1200, 1200, dpo, 2048, 2000, ratio)
            torch.cuda.synchronize()
            assert (ud, gd) == (ua, ga)
            assert np.array_equal(dpo[:, :gd].cpu().numpy().T.view(np.uint32), ya.view(np.uint32))
        outs.append(ya)


def test_fixed_ratio_output_is_block_size_invariant():
    """size-independent property: with a reduced fixed-ratio bank the output stream does not
    depend on how the input is cut into calls."""
    ch, T = 2, 380
    x, _ = noise(ch * 40000)
    x = x.reshape(-1, ch)
    streams = []
    for block in (256, 1000, 4096, 40000):
        r = HipResampler(ch, T, 380, flags=BH | INTERP | LOWPASS, fixed=(44100.0, 48000.0, 0), extra=STRICT)
        r.advance(T / 2)
        ys = []
        for p in range(0, 40000, block):
            u, g, y = r.process(x[p:p + block], int(block * 1.1) + T, 0.0)
            assert u == min(block, 40000 - p)
            ys.append(y)
        streams.append(np.concatenate(ys))
    n = min(len(s) for s in streams)
    for s in streams[1:]:
        assert np.array_equal(s[:n].view(np.uint32), streams[0][:n].view(np.uint32))


def test_default_mode_cuts_differ_by_no_more_than_the_bar():
    """include/resampler.h, the deviation stated there: in the DEFAULT mode a call's size picks the kernel (general / f32 matrix cores /
    fixed point), so a fixed-ratio stream cut into other blocks is not the same bits as the reference's is (resampler.c:533-535) — but
    every cut is within the parity bar of the double-accumulate result, two cuts within twice the bar of each other, and counts and
    positions do not depend on the cut.  8 channels x 988 taps so that every kind of kernel takes part."""
    ch, T, total = 8, 988, 300000
    x, _ = noise(ch * total)
    x = x.reshape(-1, ch)
    o = OracleResampler(ch, T, 160, flags=BH | INTERP | LOWPASS | PRECISE, fixed=(44100.0, 48000.0, 0)); o.advance(T / 2)
    uo, go, truth = o.process(x, int(total * 1.1) + T, 0.0, threads=8)
    truth = np.array(truth)
    streams, kinds = [], set()
    for block in (900, 7000, 50000, 300000):
        r = HipResampler(ch, T, 160, flags=BH | INTERP | LOWPASS, fixed=(44100.0, 48000.0, 0))
        r.advance(T / 2)
        ys, used = [], 0
        for p in range(0, total, block):
            u, g, y = r.process(x[p:p + block], int(block * 1.1) + T, 0.0)
            assert u == min(block, total - p)
            used += u
            ys.append(np.array(y))
            kinds.add((r.last_kernel(), r.fixed_point()[0]))
        streams.append(np.concatenate(ys))
        assert used == uo
    assert {(1, 0), (2, 0), (2, 1)} <= kinds, kinds              # general, f32 matrix cores and fixed point all ran
    n = min(min(len(s) for s in streams), len(truth))
    assert all(len(s) == len(streams[0]) for s in streams)      # the same number of frames whatever the cut
    for s in streams:
        assert tolerance_ok(s[:n], truth[:n])[0]
    for s in streams[1:]:
        d = np.abs(s[:n].astype(np.float64) - streams[0][:n].astype(np.float64))
        assert np.all(d <= 2.0 * 2.0 ** -23 * np.maximum(1.0, np.abs(truth[:n].astype(np.float64))))


def test_reset_and_getters():
    L = A.lib()
    r = HipResampler(2, 156, 320, flags=BH | INTERP | LOWPASS, fixed=(96000.0, 44100.0, 0))
    o = OracleResampler(2, 156, 320, flags=BH | INTERP | LOWPASS, fixed=(96000.0, 44100.0, 0))
    assert L.resampleGetNumFilters(r.p) == o.c.filters == 147
    assert L.resampleInterpolationUsed(r.p) == 0
    assert L.resampleGetLowpassRatio(r.p) == o.c.lowpass_ratio
    x, _ = noise(2 * 3000)
    x = x.reshape(-1, 2)
    Lo = load_oracle()
    for _ in range(2):
        r.advance(78.0)
        o.advance(78.0)
        for n in (10, 333):
            assert L.resampleGetRequiredSamples(r.p, n, 0.0) == Lo.ora_resample_required_input(o.p, n, 0.0)
            assert L.resampleGetExpectedOutput(r.p, n, 0.0) == Lo.ora_resample_expected_output(o.p, n, 0.0)
        u, g, y = r.process(x, 3000, 0.0)
        uo, go, yo = o.process(x, 3000, 0.0)
        assert (u, g) == (uo, go) and r.position() == o.position()
        assert tolerance_ok(y, yo)[0]
        r.reset()
        o.reset()
        assert r.state()[:2] == o.state()[:2]


# ------------------------------------------------------------------------------------------------
# biquad / decimator / ingest: bit-exact
# ------------------------------------------------------------------------------------------------

def test_biquad_host_api_and_device_bank_bit_exact():
    torch = pytest.importorskip("torch")
    L = A.lib()
    z = G.load("biquad")
    for key in [k for k in z.files if k.startswith("design/")]:
        row = z[key]
        c = A.BiquadCoefficients()
        (L.biquad_lowpass if "/lp" in key else L.biquad_highpass)(C.byref(c), float(row[0]))
        got = np.array([getattr(c, n) for n, _ in A.BiquadCoefficients._fields_], np.float32)
        assert np.array_equal(got.view(np.uint32), row[1:].astype(np.float32).view(np.uint32)), key
    ch, frames = 8, 3000
    x, _ = noise(frames * ch)
    c = A.BiquadCoefficients()
    L.biquad_lowpass(C.byref(c), 44100 * 0.45 / 96000)
    # (a) the reference's calling pattern: one strided call per channel per section (art.c:1011-1017)
    buf = x.reshape(frames, ch).copy()
    filt = [[A.Biquad(), A.Biquad()] for _ in range(ch)]
    for pair in filt:
        for b in pair:
            L.biquad_init(C.byref(b), C.byref(c), 1.0)
    for blk in range(3):
        view = buf[blk * 1000:(blk + 1) * 1000]
        for k in range(ch):
            for b in filt[k]:
                L.biquad_apply_buffer(C.byref(b), C.cast(view.ctypes.data + 4 * k, f32p), 1000, ch)
    assert np.array_equal(buf.view(np.uint32), z["cascade/y"].view(np.uint32))
    # (b) device-resident bank: all channels x both sections in one launch per block
    secs = (A.Biquad * (ch * 2))()
    for i in range(ch * 2):
        L.biquad_init(C.byref(secs[i]), C.byref(c), 1.0)
    bank = A.BiquadBank(secs, ch, 2)
    d = torch.from_numpy(x.reshape(frames, ch).copy()).cuda()
    for blk in range(3):
        bank.apply_device(d[blk * 1000:(blk + 1) * 1000], 1000)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint32), z["cascade/y"].view(np.uint32))
    state = bank.read()
    for k in range(ch):
        for s in range(2):
            assert bytes(state[k * 2 + s]) == bytes(filt[k][s])
    # (c) orders 1..4, buffer and per-sample association
    for order in (1, 2, 3, 4):
        co = A.BiquadCoefficients(*[float(v) for v in z[f"order{order}/coeffs"]])
        x1, _ = noise(600)
        bb, bs = A.Biquad(), A.Biquad()
        L.biquad_init(C.byref(bb), C.byref(co), 0.8)
        L.biquad_init(C.byref(bs), C.byref(co), 0.8)
        assert bb.order == order
        yb = x1.copy()
        L.biquad_apply_buffer(C.byref(bb), yb.ctypes.data_as(f32p), 600, 1)
        assert np.array_equal(yb.view(np.uint32), z[f"order{order}/buffer"].view(np.uint32))
        ys = np.array([L.biquad_apply_sample(C.byref(bs), float(v)) for v in x1[:64]], np.float32)
        assert np.array_equal(ys.view(np.uint32), z[f"order{order}/sample"][:64].view(np.uint32))


def test_decimator_all_combos_bit_exact():
    z = G.load("decimate")
    ch, frames, x = decimate_input()
    x2 = x.reshape(frames, ch)
    for (bits, nbytes, dither, shape, rate, want_sum, want_clips) in z["table"]:
        bits, nbytes, dither, shape, rate = int(bits), int(nbytes), int(dither), int(shape), int(rate)
        d = A.Decimator(ch, bits, nbytes, 1.0, rate, dither | shape)
        parts, clips = [], 0
        for blk in range(3):
            b, c = d.process(x2[blk * 2000:(blk + 1) * 2000])
            parts.append(b)
            clips += c
        buf = np.concatenate(parts)
        assert checksum_bytes(buf) == int(want_sum), (bits, nbytes, dither, shape, rate)
        assert clips == int(want_clips)
        key = f"bytes/{bits}_{nbytes}_{dither}_{shape}_{rate}"
        if key in z.files:
            assert np.array_equal(buf, z[key])
        d.close()


def test_decimator_planar_device_and_ingest():
    torch = pytest.importorskip("torch")
    L = A.lib()
    z = G.load("decimate")
    ch, frames, x = decimate_input()
    x2 = x.reshape(frames, ch)
    d = A.Decimator(ch, 16, 2, 1.0, 48000, DITHER_HP | SHAPE_ATH)
    outs, clips = d.process_planar([np.ascontiguousarray(x2[:, k]) for k in range(ch)])
    assert clips == int(z["planar/clips"])
    assert np.array_equal(np.stack(outs), z["planar/bytes"])
    # device-pointer form == host form
    d1 = A.Decimator(ch, 16, 2, 1.0, 48000, DITHER_HP | SHAPE_ATH)
    d2 = A.Decimator(ch, 16, 2, 1.0, 48000, DITHER_HP | SHAPE_ATH)
    want, wc = d1.process(x2)
    din = torch.from_numpy(x2.copy()).cuda()
    dout = torch.zeros(frames * ch * 2, dtype=torch.uint8, device="cuda")
    d2.process_device(din, frames, dout)
    assert d2.clipped() == wc
    assert np.array_equal(dout.cpu().numpy(), want)
    raw = z["ingest/raw"].copy()
    for bits, nbytes in ((8, 1), (16, 2), (24, 3), (24, 4), (12, 2), (20, 3)):
        o = np.zeros(50, np.float32)
        L.floatIntegersLE(raw.ctypes.data_as(u8p), 0.75, bits, nbytes, 2, o.ctypes.data_as(f32p), 50)
        assert np.array_equal(o.view(np.uint32), z[f"ingest/{bits}_{nbytes}"].view(np.uint32))


@pytest.mark.parametrize("ch,nsec", [(1, 1), (1, 2), (2, 2), (3, 1), (5, 2), (8, 1), (43, 2), (64, 2), (70, 2)])
def test_biquad_bank_pipeline_geometries_bit_exact(ch, nsec):
    """the feed-forward / four-stage biquad kernel over channel counts (lane mappings, several workgroups) and call
    lengths (one chunk, remainders of 1..3 frames, many chunks), state carried across calls — against the oracle"""
    torch = pytest.importorskip("torch")
    from _oracle import load_oracle, Biquad as OBiquad, BiquadCoeffs as OCoeffs
    L, OL = A.lib(), load_oracle()
    lengths = [64, 65, 67, 352, 353, 1000, 4099, 20000, 3]       # the last call (3 frames) takes the generic kernel
    total = sum(lengths)
    x, _ = noise(total * ch, state=0xC0FFEE1234567 | 1)
    x = x.reshape(total, ch)
    co, oc = A.BiquadCoefficients(), OCoeffs()
    secs = (A.Biquad * (ch * nsec))()
    osecs = [[OBiquad() for _ in range(nsec)] for _ in range(ch)]
    for k in range(ch):
        for s in range(nsec):
            f = 0.05 + 0.4 * ((k * 7 + s * 3) % 11) / 11.0
            (L.biquad_lowpass if (k + s) % 3 else L.biquad_highpass)(C.byref(co), f)
            (OL.ora_biquad_lowpass if (k + s) % 3 else OL.ora_biquad_highpass)(C.byref(oc), f)
            L.biquad_init(C.byref(secs[k * nsec + s]), C.byref(co), 0.9)
            OL.ora_biquad_init(C.byref(osecs[k][s]), C.byref(oc), 0.9)
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
    for k in range(ch):
        for s in range(nsec):
            a, b = state[k * nsec + s], osecs[k][s]
            hist = lambda q, arr: [arr[(q.index - i) & 3] for i in range(4)]
            assert hist(a, a.x) == hist(b, b.x) and hist(a, a.y) == hist(b, b.y), (k, s)


def test_extreme_downsampling_ratio_whose_span_does_not_fit_the_lds():
    """ratio 1/7000 with 988 taps x 8 channels: one output's window and the next are 7000 frames apart — more than a
    workgroup's LDS holds.  The reference accepts any positive ratio; the library must still consume the input and produce the
    outputs (a caller looping on input_used must not spin): evaluated by the direct, one-lane-per-sample kernel."""
    ch, T = 8, 988
    ratio = 1.0 / 7000.0
    x, _ = noise(ch * 40000)
    x = x.reshape(-1, ch)
    r = HipResampler(ch, T, T, 0.0, BH | INTERP)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE)
    s = HipResampler(ch, T, T, 0.0, BH | INTERP | STRICT)
    os_ = OracleResampler(ch, T, T, 0.0, BH | INTERP)
    for b in (r, o, s, os_):
        b.advance(T / 2)
    for blk in (x[:25000], x[25000:]):
        u, g, y = r.process(blk, 64, ratio)
        uo, go, yo = o.process(blk, 64, ratio)
        assert (u, g) == (uo, go) and u == len(blk) and (g > 0 or len(blk) < 7000)
        ok, worst, rms = tolerance_ok(y, yo)
        assert ok, (worst, rms)
        us, gs, ys = s.process(blk, 64, ratio)
        _, _, yos = os_.process(blk, 64, ratio)
        assert (us, gs) == (uo, go) and np.array_equal(ys.view(np.uint32), yos.view(np.uint32))
