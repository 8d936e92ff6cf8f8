"""The 8-byte sample path (reference PATH_WIDTH=64: art64 / artest64, reference Makefile:13/:19, resampler.h:22-26).

libartamd64.so is the product tree compiled with -DPATH_WIDTH=64; oracle/_build/liboracle64_*.so the restatement with
double samples; tests/golden/wide.npz + artest64_kat.json come from the real reference built with -DPATH_WIDTH=64
(tests/golden/make_golden64.py).  CPU tests pin the wide oracle and the library's host logic; GPU tests are the
parity tests: strict mode bit for bit, default mode within 2^-47 (relative to max(1,|y|)) of the reference-order
result — there is ONE arithmetic in this build (EXTEND_CONVOLUTION_MATH is a no-op, reference resampler.c:191), the
default mode only re-associates the double accumulation across a wave.
"""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

import _golden as G
import _oracle
import audio_resampler_amd as A
from test_gpu_fuzz import random_session, play

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = _oracle.wide()                       # oracle / reference bindings, double samples
W = A.wide()                             # product binding, double samples
f64p, u8p = W.f32p, W.u8p
STRICT = A.RESAMPLE_STRICT_ORDER
FAST_TOL = 2.0 ** -47      # two differently ordered double sums of ~1000 products; largest seen in 600 large random sessions: 34 * 2^-53

_z = {}


def gold():
    if "z" not in _z:
        _z["z"] = np.load(os.path.join(G.GOLD, "wide.npz"))
    return _z["z"]


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def replay(backend, name):
    script = G.script_of(name)
    ch = G.CTOR[name]["args"][0]
    total = sum(n for n, _, _, f in script if not f) + 16
    x, _ = O.noise(total * ch)
    x = x.reshape(-1, ch)
    outs, trace, pos = [], [], 0
    for (n, cap, ratio, flush) in script:
        if flush:
            u, g, o = backend.process(None, cap, ratio, flush=True)
        else:
            u, g, o = backend.process(x[pos:pos + n], cap, ratio)
            pos += u
        outs.append(np.array(o, copy=True))
        trace.append((u, g) + tuple(backend.state()))
    return np.concatenate(outs), np.array(trace, dtype=np.uint64)


def check_against_golden(name, y, tr, flag_mask=0xffffffff):
    z = gold()
    want = z[f"resample/{name}/trace"]
    assert np.array_equal(tr[:, :4], want[:, :4])
    assert np.array_equal(tr[:, 4] & flag_mask, want[:, 4] & flag_mask)
    assert y.dtype == np.float64 and y.shape[0] == int(z[f"resample/{name}/frames"])
    assert np.array_equal(bits(y[:256]), bits(z[f"resample/{name}/head"]))
    assert np.array_equal(bits(y[-256:]), bits(z[f"resample/{name}/tail"]))
    assert O.checksum_words(y) == int(z[f"resample/{name}/sum"])


class HipWide(W.Resampler):
    def __init__(self, channels, taps, filters, lowpass_ratio=0.0, flags=A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE,
                 fixed=None, extra=0, kernel=0):
        super().__init__(channels, taps, filters, lowpass_ratio, flags | extra, fixed)
        if kernel:
            self.set_kernel(kernel)          # 2 = force the fp64 matrix-core kernel wherever the ratio is rational


def decimate_input(ch=2, frames=6000):
    x, _ = O.noise(frames * ch)
    x = x * 1.9
    x[100:110] = 1.5
    x[200:210] = -1.5
    return ch, frames, x


# ------------------------------------------------------------------------------------------------
# CPU: the wide oracle against the reference's PATH_WIDTH=64 vectors
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", G.NAMES)
def test_wide_oracle_matches_reference64_vectors(name):
    y, tr = replay(G.make(O.OracleResampler, name), name)
    check_against_golden(name, y, tr)


@pytest.mark.parametrize("name", G.NAMES)
def test_wide_oracle_ignores_extended_math_flag(name):
    # reference resampler.c:191: the double-accumulate path is selected for 4-byte samples only
    y, tr = replay(G.make(O.OracleResampler, name, extra_flags=_oracle.PRECISE), name)
    check_against_golden(name, y, tr, flag_mask=~_oracle.PRECISE & 0xffffffff)


def test_wide_oracle_biquad_and_decimator_match_reference64_vectors():
    z, L = gold(), O.load_oracle()
    for key in [k for k in z.files if k.startswith("biquad/design/")]:
        row = z[key]
        c = O.BiquadCoeffs()
        (L.ora_biquad_lowpass if "/lp" in key else L.ora_biquad_highpass)(C.byref(c), float(row[0]))
        assert np.array_equal(bits([getattr(c, n) for n, _ in O.BiquadCoeffs._fields_]), bits(row[1:])), key
    for order in (1, 2, 3, 4):
        co = O.BiquadCoeffs(*[float(v) for v in z[f"biquad/order{order}/coeffs"]])
        x1, _ = O.noise(600)
        bb, bs = O.Biquad(), O.Biquad()
        L.ora_biquad_init(C.byref(bb), C.byref(co), 0.8)
        L.ora_biquad_init(C.byref(bs), C.byref(co), 0.8)
        yb = x1.copy()
        L.ora_biquad_buffer(C.byref(bb), yb.ctypes.data_as(f64p), 600, 1)
        assert np.array_equal(bits(yb), bits(z[f"biquad/order{order}/buffer"]))
        ys = np.array([L.ora_biquad_sample(C.byref(bs), float(v)) for v in x1])
        assert np.array_equal(bits(ys), bits(z[f"biquad/order{order}/sample"]))
    ch, frames, x = decimate_input()
    for (nbits, nbytes, dither, shape, rate, want_sum, want_clips) in z["decimate/table"]:
        nbits, nbytes = int(nbits), int(nbytes)
        d = L.ora_decimate_init(ch, nbits, nbytes, 1.0, int(rate), int(dither) | int(shape))
        buf = np.zeros(frames * ch * nbytes, np.uint8)
        clips = 0
        for blk in range(3):
            seg = x[blk * 2000 * ch:(blk + 1) * 2000 * ch]
            clips += L.ora_decimate_interleaved(d, seg.ctypes.data_as(f64p), 2000, C.cast(buf.ctypes.data + blk * 2000 * ch * nbytes, u8p))
        L.ora_decimate_free(d)
        assert O.checksum_bytes(buf) == int(want_sum) and clips == int(want_clips), (nbits, nbytes, int(dither), int(shape), int(rate))
    raw = z["ingest/raw"].copy()
    for nbits, nbytes in ((8, 1), (16, 2), (24, 3), (24, 4), (12, 2), (20, 3)):
        o = np.zeros(50, np.float64)
        L.ora_float_integers_le(raw.ctypes.data_as(u8p), 0.75, nbits, nbytes, 2, o.ctypes.data_as(f64p), 50)
        assert np.array_equal(bits(o), bits(z[f"ingest/{nbits}_{nbytes}"]))


@pytest.mark.ref
@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libartref64_strict.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(24))
def test_wide_oracle_equals_reference64_on_random_sessions(seed):
    s = random_session(seed)
    y, tr = play(O.OracleResampler, s, noise_fn=O.noise)
    yr, trr = play(O.RefResampler, s, noise_fn=O.noise)
    assert tr == trr and y.shape == yr.shape
    # A flush that arrives with the write index in the last half window of the ring makes the reference read before its
    # buffer (DESIGN.md "reference bugs"): whatever the heap holds there lands in the last T/2 * ratio frames it
    # generates.  Everything before that tail must agree bit for bit; the tail itself is not compared.
    ctor, _, calls, ch, _ = s
    T = ctor["args"][1]
    ratio = max(c[-1] for c in calls if c[0] in ("run", "flush"))
    total = y.shape[0]
    tail = min(total, int(T / 2 * ratio) + 2)
    assert np.array_equal(bits(y[:total - tail]), bits(yr[:total - tail]))


# ------------------------------------------------------------------------------------------------
# CPU: libartamd64.so host side
# ------------------------------------------------------------------------------------------------
def test_wide_library_exports_every_declared_symbol_and_has_the_wide_layouts():
    L = W.lib()
    for name in A.EXPORTED_SYMBOLS:
        assert hasattr(L, name), name
    assert set(W.EXPORTED_SYMBOLS) == set(A.EXPORTED_SYMBOLS)
    assert C.sizeof(W.Biquad) == 152 and C.sizeof(W.BiquadCoefficients) == 72
    assert b"64" in L.artamdVersion()
    # the public headers agree with the binding when compiled the way a PATH_WIDTH=64 client compiles them
    src = ('#include "resampler.h"\n#include "biquad.h"\n#include "decimator.h"\n#include "art_hip.h"\n'
           "_Static_assert (sizeof (artsample_t) == 8, \"sample\");\n_Static_assert (sizeof (Biquad) == 152, \"Biquad\");\n"
           "_Static_assert (sizeof (BiquadCoefficients) == 72, \"coeffs\");\nint main (void) { return 0; }\n")
    p = subprocess.run(["gcc", "-std=c99", "-DPATH_WIDTH=64", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", "-x", "c", "-"],
                       input=src, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


@pytest.mark.skipif(W.lib().artamdDeviceCount() > 0, reason="a GPU is present")
def test_wide_library_fails_loudly_without_a_gpu(capfd):
    L = W.lib()
    assert not L.resampleInit(2, 48, 48, 0.0, 3)
    assert not L.decimateInit(2, 16, 2, 1.0, 48000, 0)
    assert "no CPU path" in capfd.readouterr().err


@pytest.mark.parametrize("name", G.NAMES)
def test_wide_filter_bank_matches_reference64(name):
    z, rz = gold(), G.load("resample")
    F, T, flags = [int(v) for v in rz[name + "/meta"]]
    lowpass = float(rz[name + "/meta_f"][0])
    bank = np.zeros((F + 1, T), np.float64)
    W.lib().artamdBuildFilterBank(T, F, lowpass, flags, bank.ctypes.data_as(f64p))
    assert hashlib.sha256(bank.tobytes()).digest() == bytes(z[f"bank/{name}/sha256"])
    assert np.array_equal(bits(bank[z[f"bank/{name}/rows"]]), bits(z[f"bank/{name}/data"]))


# ------------------------------------------------------------------------------------------------
# GPU: parity
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", G.NAMES)
def test_wide_strict_mode_is_bit_exact_vs_reference64(name):
    y, tr = replay(G.make(HipWide, name, extra_flags=STRICT), name)
    check_against_golden(name, y, tr, flag_mask=0xffff)


def within_tolerance(y, truth):
    err = np.abs(y - truth)
    return bool(np.all(err <= FAST_TOL * np.maximum(1.0, np.abs(truth)))), float(err.max())


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 2])
@pytest.mark.parametrize("name", G.NAMES)
def test_wide_default_mode_within_tolerance_of_reference_order(name, kernel):
    y, tr = replay(G.make(HipWide, name, kernel=kernel), name)
    yo, tro = replay(G.make(O.OracleResampler, name), name)
    assert np.array_equal(tr[:, :4], tro[:, :4])
    ok, worst = within_tolerance(y, yo)
    assert ok, worst
    # and the extended-math flag changes nothing in this build
    y2, _ = replay(G.make(HipWide, name, extra_flags=A.EXTEND_CONVOLUTION_MATH, kernel=kernel), name)
    assert np.array_equal(bits(y), bits(y2))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_wide_random_session_strict_bit_exact(seed):
    s = random_session(seed)
    y, tr = play(HipWide, s, STRICT, noise_fn=O.noise)
    yo, tro = play(O.OracleResampler, s, noise_fn=O.noise)
    assert tr == tro
    assert y.shape == yo.shape and np.array_equal(bits(y), bits(yo))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 2])
@pytest.mark.parametrize("seed", range(40, 80))
def test_wide_random_session_default_mode_within_tolerance(seed, kernel):
    s = random_session(seed)
    y, tr = play(HipWide, s, noise_fn=O.noise, kernel=kernel)
    yo, tro = play(O.OracleResampler, s, noise_fn=O.noise)
    assert tr == tro
    ok, worst = within_tolerance(y, yo)
    assert ok, worst


@pytest.mark.gpu
def test_wide_planar_and_device_entry_points_equal_interleaved():
    torch = pytest.importorskip("torch")
    ch, T, n = 3, 156, 5000
    ratio = 48000 / 44100
    x, _ = O.noise(n * ch)
    x = x.reshape(n, ch)
    cap = int(n * ratio) + 64
    a = HipWide(ch, T, 320, extra=STRICT)
    a.advance(T / 2)
    _, ga, ya = a.process(x, cap, ratio)
    b = HipWide(ch, T, 320, extra=STRICT)
    b.advance(T / 2)
    _, gb, planes = b.process_planar([np.ascontiguousarray(x[:, k]) for k in range(ch)], cap, ratio)
    assert ga == gb and np.array_equal(bits(np.stack(planes, axis=1)), bits(ya))
    c = HipWide(ch, T, 320, extra=STRICT)
    c.advance(T / 2)
    d_in = torch.from_numpy(x.copy()).cuda()
    d_out = torch.zeros(cap, ch, dtype=torch.float64, device="cuda")
    _, gc = c.process_device(d_in, n, d_out, cap, ratio)
    c.synchronize()
    assert gc == ga and np.array_equal(bits(d_out[:gc].cpu().numpy()), bits(ya))
    assert a.state() == b.state() == c.state()


@pytest.mark.gpu
def test_wide_headline_shape_block_default_mode_vs_oracle():
    # 8 channels, 988 x 988 interpolating, 44.1k -> 48k: the headline configuration with double samples; this much work
    # takes the fp64 matrix-core kernel by itself
    torch = pytest.importorskip("torch")
    ch, T, n = 8, 988, 65536        # 71k output frames: beyond the general / fp64-MFMA crossover (~62k)
    ratio = 48000 / 44100
    x, _ = O.noise(n * ch)
    x = x.reshape(n, ch)
    cap = int(n * ratio) + 64
    h = HipWide(ch, T, 988)
    h.advance(T / 2)
    o = O.OracleResampler(ch, T, 988)
    o.advance(T / 2)
    _, g, y = h.process(x, cap, ratio)
    _, go, yo = o.process(x, cap, ratio, threads=8)
    assert g == go and h.last_kernel() == 2
    ok, worst = within_tolerance(y, yo)
    assert ok, worst


from test_gpu_fuzz import EDGE_CASES  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("case", EDGE_CASES, ids=lambda c: f"c{c[0]}_t{c[1]}_f{c[2]}_{c[3]}to{c[4]}{'_fixed' if c[5] else ''}")
@pytest.mark.parametrize("kernel", [0, 2])
def test_wide_edge_geometries(case, kernel):
    ch, T, F, src, dst, fixed, flags = case
    ratio = dst / src
    n1, n2 = 20 * T + 77, 9 * T + 5
    x, _ = O.noise((n1 + n2) * ch, state=0xDEADBEEFCAFEF00D | 1)
    x = x.reshape(-1, ch)
    kw = dict(flags=flags, fixed=(float(src), float(dst), 0)) if fixed else {}
    args = (ch, T, F) if fixed else (ch, T, F, 0.0, flags)
    h, o = HipWide(*args, **kw, kernel=kernel), O.OracleResampler(*args, **kw)
    for r in (h, o):
        r.advance(T / 2)
    cap = int(n1 * ratio) + 64
    for blk, n in ((x[:n1], n1), (x[n1:], n2)):
        uh, gh, yh = h.process(blk, cap, ratio)
        uo, go, yo = o.process(blk, cap, ratio, threads=8)
        assert (uh, gh) == (uo, go) and h.state()[:2] == o.state()[:2]
        ok, worst = within_tolerance(yh, yo)
        assert ok, worst
    _, gh, yh = h.process(None, cap, ratio, flush=True)
    _, go, yo = o.process(None, cap, ratio, flush=True)
    assert gh == go
    ok, worst = within_tolerance(yh, yo)
    assert ok, worst


@pytest.mark.gpu
def test_wide_biquad_host_api_and_device_bank_bit_exact():
    torch = pytest.importorskip("torch")
    L, z = W.lib(), gold()
    for key in [k for k in z.files if k.startswith("biquad/design/")]:
        row = z[key]
        c = W.BiquadCoefficients()
        (L.biquad_lowpass if "/lp" in key else L.biquad_highpass)(C.byref(c), float(row[0]))
        assert np.array_equal(bits([getattr(c, n) for n, _ in W.BiquadCoefficients._fields_]), bits(row[1:])), key
    ch, frames = 8, 3000
    x, _ = O.noise(frames * ch)
    c = W.BiquadCoefficients()
    L.biquad_lowpass(C.byref(c), 44100 * 0.45 / 96000)
    buf = x.reshape(frames, ch).copy()
    filt = [[W.Biquad(), W.Biquad()] for _ in range(ch)]
    for pair in filt:
        for b in pair:
            L.biquad_init(C.byref(b), C.byref(c), 1.0)
    for blk in range(3):
        view = buf[blk * 1000:(blk + 1) * 1000]
        for k in range(ch):
            for b in filt[k]:
                L.biquad_apply_buffer(C.byref(b), C.cast(view.ctypes.data + 8 * k, f64p), 1000, ch)
    assert np.array_equal(bits(buf), bits(z["biquad/cascade/y"]))
    secs = (W.Biquad * (ch * 2))()
    for i in range(ch * 2):
        L.biquad_init(C.byref(secs[i]), C.byref(c), 1.0)
    bank = W.BiquadBank(secs, ch, 2)
    d = torch.from_numpy(x.reshape(frames, ch).copy()).cuda()
    for blk in range(3):
        bank.apply_device(d[blk * 1000:(blk + 1) * 1000], 1000)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d.cpu().numpy()), bits(z["biquad/cascade/y"]))
    state = bank.read()
    for k in range(ch):
        for s in range(2):
            assert bytes(state[k * 2 + s]) == bytes(filt[k][s])
    for order in (1, 2, 3, 4):
        co = W.BiquadCoefficients(*[float(v) for v in z[f"biquad/order{order}/coeffs"]])
        x1, _ = O.noise(600)
        bb, bs = W.Biquad(), W.Biquad()
        L.biquad_init(C.byref(bb), C.byref(co), 0.8)
        L.biquad_init(C.byref(bs), C.byref(co), 0.8)
        assert bb.order == order
        yb = x1.copy()
        L.biquad_apply_buffer(C.byref(bb), yb.ctypes.data_as(f64p), 600, 1)
        assert np.array_equal(bits(yb), bits(z[f"biquad/order{order}/buffer"]))
        ys = np.array([L.biquad_apply_sample(C.byref(bs), float(v)) for v in x1[:64]])
        assert np.array_equal(bits(ys), bits(z[f"biquad/order{order}/sample"][:64]))


@pytest.mark.gpu
def test_wide_decimator_all_combos_planar_device_and_ingest_bit_exact():
    torch = pytest.importorskip("torch")
    L, z = W.lib(), gold()
    ch, frames, x = decimate_input()
    x2 = x.reshape(frames, ch)
    for (nbits, nbytes, dither, shape, rate, want_sum, want_clips) in z["decimate/table"]:
        nbits, nbytes, dither, shape, rate = int(nbits), int(nbytes), int(dither), int(shape), int(rate)
        d = W.Decimator(ch, nbits, nbytes, 1.0, rate, dither | shape)
        parts, clips = [], 0
        for blk in range(3):
            b, c = d.process(x2[blk * 2000:(blk + 1) * 2000])
            parts.append(b)
            clips += c
        buf = np.concatenate(parts)
        assert O.checksum_bytes(buf) == int(want_sum), (nbits, nbytes, dither, shape, rate)
        assert clips == int(want_clips)
        key = f"decimate/bytes/{nbits}_{nbytes}_{dither}_{shape}_{rate}"
        if key in z.files:
            assert np.array_equal(buf, z[key])
        d.close()
    flags = A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE
    d = W.Decimator(ch, 16, 2, 1.0, 48000, flags)
    outs, clips = d.process_planar([np.ascontiguousarray(x2[:, k]) for k in range(ch)])
    assert clips == int(z["decimate/planar/clips"]) and np.array_equal(np.stack(outs), z["decimate/planar/bytes"])
    d1, d2 = W.Decimator(ch, 16, 2, 1.0, 48000, flags), W.Decimator(ch, 16, 2, 1.0, 48000, flags)
    want, wc = d1.process(x2)
    din = torch.from_numpy(x2.copy()).cuda()
    dout = torch.zeros(frames * ch * 2, dtype=torch.uint8, device="cuda")
    d2.process_device(din, frames, dout)
    assert d2.clipped() == wc and np.array_equal(dout.cpu().numpy(), want)
    raw = z["ingest/raw"].copy()
    for nbits, nbytes in ((8, 1), (16, 2), (24, 3), (24, 4), (12, 2), (20, 3)):
        o = np.zeros(50, np.float64)
        L.floatIntegersLE(raw.ctypes.data_as(u8p), 0.75, nbits, nbytes, 2, o.ctypes.data_as(f64p), 50)
        assert np.array_equal(bits(o), bits(z[f"ingest/{nbits}_{nbytes}"]))


# ------------------------------------------------------------------------------------------------
# GPU: the reference's own 64-bit programs (artest64, art64) linked against libartamd64.so
# ------------------------------------------------------------------------------------------------
ARTEST64 = os.path.join(_oracle.ORACLE_DIR, "_ref", "artest64_amd")
ART64_AMD = os.path.join(_oracle.ORACLE_DIR, "_ref", "art64_amd")
ART64_REF = os.path.join(_oracle.ORACLE_DIR, "_ref", "art64_strict")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(ARTEST64), reason="oracle/_ref/artest64_amd not built (needs /root/reference at build time)")
@pytest.mark.parametrize("args", ["-1 -c1 -n2 -s44100 -d48000", "-3 -c2 -n2 -s44100 -d48000", "-3 -e -c2 -n2 -s44100 -d48000",
                                  "-4 -c8 -n2 -o16 -s44100 -d48000", "-4 -c8 -n2 -o24 -s44100 -d48000", "-4 -e -l -c8 -n1 -s96000 -d44100"])
def test_reference_artest64_binary_on_the_hip_library_matches_reference_checksums(args):
    with open(os.path.join(G.GOLD, "artest64_kat.json")) as f:
        want = json.load(f)["strict"][args]
    p = subprocess.run([ARTEST64] + args.split(), capture_output=True, text=True, env=dict(os.environ, ARTAMD_STRICT="1"), timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    got = {}
    for line in p.stderr.splitlines():
        m = re.search(r"(input|output|decimate) \(-w\d\): count =\s*(\d+), checksum = ([0-9a-f]{16})", line)
        if m:
            got[m.group(1)] = (int(m.group(2)), m.group(3))
            c = re.search(r"clipped samples = (\d+)", line)
            if c:
                got["clips"] = int(c.group(1))
    for stage in ("input", "output"):
        assert got[stage] == (want[stage]["count"], want[stage]["checksum"])
    if "decimate" in want:
        assert got["decimate"] == (want["decimate"]["count"], want["decimate"]["checksum"]) and got["clips"] == want["decimate"]["clips"]


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(ART64_AMD) and os.path.exists(ART64_REF)), reason="oracle/_ref/art64_* not built")
@pytest.mark.parametrize("opts,rate_in,chans", [("-4 -r48000", 44100, 2), ("-3 -r44100 -p", 96000, 2), ("-2 -r48000 -o24 -d1 -n2", 44100, 1),
                                                 ("-2 -r48000 --tempo=0.8 --pitch=200", 44100, 2)])     # time stretcher from libartamd64.so
def test_art64_cli_on_hip_library_writes_the_same_file_as_reference_art64(tmp_path, opts, rate_in, chans):
    from test_gpu_dropin import _write_wav
    src = str(tmp_path / "in.wav")
    _write_wav(src, rate_in, chans, 1.0)
    out_ref, out_amd = str(tmp_path / "ref.wav"), str(tmp_path / "amd.wav")
    r = subprocess.run([ART64_REF] + opts.split() + ["-q", "-y", src, out_ref], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    a = subprocess.run([ART64_AMD] + opts.split() + ["-q", "-y", src, out_amd], capture_output=True, text=True,
                       env=dict(os.environ, ARTAMD_STRICT="1"), timeout=600)
    assert a.returncode == 0, a.stderr[-1500:]
    with open(out_ref, "rb") as f1, open(out_amd, "rb") as f2:
        b1, b2 = f1.read(), f2.read()
    assert len(b1) == len(b2) and len(b1) > 10000
    assert b1 == b2, f"{sum(p != q for p, q in zip(b1, b2))} of {len(b1)} bytes differ"
