"""GPU: the DEFAULT (fast) numeric mode compared with the reference at the PCM level — the bytes a user's file holds.

Every other byte-identity test of the tools runs with ARTAMD_STRICT=1 (the bit-exact source-order kernel).  Here the
reference's own ART tool, compiled from its own art.c and linked against libartamd.so WITHOUT ARTAMD_STRICT, and
tools/art_gpu.py likewise, convert >= 10 s of audio in the BASELINE shapes to 16- and 24-bit files, which are compared
sample by sample with the file the reference's source-order build (oracle/_ref/art_strict) writes.

What must hold (reference decimator.c:245-283: scale, subtract shaped error, add dither, round; art.c:1011-1067):
  * headers and lengths equal; no clipped-count difference;
  * WITHOUT noise shaping a differing sample is off by exactly ONE step, and the rate is the float paths' distance in
    steps: P(flip) = E|dy| * 2^(bits-1), dy measured on the same conversion written as 32-bit float;
  * WITH noise shaping the first flip changes the error fed back, and two quantisations of (nearly) the same signal run
    apart for good — the reference's OWN two builds (Makefile flags vs source order, oracle/_ref/art_make vs art_strict) differ in
    65 - 80 % of the samples by up to 8 steps.  There: each file's error against the un-quantised signal has the same rms (1 %), and
    the difference between the files is no larger than the reference's own two builds' difference.
The reference's Makefile build is run beside every case as the yardstick: our default mode must be no further from art_strict
than the reference's shipped build is (x a margin), measured on the same input.
"""
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import _pcm as P
from _oracle import ORACLE_DIR

pytestmark = pytest.mark.gpu
REF = os.path.join(ORACLE_DIR, "_ref")
ART_AMD, ART_REF, ART_MAKE = (os.path.join(REF, n) for n in ("art_amd", "art_strict", "art_make"))
ART_GPU = os.path.join(os.path.dirname(ORACLE_DIR), "tools", "art_gpu.py")
needs_ref = pytest.mark.skipif(not all(os.path.exists(p) for p in (ART_AMD, ART_REF, ART_MAKE)), reason="oracle/_ref/art_* not built (needs /root/reference at build time)")
SECONDS = 10.0

# the BASELINE.json shapes as the ART tool runs them: name -> (options, input rate, channels)
SHAPES = {
    "P_mono_48x48": ("-1 -r48000", 44100, 1),               # configs[0]: 48 filters < 160 phases: interpolating
    "B_stereo_380": ("-3 -r48000", 44100, 2),               # configs[1] via ART: 160 x 380, nearest filter, SNAP
    "A_8ch_988": ("-4 -r48000", 44100, 8),                  # the headline's conversion via ART: 160 x 988
    "C_8ch_down_lp_biquads": ("-4 -r44100 -p", 96000, 8),   # configs[2]: 147 x 988 + low-pass + biquad cascade
    "D_32ch_988": ("-4 -r48000", 44100, 32),                # configs[3]
}


def _run(tool, opts, src, dst, env_extra=None):
    env = dict(os.environ)
    env.pop("ARTAMD_STRICT", None)
    env.pop("ARTAMD_KERNEL", None)
    env.update(env_extra or {})
    cmd = ([sys.executable, tool] if tool.endswith(".py") else [tool]) + opts.split() + ["-y", src, dst]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    m = re.search(r"(\d+) samples were clipped", p.stderr)
    return int(m.group(1)) if m else 0


def _convert_all(tmp_path, tag, tool, opts, src, env_extra=None):
    """the conversion written as float, and as 16 / 24 bits with and without dither and shaping"""
    out = {}
    for key, extra in (("f32", "-o32"), ("16", "-o16"), ("24", "-o24"), ("16_flat", "-o16 -n0"), ("24_flat", "-o24 -n0"), ("16_bare", "-o16 -d0 -n0"), ("24_bare", "-o24 -d0 -n0")):
        dst = str(tmp_path / f"{tag}_{key}.wav")
        clips = _run(tool, f"{opts} {extra}", src, dst, env_extra)
        out[key] = P.read_wav(dst) + (clips,)
    return out


_REFERENCE_FILES = {}


def _reference_files(tmp_path_factory, name):
    """the shape's input file and what the reference's two builds make of it (CPU work: once per shape and session)"""
    if name not in _REFERENCE_FILES:
        opts, rate, ch = SHAPES[name]
        d = tmp_path_factory.mktemp("pcm_" + name)
        src = str(d / "in.wav")
        P.write_float_wav(src, rate, P.signal(rate, ch, SECONDS))
        _REFERENCE_FILES[name] = (src, _convert_all(d, "strict", ART_REF, opts + " -q", src), _convert_all(d, "make", ART_MAKE, opts + " -q", src))
    return _REFERENCE_FILES[name]


def _check(name, ours, strict, make, report):
    hdr_f, y_ours, _ = ours["f32"]
    _, y_ref, _ = strict["f32"]
    _, y_make, _ = make["f32"]
    assert hdr_f == strict["f32"][0] and y_ours.shape == y_ref.shape
    dy = np.abs(y_ours.astype(np.float64) - y_ref.astype(np.float64))
    dy_make = np.abs(y_make.astype(np.float64) - y_ref.astype(np.float64))
    e_dy, e_make = float(dy.mean()), float(dy_make.mean())
    report.append(f"{name}: float E|dy| {e_dy:.3e} (max {dy.max():.3e}); reference Makefile build vs source order: {e_make:.3e} (max {dy_make.max():.3e})")
    # the float paths themselves: inside the parity bar, and no further from the source-order build than 4 x the reference's own shipped build
    # (C's biquad cascade amplifies re-association differences: there the bound is relative to the reference's own)
    assert e_dy <= max(4.0 * e_make, 2.0e-8), (e_dy, e_make)
    for bits in (16, 24):
        step = 2.0 ** (bits - 1)
        for mode in ("bare", "flat"):        # no shaping: every decision is local
            h, a, clips = ours[f"{bits}_{mode}"]
            hr, b, clips_ref = strict[f"{bits}_{mode}"]
            _, m, _ = make[f"{bits}_{mode}"]
            assert h == hr and a.shape == b.shape, name
            st, st_make = P.compare_pcm(a, b), P.compare_pcm(m, b)
            predicted = e_dy * step
            report.append(f"{name} -o{bits} {mode:5s}: {st['differ']} of {st['samples']} samples differ ({st['rate']:.3e}; predicted E|dy| x 2^{bits - 1} = {predicted:.3e}), "
                          f"max {st['max_abs']} step; reference's own builds: {st_make['rate']:.3e}, max {st_make['max_abs']}")
            assert clips == clips_ref
            # a flip needs the boundary between the two values: |dy| x step < 1 keeps it to one step
            max_steps = int(math.ceil(float(dy.max()) * step)) if dy.max() * step > 1.0 else 1
            assert st["max_abs"] <= max_steps, (name, bits, mode, st)
            # the rate IS the float distance in steps (binomial scatter: 5 sigma + 1.25 for |dy| not being uniform over the step)
            n = st["samples"]
            bound = 1.25 * predicted + 5.0 * math.sqrt(max(predicted, 1.0 / n) / n)
            assert st["rate"] <= bound, (name, bits, mode, st, predicted)
        # dither + ATH shaping (the tool's default): the quantisations part at the first flip
        h, a, clips = ours[str(bits)]
        hr, b, clips_ref = strict[str(bits)]
        _, m, _ = make[str(bits)]
        assert h == hr and a.shape == b.shape and clips == clips_ref, name
        st, st_make = P.compare_pcm(a, b), P.compare_pcm(m, b)
        err_ours = float(np.sqrt(np.mean((a / step - y_ref.astype(np.float64)) ** 2)))
        err_ref = float(np.sqrt(np.mean((b / step - y_ref.astype(np.float64)) ** 2)))
        report.append(f"{name} -o{bits} shaped: differ {st['rate']:.3f} of samples, max {st['max_abs']} steps, rms {st['rms_steps']:.3f} steps (reference's own builds: {st_make['rate']:.3f}, "
                      f"max {st_make['max_abs']}, rms {st_make['rms_steps']:.3f}); error vs the un-quantised signal: ours {err_ours * step:.4f} steps rms, reference {err_ref * step:.4f}")
        # each file is as good a rendering of the signal as the reference's: same error power against the un-quantised signal (1 %)
        assert abs(err_ours / err_ref - 1.0) < 0.01, (name, bits, err_ours, err_ref)
        # ... and the two are no further apart than the reference's own two builds (which decorrelate the same way)
        assert st["max_abs"] <= st_make["max_abs"] + 2 and st["rms_steps"] <= 1.15 * st_make["rms_steps"] + 0.05, (name, bits, st, st_make)


@needs_ref
@pytest.mark.parametrize("name", list(SHAPES))
def test_reference_art_on_the_library_in_default_mode_at_the_pcm_level(tmp_path, tmp_path_factory, name, capsys):
    opts, rate, ch = SHAPES[name]
    src, strict, make = _reference_files(tmp_path_factory, name)
    ours = _convert_all(tmp_path, "amd", ART_AMD, opts + " -q", src)
    report = []
    _check(name, ours, strict, make, report)
    with capsys.disabled():
        print("\n" + "\n".join(report))


@needs_ref
@pytest.mark.parametrize("kernel,label", [("1", "general kernel"), ("6", "f32 matrix-core streaming kernel"), ("7", "fixed-point matrix-core kernel wherever it can run")])
def test_every_pinned_kernel_at_the_pcm_level(tmp_path, tmp_path_factory, kernel, label, capsys):
    """the same comparison with the kernel pinned (ARTAMD_KERNEL): the fixed-point kernel against the f32 kernels, on the headline's conversion"""
    opts, rate, ch = SHAPES["A_8ch_988"]
    src, strict, make = _reference_files(tmp_path_factory, "A_8ch_988")
    ours = _convert_all(tmp_path, "amd", ART_AMD, opts + " -q", src, {"ARTAMD_KERNEL": kernel})
    report = []
    _check(f"A_8ch_988 [{label}]", ours, strict, make, report)
    with capsys.disabled():
        print("\n" + "\n".join(report))


@needs_ref
@pytest.mark.parametrize("name", ["A_8ch_988", "C_8ch_down_lp_biquads"])
def test_device_resident_art_tool_in_default_mode_at_the_pcm_level(tmp_path, tmp_path_factory, name, capsys):
    opts, rate, ch = SHAPES[name]
    src, strict, make = _reference_files(tmp_path_factory, name)
    ours = _convert_all(tmp_path, "gpu", ART_GPU, opts + " -q", src)
    report = []
    _check(f"{name} [tools/art_gpu.py]", ours, strict, make, report)
    with capsys.disabled():
        print("\n" + "\n".join(report))


@needs_ref
def test_clipped_counts_in_default_mode(tmp_path, capsys):
    """+5 dB on the stereo shape drives the peaks past full scale: the decimator's clipped-sample count (art.c:1066, 1148) must be the reference's.
    Without shaping a flip moves the count only where it crosses the clip level itself; with shaping the two quantisations differ sample by
    sample (above) and so may the count, by a few per cent at most."""
    opts, rate, ch = SHAPES["B_stereo_380"]
    src = str(tmp_path / "in.wav")
    P.write_float_wav(src, rate, P.signal(rate, ch, SECONDS))
    report = []
    for extra, slack in (("-o16 -d0 -n0", 0.0), ("-o16 -n0", 0.0), ("-o16", 0.05), ("-o24 -d0 -n0", 0.0), ("-o24", 0.05)):
        o = f"{opts} -g5 {extra}"
        c_ref = _run(ART_REF, o, src, str(tmp_path / "ref.wav"))
        c_amd = _run(ART_AMD, o, src, str(tmp_path / "amd.wav"))
        c_gpu = _run(ART_GPU, o, src, str(tmp_path / "gpu.wav"))
        report.append(f"B_stereo_380 -g5 {extra}: clipped samples reference {c_ref}, art.c on the library {c_amd}, tools/art_gpu.py {c_gpu}")
        assert c_ref > 100
        assert abs(c_amd - c_ref) <= 2 + slack * c_ref and abs(c_gpu - c_ref) <= 2 + slack * c_ref, report[-1]
    with capsys.disabled():
        print("\n" + "\n".join(report))


def test_asrc_and_the_headline_call_at_the_pcm_level(capsys):
    """The two BASELINE shapes no ART option reaches — configs[4] (stereo ASRC, per-block ratio, nearest filter: the general kernel) and the
    headline itself (8 ch x 988 interpolating, one device-resident call big enough for the fixed-point slab kernel) — through the Python mirror:
    the library's default mode against the oracle's source-order float loop, both decimated by the (bit-exact) decimator."""
    import audio_resampler_amd as A
    from _oracle import OracleResampler, BH, INTERP, DITHER_HP, SHAPE_ATH
    from audio_resampler_amd.synth import noise
    report = []

    def decimate(y, bits, flags):
        d = A.Decimator(y.shape[1], bits, (bits + 7) // 8, 1.0, 48000, flags)
        got, clips = d.process(np.ascontiguousarray(y))
        raw = got.reshape(-1, (bits + 7) // 8).astype(np.int64)
        v = sum(raw[:, i] << (8 * i) for i in range(raw.shape[1]))
        v -= (v >> (8 * raw.shape[1] - 1)) << (8 * raw.shape[1])
        return v.reshape(y.shape), clips

    def compare(name, y, yo):
        dy = np.abs(y.astype(np.float64) - yo.astype(np.float64))
        e_dy = float(dy.mean())
        report.append(f"{name}: float E|dy| {e_dy:.3e} (max {dy.max():.3e})")
        for bits in (16, 24):
            step = 2.0 ** (bits - 1)
            for flags, mode in ((0, "bare"), (DITHER_HP, "dither")):
                a, ca = decimate(y, bits, flags)
                b, cb = decimate(yo, bits, flags)
                st = P.compare_pcm(a, b)
                predicted = e_dy * step
                report.append(f"{name} {bits}-bit {mode}: {st['differ']} of {st['samples']} differ ({st['rate']:.3e}; predicted {predicted:.3e}), max {st['max_abs']} step")
                assert ca == cb and st["max_abs"] <= 1
                n = st["samples"]
                assert st["rate"] <= 1.25 * predicted + 5.0 * math.sqrt(max(predicted, 1.0 / n) / n), (name, bits, mode, st, predicted)
            a, ca = decimate(y, bits, DITHER_HP | SHAPE_ATH)
            b, cb = decimate(yo, bits, DITHER_HP | SHAPE_ATH)
            err_a = float(np.sqrt(np.mean((a / step - yo.astype(np.float64)) ** 2)))
            err_b = float(np.sqrt(np.mean((b / step - yo.astype(np.float64)) ** 2)))
            st = P.compare_pcm(a, b)
            report.append(f"{name} {bits}-bit shaped: differ {st['rate']:.3f}, max {st['max_abs']} steps; error vs un-quantised: ours {err_a * step:.4f} steps rms, oracle {err_b * step:.4f}")
            assert ca == cb and abs(err_a / err_b - 1.0) < 0.01 and st["max_abs"] <= 12

    # configs[4]: stereo ASRC, 380 x 380, nearest filter, ratio 48000/44100 x (1 +- 100 ppm) moving per 4,096-frame block
    ch, T, block, blocks = 2, 380, 4096, 120           # 11 s of audio
    x, _ = noise(ch * block * blocks)
    x = 0.6 * x.reshape(-1, ch)
    g = A.Resampler(ch, T, T, 0.0, A.BLACKMAN_HARRIS)
    o = OracleResampler(ch, T, T, 0.0, BH)
    g.advance(T / 2); o.advance(T / 2)
    ys, yos = [], []
    for k in range(blocks):
        ratio = 48000 / 44100 * (1.0 + 100e-6 * math.sin(2 * math.pi * k / 64))
        seg = x[k * block:(k + 1) * block]
        u, n, y = g.process(seg, 4600, ratio)
        uo, no, yo = o.process(seg, 4600, ratio)
        assert (u, n) == (uo, no)
        ys.append(y[:n]); yos.append(yo[:no])
    compare("E_asrc_2ch_380_nearest [general kernel]", np.concatenate(ys), np.concatenate(yos))

    # the headline: 8 ch x 988 x 988 interpolating, ONE call of 524,288 frames (11.9 s) — the fixed-point kernel's domain
    ch, T, frames = 8, 988, 1 << 19
    x, _ = noise(ch * frames)
    x = 0.6 * x.reshape(-1, ch)
    cap = int(frames * 48000 / 44100) + T
    for pref, label in ((0, "library's choice"), (7, "fixed point pinned"), (6, "f32 pinned")):
        g = A.Resampler(ch, T, T, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE)
        g.set_kernel(pref)
        g.advance(T / 2)
        u, n, y = g.process(x, cap, 48000 / 44100)
        if pref == 0:
            o = OracleResampler(ch, T, T, 0.0, BH | INTERP)
            o.advance(T / 2)
            uo, no, yo = o.process(x, cap, 48000 / 44100, threads=ch)
        assert (u, n) == (uo, no)
        compare(f"A_headline_8ch_988_interp, one 524,288-frame call [{label}; fixed point ran: {g.fixed_point()}]", y[:n], yo[:no])
    with capsys.disabled():
        print("\n" + "\n".join(report))
