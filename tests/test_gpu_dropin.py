"""GPU: the reference's OWN test program — compiled from its own artest.c against its own headers, linked
against libartamd.so instead of resampler.c/biquad.c/decimator.c (oracle/Makefile target `dropin`) — must
print the same statistics as when it is linked against the reference's own DSP sources.
The binary only exists where /root/reference was available at build time (it travels to the GPU box).
Likewise the reference's ART command-line tool: art.c ALONE, every DSP entry point (resampler.h, biquad.h, decimator.h,
stretch.h) resolved by the library."""
import os
import re
import subprocess

import pytest

import _golden as G
from _oracle import ORACLE_DIR

pytestmark = pytest.mark.gpu
EXE = os.path.join(ORACLE_DIR, "_ref", "artest_amd")


def run(args, strict=True):
    env = dict(os.environ)
    if strict:
        env["ARTAMD_STRICT"] = "1"
    p = subprocess.run([EXE] + args.split(), capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = {}
    for line in p.stderr.splitlines():
        m = re.search(r"(input|output|decimate) \(-w\d\): count =\s*(\d+), checksum = ([0-9a-f]{16})", line)
        if m:
            rec[m.group(1)] = (int(m.group(2)), m.group(3))
            c = re.search(r"clipped samples = (\d+)", line)
            if c:
                rec["clips"] = int(c.group(1))
    return rec


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/artest_amd not built (needs /root/reference at build time)")
@pytest.mark.parametrize("args", ["-1 -c1 -n2 -s44100 -d48000", "-3 -c2 -n2 -s44100 -d48000", "-3 -e -c2 -n2 -s44100 -d48000",
                                  "-4 -c8 -n2 -o16 -s44100 -d48000", "-4 -e -l -c8 -n1 -s96000 -d44100", "-3 -c2 -n2 -p -s44100 -d48000"])
def test_reference_artest_binary_on_the_hip_library_matches_reference_checksums(args):
    want = G.kat()["strict"][args]
    got = run(args)
    assert got["input"] == (want["input"]["count"], want["input"]["checksum"])
    assert got["output"] == (want["output"]["count"], want["output"]["checksum"])
    if "decimate" in want:
        assert got["decimate"] == (want["decimate"]["count"], want["decimate"]["checksum"])
        assert got["clips"] == want["decimate"]["clips"]


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/artest_amd not built")
def test_reference_artest_binary_planar_simulator_and_default_mode():
    # -v routes through resampleProcess / decimateProcessLE (planar); must equal the interleaved run
    a = run("-3 -c2 -n2 -o16 -s44100 -d48000")
    b = run("-3 -c2 -n2 -o16 -v -s44100 -d48000")
    assert a == b
    # default (fast) numeric mode: same frame counts, program's own self-checks pass (exit code 0)
    c = run("-4 -c8 -n2 -s44100 -d48000", strict=False)
    assert c["output"][0] == G.kat()["strict"]["-4 -c8 -n2 -o16 -s44100 -d48000"]["output"]["count"]


# ------------------------------------------------------------------------------------------------
# the whole ART command-line tool (reference art.c + stretch.c, unmodified) on the HIP library
# ------------------------------------------------------------------------------------------------
ART_AMD = os.path.join(ORACLE_DIR, "_ref", "art_amd")
ART_REF = os.path.join(ORACLE_DIR, "_ref", "art_strict")


def _write_wav(path, rate, channels, seconds, bits=16):
    import wave
    import numpy as np
    from _oracle import noise
    n = int(rate * seconds)
    x, _ = noise(n * channels)
    t = np.arange(n)[:, None] / rate
    sig = 0.35 * x.reshape(n, channels) + 0.4 * np.sin(2 * np.pi * (440.0 * (1 + np.arange(channels))[None, :]) * t)
    pcm = np.clip(np.round(sig * 32767), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(pcm.tobytes())


@pytest.mark.skipif(not (os.path.exists(ART_AMD) and os.path.exists(ART_REF)), reason="oracle/_ref/art_* not built")
@pytest.mark.parametrize("opts,rate_in,chans", [
    ("-4 -r48000", 44100, 2),                # ART default: fixed ratio 160x988 no-lerp, end-point extrapolation, 16-bit dither+ATH shaping
    ("-3 -r44100 -p", 96000, 2),             # downsample with auto low-pass + cascaded biquad pre-filter (BASELINE configs[2] shape)
    ("-2 -r48000 -o24 -d1 -n2", 44100, 1),   # 24-bit out, flat dither, 2nd-order shaping
    ("-3 -r32000 -x -o8", 48000, 2),         # no extrapolation, 8-bit output
    ("-2 --tempo=1.25", 44100, 2),           # time stretch only (stretch.h from the library too: art_amd is art.c alone)
    ("-3 -r48000 --pitch=-300", 44100, 1),   # pitch shift: stretch, then resample by the inverse
    ("-2 --tempo=0.3 -o24", 44100, 2),       # below 0.5: the cascaded (dual) stretcher
    ("-2 --duration=+0.4", 44100, 2),        # a target duration becomes a tempo ratio
    ("-2 -r22050 -p --tempo=1.25", 44100, 2),    # stretch + downsampling pre-filter (which ART applies to a buffer it then does not use)
])
def test_art_cli_on_hip_library_writes_the_same_file_as_reference_art(tmp_path, opts, rate_in, chans):
    src = str(tmp_path / "in.wav")
    _write_wav(src, rate_in, chans, 1.5)
    out_ref, out_amd = str(tmp_path / "ref.wav"), str(tmp_path / "amd.wav")
    env = dict(os.environ, ARTAMD_STRICT="1")
    r = subprocess.run([ART_REF] + opts.split() + ["-q", "-y", src, out_ref], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    a = subprocess.run([ART_AMD] + opts.split() + ["-q", "-y", src, out_amd], capture_output=True, text=True, env=env, timeout=600)
    assert a.returncode == 0, a.stderr[-1500:]
    with open(out_ref, "rb") as f1, open(out_amd, "rb") as f2:
        b1, b2 = f1.read(), f2.read()
    assert len(b1) == len(b2) and len(b1) > 10000
    assert b1 == b2, f"{sum(x != y for x, y in zip(b1, b2))} of {len(b1)} bytes differ"


# ------------------------------------------------------------------------------------------------
# tools/art_gpu.py: our own device-resident ART counterpart (ingest, biquads, resample, decimate all in HBM)
# ------------------------------------------------------------------------------------------------
@pytest.mark.skipif(not os.path.exists(ART_REF), reason="oracle/_ref/art_strict not built")
@pytest.mark.parametrize("opts,rate_in,chans", [
    ("-4 -r48000", 44100, 2), ("-3 -r44100 -p", 96000, 2), ("-2 -r48000 -o24 -d1 -n2", 44100, 1), ("-3 -r32000 -x -o8", 48000, 2),
    ("-1 -r48000", 44100, 3),            # 48 filters < 160 phases: interpolation stays on; 3 channels => extensible header
    ("-3 -r96000 -p -o24", 44100, 2),    # upsampling with the biquad POST-filter
    ("-2 --tempo=1.25", 44100, 2),       # device-resident time stretcher, no resampling
    ("-3 -r48000 --pitch=-300", 44100, 1),
    ("-2 --tempo=0.3 -o24", 44100, 2),   # cascaded stretcher
    ("-2 --duration=+0.4", 44100, 2), ("-3 -r48000 --duration=1.2", 44100, 1), ("-2 --duration=0:01.0", 44100, 2),
    ("-2 -r22050 -p --tempo=1.25", 44100, 2),    # ART's pre-filter is a no-op on the output when stretching: reproduced
])
def test_device_resident_art_tool_writes_the_same_file_as_reference_art(tmp_path, opts, rate_in, chans):
    import sys
    src = str(tmp_path / "in.wav")
    _write_wav(src, rate_in, chans, 1.5)
    out_ref, out_gpu = str(tmp_path / "ref.wav"), str(tmp_path / "gpu.wav")
    r = subprocess.run([ART_REF] + opts.split() + ["-q", "-y", src, out_ref], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    tool = os.path.join(os.path.dirname(ORACLE_DIR), "tools", "art_gpu.py")
    a = subprocess.run([sys.executable, tool] + opts.split() + ["-q", "-y", src, out_gpu], capture_output=True, text=True,
                       env=dict(os.environ, ARTAMD_STRICT="1"), timeout=600)
    assert a.returncode == 0, a.stderr[-2000:]
    with open(out_ref, "rb") as f1, open(out_gpu, "rb") as f2:
        b1, b2 = f1.read(), f2.read()
    assert len(b1) == len(b2), (len(b1), len(b2))
    assert b1 == b2, f"{sum(x != y for x, y in zip(b1, b2))} of {len(b1)} bytes differ (first at {next(i for i, (x, y) in enumerate(zip(b1, b2)) if x != y)})"
