"""The fixed-point kernel's filter rows across calls (fir_matrix_i8.hip, ArtRowsCache): a context builds them once, for the canonical
period of its stream, and every later launch runs anchored on that period (its first slots computed and not stored).  Held to the oracle
call by call over streams of equal and unequal blocks, both numeric forms of the rows (interpolating / nearest filter with pass-through
slots), every place of a launch inside its period and inside its 4-frame blocks, short periods taken several at a time, position jumps
that leave the lattice, resets, flushes, channel groups; and against the same stream with ARTAMD_ROWS_CACHE=0 (rows rebuilt by every
launch from its own positions): the first call bit for bit, the others within a fraction of the bar."""
import json, os, subprocess, sys

import numpy as np
import pytest

import audio_resampler_amd as A
from _hip import HipResampler, tolerance_ok
from _oracle import noise, OracleResampler, BH, INTERP, LOWPASS, PRECISE

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

STREAMS = [
    # (channels, taps, filters, src, dst, fixed-ratio form, flags, blocks)
    (8, 988, 988, 44100, 48000, False, BH | INTERP, (40000, 40000, 40000, 40001, 39999, 12345, 40000)),
    (4, 988, 988, 44100, 48000, False, BH | INTERP, (60000, 60000, 7001, 60000)),
    (8, 988, 988, 96000, 44100, True, BH | INTERP | LOWPASS, (80000, 80000, 80001, 40000)),        # 147 filters, nearest filter, low-pass
    (4, 512, 32, 44100, 48000, False, BH, (50000, 50000, 50003)),                                      # nearest filter, no low-pass: pass-through slots
    (8, 988, 988, 44100, 88200, False, BH | INTERP, (30000, 30000, 30001)),                            # 2 outputs per period: many periods at a time
    (8, 512, 512, 48000, 32000, False, BH | INTERP, (60000, 60000, 59999)),                            # Q = 3 per 2 outputs
    (6, 988, 988, 44100, 48000, False, BH | INTERP, (40000, 40000, 40000)),                            # a group of 8
    (16, 640, 640, 48000, 44100, False, BH | INTERP, (30000, 30000, 30002)),                           # Q % 4 == 0 after the multiple? (147 x 160)
]


F32_STREAMS = [
    # the f32 streaming kernel on kept rows (kernel preference 6, or the library's own choice for short filters / narrow streams)
    (2, 380, 380, 44100, 48000, False, BH | INTERP, (120000, 120000, 120001, 50000, 120000)),           # BASELINE configs[1]
    (8, 988, 988, 44100, 48000, False, BH | INTERP, (70000, 70000, 70003)),
    (1, 48, 48, 44100, 48000, False, BH | INTERP, (300000, 300000, 299999)),                            # configs[0]: many ring epochs per call
    (16, 156, 156, 44100, 48000, False, BH | INTERP, (100000, 100000, 100001)),
    (2, 380, 160, 44100, 48000, True, BH | INTERP | LOWPASS, (150000, 150000, 150001)),                 # ART's form: nearest filter, low-pass
    (4, 512, 32, 44100, 48000, False, BH, (90000, 90000, 90003)),                                       # nearest filter, pass-through slots
    (8, 988, 988, 44100, 88200, False, BH | INTERP, (40000, 40000, 40001)),
]


def _play(stream, make, tail=None):
    ch, T, F, src, dst, fixed, flags, blocks = stream
    r = make(ch, T, F, flags, (float(src), float(dst), 0) if fixed else None)
    r.advance(T / 2)
    ratio = dst / src
    x, _ = noise(sum(blocks) * ch, state=(ch * 1000 + T + dst) | 1)
    x = x.reshape(-1, ch)
    outs = []; pos = 0
    for n in blocks:
        u, g, y = r.process(x[pos:pos + n], int(n * ratio) + 4000, 0.0 if fixed else ratio)
        assert u == n
        outs.append(np.array(y).copy()); pos += n
        if tail is not None:
            tail(r)
    return outs


@pytest.mark.parametrize("stream", STREAMS, ids=lambda s: f"{s[0]}ch_{s[1]}x{s[2]}_{s[3]}to{s[4]}")
def test_streams_on_cached_rows_against_the_oracle(stream):
    kinds = []
    got = _play(stream, lambda ch, T, F, fl, fx: HipResampler(ch, T, F, 0.0, fl, fixed=fx, kernel=7), tail=lambda r: kinds.append(int(r.fixed_point()[0])))
    want = _play(stream, lambda ch, T, F, fl, fx: OracleResampler(ch, T, F, 0.0, fl | PRECISE, fixed=fx))
    assert all(k == 1 for k in kinds), kinds                  # (every call ran in fixed point)
    for i, (y, yo) in enumerate(zip(got, want)):
        assert y.shape == yo.shape, (i, y.shape, yo.shape)
        ok, worst, rms = tolerance_ok(y, yo)
        assert ok, (i, worst, rms)


@pytest.mark.parametrize("stream", F32_STREAMS, ids=lambda s: f"{s[0]}ch_{s[1]}x{s[2]}_{s[3]}to{s[4]}")
def test_f32_streams_on_kept_rows_against_the_oracle(stream):
    kinds = []
    got = _play(stream, lambda ch, T, F, fl, fx: HipResampler(ch, T, F, 0.0, fl, fixed=fx, kernel=6), tail=lambda r: kinds.append((int(r.last_kernel()), int(r.fixed_point()[0]))))
    want = _play(stream, lambda ch, T, F, fl, fx: OracleResampler(ch, T, F, 0.0, fl | PRECISE, fixed=fx))
    assert all(k == (2, 0) for k in kinds), kinds             # (every call on the f32 matrix kernels)
    for i, (y, yo) in enumerate(zip(got, want)):
        assert y.shape == yo.shape, (i, y.shape, yo.shape)
        ok, worst, rms = tolerance_ok(y, yo)
        assert ok, (i, worst, rms)


def test_jumps_off_the_lattice_resets_and_a_flush():
    """advance () by a fraction of a frame starts a new canonical period; reset () returns to the first one's positions; the flush takes the general kernel"""
    ch, T, ratio = 8, 988, 48000 / 44100
    x, _ = noise(5 * 40000 * ch, state=4711); x = x.reshape(-1, ch)
    r = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=7); r.advance(T / 2)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE); o.advance(T / 2)
    first = None
    for k in range(5):
        blk = x[k * 40000:(k + 1) * 40000]
        if k == 2:
            r.advance(0.37); o.advance(0.37)
        if k == 4:
            r.reset(); o.reset(); r.advance(T / 2); o.advance(T / 2)
            blk = x[:40000]
        u, g, y = r.process(blk, 48000, ratio, and_flush=(k == 3))
        uo, go, yo = o.process(blk, 48000, ratio, and_flush=(k == 3))
        assert (u, g) == (uo, go)
        ok, worst, rms = tolerance_ok(np.array(y), np.array(yo))
        assert ok, (k, worst, rms)
        if k == 0:
            first = np.array(y).copy()
        if k == 4:          # the same positions as the stream's first call: the same rows, the same bits
            assert np.array_equal(first.view(np.uint32), np.array(y).view(np.uint32))


def _sessions(**env):
    r = subprocess.run([sys.executable, os.path.join(HERE, "_rows_sessions.py")], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return np.load(r.stdout.strip().splitlines()[-1])


def test_cached_rows_against_rows_rebuilt_by_every_launch():
    a, b = _sessions(), _sessions(ARTAMD_ROWS_CACHE="0")
    try:
        keys = sorted(a.files)
        assert keys == sorted(b.files) and len(keys) >= 12
        for k in keys:
            ya, yb = a[k], b[k]
            assert ya.shape == yb.shape
            if k.endswith("_call0"):                          # the canonical period IS the first launch's: nothing differs
                assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32)), k
            else:                                             # rows from phases ~1e-13 apart: the odd last bit of an output, far inside the bar
                d = np.abs(ya.astype(np.float64) - yb.astype(np.float64))
                assert d.max() <= 2.0 ** -23 * max(1.0, float(np.abs(yb).max())), (k, d.max())      # (one float ulp of a sample near full scale)
                # (measured: rows blended at a phase 1e-8 filter steps away move ~4 % of the outputs by one ulp; the cache serves a launch up to
                # 1e-6 steps from its canonical period — an output error of a tenth of half an ulp, as the streaming kernels allow between periods)
                assert np.count_nonzero(d) <= ya.size // 2, (k, np.count_nonzero(d), ya.size)
    finally:
        for f in (a, b):
            name = f.fid.name; f.close(); os.unlink(name)


def test_every_launch_of_a_downsampling_stream_runs_on_the_kept_rows():
    """a launch anchored on the canonical period starts up to period_in frames in front of its first output; with 64 zero frames in front of the
    history, streams whose period_in exceeds T/2 + 64 (96k -> 44.1k: 320 x 3 against 494 + 64) sent such launches back to rows of their own — a rebuild per call,
    and bits that moved with the cut.  The pad now covers a period's input (MfmaGeom.head_pad, both the f32 head and the fixed-point planes): ONE build per
    stream, every later launch a hit, on the fixed-point path (1M-frame calls) and on the f32 path (300,000-frame calls)."""
    import os, re, subprocess, sys
    code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
for src, dst, block in ((96000, 44100, 1048576), (96000, 44100, 300000), (48000, 32000, 1048576)):
    ch, taps = 8, 988
    rs = A.Resampler(ch, taps, taps, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE | A.INCLUDE_LOWPASS, fixed=(float(src), float(dst), 0)); rs.advance(taps / 2.0)
    x, _ = noise(block * ch); d_in = torch.from_numpy(x.reshape(block, ch)).cuda(); cap = int((block + taps) * dst / src) + 64; d_out = torch.empty(cap, ch, device="cuda")
    print("== stream", file=sys.stderr)
    for k in range(9): rs.process_device(d_in, block, d_out, cap, 0.0)
    torch.cuda.synchronize()
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, ARTAMD_ROWS_TRACE="1"), timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    streams = p.stderr.split("== stream")[1:]
    assert len(streams) == 3
    for text in streams:
        fixed = re.findall(r"rows: launch .*?cache (on|off) .*?set (-?\d+)\s+(BUILD|hit|-)", text)
        f32 = re.findall(r"rows \(f32\): launch .*?kept (\d)\s+ready (\d)", text)
        assert len(fixed) + len(f32) == 9, text[-1500:]
        if fixed:
            assert [f[0] for f in fixed] == ["on"] * 9 and [f[2] for f in fixed] == ["BUILD"] + ["hit"] * 8, fixed
        else:
            assert [k for k, r in f32] == ["1"] * 9 and [r for k, r in f32] == ["0"] + ["1"] * 8, f32
