"""GPU (-m gpu): the persistent ("streaming") f32 form of the matrix-core kernel (fir_matrix.hip, fir_mfma_stream_kernel; kernel
preference 6 pins it: by default regular launches run the fixed-point kernel, tests/test_gpu_fixed_point.py) against the
one-tile-per-workgroup kernel (kernel preference 5 pins that one): the same tiles, K order
and flush schedule, so every bit must agree — over channel counts (all compiled column groups), tap counts with odd and
even chunk counts, nearest-filter mode with and without pass-through samples, small and multi-launch calls, streaming across
calls (history seam) and flushes."""
import numpy as np
import pytest

import audio_resampler_amd as A
from _hip import HipResampler, tolerance_ok
from _oracle import OracleResampler, noise, BH, INTERP, LOWPASS, PRECISE

pytestmark = pytest.mark.gpu

CASES = [
    # (channels, taps, filters, src, dst, fixed, flags, blocks)
    (8, 988, 988, 44100, 48000, False, BH | INTERP, (70000, 50000, 131072)),          # headline shape (32 chunks)
    (4, 988, 988, 44100, 48000, False, BH | INTERP, (90000, 90000)),                  # config D's per-GPU shard
    (2, 380, 380, 44100, 48000, False, BH | INTERP, (200000, 100001)),                # config B: 13 chunks (odd: buffer parity flips per tile)
    (1, 380, 380, 44100, 48000, False, BH | INTERP, (300000,)),                       # mono: 64 periods per tile, half the columns idle
    (16, 156, 156, 44100, 48000, False, BH | INTERP, (60000, 60000)),
    (32, 988, 988, 44100, 48000, False, BH | INTERP, (30000, 30000)),                 # config D on one GPU
    (2, 380, 380, 44100, 48000, True, BH | INTERP | LOWPASS, (150000, 150000)),       # ART form: 160 x 380 nearest filter, SNAP, low-pass
    (8, 988, 988, 96000, 44100, True, BH | INTERP | LOWPASS, (140000, 140000)),       # config C's resampler: 147 x 988 nearest filter
    (2, 380, 320, 44100, 48000, False, BH, (120000, 120000)),                         # nearest filter, no low-pass: pass-through samples (F = 2P)
    (2, 380, 32, 44100, 48000, False, BH, (120000, 120000)),                          # nearest filter, F < P: five pass-through slots per period, two of them in one tile
    (2, 64, 160, 48000, 44100, False, BH, (150000, 100000)),                          # short filter, 3 chunks, P = 147 (calls of fewer ring epochs than a launch's table holds: both kernels cut the same launches)
    (8, 988, 988, 44100, 48000, False, BH | INTERP, (2000, 3000, 500, 9000)),         # small calls: fewer tiles than workgroups
]


def _play(r, x, blocks, ratio, fixed):
    outs, pos = [], 0
    for n in blocks:
        cap = int(n * ratio) + 4000
        u, g, y = r.process(x[pos:pos + n], cap, 0.0 if fixed else ratio)
        assert u == n
        assert r.last_kernel() == 2, "the matrix-core path must be the one under test"
        outs.append(y.copy())
        pos += n
    u, g, y = r.process(None, 8000, ratio, flush=True)
    outs.append(y.copy())
    return np.concatenate(outs)


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"c{c[0]}_t{c[1]}_f{c[2]}_{c[3]}to{c[4]}{'_fixed' if c[5] else ''}_{len(c[7])}calls")
def test_streaming_kernel_equals_tile_kernel_bit_for_bit(case):
    ch, T, F, src, dst, fixed, flags, blocks = case
    ratio = dst / src
    total = sum(blocks)
    x, _ = noise(total * ch, state=0xFACADE5EED | 1)
    x = x.reshape(total, ch)

    def make(kernel):
        # (the two kernels on the SAME rows and anchoring: rows built by every launch from its own positions — kept rows anchor the streaming
        # kernel's tiles on the stream's canonical period, which the tile kernel does not follow; tests/test_gpu_rows_cache.py holds that form to the oracle)
        r = HipResampler(ch, T, F, flags=flags, fixed=(float(src), float(dst), 0), kernel=kernel, keep_rows=False) if fixed else HipResampler(ch, T, F, 0.0, flags, kernel=kernel, keep_rows=False)
        r.advance(T / 2)
        return r

    y_stream = _play(make(6), x, blocks, ratio, fixed)
    y_tile = _play(make(5), x, blocks, ratio, fixed)
    assert y_stream.shape == y_tile.shape
    assert np.array_equal(y_stream.view(np.uint32), y_tile.view(np.uint32))


def test_streaming_kernel_long_call_with_many_ring_epochs_matches_oracle():
    """one call of 2.2 M frames x 2 channels: > 128 ring epochs (several launches, n_begin > 0) — against the
    double-accumulate oracle and the tile kernel"""
    ch, T, frames = 2, 988, 2200000
    ratio = 48000 / 44100
    x, _ = noise(frames * ch)
    x = x.reshape(frames, ch)
    outs = []
    for kernel in (6, 5):
        r = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=kernel)
        r.advance(T / 2)
        u, g, y = r.process(x, int(frames * ratio) + 4000, ratio)
        assert u == frames and r.last_kernel() == 2
        outs.append(y)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE)
    o.advance(T / 2)
    uo, go, yo = o.process(x[:400000], int(400000 * ratio) + 4000, ratio, threads=2)
    ok, worst, rms = tolerance_ok(outs[0][:go], yo)
    assert ok and rms < 2.0e-8, (worst, rms)


def test_irregular_launches_take_the_replaying_kernel_and_match_the_oracle():
    """launches the host-side regularity test rejects: (a) nearest-filter phases exactly on a half step (160/147 with 80 filters:
    every other slot sits at x.5 filter steps, where the reference's own fp64 noise decides the rounding per output) and (b) a
    call long enough for the position arithmetic's rounding to exceed the tolerance (4.7M frames x 988 filters).  Both must run
    on the one-tile-per-workgroup kernel with its exact per-output replay — (a) against the oracle, every output's filter
    choice included; (b) against the general kernel (both within the bar of the truth)."""
    ratio = 48000 / 44100
    ch, T, F, frames = 2, 380, 80, 150000
    x, _ = noise(frames * ch, state=0xBADC0FFEE | 1)
    x = x.reshape(frames, ch)
    r = HipResampler(ch, T, F, 0.0, BH, kernel=2)
    o = OracleResampler(ch, T, F, 0.0, BH | PRECISE)
    for b in (r, o):
        b.advance(T / 2)
    cap = int(frames * ratio) + 4000
    u, g, y = r.process(x, cap, ratio)
    uo, go, yo = o.process(x, cap, ratio, threads=2)
    assert (u, g) == (uo, go) and r.last_kernel() == 2
    ok, worst, rms = tolerance_ok(y, yo)
    assert ok, (worst, rms)

    ch, T, frames = 1, 988, 4700000
    x, _ = noise(frames * ch)
    x = x.reshape(frames, ch)
    outs = []
    for kernel in (2, 1):
        r = HipResampler(ch, T, T, 0.0, BH | INTERP, kernel=kernel)
        r.advance(T / 2)
        u, g, y = r.process(x, int(frames * ratio) + 4000, ratio)
        assert u == frames and r.last_kernel() == kernel
        outs.append(y)
    d = np.abs(outs[0].astype(np.float64) - outs[1].astype(np.float64))
    assert np.all(d <= 2.0 * 2.0 ** -23 * np.maximum(1.0, np.abs(outs[1]))), d.max()


@pytest.mark.parametrize("shape", [(1, 48, 48, BH | INTERP, 44100, 48000), (2, 48, 160, BH, 44100, 48000), (8, 156, 156, BH | INTERP, 48000, 32000), (4, 48, 48, BH | INTERP, 44100, 88200)],
                         ids=["mono_t48", "stereo_t48_nearest_passthrough", "c8_t156_48to32", "c4_t48_2x"])
def test_a_call_of_more_ring_epochs_than_a_table_holds_is_one_launch_with_the_same_bits(shape):
    """short filters: a ring epoch is a few hundred input frames, a 1M-frame call many hundred segments, a launch's table 192 — a call
    that runs on a streaming matrix-core kernel is handed over whole (the kernels follow the lattice from the launch's first
    period); kernel preference 5, the one-tile-per-workgroup kernel, which replays positions from the table, keeps the cut
    launches (each anchors its tiles at its own first output: other K origins, other float roundings — not the same bits): all
    three against the oracle"""
    ch, T, F, flags, src, dst = shape
    frames = 700000
    ratio = dst / src
    x, _ = noise(frames * ch, state=0x5E65 | 1)
    x = x.reshape(frames, ch)
    outs = {}
    for kernel in (6, 5, 7):
        r = HipResampler(ch, T, F, 0.0, flags, kernel=kernel); r.advance(T / 2)
        u, g, y = r.process(x, int(frames * ratio) + 4000, ratio)
        assert u == frames and r.last_kernel() == 2
        outs[kernel] = np.array(y).copy()
    o = OracleResampler(ch, T, F, 0.0, flags | PRECISE); o.advance(T / 2)
    uo, go, yo = o.process(x, int(frames * ratio) + 4000, ratio, threads=8)
    for kernel in (6, 5, 7):
        assert outs[kernel].shape == np.array(yo).shape and tolerance_ok(outs[kernel], np.array(yo))[0], kernel
