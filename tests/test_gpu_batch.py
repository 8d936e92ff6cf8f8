"""GPU: resampleProcessBatchInterleavedDevice — many independent resampler contexts, one launch.  Every context has a twin
driven one call at a time through resampleProcessInterleavedDevice; counts, samples (bit for bit), kernel choice and the
final flush must agree, whatever mix of contexts is in the batch: free-ratio streams whose ratio moves every call (ASRC),
fixed-ratio ones, EXTEND mode, and the kinds that cannot share a launch (strict order, endpoint extrapolation, a block big
enough for the matrix-core path) and are made one by one inside the same batch call."""
import numpy as np
import pytest

import audio_resampler_amd as A

pytestmark = pytest.mark.gpu
BH, IN, LP = A.BLACKMAN_HARRIS, A.SUBSAMPLE_INTERPOLATE, A.INCLUDE_LOWPASS

# (channels, taps, filters, flags, fixed (src, dst) or None, free ratio, max block)
STREAMS = [
    (2, 48, 48, BH | IN, None, 48000 / 44100, 1500),
    (1, 156, 320, BH | IN, None, 0.731, 900),
    (3, 64, 64, BH, None, 2.0, 700),
    (8, 380, 380, BH | IN | A.EXTEND_CONVOLUTION_MATH, None, 44100 / 48000, 1200),
    (5, 32, 16, IN, None, 1 / 3.0, 2000),
    (2, 380, 160, BH | LP, (44100.0, 48000.0), 0.0, 1800),
    (4, 156, 147, BH | IN | LP, (96000.0, 44100.0), 0.0, 2500),
    (2, 48, 48, BH | IN | A.RESAMPLE_STRICT_ORDER, None, 48000 / 44100, 600),            # one by one: strict order
    (2, 64, 64, BH | IN | A.EXTRAPOLATE_ENDPOINTS, None, 1.25, 900),                    # one by one: extrapolation
    (8, 988, 160, BH | LP, (44100.0, 48000.0), 0.0, 62000),                              # big blocks: matrix-core path
    (9, 16, 7, BH | IN, None, 3.7, 400),
]


def bits(a, width):
    return np.ascontiguousarray(a).view(np.uint32 if width == 32 else np.uint64)


@pytest.mark.parametrize("width", [32, 64])
def test_batched_calls_equal_single_calls(width):
    torch = pytest.importorskip("torch")
    B = A.binding(width)
    dt, tdt = (np.float32, torch.float32) if width == 32 else (np.float64, torch.float64)
    rng = np.random.default_rng(11 + width)
    mk = lambda s: B.Resampler(s[0], s[1], s[2], 0.0, s[3], None if s[4] is None else (s[4][0], s[4][1], 0))
    batch, single = [mk(s) for s in STREAMS], [mk(s) for s in STREAMS]
    for r, s in zip(batch + single, STREAMS + STREAMS):
        r.advance(s[1] / 2)
    n = len(STREAMS)
    cap_max = [int(s[6] * max(s[5] if s[4] is None else s[4][1] / s[4][0], 1.0) * 1.02 + 4 * s[1] + 64) for s in STREAMS]
    d_out_b = [torch.zeros(c, s[0], device="cuda", dtype=tdt) for c, s in zip(cap_max, STREAMS)]
    d_out_s = [torch.zeros(c, s[0], device="cuda", dtype=tdt) for c, s in zip(cap_max, STREAMS)]
    kernels_seen = set()
    for rnd in range(14):
        d_in, n_in, caps, ratios = [], [], [], []
        for i, s in enumerate(STREAMS):
            k = int(rng.integers(0, s[6])) if rng.integers(0, 6) else int(rng.integers(0, 3))
            if i == 9: k = s[6] - int(rng.integers(0, 100))                # keep that stream's blocks big
            x = (rng.random((max(k, 1), s[0])) - 0.5).astype(dt)
            d_in.append(torch.from_numpy(x).cuda()); n_in.append(k)
            caps.append(cap_max[i] if rng.integers(0, 5) else int(rng.integers(1, 200)))
            base = s[5] if s[4] is None else s[4][1] / s[4][0]
            ratios.append(base * (1 + rng.uniform(-3e-4, 3e-4)) if s[4] is None and rng.integers(0, 2) else base)
        got = B.process_batch_device(batch, d_in, n_in, d_out_b, caps, ratios)
        for i in range(n):
            u, g = single[i].process_device(d_in[i], n_in[i], d_out_s[i], caps[i], ratios[i])
            assert got[i] == (u, g), (rnd, i, got[i], (u, g))
            assert np.array_equal(bits(d_out_b[i][:g].cpu().numpy(), width), bits(d_out_s[i][:g].cpu().numpy(), width)), (rnd, i)
            assert batch[i].last_kernel() == single[i].last_kernel(), (rnd, i)
            assert batch[i].state() == single[i].state(), (rnd, i)
            kernels_seen.add((i, single[i].last_kernel()))
    assert (9, 2) in kernels_seen          # the big-block stream did take the matrix-core path inside the batch call
    # the contexts are interchangeable afterwards: flush both sides one by one
    for i, s in enumerate(STREAMS):
        ub, gb, yb = batch[i].process(None, 3 * s[1], STREAMS[i][5] or 1.0, flush=True)
        us, gs, ys = single[i].process(None, 3 * s[1], STREAMS[i][5] or 1.0, flush=True)
        assert (ub, gb) == (us, gs) and np.array_equal(bits(yb, width), bits(ys, width)), i


def test_batch_argument_checks():
    torch = pytest.importorskip("torch")
    B = A.binding(32)
    r = B.Resampler(2, 48, 48)
    d = torch.zeros(64, 2, device="cuda")
    with pytest.raises(RuntimeError):
        B.process_batch_device([r, r], [d, d], [10, 10], [d, d], [32, 32], [1.0, 1.0])          # a context twice
    assert B.process_batch_device([], [], [], [], [], []) == []


def test_shared_filter_bank_lifecycle():
    """Contexts with the same preset share one device bank (reference-counted) and keep private host rows: freeing one must
    not disturb the others, a different preset gets its own bank, and writing into one context's `filters` rows (a plain
    array in the reference too) must not leak into contexts opened later."""
    import ctypes as C
    import time
    B = A.binding(32)
    x = (np.random.default_rng(2).random((3000, 2)) - 0.5).astype(np.float32)

    def run(r):
        r.advance(190)
        return r.process(x, 4000, 48000 / 44100)[2]

    t0 = time.perf_counter(); a = B.Resampler(2, 380, 380); t1 = time.perf_counter()
    b = B.Resampler(2, 380, 380); t2 = time.perf_counter()
    c = B.Resampler(2, 380, 380, 0.9)
    assert np.array_equal(a.bank(), b.bank()) and not np.array_equal(a.bank(), c.bank())
    ya = run(a)
    a.close()                                              # b keeps the shared bank alive
    yb = run(b)
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
    assert not np.array_equal(run(c).view(np.uint32), yb.view(np.uint32))
    row = np.ctypeslib.as_array(b.c.filters[5], shape=(380,)); row[:] = 0.0          # scribble on b's private host rows
    d = B.Resampler(2, 380, 380)
    assert np.array_equal(run(d).view(np.uint32), yb.view(np.uint32))
    b.close(); c.close(); d.close()
    e = B.Resampler(2, 380, 380)                            # last reference gone: rebuilt from scratch, same bank
    assert np.array_equal(run(e).view(np.uint32), yb.view(np.uint32))
    print(f"first init {1e3 * (t1 - t0):.2f} ms, second (shared bank) {1e3 * (t2 - t1):.2f} ms")


@pytest.mark.parametrize("width", [32, 64])
def test_outputs_off_the_canonical_pattern_are_evaluated_inside_the_matrix_kernel(width):
    """44.1 -> 48 kHz with 300 nearest-filter phases: thousands of outputs land on a filter boundary and round to the other
    row than their slot's canonical one.  The matrix-core kernels evaluate those themselves at their exact position (no
    follow-up launch): they must be counted (resampleHipLastHandedBack) and the whole output must meet the parity bar."""
    import _oracle
    from _hip import tolerance_ok
    B = A.binding(width); O = _oracle.binding(width)
    dt = np.float32 if width == 32 else np.float64
    ch, T, F = 2, 380, 300
    h = B.Resampler(ch, T, F, 0.0, BH); o = O.OracleResampler(ch, T, F, 0.0, _oracle.BH | _oracle.PRECISE)
    h.set_kernel(2); h.advance(T / 2); o.advance(T / 2)
    rng = np.random.default_rng(3)
    for k in range(3):
        n = 60000 + 777 * k
        x = (rng.random((n, ch)) - 0.5).astype(dt)
        cap = int(n * 48000 / 44100 * 1.01) + T
        u, g, y = h.process(x, cap, 48000 / 44100); uo, go, yo = o.process(x, cap, 48000 / 44100)
        assert (u, g) == (uo, go) and h.last_kernel() == 2
        if width == 32:
            assert tolerance_ok(np.array(y), np.array(yo))[0]
        else:
            assert np.all(np.abs(np.array(y) - np.array(yo)) <= 2.0 ** -47 * np.maximum(1.0, np.abs(np.array(yo))))
    assert h.handed_back() > 1000
