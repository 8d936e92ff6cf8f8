"""GPU (-m gpu): a fixed-ratio stream (resampleFixedRatioInit, the reference's `-e` mode) on the f32 streaming kernel (kernel preference 6) gives
the same bits however its input is cut into calls — the reference's property (resampler.c:323-335, 529-535; SURVEY appendix C: one checksum at
every -b) on a matrix-core kernel, not only under RESAMPLE_STRICT_ORDER.  What carries it: the rows kept across calls are built once, for the
stream's canonical period, and every launch's tiles are anchored on that period — an output lands in the same tile row, walks the same K chunks
and is flushed at the same points whichever call brought it.  Holds where every launch is the streaming kernel's: calls of at least one period
of outputs (1,000 frames here), host-pointer calls of any length from there, device-pointer calls whose input is aligned to a frame of one or
two channels / to 16 bytes from four channels on (random cuts of a stereo stream start at odd frames of the caller's buffer).  (The library's own choice, preference 0, takes other kernels for other call sizes: within the
parity bar, not the same bits.)"""
import hashlib
import numpy as np
import pytest

import audio_resampler_amd as A
from _hip import HipResampler
from _oracle import noise, BH, INTERP

pytestmark = pytest.mark.gpu

TOTAL = 300000
STREAMS = [
    # (channels, taps, max filters, flags): fixed 44.1k -> 48k
    (2, 380, 380, BH | INTERP),        # resolves to 320 filters = 2 per phase: nearest filter, slots ON input samples in every period
    (8, 988, 988, BH | INTERP),        # 988 filters, 160 phases: interpolating rows
    (1, 380, 380, BH | INTERP),
    (4, 256, 256, BH | INTERP),
]


def _cuts(kind, rng):
    if kind == "one":
        return [TOTAL]
    if kind.isdigit():
        k = int(kind); c = [k] * (TOTAL // k)
        return c + ([TOTAL - sum(c)] if TOTAL - sum(c) else [])
    c = []
    while sum(c) < TOTAL:
        c.append(int(min(rng.integers(1000, 90000), TOTAL - sum(c))))
    if c[-1] < 1000 and len(c) > 1:
        c[-2] += c[-1]; c.pop()
    return c


def _play(stream, cuts, x, device, policy=False, rates=(44100.0, 48000.0), planar=False, flush=False):
    ch, T, F, flags = stream
    r = HipResampler(ch, T, F, flags=flags, fixed=(rates[0], rates[1], 0), kernel=0 if policy else 6); r.advance(T / 2)
    if policy:
        r.set_cut_invariant(True)      # (kernel preference stays the library's own: the POLICY pins the arithmetic per stream)
    outs, pos = [], 0
    if device:
        import torch
        d_x = torch.from_numpy(x).cuda()
    for i, n in enumerate(cuts):
        cap = int(n * rates[1] / rates[0]) + 4000
        last = flush and i == len(cuts) - 1
        if device:
            d_y = torch.zeros(cap, ch, device="cuda")
            u, g = r.process_device(d_x[pos:pos + n], n, d_y, cap, 0.0, and_flush=last)
            y = d_y[:g].cpu().numpy()
        elif planar:
            u, g, planes = r.process_planar([np.ascontiguousarray(x[pos:pos + n, c]) for c in range(ch)], cap, 0.0, and_flush=last)
            y = np.stack(planes, axis=1) if g else np.zeros((0, ch), np.float32)
        else:
            u, g, y = r.process(x[pos:pos + n], cap, 0.0, and_flush=last)
        assert u == n and (g == 0 or last or r.last_kernel() == 2), (i, n, u, g, r.last_kernel())
        outs.append(np.array(y).copy()); pos += n
    if policy:
        assert r.cut_invariant_fallbacks() == 0
    return np.concatenate(outs)


@pytest.mark.parametrize("stream", STREAMS, ids=[f"c{s[0]}_t{s[1]}" for s in STREAMS])
def test_fixed_ratio_output_does_not_depend_on_the_cuts(stream):
    ch = stream[0]
    x, _ = noise(TOTAL * ch, state=0xC075 | 1)
    x = x.reshape(TOTAL, ch)
    rng = np.random.default_rng(11)
    ref = _play(stream, [TOTAL], x, device=True)
    want = hashlib.sha256(ref.tobytes()).hexdigest()
    for kind, device in (("65536", True), ("16384", True), ("4096", True), ("1000", True), ("random", False), ("random", True), ("random", True), ("16384", False)):
        y = _play(stream, _cuts(kind, rng), x, device)
        assert y.shape == ref.shape, (kind, device, y.shape, ref.shape)
        assert hashlib.sha256(y.tobytes()).hexdigest() == want, (kind, device, int(np.count_nonzero(y.view(np.uint32) != ref.view(np.uint32))))


# ------------------------------------------------------------------------------------------------------------------------
# The cut-invariant STREAM POLICY (art_hip.h: resampleHipSetCutInvariant; round 6): kernel preference 0 — the library's own — and the policy on.
# Any cut: calls shorter than one period (160 outputs = 147 input frames) down to single frames, planar host calls, the flush in the last call; and
# a downsampling stream, whose input period (320 frames) is longer than T/2 + 64 — round 5's anchoring test sent such launches back to rows of their own.
# ------------------------------------------------------------------------------------------------------------------------
POLICY_STREAMS = [
    ((2, 380, 380, BH | INTERP), (44100.0, 48000.0)),
    ((8, 988, 988, BH | INTERP), (44100.0, 48000.0)),
    ((1, 380, 380, BH | INTERP), (44100.0, 48000.0)),
    ((4, 256, 256, BH | INTERP), (44100.0, 48000.0)),
    ((8, 988, 988, BH | INTERP), (96000.0, 44100.0)),      # BASELINE configs[2]'s conversion: 147 x 988, period_in 320
    ((2, 380, 380, BH | INTERP), (96000.0, 44100.0)),      # period_in 320 > T/2 + 64 = 254
    ((2, 156, 156, BH | INTERP), (48000.0, 32000.0)),      # 2 outputs per 3 inputs: the kernels take 16 periods at a time
    ((6, 380, 380, BH | INTERP), (44100.0, 48000.0)),      # not a compiled width: every launch runs as a 4-wide and a 2-wide group behind copies (fir_dispatch.hip)
]


def _tiny_cuts(rng, total):
    c = []
    while sum(c) < min(total, 30000):
        c.append(int(min(rng.integers(1, 400), total - sum(c))))
    while sum(c) < total:
        c.append(int(min(rng.integers(100, 20000), total - sum(c))))
    return c


@pytest.mark.parametrize("stream,rates", POLICY_STREAMS, ids=[f"c{s[0]}_t{s[1]}_{int(r[0])}_{int(r[1])}" for s, r in POLICY_STREAMS])
def test_cut_invariant_policy_any_cut_default_kernel_preference(stream, rates):
    ch = stream[0]
    total = 200000
    x, _ = noise(total * ch, state=0xC077 | 1)
    x = x.reshape(total, ch)
    rng = np.random.default_rng(5)
    ref = _play(stream, [total], x, device=True, policy=True, rates=rates, flush=True)
    want = hashlib.sha256(ref.tobytes()).hexdigest()
    fixed = lambda k: [k] * (total // k) + ([total - k * (total // k)] if total % k else [])
    for name, cuts, kw in (("65536", fixed(65536), dict(device=True)), ("16384", fixed(16384), dict(device=False)), ("4096", fixed(4096), dict(device=True)),
                           ("1000", fixed(1000), dict(device=False)), ("tiny", _tiny_cuts(rng, total), dict(device=False)), ("tiny-dev", _tiny_cuts(rng, total), dict(device=True)),
                           ("planar", fixed(16384), dict(device=False, planar=True)), ("random", _cuts("random", rng)[:1] + fixed(7777), dict(device=True))):
        cuts = [c for c in cuts if c > 0]
        if sum(cuts) > total:                      # (the random head may overshoot: trim the tail)
            over = sum(cuts) - total
            while over > 0:
                d = min(over, cuts[-1]); cuts[-1] -= d; over -= d
                if cuts[-1] == 0: cuts.pop()
        y = _play(stream, cuts, x, policy=True, rates=rates, flush=True, **kw)
        assert y.shape == ref.shape, (name, y.shape, ref.shape)
        assert hashlib.sha256(y.tobytes()).hexdigest() == want, (name, int(np.count_nonzero(y.view(np.uint32) != ref.view(np.uint32))))


def test_reference_artest_fixed_ratio_block_size_invariance_on_the_library(tmp_path):
    """SURVEY appendix C, with the reference's own test program (oracle/_ref/artest_amd = artest.c on this library): the `-e` output of ONE raw
    source stream (written once by `artest -w1`, read back with `-r -a`) has one checksum at every -b — under the policy (ARTAMD_KERNEL=9) in the
    DEFAULT numeric mode, as it has in the reference and under ARTAMD_STRICT=1."""
    import os, re, subprocess
    from _oracle import ORACLE_DIR
    exe, gen = os.path.join(ORACLE_DIR, "_ref", "artest_amd"), os.path.join(ORACLE_DIR, "_ref", "artest_strict")
    if not (os.path.exists(exe) and os.path.exists(gen)):
        pytest.skip("oracle/_ref/artest_* not built (needs /root/reference at build time)")
    for ch, rates, preset in ((2, "-s44100 -d48000", "-3"), (8, "-s96000 -d44100", "-4 -l")):
        raw = tmp_path / f"src{ch}.raw"
        with open(raw, "wb") as f:
            p = subprocess.run([gen] + f"-3 -c{ch} -n3 {rates} -a -w1".split(), stdout=f, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0 and raw.stat().st_size > 100000, p.stderr[-1500:]
        got = set()
        for b in (256, 1000, 4096, 65536):
            env = dict(os.environ, ARTAMD_KERNEL="9"); env.pop("ARTAMD_STRICT", None)
            with open(raw, "rb") as f:
                p = subprocess.run([exe] + f"{preset} -e -c{ch} {rates} -r -a -b{b}".split(), stdin=f, capture_output=True, text=True, env=env, timeout=600)
            assert p.returncode == 0, p.stderr[-1500:]
            m = re.search(r"output \(-w\d\): count =\s*(\d+), checksum = ([0-9a-f]{16})", p.stderr)
            assert m, p.stderr[-1500:]
            got.add((m.group(1), m.group(2)))
        assert len(got) == 1, (ch, rates, got)
