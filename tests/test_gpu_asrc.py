"""Any-ratio streams (BASELINE.json configs[4]: stereo ASRC, nearest filter, a new ratio on every call) on the general kernel: its lean
tap loop (several steps' loads in flight, no address arithmetic or loop control between them) takes the same taps in the same order as
the plain loop — same bits, session by session (ARTAMD_GENERAL_LEAN=0 pins the plain loop) — and the stream is held to the oracle like
every other path."""
import json, math, os, subprocess, sys

import numpy as np
import pytest

from _hip import HipResampler, tolerance_ok
from _oracle import noise, OracleResampler, BH, INTERP, PRECISE

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _sessions(**env):
    r = subprocess.run([sys.executable, os.path.join(HERE, "_asrc_sessions.py")], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_lean_and_plain_tap_loops_leave_the_same_bits():
    a, b = _sessions(), _sessions(ARTAMD_GENERAL_LEAN="0")
    assert len(a) == len(b) >= 15
    for sa, sb in zip(a, b):
        assert sa["frames"] == sb["frames"] > 0
        assert sa["sha256"] == sb["sha256"], (sa, sb)


@pytest.mark.parametrize("interp", [False, True])
def test_asrc_blocks_against_the_oracle(interp):
    """configs[4]'s call shape: stereo, 380 x 380, 65,536-frame blocks, the ratio moved by up to 100 ppm on every block"""
    ch, T = 2, 380
    flags = BH | (INTERP if interp else 0)
    blocks = [65536, 65536, 20000, 65536]
    x, _ = noise(sum(blocks) * ch, state=777); x = x.reshape(-1, ch)
    r = HipResampler(ch, T, T, 0.0, flags); r.advance(T / 2)
    o = OracleResampler(ch, T, T, 0.0, flags | PRECISE); o.advance(T / 2)
    pos = 0
    for i, n in enumerate(blocks):
        ratio = 48000 / 44100 * (1 + 100e-6 * math.sin(2 * math.pi * i / 7 + 0.3))
        cap = int(n * ratio) + 4000
        u, g, y = r.process(x[pos:pos + n], cap, ratio)
        uo, go, yo = o.process(x[pos:pos + n], cap, ratio)
        assert (u, g) == (uo, go) and r.last_kernel() == 1
        ok, worst, rms = tolerance_ok(np.array(y), np.array(yo))
        assert ok, (i, worst, rms)
        pos += n
