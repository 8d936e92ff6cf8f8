"""The general kernel's pipelined tap loop (long filters, four channels or more: a round of taps loaded before the first multiply-add,
the next output's round under this output's reduction) takes the same taps in the same order as the plain loop: same bits, session by
session, whichever the host launches (ARTAMD_GENERAL_PIPE=0 pins the plain loop)."""
import json, os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _sessions(**env):
    r = subprocess.run([sys.executable, os.path.join(HERE, "_general_sessions.py")], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_pipelined_and_plain_tap_loops_leave_the_same_bits():
    a, b = _sessions(), _sessions(ARTAMD_GENERAL_PIPE="0")
    assert len(a) == len(b) >= 8
    for sa, sb in zip(a, b):
        assert sa["frames"] == sb["frames"] > 0
        assert sa["sha256"] == sb["sha256"], (sa, sb)
