"""Nearest-filter mode without a low-pass: big launches run the matrix kernels' plain instantiations and overwrite the pass-through slots
in a pass of their own (artfir_pass_fixup); small ones substitute them in the kernels' epilogues (the PASS instantiations).  Same bits
either way, whichever ARTAMD_PASS_FIXUP_MIN (samples per launch) decides."""
import json, os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _sessions(**env):
    r = subprocess.run([sys.executable, os.path.join(HERE, "_pass_sessions.py")], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_pass_through_slots_by_epilogue_or_by_their_own_pass_are_the_same_bits():
    never, always, default = _sessions(ARTAMD_PASS_FIXUP_MIN="1000000000000"), _sessions(ARTAMD_PASS_FIXUP_MIN="0"), _sessions()
    assert len(never) == len(always) == len(default) >= 7
    for again in (_sessions(ARTAMD_PASS_FIXUP_MIN="0"), _sessions(ARTAMD_PASS_FIXUP_MIN="0", ARTAMD_I8_SLAB_MIN="1")):     # (the pass shares its work out by list index: every workgroup must build the same list — it did not at first, and one run in two showed it)
        assert [s["sha256"] for s in again] == [s["sha256"] for s in never]
    for a, b, c in zip(never, always, default):
        assert a["frames"] == b["frames"] == c["frames"] > 0
        assert a["kernels"] == b["kernels"] == c["kernels"], (a, b)
        assert any(k[0] == 2 for k in a["kernels"]), a                       # (the matrix path ran)
        assert a["sha256"] == b["sha256"] == c["sha256"], (a, b, c)
