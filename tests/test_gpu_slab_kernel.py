"""GPU (-m gpu): fir_i8_slab_kernel — the fixed-point kernel's form for big launches (tiles of 64 slots x 256 columns, one
workgroup per CU, K walked 64 taps per barrier, tiles cut between workgroups with exact 64-bit partial sums).  Every
fixed-point kernel sums integers exactly and rounds once, the same way: whichever of them runs a launch, the bits are the same.
The switches are read once per process, so each variant runs in a process of its own:
  default                 slabs from 8 tiles per XCD on, the 32-slot kernels below;
  ARTAMD_I8_SLAB_MIN=1    slabs wherever they can run (small launches: most tiles cut into several parts, idle workgroups);
  ARTAMD_I8_SLAB=0        no slabs, periods taken to fill 32-slot tiles (other rows: the parity bar, not the same bits)."""
import json, os, subprocess, sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _sessions(**env):
    r = subprocess.run([sys.executable, os.path.join(HERE, "_slab_sessions.py")], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_slabs_and_32_slot_tiles_leave_the_same_bits():
    a, b = _sessions(), _sessions(ARTAMD_I8_SLAB_MIN="1")
    c = _sessions(ARTAMD_I8_SLAB_MIN="1000000")               # (same period rule, never a slab)
    assert len(a) == len(b) == len(c) >= 8
    for sa, sb, sc in zip(a, b, c):
        assert all(sa["fixed_point"]) and all(sb["fixed_point"]) and all(sc["fixed_point"]), (sa, sb, sc)
        assert sa["frames"] == sb["frames"] == sc["frames"]
        assert sa["sha256"] == sb["sha256"] == sc["sha256"], (sa, sb, sc)


def test_the_same_twice():
    """parts arrive in whatever order the hardware runs them: integer sums do not care"""
    a, b = _sessions(ARTAMD_I8_SLAB_MIN="1"), _sessions(ARTAMD_I8_SLAB_MIN="1")
    assert [s["sha256"] for s in a] == [s["sha256"] for s in b]


@pytest.mark.parametrize("env", [{"ARTAMD_I8_SLAB_MIN": "1"}, {"ARTAMD_I8_SLAB": "0"}], ids=["slabs_everywhere", "no_slabs"])
def test_fixed_point_files_under_the_switch(env):
    """every test of test_gpu_fixed_point.py and test_gpu_short_periods.py (oracle bars, scale-free behaviour, stand-by, shards)
    in a process with the switch set"""
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(HERE, "test_gpu_fixed_point.py"), os.path.join(HERE, "test_gpu_short_periods.py"),
                        "-k", "not the_rule"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
