"""GPU: the reference's OWN test program — compiled from its own artest.c against its own headers, linked
against libartamd.so instead of resampler.c/biquad.c/decimator.c (oracle/Makefile target `dropin`) — must
print the same statistics as when it is linked against the reference's own DSP sources.
The binary only exists where /root/reference was available at build time (it travels to the GPU box)."""
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
