"""GPU: the host's failure path.  The reference's ABI has no error codes: a FIR launch that fails must leave the stream exactly where it was —
{0, 0} returned, position and history untouched (both are committed behind the call's LAST launch), the failure counted (artamdErrorCount /
artamdLastError, art_hip.h) — so that a caller who repeats the call gets what an undisturbed stream would have produced.
ARTAMD_TEST_FAIL_FIR=k (a test hook in arthip_fir) makes the k-th FIR launch of the process fail before anything is enqueued."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = r'''
import sys, json, hashlib, ctypes
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
import audio_resampler_amd as A
from _oracle import noise
ch, T, n = %(ch)d, %(T)d, %(n)d
x, _ = noise(3 * n * ch); x = x.reshape(-1, ch)
L = A.lib(); L.artamdLastError.restype = ctypes.c_char_p
r = A.Resampler(ch, T, T, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE); r.advance(T / 2)
log, outs = [], []
for k in range(3):
    seg = x[k * n:(k + 1) * n]
    before = r.state()
    u, g, y = r.process(seg, 2 * n, 48000 / 44100)
    if (u, g) == (0, 0):                       # failed: nothing moved — repeat the call
        assert tuple(r.state()) == tuple(before), (before, r.state())
        log.append(("failed", k, L.artamdErrorCount(), (L.artamdLastError() or b"").decode()))
        u, g, y = r.process(seg, 2 * n, 48000 / 44100)
    assert u == n, (k, u, g)
    outs.append(np.array(y).copy())
y = np.concatenate(outs)
print(json.dumps({"sha256": hashlib.sha256(y.tobytes()).hexdigest(), "frames": int(y.shape[0]), "errors": L.artamdErrorCount(), "log": log}))
'''


def _run(fail_at, ch, T, n):
    env = dict(os.environ)
    env.pop("ARTAMD_TEST_FAIL_FIR", None)
    if fail_at:
        env["ARTAMD_TEST_FAIL_FIR"] = str(fail_at)
    code = CHILD % dict(root=os.path.dirname(HERE), tests=HERE, ch=ch, T=T, n=n)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1]), p.stderr


@pytest.mark.parametrize("ch,T,n", [(2, 380, 5000), (8, 988, 40000), (8, 988, 300000)], ids=["general_kernel", "f32_matrix", "fixed_point"])
def test_a_failed_fir_launch_moves_nothing_and_is_counted(ch, T, n):
    clean, _ = _run(0, ch, T, n)
    assert clean["errors"] == 0 and clean["log"] == []
    for fail_at in (1, 2, 3):
        got, err = _run(fail_at, ch, T, n)
        assert got["errors"] == 1 and len(got["log"]) == 1 and got["log"][0][1] == fail_at - 1, got
        assert "FIR launch failed" in got["log"][0][3] and "FIR launch failed" in err
        # the repeated call continues the stream as if nothing had happened: the undisturbed run's bits
        assert got["frames"] == clean["frames"] and got["sha256"] == clean["sha256"], (fail_at, got, clean)
