"""GPU (-m gpu): bench.py's multi-rank path, executed.  The driver runs `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N` on an 8-GPU node at round end; on the one-GPU box the same command line runs with two ranks sharing the device and the
few scalars of the reduction travelling over gloo (ARTAMD_BENCH_BACKEND=gloo, bench.py) — rendezvous, channel slices of ONE
stream, barrier + max-over-ranks timing, frame-count agreement and the JSON line are what is under test, not the numbers."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname() [1]; s.close()
    return p


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_runs_with_two_ranks(scaling):
    env = dict(os.environ, ARTAMD_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--block-frames", "262144", "--preroll-ms", "20", "--scaling", scaling, "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr [-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout [-2000:]                # rank 0 alone prints
    line = json.loads(lines [0])
    assert line ["n_gpus"] == 2 and line ["steps"] == 4 and line ["scaling"] == scaling
    cfg = line ["config"]
    if scaling == "strong":
        assert cfg ["stream_channels"] == 32 and cfg ["channels_per_gpu"] == 16
    else:
        assert cfg ["stream_channels"] == 16 and cfg ["channels_per_gpu"] == 8
    # whole-job throughput: both ranks' samples over the slower rank's time
    frames_per_step = 262144 * 48000 / 44100
    assert abs(line ["value"] * 1e6 * line ["ms_per_step"] * 1e-3 / (frames_per_step * cfg ["stream_channels"]) - 1.0) < 0.01
    assert line ["roofline"] ["launches"] == 4
    # BASELINE.json configs[3] rides along in the default (weak) command line: one 32-channel stream shared among the ranks
    if scaling == "weak":
        d = line ["config_d"]
        assert d ["scaling"] == "strong" and d ["stream_channels"] == 32 and d ["channels_per_gpu"] == 16 and d ["n_gpus"] == 2 and d ["frames_consistent"]
        assert abs(d ["value"] * 1e6 * d ["ms_per_step"] * 1e-3 / (frames_per_step * 32) - 1.0) < 0.01
    else:
        assert "config_d" not in line
