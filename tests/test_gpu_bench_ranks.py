"""GPU (-m gpu): bench.py's multi-rank path, executed.  The driver runs `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N` on an 8-GPU node at round end; on the one-GPU box the same command line runs with two ranks sharing the device and the
few scalars of the reduction travelling over gloo (ARTAMD_BENCH_BACKEND=gloo, bench.py) — rendezvous, channel slices of ONE
stream, barrier + max-over-ranks timing, frame-count agreement and the JSON line are what is under test, not the numbers."""
import json

import pytest

from _spawn import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_runs_with_two_ranks(scaling):
    out, failed = run_ranks(2, ["--gpus", "2", "--steps", "4", "--warmup", "1", "--block-frames", "262144", "--preroll-ms", "20",
                                "--scaling", scaling, "--no-cpu-baseline"], ARTAMD_BENCH_BACKEND="gloo")
    assert out.returncode == 0, "\n=====\n".join(failed)
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


def test_bench_runs_with_eight_ranks_as_the_driver_launches_it():
    """N = 8, the driver's scaling tier: BASELINE.json configs[3]'s partition — one 32-channel stream, 4 channels per rank (config_d in the default, weak
    line; the headline keeps 8 channels per rank = a 64-channel stream).  Eight ranks share the box's one device here: partition, rendezvous,
    barrier, agreement and the JSON line at N = 8 are under test, not the numbers."""
    out, failed = run_ranks(8, ["--gpus", "8", "--steps", "3", "--warmup", "1", "--block-frames", "131072", "--preroll-ms", "10",
                                "--no-cpu-baseline"], ARTAMD_BENCH_BACKEND="gloo")
    assert out.returncode == 0, "\n=====\n".join(failed)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout [-2000:]
    line = json.loads(lines [0])
    assert line ["n_gpus"] == 8 and line ["scaling"] == "weak"
    cfg = line ["config"]
    assert cfg ["stream_channels"] == 64 and cfg ["channels_per_gpu"] == 8
    frames_per_step = 131072 * 48000 / 44100
    assert abs(line ["value"] * 1e6 * line ["ms_per_step"] * 1e-3 / (frames_per_step * 64) - 1.0) < 0.01
    d = line ["config_d"]
    assert d ["scaling"] == "strong" and d ["stream_channels"] == 32 and d ["channels_per_gpu"] == 4 and d ["n_gpus"] == 8 and d ["frames_consistent"]
