"""CPU, world_size 2 over gloo: the N>1 plumbing bench.py uses — channel slicing, the timing/consistency
reduction — plus the property that makes sharding legal: resampling a channel slice equals slicing the
resampled stream (checked with the oracle, which stands in for the per-rank GPU context here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audio_resampler_amd.shard import channel_slice, scatter_interleaved, agree_and_aggregate


def test_channel_slices_partition_the_stream():
    for total in (1, 2, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = channel_slice(total, world, r)
                assert 0 <= lo <= hi <= total
                cover += list(range(lo, hi))
            assert cover == list(range(total))
    assert [channel_slice(32, 8, r) for r in (0, 7)] == [(0, 4), (28, 32)]        # BASELINE.json configs[3]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, hip=False):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from _oracle import OracleResampler, noise, BH, INTERP
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total_ch, T = 6, 48
    x, _ = noise(total_ch * 3000)
    x = x.reshape(-1, total_ch)
    mine = scatter_interleaved(torch.from_numpy(x), world, rank).numpy()
    if hip:         # the PRODUCT as the per-rank context: strict order, so the comparison below stays bit for bit
        import audio_resampler_amd as A
        from _hip import HipResampler
        torch.cuda.set_device(rank % torch.cuda.device_count())
        r = HipResampler(mine.shape[1], T, 48, 0.0, BH | INTERP | A.RESAMPLE_STRICT_ORDER)
    else:
        r = OracleResampler(mine.shape[1], T, 48, 0.0, BH | INTERP)
    r.advance(T / 2)
    u, g, y = r.process(mine, 4000, 48000 / 44100, and_flush=True)
    agg = agree_and_aggregate(dist, "cpu", 0.5 + rank, g, mine.shape[1], kernel_ms=1.0 + rank, launches=3)
    # gather the slices back to compare with the un-sharded run on rank 0
    parts = [None] * world
    dist.all_gather_object(parts, y)
    if rank == 0:
        full = OracleResampler(total_ch, T, 48, 0.0, BH | INTERP)
        full.advance(T / 2)
        _, gf, yf = full.process(x, 4000, 48000 / 44100, and_flush=True)
        q.put((agg, gf, bool(np.array_equal(np.concatenate(parts, axis=1).view(np.uint32), yf.view(np.uint32)))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_over_gloo_with_hip_contexts():
    """the same world-2 run with the HIP library as every rank's context (ranks share the box's GPU(s)): the sharded product
    equals the un-sharded reference-order stream bit for bit"""
    test_two_ranks_over_gloo(hip=True)


def test_two_ranks_over_gloo(hip=False):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, hip)) for r in range(world)]
    for p in procs:
        p.start()
    agg, frames, same = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert agg["frames_consistent"] and agg["seconds_max"] == 1.5 and agg["kernel_ms_max"] == 2.0
    assert agg["samples_total"] == frames * 6
    assert same, "sharded channels must equal the un-sharded stream bit for bit"
