"""Channel sharding across the GPUs of one node.

Channels of a multichannel stream are independent (own history, own decimator/biquad state); the only
shared state is the scalar position, which evolves identically everywhere.  So rank r owns a contiguous
channel slice and runs its own context on its own GPU; no sample ever crosses a link.  The collective
below exists only to (a) agree on timing (barrier + max) and (b) cross-check that every rank generated
the same number of frames — it is a handful of scalars, latency-bound, on RCCL over xGMI (or gloo in
the CPU tests).
"""
import torch


def channel_slice(total_channels, world, rank):
    """contiguous, balanced: the first (total % world) ranks get one extra channel"""
    base, extra = divmod(total_channels, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def scatter_interleaved(x, world, rank):
    """x: [frames, channels] interleaved stream -> this rank's [frames, own_channels] contiguous slice"""
    lo, hi = channel_slice(x.shape[1], world, rank)
    return x[:, lo:hi].contiguous()


def agree_and_aggregate(dist, device, seconds, out_frames, channels, kernel_ms=0.0, launches=0):
    """Returns dict(seconds_max, samples_total, frames_consistent, kernel_ms_max, launches_max).
    `dist` is torch.distributed (initialised) or None for a single process."""
    mine = torch.tensor([seconds, float(out_frames) * channels, float(out_frames), kernel_ms, float(launches)],
                        dtype=torch.float64, device=device)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(seconds_max=seconds, samples_total=float(out_frames) * channels, frames_consistent=True,
                    kernel_ms_max=kernel_ms, launches_max=launches)
    mx, mn, sm = mine.clone(), mine.clone(), mine.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    return dict(seconds_max=mx[0].item(), samples_total=sm[1].item(), frames_consistent=bool(mx[2].item() == mn[2].item()),
                kernel_ms_max=mx[3].item(), launches_max=int(mx[4].item()))
