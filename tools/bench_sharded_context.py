"""ONE 32-channel context whose channels run as shards (RESAMPLE_MULTITHREADED; ARTAMD_SHARDS=8 on a one-GPU box puts all eight
shards on device 0) next to an ordinary 32-channel context: host-pointer and device-pointer calls, BASELINE configs[3] shape.
What this measures on one GPU is the cost of the sharding machinery (per-shard staging, streams, events), not a speed-up."""
import math, os, sys, time
os.environ.setdefault("ARTAMD_SHARDS", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
T, C, ratio = 988, 32, 48000 / 44100
flags = A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE
for block in (16384, 262144):
    x, _ = noise(block * C); x = x.reshape(block, C); cap = int((block + T // 2) * ratio + 10)
    out = np.zeros((cap, C), np.float32); d_in = torch.from_numpy(x).cuda(); d_out = torch.empty(cap, C, device="cuda")
    for name, fl in (("ordinary", flags), ("8 shards", flags | A.RESAMPLE_MULTITHREADED)):
        rs = A.Resampler(C, T, T, 0.0, fl); rs.advance(T / 2)
        xp, op = x.ctypes.data_as(A.api.f32p), out.ctypes.data_as(A.api.f32p)
        for _ in range(3): rs.L.resampleProcessInterleaved(rs.p, xp, block, op, cap, ratio)
        n = 20; t0 = time.perf_counter()
        for _ in range(n): rs.L.resampleProcessInterleaved(rs.p, xp, block, op, cap, ratio)
        host_us = (time.perf_counter() - t0) / n * 1e6
        for _ in range(3): rs.process_device(d_in, block, d_out, cap, ratio)
        rs.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): rs.process_device(d_in, block, d_out, cap, ratio)
        rs.synchronize(); torch.cuda.synchronize(); dev_us = (time.perf_counter() - t0) / n * 1e6
        print(f"32 ch x {block:7d} frames, {name:9s} ({len(rs.shards())} shards): host-pointer call {host_us:9.1f} us, device-pointer call {dev_us:9.1f} us "
              f"({block * ratio * C / dev_us:8.1f} Msamples/s)", flush=True)
