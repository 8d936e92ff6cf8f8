"""Device-pointer API: general vs MFMA kernel by block size (headline config) — to place the dispatch threshold."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
ratio = 48000 / 44100
for (C, T) in ((8, 988), (2, 380), (2, 988)):
    for block in (1024, 4096, 16384, 65536, 262144):
        row = []
        for kern in (1, 2):
            rs = A.Resampler(C, T, T, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE); rs.advance(T / 2); rs.set_kernel(kern)
            rs.set_stream(torch.cuda.current_stream().cuda_stream)
            x, _ = noise(block * C); d_in = torch.from_numpy(x.reshape(block, C)).cuda()
            cap = int(math.floor((block + T // 2) * ratio + 10)); d_out = torch.empty(cap, C, device="cuda")
            for _ in range(3): rs.process_device(d_in, block, d_out, cap, ratio)
            torch.cuda.synchronize()
            n = max(10, min(300, int(3e7 / (block * C))))
            t0 = time.perf_counter()
            for _ in range(n):
                rs.process_device(d_in, block, d_out, cap, ratio); torch.cuda.synchronize()     # latency per call (sync each)
            dt = (time.perf_counter() - t0) / n
            row.append(dt * 1e6)
        print(f"C={C} T={T} block {block:7d}: general {row[0]:8.1f} us   mfma {row[1]:8.1f} us   -> {'mfma' if row[1] < row[0] else 'general'}", flush=True)
