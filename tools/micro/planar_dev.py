"""Device-resident PLANAR calls (resampleProcessPlanarDevice) against interleaved ones of the same stream: 8 ch x 988 taps 44.1k -> 48k."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
ch, T = 8, 988
ratio = 48000 / 44100
for block in (65536, 262144, 1048576):
    for planar in (0, 1):
        rs = A.Resampler(ch, T, T, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE); rs.advance(T / 2.0); rs.set_stream(torch.cuda.current_stream().cuda_stream)
        x, _ = noise(block * ch)
        cap = int((block + T // 2) * ratio * 1.001 + 10)
        if planar:
            d_in = torch.from_numpy(np.ascontiguousarray(x.reshape(block, ch).T)).cuda(); d_out = torch.empty(ch, cap, device="cuda")
            step = lambda: rs.process_planar_device(d_in, block, block, d_out, cap, cap, ratio)
        else:
            d_in = torch.from_numpy(x.reshape(block, ch)).cuda(); d_out = torch.empty(cap, ch, device="cuda")
            step = lambda: rs.process_device(d_in, block, d_out, cap, ratio)
        for _ in range(10): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print(f"block {block:8d} {'planar     ' if planar else 'interleaved'}: {dt * 1e6:8.1f} us per call  kernel {rs.last_kernel()}", flush=True)
