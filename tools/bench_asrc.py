"""Device-resident throughput of an any-ratio (ASRC) stream: 44.1k -> 48k x (1 +- 100 ppm), the ratio changed on every call.
Usage: python tools/bench_asrc.py CH TAPS FILTERS INTERP(0/1) BLOCK"""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
ch, taps, filters, interp, block = [int(v) for v in sys.argv[1:6]]
rs = A.Resampler(ch, taps, filters, 0.0, A.BLACKMAN_HARRIS | (A.SUBSAMPLE_INTERPOLATE if interp else 0))
rs.advance(taps / 2.0); rs.set_stream(torch.cuda.current_stream().cuda_stream)
x, _ = noise(block * ch); d_in = torch.from_numpy(x.reshape(block, ch)).cuda()
R = 48000 / 44100; cap = int((block + taps // 2) * R * 1.001 + 10); d_out = torch.empty(cap, ch, device="cuda")
k = [0]
def step():
    r = R * (1 + 100e-6 * math.sin(2 * math.pi * k[0] / 64)); k[0] += 1
    u, g = rs.process_device(d_in, block, d_out, cap, r); return g * ch
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    for _ in range(8): step()
    torch.cuda.synchronize()
rs.set_timing(True)
n = 0; t0 = time.perf_counter()
for _ in range(100): n += step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ms, launches = rs.read_timing()
print(f"ch {ch} T {taps} F {filters} interp {interp} block {block} kernel {rs.last_kernel()}: {n / dt / 1e6:9.1f} Msamples/s  step {dt / 100 * 1e3:.4f} ms  fir kernel {ms / max(launches, 1):.4f} ms")
