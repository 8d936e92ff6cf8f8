"""Device-resident throughput of one resampler shape, for A/B runs under environment switches.
Usage: python tools/bench_shapes.py CH TAPS FILTERS SRC DST FIXED(0/1) INTERP(0/1) BLOCK [KERNEL]"""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
ch, taps, filters, src, dst, fixed, interp, block = [int(v) for v in sys.argv[1:9]]
kernel = int(sys.argv[9]) if len(sys.argv) > 9 else 0
flags = A.BLACKMAN_HARRIS | (A.SUBSAMPLE_INTERPOLATE if interp else 0) | (A.INCLUDE_LOWPASS if fixed else 0)
rs = A.Resampler(ch, taps, filters, 0.0, flags, fixed=(src, dst, 0) if fixed else None)
rs.advance(taps / 2.0); rs.set_stream(torch.cuda.current_stream().cuda_stream)
if kernel: rs.set_kernel(kernel)
x, _ = noise(block * ch); d_in = torch.from_numpy(x.reshape(block, ch)).cuda()
ratio = dst / src; cap = int((block + taps // 2) * ratio * 1.001 + 10); d_out = torch.empty(cap, ch, device="cuda")
def step():
    u, g = rs.process_device(d_in, block, d_out, cap, 0.0 if fixed else ratio); return g * ch
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    for _ in range(8): step()
    torch.cuda.synchronize()
rs.set_timing(True)
n = 0; t0 = time.perf_counter()
for _ in range(50): n += step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ms, launches = rs.read_timing()
print(f"ch {ch} T {taps} F {rs.L.resampleGetNumFilters(rs.p)} {src}->{dst} interp {interp} block {block} kernel {rs.last_kernel()} pref {kernel} "
      f"tiles/wg {os.environ.get('ARTAMD_TILES_PER_WG', 'auto')}: {n / dt / 1e6:9.1f} Msamples/s  step {dt / 50 * 1e3:.4f} ms  fir kernel {ms / max(launches, 1):.4f} ms")
