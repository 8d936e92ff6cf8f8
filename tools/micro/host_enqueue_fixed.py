"""host time per device-resident call of ART's form of a stream (fixed ratio, nearest filter): the enqueue loop alone, and enqueue + drain.  usage: host_enqueue_fixed.py CH TAPS BLOCK"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
ch, taps, block = [int(v) for v in sys.argv[1:4]]
rs = A.Resampler(ch, taps, taps, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE | A.INCLUDE_LOWPASS, fixed=(44100.0, 48000.0, 0)); rs.advance(taps / 2.0)
rs.set_stream(torch.cuda.current_stream().cuda_stream)
x, _ = noise(block * ch); d_in = torch.from_numpy(x.reshape(block, ch)).cuda(); cap = int((block + taps) * 48000 / 44100) + 64; d_out = torch.empty(cap, ch, device="cuda")
for _ in range(400): rs.process_device(d_in, block, d_out, cap, 0.0)
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter()
for _ in range(N): rs.process_device(d_in, block, d_out, cap, 0.0)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"ch {ch} T {taps} block {block} interp {rs.L.resampleInterpolationUsed(rs.p)} kernel {rs.last_kernel()}: enqueue {1e6 * (t1 - t0) / N:.1f} us/call, enqueue + drain {1e6 * (t2 - t0) / N:.1f} us/call")
