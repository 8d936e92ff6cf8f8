"""Is a stream of device-resident calls bound by the host (enqueue time per call) or by the GPU?  Times the enqueue loop alone and the loop + drain.
Usage: python tools/micro/host_rate.py CH TAPS BLOCK [PREF] [TIMING 0/1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
ch, taps, block = [int(v) for v in sys.argv[1:4]]
pref = int(sys.argv[4]) if len(sys.argv) > 4 else 0
timing = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rs = A.Resampler(ch, taps, taps, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE)
rs.advance(taps / 2.0); rs.set_stream(torch.cuda.current_stream().cuda_stream)
if pref: rs.set_kernel(pref)
x, _ = noise(block * ch); d_in = torch.from_numpy(x.reshape(block, ch)).cuda()
ratio = 48000 / 44100; cap = int((block + taps // 2) * ratio * 1.001 + 10); d_out = torch.empty(cap, ch, device="cuda")
for _ in range(200): rs.process_device(d_in, block, d_out, cap, ratio)
torch.cuda.synchronize()
if timing: rs.set_timing(True)
N = 200
t0 = time.perf_counter()
for _ in range(N): rs.process_device(d_in, block, d_out, cap, ratio)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"ch {ch} T {taps} block {block} pref {pref} timing {timing}: enqueue {1e6 * (t1 - t0) / N:.1f} us/call, enqueue + drain {1e6 * (t2 - t0) / N:.1f} us/call, kernel {rs.last_kernel()}")
