import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
ch, taps, cut, ncalls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
total = cut * ncalls
x, _ = noise(600000 * ch); d_in = torch.from_numpy(x.reshape(600000, ch)).cuda()
rs = A.Resampler(ch, taps, taps, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE, fixed=(44100, 48000, 0))
rs.advance(taps / 2.0); rs.set_kernel(6)
pos = 0
for k in range(ncalls):
    cap = int(cut * 48000 / 44100) + 4000
    d_out = torch.zeros(cap, ch, device="cuda")
    print(f"--- call {k}: {cut} frames from {pos}", file=sys.stderr, flush=True)
    u, g = rs.process_device(d_in[pos:pos + cut], cut, d_out, cap, 0.0); pos += cut
    torch.cuda.synchronize()
