"""which launches of a fixed-point (big-call) downsampling stream run on the rows kept across calls?  ARTAMD_ROWS_TRACE=1 python tools/micro/rows_trace_down.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
for src, dst, block in ((96000, 44100, 1048576), (48000, 32000, 1048576), (44100, 48000, 1048576)):
    ch, taps = 8, 988
    rs = A.Resampler(ch, taps, taps, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE | A.INCLUDE_LOWPASS, fixed=(float(src), float(dst), 0)); rs.advance(taps / 2.0)
    rs.set_stream(torch.cuda.current_stream().cuda_stream)
    x, _ = noise(block * ch); d_in = torch.from_numpy(x.reshape(block, ch)).cuda(); cap = int((block + taps) * dst / src) + 64; d_out = torch.empty(cap, ch, device="cuda")
    print("==", src, dst, block, file=sys.stderr)
    for k in range(8): rs.process_device(d_in, block, d_out, cap, 0.0)
    torch.cuda.synchronize()
