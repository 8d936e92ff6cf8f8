import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
T=988; C=8
for block in (65536, 1<<20):
    r = A.Resampler(C, T, T, 0.0, A.BLACKMAN_HARRIS|A.SUBSAMPLE_INTERPOLATE); r.advance(T/2)
    x,_ = noise(block*C); d_in = torch.from_numpy(x.reshape(block,C)).cuda(); cap=int(block*1.09+600); d_out=torch.empty(cap,C,device='cuda')
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    for it in range(3):
        u,g = r.process_device(d_in, block, d_out, cap, 48000/44100)
        print(block, 'call', it, 'outputs', g, 'kernel', r.last_kernel(), 'handed back (last launch)', r.handed_back())
