"""The serial-recurrence stages (biquad cascade, noise-shaped decimator) against the number of independent channels in
one call: one lane per channel, 64 channels per workgroup — a single 8-channel stream keeps one wave of one CU busy, a
batch of streams fills the chip.  Device-resident, one JSON line per point."""
import ctypes as C, json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import audio_resampler_amd as A

L = A.lib()
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, steps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for ch in (8, 64, 512, 4096, 16384):
    frames = max(4096, (1 << 25) // ch)                 # ~32M samples per call
    x = (torch.rand(frames, ch, device="cuda") - 0.5)
    co = A.BiquadCoefficients(); L.biquad_lowpass(C.byref(co), 0.2)
    secs = (A.Biquad * (ch * 2))()
    for i in range(ch * 2): L.biquad_init(C.byref(secs[i]), C.byref(co), 1.0)
    bank = A.BiquadBank(secs, ch, 2); bank.set_stream(stream)
    buf = x.clone()
    tb = timed(lambda: bank.apply_device(buf, frames))
    dec = A.Decimator(ch, 16, 2, 1.0, 44100, A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE); dec.set_stream(stream)
    pcm = torch.empty(frames * ch * 2, dtype=torch.uint8, device="cuda")
    td = timed(lambda: dec.process_device(x, frames, pcm))
    n = frames * ch
    print(json.dumps({"channels": ch, "frames": frames,
                      "biquad_2x_order2_Msamples_per_s": round(n / tb / 1e6, 1), "biquad_GBps": round(n * 8 / tb / 1e9, 1),
                      "decimate_16bit_ATH_Msamples_per_s": round(n / td / 1e6, 1), "decimate_GBps": round(n * 6 / td / 1e9, 1)}), flush=True)
