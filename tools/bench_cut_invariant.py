"""What the cut-invariant stream policy (art_hip.h: resampleHipSetCutInvariant) costs against the library's own per-call choice: device-resident calls of a
fixed-ratio stream (resampleFixedRatioInit), microseconds per call (enqueue + drain over 200 calls, no events), preference 0 | policy | general kernel only
(preference 1: the other cut-invariant arithmetic).  Usage: python tools/bench_cut_invariant.py  -> profiles/r6_cut_invariant.txt"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise

STREAMS = [(8, 988, 44100.0, 48000.0), (2, 380, 44100.0, 48000.0), (8, 988, 96000.0, 44100.0), (32, 988, 44100.0, 48000.0)]
BLOCKS = [100, 1000, 4096, 16384, 65536, 262144, 1048576]


def run(ch, taps, src, dst, block, mode):
    rs = A.Resampler(ch, taps, taps, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE, fixed=(src, dst, 0))
    rs.advance(taps / 2.0); rs.set_stream(torch.cuda.current_stream().cuda_stream)
    if mode == "policy": rs.set_cut_invariant(True)
    elif mode == "general": rs.set_kernel(1)
    x, _ = noise(block * ch); d_in = torch.from_numpy(x.reshape(block, ch)).cuda()
    cap = int((block + taps) * dst / src + 64); d_out = torch.empty(cap, ch, device="cuda")
    n = 200 if block <= 65536 else 40
    for _ in range(max(20, n // 2)): rs.process_device(d_in, block, d_out, cap, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): rs.process_device(d_in, block, d_out, cap, 0.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fb = rs.cut_invariant_fallbacks() if mode == "policy" else 0
    return dt * 1e6, rs.last_kernel(), fb


for ch, taps, src, dst in STREAMS:
    print(f"{ch} ch x {taps} taps, {int(src)} -> {int(dst)} (fixed ratio): us per device-resident call   library's choice | policy | general kernel only")
    for block in BLOCKS:
        if block * ch > 1 << 24: continue
        a, ka, _ = run(ch, taps, src, dst, block, "auto")
        b, kb, fb = run(ch, taps, src, dst, block, "policy")
        c, kc, _ = run(ch, taps, src, dst, block, "general") if block * ch <= 1 << 22 else (float("nan"), 1, 0)
        print(f"   {block:8d} frames: {a:8.1f} (kernel {ka}) | {b:8.1f} ({b / a:4.2f} x, fallbacks {fb}) | {c:8.1f} ({c / a:4.2f} x)")
