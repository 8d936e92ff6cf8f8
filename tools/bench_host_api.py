"""Host-pointer API (resampleProcessInterleaved on numpy/pageable memory: H2D + kernels + D2H + sync per call),
the path ART/artest use.  Headline config, several block sizes."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
T, C, ratio = 988, 8, 48000 / 44100
for block in (4096, 16384, 65536, 262144, 1048576):
    rs = A.Resampler(C, T, T, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE); rs.advance(T / 2)
    x, _ = noise(block * C); x = x.reshape(block, C)
    cap = int(math.floor((block + T // 2) * ratio + 10))
    out = np.zeros((cap, C), np.float32)
    xp, op = x.ctypes.data_as(A.api.f32p), out.ctypes.data_as(A.api.f32p)
    L, p = rs.L, rs.p
    for _ in range(3): L.resampleProcessInterleaved(p, xp, block, op, cap, ratio)
    n = max(5, min(400, int(2e7 / (block * C))))
    t0 = time.perf_counter(); frames = 0
    for _ in range(n): frames += L.resampleProcessInterleaved(p, xp, block, op, cap, ratio).output_generated
    dt = time.perf_counter() - t0
    print(f"block {block:8d} frames: {dt / n * 1e6:9.1f} us/call  {frames * C / dt / 1e6:9.1f} Msamples/s  "
          f"({(block * C * 4 + frames / n * C * 4) / (dt / n) / 1e9:6.2f} GB/s over PCIe)", flush=True)
