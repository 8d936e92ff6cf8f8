"""one block size of tools/bench_host_api.py: python tools/micro/host_block.py BLOCK [channels taps]"""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
block = int(sys.argv[1]); C = int(sys.argv[2]) if len(sys.argv) > 2 else 8; T = int(sys.argv[3]) if len(sys.argv) > 3 else 988
ratio = 48000 / 44100
rs = A.Resampler(C, T, T, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE); rs.advance(T / 2)
x, _ = noise(block * C); x = x.reshape(block, C)
cap = int(math.floor((block + T // 2) * ratio + 10)); out = np.zeros((cap, C), np.float32)
xp, op = x.ctypes.data_as(A.api.f32p), out.ctypes.data_as(A.api.f32p)
L, p = rs.L, rs.p
for _ in range(5): L.resampleProcessInterleaved(p, xp, block, op, cap, ratio)
n = max(20, min(400, int(2e7 / (block * C))))
t0 = time.perf_counter()
for _ in range(n): L.resampleProcessInterleaved(p, xp, block, op, cap, ratio)
dt = time.perf_counter() - t0
print(f"block {block} ch {C} taps {T}: {dt / n * 1e6:.1f} us/call", flush=True)
