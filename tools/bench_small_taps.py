"""general vs matrix-core FIR kernel on small calls / small tap counts: kernel preference 1 = general, 2 = MFMA forced,
0 = the library's own choice (which should track the faster of the two)."""
import sys, math, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import audio_resampler_amd as A
WIDE = "--wide" in sys.argv
if WIDE:
    A = A.wide()            # the 8-byte sample build (libartamd64.so)
from audio_resampler_amd.synth import noise
stream = torch.cuda.current_stream().cuda_stream
def run(ch, taps, filters, block, kern, flags=A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE, steps=20):
    rs = A.Resampler(ch, taps, filters, 0.0, flags); rs.advance(taps / 2.0); rs.set_stream(stream)
    if kern: rs.set_kernel(kern)
    x, _ = noise(block * ch); d_in = torch.from_numpy(x.reshape(block, ch).astype(np.float64 if WIDE else np.float32)).cuda()
    ratio = 48000 / 44100
    cap = int(math.floor((block + taps // 2) * ratio * 1.001 + 10)); d_out = torch.empty(cap, ch, device="cuda", dtype=torch.float64 if WIDE else torch.float32)
    for _ in range(3): rs.process_device(d_in, block, d_out, cap, ratio)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for _ in range(steps):
        u, g = rs.process_device(d_in, block, d_out, cap, ratio); n += g * ch
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"ch {ch} T {taps} block {block} kernel pref {kern} used {rs.last_kernel()}: {n / dt / 1e6:9.1f} Msamples/s  {dt / steps * 1e3:.3f} ms/step", flush=True)
for (ch, T) in ((8, 988), (2, 380), (8, 48), (1, 48), (2, 156), (4, 380)):
    for block in (1024, 4096, 16384, 65536, 1 << 20):
        for kern in (1, 2, 0):
            run(ch, T, T, block, kern, steps=30)
