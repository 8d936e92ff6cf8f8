import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
ch, taps = int(sys.argv[1]), int(sys.argv[2])
total = 600000
x, _ = noise(600000 * ch); d_in = torch.from_numpy(x.reshape(600000, ch)).cuda()
def run(cuts, trace_from=None):
    rs = A.Resampler(ch, taps, taps, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE, fixed=(44100, 48000, 0))
    rs.advance(taps / 2.0); rs.set_kernel(6)
    outs, pos, starts = [], 0, []
    for k, n in enumerate(cuts):
        cap = int(n * 48000 / 44100) + 4000
        d_out = torch.zeros(cap, ch, device="cuda")
        if trace_from is not None and k >= trace_from - 1 and k <= trace_from + 1: print(f"--- call {k}: {n} frames from {pos}", file=sys.stderr, flush=True)
        u, g = rs.process_device(d_in[pos:pos + n], n, d_out, cap, 0.0); pos += n
        torch.cuda.synchronize()
        starts.append(sum(len(o) for o in outs)); outs.append(d_out[:g].cpu().numpy().copy())
    return np.concatenate(outs), starts
rng = np.random.default_rng(7)
c = []
while sum(c) < total: c.append(int(min(rng.integers(3000, 120000), total - sum(c))))
a, _ = run([total])
if len(sys.argv) > 3:
    run(c, int(sys.argv[3])); sys.exit(0)
b, sb = run(c)
d = (b.view(np.uint32) != a.view(np.uint32)).any(axis=1)
idx = np.nonzero(d)[0]
print("cuts", c[:12], "...")
print("differing frames", len(idx), "first", idx[:5], "call starts", sb[:12])
k = int(np.searchsorted(sb, idx[0], side="right") - 1)
print("first differing call", k, "frames", c[k], "starts at output", sb[k], "per call:", [int(d[sb[j]:(sb[j + 1] if j + 1 < len(sb) else len(d))].sum()) for j in range(len(sb))])
