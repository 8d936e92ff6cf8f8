import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
ch, taps = int(sys.argv[1]), int(sys.argv[2])
total = 200000
x, _ = noise(600000 * ch); d_in = torch.from_numpy(x.reshape(600000, ch)).cuda()
def run(cuts):
    rs = A.Resampler(ch, taps, taps, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE, fixed=(44100, 48000, 0))
    rs.advance(taps / 2.0); rs.set_kernel(6)
    outs, pos, starts = [], 0, []
    for n in cuts:
        cap = int(n * 48000 / 44100) + 4000
        d_out = torch.zeros(cap, ch, device="cuda")
        u, g = rs.process_device(d_in[pos:pos + n], n, d_out, cap, 0.0); pos += n
        starts.append(sum(len(o) for o in outs)); outs.append(d_out[:g].cpu().numpy().copy())
    return np.concatenate(outs), starts
a, sa = run([total]); b, sb = run([16384] * 12 + [total - 16384 * 12]); c, sc = run([65536] * 3 + [total - 65536 * 3])
for name, y, st in (("16384", b, sb), ("65536", c, sc)):
    d = (y.view(np.uint32) != a.view(np.uint32)).any(axis=1)
    idx = np.nonzero(d)[0]
    print(name, "differing frames", len(idx), "of", len(d), "first", idx[:10], "last", idx[-5:])
    print("  call starts", st[:6])
    h = np.bincount(idx % 320, minlength=320)
    print("  by slot of the period (n mod 320), top:", np.argsort(-h)[:12], h[np.argsort(-h)[:12]], "zero slots:", int((h == 0).sum()))
    print("  per call:", [int(d[st[k]:(st[k + 1] if k + 1 < len(st) else len(d))].sum()) for k in range(len(st))])
    ulp = np.abs(y.view(np.int32).astype(np.int64) - a.view(np.int32).astype(np.int64))
    print("  max ulp distance", int(ulp.max()))
