"""where do two cuts of one fixed-ratio stream differ under the cut-invariant policy?  usage: cut_diff3.py CH TAPS SRC DST CUT"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import audio_resampler_amd as A
from _oracle import noise
ch, taps, src, dst, cut = [int(v) for v in sys.argv[1:6]]
total = 200000
x, _ = noise(total * ch, state=0xC077 | 1); x = x.reshape(total, ch)
def play(cuts):
    r = A.Resampler(ch, taps, taps, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE, fixed=(float(src), float(dst), 0)); r.advance(taps / 2)
    r.set_cut_invariant(True)
    outs, pos, starts = [], 0, []
    for n in cuts:
        u, g, y = r.process(x[pos:pos + n], int(n * dst / src) + 4000, 0.0)
        starts.append(sum(len(o) for o in outs)); outs.append(np.array(y).copy()); pos += n
    print("fallbacks", r.cut_invariant_fallbacks(), "filters", r.L.resampleGetNumFilters(r.p), "interp", r.L.resampleInterpolationUsed(r.p), file=sys.stderr)
    return np.concatenate(outs), starts
a, _ = play([total])
cuts = [cut] * (total // cut) + ([total % cut] if total % cut else [])
b, starts = play(cuts)
d = np.flatnonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1))
print("frames", a.shape[0], "differing frames", d.size, "first", d[:5], "last", d[-5:] if d.size else None)
print("call starts", starts)
if d.size:
    h, e = np.histogram(d, bins=40, range=(0, a.shape[0])); print("histogram over output frames (40 bins):", h.tolist())
    print("max |a-b|", float(np.abs(a - b).max()))
    per = 147 if (src, dst) == (96000, 44100) else 160
    print("differing slots mod period (first 40):", sorted(set((d % per).tolist()))[:40], "count distinct", len(set((d % per).tolist())))
