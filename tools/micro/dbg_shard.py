import os, sys
os.environ["ARTAMD_SHARDS"] = "5"
os.environ["ARTAMD_ROWS_TRACE"] = "1"
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import audio_resampler_amd as A
from _hip import HipResampler
from _oracle import noise, BH, INTERP
MT = A.RESAMPLE_MULTITHREADED
ch, T, R = 12, 988, 48000 / 44100
sizes = [300, 9000, 30000, 140000]
x, _ = noise(sum(sizes) * ch, state=5); x = x.reshape(-1, ch)
plain = HipResampler(ch, T, T, 0.0, BH | INTERP); sharded = HipResampler(ch, T, T, 0.0, BH | INTERP | MT)
for r in (plain, sharded): r.advance(T / 2)
pos = 0
for n in sizes:
    cap = int(n * R) + 2000
    print("--- plain", n, file=sys.stderr); a = plain.process(x[pos:pos + n], cap, R)
    print("--- sharded", n, file=sys.stderr); b = sharded.process(x[pos:pos + n], cap, R)
    pos += n
    ya, yb = np.array(a[2]), np.array(b[2])
    d = (ya.view(np.uint32) != yb.view(np.uint32))
    print(n, "differing per channel:", d.sum(axis=0), "first rows:", np.nonzero(d.any(axis=1))[0][:5], file=sys.stderr)
