import sys, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np
import test_gpu_fuzz as T
from _hip import HipResampler
from _oracle import OracleResampler, noise, PRECISE
seed, kernel = 17, 2
level = T.LEVELS[seed % 6]; unit = 2.0 ** np.ceil(np.log2(level))
def scaled(count, state):
    x, st = noise(count, state=state); return (x * np.float32(2.0 * level)).astype(np.float32), st
s = T.random_session(1000 + seed)
print("ctor", s[0], "adv", s[1], "calls", s[2])
y, tr = T.play(HipResampler, s, noise_fn=scaled, kernel=kernel)
yo, tro = T.play(OracleResampler, s, PRECISE, noise_fn=scaled)
e = np.abs(y.astype(np.float64) - yo.astype(np.float64)) / unit
tol = 2.0 ** -23 * np.maximum(1.0, np.abs(yo.astype(np.float64) / unit))
bad = np.argwhere(e > tol)
print("bad", bad[:10], "count", len(bad), "of", e.size)
for i, c in bad[:5]:
    print("frame", i, "ch", c, "y/unit", yo[i, c] / unit, "err", e[i, c], "err/ulp", e[i, c] / (np.spacing(np.float32(abs(yo[i, c]))) / unit))
print("rms", np.sqrt(np.mean(e ** 2)), "trace", tr)
