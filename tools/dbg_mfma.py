import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
from _hip import HipResampler
from _oracle import OracleResampler, noise, BH, INTERP, PRECISE
ch, T, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
interp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
flags = BH | (INTERP if interp else 0)
x, _ = noise(ch * 8000); x = x.reshape(-1, ch)
ratio = 48000/44100
g = HipResampler(ch, T, F, 0.0, flags, kernel=2); o = OracleResampler(ch, T, F, 0.0, flags | PRECISE)
g.advance(T/2); o.advance(T/2)
u, n, y = g.process(x, 9000, ratio); uo, no, yo = o.process(x, 9000, ratio)
print('counts', (u, n), (uo, no), 'kernel', g.last_kernel())
err = np.abs(y.astype(np.float64) - yo)
print('max err', err.max(), 'rms', np.sqrt((err**2).mean()), 'rms y', np.sqrt((yo.astype(np.float64)**2).mean()))
bad = err > 2e-7
print('bad frac', bad.mean())
slot = np.arange(n) % 160
print('bad by slot tile:', [float(bad[(slot//32)==t].mean()) for t in range(5)])
print('bad by slot (first 40):', [round(float(bad[slot==s].mean()),2) for s in range(40)])
print('bad by channel:', bad.mean(axis=0))
per = np.arange(n)//160
print('bad by period (first 20):', [round(float(bad[per==p].mean()),2) for p in range(20)])
i = np.argwhere(bad)[:6]
for a,b in i: print(a, b, y[a,b], yo[a,b])
