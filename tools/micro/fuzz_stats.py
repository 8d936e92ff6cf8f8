"""how often does a default-mode sample of the fuzz sessions leave the bar, at the stress level (peak 30 / unit 32: overshoots past 1.0)?  usage: fuzz_stats.py KERNEL N"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import test_gpu_fuzz as T
from _hip import HipResampler
from _oracle import OracleResampler, noise, PRECISE
kernel, N = int(sys.argv[1]), int(sys.argv[2])
level = 30.0; unit = 32.0
def scaled(count, state):
    x, st = noise(count, state=state); return (x * np.float32(2.0 * level)).astype(np.float32), st
bad = tot = 0; worst = 0.0; sess = []; rbad = 0; rworst = 0.0
for seed in range(N):
    s = T.random_session(5000 + seed)
    y, tr = T.play(HipResampler, s, noise_fn=scaled, kernel=kernel)
    yo, tro = T.play(OracleResampler, s, PRECISE, noise_fn=scaled)
    assert tr == tro
    e = np.abs(y.astype(np.float64) - yo.astype(np.float64)) / unit
    tol = 2.0 ** -23 * np.maximum(1.0, np.abs(yo.astype(np.float64) / unit))
    b = int((e > tol).sum()); bad += b; tot += e.size; worst = max(worst, float((e / tol).max()) if e.size else 0.0)
    yr, _ = T.play(OracleResampler, s, noise_fn=scaled)          # the reference's own float loop (source order)
    er = np.abs(yr.astype(np.float64) - yo.astype(np.float64)) / unit
    rbad += int((er > tol).sum()); rworst = max(rworst, float((er / tol).max()) if er.size else 0.0)
    if b: sess.append((seed, b, s[0]["args"][:3], s[0]["kw"].get("fixed")))
print(f"reference float loop (source order): {rbad} outside the bar ({rbad / max(tot, 1):.2e}), worst {rworst:.2f} x")
print(f"kernel {kernel}: {bad} of {tot} samples outside the bar ({bad / max(tot, 1):.2e}), worst {worst:.2f} x the bar; sessions: {sess[:12]}")
