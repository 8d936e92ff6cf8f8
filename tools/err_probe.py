"""worst error of the default-mode FIR against the double-accumulate oracle, in units of the parity bar (2^-23 max(1,|y|)),
on signals chosen to make partial sums large: full-scale DC, a full-scale low-frequency sine, full-scale noise, an impulse
train.  Usage: python tools/err_probe.py  (ARTAMD_LIB selects a build)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import audio_resampler_amd as A, _oracle
BH, IN, PRECISE = _oracle.BH, _oracle.INTERP, _oracle.PRECISE
n = 70000
t = np.arange(n)
rng = np.random.default_rng(5)
signals = {"dc": np.ones(n), "-dc": -np.ones(n), "sine_lo": np.sin(2 * np.pi * t / 3000.0), "sine_mid": np.sin(2 * np.pi * t / 37.3),
           "noise_fs": rng.choice([-1.0, 1.0], n) * rng.random(n) ** 0.2, "square": np.sign(np.sin(2 * np.pi * t / 500.0)), "impulses": (t % 1009 == 0) * 1.0}
worst_all = 0
for (ch, T, F, interp) in ((8, 988, 988, 1), (8, 988, 160, 0), (2, 380, 380, 1), (4, 156, 160, 0), (8, 48, 48, 1), (1, 1024, 1024, 1)):
    for name, sig in signals.items():
        x = np.repeat(sig[:, None], ch, axis=1).astype(np.float32)
        x[:, 1::2] *= -1 if ch > 1 else 1
        flags = BH | (IN if interp else 0)
        h = A.Resampler(ch, T, F, 0.0, flags); o = _oracle.OracleResampler(ch, T, F, 0.0, flags | PRECISE)
        K = int(os.environ.get("PROBE_KERNEL", "2")); h.set_kernel(K); h.advance(T / 2); o.advance(T / 2)
        cap = int(n * 48000 / 44100 * 1.01) + T
        u, g, y = h.process(x, cap, 48000 / 44100); uo, go, yo = o.process(x, cap, 48000 / 44100)
        assert (u, g) == (uo, go) and h.last_kernel() == K
        y64, t64 = np.array(y, np.float64), np.array(yo, np.float64)
        rf = _oracle.OracleResampler(ch, T, F, 0.0, flags); rf.advance(T / 2)            # the reference's own float arithmetic
        ref64 = np.array(rf.process(x, cap, 48000 / 44100)[2], np.float64)
        relref = np.abs(ref64 - t64) / (2.0 ** -23 * np.maximum(1.0, np.abs(t64)))
        rel = np.abs(y64 - t64) / (2.0 ** -23 * np.maximum(1.0, np.abs(t64)))
        worst_all = max(worst_all, rel.max())
        print(f"ch {ch} T {T} F {F} interp {interp} {name:9s}: worst {rel.max():.3f} of the bar, rms {np.sqrt(np.mean(rel**2)):.4f}   | reference float build: worst {relref.max():.3f}, rms {np.sqrt(np.mean(relref**2)):.4f}", flush=True)
print("worst overall", round(float(worst_all), 3))
