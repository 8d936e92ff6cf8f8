"""Wall clock of the reference's OWN ART command-line tool on a WAV file: built from its own sources (oracle/_ref/art_strict,
-O2 -ffp-contract=off, one thread) versus art.c alone linked against libartamd.so (oracle/_ref/art_amd: host-pointer API,
PCIe both ways and a sync per call) and the device-resident tools/art_gpu.py.  Usage: python tools/art_timing.py [seconds]"""
import os, subprocess, sys, tempfile, time, wave
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
R = os.path.join(ROOT, "oracle", "_ref")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
long_secs = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
tmp = tempfile.mkdtemp()


def make(rate, ch, secs=secs):
    path = os.path.join(tmp, f"in_{rate}_{ch}.wav")
    n = int(rate * secs)
    rng = np.random.default_rng(1)
    t = np.arange(n)[:, None] / rate
    sig = 0.3 * rng.standard_normal((n, ch)) * 0.3 + 0.4 * np.sin(2 * np.pi * 440.0 * (1 + np.arange(ch))[None, :] * t)
    with wave.open(path, "wb") as w:
        w.setnchannels(ch); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(np.clip(np.round(sig * 32767), -32768, 32767).astype("<i2").tobytes())
    return path


for opts, rate, ch, dur in (("-4 -r48000", 44100, 2, secs), ("-3 -r44100 -p", 96000, 2, secs), ("-2 --tempo=1.25", 44100, 2, secs),
                            ("-4 -r48000", 44100, 8, long_secs), ("-4 -r44100 -p -o16", 96000, 8, long_secs)):
    src = make(rate, ch, dur)
    row = []
    for name, cmd in (("reference art", [os.path.join(R, "art_strict")]), ("art.c on libartamd", [os.path.join(R, "art_amd")]),
                      ("tools/art_gpu.py", [sys.executable, os.path.join(ROOT, "tools", "art_gpu.py")])):
        out = os.path.join(tmp, "out.wav")
        t0 = time.perf_counter()
        p = subprocess.run(cmd + opts.split() + ["-q", "-y", src, out], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        row.append(f"{name}: {dt:6.2f} s" + ("" if p.returncode == 0 else " (FAILED)"))
    print(f"{dur:.0f} s of {ch}-ch {rate} Hz audio, {opts:18s} | " + " | ".join(row), flush=True)
