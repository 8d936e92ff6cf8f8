"""Wall-clock of the reference's OWN artest program: reference DSP sources (its Makefile flags; -m = its worker
threads) versus the same program linked against libartamd.so (host-pointer API: PCIe both ways + sync per call)."""
import os, re, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
R = os.path.join(ROOT, "oracle", "_ref")
CASES = ["-4 -c8 -n60 -s44100 -d48000", "-4 -c8 -n60 -s44100 -d48000 -b65536", "-4 -e -l -c8 -n60 -s96000 -d44100 -b65536 -o16",
         "-3 -c2 -n120 -s44100 -d48000 -b65536"]
for args in CASES:
    for exe, extra in (("artest_make", ""), ("artest_make", "-m"), ("artest_amd", "")):
        cmd = [os.path.join(R, exe)] + (extra.split() + args.split())
        t0 = time.perf_counter(); p = subprocess.run(cmd, capture_output=True, text=True); dt = time.perf_counter() - t0
        m = re.search(r"output \(-w2\): count =\s*(\d+)", p.stderr)
        ch = int(re.search(r"-c(\d+)", args).group(1))
        frames = int(m.group(1)) if m else 0
        print(f"{dt:7.2f} s  {frames * ch / dt / 1e6:9.1f} Msamples/s  {exe:12s} {extra:3s} {args}", flush=True)
