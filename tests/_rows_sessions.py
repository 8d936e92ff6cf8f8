"""Helper of test_gpu_rows_cache.py: plays fixed streams through the fixed-point matrix path (kernel preference 7) in THIS process's
environment, saves every call's output to an .npz in the temporary directory and prints its path, so that processes with and without
ARTAMD_ROWS_CACHE=0 can be compared sample by sample."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from _hip import HipResampler
from _oracle import noise, BH, INTERP, LOWPASS

STREAMS = [
    (8, 988, 988, 44100, 48000, False, BH | INTERP, (300000, 150000, 150000, 150001, 70000)),
    (4, 988, 988, 44100, 48000, False, BH | INTERP, (400000, 123457, 400000)),
    (8, 988, 988, 96000, 44100, True, BH | INTERP | LOWPASS, (280000, 140000, 280000)),
    (4, 380, 32, 44100, 48000, False, BH, (250000, 120000, 250000)),
]
F32_STREAMS = [
    (2, 380, 380, 44100, 48000, False, BH | INTERP, (200000, 200000, 90000, 200001)),
    (8, 988, 988, 44100, 48000, False, BH | INTERP, (90000, 90000, 40000)),
    (16, 156, 156, 44100, 48000, False, BH | INTERP, (100000, 100000)),
]


def main():
    out = {}
    for si, (ch, T, F, src, dst, fixed, flags, blocks) in enumerate(STREAMS):
        r = HipResampler(ch, T, F, flags=flags, fixed=(float(src), float(dst), 0), kernel=7) if fixed else HipResampler(ch, T, F, 0.0, flags, kernel=7)
        r.advance(T / 2)
        ratio = dst / src
        x, _ = noise(sum(blocks) * ch, state=(ch * 1000 + T) | 1)
        x = x.reshape(-1, ch)
        pos = 0
        for ci, n in enumerate(blocks):
            u, g, y = r.process(x[pos:pos + n], int(n * ratio) + 4000, 0.0 if fixed else ratio)
            assert u == n and r.fixed_point()[0] == 1
            out[f"s{si}_call{ci}"] = np.array(y).copy(); pos += n
    for si, (ch, T, F, src, dst, fixed, flags, blocks) in enumerate(F32_STREAMS):
        r = HipResampler(ch, T, F, 0.0, flags, kernel=6)
        r.advance(T / 2)
        ratio = dst / src
        x, _ = noise(sum(blocks) * ch, state=(ch * 1000 + T + 7) | 1)
        x = x.reshape(-1, ch)
        pos = 0
        for ci, n in enumerate(blocks):
            u, g, y = r.process(x[pos:pos + n], int(n * ratio) + 4000, ratio)
            assert u == n and r.last_kernel() == 2 and r.fixed_point()[0] == 0
            out[f"f{si}_call{ci}"] = np.array(y).copy(); pos += n
    fd, path = tempfile.mkstemp(suffix=".npz"); os.close(fd)
    np.savez(path, **out)
    print(path)


if __name__ == "__main__":
    sys.exit(main())
