"""Helper of test_gpu_slab_kernel.py: plays fixed sessions through the fixed-point matrix path (kernel preference 7) in THIS
process's environment and prints one sha256 per session (and which kernel form ran), so that processes with different
ARTAMD_I8_SLAB* switches can be compared bit for bit."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import audio_resampler_amd as A
from _hip import HipResampler
from _oracle import noise, BH, INTERP, LOWPASS

SESSIONS = [
    # (channels, taps, filters, src, dst, fixed-ratio form, flags, blocks)
    (8, 988, 988, 44100, 48000, False, BH | INTERP, (300000, 150000, 70000, 9000)),
    (4, 988, 988, 44100, 48000, False, BH | INTERP, (400000, 123457)),
    (16, 156, 156, 44100, 48000, False, BH | INTERP, (200000, 60000)),
    (32, 988, 988, 44100, 48000, False, BH | INTERP, (60000, 60000)),
    (8, 988, 988, 96000, 44100, True, BH | INTERP | LOWPASS, (280000, 140000)),
    (4, 380, 32, 44100, 48000, False, BH, (250000, 120000)),        # nearest filter, F < P: pass-through slots
    (8, 988, 988, 44100, 88200, False, BH | INTERP, (200000,)),      # 2 outputs per period: many periods at a time
    (8, 512, 512, 48000, 32000, False, BH | INTERP, (300000,)),
]


def main():
    out = []
    for ch, T, F, src, dst, fixed, flags, blocks in SESSIONS:
        r = HipResampler(ch, T, F, flags=flags, fixed=(float(src), float(dst), 0), kernel=7) if fixed else HipResampler(ch, T, F, 0.0, flags, kernel=7)
        r.advance(T / 2)
        ratio = dst / src
        x, _ = noise(sum(blocks) * ch, state=(ch * 1000 + T) | 1)
        x = x.reshape(-1, ch)
        h = hashlib.sha256(); pos = 0; made = 0; fixed_point = []
        for n in blocks:
            u, g, y = r.process(x[pos:pos + n], int(n * ratio) + 4000, 0.0 if fixed else ratio)
            assert u == n
            h.update(np.ascontiguousarray(y).tobytes()); pos += n; made += g
            fixed_point.append(int(r.fixed_point()[0]))
        out.append({"session": [ch, T, F, src, dst], "frames": made, "fixed_point": fixed_point, "sha256": h.hexdigest()})
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
