"""Helper of test_gpu_general_pipe.py: plays fixed sessions through the GENERAL kernel (kernel preference 1) in THIS process's
environment and prints one sha256 per session, so that processes with and without ARTAMD_GENERAL_PIPE=0 can be compared bit for bit."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import audio_resampler_amd as A
from _hip import HipResampler
from _oracle import noise, BH, INTERP, LOWPASS, PRECISE

SESSIONS = [
    # (channels, taps, filters, ratio, flags, blocks)
    (8, 988, 988, 48000 / 44100, BH | INTERP, (30000, 4096, 1000, 17, 9000)),
    (4, 1024, 256, 44100 / 48000 * 1.0001, BH | INTERP, (20000, 5000)),
    (5, 988, 988, 1.37, BH | INTERP, (12000, 3000)),                       # five channels: a column group of eight, three idle
    (8, 988, 32, 2.0, BH, (9000, 2000)),                                   # nearest filter without a low-pass: pass-through outputs among the others
    (8, 600, 600, 0.731, BH, (16000, 800)),
    (16, 512, 512, 48000 / 44100, BH | INTERP | PRECISE, (6000, 2500)),    # double accumulators
    (32, 988, 988, 0.5, BH | INTERP, (5000, 1200)),
    (8, 1024, 64, 1.25, BH | INTERP, (8000, 3000)),                        # the longest filter: two full rounds of taps per output
]


def main():
    out = []
    for ch, T, F, ratio, flags, blocks in SESSIONS:
        r = HipResampler(ch, T, F, 0.0, flags, kernel=1)
        r.advance(T / 2)
        x, _ = noise(sum(blocks) * ch, state=(ch * 1000 + T) | 1)
        x = x.reshape(-1, ch)
        h = hashlib.sha256(); pos = 0; made = 0
        for n in blocks:
            u, g, y = r.process(x[pos:pos + n], int(n * ratio) + 4000, ratio)
            assert u == n and r.last_kernel() == 1
            h.update(np.ascontiguousarray(y).tobytes()); pos += n; made += g
        out.append({"session": [ch, T, F, ratio], "frames": made, "sha256": h.hexdigest()})
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
