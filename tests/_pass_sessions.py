"""Helper of test_gpu_pass_fixup.py: plays fixed sessions of the nearest-filter mode WITHOUT a low-pass (outputs that fall exactly on an
input sample are copies of it) in THIS process's environment and prints one sha256 per session, so that processes with different
ARTAMD_PASS_FIXUP_MIN can be compared bit for bit."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import audio_resampler_amd as A
from _hip import HipResampler
from _oracle import noise, BH

SESSIONS = [
    # (channels, taps, filters, src, dst, kernel preference, blocks)
    (2, 380, 160, 44100, 48000, 0, (600000, 150000, 9000)),          # B': f32 matrix kernel (streaming), then smaller calls
    (8, 988, 160, 44100, 48000, 0, (300000, 160000, 70000)),         # A': fixed point, slabs and 32-slot tiles
    (8, 988, 160, 44100, 48000, 7, (40000, 12000)),                  # fixed point forced on small launches
    (4, 380, 32, 44100, 48000, 0, (400000, 120000)),                 # F < P: several pass-through slots per period
    (1, 156, 160, 44100, 48000, 0, (700000, 20000)),                 # mono
    (16, 512, 147, 96000, 44100, 0, (200000, 50000)),                # down-sampling, 147 phases
    (8, 988, 2, 44100, 88200, 0, (200000,)),                         # 2 outputs per period: every other output is a copy
    (8, 988, 160, 44100, 48000, 8, (30000, 20000)),                  # the K-split kernel forced (few tiles)
    (4, 380, 32, 44100, 48000, 7, (250000, 120000)),                 # fixed point, three pass-through slots per period
    (2, 64, 1, 44100, 48000, 0, (300000,)),                          # ONE filter: every slot whose nearest filter is a whole sample — most of them
]


def main():
    out = []
    for ch, T, F, src, dst, kernel, blocks in SESSIONS:
        r = HipResampler(ch, T, F, 0.0, BH, kernel=kernel)
        r.advance(T / 2)
        ratio = dst / src
        x, _ = noise(sum(blocks) * ch, state=(ch * 1000 + T) | 1)
        x = x.reshape(-1, ch)
        h = hashlib.sha256(); pos = 0; made = 0; kinds = []
        for n in blocks:
            u, g, y = r.process(x[pos:pos + n], int(n * ratio) + 4000, ratio)
            assert u == n
            h.update(np.ascontiguousarray(y).tobytes()); pos += n; made += g
            kinds.append((int(r.last_kernel()), int(r.fixed_point()[0])))
        out.append({"session": [ch, T, F, src, dst, kernel], "frames": made, "kernels": kinds, "sha256": h.hexdigest()})
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
