"""Helper of test_gpu_asrc.py: plays fixed any-ratio sessions (the general kernel: BASELINE.json configs[4]'s stereo ASRC stream and its
neighbours — mono, short and long filters, wider streams below the pipelined loop's threshold) in THIS process's environment and prints one
sha256 per session, so that processes with and without ARTAMD_GENERAL_LEAN=0 can be compared bit for bit."""
import hashlib, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import audio_resampler_amd as A
from _hip import HipResampler
from _oracle import noise, BH, INTERP, LOWPASS, PRECISE

R = 48000 / 44100
SESSIONS = [
    # (channels, taps, filters, ratios per block, flags, blocks, kernel preference)
    (2, 380, 380, [R * (1 + 100e-6 * math.sin(2 * math.pi * i / 5 + 0.3)) for i in range(5)], BH, (65536, 30000, 65536, 1000, 40000), 0),    # BASELINE configs[4]
    (2, 380, 380, [R * 1.00003] * 3, BH | INTERP, (50000, 4096, 33000), 0),                 # interpolating: rows fi and fi + 1, the last of a cell from the next
    (1, 988, 988, [R * 0.99991] * 2, BH | INTERP, (70000, 25000), 0),                       # mono, preset -4
    (1, 988, 988, [1.3700013] * 2, BH, (60000, 20000), 0),
    (2, 156, 156, [0.731] * 2, BH | INTERP, (90000, 30000), 0),                             # down-sampling: long input span per output range
    (2, 380, 64, [2.0 * 1.000013] * 2, BH, (40000, 20000), 0),                              # nearest filter without a low-pass, near a 2x ratio: pass-through outputs among the others
    (2, 380, 380, [R * 1.00002] * 2, BH | INTERP | PRECISE, (40000, 20000), 0),             # double accumulators
    (2, 512, 512, [R * 1.0000001] * 2, BH | INTERP, (60000, 30000), 0),                       # all but a rational ratio
    (2, 380, 380, [R * 1.00004] * 2, BH | INTERP | LOWPASS, (40000, 18000), 0),
    (8, 380, 380, [R * 1.00002] * 2, BH | INTERP, (20000, 6000), 0),                        # wider streams below the pipelined loop's 512 taps
    (4, 256, 256, [0.9131] * 2, BH, (20000, 6000), 0),                                      # 16 lanes per output
    (1, 16, 16, [1.0713] * 2, BH | INTERP, (30000, 5000), 0),                               # a filter shorter than a lane group
    (2, 48, 48, [R * 1.0001] * 2, BH | INTERP, (50000, 7000), 0),
    (3, 988, 988, [1.0000317] * 2, BH | INTERP, (9000, 3000), 0),
    (16, 156, 156, [R * 0.9999] * 2, BH | INTERP, (12000, 3000), 0),
]


def main():
    out = []
    for ch, T, F, ratios, flags, blocks, pref in SESSIONS:
        r = HipResampler(ch, T, F, 0.0, flags, kernel=pref)
        r.advance(T / 2)
        x, _ = noise(sum(blocks) * ch, state=(ch * 1000 + T + F) | 1)
        x = x.reshape(-1, ch)
        h = hashlib.sha256(); pos = 0; made = 0
        for i, n in enumerate(blocks):
            ratio = ratios[i % len(ratios)]
            u, g, y = r.process(x[pos:pos + n], int(n * ratio) + 4000, ratio)
            assert u == n and r.last_kernel() == 1, (u, n, r.last_kernel())
            h.update(np.ascontiguousarray(y).tobytes()); pos += n; made += g
        out.append({"session": [ch, T, F, ratios[0]], "frames": made, "sha256": h.hexdigest()})
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
