"""PCM-level comparison of two WAV files written by ART-compatible tools (test infrastructure).

The last stage of the reference tool is its decimator (reference decimator.c:245-283): scale, subtract the shaped error,
add TPDF dither, round.  Two resamplers whose float outputs differ by dy give different integers where the rounding
boundary falls between them: with no noise shaping a differing sample is off by exactly one step and the rate is about
E|dy| / step; with noise shaping the first such flip changes the error fed back and the two quantisations, each a
valid noise-shaped rendering of (nearly) the same signal, run apart for good — what can be compared there is the
moment of the first flip and the error of each file against the un-quantised signal.
"""
import struct
import wave

import numpy as np


def write_float_wav(path, rate, x):
    """x: float32 [frames, channels] -> WAVE_FORMAT_IEEE_FLOAT file"""
    x = np.ascontiguousarray(x, "<f4")
    frames, ch = x.shape
    data = x.tobytes()
    fmt = struct.pack("<HHIIHH", 3, ch, rate, rate * ch * 4, ch * 4, 32)
    if ch > 2:      # extensible header with the "first ch speakers" mask (what art.c assumes for a plain header anyway)
        guid = bytes([3, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xAA, 0, 0x38, 0x9B, 0x71])
        fmt = struct.pack("<HHIIHH", 0xFFFE, ch, rate, rate * ch * 4, ch * 4, 32) + struct.pack("<HHI", 22, 32, (1 << ch) - 1 if ch <= 18 else 0xFFFFFFFF) + guid
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        f.write(b"data" + struct.pack("<I", len(data)) + data)


def read_wav(path):
    """-> (header bytes up to and including the data chunk's size field, samples [frames, channels]: int64 for PCM, float32 for float files)"""
    with open(path, "rb") as f:
        b = f.read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE", path
    pos, fmt = 12, None
    while pos + 8 <= len(b):
        cid, size = b[pos:pos + 4], struct.unpack_from("<I", b, pos + 4)[0]
        if cid == b"fmt ":
            tag, ch, rate, _, align, bits = struct.unpack_from("<HHIIHH", b, pos + 8)
            if tag == 0xFFFE:
                tag = struct.unpack_from("<H", b, pos + 8 + 24)[0]
            fmt = (tag, ch, align // ch, bits)
        elif cid == b"data":
            tag, ch, nbytes, bits = fmt
            raw = np.frombuffer(b, np.uint8, size, pos + 8)
            if tag == 3:
                x = raw.view("<f4").reshape(-1, ch)
            elif nbytes == 1:
                x = raw.astype(np.int64).reshape(-1, ch) - 128
            else:
                r = raw.reshape(-1, nbytes).astype(np.int64)
                v = sum(r[:, i] << (8 * i) for i in range(nbytes))
                v -= (v >> (8 * nbytes - 1)) << (8 * nbytes)
                x = (v >> (8 * nbytes - bits)).reshape(-1, ch)      # (left-justified when bits is not a multiple of 8)
            return b[:pos + 8], x
        pos += 8 + size + (size & 1)
    raise AssertionError(f"no data chunk in {path}")


def signal(rate, channels, seconds, seed=1):
    """programme-like test signal: per channel a few sines + a decaying noise burst pattern, peak ~0.7"""
    n = int(rate * seconds)
    rng = np.random.default_rng(seed)
    t = np.arange(n)[:, None] / rate
    f = 110.0 * (1 + np.arange(channels))[None, :] * np.array([1.0, 2.51, 5.03])[:, None, None]
    x = 0.25 * np.sin(2 * np.pi * f[0] * t) + 0.15 * np.sin(2 * np.pi * f[1] * t + 1.0) + 0.08 * np.sin(2 * np.pi * f[2] * t + 2.0)
    env = np.exp(-((np.arange(n) % (rate // 3)) / (0.05 * rate)))[:, None]
    x = x + 0.2 * env * rng.uniform(-1.0, 1.0, (n, channels))
    return x.astype(np.float32)


def compare_pcm(a, b):
    """two int sample arrays of equal shape -> dict of the statistics the tests assert on"""
    assert a.shape == b.shape, (a.shape, b.shape)
    d = (a - b).ravel()
    nz = np.flatnonzero(d)
    return {
        "samples": int(d.size),
        "differ": int(nz.size),
        "rate": float(nz.size) / max(1, d.size),
        "max_abs": int(np.abs(d).max()) if d.size else 0,
        "first": int(nz[0]) if nz.size else -1,
        "rms_steps": float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) if d.size else 0.0,
    }
