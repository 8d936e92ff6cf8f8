"""Bindings and drivers for the time-stretcher: oracle restatement (oracle/stretch_oracle.c), the real
reference (oracle/_ref, build container only) and the product (libartamd*.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

import _oracle

FAST, DUAL = 0x1, 0x2           # reference stretch.h:33-34


def signal(frames, channels, rate=44100, seed=1, dtype=np.float32):
    """pitched, slowly gliding tone + harmonics + a little noise, a silent gap and a burst of pure noise: exercises the
    period search, the silence shortcut and near-ties"""
    rng = np.random.default_rng(seed)
    t = np.arange(frames) / rate
    f0 = 110.0 * (1.0 + 0.3 * np.sin(2 * np.pi * 0.7 * t))
    ph = 2 * np.pi * np.cumsum(f0) / rate
    x = 0.5 * np.sin(ph) + 0.25 * np.sin(2 * ph + 0.3) + 0.12 * np.sin(3 * ph + 1.1) + 0.02 * rng.standard_normal(frames)
    x[frames // 3: frames // 3 + rate // 20] = 0.0                        # 50 ms of digital silence
    nb = frames // 2
    x[nb: nb + rate // 25] = 0.3 * rng.standard_normal(rate // 25)[: max(0, min(rate // 25, frames - nb))]
    cols = [x]
    if channels == 2:
        cols.append(0.8 * np.roll(x, 7) + 0.01 * rng.standard_normal(frames))
    return np.ascontiguousarray(np.stack(cols, axis=1), dtype=dtype)


class _Base:
    def run(self, x, blocks, ratios):
        """feed x in blocks (frames per call), ratio per call cycling through `ratios`; then drain.  Returns
        (output [frames, ch], per-call frame counts)."""
        ch = x.shape[1]
        cap = self.capacity(max(blocks), max(max(ratios), 1.0))
        out = np.zeros((cap, ch), x.dtype)
        ys, counts, pos, k = [], [], 0, 0
        while pos < x.shape[0]:
            n = min(blocks[k % len(blocks)], x.shape[0] - pos)
            g = self.feed(x[pos:pos + n], out, ratios[k % len(ratios)])
            ys.append(out[:g].copy()); counts.append(g); pos += n; k += 1
        for _ in range(4):
            g = self.drain(out)
            ys.append(out[:g].copy()); counts.append(g)
            if not g:
                break
        return np.concatenate(ys), counts


def _ptr(B, a):
    return a.ctypes.data_as(B.f32p)


class OracleStretch(_Base):
    def __init__(self, shortest, longest, channels, flags=0, width=32):
        self.B = _oracle.binding(width)
        L = self.L = self.B.load_oracle()
        L.ora_stretch_init.restype = C.c_void_p
        L.ora_stretch_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.ora_stretch_capacity.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.ora_stretch_feed.argtypes = [C.c_void_p, self.B.f32p, C.c_int, self.B.f32p, C.c_double]
        L.ora_stretch_drain.argtypes = [C.c_void_p, self.B.f32p]
        L.ora_stretch_reset.argtypes = [C.c_void_p]
        L.ora_stretch_free.argtypes = [C.c_void_p]
        self.p = L.ora_stretch_init(shortest, longest, channels, flags)
        if not self.p:
            raise ValueError("invalid periods")

    def capacity(self, n, r): return self.L.ora_stretch_capacity(self.p, n, r)
    def feed(self, x, out, r): return self.L.ora_stretch_feed(self.p, _ptr(self.B, x), x.shape[0], _ptr(self.B, out), r)
    def drain(self, out): return self.L.ora_stretch_drain(self.p, _ptr(self.B, out))
    def reset(self): self.L.ora_stretch_reset(self.p)

    def __del__(self):
        if getattr(self, "p", None):
            self.L.ora_stretch_free(self.p); self.p = None


class _CApi(_Base):
    """the reference's own API names (stretch.h:47-52): used for the real reference and for the product library"""

    def _bind(self, L, fp):
        L.stretchInit.restype = C.c_void_p
        L.stretchInit.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.stretchGetOutputCapacity.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.stretchProcess.argtypes = [C.c_void_p, fp, C.c_int, fp, C.c_double]
        L.stretchFlush.argtypes = [C.c_void_p, fp]
        L.stretchReset.argtypes = [C.c_void_p]
        L.stretchFree.argtypes = [C.c_void_p]

    def capacity(self, n, r): return self.L.stretchGetOutputCapacity(self.p, n, r)
    def feed(self, x, out, r): return self.L.stretchProcess(self.p, x.ctypes.data_as(self.fp), x.shape[0], out.ctypes.data_as(self.fp), r)
    def drain(self, out): return self.L.stretchFlush(self.p, out.ctypes.data_as(self.fp))
    def reset(self): self.L.stretchReset(self.p)

    def __del__(self):
        if getattr(self, "p", None):
            self.L.stretchFree(self.p); self.p = None


class RefStretch(_CApi):
    def __init__(self, shortest, longest, channels, flags=0, width=32):
        B = _oracle.binding(width)
        self.L, self.fp = B.load_ref("strict"), B.f32p
        self._bind(self.L, self.fp)
        self.p = self.L.stretchInit(shortest, longest, channels, flags)
        if not self.p:
            raise ValueError("invalid periods")


class HipStretch(_CApi):
    def __init__(self, shortest, longest, channels, flags=0, width=32):
        import audio_resampler_amd as A
        B = A.binding(width)
        self.L, self.fp = B.lib(), B.f32p
        self._bind(self.L, self.fp)
        self.p = self.L.stretchInit(shortest, longest, channels, flags)
        if not self.p:
            raise RuntimeError("stretchInit failed (bad periods, or no MI355X visible — there is no CPU path)")


# (name, rate, channels, flags, blocks, ratios, seconds)
CASES = [
    ("mono_slow", 44100, 1, 0, [16384], [1.25], 1.5),
    ("mono_fast", 44100, 1, 0, [16384], [0.8], 1.5),
    ("stereo_half", 44100, 2, 0, [16384], [0.5], 1.2),
    ("stereo_double", 44100, 2, 0, [16384], [2.0], 1.0),
    ("mono_blocks", 22050, 1, 0, [1000, 37, 5000, 1], [1.1, 0.9, 1.0, 1.37, 0.61], 1.5),
    ("stereo_unity_then_stretch", 32000, 2, 0, [4096], [1.0, 1.0, 1.0, 1.3, 1.0, 0.77], 1.5),
    ("mono_quick", 44100, 1, FAST, [16384], [1.4], 1.2),
    ("stereo_quick", 44100, 2, FAST, [8192], [0.7, 1.9], 1.2),
    ("mono_dual_slow", 44100, 1, DUAL, [16384], [3.1], 0.8),
    ("stereo_dual_fast", 44100, 2, DUAL, [16384], [0.3], 1.5),
    ("mono_dual_mid", 44100, 1, DUAL, [5000], [1.5, 2.6, 0.4], 1.0),
    ("mono_96k", 96000, 1, 0, [16384], [1.2], 0.5),
]


def case_setup(case, dtype=np.float32):
    name, rate, ch, flags, blocks, ratios, secs = case
    x = signal(int(rate * secs), ch, rate, seed=len(name), dtype=dtype)
    return x, (rate // 350, rate // 50, ch, flags), blocks, ratios
