"""ctypes bindings for the CPU oracle (oracle/_build) and, where it exists, the real
reference built by oracle/Makefile (oracle/_ref).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess
import types
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# reference flag values (resampler.h:28-38, decimator.h:29-40)
INTERP, BH, LOWPASS, MT, NO_REDUCTION, FIXED, EXTRAP, PREFILL, PRECISE, FLUSHED, SNAP = (
    0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40, 0x80, 0x100, 0x200, 0x400)
DITHER_HP, DITHER_FLAT, DITHER_LP = 1, 2, 4
SHAPE_1, SHAPE_2, SHAPE_3, SHAPE_ATH = 0x100, 0x200, 0x400, 0x800
DEC_MT = 0x1000

NOISE_SEED = 0x3141592653589793

def _bind(width):
    """one set of bindings per sample width: 32 (float) and 64 (double; liboracle64_*.so / libartref64_*.so, the
    reference's PATH_WIDTH=64 builds)"""
    smp_c = C.c_double if width == 64 else C.c_float
    smp_np = np.float64 if width == 64 else np.float32
    suffix = "64" if width == 64 else ""
    f32p = C.POINTER(smp_c)          # (name kept from the 4-byte build)
    u8p = C.POINTER(C.c_ubyte)


    class Result(C.Structure):
        _fields_ = [("used", C.c_uint), ("generated", C.c_uint)]


    class OraResampler(C.Structure):
        _fields_ = [("channels", C.c_int), ("taps", C.c_int), ("filters", C.c_int), ("ring_len", C.c_int),
                    ("write_pos", C.c_int), ("flags", C.c_int),
                    ("read_pos", C.c_double), ("fixed_ratio", C.c_double), ("lowpass_ratio", C.c_double),
                    ("bank", f32p), ("ring", f32p), ("ring_store", f32p)]


    class BiquadCoeffs(C.Structure):
        _fields_ = [(n, smp_c) for n in ("a0", "a1", "a2", "a3", "a4", "b1", "b2", "b3", "b4")]


    class Biquad(C.Structure):
        _fields_ = [("a", smp_c * 5), ("b", smp_c * 5), ("x", smp_c * 4), ("y", smp_c * 4),
                    ("order", C.c_int), ("index", C.c_int)]


    def build_oracle():
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


    def _fptr(a):
        return a.ctypes.data_as(f32p)


    _oracle_cache = {}


    def load_oracle(kind="strict"):
        """kind: 'strict' (parity) or 'fast' (reference Makefile flags; CPU baseline port)."""
        if kind in _oracle_cache:
            return _oracle_cache[kind]
        path = os.path.join(ORACLE_DIR, "_build", f"liboracle{suffix}_{kind}.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        P = C.POINTER(OraResampler)
        L.ora_resample_init.restype = P
        L.ora_resample_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        L.ora_resample_fixed_init.restype = P
        L.ora_resample_fixed_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        L.ora_resample_free.argtypes = [P]
        L.ora_resample_reset.argtypes = [P]
        L.ora_resample_advance.argtypes = [P, C.c_double]
        L.ora_resample_position.restype = C.c_double
        L.ora_resample_position.argtypes = [P]
        L.ora_resample_required_input.restype = C.c_uint
        L.ora_resample_required_input.argtypes = [P, C.c_int, C.c_double]
        L.ora_resample_expected_output.restype = C.c_uint
        L.ora_resample_expected_output.argtypes = [P, C.c_int, C.c_double]
        for name in ("ora_resample_interleaved", "ora_resample_interleaved_flush"):
            fn = getattr(L, name)
            fn.restype = Result
            fn.argtypes = [P, f32p, C.c_int, f32p, C.c_int, C.c_double, C.c_int]
        L.ora_resample_planar.restype = Result
        L.ora_resample_planar.argtypes = [P, C.POINTER(f32p), C.c_int, C.POINTER(f32p), C.c_int, C.c_double, C.c_int]
        L.ora_dot_outside_in.restype = C.c_double
        L.ora_dot_outside_in.argtypes = [f32p, f32p, C.c_int]
        L.ora_dot_precise.restype = C.c_double
        L.ora_dot_precise.argtypes = [f32p, f32p, C.c_int]
        L.ora_biquad_lowpass.argtypes = [C.POINTER(BiquadCoeffs), C.c_double]
        L.ora_biquad_highpass.argtypes = [C.POINTER(BiquadCoeffs), C.c_double]
        L.ora_biquad_init.argtypes = [C.POINTER(Biquad), C.POINTER(BiquadCoeffs), C.c_double]
        L.ora_biquad_sample.restype = smp_c
        L.ora_biquad_sample.argtypes = [C.POINTER(Biquad), smp_c]
        L.ora_biquad_buffer.argtypes = [C.POINTER(Biquad), f32p, C.c_int, C.c_int]
        L.ora_decimate_init.restype = C.c_void_p
        L.ora_decimate_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        L.ora_decimate_free.argtypes = [C.c_void_p]
        L.ora_decimate_interleaved.restype = C.c_int
        L.ora_decimate_interleaved.argtypes = [C.c_void_p, f32p, C.c_int, u8p]
        L.ora_decimate_planar.restype = C.c_int
        L.ora_decimate_planar.argtypes = [C.c_void_p, C.POINTER(f32p), C.c_int, C.POINTER(u8p)]
        L.ora_float_integers_le.argtypes = [u8p, C.c_double, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
        L.ora_noise_fill.restype = C.c_uint64
        L.ora_noise_fill.argtypes = [f32p, C.c_long, C.c_uint64]
        L.ora_fade_in.argtypes = [f32p, C.c_int]
        L.ora_fade_out.argtypes = [f32p, C.c_int]
        L.ora_checksum_words.restype = C.c_uint64
        L.ora_checksum_words.argtypes = [C.c_uint64, C.c_void_p, C.c_long]
        L.ora_checksum_bytes.restype = C.c_uint64
        L.ora_checksum_bytes.argtypes = [C.c_uint64, u8p, C.c_long]
        _oracle_cache[kind] = L
        return L


    # ------------------------------------------------------------------------------------------
    # oracle convenience wrappers
    # ------------------------------------------------------------------------------------------

    class OracleResampler:
        def __init__(self, channels, taps, filters, lowpass_ratio=0.0, flags=BH | INTERP, fixed=None, kind="strict"):
            self.L = load_oracle(kind)
            if fixed is None:
                self.p = self.L.ora_resample_init(channels, taps, filters, lowpass_ratio, flags)
            else:
                src, dst, lpf = fixed
                self.p = self.L.ora_resample_fixed_init(channels, taps, filters, src, dst, lpf, flags)
            if not self.p:
                raise ValueError("oracle init failed")
            self.channels = channels

        def __del__(self):
            if getattr(self, "p", None):
                self.L.ora_resample_free(self.p)
                self.p = None

        @property
        def c(self):
            return self.p.contents

        def bank(self):
            c = self.c
            return np.ctypeslib.as_array(c.bank, shape=(c.filters + 1, c.taps)).copy()

        def state(self):
            c = self.c
            return (np.float64(c.read_pos).view(np.uint64).item(), c.write_pos, c.flags)

        def advance(self, d):
            self.L.ora_resample_advance(self.p, d)

        def reset(self):
            self.L.ora_resample_reset(self.p)

        def position(self):
            return self.L.ora_resample_position(self.p)

        def process(self, x, out_cap, ratio, flush=False, threads=1, and_flush=False):
            """x: float32 array [frames, channels] (interleaved) or None with flush=True."""
            out = np.zeros((out_cap, self.channels), smp_np)
            if flush:
                r = self.L.ora_resample_interleaved(self.p, None, -1, _fptr(out), out_cap, ratio, threads)
            else:
                x = np.ascontiguousarray(x, smp_np)
                fn = self.L.ora_resample_interleaved_flush if and_flush else self.L.ora_resample_interleaved
                r = fn(self.p, _fptr(x), x.shape[0], _fptr(out), out_cap, ratio, threads)
            return r.used, r.generated, out[:r.generated]


    def noise(count, state=NOISE_SEED, kind="strict"):
        """artest's white-noise generator; returns (float32[count], next_state)."""
        L = load_oracle(kind)
        a = np.empty(count, smp_np)
        s = L.ora_noise_fill(_fptr(a), count, state)
        return a, s


    def checksum_words(a, c=0):
        a = np.ascontiguousarray(a)
        return load_oracle().ora_checksum_words(c, a.ctypes.data, a.size)


    def checksum_bytes(a, c=0):
        a = np.ascontiguousarray(a, np.uint8)
        return load_oracle().ora_checksum_bytes(c, a.ctypes.data_as(u8p), a.size)


    # ------------------------------------------------------------------------------------------
    # the real reference (only where oracle/_ref exists)
    # ------------------------------------------------------------------------------------------

    class RefResample(C.Structure):
        """Prefix of the reference's `Resample` (resampler.h:44-48) — enough to read state and the bank."""
        _fields_ = [("numChannels", C.c_int), ("numSamples", C.c_int), ("numFilters", C.c_int), ("numTaps", C.c_int),
                    ("inputIndex", C.c_int), ("flags", C.c_int),
                    ("tempFilter", C.c_void_p), ("outputOffset", C.c_double), ("fixedRatio", C.c_double),
                    ("lowpassRatio", C.c_double), ("subsample", C.c_void_p),
                    ("buffers", C.POINTER(f32p)), ("filters", C.POINTER(f32p))]


    def ref_path(kind):
        return os.path.join(ORACLE_DIR, "_ref", f"libartref{suffix}_{kind}.so")


    def have_ref(kind="strict"):
        return os.path.exists(ref_path(kind))


    _ref_cache = {}


    def load_ref(kind="strict"):
        if kind in _ref_cache:
            return _ref_cache[kind]
        L = C.CDLL(ref_path(kind))
        P = C.POINTER(RefResample)
        L.resampleInit.restype = P
        L.resampleInit.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        L.resampleFixedRatioInit.restype = P
        L.resampleFixedRatioInit.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        for n in ("resampleProcessInterleaved", "resampleProcessAndFlushInterleaved"):
            fn = getattr(L, n)
            fn.restype = Result
            fn.argtypes = [P, f32p, C.c_int, f32p, C.c_int, C.c_double]
        for n in ("resampleProcess", "resampleProcessAndFlush"):
            fn = getattr(L, n)
            fn.restype = Result
            fn.argtypes = [P, C.POINTER(f32p), C.c_int, C.POINTER(f32p), C.c_int, C.c_double]
        L.resampleGetRequiredSamples.restype = C.c_uint
        L.resampleGetRequiredSamples.argtypes = [P, C.c_int, C.c_double]
        L.resampleGetExpectedOutput.restype = C.c_uint
        L.resampleGetExpectedOutput.argtypes = [P, C.c_int, C.c_double]
        L.resampleAdvancePosition.argtypes = [P, C.c_double]
        L.resampleGetPosition.restype = C.c_double
        L.resampleGetPosition.argtypes = [P]
        L.resampleGetLowpassRatio.restype = C.c_double
        L.resampleGetLowpassRatio.argtypes = [P]
        L.resampleGetNumFilters.argtypes = [P]
        L.resampleInterpolationUsed.argtypes = [P]
        L.resampleReset.argtypes = [P]
        L.resampleFree.argtypes = [P]
        L.biquad_lowpass.argtypes = [C.POINTER(BiquadCoeffs), C.c_double]
        L.biquad_highpass.argtypes = [C.POINTER(BiquadCoeffs), C.c_double]
        L.biquad_init.argtypes = [C.POINTER(Biquad), C.POINTER(BiquadCoeffs), C.c_double]
        L.biquad_apply_sample.restype = smp_c
        L.biquad_apply_sample.argtypes = [C.POINTER(Biquad), smp_c]
        L.biquad_apply_buffer.argtypes = [C.POINTER(Biquad), f32p, C.c_int, C.c_int]
        L.decimateInit.restype = C.c_void_p
        L.decimateInit.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        L.decimateFree.argtypes = [C.c_void_p]
        L.decimateProcessInterleavedLE.restype = C.c_int
        L.decimateProcessInterleavedLE.argtypes = [C.c_void_p, f32p, C.c_int, u8p]
        L.decimateProcessLE.restype = C.c_int
        L.decimateProcessLE.argtypes = [C.c_void_p, C.POINTER(f32p), C.c_int, C.POINTER(u8p)]
        L.floatIntegersLE.argtypes = [u8p, C.c_double, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
        _ref_cache[kind] = L
        return L


    class RefResampler:
        """Same surface as OracleResampler, backed by the real reference."""

        def __init__(self, channels, taps, filters, lowpass_ratio=0.0, flags=BH | INTERP, fixed=None, kind="strict"):
            self.L = load_ref(kind)
            if fixed is None:
                self.p = self.L.resampleInit(channels, taps, filters, lowpass_ratio, flags)
            else:
                src, dst, lpf = fixed
                self.p = self.L.resampleFixedRatioInit(channels, taps, filters, src, dst, lpf, flags)
            if not self.p:
                raise ValueError("reference init failed")
            self.channels = channels

        def __del__(self):
            if getattr(self, "p", None):
                self.L.resampleFree(self.p)
                self.p = None

        @property
        def c(self):
            return self.p.contents

        def bank(self):
            c = self.c
            return np.stack([np.ctypeslib.as_array(c.filters[i], shape=(c.numTaps,)).copy() for i in range(c.numFilters + 1)])

        def state(self):
            c = self.c
            return (np.float64(c.outputOffset).view(np.uint64).item(), c.inputIndex, c.flags)

        def advance(self, d):
            self.L.resampleAdvancePosition(self.p, d)

        def reset(self):
            self.L.resampleReset(self.p)

        def position(self):
            return self.L.resampleGetPosition(self.p)

        def process(self, x, out_cap, ratio, flush=False, threads=1, and_flush=False):
            out = np.zeros((out_cap, self.channels), smp_np)
            if flush:
                r = self.L.resampleProcessInterleaved(self.p, None, -1, _fptr(out), out_cap, ratio)
            else:
                x = np.ascontiguousarray(x, smp_np)
                fn = self.L.resampleProcessAndFlushInterleaved if and_flush else self.L.resampleProcessInterleaved
                r = fn(self.p, _fptr(x), x.shape[0], _fptr(out), out_cap, ratio)
            return r.used, r.generated, out[:r.generated]

    return types.SimpleNamespace(**{k: v for k, v in locals().items() if not k.startswith("_") and k != "width"}, width=width)


_bound = {}


def binding(width=32):
    if width not in _bound:
        _bound[width] = _bind(width)
    return _bound[width]


def wide():
    return binding(64)


globals().update({k: v for k, v in vars(binding(32)).items() if k != "width"})
