"""ctypes mirror of include/{resampler,biquad,decimator,art_hip}.h.

`lib()` returns the loaded libartamd.so with argtypes set (`wide().lib()`: libartamd64.so, 8-byte samples); the functions keep the reference's names
(resampleInit, resampleProcessInterleaved, biquad_apply_buffer, decimateProcessInterleavedLE, ...).
`Resampler` / `Decimator` / `BiquadBank` are thin object wrappers used by tests and bench.py.
There is no fallback: if the library is missing or no GPU is present the constructors raise.
"""
import ctypes as C
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# resampler.h flags
SUBSAMPLE_INTERPOLATE, BLACKMAN_HARRIS, INCLUDE_LOWPASS, RESAMPLE_MULTITHREADED, NO_FILTER_REDUCTION = 0x1, 0x2, 0x4, 0x8, 0x10
RESAMPLE_FIXED_RATIO, EXTRAPOLATE_ENDPOINTS, EXTRAPOLATE_PREFILL, EXTEND_CONVOLUTION_MATH = 0x20, 0x40, 0x80, 0x100
RESAMPLER_FLUSHED, RESAMPLER_SNAP_OFFSET, RESAMPLE_STRICT_ORDER = 0x200, 0x400, 0x10000
# decimator.h flags
DITHER_HIGHPASS, DITHER_FLAT, DITHER_LOWPASS = 0x1, 0x2, 0x4
SHAPING_1ST_ORDER, SHAPING_2ND_ORDER, SHAPING_3RD_ORDER, SHAPING_ATH_CURVE = 0x100, 0x200, 0x400, 0x800
DECIMATE_MULTITHREADED = 0x1000

def _bind(width):
    """everything below exists once per sample width: 32 (libartamd.so, float) and 64 (libartamd64.so, double —
    the reference's PATH_WIDTH=64 builds, reference resampler.h:22-26)"""
    smp_c = C.c_double if width == 64 else C.c_float            # artsample_t of this build
    smp_np = np.float64 if width == 64 else np.float32
    smp_torch = "float64" if width == 64 else "float32"         # torch dtype name for device tensors
    LIB_PATH = os.environ.get("ARTAMD_LIB64" if width == 64 else "ARTAMD_LIB") or \
        os.path.join(HERE, "libartamd64.so" if width == 64 else "libartamd.so")     # ARTAMD_LIB*: ablation builds only
    f32p = C.POINTER(smp_c)                                      # (name kept from the 4-byte build)
    u8p = C.POINTER(C.c_ubyte)


    class ResampleResult(C.Structure):
        _fields_ = [("input_used", C.c_uint), ("output_generated", C.c_uint)]


    class Resample(C.Structure):
        _fields_ = [("numChannels", C.c_int), ("numSamples", C.c_int), ("numFilters", C.c_int), ("numTaps", C.c_int),
                    ("inputIndex", C.c_int), ("flags", C.c_int), ("tempFilter", C.c_void_p),
                    ("outputOffset", C.c_double), ("fixedRatio", C.c_double), ("lowpassRatio", C.c_double),
                    ("subsample", C.c_void_p), ("buffers", C.c_void_p), ("filters", C.POINTER(f32p)), ("hip", C.c_void_p)]


    class BiquadCoefficients(C.Structure):
        _fields_ = [(n, smp_c) for n in ("a0", "a1", "a2", "a3", "a4", "b1", "b2", "b3", "b4")]


    class Biquad(C.Structure):
        _fields_ = [("a", smp_c * 5), ("b", smp_c * 5), ("x", smp_c * 4), ("y", smp_c * 4),
                    ("order", C.c_int), ("index", C.c_int)]


    class Decimate(C.Structure):
        _fields_ = [("numChannels", C.c_int), ("outputBits", C.c_int), ("outputBytes", C.c_int), ("dither_type", C.c_int),
                    ("flags", C.c_int), ("outputGain", C.c_double), ("feedback", f32p), ("tpdf_generators", C.POINTER(C.c_uint32)),
                    ("noise_shapers", C.POINTER(Biquad)), ("hip", C.c_void_p)]


    class ArtamdSegment(C.Structure):
        _fields_ = [("first_output", C.c_uint), ("lin_base", C.c_int), ("base_offset", C.c_double)]


    class ArtamdPosition(C.Structure):
        _fields_ = [("numTaps", C.c_int), ("numFilters", C.c_int), ("flags", C.c_int), ("inputIndex", C.c_int),
                    ("floorActive", C.c_int), ("outputOffset", C.c_double), ("fixedRatio", C.c_double)]


    RP, DP = C.POINTER(Resample), C.POINTER(Decimate)
    ptr = C.c_void_p      # device pointers travel as plain addresses

    # every symbol the public headers declare: name -> (restype, argtypes)
    EXPORTED_SYMBOLS = {
        # resampler.h
        "resampleInit": (RP, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]),
        "resampleFixedRatioInit": (RP, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]),
        "resampleProcess": (ResampleResult, [RP, C.POINTER(f32p), C.c_int, C.POINTER(f32p), C.c_int, C.c_double]),
        "resampleProcessInterleaved": (ResampleResult, [RP, f32p, C.c_int, f32p, C.c_int, C.c_double]),
        "resampleProcessAndFlush": (ResampleResult, [RP, C.POINTER(f32p), C.c_int, C.POINTER(f32p), C.c_int, C.c_double]),
        "resampleProcessAndFlushInterleaved": (ResampleResult, [RP, f32p, C.c_int, f32p, C.c_int, C.c_double]),
        "resampleGetRequiredSamples": (C.c_uint, [RP, C.c_int, C.c_double]),
        "resampleGetExpectedOutput": (C.c_uint, [RP, C.c_int, C.c_double]),
        "resampleAdvancePosition": (None, [RP, C.c_double]),
        "resampleGetLowpassRatio": (C.c_double, [RP]),
        "resampleGetPosition": (C.c_double, [RP]),
        "resampleGetNumFilters": (C.c_int, [RP]),
        "resampleInterpolationUsed": (C.c_int, [RP]),
        "resampleReset": (None, [RP]),
        "resampleFree": (None, [RP]),
        # biquad.h
        "biquad_init": (None, [C.POINTER(Biquad), C.POINTER(BiquadCoefficients), C.c_double]),
        "biquad_lowpass": (None, [C.POINTER(BiquadCoefficients), C.c_double]),
        "biquad_highpass": (None, [C.POINTER(BiquadCoefficients), C.c_double]),
        "biquad_apply_buffer": (None, [C.POINTER(Biquad), f32p, C.c_int, C.c_int]),
        "biquad_apply_sample": (smp_c, [C.POINTER(Biquad), smp_c]),
        # decimator.h
        "floatIntegersLE": (None, [u8p, C.c_double, C.c_int, C.c_int, C.c_int, f32p, C.c_int]),
        "decimateInit": (DP, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]),
        "decimateProcessLE": (C.c_int, [DP, C.POINTER(f32p), C.c_int, C.POINTER(u8p)]),
        "decimateProcessInterleavedLE": (C.c_int, [DP, f32p, C.c_int, u8p]),
        "decimateFree": (None, [DP]),
        # art_hip.h
        "artamdDeviceCount": (C.c_int, []),
        "artamdDeviceAlloc": (ptr, [C.c_size_t]),
        "artamdDeviceFree": (None, [ptr]),
        "artamdUpload": (C.c_int, [ptr, ptr, C.c_size_t, ptr]),
        "artamdDownload": (C.c_int, [ptr, ptr, C.c_size_t, ptr]),
        "artamdDeviceZero": (C.c_int, [ptr, C.c_size_t, ptr]),
        "artamdStreamSynchronize": (C.c_int, [ptr]),
        "artamdVersion": (C.c_char_p, []),
        "artamdSetDevices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
        "resampleHipGetDevice": (C.c_int, [RP]),
        "resampleHipNumShards": (C.c_int, [RP]),
        "resampleHipShardInfo": (C.c_int, [RP, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "resampleHipSetStream": (None, [RP, ptr]),
        "resampleHipSynchronize": (None, [RP]),
        "resampleHipSetKernel": (None, [RP, C.c_int]),
        "resampleHipKeepRows": (None, [RP, C.c_int]),
        "resampleHipSetCutInvariant": (None, [RP, C.c_int]),
        "resampleHipCutInvariantFallbacks": (C.c_uint, [RP]),
        "resampleHipLastKernel": (C.c_int, [RP]),
        "resampleHipLastHandedBack": (C.c_uint, [RP]),
        "resampleHipLastFixedPoint": (C.c_int, [RP, C.POINTER(C.c_double)]),
        "resampleHipLastFixedPointKernel": (C.c_int, [RP]),
        "resampleHipSetTiming": (None, [RP, C.c_int]),
        "resampleHipReadTiming": (C.c_double, [RP, C.POINTER(C.c_int)]),
        "resampleHipReadPrepTiming": (C.c_double, [RP]),
        "resampleProcessInterleavedDevice": (ResampleResult, [RP, ptr, C.c_int, ptr, C.c_int, C.c_double]),
        "resampleProcessAndFlushInterleavedDevice": (ResampleResult, [RP, ptr, C.c_int, ptr, C.c_int, C.c_double]),
        "resampleProcessPlanarDevice": (ResampleResult, [RP, ptr, C.c_long, C.c_int, ptr, C.c_long, C.c_int, C.c_double]),
        "artamdBuildFilterBank": (None, [C.c_int, C.c_int, C.c_double, C.c_int, f32p]),
        "artamdPlanCall": (C.c_int, [C.POINTER(ArtamdPosition), C.c_int, C.c_int, C.c_double, C.POINTER(ResampleResult),
                                     C.POINTER(ArtamdSegment), C.c_int, C.POINTER(C.c_int)]),
        "biquadBankCreate": (ptr, [C.POINTER(Biquad), C.c_int, C.c_int]),
        "biquadBankCreateMulti": (ptr, [C.POINTER(Biquad), C.c_int, C.c_int]),
        "biquadBankShardCount": (C.c_int, [ptr]),
        "biquadBankSetStream": (None, [ptr, ptr]),
        "biquadBankApplyInterleavedDevice": (None, [ptr, ptr, C.c_int]),
        "biquadBankRead": (None, [ptr, C.POINTER(Biquad)]),
        "biquadBankFree": (None, [ptr]),
        "biquadBankRepairs": (C.c_uint, [ptr]),
        "artamdBiquadRepairs": (C.c_uint, []),
        "decimateHipSetStream": (None, [DP, ptr]),
        "decimateProcessInterleavedLEDevice": (None, [DP, ptr, C.c_int, ptr]),
        "decimateHipClipped": (C.c_long, [DP]),
        "decimateHipShardCount": (C.c_int, [DP]),
        "artamdErrorCount": (C.c_int, []),
        "artamdPeriodMultiple": (C.c_int, [C.c_int]),
        "artamdPeriodMultipleRows": (C.c_int, [C.c_int, C.c_int]),
        "artamdLastError": (C.c_char_p, []),
        "floatIntegersLEDevice": (None, [ptr, C.c_double, C.c_int, C.c_int, C.c_int, ptr, C.c_int, ptr]),
        # stretch.h
        "stretchInit": (ptr, [C.c_int, C.c_int, C.c_int, C.c_int]),
        "stretchGetOutputCapacity": (C.c_int, [ptr, C.c_int, C.c_double]),
        "stretchProcess": (C.c_int, [ptr, f32p, C.c_int, f32p, C.c_double]),
        "stretchFlush": (C.c_int, [ptr, f32p]),
        "stretchReset": (None, [ptr]),
        "stretchFree": (None, [ptr]),
        "stretchHipSetStream": (None, [ptr, ptr]),
        "stretchProcessDevice": (C.c_int, [ptr, ptr, C.c_int, ptr, C.c_double]),
        "stretchFlushDevice": (C.c_int, [ptr, ptr]),
        "resampleProcessBatchInterleavedDevice": (C.c_int, [ptr, C.c_int, ptr, ptr, ptr, ptr, ptr, ptr]),
        "stretchProcessBatchDevice": (C.c_int, [ptr, C.c_int, ptr, ptr, ptr, ptr, ptr]),
        "stretchFlushBatchDevice": (C.c_int, [ptr, C.c_int, ptr, ptr]),
    }

    _state = {"lib": None}


    def load_library(path=LIB_PATH):
        if _state["lib"] is None:
            # torch bundles its own HIP runtime; whichever libamdhip64 is loaded first owns the process.
            # Import torch first so that device memory, streams and RCCL handed in from torch and the
            # kernels launched by libartamd.so live in ONE runtime (set ARTAMD_NO_TORCH=1 for torch-free use).
            if not os.environ.get("ARTAMD_NO_TORCH"):
                try:
                    import torch  # noqa: F401
                except Exception:
                    pass
            if not os.path.exists(path):
                raise RuntimeError(f"{path} is missing — build it with `python -m audio_resampler_amd.build` "
                                   "(there is no CPU fallback)")
            L = C.CDLL(path)
            for name, (res, args) in EXPORTED_SYMBOLS.items():
                fn = getattr(L, name)
                fn.restype, fn.argtypes = res, args
            _state["lib"] = L
        return _state["lib"]


    def lib():
        return load_library()


    def _np_f32(a):
        return np.ascontiguousarray(a, dtype=smp_np)


    def _dev_ptr(t):
        """address of a torch CUDA tensor (or a raw int)."""
        return t if isinstance(t, int) else (t.data_ptr() if t is not None else None)


    class Resampler:
        """Object wrapper over the C API.  Host arrays are numpy [frames, channels] (interleaved);
        device tensors are torch float32 CUDA tensors of the same shape."""

        def __init__(self, channels, taps, filters, lowpass_ratio=0.0, flags=BLACKMAN_HARRIS | SUBSAMPLE_INTERPOLATE, fixed=None):
            L = lib()
            if fixed is None:
                self.p = L.resampleInit(channels, taps, filters, lowpass_ratio, flags)
            else:
                src, dst, lowpass_freq = fixed
                self.p = L.resampleFixedRatioInit(channels, taps, filters, float(src), float(dst), int(lowpass_freq), flags)
            if not self.p:
                raise RuntimeError("resampleInit failed (bad parameters, or no MI355X visible — there is no CPU path)")
            self.channels = channels
            self.L = L

        def close(self):
            if getattr(self, "p", None):
                self.L.resampleFree(self.p)
                self.p = None

        __del__ = close

        @property
        def c(self):
            return self.p.contents

        def state(self):
            c = self.c
            return (np.float64(c.outputOffset).view(np.uint64).item(), c.inputIndex, c.flags & 0xffff)

        def bank(self):
            c = self.c
            return np.stack([np.ctypeslib.as_array(c.filters[i], shape=(c.numTaps,)).copy() for i in range(c.numFilters + 1)])

        def advance(self, delta):
            self.L.resampleAdvancePosition(self.p, delta)

        def reset(self):
            self.L.resampleReset(self.p)

        def position(self):
            return self.L.resampleGetPosition(self.p)

        def set_stream(self, stream_ptr):
            self.L.resampleHipSetStream(self.p, stream_ptr)

        def shards(self):
            """[(device, first_channel, channels), ...] of a RESAMPLE_MULTITHREADED context spread over devices; [] otherwise"""
            out = []
            for k in range(self.L.resampleHipNumShards(self.p)):
                d, f, n = C.c_int(), C.c_int(), C.c_int()
                self.L.resampleHipShardInfo(self.p, k, C.byref(d), C.byref(f), C.byref(n))
                out.append((d.value, f.value, n.value))
            return out

        def set_kernel(self, which):
            self.L.resampleHipSetKernel(self.p, which)

        def set_cut_invariant(self, on=True):
            """the cut-invariant stream policy (art_hip.h): one arithmetic per stream, the same bits for any cut of the input into calls"""
            self.L.resampleHipSetCutInvariant(self.p, 1 if on else 0)

        def cut_invariant_fallbacks(self):
            return self.L.resampleHipCutInvariantFallbacks(self.p)

        def keep_rows(self, on):
            self.L.resampleHipKeepRows(self.p, 1 if on else 0)

        def last_kernel(self):
            return self.L.resampleHipLastKernel(self.p)

        def synchronize(self):
            self.L.resampleHipSynchronize(self.p)

        def fixed_point(self):
            """(state, pairs): state 0 = the last call's FIR was not the fixed-point matrix kernel, 1 = it was, 2 = it stood down
            for the f32 kernel; pairs = digit-pair products per 32-tap chunk (9..13)"""
            pairs = C.c_double(0.0)
            state = self.L.resampleHipLastFixedPoint(self.p, C.byref(pairs))
            return state, pairs.value

        def fixed_point_kernel(self):
            """the form of the fixed-point kernel the last call's last launch was given to (art_hip.h), as its name; None: not fixed point"""
            return {1: "fir_i8_stream_kernel", 2: "fir_i8_dma_kernel", 3: "fir_i8_slab_kernel"}.get(self.L.resampleHipLastFixedPointKernel(self.p))

        def handed_back(self):
            return self.L.resampleHipLastHandedBack(self.p)

        def set_timing(self, on=True):
            self.L.resampleHipSetTiming(self.p, int(on))

        def read_timing(self):
            """(total FIR-kernel milliseconds, launches) since timing was enabled / last read"""
            n = C.c_int()
            ms = self.L.resampleHipReadTiming(self.p, C.byref(n))
            return ms, n.value

        def read_prep_timing(self):
            """milliseconds the launches of the last read_timing() spent before their dominant kernel (table / staging passes)"""
            return self.L.resampleHipReadPrepTiming(self.p)

        # -- host-pointer API (numpy) --
        def process(self, x, out_cap, ratio, flush=False, and_flush=False, threads=1):
            out = np.zeros((out_cap, self.channels), smp_np)
            op = out.ctypes.data_as(f32p)
            if flush:
                r = self.L.resampleProcessInterleaved(self.p, None, -1, op, out_cap, ratio)
            else:
                x = _np_f32(x)
                fn = self.L.resampleProcessAndFlushInterleaved if and_flush else self.L.resampleProcessInterleaved
                r = fn(self.p, x.ctypes.data_as(f32p), x.shape[0], op, out_cap, ratio)
            return r.input_used, r.output_generated, out[:r.output_generated]

        def process_planar(self, planes, out_cap, ratio, flush=False, and_flush=False):
            """planes: list of C float32 arrays [frames]; returns (used, generated, [C arrays])."""
            Cn = self.channels
            outs = [np.zeros(out_cap, smp_np) for _ in range(Cn)]
            op = (f32p * Cn)(*[o.ctypes.data_as(f32p) for o in outs])
            if flush:
                r = self.L.resampleProcess(self.p, None, -1, op, out_cap, ratio)
            else:
                planes = [_np_f32(p) for p in planes]
                ip = (f32p * Cn)(*[p.ctypes.data_as(f32p) for p in planes])
                fn = self.L.resampleProcessAndFlush if and_flush else self.L.resampleProcess
                r = fn(self.p, ip, len(planes[0]), op, out_cap, ratio)
            return r.input_used, r.output_generated, [o[:r.output_generated] for o in outs]

        # -- device-pointer API (torch tensors or raw addresses); asynchronous --
        def process_device(self, d_in, n_in, d_out, out_cap, ratio, and_flush=False):
            fn = self.L.resampleProcessAndFlushInterleavedDevice if and_flush else self.L.resampleProcessInterleavedDevice
            r = fn(self.p, _dev_ptr(d_in), n_in, _dev_ptr(d_out), out_cap, ratio)
            return r.input_used, r.output_generated

        def process_planar_device(self, d_in, in_pitch, n_in, d_out, out_pitch, out_cap, ratio):
            r = self.L.resampleProcessPlanarDevice(self.p, _dev_ptr(d_in), in_pitch, n_in, _dev_ptr(d_out), out_pitch, out_cap, ratio)
            return r.input_used, r.output_generated


    class Decimator:
        def __init__(self, channels, bits, nbytes, gain, rate, flags):
            self.L = lib()
            self.p = self.L.decimateInit(channels, bits, nbytes, gain, rate, flags)
            if not self.p:
                raise RuntimeError("decimateInit failed (bad parameters, or no MI355X visible — there is no CPU path)")
            self.channels, self.nbytes = channels, nbytes

        def close(self):
            if getattr(self, "p", None):
                self.L.decimateFree(self.p)
                self.p = None

        __del__ = close

        def process(self, x):
            """x float32 [frames, channels] -> (uint8 [frames*channels*nbytes], clipped)"""
            x = _np_f32(x)
            out = np.zeros(x.size * self.nbytes, np.uint8)
            clips = self.L.decimateProcessInterleavedLE(self.p, x.ctypes.data_as(f32p), x.shape[0], out.ctypes.data_as(u8p))
            return out, clips

        def process_planar(self, planes):
            Cn = self.channels
            planes = [_np_f32(p) for p in planes]
            n = len(planes[0])
            outs = [np.zeros(n * self.nbytes, np.uint8) for _ in range(Cn)]
            ip = (f32p * Cn)(*[p.ctypes.data_as(f32p) for p in planes])
            op = (u8p * Cn)(*[o.ctypes.data_as(u8p) for o in outs])
            clips = self.L.decimateProcessLE(self.p, ip, n, op)
            return outs, clips

        def process_device(self, d_in, frames, d_out):
            self.L.decimateProcessInterleavedLEDevice(self.p, _dev_ptr(d_in), frames, _dev_ptr(d_out))

        def clipped(self):
            return self.L.decimateHipClipped(self.p)

        def shards(self):
            return self.L.decimateHipShardCount(self.p)

        def set_stream(self, s):
            self.L.decimateHipSetStream(self.p, s)


    class BiquadBank:
        def __init__(self, sections, channels, nsections, multi=False):
            """sections: ctypes array (Biquad * (channels*nsections)), channel-major; multi: spread over the listed devices"""
            self.L = lib()
            self.p = (self.L.biquadBankCreateMulti if multi else self.L.biquadBankCreate)(sections, channels, nsections)
            if not self.p:
                raise RuntimeError("biquadBankCreate failed (no MI355X visible — there is no CPU path)")
            self.n = channels * nsections

        def close(self):
            if getattr(self, "p", None):
                self.L.biquadBankFree(self.p)
                self.p = None

        __del__ = close

        def apply_device(self, d_buf, frames):
            self.L.biquadBankApplyInterleavedDevice(self.p, _dev_ptr(d_buf), frames)

        def set_stream(self, s):
            self.L.biquadBankSetStream(self.p, s)

        def repairs(self):
            return self.L.biquadBankRepairs(self.p)

        def shards(self):
            return self.L.biquadBankShardCount(self.p)

        def read(self):
            out = (Biquad * self.n)()
            self.L.biquadBankRead(self.p, out)
            return out

    def process_batch_device(resamplers, d_ins, n_ins, d_outs, out_caps, ratios):
        """resampleProcessBatchInterleavedDevice over a list of Resampler objects: one call per context, one launch for those
        the general kernel runs.  Returns [(input_used, output_generated), ...] (raises if a launch failed)."""
        n = len(resamplers)
        ctx = (C.c_void_p * n)(*[C.cast(r.p, C.c_void_p) for r in resamplers])
        res = (ResampleResult * n)()
        rc = lib().resampleProcessBatchInterleavedDevice(
            ctx, n, (C.c_void_p * n)(*[_dev_ptr(d) for d in d_ins]), (C.c_int * n)(*[int(v) for v in n_ins]),
            (C.c_void_p * n)(*[_dev_ptr(d) for d in d_outs]), (C.c_int * n)(*[int(v) for v in out_caps]),
            (C.c_double * n)(*[float(v) for v in ratios]), res)
        if rc:
            raise RuntimeError("resampleProcessBatchInterleavedDevice failed")
        return [(r.input_used, r.output_generated) for r in res]

    return types.SimpleNamespace(**{k: v for k, v in locals().items() if not k.startswith("_") and k != "width"}, width=width)


_bound = {}


def binding(width=32):
    if width not in _bound:
        ns = _bind(width)
        # the flag values (resampler.h / decimator.h) are the same for both builds
        vars(ns).update({k: v for k, v in globals().items() if k.isupper() and isinstance(v, int)})
        _bound[width] = ns
    return _bound[width]


def wide():
    """the PATH_WIDTH=64 binding: same names (lib, Resampler, Decimator, BiquadBank, Biquad, ...), double samples"""
    return binding(64)


globals().update({k: v for k, v in vars(binding(32)).items() if k != "width"})      # module level = the 4-byte build
