"""audio_resampler_amd — MI355X-native windowed-sinc resampler, biquad and decimator.

The product is the C-ABI shared library `libartamd.so` (and `libartamd64.so`, the same tree compiled for
8-byte samples, reached through `wide()`): C host layer + hand-written gfx950 HIP
kernels, see csrc/ and ../include/).  This Python package is only plumbing: a ctypes binding that
mirrors the reference's C API name for name (`api.py`), helpers to hand torch device tensors to
the device-pointer entry points, and the build script.
"""
from .api import (  # noqa: F401
    lib, load_library, Resampler, Decimator, BiquadBank, ResampleResult, Resample, Decimate, Biquad, BiquadCoefficients,
    SUBSAMPLE_INTERPOLATE, BLACKMAN_HARRIS, INCLUDE_LOWPASS, RESAMPLE_MULTITHREADED, NO_FILTER_REDUCTION,
    RESAMPLE_FIXED_RATIO, EXTRAPOLATE_ENDPOINTS, EXTEND_CONVOLUTION_MATH, RESAMPLER_FLUSHED, RESAMPLER_SNAP_OFFSET,
    RESAMPLE_STRICT_ORDER, DITHER_HIGHPASS, DITHER_FLAT, DITHER_LOWPASS, SHAPING_1ST_ORDER, SHAPING_2ND_ORDER,
    SHAPING_3RD_ORDER, SHAPING_ATH_CURVE, DECIMATE_MULTITHREADED, EXPORTED_SYMBOLS, wide, binding,
)
