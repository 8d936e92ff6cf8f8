"""Adapters so the golden/artest harnesses can drive the HIP library like the oracle."""
import numpy as np

import audio_resampler_amd as A


class HipResampler(A.Resampler):
    """same constructor shape as _oracle.OracleResampler"""

    def __init__(self, channels, taps, filters, lowpass_ratio=0.0, flags=A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE,
                 fixed=None, extra=0, kernel=0, keep_rows=True):
        super().__init__(channels, taps, filters, lowpass_ratio, flags | extra, fixed)
        if kernel:
            self.set_kernel(kernel)
        if not keep_rows:          # (comparisons of kernel forms bit for bit: every launch on rows built from its own positions, include/art_hip.h)
            self.keep_rows(False)


def strict(**kw):
    return lambda *a, **k: HipResampler(*a, **{**k, **kw, "extra": A.RESAMPLE_STRICT_ORDER | kw.get("extra", 0)})


def tolerance_ok(y, truth):
    """SURVEY 7.3-1 / north_star: within one float32 ulp at full scale of the double-accumulate result."""
    y64, t64 = y.astype(np.float64), truth.astype(np.float64)
    tol = 2.0 ** -23 * np.maximum(1.0, np.abs(t64))
    return bool(np.all(np.abs(y64 - t64) <= tol)), float(np.abs(y64 - t64).max()), float(np.sqrt(np.mean((y64 - t64) ** 2)))
