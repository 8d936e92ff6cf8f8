"""The reference's own CPU path, timed beside the GPU figures (tools/bench_configs.py, bench.py's cpu_baseline): the REAL
reference built by oracle/Makefile from the sources where they lie (oracle/_ref/libartref_make.so: the reference Makefile's flags),
driven the way the reference drives itself:
  * resampler: RESAMPLE_MULTITHREADED (workers.c:249-371 — one job per channel, the last on the caller, joined every call;
    resampler.c:447-464) with 65,536-frame blocks, its best setting;
  * decimator: planar entry point + DECIMATE_MULTITHREADED (decimator.c:92-93, 119-136: one job per channel) — the interleaved
    entry point has no threaded form in the reference; both are timed;
  * biquad: biquad_apply_buffer per channel and section in a loop, as art.c:1011-1017 calls it (the reference has no threaded form).
Where oracle/_ref does not exist (the GPU box always has it: built files travel), the oracle restatement built with the same flags
stands in ("port").  TEST / MEASUREMENT INFRASTRUCTURE: never imported by the product."""
import ctypes as C
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import _oracle as O                                            # noqa: E402
from audio_resampler_amd.synth import noise                   # noqa: E402

CORES = os.cpu_count() or 1


def _loop(call, budget, min_calls=3, max_calls=400):
    call()                                                     # warm-up (thread pool, caches)
    t0 = time.perf_counter()
    units = calls = 0
    while calls < min_calls or time.perf_counter() - t0 < budget:
        units += call()
        calls += 1
        if calls >= max_calls:
            break
    return units, calls, time.perf_counter() - t0


def resample(ch, taps, filters, src, dst, flags, fixed=False, block=65536, budget=4.0, ratio_fn=None, threaded=True):
    """flags: reference flag bits (BH, INTERP, LOWPASS ...); returns the cpu_reference record"""
    kind = "reference" if O.have_ref("make") else "port"
    ratio = dst / src
    cap = int(math.floor((block + taps // 2) * ratio * 1.001 + 10))
    x, _ = noise(block * ch)
    x = np.ascontiguousarray(x.reshape(block, ch))
    out = np.zeros((cap, ch), np.float32)
    threads = min(ch, CORES) if threaded and ch > 1 else 1
    k = [0]
    if kind == "reference":
        L = O.load_ref("make")
        fl = flags | (O.MT if threads > 1 else 0)
        p = L.resampleFixedRatioInit(ch, taps, filters, float(src), float(dst), 0, fl) if fixed else L.resampleInit(ch, taps, filters, 0.0, fl)
        L.resampleAdvancePosition(p, taps / 2.0)
        def call():
            r = ratio_fn(k[0]) if ratio_fn else ratio
            k[0] += 1
            return L.resampleProcessInterleaved(p, x.ctypes.data_as(O.f32p), block, out.ctypes.data_as(O.f32p), cap, r).generated * ch
        free = lambda: L.resampleFree(p)
    else:
        L = O.load_oracle("fast")
        p = L.ora_resample_fixed_init(ch, taps, filters, float(src), float(dst), 0, flags) if fixed else L.ora_resample_init(ch, taps, filters, 0.0, flags)
        L.ora_resample_advance(p, taps / 2.0)
        def call():
            r = ratio_fn(k[0]) if ratio_fn else ratio
            k[0] += 1
            return L.ora_resample_interleaved(p, x.ctypes.data_as(O.f32p), block, out.ctypes.data_as(O.f32p), cap, r, threads).generated * ch
        free = lambda: L.ora_resample_free(p)
    n, calls, dt = _loop(call, budget)
    free()
    return {"Msamples_per_s": round(n / dt / 1e6, 2), "threads": threads, "host_cores": CORES, "kind": kind, "block_frames": block,
            "sample": f"{calls} calls, {dt:.1f} s wall", "how": "RESAMPLE_MULTITHREADED, one worker per channel (workers.c)" if threads > 1 else "one thread"}


def biquad_cascade(ch, sections, cutoff, block=16384, budget=3.0):
    """`sections` cascaded low-pass sections per channel over an interleaved block, as art.c:1011-1017 (one thread: the reference
    has no threaded biquad)"""
    if not O.have_ref("make"):
        return None
    L = O.load_ref("make")
    co = O.BiquadCoeffs()
    L.biquad_lowpass(C.byref(co), cutoff)
    secs = (O.Biquad * (ch * sections))()
    for i in range(ch * sections):
        L.biquad_init(C.byref(secs[i]), C.byref(co), 1.0)
    x, _ = noise(block * ch)
    x = np.ascontiguousarray(x.reshape(block, ch))
    def call():
        for c in range(ch):
            ptr = C.cast(C.addressof(x.ctypes.data_as(O.f32p).contents) + 4 * c, O.f32p)
            for s in range(sections):
                L.biquad_apply_buffer(C.byref(secs[c * sections + s]), ptr, block, ch)
        return block * ch
    n, calls, dt = _loop(call, budget)
    return {"Msamples_per_s": round(n / dt / 1e6, 2), "threads": 1, "host_cores": CORES, "kind": "reference", "block_frames": block,
            "sample": f"{calls} calls, {dt:.1f} s wall", "how": "biquad_apply_buffer per channel and section (art.c:1011-1017)"}


def decimate(ch, bits, out_bytes, rate, flags, block=16384, budget=3.0, threaded=False):
    """interleaved entry point (one thread), or the planar one with DECIMATE_MULTITHREADED (decimator.c:119-136)"""
    if not O.have_ref("make"):
        return None
    L = O.load_ref("make")
    x, _ = noise(block * ch)
    threads = min(ch, CORES) if threaded and ch > 1 else 1
    d = L.decimateInit(ch, bits, out_bytes, 1.0, rate, flags | (O.DEC_MT if threads > 1 else 0))
    if threads > 1:
        planes = np.ascontiguousarray(x.reshape(block, ch).T)
        outs = np.zeros((ch, block * out_bytes), np.uint8)
        ins_p = (O.f32p * ch)(*[planes[c].ctypes.data_as(O.f32p) for c in range(ch)])
        outs_p = (O.u8p * ch)(*[outs[c].ctypes.data_as(O.u8p) for c in range(ch)])
        def call():
            L.decimateProcessLE(d, ins_p, block, outs_p)
            return block * ch
    else:
        xi = np.ascontiguousarray(x.reshape(block, ch))
        out = np.zeros(block * ch * out_bytes, np.uint8)
        def call():
            L.decimateProcessInterleavedLE(d, xi.ctypes.data_as(O.f32p), block, out.ctypes.data_as(O.u8p))
            return block * ch
    n, calls, dt = _loop(call, budget)
    L.decimateFree(d)
    return {"Msamples_per_s": round(n / dt / 1e6, 2), "threads": threads, "host_cores": CORES, "kind": "reference", "block_frames": block,
            "sample": f"{calls} calls, {dt:.1f} s wall",
            "how": "decimateProcessLE (planar) + DECIMATE_MULTITHREADED, one worker per channel" if threads > 1 else "decimateProcessInterleavedLE, one thread"}


def config_c_pipeline(ch=8, taps=988, src=96000, dst=44100, block=16384, budget=5.0):
    """BASELINE configs[2] end to end as ART runs it (art.c:1011-1067): 2 x biquad pre-filter per channel, fixed-ratio resampler
    (RESAMPLE_MULTITHREADED), 16-bit HP-TPDF + ATH decimation (interleaved entry point), 16,384-frame blocks"""
    if not O.have_ref("make"):
        return None
    L = O.load_ref("make")
    threads = min(ch, CORES)
    p = L.resampleFixedRatioInit(ch, taps, taps, float(src), float(dst), 0, O.BH | O.INTERP | O.LOWPASS | O.MT)
    L.resampleAdvancePosition(p, taps / 2.0)
    co = O.BiquadCoeffs()
    L.biquad_lowpass(C.byref(co), dst * 0.45 / src)
    secs = (O.Biquad * (ch * 2))()
    for i in range(ch * 2):
        L.biquad_init(C.byref(secs[i]), C.byref(co), 1.0)
    d = L.decimateInit(ch, 16, 2, 1.0, dst, O.DITHER_HP | O.SHAPE_ATH)
    x0, _ = noise(block * ch)
    x0 = np.ascontiguousarray(x0.reshape(block, ch))
    x = x0.copy()
    cap = int(math.floor((block + taps // 2) * dst / src + 10))
    out = np.zeros((cap, ch), np.float32)
    pcm = np.zeros(cap * ch * 2, np.uint8)
    def call():
        np.copyto(x, x0)
        for c in range(ch):
            ptr = C.cast(C.addressof(x.ctypes.data_as(O.f32p).contents) + 4 * c, O.f32p)
            for s in range(2):
                L.biquad_apply_buffer(C.byref(secs[c * 2 + s]), ptr, block, ch)
        g = L.resampleProcessInterleaved(p, x.ctypes.data_as(O.f32p), block, out.ctypes.data_as(O.f32p), cap, 0.0).generated
        L.decimateProcessInterleavedLE(d, out.ctypes.data_as(O.f32p), g, pcm.ctypes.data_as(O.u8p))
        return g * ch
    n, calls, dt = _loop(call, budget)
    L.resampleFree(p)
    L.decimateFree(d)
    return {"Msamples_per_s": round(n / dt / 1e6, 2), "threads": threads, "host_cores": CORES, "kind": "reference", "block_frames": block,
            "sample": f"{calls} blocks, {dt:.1f} s wall", "how": "ART's block loop: biquads (1 thread), resampler (RESAMPLE_MULTITHREADED), decimator (1 thread)"}


if __name__ == "__main__":
    import json
    print(json.dumps(resample(8, 988, 988, 44100, 48000, O.BH | O.INTERP, budget=2.0)))
    print(json.dumps(biquad_cascade(8, 2, 44100 * 0.45 / 96000, budget=1.0)))
    print(json.dumps(decimate(8, 16, 2, 44100, O.DITHER_HP | O.SHAPE_ATH, budget=1.0)))
    print(json.dumps(decimate(8, 16, 2, 44100, O.DITHER_HP | O.SHAPE_ATH, budget=1.0, threaded=True)))
    print(json.dumps(config_c_pipeline(budget=2.0)))
