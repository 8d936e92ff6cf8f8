"""One named workload, a few dozen launches, for rocprofv3 (kernel stats and --pmc passes): the kernels bench.py does not reach.
Usage: python tools/profile_case.py CASE [steps]     CASE in: general_E general_P matrix_P general_A matrix_B matrix_D4 matrix_D32 fixed_D4 fixed_D32 wide biquad biquad_serial decimate strict
Prints one JSON line: what ran, samples per launch, algorithmic flop and bytes per sample (tools/roofline_report.py reads it)."""
import ctypes as C, json, math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
case = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if case == "biquad_serial": os.environ["ARTAMD_BIQUAD_SERIAL"] = "1"
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
B = A.wide() if case == "wide" else A
stream = torch.cuda.current_stream().cuda_stream
BH, IN = A.BLACKMAN_HARRIS, A.SUBSAMPLE_INTERPOLATE


def resampler_case(ch, taps, filters, flags, block, kernel=0, ratio_fn=None, dtype=np.float32, mod=A, ring=1):
    """ring: distinct input / output buffer pairs the steps walk round — with ring x (in + out) bytes past the 256 MiB Infinity Cache every
    launch's input comes from HBM and the memory-side counters see it (a 4 MB input re-read every step never leaves the cache)"""
    rs = mod.Resampler(ch, taps, filters, 0.0, flags); rs.advance(taps / 2.0); rs.set_stream(stream)
    if kernel: rs.set_kernel(kernel)
    x, _ = noise(block * ch); d_ins = [torch.from_numpy(x.reshape(block, ch).astype(dtype)).cuda() for _ in range(ring)]
    ratio = 48000 / 44100; cap = int((block + taps // 2) * ratio * 1.001 + 10); d_outs = [torch.empty(cap, ch, device="cuda", dtype=d_ins [0].dtype) for _ in range(ring)]
    k = [0]
    def step():
        r = ratio_fn(k[0]) if ratio_fn else ratio; k[0] += 1
        u, g = rs.process_device(d_ins [k[0] % ring], block, d_outs [k[0] % ring], cap, r); return g * ch
    return step, rs


info = {"case": case}
if case == "general_E":        # BASELINE configs[4]: stereo ASRC, preset -3, nearest filter, ratio moving +-100 ppm per 65,536-frame block
    step, rs = resampler_case(2, 380, 380, BH, 65536, ratio_fn=lambda k: 48000 / 44100 * (1 + 100e-6 * math.sin(2 * math.pi * k / 64)))
    info.update(kernel="fir_general_kernel", flop_per_sample=2 * 380, bytes_per_sample=4 * 44100 / 48000 + 4, peak="fp32")
elif case == "general_P":      # BASELINE configs[0]: mono preset -1 (48 x 48) interpolating, the general kernel pinned; 40 buffer pairs = 344 MB: past the Infinity Cache
    step, rs = resampler_case(1, 48, 48, BH | IN, 1 << 20, kernel=1, ring=40)         # (1: the general kernel pinned — at 1M frames the library itself takes the matrix path: matrix_P)
    info.update(kernel="fir_general_kernel", flop_per_sample=4 * 48 + 3, bytes_per_sample=4 * 44100 / 48000 + 4, peak="fp32")
elif case == "matrix_P":       # the same config as the library runs it: the f32 matrix kernel, one launch for the call's 1,456 ring epochs (2 chunks: K = 64)
    step, rs = resampler_case(1, 48, 48, BH | IN, 1 << 20, ring=40)
    info.update(kernel="fir_mfma", flop_per_sample=2 * 64, bytes_per_sample=4 * 44100 / 48000 + 4, peak="fp32")
elif case == "general_A":      # the headline shape on the general kernel
    step, rs = resampler_case(8, 988, 988, BH | IN, 1 << 20, kernel=1)
    info.update(kernel="fir_general_kernel", flop_per_sample=4 * 988 + 3, bytes_per_sample=4 * 44100 / 48000 + 4, peak="fp32")
elif case == "matrix_B":       # BASELINE configs[1]: stereo preset -3 (380 x 380) interpolating, on the matrix cores (13 chunks: K = 416)
    step, rs = resampler_case(2, 380, 380, BH | IN, 1 << 20)
    info.update(kernel="fir_mfma_stream_kernel", flop_per_sample=2 * 416, bytes_per_sample=4 * 44100 / 48000 + 4, peak="fp32")
elif case == "matrix_D4":      # BASELINE configs[3]: the 4-channel shard one GPU of 8 owns (988 x 988 interpolating)
    step, rs = resampler_case(4, 988, 988, BH | IN, 1 << 20, kernel=6)          # (6: the f32 streaming kernel; left alone the call runs in fixed point)
    info.update(kernel="fir_mfma_stream_kernel", flop_per_sample=2 * 1024, bytes_per_sample=4 * 44100 / 48000 + 4, peak="fp32")
elif case == "matrix_D32":     # BASELINE configs[3] on ONE GPU: all 32 channels
    step, rs = resampler_case(32, 988, 988, BH | IN, 1 << 18, kernel=6)
    info.update(kernel="fir_mfma_stream_kernel", flop_per_sample=2 * 1024, bytes_per_sample=4 * 44100 / 48000 + 4, peak="fp32")
elif case in ("fixed_D4", "fixed_D32"):     # the same two shapes as the library runs them: the fixed-point kernel (integer matrix cores)
    step, rs = resampler_case(4, 988, 988, BH | IN, 1 << 20) if case == "fixed_D4" else resampler_case(32, 988, 988, BH | IN, 1 << 18)
    step(); state, pairs = rs.fixed_point(); assert state == 1, state
    # (K columns a tile walks: 33 images for the 64-slot tiles of the slab kernel, 32 for the 32-slot kernels — as bench.py prices them since round 6)
    kcols = 1056 if rs.fixed_point_kernel() == "fir_i8_slab_kernel" else 1024
    info.update(kernel="fir_i8_", flop_per_sample=round(2 * kcols * pairs, 1), bytes_per_sample=4 * 44100 / 48000 + 4, peak="i8")
elif case == "strict":         # RESAMPLE_STRICT_ORDER: the parity instrument
    step, rs = resampler_case(8, 988, 988, BH | IN | A.RESAMPLE_STRICT_ORDER, 1 << 16)
    info.update(kernel="fir_strict_kernel", flop_per_sample=4 * 988 + 3, bytes_per_sample=4 * 44100 / 48000 + 4, peak="fp32")
elif case == "wide":           # the 8-byte sample build on the headline shape: fp64 matrix cores, both interpolation rows carried
    step, rs = resampler_case(8, 988, 988, BH | IN, 1 << 20, dtype=np.float64, mod=B)
    info.update(kernel="fir_mfma64_kernel", flop_per_sample=2 * 2 * 1024, bytes_per_sample=8 * 44100 / 48000 + 8, peak="fp64_mfma")
elif case in ("biquad", "biquad_serial"):     # config C's pre-filter: 8 ch x 2 low-pass sections, 1M frames
    ch, block = 8, 1 << 20
    L = A.lib(); co = A.BiquadCoefficients(); L.biquad_lowpass(C.byref(co), 44100 * 0.45 / 96000)
    secs = (A.Biquad * (ch * 2))()
    for i in range(ch * 2): L.biquad_init(C.byref(secs[i]), C.byref(co), 1.0)
    bank = A.BiquadBank(secs, ch, 2); bank.set_stream(stream)
    x, _ = noise(block * ch); d = torch.from_numpy(x.reshape(block, ch)).cuda()
    def step():
        bank.apply_device(d, block); return block * ch
    info.update(kernel="biquad_spec_kernel" if case == "biquad" else "biquad_order2_ff_kernel", flop_per_sample=18, bytes_per_sample=8, peak="hbm")
elif case == "decimate":       # config C's output stage: 16-bit, high-pass TPDF dither, ATH-curve noise shaping (serial per channel)
    ch, block = 8, 1 << 18
    dec = A.Decimator(ch, 16, 2, 1.0, 44100, A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE); dec.set_stream(stream)
    x, _ = noise(block * ch); d = torch.from_numpy(x.reshape(block, ch)).cuda(); out = torch.empty(block * ch * 2, dtype=torch.uint8, device="cuda")
    def step():
        dec.process_device(d, block, out); return block * ch
    info.update(kernel="decimate_pipe_kernel", flop_per_sample=30, bytes_per_sample=6, peak="hbm")
    steps = min(steps, 6)
else:
    raise SystemExit("unknown case")

t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:
    step(); torch.cuda.synchronize()
n = 0; t0 = time.perf_counter()
for _ in range(steps): n += step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
info.update(steps=steps, samples_per_launch=n / steps, Msamples_per_s=round(n / dt / 1e6, 1), ms_per_step=round(dt / steps * 1e3, 4))
print(json.dumps(info), flush=True)
