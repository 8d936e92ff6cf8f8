"""Device-resident throughput of the 8-byte sample build (libartamd64.so, reference PATH_WIDTH=64) with the CPU
reference64 timed beside it when oracle/_ref/libartref64_make.so exists.  One JSON line per stage.
Usage: python tools/bench_wide.py [--steps N] [--block FRAMES]"""
import argparse, ctypes as C, json, math, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import audio_resampler_amd as A

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--block", type=int, default=1 << 18)
args = ap.parse_args()
W = A.wide()
stream = torch.cuda.current_stream().cuda_stream
BH, IN, LP = A.BLACKMAN_HARRIS, A.SUBSAMPLE_INTERPOLATE, A.INCLUDE_LOWPASS


def timed(fn, steps):
    # untimed pre-roll: the device reaches its steady-state clocks only after tens of milliseconds of load (see bench.py)
    t_pre = time.perf_counter(); fn(); torch.cuda.synchronize()
    if time.perf_counter() - t_pre < 0.005:                  # (the slow serial stages are their own pre-roll)
        while time.perf_counter() - t_pre < 0.15:
            for _ in range(4): fn()
            torch.cuda.synchronize()
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for _ in range(steps): n += fn()
    torch.cuda.synchronize()
    return n, time.perf_counter() - t0


def signal(frames, ch):
    rng = np.random.default_rng(7)
    return (rng.random((frames, ch)) - 0.5)


def cpu_resample(ch, taps, filters, src, dst, flags, fixed, frames=16384):
    import _oracle
    O = _oracle.wide()
    if not O.have_ref("make"):
        return None
    r = O.RefResampler(ch, taps, filters, 0.0, flags | 0x8, fixed=(float(src), float(dst), 0) if fixed else None, kind="make")
    r.advance(taps / 2.0)
    x = signal(frames, ch)
    cap = int(frames * dst / src) + 64
    r.process(x, cap, dst / src)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 3.0:
        _, g, _ = r.process(x, cap, dst / src); n += g * ch
    return round(n / (time.perf_counter() - t0) / 1e6, 2)


def resample(name, ch, taps, filters, src, dst, flags, fixed, block):
    rs = W.Resampler(ch, taps, filters, 0.0, flags, fixed=(src, dst, 0) if fixed else None)
    rs.advance(taps / 2.0); rs.set_stream(stream)
    d_in = torch.from_numpy(signal(block, ch)).cuda()
    ratio = dst / src
    cap = int(math.floor((block + taps // 2) * ratio * 1.001 + 10)); d_out = torch.empty(cap, ch, dtype=torch.float64, device="cuda")
    def step():
        u, g = rs.process_device(d_in, block, d_out, cap, 0.0 if fixed else ratio)
        return g * ch
    n, dt = timed(step, args.steps)
    T = rs.c.numTaps
    flops = 2.0 * T * (2 if rs.L.resampleInterpolationUsed(rs.p) else 1)
    print(json.dumps({"stage": name, "Msamples_per_s": round(n / dt / 1e6, 1), "ms_per_step": round(dt / args.steps * 1e3, 3),
                      "fp64_TFLOPs": round(n / dt * flops / 1e12, 2), "kernel": rs.last_kernel(), "block_frames": block,
                      "cpu_reference64_Msamples_per_s": cpu_resample(ch, taps, filters, src, dst, flags, fixed)}), flush=True)


blk = args.block
resample("wide A  8ch -4 988x988 interp 44.1k->48k", 8, 988, 988, 44100, 48000, BH | IN, False, blk)
resample("wide A' 8ch -4 ART form (160x988 no-lerp)", 8, 988, 988, 44100, 48000, BH | IN | LP, True, blk)
resample("wide B  stereo -3 380x380 interp", 2, 380, 380, 44100, 48000, BH | IN, False, blk)

ch, block, dst, src = 8, blk, 44100, 96000
L = W.lib()
co = W.BiquadCoefficients(); L.biquad_lowpass(C.byref(co), dst * 0.45 / src)
secs = (W.Biquad * (ch * 2))()
for i in range(ch * 2): L.biquad_init(C.byref(secs[i]), C.byref(co), 1.0)
bank = W.BiquadBank(secs, ch, 2); bank.set_stream(stream)
d_buf = torch.from_numpy(signal(block, ch)).cuda()
d_pcm = torch.empty(block * ch * 2, dtype=torch.uint8, device="cuda")
def biq(): bank.apply_device(d_buf, block); return block * ch
n, dt = timed(biq, args.steps)
print(json.dumps({"stage": "wide biquad 2 x order-2, 8 ch", "Msamples_per_s": round(n / dt / 1e6, 1)}), flush=True)
for nm, fl in (("ATH shaping + HP dither", A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE), ("HP dither only", A.DITHER_HIGHPASS)):
    dec = W.Decimator(ch, 16, 2, 1.0, dst, fl); dec.set_stream(stream)
    def de(): dec.process_device(d_buf, block, d_pcm); return block * ch
    n, dt = timed(de, args.steps)
    print(json.dumps({"stage": f"wide decimate 16-bit, {nm}, 8 ch", "Msamples_per_s": round(n / dt / 1e6, 1)}), flush=True)
