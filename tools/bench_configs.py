"""Device-resident throughput of the BASELINE.json configs other than the headline (which bench.py covers).
Prints one JSON line per config.  Usage: python tools/bench_configs.py [--steps N]"""
import argparse, ctypes as C, json, math, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--block", type=int, default=1 << 20)
args = ap.parse_args()
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, steps):
    # untimed pre-roll: the device reaches its steady-state clocks only after tens of milliseconds of load (see bench.py)
    t_pre = time.perf_counter(); fn(); torch.cuda.synchronize()
    if time.perf_counter() - t_pre < 0.005:                  # (the slow serial stages are their own pre-roll)
        while time.perf_counter() - t_pre < 0.15:
            for _ in range(4): fn()
            torch.cuda.synchronize()
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for _ in range(steps): n += fn()
    torch.cuda.synchronize()
    return n, time.perf_counter() - t0


def resample_only(name, ch, taps, filters, src, dst, flags, fixed, block, ratio_fn=None):
    rs = A.Resampler(ch, taps, filters, 0.0, flags, fixed=(src, dst, 0) if fixed else None)
    rs.advance(taps / 2.0); rs.set_stream(stream)
    x, _ = noise(block * ch); d_in = torch.from_numpy(x.reshape(block, ch)).cuda()
    ratio = dst / src
    cap = int(math.floor((block + taps // 2) * ratio * 1.001 + 10)); d_out = torch.empty(cap, ch, device="cuda")
    k = [0]
    def step():
        r = ratio_fn(k[0]) if ratio_fn else ratio; k[0] += 1
        u, g = rs.process_device(d_in, block, d_out, cap, 0.0 if fixed else r)
        assert u == block
        return g * ch
    n, dt = timed(step, args.steps)
    print(json.dumps({"config": name, "Msamples_per_s": round(n / dt / 1e6, 1), "ms_per_step": round(dt / args.steps * 1e3, 3),
                      "kernel": rs.last_kernel(), "block_frames": block, "channels": ch, "filters": rs.L.resampleGetNumFilters(rs.p),
                      "interp": bool(rs.L.resampleInterpolationUsed(rs.p))}), flush=True)


BH, IN, LP = A.BLACKMAN_HARRIS, A.SUBSAMPLE_INTERPOLATE, A.INCLUDE_LOWPASS
blk = args.block
resample_only("B  stereo -3 380x380 interp 44.1k->48k (artest form)", 2, 380, 380, 44100, 48000, BH | IN, False, blk)
resample_only("B' stereo -3 ART form (160x380 no-lerp, SNAP)", 2, 380, 380, 44100, 48000, BH | IN | LP, True, blk)
resample_only("A' 8ch -4 ART form (160x988 no-lerp, SNAP)", 8, 988, 988, 44100, 48000, BH | IN | LP, True, blk)
resample_only("D  32ch -4 988x988 interp 44.1k->48k on ONE GPU", 32, 988, 988, 44100, 48000, BH | IN, False, blk // 4)
resample_only("D/8 4ch -4 (one GPU's shard of D)", 4, 988, 988, 44100, 48000, BH | IN, False, blk)
resample_only("E  stereo ASRC -3 no-lerp, ratio +-100ppm per block", 2, 380, 380, 44100, 48000, BH, False, 65536,
              ratio_fn=lambda k: 48000 / 44100 * (1 + 100e-6 * math.sin(2 * math.pi * k / 64)))
resample_only("P  mono -1 48x48 interp", 1, 48, 48, 44100, 48000, BH | IN, False, blk)

# ---- C: 8 ch 96k -> 44.1k, preset -4 fixed ratio (147x988 no-lerp, auto low-pass), 2x biquad LP pre-filter, 16-bit decimation
ch, taps, src, dst = 8, 988, 96000, 44100
block = blk
rs = A.Resampler(ch, taps, taps, 0.0, BH | IN | LP, fixed=(src, dst, 0)); rs.advance(taps / 2.0); rs.set_stream(stream)
L = A.lib()
co = A.BiquadCoefficients(); L.biquad_lowpass(C.byref(co), dst * 0.45 / src)
secs = (A.Biquad * (ch * 2))()
for i in range(ch * 2): L.biquad_init(C.byref(secs[i]), C.byref(co), 1.0)
bank = A.BiquadBank(secs, ch, 2); bank.set_stream(stream)
dec = A.Decimator(ch, 16, 2, 1.0, dst, A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE); dec.set_stream(stream)
x, _ = noise(block * ch); d_src = torch.from_numpy(x.reshape(block, ch)).cuda(); d_in = torch.empty_like(d_src)
cap = int(math.floor((block + taps // 2) * dst / src + 10)); d_out = torch.empty(cap, ch, device="cuda")
d_pcm = torch.empty(cap * ch * 2, dtype=torch.uint8, device="cuda")
def stage(which):
    def step():
        g = 0
        if which in ("biquad", "all"):
            d_in.copy_(d_src); bank.apply_device(d_in, block); g = block
        if which in ("resample", "all"):
            u, g = rs.process_device(d_in if which == "all" else d_src, block, d_out, cap, 0.0)
        if which in ("decimate", "all"):
            gg = g if which == "all" else cap - 16
            dec.process_device(d_out, gg, d_pcm); g = gg
        return g * ch
    return step
dec0 = A.Decimator(ch, 16, 2, 1.0, dst, A.DITHER_HIGHPASS); dec0.set_stream(stream)
def dither_only():
    dec0.process_device(d_out, cap - 16, d_pcm)
    return (cap - 16) * ch
n, dt = timed(dither_only, args.steps)
print(json.dumps({"config": "C' 16-bit decimation, HP-TPDF dither, no noise shaping (no recurrence => fully parallel)", "Msamples_per_s": round(n / dt / 1e6, 1),
                  "ms_per_step": round(dt / args.steps * 1e3, 3)}), flush=True)
for which in ("resample", "biquad", "decimate", "all"):
    n, dt = timed(stage(which), max(2, args.steps // 3))
    print(json.dumps({"config": f"C  8ch 96k->44.1k -4 fixed (147x988 no-lerp, LP) + 2x biquad + 16-bit ATH decimate: stage={which}",
                      "Msamples_per_s": round(n / dt / 1e6, 1), "ms_per_step": round(dt / max(2, args.steps // 3) * 1e3, 3), "block_frames": block}), flush=True)


# ---- C pipelined: the three stages of successive blocks overlap on three HIP streams (block k+1 in the biquads
# while block k is resampled and block k-1 decimated); buffers are double-buffered and ordered with events.
sA, sB, sC = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
bank.set_stream(sA.cuda_stream); rs.set_stream(sB.cuda_stream); dec.set_stream(sC.cuda_stream)
ins = [torch.empty_like(d_src) for _ in range(2)]
outs = [torch.empty(cap, ch, device="cuda") for _ in range(2)]
ev_in = [torch.cuda.Event() for _ in range(2)]; ev_out = [torch.cuda.Event() for _ in range(2)]
ev_in_free = [torch.cuda.Event() for _ in range(2)]; ev_out_free = [torch.cuda.Event() for _ in range(2)]
torch.cuda.synchronize()
for e in ev_in_free + ev_out_free: e.record()
def run_pipeline(nblocks):
    total = 0
    for k in range(nblocks):
        b = k & 1
        with torch.cuda.stream(sA):
            sA.wait_event(ev_in_free[b])                  # resampler finished with this input buffer
            ins[b].copy_(d_src, non_blocking=True); bank.apply_device(ins[b], block); ev_in[b].record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(ev_in[b]); sB.wait_event(ev_out_free[b])
            u, g = rs.process_device(ins[b], block, outs[b], cap, 0.0)
            ev_in_free[b].record(sB); ev_out[b].record(sB)
        with torch.cuda.stream(sC):
            sC.wait_event(ev_out[b])
            dec.process_device(outs[b], g, d_pcm); ev_out_free[b].record(sC)
        total += g * ch
    torch.cuda.synchronize()
    return total
run_pipeline(3)
t0 = time.perf_counter(); n = run_pipeline(8); dt = time.perf_counter() - t0
print(json.dumps({"config": "C  8ch 96k->44.1k -4 fixed + 2x biquad + 16-bit ATH decimate: stages PIPELINED on 3 streams (8 blocks)",
                  "Msamples_per_s": round(n / dt / 1e6, 1), "ms_per_block": round(dt / 8 * 1e3, 3), "block_frames": block}), flush=True)
