"""Device-resident throughput of the BASELINE.json configs other than the headline (which bench.py covers), every stage with the
reference's own CPU path timed beside it in the same run (tools/cpu_ref.py: the real reference from oracle/_ref, its own
threading — workers.c:249-371 through RESAMPLE_MULTITHREADED / DECIMATE_MULTITHREADED, decimator.c:119-136 — threads and host
cores stated) and the roofline fraction of the stage's dominant kernel (HIP events around the FIR launches; executed matrix
operations over that instruction's dense peak, or algorithmic bytes over the HBM peak for the streaming stages).
Prints one JSON line per config.  Usage: python tools/bench_configs.py [--steps N] [--no-cpu]"""
import argparse, ctypes as C, json, math, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--block", type=int, default=1 << 20)
ap.add_argument("--no-cpu", action="store_true"); ap.add_argument("--cpu-budget", type=float, default=3.0)
args = ap.parse_args()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cpu_ref
PEAK_F32, PEAK_I8, PEAK_HBM = 157.3, 5033.0, 8000.0


def fir_roofline(rs, out_samples, taps, src, dst, interp_used):
    """executed operations of the FIR kernel that ran (HIP events around its launches) over the peak that bounds it"""
    ms, launches = rs.read_timing()
    if not launches or ms <= 0:
        return None
    rate = out_samples / (ms * 1e-3)
    kernel = rs.last_kernel()
    fixed_state, pairs = rs.fixed_point()
    g = math.gcd(src, dst); P, Q = dst // g, src // g
    kpad = ((taps + int(31.0 * Q / P) + 2 + 3 + 31) // 32) * 32
    if kernel == 2 and fixed_state == 1:
        ops, peak, name, unit = 2 * kpad * pairs, PEAK_I8, "fir_i8 (fixed point, int8 matrix cores)", "TOP/s"
    elif kernel == 2:
        ops, peak, name, unit = 2 * kpad, PEAK_F32, "fir_mfma (f32 matrix cores)", "TFLOP/s"
    else:
        ops, peak, name, unit = (4 * taps + 3) if interp_used else 2 * taps, PEAK_F32, "fir_general (vector, algorithmic flop)", "TFLOP/s"
    ach = rate * ops / 1e12
    bytes_per = 4.0 * src / dst + 4.0
    return {"kernel": name, "bound": "mfma" if kernel == 2 else "valu", "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
            "avg_kernel_ms": round(ms / launches, 4), "hbm_frac_algorithmic": round(rate * bytes_per / 1e9 / PEAK_HBM, 4)}


def hbm_roofline(samples_per_s, bytes_per_sample, kernel):
    gbs = samples_per_s * bytes_per_sample / 1e9
    return {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM, "unit": "GB/s", "frac": round(gbs / PEAK_HBM, 5),
            "note": "algorithmic bytes / wall time of the whole stage (all its launches)"}
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
    rs.set_timing(True); m = 0
    for _ in range(max(3, args.steps // 2)): m += step()
    roof = fir_roofline(rs, m, taps, src, dst, bool(rs.L.resampleInterpolationUsed(rs.p)))
    rs.set_timing(False)
    cpu = None if args.no_cpu else cpu_ref.resample(ch, taps, filters, src, dst, flags, fixed=fixed, budget=args.cpu_budget, ratio_fn=ratio_fn)
    print(json.dumps({"config": name, "Msamples_per_s": round(n / dt / 1e6, 1), "ms_per_step": round(dt / args.steps * 1e3, 3),
                      "kernel": rs.last_kernel(), "block_frames": block, "channels": ch, "filters": rs.L.resampleGetNumFilters(rs.p),
                      "interp": bool(rs.L.resampleInterpolationUsed(rs.p)), "roofline": roof, "cpu_reference": cpu,
                      "gpu_over_cpu": round(n / dt / 1e6 / cpu["Msamples_per_s"], 1) if cpu else None}), flush=True)


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
# short periods (2x / 4x / 2:3 conversions: 1-3 outputs per period, taken several periods at a time) and many ring epochs per call
resample_only("R2 8ch -4 988x988 interp 44.1k->88.2k (2 outputs per period)", 8, 988, 988, 44100, 88200, BH | IN, False, blk // 2)
resample_only("R3 8ch -4 988x988 interp 48k->32k (2 outputs per 3 inputs)", 8, 988, 988, 48000, 32000, BH | IN, False, blk)
resample_only("R4 8ch -4 ART form 192k->48k (1 filter, low-pass)", 8, 988, 988, 192000, 48000, BH | IN | LP, True, blk)
resample_only("S  16ch -2 156x156 interp 44.1k->48k (short filter: 448 ring epochs per call)", 16, 156, 156, 44100, 48000, BH | IN, False, blk // 2)

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
# (one thread: the reference's threaded decimator dereferences its noise shapers even when shaping is off — decimator.c:129-130, a NULL pointer here)
cpu = None if args.no_cpu else cpu_ref.decimate(ch, 16, 2, dst, A.DITHER_HIGHPASS, block=65536, budget=args.cpu_budget, threaded=False)
print(json.dumps({"config": "C' 16-bit decimation, HP-TPDF dither, no noise shaping (no recurrence => fully parallel)", "Msamples_per_s": round(n / dt / 1e6, 1),
                  "ms_per_step": round(dt / args.steps * 1e3, 3), "roofline": hbm_roofline(n / dt, 6.0, "decimate_parallel_kernel"),
                  "cpu_reference": cpu, "gpu_over_cpu": round(n / dt / 1e6 / cpu["Msamples_per_s"], 1) if cpu else None}), flush=True)
for which in ("resample", "biquad", "decimate", "all"):
    n, dt = timed(stage(which), max(2, args.steps // 3))
    roof = cpu = cpu2 = None
    if which == "resample":
        rs.set_timing(True); m = 0
        for _ in range(3): m += stage(which)()
        roof = fir_roofline(rs, m, taps, src, dst, False); rs.set_timing(False)
        cpu = None if args.no_cpu else cpu_ref.resample(ch, taps, taps, src, dst, BH | IN | LP, fixed=True, budget=args.cpu_budget)
    elif which == "biquad":
        roof = hbm_roofline(n / dt, 8.0, "biquad_spec_kernel (+ check, commit, copy aside)")
        cpu = None if args.no_cpu else cpu_ref.biquad_cascade(ch, 2, dst * 0.45 / src, budget=args.cpu_budget)
    elif which == "decimate":
        roof = hbm_roofline(n / dt, 6.0, "decimate_pipe_kernel (serial error feedback per channel)")
        cpu = None if args.no_cpu else cpu_ref.decimate(ch, 16, 2, dst, A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE, budget=args.cpu_budget)
        cpu2 = None if args.no_cpu else cpu_ref.decimate(ch, 16, 2, dst, A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE, block=65536, budget=args.cpu_budget, threaded=True)
    else:
        cpu = None if args.no_cpu else cpu_ref.config_c_pipeline(ch, taps, src, dst, budget=args.cpu_budget)
    line = {"config": f"C  8ch 96k->44.1k -4 fixed (147x988 no-lerp, LP) + 2x biquad + 16-bit ATH decimate: stage={which}",
            "Msamples_per_s": round(n / dt / 1e6, 1), "ms_per_step": round(dt / max(2, args.steps // 3) * 1e3, 3), "block_frames": block,
            "roofline": roof, "cpu_reference": cpu, "gpu_over_cpu": round(n / dt / 1e6 / cpu["Msamples_per_s"], 1) if cpu else None}
    if cpu2: line["cpu_reference_threaded"] = cpu2
    print(json.dumps(line), flush=True)


# ---- the shaped decimator is a serial recurrence per channel (one lane each): it scales with channels, not with the chip
for chn in (64, 512):
    decn = A.Decimator(chn, 16, 2, 1.0, dst, A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE); decn.set_stream(stream)
    nf = 1 << 18
    xn, _ = noise(nf * chn); d_xn = torch.from_numpy(xn.reshape(nf, chn)).cuda(); d_pn = torch.empty(nf * chn * 2, dtype=torch.uint8, device="cuda")
    def many():
        decn.process_device(d_xn, nf, d_pn)
        return nf * chn
    n, dt = timed(many, max(2, args.steps // 3))
    cpu = None if args.no_cpu else cpu_ref.decimate(chn, 16, 2, dst, A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE, block=65536, budget=args.cpu_budget, threaded=True)
    print(json.dumps({"config": f"C'' 16-bit ATH-shaped decimation of {chn} channels (one serial lane per channel)", "Msamples_per_s": round(n / dt / 1e6, 1),
                      "ms_per_step": round(dt / max(2, args.steps // 3) * 1e3, 3), "block_frames": nf, "channels": chn,
                      "roofline": hbm_roofline(n / dt, 6.0, "decimate_pipe_kernel / decimate_lds_kernel"), "cpu_reference": cpu,
                      "gpu_over_cpu": round(n / dt / 1e6 / cpu["Msamples_per_s"], 1) if cpu else None}), flush=True)
    del decn, d_xn, d_pn

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
