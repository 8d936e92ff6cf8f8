#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X sinc resampler.

Metric (BASELINE.json): output Msamples/s, 44.1 kHz -> 48 kHz, preset -4 (988 filters x 988 taps,
Blackman-Harris, interpolating = what `artest -4 -c8 -s44100 -d48000` runs), 8 channels, float32.

A "step" is ONE streaming call of the drop-in entry point on device-resident buffers:
    resampleProcessInterleavedDevice(cxt, d_in[block x C], block, d_out, cap, ratio)
i.e. the whole hot path (host position planning, FIR kernel, history roll) for one block of
`--block-frames` input frames x C channels, with the input already in HBM when timing starts.

Multi-GPU (`--gpus N`, launched by torch.distributed.run): channels shard across ranks — every rank
owns a contiguous channel slice of ONE stream (its own context, filter-bank replica and history in its
own HBM).  No data-path collective exists or is needed; RCCL is used only for the
timing barrier and the max-over-ranks reduction.
  --scaling weak   (default) 8 channels per GPU, the stream has 8N channels; N = 1 is the metric's config
  --scaling strong ONE 32-channel stream (--total-channels), 32/N channels per GPU: BASELINE.json configs[3]
                   (4 channels per GPU at N = 8)
Whatever --scaling says, the line also carries "config_d": the strong-scaling figure of configs[3] (one 32-channel stream shared
among the N ranks), measured in the same run — the driver's default command line reports both.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TAPS = FILTERS = 988                      # preset -4 (reference art.c:163-166 / artest.c:166-169)
SRC, DST = 44100, 48000
FLOP_PER_SAMPLE = 4 * TAPS + 3            # two T-tap dot products + lerp           
BYTES_PER_SAMPLE = 4.0 * SRC / DST + 4.0  # float32 in (1/R frames per out frame) + float32 out
PEAK_I8_TOPS = 5033.0                     # MI355X dense int8 MFMA peak: twice the 2.5 PF bf16 rate, = the ~5 PF dense fp8 figure (MI355X_MICROARCH.md)
PEAK_FP32_TFLOPS = 157.3                  # MI355X f32 FMA / f32-MFMA dense peak (MI355X_MICROARCH.md)
# What v_mfma_i32_32x32x32_i8 SUSTAINS on this chip with live (pseudo-random) operands, registers only, nothing else running: 3,030-3,190 TOP/s
# at a shader clock of 1.70-1.93 GHz, against 4,530 at 2.39 GHz on zero operands (tools/micro/mfma_sustain.hip, profiles/r4_mfma_sustain.txt):
# power management, not the schedule, bounds the integer matrix cores on real data.  Reported BESIDE the nominal peak, never instead of it.
SUSTAINED_LIVE_I8_TOPS = 3190.0
PEAK_HBM_GBS = 8000.0


def cpu_baseline(channels, seconds_budget=12.0):
    """Reference CPU path on this host's cores, bounded sample of the same workload.
    kind "reference": oracle/_ref/libartref_make.so — the real reference (its own Makefile flags) with
    RESAMPLE_MULTITHREADED (workers.c, one thread per channel) and 65,536-frame blocks, its best setting.
    kind "port": the oracle restatement built with the same flags, one pthread per channel."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from audio_resampler_amd.synth import noise
    block = 65536
    ratio = DST / SRC
    cap = int(math.floor((block + TAPS // 2) * ratio + 10))
    x, _ = noise(block * channels)
    x = np.ascontiguousarray(x.reshape(block, channels))
    out = np.zeros((cap, channels), np.float32)
    cores = os.cpu_count() or 1
    threads = min(channels, cores)
    if O.have_ref("make"):
        L = O.load_ref("make")
        p = L.resampleInit(channels, TAPS, FILTERS, 0.0, O.BH | O.INTERP | O.MT)
        L.resampleAdvancePosition(p, TAPS / 2.0)
        call = lambda: L.resampleProcessInterleaved(p, x.ctypes.data_as(O.f32p), block, out.ctypes.data_as(O.f32p), cap, ratio).generated
        kind, free = "reference", lambda: L.resampleFree(p)
    else:
        L = O.load_oracle("fast")
        p = L.ora_resample_init(channels, TAPS, FILTERS, 0.0, O.BH | O.INTERP)
        L.ora_resample_advance(p, TAPS / 2.0)
        call = lambda: L.ora_resample_interleaved(p, x.ctypes.data_as(O.f32p), block, out.ctypes.data_as(O.f32p), cap, ratio, threads).generated
        kind, free = "port", lambda: L.ora_resample_free(p)
    call()                                   # warm-up (thread pool, caches)
    t0 = time.perf_counter()
    frames = blocks = 0
    while blocks < 3 or time.perf_counter() - t0 < seconds_budget:
        frames += call()
        blocks += 1
        if blocks >= 200:
            break
    dt = time.perf_counter() - t0
    free()
    return {"value": round(frames * channels / dt / 1e6, 3), "unit": "Msamples/s", "cores": threads, "host_cores": cores,
            "kind": kind, "sample": f"{blocks} blocks x {block} frames x {channels} ch ({blocks * block / SRC:.1f} s of audio), "
                                    f"{dt:.1f} s wall; 988x988 interpolating, one thread per channel"}


def stream_slice(block, lo, hi):
    """channels [lo, hi) of THE synthetic stream, interleaved [block, hi - lo]: channel c of the stream is artest's noise generator
    started from seed + 2c — every rank cuts its slice out of the same stream, whatever the number of ranks"""
    from audio_resampler_amd.synth import noise, SEED
    cols = [noise(block, state=(SEED + 2 * c) | 1)[0] for c in range(lo, hi)]
    return np.ascontiguousarray(np.stack(cols, axis=1))


def other_configs(A, torch, steps, warmup):
    """BASELINE.json configs[1], [2] and [4] beside the headline, in the same run (N = 1 only; never `value`): each W warmup + K timed
    steps of device-resident calls between synchronisations, with the FIR kernel's own time from HIP events and the fraction of the
    peak that bounds it.  Config C is timed END TO END (2 x biquad low-pass pre-filter, FIR, 16-bit HP-TPDF + ATH-shaped decimation:
    the shaped decimator is a serial recurrence per channel — art.c:1011-1067 / decimator.c:255-283 — and the floor of that pipeline)
    and stage by stage."""
    from audio_resampler_amd.synth import noise
    stream = torch.cuda.current_stream().cuda_stream
    BH, IN, LP = A.BLACKMAN_HARRIS, A.SUBSAMPLE_INTERPOLATE, A.INCLUDE_LOWPASS
    out = {}

    def run(fn, k):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in range(k):
            n += fn()
        torch.cuda.synchronize()
        return n, time.perf_counter() - t0

    def fir_leg(ch, taps, filters, src, dst, flags, fixed, block, k, ratio_fn=None):
        rs = A.Resampler(ch, taps, filters, 0.0, flags, fixed=(src, dst, 0) if fixed else None)
        rs.advance(taps / 2.0)
        rs.set_stream(stream)
        x, _ = noise(block * ch)
        d_in = torch.from_numpy(x.reshape(block, ch)).cuda()
        ratio = dst / src
        cap = int(math.floor((block + taps // 2) * ratio * 1.001 + 10))
        d_out = torch.empty(cap, ch, device="cuda", dtype=torch.float32)
        calls = [0]

        def step():
            r = ratio_fn(calls[0]) if ratio_fn else ratio
            calls[0] += 1
            used, made = rs.process_device(d_in, block, d_out, cap, 0.0 if fixed else r)
            assert used == block and 0 < made < cap, (used, made)
            return made * ch
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in range(k):
            n += step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rs.set_timing(True)                                # (the kernel's own time: the same steps again, with events — see timed_region)
        n_ev = 0
        for _ in range(k):
            n_ev += step()
        torch.cuda.synchronize()
        ms, launches = rs.read_timing()
        rs.set_timing(False)
        kernel = rs.last_kernel()
        fixed_state, pairs = rs.fixed_point()
        interp = bool(rs.L.resampleInterpolationUsed(rs.p))
        g = math.gcd(src, dst)
        # (K columns a tile walks: 64-slot tiles of the slab kernel spread their rows' window starts twice as far as 32-slot tiles)
        rows = 64 if rs.fixed_point_kernel() == "fir_i8_slab_kernel" else 32
        kpad = ((taps + int((rows - 1.0) * (src // g) / (dst // g)) + 2 + 3 + 31) // 32) * 32
        rate = n_ev / (ms * 1e-3) if ms > 0 else 0.0
        if kernel == 2 and fixed_state == 1:
            name, ops, peak, unit = rs.fixed_point_kernel() or "fir_i8", 2 * kpad * pairs, PEAK_I8_TOPS, "TOP/s"
        elif kernel == 2:
            name, ops, peak, unit = "fir_mfma_stream_kernel (f32 matrix cores)", 2 * kpad, PEAK_FP32_TFLOPS, "TFLOP/s"
        else:
            name, ops, peak, unit = "fir_general_kernel", ((4 * taps + 3) if interp else 2 * taps), PEAK_FP32_TFLOPS, "TFLOP/s (algorithmic flop, vector units)"
        return {"value": round(n / dt / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt / k * 1e3, 4), "steps": k, "block_frames": block,
                "channels": ch, "fir_kernel": name, "avg_kernel_ms": round(ms / max(launches, 1), 4), "launches": launches,
                "roofline": {"bound": "mfma" if kernel == 2 else "valu", "achieved": round(rate * ops / 1e12, 3), "peak": peak, "unit": unit,
                             "frac": round(rate * ops / 1e12 / peak, 4)}}, rs, d_out, cap

    # ---- configs[1]: stereo 44.1k -> 48k, preset -3 (380 x 380), Blackman-Harris, interpolating, 1M-frame calls
    leg, rs_b, _, _ = fir_leg(2, 380, 380, SRC, DST, BH | IN, False, 1 << 20, steps)
    leg["workload"] = "BASELINE.json configs[1]: stereo 44.1k->48k preset -3 = 380 filters x 380 taps Blackman-Harris interpolating, 1,048,576 input frames per call, device-resident"
    out["config_b"] = leg
    del rs_b

    # ---- configs[4]: stereo ASRC, ratio 48000/44100 x (1 +- 100 ppm) updated every block, preset -3, nearest filter (no lerp), 65,536-frame blocks
    leg, rs_e, _, _ = fir_leg(2, 380, 380, SRC, DST, BH, False, 65536, max(steps, 32),
                              ratio_fn=lambda i: DST / SRC * (1.0 + 100e-6 * math.sin(2.0 * math.pi * i / 64.0)))
    leg["workload"] = ("BASELINE.json configs[4]: stereo ASRC, ratio 48000/44100 x (1 +- 100 ppm) changed on every call, preset -3 = 380 x 380 nearest filter "
                       "(no interpolation), 65,536 input frames per call, device-resident")
    out["config_e"] = leg
    del rs_e

    # ---- configs[2]: 8 ch 96k -> 44.1k, preset -4 fixed ratio (-l implicit low-pass: 147 filters x 988 taps, nearest filter), -p biquad
    # cascade (2 low-pass sections per channel in front), 16-bit output with HP-TPDF dither + ATH noise shaping
    ch, taps, src, dst, block = 8, 988, 96000, 44100, 1 << 20
    k = max(2, min(steps, 5))
    leg, rs_c, d_out, cap = fir_leg(ch, taps, taps, src, dst, BH | IN | LP, True, block, k)
    L = A.lib()
    co = A.BiquadCoefficients()
    L.biquad_lowpass(C.byref(co), dst * 0.45 / src)
    secs = (A.Biquad * (ch * 2))()
    for i in range(ch * 2):
        L.biquad_init(C.byref(secs[i]), C.byref(co), 1.0)
    bank = A.BiquadBank(secs, ch, 2)
    bank.set_stream(stream)
    dec = A.Decimator(ch, 16, 2, 1.0, dst, A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE)
    dec.set_stream(stream)
    x, _ = noise(block * ch)
    d_src = torch.from_numpy(x.reshape(block, ch)).cuda()
    d_in = torch.empty_like(d_src)
    d_pcm = torch.empty(cap * ch * 2, dtype=torch.uint8, device="cuda")
    made_c = [0]

    def pre():
        d_in.copy_(d_src)
        bank.apply_device(d_in, block)
        return block * ch

    def fir():
        used, made = rs_c.process_device(d_in, block, d_out, cap, 0.0)
        assert used == block and 0 < made < cap, (used, made)
        made_c[0] = made
        return made * ch

    def post():
        dec.process_device(d_out, made_c[0], d_pcm)
        return made_c[0] * ch

    def whole():
        pre()
        n = fir()
        post()
        return n
    n_all, dt_all = run(whole, k)
    stage_ms = {}
    for name, fn in (("biquad_prefilter", pre), ("fir", fir), ("decimate", post)):
        _, dts = run(fn, k)
        stage_ms[name] = round(dts / k * 1e3, 4)
    tot = sum(stage_ms.values())
    out["config_c"] = {"value": round(n_all / dt_all / 1e6, 2), "unit": "Msamples/s (out)", "ms_per_step": round(dt_all / k * 1e3, 4), "steps": k, "block_frames": block,
                       "channels": ch, "stage_ms": stage_ms, "stage_share": {kk: round(v / tot, 4) for kk, v in stage_ms.items()},
                       "fir": {kk: leg[kk] for kk in ("value", "fir_kernel", "avg_kernel_ms", "roofline")},
                       "floor": "decimate_pipe_kernel: the ATH-shaped 16-bit decimator is a serial error-feedback recurrence per channel (one lane per channel, "
                                "8 lanes here): it bounds this pipeline whatever the FIR stage does; without noise shaping the decimator is fully parallel",
                       "workload": "BASELINE.json configs[2]: 8 ch 96k->44.1k, preset -4 fixed ratio with -l (147 filters x 988 taps, nearest filter, implicit low-pass), "
                                   "-p biquad cascade (2 low-pass sections per channel), 16-bit HP-TPDF + ATH-shaped decimation; END TO END per 1,048,576-frame block, device-resident"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=8, help="weak scaling: channels per GPU")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --channels per GPU, the stream has channels x N (default; N = 1 is BASELINE.json's metric config); "
                         "strong: ONE stream of --total-channels, total/N per GPU (BASELINE.json configs[3]: 32 channels, 4 per GPU at N = 8)")
    ap.add_argument("--total-channels", type=int, default=32, help="strong scaling: channels of the stream")
    ap.add_argument("--block-frames", type=int, default=1 << 20, help="input frames per call")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 general, 2 matrix path, 5 f32 tile kernel, 6 f32 streaming kernel, 7 fixed point wherever possible")
    ap.add_argument("--preroll-ms", type=float, default=200.0, help="untimed device pre-roll before the warmup steps (clock ramp); 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config_b / config_c / config_e legs (BASELINE.json configs[1], [2], [4]) behind the headline")
    ap.add_argument("--pmc-json", default=None, help="traffic file written by tools/roofline_report.py from the rocprofv3 --pmc passes of the same lease (tools/refresh_evidence.sh; else the committed profiles/ figure is reported, labelled as such)")
    args = ap.parse_args()

    import torch
    import audio_resampler_amd as A

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        args.gpus = world
    # one rank per GPU.  (ARTAMD_BENCH_BACKEND=gloo lets a multi-rank run be smoke-tested on a box with fewer GPUs
    # than ranks: ranks then share devices and the few scalars of the reduction travel over gloo.)
    backend = os.environ.get("ARTAMD_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and backend == "nccl":
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local_rank} but only {ndev} device(s) are visible")
    torch.cuda.set_device(local_rank % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            # gloo picks its interface by resolving the box's hostname, which a fresh container may not be able to do: one node,
            # so the loopback is the right device whatever the hostname says
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group(backend)

    from audio_resampler_amd.shard import agree_and_aggregate, channel_slice
    block = args.block_frames
    total_ch = args.total_channels if args.scaling == "strong" else args.channels * world
    lo, hi = channel_slice(total_ch, world, rank)
    Cn = hi - lo
    if Cn < 1:
        raise SystemExit(f"bench.py: {total_ch} channels cannot be shared among {world} ranks")
    ratio = DST / SRC
    cap = int(math.floor((block + TAPS // 2) * ratio + 10))

    # this rank's channel slice of the ONE stream
    d_in = torch.from_numpy(stream_slice(block, lo, hi)).cuda()
    d_out = torch.empty(cap, Cn, device="cuda", dtype=torch.float32)

    rs = A.Resampler(Cn, TAPS, FILTERS, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE)
    rs.advance(TAPS / 2.0)
    rs.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.kernel:
        rs.set_kernel(args.kernel)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(blk=None, steps=None):
        """W untimed warmup steps, then EXACTLY K steps between barrier + synchronize on both sides: the clean region `value` is taken
        from — no event is recorded inside it.  Behind it, between barriers of its own, the same K steps once more with HIP events around
        every launch's dominant kernel (on the kernel's own stream): the roofline's per-launch duration.  Events are not free — each is a
        packet in the queue between two kernels, three per launch: measured (tools/micro/host_rate.py) 117.5 against 123.1 us per headline
        call, 27.1 against 32.9 at 65,536 frames — which is why the two regions are separate; the instrumented region's own rate rides in
        the line as ms_per_step_instrumented."""
        blk = block if blk is None else blk
        steps = args.steps if steps is None else steps
        cap_b = int(math.floor((blk + TAPS // 2) * ratio + 10))
        for _ in range(args.warmup):
            used, made = rs.process_device(d_in, blk, d_out, cap_b, ratio)
            assert used == blk and made < cap_b
        barrier()
        frames = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            used, made = rs.process_device(d_in, blk, d_out, cap_b, ratio)
            frames += made
        barrier()
        dt = time.perf_counter() - t0
        rs.set_timing(True)
        t1 = time.perf_counter()
        for _ in range(steps):
            rs.process_device(d_in, blk, d_out, cap_b, ratio)
        barrier()
        dt_events = time.perf_counter() - t1
        kernel_ms, launches = rs.read_timing()
        prep_ms = rs.read_prep_timing()
        rs.set_timing(False)
        timed_region.instrumented_ms_per_step = dt_events / steps * 1e3
        return dt, frames, kernel_ms, launches, prep_ms

    # From cold first (reported as value_cold): MI355X raises its clocks over the first tens of milliseconds of sustained load —
    # measured on this workload: 0.213 ms per kernel in the first 5 ms, 0.179 ms after 100 ms — so a K of a few dozen 0.2 ms
    # steps from cold times the ramp, not the steady state the metric means.
    cold = timed_region()
    # Device pre-roll (NOT part of the W warmup steps, never timed), then the run `value` reports.  Disclosed in the JSON line
    # ("preroll_ms"); --preroll-ms 0 disables it (value == a second cold-ish run then).
    if args.preroll_ms > 0:
        t_pre = time.perf_counter()
        pre_calls = 0
        while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
            for _ in range(8):
                rs.process_device(d_in, block, d_out, cap, ratio)
            pre_calls += 8
            torch.cuda.synchronize()
        if dist is not None:
            # (the pre-roll is bounded by time, so ranks make different numbers of calls — and a call's frame count depends on the stream's phase
            # (1,048,576 x 160 / 147 is not whole): every rank goes on to the longest rank's count, so that all streams enter the timed region at the
            # same position and the ranks' frame counts agree whatever K is)
            n_pre = torch.tensor([pre_calls], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(n_pre, op=dist.ReduceOp.MAX)
            for _ in range(int(n_pre.item()) - pre_calls):
                rs.process_device(d_in, block, d_out, cap, ratio)
            torch.cuda.synchronize()
    dt, out_frames, kernel_ms, launches, prep_ms = timed_region()
    ms_instrumented = timed_region.instrumented_ms_per_step
    kernel_used = rs.last_kernel()
    fixed_state, fixed_pairs = rs.fixed_point()           # 1: the matrix path ran in fixed point on the integer matrix cores
    fixed_kernel_name = rs.fixed_point_kernel()           # which form of the fixed-point kernel (asked of the library, not inferred)

    # Beside the headline, in the same run and on the same context (never `value`):
    #  * the same call at the block size the CPU baseline uses (65,536 frames): the like-for-like figure;
    #  * the f32 matrix-core kernel pinned (kernel preference 6): exact-f32 FMA chains, the path of launches below the
    #    fixed-point kernel's threshold and its stand-by.
    small_block = min(65536, block)
    small = timed_region(small_block, max(args.steps, 50)) if not args.kernel else None
    f32 = None
    if not args.kernel and kernel_used == 2 and fixed_state == 1:
        rs.set_kernel(6)
        f32 = timed_region()
        rs.set_kernel(0)

    # BASELINE.json configs[3] in the same run, whatever --scaling says: ONE 32-channel stream (--total-channels) whose channels are
    # shared among the ranks, 32/N per GPU (4 per GPU at N = 8) — the STRONG-scaling figure beside the weak-scaling `value`
    # (reference fan-out: resampler.c:442-470, one worker per channel of one context).  Its own context and buffers, W warmup + K
    # timed steps between barriers like the headline; never `value`.
    config_d = None
    config_d_error = None
    dev = "cuda" if backend == "nccl" else "cpu"

    def all_ranks_ok(ok):
        """every rank enters the barriers of the config_d leg or none does: one rank's failure must not hang the others"""
        if dist is None:
            return ok
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item() == 1.0)

    if args.scaling == "weak" and not args.kernel and args.total_channels >= world:
        rs_d = d_in_d = d_out_d = None
        Cd = 0
        try:
            lo_d, hi_d = channel_slice(args.total_channels, world, rank)
            Cd = hi_d - lo_d
            d_in_d = torch.from_numpy(stream_slice(block, lo_d, hi_d)).cuda()
            d_out_d = torch.empty(cap, Cd, device="cuda", dtype=torch.float32)
            rs_d = A.Resampler(Cd, TAPS, FILTERS, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE)
            rs_d.advance(TAPS / 2.0)
            rs_d.set_stream(torch.cuda.current_stream().cuda_stream)
            for _ in range(args.warmup):
                used, made = rs_d.process_device(d_in_d, block, d_out_d, cap, ratio)
                assert used == block and 0 < made < cap, (used, made)
            torch.cuda.synchronize()
            ok = True
        except Exception as e:                       # reported in the line ("config_d_error"), never fatal to `value`
            config_d_error = f"rank {rank}: {type(e).__name__}: {e}"
            print(f"bench.py: config_d leg skipped: {config_d_error}", file=sys.stderr, flush=True)
            ok = False
        if all_ranks_ok(ok):
            barrier()
            frames_d = 0
            t0 = time.perf_counter()
            for _ in range(args.steps):
                used, made = rs_d.process_device(d_in_d, block, d_out_d, cap, ratio)
                assert used == block and 0 < made < cap, (used, made)
                frames_d += made
            barrier()
            dt_d = time.perf_counter() - t0
            rs_d.set_timing(True)                        # (the kernel's own time: the same steps again, with events — see timed_region)
            for _ in range(args.steps):
                rs_d.process_device(d_in_d, block, d_out_d, cap, ratio)
            barrier()
            k_ms_d, launches_d = rs_d.read_timing()
            rs_d.set_timing(False)
            config_d = (dt_d, frames_d, Cd, k_ms_d, launches_d, rs_d.fixed_point_kernel() or {1: "general", 2: "mfma (f32)"}.get(rs_d.last_kernel()))
        elif config_d_error is None:
            config_d_error = "another rank could not set the leg up"
        del rs_d, d_in_d, d_out_d

    dev = "cuda" if backend == "nccl" else "cpu"
    agg_d = agree_and_aggregate(dist, dev, config_d[0], config_d[1], config_d[2], config_d[3], config_d[4]) if config_d else None
    agg = agree_and_aggregate(dist, dev, dt, out_frames, Cn, kernel_ms, launches)
    agg_cold = agree_and_aggregate(dist, dev, cold[0], cold[1], Cn, cold[2], cold[3])
    agg_small = agree_and_aggregate(dist, dev, small[0], small[1], Cn, small[2], small[3]) if small else None
    agg_f32 = agree_and_aggregate(dist, dev, f32[0], f32[1], Cn, f32[2], f32[3]) if f32 else None
    dt_max, samples_total = agg["seconds_max"], agg["samples_total"]
    assert agg["frames_consistent"] and agg_cold["frames_consistent"], "ranks disagree on the number of generated frames"

    if rank == 0:
        # Roofline of the dominant kernel (the FIR), from HIP events recorded around its launches on its stream.
        # The matrix-core kernel folds the lerp into ONE effective row per phase (the lerp is linear), so per output sample it
        # EXECUTES 2 x Kpad flop on the matrix cores (Kpad = the tile's K columns: 1024 for T = 988; 2 x T of them useful) —
        # that, over the f32-MFMA peak, is `frac` (<= 1 by construction).  The reference FORMULATION (two T-tap dot products +
        # lerp = 4T+3 flop) at the same speed is reported separately and may exceed the peak.
        per_launch_samples = out_frames * Cn / max(launches, 1)
        avg_ms = kernel_ms / max(launches, 1)
        # K columns a tile walks: T + the spread of its rows' window starts + alignment, a whole number of 32-tap images — 1,024 for the 32-slot tiles
        # (f32 kernels, fir_i8_dma / _stream), 1,056 = 33 images for the 64-slot tiles of fir_i8_slab_kernel (the slab kernel had been priced at
        # 1,024, understating what it executes by 3 %)
        def k_columns(rows):
            return ((TAPS + int((rows - 1.0) * 147 / 160) + 2 + 3 + 31) // 32) * 32
        kpad_f32 = k_columns(32)
        kpad = k_columns(64) if fixed_kernel_name == "fir_i8_slab_kernel" else kpad_f32
        fixed = kernel_used == 2 and fixed_state == 1
        # fixed-point kernel: every (sample, K column) costs one integer multiply-add per digit pair issued (13, or 9 where the
        # rows' most significant digit plane is all zero: resampleHipLastFixedPoint reports the average)
        executed_per_sample = (2 * kpad * fixed_pairs if fixed else 2 * kpad) if kernel_used == 2 else FLOP_PER_SAMPLE
        peak = PEAK_I8_TOPS if fixed else PEAK_FP32_TFLOPS
        rate = per_launch_samples / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
        tflops_exec = rate * executed_per_sample / 1e12
        tflops_useful = rate * ((2 * TAPS * fixed_pairs if fixed else 2 * TAPS) if kernel_used == 2 else FLOP_PER_SAMPLE) / 1e12
        tflops_ref_form = rate * FLOP_PER_SAMPLE / 1e12
        gbs = rate * BYTES_PER_SAMPLE / 1e9
        # HBM traffic of the dominant kernel is a PMC measurement (tools/refresh_evidence.sh: separate rocprofv3 --pmc passes, FETCH_SIZE
        # with the gfx950 wide-read correction + WRITE_SIZE); it cannot be taken inside this process, so the committed
        # per-launch figure is reported when — and only when — this run is the workload it was measured on.
        traffic = traffic_source = None
        for name in ([args.pmc_json] if args.pmc_json else []) + ["r6_traffic.json", "r5_traffic.json", "r4_traffic.json"]:
            try:
                tr = json.load(open(name if os.path.isabs(name) or os.path.exists(name) else os.path.join(ROOT, "profiles", name)))
                w = tr["workload"]
                if kernel_used == 2 and bool(tr.get("fixed_point", False)) == fixed and (w["block_frames"], w["channels"], w["taps"]) == (block, Cn, TAPS) and launches == args.steps \
                        and (fixed_kernel_name or "fir_mfma") in tr.get("kernel", fixed_kernel_name or "fir_mfma"):
                    traffic = tr["traffic_bytes_per_launch"]
                    traffic_source = (f"--pmc-json {name}: rocprofv3 --pmc passes of this command in the same lease" if name == args.pmc_json else
                                      f"committed profiles/{name}: rocprofv3 --pmc passes of this command on an earlier box (not measured in this run)")
                break
            except Exception:
                continue
        mode = (f"weak scaling: {Cn} channels per GPU" if args.scaling == "weak"
                else f"STRONG scaling: one {total_ch}-channel stream, {total_ch}/{world} channels per GPU")
        line = {
            "metric": "Msamples/s (out) 44.1k->48k preset -4, 8ch float32",
            "value": round(samples_total / dt_max / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4),
            "ms_per_step_instrumented": round(ms_instrumented, 4),
            "instrumentation_note": "value / ms_per_step: K steps with NO event recorded inside the timed region; roofline.avg_kernel_ms / avg_prep_ms: the same K steps "
                                    "run again right behind it with HIP events around every launch's dominant kernel (three event packets per launch cost the queue "
                                    "4-6 us per call), at ms_per_step_instrumented",
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "preroll_ms": args.preroll_ms,
            "value_cold": round(agg_cold["samples_total"] / agg_cold["seconds_max"] / 1e6, 2),
            "value_cold_note": "the same W warmup + K timed steps run first, from cold clocks, before the untimed pre-roll",
            "dtype": "i8 x 4 digits (32-bit fixed point of the f32 samples and the fp64-blended filter rows, exact i32 accumulation, one float rounding per output)" if fixed else "f32",
            "data": "synthetic",
            "config": {"workload": f"{mode}; ranks take contiguous channel slices of ONE {total_ch}-channel stream; 44100->48000 Hz, "
                                   f"preset -4 = 988 filters x 988 taps Blackman-Harris interpolating (artest -4 -c8), float32 interleaved, "
                                   f"{block} input frames per call, device-resident in/out, resampleProcessInterleavedDevice",
                       "stream_channels": total_ch, "channels_per_gpu": Cn, "block_frames": block, "taps": TAPS, "filters": FILTERS,
                       "fir_kernel": "mfma-i8 (fixed point)" if fixed else {1: "general", 2: "mfma"}.get(kernel_used, str(kernel_used)),
                       "parallelism": f"channel-shard x{world}, no data-path collective",
                       "accuracy": ("default mode, fixed-point matrix kernel, block floating point (one power-of-two exponent per channel and EXPONENT "
                                    "BLOCK of ~9,400 input frames — 0.2 s — from the channel's peak |x| over the frames the block's outputs read): "
                                    "|y - fp64-accumulate| <= half a float ulp of y + 2^-26 x that peak — RELATIVE to the channel's level around the "
                                    "output like float arithmetic (rms error ~0.5 x the reference float loop's at any amplitude), not an absolute "
                                    "grid; inside an exponent block a passage far below the channel's peak keeps the peak's grid (floor 2^-31 x "
                                    "peak); RESAMPLE_STRICT_ORDER is bit-exact") if fixed else
                                   ("default mode: |y - fp64-accumulate| <= 2^-23 max(1,|y|) (2 ulp on < 0.1 % of samples when |y| > 1, as the "
                                    "reference's own float build); RESAMPLE_STRICT_ORDER is bit-exact")},
            "roofline": {"bound": "mfma", "achieved": round(tflops_exec, 3), "peak": peak, "unit": "TFLOP/s",
                         "unit_note": "integer multiply-adds of v_mfma_i32_32x32x32_i8, 2 ops each (TOP/s), against the dense int8 MFMA peak" if fixed else "f32 MFMA",
                         "frac": round(tflops_exec / peak, 4),
                         "frac_of_sustained_live_peak": round(tflops_exec / SUSTAINED_LIVE_I8_TOPS, 4) if fixed else None,
                         "frac_at_round3_products": round(rate * 2 * kpad_f32 * 9.5 / 1e12 / peak, 4) if fixed else None,
                         "frac_at_round3_products_note": ("the same samples/s priced at round 3's 9.5 products per chunk (this round skips the products with the "
                                                          "rows' second digit plane where it is all zero: fewer operations EXECUTED, so `frac` does not rise "
                                                          "with the speed-up; this figure is the one comparable with rounds 2-3)") if fixed else None,
                         "sustained_live_peak": SUSTAINED_LIVE_I8_TOPS if fixed else None,
                         "sustained_live_peak_note": ("v_mfma_i32_32x32x32_i8 alone, operands in registers, pseudo-random bytes, 0.6 s: 3,030-3,190 TOP/s at "
                                                      "1.70-1.93 GHz (zero operands: 4,530 at 2.39 GHz) — tools/micro/mfma_sustain.hip, profiles/r4_mfma_sustain.txt: "
                                                      "the chip's power management bounds the integer matrix cores on live data well below the nominal dense peak") if fixed else None,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_unit": "bytes/launch (HBM, PMC)", "algorithmic_bytes_per_launch": int(per_launch_samples * BYTES_PER_SAMPLE),
                         "kernel": fixed_kernel_name if fixed else "fir_mfma_stream_kernel", "avg_kernel_ms": round(avg_ms, 4), "launches": launches,
                         "avg_prep_ms": round(prep_ms / max(launches, 1), 4),
                         "prep_note": "HIP events: what each launch spends before its dominant kernel (fixed point: peak pass + digit-plane staging pass; f32: row-table pass)",
                         "flop_per_sample_executed": round(executed_per_sample, 1), "bytes_per_sample": round(BYTES_PER_SAMPLE, 3),
                         "digit_pairs_per_chunk": round(fixed_pairs, 3) if fixed else None,
                         "f32_mfma_equivalent_frac": round(rate * 2 * kpad_f32 / 1e12 / PEAK_FP32_TFLOPS, 4) if kernel_used == 2 else None,
                         "useful_frac": round(tflops_useful / peak, 4),
                         "algorithmic_vs_reference_formulation": round(tflops_ref_form / PEAK_FP32_TFLOPS, 4),
                         "note": "achieved/frac = operations the kernel EXECUTES on the matrix cores (2 x Kpad per sample, lerp folded into one "
                                 "row per phase; x digit pairs issued for the fixed-point kernel) over that instruction's dense peak; "
                                 "f32_mfma_equivalent_frac prices the same samples/s as the f32 kernel's 2 x Kpad flop over the f32-MFMA peak "
                                 "(the figure of earlier rounds; not a roofline fraction for the integer kernel); "
                                 "useful_frac counts only the 2 x T non-padding columns; "
                                 "algorithmic_vs_reference_formulation prices the same samples/s at the reference's 4T+3 flop per "
                                 "sample (SURVEY 8d) and is not a roofline fraction",
                         "hbm_algorithmic_GBps": round(gbs, 2), "hbm_frac": round(gbs / PEAK_HBM_GBS, 5)},
        }
        if agg_small:
            line["value_block65536"] = round(agg_small["samples_total"] / agg_small["seconds_max"] / 1e6, 2)
            line["value_block65536_note"] = (f"the same call with {small_block} input frames (the CPU baseline's block size), {max(args.steps, 50)} timed steps "
                                             "after the headline's, same context; the like-for-like figure beside cpu_baseline")
        if agg_f32:
            f_launches = max(f32[3], 1)
            f_rate = f32[1] * Cn / f_launches / (f32[2] / f_launches * 1e-3)
            line["value_f32"] = round(agg_f32["samples_total"] / agg_f32["seconds_max"] / 1e6, 2)
            line["f32_frac"] = round(f_rate * 2 * kpad_f32 / 1e12 / PEAK_FP32_TFLOPS, 4)
            line["f32_avg_kernel_ms"] = round(f32[2] / f_launches, 4)
            line["value_f32_note"] = ("same W + K steps with the f32 matrix-core kernel pinned (kernel preference 6: exact-f32 FMA chains with fp64 flushes — "
                                      "what launches below the fixed-point threshold, and the fixed-point kernel's stand-by, run); f32_frac = executed "
                                      "2 x Kpad flop per sample over the f32-MFMA peak")
        if agg_d:
            line["config_d"] = {"value": round(agg_d["samples_total"] / agg_d["seconds_max"] / 1e6, 2), "unit": "Msamples/s", "scaling": "strong",
                                "stream_channels": args.total_channels, "channels_per_gpu": config_d[2], "n_gpus": world,
                                "ms_per_step": round(agg_d["seconds_max"] / args.steps * 1e3, 4), "fir_kernel": config_d[5],
                                "frames_consistent": bool(agg_d["frames_consistent"]),
                                "workload": f"BASELINE.json configs[3]: ONE {args.total_channels}-channel 44.1k->48k preset -4 stream, its channels shared among "
                                            f"the {world} rank(s) ({config_d[2]} per GPU here; 4 per GPU at N = 8), {block} input frames per call, same W + K steps "
                                            "between barriers as the headline; the strong-scaling figure (total work fixed as N grows), never `value`"}
        if config_d_error:
            line["config_d_error"] = config_d_error
        if world == 1 and not args.kernel and not args.no_other_configs:
            try:
                line.update(other_configs(A, torch, args.steps, args.warmup))
            except Exception as e:                           # (never fatal to the headline; the reason rides in the line)
                line["other_configs_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(Cn)
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        # a rank that dies under torch.distributed.run leaves only the launcher's summary behind unless it says why itself: the
        # traceback goes to BOTH streams, tagged with the rank, so that whoever captured either can read the cause
        import traceback
        text = f"bench.py: rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')} FAILED\n" + traceback.format_exc()
        for stream in (sys.stderr, sys.stdout):
            print("\n".join("bench.py[FAILED] " + l for l in text.splitlines()), file=stream, flush=True)
        raise
