"""Many small-block streams on one GPU: N independent stereo resampler contexts (44.1 -> 48 kHz, preset -3 = 380 taps, ratio
drifting per call like an ASRC), 10 ms blocks, device-resident.  One call per context per tick
(resampleProcessInterleavedDevice in a loop) next to one batched call per tick (resampleProcessBatchInterleavedDevice).
Prints one JSON line per N: ticks/s, aggregate Msamples/s and how many real-time streams that sustains."""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
import audio_resampler_amd as A
B = A.binding(32)
src, dst, ch, T, block = 44100, 48000, 2, 380, 441                 # 441 frames = 10 ms at 44.1 kHz
ticks = 200
for N in (16, 128, 1024):
    rs = [B.Resampler(ch, T, T, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE) for _ in range(N)]
    for r in rs: r.advance(T / 2)
    x = torch.from_numpy((np.random.default_rng(1).random((block, ch)) - 0.5).astype(np.float32)).cuda()
    cap = int(block * dst / src * 1.01) + 16
    outs = [torch.empty(cap, ch, device="cuda") for _ in range(N)]
    ratios = [dst / src * (1 + 1e-5 * ((i * 7) % 11 - 5)) for i in range(N)]
    row = {"streams": N, "block_frames": block, "taps": T, "channels": ch}
    # argument arrays built once (the pointers do not change from tick to tick): the loop below times the library, not ctypes
    import ctypes as C
    L = B.lib()
    ctx = (C.c_void_p * N)(*[C.cast(r.p, C.c_void_p) for r in rs])
    ins = (C.c_void_p * N)(*([x.data_ptr()] * N)); outp = (C.c_void_p * N)(*[o.data_ptr() for o in outs])
    nin = (C.c_int * N)(*([block] * N)); caps = (C.c_int * N)(*([cap] * N)); rat = (C.c_double * N)(*ratios)
    res = (B.ResampleResult * N)()
    single_args = [(r.p, x.data_ptr(), block, outs[i].data_ptr(), cap, ratios[i]) for i, r in enumerate(rs)]
    for mode in ("single", "batched"):
        def tick():
            if mode == "single":
                g = 0
                for a in single_args: g += L.resampleProcessInterleavedDevice(*a).output_generated
                return g
            assert L.resampleProcessBatchInterleavedDevice(ctx, N, ins, nin, outp, caps, rat, res) == 0
            return sum(r.output_generated for r in res)
        for warm in range(3): tick()
        torch.cuda.synchronize(); t0 = time.perf_counter(); gen = 0
        nt = ticks if N < 1024 or mode == "batched" else 20
        for t in range(nt): gen += tick()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        row[mode + "_ms_per_tick"] = round(dt / nt * 1e3, 3)
        row[mode + "_Msamples_per_s"] = round(gen * ch / dt / 1e6, 2)
        row[mode + "_realtime_streams"] = int(gen / dt / dst)
    print(json.dumps(row), flush=True)
    for r in rs: r.close()
