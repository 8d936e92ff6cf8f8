"""Time stretcher throughput: libartamd.so (one persistent workgroup per stream) next to the real reference on one CPU
core (oracle/_ref/libartref_make.so when present, else the oracle port).  The stretcher is a chain of decisions per stream
("unrelated effect", mono/stereo only): the GPU figure that scales is streams in flight, so N independent
contexts are driven round-robin on N HIP streams.  One JSON line per case."""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import audio_resampler_amd as A
import _oracle, _stretch as S

L = A.lib()
rate, secs, blk = 44100, 4.0, 16384


def cpu_rate(ch, flags, ratio, x):
    B = _oracle.binding(32)
    try:
        lib = B.load_ref("make"); cls, kind = S.RefStretch, "reference"
    except OSError:
        cls, kind = S.OracleStretch, "port"
    st = cls(rate // 350, rate // 50, ch, flags)
    if kind == "reference":
        st.L = lib; st._bind(lib, B.f32p); st.p = lib.stretchInit(rate // 350, rate // 50, ch, flags)
    t0 = time.perf_counter(); st.run(x, [blk], [ratio]); dt = time.perf_counter() - t0
    return x.shape[0] / rate / dt, kind


for ch, flags, ratio, name in ((1, 0, 1.25, "mono x1.25"), (2, 0, 0.8, "stereo x0.8"), (2, S.FAST, 1.6, "stereo fast x1.6"), (2, S.DUAL, 3.0, "stereo dual x3.0")):
    x = S.signal(int(rate * secs), ch, rate, seed=3)
    cpu, kind = cpu_rate(ch, flags, ratio, x)
    row = {"case": name, "cpu_realtime_factor": round(cpu, 1), "cpu_kind": kind}
    for nstreams in (1, 16, 64):
        ctxs = [S.HipStretch(rate // 350, rate // 50, ch, flags) for _ in range(nstreams)]
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        cap = ctxs[0].capacity(blk, ratio)
        d_in = torch.from_numpy(x).cuda()
        d_outs = [torch.empty(cap, ch, device="cuda") for _ in range(nstreams)]
        for c, s in zip(ctxs, streams):
            L.stretchHipSetStream(c.p, s.cuda_stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if nstreams == 1:
            for pos in range(0, x.shape[0], blk):
                n = min(blk, x.shape[0] - pos)
                L.stretchProcessDevice(ctxs[0].p, d_in[pos:].data_ptr(), n, d_outs[0].data_ptr(), ratio)
        else:
            # the synchronous entry point serialises the host; independent streams are driven from threads
            import threading
            def work(i):
                for pos in range(0, x.shape[0], blk):
                    n = min(blk, x.shape[0] - pos)
                    L.stretchProcessDevice(ctxs[i].p, d_in[pos:].data_ptr(), n, d_outs[i].data_ptr(), ratio)
            th = [threading.Thread(target=work, args=(i,)) for i in range(nstreams)]
            [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        row[f"gpu_realtime_factor_{nstreams}_streams"] = round(nstreams * secs / dt, 1)
    # the batched entry point: one launch per round for all streams, one workgroup per stream
    import ctypes as C
    for nstreams in (64, 512):
        ctxs = [S.HipStretch(rate // 350, rate // 50, ch, flags) for _ in range(nstreams)]
        cap = ctxs[0].capacity(blk, ratio)
        d_in = torch.from_numpy(x).cuda()
        d_outs = [torch.empty(cap, ch, device="cuda") for _ in range(nstreams)]
        ctx = (C.c_void_p * nstreams)(*[c.p for c in ctxs]); outs = (C.c_void_p * nstreams)(*[d.data_ptr() for d in d_outs])
        made = (C.c_int * nstreams)(); rat = (C.c_double * nstreams)(*([ratio] * nstreams))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for pos in range(0, x.shape[0], blk):
            n = min(blk, x.shape[0] - pos)
            ins = (C.c_void_p * nstreams)(*([d_in[pos:].data_ptr()] * nstreams)); fr = (C.c_int * nstreams)(*([n] * nstreams))
            assert L.stretchProcessBatchDevice(ctx, nstreams, ins, fr, outs, rat, made) == 0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        row[f"gpu_realtime_factor_batched_{nstreams}_streams"] = round(nstreams * secs / dt, 1)
    print(json.dumps(row), flush=True)
