"""General kernel against the matrix path by call size, per shape: pipelined step time (calls enqueued back to back) and the time of a
synchronised call, with what the dispatch rule chooses — to re-fit artfir_takes_matrix_path's cost model.  GPU box."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
ratio = 48000 / 44100
shapes = [(8, 988), (2, 380), (2, 988), (1, 48), (16, 156), (4, 988), (32, 988), (8, 380), (1, 988)] if len(sys.argv) < 2 else [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
for (C, T) in shapes:
    for block in (1024, 2048, 2896, 4096, 5793, 8192, 11585, 16384, 23170, 32768, 65536):
        row = []
        for kern in (1, 6, 0):
            rs = A.Resampler(C, T, T, 0.0, A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE); rs.advance(T / 2)
            if kern: rs.set_kernel(kern)
            rs.set_stream(torch.cuda.current_stream().cuda_stream)
            x, _ = noise(block * C); d_in = torch.from_numpy(x.reshape(block, C)).cuda()
            cap = int(math.floor((block + T // 2) * ratio + 10)); d_out = torch.empty(cap, C, device="cuda")
            for _ in range(20): rs.process_device(d_in, block, d_out, cap, ratio)
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for _ in range(n): rs.process_device(d_in, block, d_out, cap, ratio)
            torch.cuda.synchronize()
            piped = (time.perf_counter() - t0) / n * 1e6
            t0 = time.perf_counter()
            for _ in range(50):
                rs.process_device(d_in, block, d_out, cap, ratio); torch.cuda.synchronize()
            synced = (time.perf_counter() - t0) / 50 * 1e6
            row += [piped, synced, rs.last_kernel()]
        print(f"C={C} T={T} block {block:6d}: general {row[0]:6.1f} / {row[1]:6.1f} us   matrix {row[3]:6.1f} / {row[4]:6.1f} us   library's choice: kernel {row[8]} {row[6]:6.1f} / {row[7]:6.1f}   best piped: {'matrix' if row[3] < row[0] else 'general'}, synced: {'matrix' if row[4] < row[1] else 'general'}", flush=True)
