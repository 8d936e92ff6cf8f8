"""Is a fixed-ratio stream's output on the f32 streaming kernel (preference 6, rows kept across calls) independent of how the input is cut into calls?
usage: python tools/micro/cut_invariance.py [kernel=6] [channels=2] [taps=380]"""
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import audio_resampler_amd as A
from audio_resampler_amd.synth import noise
kernel = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
taps = int(sys.argv[3]) if len(sys.argv) > 3 else 380
total = 200000 if os.environ.get('CUT_SMALL') else 600000
x, _ = noise(total * ch); d_in = torch.from_numpy(x.reshape(total, ch)).cuda()
ratio = 48000 / 44100
def run(cuts, fixed, host=False):
    flags = A.BLACKMAN_HARRIS | A.SUBSAMPLE_INTERPOLATE
    rs = A.Resampler(ch, taps, taps, 0.0, flags, fixed=(44100, 48000, 0) if fixed else None)
    rs.advance(taps / 2.0)
    if kernel: rs.set_kernel(kernel)
    outs, pos, kernels = [], 0, set()
    for n in cuts:
        cap = int(n * ratio) + 4000
        d_out = torch.zeros(cap, ch, device="cuda")
        if host:
            u, g, y = rs.process(x.reshape(total, ch)[pos:pos + n], cap, 0.0 if fixed else ratio)
            assert u == n
            kernels.add(rs.last_kernel()); outs.append(np.array(y).copy()); pos += n
            continue
        u, g = rs.process_device(d_in[pos:pos + n], n, d_out, cap, 0.0 if fixed else ratio)
        assert u == n
        kernels.add(rs.last_kernel())
        outs.append(d_out[:g].cpu().numpy().copy()); pos += n
    return np.concatenate(outs), kernels
rng = np.random.default_rng(7)
def cuts_of(kind):
    if kind == "one": return [total]
    if kind == "65536": c = [65536] * (total // 65536); return c + [total - sum(c)]
    if kind == "16384": c = [16384] * (total // 16384); return c + [total - sum(c)]
    if kind == "4096": c = [4096] * (total // 4096); return c + [total - sum(c)]
    if kind in ("1000", "256"): k = int(kind); c = [k] * (total // k); return c + ([total - sum(c)] if total - sum(c) else [])
    if kind == "host-small":
        c = []
        while sum(c) < total: c.append(int(min(rng.integers(1, 3000), total - sum(c))))
        return c
    c = []
    while sum(c) < total: c.append(int(min((rng.integers(3000, 120000) // 4 * 4 if kind == "random4" else rng.integers(3000, 120000)), total - sum(c))))
    return c
for fixed in (False, True):
    ref = None
    for kind in (("one", "1000", "256", "host-small") if os.environ.get("CUT_SMALL") else ("one", "65536", "16384", "4096", "random4", "random", "host-random")):
        y, ks = run(cuts_of("random" if kind == "host-random" else kind), fixed, host=kind.startswith("host"))
        h = hashlib.sha256(y.tobytes()).hexdigest()[:16]
        if ref is None: ref = y
        same = y.shape == ref.shape and np.array_equal(y.view(np.uint32), ref.view(np.uint32))
        diff = int(np.count_nonzero(y.view(np.uint32) != ref.view(np.uint32))) if y.shape == ref.shape else -1
        print(f"kernel pref {kernel} ch {ch} taps {taps} fixed-ratio-init {int(fixed)} cuts {kind:7s}: frames {y.shape[0]} sha {h} kernels used {sorted(ks)}  same bits as one call: {same} (differing samples {diff})")
