"""GPU (-m gpu): BASELINE.json configs[3] — a 32-channel 44.1k -> 48k preset -4 stream whose channels shard 4 per GPU — and the
multi-device context behind the C ABI (RESAMPLE_MULTITHREADED, reference resampler.c:185-186, :442-470).

On the one-GPU box every shard lives on device 0 (ARTAMD_SHARDS=8 forces the shard count): the sharding logic — channel
slices, per-shard de-interleaving on the way into HBM, per-shard streams, the position mirrored into the parent — is what is
under test; which device a shard sits on does not change a sample.
"""
import os

import numpy as np
import pytest
import torch

import _golden as G
import audio_resampler_amd as A
from _hip import HipResampler, tolerance_ok
from _oracle import OracleResampler, noise, BH, INTERP, PRECISE, EXTRAP

pytestmark = pytest.mark.gpu

STRICT = A.RESAMPLE_STRICT_ORDER
MT = A.RESAMPLE_MULTITHREADED
R = 48000 / 44100
T = 988


@pytest.fixture
def eight_shards(monkeypatch):
    monkeypatch.setenv("ARTAMD_SHARDS", "8")


def _stream(ch, frames, seed_skip=0):
    x, _ = noise(ch * (frames + seed_skip))
    return np.ascontiguousarray(x[ch * seed_skip:].reshape(frames, ch))


def _run(r, x, blocks, cap):
    outs, pos = [], 0
    for n in blocks:
        u, g, y = r.process(x[pos:pos + n], cap, R)
        assert u == n
        outs.append(y.copy())
        pos += n
    u, g, y = r.process(None, cap, R, flush=True)
    outs.append(y.copy())
    return np.concatenate(outs)


@pytest.mark.parametrize("mode", ["strict", "general", "mfma", "fixed_point"])
def test_eight_four_channel_contexts_equal_one_32_channel_context(mode):
    """slice-then-resample == resample-then-slice for the PRODUCT: 8 HIP contexts of 4 channels (what 8 GPUs would run) against
    one 32-channel HIP context, bit for bit — strict order, and each fast kernel with itself (6: the f32 matrix kernels — left alone
    the 32-channel calls are big enough for the fixed-point kernel, the 4-channel ones are not; 7: the fixed-point kernel on both
    sides, whose exact sums do not depend on how channels are grouped into tiles)"""
    flags, kernel = (BH | INTERP | STRICT, 0) if mode == "strict" else (BH | INTERP, {"general": 1, "mfma": 6, "fixed_point": 7} [mode])
    frames = 6000 if mode == "strict" else 40000
    x = _stream(32, frames)
    blocks = [frames // 3, frames - frames // 3]
    cap = int(frames * R) + 2000
    whole = HipResampler(32, T, T, 0.0, flags, kernel=kernel)
    whole.advance(T / 2)
    y = _run(whole, x, blocks, cap)
    if mode != "strict":
        assert whole.last_kernel() in (1, 2)
    for s in range(8):
        part = HipResampler(4, T, T, 0.0, flags, kernel=kernel)
        part.advance(T / 2)
        ys = _run(part, np.ascontiguousarray(x[:, 4 * s:4 * s + 4]), blocks, cap)
        assert ys.shape == (y.shape[0], 4)
        assert np.array_equal(ys.view(np.uint32), y[:, 4 * s:4 * s + 4].view(np.uint32)), (mode, s)


def test_config_d_at_full_block_size_matches_the_oracle_on_a_shard():
    """the 4-channel shard of configs[3] on the matrix-core path at a bench-sized block, every sample against the
    double-accumulate oracle (the CG = 4 instantiation at T = 988)"""
    ch, frames = 4, 150000
    x = _stream(ch, frames)
    r = HipResampler(ch, T, T, 0.0, BH | INTERP)
    o = OracleResampler(ch, T, T, 0.0, BH | INTERP | PRECISE)
    for b in (r, o):
        b.advance(T / 2)
    cap = int(frames * R) + 2000
    u, g, y = r.process(x, cap, R)
    uo, go, yo = o.process(x, cap, R, threads=4)
    assert (u, g) == (uo, go) and r.last_kernel() == 2
    ok, worst, rms = tolerance_ok(y, yo)
    assert ok and rms < 2.0e-8, (worst, rms)


def _pair(ch, flags, fixed=None, F=T, taps=T):
    """(ordinary context, RESAMPLE_MULTITHREADED context) with the same parameters"""
    if fixed is None:
        return (A.Resampler(ch, taps, F, 0.0, flags), A.Resampler(ch, taps, F, 0.0, flags | MT))
    return (A.Resampler(ch, taps, F, flags=flags, fixed=fixed), A.Resampler(ch, taps, F, flags=flags | MT, fixed=fixed))


def test_multithreaded_flag_alone_changes_nothing_on_one_device(monkeypatch):
    monkeypatch.delenv("ARTAMD_SHARDS", raising=False)
    monkeypatch.delenv("ARTAMD_DEVICES", raising=False)
    if A.lib().artamdDeviceCount() > 1:
        pytest.skip("several devices visible: the flag shards for real")
    r = A.Resampler(8, 48, 48, 0.0, BH | INTERP | MT)
    assert r.shards() == []


def test_sharded_context_layout(eight_shards):
    r = A.Resampler(32, 48, 48, 0.0, BH | INTERP | MT)
    assert [(f, n) for _, f, n in r.shards()] == [(4 * s, 4) for s in range(8)]       # configs[3]: 4 channels per GPU
    r7 = A.Resampler(7, 48, 48, 0.0, BH | INTERP | MT)                               # more shards than channels allow: clamped
    assert [(f, n) for _, f, n in r7.shards()] == [(s, 1) for s in range(7)]
    r10 = A.Resampler(10, 48, 48, 0.0, BH | INTERP | MT)
    assert [(f, n) for _, f, n in r10.shards()] == [(0, 2), (2, 2), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1), (9, 1)]
    import ctypes
    assert A.lib().artamdSetDevices((ctypes.c_int * 1)(99), 1) == -1
    assert A.lib().artamdSetDevices(None, 0) == 0


@pytest.mark.parametrize("strict", [True, False])
def test_sharded_context_host_api_equals_ordinary_context(eight_shards, strict):
    """art -c32 -m on this library: resampleProcessInterleaved / resampleProcess on ONE context whose channels run as 8 shards"""
    ch, frames = 32, 9000
    flags = BH | INTERP | (STRICT if strict else 0)
    x = _stream(ch, frames)
    plain, sharded = _pair(ch, flags)
    assert len(sharded.shards()) == 8 and plain.shards() == []
    for r in (plain, sharded):
        r.advance(T / 2)
    cap = 6000
    pos = 0
    for n in (4096, 1, 3000, 1903):
        ua, ga, ya = plain.process(x[pos:pos + n], cap, R)
        ub, gb, yb = sharded.process(x[pos:pos + n], cap, R)
        assert (ua, ga) == (ub, gb) and plain.state() == (sharded.state()[0], sharded.state()[1], sharded.state()[2] & ~MT)
        assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
        assert plain.position() == sharded.position()
        pos += ua
    # planar entry point, then the flush through the planar AndFlush form
    planes = [np.ascontiguousarray(x[:500, c]) for c in range(ch)]
    ua, ga, pa = plain.process_planar(planes, cap, R, and_flush=True)
    ub, gb, pb = sharded.process_planar(planes, cap, R, and_flush=True)
    assert (ua, ga) == (ub, gb) and ga > 500
    for c in range(ch):
        assert np.array_equal(pa[c].view(np.uint32), pb[c].view(np.uint32))
    # reset re-arms both
    plain.reset(), sharded.reset()
    ua, ga, ya = plain.process(x[:2000], cap, R)
    ub, gb, yb = sharded.process(x[:2000], cap, R)
    assert (ua, ga) == (ub, gb) and np.array_equal(ya.view(np.uint32), yb.view(np.uint32))


@pytest.mark.parametrize("stage_limit", [None, "0"])
def test_sharded_context_fixed_ratio_extrapolation_and_large_blocks(eight_shards, monkeypatch, stage_limit):
    """the ART form (fixed ratio, SNAP, low-pass, end-point extrapolation) on 10 channels (uneven shards), and a block large
    enough to bypass the page-locked staging (strided 2-D copies straight from the caller's buffers; ARTAMD_STAGE_LIMIT=0
    forces that path for every call and every shard)"""
    if stage_limit is not None:
        monkeypatch.setenv("ARTAMD_STAGE_LIMIT", stage_limit)
    ch = 10
    x = _stream(ch, 300000)
    plain, sharded = _pair(ch, BH | INTERP | A.INCLUDE_LOWPASS | EXTRAP | STRICT, fixed=(96000., 44100., 0), F=320, taps=156)
    for r in (plain, sharded):
        r.advance(78.0)
    cap = 200000
    pos = 0
    for n in (40, 3000, 280000):            # 280000 x 10 ch x 4 B = 11 MB > the 8 MB staging limit
        ua, ga, ya = plain.process(x[pos:pos + n], cap, 1.0)
        ub, gb, yb = sharded.process(x[pos:pos + n], cap, 1.0)
        assert (ua, ga) == (ub, gb)
        assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
        pos += ua
    ua, ga, ya = plain.process(None, cap, 1.0, flush=True)
    ub, gb, yb = sharded.process(None, cap, 1.0, flush=True)
    assert (ua, ga) == (ub, gb) and np.array_equal(ya.view(np.uint32), yb.view(np.uint32))


def test_sharded_context_device_pointer_calls(eight_shards):
    """device-resident 32-channel buffers on the context's device: every shard pulls its strided slice, runs on its own stream,
    pushes its slice back; the context's stream is ordered before and after"""
    ch, frames = 32, 50000
    x = _stream(ch, frames)
    plain, sharded = _pair(ch, BH | INTERP)
    for r in (plain, sharded):
        r.advance(T / 2)      # (kernel choice left to the library: a shard decides as its whole stream would — here the fixed-point kernel on both sides)
    cap = int(frames * R) + 2000
    d_in = torch.from_numpy(x).cuda()
    outs = []
    for r in (plain, sharded):
        d_out = torch.zeros(cap, ch, device="cuda")
        u, g = r.process_device(d_in, frames, d_out, cap, R, and_flush=True)
        r.synchronize()
        torch.cuda.synchronize()
        outs.append((u, g, d_out[:g].cpu().numpy()))
    assert outs[0][:2] == outs[1][:2]
    assert np.array_equal(outs[0][2].view(np.uint32), outs[1][2].view(np.uint32))
    # planar device buffers: a shard's channels are a contiguous run of planes, no copies
    plain, sharded = _pair(ch, BH | INTERP | STRICT)
    pl_in = torch.from_numpy(np.ascontiguousarray(x[:3000].T)).cuda()
    res = []
    for r in (plain, sharded):
        r.advance(T / 2)
        pl_out = torch.zeros(ch, 4000, device="cuda")
        u, g = r.process_planar_device(pl_in, 3000, 3000, pl_out, 4000, 4000, R)
        r.synchronize()
        torch.cuda.synchronize()
        res.append((u, g, pl_out[:, :g].cpu().numpy()))
    assert res[0][:2] == res[1][:2] and res[0][1] > 0
    assert np.array_equal(res[0][2].view(np.uint32), res[1][2].view(np.uint32))


def test_sharded_context_matches_golden_d_32ch(eight_shards):
    """configs[3] golden (reference-generated) replayed on a sharded context, strict order: every bit"""
    name = "D_32ch_988"
    r = G.make(HipResampler, name, STRICT | MT)
    assert len(r.shards()) == 8
    y, trace = G.replay(r, name)
    full, head, tail, csum = G.expected(name, "strict")
    assert np.array_equal(trace[:, :4], G.load("resample")[name + "/trace"][:, :4])
    assert np.array_equal(y[:256].view(np.uint32), head.view(np.uint32))
    assert np.array_equal(y[-256:].view(np.uint32), tail.view(np.uint32))
    from _oracle import checksum_words
    assert checksum_words(y) == csum


def test_context_uses_its_own_device_whatever_the_caller_selected():
    """contexts remember their device (here: the only one) and restore the caller's afterwards"""
    r = A.Resampler(2, 48, 48, 0.0, BH | INTERP)
    assert A.lib().resampleHipGetDevice(r.p) == torch.cuda.current_device()
    x = _stream(2, 500)
    u, g, y = r.process(x, 2000, R)
    assert u == 500 and g > 0 and torch.cuda.current_device() == A.lib().resampleHipGetDevice(r.p)


def test_sharded_context_runs_its_shards_in_fixed_point_with_the_same_bits(eight_shards):
    """a 32-channel RESAMPLE_MULTITHREADED context whose 4-channel shards are each big enough for the fixed-point matrix kernel
    (kernel preference 7 on both sides): exact integer sums do not care how the channels are grouped, so the sharded context —
    eight leaves, eight streams, eight sets of digit planes — equals the ordinary one bit for bit, host-pointer calls and
    device-pointer calls alike"""
    ch, frames = 32, 120000
    x = _stream(ch, frames)
    plain, sharded = _pair(ch, BH | INTERP)
    assert len(sharded.shards()) == 8
    cap = int(frames * R) + 2000
    outs = []
    for r in (plain, sharded):
        r.advance(T / 2)
        r.set_kernel(7)
        u, g, y = r.process(x, cap, R)
        assert u == frames and r.last_kernel() == 2 and r.fixed_point() [0] == 1
        d_in, d_out = torch.from_numpy(x).cuda(), torch.zeros(cap, ch, device="cuda")
        u2, g2 = r.process_device(d_in, frames, d_out, cap, R)
        r.synchronize(); torch.cuda.synchronize()
        outs.append((g, np.array(y).copy(), g2, d_out [:g2].cpu().numpy()))
    assert outs [0] [0] == outs [1] [0] and outs [0] [2] == outs [1] [2]
    assert np.array_equal(outs [0] [1].view(np.uint32), outs [1] [1].view(np.uint32))
    assert np.array_equal(outs [0] [3].view(np.uint32), outs [1] [3].view(np.uint32))


def test_sharded_and_ordinary_context_choose_the_same_kernel_at_every_call_size(eight_shards):
    """kernel choice left to the library: a 4-channel shard decides as its 32-channel stream would on one device (general kernel
    for small calls, f32 matrix kernel for middling ones, fixed point for big ones), so the two contexts agree bit for bit at
    every size — the default mode, no kernel pinned"""
    ch = 32
    sizes = [300, 1200, 2500, 9000, 30000, 61000]
    x = _stream(ch, sum(sizes))
    plain, sharded = _pair(ch, BH | INTERP)
    kinds = []
    pos = 0
    for r in (plain, sharded):
        r.advance(T / 2)
    for n in sizes:
        cap = int(n * R) + 2000
        a = plain.process(x [pos:pos + n], cap, R); ka = (plain.last_kernel(), plain.fixed_point() [0])
        b = sharded.process(x [pos:pos + n], cap, R); kb = (sharded.last_kernel(), sharded.fixed_point() [0])
        pos += n
        assert a [:2] == b [:2] and ka == kb, (n, ka, kb)
        assert np.array_equal(np.array(a [2]).view(np.uint32), np.array(b [2]).view(np.uint32)), (n, ka)
        kinds.append(ka)
    assert (1, 0) in kinds and (2, 0) in kinds and (2, 1) in kinds, kinds      # every kind of kernel was exercised


def test_worker_threads_enqueue_the_shards_with_the_same_result(monkeypatch):
    """on several devices every shard's launch sequence is enqueued by a worker thread of its own (resampler_host.c, shard_pool); on the
    one-GPU box ARTAMD_SHARD_THREADS=1 forces the threads although the shards share the device: the same bits, counts and positions as
    the calling thread's own enqueues, call after call, and a clean shutdown"""
    monkeypatch.setenv("ARTAMD_SHARDS", "8")
    ch = 32
    sizes = [300, 2500, 9000, 30000, 61000, 5, 0, 1200]
    x = _stream(ch, sum(sizes))
    outs = {}
    for threads in ("0", "1"):
        monkeypatch.setenv("ARTAMD_SHARD_THREADS", threads)
        r = HipResampler(ch, T, T, 0.0, BH | INTERP | MT); r.advance(T / 2)
        pos, ys = 0, []
        for n in sizes:
            u, g, y = r.process(x [pos:pos + n], int(n * R) + 2000, R)
            pos += u
            ys.append((u, g, np.array(y).view(np.uint32).copy()) + tuple(r.state()) [:2])
        u, g, y = r.process(None, 4000, R, flush=True)
        ys.append((u, g, np.array(y).view(np.uint32).copy()))
        outs [threads] = ys
        del r
    for a, b in zip(outs ["0"], outs ["1"]):
        assert a [:2] == b [:2] and np.array_equal(a [2], b [2]) and a [3:] == b [3:]


def test_slices_are_widths_the_matrix_kernels_are_compiled_for(monkeypatch):
    """a shard decides as its stream would, and the matrix-core kernels exist for 1, 2, 4, 8, 16, 32 channels: the channels are cut into
    such widths wherever the shard count allows (12 over 5 = 4 2 2 2 2, not 3 3 2 2 2) — all shards then run the same kernels; where
    it does not (7 channels on 2 shards), and where the stream's own channel count is not such a width (an ordinary 12-channel
    context runs the generic f32 matrix kernel), the shards keep to the f32 kernels.  Either way: the ordinary context's bits at
    every call size, kernel choice left to the library (round 3's advice: 3-channel shards ran f32 beside fixed-point siblings)"""
    for ch, shards, want in ((12, 5, [4, 2, 2, 2, 2]), (8, 3, [4, 2, 2]), (7, 2, [4, 3]), (24, 3, [8, 8, 8]), (6, 4, [2, 2, 1, 1]), (16, 3, [8, 4, 4])):
        monkeypatch.setenv("ARTAMD_SHARDS", str(shards))
        r = A.Resampler(ch, 48, 48, 0.0, BH | INTERP | MT)
        assert [n for _, _, n in r.shards()] == want, (ch, shards, r.shards())
    for ch, shards in ((12, 5), (7, 2), (16, 3)):
        monkeypatch.setenv("ARTAMD_SHARDS", str(shards))
        sizes = [300, 9000, 30000, 140000]
        x = _stream(ch, sum(sizes))
        plain = HipResampler(ch, T, T, 0.0, BH | INTERP); sharded = HipResampler(ch, T, T, 0.0, BH | INTERP | MT)
        assert len(sharded.shards()) == shards
        pos = 0
        for r in (plain, sharded):
            r.advance(T / 2)
        kinds = []
        for n in sizes:
            cap = int(n * R) + 2000
            a = plain.process(x [pos:pos + n], cap, R); ka = (plain.last_kernel(), plain.fixed_point() [0])
            b = sharded.process(x [pos:pos + n], cap, R); kb = (sharded.last_kernel(), sharded.fixed_point() [0])
            pos += n
            assert a [:2] == b [:2] and ka == kb, (ch, n, ka, kb)
            assert np.array_equal(np.array(a [2]).view(np.uint32), np.array(b [2]).view(np.uint32)), (ch, n, ka, kb)
            kinds.append(kb)
        # the big call runs in fixed point on every shard of every one of the three streams — 12 channels as 4 2 2 2 2, 7 as 4 3 (the
        # 3-channel shard in a group of 4: fir_in_groups), 16 as 8 4 4 — and on the ordinary context (12 in a group of 16, 7 in 8)
        assert (2, 1) in kinds, (ch, kinds)


# ---- the other two stages spread the same way: DECIMATE_MULTITHREADED (reference decimator.c:92-93, 119-136) and a multi-device biquad bank ----

DEC_FLAGS = [A.DITHER_HIGHPASS | A.SHAPING_ATH_CURVE, A.DITHER_FLAT, A.SHAPING_3RD_ORDER, 0]


@pytest.mark.parametrize("flags", DEC_FLAGS, ids=["hp_dither_ath", "flat_dither", "shaping3", "plain"])
@pytest.mark.parametrize("ch,bits,nbytes", [(32, 16, 2), (8, 24, 3), (5, 8, 1), (9, 20, 4)], ids=["32ch_16bit", "8ch_24bit_3B", "5ch_8bit", "9ch_20bit_4B"])
def test_sharded_decimator_equals_ordinary_decimator(eight_shards, flags, ch, bits, nbytes):
    """DECIMATE_MULTITHREADED on 8 shards (uneven slices where the channels do not divide; packed bytes of a slice that are not
    whole words) against an ordinary context: the same bytes, clip counts and host-visible state (feedback, dither generators
    — seeded channel after channel from one byte stream — noise shapers) through the interleaved, the planar and the
    device-pointer entry points, state carried across calls"""
    frames = 20000
    x = (_stream(ch, 3 * frames) * 2.3).astype(np.float32)          # (loud enough to clip now and then)
    plain = A.Decimator(ch, bits, nbytes, 1.0, 48000, flags)
    multi = A.Decimator(ch, bits, nbytes, 1.0, 48000, flags | A.DECIMATE_MULTITHREADED)
    assert plain.shards() == 0 and multi.shards() == min(8, ch)
    # interleaved host call
    a, ca = plain.process(x [:frames]); b, cb = multi.process(x [:frames])
    assert ca == cb and np.array_equal(a, b)
    # planar host call
    planes = [np.ascontiguousarray(x [frames:2 * frames, c]) for c in range(ch)]
    pa, ca = plain.process_planar(planes); pb, cb = multi.process_planar(planes)
    assert ca == cb and all(np.array_equal(u, v) for u, v in zip(pa, pb))
    # device-pointer call
    d_in = torch.from_numpy(x [2 * frames:]).cuda()
    oa = torch.zeros(frames * ch * nbytes, dtype=torch.uint8, device="cuda"); ob = torch.zeros_like(oa)
    plain.process_device(d_in, frames, oa); multi.process_device(d_in, frames, ob)
    assert plain.clipped() == multi.clipped()
    assert torch.equal(oa, ob)
    # host-visible state after one more host call (the mirrors are refreshed by host-pointer calls)
    a, ca = plain.process(x [:777]); b, cb = multi.process(x [:777])
    assert ca == cb and np.array_equal(a, b)
    pc, mc = plain.p.contents, multi.p.contents
    assert np.array_equal(np.ctypeslib.as_array(pc.feedback, (ch,)), np.ctypeslib.as_array(mc.feedback, (ch,)))
    if flags & (A.DITHER_HIGHPASS | A.DITHER_FLAT | A.DITHER_LOWPASS):
        assert np.array_equal(np.ctypeslib.as_array(pc.tpdf_generators, (ch,)), np.ctypeslib.as_array(mc.tpdf_generators, (ch,)))


def test_decimate_multithreaded_alone_changes_nothing_on_one_device(monkeypatch):
    monkeypatch.delenv("ARTAMD_SHARDS", raising=False)
    d = A.Decimator(8, 16, 2, 1.0, 48000, A.DITHER_HIGHPASS | A.DECIMATE_MULTITHREADED)
    assert d.shards() == 0


@pytest.mark.parametrize("ch,nsec,cutoff", [(32, 2, 44100 * 0.45 / 96000), (7, 1, 0.01), (8, 4, 0.2)], ids=["32ch_2sec_config_c", "7ch_1sec_slow", "8ch_4sec"])
def test_multi_device_biquad_bank_equals_ordinary_bank(eight_shards, ch, nsec, cutoff):
    """biquadBankCreateMulti on 8 shards against an ordinary bank: the same samples bit for bit (long runs: the time-parallel form
    inside every shard; short runs: the serial kernels), and the same filter state read back, across calls"""
    import ctypes as C
    L = A.lib()
    co = A.BiquadCoefficients(); L.biquad_lowpass(C.byref(co), cutoff)
    secs = (A.Biquad * (ch * nsec))()
    for i in range(ch * nsec):
        L.biquad_init(C.byref(secs [i]), C.byref(co), 1.0 + 0.01 * (i % 3))
    plain, multi = A.BiquadBank(secs, ch, nsec), A.BiquadBank(secs, ch, nsec, multi=True)
    assert plain.shards() == 0 and multi.shards() == min(8, ch)
    for frames in (150000, 500, 70000):
        x = torch.from_numpy(_stream(ch, frames, seed_skip=frames % 97)).cuda()
        a, b = x.clone(), x.clone()
        plain.apply_device(a, frames); multi.apply_device(b, frames)
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), frames
    sa, sb = plain.read(), multi.read()
    assert bytes(sa) == bytes(sb)
    multi.repairs(); plain.repairs()                          # (how many chunks needed a repair depends on the chunking: diagnostics only)
