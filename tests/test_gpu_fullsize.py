"""GPU: parity at BASELINE.json's full sizes (the bench workload: 1,048,576 input frames x 8 channels per call).
The oracle is fast enough (a few seconds) to check EVERY sample of one such call; the rest are
size-independent properties (block-size invariance, device == host entry points, checksum equalities)."""
import math

import numpy as np
import pytest

import audio_resampler_amd as A
from _hip import HipResampler, tolerance_ok
from _oracle import OracleResampler, load_oracle, checksum_bytes, BH, INTERP, LOWPASS, PRECISE, f32p, u8p, DITHER_HP, SHAPE_ATH
from audio_resampler_amd.synth import noise

pytestmark = pytest.mark.gpu
T, C, BLOCK, RATIO = 988, 8, 1 << 20, 48000 / 44100


def test_headline_block_every_sample_within_tolerance_of_oracle():
    torch = pytest.importorskip("torch")
    x, _ = noise(BLOCK * C)
    x = x.reshape(BLOCK, C)
    cap = int(math.floor((BLOCK + T // 2) * RATIO + 10))
    g = HipResampler(C, T, T, 0.0, BH | INTERP)
    g.advance(T / 2)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty(cap, C, device="cuda")
    used, made = g.process_device(d_in, BLOCK, d_out, cap, RATIO)
    assert g.last_kernel() == 2 and g.handed_back() == 0
    y = d_out[:made].cpu().numpy()
    # a second call continues the stream across the history seam
    used2, made2 = g.process_device(d_in, 65536, d_out, cap, RATIO)
    y2 = d_out[:made2].cpu().numpy()

    o = OracleResampler(C, T, T, 0.0, BH | INTERP | PRECISE)
    o.advance(T / 2)
    uo, go, yo = o.process(x, cap, RATIO, threads=C)
    uo2, go2, yo2 = o.process(x[:65536], cap, RATIO, threads=C)
    assert (used, made, used2, made2) == (uo, go, uo2, go2) and g.state()[:2] == o.state()[:2]
    ok, worst, rms = tolerance_ok(y, yo)
    assert ok and rms < 2.0e-8, (worst, rms)
    ok2, worst2, _ = tolerance_ok(y2, yo2)
    assert ok2, worst2


def test_fixed_ratio_full_size_block_equals_small_blocks_bit_for_bit():
    """ART's form of the conversion (160 x 988, no interpolation, SNAP), strict numeric mode: one 262,144-frame call
    == four 65,536-frame calls == sixty-four 4,096-frame calls, bit for bit (size-independent property)."""
    x, _ = noise((BLOCK // 4) * C)
    x = x.reshape(-1, C)
    n = x.shape[0]
    outs = []
    for block in (n, 65536, 4096):
        r = HipResampler(C, T, T, flags=BH | INTERP | LOWPASS, fixed=(44100.0, 48000.0, 0), extra=A.RESAMPLE_STRICT_ORDER)
        r.advance(T / 2)
        ys = []
        for p in range(0, n, block):
            u, g, y = r.process(x[p:p + block], int(block * 1.09) + T, 0.0)
            assert u == min(block, n - p)
            ys.append(y)
        outs.append(np.concatenate(ys))
    m = min(len(o) for o in outs)
    assert np.array_equal(outs[0][:m].view(np.uint32), outs[1][:m].view(np.uint32))
    assert np.array_equal(outs[0][:m].view(np.uint32), outs[2][:m].view(np.uint32))


def test_full_size_decimation_checksum_equals_oracle():
    L = load_oracle()
    frames = 1 << 19
    x, _ = noise(frames * C)
    x = (x * 1.6).reshape(frames, C)
    for flags, nbytes, bits in ((DITHER_HP | SHAPE_ATH, 2, 16), (DITHER_HP, 3, 24), (0, 1, 8)):
        d = A.Decimator(C, bits, nbytes, 1.0, 48000, flags)
        got, clips = d.process(x)
        od = L.ora_decimate_init(C, bits, nbytes, 1.0, 48000, flags)
        want = np.zeros(x.size * nbytes, np.uint8)
        wclips = L.ora_decimate_interleaved(od, x.ctypes.data_as(f32p), frames, want.ctypes.data_as(u8p))
        L.ora_decimate_free(od)
        assert clips == wclips
        assert checksum_bytes(got) == checksum_bytes(want)
        assert np.array_equal(got, want)


def test_bench_line_keeps_its_contract():
    """bench.py prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup /
    ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus `roofline` and — unless
    switched off — `cpu_baseline`; the workload is BASELINE.json's, the kernel the matrix-core one, value = samples / time."""
    import json
    from _spawn import run_bench, report
    out = run_bench(["--steps", "5", "--warmup", "2", "--preroll-ms", "20", "--no-cpu-baseline"])
    assert out.returncode == 0, report(out)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    # (the arithmetic type the path computes in: regular launches of the matrix path run in 32-bit fixed point on the int8 matrix cores)
    assert d["unit"] == "Msamples/s" and d["dtype"].startswith("i8 x 4 digits") and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["config"]["fir_kernel"] == "mfma-i8 (fixed point)" and d["config"]["block_frames"] == 1 << 20
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["peak"] > 5000 and 5.0 <= r["digit_pairs_per_chunk"] < 13.0
    # a roofline fraction: what the matrix cores execute over their peak, never above 1; the reference-formulation figure is separate
    assert 0.0 < r["frac"] <= 1.0 and r["useful_frac"] <= r["frac"] and "algorithmic_vs_reference_formulation" in r
    assert "value_cold" in d and d["value_cold"] > 0
    # the kernel's name is asked of the library; the fraction of what the integer matrix cores SUSTAIN on live operands (measured:
    # profiles/r4_mfma_sustain.txt) rides beside the fraction of the nominal peak
    assert r["kernel"] in ("fir_i8_slab_kernel", "fir_i8_dma_kernel") and r["frac"] < r["frac_of_sustained_live_peak"] <= 1.0
    # BASELINE.json configs[3] in the same line (one 32-channel stream; on one GPU the rank owns all of it)
    assert d["config_d"]["stream_channels"] == 32 and d["config_d"]["channels_per_gpu"] == 32 and d["config_d"]["scaling"] == "strong" and d["config_d"]["value"] > 0
    # value is whole-job samples over the timed region: consistent with ms_per_step and the workload's size
    per_step = d["value"] * 1e6 * d["ms_per_step"] * 1e-3
    assert abs(per_step - (1 << 20) * 8 * 48000 / 44100) / per_step < 0.01
    # every other BASELINE.json config rides in the same line (N = 1): B (stereo -3), C (8 ch 96k -> 44.1k -4 + biquads + 16-bit ATH decimation,
    # end to end, the decimator's share stated), E (stereo ASRC, nearest filter, the ratio changed on every call) — each with its kernel and fraction
    assert "other_configs_error" not in d, d.get("other_configs_error")
    b, c, e = d["config_b"], d["config_c"], d["config_e"]
    assert b["channels"] == 2 and b["block_frames"] == 1 << 20 and b["value"] > 0 and 0 < b["roofline"]["frac"] <= 1.0 and b["roofline"]["bound"] == "mfma"
    assert abs(b["value"] * 1e6 * b["ms_per_step"] * 1e-3 - (1 << 20) * 2 * 48000 / 44100) / ((1 << 20) * 2.2) < 0.01
    assert e["channels"] == 2 and e["block_frames"] == 65536 and e["value"] > 0 and e["fir_kernel"] and 0 < e["roofline"]["frac"] <= 1.0
    assert c["channels"] == 8 and c["value"] > 0 and set(c["stage_ms"]) == {"biquad_prefilter", "fir", "decimate"} and abs(sum(c["stage_share"].values()) - 1.0) < 1e-3
    assert abs(c["value"] * 1e6 * c["ms_per_step"] * 1e-3 - (1 << 20) * 8 * 44100 / 96000) / ((1 << 20) * 3.7) < 0.01 and "floor" in c and c["fir"]["roofline"]["frac"] <= 1.0


def test_bench_strong_mode_is_one_32_channel_stream():
    """--scaling strong: ONE 32-channel stream (BASELINE.json configs[3]) cut 32/N per GPU; on one GPU the rank owns all 32"""
    import json
    from _spawn import run_bench, report
    out = run_bench(["--scaling", "strong", "--steps", "3", "--warmup", "1", "--preroll-ms", "0", "--block-frames", "262144", "--no-cpu-baseline"])
    assert out.returncode == 0, report(out)
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["stream_channels"] == 32 and d["config"]["channels_per_gpu"] == 32
    assert "STRONG" in d["config"]["workload"] and d["config"]["fir_kernel"] == "mfma-i8 (fixed point)" and d["roofline"]["frac"] <= 1.0
    per_step = d["value"] * 1e6 * d["ms_per_step"] * 1e-3
    assert abs(per_step - 262144 * 32 * 48000 / 44100) / per_step < 0.01
