#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the REAL reference.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

It loads oracle/_ref/libartref_{strict,make}.so (the reference's own sources compiled by
oracle/Makefile with C source-order flags resp. the reference Makefile's flags), drives them
through ctypes with artest's synthetic noise (artest.c:744-754) and stores inputs-by-seed and
expected outputs.  Fixtures are data only: no reference source text is stored.

Files written:
  resample.npz   per-config call scripts, per-call (used, generated, outputOffset bits, inputIndex, flags)
                 traces and outputs (ref-strict default math, ref-strict EXTEND_CONVOLUTION_MATH, ref-make)
  bank.npz       selected filter-bank rows + sha256 of whole banks (ref-strict)
  biquad.npz     design coefficients and cascade outputs (ref-strict)
  decimate.npz   byte streams / checksums for bit depths x dither x shaping (ref-strict == ref-make)
  artest_kat.json  stderr statistics of the reference's own artest program (both builds)
"""
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _oracle import (BH, INTERP, LOWPASS, PRECISE, EXTRAP, NO_REDUCTION, DITHER_HP, DITHER_FLAT, DITHER_LP,  # noqa: E402
                     SHAPE_1, SHAPE_2, SHAPE_3, SHAPE_ATH, RefResampler, load_ref, noise, checksum_words,
                     checksum_bytes, BiquadCoeffs, Biquad, f32p, u8p, ORACLE_DIR)
import ctypes as C  # noqa: E402

R4448 = 48000 / 44100
R9644 = 44100 / 96000

# name -> (ctor args, ctor kwargs, advance, call script [(n_in, out_cap, ratio, flush)])
SCRIPT_SHORT = lambda r: [(700, 2000, r, False), (1, 50, r, False), (1500, 3000, r, False), (900, 64, r, False),
                          (900, 2000, r * 1.00013, False), (0, 3000, r, True), (10, 100, r, False)]
SCRIPT_WRAP = lambda r, T: [(15 * T + 37, 40 * T, r, False), (3 * T, 10 * T, r, False), (16 * T, 40 * T, r, False),
                            (0, 4 * T, r, True)]

CONFIGS = {
    # BASELINE.json configs[0]: mono -1 48x48 interpolating
    "P_mono_48x48": dict(args=(1, 48, 48, 0.0, BH | INTERP), adv=24.0, script=SCRIPT_SHORT(R4448), full=True),
    "P_wrap": dict(args=(1, 48, 48, 0.0, BH | INTERP), adv=24.0, script=SCRIPT_WRAP(R4448, 48), full=True),
    # configs[1]: stereo -3 380x380 BH interpolating (artest form) and the ART form (160x380 no-lerp, SNAP)
    "B_stereo_380": dict(args=(2, 380, 380, 0.0, BH | INTERP), adv=190.0, script=SCRIPT_SHORT(R4448), full=True),
    "B_fixed_160x380": dict(args=(2, 380, 380), kw=dict(flags=BH | INTERP | LOWPASS, fixed=(44100., 48000., 0)), adv=190.0,
                            script=SCRIPT_SHORT(R4448), full=True),
    "B_hann": dict(args=(2, 380, 380, 0.0, INTERP), adv=190.0, script=SCRIPT_SHORT(R4448), full=False),
    # configs[2]: 8ch 96k->44.1k -4 auto-lowpass (147x988 no-lerp, SNAP, LP)
    "C_8ch_147x988_lp": dict(args=(8, 988, 988), kw=dict(flags=BH | INTERP | LOWPASS, fixed=(96000., 44100., 0)), adv=494.0,
                             script=[(4000, 4000, R9644, False), (16384, 9000, R9644, False), (0, 2000, R9644, True)], full=False),
    "C_small_147x156_lp": dict(args=(3, 156, 320), kw=dict(flags=BH | INTERP | LOWPASS, fixed=(96000., 44100., 0)), adv=78.0,
                               script=SCRIPT_SHORT(R9644), full=True),
    # headline (metric): 8ch -4 988x988 interpolating; and its 2ch variant
    "A_8ch_988": dict(args=(8, 988, 988, 0.0, BH | INTERP), adv=494.0,
                      script=[(4096, 4962, R4448, False), (4096, 4962, R4448, False), (0, 4962, R4448, True)], full=False),
    "A_2ch_988_wrap": dict(args=(2, 988, 988, 0.0, BH | INTERP), adv=494.0, script=SCRIPT_WRAP(R4448, 988), full=False),
    # configs[4]: stereo ASRC, nearest-filter mode, ratio changes per block
    "E_asrc_380_nolerp": dict(args=(2, 380, 380, 0.0, BH), adv=190.0,
                              script=[(512, 700, R4448 * (1 + 100e-6 * np.sin(2 * np.pi * k / 64)), False) for k in range(24)]
                              + [(0, 700, R4448, True)], full=True),
    # explicit low-pass, fractional advance, odd channel count, tiny filter
    "lp_frac": dict(args=(3, 64, 32, 0.7, BH | INTERP), adv=0.3, script=SCRIPT_SHORT(R4448), full=True),
    "tiny_4x1": dict(args=(1, 4, 1, 0.0, INTERP), adv=None, script=SCRIPT_SHORT(0.5), full=True),
    "no_reduction": dict(args=(2, 32, 64), kw=dict(flags=BH | INTERP | NO_REDUCTION, fixed=(44100., 48000., 0)), adv=16.0,
                         script=SCRIPT_SHORT(R4448), full=True),
    "down_3x": dict(args=(2, 128, 256, 0.0, BH | INTERP), adv=64.0, script=SCRIPT_SHORT(1 / 3.0), full=True),
    "up_4x_pow2": dict(args=(2, 64, 4), kw=dict(flags=BH | INTERP, fixed=(12000., 48000., 0)), adv=32.0,
                       script=SCRIPT_SHORT(4.0), full=True),
    # EXTRAPOLATE_ENDPOINTS (LPC prefill before the first output + extrapolated flush), the way ART calls it (art.c:821-827)
    "X_art_160x380": dict(args=(2, 380, 380), kw=dict(flags=BH | INTERP | LOWPASS | EXTRAP, fixed=(44100., 48000., 0)), adv=190.0,
                          script=SCRIPT_SHORT(R4448), full=True),
    "X_art_147x156_lp": dict(args=(3, 156, 320), kw=dict(flags=BH | INTERP | LOWPASS | EXTRAP, fixed=(96000., 44100., 0)), adv=78.0,
                             script=SCRIPT_SHORT(R9644), full=True),
    "X_interp_48": dict(args=(1, 48, 48, 0.0, BH | INTERP | EXTRAP), adv=24.0, script=SCRIPT_SHORT(R4448), full=True),
    "X_8ch_988": dict(args=(8, 988, 988, 0.0, BH | INTERP | EXTRAP), adv=494.0,
                      script=[(4096, 4962, R4448, False), (4096, 4962, R4448, False), (0, 4962, R4448, True)], full=False),
    # BASELINE.json configs[3]: 32 channels 44.1k->48k preset -4 (988x988 interpolating), and the 4-channel shard one GPU of 8 owns
    "D_4ch_988": dict(args=(4, 988, 988, 0.0, BH | INTERP), adv=494.0, script=SCRIPT_WRAP(R4448, 988), full=False),
    "D_32ch_988": dict(args=(32, 988, 988, 0.0, BH | INTERP), adv=494.0, script=SCRIPT_WRAP(R4448, 988), full=False),
    # EXTRAPOLATE_ENDPOINTS corners (resampler.c:691-698, :775-791, :812-819): a stream shorter than T/2 whose FIRST output is
    # produced by the flush call (the prefill then runs over real samples ++ the forward-extrapolated tail) ...
    "X_short_flush": dict(args=(2, 380, 380, 0.0, BH | INTERP | EXTRAP), adv=190.0,
                          script=[(60, 500, R4448, False), (40, 500, R4448, False), (0, 2000, R4448, True)], full=True),
    "X_short_flush_fixed": dict(args=(3, 156, 320), kw=dict(flags=BH | INTERP | LOWPASS | EXTRAP, fixed=(96000., 44100., 0)), adv=78.0,
                                script=[(50, 500, R9644, False), (0, 7, R9644, True), (0, 500, R9644, True)], full=True),
    # ... and a first output that comes only after the ring has rewound (position advanced by more than 15*T): the
    # reference then extrapolates backwards from the samples since the rewind, over real history
    "X_late_first": dict(args=(2, 380, 380, 0.0, BH | INTERP | EXTRAP), adv=190.0 + 15 * 380 + 100,
                         script=[(16 * 380 + 500, 4000, R4448, False), (700, 2000, R4448, False), (0, 2000, R4448, True)], full=True),
    "X_late_first_split": dict(args=(2, 380, 380, 0.0, BH | INTERP | EXTRAP), adv=190.0 + 15 * 380 + 40,
                               script=[(10 * 380, 4000, R4448, False), (5 * 380 + 7, 4000, R4448, False), (3 * 380, 4000, R4448, False),
                                       (0, 2000, R4448, True)], full=True),
}


def total_in(script):
    return sum(n for n, _, _, f in script if not f) + 16


def drive(kind, cfg, extra_flags, x):
    args, kw = list(cfg["args"]), dict(cfg.get("kw", {}))
    if "flags" in kw:
        kw["flags"] |= extra_flags
    else:
        args[4] |= extra_flags
    r = RefResampler(*args, **kw, kind=kind)
    meta = (r.c.numFilters, r.c.numTaps, r.c.flags, r.c.lowpassRatio, r.c.fixedRatio)
    if cfg["adv"] is not None:
        r.advance(cfg["adv"])
    outs, trace, pos = [], [], 0
    for (n, cap, ratio, flush) in cfg["script"]:
        if flush:
            u, g, o = r.process(None, cap, ratio, flush=True)
        else:
            u, g, o = r.process(x[pos:pos + n], cap, ratio)
            pos += u
        outs.append(o)
        trace.append((u, g) + r.state())
    bank = r.bank()
    return np.concatenate(outs), np.array(trace, dtype=np.uint64), bank, meta


def gen_resample():
    out, banks = {}, {}
    for name, cfg in CONFIGS.items():
        ch = cfg["args"][0]
        x, _ = noise(total_in(cfg["script"]) * ch)
        x = x.reshape(-1, ch)
        y_s, tr_s, bank, meta = drive("strict", cfg, 0, x)
        y_p, tr_p, _, _ = drive("strict", cfg, PRECISE, x)
        y_m, tr_m, _, _ = drive("make", cfg, 0, x)
        assert np.array_equal(tr_s[:, :4], tr_p[:, :4]) and np.array_equal(tr_s, tr_m), name
        out[name + "/script"] = np.array([(n, cap, np.float64(r).view(np.uint64), int(f)) for n, cap, r, f in cfg["script"]], dtype=np.uint64)
        out[name + "/trace"] = tr_s
        out[name + "/meta"] = np.array([meta[0], meta[1], meta[2]], dtype=np.int64)
        out[name + "/meta_f"] = np.array([meta[3], meta[4]], dtype=np.float64)
        out[name + "/sum_strict"] = np.uint64(checksum_words(y_s))
        out[name + "/sum_precise"] = np.uint64(checksum_words(y_p))
        out[name + "/sum_make"] = np.uint64(checksum_words(y_m))
        if cfg["full"]:
            out[name + "/y_strict"] = y_s
            out[name + "/y_precise"] = y_p
            out[name + "/y_make"] = y_m
        else:
            k = 256
            for tag, y in (("strict", y_s), ("precise", y_p), ("make", y_m)):
                out[name + f"/head_{tag}"] = y[:k]
                out[name + f"/tail_{tag}"] = y[-k:]
        F, T = meta[0], meta[1]
        rows = sorted({0, 1, F // 2, F - 1, F})
        banks[name + "/rows"] = np.array(rows)
        banks[name + "/data"] = bank[rows]
        banks[name + "/sha256"] = np.frombuffer(hashlib.sha256(bank.tobytes()).digest(), dtype=np.uint8)
        banks[name + "/shape"] = np.array(bank.shape)
        print(f"{name:22s} F={F} T={T} flags={meta[2]:#x} out={y_s.shape} strict={int(out[name + '/sum_strict']):016x}")
    np.savez_compressed(os.path.join(HERE, "resample.npz"), **out)
    np.savez_compressed(os.path.join(HERE, "bank.npz"), **banks)


def gen_biquad():
    L = load_ref("strict")
    out = {}
    for i, f in enumerate((44100 * 0.45 / 96000, 0.1, 0.45)):
        for kind, fn in (("lp", L.biquad_lowpass), ("hp", L.biquad_highpass)):
            c = BiquadCoeffs()
            fn(C.byref(c), f)
            out[f"design/{kind}{i}"] = np.array([f] + [getattr(c, n) for n, _ in BiquadCoeffs._fields_], dtype=np.float64)
    # two cascaded LP sections per channel, 8 ch interleaved, 3 blocks of 1000 frames (art.c:1011-1017 usage)
    ch, frames = 8, 3000
    x, _ = noise(frames * ch)
    buf = x.reshape(frames, ch).copy()
    c = BiquadCoeffs()
    L.biquad_lowpass(C.byref(c), 44100 * 0.45 / 96000)
    filt = [[Biquad(), Biquad()] for _ in range(ch)]
    for pair in filt:
        for b in pair:
            L.biquad_init(C.byref(b), C.byref(c), 1.0)
    for blk in range(3):
        view = buf[blk * 1000:(blk + 1) * 1000]
        for k in range(ch):
            for b in filt[k]:
                L.biquad_apply_buffer(C.byref(b), C.cast(view.ctypes.data + 4 * k, f32p), 1000, ch)
    out["cascade/y"] = buf
    out["cascade/sum"] = np.uint64(checksum_words(buf))
    # orders 1..4 via apply_buffer and apply_sample with hand-made coefficient sets
    for order in (1, 2, 3, 4):
        c = BiquadCoeffs(a0=0.2, a1=0.15, a2=0.1 if order >= 2 else 0.0, a3=-0.05 if order >= 3 else 0.0, a4=0.02 if order >= 4 else 0.0,
                         b1=-0.5, b2=0.2 if order >= 2 else 0.0, b3=-0.1 if order >= 3 else 0.0, b4=0.03 if order >= 4 else 0.0)
        x1, _ = noise(600)
        bb, bs = Biquad(), Biquad()
        L.biquad_init(C.byref(bb), C.byref(c), 0.8)
        L.biquad_init(C.byref(bs), C.byref(c), 0.8)
        yb = x1.copy()
        L.biquad_apply_buffer(C.byref(bb), yb.ctypes.data_as(f32p), 600, 1)
        ys = np.array([L.biquad_apply_sample(C.byref(bs), float(v)) for v in x1], dtype=np.float32)
        out[f"order{order}/coeffs"] = np.array([getattr(c, n) for n, _ in BiquadCoeffs._fields_], dtype=np.float32)
        out[f"order{order}/buffer"] = yb
        out[f"order{order}/sample"] = ys
    np.savez_compressed(os.path.join(HERE, "biquad.npz"), **out)
    print("biquad cascade sum %016x" % int(out["cascade/sum"]))


def gen_decimate():
    out = {}
    Ls, Lm = load_ref("strict"), load_ref("make")
    ch, frames = 2, 6000
    x, _ = noise(frames * ch)
    x = (x * 1.9).astype(np.float32)            # a little over full scale in places => exercises clipping
    x[100:110] = 1.5
    x[200:210] = -1.5
    combos = []
    for bits, nbytes in ((8, 1), (12, 2), (16, 2), (20, 3), (24, 3), (24, 4), (16, 4)):
        for dither in (0, DITHER_FLAT, DITHER_HP, DITHER_LP):
            for shape, rate in ((0, 48000), (SHAPE_1, 48000), (SHAPE_2, 48000), (SHAPE_3, 48000), (SHAPE_ATH, 44100),
                                (SHAPE_ATH, 48000), (SHAPE_ATH, 96000), (SHAPE_ATH, 50000), (SHAPE_ATH, 32000), (SHAPE_ATH, 88200)):
                combos.append((bits, nbytes, dither, shape, rate))
    table = []
    for (bits, nbytes, dither, shape, rate) in combos:
        res = []
        for L in (Ls, Lm):
            d = L.decimateInit(ch, bits, nbytes, 1.0, rate, dither | shape)
            buf = np.zeros(frames * ch * nbytes, np.uint8)
            clips = 0
            for blk in range(3):                   # 3 calls => state carries across calls
                seg = x[blk * 2000 * ch:(blk + 1) * 2000 * ch]
                clips += L.decimateProcessInterleavedLE(d, seg.ctypes.data_as(f32p), 2000,
                                                        C.cast(buf.ctypes.data + blk * 2000 * ch * nbytes, u8p))
            L.decimateFree(d)
            res.append((checksum_bytes(buf), clips, buf))
        assert res[0][0] == res[1][0] and res[0][1] == res[1][1], (bits, dither, shape)
        table.append((bits, nbytes, dither, shape, rate, res[0][0], res[0][1]))
        if (bits, nbytes) in ((16, 2), (24, 3), (8, 1)) and dither in (0, DITHER_HP) and (shape, rate) in ((0, 48000), (SHAPE_ATH, 48000)):
            out[f"bytes/{bits}_{nbytes}_{dither}_{shape}_{rate}"] = res[0][2]
    out["table"] = np.array(table, dtype=np.uint64)
    # planar entry point on one combo
    d = Ls.decimateInit(ch, 16, 2, 1.0, 48000, DITHER_HP | SHAPE_ATH)
    planes = [np.ascontiguousarray(x.reshape(frames, ch)[:, k]) for k in range(ch)]
    outs = [np.zeros(frames * 2, np.uint8) for _ in range(ch)]
    ip = (f32p * ch)(*[p.ctypes.data_as(f32p) for p in planes])
    op = (u8p * ch)(*[o.ctypes.data_as(u8p) for o in outs])
    out["planar/clips"] = np.int64(Ls.decimateProcessLE(d, ip, frames, op))
    out["planar/bytes"] = np.stack(outs)
    Ls.decimateFree(d)
    # inverse: floatIntegersLE
    raw = (np.arange(3 * 4 * 50, dtype=np.uint32) * 2654435761 >> 13).astype(np.uint8)
    for bits, nbytes in ((8, 1), (16, 2), (24, 3), (24, 4), (12, 2), (20, 3)):
        n = 50
        o = np.zeros(n, np.float32)
        Ls.floatIntegersLE(raw.ctypes.data_as(u8p), 0.75, bits, nbytes, 2, o.ctypes.data_as(f32p), n)
        out[f"ingest/{bits}_{nbytes}"] = o
    out["ingest/raw"] = raw
    np.savez_compressed(os.path.join(HERE, "decimate.npz"), **out)
    print("decimate combos:", len(table))


ARTEST_RUNS = [
    "-1 -c1 -n2 -s44100 -d48000", "-3 -c2 -n2 -s44100 -d48000", "-3 -c2 -n2 -z -s44100 -d48000",
    "-3 -c2 -n2 -p -s44100 -d48000", "-3 -e -c2 -n2 -s44100 -d48000", "-4 -c2 -n2 -s44100 -d48000",
    "-4 -c8 -n2 -o16 -s44100 -d48000", "-4 -c8 -n2 -o24 -s44100 -d48000", "-4 -c8 -n2 -o8 -s44100 -d48000",
    "-4 -e -l -c8 -n2 -s96000 -d44100", "-4 -l20k -c8 -n2 -s96000 -d44100", "-4 -e -c8 -n2 -s44100 -d48000",
    "-2 -c2 -n2 -s44100 -d48000", "-3 -c2 -n2 -s44100 -d48000 -b1000", "-3 -e -c2 -n2 -s48000 -d44100 -l",
    # the -n2 run of the 96k->44.1k config above ends with inputIndex in the last half-window of the ring, where the
    # reference's flush reads before buffers[c][0] (UB, see DESIGN.md "reference bugs"); -n1/-n3 do not
    "-4 -e -l -c8 -n1 -s96000 -d44100", "-4 -e -l -c8 -n3 -s96000 -d44100",
]


def gen_artest():
    kat = {}
    for build in ("make", "strict"):
        exe = os.path.join(ORACLE_DIR, "_ref", f"artest_{build}")
        for args in ARTEST_RUNS:
            p = subprocess.run([exe] + args.split(), capture_output=True, text=True)
            rec = {}
            for line in p.stderr.splitlines():
                m = re.search(r"(input|output|decimate) \(-w\d\): count =\s*(\d+), checksum = ([0-9a-f]{16})", line)
                if m:
                    rec[m.group(1)] = {"count": int(m.group(2)), "checksum": m.group(3)}
                    c = re.search(r"clipped samples = (\d+)", line)
                    if c:
                        rec[m.group(1)]["clips"] = int(c.group(1))
                if "w1 --> w2" in line:
                    rec["banner"] = line.strip()
            kat.setdefault(build, {})[args] = rec
            print(build, args, rec.get("output", {}).get("checksum"), rec.get("decimate", {}).get("checksum"))
    with open(os.path.join(HERE, "artest_kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn in (gen_resample, gen_biquad, gen_decimate, gen_artest):
        if not only or fn.__name__[4:] in only:
            fn()
