#!/usr/bin/env python3
"""Golden vectors for the 8-byte sample path, from the REAL reference compiled with -DPATH_WIDTH=64.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden64.py

Loads oracle/_ref/libartref64_{strict,make}.so (the reference's own sources, reference Makefile:13/:19 add
-DPATH_WIDTH=64) and runs the SAME configs, call scripts and noise seeds as make_golden.py, with double samples.
Fixtures are data only.

Files written:
  wide.npz            resample: per-config traces, checksums, head/tail (256 frames) of ref64-strict output;
                      bank rows + sha256; biquad designs / cascade / orders 1-4; decimator table for all
                      bit-depth x dither x shaping combos, planar bytes, ingest
  artest64_kat.json   stderr statistics of the reference's own artest64 program (strict build)
"""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _oracle  # noqa: E402
from make_golden import CONFIGS, ARTEST_RUNS, total_in  # noqa: E402

W = _oracle.wide()
DITHER_HP, DITHER_FLAT, DITHER_LP = _oracle.DITHER_HP, _oracle.DITHER_FLAT, _oracle.DITHER_LP
SHAPE_1, SHAPE_2, SHAPE_3, SHAPE_ATH = _oracle.SHAPE_1, _oracle.SHAPE_2, _oracle.SHAPE_3, _oracle.SHAPE_ATH
f64p, u8p = W.f32p, W.u8p


def drive(kind, cfg, x):
    args, kw = list(cfg["args"]), dict(cfg.get("kw", {}))
    r = W.RefResampler(*args, **kw, kind=kind)
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
    return np.concatenate(outs), np.array(trace, dtype=np.uint64), r.bank()


def gen_resample(out):
    narrow = np.load(os.path.join(HERE, "resample.npz"))
    for name, cfg in CONFIGS.items():
        ch = cfg["args"][0]
        x, _ = W.noise(total_in(cfg["script"]) * ch)
        x = x.reshape(-1, ch)
        y, tr, bank = drive("strict", cfg, x)
        # positions do not depend on the sample type: the call-by-call trace equals the 4-byte build's
        assert np.array_equal(tr, narrow[name + "/trace"]), name
        out[f"resample/{name}/trace"] = tr
        out[f"resample/{name}/sum"] = np.uint64(W.checksum_words(y))
        out[f"resample/{name}/frames"] = np.int64(y.shape[0])
        out[f"resample/{name}/head"] = y[:256]
        out[f"resample/{name}/tail"] = y[-256:]
        F = bank.shape[0] - 1
        rows = sorted({0, 1, F // 2, F - 1, F})
        out[f"bank/{name}/rows"] = np.array(rows)
        out[f"bank/{name}/data"] = bank[rows]
        out[f"bank/{name}/sha256"] = np.frombuffer(hashlib.sha256(bank.tobytes()).digest(), dtype=np.uint8)
        print(f"{name:22s} out={y.shape} sum={int(out[f'resample/{name}/sum']):016x}")


def gen_biquad(out):
    L = W.load_ref("strict")
    for i, f in enumerate((44100 * 0.45 / 96000, 0.1, 0.45)):
        for kind, fn in (("lp", L.biquad_lowpass), ("hp", L.biquad_highpass)):
            c = W.BiquadCoeffs()
            fn(C.byref(c), f)
            out[f"biquad/design/{kind}{i}"] = np.array([f] + [getattr(c, n) for n, _ in W.BiquadCoeffs._fields_], dtype=np.float64)
    ch, frames = 8, 3000
    x, _ = W.noise(frames * ch)
    buf = x.reshape(frames, ch).copy()
    c = W.BiquadCoeffs()
    L.biquad_lowpass(C.byref(c), 44100 * 0.45 / 96000)
    filt = [[W.Biquad(), W.Biquad()] for _ in range(ch)]
    for pair in filt:
        for b in pair:
            L.biquad_init(C.byref(b), C.byref(c), 1.0)
    for blk in range(3):
        view = buf[blk * 1000:(blk + 1) * 1000]
        for k in range(ch):
            for b in filt[k]:
                L.biquad_apply_buffer(C.byref(b), C.cast(view.ctypes.data + 8 * k, f64p), 1000, ch)
    out["biquad/cascade/y"] = buf
    for order in (1, 2, 3, 4):
        c = W.BiquadCoeffs(a0=0.2, a1=0.15, a2=0.1 if order >= 2 else 0.0, a3=-0.05 if order >= 3 else 0.0, a4=0.02 if order >= 4 else 0.0,
                           b1=-0.5, b2=0.2 if order >= 2 else 0.0, b3=-0.1 if order >= 3 else 0.0, b4=0.03 if order >= 4 else 0.0)
        x1, _ = W.noise(600)
        bb, bs = W.Biquad(), W.Biquad()
        L.biquad_init(C.byref(bb), C.byref(c), 0.8)
        L.biquad_init(C.byref(bs), C.byref(c), 0.8)
        yb = x1.copy()
        L.biquad_apply_buffer(C.byref(bb), yb.ctypes.data_as(f64p), 600, 1)
        ys = np.array([L.biquad_apply_sample(C.byref(bs), float(v)) for v in x1], dtype=np.float64)
        out[f"biquad/order{order}/coeffs"] = np.array([getattr(c, n) for n, _ in W.BiquadCoeffs._fields_], dtype=np.float64)
        out[f"biquad/order{order}/buffer"] = yb
        out[f"biquad/order{order}/sample"] = ys


def decimate_input(ch=2, frames=6000):
    x, _ = W.noise(frames * ch)
    x = x * 1.9                                   # a little over full scale in places => exercises clipping
    x[100:110] = 1.5
    x[200:210] = -1.5
    return x


def gen_decimate(out):
    Ls, Lm = W.load_ref("strict"), W.load_ref("make")
    ch, frames = 2, 6000
    x = decimate_input(ch, frames)
    table = []
    for bits, nbytes in ((8, 1), (12, 2), (16, 2), (20, 3), (24, 3), (24, 4), (16, 4)):
        for dither in (0, DITHER_FLAT, DITHER_HP, DITHER_LP):
            for shape, rate in ((0, 48000), (SHAPE_1, 48000), (SHAPE_2, 48000), (SHAPE_3, 48000), (SHAPE_ATH, 44100),
                                (SHAPE_ATH, 48000), (SHAPE_ATH, 96000), (SHAPE_ATH, 50000), (SHAPE_ATH, 32000), (SHAPE_ATH, 88200)):
                res = []
                for L in (Ls, Lm):
                    d = L.decimateInit(ch, bits, nbytes, 1.0, rate, dither | shape)
                    buf = np.zeros(frames * ch * nbytes, np.uint8)
                    clips = 0
                    for blk in range(3):
                        seg = x[blk * 2000 * ch:(blk + 1) * 2000 * ch]
                        clips += L.decimateProcessInterleavedLE(d, seg.ctypes.data_as(f64p), 2000,
                                                                C.cast(buf.ctypes.data + blk * 2000 * ch * nbytes, u8p))
                    L.decimateFree(d)
                    res.append((W.checksum_bytes(buf), clips, buf))
                assert res[0][0] == res[1][0] and res[0][1] == res[1][1], (bits, dither, shape)
                table.append((bits, nbytes, dither, shape, rate, res[0][0], res[0][1]))
                if (bits, nbytes) in ((16, 2), (24, 3)) and dither == DITHER_HP and (shape, rate) == (SHAPE_ATH, 48000):
                    out[f"decimate/bytes/{bits}_{nbytes}_{dither}_{shape}_{rate}"] = res[0][2]
    out["decimate/table"] = np.array(table, dtype=np.uint64)
    d = Ls.decimateInit(ch, 16, 2, 1.0, 48000, DITHER_HP | SHAPE_ATH)
    planes = [np.ascontiguousarray(x.reshape(frames, ch)[:, k]) for k in range(ch)]
    outs = [np.zeros(frames * 2, np.uint8) for _ in range(ch)]
    ip = (f64p * ch)(*[p.ctypes.data_as(f64p) for p in planes])
    op = (u8p * ch)(*[o.ctypes.data_as(u8p) for o in outs])
    out["decimate/planar/clips"] = np.int64(Ls.decimateProcessLE(d, ip, frames, op))
    out["decimate/planar/bytes"] = np.stack(outs)
    Ls.decimateFree(d)
    raw = (np.arange(3 * 4 * 50, dtype=np.uint32) * 2654435761 >> 13).astype(np.uint8)
    for bits, nbytes in ((8, 1), (16, 2), (24, 3), (24, 4), (12, 2), (20, 3)):
        o = np.zeros(50, np.float64)
        Ls.floatIntegersLE(raw.ctypes.data_as(u8p), 0.75, bits, nbytes, 2, o.ctypes.data_as(f64p), 50)
        out[f"ingest/{bits}_{nbytes}"] = o
    out["ingest/raw"] = raw
    print("decimate combos:", len(table))


def gen_artest():
    kat = {}
    exe = os.path.join(_oracle.ORACLE_DIR, "_ref", "artest64_strict")
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
        kat[args] = rec
        print(args, rec.get("output", {}).get("checksum"), rec.get("decimate", {}).get("checksum"))
    with open(os.path.join(HERE, "artest64_kat.json"), "w") as f:
        json.dump({"strict": kat}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    out = {}
    gen_resample(out)
    gen_biquad(out)
    gen_decimate(out)
    np.savez_compressed(os.path.join(HERE, "wide.npz"), **out)
    gen_artest()
