"""Load the golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the real
reference) and replay their call scripts against any backend."""
import json
import os

import numpy as np

from _oracle import noise

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# constructor recipes of the golden configs (must mirror tests/golden/make_golden.py CONFIGS)
from _oracle import BH, INTERP, LOWPASS, NO_REDUCTION, EXTRAP  # noqa: E402

CTOR = {
    "P_mono_48x48": dict(args=(1, 48, 48, 0.0, BH | INTERP), adv=24.0),
    "P_wrap": dict(args=(1, 48, 48, 0.0, BH | INTERP), adv=24.0),
    "B_stereo_380": dict(args=(2, 380, 380, 0.0, BH | INTERP), adv=190.0),
    "B_fixed_160x380": dict(args=(2, 380, 380), kw=dict(flags=BH | INTERP | LOWPASS, fixed=(44100., 48000., 0)), adv=190.0),
    "B_hann": dict(args=(2, 380, 380, 0.0, INTERP), adv=190.0),
    "C_8ch_147x988_lp": dict(args=(8, 988, 988), kw=dict(flags=BH | INTERP | LOWPASS, fixed=(96000., 44100., 0)), adv=494.0),
    "C_small_147x156_lp": dict(args=(3, 156, 320), kw=dict(flags=BH | INTERP | LOWPASS, fixed=(96000., 44100., 0)), adv=78.0),
    "A_8ch_988": dict(args=(8, 988, 988, 0.0, BH | INTERP), adv=494.0),
    "A_2ch_988_wrap": dict(args=(2, 988, 988, 0.0, BH | INTERP), adv=494.0),
    "E_asrc_380_nolerp": dict(args=(2, 380, 380, 0.0, BH), adv=190.0),
    "lp_frac": dict(args=(3, 64, 32, 0.7, BH | INTERP), adv=0.3),
    "tiny_4x1": dict(args=(1, 4, 1, 0.0, INTERP), adv=None),
    "no_reduction": dict(args=(2, 32, 64), kw=dict(flags=BH | INTERP | NO_REDUCTION, fixed=(44100., 48000., 0)), adv=16.0),
    "down_3x": dict(args=(2, 128, 256, 0.0, BH | INTERP), adv=64.0),
    "up_4x_pow2": dict(args=(2, 64, 4), kw=dict(flags=BH | INTERP, fixed=(12000., 48000., 0)), adv=32.0),
    "X_art_160x380": dict(args=(2, 380, 380), kw=dict(flags=BH | INTERP | LOWPASS | EXTRAP, fixed=(44100., 48000., 0)), adv=190.0),
    "X_art_147x156_lp": dict(args=(3, 156, 320), kw=dict(flags=BH | INTERP | LOWPASS | EXTRAP, fixed=(96000., 44100., 0)), adv=78.0),
    "X_interp_48": dict(args=(1, 48, 48, 0.0, BH | INTERP | EXTRAP), adv=24.0),
    "X_8ch_988": dict(args=(8, 988, 988, 0.0, BH | INTERP | EXTRAP), adv=494.0),
    "D_4ch_988": dict(args=(4, 988, 988, 0.0, BH | INTERP), adv=494.0),
    "D_32ch_988": dict(args=(32, 988, 988, 0.0, BH | INTERP), adv=494.0),
    "X_short_flush": dict(args=(2, 380, 380, 0.0, BH | INTERP | EXTRAP), adv=190.0),
    "X_short_flush_fixed": dict(args=(3, 156, 320), kw=dict(flags=BH | INTERP | LOWPASS | EXTRAP, fixed=(96000., 44100., 0)), adv=78.0),
    "X_late_first": dict(args=(2, 380, 380, 0.0, BH | INTERP | EXTRAP), adv=190.0 + 15 * 380 + 100),
    "X_late_first_split": dict(args=(2, 380, 380, 0.0, BH | INTERP | EXTRAP), adv=190.0 + 15 * 380 + 40),
}
NAMES = list(CTOR)

_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLD, name + ".npz"))
    return _cache[name]


def kat():
    with open(os.path.join(GOLD, "artest_kat.json")) as f:
        return json.load(f)


def script_of(name):
    z = load("resample")
    s = z[name + "/script"]
    return [(int(n), int(cap), float(np.uint64(r).view(np.float64)), bool(f)) for n, cap, r, f in s]


def make(cls, name, extra_flags=0, **kw_extra):
    cfg = CTOR[name]
    args, kw = list(cfg["args"]), dict(cfg.get("kw", {}))
    if "flags" in kw:
        kw["flags"] |= extra_flags
    else:
        args[4] |= extra_flags
    kw.update(kw_extra)
    r = cls(*args, **kw)
    if cfg["adv"] is not None:
        r.advance(cfg["adv"])
    return r


def replay(backend, name):
    """Run the golden call script; returns (y float32[frames, ch], trace uint64[calls, 5])."""
    script = script_of(name)
    ch = CTOR[name]["args"][0]
    total = sum(n for n, _, _, f in script if not f) + 16
    x, _ = noise(total * ch)
    x = x.reshape(-1, ch)
    outs, trace, pos = [], [], 0
    for (n, cap, ratio, flush) in script:
        if flush:
            u, g, o = backend.process(None, cap, ratio, flush=True)
        else:
            u, g, o = backend.process(x[pos:pos + n], cap, ratio)
            pos += u
        outs.append(np.array(o, copy=True))
        trace.append((u, g) + tuple(backend.state()))
    return np.concatenate(outs), np.array(trace, dtype=np.uint64)


def expected(name, tag):
    """tag in strict|precise|make -> (full y or None, head, tail, checksum)."""
    z = load("resample")
    full = z[name + "/y_" + tag] if (name + "/y_" + tag) in z.files else None
    if full is not None:
        head, tail = full[:256], full[-256:]
    else:
        head, tail = z[name + "/head_" + tag], z[name + "/tail_" + tag]
    return full, head, tail, int(z[name + "/sum_" + tag])
