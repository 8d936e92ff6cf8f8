#!/usr/bin/env python3
"""Golden vectors of the time stretcher from the REAL reference (stretch.c compiled into oracle/_ref/libartref*_strict.so by
oracle/Makefile).  Run in the build container only:  make -C oracle ref && python tests/golden/make_golden_stretch.py

Inputs are synthetic (tests/_stretch.py: signal(), regenerated from a seed); stored per case and sample width: the frame
count of every call, a checksum of the whole output, and its first / last 512 frames.  Data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle  # noqa: E402
import _stretch as S  # noqa: E402

out = {}
for width, dt in ((32, np.float32), (64, np.float64)):
    B = _oracle.binding(width)
    for case in S.CASES:
        x, ctor, blocks, ratios = S.case_setup(case, dt)
        y, counts = S.RefStretch(*ctor, width=width).run(x, blocks, ratios)
        key = f"w{width}/{case[0]}"
        out[key + "/counts"] = np.array(counts, np.int64)
        out[key + "/sum"] = np.uint64(B.checksum_words(y) if width == 32 else B.checksum_words(y))
        out[key + "/head"] = y[:512]
        out[key + "/tail"] = y[-512:]
        out[key + "/in_sum"] = np.uint64(B.checksum_words(x))
        print(key, x.shape, "->", y.shape, f"{int(out[key + '/sum']):016x}")
np.savez_compressed(os.path.join(HERE, "stretch.npz"), **out)
