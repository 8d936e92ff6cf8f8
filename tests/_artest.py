"""The reference's test program (artest.c:116-612) restated as a harness that can drive ANY
backend exposing .process(x, out_cap, ratio, flush=..., and_flush=...) -> (used, generated, y):
the oracle, the real reference, or the HIP library.  Returns the same statistics artest prints."""
import math

import numpy as np

from _oracle import load_oracle, NOISE_SEED, f32p, checksum_words, checksum_bytes

PRESETS = {1: (48, 48), 2: (156, 320), 3: (380, 380), 4: (988, 988)}       # taps, filters (artest.c:154-169)


def workload_blocks(chans, src_rate, seconds, block=4096, fades=True):
    """Yield (index, is_last, float32[block, chans]) exactly as artest generates them (artest.c:446-462)."""
    L = load_oracle()
    nblocks = math.ceil(seconds * src_rate / block)
    state = NOISE_SEED
    for bi in range(nblocks):
        buf = np.empty(block * chans, np.float32)
        state = L.ora_noise_fill(buf.ctypes.data_as(f32p), buf.size, state)
        if fades and bi == 0:
            L.ora_fade_in(buf.ctypes.data_as(f32p), buf.size)
        elif fades and bi == nblocks - 1:
            L.ora_fade_out(buf.ctypes.data_as(f32p), buf.size)
        yield bi, bi == nblocks - 1, buf.reshape(block, chans)


def run_artest(make_resampler, chans, taps, src_rate, dst_rate, seconds, block=4096, ratio_arg=None,
               decimator=None, collect=False):
    """make_resampler() -> backend (already advanced by taps/2, artest.c:433).
    ratio_arg: ratio passed per call (0.0 for -e fixed-ratio contexts, artest.c:401).
    decimator: optional callable(y float32[frames, ch]) -> (bytes uint8 array, clips)."""
    ratio = dst_rate / src_rate
    out_cap = int(math.floor((block + taps // 2) * ratio + 10))            # artest.c:369
    rs = make_resampler()
    call_ratio = ratio if ratio_arg is None else ratio_arg
    in_sum = out_sum = dec_sum = 0
    out_frames = clips = dec_bytes = 0
    chunks = []
    for bi, last, x in workload_blocks(chans, src_rate, seconds, block):
        in_sum = checksum_words(x, in_sum)
        used, gen, y = rs.process(x, out_cap, call_ratio, and_flush=last)
        assert used == block and gen != out_cap, "fatal error in resample results! (artest.c:486)"
        out_sum = checksum_words(y, out_sum)
        out_frames += gen
        if collect:
            chunks.append(y.copy())
        if decimator is not None:
            b, c = decimator(y)
            dec_sum = checksum_bytes(b, dec_sum)
            dec_bytes += b.size
            clips += c
    res = {"out_frames": out_frames, "in_checksum": "%016x" % in_sum, "out_checksum": "%016x" % out_sum}
    if decimator is not None:
        res.update(dec_checksum="%016x" % dec_sum, dec_bytes=dec_bytes, clips=clips)
    if collect:
        res["y"] = np.concatenate(chunks)
    return res
