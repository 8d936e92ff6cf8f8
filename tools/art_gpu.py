#!/usr/bin/env python3
"""art_gpu — sample-rate conversion of WAV files with every DSP stage resident on the MI355X.

The counterpart of the reference's ART command line (reference art.c:96-1155) built on the
device-pointer entry points of libartamd.so: the PCM bytes of a block are uploaded once, then

    floatIntegersLEDevice -> [stretchProcessDevice] -> [biquadBank x2, pre-filter] -> resampleProcessInterleavedDevice
                          -> [biquadBank x2, post-filter] -> decimateProcessInterleavedLEDevice

run back to back on one HIP stream and only the packed output bytes come back.  Block size, flag choices, the
position advance and the output-length rule follow ART (art.c:717, 808-830, 849-874, 924, 933-1067), so with
ARTAMD_STRICT=1 the output file is byte-identical to the reference tool's (tests/test_gpu_dropin.py).

usage: art_gpu.py [-1|-2|-3|-4] [-r<Hz>] [-g<dB>] [-l<Hz>] [-f<n>] [-t<n>] [-o<bits>] [-d<0|1|2>] [-n<0..3>]
                  [-a] [-b] [-h] [-e] [-p] [-x] [-y] [-q] [--tempo=<ratio>] [--pitch=<cents>] [--duration=<[+|-][[hh:]mm:]ss.ss>] in.wav out.wav

--tempo / --pitch put the time stretcher (stretchProcessDevice, mono or stereo) between ingest and resampler, as ART does
(art.c:769-797, 1002-1007).
"""
import ctypes as C
import math
import os
import struct
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

BLOCK = 16384                                   # frames per block (art.c:717)
PRESETS = {"1": (48, 48), "2": (156, 320), "3": (380, 380), "4": (988, 988)}     # taps, filters (art.c:151-166)


def read_wav(path):
    """-> (rate, channels, bits, is_float, channel_mask, raw little-endian sample bytes)"""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise SystemExit(f'"{path}" is not a valid .WAV file!')
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, rate, _, align, bits = struct.unpack_from("<HHIIHH", body, 0)
            # default speaker mask when the file does not carry one: front L/R (or centre) for <= 2 channels, the first
            # `ch` positions up to 18 channels, "all" beyond (art.c:540-547)
            mask, valid = (0x5 - ch) if ch <= 2 else ((1 << ch) - 1 if ch <= 18 else 0xFFFFFFFF), bits
            if tag == 0xFFFE and size >= 40:
                valid, mask, tag = struct.unpack_from("<HIH", body, 18)
            fmt = (rate, ch, valid if valid else bits, tag == 3, mask, align // ch)
        elif cid == b"data":
            payload = body
            break
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise SystemExit(f'"{path}" is not a valid .WAV file!')
    rate, ch, bits, is_float, mask, bytes_per = fmt
    return rate, ch, bits, is_float, mask, bytes_per, payload


def wav_header(bits, channels, frames, rate, mask):
    """same layout decisions as the reference writer (art.c:1157-1215): plain header for mono/stereo with the default
    mask, WAVE_FORMAT_EXTENSIBLE otherwise"""
    bps = (bits + 7) // 8
    fmt_tag = 3 if bits >= 32 else 1
    data_bytes = frames * bps * channels
    base = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * channels * bps, bps * channels, bits)
    if channels > 2 or mask != 0x5 - channels:
        guid = bytearray(14)
        guid[4], guid[6], guid[9], guid[11], guid[12], guid[13] = 0x10, 0x80, 0xAA, 0x38, 0x9B, 0x71
        base = struct.pack("<HHIIHH", 0xFFFE, channels, rate, rate * channels * bps, bps * channels, bits)
        base += struct.pack("<HHIH", 22, bits, mask & 0xFFFFFFFF, fmt_tag) + bytes(guid)
    riff_size = (12 + len(base) + 8 + data_bytes + 1) & ~1
    return b"RIFF" + struct.pack("<I", riff_size) + b"WAVE" + b"fmt " + struct.pack("<I", len(base)) + base + \
        b"data" + struct.pack("<I", data_bytes)


def parse_time_spec(text):
    """--duration=[+|-][[hh:]mm:]ss.ss -> (relative sign or 0, seconds), None when malformed (reference art.c:395-428)"""
    sign = 0
    if text[:1] in "+-":
        sign = 1 if text[0] == "+" else -1
        text = text[1:]
    value, colons = 0.0, 0
    for i, part in enumerate(text.split(":")):
        if i:
            colons += 1
            if colons == 3 or value != math.floor(value):
                return None
            value *= 60.0
        if part == "":
            continue
        try:
            field = float(part)
        except ValueError:
            return None
        if field < 0.0 or (colons and field >= 60.0):
            return None
        value += field
    return sign, value


def main(argv):
    # (no torch: the library's own device-memory entry points — a second less start-up per file than importing a framework)
    os.environ.setdefault("ARTAMD_NO_TORCH", "1")
    import numpy as np
    import audio_resampler_amd as A
    L = A.lib()

    taps, filters = PRESETS["3"]
    rate_out = lowpass = outbits = 0
    gain, phase = 1.0, 0.0
    dither, shaping = A.DITHER_HIGHPASS, A.SHAPING_ATH_CURVE
    bh = hann = allpass = extended = prepost = overwrite = quiet = False
    extrapolate = True
    pitch_ratio = tempo_ratio = 1.0
    duration = None
    files = []
    for arg in argv:
        if arg.startswith("--"):
            key, _, val = arg[2:].partition("=")
            if key.startswith("pitch"): pitch_ratio = 2.0 ** (float(val) / 1200.0)
            elif key.startswith("tempo"): tempo_ratio = float(val)
            elif key.startswith("durat"):
                duration = parse_time_spec(val)
                if duration is None:
                    raise SystemExit("invalid --duration parameter!")
            else: raise SystemExit(f"unknown option {arg} !")
        elif arg.startswith("-") and len(arg) > 1:
            o, v = arg[1], arg[2:]
            num = lambda: float(v[:-1]) * 1000 if v[-1:] in "kK" else float(v)
            if o in PRESETS: taps, filters = PRESETS[o]
            elif o == "r": rate_out = int(num())
            elif o == "g": gain = 10.0 ** (float(v) / 20.0)
            elif o == "s": phase = float(v) / 360.0
            elif o == "l": lowpass = int(num())
            elif o == "f": filters = int(v)
            elif o == "t": taps = int(v)
            elif o == "o": outbits = int(v)
            elif o == "d": dither = {0: 0, 1: A.DITHER_FLAT, 2: A.DITHER_LOWPASS}[int(v)]
            elif o == "n": shaping = {0: 0, 1: A.SHAPING_1ST_ORDER, 2: A.SHAPING_2ND_ORDER, 3: A.SHAPING_3RD_ORDER}[int(v)]
            elif o == "a": allpass = True
            elif o == "b": bh = True
            elif o == "h": hann = True
            elif o == "e": extended = True
            elif o == "p": prepost = True
            elif o == "x": extrapolate = False
            elif o == "y": overwrite = True
            elif o == "q": quiet = True
            else: raise SystemExit(f"illegal option: {o} !")
        else:
            files.append(arg)
    if len(files) != 2:
        raise SystemExit(__doc__)
    src, dst = files
    if os.path.exists(dst) and not overwrite:
        raise SystemExit(f"{dst} exists (use -y to overwrite)")

    rate_in, ch, inbits, is_float, mask, in_bytes, payload = read_wav(src)
    frames_in = len(payload) // (in_bytes * ch)
    rate_out = rate_out or rate_in
    outbits = outbits or inbits
    out_bytes = (outbits + 7) // 8
    ratio = rate_out / rate_in
    stream = None                                            # the null stream

    # ---- a target duration, absolute or relative, becomes a tempo ratio (art.c:740-765)
    if duration is not None:
        if tempo_ratio != 1.0:
            raise SystemExit("error: can't specify BOTH a tempo change and a target duration!")
        source_seconds = frames_in / rate_in
        target_seconds = source_seconds + duration[0] * duration[1] if duration[0] else duration[1]
        if target_seconds <= 0.0:
            raise SystemExit("error: invalid relative duration specified!")
        tempo_ratio = source_seconds / target_seconds

    # ---- time stretcher (art.c:769-797): pitch is a stretch followed by resampling with the inverse ratio
    stretch_ratio, st, stretch_cap = 1.0, None, BLOCK
    if pitch_ratio != 1.0 or tempo_ratio != 1.0:
        stretch_ratio = pitch_ratio / tempo_ratio
        ratio /= pitch_ratio
        if stretch_ratio != 1.0:
            if ch > 2:
                raise SystemExit(f"error: audio stretch only works with mono or stereo, not {ch}-channel")
            if not 0.25 <= stretch_ratio <= 4.0:
                raise SystemExit(f"error: audio stretch requires excessive ratio {stretch_ratio:g}")
            st = L.stretchInit(rate_in // 350, rate_in // 50, ch, 2 if (stretch_ratio < 0.5 or stretch_ratio > 2.0) else 0)
            if not st:
                raise SystemExit("stretchInit failed")
            L.stretchHipSetStream(st, stream)
            stretch_cap = L.stretchGetOutputCapacity(st, BLOCK, stretch_ratio)

    # ---- contexts (art.c:808-890)
    rs = None
    if filters and (ratio != 1.0 or lowpass or phase != 0.0):
        flags = A.SUBSAMPLE_INTERPOLATE | A.INCLUDE_LOWPASS
        if bh or not hann: flags |= A.BLACKMAN_HARRIS
        if phase != 0.0: flags |= A.NO_FILTER_REDUCTION
        if allpass: flags &= ~A.INCLUDE_LOWPASS
        if extrapolate: flags |= A.EXTRAPOLATE_ENDPOINTS
        if extended: flags |= A.EXTEND_CONVOLUTION_MATH
        rs = A.Resampler(ch, taps, filters, flags=flags, fixed=(rate_in * pitch_ratio, float(rate_out), lowpass))
        rs.set_stream(stream)
        rs.advance(taps / 2.0 + phase)
    pre = post = None
    if prepost:
        co = A.BiquadCoefficients()
        cutoff = rate_out * 0.45 / rate_in if rate_out <= rate_in else rate_in * 0.45 / rate_out
        L.biquad_lowpass(C.byref(co), cutoff)
        secs = (A.Biquad * (ch * 2))()
        for i in range(ch * 2):
            L.biquad_init(C.byref(secs[i]), C.byref(co), 1.0)
        bank = A.BiquadBank(secs, ch, 2)
        bank.set_stream(stream)
        pre, post = (bank, None) if rate_out <= rate_in else (None, bank)
    dec = None
    if outbits < 32:
        dec = A.Decimator(ch, outbits, out_bytes, 1.0, rate_out, dither | shaping)
        dec.set_stream(stream)

    cap = int(math.floor((stretch_cap + taps // 2) * ratio + 100.0))
    target = int(math.floor(frames_in * stretch_ratio * ratio + 0.5))
    def dev(nbytes):
        p = L.artamdDeviceAlloc(max(int(nbytes), 16))
        if not p:
            raise SystemExit("device allocation failed")
        return p
    d_raw, d_in = dev(BLOCK * ch * in_bytes), dev(BLOCK * ch * 4)
    d_st = dev(stretch_cap * ch * 4) if st else None
    d_out, d_pcm = dev(cap * ch * 4), dev(cap * ch * out_bytes)
    raw = np.frombuffer(payload, dtype=np.uint8)
    host_out = np.empty(cap * ch * max(out_bytes, 4), np.uint8)

    out_chunks, produced, pos = [], 0, 0
    while produced < target:
        n = min(BLOCK, frames_in - pos)
        if n > 0:
            nbytes = n * ch * in_bytes
            piece = raw[pos * ch * in_bytes:pos * ch * in_bytes + nbytes]
            if inbits > 24:                     # 32-bit float input: samples as they are (x gain, art.c:949-957)
                fl = piece.view(np.float32)
                if gain != 1.0:
                    fl = fl * np.float32(gain)
                fl = np.ascontiguousarray(fl)
                L.artamdUpload(d_in, fl.ctypes.data, nbytes, stream)
                L.artamdStreamSynchronize(stream)       # (`fl` may be a temporary)
            else:
                L.artamdUpload(d_raw, piece.ctypes.data, nbytes, stream)
                L.floatIntegersLEDevice(d_raw, gain, inbits, in_bytes, 1, d_in, n * ch, stream)
            pos += n
        src = d_in
        if st:                                  # stretch (or drain the stretcher once the file is exhausted)
            n = L.stretchProcessDevice(st, d_in, n, d_st, stretch_ratio) if n > 0 else L.stretchFlushDevice(st, d_st)
            src = d_st
        # ART filters `inbuffer` here although, when stretching, the resampler reads the stretcher's own buffer (art.c:1009-1016
        # against :1023): with a stretch the pre-filter has no effect on the output.  Reproduced: the file is the reference's.
        if n > 0 and pre is not None and not st:
            pre.apply_device(src, n)
        if rs is not None:
            _, made = rs.process_device(src if n > 0 else None, n if n > 0 else -1, d_out, cap, ratio)
            if made == cap:
                raise SystemExit("fatal error: outputbuffer too small!")
            buf = d_out
        else:
            made, buf = max(n, 0), src
        if n <= 0 and made == 0:
            if not st or produced >= target:
                break
            # the stretcher can come up short of the rounded target: pad with silence (art.c:1036-1047)
            made = min(target - produced, cap)
            buf = d_out
            L.artamdDeviceZero(d_out, made * ch * 4, stream)
        if post is not None and made:
            post.apply_device(buf, made)
        made = min(made, target - produced)
        if dec is not None:
            dec.process_device(buf, made, d_pcm)
            nout, d_src = made * ch * out_bytes, d_pcm
        else:
            nout, d_src = made * ch * 4, buf
        L.artamdDownload(host_out.ctypes.data, d_src, nout, stream)
        L.artamdStreamSynchronize(stream)
        out_chunks.append(host_out[:nout].tobytes())
        produced += made

    if st:
        L.stretchFree(st)
    body = b"".join(out_chunks)
    with open(dst, "wb") as f:
        f.write(wav_header(outbits, ch, produced, rate_out, mask))
        f.write(body)
        if len(body) & 1:
            f.write(b"\0")
    clipped = dec.clipped() if dec is not None else 0
    if not quiet:
        print(f"{produced} frames x {ch} ch written to {dst}", file=sys.stderr)
    if clipped:                                               # (the reference's warning, quiet or not: art.c:1148-1149)
        print(f"warning: {clipped} samples were clipped, suggest reducing gain!", file=sys.stderr)
    for p in (d_raw, d_in, d_st, d_out, d_pcm):
        if p:
            L.artamdDeviceFree(p)
    if L.artamdErrorCount():                                  # (the reference's void entry points cannot report: counted instead)
        raise SystemExit(f"error: {L.artamdLastError().decode()} — {dst} is not to be trusted")


if __name__ == "__main__":
    main(sys.argv[1:])
