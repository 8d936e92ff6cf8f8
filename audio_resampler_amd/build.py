"""Build libartamd.so and libartamd64.so (C host layer + gfx950 HIP kernels) in-tree.

libartamd64.so is the same source tree compiled with -DPATH_WIDTH=64 (double-precision samples, the
reference's art64 / artest64 builds, reference Makefile:13/:19).

    python -m audio_resampler_amd.build        # or: from audio_resampler_amd.build import build; build()

hipcc cross-compiles gfx950 without a GPU.  -ffp-contract=off is load-bearing: the reference's
position arithmetic, fp64 lerp, biquad and decimator are un-fused; the only FMAs are explicit.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(HERE, "libartamd.so")
OUT64 = os.path.join(HERE, "libartamd64.so")

C_SOURCES = ["resampler_host.c", "pcm_host.c", "extrapolate_host.c", "stretch_host.c"]
HIP_SOURCES = ["device_rt.hip", "fir_general.hip", "fir_matrix.hip", "fir_matrix_i8.hip", "fir_matrix64.hip", "fir_dispatch.hip", "pcm_kernels.hip", "stretch_kernels.hip"]
# per-file extra flags (tried: -fno-slp-vectorize on pcm_kernels.hip — 10 % slower, so none)
EXTRA_FLAGS = {}
HEADERS = [os.path.join(CSRC, h) for h in ("art_internal.h", "fir_internal.h", "fir_common.hip.h", "fir_matrix_common.hip.h", "fir_matrix_stream.hip.h", "fir_matrix_stream_body.inc")] + [os.path.join(INC, h) for h in ("art_hip.h", "resampler.h", "biquad.h", "decimator.h", "stretch.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    build_one(OUT64, "_obj64", ["-DPATH_WIDTH=64"], force, verbose)
    return build_one(OUT, "_obj", [], force, verbose)


def build_one(out, objsub, defs, force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, objsub)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in C_SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(objdir, src + ".o")
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = ["gcc", "-std=c99", "-O2", "-ffp-contract=off", "-fPIC", "-Wall", "-I", INC, "-I", CSRC, "-c", s, "-o", o] + defs
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    for src in HIP_SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(objdir, src + ".o")
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
                   "-I", INC, "-I", CSRC, "-c", s, "-o", o] + defs + EXTRA_FLAGS.get(src, [])
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    if force or _stale(out, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lm", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
