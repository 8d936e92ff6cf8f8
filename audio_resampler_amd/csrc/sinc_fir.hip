// sinc_fir.hip — gfx950 kernels for the windowed-sinc interpolator.
//
// Reference semantics restated (not translated): reference resampler.c:1135-1181 (subsample_*),
// :1033-1057 (apply_filter*), with positions per reference resampler.c:526/:643/:822 (offset2 = n/ratio).
//
// This translation unit is compiled with -ffp-contract=off: the only fused multiply-adds are the
// explicit ones in the FAST accumulation; position arithmetic and the fp64 lerp round exactly where
// the reference's C does.
//
// Data in HBM (all float32):
//   bank  (F+1) x T            filter rows, row-major
//   hist  H x C                frames kept from previous calls, frame-major (H = 1.5 T)
//   in    n x C  (or planar)   this call's new frames
//   "linear index" lin addresses the concatenation hist ++ in; ring index + lin_base = lin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "art_internal.h"

namespace {

struct Pos { int ip; int fi; double frac; };

__device__ __forceinline__ float load_frame (const ArtFirArgs &a, int lin_floor, int lin, int ch)
{
    if (lin < lin_floor || lin < 0 || ch >= a.C) return 0.0f;
    if (lin < a.H) return a.hist [(size_t) lin * a.C + ch];
    int f = lin - a.H;
    if (f >= a.in_frames) return 0.0f;
    return a.in_pitch ? a.in [(size_t) ch * a.in_pitch + f] : a.in [(size_t) f * a.C + ch];
}

// exact replay of the reference's per-output position arithmetic (fp64, un-fused)
template <bool INTERP>
__device__ __forceinline__ Pos locate (const ArtFirArgs &a, const ArtSegTable &segs, unsigned int n)
{
    int e = 0;
    for (int k = 1; k < segs.count; ++k)
        if (segs.first [k] <= n) e = k;

    const double step = n ? (double) n / a.ratio : 0.0;
    const double off = segs.base [e] + step;
    const double whole = floor (off);
    Pos p;

    if (INTERP) {
        double fr = off - whole;
        fr = fr * (double) a.F;
        p.fi = (int) floor (fr);
        p.frac = fr - (double) p.fi;
    }
    else {
        double fr = off - whole;
        fr = fr * (double) a.F;
        p.fi = (int) floor (fr + 0.5);
        p.frac = 0.0;
    }

    p.ip = (int) whole + segs.lin_base [e];
    return p;
}

// Cross-lane reduction of NV per-lane partial sums, carried out in fp64 so that the handful of
// large-magnitude additions near the root of the tree do not each cost half a float ulp.
// Halving butterfly: at every level half of the values change hands, so NV values cost
// NV-1 (+ 6 - log2 NV) shuffle-adds instead of 6*NV.  On return lane L holds the complete sum of
// value (L >> (6 - log2 NV)) in v[0].
template <int NV>
__device__ __forceinline__ void wave_reduce (double (&v) [NV], int lane)
{
    int n = NV;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        if (n > 1) {
            const bool upper = (lane & m) != 0;
#pragma unroll
            for (int j = 0; j < NV / 2; ++j)
                if (j < n / 2) {
                    const double keep = upper ? v [j + n / 2] : v [j];
                    const double send = upper ? v [j] : v [j + n / 2];
                    v [j] = keep + __shfl_xor (send, m);
                }
            n >>= 1;
        }
        else
            v [0] = v [0] + __shfl_xor (v [0], m);
    }
}

constexpr int GEN_THREADS = 256;
constexpr int GEN_MAX_TILE = 32;

// General kernel: one workgroup per tile of consecutive output frames; the tile's input span is
// staged once in LDS (coalesced frame-major reads), then each wave evaluates whole output frames:
// lanes stride the taps, every lane feeds CG channels and both interpolation rows from one LDS read.
template <int CG, bool INTERP, bool PRECISE>
__global__ __launch_bounds__ (GEN_THREADS)
void fir_general_kernel (ArtFirArgs a, ArtSegTable segs, int tile)
{
    using Acc = typename std::conditional<PRECISE, double, float>::type;
    extern __shared__ __attribute__ ((aligned (16))) float xs [];
    __shared__ int s_ip [GEN_MAX_TILE], s_fi [GEN_MAX_TILE];
    __shared__ double s_frac [GEN_MAX_TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch0 = blockIdx.y * CG;
    const unsigned int n0 = a.n_begin + blockIdx.x * (unsigned int) tile;
    const int cnt = min ((unsigned int) tile, a.n_end - n0);
    const int half = a.T / 2;

    if (tid < cnt) {
        Pos p = locate<INTERP> (a, segs, n0 + tid);
        s_ip [tid] = p.ip; s_fi [tid] = p.fi; s_frac [tid] = p.frac;
    }
    __syncthreads ();

    const int lin_lo = s_ip [0] - half + 1;
    const int span = s_ip [cnt - 1] + half + 1 - lin_lo;

    for (int e = tid; e < span * CG; e += GEN_THREADS) {
        int f = e / CG, c = e - f * CG;
        xs [e] = load_frame (a, segs.lin_floor, lin_lo + f, ch0 + c);
    }
    __syncthreads ();

    for (int i = wave; i < cnt; i += GEN_THREADS / 64) {
        const int ip = s_ip [i], fi = s_fi [i];
        const float *x = xs + (size_t)(ip - half + 1 - lin_lo) * CG;
        float result [CG];

        if (!INTERP && !a.lowpass && (fi % a.F) == 0) {
            // exact sample hit in nearest-filter mode: the reference copies the sample through
#pragma unroll
            for (int c = 0; c < CG; ++c) result [c] = x [(size_t)(half - 1 + fi / a.F) * CG + c];
        }
        else {
            const float *h0 = a.bank + (size_t) fi * a.T;
            const float *h1 = h0 + a.T;
            Acc acc0 [CG], acc1 [CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) { acc0 [c] = 0; acc1 [c] = 0; }

            // Taps are visited in mirrored pairs from the window edges towards the centre (as the
            // reference does): partial sums stay small until the dominant central taps arrive, which
            // keeps the float accumulation error at or below the reference's.
            for (int p = lane; p < half; p += 64) {
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int k = side ? a.T - 1 - p : p;
                    const float c0 = h0 [k];
                    const float c1 = INTERP ? h1 [k] : 0.0f;
#pragma unroll
                    for (int c = 0; c < CG; ++c) {
                        const float v = x [(size_t) k * CG + c];
                        if (PRECISE) {
                            acc0 [c] = acc0 [c] + (Acc) c0 * (Acc) v;
                            if (INTERP) acc1 [c] = acc1 [c] + (Acc) c1 * (Acc) v;
                        }
                        else {
                            acc0 [c] = __builtin_fmaf (c0, v, acc0 [c]);
                            if (INTERP) acc1 [c] = __builtin_fmaf (c1, v, acc1 [c]);
                        }
                    }
                }
            }

            // interleave rows per channel: value index 2c (+1) = row fi (fi+1) of channel c
            constexpr int NV = INTERP ? 2 * CG : CG;
            double part [NV];
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                if (INTERP) { part [2 * c] = (double) acc0 [c]; part [2 * c + 1] = (double) acc1 [c]; }
                else part [c] = (double) acc0 [c];
            }
            wave_reduce<NV> (part, lane);

            constexpr int GROUP = 64 / NV;                       // lanes holding the same reduced value
            const double mine = part [0];
            const double frac = s_frac [i];
            float y;
            if (INTERP) {
                // the lane group of row fi fetches row fi+1 from the neighbouring group; fp64 lerp, un-fused
                const double s1 = __shfl_xor (mine, GROUP);
                const double left = mine * (1.0 - frac);
                const double right = s1 * frac;
                y = (float)(left + right);
            }
            else
                y = (float) mine;

            const int owner = INTERP ? (lane / GROUP) >> 1 : lane / GROUP;
            const bool writer = (lane % GROUP) == 0 && (!INTERP || ((lane / GROUP) & 1) == 0);
            if (writer && ch0 + owner < a.C) {
                const size_t n = n0 + i;
                if (a.out_pitch) a.out [(size_t)(ch0 + owner) * a.out_pitch + n] = y;
                else a.out [n * a.C + ch0 + owner] = y;
            }
            continue;
        }

#pragma unroll
        for (int c = 0; c < CG; ++c)
            if (lane == c && ch0 + c < a.C) {
                const size_t n = n0 + i;
                if (a.out_pitch) a.out [(size_t)(ch0 + c) * a.out_pitch + n] = result [c];
                else a.out [n * a.C + ch0 + c] = result [c];
            }
    }
}

// Strict kernel: one lane per output sample, taps visited in the reference's source order
// (pairs from both ends towards the middle, float accumulator; or in order with a double accumulator),
// no fused operations.  Bit-identical to the reference compiled with -O2 -ffp-contract=off.
template <bool INTERP>
__global__ __launch_bounds__ (256)
void fir_strict_kernel (ArtFirArgs a, ArtSegTable segs, int precise)
{
    const size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n = a.n_begin + (unsigned int)(idx / a.C);
    const int ch = (int)(idx % a.C);
    if (n >= a.n_end) return;

    const Pos p = locate<INTERP> (a, segs, n);
    const int T = a.T, half = T / 2, w = p.ip - half + 1;
    float y;

    auto dot = [&] (const float *h) -> double {
        if (precise) {
            double acc = 0.0;
            for (int k = 0; k < T; ++k) {
                double prod = (double) h [k] * (double) load_frame (a, segs.lin_floor, w + k, ch);
                acc = acc + prod;
            }
            return acc;
        }
        float acc = 0.0f;
        for (int lo = 0, hi = T - 1; lo < hi; ++lo, --hi) {
            float pl = h [lo] * load_frame (a, segs.lin_floor, w + lo, ch);
            float ph = h [hi] * load_frame (a, segs.lin_floor, w + hi, ch);
            float pair = pl + ph;
            acc = acc + pair;
        }
        return (double) acc;
    };

    if (INTERP) {
        double s0 = dot (a.bank + (size_t) p.fi * T);
        double s1 = dot (a.bank + (size_t)(p.fi + 1) * T);
        double left = s0 * (1.0 - p.frac);
        double right = s1 * p.frac;
        y = (float)(left + right);
    }
    else if (!a.lowpass && (p.fi % a.F) == 0)
        y = load_frame (a, segs.lin_floor, p.ip + p.fi / a.F, ch);
    else
        y = (float) dot (a.bank + (size_t) p.fi * T);

    if (a.out_pitch) a.out [(size_t) ch * a.out_pitch + n] = y;
    else a.out [(size_t) n * a.C + ch] = y;
}

__global__ void roll_history_kernel (float *dst, const float *hist, const float *in, long in_pitch, int appended, int H, int C)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= H * C) return;
    const int f = e / C, c = e - f * C, lin = appended + f;
    float v = 0.0f;
    if (lin < H) v = hist [(size_t) lin * C + c];
    else if (in) { const int g = lin - H; v = in_pitch ? in [(size_t) c * in_pitch + g] : in [(size_t) g * C + c]; }
    dst [e] = v;
}

__global__ void interleave_kernel (float *dst, const float *src, long pitch, int frames, int C)
{
    const size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) frames * C) return;
    const size_t f = e / C; const int c = (int)(e - f * C);
    dst [e] = src [(size_t) c * pitch + f];
}

__global__ void deinterleave_kernel (float *dst, long pitch, const float *src, int frames, int C)
{
    const size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) frames * C) return;
    const size_t f = e / C; const int c = (int)(e - f * C);
    dst [(size_t) c * pitch + f] = src [e];
}

template <int CG>
int launch_general (const ArtFirArgs &a, const ArtSegTable &segs, hipStream_t st)
{
    // tile size: as many consecutive outputs as keep the staged span within the LDS budget
    const int lds_budget = 64 * 1024;
    const int max_span = lds_budget / (4 * CG);
    int tile = (int) floor ((max_span - a.T - 3) * a.ratio);
    if (tile > GEN_MAX_TILE) tile = GEN_MAX_TILE;
    if (tile < 1) tile = 1;
    long span = a.T + (long) ceil (tile / a.ratio) + 3;
    size_t lds = (size_t) span * CG * 4;
    if (lds > 160 * 1024 - 1024) return -1;                 // absurd ratio/taps combination
    const unsigned int total = a.n_end - a.n_begin;
    dim3 grid ((total + tile - 1) / tile, (a.C + CG - 1) / CG);
    const bool precise = (a.mode & 3) == ART_MODE_PRECISE;

#define GO(I, P) do { auto k = fir_general_kernel<CG, I, P>; \
        if (lds > 48 * 1024) (void) hipFuncSetAttribute ((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
        hipLaunchKernelGGL (k, grid, dim3 (GEN_THREADS), lds, st, a, segs, tile); } while (0)
    if (a.interpolate) { if (precise) GO (true, true); else GO (true, false); }
    else               { if (precise) GO (false, true); else GO (false, false); }
#undef GO
    return 0;
}

} // namespace

extern "C" {

int arthip_fir (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref, void *stream)
{
    hipStream_t st = (hipStream_t) stream;
    (void) kernel_pref;

    if (a->n_end <= a->n_begin) return ART_KERNEL_GENERAL;

    if ((a->mode & 3) == ART_MODE_STRICT) {
        const size_t total = (size_t)(a->n_end - a->n_begin) * a->C;
        dim3 grid ((unsigned int)((total + 255) / 256));
        if (a->interpolate) hipLaunchKernelGGL (fir_strict_kernel<true>, grid, dim3 (256), 0, st, *a, *segs, (a->mode & 4) != 0);
        else hipLaunchKernelGGL (fir_strict_kernel<false>, grid, dim3 (256), 0, st, *a, *segs, (a->mode & 4) != 0);
        return hipGetLastError () == hipSuccess ? ART_KERNEL_GENERAL : -1;
    }

    int rc;
    if (a->C >= 8 || a->C > 4) rc = launch_general<8> (*a, *segs, st);
    else if (a->C > 2) rc = launch_general<4> (*a, *segs, st);
    else if (a->C == 2) rc = launch_general<2> (*a, *segs, st);
    else rc = launch_general<1> (*a, *segs, st);
    if (rc) return rc;
    return hipGetLastError () == hipSuccess ? ART_KERNEL_GENERAL : -1;
}

int arthip_roll_history (float *new_hist, const float *hist, const float *in, long in_pitch, int appended, int H, int C, void *stream)
{
    const int total = H * C;
    hipLaunchKernelGGL (roll_history_kernel, dim3 ((total + 255) / 256), dim3 (256), 0, (hipStream_t) stream, new_hist, hist, in, in_pitch, appended, H, C);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

int arthip_interleave (float *dst, const float *src, long pitch, int frames, int C, void *stream)
{
    const size_t total = (size_t) frames * C;
    if (!total) return 0;
    hipLaunchKernelGGL (interleave_kernel, dim3 ((unsigned int)((total + 255) / 256)), dim3 (256), 0, (hipStream_t) stream, dst, src, pitch, frames, C);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

int arthip_deinterleave (float *dst, long pitch, const float *src, int frames, int C, void *stream)
{
    const size_t total = (size_t) frames * C;
    if (!total) return 0;
    hipLaunchKernelGGL (deinterleave_kernel, dim3 ((unsigned int)((total + 255) / 256)), dim3 (256), 0, (hipStream_t) stream, dst, pitch, src, frames, C);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

}
