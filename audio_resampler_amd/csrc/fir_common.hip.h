#pragma once
// fir_common.hip.h — shared by the FIR translation units (fir_general.hip, fir_matrix.hip, fir_matrix64.hip, fir_dispatch.hip):
// the linear-index view of (history ++ input), the exact replay of the reference's position arithmetic, the direct
// per-sample evaluation and the cross-lane reduction.
//
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
#include <climits>
#include <cstdint>
#include <cstdio>
#include <type_traits>
#include "art_internal.h"

#include "fir_internal.h"

namespace {

struct Pos { int ip; int fi; double frac; };

__device__ __forceinline__ art_s load_frame (const ArtFirArgs &a, int lin_floor, int lin, int ch)
{
    if (lin < lin_floor || lin < 0 || ch >= a.C) return 0.0f;
    if (lin < a.H) return a.hist [(size_t) lin * a.C + ch];
    int f = lin - a.H;
    if (f >= a.in_frames) return 0.0f;
    return a.in_pitch ? a.in [(size_t) ch * a.in_pitch + f] : a.in [(size_t) f * a.C + ch];
}

// the same value through ONE load (address and validity selected first): several of these can be in flight from a loop, where
// load_frame's separate loads meet at a merge and each waits for its own
__device__ __forceinline__ art_s load_frame_flat (const ArtFirArgs &a, int lin_floor, int lin, int ch)
{
    const bool in_hist = lin < a.H;
    const int f = lin - a.H;
    const bool ok = lin >= lin_floor && lin >= 0 && ch < a.C && (in_hist || f < a.in_frames);
    const art_s *ptr = in_hist ? a.hist + ((size_t) lin * a.C + ch) : a.in_pitch ? a.in + ((size_t) ch * a.in_pitch + f) : a.in + ((size_t) f * a.C + ch);
    art_s v = 0;
    if (ok) v = *ptr;
    return v;
}

// last segment whose first output is <= n
__device__ __forceinline__ int find_segment (const ArtSegTable &segs, unsigned int n)
{
    int lo = 0, hi = segs.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs.first [mid] <= n) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// exact replay of the reference's per-output position arithmetic (fp64, un-fused)
template <bool INTERP>
__device__ __forceinline__ Pos locate (const ArtFirArgs &a, const ArtSegTable &segs, unsigned int n)
{
    const int e = find_segment (segs, n);
    const double step = n ? (double) n / a.ratio : 0.0;
    const double off = segs.base [e] + step;
    const double whole = floor (off);
    Pos p;

    double fr = off - whole;
    fr = fr * (double) a.F;

    if (INTERP) {
        p.fi = (int) floor (fr);
        p.frac = fr - (double) p.fi;
    }
    else {
        p.fi = (int) floor (fr + 0.5);
        p.frac = 0.0;
    }

    p.ip = (int) whole + segs.lin_base [e];
    return p;
}

// One output sample evaluated by ONE lane, fp64 accumulation, the reference's lerp: what the matrix-core kernels do with
// an output whose exact position is not its slot's canonical one (rare — a phase on a filter boundary that rounds the other
// way): cheaper than a follow-up launch for a list that is almost always empty.
template <bool INTERP>
__device__ __forceinline__ art_s direct_sample (const ArtFirArgs &a, int lin_floor, Pos p, int ch)
{
    const int half = a.T / 2, w = p.ip - half + 1;
    if (!INTERP && !a.lowpass && (p.fi % a.F) == 0) return load_frame (a, lin_floor, w + half - 1 + p.fi / a.F, ch);
    const art_s *h0 = a.bank + (size_t) p.fi * a.T;
    double s0 = 0.0, s1 = 0.0;
    for (int q = 0; q < half; ++q)                            // mirrored pairs from the edges inwards, as everywhere
        for (int side = 0; side < 2; ++side) {
            const int k = side ? a.T - 1 - q : q;
            const double v = (double) load_frame (a, lin_floor, w + k, ch);
            s0 = s0 + (double) h0 [k] * v;
            if (INTERP) s1 = s1 + (double) h0 [k + a.T] * v;
        }
    if (!INTERP) return (art_s) s0;
    const double left = s0 * (1.0 - p.frac), right = s1 * p.frac;
    return (art_s)(left + right);
}

// Cross-lane reduction of NV per-lane partial sums, carried out in fp64 so that the handful of
// large-magnitude additions near the root of the tree do not each cost half a float ulp.
// Halving butterfly: at every level half of the values change hands, so NV values cost
// NV-1 (+ 6 - log2 NV) shuffle-adds instead of 6*NV.  On return lane L holds the complete sum of
// value (L >> (6 - log2 NV)) in v[0].
// (Levels are unrolled at compile time — with a run-time count of live values the register array is indexed
// dynamically and every exchange turns into a chain of compares and selects over the whole array: 1,400 VALU
// instructions per output for 16 values instead of ~80.)
// lane L <- the value of lane L ^ M, through the data-parallel paths of the vector unit instead of the LDS crossbar (__shfl_xor =
// ds_bpermute_b32: an LDS round trip of ~100+ cycles per 32-bit half, and the five dependent levels of a reduction were close to half
// of a pass of the any-ratio kernels): quad permutes (M = 1, 2), row shifts picked by the lane's bit (4), a row rotation (8), the
// gfx950 row / half swaps (16, 32).  The same pairs exchange the same values: bit-neutral.
template <int M>
__device__ __forceinline__ unsigned int xor_lane_b32 (unsigned int v, int lane)
{
    if constexpr (M == 1) return (unsigned int) __builtin_amdgcn_mov_dpp ((int) v, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return (unsigned int) __builtin_amdgcn_mov_dpp ((int) v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    else if constexpr (M == 4) {
        const unsigned int up = (unsigned int) __builtin_amdgcn_mov_dpp ((int) v, 0x104, 0xf, 0xf, true);        // row_shl:4 — lane L reads lane L + 4
        const unsigned int dn = (unsigned int) __builtin_amdgcn_mov_dpp ((int) v, 0x114, 0xf, 0xf, true);        // row_shr:4 — lane L reads lane L - 4
        return (lane & 4) ? dn : up;
    }
    else if constexpr (M == 8) return (unsigned int) __builtin_amdgcn_mov_dpp ((int) v, 0x128, 0xf, 0xf, true);  // row_ror:8 (rows of 16: L <-> L ^ 8)
    else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap (v, v, false, false);      // r[0] = rows {0,0,2,2}, r[1] = rows {1,1,3,3} of v
        return (lane & 16) ? r [0] : r [1];
    }
    else if constexpr (M >= 64) return v;                     // (no such lane: instantiated only in branches that are never taken)
    else {
        static_assert (M == 32, "lane masks 1 .. 32");
        const auto r = __builtin_amdgcn_permlane32_swap (v, v, false, false);      // r[0] = halves {lo,lo}, r[1] = halves {hi,hi} of v
        return (lane & 32) ? r [0] : r [1];
    }
}
template <int M>
__device__ __forceinline__ double xor_lane (double v, int lane)
{
    const unsigned long long b = (unsigned long long) __double_as_longlong (v);
    const unsigned int lo = xor_lane_b32<M> ((unsigned int) b, lane), hi = xor_lane_b32<M> ((unsigned int)(b >> 32), lane);
    return __longlong_as_double ((long long)(((unsigned long long) hi << 32) | lo));
}

template <int N, int M>                            // N live values, lane mask M
__device__ __forceinline__ void reduce_level (double *v, int lane)
{
    if constexpr (M >= 1) {
        if constexpr (N > 1) {
            const bool upper = (lane & M) != 0;
#pragma unroll
            for (int j = 0; j < N / 2; ++j) {
                const double keep = upper ? v [j + N / 2] : v [j];
                const double send = upper ? v [j] : v [j + N / 2];
                v [j] = keep + xor_lane<M> (send, lane);
            }
            reduce_level<N / 2, M / 2> (v, lane);
        }
        else {
            v [0] = v [0] + xor_lane<M> (v [0], lane);
            reduce_level<1, M / 2> (v, lane);
        }
    }
}

template <int NV>
__device__ __forceinline__ void wave_reduce (double (&v) [NV], int lane)
{
    reduce_level<NV, 32> (v, lane);
}

__attribute__ ((unused)) __device__ __forceinline__ float fused (float a, float b, float c) { return __builtin_fmaf (a, b, c); }
__attribute__ ((unused)) __device__ __forceinline__ double fused (double a, double b, double c) { return __builtin_fma (a, b, c); }

// lanes per output frame: 16 up to 512 taps (256 until round 5), 32 above (64 — one frame per wave — is what the kernel started with and still
// instantiates for experiments).  Measured at 1M-frame blocks: 8 ch x 48 taps 15 -> 35 Gsamples/s, stereo x 156 taps 7.7 ->
// 13.9 (16 lanes); stereo x 380 taps 6.7 -> 8.2, 8 ch x 988 taps 7.5 -> 8.4 (32 lanes): several frames per wave keep more
// coefficient loads in flight and share the reduction.  The price is latency on calls too small to fill the chip — a
// group walks more tap pairs than a wave did: 12 -> 14 us for 1,024 frames at 8 ch x 988 taps — which is why 16 lanes stop
// at 256 taps (at 380 they gain no more than 32 and cost a 10 ms block 2 us).
// The matrix-core kernels' tiles hold `rows` consecutive slots of ONE period.  A short period (48k -> 32k: 2 outputs per 3 inputs;
// 44.1k -> 88.2k: 2 per 1) would leave a tile all but empty, and one that is a little more than a whole number of tiles leaves its
// last tile mostly empty — but any multiple of a period is a period: such ratios are taken several periods at a time (32 per 48,
// 32 per 16).  Returns the multiple: 1 while the padding stays within 15 %, else the one that fills whole tiles, or — where that takes more
// than 16 tiles — the smallest that brings the padding within 4 % (the best below 16 tiles).  A function of the ratio alone: every context of a stream, sharded or not, chooses the same.
static inline int artfir_period_multiple (int P, int rows)
{
    const int tiles = (P + rows - 1) / rows;
    if (tiles * rows * 100 <= P * 115) return 1;
    static const bool off = [] { const char *e = getenv ("ARTAMD_PERIOD_MULTIPLE"); return e && *e == '0'; } ();     // (A/B runs)
    if (off) return 1;
    {   // (a multiple that fills whole tiles, if a small one does)
        int a = P, b = rows;
        while (b) { const int t = a % b; a = b; b = t; }
        if ((rows / a) * P <= 16 * rows) return rows / a;
    }
    int best = 1;
    double best_waste = (double) tiles * rows / P;
    for (int mu = 2; mu * P <= 16 * rows; ++mu) {
        const double w = (double)(((mu * P + rows - 1) / rows) * rows) / (mu * P);
        if (w < best_waste - 1e-9) { best = mu; best_waste = w; if (w <= 1.04) break; }
    }
    return best;
}

// (round 5, with the lean tap loop and the DPP reduction: 16 lanes win up to 512 taps — kernel time, 32 | 16 lanes, 65,536-frame calls: stereo x 380 no-lerp
// 15.1 | 13.1 us, mono 13.3 | 11.4, 4 ch 19.2 | 16.1, 8 ch x 380 interpolating 33.2 | 28.5, stereo x 512 21.4 | 20.1, stereo x 380 interpolating 17.2 | 17.3;
// a 4,096-frame stereo call 5.3 | 5.6)
__host__ __device__ constexpr int general_group (int taps) { return taps <= 512 ? 16 : 32; }

} // namespace
