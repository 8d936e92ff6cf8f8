// fir_general.hip — the any-ratio kernels of the windowed-sinc interpolator (gfx950): the wave-per-frame-group general kernel
// (default and EXTEND_CONVOLUTION_MATH modes, single and batched launches), the strict-order kernel (RESAMPLE_STRICT_ORDER: the
// reference's C source order, bit for bit), and the small data movers (history roll, planar <-> interleaved).
//
// Reference semantics restated (not translated): reference resampler.c:1135-1181 (subsample_*), :1033-1057 (apply_filter*).
// Compiled with -ffp-contract=off: the only fused multiply-adds are the explicit ones of the default mode.
#include "fir_common.hip.h"

namespace {

constexpr int GEN_THREADS = 256;
// (outputs per workgroup tile at most: 32 until round 6 — three passes of the four waves amortise a tile's positions, staging and barriers better than two, four
// lose again: tools/micro/general_tile_sweep.sh, profiles/r6_config_e.txt.  An output's bits do not depend on its tile.)
constexpr int GEN_MAX_TILE = 48;

// General kernel: one workgroup per tile of consecutive output frames; the tile's input span is
// staged once in LDS (coalesced frame-major reads), then each wave evaluates whole output frames:
// lanes stride the taps, every lane feeds CG channels and both interpolation rows from one LDS read.
// G: lanes that share one output frame (64 = a whole wave, or 16: four output frames per wave side by side — the cross-lane
// reduction and the per-output bookkeeping are then paid once per FOUR outputs, which is most of the cost when taps x
// channels is small).  G depends on the tap count only, never on the tile, so a frame's value does not depend on how a
// call is cut up.
template <int CG, bool INTERP, bool PRECISE, int G, bool PIPE = false, int LEAN = 0>
__device__ __forceinline__ void fir_general_body (const ArtFirArgs &a, const ArtSegTable &segs, int tile, unsigned int bx, unsigned int by)
{
    constexpr int SUBS = 64 / G;
    using Acc = typename std::conditional<PRECISE || ART_WIDE, double, float>::type;   // 8-byte samples accumulate in double
    extern __shared__ __attribute__ ((aligned (16))) art_s xs [];
    __shared__ int s_ip [GEN_MAX_TILE], s_fi [GEN_MAX_TILE];
    __shared__ double s_frac [GEN_MAX_TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane / G, l = lane % G;       // output within the wave, lane within the output's group
    const int ch0 = by * CG;
    const int half = a.T / 2;
    // Blocks [0, 8 * per_xcd) evaluate one tile of outputs each, XCD-aware: workgroup b is dispatched to XCD b % 8, consecutive
    // tiles stage overlapping input spans (a tile's span is T + ~30 frames, its neighbour's starts ~30 frames later), so
    // XCD x takes the CONTIGUOUS tiles [x * per_xcd, (x + 1) * per_xcd): its L2 then sees one eighth of the call's input
    // instead of all of it (measured with round-robin tiles: 4.5x the algorithmic bytes from the fabric on the headline shape,
    // 8.7x on the 65,536-frame stereo ASRC call).  Placement only affects speed.  Any
    // further blocks (x only, y == 0) roll the history for the next call (reads hist ++ in, writes the OTHER history
    // buffer: independent of everything else in flight) — one launch less per call.
    const unsigned int tiles_total = (a.n_end - a.n_begin + (unsigned int) tile - 1) / (unsigned int) tile;
    const unsigned int per_xcd = (tiles_total + 7u) / 8u, workers = 8u * per_xcd;
    if (bx >= workers) {
        if (by) return;
        const int e = (int)(bx - workers) * GEN_THREADS + tid;
        if (e < a.H * a.C) {
            const int f = e / a.C, c = e - f * a.C, lin = a.roll_appended + f;
            art_s v = 0;
            if (lin < a.H) v = a.hist [(size_t) lin * a.C + c];
            else if (a.in) { const int gi = lin - a.H; v = a.in_pitch ? a.in [(size_t) c * a.in_pitch + gi] : a.in [(size_t) gi * a.C + c]; }
            a.roll_dst [e] = v;
        }
        return;
    }
  {
    const unsigned int tile_index = (bx & 7u) * per_xcd + (bx >> 3);
    if (tile_index >= tiles_total) return;
    const unsigned int n0 = a.n_begin + tile_index * (unsigned int) tile;
    const int cnt = (int) min ((unsigned int) tile, a.n_end - n0);

    __syncthreads ();
    if (tid < cnt) {
        Pos p = locate<INTERP> (a, segs, n0 + tid);
        s_ip [tid] = p.ip; s_fi [tid] = p.fi; s_frac [tid] = p.frac;
    }
    __syncthreads ();

    const int lin_lo = s_ip [0] - half + 1;
    const int span = s_ip [cnt - 1] + half + 1 - lin_lo;

    // LEAN == 2 (round 6, stereo streams; tools/micro/general_lean_first.sh): the lean loop with the FIRST round's coefficient loads of the wave's first output issued
    // before the tile's span is staged — they need the filter index only, so that trip to the L2 and the staging's overlap; 2 x R registers, not the 2 x steps
    // of the all-at-once form that lost its occupancy (profiles/r6_config_e.txt).  Same taps, same order: same bits.
    constexpr int RL = CG >= 2 ? 3 : 6;
    art_s pre0 [LEAN == 2 ? RL : 1] [2], pre1 [LEAN == 2 && INTERP ? RL : 1] [2];
    if constexpr (LEAN == 2) {
        constexpr unsigned int SZ = sizeof (art_s);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc (const_cast<art_s *> (a.bank), 0, (int)((unsigned int)(a.F + 1) * (unsigned int) a.T * SZ), 0x00020000);
        constexpr int SUBS_ = 64 / G;
        const int i_first = wave * SUBS_ + sub < cnt ? wave * SUBS_ + sub : cnt - 1;
        const int steps = (half + G - 1) / G, le = l < half ? l : 0;
        const unsigned int row = (unsigned int) s_fi [i_first] * (unsigned int) a.T;
        const unsigned int lo = (row + (unsigned int) le) * SZ, hi = (row + (unsigned int)(a.T - 1 - le)) * SZ, next_row = INTERP ? (unsigned int) a.T * SZ : 0u;
        static_assert (sizeof (art_s) == 4 || LEAN != 2, "4-byte samples");
#pragma unroll
        for (int u = 0; u < RL; ++u)
            if (u < steps) {
                pre0 [u] [0] = __uint_as_float (__builtin_amdgcn_raw_buffer_load_b32 (rb, (int)(lo + (unsigned int)(u * G) * SZ), 0, 0));
                pre0 [u] [1] = __uint_as_float (__builtin_amdgcn_raw_buffer_load_b32 (rb, (int)(hi - (unsigned int)(u * G) * SZ), 0, 0));
                if constexpr (INTERP) {
                    pre1 [u] [0] = __uint_as_float (__builtin_amdgcn_raw_buffer_load_b32 (rb, (int)(lo + (unsigned int)(u * G) * SZ), (int) next_row, 0));
                    pre1 [u] [1] = __uint_as_float (__builtin_amdgcn_raw_buffer_load_b32 (rb, (int)(hi - (unsigned int)(u * G) * SZ), (int) next_row, 0));
                }
            }
    }

    if constexpr (PIPE) {
        // (four loads in flight per thread: one at a time, a long span's rounds are as many trips to memory)
        for (int e0 = tid; e0 < span * CG; e0 += 4 * GEN_THREADS) {
            art_s v [4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = e0 + j * GEN_THREADS, f = e / CG, c = e - f * CG;
                v [j] = load_frame_flat (a, segs.lin_floor, e < span * CG ? lin_lo + f : -1, ch0 + c);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (e0 + j * GEN_THREADS < span * CG) xs [e0 + j * GEN_THREADS] = v [j];
        }
    }
    else
    for (int e = tid; e < span * CG; e += GEN_THREADS) {
        int f = e / CG, c = e - f * CG;
        xs [e] = load_frame (a, segs.lin_floor, lin_lo + f, ch0 + c);
    }
    __syncthreads ();

    // PIPE (long filters, four channels or more — the host's choice, never a tile's: the bits are the same either way): a lane's taps
    // of one output U mirrored pairs (x 2 rows) at a time, ALL of a round's loads issued before the first multiply-add waits for one,
    // and the first round of the wave's NEXT output issued before this output's reduction.  The plain loop below waits for each
    // pair: 31 dependent trips to the L2 per output at 988 taps, which is what a call of a few thousand frames spends its time on
    // (profiles/r4_general_kernel_experiment.txt: 8 ch x 988 taps, 65,536 frames 80 -> 68 us; short filters and stereo lose).
    // Which taps a lane takes, and in which order, is unchanged.
    constexpr int U = 8, STRIDE = (GEN_THREADS / 64) * SUBS;
    const __amdgpu_buffer_rsrc_t r_bank = __builtin_amdgcn_make_buffer_rsrc (const_cast<art_s *> (a.bank), 0, (int)((unsigned int)(a.F + 1) * (unsigned int) a.T * (unsigned int) sizeof (art_s)), 0x00020000);
    art_s c0v [U] [2], c1v [U] [2];
    // (buffer loads: a lane past the row's half reads what it then drops — possibly past the bank's end, which reads as zero)
    auto tap = [&] (unsigned int voff, unsigned int soff) -> art_s {
        if constexpr (sizeof (art_s) == 4) return __uint_as_float (__builtin_amdgcn_raw_buffer_load_b32 (r_bank, (int) voff, (int) soff, 0));
        else {
            typedef unsigned int u32x2_ __attribute__ ((ext_vector_type (2)));
            const u32x2_ w = __builtin_amdgcn_raw_buffer_load_b64 (r_bank, (int) voff, (int) soff, 0);
            return (art_s) __longlong_as_double ((long long)(((unsigned long long) w.y << 32) | w.x));
        }
    };
    auto fetch = [&] (int i, int round) {
        constexpr unsigned int SZ = sizeof (art_s);
        const unsigned int row = (unsigned int) s_fi [i] * (unsigned int) a.T;
        const unsigned int lo = (row + (unsigned int)(l + round * U * G)) * SZ;                  // side 0: tap l + round * U * G, + u * G
        const unsigned int hi = (row + (unsigned int)(a.T - 1 - l - round * U * G)) * SZ;        // side 1: its mirror, - u * G
        const unsigned int next_row = INTERP ? (unsigned int) a.T * SZ : 0u;
        const int steps = (half - round * U * G + G - 1) / G;                                    // (the same for every lane: a short round issues no idle loads)
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u < steps) {
                c0v [u] [0] = tap (lo, (unsigned int)(u * G) * SZ); c0v [u] [1] = tap (hi - (unsigned int)(u * G) * SZ, 0u);
                if (INTERP) { c1v [u] [0] = tap (lo, next_row + (unsigned int)(u * G) * SZ); c1v [u] [1] = tap (hi - (unsigned int)(u * G) * SZ, next_row); }
            }
    };
    auto index_of = [&] (int i0) { return i0 + sub < cnt ? i0 + sub : cnt - 1; };
    if constexpr (PIPE) { if (wave * SUBS < cnt) fetch (index_of (wave * SUBS), 0); }

    for (int i0 = wave * SUBS; i0 < cnt; i0 += (GEN_THREADS / 64) * SUBS) {
        const bool live = i0 + sub < cnt;                       // (a dead group recomputes the tile's last frame and drops it)
        const int i = live ? i0 + sub : cnt - 1;
        const int ip = s_ip [i], fi = s_fi [i];
        const art_s *x = xs + (size_t)(ip - half + 1 - lin_lo) * CG;
        art_s result [CG];

        if (!INTERP && !a.lowpass && (fi == 0 || fi == a.F)) {      // (fi % F == 0 with fi in [0, F]: no integer division in every pass)
            // exact sample hit in nearest-filter mode: the reference copies the sample through
#pragma unroll
            for (int c = 0; c < CG; ++c) result [c] = x [(size_t)(half - 1 + (fi == a.F ? 1 : 0)) * CG + c];
            if constexpr (PIPE) { if (i0 + (GEN_THREADS / 64) * SUBS < cnt) fetch (index_of (i0 + (GEN_THREADS / 64) * SUBS), 0); }     // (these lanes' next output)
        }
        else {
            const art_s *h0 = a.bank + (size_t) fi * a.T;
            const art_s *h1 = h0 + a.T;
            Acc acc0 [CG], acc1 [CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) { acc0 [c] = 0; acc1 [c] = 0; }

            // Taps are visited in mirrored pairs from the window edges towards the centre (as the
            // reference does): partial sums stay small until the dominant central taps arrive, which
            // keeps the float accumulation error at or below the reference's.
            if constexpr (PIPE) {
                const int rounds = (half + G * U - 1) / (G * U);
                for (int r = 0; r < rounds; ++r) {
                    if (r) fetch (i, r);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int p = l + (r * U + u) * G;
                        if (p < half) {
#pragma unroll
                            for (int side = 0; side < 2; ++side) {
                                const int k = side ? a.T - 1 - p : p;
                                const art_s c0 = c0v [u] [side];
                                const art_s c1 = INTERP ? c1v [u] [side] : 0.0f;
#pragma unroll
                                for (int c = 0; c < CG; ++c) {
                                    const art_s v = x [(size_t) k * CG + c];
                                    if (PRECISE) {
                                        acc0 [c] = acc0 [c] + (Acc) c0 * (Acc) v;
                                        if (INTERP) acc1 [c] = acc1 [c] + (Acc) c1 * (Acc) v;
                                    }
                                    else {
                                        acc0 [c] = fused ((Acc) c0, (Acc) v, acc0 [c]);
                                        if (INTERP) acc1 [c] = fused ((Acc) c1, (Acc) v, acc1 [c]);
                                    }
                                }
                            }
                        }
                    }
                }
                if (i0 + STRIDE < cnt) fetch (index_of (i0 + STRIDE), 0);      // (the next output's first round travels under this one's reduction)
            }
            else if constexpr (LEAN != 0) {
                // The plain loop below, instruction for instruction leaner (the kernel is bound by vector-instruction ISSUE — ~210 wave
                // instructions per pass of two outputs for its 12 packed multiply-adds at 380 taps, profiles/r5_config_e.txt; the shelved cell kernel is in git history, tools/attic/ up to round 5 —
                // not by the trips its loads make): R steps at a time, their coefficient loads (buffer loads: one address per lane and
                // side, the step in the instruction's offset) and LDS reads (likewise) all issued before the first multiply-add, no loop
                // control or address arithmetic between them.  A lane takes the same taps in the same order: the same bits.
                constexpr int R = CG >= 2 ? 3 : 6;        // (steps in flight: their samples and coefficients are registers — occupancy — of every wave)
                constexpr unsigned int SZ = sizeof (art_s);
                const int steps = (half + G - 1) / G;                // (uniform)
                const int le = l < half ? l : 0;                     // (lanes past the row's half — short filters — read lane 0's taps and drop them)
                const unsigned int row = (unsigned int) fi * (unsigned int) a.T;
                unsigned int lo = (row + (unsigned int) le) * SZ, hi = (row + (unsigned int)(a.T - 1 - le)) * SZ;
                const art_s *xl = x + (size_t) le * CG, *xh = x + (size_t)(a.T - 1 - le) * CG;
                const unsigned int next_row = INTERP ? (unsigned int) a.T * SZ : 0u;
                for (int r0 = 0; r0 < steps; r0 += R) {
                    art_s c0v [R] [2], c1v [R] [2], xv [R] [2] [CG];
#pragma unroll
                    for (int u = 0; u < R; ++u)
                        if (r0 + u < steps) {
                            // (the last step of a row may lie past its half for the upper lanes: what they read — taps of the row's other half, inside
                            // the row and the window — is dropped below)
                            if (LEAN == 2 && r0 == 0 && i0 == wave * SUBS) {       // (the wave's first output: its first round came in before the staging)
                                c0v [u] [0] = pre0 [u] [0]; c0v [u] [1] = pre0 [u] [1];
                                if (INTERP) { c1v [u] [0] = pre1 [u] [0]; c1v [u] [1] = pre1 [u] [1]; }
                            }
                            else {
                            c0v [u] [0] = tap (lo + (unsigned int)(u * G) * SZ, 0u);
                            c0v [u] [1] = tap (hi - (unsigned int)(u * G) * SZ, 0u);
                            if (INTERP) { c1v [u] [0] = tap (lo + (unsigned int)(u * G) * SZ, next_row); c1v [u] [1] = tap (hi - (unsigned int)(u * G) * SZ, next_row); }
                            }
#pragma unroll
                            for (int c = 0; c < CG; ++c) { xv [u] [0] [c] = xl [(size_t)(u * G) * CG + c]; xv [u] [1] [c] = (xh - (size_t)(u * G) * CG) [c]; }
                        }
#pragma unroll
                    for (int u = 0; u < R; ++u)
                        if (r0 + u < steps && l + (r0 + u) * G < half) {
#pragma unroll
                            for (int side = 0; side < 2; ++side) {
                                const art_s c0 = c0v [u] [side];
                                const art_s c1 = INTERP ? c1v [u] [side] : 0.0f;
#pragma unroll
                                for (int c = 0; c < CG; ++c) {
                                    const art_s v = xv [u] [side] [c];
                                    if (PRECISE) {
                                        acc0 [c] = acc0 [c] + (Acc) c0 * (Acc) v;
                                        if (INTERP) acc1 [c] = acc1 [c] + (Acc) c1 * (Acc) v;
                                    }
                                    else {
                                        acc0 [c] = fused ((Acc) c0, (Acc) v, acc0 [c]);
                                        if (INTERP) acc1 [c] = fused ((Acc) c1, (Acc) v, acc1 [c]);
                                    }
                                }
                            }
                        }
                    lo += (unsigned int)(R * G) * SZ; hi -= (unsigned int)(R * G) * SZ;
                    xl += (size_t)(R * G) * CG; xh -= (size_t)(R * G) * CG;
                }
            }
            else
            for (int p = l; p < half; p += G) {
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int k = side ? a.T - 1 - p : p;
                    const art_s c0 = h0 [k];
                    const art_s c1 = INTERP ? h1 [k] : 0.0f;
#pragma unroll
                    for (int c = 0; c < CG; ++c) {
                        const art_s v = x [(size_t) k * CG + c];
                        if (PRECISE) {
                            acc0 [c] = acc0 [c] + (Acc) c0 * (Acc) v;
                            if (INTERP) acc1 [c] = acc1 [c] + (Acc) c1 * (Acc) v;
                        }
                        else {
                            acc0 [c] = fused ((Acc) c0, (Acc) v, acc0 [c]);
                            if (INTERP) acc1 [c] = fused ((Acc) c1, (Acc) v, acc1 [c]);
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
            reduce_level<NV, G / 2> (part, lane);

            static_assert (NV <= G, "one lane group must hold every value");
            constexpr int GROUP = G / NV;                        // lanes holding the same reduced value
            const double mine = part [0];
            const double frac = s_frac [i];
            art_s y;
            if (INTERP) {
                // the lane group of row fi fetches row fi+1 from the neighbouring group; fp64 lerp, un-fused
                const double s1 = xor_lane<GROUP> (mine, lane);
                const double left = mine * (1.0 - frac);
                const double right = s1 * frac;
                y = (art_s)(left + right);
            }
            else
                y = (art_s) mine;

            const int owner = INTERP ? (l / GROUP) >> 1 : l / GROUP;
            const bool writer = live && (l % GROUP) == 0 && (!INTERP || ((l / GROUP) & 1) == 0);
            if (writer && ch0 + owner < a.C) {
                const size_t n = n0 + i;
                if (a.out_pitch) a.out [(size_t)(ch0 + owner) * a.out_pitch + n] = y;
                else a.out [n * a.C + ch0 + owner] = y;
            }
            continue;
        }

#pragma unroll
        for (int c = 0; c < CG; ++c)
            if (live && l == c && ch0 + c < a.C) {
                const size_t n = n0 + i;
                if (a.out_pitch) a.out [(size_t)(ch0 + c) * a.out_pitch + n] = result [c];
                else a.out [n * a.C + ch0 + c] = result [c];
            }
    }
  }
}

template <int CG, bool INTERP, bool PRECISE, int G, bool PIPE, int LEAN>
__global__ __launch_bounds__ (GEN_THREADS)
void fir_general_kernel (ArtFirArgs a, ArtSegTable segs, int tile)
{
    fir_general_body<CG, INTERP, PRECISE, G, PIPE, LEAN> (a, segs, tile, blockIdx.x, blockIdx.y);
}


// Many independent streams, one launch: blockIdx.z picks a stream's call (its arguments sit in a table in device memory,
// exactly what the single-stream launch would have passed by value), x / y are that call's own grid.  Same body, same
// tile geometry => the samples are identical to n separate launches.
constexpr int BATCH_SEGS = 4;                       // ring-epoch segments a batched call may have (small blocks have 1 or 2)
struct FirBatchItem {
    ArtFirArgs a;
    int seg_count, lin_floor;
    unsigned int first [BATCH_SEGS]; int lin_base [BATCH_SEGS]; double base [BATCH_SEGS];
    int tile; unsigned int blocks_x, blocks_y; int pad;
};

template <int CG, bool INTERP, bool PRECISE, int G>
__global__ __launch_bounds__ (GEN_THREADS)
void fir_general_batch_kernel (const FirBatchItem *items)
{
    __shared__ ArtSegTable s_tab;                   // the table the body expects, rebuilt from the item's few entries
    const FirBatchItem &it = items [blockIdx.z];
    if (blockIdx.x >= it.blocks_x || blockIdx.y >= it.blocks_y) return;
    if (threadIdx.x < BATCH_SEGS) {
        s_tab.first [threadIdx.x] = it.first [threadIdx.x]; s_tab.lin_base [threadIdx.x] = it.lin_base [threadIdx.x];
        s_tab.base [threadIdx.x] = it.base [threadIdx.x];
    }
    if (threadIdx.x == 0) { s_tab.count = it.seg_count; s_tab.lin_floor = it.lin_floor; }
    __syncthreads ();
    fir_general_body<CG, INTERP, PRECISE, G> (it.a, s_tab, it.tile, blockIdx.x, blockIdx.y);
}

// Strict kernel: one lane per output sample, taps visited in the reference's source order
// (pairs from both ends towards the middle, sample-type accumulator; or in order with a double accumulator),
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
    art_s y;

    auto dot = [&] (const art_s *h) -> double {
        if (precise) {
            double acc = 0.0;
            for (int k = 0; k < T; ++k) {
                double prod = (double) h [k] * (double) load_frame (a, segs.lin_floor, w + k, ch);
                acc = acc + prod;
            }
            return acc;
        }
        art_s acc = 0.0f;
        for (int lo = 0, hi = T - 1; lo < hi; ++lo, --hi) {
            art_s pl = h [lo] * load_frame (a, segs.lin_floor, w + lo, ch);
            art_s ph = h [hi] * load_frame (a, segs.lin_floor, w + hi, ch);
            art_s pair = pl + ph;
            acc = acc + pair;
        }
        return (double) acc;
    };

    if (INTERP) {
        double s0 = dot (a.bank + (size_t) p.fi * T);
        double s1 = dot (a.bank + (size_t)(p.fi + 1) * T);
        double left = s0 * (1.0 - p.frac);
        double right = s1 * p.frac;
        y = (art_s)(left + right);
    }
    else if (!a.lowpass && (p.fi % a.F) == 0)
        y = load_frame (a, segs.lin_floor, p.ip + p.fi / a.F, ch);
    else
        y = (art_s) dot (a.bank + (size_t) p.fi * T);

    if (a.out_pitch) a.out [(size_t) ch * a.out_pitch + n] = y;
    else a.out [(size_t) n * a.C + ch] = y;
}

__global__ void roll_history_kernel (art_s *dst, const art_s *hist, const art_s *in, long in_pitch, int appended, int H, int C)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= H * C) return;
    const int f = e / C, c = e - f * C, lin = appended + f;
    art_s v = 0.0f;
    if (lin < H) v = hist [(size_t) lin * C + c];
    else if (in) { const int g = lin - H; v = in_pitch ? in [(size_t) c * in_pitch + g] : in [(size_t) g * C + c]; }
    dst [e] = v;
}

__global__ void interleave_kernel (art_s *dst, const art_s *src, long pitch, int frames, int C)
{
    const size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) frames * C) return;
    const size_t f = e / C; const int c = (int)(e - f * C);
    dst [e] = src [(size_t) c * pitch + f];
}

__global__ void deinterleave_kernel (art_s *dst, long pitch, const art_s *src, int frames, int C)
{
    const size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) frames * C) return;
    const size_t f = e / C; const int c = (int)(e - f * C);
    dst [(size_t) c * pitch + f] = src [e];
}

// tile size, LDS bytes and grid of one general-kernel launch (shared by the single and the batched launch)
template <int CG>
bool general_geometry (const ArtFirArgs &a, int *tile_out, size_t *lds_out, dim3 *grid_out, unsigned int crowd = 1)
{
    // tile size: as many consecutive outputs as keep the staged span within the LDS budget
    const int lds_budget = 64 * 1024;
    const int max_span = lds_budget / ((int) sizeof (art_s) * CG);
    int tile = (int) floor ((max_span - a.T - 3) * a.ratio);
    if (tile > GEN_MAX_TILE) tile = GEN_MAX_TILE;
    {   // (tile-size sweeps: tools/micro/general_tile_sweep.sh)
        static const int t_env = [] { const char *e = getenv ("ARTAMD_GENERAL_TILE"); return e && *e ? atoi (e) : 0; } ();
        if (t_env > 0 && t_env < tile) tile = t_env;
    }
    // small calls: prefer many small tiles (each wave walks its tile's outputs serially, so latency ~ tile/4
    // outputs) over staging efficiency, until there are about four workgroups per CU
    // (`crowd` = launches of this size sharing the grid — the batched entry point: many streams fill the chip together, so
    // each keeps larger tiles.  An output's value does not depend on the tile it is computed in.)
    const unsigned int total_outputs = a.n_end - a.n_begin;
    // (not below one pass of the workgroup's four waves — 8 outputs at 32 lanes per output, 16 at 16: a smaller tile idles waves and
    // doubles the workgroups for nothing; at 4 outputs a 4,096-frame call of 8 ch x 988 taps was 1,115 workgroups, more than the
    // 1,024 the chip holds at once: 20.7 us against 13.7 at 2,896 frames)
    const int pass = (GEN_THREADS / 64) * (64 / general_group (a.T));
    // (whole passes: a tile of 40 outputs costs its first wave three passes like one of 48)
    if (tile > pass) {
        int k = tile / pass;
        while (k > 1 && (unsigned long long)((total_outputs + (unsigned int)(k * pass) - 1) / (unsigned int)(k * pass)) * crowd < 1024u) --k;
        tile = k * pass;
    }
    if (tile < 1) tile = 1;
    long span = a.T + (long) ceil (tile / a.ratio) + 3;
    size_t lds = (size_t) span * CG * sizeof (art_s);
    if (lds > 160 * 1024 - 1024) return false;              // absurd ratio/taps combination
    const unsigned int total = a.n_end - a.n_begin;
    const unsigned int roll_blocks = a.roll_dst ? (unsigned int)((a.H * a.C + GEN_THREADS - 1) / GEN_THREADS) : 0u;
    *tile_out = tile; *lds_out = lds;
    *grid_out = dim3 (8u * (((total + tile - 1) / tile + 7u) / 8u) + roll_blocks, (a.C + CG - 1) / CG);      // (tiles rounded up to the 8 XCDs)
    return true;
}

template <int CG>
int launch_general (const ArtFirArgs &a, const ArtSegTable &segs, hipStream_t st)
{
    int tile; size_t lds; dim3 grid;
    if (!general_geometry<CG> (a, &tile, &lds, &grid)) return -1;
    const bool precise = (a.mode & 3) == ART_MODE_PRECISE;
    static const bool pipe_on = [] { const char *e = getenv ("ARTAMD_GENERAL_PIPE"); return !(e && *e == '0'); } ();      // (A/B runs and the bit-identity test)

#define GO(I, P) do { const int gg = general_group (a.T); if (gg == 16) GO_ (I, P, 16); else if (gg == 32) GO_ (I, P, 32); else GO_ (I, P, 64); } while (0)
    // (ARTAMD_GENERAL_LEAN: 0 pins the plain loop, 1 round 5's lean loop; default 2 = the lean loop, and for STEREO streams its first round's coefficient loads issued
    // before the staging — config E 12.5 -> 12.0 us a call, stereo interpolating 16.6 -> 16.4; mono loses (68 -> 78 registers: 11.0 against 10.2 us) and keeps 1:
    // tools/micro/general_lean_first.sh, profiles/r6_config_e.txt)
    static const int lean_on = [] { const char *e = getenv ("ARTAMD_GENERAL_LEAN"); return e && *e >= '0' && *e <= '2' ? *e - '0' : 2; } ();
#define GO_(I, P, GG) do { if (CG >= 4 && a.T >= 512 && pipe_on) GO__ (I, P, GG, (CG >= 4), 0); else if (lean_on == 2 && CG == 2 && sizeof (art_s) == 4) GO__ (I, P, GG, false, (CG == 2 && sizeof (art_s) == 4 ? 2 : 0)); \
        else if (lean_on && CG <= 2) GO__ (I, P, GG, false, (CG <= 2 ? 1 : 0)); else GO__ (I, P, GG, false, 0); } while (0)
#define GO__(I, P, GG, PP, LL) do { auto k = fir_general_kernel<CG, I, P, GG, PP, LL>; \
        if (lds > 48 * 1024) (void) hipFuncSetAttribute ((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
        hipLaunchKernelGGL (k, grid, dim3 (GEN_THREADS), lds, st, a, segs, tile); } while (0)
    if (a.interpolate) { if (precise) GO (true, true); else GO (true, false); }
    else               { if (precise) GO (false, true); else GO (false, false); }
#undef GO
#undef GO_
#undef GO__
    return 0;
}

template <int CG>
int batch_variant (const ArtFirArgs *a, const ArtSegTable *segs, const int *which, int count, bool interp, bool precise, int group,
                          FirBatchItem *host, FirBatchItem *dev, hipStream_t st)
{
    size_t lds_max = 0; unsigned int gx = 0, gy = 0;
    for (int k = 0; k < count; ++k) {
        const int i = which [k];
        int tile; size_t lds; dim3 grid;
        if (!general_geometry<CG> (a [i], &tile, &lds, &grid, (unsigned int) count)) return -1;
        if (segs [i].count > BATCH_SEGS) return -1;
        host [k].a = a [i]; host [k].seg_count = segs [i].count; host [k].lin_floor = segs [i].lin_floor;
        for (int q = 0; q < BATCH_SEGS; ++q) {
            const bool used = q < segs [i].count;
            host [k].first [q] = used ? segs [i].first [q] : 0u; host [k].lin_base [q] = used ? segs [i].lin_base [q] : 0; host [k].base [q] = used ? segs [i].base [q] : 0.0;
        }
        host [k].tile = tile; host [k].blocks_x = grid.x; host [k].blocks_y = grid.y; host [k].pad = 0;
        if (lds > lds_max) lds_max = lds;
        if (grid.x > gx) gx = grid.x;
        if (grid.y > gy) gy = grid.y;
    }
    if (hipMemcpyAsync (dev, host, sizeof (FirBatchItem) * (size_t) count, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
#define GOB(I, P) do { if (group == 16) GOB_ (I, P, 16); else if (group == 32) GOB_ (I, P, 32); else GOB_ (I, P, 64); } while (0)
#define GOB_(I, P, GG) do { auto k = fir_general_batch_kernel<CG, I, P, GG>; \
        if (lds_max > 48 * 1024) (void) hipFuncSetAttribute ((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_max); \
        hipLaunchKernelGGL (k, dim3 (gx, gy, (unsigned int) count), dim3 (GEN_THREADS), lds_max, st, (const FirBatchItem *) dev); } while (0)
    if (interp) { if (precise) GOB (true, true); else GOB (true, false); }
    else        { if (precise) GOB (false, true); else GOB (false, false); }
#undef GOB
#undef GOB_
    return hipGetLastError () == hipSuccess ? 0 : -1;
}


} // namespace

// the general kernel on one call (default / precise mode); -1 when the tile's input span cannot fit the LDS
int artfir_general (const ArtFirArgs &a, const ArtSegTable &segs, hipStream_t st)
{
    if (a.C > 4) return launch_general<8> (a, segs, st);
    if (a.C > 2) return launch_general<4> (a, segs, st);
    if (a.C == 2) return launch_general<2> (a, segs, st);
    return launch_general<1> (a, segs, st);
}

// one lane per output sample, reference source order; precise: double accumulator (reference apply_filter_precise)
void artfir_strict (const ArtFirArgs &a, const ArtSegTable &segs, int precise, hipStream_t st)
{
    const size_t total = (size_t)(a.n_end - a.n_begin) * a.C;
    const dim3 grid ((unsigned int)((total + 255) / 256));
    if (a.interpolate) hipLaunchKernelGGL (fir_strict_kernel<true>, grid, dim3 (256), 0, st, a, segs, precise);
    else hipLaunchKernelGGL (fir_strict_kernel<false>, grid, dim3 (256), 0, st, a, segs, precise);
}

int artfir_general_group (int taps) { return general_group (taps); }

extern "C" {

// n general-kernel calls of independent streams as ONE launch per kernel variant (column group x interpolation x
// accumulator type; streams of one service normally share it).  d_table: device scratch of at least
// n * arthip_fir_batch_item_bytes () bytes.  Returns 0, or -1 (nothing usable was launched for some item).
size_t arthip_fir_batch_item_bytes (void) { return sizeof (FirBatchItem); }
int arthip_fir_batch_max_segments (void) { return BATCH_SEGS; }

int arthip_fir_batch (const ArtFirArgs *a, const ArtSegTable *segs, int n, void *d_table, void *stream)
{
    hipStream_t st = (hipStream_t) stream;
    if (n <= 0) return 0;
    // pinned staging (per calling thread, kept): the table goes to the device without the runtime's bounce through its own
    // pinned buffers.  Two tables take turns, each guarded by an event recorded after the copies out of it: the call
    // returns without waiting for the stream, and the host plans the next tick while this one runs.
    struct Staging { FirBatchItem *host; size_t cap; hipEvent_t ev; bool pending; };
    static thread_local Staging tl [2] = { { nullptr, 0, nullptr, false }, { nullptr, 0, nullptr, false } };
    static thread_local int tl_turn = 0;
    Staging &sg = tl [tl_turn ^= 1];
    if (sg.pending) { (void) hipEventSynchronize (sg.ev); sg.pending = false; }
    if (!sg.ev && hipEventCreateWithFlags (&sg.ev, hipEventDisableTiming) != hipSuccess) { sg.ev = nullptr; return -1; }
    if ((size_t) n > sg.cap) {
        if (sg.host) (void) hipHostFree (sg.host);
        sg.cap = (size_t) n + (size_t) n / 2 + 64;
        if (hipHostMalloc ((void **) &sg.host, sizeof (FirBatchItem) * sg.cap, hipHostMallocDefault) != hipSuccess) { sg.host = nullptr; sg.cap = 0; return -1; }
    }
    FirBatchItem *host = sg.host;
    int *which = (int *) malloc (sizeof (int) * (size_t) n);
    if (!which) return -1;
    int rc = 0, done = 0;
    // group by kernel variant; each group takes its own slice of the table (the copies are asynchronous, the slices must
    // not be reused inside one call)
    for (int cgi = 0; cgi < 4 && !rc; ++cgi)
        for (int v = 0; v < 12 && !rc; ++v) {
            const bool interp = (v & 1) != 0, precise = (v & 2) != 0;
            const int group = 16 << (v >> 2);                     // 16, 32, 64 lanes per output frame
            int count = 0;
            for (int i = 0; i < n; ++i) {
                const int cls = a [i].C > 4 ? 3 : a [i].C > 2 ? 2 : a [i].C == 2 ? 1 : 0;
                if (cls == cgi && (a [i].interpolate != 0) == interp && (((a [i].mode & 3) == ART_MODE_PRECISE) == precise) &&
                    general_group (a [i].T) == group && a [i].n_end > a [i].n_begin)
                    which [count++] = i;
            }
            if (!count) continue;
            FirBatchItem *hslice = host + done, *dslice = (FirBatchItem *) d_table + done;
            switch (cgi) {
                case 3: rc = batch_variant<8> (a, segs, which, count, interp, precise, group, hslice, dslice, st); break;
                case 2: rc = batch_variant<4> (a, segs, which, count, interp, precise, group, hslice, dslice, st); break;
                case 1: rc = batch_variant<2> (a, segs, which, count, interp, precise, group, hslice, dslice, st); break;
                default: rc = batch_variant<1> (a, segs, which, count, interp, precise, group, hslice, dslice, st); break;
            }
            done += count;
        }
    // the host table must outlive the asynchronous copies out of it: marked here, waited for before its next turn
    if (hipEventRecord (sg.ev, st) == hipSuccess) sg.pending = true;
    else if (hipStreamSynchronize (st) != hipSuccess) rc = -1;
    free (which);
    return rc;
}

int arthip_roll_history (art_s *new_hist, const art_s *hist, const art_s *in, long in_pitch, int appended, int H, int C, void *stream)
{
    const int total = H * C;
    hipLaunchKernelGGL (roll_history_kernel, dim3 ((total + 255) / 256), dim3 (256), 0, (hipStream_t) stream, new_hist, hist, in, in_pitch, appended, H, C);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

int arthip_interleave (art_s *dst, const art_s *src, long pitch, int frames, int C, void *stream)
{
    const size_t total = (size_t) frames * C;
    if (!total) return 0;
    hipLaunchKernelGGL (interleave_kernel, dim3 ((unsigned int)((total + 255) / 256)), dim3 (256), 0, (hipStream_t) stream, dst, src, pitch, frames, C);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

int arthip_deinterleave (art_s *dst, long pitch, const art_s *src, int frames, int C, void *stream)
{
    const size_t total = (size_t) frames * C;
    if (!total) return 0;
    hipLaunchKernelGGL (deinterleave_kernel, dim3 ((unsigned int)((total + 255) / 256)), dim3 (256), 0, (hipStream_t) stream, dst, pitch, src, frames, C);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

}
