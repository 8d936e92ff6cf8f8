#pragma once
// fir_matrix_stream.hip.h — the f32 matrix-core K walk shared by the f32 matrix kernels (fir_matrix.hip) and the fixed-point
// kernel's stand-by (fir_matrix_i8.hip); the streaming tile loop built on it is fir_matrix_stream_body.inc.
#include "fir_matrix_common.hip.h"

#if !ART_WIDE

namespace {

// The K walk of one 32-slot x 128-column tile, shared by both matrix-core kernels (sum += A[32 x K] * X[K x cols], per wave its 32
// columns), chunk by chunk in K order as the staging waves deliver them; one workgroup barrier per chunk.
// Where the f32 accumulator is flushed into the fp64 sums decides the accuracy — and every flush is 32 vector instructions
// beside the matrix pipe.  Measured on the CPU model of this chain (tools/sim/flush_schemes.py: noise, full-scale sines, square
// waves, DC through all five slot tiles, against the fp64 dot product and the reference's own float loop): what matters is
// that no f32 partial sum carries a row's CENTRAL taps for long — the band (the chunks holding any row's central taps) is
// flushed every 4 k, the chunk on either side of it on its own — while everything left of that can share ONE accumulator
// and everything right of it one per four chunks with no measurable change (rms 0.98 x the reference float loop's either
// way; flushing the band every 8 k instead: 1.23 x and out of tolerance).
// PAR: parity of the tile's first chunk in the workgroup's chunk stream (which LDS buffer holds chunk 0).
// (Tried, same box, same run: a frame-major X tile in LDS — the staging waves then write whole dwordx4 loads, 5 LDS writes per
// chunk and thread instead of 9, the matrix waves pick their k's with ds_read2_b32 — 0.1547 vs 0.1541 ms: the staging waves' LDS
// writes are not what the matrix waves wait for.  Raised wave priority for either role: no change.)
// c_from, c_to: the chunks to walk ([0, nchunks): the whole tile; fir_mfma_split_kernel hands a tile's K range to several
// workgroups) — chunk c sits in LDS buffer (c & 1) ^ PAR
template <int PAR>
__device__ __forceinline__ void mf_k_walk (const float (*As_) [32 * MF_LD], const float (*Bs_) [MF_COLS * MF_LD], int arow, int brow,
                                           int nchunks, int band_lo, int band_hi, double (&sum) [16], int c_from = 0, int c_to = 0x7fffffff)
{
    auto b_of = [&] (const float *Bs, int grp) -> f32x4 { return *reinterpret_cast<const f32x4 *> (&Bs [brow + grp * 8]); };
    const int lo_band = band_lo / MF_KC, hi_band = (band_hi + MF_KC - 1) / MF_KC;       // band chunks [lo_band, hi_band)
    const int left_end = lo_band > 1 ? lo_band - 1 : 0;                                 // [0, left_end): one accumulator
    const int right_from = hi_band + 1 < nchunks ? hi_band + 1 : nchunks;               // [right_from, nchunks): one per four chunks

    auto chunk_into = [&] (auto buf_tag, f32x16 &acc) {          // 16 MFMAs of the chunk in LDS buffer BUF ^ PAR, onto acc
        constexpr int BUF = decltype (buf_tag)::value ^ PAR;
        const float *As = As_ [BUF], *Bs = Bs_ [BUF];
#pragma unroll
        for (int grp = 0; grp < MF_KC / 8; ++grp) {
            const f32x4 bv = b_of (Bs, grp);
            const f32x4 av = *reinterpret_cast<const f32x4 *> (&As [arow + grp * 8]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32 (av [q], bv [q], acc, 0, 0, 0);
            if (grp == 1) __builtin_amdgcn_sched_barrier (0);       // operands of two groups at a time (registers)
        }
    };
    auto flush = [&] (f32x16 &acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sum [r] = sum [r] + (double) acc [r]; acc [r] = 0.0f; }
    };
    // chunks [from, to) onto one accumulator, flushed every `every` chunks (0: once at the end)
    auto run = [&] (int from, int to, int every) {
        if (from >= to) return;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc [r] = 0.0f;
        int chunk = from, held = 0;
        auto after = [&] () { if (every && ++held == every) { flush (acc); held = 0; } __syncthreads (); };
        if (chunk & 1) { chunk_into (std::integral_constant<int, 1> {}, acc); after (); ++chunk; }
        for (; chunk + 2 <= to; chunk += 2) {
            chunk_into (std::integral_constant<int, 0> {}, acc); after ();
            chunk_into (std::integral_constant<int, 1> {}, acc); after ();
        }
        if (chunk < to) { chunk_into (std::integral_constant<int, 0> {}, acc); after (); }
        if (!every || held) flush (acc);
    };
    // chunks [from, to) of the band and its two neighbours: band chunks flushed every 4 k, the others once
    auto centre = [&] (int from, int to) {
        for (int chunk = from; chunk < to; ++chunk) {
            const float *As = As_ [((chunk & 1) ^ PAR)], *Bs = Bs_ [((chunk & 1) ^ PAR)];
            const int k0 = chunk * MF_KC;
            if (k0 < band_hi && k0 + MF_KC > band_lo) {
#pragma unroll
                for (int grp = 0; grp < MF_KC / 8; ++grp) {
                    const f32x4 bv = b_of (Bs, grp);
                    const f32x4 av = *reinterpret_cast<const f32x4 *> (&As [arow + grp * 8]);
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        f32x16 acc;
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc [r] = 0.0f;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32 (av [q], bv [q], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32 (av [q + 1], bv [q + 1], acc, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum [r] = sum [r] + (double) acc [r];
                    }
                }
            }
            else {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc [r] = 0.0f;
#pragma unroll
                for (int grp = 0; grp < MF_KC / 8; ++grp) {
                    const f32x4 bv = b_of (Bs, grp);
                    const f32x4 av = *reinterpret_cast<const f32x4 *> (&As [arow + grp * 8]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32 (av [q], bv [q], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sum [r] = sum [r] + (double) acc [r];
            }
            __syncthreads ();
        }
    };
    run (max (0, c_from), min (left_end, c_to), 0);
    centre (max (left_end, c_from), min (right_from, c_to));
    run (max (right_from, c_from), min (nchunks, c_to), 4);
}


} // namespace

#endif  // !ART_WIDE
