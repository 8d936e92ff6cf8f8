#pragma once
// fir_matrix_stream.hip.h — the f32 matrix-core K walk and the persistent ("streaming") tile loop built on it, as device code
// two kernels share: fir_mfma_stream_kernel (fir_matrix.hip) and, as its on-device stand-by, fir_i8_stream_kernel
// (fir_matrix_i8.hip: a launch whose samples the fixed-point digits cannot hold is produced by this loop inside the same
// kernel — no second launch behind every fixed-point call).
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
template <int PAR>
__device__ __forceinline__ void mf_k_walk (const float (*As_) [32 * MF_LD], const float (*Bs_) [MF_COLS * MF_LD], int arow, int brow,
                                           int nchunks, int band_lo, int band_hi, double (&sum) [16])
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
    run (0, left_end, 0);
    centre (left_end, right_from);
    run (right_from, nchunks, 4);
}


// The tile loop of fir_mfma_stream_kernel (see there): workgroup of 8 waves (4 matrix, 4 staging), LDS buffers handed in
// (2 x 32 rows and 2 x 128 columns of MF_LD floats), every workgroup of the grid's first 8 * wgs_per_xcd takes part.
template <int CG, bool PASS>
__device__ __forceinline__ void mfma_stream_tiles (const ArtFirArgs &a, const MfmaGeom &g, int wgs_per_xcd,
                                                   float (*As_) [32 * MF_LD], float (*Bs_) [MF_COLS * MF_LD])
{
    constexpr int THREADS = 2 * MF_THREADS;
    constexpr int PPW = MF_COLS / CG > MF_MAX_PPW ? MF_MAX_PPW : MF_COLS / CG;
    constexpr int NCOLS = PPW * CG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 4;
    const int pt = tid & (MF_THREADS - 1);
    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;
    const int tiles_per_xcd = g.groups_per_xcd * g.slot_tiles;
    const int nchunks = g.ktot / MF_KC;

    // tile `within` of this XCD's list -> (slot tile, period group); false past the last valid tile (validity is monotone)
    auto tile_at = [&] (int within, int &st, int &jg) -> bool {
        if (within >= tiles_per_xcd) return false;
        st = within % g.slot_tiles; jg = xcd * g.groups_per_xcd + within / g.slot_tiles;
        if (jg >= g.period_groups) return false;
        return a.n_begin + (unsigned int)(jg * PPW) * g.P + (unsigned int)(st * 32) < a.n_end;
    };

    if (NCOLS < MF_COLS)                                      // unused columns stay zero for the whole kernel
        for (int e = tid; e < (MF_COLS - NCOLS) * MF_LD; e += THREADS)
            for (int b = 0; b < 2; ++b) Bs_ [b] [NCOLS * MF_LD + e] = 0.0f;

    if (loader) {
        constexpr int VEC = CG >= 4 ? 4 : (CG == 2 ? 2 : 1);
        constexpr int VPF = CG / VEC, VPP = MF_KC * VPF, NB = (PPW * VPP) / MF_THREADS;
        constexpr unsigned int A_STEP = MF_KC * 4u, B_STEP = MF_KC * CG * 4u;
        const int a_row = pt >> 3, a_kseg = (pt & 7) * 4;
        const unsigned int a_off0 = (unsigned int)(a_row * g.ktot + a_kseg) * 4u;
        const int adst = a_row * MF_LD + a_kseg;
        unsigned int boff [NB]; int bdst [NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int v = pt + u * MF_THREADS;
            const int jl = v / VPP, rem = v % VPP, kk = rem / VPF, cv = rem % VPF;
            boff [u] = (unsigned int)((jl * g.Q + kk) * CG + cv * VEC) * 4u;       // (the tile's window origin sits in the resource base)
            bdst [u] = (jl * CG + cv * VEC) * MF_LD + kk;
        }
        float ra0 [4], rb0 [NB * VEC];

        // the fetch stream: tile being fetched, its bases, the chunk to fetch next (all uniform)
        int f_within = rank, f_chunk = 0;
        bool f_live = false;
        const char *fa_base = nullptr, *fb_base = nullptr;
        unsigned int fa_bytes = 0, fb_bytes = 0;
        auto open_tile = [&] () {
            int st, jg;
            f_live = tile_at (f_within, st, jg);
            if (!f_live) return;
            const int w0 = g.tile_w0 [3 * st] + jg * PPW * g.Q;
            const bool touches_hist = w0 < a.H;              // (first period group of a call: staged from the gathered head)
            const int origin = touches_hist ? -MF_HEAD_PAD : a.H;
            const char *base = touches_hist ? reinterpret_cast<const char *> (g.head) : reinterpret_cast<const char *> (a.in);
            const size_t total = touches_hist ? (size_t) g.head_frames * a.C * 4 : (size_t) a.in_frames * a.C * 4;
            size_t skip = (size_t) max (w0 - origin, 0) * CG * 4;
            if (skip > total) skip = total;
            fb_base = base + skip; fb_bytes = (unsigned int)(total - skip);
            fa_base = reinterpret_cast<const char *> (g.eff + (size_t) st * 32 * g.ktot);
            fa_bytes = (unsigned int)((size_t) 32 * g.ktot * 4);
        };
        auto fetch_next = [&] () {
            if (f_live) {
                const unsigned int sa = min ((unsigned int) f_chunk * A_STEP, fa_bytes), sb = min ((unsigned int) f_chunk * B_STEP, fb_bytes);
                const __amdgpu_buffer_rsrc_t ra_ = make_rsrc (fa_base + sa, fa_bytes - sa), rb_ = make_rsrc (fb_base + sb, fb_bytes - sb);
                VecLoad<4>::load (ra0, ra_, a_off0);
#pragma unroll
                for (int u = 0; u < NB; ++u) VecLoad<VEC>::load (&rb0 [u * VEC], rb_, boff [u]);
                if (++f_chunk == nchunks) { f_chunk = 0; f_within += wgs_per_xcd; open_tile (); }
            }
        };
        auto commit = [&] (auto buf_tag) {
            constexpr int BUF = decltype (buf_tag)::value;
            f32x4 v; v [0] = ra0 [0]; v [1] = ra0 [1]; v [2] = ra0 [2]; v [3] = ra0 [3];
            *reinterpret_cast<f32x4 *> (&As_ [BUF] [adst]) = v;
#pragma unroll
            for (int u = 0; u < NB; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) Bs_ [BUF] [bdst [u] + e * MF_LD] = rb0 [u * VEC + e];
        };

        // chunks this workgroup will consume in total (the matrix waves count the same way)
        int my_tiles = 0;
        { int st, jg; for (int w = rank; tile_at (w, st, jg); w += wgs_per_xcd) ++my_tiles; }
        const int total = my_tiles * nchunks;
        if (total == 0) return;

        open_tile ();
        fetch_next (); commit (std::integral_constant<int, 0> {}); fetch_next ();
        __syncthreads ();
        for (int q = 0; q < total; q += 2) {
            commit (std::integral_constant<int, 1> {}); fetch_next ();       // (past the end: registers are stale, the LDS is not read)
            __syncthreads ();
            if (q + 1 < total) {
                commit (std::integral_constant<int, 0> {}); fetch_next ();
                __syncthreads ();
            }
        }
        return;
    }

    // ---- matrix waves ----
    int my_tiles = 0;
    { int st, jg; for (int w = rank; tile_at (w, st, jg); w += wgs_per_xcd) ++my_tiles; }
    if (my_tiles == 0) return;

    const int arow = (lane & 31) * MF_LD + 4 * (lane >> 5);
    const int col = wave * 32 + (lane & 31);
    const bool col_live = col < NCOLS;
    const int jl = col / CG, c = col - jl * CG;
    const int brow = col * MF_LD + 4 * (lane >> 5);
    // output offset of this lane inside a tile: (period jl, slot 4 * (lane >> 5), channel c); the row's own 0..3 / +8 / +16 / +24
    // slots are immediates of the store
    const unsigned int out_off = (unsigned int)((jl * g.P + 4 * (lane >> 5)) * CG + c) * 4u;

    double sum [16];

    __syncthreads ();                                        // the staging waves have committed chunk 0
    int parity = 0;
    for (int within = rank, t = 0; t < my_tiles; within += wgs_per_xcd, ++t) {
        int st, jg;
        (void) tile_at (within, st, jg);
#pragma unroll
        for (int r = 0; r < 16; ++r) sum [r] = 0.0;

        if (parity) mf_k_walk<1> (As_, Bs_, arow, brow, nchunks, g.band_lo, g.band_hi, sum);
        else mf_k_walk<0> (As_, Bs_, arow, brow, nchunks, g.band_lo, g.band_hi, sum);
        parity ^= nchunks & 1;

        // ---- the tile's outputs: C/D layout of 32x32: row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31
        const unsigned int n_tile = a.n_begin + (unsigned int)(jg * PPW) * g.P + (unsigned int)(st * 32);
        const int rows_valid = min (32, g.P - st * 32);
        const size_t left = (size_t)(a.n_end - n_tile) * CG * 4;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc (a.out + (size_t) n_tile * CG, left > 0xffffff00ull ? 0xffffff00u : (unsigned int) left);
        const int pass_row = PASS ? g.tile_w0 [3 * st + 1] : -1, pass_lin = PASS ? g.tile_w0 [3 * st + 2] : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i_const = (r & 3) + 8 * (r >> 2);      // compile-time part of the slot
            float y = (float) sum [r];
            const int i = i_const + 4 * (lane >> 5);
            if constexpr (PASS) {
                // nearest-filter mode, the position falls exactly on an input sample: the reference copies it (resampler.c:1166-1170)
                if (pass_row == i) y = load_frame (a, INT_MIN, pass_lin + (jg * PPW + jl) * g.Q, c);
            }
            if (col_live && i < rows_valid)                  // (frames at or past n_end: out of the resource's range, dropped)
                __builtin_amdgcn_raw_buffer_store_b32 (__float_as_uint (y), rs_out, (int)(out_off + (unsigned int)(i_const * CG) * 4u), 0, 0);
        }
    }
}

} // namespace

#endif  // !ART_WIDE
