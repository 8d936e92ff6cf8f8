// fir_matrix_i8.hip — the fixed-point form of the matrix-core path (4-byte samples, gfx950): regular launches of the rational-ratio
// GEMM (see fir_matrix.hip) evaluated EXACTLY on the integer matrix cores instead of in chained f32.
//
// Effective filter rows (lerp folded in, fp64) are rounded once to 32-bit fixed point with 30 fraction bits; input samples are
// rounded once to 32-bit BLOCK floating point: the launch's periods are cut into exponent blocks (~10k input frames: a fraction
// of a second; the length depends on the ratio and the filter only, never on the channel count, so a channel's bits do not
// depend on which other channels share its context), and inside a block every channel has its own binary exponent, taken
// from the channel's peak |x| over the frames the block's outputs read, so that the peak lands in [2^29, 2^31 - 2^24) — exact
// for every float sample within 2^-6 of that peak, (peak) x 2^-31 absolute below it: the error model is RELATIVE to the
// channel's level around the output, as float arithmetic's is, not tied to full scale.  Both are written as four signed
// base-256 digits each (d0 most significant: value = sum_i d_i 256^(3-i)).  The dot product of two such numbers is
//     sum_k h_k x_k = sum_{i,j} 256^(6-i-j) sum_k a_i[k] b_j[k],
// and each inner sum over k is one v_mfma_i32_32x32x32_i8 chain: integer, exact, order-free (|sum| < 2^14 * K * pairs < 2^31).
// The 13 digit pairs with i + j <= 4 are kept (five accumulators, one per weight class i + j); the three dropped pairs carry
// less than 2^-36 per tap.  A tile's result is the five class sums combined in fp64 and rounded ONCE to float: half a float ulp
// plus the rows' quantisation (2^-31 per tap: ~3e-9 rms at +-0.5 noise, against the parity bar of 1.2e-7) from the infinitely
// precise dot product, where the reference's own float loop — and the f32 matrix kernel — accumulate ~T roundings
// (tools/sim/int8_scheme.py, tests/test_gpu_fixed_point.py: rms error 0.4 x theirs).  No fp64 flush schedule, no dependence
// on tile shape or summation order.
//
// Cost: 13 integer MFMAs per 32 taps and 32 x 32 outputs where the f32 form needs 16 four-times-slower f32 MFMAs (measured,
// tools/micro/i8_probe.hip: 3.5 Pop/s sustained with operands from LDS beside the staging traffic), paid for with one extra
// pass over the call's input (i8_stage_kernel: quantise + digit planes, memory-bound).
//
// Infinities and NaNs cannot be represented (any finite amplitude can: the exponent follows the peak): the staging pass then
// raises a flag in device memory and the fixed-point kernel's workgroups run the f32 streaming kernel's tile loop instead (no
// host round trip: the device-pointer calls stay asynchronous).
//
// Data layout.  X digit planes, per exponent block e (its own copy of the frames its periods read: neighbouring blocks overlap by
// one period + one window, quantised with each block's own exponents; a tile column stages from ITS period's block): plane p
// (digit d_p), 4-frame block b, channel c -> one
// dword holding frames 4b..4b+3 of that channel (byte q = frame 4b + q): [e][p][b - b0 - e * step][c].  Linear frame lin
// (history ++ input) lives in block (lin + pad) / 4, pad = MfmaGeom.head_pad zero frames in front of linear frame 0 (64, more where a period's input
// is longer than half a window: a launch anchored on the canonical period starts up to Q frames in front of its first output).  A tile
// takes every g-th period (g = 4 / gcd (Q, 4)) so that all its columns start at the same offset r in their first block; r is
// absorbed by the tile's filter rows, which exist once per (slot tile, residue) shifted r taps to the right.  A digit planes:
// [slot tile * g + residue][chunk][p][row][32 taps]: the 4 KB a workgroup stages per chunk are contiguous.
#include "fir_matrix_stream.hip.h"
#include <atomic>
#include <cstdlib>

#if !ART_WIDE

namespace {

typedef int i32x4 __attribute__ ((ext_vector_type (4)));
typedef int i32x16 __attribute__ ((ext_vector_type (16)));

constexpr int I8_KC = 32;                 // taps per staged chunk = K of one integer MFMA
constexpr int I8_PITCH = 48;              // LDS bytes per (row or column, plane) of a chunk: 32 + 16 pad, conflict-free b128 reads
constexpr int I8_COLS = 128;              // columns per workgroup
constexpr int I8_MAX_PPW = 64;
constexpr float I8_SCALE = 1073741824.0f; // 2^30: the filter rows' fixed point
constexpr float I8_LIMIT = 1.98f;         // |row value| the digits can hold: 0x7f7f7f7f / 2^30 = 1.98437..., rounded down

struct I8Geom {
    int tr;                               // rows (slots) per tile: 32, or 64 (fir_i8_slab_kernel: two 32-row register tiles on one K origin)
    int cols;                             // columns per tile: I8_COLS, or SL_COLS (fir_i8_slab_kernel)
    int tiles;                            // ceil (P / tr)
    int ktot;                             // K columns of a tile: T + the span of its rows' window starts + alignment, a multiple of I8_KC
    int g;                                // period stride inside a tile
    int gq4;                              // g * Q / 4: blocks between consecutive columns' periods
    int super_groups;                     // groups of g * ppw consecutive periods
    int sg_per_xcd;
    unsigned char *a_planes;
    unsigned long long *a_masks;          // [digit plane 0, 1][variant][row]: bit c set = chunk c of the row has a non-zero digit in that plane
    unsigned long long *tile_masks;       // [digit plane 0, 1][variant][32-row half]: the OR over the half's rows (quantise launch; read by fir_i8_slab_kernel)
    int mask_words;                       // tiles * g * tr: the second plane's masks sit this many words behind the first's
    const unsigned char *x_planes;        // (written through x_planes_w by the staging pass)
    unsigned int *x_planes_w;
    // exponent blocks: block e = periods [e * eb_periods, (e + 1) * eb_periods) of the launch; its planes hold 4-frame blocks
    // [b0 + e * eb_step, b0 + e * eb_step + eb_blocks) — everything those periods' windows read, from any slot tile
    int eb_periods, ebs;
    int eb_blocks, eb_step;
    int b0;                               // block of the launch's first window start (host: the reference's position arithmetic)
    int cgrp;                             // channels one staging workgroup handles (a power of two <= C)
    int slices;                           // staging workgroups per (exponent block, channel group): consecutive slices of its 4-frame blocks
    int slice_blocks;                     // blocks per slice
    unsigned int *peaks;                  // [ebs][slices][C] bits of each slice's peak |x| per channel (peak pass -> quantise pass)
    unsigned int eb_plane_bytes;          // eb_blocks * C * 4
    size_t x_bytes;                       // ebs * 4 * eb_plane_bytes
    int *flag; int epoch;                 // *flag == epoch: this launch cannot run in fixed point (set by the staging pass)
    int *shifts;                          // [ebs][C]: the block's samples of channel c are quantised as rint (x * 2^shift)
    unsigned char *parts;                 // slabs: parts of tiles cut between workgroups (fir_i8_slab_kernel), behind the planes
    int rows_cached;                      // the rows' planes, masks and tables are in place (no row workgroups in the peak launch)
    int rows_table;                       // the row workgroups build the rows of the CANONICAL period (rows kept across calls): slot k sits where the reference's
    double tb_base; int tb_lin, tb_w;     // arithmetic puts output tb_n0 + k of an epoch of offset tb_base and ring-to-linear shift tb_lin, tb_w frames further on
    unsigned int tb_n0;
    int jr_rot;                           // ... built by a launch whose windows started jr_rot residues further on: period residue jr of this launch
                                          // stages the rows of residue (jr + jr_rot) mod g
};

// binary exponent for a channel whose peak magnitude has these float bits: peak * 2^shift in [2^29, 2^31 - 2^24) — as large as the
// digits hold: four signed digits reach 0x7f7f7f7f = (2^31 - 2^23) - 0x8081 and no further (q + 0x80808080 must not carry out of
// the dword), so a peak whose six leading mantissa bits are all ones (>= 2^31 - 2^24 at the larger exponent) takes one bit less.
// Zero and denormal peaks take the smallest normal's exponent.
__device__ __forceinline__ int shift_of_peak (unsigned int bits)
{
    const int e = max ((int)(bits >> 23), 1);
    return 157 - e - ((bits & 0x7e0000u) == 0x7e0000u ? 1 : 0);
}

constexpr int I8_STAGE_THREADS = 256;     // workgroup of the two staging passes
#ifndef I8_STAGE_UNITS
#define I8_STAGE_UNITS 4
#endif
constexpr int I8_STAGE_K = I8_STAGE_UNITS;  // units (4 frames of one channel) per staging thread (-DI8_STAGE_UNITS=n: tools/micro/stage_units_ab.sh)

// digits of a fixed-point value as one dword: byte 3 = d0 ... byte 0 = d3, each signed
__device__ __forceinline__ unsigned int digits_of (int q) { return ((unsigned int) q + 0x80808080u) ^ 0x80808080u; }

// the class sums of one output as ONE exact integer (|sum_s| < 2^31, class 0 < 2^24: |total| < 2^57), and its single rounding
__device__ __forceinline__ long long i8_total (int s0, int s1, int s2, int s3, int s4)
{
    long long v = (long long) s0;
    v = (v << 8) + (long long) s1; v = (v << 8) + (long long) s2; v = (v << 8) + (long long) s3; v = (v << 8) + (long long) s4;
    return v;
}
__device__ __forceinline__ float i8_round (long long v, int out_exp) { return (float) __builtin_ldexp ((double) v, out_exp); }

// four consecutive taps' digit dwords -> four plane dwords (plane p, byte q = digit p of tap q)
__device__ __forceinline__ void to_planes (const unsigned int (&s) [4], unsigned int (&pl) [4])
{
    const unsigned int t01_hi = __builtin_amdgcn_perm (s [1], s [0], 0x06020703u);     // [s0.b3, s1.b3, s0.b2, s1.b2]
    const unsigned int t01_lo = __builtin_amdgcn_perm (s [1], s [0], 0x04000501u);     // [s0.b1, s1.b1, s0.b0, s1.b0]
    const unsigned int t23_hi = __builtin_amdgcn_perm (s [3], s [2], 0x06020703u);
    const unsigned int t23_lo = __builtin_amdgcn_perm (s [3], s [2], 0x04000501u);
    pl [0] = __builtin_amdgcn_perm (t23_hi, t01_hi, 0x05040100u);
    pl [1] = __builtin_amdgcn_perm (t23_hi, t01_hi, 0x07060302u);
    pl [2] = __builtin_amdgcn_perm (t23_lo, t01_lo, 0x05040100u);
    pl [3] = __builtin_amdgcn_perm (t23_lo, t01_lo, 0x07060302u);
}

// The X side of the two staging passes: a workgroup takes one slice of one exponent block's frames (x a channel group), a thread
// I8_STAGE_K units of 4 frames x 1 channel.  All addresses are one per-thread offset + a scalar: raw buffer loads / stores, whose
// range check also supplies the zeros past the call's end.
//   PEAK pass: per-channel peak |x| of the slice (magnitudes compared as unsigned bit patterns: monotone for finite values,
//       infinities above them, NaNs above those) -> a table entry per (block, slice, channel);
//   quantise pass (the launch behind it): the block's exponent from the maximum over its slices' entries (no device-wide
//       atomics: contended ones cost the peak pass 12 us), digit planes of the slice.
// HEAD: the slice starts inside the history (two source arrays) or inside the head the stand-by's f32 tiles stage from.
template <bool HEAD, bool PEAK>
__device__ __forceinline__ void stage_slice (const ArtFirArgs &a, const MfmaGeom &g, const I8Geom &q, int eb, int cgi, int slice, int gb0)
{
    __shared__ unsigned int s_peak [32];
    const int tid = threadIdx.x;
    const int c = cgi * q.cgrp + (tid & (q.cgrp - 1));
    const int per_k = I8_STAGE_THREADS / q.cgrp;               // blocks between a thread's consecutive units
    const int bl0 = slice * q.slice_blocks + tid / q.cgrp;      // (block index inside the region)
    const int bl_end = min ((slice + 1) * q.slice_blocks, q.eb_blocks);
    const int lin0 = 4 * (gb0 + bl0) - g.head_pad;              // the thread's first frame (>= -head_pad)
    const unsigned int row = (unsigned int) a.C * 4u;           // bytes per frame
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc (a.in, (unsigned int)((size_t) a.in_frames * row));
    const __amdgpu_buffer_rsrc_t r_hist = make_rsrc (a.hist, (unsigned int) a.H * row);
    if (tid < 32) s_peak [tid] = 0u;
    __syncthreads ();
    if constexpr (!PEAK) {                                      // the block's peaks: maximum over its slices (issued ahead of the sample loads)
        for (int i = tid; i < q.slices * q.cgrp; i += I8_STAGE_THREADS) {
            const int sl = i / q.cgrp, cc = i - sl * q.cgrp;
            const unsigned int pkv = q.peaks [(size_t)(eb * q.slices + sl) * a.C + cgi * q.cgrp + cc];
            if (pkv) atomicMax (&s_peak [cc], pkv);
        }
    }
    if constexpr (PEAK && !HEAD) {
        // the peak pass of a plain slice needs no frame x channel transposition: whole 16-byte vectors (4 channels of one frame)
        if (q.cgrp >= 4) {
            const int vpf = q.cgrp >> 2, cq = tid & (vpf - 1), per_f = I8_STAGE_THREADS / vpf;
            const int f0 = 4 * (gb0 + slice * q.slice_blocks) - g.head_pad - a.H + tid / vpf, f_end = 4 * (gb0 + bl_end) - g.head_pad - a.H;
            const unsigned int voff = (unsigned int)(f0 * a.C + cgi * q.cgrp + 4 * cq) * 4u;
            unsigned int m4 [4] = { 0u, 0u, 0u, 0u };
#pragma unroll
            for (int k = 0; k < I8_STAGE_K; ++k)
                if (f0 + k * per_f < f_end) {
                    const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128 (r_in, (int) voff, (int)((unsigned int)(k * per_f) * row), 0);
                    m4 [0] = max (m4 [0], x.x & 0x7fffffffu); m4 [1] = max (m4 [1], x.y & 0x7fffffffu);
                    m4 [2] = max (m4 [2], x.z & 0x7fffffffu); m4 [3] = max (m4 [3], x.w & 0x7fffffffu);
                }
            for (int off = 32; off >= vpf; off >>= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) m4 [e] = max (m4 [e], (unsigned int) __shfl_xor ((int) m4 [e], off));
            if ((tid & 63) < vpf)
#pragma unroll
                for (int e = 0; e < 4; ++e) if (m4 [e]) atomicMax (&s_peak [4 * cq + e], m4 [e]);
            __syncthreads ();
            if (tid < q.cgrp) q.peaks [(size_t)(eb * q.slices + slice) * a.C + c] = s_peak [tid];
            return;
        }
    }
    unsigned int v [I8_STAGE_K] [4];
    unsigned int m = 0u;
    const unsigned int voff_in = (unsigned int)((lin0 - a.H) * a.C + c) * 4u;       // (plain slices: lin0 >= H)
#pragma unroll
    for (int k = 0; k < I8_STAGE_K; ++k) {
        const bool live = bl0 + k * per_k < bl_end;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            unsigned int x = 0u;
            if (live) {
                if constexpr (HEAD) {
                    const int lin = lin0 + 4 * k * per_k + t;
                    const unsigned int xi = __builtin_amdgcn_raw_buffer_load_b32 (r_in, lin >= a.H ? (int)((unsigned int)((lin - a.H) * a.C + c) * 4u) : -16, 0, 0);
                    const unsigned int xh = __builtin_amdgcn_raw_buffer_load_b32 (r_hist, lin >= 0 && lin < a.H ? (int)((unsigned int)(lin * a.C + c) * 4u) : -16, 0, 0);
                    x = xi | xh;                                // (one of the two is out of its array's range: zero)
                }
                else x = __builtin_amdgcn_raw_buffer_load_b32 (r_in, (int) voff_in, (int)((unsigned int)(4 * k * per_k + t) * row), 0);
            }
            v [k] [t] = x;
            m = max (m, x & 0x7fffffffu);
        }
    }
    if constexpr (PEAK) {
        // threads of one channel: lanes cgrp apart first, then one LDS atomic per wave and channel
        for (int off = 32; off >= q.cgrp; off >>= 1) m = max (m, (unsigned int) __shfl_xor ((int) m, off));
        if ((tid & 63) < q.cgrp && m) atomicMax (&s_peak [tid & (q.cgrp - 1)], m);
        __syncthreads ();
        if (tid < q.cgrp) q.peaks [(size_t)(eb * q.slices + slice) * a.C + c] = s_peak [tid];
        return;
    }
    __syncthreads ();
    const unsigned int pk = s_peak [tid & (q.cgrp - 1)];
    // the channel's exponent in this block: |x| <= peak, so |x * 2^shift| < 2^31 - 2^24 <= 0x7f7f7f7f and the scaling itself is exact (v_ldexp_f32)
    const int shift = shift_of_peak (pk);
    if (tid < q.cgrp && slice == 0) {
        q.shifts [eb * a.C + c] = shift;
        if (pk >= 0x7f800000u) *q.flag = q.epoch;               // an infinity or a NaN among the block's frames: no exponent holds it
    }
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc (q.x_planes_w + (size_t) eb * q.eb_plane_bytes, 4u * q.eb_plane_bytes);    // (the block's 4 planes)
    const unsigned int voff_out = (unsigned int)(bl0 * a.C + c) * 4u;       // (bl0 counts from the region's start)
    // (the call's head as one contiguous float array, for the stand-by's tiles that reach into the history)
    const __amdgpu_buffer_rsrc_t r_head = make_rsrc (g.head, g.head ? (unsigned int) g.head_frames * row : 0u);
    const unsigned int voff_head = (unsigned int)((lin0 + g.head_pad) * a.C + c) * 4u;
#pragma unroll
    for (int k = 0; k < I8_STAGE_K; ++k) {
        if (bl0 + k * per_k < bl_end) {
            unsigned int sd [4], pl [4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (HEAD) __builtin_amdgcn_raw_buffer_store_b32 (v [k] [t], r_head, (int) voff_head, (int)((unsigned int)(4 * k * per_k + t) * row), 0);
                sd [t] = digits_of (__float2int_rn (ldexpf (__uint_as_float (v [k] [t]), shift)));   // (a channel with an infinity or a NaN: garbage, the launch is flagged)
            }
            to_planes (sd, pl);
#pragma unroll
            for (int pn = 0; pn < 4; ++pn) __builtin_amdgcn_raw_buffer_store_b32 (pl [pn], r_out, (int) voff_out, (int)((unsigned int) pn * q.eb_plane_bytes + (unsigned int)(k * per_k) * row), 0);
        }
    }
}

// Staging, two launches.  The first (PEAK): X workgroups find the exponent blocks' peaks; behind them in the same grid, one
// workgroup per effective row (same arithmetic as mfma_prepare_kernel up to the rounding: the fp64 blend goes straight to fixed
// point, not through float) -> A digit planes, masks and the tables the streaming kernels read.  The second: X workgroups only,
// the same slices again (their second read comes from the Infinity Cache) -> X digit planes.
template <bool INTERP, bool PEAK>
__global__ __launch_bounds__ (I8_STAGE_THREADS)
void i8_stage_kernel (ArtFirArgs a, ArtSegTable segs, MfmaGeom g, I8Geom q)
{
    const int tid = threadIdx.x;
    // (PEAK launch: the row workgroups come first in the grid — latency chains, they finish under the X workgroups' traffic)
    // (the residue-0 row workgroups write the f32 tables too — 320 workgroups fewer, 1.1 us of the peak pass)
    const unsigned int t_wgs = 0u;
    // slot k of the launch's first period: its position by the reference's arithmetic — or, for rows kept across calls, from the canonical
    // period's constants (artfir_i8_launch), which makes the rows a function of those alone
    auto slot_pos = [&] (int k) -> Pos {
        if (q.rows_table) {                                   // (locate () on the canonical period's own constants: what the host evaluated, to the bit)
            const unsigned int n = q.tb_n0 + (unsigned int) k;
            const double step = n ? (double) n / a.ratio : 0.0;
            const double off = q.tb_base + step;
            const double whole = floor (off);
            Pos t;
            double fr = off - whole;
            fr = fr * (double) a.F;
            if (INTERP) { t.fi = (int) floor (fr); t.frac = fr - (double) t.fi; }
            else { t.fi = (int) floor (fr + 0.5); t.frac = 0.0; }
            t.ip = (int) whole + q.tb_lin + q.tb_w;
            return t;
        }
        return locate<INTERP> (a, segs, a.n_begin + (unsigned int) k);
    };
    const unsigned int a_wgs = PEAK && !q.rows_cached ? t_wgs + (unsigned int)(q.tiles * q.g * q.tr) : 0u;
    // what mfma_prepare_kernel leaves for the f32 streaming kernels (that kernel is not launched at all then), for row `row` of the
    // 32-row slot tile `st`: the effective row in float, the canonical position, the tile's origin and pass-through rows
    auto write_tables = [&] (int st, int row) {
        const int rows_valid = min (32, g.P - st * 32);
        const Pos p0 = slot_pos (st * 32);
        const Pos p = slot_pos (st * 32 + min (row, rows_valid - 1));
        const float *h0 = a.bank + (size_t) p.fi * a.T;
        __shared__ unsigned int s_pass;
        if (row == 0) {
            if (tid == 0) s_pass = 0u;
            __syncthreads ();
            if (tid == 0) {
                g.tile_w0 [3 * st] = p0.ip - a.T / 2 + 1;
                if (st == 0) a.fix_count [0] = 0;
            }
            // nearest-filter mode without a low-pass: the slots whose position falls exactly on an input sample (one bit per row;
            // their sample index is the row's canonical ip + fi / F)
            if (!INTERP && !a.lowpass && tid < rows_valid) {
                const Pos pq = slot_pos (st * 32 + tid);
                if ((pq.fi % a.F) == 0) atomicOr (&s_pass, 1u << tid);
            }
            __syncthreads ();
            if (tid == 0) { g.tile_w0 [3 * st + 1] = (int) s_pass; g.tile_w0 [3 * st + 2] = 0; }
        }
        if (tid == 0) { g.canon_ip [st * 32 + row] = p.ip; g.canon_fi [st * 32 + row] = p.fi; g.canon_frac [st * 32 + row] = p.frac; }
        float *dst = g.eff + (size_t)(st * 32 + row) * g.ktot;
        for (int k = tid; k < g.ktot; k += 256) {
            const int tap = k - (p.ip - p0.ip);
            float cf = 0.0f;
            if (tap >= 0 && tap < a.T) {
                if (INTERP) {
                    const double left = (double) h0 [tap] * (1.0 - p.frac);
                    const double right = (double) h0 [tap + a.T] * p.frac;
                    cf = (float)(left + right);
                }
                else cf = h0 [tap];
            }
            dst [k] = cf;
        }
    };
    if (blockIdx.x < t_wgs) {                                 // ---- table role (64-slot tiles: one workgroup per row of the 32-row slot tiles)
        write_tables ((int) blockIdx.x >> 5, (int) blockIdx.x & 31);
        return;
    }
    if (blockIdx.x < a_wgs) {
        // ---- planes role: one workgroup per (tile, residue) variant and row -> the row's digit planes (same arithmetic as
        // mfma_prepare_kernel up to the rounding: the fp64 blend goes straight to fixed point, not through float)
        const int ab = (int)(blockIdx.x - t_wgs);
        const int variant = ab / q.tr, row = ab - variant * q.tr;
        const int st = variant / q.g, jr = variant - st * q.g;
        if (jr == 0 && (q.tr == 32 ? st : 2 * st + (row >> 5)) < g.slot_tiles) write_tables (q.tr == 32 ? st : 2 * st + (row >> 5), row & 31);
        const int rows_valid = min (q.tr, g.P - st * q.tr);
        const Pos p0 = slot_pos (st * q.tr);
        const Pos p = slot_pos (st * q.tr + min (row, rows_valid - 1));
        const float *h0 = a.bank + (size_t) p.fi * a.T;
        // K column 0 of this tile family sits r frames before the first slot's window (the start of its 4-frame block)
        const int r = max (p0.ip - a.T / 2 + 1 + jr * g.Q + g.head_pad, 0) & 3;
        const int shift = p.ip - p0.ip + r;
        bool bad = false;
        __shared__ unsigned long long s_mask, s_mask1;
        if (tid == 0) { s_mask = 0ull; s_mask1 = 0ull; }
        __syncthreads ();
        unsigned long long mine = 0ull, mine1 = 0ull;
        // [variant][chunk][plane][16-tap half][row][16 taps]: the tr * 128 bytes a workgroup stages per chunk are contiguous
        const int chunk_bytes = q.tr * 128, plane_bytes = q.tr * 32, half_bytes = q.tr * 16;
        unsigned char *base = q.a_planes + (size_t) variant * (q.ktot / I8_KC) * chunk_bytes + row * 16;
        for (int k4 = tid * 4; k4 < q.ktot; k4 += 1024) {
            unsigned int s [4], pl [4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int tap = k4 + t - shift;
                double v = 0.0;
                if (row < rows_valid && tap >= 0 && tap < a.T) {
                    if (INTERP) {
                        const double left = (double) h0 [tap] * (1.0 - p.frac);
                        const double right = (double) h0 [tap + a.T] * p.frac;
                        v = left + right;
                    }
                    else v = (double) h0 [tap];
                }
                if (!(fabs (v) < (double) I8_LIMIT)) { bad = true; v = 0.0; }
                s [t] = digits_of ((int) rint (v * (double) I8_SCALE));
            }
            to_planes (s, pl);
#pragma unroll
            for (int pn = 0; pn < 4; ++pn) *reinterpret_cast<unsigned int *> (base + (size_t)(k4 >> 5) * chunk_bytes + pn * plane_bytes + ((k4 >> 4) & 1) * half_bytes + (k4 & 15)) = pl [pn];
            if (pl [0]) mine |= 1ull << (k4 >> 5);
            if (pl [1]) mine1 |= 1ull << (k4 >> 5);
        }
        if (mine) atomicOr (&s_mask, mine);
        if (mine1) atomicOr (&s_mask1, mine1);
        __syncthreads ();
        if (tid == 0) { q.a_masks [variant * q.tr + row] = s_mask; q.a_masks [q.mask_words + variant * q.tr + row] = s_mask1; }
        if (bad) *q.flag = q.epoch;
        return;
    }
    // ---- X role ----
    const int groups = a.C / q.cgrp;
    const int xid = (int)(blockIdx.x - a_wgs);
    if (!PEAK && xid >= q.ebs * groups * q.slices) {
        // (behind the X workgroups of the quantise launch: the rows' masks of the launch before it, per 32-row register tile)
        const int e = (xid - q.ebs * groups * q.slices) * I8_STAGE_THREADS + tid, halves = q.tr / 32;
        if (e < q.tiles * q.g * halves) {
            unsigned long long m = 0ull, m1 = 0ull;
            for (int r = 0; r < 32; ++r) { m |= q.a_masks [(e / halves) * q.tr + (e % halves) * 32 + r]; m1 |= q.a_masks [q.mask_words + (e / halves) * q.tr + (e % halves) * 32 + r]; }
            q.tile_masks [e] = m; q.tile_masks [q.tiles * q.g * halves + e] = m1;
        }
        return;
    }
    const int slice = xid % q.slices, xb = xid / q.slices, eb = xb / groups, cgi = xb - eb * groups;
    const int gb0 = q.b0 + eb * q.eb_step;                      // the region's first block
    // (slices that start inside the history ++ head span — the first few — read two arrays and leave the stand-by its head)
    if (4 * (gb0 + slice * q.slice_blocks) - g.head_pad < max (a.H, g.head_frames - g.head_pad)) stage_slice<true, PEAK> (a, g, q, eb, cgi, slice, gb0);
    else stage_slice<false, PEAK> (a, g, q, eb, cgi, slice, gb0);
}

template <int VEC> struct PlaneLoad;
template <> struct PlaneLoad<1> { static __device__ __forceinline__ void load (unsigned int *dst, __amdgpu_buffer_rsrc_t r, unsigned int off, unsigned int soff) {
    dst [0] = __builtin_amdgcn_raw_buffer_load_b32 (r, (int) off, (int) soff, 0); } };
template <> struct PlaneLoad<2> { static __device__ __forceinline__ void load (unsigned int *dst, __amdgpu_buffer_rsrc_t r, unsigned int off, unsigned int soff) {
    u32x2 v = __builtin_amdgcn_raw_buffer_load_b64 (r, (int) off, (int) soff, 0); dst [0] = v.x; dst [1] = v.y; } };
template <> struct PlaneLoad<4> { static __device__ __forceinline__ void load (unsigned int *dst, __amdgpu_buffer_rsrc_t r, unsigned int off, unsigned int soff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128 (r, (int) off, (int) soff, 0); dst [0] = v.x; dst [1] = v.y; dst [2] = v.z; dst [3] = v.w; } };

// Persistent workgroups over the tiles of a regular launch, as fir_mfma_stream_kernel: waves 4-7 stage (global -> registers ->
// LDS, two chunks ahead, one chunk stream across all of the workgroup's tiles), waves 0-3 multiply (32 slots x 32 columns each).
// A tile = (slot tile st, residue jr, super group sg): slots st*32.., periods sg*g*PPW + jr + g*m for column group m.
// the f32 streaming kernel's tile loop on this kernel's workgroups (the LDS handed in: the digit buffers, big enough for its
// 2 x 32 rows and 2 x 128 columns of MF_LD floats); inlined: as a real call it cost the main loop 12 spilt registers and a stack frame, 72 -> 59 Gsamples/s
template <int CG, bool PASS>
__device__ __forceinline__ void stand_by_tiles (const ArtFirArgs &a, const MfmaGeom &g, int wgs_per_xcd, float (&As_) [2] [32 * MF_LD], float (&Bs_) [2] [MF_COLS * MF_LD])
{
    constexpr int THREADS = 2 * MF_THREADS;
    constexpr int PPW = MF_COLS / CG > MF_MAX_PPW ? MF_MAX_PPW : MF_COLS / CG;
    constexpr int NCOLS = PPW * CG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 4;
    const int pt = tid & (MF_THREADS - 1);
    (void) THREADS;
#include "fir_matrix_stream_body.inc"
}

template <int CG, bool PASS>
__global__ __launch_bounds__ (2 * MF_THREADS) __attribute__ ((amdgpu_waves_per_eu (4, 4)))
void fir_i8_stream_kernel (ArtFirArgs a, MfmaGeom g, I8Geom q, int wgs_per_xcd)
{
    constexpr int THREADS = 2 * MF_THREADS;
    constexpr int PPW = I8_COLS / CG > I8_MAX_PPW ? I8_MAX_PPW : I8_COLS / CG;
    constexpr int NCOLS = PPW * CG;
    __shared__ __attribute__ ((aligned (16))) unsigned char As_ [2] [4] [32 * I8_PITCH];
    __shared__ __attribute__ ((aligned (16))) unsigned char Bs_ [2] [4] [I8_COLS * I8_PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 4;
    const int pt = tid & (MF_THREADS - 1);

    const unsigned int stream_blocks = 8u * (unsigned int) wgs_per_xcd;
    if (blockIdx.x >= stream_blocks) {                        // extra workgroups: the history roll (as in fir_mfma_kernel)
        if (a.roll_dst) {
            const int e = (int)(blockIdx.x - stream_blocks) * THREADS + tid;
            if (e < a.H * a.C) {
                const int f = e / a.C, c = e - f * a.C, lin = a.roll_appended + f;
                float v = 0.0f;
                if (lin < a.H) v = a.hist [(size_t) lin * a.C + c];
                else if (a.in && lin - a.H < a.in_frames) v = a.in [(size_t)(lin - a.H) * a.C + c];
                a.roll_dst [e] = v;
            }
        }
        return;
    }
    // Samples the digits cannot hold (flag raised by the staging pass; uniform): the launch is produced in f32 by the streaming
    // kernel's own tile loop on this kernel's workgroups and LDS (its 2 x 32 rows and 2 x 128 columns of 36 floats fit the
    // digit buffers), from the tables the staging pass has left for it — the bits of fir_mfma_stream_kernel.
    if (*q.flag == q.epoch) {
        static_assert (sizeof (As_) >= 2 * 32 * MF_LD * sizeof (float) && sizeof (Bs_) >= 2 * MF_COLS * MF_LD * sizeof (float), "the f32 tiles live in the digit buffers");
        stand_by_tiles<CG, PASS> (a, g, wgs_per_xcd, *reinterpret_cast<float (*) [2] [32 * MF_LD]> (&As_ [0] [0] [0]), *reinterpret_cast<float (*) [2] [MF_COLS * MF_LD]> (&Bs_ [0] [0] [0]));
        return;
    }

    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;
    const int tiles_per_xcd = q.sg_per_xcd * q.g * q.tiles;
    const int nchunks = q.ktot / I8_KC;

    // tile `within` of this XCD's list -> (slot tile, first period); false if the tile holds no output of the launch
    auto tile_at = [&] (int within, int &st, int &j0) -> bool {
        st = within % q.tiles;
        const int t2 = within / q.tiles, jr = t2 % q.g, sg = xcd * q.sg_per_xcd + t2 / q.g;
        if (sg >= q.super_groups) return false;
        j0 = sg * q.g * PPW + jr;
        return a.n_begin + (unsigned int) j0 * g.P + (unsigned int)(st * 32) < a.n_end;
    };
    int my_tiles = 0;
    { int st, j0; for (int w = rank; w < tiles_per_xcd; w += wgs_per_xcd) my_tiles += tile_at (w, st, j0) ? 1 : 0; }
    if (my_tiles == 0) return;

    if (NCOLS < I8_COLS)                                      // unused columns stay zero for the whole kernel
        for (int e = tid; e < 2 * 4 * (I8_COLS - NCOLS) * I8_PITCH / 4; e += THREADS) {
            const int per = (I8_COLS - NCOLS) * I8_PITCH / 4, bp = e / per, r = e - bp * per;
            reinterpret_cast<unsigned int *> (&Bs_ [bp >> 2] [bp & 3] [NCOLS * I8_PITCH]) [r] = 0u;
        }

    if (loader) {
        constexpr int VEC = CG >= 4 ? 4 : (CG == 2 ? 2 : 1);
        constexpr int VPF = CG / VEC, VPP = (I8_KC / 4) * VPF, NB = (PPW * VPP) / MF_THREADS;
        static_assert ((PPW * VPP) % MF_THREADS == 0 && NB >= 1, "every staging thread moves NB vectors per plane and chunk");
        constexpr unsigned int A_STEP = 4096u, B_STEP = (I8_KC / 4) * CG * 4u;
        // A: thread -> (plane, 16-tap half, row), consecutive threads on consecutive 16 bytes of the chunk's 4 KB
        const int a_plane = pt >> 6, a_half = (pt >> 5) & 1, a_row = pt & 31;
        const unsigned int a_off0 = (unsigned int) pt * 16u;
        const int adst = a_plane * (32 * I8_PITCH) + a_row * I8_PITCH + a_half * 16;
        unsigned int boff [NB], bdel [NB]; int bdst [NB], bper [NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int v = pt + u * MF_THREADS;
            const int m = v / VPP, rem = v % VPP, kb = rem / VPF, cv = rem % VPF;
            boff [u] = (unsigned int)((m * q.gq4 + kb) * CG + cv * VEC) * 4u;       // (the tile's first block sits in the resource base)
            bdst [u] = (m * CG + cv * VEC) * I8_PITCH + kb * 4;
            bper [u] = m * q.g;                                                   // the unit's column: this many periods behind the tile's first
            bdel [u] = 0u;
        }
        // a column whose period lies d exponent blocks behind the tile's first column stages from that block's own planes: d regions
        // further on, where the same 4-frame block sits d * eb_step blocks earlier
        const unsigned int x_total = 4u * q.eb_plane_bytes;
        const unsigned int eb_hop = x_total - (unsigned int) q.eb_step * (unsigned int)(CG * 4);
        // two register stages: the loads of chunk c + 3 are issued while those of c + 2 are still in flight (a chunk is ~0.5 us
        // of matrix work, less than a loaded L2 round trip: with one stage the kernel ran at the memory latency, 139 us)
        unsigned int ra0 [2] [4], rb0 [2] [4] [NB * VEC];

        int f_within = rank - wgs_per_xcd, f_chunk = 0;
        bool f_live = false;
        const unsigned char *fa_base = nullptr, *fb_base = nullptr;
        unsigned int fa_bytes = 0, fb_bytes = 0;
        auto open_tile = [&] () {                             // next tile of this workgroup's list that holds outputs
            int st = 0, j0 = 0;
            f_live = false;
            for (f_within += wgs_per_xcd; f_within < tiles_per_xcd; f_within += wgs_per_xcd)
                if (tile_at (f_within, st, j0)) { f_live = true; break; }
            if (!f_live) return;
            // (readfirstlane: the table entry arrives in a vector register, and a resource built from it would make every load a
            // waterfall loop; the value is the same in all lanes)
            const int la = max (__builtin_amdgcn_readfirstlane (g.tile_w0 [3 * st]) + g.w_shift + j0 * g.Q + g.head_pad, 0);
            // the tile's exponent block and its first 4-frame block inside that block's own planes
            const int eb = j0 / q.eb_periods;
            unsigned int skip = (unsigned int) max ((la >> 2) - q.b0 - eb * q.eb_step, 0) * (unsigned int)(CG * 4);
            if (skip > q.eb_plane_bytes) skip = q.eb_plane_bytes;
            const size_t from = (size_t) eb * x_total + skip;
            fb_base = q.x_planes + from; fb_bytes = (unsigned int) min (q.x_bytes - from, (size_t) 0xfffffff0u);
#pragma unroll
            for (int u = 0; u < NB; ++u) bdel [u] = (unsigned int)((j0 + bper [u]) / q.eb_periods - eb) * eb_hop;
            fa_bytes = (unsigned int) nchunks * 4096u;
            fa_base = q.a_planes + (size_t)(st * q.g + (j0 + q.jr_rot) % q.g) * fa_bytes;
        };
        auto fetch_next = [&] (auto set_tag) {
            constexpr int SET = decltype (set_tag)::value;
            if (f_live) {
                const unsigned int sa = (unsigned int) f_chunk * A_STEP, sb = min ((unsigned int) f_chunk * B_STEP, fb_bytes);
                const __amdgpu_buffer_rsrc_t ra_ = make_rsrc (fa_base + sa, fa_bytes - sa), rb_ = make_rsrc (fb_base + sb, fb_bytes - sb);
                {
                    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128 (ra_, (int) a_off0, 0, 0);
                    ra0 [SET] [0] = v.x; ra0 [SET] [1] = v.y; ra0 [SET] [2] = v.z; ra0 [SET] [3] = v.w;
                }
#pragma unroll
                for (int pn = 0; pn < 4; ++pn)
#pragma unroll
                    for (int u = 0; u < NB; ++u) PlaneLoad<VEC>::load (&rb0 [SET] [pn] [u * VEC], rb_, boff [u] + bdel [u], (unsigned int) pn * q.eb_plane_bytes);
                if (++f_chunk == nchunks) { f_chunk = 0; open_tile (); }
            }
        };
        // register stage s -> LDS buffer s (chunk c lives in stage and buffer c & 1)
        auto commit = [&] (auto set_tag) {
            constexpr int SET = decltype (set_tag)::value;
            i32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v [i] = (int) ra0 [SET] [i];
            *reinterpret_cast<i32x4 *> (&As_ [SET] [0] [adst]) = v;
#pragma unroll
            for (int pn = 0; pn < 4; ++pn)
#pragma unroll
                for (int u = 0; u < NB; ++u)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) *reinterpret_cast<unsigned int *> (&Bs_ [SET] [pn] [bdst [u] + e * I8_PITCH]) = rb0 [SET] [pn] [u * VEC + e];
        };
        const std::integral_constant<int, 0> S0 {};
        const std::integral_constant<int, 1> S1 {};

        const int total = my_tiles * nchunks;
        open_tile ();
        fetch_next (S0); fetch_next (S1);
        commit (S0); fetch_next (S0);
        __syncthreads ();
        for (int c = 0; c < total; c += 2) {
            commit (S1); fetch_next (S1);                     // (past the end: registers are stale, the LDS is not read)
            __syncthreads ();
            if (c + 1 < total) {
                commit (S0); fetch_next (S0);
                __syncthreads ();
            }
        }
        return;
    }

    // ---- matrix waves ----
    {   // the two workgroups of a CU share each SIMD's matrix pipe: left alone their matrix waves fall into step (both multiply, then
        // both wait for the LDS and the barrier); different issue priorities make them alternate instead
        const unsigned int hw_id = __builtin_amdgcn_s_getreg ((4 - 1) << 11 | 16 << 6 | 4);      // HW_ID.TG_ID: bits 19:16
        if (hw_id & 1u) __builtin_amdgcn_s_setprio (3); else __builtin_amdgcn_s_setprio (0);
    }
    const int aoff = (lane & 31) * I8_PITCH + (lane >> 5) * 16;
    const int col = wave * 32 + (lane & 31);
    const bool col_live = col < NCOLS;
    const int jl = col / CG, c = col - jl * CG;
    const int boff = col * I8_PITCH + (lane >> 5) * 16;
    // output offset of this lane inside a tile: (period jl * g, slot 4 * (lane >> 5), channel c); the row's own 0..3 / +8 / +16 / +24
    // slots are immediates of the store
    const unsigned int out_off = (unsigned int)((jl * q.g * g.P + 4 * (lane >> 5)) * CG + c) * 4u;

    __syncthreads ();                                        // the staging waves have committed chunk 0
    int qn = 0;                                              // chunks consumed so far: chunk qn sits in LDS buffer qn & 1
    for (int within = rank; within < tiles_per_xcd; within += wgs_per_xcd) {
        int st, j0;
        if (!tile_at (within, st, j0)) continue;
        // rows carry 30 fraction bits, this lane's channel 2^shift in its period's exponent block; the class sums are combined at
        // weight 256^(4 - s) in units of 2^16: the result is scaled by 2^(-14 - shift) (loaded now, used after the K loop)
        const int out_exp = -14 - q.shifts [((j0 + jl * q.g) / q.eb_periods) * CG + c];
        i32x16 acc [5];
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc [s] [r] = 0;
        // chunks in which some row of this tile has a non-zero most significant digit (the few around the rows' centres: taps
        // fall off as 1 / distance): everywhere else the four products with that digit plane are exactly zero and not issued
        // (and likewise the second digit plane — zero in the window's tails, where the taps are below 2^-15: its four products too)
        unsigned long long top = q.a_masks [(st * q.g + (j0 + q.jr_rot) % q.g) * 32 + (lane & 31)], sec = q.a_masks [q.mask_words + (st * q.g + (j0 + q.jr_rot) % q.g) * 32 + (lane & 31)];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) { top |= __shfl_xor (top, off); sec |= __shfl_xor (sec, off); }
        const unsigned int top_lo = __builtin_amdgcn_readfirstlane ((unsigned int) top), top_hi = __builtin_amdgcn_readfirstlane ((unsigned int)(top >> 32));
        const unsigned int sec_lo = __builtin_amdgcn_readfirstlane ((unsigned int) sec), sec_hi = __builtin_amdgcn_readfirstlane ((unsigned int)(sec >> 32));

        for (int ch = 0; ch < nchunks; ++ch, ++qn) {
            // (one loop body, the LDS buffer chosen by address: two bodies made the compiler keep two copies of the accumulators)
            const unsigned char *Ab = &As_ [0] [0] [0] + (qn & 1) * (int) sizeof (As_ [0]) + aoff;
            const unsigned char *Bb = &Bs_ [0] [0] [0] + (qn & 1) * (int) sizeof (Bs_ [0]) + boff;
            i32x4 av [4], bv [4];
#pragma unroll
            for (int pn = 0; pn < 4; ++pn) {
                av [pn] = *reinterpret_cast<const i32x4 *> (Ab + pn * (32 * I8_PITCH));
                bv [pn] = *reinterpret_cast<const i32x4 *> (Bb + pn * (I8_COLS * I8_PITCH));
            }
            // all eight operand reads are issued together (left to itself the compiler re-used one operand register and paid an
            // LDS round trip per plane), and the buffer is handed back as soon as they have landed
            __builtin_amdgcn_sched_group_barrier (0x100, 8, 0);
            __syncthreads ();
#pragma unroll
            for (int i = 2; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j <= 4) acc [i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (av [i], bv [j], acc [i + j], 0, 0, 0);
            if (((ch < 32 ? sec_lo >> ch : sec_hi >> (ch - 32)) & 1u) != 0u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc [1 + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (av [1], bv [j], acc [1 + j], 0, 0, 0);
            }
            if (((ch < 32 ? top_lo >> ch : top_hi >> (ch - 32)) & 1u) != 0u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc [j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (av [0], bv [j], acc [j], 0, 0, 0);
            }
        }

        // ---- the tile's outputs: C/D layout of 32x32: row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31
        const unsigned int n_tile = a.n_begin + (unsigned int) j0 * g.P + (unsigned int)(st * 32);
        const int rows_valid = min (32, g.P - st * 32);
        const size_t left = (size_t)(a.n_end - n_tile) * CG * 4;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc (a.out + (size_t) n_tile * CG, left > 0xffffff00ull ? 0xffffff00u : (unsigned int) left);
        const unsigned int pass_rows = PASS ? (unsigned int) g.tile_w0 [3 * st + 1] : 0u;
        // (a launch on rows kept across calls starts mid-period: the slots of its first period in front of its first output are not stored.
        // They sit in the first CG columns of the first matrix wave of the launch's first period group: a scalar bound — 0 everywhere else —
        // and a test on the lane's own number, nothing kept live through the tile loop)
        const int lo = a.n_skip != 0 && j0 == 0 && wave == 0 ? a.n_skip - st * 32 : 0;
        // (the lane's half, opaque and per tile: as loop invariants the slot numbers below were computed in front of the tile loop, spilled and read back per tile)
        int half = lane >> 5;
        asm volatile ("" : "+v" (half));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i_const = (r & 3) + 8 * (r >> 2);      // compile-time part of the slot
            // class sums, weights 256^(4 - s), as one exact 64-bit integer, scaled back by the rows' and the channel's exponents (a
            // power of two: exact) and rounded ONCE to float — the same arithmetic in every fixed-point kernel: the same bits
            float y = i8_round (i8_total (acc [0] [r], acc [1] [r], acc [2] [r], acc [3] [r], acc [4] [r]), out_exp);
            const int i = i_const + 4 * half;
            if constexpr (PASS) {
                // nearest-filter mode, the position falls exactly on an input sample: the reference copies it (resampler.c:1141-1142)
                if ((pass_rows >> i) & 1u)
                    y = load_frame (a, INT_MIN, g.canon_ip [st * 32 + i] + g.w_shift + g.canon_fi [st * 32 + i] / a.F + (j0 + jl * q.g) * g.Q, c);
            }
            if (col_live && i < rows_valid && (i >= lo || (lane & 31) >= CG))      // (frames at or past n_end: out of the resource's range, dropped)
                __builtin_amdgcn_raw_buffer_store_b32 (__float_as_uint (y), rs_out, (int)(out_off + (unsigned int)(i_const * CG) * 4u), 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same kernel with the staging done by the LDS-DMA path (buffer_load ... lds, 16 bytes per lane: gfx950), for streams of 4
// channels and more.  fir_i8_stream_kernel above spends its LDS time on the staging waves' writes — per chunk and workgroup 64
// ds_write_b32 wave-instructions (a loaded dword = 4 taps x 1 column, 4 columns per 16-byte load, each to another LDS row), two-way
// conflicting — and holds every chunk in registers on the way.  Here the loaded 16 bytes land in the LDS as they are:
//   B image  [plane][4-tap block kb][column][4 taps]   (a wave's 64 lanes = 2 kb x 32 column quads = 1 KB, lane-linear, which is
//            what the hardware writes: base + lane x 16); a matrix lane picks its column's four dwords 512 bytes apart
//            (two ds_read2st64_b32 per plane, conflict-free);
//   A image  [plane][16-tap half][row][16 taps] — the order the staging pass already leaves in memory.
// No staging registers, no ds_write; three LDS buffers (chunk c being multiplied, c + 1 landed, c + 2 in flight: the depth the
// two register stages gave), the same 60 KB.  The four staging waves only issue (5 DMA instructions per chunk each), count their
// own landings (s_waitcnt vmcnt) and meet the matrix waves at the one barrier per chunk — raw s_barrier: __syncthreads () would
// drain the DMA in flight.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__ ((address_space (3))) void *lds_ptr_t;
#ifndef I8_DMA_BUFS
#define I8_DMA_BUFS 3
#endif

template <int CG, bool PASS>
__global__ __launch_bounds__ (2 * MF_THREADS) __attribute__ ((amdgpu_waves_per_eu (4, 4)))
void fir_i8_dma_kernel (ArtFirArgs a, MfmaGeom g, I8Geom q, int wgs_per_xcd)
{
    static_assert (CG >= 4 && I8_COLS % CG == 0, "16-byte vectors of 4 channels");
    constexpr int THREADS = 2 * MF_THREADS;
    constexpr int PPW = I8_COLS / CG;
    constexpr int NBUF = I8_DMA_BUFS, A_BUF = 4096, B_BUF = 16384;
    __shared__ __attribute__ ((aligned (16))) unsigned char As_ [NBUF * A_BUF];
    __shared__ __attribute__ ((aligned (16))) unsigned char Bs_ [NBUF * B_BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 4;

    const unsigned int stream_blocks = 8u * (unsigned int) wgs_per_xcd;
    if (blockIdx.x >= stream_blocks) {                        // extra workgroups: the history roll (as in fir_mfma_kernel)
        if (a.roll_dst) {
            const int e = (int)(blockIdx.x - stream_blocks) * THREADS + tid;
            if (e < a.H * a.C) {
                const int f = e / a.C, c = e - f * a.C, lin = a.roll_appended + f;
                float v = 0.0f;
                if (lin < a.H) v = a.hist [(size_t) lin * a.C + c];
                else if (a.in && lin - a.H < a.in_frames) v = a.in [(size_t)(lin - a.H) * a.C + c];
                a.roll_dst [e] = v;
            }
        }
        return;
    }
    // (stand-by: as fir_i8_stream_kernel)
    if (*q.flag == q.epoch) {
        static_assert (sizeof (As_) >= 2 * 32 * MF_LD * sizeof (float) && sizeof (Bs_) >= 2 * MF_COLS * MF_LD * sizeof (float), "the f32 tiles live in the digit buffers");
        stand_by_tiles<CG, PASS> (a, g, wgs_per_xcd, *reinterpret_cast<float (*) [2] [32 * MF_LD]> (&As_ [0]), *reinterpret_cast<float (*) [2] [MF_COLS * MF_LD]> (&Bs_ [0]));
        return;
    }

    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;
    const int tiles_per_xcd = q.sg_per_xcd * q.g * q.tiles;
    const int nchunks = q.ktot / I8_KC;

    // tile `within` of this XCD's list -> (slot tile, first period); false if the tile holds no output of the launch
    auto tile_at = [&] (int within, int &st, int &j0) -> bool {
        st = within % q.tiles;
        const int t2 = within / q.tiles, jr = t2 % q.g, sg = xcd * q.sg_per_xcd + t2 / q.g;
        if (sg >= q.super_groups) return false;
        j0 = sg * q.g * PPW + jr;
        return a.n_begin + (unsigned int) j0 * g.P + (unsigned int)(st * 32) < a.n_end;
    };
    int my_tiles = 0;
    { int st, j0; for (int w = rank; w < tiles_per_xcd; w += wgs_per_xcd) my_tiles += tile_at (w, st, j0) ? 1 : 0; }
    if (my_tiles == 0) return;
    const int total = my_tiles * nchunks;

    if (loader) {
        constexpr int VPF = CG / 4;                           // 16-byte vectors per 4-frame block of the stream
        constexpr unsigned int A_STEP = 4096u, B_STEP = (I8_KC / 4) * CG * 4u;
        const int lw = wave - 4;                              // this wave: A plane lw, B 4-tap blocks 2 lw and 2 lw + 1 of every plane
        const int kb = 2 * lw + (lane >> 5), colquad = lane & 31, m = colquad / VPF, cv = colquad - m * VPF;
        const unsigned int boff = (unsigned int)((m * q.gq4 + kb) * CG + cv * 4) * 4u;       // (the tile's first block sits in the resource base)
        const unsigned int a_off = (unsigned int)(lw * 1024 + lane * 16);
        // a column whose period lies d exponent blocks behind the tile's first column stages from that block's own planes: d regions
        // further on, where the same 4-frame block sits d * eb_step blocks earlier
        const unsigned int x_total = 4u * q.eb_plane_bytes;
        const unsigned int eb_hop = x_total - (unsigned int) q.eb_step * (unsigned int)(CG * 4);
        unsigned int bdel = 0u;
        // (the tile table through the scalar cache: a vector load here would sit on the VM counter among the DMAs)
        const __attribute__ ((address_space (4))) int *tile_w0 = (const __attribute__ ((address_space (4))) int *) g.tile_w0;

        int f_within = rank - wgs_per_xcd, f_chunk = 0;
        bool f_live = false;
        const unsigned char *fa_base = nullptr, *fb_base = nullptr;
        unsigned int fa_bytes = 0, fb_bytes = 0;
        auto open_tile = [&] () {                             // next tile of this workgroup's list that holds outputs
            int st = 0, j0 = 0;
            f_live = false;
            for (f_within += wgs_per_xcd; f_within < tiles_per_xcd; f_within += wgs_per_xcd)
                if (tile_at (f_within, st, j0)) { f_live = true; break; }
            if (!f_live) return;
            const int la = max (tile_w0 [3 * st] + g.w_shift + j0 * g.Q + g.head_pad, 0);
            const int eb = j0 / q.eb_periods;
            unsigned int skip = (unsigned int) max ((la >> 2) - q.b0 - eb * q.eb_step, 0) * (unsigned int)(CG * 4);
            if (skip > q.eb_plane_bytes) skip = q.eb_plane_bytes;
            const size_t from = (size_t) eb * x_total + skip;
            fb_base = q.x_planes + from; fb_bytes = (unsigned int) min (q.x_bytes - from, (size_t) 0xfffffff0u);
            bdel = (unsigned int)((j0 + m * q.g) / q.eb_periods - eb) * eb_hop;
            fa_bytes = (unsigned int) nchunks * 4096u;
            fa_base = q.a_planes + (size_t)(st * q.g + (j0 + q.jr_rot) % q.g) * fa_bytes;
        };
        // the next chunk of the workgroup's stream -> LDS buffer `buf`: 5 DMA instructions of this wave, or none past the end
        auto issue = [&] (int buf) -> bool {
            if (!f_live) return false;
            const unsigned int sa = (unsigned int) f_chunk * A_STEP, sb = min ((unsigned int) f_chunk * B_STEP, fb_bytes);
            const __amdgpu_buffer_rsrc_t ra_ = make_rsrc (fa_base + sa, fa_bytes - sa), rb_ = make_rsrc (fb_base + sb, fb_bytes - sb);
            __builtin_amdgcn_raw_ptr_buffer_load_lds (ra_, (lds_ptr_t)(As_ + buf * A_BUF + lw * 1024), 16, (int) a_off, 0, 0, 0);
#pragma unroll
            for (int pn = 0; pn < 4; ++pn)
                __builtin_amdgcn_raw_ptr_buffer_load_lds (rb_, (lds_ptr_t)(Bs_ + buf * B_BUF + pn * 4096 + lw * 1024), 16, (int)(boff + bdel), (int)((unsigned int) pn * q.eb_plane_bytes), 0, 0);
            if (++f_chunk == nchunks) { f_chunk = 0; open_tile (); }
            return true;
        };

        open_tile ();
        bool all = true;
#pragma unroll
        for (int b = 0; b < NBUF - 1; ++b) all = issue (b) && all;
        // (NBUF - 1 chunks in flight: chunk 0 has landed once at most the NBUF - 2 behind it are outstanding, 5 instructions each)
        if (all) { if constexpr (NBUF == 3) asm volatile ("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile ("s_waitcnt vmcnt(10)" ::: "memory"); }
        else asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier ();                        // chunk 0 has landed
        int fill = NBUF - 1;                                 // the buffer chunk c + NBUF - 1 goes to
        for (int c = 0; c < total; ++c) {
            const bool more = issue (fill);
            fill = fill == NBUF - 1 ? 0 : fill + 1;
            // chunk c + 1 has landed once at most the NBUF - 2 chunks behind it are outstanding (at the end of the stream: none)
            if (more) { if constexpr (NBUF == 3) asm volatile ("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile ("s_waitcnt vmcnt(10)" ::: "memory"); }
            else asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier ();
        }
        return;
    }

    // ---- matrix waves ----
    {   // the two workgroups of a CU share each SIMD's matrix pipe: left alone their matrix waves fall into step (both multiply, then
        // both wait for the LDS and the barrier); different issue priorities make them alternate instead
        const unsigned int hw_id = __builtin_amdgcn_s_getreg ((4 - 1) << 11 | 16 << 6 | 4);      // HW_ID.TG_ID: bits 19:16
        if (hw_id & 1u) __builtin_amdgcn_s_setprio (3); else __builtin_amdgcn_s_setprio (0);
    }
    const int col = wave * 32 + (lane & 31);
    const int jl = col / CG, c = col - jl * CG;
    const unsigned char *Ab0 = As_ + (lane & 31) * 16 + (lane >> 5) * 512;
    const unsigned char *Bb0 = Bs_ + (lane >> 5) * 2048 + col * 4;
    // output offset of this lane inside a tile: (period jl * g, slot 4 * (lane >> 5), channel c); the row's own 0..3 / +8 / +16 / +24
    // slots are immediates of the store
    const unsigned int out_off = (unsigned int)((jl * q.g * g.P + 4 * (lane >> 5)) * CG + c) * 4u;

    __builtin_amdgcn_s_barrier ();                            // the staging waves have seen chunk 0 land
    asm volatile ("" ::: "memory");
    int qb = 0;                                              // LDS buffer of the next chunk
    for (int within = rank; within < tiles_per_xcd; within += wgs_per_xcd) {
        int st, j0;
        if (!tile_at (within, st, j0)) continue;
        // rows carry 30 fraction bits, this lane's channel 2^shift in its period's exponent block; the class sums are combined at
        // weight 256^(4 - s) in units of 2^16: the result is scaled by 2^(-14 - shift) (loaded now, used after the K loop)
        const int out_exp = -14 - q.shifts [((j0 + jl * q.g) / q.eb_periods) * CG + c];
        i32x16 acc [5];
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc [s] [r] = 0;
        // chunks in which some row of this tile has a non-zero most significant digit (the few around the rows' centres: taps
        // fall off as 1 / distance): everywhere else the four products with that digit plane are exactly zero and not issued
        // (and likewise the second digit plane — zero in the window's tails, where the taps are below 2^-15: its four products too)
        unsigned long long top = q.a_masks [(st * q.g + (j0 + q.jr_rot) % q.g) * 32 + (lane & 31)], sec = q.a_masks [q.mask_words + (st * q.g + (j0 + q.jr_rot) % q.g) * 32 + (lane & 31)];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) { top |= __shfl_xor (top, off); sec |= __shfl_xor (sec, off); }
        const unsigned int top_lo = __builtin_amdgcn_readfirstlane ((unsigned int) top), top_hi = __builtin_amdgcn_readfirstlane ((unsigned int)(top >> 32));
        const unsigned int sec_lo = __builtin_amdgcn_readfirstlane ((unsigned int) sec), sec_hi = __builtin_amdgcn_readfirstlane ((unsigned int)(sec >> 32));

        for (int ch = 0; ch < nchunks; ++ch) {
            const unsigned char *Ab = Ab0 + qb * A_BUF, *Bb = Bb0 + qb * B_BUF;
            qb = qb == NBUF - 1 ? 0 : qb + 1;
            i32x4 av [4], bv [4];
#pragma unroll
            for (int pn = 0; pn < 4; ++pn) {
                av [pn] = *reinterpret_cast<const i32x4 *> (Ab + pn * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) bv [pn] [i] = *reinterpret_cast<const int *> (Bb + pn * 4096 + i * 512);
            }
            // all twelve operand reads are issued together and, once they have landed, the buffer is handed back (its next writer is
            // the DMA of three chunks on)
            __builtin_amdgcn_sched_group_barrier (0x100, 12, 0);
            asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier ();
            asm volatile ("" ::: "memory");
#pragma unroll
            for (int i = 2; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j <= 4) acc [i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (av [i], bv [j], acc [i + j], 0, 0, 0);
            if (((ch < 32 ? sec_lo >> ch : sec_hi >> (ch - 32)) & 1u) != 0u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc [1 + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (av [1], bv [j], acc [1 + j], 0, 0, 0);
            }
            if (((ch < 32 ? top_lo >> ch : top_hi >> (ch - 32)) & 1u) != 0u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc [j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (av [0], bv [j], acc [j], 0, 0, 0);
            }
        }

        // ---- the tile's outputs: C/D layout of 32x32: row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31
        const unsigned int n_tile = a.n_begin + (unsigned int) j0 * g.P + (unsigned int)(st * 32);
        const int rows_valid = min (32, g.P - st * 32);
        const size_t left = (size_t)(a.n_end - n_tile) * CG * 4;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc (a.out + (size_t) n_tile * CG, left > 0xffffff00ull ? 0xffffff00u : (unsigned int) left);
        const unsigned int pass_rows = PASS ? (unsigned int) g.tile_w0 [3 * st + 1] : 0u;
        // (a launch on rows kept across calls starts mid-period: the slots of its first period in front of its first output are not stored.
        // They sit in the first CG columns of the first matrix wave of the launch's first period group: a scalar bound — 0 everywhere else —
        // and a test on the lane's own number, nothing kept live through the tile loop)
        const int lo = a.n_skip != 0 && j0 == 0 && wave == 0 ? a.n_skip - st * 32 : 0;
        // (the lane's half, opaque and per tile: as loop invariants the slot numbers below were computed in front of the tile loop, spilled and read back per tile)
        int half = lane >> 5;
        asm volatile ("" : "+v" (half));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i_const = (r & 3) + 8 * (r >> 2);      // compile-time part of the slot
            float y = i8_round (i8_total (acc [0] [r], acc [1] [r], acc [2] [r], acc [3] [r], acc [4] [r]), out_exp);
            const int i = i_const + 4 * half;
            if constexpr (PASS) {
                if ((pass_rows >> i) & 1u)
                    y = load_frame (a, INT_MIN, g.canon_ip [st * 32 + i] + g.w_shift + g.canon_fi [st * 32 + i] / a.F + (j0 + jl * q.g) * g.Q, c);
            }
            if (i < rows_valid && (i >= lo || (lane & 31) >= CG))                  // (frames at or past n_end: out of the resource's range, dropped)
                __builtin_amdgcn_raw_buffer_store_b32 (__float_as_uint (y), rs_out, (int)(out_off + (unsigned int)(i_const * CG) * 4u), 0, 0);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Slabs: tiles of 64 slots x 256 columns, one eight-wave workgroup per CU.  What bounds the 32-slot kernel above is the bytes a tile
// pulls through the L2 and the LDS-DMA path per product — 20 KB per chunk for 38 products per workgroup, two workgroups per CU
// each staging tiles of their own: 2.0 GB of L2 requests per headline launch, ~60 % of what the eight L2s deliver, and every wait
// of the kernel is behind that queue.  Here ALL eight waves multiply, each holding two 32-row register tiles on one K origin (64
// slots x 32 columns, 2 x 5 x 16 accumulators, 256 registers: two waves per SIMD, which cover each other's operand reads), and a
// workgroup stages 40 KB per 32 taps for 152 products: half the bytes per product (0.26 KB against 0.53), half the DMA pieces.
// K is walked 64 taps per barrier: two LDS buffers of two 32-tap images each (2 x 80 KB: all of the CU's LDS); behind the barrier
// of chunk c every wave issues its 10 DMA pieces of chunk c + 1 (one piece of the rows and one of each X plane per 32-tap image)
// and then multiplies chunk c.  One barrier per chunk: at it every wave's pieces of the chunk have landed (counted per wave,
// s_waitcnt vmcnt) and every wave has read the chunk before, whose buffer the DMA issued behind the barrier overwrites.
//   LDS image of 32 taps: rows  [plane][16-tap half][row 0..63][16 taps]           (8 KB, the order the staging pass leaves in memory)
//                         X     [plane][column half][4-tap block][column 0..127][4 taps]   (32 KB; a DMA piece = 2 blocks x 128 columns)
// Tiles cut to the launch ("two-tile stream-K").  An XCD's list of L tiles is walked by its W workgroups: whole tiles, strided,
// for all but the last full round; the remaining S = W + (L mod W) tiles are a run of S x chunks K-chunks cut into W equal
// pieces, so a workgroup takes the tail of one tile and the head of the next.  The sums are integers: a part leaves its class
// sums combined to one exact 64-bit integer per output in device memory (16-byte stores written through), and the WAVE that brings
// the tile's per-wave arrival count to the number of parts adds the others' to its own (coherent loads) and writes the outputs —
// no workgroup waits for another, any order of arrival gives the same bits, and they are the bits of the uncut tile.
// The history roll is done by the stream's own workgroups (no extra workgroups: there is one slot per CU).
// ---------------------------------------------------------------------------------------------------
constexpr int SL_COLS = 256, SL_THREADS = 512;
constexpr int SL_A_IMG = 8192, SL_B_IMG = 32768, SL_IMG = SL_A_IMG + SL_B_IMG, SL_BUF = 2 * SL_IMG;
constexpr int SL_WGS = 32;                // workgroups per XCD: one per CU
constexpr int SL_MAX_SK = 64;             // stream-K tiles per XCD (< 2 W), arrival counters per XCD
constexpr size_t SL_PART_BYTES = (size_t) 8 * 16 * 64 * 16;      // one part of one tile: [wave][register pair][lane][2 x int64] = 128 KB

struct I8Slab {
    int wgs_per_xcd;                      // W
    int tail;                             // 0: the run behind the whole rounds is W + (L mod W) tiles (each tile cut in two at most); 1: L mod W tiles, cut into W pieces
    int live [8];                         // tiles of each XCD's list that hold outputs (they are a prefix of the list)
    unsigned char *parts;                 // [xcd][rank][2] parts of SL_PART_BYTES
    unsigned int *arrivals;               // [xcd][SL_MAX_SK][8 waves], zero between launches
#ifdef I8_SLAB_TRACE
    long long *trace;                     // (debug build: per workgroup and wave, cycles spent in each phase of the chunk loop)
#endif
};


template <int CG, bool PASS>
__global__ __launch_bounds__ (SL_THREADS) __attribute__ ((amdgpu_waves_per_eu (2, 2)))
void fir_i8_slab_kernel (ArtFirArgs a, MfmaGeom g, I8Geom q, I8Slab sl)
{
    static_assert (CG >= 4 && SL_COLS % CG == 0 && SL_COLS / CG <= I8_MAX_PPW, "16-byte vectors of 4 channels");
    constexpr int THREADS = SL_THREADS;
    constexpr int PPW = SL_COLS / CG;
    // (ONE __shared__ object: with a second one the compiler waits vmcnt(0) before the first LDS read behind a DMA)
    __shared__ __attribute__ ((aligned (16))) unsigned char smem_ [2 * SL_BUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane (tid >> 6);       // (uniform, and known to be: LDS-DMA destinations are scalars)

    // the history roll: every workgroup its share
    if (a.roll_dst) {
        for (int e = (int) blockIdx.x * THREADS + tid; e < a.H * a.C; e += (int) gridDim.x * THREADS) {
            const int f = e / a.C, c = e - f * a.C, lin = a.roll_appended + f;
            float v = 0.0f;
            if (lin < a.H) v = a.hist [(size_t) lin * a.C + c];
            else if (a.in && lin - a.H < a.in_frames) v = a.in [(size_t)(lin - a.H) * a.C + c];
            a.roll_dst [e] = v;
        }
    }
    const int W = sl.wgs_per_xcd;
    // (stand-by: as fir_i8_stream_kernel — the f32 streaming kernel's tile loop on this kernel's workgroups and LDS)
    if (*q.flag == q.epoch) {
        constexpr int MF_PPW = MF_COLS / CG > MF_MAX_PPW ? MF_MAX_PPW : MF_COLS / CG;
        (void) MF_PPW;
        stand_by_tiles<CG, PASS> (a, g, W, *reinterpret_cast<float (*) [2] [32 * MF_LD]> (&smem_ [0]), *reinterpret_cast<float (*) [2] [MF_COLS * MF_LD]> (&smem_ [SL_BUF]));
        return;
    }

    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;
    const int nsub = q.ktot / I8_KC, nch = (nsub + 1) >> 1;    // 32-tap images per tile; 64-tap chunks (the last may hold one image)
    const int L = sl.live [xcd];
    if (L <= 0) return;
    // ---- this workgroup's work: D whole tiles (rank, rank + W, ...), then chunks [lo, hi) of the run of S tiles behind them
    int D, S;
    {
        const int R = L / W, rem = L - R * W;
        D = rem && !sl.tail ? (R > 0 ? R - 1 : 0) : R;
        S = L - D * W;
    }
    const int Ct = S * nch, Weff = S ? (W < Ct ? W : Ct) : 1;
    const int lo = S && rank < Weff ? (rank * Ct) / Weff : 0, hi = S && rank < Weff ? ((rank + 1) * Ct) / Weff : 0;
    const int t_first = lo / nch, t_last = hi > lo ? (hi - 1) / nch : t_first - 1;
    const int nseg = D + (t_last - t_first + 1);
    if (nseg == 0) return;
    // segment k -> tile of the XCD's list and its chunks [c0, c1)
    auto segment = [&] (int k, int &within, int &c0, int &c1) {
        if (k < D) { within = rank + k * W; c0 = 0; c1 = nch; }
        else {
            const int t = t_first + (k - D);
            within = D * W + t;
            c0 = lo - t * nch; if (c0 < 0) c0 = 0;
            c1 = hi - t * nch; if (c1 > nch) c1 = nch;
        }
    };
    const int g_mask = q.g - 1, g_log2 = q.g >> 1;            // (g = 4 / gcd (Q, 4) is 1, 2 or 4: no divisions by it)
    auto tile_of = [&] (int within, int &st, int &j0) {
        st = within % q.tiles;
        const int t2 = within / q.tiles, jr = t2 & g_mask, sg = xcd * q.sg_per_xcd + (t2 >> g_log2);
        j0 = ((sg * PPW) << g_log2) + jr;
    };

    // ---- this wave's share of the staging, per 32-tap image: piece `wave` of the rows' 8 KB; of every X plane, 4-tap blocks
    // 2 (wave & 3) and + 1 of column half wave >> 2 (64 lanes = 2 blocks x 32 column quads = 1 KB, lane-linear in the LDS)
    constexpr int VPF = CG / 4;                               // 16-byte vectors per 4-frame block of the stream
#ifdef I8_ABL_CONTIG     // (TIMING ONLY: every X piece one contiguous kilobyte — what the DMA path delivers when its 64 lanes read consecutive memory)
    constexpr unsigned int A_STEP = 8192u, B_STEP = 8192u;
    const int kb = 2 * (wave & 3) + (lane >> 5), colquad = (wave >> 2) * 32 + (lane & 31), m = colquad / VPF, cv = colquad - m * VPF;
    const unsigned int boff = (unsigned int)(wave * 1024 + lane * 16) + 0u * (unsigned int)(kb + cv);
#else
    constexpr unsigned int A_STEP = 8192u, B_STEP = (I8_KC / 4) * CG * 4u;
    const int kb = 2 * (wave & 3) + (lane >> 5), colquad = (wave >> 2) * 32 + (lane & 31), m = colquad / VPF, cv = colquad - m * VPF;
    const unsigned int boff = (unsigned int)((m * q.gq4 + kb) * CG + cv * 4) * 4u;           // (the tile's first block sits in the resource base)
#endif
    const unsigned int a_off = (unsigned int)(wave * 1024 + lane * 16);
    // a column whose period lies d exponent blocks behind the tile's first column stages from that block's own planes: d regions
    // further on, where the same 4-frame block sits d * eb_step blocks earlier
    const unsigned int x_total = 4u * q.eb_plane_bytes;
    const unsigned int eb_hop = x_total - (unsigned int) q.eb_step * (unsigned int)(CG * 4);
    const int plane_step = __builtin_amdgcn_readfirstlane ((int) q.eb_plane_bytes);
    // (the tile table through the scalar cache: a vector load here would sit on the VM counter among the DMAs)
    const __attribute__ ((address_space (4))) int *tile_w0 = (const __attribute__ ((address_space (4))) int *) g.tile_w0;

    int f_seg = -1, f_ch = 0, f_c1 = 0;
    unsigned long long f_live [2] = { 0ull, 0ull };           // the fetched tile's images in which the rows' digit plane 0 / 1 is not all zero (either register tile)
    __amdgpu_buffer_rsrc_t f_ra = make_rsrc (nullptr, 0u), f_rb = f_ra;
    unsigned int f_va = 0u, f_vb = 0u;                        // the lanes' offsets of the stream's next chunk
    auto open_segment = [&] () {                              // the next segment of this workgroup's list (there is one)
        int within, st, j0, c0;
        ++f_seg;
        segment (f_seg, within, c0, f_c1);
        tile_of (within, st, j0);
        const int la = max (tile_w0 [3 * (2 * st)] + g.w_shift + j0 * g.Q + g.head_pad, 0);     // (the origin of the pair's first 32-row slot tile)
        const int eb = j0 / q.eb_periods;
        unsigned int skip = (unsigned int) max ((la >> 2) - q.b0 - eb * q.eb_step, 0) * (unsigned int)(CG * 4);
        if (skip > q.eb_plane_bytes) skip = q.eb_plane_bytes;
        const size_t from = (size_t) eb * x_total + skip;
        f_rb = make_rsrc (q.x_planes + from, (unsigned int) min (q.x_bytes - from, (size_t) 0xfffffff0u));
        const unsigned int fa_bytes = (unsigned int) nsub * A_STEP;
        f_ra = make_rsrc (q.a_planes + (size_t)(st * q.g + ((j0 + q.jr_rot) & g_mask)) * fa_bytes, fa_bytes);
        {
            const __attribute__ ((address_space (4))) unsigned long long *tm = (const __attribute__ ((address_space (4))) unsigned long long *) q.tile_masks + (st * q.g + ((j0 + q.jr_rot) & g_mask)) * 2;
            f_live [0] = tm [0] | tm [1];
            f_live [1] = tm [q.tiles * q.g * 2] | tm [q.tiles * q.g * 2 + 1];
        }
        f_ch = c0;
        f_va = a_off + (unsigned int)(2 * c0) * A_STEP;
        f_vb = boff + (unsigned int)((j0 + m * q.g) / q.eb_periods - eb) * eb_hop + (unsigned int)(2 * c0) * B_STEP;
    };
    // The stream's next chunk -> LDS buffer `buf`: 10 DMA pieces of this wave, piece 5 im + 0 = the rows' of 32-tap image im, 5 im + 1 + pn
    // = X plane pn's (an image past the tile's K range: out of the rows' resource, zeros; it is not multiplied).  The pieces are
    // issued one by one BETWEEN the products of the current chunk's first image: a piece costs the issuing wave tens of cycles, and
    // all eight waves issuing theirs together right behind the barrier left the matrix pipes idle for a third of every chunk.
    // next_chunk () moves the stream on (scalar work, behind the barrier) and says where the pieces go; past the end of the stream
    // the resources are empty and the pieces fetch nothing.
    int total = D * nch + (hi - lo), issued = 0;              // chunks of this workgroup's stream; handed to the DMA so far
    unsigned int d_va = 0u, d_vb = 0u;                        // the lanes' offsets of the chunk being issued
    // (this wave's piece of the rows belongs to digit plane wave >> 1: where that plane is all zero in an image — its products are not
    // issued — the piece is not fetched either: a scalar offset past the resource's end, zeros into the LDS, nothing through the L2.
    // The rows' first plane lives in 4 of 33 images, the second in 24: a quarter of the rows' bytes stay where they are)
    int d_skip [2] = { 0, 0 };
    // (likewise the samples' LEAST significant digit plane: it only meets the rows' two upper planes — pairs (0,3) and (1,3) — so an image in
    // which both of those are all zero does not read it, and its four pieces per image are not fetched: 9 of 33 images at 988 taps, 5 % of the staged bytes)
    int d_skip3 [2] = { 0, 0 };
    auto next_chunk = [&] () {
        if (issued < total) {
            if (f_seg < 0 || f_ch == f_c1) open_segment ();
            d_va = f_va; d_vb = f_vb;
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                d_skip [im] = (wave >> 1) < 2 && !((f_live [wave >> 1] >> (2 * f_ch + im)) & 1ull) ? 0x7ffffff0 : 0;
                d_skip3 [im] = !(((f_live [0] | f_live [1]) >> (2 * f_ch + im)) & 1ull) ? 0x7ffffff0 : 0;
            }
            f_va += 2 * A_STEP; f_vb += 2 * B_STEP;
            ++f_ch; ++issued;
        }
        else { f_ra = make_rsrc (nullptr, 0u); f_rb = f_ra; }
    };
    // (TIMING-ONLY ablation builds, tools/micro/slab_ablation.sh — wrong samples, the schedule with one ingredient taken out:
    //  I8_ABL_NO_DMA no staging pieces, I8_ABL_NO_READ no LDS operand reads, I8_ABL_NO_MFMA no products, I8_ABL_NO_XCHG no exchange of parts)
#ifdef I8_ABL_XRES
    int abl_n = 0;
#endif
    auto piece = [&] (int buf, int idx) {
#ifdef I8_ABL_NO_DMA
        return;
#endif
        const int im = idx / 5, pc = idx % 5;
        unsigned char *img = smem_ + buf * SL_BUF + im * SL_IMG;
#ifdef I8_ABL_XRES
        // (TIMING ONLY, round 6: the staging traffic of an X-RESIDENT tile of 64 slots x 128 columns cut in two along K between the wave groups —
        // per image-step of 8 x 17 products TWO images of the rows (16 KB less their zero planes) and 1 / 41 of a 107 KB span of X (one 1 KB piece
        // per wave every third image) instead of 8 + 32 KB.  Wrong samples.)
        // (I8_ABL_XRES == 2: ONE image of the rows per step — a 256-column resident tile, which no LDS holds: the floor of the idea)
        if (pc == 0 || (pc == 1 && I8_ABL_XRES == 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds (f_ra, (lds_ptr_t)(img + pc * SL_A_IMG + wave * 1024), 16, (int)(d_va + (unsigned int)(im + 2 * pc) * A_STEP), d_skip [im], 0, 0);
        else if (pc == 2 && (abl_n++ % 3) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds (f_rb, (lds_ptr_t)(img + 2 * SL_A_IMG + wave * 1024), 16, (int)(d_vb + (unsigned int) im * B_STEP), 0, 0, 0);
        return;
#endif
#ifndef I8_SLAB_A_AUX
#define I8_SLAB_A_AUX 0
#endif
#ifndef I8_SLAB_X_AUX
#define I8_SLAB_X_AUX 0
#endif
        if (pc == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds (f_ra, (lds_ptr_t)(img + wave * 1024), 16, (int)(d_va + (unsigned int) im * A_STEP), d_skip [im], 0, I8_SLAB_A_AUX);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds (f_rb, (lds_ptr_t)(img + SL_A_IMG + (pc - 1) * 8192 + wave * 1024), 16, (int)(d_vb + (unsigned int) im * B_STEP), pc == 4 ? (d_skip3 [im] ? d_skip3 [im] : 3 * plane_step) : (pc - 1) * plane_step, 0, I8_SLAB_X_AUX);
    };

    {   // the two waves of a SIMD share its matrix pipe: left alone they fall into step (both read, then both multiply);
        // different issue priorities make them alternate instead
#ifndef I8_SLAB_PRIO
#define I8_SLAB_PRIO 1
#endif
        if (I8_SLAB_PRIO == 1) { if (wave & 4) __builtin_amdgcn_s_setprio (2); else __builtin_amdgcn_s_setprio (0); }
    }
    const int col = wave * 32 + (lane & 31);
    const int jl = col / CG, c = col - jl * CG;
    // rows' image [plane][16-tap half][row 0..63][16 taps]; X image [plane][column half][4-tap block][column 0..127][4 taps]
    const unsigned char *Ab0 = smem_ + (lane >> 5) * 1024 + (lane & 31) * 16;
    const unsigned char *Bb0 = smem_ + SL_A_IMG + (wave >> 2) * 4096 + (lane >> 5) * 2048 + (col & 127) * 4;

    // ---- the stream.  Chunk s of it sits in LDS buffer s & 1.  Between two barriers a wave multiplies the SECOND image of a chunk,
    // whose operands it read before the barrier, and the FIRST image of the next: at the barrier every wave has read the whole of
    // chunk s — its buffer is free for the pieces of chunk s + 2, issued between the products behind the barrier — and has seen its
    // own pieces of chunk s + 1 land.  (With the barrier between a chunk's reads and the chunk before, all eight waves read
    // operands at the same moment and the matrix pipes idled through every chunk's first read.)
    next_chunk ();
#pragma unroll
    for (int idx = 0; idx < 10; ++idx) piece (0, idx);
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier ();                            // chunk 0 has landed
    asm volatile ("" ::: "memory");
    next_chunk ();
#pragma unroll
    for (int idx = 0; idx < 5; ++idx) piece (1, idx);
    int cur = 0;                                              // LDS buffer of the current chunk
#ifdef I8_SLAB_TRACE
    long long tr_ [16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, tp_ = (long long) __builtin_readcyclecounter ();
    const long long tr_begin = tp_, tr_real = (long long) __builtin_amdgcn_s_memrealtime ();
#define TR(i) do { const long long n_ = (long long) __builtin_readcyclecounter (); tr_ [i] += n_ - tp_; tp_ = n_; } while (0)
#else
#define TR(i) do { } while (0)
#endif

    for (int k = 0; k < nseg; ++k) {
        int within, c0, c1, st, j0;
        segment (k, within, c0, c1);
        tile_of (within, st, j0);
        i32x16 acc [2] [5];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc [h] [s] [r] = 0;
        // images in which some row of a register tile has a non-zero most significant digit (the few around the rows' centres:
        // taps fall off as 1 / distance): everywhere else the four products with that digit plane are exactly zero and not issued
        // (through the scalar cache, like the tile table)
        unsigned long long top [2], sec [2];                  // (sec: the same for the second digit plane — zero in the window's tails)
        {
            const __attribute__ ((address_space (4))) unsigned long long *tm = (const __attribute__ ((address_space (4))) unsigned long long *) q.tile_masks + (st * q.g + ((j0 + q.jr_rot) & g_mask)) * 2;
            top [0] = tm [0]; top [1] = tm [1];
            sec [0] = tm [q.tiles * q.g * 2]; sec [1] = tm [q.tiles * q.g * 2 + 1];
        }
        // this lane's channel's exponent in its period's block: loaded now, taken up behind the wait in front of the tile's first
        // barrier (where this wave drains its memory operations anyway) — nothing in flight is waited for on its account later
        int shift_v = q.shifts [((j0 + jl * q.g) / q.eb_periods) * CG + c];

        i32x4 av [2] [4], bv [4];
#ifdef I8_ABL_NO_READ
#pragma unroll
        for (int pn = 0; pn < 4; ++pn) { av [0] [pn] = i32x4 {lane, pn, 3, 4}; av [1] [pn] = i32x4 {lane, pn, 5, 6}; bv [pn] = i32x4 {pn, lane, 7, 8}; }
#endif
        auto read_image = [&] (int im, int sub) {
#ifdef I8_ABL_NO_READ
            if (im >= 0) { asm volatile ("" : "+v" (av [0] [0]), "+v" (av [1] [0]), "+v" (bv [0])); return; }
#endif
            const unsigned char *Ab = Ab0 + cur * SL_BUF + im * SL_IMG, *Bb = Bb0 + cur * SL_BUF + im * SL_IMG;
            // (only the digit planes this image multiplies: the rows' first plane lives in 4 of 33 images at 988 taps, the second in 24, and the samples'
            // last plane meets nothing else — the CU's LDS is as busy as its matrix pipes (80 KB of DMA writes and 8 waves x 12 KB of operand reads per
            // chunk: ~2,200 clocks at 128 B a clock against ~2,150 of products), and two thirds of the reads are the rows, which every wave reads whole)
            const bool l0 = ((top [0] | top [1]) >> sub) & 1ull, l1 = ((sec [0] | sec [1]) >> sub) & 1ull;
            // (the conditional reads FIRST: the products that use them come last, and the first products' waits can then be counted — behind a
            // branch of unknown length the compiler waits for everything)
            if (l0 | l1) {
                av [0] [1] = *reinterpret_cast<const i32x4 *> (Ab + 2048);
                av [1] [1] = *reinterpret_cast<const i32x4 *> (Ab + 2048 + 512);
#if defined (I8_ABL_XRES) && !defined (I8_ABL_XRES_NOALIGN)
                {
                    int d5 [5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) d5 [i] = *reinterpret_cast<const int *> (Bb + 3 * 8192 + i * 512);
#pragma unroll
                    for (int i = 0; i < 4; ++i) bv [3] [i] = (int) __builtin_amdgcn_alignbyte ((unsigned int) d5 [i + 1], (unsigned int) d5 [i], (unsigned int)(lane & 3));
                }
#else
#pragma unroll
                for (int i = 0; i < 4; ++i) bv [3] [i] = *reinterpret_cast<const int *> (Bb + 3 * 8192 + i * 512);
#endif
                if (l0) {
                    av [0] [0] = *reinterpret_cast<const i32x4 *> (Ab);
                    av [1] [0] = *reinterpret_cast<const i32x4 *> (Ab + 512);
                }
            }
#pragma unroll
            for (int pn = 2; pn < 4; ++pn) {
                av [0] [pn] = *reinterpret_cast<const i32x4 *> (Ab + pn * 2048);
                av [1] [pn] = *reinterpret_cast<const i32x4 *> (Ab + pn * 2048 + 512);
            }
#if defined (I8_ABL_XRES) && !defined (I8_ABL_XRES_NOALIGN)
            // (a column of a resident span starts at any frame: five dwords and four v_alignbyte_b32 per plane instead of four dwords)
#pragma unroll
            for (int pn = 0; pn < 3; ++pn) {
                int d5 [5];
#pragma unroll
                for (int i = 0; i < 5; ++i) d5 [i] = *reinterpret_cast<const int *> (Bb + pn * 8192 + i * 512);
#pragma unroll
                for (int i = 0; i < 4; ++i) bv [pn] [i] = (int) __builtin_amdgcn_alignbyte ((unsigned int) d5 [i + 1], (unsigned int) d5 [i], (unsigned int)(lane & 3));
            }
#else
#pragma unroll
            for (int pn = 0; pn < 3; ++pn)
#pragma unroll
                for (int i = 0; i < 4; ++i) bv [pn] [i] = *reinterpret_cast<const int *> (Bb + pn * 8192 + i * 512);
#endif
        };
        // the products of one image, and five DMA pieces of the chunk being issued (pieces first .. first + 4 -> buffer `to`) spread
        // between them, one behind every fourth product or so: issued in a burst the pieces of eight waves queue up in front of the
        // CU's one address unit (~20 cycles a piece), and a wave stuck behind them multiplies nothing
        auto products = [&] (int sub, int to, int first) {
#ifdef I8_ABL_NO_MFMA
            for (int i = 0; i < 5; ++i) piece (to, first + i);
            return;
#endif
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int n = 0;                                    // products issued so far in this register tile's block
                // the rows' two lower digit planes: always (five products, the pieces between them)
#pragma unroll
                for (int i = 2; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (i + j <= 4) {
                            acc [h] [i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (av [h] [i], bv [j], acc [h] [i + j], 0, 0, 0);
                            ++n;
                            if (h == 0 && n == 2) piece (to, first);
                            if (h == 0 && n == 4) piece (to, first + 1);
                            if (h == 1 && n == 1) piece (to, first + 2);
                            if (h == 1 && n == 3) piece (to, first + 3);
                            if (h == 1 && n == 5) piece (to, first + 4);
                        }
                if (h == 0) {
                    __builtin_amdgcn_sched_group_barrier (0x008, 2, 0); __builtin_amdgcn_sched_group_barrier (0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier (0x008, 2, 0); __builtin_amdgcn_sched_group_barrier (0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier (0x008, 1, 0);
                }
                else {
                    __builtin_amdgcn_sched_group_barrier (0x008, 1, 0); __builtin_amdgcn_sched_group_barrier (0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier (0x008, 2, 0); __builtin_amdgcn_sched_group_barrier (0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier (0x008, 2, 0); __builtin_amdgcn_sched_group_barrier (0x020, 1, 0);
                }
                // the second plane: zero in the window's tails; the first: zero all but around the rows' centres
                if ((sec [h] >> sub) & 1ull) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc [h] [1 + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (av [h] [1], bv [j], acc [h] [1 + j], 0, 0, 0);
                }
                if ((top [h] >> sub) & 1ull) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc [h] [j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (av [h] [0], bv [j], acc [h] [j], 0, 0, 0);
                }
            }
        };

        for (int ch = c0; ch < c1; ++ch) {
            TR (0);                                           // (0: everything between chunks — tile set-up, epilogue, exchange)
            // ---- the chunk's first image
            read_image (0, 2 * ch);
            __builtin_amdgcn_sched_group_barrier (0x100, 16, 0);
            products (2 * ch, cur ^ 1, 5);                    // (with the second five pieces of the chunk announced behind the last barrier)
            TR (4);
            // ---- its second image (a tile's last chunk may have none): operands now, products behind the barrier
            const bool two = 2 * ch + 1 < nsub;
            if (two) read_image (1, 2 * ch + 1);
            asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the chunk has been read: its buffer may be written again)
            asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");         // (this wave's pieces of the next chunk have landed)
            if (ch == c0) asm volatile ("" : "+v" (shift_v));   // (the exponent's load is waited for here)
            TR (1);                                           // (1: reading the second image, waiting for this wave's pieces)
            __builtin_amdgcn_s_barrier ();
            asm volatile ("" ::: "memory");
            TR (2);                                           // (2: the barrier)
            next_chunk ();                                    // the stream's next-but-one chunk -> the buffer just read
            // (moving this scalar upkeep in front of the barrier, where the faster waves wait anyway, was measured: + 1 .. 2 us, profiles/r5_slab_kernel_trims.txt)
            TR (3);                                           // (3: moving the stream on)
            if (two) products (2 * ch + 1, cur, 0);             // (with the first five pieces)
            else {
#pragma unroll
                for (int idx = 0; idx < 5; ++idx) piece (cur, idx);
            }
            TR (5);
            asm volatile ("" ::: "memory");
            cur ^= 1;
        }

        // ---- the tile's outputs.  (In flight: the 10 pieces issued between the last products; they stay in flight.)
        TR (0);
        long long tot [2] [16];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot [h] [r] = i8_total (acc [h] [0] [r], acc [h] [1] [r], acc [h] [2] [r], acc [h] [3] [r], acc [h] [4] [r]);
        TR (9);
#ifdef I8_ABL_NO_XCHG
        if (false) {
#else
        if (c0 != 0 || c1 != nch) {
#endif
            // ---- part of a tile: leave the sums, count the arrival; the last wave to arrive goes on with everybody's
            constexpr int COHERENT = 1 | 16;                 // (aux bits of the raw buffer instructions on gfx940+: sc0, sc1)
            const int t = within - D * W;
            auto owner_of = [&] (int chunk) {                 // the workgroup whose range holds this chunk of the run
                int r = (int)(((long long) chunk * Weff) / Ct);
                while (r + 1 < Weff && ((r + 1) * Ct) / Weff <= chunk) ++r;
                return r;
            };
            const int r_first = owner_of (t * nch), r_last = owner_of ((t + 1) * nch - 1);
            const unsigned int wave_bytes = 16u * 64u * 16u;
            unsigned int *count = sl.arrivals + ((size_t)(xcd * SL_MAX_SK + t) * 8 + wave);
            // (the part a workgroup walks LAST — the head of the run's next tile — usually finds the tile's other part long arrived: it
            // looks first, and if only itself is missing it is the one that finishes and leaves nothing behind: half of the parts'
            // bytes, and a written-through store's round trip, saved)
            unsigned int before = 0u;
            bool look = k == nseg - 1 && k != D;
            if (look) {
                if (lane == 0) before = __hip_atomic_load (count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                before = (unsigned int) __builtin_amdgcn_readfirstlane ((int) before);
                look = before == (unsigned int)(r_last - r_first);
            }
            if (!look) {
                const int which = k == D ? 0 : 1;
                const __amdgpu_buffer_rsrc_t rs_part = make_rsrc (sl.parts + ((size_t)((xcd * W + rank) * 2 + which) * 8 + wave) * wave_bytes, wave_bytes);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    u32x4 v;
                    const unsigned long long x0 = (unsigned long long) tot [0] [r], x1 = (unsigned long long) tot [1] [r];
                    v.x = (unsigned int) x0; v.y = (unsigned int)(x0 >> 32); v.z = (unsigned int) x1; v.w = (unsigned int)(x1 >> 32);
                    __builtin_amdgcn_raw_buffer_store_b128 (v, rs_part, (r * 64 + lane) * 16, 0, COHERENT);
                }
                asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");       // (written through: the count below is only seen behind them)
                if (lane == 0) before = __hip_atomic_fetch_add (count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                before = (unsigned int) __builtin_amdgcn_readfirstlane ((int) before);
            }
            if (before != (unsigned int)(r_last - r_first)) { TR (10); continue; }
            if (lane == 0) __hip_atomic_store (count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (for the next launch)
            for (int rr = r_first; rr <= r_last; ++rr) {
                if (rr == rank) continue;
                const int which = (rr * Ct) / Weff / nch == t ? 0 : 1;      // (that workgroup's first tile of the run, or its last)
                const __amdgpu_buffer_rsrc_t rs_part = make_rsrc (sl.parts + ((size_t)((xcd * W + rr) * 2 + which) * 8 + wave) * wave_bytes, wave_bytes);
                u32x4 v [16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v [r] = __builtin_amdgcn_raw_buffer_load_b128 (rs_part, (r * 64 + lane) * 16, 0, COHERENT);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    tot [0] [r] += (long long)((unsigned long long) v [r].x | (unsigned long long) v [r].y << 32);
                    tot [1] [r] += (long long)((unsigned long long) v [r].z | (unsigned long long) v [r].w << 32);
                }
            }
        }
        TR (10);
        // rows carry 30 fraction bits, this lane's channel 2^shift in its period's exponent block; the total is in units of
        // 2^-30 x 2^-shift x 2^16 (the class weights 256^(4 - s) are relative to the least significant kept class): scaled by
        // 2^(-14 - shift), a power of two, and rounded ONCE to float
        const int out_exp = -14 - shift_v;
        const unsigned int n_tile = a.n_begin + (unsigned int) j0 * g.P + (unsigned int)(st * 64);
        const int rows_valid = min (64, g.P - st * 64);
        const size_t left = (size_t)(a.n_end - n_tile) * CG * 4;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc (a.out + (size_t) n_tile * CG, left > 0xffffff00ull ? 0xffffff00u : (unsigned int) left);
        float y [2] [16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned int pass_rows = PASS && 2 * st + h < g.slot_tiles ? (unsigned int) tile_w0 [3 * (2 * st + h) + 1] : 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y [h] [r] = i8_round (tot [h] [r], out_exp);
                if constexpr (PASS) {
                    const int i = h * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    // nearest-filter mode, the position falls exactly on an input sample: the reference copies it (resampler.c:1141-1142)
                    if ((pass_rows >> (i & 31)) & 1u)
                        y [h] [r] = load_frame (a, INT_MIN, g.canon_ip [st * 64 + i] + g.w_shift + g.canon_fi [st * 64 + i] / a.F + (j0 + jl * q.g) * g.Q, c);
                }
            }
        }
        // ---- the stores.  A lane holds, per block of four registers, four consecutive slots (frames) of ONE channel: stored as they
        // are, an instruction writes 64 scattered dwords — 32-byte runs, eight of them per wave — and the tile's 256 such instructions
        // queue in front of the CU's address unit long after the products have ended (the next tile's first pieces behind them).  The
        // four lanes of a quad hold four consecutive channels: a 4 x 4 transposition inside the quad (two DPP exchanges) leaves every
        // lane one frame's four channels, 16 contiguous bytes — eight 16-byte stores per tile instead of 32 dword stores, whole
        // 128-byte lines per eight lanes of an 8-channel stream.
        TR (11);
        {
            int lane_e = lane;                                // (an opaque copy: see q_off below)
            asm volatile ("" : "+v" (lane_e));
            const int qm = lane_e & 3;                        // this lane's place in its quad = the slot it ends up with
            // (a launch on rows kept across calls starts mid-period: the slots of its first period in front of its first output are not stored —
            // the first CG columns of the first wave of the launch's first period group: a scalar bound and a test on the lane's own number)
            const int lo = a.n_skip != 0 && j0 == 0 && wave == 0 ? a.n_skip - st * 64 : 0;
            // (the lane's period and channel worked out afresh from its number: carried from the top of the kernel, jl * g was the one register the tile loop
            // left no room for — a spill stored in front of the loop and read back here for every tile)
            const int col_e = wave * 32 + (lane_e & 31), jl_e = col_e / CG, c_e = col_e - jl_e * CG;
            const unsigned int q_off = (unsigned int)((((jl_e << g_log2) * g.P) + 4 * (lane_e >> 5) + qm) * CG + (c_e & ~3)) * 4u;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    unsigned int t [4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) t [u] = __float_as_uint (y [h] [4 * rb + u]);
                    // (lane bit 0 <-> register bit 0, then lane bit 1 <-> register bit 1)
#pragma unroll
                    for (int pr = 0; pr < 4; pr += 2) {
                        const unsigned int send = (qm & 1) ? t [pr] : t [pr + 1];
                        const unsigned int recv = (unsigned int) __builtin_amdgcn_mov_dpp ((int) send, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
                        if (qm & 1) t [pr] = recv; else t [pr + 1] = recv;
                    }
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const unsigned int send = (qm & 2) ? t [pr] : t [pr + 2];
                        const unsigned int recv = (unsigned int) __builtin_amdgcn_mov_dpp ((int) send, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
                        if (qm & 2) t [pr] = recv; else t [pr + 2] = recv;
                    }
                    u32x4 v; v.x = t [0]; v.y = t [1]; v.z = t [2]; v.w = t [3];
                    const int i_const = h * 32 + 8 * rb;      // compile-time part of the slot
                    const int i = i_const + 4 * (lane >> 5) + qm;
                    // (a slot past the period goes out of the resource's range, as frames at or past n_end do, and is dropped)
                    const unsigned int off = i < rows_valid && (i >= lo || (lane & 31) >= CG) ? q_off + (unsigned int)(i_const * CG) * 4u : 0xfffffff0u;
                    __builtin_amdgcn_raw_buffer_store_b128 (v, rs_out, (int) off, 0, 0);
                }
        }
        TR (12);
    }
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef I8_SLAB_TRACE
    TR (0);
    tr_ [6] = tp_ - tr_begin; tr_ [7] = (long long) total; tr_ [15] = (long long) __builtin_amdgcn_s_memrealtime () - tr_real;
    if (lane == 0) for (int i = 0; i < 16; ++i) sl.trace [((size_t) blockIdx.x * 8 + wave) * 16 + i] = tr_ [i];
#endif
#undef TR
}


} // namespace

// Slabs (fir_i8_slab_kernel: tiles of 64 slots x 256 columns, one workgroup per CU, tiles cut to the launch) for streams whose
// 4-frame blocks are whole 16-byte vectors.  ARTAMD_I8_SLAB=0 (or ARTAMD_I8_DMA=0) switches them off — the 32-slot kernels then
// run every launch, and periods are taken so as to fill THEIR tiles (A/B runs; the results are the same bits either way).
bool artfir_i8_slab_enabled ()
{
    static const bool on = [] { const char *e = getenv ("ARTAMD_I8_SLAB"); const char *d = getenv ("ARTAMD_I8_DMA"); return !(e && *e == '0') && !(d && *d == '0'); } ();
    return on;
}
// slabs per XCD from which a launch is given to the slab kernel (below — measured, tools/micro/slab_sizes.sh — more, smaller tiles
// fill the chip better and walk their K range sooner)
static int i8_slab_min_tiles ()
{
    // (16 until round 5: re-measured with the rows kept across calls, tools/micro/fixed_crossover_r5.sh + ARTAMD_I8_SLAB_MIN — at 20 slabs per XCD the 32-slot kernels are
    // ahead: 8 ch x 988 taps at 196,608 frames 42.0 against 48.0 us a call, 16 ch at 98,304 40.0 against 56.9, 4 ch at 393,216 46.1 against 49.4; at 30 they are level)
    static const int v = [] { const char *e = getenv ("ARTAMD_I8_SLAB_MIN"); return e && *e ? atoi (e) : 24; } ();
    return v;
}

// The planes buffer of a launch: [header: flag (art_internal.h)][row masks][exponents][A digit planes][X digit planes per exponent block];
// returns its size, 0 if the launch is not for this path
static size_t i8_layout (const ArtFirArgs *a, const MfmaGeom &g, int cgt, I8Geom &q, char *base, unsigned int outputs = 0)
{
    if (!cgt || g.tile_rows != 32 || (g.ktot % I8_KC)) return 0;
    // (outputs != 0: sizing a call's buffer before its launches are cut — any launch of the call has at most this many periods)
    const unsigned int total = outputs ? outputs + (unsigned int) g.P : a->n_end - a->n_begin, periods = (total + g.P - 1) / g.P;
    const int gg = (g.Q % 4 == 0) ? 1 : (g.Q % 2 == 0) ? 2 : 4;
    q.tr = 32; q.cols = I8_COLS; q.rows_cached = 0; q.jr_rot = 0; q.rows_table = 0; q.tb_base = 0.0; q.tb_lin = q.tb_w = 0; q.tb_n0 = 0u;
    if (artfir_i8_slab_enabled () && cgt >= 4) {
        // slabs where the launch has enough of them (decided from this context's own columns: every fixed-point kernel leaves the
        // same bits, so a shard need not decide as its stream would)
        const int ppw64 = SL_COLS / cgt;
        const int sgs = (int)((periods + (unsigned int)(gg * ppw64) - 1) / (unsigned int)(gg * ppw64));
        const int per_xcd = ((sgs + 7) / 8) * gg * ((g.P + 63) / 64);
        if (per_xcd >= i8_slab_min_tiles ()) { q.tr = 64; q.cols = SL_COLS; }
    }
    for (;;) {  // (as matrix_geometry's ktot, for tiles of tr rows; + 3: a tile's K columns start on a 4-frame block, up to 3 frames early)
        const int shift_max = (int)((q.tr - 1.0) * g.Q / g.P) + 2;
        q.ktot = ((a->T + shift_max + 3 + I8_KC - 1) / I8_KC) * I8_KC;
        // (the slab kernel wants a few chunks per tile; and one mask bit per 32-tap image)
        if (q.tr == 64 && (q.ktot / I8_KC < 4 || q.ktot / I8_KC > 64)) { q.tr = 32; q.cols = I8_COLS; continue; }
        break;
    }
    q.tiles = (g.P + q.tr - 1) / q.tr;
    const int ppw = q.cols / cgt > I8_MAX_PPW ? I8_MAX_PPW : q.cols / cgt;
    q.g = (g.Q % 4 == 0) ? 1 : (g.Q % 2 == 0) ? 2 : 4;
    q.gq4 = q.g * g.Q / 4;
    q.super_groups = (int)((periods + (unsigned int)(q.g * ppw) - 1) / (unsigned int)(q.g * ppw));
    q.sg_per_xcd = (q.super_groups + 7) / 8;
    if (q.ktot / I8_KC > 64) return 0;                                  // (one mask bit per chunk)
    // Exponent blocks: eb_periods periods each — a multiple of g (block starts stay on 4-frame blocks) chosen from the ratio alone
    // so that a block spans ~9,400 input frames (one 8-channel tile's periods at 44.1k -> 48k), whatever the channel count.  A
    // block's planes hold every frame its periods read: their input span, + the span of one period's slot tiles (< Q + 2 frames),
    // + the K columns, + alignment.  One staging workgroup holds a block (x channel group) in registers: blocks x channels <=
    // I8_STAGE_K x 1024 units.
    {
        const int unit = 16 * q.g * g.Q;
        const int k = (9408 + unit / 2) / unit;
        q.eb_periods = 16 * q.g * (k > 1 ? k : 1);
    }
    q.eb_step = q.eb_periods * g.Q / 4;                                 // (g * Q is a multiple of 4)
    const int over_blocks = (g.Q + 2 + q.ktot + 3 + 3) / 4 + 2;
    q.eb_blocks = q.eb_step + over_blocks;
    q.ebs = (int)((periods + (unsigned int) q.eb_periods - 1) / (unsigned int) q.eb_periods);
    // Staging workgroups: a block's region is cut into slices of consecutive 4-frame blocks, 32-byte runs per frame (8 channels
    // per workgroup where the stream has them), I8_STAGE_K x 256 units each.  (Measured on the way here: one 1024-thread
    // workgroup per exponent block holding its region in registers — one pass, no peak kernel — took 37 us for 67 MB on the
    // headline call: a CU draws from memory at ~18 GB/s however many loads it has in flight, so 111 busy CUs are not enough; teams
    // of such workgroups agreeing on the peak through device-wide atomics and a bounded wait, all CUs busy: 31 us — every
    // workgroup reads, then every workgroup writes.  Two massively parallel launches do better.)
    q.cgrp = a->C < 8 ? a->C : 8;
    q.slice_blocks = I8_STAGE_K * I8_STAGE_THREADS / q.cgrp;
    q.slices = (q.eb_blocks + q.slice_blocks - 1) / q.slice_blocks;
    q.eb_plane_bytes = (unsigned int) q.eb_blocks * (unsigned int) a->C * 4u;
    q.x_bytes = (size_t) q.ebs * 4 * q.eb_plane_bytes;
    q.b0 = 0;
    const size_t a_bytes = (size_t) q.tiles * q.g * (q.ktot / I8_KC) * (size_t)(q.tr * 128);
    const size_t masks = 2 * ((size_t) q.tiles * q.g * q.tr * 8 + (size_t) q.tiles * q.g * (q.tr / 32) * 8), shifts = (size_t) q.ebs * a->C * 4;
    const size_t teams = (size_t) q.ebs * q.slices * a->C * 4;
    const size_t head = (ART_I8_HEAD_BYTES + masks + teams + shifts + 255) & ~(size_t) 255;
    q.flag = (int *) base; q.epoch = 0;
    q.a_masks = (unsigned long long *)(base + ART_I8_HEAD_BYTES);
    q.mask_words = q.tiles * q.g * q.tr;
    q.tile_masks = q.a_masks + 2 * (size_t) q.mask_words;
    q.peaks = (unsigned int *)(base + ART_I8_HEAD_BYTES + masks);
    q.shifts = (int *)(base + ART_I8_HEAD_BYTES + masks + teams);
    q.a_planes = (unsigned char *) base + head;
    q.x_planes_w = (unsigned int *)(base + head + a_bytes); q.x_planes = (const unsigned char *) q.x_planes_w;
    // (slabs: behind the planes, the parts of the tiles that are cut between workgroups — two per workgroup)
    q.parts = q.tr == 64 ? (unsigned char *) base + ((head + a_bytes + q.x_bytes + 255) & ~(size_t) 255) : nullptr;
    return q.tr == 64 ? ((head + a_bytes + q.x_bytes + 255) & ~(size_t) 255) + (size_t) 8 * SL_WGS * 2 * SL_PART_BYTES : head + a_bytes + q.x_bytes;
}

size_t artfir_i8_bytes (const ArtFirArgs *a, const MfmaGeom &g, int cgt, unsigned int outputs)
{
    I8Geom q;
    return i8_layout (a, g, cgt, q, nullptr, outputs ? outputs : 1u);
}

// ---------------------------------------------------------------------------------------------------
// The rows across calls.  A launch's filter rows — digit planes, masks, the stand-by's f32 tables — depend on the stream's ratio and on
// where in its period the launch starts, not on its samples: until round 5 every launch rebuilt them (640 workgroups of a latency
// chain — two position evaluations, the bank's rows, the blend: 11.5 us in front of every call's main kernel, the long pole of the peak
// pass at every call size).  A context now keeps them (ArtFirArgs.rows on the device, an ArtRowsCache on the host):
//   * the CANONICAL PERIOD: the slot positions (window index, filter index, phase) of the first period of the stream's first fixed-point
//     launch, evaluated on the host with the reference's position arithmetic (host_locate = locate ()).  Every later launch looks its
//     first output up among them: it is slot s of that period, a whole number of frames w further on.  Rows are only ever built FOR THE
//     CANONICAL PERIOD, from this table (uploaded; the row workgroups read it instead of evaluating positions): they are a function of the
//     table alone, whichever launch builds them, whatever its tile height — so the kernel forms still leave the same bits;
//   * a launch runs as if it started s outputs earlier, on the canonical period's first slot: tiles anchored there, the s slots in front of
//     its first output computed and not stored (ArtFirArgs.n_skip), output indices carried one period higher so that the virtual start is not
//     negative, and w added to the table's linear indices (MfmaGeom.w_shift);
//   * the fixed-point tiles start their K columns on 4-frame blocks, the offset inside the block absorbed by the rows (one variant per
//     period residue): a SET of rows built at shift w_b serves the launches whose w - w_b is a multiple of 4 / g frames — residue jr of the
//     launch then stages the variant (jr + rot) mod g, rot Q = w - w_b (mod 4) (I8Geom.jr_rot) — and there are at most 4 / g sets;
//   * interpolating streams accept a first output whose phase is within 1e-6 filter steps of its canonical slot's (the blend moves by
//     < 4e-9 relative; the streaming kernels already share a row between periods 2e-6 apart); nearest-filter streams must round to the
//     canonical filter index in every slot; a stream that left the lattice (advance, reset to another phase, another ratio) starts a new
//     canonical period.  ARTAMD_ROWS_CACHE=0: off — every launch builds its rows from its own positions, as before.
// ---------------------------------------------------------------------------------------------------
namespace {

struct RowsSetPtrs {
    unsigned long long *a_masks, *tile_masks; unsigned char *a_planes;
    float *eff; double *canon_frac; int *canon_ip, *canon_fi, *tile_w0;
};
static size_t rows_set_layout (const MfmaGeom &g, const I8Geom &q, char *base, RowsSetPtrs *out)
{
    size_t off = 0;
    auto take = [&] (size_t bytes) { const size_t at = off; off = (off + bytes + 255) & ~(size_t) 255; return at; };
    const size_t rows32 = (size_t) g.slot_tiles * 32;
    const size_t o_masks = take (2 * (size_t) q.tiles * q.g * q.tr * 8), o_tm = take (2 * (size_t) q.tiles * q.g * (q.tr / 32) * 8);
    const size_t o_planes = take ((size_t) q.tiles * q.g * (q.ktot / I8_KC) * (size_t)(q.tr * 128));
    const size_t o_eff = take (rows32 * g.ktot * sizeof (float)), o_frac = take (rows32 * sizeof (double));
    const size_t o_ip = take (rows32 * sizeof (int)), o_fi = take (rows32 * sizeof (int)), o_w0 = take ((size_t) 3 * g.slot_tiles * sizeof (int));
    if (out && base) {
        out->a_masks = (unsigned long long *)(base + o_masks); out->tile_masks = (unsigned long long *)(base + o_tm); out->a_planes = (unsigned char *)(base + o_planes);
        out->eff = (float *)(base + o_eff); out->canon_frac = (double *)(base + o_frac); out->canon_ip = (int *)(base + o_ip); out->canon_fi = (int *)(base + o_fi);
        out->tile_w0 = (int *)(base + o_w0);
    }
    return off;
}

} // namespace

extern "C" {
size_t arthip_fir_rows_cache_bytes (void) { return sizeof (ArtRowsCache); }
void arthip_fir_rows_cache_reset (void *cache)               // (the device buffer was replaced: no set in it is valid; the canonical period stays)
{
    ArtRowsCache *rc = (ArtRowsCache *) cache;
    if (rc) { for (int k = 0; k < 4; ++k) rc->valid [k] = 0; rc->f_valid = 0; }
}
void arthip_fir_rows_cache_free (void *cache)
{
    ArtRowsCache *rc = (ArtRowsCache *) cache;
    if (rc) {
        free (rc->c_ph); free (rc->c_ip); free (rc->c_fi); rc->c_ph = nullptr; rc->c_ip = rc->c_fi = nullptr; rc->cap = 0; rc->canon_valid = 0;
    }
}
}

// device bytes the rows of a call of this shape want across calls (all sets; 0: no cache for it)
size_t artfir_i8_rows_bytes (const ArtFirArgs *a, const MfmaGeom &g, int cgt, unsigned int outputs)
{
    I8Geom q;
    if (!i8_layout (a, g, cgt, q, nullptr, outputs ? outputs : 1u)) return 0;
    // (4 / g sets: one per place of the windows inside their 4-frame blocks that the period residues do not cover)
    return (size_t)(4 / q.g) * (rows_set_layout (g, q, nullptr, nullptr) + 4096);      // (behind the f32 kernel's set: artfir_rows_bytes adds it)
}

int artfir_i8_launch (const ArtFirArgs *a_in, const ArtSegTable *segs, const MfmaGeom &g_in, int cgt, unsigned int roll_blocks, hipStream_t st)
{
    static std::atomic<int> launches {0};
    I8Geom q;
    if (!a_in->planes) return 0;
    ArtFirArgs a_v = *a_in; const ArtFirArgs *a = &a_v;
    MfmaGeom g = g_in;

    // ---- the launch's place in the canonical period
    ArtRowsCache *rc = artfir_rows_cache_enabled () && a_in->rows ? (ArtRowsCache *) a_in->rows_cache : nullptr;
    HostPos pos0;
    int slot0 = 0, w = 0;
    if (!artfir_rows_canonical (a_in, segs, g.P, g.Q, rc, &pos0, &slot0, &w)) rc = nullptr;
    // (the virtual start's window must not begin in front of the zero frames the planes hold before linear frame 0: head_pad covers a whole period's input —
    // round 5's fixed 64 frames sent a third of the launches of a 96k -> 44.1k stream back to rows of their own)
    if (rc && rc->c_ip [0] + w - a_in->T / 2 + 1 + g.head_pad < 0) rc = nullptr;
    if (rc) {
        a_v.n_begin = a_in->n_begin + (unsigned int)(g.P - slot0); a_v.n_end = a_in->n_end + (unsigned int) g.P;
        a_v.out = a_in->out - (size_t) g.P * a_in->C; a_v.n_skip = slot0;
    }
    size_t need = i8_layout (a, g, cgt, q, (char *) a_in->planes);
    if (rc && (!need || need > a_in->planes_bytes)) {         // (the partial period in front does not fit: as before)
        rc = nullptr; a_v = *a_in;
        need = i8_layout (a, g, cgt, q, (char *) a_in->planes);
    }
    if (!need || need > a_in->planes_bytes) return 0;

    // ---- the set of rows that serves it, or that it builds
    bool build_from_table = false;
    int set = -1;
    if (rc) {
        const size_t set_bytes = rows_set_layout (g, q, nullptr, nullptr);
        const int nsets = 4 / q.g;
        const bool same = rc->lowpass == a_in->lowpass && rc->tr == q.tr && rc->ktot == q.ktot && rc->tiles == q.tiles && rc->g == q.g && rc->slot_tiles == g.slot_tiles &&
                          rc->ktot32 == g.ktot && rc->set_bytes == set_bytes && rc->nsets == nsets;
        if (artfir_f32_set_bytes (g) + (size_t) nsets * (set_bytes + 4096) > a_in->rows_bytes) {          // (no room: this launch alone, in its own buffers, from its own positions)
            rc = nullptr; a_v = *a_in;
            need = i8_layout (a, g, cgt, q, (char *) a_in->planes);
            if (!need || need > a_in->planes_bytes) return 0;
        }
        else {
            if (!same) {
                for (int k = 0; k < 4; ++k) rc->valid [k] = 0;
                rc->lowpass = a_in->lowpass; rc->tr = q.tr; rc->ktot = q.ktot; rc->tiles = q.tiles; rc->g = q.g; rc->slot_tiles = g.slot_tiles; rc->ktot32 = g.ktot;
                rc->set_bytes = set_bytes; rc->nsets = nsets; rc->victim = 0;
            }
            const int step = 4 / q.g;
            for (int k = 0; k < nsets && set < 0; ++k) {
                if (!rc->valid [k] || ((w - rc->w_build [k]) % step) != 0) continue;
                const int dw = w - rc->w_build [k];
                for (int r = 0; r < q.g; ++r) if ((((long long) r * g.Q - dw) & 3) == 0) { set = k; g.w_shift = dw; q.jr_rot = r; q.rows_cached = 1; break; }
            }
            if (set < 0) {                                    // build: an empty set, else the next victim in turn
                for (int k = 0; k < nsets && set < 0; ++k) if (!rc->valid [k]) set = k;
                if (set < 0) { set = rc->victim; rc->victim = (rc->victim + 1) % nsets; }
                rc->valid [set] = 1; rc->w_build [set] = w;
                g.w_shift = 0; q.jr_rot = 0; q.rows_cached = 0; build_from_table = true;
            }
            RowsSetPtrs sp;
            rows_set_layout (g, q, (char *) a_in->rows + artfir_f32_set_bytes (g) + (size_t) set * (set_bytes + 4096), &sp);      // (behind the f32 kernel's own set)
            q.a_masks = sp.a_masks; q.tile_masks = sp.tile_masks; q.a_planes = sp.a_planes;
            g.eff = sp.eff; g.canon_frac = sp.canon_frac; g.canon_ip = sp.canon_ip; g.canon_fi = sp.canon_fi; g.tile_w0 = sp.tile_w0;
        }
    }
    q.rows_table = build_from_table ? 1 : 0;
    if (build_from_table) { q.tb_base = rc->c_base; q.tb_lin = rc->c_lin; q.tb_n0 = rc->c_n0; q.tb_w = w; }
    {   static const bool trace = [] { const char *e = getenv ("ARTAMD_ROWS_TRACE"); return e && *e == '1'; } ();
        if (trace) fprintf (stderr, "rows: launch n %u..%u  cache %s  slot0 %d  w %d  set %d  %s  w_shift %d rot %d  tr %d  ph0 %.12f ip0 %d\n", a_in->n_begin, a_in->n_end,
                            rc ? "on" : "off", slot0, w, set, rc ? (build_from_table ? "BUILD" : "hit") : "-", g.w_shift, q.jr_rot, q.tr, pos0.ph, pos0.ip);
    }
    if (a_in->rows_masks_out) *a_in->rows_masks_out = (void *) q.a_masks;

    int ep = ++launches;
    if (ep <= 0) { launches = 1; ep = 1; }                             // (the flag word is zero when the buffer is allocated)
    q.epoch = ep;
    static const bool dma = [] { const char *e = getenv ("ARTAMD_I8_DMA"); return !(e && *e == '0'); } ();
    if (a->fixed_out) { a->fixed_out [0] = ep; a->fixed_out [1] = q.tiles * q.g * q.tr; a->fixed_out [2] = q.ktot / I8_KC; a->fixed_out [3] = q.tr == 64 ? 3 : (dma && cgt >= 4) ? 2 : 1; }

    {   // block of the first tile's window start: the launch's first output's position (the reference's arithmetic, host_locate), or the
        // canonical period's first slot carried to this launch
        const int ip = rc ? rc->c_ip [0] + w : pos0.ip;
        const int la = ip - a->T / 2 + 1 + g.head_pad;
        q.b0 = (la > 0 ? la : 0) >> 2;
    }
    const unsigned int x_wgs = (unsigned int)(q.ebs * (a->C / q.cgrp) * q.slices);
    const dim3 pgrid (x_wgs + (q.rows_cached ? 0u : (unsigned int)(q.tiles * q.g * q.tr))),
               xgrid (x_wgs + (q.rows_cached ? 0u : (unsigned int)((q.tiles * q.g * (q.tr / 32) + I8_STAGE_THREADS - 1) / I8_STAGE_THREADS)));
    if (a->interpolate) {
        hipLaunchKernelGGL ((i8_stage_kernel<true, true>), pgrid, dim3 (I8_STAGE_THREADS), 0, st, *a, *segs, g, q);
        hipLaunchKernelGGL ((i8_stage_kernel<true, false>), xgrid, dim3 (I8_STAGE_THREADS), 0, st, *a, *segs, g, q);
    }
    else {
        hipLaunchKernelGGL ((i8_stage_kernel<false, true>), pgrid, dim3 (I8_STAGE_THREADS), 0, st, *a, *segs, g, q);
        hipLaunchKernelGGL ((i8_stage_kernel<false, false>), xgrid, dim3 (I8_STAGE_THREADS), 0, st, *a, *segs, g, q);
    }
    if (a->ev_start) arthip_event_record (a->ev_start, (void *) st);

    const int tiles_per_xcd = q.sg_per_xcd * q.g * q.tiles;
    // As many workgroups as the XCD holds (32 CUs x 2: 60 / 72 KB of LDS, 128 / 256 registers each) less two slots for the history-roll
    // workgroups of the same grid, each striding the XCD's tile list; the last, partly filled round then runs with one
    // workgroup per CU and its tiles finish sooner.  (Equal shares — 56 workgroups x 5 tiles for the headline's 280 — kept 8
    // slots idle for the whole launch: 0.1078 ms, 64 workgroups 0.1039, 62 0.1029; 4 and 32 channels, 256k..1M frames,
    // 96k -> 44.1k: 3..9 % the same way.  ARTAMD_I8_WGS overrides, for experiments.)
    const int resident = 62;
    int wgs_per_xcd = tiles_per_xcd < resident ? tiles_per_xcd : resident;
    { static const int k_env = [] { const char *e = getenv ("ARTAMD_I8_WGS"); return e && *e ? atoi (e) : 0; } (); if (k_env > 0 && k_env < tiles_per_xcd) wgs_per_xcd = k_env; }
    // (big launches of the nearest-filter mode: the plain instantiation and the pass-through pass behind it, artfir_pass_fixup_wanted)
    const bool fixup = artfir_pass_fixup_wanted (a);
    const bool pass = !a->interpolate && !a->lowpass && !fixup;
    if (q.tr == 64) {
        // slabs: one eight-wave workgroup per CU (all of its LDS), which also rolls the history and carries the stand-by
        I8Slab sl;
        sl.wgs_per_xcd = SL_WGS;
        { static const int tail_env = [] { const char *e = getenv ("ARTAMD_I8_SLAB_TAIL"); return e && *e ? atoi (e) : 0; } (); sl.tail = tail_env; }
        sl.parts = q.parts;
        sl.arrivals = (unsigned int *)((char *) a->planes + ART_I8_FLAG_BYTES);
        {   // tiles of each XCD's list that hold outputs: a prefix of the list (a tile holds outputs iff its first column's period does)
            const int ppw = q.cols / cgt > I8_MAX_PPW ? I8_MAX_PPW : q.cols / cgt;
            const unsigned int n_out = a->n_end - a->n_begin;
            for (int x = 0; x < 8; ++x) {
                int live = 0; bool prefix = true;
                for (int w = 0; w < tiles_per_xcd; ++w) {
                    const int stt = w % q.tiles, t2 = w / q.tiles, jr = t2 % q.g, sg = x * q.sg_per_xcd + t2 / q.g;
                    const bool ok = sg < q.super_groups && (unsigned long long)(sg * q.g * ppw + jr) * (unsigned int) g.P + (unsigned int)(stt * 64) < n_out;
                    if (ok) { if (live != w) prefix = false; ++live; }
                }
                if (!prefix) return 0;                        // (cannot happen: see the kernel's comment; the 32-slot path would be taken by the caller)
                sl.live [x] = live;
            }
        }
        const dim3 wgrid ((unsigned int)(8 * SL_WGS));
#ifdef I8_SLAB_TRACE
        static long long *d_trace = nullptr;
        if (!d_trace) (void) hipMalloc (&d_trace, (size_t) 8 * SL_WGS * 8 * 16 * sizeof (long long));
        sl.trace = d_trace;
#endif
#define I8_SLAB(CGT) do { if (pass) hipLaunchKernelGGL ((fir_i8_slab_kernel<CGT, true>), wgrid, dim3 (SL_THREADS), 0, st, *a, g, q, sl); \
                          else hipLaunchKernelGGL ((fir_i8_slab_kernel<CGT, false>), wgrid, dim3 (SL_THREADS), 0, st, *a, g, q, sl); } while (0)
        switch (cgt) { case 32: I8_SLAB (32); break; case 16: I8_SLAB (16); break; case 8: I8_SLAB (8); break; default: I8_SLAB (4); }
#undef I8_SLAB
        if (a->ev_stop) arthip_event_record (a->ev_stop, (void *) st);
        if (fixup && artfir_pass_fixup (a, g, st)) return -1;
#ifdef I8_SLAB_TRACE
        {
            static int n_launch = 0;
            if (++n_launch == 40) {
                static long long h [8 * SL_WGS * 8 * 16];
                (void) hipStreamSynchronize (st);
                (void) hipMemcpy (h, d_trace, sizeof (h), hipMemcpyDeviceToHost);
                const char *names [16] = { "tile set-up", "wait own pieces", "barrier", "stream on", "image 0 (+ pieces)", "image 1", "whole kernel", "chunks",
                                           "drain before outputs", "totals", "exchange", "round", "stores", "-", "-", "whole kernel (100 MHz)" };
                for (int wv = 0; wv < 8; wv += 4) {
                    fprintf (stderr, "slab trace, wave %d, mean over %d workgroups (cycles; per chunk in brackets):\n", wv, 8 * SL_WGS);
                    double chunks = 0; for (int b = 0; b < 8 * SL_WGS; ++b) chunks += (double) h [((size_t) b * 8 + wv) * 16 + 7];
                    for (int i = 0; i < 16; ++i) {
                        double sum = 0, mx = 0; for (int b = 0; b < 8 * SL_WGS; ++b) { const double v = (double) h [((size_t) b * 8 + wv) * 16 + i]; sum += v; if (v > mx) mx = v; }
                        fprintf (stderr, "   %-20s %10.0f  max %10.0f  [%8.1f]\n", names [i], sum / (8 * SL_WGS), mx, sum / chunks);
                    }
                }
            }
        }
#endif
        return 1;
    }
    const dim3 sgrid ((unsigned int)(8 * wgs_per_xcd) + roll_blocks);
#define I8_GO(CGT) do { if (pass) hipLaunchKernelGGL ((fir_i8_stream_kernel<CGT, true>), sgrid, dim3 (2 * MF_THREADS), 0, st, *a, g, q, wgs_per_xcd); \
                        else hipLaunchKernelGGL ((fir_i8_stream_kernel<CGT, false>), sgrid, dim3 (2 * MF_THREADS), 0, st, *a, g, q, wgs_per_xcd); } while (0)
#define I8_DMA(CGT) do { if (pass) hipLaunchKernelGGL ((fir_i8_dma_kernel<CGT, true>), sgrid, dim3 (2 * MF_THREADS), 0, st, *a, g, q, wgs_per_xcd); \
                         else hipLaunchKernelGGL ((fir_i8_dma_kernel<CGT, false>), sgrid, dim3 (2 * MF_THREADS), 0, st, *a, g, q, wgs_per_xcd); } while (0)
    // (ARTAMD_I8_DMA=0: the register-staged kernel for every channel count — comparisons; the results are the same bits)
    if (dma && cgt >= 4) switch (cgt) { case 32: I8_DMA (32); break; case 16: I8_DMA (16); break; case 8: I8_DMA (8); break; default: I8_DMA (4); }
    else switch (cgt) { case 32: I8_GO (32); break; case 16: I8_GO (16); break; case 8: I8_GO (8); break; case 4: I8_GO (4); break; case 2: I8_GO (2); break; default: I8_GO (1); }
#undef I8_DMA
#undef I8_GO
    if (a->ev_stop) arthip_event_record (a->ev_stop, (void *) st);
    if (fixup && artfir_pass_fixup (a, g, st)) return -1;
    return 1;
}

#endif  // !ART_WIDE
