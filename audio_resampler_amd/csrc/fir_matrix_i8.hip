// fir_matrix_i8.hip — the fixed-point form of the matrix-core path (4-byte samples, gfx950): regular launches of the rational-ratio
// GEMM (see fir_matrix.hip) evaluated EXACTLY on the integer matrix cores instead of in chained f32.
//
// Effective filter rows (lerp folded in, fp64) are rounded once to 32-bit fixed point with 30 fraction bits; input samples are
// rounded once to 32-bit BLOCK floating point: every channel of a launch has its own binary exponent, taken from the channel's
// peak |x| over the launch's history ++ input (i8_peak_kernel) so that the peak lands in [2^29, 2^31 - 2^23) — exact for every
// float sample within 2^-6 of its channel's peak, (peak) x 2^-31 absolute below that: the error model is RELATIVE to the
// channel's level in the launch, as float arithmetic's is, not tied to full scale.  Both are written as four signed base-256
// digits each (d0 most significant: value = sum_i d_i 256^(3-i)).  The dot product of two such numbers is
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
// Data layout.  X digit planes: plane p (digit d_p), 4-frame block b, channel c -> one dword holding frames 4b..4b+3 of that
// channel (byte q = frame 4b + q): [p][b][c].  Linear frame lin (history ++ input) lives in block (lin + I8_PADF) / 4.  A tile
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
constexpr int I8_PADF = 64;               // zero frames in front of linear frame 0 in the digit planes
constexpr float I8_SCALE = 1073741824.0f; // 2^30: the filter rows' fixed point
constexpr float I8_LIMIT = 1.98f;         // |row value| the digits can hold: (2^31 - 2^23) / 2^30, rounded down

struct I8Geom {
    int g;                                // period stride inside a tile
    int gq4;                              // g * Q / 4: blocks between consecutive columns' periods
    int super_groups;                     // groups of g * ppw consecutive periods
    int sg_per_xcd;
    unsigned char *a_planes;
    unsigned long long *a_masks;          // [variant][row]: bit c set = chunk c of the row has a non-zero most significant digit
    const unsigned char *x_planes;        // (written through x_planes_w by the staging pass)
    unsigned int *x_planes_w;
    unsigned int x_blocks;                // 4-frame blocks per plane
    size_t x_plane_bytes;                 // x_blocks * C * 4
    int *flag; int epoch;                 // *flag == epoch: this launch cannot run in fixed point (set by the staging pass)
    unsigned int *peak;                   // [C] bits of the channel's peak |x| over history ++ input (i8_peak_kernel; zeroed again by the main kernel)
    int *shifts;                          // [C] the channel's samples are quantised as rint (x * 2^shift) (written by the staging pass)
};

// binary exponent for a channel whose peak magnitude has these float bits: peak * 2^shift in [2^29, 2^31 - 2^23) — as large as the
// digits hold (the top digit of (q + 0x80808080) must not overflow).  Zero and denormal peaks take the smallest normal's exponent.
__device__ __forceinline__ int shift_of_peak (unsigned int bits)
{
    const int e = max ((int)(bits >> 23), 1);
    return 157 - e - ((bits & 0x7f0000u) == 0x7f0000u ? 1 : 0);
}

// Per-channel peak |x| of a launch's history ++ input (both interleaved, 16-byte aligned, C a power of two <= 32): magnitudes
// compared as unsigned bit patterns — monotone for finite values, infinities above them, NaNs above those.  One pass at HBM
// speed; the staging pass behind it re-reads the same bytes from the Infinity Cache.
constexpr int I8_PEAK_THREADS = 1024;
__global__ __launch_bounds__ (I8_PEAK_THREADS)
void i8_peak_kernel (ArtFirArgs a, unsigned int *peak)
{
    __shared__ unsigned int s_peak [32];
    const int tid = threadIdx.x;
    if (tid < 32) s_peak [tid] = 0u;
    __syncthreads ();
    const size_t stride = (size_t) gridDim.x * I8_PEAK_THREADS;   // (in 4-float vectors: 4 * stride is a multiple of C, a thread's channels never change)
    const size_t v0 = (size_t) blockIdx.x * I8_PEAK_THREADS + tid;
    unsigned int m [4] = { 0u, 0u, 0u, 0u };
    auto take = [&] (const u32x4 &v) {
        m [0] = max (m [0], v.x & 0x7fffffffu); m [1] = max (m [1], v.y & 0x7fffffffu);
        m [2] = max (m [2], v.z & 0x7fffffffu); m [3] = max (m [3], v.w & 0x7fffffffu);
    };
    for (int part = 0; part < 2; ++part) {
        const float *src = part ? a.in : a.hist;
        const size_t n = part ? (size_t) a.in_frames * a.C : (size_t) a.H * a.C;
        if (!src || !n) continue;
        const u32x4 *sv = reinterpret_cast<const u32x4 *> (src);
        const size_t nv = n / 4;
        size_t v = v0;
        for (; v + 3 * stride < nv; v += 4 * stride) {         // four loads in flight per thread
            const u32x4 x0 = sv [v], x1 = sv [v + stride], x2 = sv [v + 2 * stride], x3 = sv [v + 3 * stride];
            take (x0); take (x1); take (x2); take (x3);
        }
        for (; v < nv; v += stride) take (sv [v]);
        if (v0 == 0)                                            // (up to 3 values behind the last whole vector)
            for (size_t i = nv * 4; i < n; ++i) atomicMax (&s_peak [i % a.C], __float_as_uint (src [i]) & 0x7fffffffu);
    }
    // lanes that share a channel set first (xor offsets that are multiples of C / 4 lanes), then one LDS atomic per wave and channel
    const int lanes_per_set = a.C >= 4 ? a.C / 4 : 1;
    for (int off = 32; off >= lanes_per_set; off >>= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) m [e] = max (m [e], (unsigned int) __shfl_xor ((int) m [e], off));
    if ((tid & 63) < lanes_per_set)
#pragma unroll
        for (int e = 0; e < 4; ++e) if (m [e]) atomicMax (&s_peak [(int)((v0 * 4 + e) % a.C)], m [e]);
    __syncthreads ();
    // (most workgroups find a peak some other has already reported: the plain read spares the contended atomic)
    if (tid < a.C && s_peak [tid] > __builtin_nontemporal_load (&peak [tid])) atomicMax (&peak [tid], s_peak [tid]);
}

// digits of a fixed-point value as one dword: byte 3 = d0 ... byte 0 = d3, each signed
__device__ __forceinline__ unsigned int digits_of (int q) { return ((unsigned int) q + 0x80808080u) ^ 0x80808080u; }

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

// Staging pass, one launch, two roles by block index:
//   blocks [0, slot_tiles * 32): one effective row each (same arithmetic as mfma_prepare_kernel up to the rounding: the fp64
//       blend goes straight to fixed point, not through float) -> A digit planes;
//   the rest: X digit planes of history ++ input, one thread per (4-frame block, channel).
template <bool INTERP>
__global__ __launch_bounds__ (256)
void i8_stage_kernel (ArtFirArgs a, ArtSegTable segs, MfmaGeom g, I8Geom q)
{
    const int tid = threadIdx.x;
    const unsigned int a_blocks = (unsigned int)(g.slot_tiles * q.g) * 32u;
    if (blockIdx.x < a_blocks) {
        const int variant = blockIdx.x >> 5, row = blockIdx.x & 31;
        const int st = variant / q.g, jr = variant - st * q.g;
        const int rows_valid = min (32, g.P - st * 32);
        const Pos p0 = locate<INTERP> (a, segs, a.n_begin + st * 32);
        const Pos p = locate<INTERP> (a, segs, a.n_begin + st * 32 + min (row, rows_valid - 1));
        const float *h0 = a.bank + (size_t) p.fi * a.T;
        // K column 0 of this tile family sits r frames before the first slot's window (the start of its 4-frame block)
        const int r = max (p0.ip - a.T / 2 + 1 + jr * g.Q + I8_PADF, 0) & 3;
        const int shift = p.ip - p0.ip + r;
        bool bad = false;
        __shared__ unsigned long long s_mask;
        __shared__ int s_pass [2];
        if (tid == 0) { s_mask = 0ull; s_pass [0] = -1; s_pass [1] = 0; }
        __syncthreads ();
        // what mfma_prepare_kernel leaves for the streaming kernels is written here: that kernel is not launched at all then
        if (jr == 0 && row == 0) {
            if (tid == 0) {
                g.tile_w0 [3 * st] = p0.ip - a.T / 2 + 1;
                if (st == 0) a.fix_count [0] = 0;
            }
            if (st == 0 && tid < a.C) {                        // the launch's per-channel exponents, for the main kernel's final scaling
                const unsigned int pk = q.peak [tid];
                q.shifts [tid] = shift_of_peak (pk);
                if (pk >= 0x7f800000u) bad = true;              // an infinity or a NaN somewhere in the channel: no exponent holds it
            }
            if (!INTERP && !a.lowpass && tid < rows_valid) {
                const Pos pq = locate<INTERP> (a, segs, a.n_begin + st * 32 + tid);
                if ((pq.fi % a.F) == 0) { s_pass [0] = tid; s_pass [1] = pq.ip + pq.fi / a.F; }
            }
        }
        if (jr == 0) {
            // ... including the effective rows in float and the canonical positions (what the f32 streaming kernel stages)
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
        }
        unsigned long long mine = 0ull;
        unsigned char *base = q.a_planes + (size_t) variant * (g.ktot / I8_KC) * 4096 + row * 32;
        for (int k4 = tid * 4; k4 < g.ktot; k4 += 1024) {
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
            for (int pn = 0; pn < 4; ++pn) *reinterpret_cast<unsigned int *> (base + (size_t)(k4 >> 5) * 4096 + pn * 1024 + (k4 & 31)) = pl [pn];
            if (pl [0]) mine |= 1ull << (k4 >> 5);
        }
        if (mine) atomicOr (&s_mask, mine);
        __syncthreads ();
        if (tid == 0) {
            q.a_masks [variant * 32 + row] = s_mask;
            if (jr == 0 && row == 0) { g.tile_w0 [3 * st + 1] = s_pass [0]; g.tile_w0 [3 * st + 2] = s_pass [1]; }
        }
        if (bad) *q.flag = q.epoch;
        return;
    }
    const size_t e = (size_t)(blockIdx.x - a_blocks) * 256 + tid;
    const size_t total = (size_t) q.x_blocks * a.C;
    if (e >= total) return;
    const int b = (int)(e / a.C), c = (int)(e - (size_t) b * a.C);
    unsigned int s [4], pl [4];
    // the channel's block exponent: |x| <= peak, so |x * 2^shift| < 2^31 - 2^23 and the scaling itself is exact (v_ldexp_f32)
    const int shift = shift_of_peak (q.peak [c]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int lin = 4 * b + t - I8_PADF;
        float v = 0.0f;
        if (lin >= 0 && lin < a.H) v = a.hist [(size_t) lin * a.C + c];
        else if (lin >= a.H && lin - a.H < a.in_frames) v = a.in [(size_t)(lin - a.H) * a.C + c];
        // (the call's head as one contiguous float array, for the stand-by kernel's tiles that reach into the history)
        if (g.head && lin + MF_HEAD_PAD >= 0 && lin + MF_HEAD_PAD < g.head_frames) g.head [(size_t)(lin + MF_HEAD_PAD) * a.C + c] = v;
        s [t] = digits_of (__float2int_rn (ldexpf (v, shift)));   // (a channel with an infinity or a NaN: garbage, the launch is flagged)
    }
    to_planes (s, pl);
#pragma unroll
    for (int pn = 0; pn < 4; ++pn) q.x_planes_w [(size_t) pn * (q.x_plane_bytes / 4) + e] = pl [pn];
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
    // (the staging pass has turned the peaks into exponents: the next launch's peak pass finds them at zero again)
    if (blockIdx.x == 0 && tid < 32) q.peak [tid] = 0u;
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
    const int tiles_per_xcd = q.sg_per_xcd * q.g * g.slot_tiles;
    const int nchunks = g.ktot / I8_KC;

    // tile `within` of this XCD's list -> (slot tile, first period); false if the tile holds no output of the launch
    auto tile_at = [&] (int within, int &st, int &j0) -> bool {
        st = within % g.slot_tiles;
        const int t2 = within / g.slot_tiles, jr = t2 % q.g, sg = xcd * q.sg_per_xcd + t2 / q.g;
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
        // A: thread -> (plane, row, 16-tap half), consecutive threads on consecutive 16 bytes of the chunk's 4 KB
        const int a_plane = pt >> 6, a_row = (pt >> 1) & 31, a_half = pt & 1;
        const unsigned int a_off0 = (unsigned int) pt * 16u;
        const int adst = a_plane * (32 * I8_PITCH) + a_row * I8_PITCH + a_half * 16;
        unsigned int boff [NB]; int bdst [NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int v = pt + u * MF_THREADS;
            const int m = v / VPP, rem = v % VPP, kb = rem / VPF, cv = rem % VPF;
            boff [u] = (unsigned int)((m * q.gq4 + kb) * CG + cv * VEC) * 4u;       // (the tile's first block sits in the resource base)
            bdst [u] = (m * CG + cv * VEC) * I8_PITCH + kb * 4;
        }
        // two register stages: the loads of chunk c + 3 are issued while those of c + 2 are still in flight (a chunk is ~0.5 us
        // of matrix work, less than a loaded L2 round trip: with one stage the kernel ran at the memory latency, 139 us)
        unsigned int ra0 [2] [4], rb0 [2] [4] [NB * VEC];

        int f_within = rank - wgs_per_xcd, f_chunk = 0;
        bool f_live = false;
        const unsigned char *fa_base = nullptr, *fb_base = nullptr;
        unsigned int fa_bytes = 0, fb_bytes = 0;
        const size_t x_total = 4 * q.x_plane_bytes;
        auto open_tile = [&] () {                             // next tile of this workgroup's list that holds outputs
            int st = 0, j0 = 0;
            f_live = false;
            for (f_within += wgs_per_xcd; f_within < tiles_per_xcd; f_within += wgs_per_xcd)
                if (tile_at (f_within, st, j0)) { f_live = true; break; }
            if (!f_live) return;
            // (readfirstlane: the table entry arrives in a vector register, and a resource built from it would make every load a
            // waterfall loop; the value is the same in all lanes)
            const int la = max (__builtin_amdgcn_readfirstlane (g.tile_w0 [3 * st]) + j0 * g.Q + I8_PADF, 0);
            size_t skip = (size_t)(la >> 2) * CG * 4;
            if (skip > q.x_plane_bytes) skip = q.x_plane_bytes;
            fb_base = q.x_planes + skip; fb_bytes = (unsigned int) min (x_total - skip, (size_t) 0xfffffff0u);
            fa_bytes = (unsigned int) nchunks * 4096u;
            fa_base = q.a_planes + (size_t)(st * q.g + j0 % q.g) * fa_bytes;
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
                    for (int u = 0; u < NB; ++u) PlaneLoad<VEC>::load (&rb0 [SET] [pn] [u * VEC], rb_, boff [u], (unsigned int)(pn * q.x_plane_bytes));
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
    // rows carry 30 fraction bits, this lane's channel `shift`; the class sums are combined at weight 256^(4 - s) of 2^16 units
    const double out_scale = __builtin_ldexp (1.0, -14 - q.shifts [c]);

    __syncthreads ();                                        // the staging waves have committed chunk 0
    int qn = 0;                                              // chunks consumed so far: chunk qn sits in LDS buffer qn & 1
    for (int within = rank; within < tiles_per_xcd; within += wgs_per_xcd) {
        int st, j0;
        if (!tile_at (within, st, j0)) continue;
        i32x16 acc [5];
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc [s] [r] = 0;
        // chunks in which some row of this tile has a non-zero most significant digit (the few around the rows' centres: taps
        // fall off as 1 / distance): everywhere else the four products with that digit plane are exactly zero and not issued
        unsigned long long top = q.a_masks [(st * q.g + j0 % q.g) * 32 + (lane & 31)];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) top |= __shfl_xor (top, off);
        const unsigned int top_lo = __builtin_amdgcn_readfirstlane ((unsigned int) top), top_hi = __builtin_amdgcn_readfirstlane ((unsigned int)(top >> 32));

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
            for (int i = 1; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j <= 4) acc [i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (av [i], bv [j], acc [i + j], 0, 0, 0);
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
        const int pass_row = PASS ? g.tile_w0 [3 * st + 1] : -1, pass_lin = PASS ? g.tile_w0 [3 * st + 2] : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i_const = (r & 3) + 8 * (r >> 2);      // compile-time part of the slot
            // class sums, weights 256^(4 - s), in fp64 (|total| < 2^58: the four roundings are 2^-53 relative), scaled back by the
            // rows' and the channel's exponents (a power of two: exact) and rounded ONCE to float
            double v = (double) acc [0] [r];
#pragma unroll
            for (int s = 1; s < 5; ++s) v = v * 256.0 + (double) acc [s] [r];
            float y = (float)(v * out_scale);
            const int i = i_const + 4 * (lane >> 5);
            if constexpr (PASS) {
                // nearest-filter mode, the position falls exactly on an input sample: the reference copies it (resampler.c:1166-1170)
                if (pass_row == i) y = load_frame (a, INT_MIN, pass_lin + (j0 + jl * q.g) * g.Q, c);
            }
            if (col_live && i < rows_valid)                  // (frames at or past n_end: out of the resource's range, dropped)
                __builtin_amdgcn_raw_buffer_store_b32 (__float_as_uint (y), rs_out, (int)(out_off + (unsigned int)(i_const * CG) * 4u), 0, 0);
        }
    }
}

} // namespace

// The planes buffer of a launch: [header: flag, peaks, exponents (art_internal.h)][row masks][A digit planes][X digit planes];
// returns its size, 0 if the launch is not for this path
static size_t i8_layout (const ArtFirArgs *a, const MfmaGeom &g, int cgt, I8Geom &q, char *base)
{
    if (!cgt || g.tile_rows != 32 || (g.ktot % I8_KC)) return 0;
    const int ppw = I8_COLS / cgt > I8_MAX_PPW ? I8_MAX_PPW : I8_COLS / cgt;
    q.g = (g.Q % 4 == 0) ? 1 : (g.Q % 2 == 0) ? 2 : 4;
    q.gq4 = q.g * g.Q / 4;
    const unsigned int total = a->n_end - a->n_begin, periods = (total + g.P - 1) / g.P;
    q.super_groups = (int)((periods + (unsigned int)(q.g * ppw) - 1) / (unsigned int)(q.g * ppw));
    q.sg_per_xcd = (q.super_groups + 7) / 8;
    const size_t a_bytes = (size_t) g.slot_tiles * q.g * (g.ktot / I8_KC) * 4096;
    // every frame a tile with an output in range can stage: the call's frames, then (ppw - 1) * g periods, K columns, slack
    const size_t frames = (size_t) I8_PADF + a->H + a->in_frames + (size_t) ppw * q.g * g.Q + g.ktot + 160;
    q.x_blocks = (unsigned int)((frames + 3) / 4);
    q.x_plane_bytes = (size_t) q.x_blocks * a->C * 4;
    if (4 * q.x_plane_bytes >= 0xffff0000ull) return 0;               // (plane offsets are 32-bit)
    if (g.ktot / I8_KC > 64) return 0;                                  // (one mask bit per chunk)
    const size_t head = (ART_I8_HEAD_BYTES + (size_t) g.slot_tiles * q.g * 32 * 8 + 255) & ~(size_t) 255;
    q.flag = (int *) base; q.epoch = 0;
    q.peak = (unsigned int *)(base + ART_I8_PEAK_OFFSET); q.shifts = (int *)(base + ART_I8_SHIFT_OFFSET);
    q.a_masks = (unsigned long long *)(base + ART_I8_HEAD_BYTES);
    q.a_planes = (unsigned char *) base + head;
    q.x_planes_w = (unsigned int *)(base + head + a_bytes); q.x_planes = (const unsigned char *) q.x_planes_w;
    return head + a_bytes + 4 * q.x_plane_bytes;
}

size_t artfir_i8_bytes (const ArtFirArgs *a, const MfmaGeom &g, int cgt)
{
    I8Geom q;
    return i8_layout (a, g, cgt, q, nullptr);
}

int artfir_i8_launch (const ArtFirArgs *a, const ArtSegTable *segs, const MfmaGeom &g, int cgt, unsigned int roll_blocks, hipStream_t st)
{
    static std::atomic<int> launches {0};
    I8Geom q;
    if (!a->planes) return 0;
    const size_t need = i8_layout (a, g, cgt, q, (char *) a->planes);
    if (!need || need > a->planes_bytes) return 0;
    int ep = ++launches;
    if (ep <= 0) { launches = 1; ep = 1; }                             // (the flag word is zero when the buffer is allocated)
    q.epoch = ep;
    if (a->fixed_out) { a->fixed_out [0] = ep; a->fixed_out [1] = g.slot_tiles * q.g * 32; a->fixed_out [2] = g.ktot / I8_KC; }

    {   // per-channel peaks first: enough workgroups to pull at HBM speed, each thread a few vectors deep
        const size_t vecs = ((size_t) a->in_frames + a->H) * a->C / 4;
        unsigned int wgs = (unsigned int)((vecs + I8_PEAK_THREADS * 8 - 1) / (I8_PEAK_THREADS * 8));
        if (wgs > 512u) wgs = 512u;                                     // two per CU: 32 waves per CU, four 16-byte loads in flight each
        if (wgs < 1u) wgs = 1u;
        hipLaunchKernelGGL (i8_peak_kernel, dim3 (wgs), dim3 (I8_PEAK_THREADS), 0, st, *a, q.peak);
    }
    const unsigned int x_wgs = (unsigned int)(((size_t) q.x_blocks * a->C + 255) / 256);
    const dim3 pgrid ((unsigned int)(g.slot_tiles * q.g) * 32u + x_wgs);
    if (a->interpolate) hipLaunchKernelGGL (i8_stage_kernel<true>, pgrid, dim3 (256), 0, st, *a, *segs, g, q);
    else hipLaunchKernelGGL (i8_stage_kernel<false>, pgrid, dim3 (256), 0, st, *a, *segs, g, q);
    if (a->ev_start) arthip_event_record (a->ev_start, (void *) st);

    const int tiles_per_xcd = q.sg_per_xcd * q.g * g.slot_tiles;
    // As many workgroups as the XCD holds (32 CUs x 2: 60 KB of LDS, 128 registers each) less two slots for the history-roll
    // workgroups of the same grid, each striding the XCD's tile list; the last, partly filled round then runs with one
    // workgroup per CU and its tiles finish sooner.  (Equal shares — 56 workgroups x 5 tiles for the headline's 280 — kept 8
    // slots idle for the whole launch: 0.1078 ms, 64 workgroups 0.1039, 62 0.1029; 4 and 32 channels, 256k..1M frames,
    // 96k -> 44.1k: 3..9 % the same way.  ARTAMD_I8_WGS overrides, for experiments.)
    const int resident = 62;
    int wgs_per_xcd = tiles_per_xcd < resident ? tiles_per_xcd : resident;
    { static const int k_env = [] { const char *e = getenv ("ARTAMD_I8_WGS"); return e && *e ? atoi (e) : 0; } (); if (k_env > 0 && k_env < tiles_per_xcd) wgs_per_xcd = k_env; }
    const dim3 sgrid ((unsigned int)(8 * wgs_per_xcd) + roll_blocks);
    const bool pass = !a->interpolate && !a->lowpass;
#define I8_GO(CGT) do { if (pass) hipLaunchKernelGGL ((fir_i8_stream_kernel<CGT, true>), sgrid, dim3 (2 * MF_THREADS), 0, st, *a, g, q, wgs_per_xcd); \
                        else hipLaunchKernelGGL ((fir_i8_stream_kernel<CGT, false>), sgrid, dim3 (2 * MF_THREADS), 0, st, *a, g, q, wgs_per_xcd); } while (0)
    switch (cgt) { case 32: I8_GO (32); break; case 16: I8_GO (16); break; case 8: I8_GO (8); break; case 4: I8_GO (4); break; case 2: I8_GO (2); break; default: I8_GO (1); }
#undef I8_GO
    if (a->ev_stop) arthip_event_record (a->ev_stop, (void *) st);
    return 1;
}

#endif  // !ART_WIDE
