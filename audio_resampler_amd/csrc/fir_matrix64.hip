// fir_matrix64.hip — the matrix-core path of the 8-byte sample build (libartamd64.so, reference PATH_WIDTH=64): fp64 MFMA
// (v_mfma_f64_16x16x4_f64) over the same periodic-phase GEMM as fir_matrix.hip.
#include "fir_common.hip.h"

#if ART_WIDE

namespace {

// ---------------------------------------------------------------------------------------------------
// fp64 matrix-core kernel for rational ratios (the 8-byte sample build).
//
// Same periodic-phase GEMM as the single-precision kernel above (ratio = P/Q: output n + P sits Q input frames after
// output n at the same filter phase; 32 consecutive slots x 128 columns = periods x channels per workgroup), with
// two differences that come with the precision target (default mode within 2^-48 of the reference-order result):
//
//  * the lerp is NOT folded into one row per slot.  The reference's position arithmetic is quantised to ~1e-7
//    filter steps after a million frames, so the fractions of one slot differ from period to period by far more
//    than a double ulp.  The A tile carries both rows of every slot — rows 0-31: h[fi_i], rows 32-63: h[fi_i + 1],
//    each shifted to the tile's K origin — and the epilogue blends s0, s1 with the output's OWN fraction, taken
//    from the exact fp64 replay of its position.  An output may use the tile whenever its integer position and
//    filter index equal the slot's; anything else is evaluated directly in the epilogue (direct_sample).
//  * accumulation is fp64 throughout (v_mfma_f64_16x16x4_f64), so there is no flush scheme.
//
// 4 waves; wave w owns columns [32w, 32w+32) x all rows: 2 column tiles x (4 | 2) row tiles of 16x16.  K is staged
// through LDS in chunks of 16 ([row][k], pitch 18 doubles: conflict-free ds_read_b128); within a group of 8 k's
// the four lane groups of a wave take k pairs (0,1),(2,3),(4,5),(6,7) and two MFMAs consume first/second element —
// a fixed permutation of the summation order.  C/D layout of the f64 form: col = lane & 15, row = (lane >> 4) + 4 reg.
// ---------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__ ((ext_vector_type (4)));
typedef double f64x2 __attribute__ ((ext_vector_type (2)));
typedef unsigned int w_u32x4 __attribute__ ((ext_vector_type (4)));
typedef unsigned int w_u32x2 __attribute__ ((ext_vector_type (2)));

constexpr int MW_THREADS = 256;
constexpr int MW_KC = 16;                 // k's per staged chunk
constexpr int MW_LD = MW_KC + 2;          // LDS row pitch in doubles (144 B)
constexpr int MW_COLS = 128;
constexpr int MW_ROWS = 32;               // slots per workgroup
constexpr int MW_MAX_PPW = 64;

constexpr int MW_HEAD_PAD = 64;
struct WideGeom {
    int P, Q;
    int slot_tiles;                       // ceil (P / 32)
    int nrows;                            // A rows per slot tile: 64 (interpolating) or 32
    int ktot;                             // K columns, multiple of MW_KC
    int period_groups, groups_per_xcd;
    double *rows;                         // [slot_tiles][nrows][ktot]  filter rows shifted to the tile's K origin, zero padded
    int *canon_ip, *canon_fi;             // [slot_tiles*32]  canonical position of each slot (period 0 of the launch)
    double *head; int head_frames;        // the call's head as one array (history ++ first input frames, MW_HEAD_PAD zero frames in front)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc (const void *base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc (const_cast<void *> (base), 0, (int) bytes, 0x00020000);
}
__device__ __forceinline__ double u2d (unsigned int lo, unsigned int hi) { return __hiloint2double ((int) hi, (int) lo); }

// grid (slot tile, slot): canonical (ip, fi) of the slot and its one or two filter rows, laid out as the main kernel stages them
template <bool INTERP>
__global__ __launch_bounds__ (256)
void wide_prepare_kernel (ArtFirArgs a, ArtSegTable segs, WideGeom g)
{
    const int st = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
    const int rows_valid = min (MW_ROWS, g.P - st * MW_ROWS);
    const Pos p0 = locate<INTERP> (a, segs, a.n_begin + st * MW_ROWS);
    const Pos p = locate<INTERP> (a, segs, a.n_begin + st * MW_ROWS + min (row, rows_valid - 1));
    if (tid == 0) {
        g.canon_ip [st * MW_ROWS + row] = p.ip; g.canon_fi [st * MW_ROWS + row] = p.fi;
        if (st == 0 && row == 0) a.fix_count [0] = 0;
    }
    const double *h0 = a.bank + (size_t) p.fi * a.T;
    const int shift = p.ip - p0.ip;
    double *d0 = g.rows + ((size_t) st * g.nrows + row) * g.ktot;
    double *d1 = d0 + (size_t) MW_ROWS * g.ktot;
    for (int k = tid; k < g.ktot; k += 256) {
        const int tap = k - shift;
        const bool in = tap >= 0 && tap < a.T;
        d0 [k] = in ? h0 [tap] : 0.0;
        if (INTERP) d1 [k] = in ? h0 [tap + a.T] : 0.0;
    }
    // the call's head, gathered by the whole grid (see mfma_prepare_kernel)
    const int blocks = gridDim.x * gridDim.y, me = blockIdx.y * gridDim.x + blockIdx.x;
    const long total = (long) g.head_frames * a.C;
    for (long e = (long) me * 256 + tid; e < total; e += (long) blocks * 256) {
        const int f = (int)(e / a.C), c = (int)(e - (long) f * a.C), lin = f - MW_HEAD_PAD;
        double v = 0.0;
        if (lin >= 0 && lin < a.H) v = a.hist [(size_t) lin * a.C + c];
        else if (lin >= a.H && lin - a.H < a.in_frames) v = a.in [(size_t)(lin - a.H) * a.C + c];
        g.head [e] = v;
    }
}

template <bool INTERP, int CG>
__global__ __launch_bounds__ (MW_THREADS, 2)
void fir_mfma64_kernel (ArtFirArgs a, ArtSegTable segs, WideGeom g)
{
    constexpr int NROWS = INTERP ? 64 : 32, RT = NROWS / 16;           // A rows, row tiles
    constexpr int PPW = MW_COLS / CG > MW_MAX_PPW ? MW_MAX_PPW : MW_COLS / CG;
    __shared__ __attribute__ ((aligned (16))) double As [NROWS * MW_LD];
    __shared__ __attribute__ ((aligned (16))) double Bs [MW_COLS * MW_LD];
    __shared__ double s_frac [INTERP ? MW_ROWS * PPW : 1];             // the exact fraction of every (slot, period)
    __shared__ unsigned char s_status [MW_ROWS * PPW];                 // 0 ok, 1 off the pattern (evaluated directly), 2 masked, 3 pass-through
    __shared__ int s_fi [MW_ROWS], s_ip [MW_ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware mapping as in the single-precision kernel: XCD x takes a contiguous range of period groups
    // workgroups past the tile grid roll the history for the next call (see fir_mfma_kernel)
    const unsigned int tile_blocks = 8u * (unsigned int) g.groups_per_xcd * (unsigned int) g.slot_tiles;
    if (blockIdx.x >= tile_blocks) {
        if (a.roll_dst) {
            const int e = (int)(blockIdx.x - tile_blocks) * MW_THREADS + (int) threadIdx.x;
            if (e < a.H * a.C) {
                const int f = e / a.C, c = e - f * a.C, lin = a.roll_appended + f;
                double v = 0.0;
                if (lin < a.H) v = a.hist [(size_t) lin * a.C + c];
                else if (a.in && lin - a.H < a.in_frames) v = a.in [(size_t)(lin - a.H) * a.C + c];
                a.roll_dst [e] = v;
            }
        }
        return;
    }
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int st = within % g.slot_tiles, jg = xcd * g.groups_per_xcd + within / g.slot_tiles;
    if (jg >= g.period_groups) return;
    const int half = a.T / 2;
    const int r0 = st * MW_ROWS;
    const int rows_valid = min (MW_ROWS, g.P - r0);
    const unsigned int n_tile = a.n_begin + (unsigned int)(jg * PPW) * g.P + r0;
    if (n_tile >= a.n_end) return;

    const int j_first = jg * PPW;
    if (tid < MW_ROWS) { s_ip [tid] = g.canon_ip [st * MW_ROWS + tid] + j_first * g.Q; s_fi [tid] = g.canon_fi [st * MW_ROWS + tid]; }
    __syncthreads ();
    const int w0 = s_ip [0] - half + 1;                      // linear index of K column 0 (first period)

    for (int e = tid; e < MW_ROWS * PPW; e += MW_THREADS) {
        const int i = e & (MW_ROWS - 1), jl = e / MW_ROWS;
        const unsigned int n = n_tile + (unsigned int) jl * g.P + i;
        unsigned char status = 2;
        if (i < rows_valid && n < a.n_end) {
            const Pos p = locate<INTERP> (a, segs, n);
            const int dip = p.ip - (s_ip [i] + jl * g.Q), dfi = p.fi - s_fi [i];
            if (INTERP) { status = (dip == 0 && dfi == 0) ? 0 : 1; s_frac [e] = p.frac; }
            else {
                // (ip-1, fi=F) and (ip, fi=0) are the same position: row F is row 0 one tap later (resampler.c:156-168)
                status = (dip * a.F + dfi == 0) ? 0 : 1;
                if (status == 0 && !a.lowpass && (p.fi % a.F) == 0) status = 3;
            }
            if (status == 1) {
                atomicAdd (a.fix_count, 1u); atomicAdd (a.fix_count + 1, 1u);       // (diagnostics: resampleHipLastHandedBack)
            }
        }
        s_status [e] = status;
    }

    // ---- staging plan: raw buffer loads, everything out of range reads as 0.  Fixed per-thread offsets; the chunk moves the
    // resource bases on the scalar unit — no vector arithmetic per chunk beside the matrix pipe (see fir_mfma_kernel).  A tile
    // whose window starts inside the history stages from the call's contiguous head (wide_prepare_kernel), all others from `in`.
    const bool touches_hist = w0 < a.H;
    const int origin = touches_hist ? -MW_HEAD_PAD : a.H;                      // linear index of the base's first frame
    constexpr int NA = NROWS / 32;
    const int a_row = tid >> 3, a_kseg = (tid & 7) * 2;
    constexpr int VEC = CG >= 2 ? 2 : 1;
    constexpr int VPF = CG / VEC, VPP = MW_KC * VPF;
    constexpr int NB = (PPW * VPP) / MW_THREADS;
    static_assert ((PPW * VPP) % MW_THREADS == 0, "staging plan");
    static_assert (MW_THREADS % VPP == 0, "per-vector period step must be uniform");

    double ra [NA * 2], rb [NB * VEC];
    const unsigned int rows_bytes = (unsigned int)((size_t) NROWS * g.ktot * 8);
    const unsigned int in_bytes = touches_hist ? (unsigned int)((size_t) g.head_frames * CG * 8) : (unsigned int)((size_t) a.in_frames * CG * 8);
    const char *rows_base = reinterpret_cast<const char *> (g.rows + (size_t) st * NROWS * g.ktot);
    const char *in_base = touches_hist ? reinterpret_cast<const char *> (g.head) : reinterpret_cast<const char *> (a.in);
    const int aoff = (a_row * g.ktot + a_kseg) * 8;          // vector u / row tile m differ from the first by a UNIFORM step: scalar too
    const int boff = (max (w0 + (tid / VPP) * g.Q + (tid % VPP) / VPF - origin, 0) * CG + ((tid % VPP) % VPF) * VEC) * 8;
    auto lean_fetch = [&] (int chunk) {
#pragma unroll
        for (int m = 0; m < NA; ++m) {
            const unsigned int sa = min ((unsigned int) chunk * (MW_KC * 8u) + (unsigned int)(m * 32 * g.ktot) * 8u, rows_bytes);
            const w_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128 (wide_rsrc (rows_base + sa, rows_bytes - sa), aoff, 0, 0);
            ra [m * 2] = u2d (v.x, v.y); ra [m * 2 + 1] = u2d (v.z, v.w);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const unsigned int sb = min ((unsigned int) chunk * (MW_KC * CG * 8u) + (unsigned int)(u * (MW_THREADS / VPP) * g.Q) * (CG * 8u), in_bytes);
            const __amdgpu_buffer_rsrc_t r_in = wide_rsrc (in_base + sb, in_bytes - sb);
            if (VEC == 2) {
                const w_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128 (r_in, boff, 0, 0);
                rb [u * VEC] = u2d (x.x, x.y); rb [u * VEC + (VEC - 1)] = u2d (x.z, x.w);
            }
            else {
                const w_u32x2 x = __builtin_amdgcn_raw_buffer_load_b64 (r_in, boff, 0, 0);
                rb [u * VEC] = u2d (x.x, x.y);
            }
        }
    };
    auto commit = [&] () {
#pragma unroll
        for (int m = 0; m < NA; ++m) {
            f64x2 v; v [0] = ra [m * 2]; v [1] = ra [m * 2 + 1];
            *reinterpret_cast<f64x2 *> (&As [(m * 32 + a_row) * MW_LD + a_kseg]) = v;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int vi = tid + u * MW_THREADS;
            const int jl = vi / VPP, rem = vi % VPP, kk = rem / VPF, cv = rem % VPF;
#pragma unroll
            for (int e = 0; e < VEC; ++e) Bs [(jl * CG + cv * VEC + e) * MW_LD + kk] = rb [u * VEC + e];
        }
    };

    constexpr int ncols = PPW * CG;
    if (ncols < MW_COLS)                                     // unused columns stay zero for the whole kernel
        for (int e = tid; e < (MW_COLS - ncols) * MW_LD; e += MW_THREADS) Bs [ncols * MW_LD + e] = 0.0;

    f64x4 acc [RT][2];
#pragma unroll
    for (int m = 0; m < RT; ++m)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc [m][c] = f64x4 { 0.0, 0.0, 0.0, 0.0 };

    const int nchunks = g.ktot / MW_KC;
    const int frag = (lane & 15) * MW_LD + 2 * (lane >> 4);      // this lane's (row | column, k pair) inside a 16-wide tile
    lean_fetch (0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads ();                                    // previous chunk fully consumed (first pass: status table complete)
        commit ();
        __syncthreads ();
        if (chunk + 1 < nchunks) lean_fetch (chunk + 1);     // global loads fly while the matrix cores work
#pragma unroll
        for (int grp = 0; grp < MW_KC / 8; ++grp) {
            f64x2 av [RT], bv [2];
#pragma unroll
            for (int m = 0; m < RT; ++m) av [m] = *reinterpret_cast<const f64x2 *> (&As [m * 16 * MW_LD + frag + grp * 8]);
#pragma unroll
            for (int c = 0; c < 2; ++c) bv [c] = *reinterpret_cast<const f64x2 *> (&Bs [(wave * 32 + c * 16) * MW_LD + frag + grp * 8]);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < RT; ++m)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc [m][c] = __builtin_amdgcn_mfma_f64_16x16x4f64 (av [m][h], bv [c][h], acc [m][c], 0, 0, 0);
        }
    }

    // ---- epilogue.  Row tiles 0,1 hold row fi of slots 0-15 / 16-31, tiles 2,3 row fi+1 of the same slots.
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int col = wave * 32 + c * 16 + (lane & 15);
        if (col >= ncols) continue;
        const int jl = col / CG, ch = col - jl * CG;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = t * 16 + (lane >> 4) + 4 * r;
                const unsigned char status = s_status [jl * MW_ROWS + i];
                if (status == 0 || status == 3) {
                    const size_t n = (size_t) n_tile + (size_t) jl * g.P + i;
                    double y;
                    if (INTERP) {
                        const double frac = s_frac [jl * MW_ROWS + i];
                        const double left = acc [t][c][r] * (1.0 - frac);
                        const double right = acc [INTERP ? t + 2 : t][c][r] * frac;
                        y = left + right;
                    }
                    else if (status == 3) y = load_frame (a, INT_MIN, s_ip [i] + jl * g.Q + s_fi [i] / a.F, ch);
                    else y = acc [t][c][r];
                    a.out [n * CG + ch] = y;
                }
                else if (status == 1) {                        // off the canonical pattern: evaluated here at its exact position
                    const size_t n = (size_t) n_tile + (size_t) jl * g.P + i;
                    a.out [n * CG + ch] = direct_sample<INTERP> (a, INT_MIN, locate<INTERP> (a, segs, (unsigned int) n), ch);
                }
            }
    }
}
} // namespace

// does this call take the matrix-core path (arthip_fir), or the general kernel?  One rule, also asked by the batched entry
// point, which only gathers calls the general kernel would have run anyway.
bool artfir_takes_matrix_path (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref)
{
    if (a->n_end <= a->n_begin || (a->mode & 3) == ART_MODE_STRICT) return false;
    const unsigned int total = a->n_end - a->n_begin;
    const int Cs = a->stream_C > a->C ? a->stream_C : a->C;     // (a group of a wider stream decides as the stream does)
    const double k_ns = ((0.2 + 0.05 * Cs) + 0.00021 * Cs * a->T) * (a->T <= 256 ? 0.55 : a->T <= 512 ? 0.85 : 0.95);
    const double floor_ns = (15000.0 + 4100.0 * ((a->T + 63) / 32)) * ((Cs + 31) / 32);
    const bool enough = total * k_ns >= floor_ns - 5000.0;
    const bool small = (size_t) a->in_frames * a->C * 8 < 0x7fff0000ull && (size_t) a->H * a->C * 8 < 0x7fff0000ull &&
                       ((uintptr_t) a->in % 16) == 0 && ((uintptr_t) a->hist % 16) == 0;
    const int cgt = (small && (a->C == 1 || a->C == 2 || a->C == 4 || a->C == 8 || a->C == 16 || a->C == 32)) ? a->C : 0;
    // (a channel count the kernel is not compiled for: in groups of a compiled width, where the launch has the buffer for them — fir_dispatch.hip)
    const bool grouped = a->pad != nullptr && (a->C > 32 || (a->C & (a->C - 1)) != 0);
    return a->mode == ART_MODE_FAST && a->period_out > 0 && a->fix_list && a->scratch && a->in_pitch == 0 && a->out_pitch == 0 &&
                    segs->lin_floor == INT_MIN && kernel_pref != ART_KERNEL_GENERAL && (cgt != 0 || grouped) &&
                    (enough || kernel_pref >= ART_KERNEL_MFMA) && total >= (unsigned int) a->period_out;
}

// Launch the fp64 matrix-core path for this call if it applies: ART_KERNEL_MFMA (| ART_FIR_ROLLED), -1 on a launch failure,
// 0 when the call is for the general kernel.
size_t artfir_planes_bytes (const ArtFirArgs *, unsigned int, int) { return 0; }
size_t artfir_rows_bytes (const ArtFirArgs *, unsigned int, int) { return 0; }
void artfir_rows_touch (const ArtFirArgs *, const ArtSegTable *) { }
extern "C" {      // (the fixed-point kernel's rows across calls: 4-byte samples only)
size_t arthip_fir_rows_cache_bytes (void) { return 0; }
void arthip_fir_rows_cache_reset (void *) { }
void arthip_fir_rows_cache_free (void *) { }
}
size_t artfir_split_bytes (const ArtFirArgs *, unsigned int, int) { return 0; }
bool artfir_matrix_spans_segments (const ArtFirArgs *, const ArtSegTable *, int) { return false; }    // (the fp64 kernel checks every output's position against the table)

int artfir_matrix (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref, void *stream)
{
    hipStream_t st = (hipStream_t) stream;
    // fp64 matrix-core path: exact rational ratio, interleaved buffers, the stream's channel count one of the
    // compiled column groups, no history floor — and enough work: cost models as in the float build, fitted to this
    // build (tools/bench_small_taps.py --wide): general ~ 5 us + n (0.2 + 0.05 C + 0.00021 C T) ns, fp64 MFMA floor
    // ~ 15 us + 4.1 us per 32-tap chunk
    {
        const unsigned int total = a->n_end - a->n_begin;
        const bool small = (size_t) a->in_frames * a->C * 8 < 0x7fff0000ull && (size_t) a->H * a->C * 8 < 0x7fff0000ull &&
                           ((uintptr_t) a->in % 16) == 0 && ((uintptr_t) a->hist % 16) == 0;
        const int cgt = (small && (a->C == 1 || a->C == 2 || a->C == 4 || a->C == 8 || a->C == 16 || a->C == 32)) ? a->C : 0;
        const bool ok = cgt != 0 && artfir_takes_matrix_path (a, segs, kernel_pref);     // (cgt == 0 here: the groups could not run — no buffer — and the general kernel takes the call)
        if (ok) {
            WideGeom g;
            {   // (short or badly fitting periods: several at a time, fir_common.hip.h)
                const int mu = artfir_period_multiple (a->period_out, MW_ROWS);
                g.P = mu * a->period_out; g.Q = mu * a->period_in;
            }
            g.slot_tiles = (g.P + MW_ROWS - 1) / MW_ROWS;
            g.nrows = a->interpolate ? 64 : 32;
            const int shift_max = (int)((MW_ROWS - 1.0) * g.Q / g.P) + 2;
            g.ktot = ((a->T + shift_max + MW_KC - 1) / MW_KC) * MW_KC;
            const int ppw = MW_COLS / cgt > MW_MAX_PPW ? MW_MAX_PPW : MW_COLS / cgt;
            const unsigned int periods = (total + g.P - 1) / g.P;
            g.period_groups = (int)((periods + ppw - 1) / ppw);
            g.groups_per_xcd = (g.period_groups + 7) / 8;
            const size_t row_bytes = (size_t) g.slot_tiles * g.nrows * g.ktot * sizeof (double);
            char *base = (char *) a->scratch;
            g.rows = (double *) base;
            g.canon_ip = (int *)(base + ((row_bytes + 15) & ~(size_t) 15));
            g.canon_fi = g.canon_ip + (size_t) g.slot_tiles * MW_ROWS;
            const size_t used = (((size_t)((char *)(g.canon_fi + (size_t) g.slot_tiles * MW_ROWS) - base)) + 255) & ~(size_t) 255;
            g.head_frames = MW_HEAD_PAD + a->H + (ppw - 1) * g.Q + g.ktot + 3 * MW_KC;
            g.head = (double *)(base + used);
            if (base && used + (size_t) g.head_frames * a->C * sizeof (double) <= a->scratch_bytes &&
                (size_t) g.head_frames * a->C * 8 < 0x7fff0000ull && (size_t) g.nrows * g.ktot * 8 < 0x7fff0000ull) {
                const unsigned int roll_blocks = a->roll_dst ? (unsigned int)((a->H * a->C + MW_THREADS - 1) / MW_THREADS) : 0u;
                const dim3 grid ((unsigned int)(8 * g.groups_per_xcd * g.slot_tiles) + roll_blocks);
                if (a->interpolate) hipLaunchKernelGGL (wide_prepare_kernel<true>, dim3 (g.slot_tiles, MW_ROWS), dim3 (256), 0, st, *a, *segs, g);
                else hipLaunchKernelGGL (wide_prepare_kernel<false>, dim3 (g.slot_tiles, MW_ROWS), dim3 (256), 0, st, *a, *segs, g);
                if (a->ev_start) arthip_event_record (a->ev_start, stream);
#define MW_GO(I, CGT) hipLaunchKernelGGL ((fir_mfma64_kernel<I, CGT>), grid, dim3 (MW_THREADS), 0, st, *a, *segs, g)
                if (a->interpolate) switch (cgt) { case 32: MW_GO (true, 32); break; case 16: MW_GO (true, 16); break; case 8: MW_GO (true, 8); break;
                                                    case 4: MW_GO (true, 4); break; case 2: MW_GO (true, 2); break; default: MW_GO (true, 1); }
                else                switch (cgt) { case 32: MW_GO (false, 32); break; case 16: MW_GO (false, 16); break; case 8: MW_GO (false, 8); break;
                                                    case 4: MW_GO (false, 4); break; case 2: MW_GO (false, 2); break; default: MW_GO (false, 1); }
#undef MW_GO
                if (a->ev_stop) arthip_event_record (a->ev_stop, stream);
                return hipGetLastError () == hipSuccess ? (ART_KERNEL_MFMA | (a->roll_dst ? ART_FIR_ROLLED : 0)) : -1;
            }
        }
    }
    return 0;
}

#endif  // ART_WIDE
