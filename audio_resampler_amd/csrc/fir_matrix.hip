// fir_matrix.hip — the matrix-core (MFMA) path of the windowed-sinc interpolator for rational ratios, 4-byte samples (gfx950):
// mfma_prepare_kernel (per-launch effective rows and canonical slot positions), fir_mfma_stream_kernel (persistent workgroups,
// regular launches: the headline path), fir_mfma_kernel (one tile per workgroup with per-output position replay: every other
// launch), their shared K walk, and the host-side rules that pick between them and the general kernel.
#include "fir_matrix_stream.hip.h"

#if !ART_WIDE          // the 8-byte sample build has its own matrix-core kernel (fir_matrix64.hip)

namespace {

// ---------------------------------------------------------------------------------------------------
// MFMA kernel for rational ratios (the headline path: 44.1k -> 48k is 160 outputs per 147 inputs).
//
// When ratio = P/Q exactly, output n+P sits exactly Q input frames after output n, at the same filter
// phase.  Two facts turn the per-sample "two dot products + lerp" into one GEMM on the matrix cores:
//
//  (1) The lerp is linear, so for a fixed phase the two rows can be blended ONCE into an effective row
//          g_i[k] = h[fi_i][k] * (1 - frac_i) + h[fi_i + 1][k] * frac_i        (fp64, rounded once to f32)
//      shared by every period and channel — half the multiply-adds of the reference formulation.
//  (2) 32 consecutive outputs ("slots" r0..r0+31 of the P-periodic pattern) have windows shifted by
//      0/1 frames each, so their rows fit one zero-padded matrix over a common K = T + shift_max span:
//
//          Y[i, (j, c)] = sum_k A[i, k] * X[k, (j, c)],   A[i, k] = g_i[k - shift_i],  X[k, (j, c)] = x_c[w0 + j*Q + k]
//
//      a (32 x K) * (K x N) product, N = periods x channels, on exact-f32 v_mfma_f32_32x32x2_f32.
//
// A workgroup (4 waves) owns one slot tile and 128 columns, 32 per wave.  A and X chunks (32 k's) are
// staged through LDS in [row][k] layout (+4 pad: conflict-free ds_read_b128; the 8 k's of a group are
// split 4/4 over the two half-waves, a fixed permutation of the summation order).
//
// Accuracy.  An f32 accumulator that has swallowed the big central taps loses half an ulp on every
// further add, so no f32 partial sum is carried past a chunk: after each 32-k chunk the MFMA
// accumulator is added into an fp64 running sum and cleared, and inside the "centre band" (the chunks
// holding the central taps of any row) after every 4 k's.  At most 3 adds follow a row's peak tap in
// f32 — fewer full-magnitude roundings than the reference's own float loop.
//
// Exactness of positions.  Every output's (ip, fi, frac) is recomputed with the reference's own fp64
// arithmetic and compared with its slot's canonical values (taken from the workgroup's first period).
// Equal, or the same position within MF_PHASE_TOL filter steps (phase values on a filter boundary can round to
// the neighbouring (fi-1, frac ~ 1) representation; the effective rows differ by < 1e-10 relative):
// the tile's row is used.  Anything else (ratio drift, ring-epoch seams) is evaluated in the epilogue by the lane that
// owns the output, at its exact position (direct_sample) and counted (fix_count: diagnostics).
// ---------------------------------------------------------------------------------------------------

// Grid (slot tile, row): canonical (ip, fi, frac) of the slot from the first period of the launch, and its
// effective row g_i[k - shift_i] (lerp folded in, fp64, one rounding) laid out exactly as the main kernel
// stages it: [row][ktot], zero outside the row's T taps.
// the call's head, gathered by a whole grid: linear frame lin = index - MF_HEAD_PAD; history below H, input above
__device__ __forceinline__ void gather_head (const ArtFirArgs &a, const MfmaGeom &g, int blocks, int me, int tid)
{
    const long total = (long) g.head_frames * a.C;
    for (long e = (long) me * 256 + tid; e < total; e += (long) blocks * 256) {
        const int f = (int)(e / a.C), c = (int)(e - (long) f * a.C), lin = f - g.head_pad;
        float v = 0.0f;
        if (lin >= 0 && lin < a.H) v = a.hist [(size_t) lin * a.C + c];
        else if (lin >= a.H && lin - a.H < a.in_frames) v = a.in [(size_t)(lin - a.H) * a.C + c];
        g.head [e] = v;
    }
}
// launches on rows kept across calls: the head is all a call still has to prepare
// (interleaved frames on both sides: the head is MF_HEAD_PAD zero frames, the history, the call's first input frames — three runs of
// consecutive floats; one element per thread, no division, as many workgroups as it takes: the launch is a latency chain of one load)
__global__ __launch_bounds__ (256)
void mfma_head_kernel (ArtFirArgs a, MfmaGeom g)
{
    const unsigned int total = (unsigned int) g.head_frames * (unsigned int) a.C, e = blockIdx.x * 256u + threadIdx.x;
    if (e >= total) return;
    const unsigned int pad = (unsigned int) g.head_pad * (unsigned int) a.C, hist = (unsigned int) a.H * (unsigned int) a.C;
    float v = 0.0f;
    if (e >= pad) {
        unsigned int k = e - pad;
        if (k < hist) v = a.hist [k];
        else { k -= hist; if (k < (unsigned int) a.in_frames * (unsigned int) a.C) v = a.in [k]; }
    }
    g.head [e] = v;
}

template <bool INTERP>
__global__ __launch_bounds__ (256)
void mfma_prepare_kernel (ArtFirArgs a, ArtSegTable segs, MfmaGeom g, ArtRowsTable tb)
{
    const int st = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
    const int R = g.tile_rows;
    const int rows_valid = min (R, g.P - st * R);
    // slot k of the launch's first period by the reference's arithmetic — or, for rows kept across calls, slot k of the stream's CANONICAL
    // period from its constants (what the host evaluated, to the bit: fir_matrix_i8.hip, "The rows across calls")
    auto slot_pos = [&] (int k) -> Pos {
        if (tb.on) {
            const unsigned int n = tb.n0 + (unsigned int) k;
            const double step = n ? (double) n / a.ratio : 0.0;
            const double off = tb.base + step;
            const double whole = floor (off);
            Pos t;
            double fr = off - whole;
            fr = fr * (double) a.F;
            if (INTERP) { t.fi = (int) floor (fr); t.frac = fr - (double) t.fi; }
            else { t.fi = (int) floor (fr + 0.5); t.frac = 0.0; }
            t.ip = (int) whole + tb.lin + tb.w;
            return t;
        }
        return locate<INTERP> (a, segs, a.n_begin + (unsigned int) k);
    };
    // every thread derives the two positions it needs (uniform, a few dozen fp64 ops)
    const Pos p0 = slot_pos (st * R);
    const Pos p = slot_pos (st * R + min (row, rows_valid - 1));
    if (tid == 0) {
        g.canon_ip [st * R + row] = p.ip; g.canon_fi [st * R + row] = p.fi; g.canon_frac [st * R + row] = p.frac;
        if (st == 0 && row == 0) a.fix_count [0] = 0;       // per-launch count of off-pattern outputs (the main kernel follows in-stream)
        if (row == 0 && g.tile_w0) g.tile_w0 [3 * st] = p0.ip - a.T / 2 + 1;
    }
    if (row == 0 && g.tile_w0) {
        // nearest-filter mode without a low-pass: the slots of this tile whose ROUNDED filter index is a whole sample (reference
        // resampler.c:1141: fi % F == 0 — one slot per period when F = P, several when F < P), one bit per row; such a slot's
        // sample index is its canonical ip + fi / F
        __shared__ unsigned int s_pass;
        if (tid == 0) s_pass = 0u;
        __syncthreads ();
        if (!INTERP && !a.lowpass && tid < rows_valid && tid < 32) {
            const Pos q = slot_pos (st * R + tid);
            if ((q.fi % a.F) == 0) atomicOr (&s_pass, 1u << tid);
        }
        __syncthreads ();
        if (tid == 0) { g.tile_w0 [3 * st + 1] = (int) s_pass; g.tile_w0 [3 * st + 2] = 0; }
    }
    const float *h0 = a.bank + (size_t) p.fi * a.T;
    const int shift = p.ip - p0.ip;
    float *dst = g.eff + (size_t)(st * R + row) * g.ktot;
    for (int k = tid; k < g.ktot; k += 256) {
        const int tap = k - shift;
        float c = 0.0f;
        if (tap >= 0 && tap < a.T) {
            if (INTERP) {
                const double left = (double) h0 [tap] * (1.0 - p.frac);
                const double right = (double) h0 [tap + a.T] * p.frac;
                c = (float)(left + right);
            }
            else c = h0 [tap];
        }
        dst [k] = c;
    }
    // the call's head, gathered by the whole grid
    if (g.head) gather_head (a, g, (int)(gridDim.x * gridDim.y), (int)(blockIdx.y * gridDim.x + blockIdx.x), tid);
}

// the flagged slots of every period of a launch <- their input samples (see artfir_pass_fixup_wanted)
__global__ __launch_bounds__ (256)
void pass_fixup_kernel (ArtFirArgs a, MfmaGeom g)
{
    // the flagged slots in ascending order — the SAME list in every workgroup: the items below are shared out by list index
    // (built with atomics the order differed from workgroup to workgroup and a slot's copy could be made twice here, never there)
    __shared__ int s_slots [1024];
    __shared__ int s_off [1025];
    const int tid = threadIdx.x;
    const int tiles = (g.P + 31) >> 5;                        // (<= 1024: artfir_pass_fixup)
    auto word = [&] (int st) -> unsigned int {
        const int valid = min (32, g.P - st * 32);
        return (unsigned int) g.tile_w0 [3 * st + 1] & (valid >= 32 ? 0xffffffffu : (1u << valid) - 1u);
    };
    for (int st = tid; st < tiles; st += 256) s_off [st + 1] = __popc (word (st));
    __syncthreads ();
    if (tid == 0) { s_off [0] = 0; for (int st = 0; st < tiles; ++st) s_off [st + 1] += s_off [st]; }
    __syncthreads ();
    for (int st = tid; st < tiles; st += 256) {
        unsigned int w = word (st); int o = s_off [st];
        while (w) { const int b = __ffs ((int) w) - 1; if (o < 1024) s_slots [o] = st * 32 + b; ++o; w &= w - 1u; }
    }
    __syncthreads ();
    if (s_off [tiles] == 0) return;
    const unsigned int total = a.n_end - a.n_begin;
    const unsigned int periods = (total + (unsigned int) g.P - 1u) / (unsigned int) g.P;
    if (s_off [tiles] > 1024) {
        // (more flagged slots than the list holds — a period of thousands of slots with few filters: every slot of every period is looked at)
        const unsigned long long all = (unsigned long long) periods * (unsigned int) g.P * a.C;
        for (unsigned long long e = (unsigned long long) blockIdx.x * 256 + tid; e < all; e += (unsigned long long) gridDim.x * 256) {
            const int c = (int)(e % a.C);
            const unsigned long long r = e / a.C;
            const int slot = (int)(r % (unsigned int) g.P);
            const unsigned int j = (unsigned int)(r / (unsigned int) g.P);
            const unsigned int n = a.n_begin + j * (unsigned int) g.P + (unsigned int) slot;
            if (((word (slot >> 5) >> (slot & 31)) & 1u) && n < a.n_end && (j || slot >= a.n_skip))
                a.out [(size_t) n * a.C + c] = load_frame (a, INT_MIN, g.canon_ip [slot] + g.w_shift + g.canon_fi [slot] / a.F + (int) j * g.Q, c);
        }
        return;
    }
    const int nflag = s_off [tiles];
    const unsigned long long items = (unsigned long long) periods * nflag * a.C;
    for (unsigned long long e = (unsigned long long) blockIdx.x * 256 + tid; e < items; e += (unsigned long long) gridDim.x * 256) {
        const int c = (int)(e % a.C);
        const unsigned long long r = e / a.C;
        const int f = (int)(r % nflag);
        const unsigned int j = (unsigned int)(r / nflag);
        const int slot = s_slots [f];
        const unsigned int n = a.n_begin + j * (unsigned int) g.P + (unsigned int) slot;
        if (n < a.n_end && (j || slot >= a.n_skip))
            a.out [(size_t) n * a.C + c] = load_frame (a, INT_MIN, g.canon_ip [slot] + g.w_shift + g.canon_fi [slot] / a.F + (int) j * g.Q, c);
    }
}

// CG > 0: the stream has exactly CG channels (compile-time index math, vector loads);  CG == 0: any count.
// WS (wave specialisation, needs CG > 0): the workgroup has 8 waves.  Waves 4-7 are LOADERS — they prefetch
// chunk c+2 from global memory into registers and commit chunk c+1 (lerp folded in) to the other LDS buffer;
// waves 0-3 are the MATRIX waves — LDS operand reads, the MFMA chain and its fp64 flush, nothing else.
// Each SIMD hosts one wave of each kind per workgroup, so staging (VALU/VMEM/LDS-write) and matrix work
// overlap in hardware with one barrier per chunk, instead of relying on instruction scheduling.
// MT: MFMA m-tiles (32 slots each) per workgroup.  With MT = 2 the workgroup owns 64 consecutive slots whose two
// A tiles multiply the SAME X tile: X staging and barriers per MFMA halve and every matrix wave runs two
// independent accumulator chains.
template <bool INTERP, int CG, bool WS, int MT>
// (six waves per SIMD = three workgroups per CU for the shipped form: stated as waves per EU, which the register allocator
// honours — 80 VGPRs, no spills — where the launch-bounds hint alone let it drift to 86; the stereo instantiation needs
// those 86 and runs two workgroups per CU rather than spill)
__global__ __launch_bounds__ (WS ? 2 * MF_THREADS : MF_THREADS) __attribute__ ((amdgpu_waves_per_eu ((WS && MT == 1) ? 6 : 2)))
void fir_mfma_kernel (ArtFirArgs a, ArtSegTable segs, MfmaGeom g)
{
    constexpr int THREADS = WS ? 2 * MF_THREADS : MF_THREADS;
    constexpr int NBUF = WS ? 2 : 1;
    constexpr int ROWS = 32 * MT;
    __shared__ __attribute__ ((aligned (16))) float As_ [NBUF] [ROWS * MF_LD];
    __shared__ __attribute__ ((aligned (16))) float Bs_ [NBUF] [MF_COLS * MF_LD];
    __shared__ unsigned char s_status [ROWS * MF_MAX_PPW];    // 0 ok, 1 off the pattern (evaluated directly), 2 masked, 3 pass-through
    __shared__ int s_fi [ROWS], s_shift [ROWS], s_ip [ROWS];
    __shared__ double s_frac [ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = WS && wave >= 4;
    const int pt = WS ? (tid & (MF_THREADS - 1)) : tid;        // staging thread index (loaders, or everybody)
    // XCD-aware tile mapping.  Workgroup b is dispatched to XCD b % 8, each with a private 4 MiB L2.  The
    // slot tiles of one period group read the same input span and consecutive period groups overlap, so
    // XCD x takes the contiguous period groups [x*gpx, (x+1)*gpx) and all their slot tiles: its L2 then
    // holds ~2 MB of input + the phase rows instead of seeing the whole call (measured before the remap:
    // 46 % L2 misses, 16x the algorithmic bytes fetched from the fabric; after: 4 % and 1.6x).
    // Placement only affects speed.
    // Workgroups past the tile grid (x only, y == 0) roll the history for the next call — reads hist ++ in, writes the OTHER
    // history buffer: independent of everything else in flight, and one launch less per call.
    const unsigned int tile_blocks = 8u * (unsigned int) g.groups_per_xcd * (unsigned int) g.slot_tiles;
    if (blockIdx.x >= tile_blocks) {
        if (blockIdx.y == 0 && a.roll_dst) {
            const int e = (int)(blockIdx.x - tile_blocks) * THREADS + tid;
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
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int st = within % g.slot_tiles, jg = xcd * g.groups_per_xcd + within / g.slot_tiles;
    if (jg >= g.period_groups) return;
    const int cg = CG ? CG : g.cg, ppw = CG ? (MF_COLS / CG > MF_MAX_PPW ? MF_MAX_PPW : MF_COLS / CG) : g.ppw;
    const int ch_base = blockIdx.y * cg;
    const int half = a.T / 2;
    const int r0 = st * ROWS;
    const int rows_valid = min (ROWS, g.P - r0);
    const unsigned int n_tile = a.n_begin + (unsigned int)(jg * ppw) * g.P + r0;       // slot 0, first period
    if (n_tile >= a.n_end) return;

    // ---- canonical (ip, fi, frac) of the 32 slots: period 0 of the launch (mfma_prepare_kernel), moved to
    // this workgroup's first period
    const int j_first = jg * ppw;
    if (tid < ROWS) {
        s_ip [tid] = g.canon_ip [st * ROWS + tid] + j_first * g.Q;
        s_fi [tid] = g.canon_fi [st * ROWS + tid]; s_frac [tid] = g.canon_frac [st * ROWS + tid];
    }
    __syncthreads ();
    const int w0 = s_ip [0] - half + 1;                      // linear index of K column 0 (first period)
    if (tid < ROWS) s_shift [tid] = s_ip [tid] - s_ip [0];

    // ---- exact position of every (slot, period) of the tile, checked against the canonical pattern
    for (int e = tid; e < ROWS * ppw; e += THREADS) {
        const int i = e & (ROWS - 1), jl = e / ROWS;
        const unsigned int n = n_tile + (unsigned int) jl * g.P + i;
        unsigned char status = 2;
        if (i < rows_valid && n < a.n_end) {
            const Pos p = locate<INTERP> (a, segs, n);
            const int dip = p.ip - (s_ip [i] + jl * g.Q), dfi = p.fi - s_fi [i];
            if (dip == 0 && dfi == 0 && p.frac == s_frac [i]) status = 0;
            else if (INTERP) {
                const double d = (double)(dip * a.F + dfi) + (p.frac - s_frac [i]);   // signed distance in filter steps
                status = (d >= -MF_PHASE_TOL && d <= MF_PHASE_TOL) ? 0 : 1;
            }
            // nearest-filter mode: (ip-1, fi=F) and (ip, fi=0) are the same position — row F is row 0 one tap later
            // (resampler.c:156-168), so window and products are identical, as is the pass-through sample
            else status = (dip * a.F + dfi == 0) ? 0 : 1;
            if (!INTERP && status == 0 && !a.lowpass && (p.fi % a.F) == 0) status = 3;
            if (status == 1) { atomicAdd (a.fix_count, 1u); atomicAdd (a.fix_count + 1, 1u); }       // (diagnostics: resampleHipLastHandedBack)
        }
        s_status [e] = status;
    }
    __syncthreads ();

    // ---- staging plan.  All loads are raw buffer loads; everything out of range reads as 0.
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc (a.in, (unsigned int)((size_t) a.in_frames * a.C * 4));
    const __amdgpu_buffer_rsrc_t rs_hist = make_rsrc (a.hist, (unsigned int)((size_t) a.H * a.C * 4));
    const bool touches_hist = w0 < a.H;                      // only the first period group of a call

    // A: thread -> (row, 4 consecutive k) of the prepared effective rows: one aligned dwordx4 per chunk
    const int a_row = pt >> 3, a_kseg = (pt & 7) * 4;
    const __amdgpu_buffer_rsrc_t rs_eff = make_rsrc (g.eff + (size_t) st * ROWS * g.ktot, (unsigned int)((size_t) ROWS * g.ktot * 4));
    const unsigned int a_off0 = (unsigned int)(a_row * g.ktot + a_kseg) * 4u;
    const unsigned int a_tile_stride = (unsigned int)(32 * g.ktot) * 4u;       // second m-tile: 32 rows further

    // B: thread -> NB vectors of VEC channels of one frame of one period
    constexpr int VEC = CG >= 4 ? 4 : (CG == 2 ? 2 : 1);
    constexpr int VPF = CG ? CG / VEC : 1;                   // vectors per frame
    constexpr int VPP = MF_KC * VPF;                         // vectors per period-chunk
    constexpr int PPW_C = CG ? (MF_COLS / CG > MF_MAX_PPW ? MF_MAX_PPW : MF_COLS / CG) : 1;
    constexpr int NB = CG ? (PPW_C * VPP) / MF_THREADS : 1;

    float ra0 [MT * 4];
    float rb0 [NB * VEC];

    auto fetch = [&] (int chunk, float (&ra) [MT * 4], float (&rb) [NB * VEC]) {
        const int k0 = chunk * MF_KC;
#pragma unroll
        for (int m = 0; m < MT; ++m)                        // past ktot / past the last row: out of range => 0
            VecLoad<4>::load (&ra [m * 4], rs_eff, a_off0 + m * a_tile_stride + (unsigned int) k0 * 4u);
        if (CG) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int v = pt + u * MF_THREADS;
                const int jl = v / VPP, rem = v % VPP, kk = rem / VPF, cv = rem % VPF;
                const int lin = w0 + jl * g.Q + k0 + kk;
                // Frames below H belong to the history buffer: their offset into `in` is forced far out of range by a
                // select, not left to wrap as a negative number — the compiler may split an offset into register + immediate,
                // and the hardware's range check does not wrap that sum to 32 bits (a "negative" base plus a positive
                // immediate would be rejected although the true offset is valid).
                const unsigned int oi = lin >= a.H ? (unsigned int)((lin - a.H) * CG + cv * VEC) * 4u : 0xfffffff0u;
                VecLoad<VEC>::load (&rb [u * VEC], rs_in, oi);
                if (touches_hist) {
                    float hv [VEC];
                    VecLoad<VEC>::load (hv, rs_hist, (lin >= 0 && lin < a.H) ? (unsigned int)(lin * CG + cv * VEC) * 4u : 0xfffffff0u);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) rb [u * VEC + e] = __uint_as_float (__float_as_uint (rb [u * VEC + e]) | __float_as_uint (hv [e]));
                }
            }
        }
    };

    auto commit = [&] (int chunk, int buf, float (&ra) [MT * 4], float (&rb) [NB * VEC]) {
        float *As = As_ [buf], *Bs = Bs_ [buf];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 v;
            v [0] = ra [m * 4]; v [1] = ra [m * 4 + 1]; v [2] = ra [m * 4 + 2]; v [3] = ra [m * 4 + 3];
            *reinterpret_cast<f32x4 *> (&As [(m * 32 + a_row) * MF_LD + a_kseg]) = v;
        }

        if (CG) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int vi = pt + u * MF_THREADS;
                const int jl = vi / VPP, rem = vi % VPP, kk = rem / VPF, cv = rem % VPF;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    Bs [(jl * CG + cv * VEC + e) * MF_LD + kk] = rb [u * VEC + e];
            }
        }
        else {
            // generic channel count: plain loop, four loads in flight at a time
            const int k0 = chunk * MF_KC, per = MF_KC * cg, total = ppw * per;
            for (int e0 = tid; e0 < total; e0 += 4 * MF_THREADS) {
                float tmp [4]; int dst [4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * MF_THREADS;
                    const int jl = e / per, rem = e - jl * per, kk = rem / cg, c = rem - kk * cg;
                    dst [u] = e < total ? (jl * cg + c) * MF_LD + kk : -1;
                    tmp [u] = e < total ? load_frame (a, INT_MIN, w0 + jl * g.Q + k0 + kk, ch_base + c) : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (dst [u] >= 0) Bs [dst [u]] = tmp [u];
            }
        }
    };

    const int ncols = ppw * cg;
    if (ncols < MF_COLS)                                     // unused columns stay zero for the whole kernel
        for (int e = tid; e < (MF_COLS - ncols) * MF_LD; e += THREADS)
            for (int b = 0; b < NBUF; ++b) Bs_ [b] [ncols * MF_LD + e] = 0.0f;

    double sum [MT] [16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum [m] [r] = 0.0;
    const bool second_tile = MT > 1 && rows_valid > 32;       // (uniform) the last super-tile of a period may be half empty

    const int nchunks = g.ktot / MF_KC;
    const int arow = (lane & 31) * MF_LD + 4 * (lane >> 5);
    const int brow = (wave * 32 + (lane & 31)) * MF_LD + 4 * (lane >> 5);

    // one chunk of matrix work on LDS buffer `buf`: 32 k's = 4 groups of 8; lanes 0-31 take k 0-3 of a
    // group, lanes 32-63 k 4-7.  NT = m-tiles actually computed (1 or MT).
    auto matrix_rows = [&] (auto nt_tag, int chunk, int buf) {
        constexpr int NT = decltype (nt_tag)::value;
        const float *As = As_ [buf], *Bs = Bs_ [buf];
        const int k0 = chunk * MF_KC;
        const bool band = k0 < g.band_hi && k0 + MF_KC > g.band_lo;
        if (!band) {
            f32x16 acc [NT];
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc [m] [r] = 0.0f;
#pragma unroll
            for (int grp = 0; grp < MF_KC / 8; ++grp) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *> (&Bs [brow + grp * 8]);
                f32x4 av [NT];
#pragma unroll
                for (int m = 0; m < NT; ++m) av [m] = *reinterpret_cast<const f32x4 *> (&As [m * 32 * MF_LD + arow + grp * 8]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int m = 0; m < NT; ++m) {
                        acc [m] = __builtin_amdgcn_mfma_f32_32x32x2f32 (av [m] [q], bv [q], acc [m], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int m = 0; m < NT; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sum [m] [r] = sum [m] [r] + (double) acc [m] [r];
            }
        }
        else {
#pragma unroll
            for (int grp = 0; grp < MF_KC / 8; ++grp) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *> (&Bs [brow + grp * 8]);
                f32x4 av [NT];
#pragma unroll
                for (int m = 0; m < NT; ++m) av [m] = *reinterpret_cast<const f32x4 *> (&As [m * 32 * MF_LD + arow + grp * 8]);
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    f32x16 acc [NT];
#pragma unroll
                    for (int m = 0; m < NT; ++m) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc [m] [r] = 0.0f;
                        acc [m] = __builtin_amdgcn_mfma_f32_32x32x2f32 (av [m] [q], bv [q], acc [m], 0, 0, 0);
                        acc [m] = __builtin_amdgcn_mfma_f32_32x32x2f32 (av [m] [q + 1], bv [q + 1], acc [m], 0, 0, 0);
                    }
#pragma unroll
                    for (int m = 0; m < NT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum [m] [r] = sum [m] [r] + (double) acc [m] [r];
                }
            }
        }
    };
    auto matrix_chunk = [&] (int chunk, int buf) {
        if (second_tile) matrix_rows (std::integral_constant<int, MT> {}, chunk, buf);
        else matrix_rows (std::integral_constant<int, 1> {}, chunk, buf);
    };

    if (WS) {
        // two separate loops (disjoint live ranges => registers = max of the two roles, not the sum);
        // both execute exactly nchunks + 1 barriers
        if (loader) {
            if constexpr (CG != 0 && MT == 1) {
                // No vector arithmetic per chunk (measured +10 %: vector-unit instructions of ANY wave on a SIMD take issue
                // slots from its matrix pipe): every thread's offsets and LDS addresses are fixed, the chunk moves the
                // resource BASES (scalar unit; the range check moves with them, so past-the-end reads stay 0), and the
                // loop is unrolled by the two LDS buffers so their addresses are immediates.
                constexpr unsigned int A_STEP = MF_KC * 4u, B_STEP = MF_KC * CG * 4u;
                // tiles that reach into the history read the call's contiguous head instead of `in` (same loop, other base)
                const int origin = touches_hist ? -g.head_pad : a.H;                  // linear index of the base's first frame
                const unsigned int a_bytes = (unsigned int)((size_t) ROWS * g.ktot * 4);
                const unsigned int b_bytes = touches_hist ? (unsigned int)((size_t) g.head_frames * a.C * 4) : (unsigned int)((size_t) a.in_frames * a.C * 4);
                const char *a_base = reinterpret_cast<const char *> (g.eff + (size_t) st * ROWS * g.ktot);
                const char *b_base = touches_hist ? reinterpret_cast<const char *> (g.head) : reinterpret_cast<const char *> (a.in);
                unsigned int boff [NB]; int bdst [NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int v = pt + u * MF_THREADS;
                    const int jl = v / VPP, rem = v % VPP, kk = rem / VPF, cv = rem % VPF;
                    boff [u] = (unsigned int)(max (w0 + jl * g.Q + kk - origin, 0) * CG + cv * VEC) * 4u;
                    bdst [u] = (jl * CG + cv * VEC) * MF_LD + kk;
                }
                const int adst = a_row * MF_LD + a_kseg;
                auto lean_fetch = [&] (int chunk) {
                    const unsigned int sa = min ((unsigned int) chunk * A_STEP, a_bytes), sb = min ((unsigned int) chunk * B_STEP, b_bytes);
                    const __amdgpu_buffer_rsrc_t ra_ = make_rsrc (a_base + sa, a_bytes - sa), rb_ = make_rsrc (b_base + sb, b_bytes - sb);
                    VecLoad<4>::load (ra0, ra_, a_off0);
#pragma unroll
                    for (int u = 0; u < NB; ++u) VecLoad<VEC>::load (&rb0 [u * VEC], rb_, boff [u]);
                };
                auto lean_commit = [&] (auto buf_tag) {
                    constexpr int BUF = decltype (buf_tag)::value % NBUF;      // (NBUF = 1 only in the instantiations that never get here)
                    f32x4 v; v [0] = ra0 [0]; v [1] = ra0 [1]; v [2] = ra0 [2]; v [3] = ra0 [3];
                    *reinterpret_cast<f32x4 *> (&As_ [BUF] [adst]) = v;
#pragma unroll
                    for (int u = 0; u < NB; ++u)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) Bs_ [BUF] [bdst [u] + e * MF_LD] = rb0 [u * VEC + e];
                };
                lean_fetch (0); lean_commit (std::integral_constant<int, 0> {}); lean_fetch (1);
                __syncthreads ();
                for (int chunk = 0; chunk < nchunks; chunk += 2) {
                    lean_commit (std::integral_constant<int, 1> {}); lean_fetch (chunk + 2);
                    __syncthreads ();
                    if (chunk + 1 < nchunks) {
                        lean_commit (std::integral_constant<int, 0> {}); lean_fetch (chunk + 3);
                        __syncthreads ();
                    }
                }
                return;
            }
            else {
                // (the two-m-tile experiment, kernel preference 4: the general staging code, one register stage)
                fetch (0, ra0, rb0); commit (0, 0, ra0, rb0); fetch (1, ra0, rb0);
                __syncthreads ();
                for (int chunk = 0; chunk < nchunks; ++chunk) {
                    commit (chunk + 1, (chunk & 1) ^ 1, ra0, rb0);       // past-the-end chunks: loads return 0 / LDS unread
                    fetch (chunk + 2, ra0, rb0);
                    __syncthreads ();
                }
                return;
            }
        }
        __syncthreads ();
        if constexpr (MT == 1 && WS) mf_k_walk<0> (As_, Bs_, arow, brow, nchunks, g.band_lo, g.band_hi, sum [0]);
        else
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            matrix_chunk (chunk, chunk & 1);
            __syncthreads ();
        }
    }
    else {
        fetch (0, ra0, rb0);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            __syncthreads ();                                // previous chunk fully consumed
            commit (chunk, 0, ra0, rb0);
            __syncthreads ();
            if (chunk + 1 < nchunks) fetch (chunk + 1, ra0, rb0);      // global loads fly while the matrix cores work
            matrix_chunk (chunk, 0);
        }
    }

    // ---- epilogue: C/D layout of 32x32: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
    const int col = wave * 32 + (lane & 31);
    if (col >= ncols) return;
    const int jl = col / cg, c = col - jl * cg;
    if (ch_base + c >= a.C) return;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m && !second_tile) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const unsigned char status = s_status [jl * ROWS + i];
            if (status == 0 || status == 3) {
                const size_t n = (size_t) n_tile + (size_t) jl * g.P + i;
                float y = (float) sum [m] [r];
                if (!INTERP && status == 3)
                    y = load_frame (a, INT_MIN, s_ip [i] + jl * g.Q + s_fi [i] / a.F, ch_base + c);
                a.out [n * a.C + ch_base + c] = y;
            }
            else if (status == 1) {                            // off the canonical pattern: evaluated here at its exact position
                const size_t n = (size_t) n_tile + (size_t) jl * g.P + i;
                a.out [n * a.C + ch_base + c] = direct_sample<INTERP> (a, INT_MIN, locate<INTERP> (a, segs, (unsigned int) n), ch_base + c);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Streaming form of the matrix-core kernel: PERSISTENT workgroups, for REGULAR launches.
//
// fir_mfma_kernel spends 14 of its 59 us per tile outside the K loop — the per-output position replay in front of it, the
// branchy store loop behind it, the pipeline filling up again for every tile — while three workgroups per CU cover only part
// of that for each other.  Here a workgroup takes a strided list of tiles of its XCD (same XCD-contiguous order as the
// one-tile-per-workgroup grid) and the two roles never stop between tiles:
//   * the staging waves run ONE chunk stream across all of the workgroup's tiles (chunk c+2 of the flattened sequence is
//     in flight, c+1 is being committed, while the matrix waves consume c): the first chunks of the next tile are staged
//     while the last ones of the current tile are multiplied;
//   * the matrix waves finish a tile with 16 conversions and 16 buffer stores (addresses: one per-lane offset + immediates,
//     tile base and range check in the buffer resource: out-of-range rows are dropped by the hardware) and go straight on.
// What makes the position replay unnecessary is decided on the HOST, per launch (mfma_launch_is_regular): every output's
// position differs from its slot's canonical lattice position by less than MF_PHASE_TOL filter steps — a bound on the
// reference's own fp64 arithmetic (two roundings of n/ratio, one of the addition), so all outputs would have taken the
// status-0 path of fir_mfma_kernel — and, in nearest-filter mode, no slot sits close enough to a half step for a period to
// round to another filter.  Launches that fail the test (calls beyond a few million frames, drifting ratios do not get here
// at all) run fir_mfma_kernel.  Same tiles, same K order, same flush schedule: the two kernels produce identical bits.
// ---------------------------------------------------------------------------------------------------
// PASS: nearest-filter mode without a low-pass — outputs that fall exactly on an input sample are copied through (its own
// instantiation: the extra loads of that epilogue cost the common ones a spilled register otherwise).
template <bool INTERP, int CG, bool PASS>
__global__ __launch_bounds__ (2 * MF_THREADS) __attribute__ ((amdgpu_waves_per_eu (6)))
void fir_mfma_stream_kernel (ArtFirArgs a, MfmaGeom g, int wgs_per_xcd)
{
    constexpr int THREADS = 2 * MF_THREADS;
    constexpr int PPW = MF_COLS / CG > MF_MAX_PPW ? MF_MAX_PPW : MF_COLS / CG;
    constexpr int NCOLS = PPW * CG;
    __shared__ __attribute__ ((aligned (16))) float As_ [2] [32 * MF_LD];
    __shared__ __attribute__ ((aligned (16))) float Bs_ [2] [MF_COLS * MF_LD];

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

    // (the tile loop, shared as text with the fixed-point kernel's stand-by)
#include "fir_matrix_stream_body.inc"
}

// ---------------------------------------------------------------------------------------------------
// The streaming kernel for launches of FEW tiles (calls of some ten thousand frames: fewer tiles than the chip has CUs).  One
// workgroup per tile walks the whole K range at the pace ONE CU draws rows and samples through a cold L2 (~20 us for 32 chunks
// whatever the call's size: profiles/r3_small_launch_experiment.txt).  Here a tile's K range is cut into KS parts, each a work item
// of its own (the same staging, the same walk over its chunks, the same flush schedule inside them): KS times the workgroups,
// each fetching a KS-th.  A part leaves its 32 x 128 fp64 partial sums in device memory; the part that finishes LAST — per matrix
// wave: a wave's 32 columns are its own, so the four waves of a workgroup never wait for each other — adds the KS partials in
// the order of the parts and writes the outputs: the result does not depend on which part came last.  All parts of a tile are
// neighbours in one XCD's work list (one L2); partials and counters still go through agent-scope accesses (write-through
// stores, L1-bypassing loads), so nothing rests on where a workgroup runs.
// Not the bits of the unsplit kernel in general (the same flushed values, added in another association: a float in ~2^29 may
// round the other way): the library's own choice only (kernel preference 0 / 2; 8 forces it), decided from the stream's size, never
// with preference 5 / 6, whose bit-for-bit agreement the tests pin.
// ---------------------------------------------------------------------------------------------------
template <bool INTERP, int CG, bool PASS>
__global__ __launch_bounds__ (2 * MF_THREADS) __attribute__ ((amdgpu_waves_per_eu (6)))
void fir_mfma_split_kernel (ArtFirArgs a, MfmaGeom g, int wgs_per_xcd, int KS, double *partials, unsigned int *arrivals)
{
    constexpr int THREADS = 2 * MF_THREADS;
    constexpr int PPW = MF_COLS / CG > MF_MAX_PPW ? MF_MAX_PPW : MF_COLS / CG;
    constexpr int NCOLS = PPW * CG;
    __shared__ __attribute__ ((aligned (16))) float As_ [2] [32 * MF_LD];
    __shared__ __attribute__ ((aligned (16))) float Bs_ [2] [MF_COLS * MF_LD];

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

    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;
    const int tiles_per_xcd = g.groups_per_xcd * g.slot_tiles;
    const int items_per_xcd = tiles_per_xcd * KS;
    const int nchunks = g.ktot / MF_KC;

    // work item `within` of this XCD's list -> (slot tile, period group, part); false past the last valid tile (validity is monotone)
    auto item_at = [&] (int within, int &st, int &jg, int &ks) -> bool {
        if (within >= items_per_xcd) return false;
        const int tile = within / KS;
        ks = within - tile * KS;
        st = tile % g.slot_tiles; jg = xcd * g.groups_per_xcd + tile / g.slot_tiles;
        if (jg >= g.period_groups) return false;
        return a.n_begin + (unsigned int)(jg * PPW) * g.P + (unsigned int)(st * 32) < a.n_end;
    };
    auto chunks_of = [&] (int ks, int &c0, int &c1) { c0 = ks * nchunks / KS; c1 = (ks + 1) * nchunks / KS; };

    if (NCOLS < MF_COLS)                                      // unused columns stay zero for the whole kernel
        for (int e = tid; e < (MF_COLS - NCOLS) * MF_LD; e += THREADS)
            for (int b = 0; b < 2; ++b) Bs_ [b] [NCOLS * MF_LD + e] = 0.0f;

    // chunks this workgroup will consume in total (both roles count the same way)
    int total = 0;
    { int st, jg, ks, c0, c1; for (int w = rank; item_at (w, st, jg, ks); w += wgs_per_xcd) { chunks_of (ks, c0, c1); total += c1 - c0; } }
    if (total == 0) return;

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

        // the fetch stream: item being fetched, its bases, the chunk to fetch next and the item's last (all uniform)
        int f_within = rank, f_chunk = 0, f_end = 0;
        bool f_live = false;
        const char *fa_base = nullptr, *fb_base = nullptr;
        unsigned int fa_bytes = 0, fb_bytes = 0;
        auto open_item = [&] () {
            int st, jg, ks;
            f_live = item_at (f_within, st, jg, ks);
            if (!f_live) return;
            chunks_of (ks, f_chunk, f_end);
            const int w0 = g.tile_w0 [3 * st] + g.w_shift + jg * PPW * g.Q;
            const bool touches_hist = w0 < a.H;              // (first period group of a call: staged from the gathered head)
            const int origin = touches_hist ? -g.head_pad : a.H;
            const char *base = touches_hist ? reinterpret_cast<const char *> (g.head) : reinterpret_cast<const char *> (a.in);
            const size_t total_b = touches_hist ? (size_t) g.head_frames * a.C * 4 : (size_t) a.in_frames * a.C * 4;
            size_t skip = (size_t) max (w0 - origin, 0) * CG * 4;
            if (skip > total_b) skip = total_b;
            fb_base = base + skip; fb_bytes = (unsigned int)(total_b - skip);
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
                if (++f_chunk == f_end) { f_within += wgs_per_xcd; open_item (); }
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

        open_item ();
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
    const int arow = (lane & 31) * MF_LD + 4 * (lane >> 5);
    const int col = wave * 32 + (lane & 31);
    const bool col_live = col < NCOLS;
    const int jl = col / CG, c = col - jl * CG;
    const int brow = col * MF_LD + 4 * (lane >> 5);
    const unsigned int out_off = (unsigned int)((jl * g.P + 4 * (lane >> 5)) * CG + c) * 4u;

    double sum [16];

    __syncthreads ();                                        // the staging waves have committed chunk 0
    int consumed = 0;                                        // chunks walked so far: the stream's chunk s sits in LDS buffer s & 1
    for (int within = rank; ; within += wgs_per_xcd) {
        int st, jg, ks, c0, c1;
        if (!item_at (within, st, jg, ks)) break;
        chunks_of (ks, c0, c1);
#pragma unroll
        for (int r = 0; r < 16; ++r) sum [r] = 0.0;

        if ((consumed ^ c0) & 1) mf_k_walk<1> (As_, Bs_, arow, brow, nchunks, g.band_lo, g.band_hi, sum, c0, c1);
        else mf_k_walk<0> (As_, Bs_, arow, brow, nchunks, g.band_lo, g.band_hi, sum, c0, c1);
        consumed += c1 - c0;

        // ---- this part's sums -> device memory; the wave that brings the tile's count to KS adds the parts up.  A part's image:
        // [register pair][thread][2 doubles] — 16 bytes per lane and instruction, a wave's lanes side by side; stores written
        // through and loads past the L1 (cache policy sc0 sc1: coherent wherever the other parts ran)
        constexpr int COHERENT = 1 | 16;                     // (aux bits of the raw buffer instructions on gfx940+: sc0, sc1)
        const int tile_g = xcd * tiles_per_xcd + within / KS;
        {
        const size_t part_bytes = (size_t) 16 * MF_THREADS * sizeof (double);
        {
            const __amdgpu_buffer_rsrc_t rs_part = make_rsrc (reinterpret_cast<char *> (partials) + (size_t)(tile_g * KS + ks) * part_bytes, (unsigned int) part_bytes);
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                u32x4 v;
                const unsigned long long lo = (unsigned long long) __double_as_longlong (sum [2 * r2]), hi = (unsigned long long) __double_as_longlong (sum [2 * r2 + 1]);
                v.x = (unsigned int) lo; v.y = (unsigned int)(lo >> 32); v.z = (unsigned int) hi; v.w = (unsigned int)(hi >> 32);
                __builtin_amdgcn_raw_buffer_store_b128 (v, rs_part, (r2 * MF_THREADS + pt) * 16, 0, COHERENT);
            }
        }
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");       // (written through: the count below is only seen behind them)
        unsigned int before = 0u;
        if (lane == 0) before = __hip_atomic_fetch_add (arrivals + tile_g * 4 + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        before = (unsigned int) __builtin_amdgcn_readfirstlane ((int) before);
        if (before != (unsigned int)(KS - 1)) continue;
        if (lane == 0) __hip_atomic_store (arrivals + tile_g * 4 + wave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (for the next launch)

        {
        // every part's sums, in the order of the parts (all loads of a register pair in flight together)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum [r] = 0.0;
        for (int k0 = 0; k0 < KS; k0 += 2) {                  // (two parts at a time, 16 loads in flight; an odd count's last pair: the second reads an empty resource — zeros)
            u32x4 v [2] [8];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const __amdgpu_buffer_rsrc_t rs_part = make_rsrc (reinterpret_cast<char *> (partials) + (size_t)(tile_g * KS + k0 + kk) * part_bytes, k0 + kk < KS ? (unsigned int) part_bytes : 0u);
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) v [kk] [r2] = __builtin_amdgcn_raw_buffer_load_b128 (rs_part, (r2 * MF_THREADS + pt) * 16, 0, COHERENT);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    sum [2 * r2] = sum [2 * r2] + __longlong_as_double ((long long)((unsigned long long) v [kk] [r2].x | (unsigned long long) v [kk] [r2].y << 32));
                    sum [2 * r2 + 1] = sum [2 * r2 + 1] + __longlong_as_double ((long long)((unsigned long long) v [kk] [r2].z | (unsigned long long) v [kk] [r2].w << 32));
                }
        }
        }
        }
        const unsigned int n_tile = a.n_begin + (unsigned int)(jg * PPW) * g.P + (unsigned int)(st * 32);
        const int rows_valid = min (32, g.P - st * 32);
        const size_t left = (size_t)(a.n_end - n_tile) * CG * 4;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc (a.out + (size_t) n_tile * CG, left > 0xffffff00ull ? 0xffffff00u : (unsigned int) left);
        const unsigned int pass_rows = PASS ? (unsigned int) g.tile_w0 [3 * st + 1] : 0u;
        const int lo = a.n_skip != 0 && jg == 0 && wave == 0 ? a.n_skip - st * 32 : 0;      // (a launch on rows kept across calls starts mid-period: fir_i8_stream_kernel's epilogue)
        // (the lane's half, opaque and per item: as a loop invariant the sixteen slot numbers below were computed in front of the item loop, spilled — this kernel has
        // no register to spare — and read back from scratch one by one, a round trip per output register: ~3 us a tile)
        int half = lane >> 5;
        asm volatile ("" : "+v" (half));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i_const = (r & 3) + 8 * (r >> 2);      // compile-time part of the slot
            float y = (float) sum [r];
            const int i = i_const + 4 * half;
            if constexpr (PASS) {
                if ((pass_rows >> i) & 1u)
                    y = load_frame (a, INT_MIN, g.canon_ip [st * 32 + i] + g.w_shift + g.canon_fi [st * 32 + i] / a.F + (jg * PPW + jl) * g.Q, c);
            }
            if (col_live && i < rows_valid && (i >= lo || (lane & 31) >= CG))
                __builtin_amdgcn_raw_buffer_store_b32 (__float_as_uint (y), rs_out, (int)(out_off + (unsigned int)(i_const * CG) * 4u), 0, 0);
        }
    }
}
} // namespace

// May fir_mfma_stream_kernel run this launch (it never replays an output's position)?  Every output n of the launch sits at
// fl (base_e + fl (n / ratio)) (reference resampler.c:526, :1149); with ratio = fl (P / Q) the lattice the tiles assume is
// base_e + n Q / P, and the three roundings in between move a position by at most
//     dev = frames * 2^-52 + 2e-12   input frames   (frames = input frames from the start of the call; ring positions < 2^14)
// — canonical slots included, so two positions of one slot differ by at most 2 dev.  Interpolating: that is inside
// MF_PHASE_TOL filter steps, i.e. every output would have passed fir_mfma_kernel's own test.  Nearest-filter mode: the
// rounded filter index must be the same in every period, so no slot may sit within 2 dev of a half step.
static bool mfma_launch_is_regular (const ArtFirArgs *a, const ArtSegTable *segs)
{
    const double frames = (double) a->n_end / a->ratio + a->T + 2.0;
    const double dev_steps = 2.0 * (frames * 0x1p-52 + 2e-12) * a->F;
    if (!(dev_steps <= MF_PHASE_TOL)) return false;
    if (a->interpolate) return true;
    for (int i = 0; i < a->period_out; ++i) {
        const unsigned int n = a->n_begin + (unsigned int) i;
        if (n >= a->n_end) break;
        int e = 0;
        while (e + 1 < segs->count && segs->first [e + 1] <= n) ++e;
        const double step = n ? (double) n / a->ratio : 0.0;
        const double off = segs->base [e] + step;
        double fr = off - floor (off);
        fr = fr * (double) a->F;
        const double t = fr + 0.5, d = t - floor (t);
        if (d < dev_steps + 1e-9 || 1.0 - d < dev_steps + 1e-9) return false;
    }
    return true;
}

// does this call take the matrix-core path (arthip_fir), or the general kernel?  One rule, also asked by the batched entry
// point, which only gathers calls the general kernel would have run anyway.
static size_t pass_fixup_min ()
{
    static const size_t v = [] { const char *e = getenv ("ARTAMD_PASS_FIXUP_MIN"); return e && *e ? (size_t) strtoull (e, nullptr, 10) : (size_t) 0; } ();
    return v;                                                 // (samples of a launch from which it is done; default: every launch of the streaming kernels — measured down to 65,536 frames x 8 ch and 131,072 x 2: the spills cost more than the launch; a huge number: never — A/B runs, the bit-identity test)
}
bool artfir_pass_fixup_wanted (const ArtFirArgs *a)
{
    // (the pass's slot list holds 1024 x 32 slots of a period: a longer period — the period multiple included — keeps the kernels' own
    // PASS epilogues, decided HERE, before the instantiation is chosen, so that no launch can end up with neither)
    // (the period multiple of the launch's own geometry — matrix_geometry: 64-row tiles where the slab kernel is enabled — so that the bound here IS the
    // bound on g.P the pass sees)
    const long period = (long) a->period_out * (a->period_out > 0 ? artfir_period_multiple (a->period_out, artfir_i8_slab_enabled () ? 64 : 32) : 1);
    return !a->interpolate && !a->lowpass && a->out_pitch == 0 && a->in_pitch == 0 && period > 0 && period <= 8192 && (size_t)(a->n_end - a->n_begin) * a->C >= pass_fixup_min ();
}
int artfir_pass_fixup (const ArtFirArgs *a, const MfmaGeom &g, hipStream_t st)
{
    // (cannot be — artfir_pass_fixup_wanted bounds the same period — but a library does not abort its host: the launch fails, counted by the caller)
    if (g.P > 1024 * 32) { fprintf (stderr, "artamd: pass-through pass: a period of %d slots\n", g.P); return -1; }
    const unsigned int total = a->n_end - a->n_begin;
    const unsigned long long items = (unsigned long long)((total + g.P - 1) / g.P) * a->C;       // (per flagged slot)
    unsigned int blocks = (unsigned int)((items + 255) / 256);
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0u) blocks = 1u;
    hipLaunchKernelGGL (pass_fixup_kernel, dim3 (blocks), dim3 (256), 0, st, *a, g);
    return 0;
}

bool artfir_takes_matrix_path (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref)
{
    if (a->n_end <= a->n_begin || (a->mode & 3) == ART_MODE_STRICT) return false;
    const unsigned int total = a->n_end - a->n_begin;
    // (a shard of a multi-device context decides as its whole stream would on one device: same kernels, same bits either way)
    const int C = a->stream_C > a->C ? a->stream_C : a->C;
    bool enough;
    if (C == 1 || C == 2 || C == 4 || C == 8 || C == 16 || C == 32) {
        // (up to 256 taps the general kernel works four frames per wave: its per-frame cost roughly halves; with long filters
        // a mid-sized call is cut into 16-frame tiles that each stage ~T frames, and the more column groups the grid has the
        // fewer frames each of its workgroups shares that staging with: the per-frame cost below ~40k outputs is 1.5x at 4
        // channels, 2.3x at 8, 5x at 32; one or two channels pay 1.7x at any size — re-fitted to tools/bench_crossover2.py,
        // profiles/r4_dispatch_crossover.txt: the rule had 32 ch x 988 taps on the general kernel up to 5k frames where the
        // matrix path is ahead from 2k on, 1-2 ch x 988 taps up to 100k frames where it is ahead from 45k on)
        const double k_base = ((0.2 + 0.04 * C) + 0.00007 * C * a->T) * (a->T <= 256 ? 0.47 : a->T <= 512 ? 0.74 : 0.85) * (a->T >= 512 && C <= 2 ? 1.7 : 1.0);
        const double k_mid = k_base * (a->T >= 512 && C > 2 ? 1.5 * pow ((double) C / 4.0, 0.6) : 1.0);      // below ~40k outputs
        const double chunks = (a->T + 63) / 32;
        const double floor_ns = 13500.0 + 550.0 * chunks + (C <= 2 ? 2000.0 : 0.0);
        // (one threshold, so that a longer call never goes back to the general kernel)
        double need = (floor_ns - 5000.0) / k_mid < 40000.0 ? (floor_ns - 5000.0) / k_mid : (floor_ns - 5000.0) / k_base;
        // (round 5: filters of 24 chunks and more — 704 taps — have their mid-sized launches cut into three parts (matrix_split_parts) and their rows kept across
        // calls: the matrix path's floor fell from ~23 to ~19.3 us a call and the crossover with it — tools/micro/crossover_r5.sh, profiles/r5_crossover.txt:
        // 8 ch x 988 taps from ~7k frames (was ~11k: the 8,192-frame call 22.6 -> 19.1 us), 4 ch ~17k (was ~30k: 24,576 frames 26.0 -> 19.9), 2 ch ~27k, mono ~38k, 16 ch ~3.5k)
        if ((a->T + 64) / 32 >= 24) need *= C >= 16 ? 0.8 : 0.62;
        // (shorter filters, tools/micro/crossover_short_r5.sh, profiles/r5_crossover.txt: without the prepare launch the matrix path's floor is ~14 us at 380 taps,
        // ~12.7 at 256 — 8 ch x 380 taps from ~19k frames instead of ~27k (24,576 frames 16.6 -> 14.2 us), 32 ch from ~6k (8,192 frames 22.0 -> 15.0), 8 ch x 256 from
        // ~29k (32,768 frames 15.0 -> 12.7); 512 .. 703 taps the other way: 8 ch x 512 at 12,288 frames is the general kernel's, 13.5 against 16.9 us)
        else need *= a->T < 512 ? (C <= 2 ? 0.85 : 0.7) : 1.15;
        enough = (double) total >= need;
    }
    else {
        // (not a compiled width: the launch runs in groups of compiled widths behind two copies, fir_dispatch.hip — when it has the
        // buffer for them; the old rule stays: such streams are rare, and below it the general kernel is at home)
        // (measured with the groups in place, tools/bench_crossover2.py 6x988 6x380 3x988 12x988 64x988 24x156: the rule sits on the
        // crossover for one group; a stream of several groups pays the matrix path's floor once per group)
        enough = (double) total * C * a->T >= 1.2e8 * ((C + 31) / 32);
    }
    return a->mode == ART_MODE_FAST && a->period_out > 0 && a->fix_list && a->scratch && a->in_pitch == 0 && a->out_pitch == 0 &&
                         segs->lin_floor == INT_MIN && kernel_pref != ART_KERNEL_GENERAL &&
                         (enough || kernel_pref >= ART_KERNEL_MFMA) &&
                         // (a launch of less than a period is the general kernel's — unless the stream runs under the cut-invariant policy: anchored on the
                         // canonical period it is the same tiles as any other launch, the slots in front of its first output computed and not stored)
                         (total >= (unsigned int) a->period_out || (kernel_pref == ART_KERNEL_INVARIANT && a->rows_cache));
}

// Tile geometry of a launch (everything but the tables in device scratch); returns the compile-time channel count of the
// wave-specialised kernels, 0 for the generic instantiation.
static int matrix_geometry (const ArtFirArgs *a, MfmaGeom &g)
{
    const unsigned int total = a->n_end - a->n_begin;
    {   // (short or badly fitting periods: several at a time, fir_common.hip.h — one rule for every kernel of the path and every
        // channel count, so that all contexts of a stream have the same rows; so as to fill the 64-slot tiles of the
        // fixed-point slab kernel, which is also whole 32-slot tiles for every other kernel)
        const int mu = artfir_period_multiple (a->period_out, artfir_i8_slab_enabled () ? 64 : 32);
        g.P = mu * a->period_out; g.Q = mu * a->period_in;
    }
    // compile-time channel count where the whole stream is one column group and the buffers allow vector loads
    // (the staging waves load whole frames of one or two channels, 16-byte vectors of four channels: the input wants that alignment, not more — a stereo
    // stream handed over at an odd frame of the caller's buffer is still the specialised kernels')
    const size_t in_align = a->C >= 4 ? 16 : (size_t) a->C * 4;
    const bool small = (size_t) a->in_frames * a->C * 4 < 0xffff0000ull && ((uintptr_t) a->in % in_align) == 0 && ((uintptr_t) a->hist % 16) == 0;
    const int cgt = (small && (a->C == 1 || a->C == 2 || a->C == 4 || a->C == 8 || a->C == 16 || a->C == 32)) ? a->C : 0;
    g.tile_rows = 32;
    g.slot_tiles = (g.P + g.tile_rows - 1) / g.tile_rows;
    g.cg = a->C < 32 ? a->C : 32;
    g.ppw = MF_COLS / g.cg;
    if (g.ppw > MF_MAX_PPW) g.ppw = MF_MAX_PPW;
    // (+ 3: the fixed-point kernel starts a tile's K columns on a 4-frame block, up to 3 frames early)
    const int shift_max = (int)((g.tile_rows - 1.0) * g.Q / g.P) + 2;
    g.ktot = ((a->T + shift_max + 3 + MF_KC - 1) / MF_KC) * MF_KC;
    g.band_lo = a->T / 2 - 1 - 6;                       // central taps of the first row ...
    g.band_hi = a->T / 2 + shift_max + 6;               // ... to those of the last
    const unsigned int periods = (total + g.P - 1) / g.P;
    g.period_groups = (int)((periods + g.ppw - 1) / g.ppw);
    g.groups_per_xcd = (g.period_groups + 7) / 8;
    g.eff = nullptr; g.canon_ip = g.canon_fi = nullptr; g.canon_frac = nullptr; g.head = nullptr; g.head_frames = 0; g.tile_w0 = nullptr; g.w_shift = 0;
    // (a launch anchored on the canonical period starts its first period's tiles at slot 0: up to Q frames in front of the first output, whose own window
    // starts about T/2 frames into the history — ADVICE r5: with 64 zero frames, streams whose period_in exceeds T/2 + 64 fell back to rows of their own)
    g.head_pad = MF_HEAD_PAD + (g.Q > a->T / 2 ? ((g.Q - a->T / 2 + 3) & ~3) : 0);
    return cgt;
}

// Parts a tile's K range is cut into for this launch (fir_mfma_split_kernel), 1: not split.  Decided from the STREAM's size (a shard
// of a multi-device context as its whole stream would: the same parts, the same bits): tiles = slot tiles x period groups of 128
// columns.  Measured (profiles/r3_split_k_experiment.txt): cutting K does not cut a part's sample fetch (a tile's 16 periods are
// 147 frames apart: a quarter of the K range still spans 3/4 of the tile's input), so more than two parts lose; two parts win
// where half the CUs would otherwise idle through a whole K walk (50-110 tiles: the 32,768-frame call of 8 ch x 988 taps,
// 21.7 -> 16.2 us) and nowhere else.  Kernel preference 8 forces 2 / 4 / 8 parts (tests, experiments: ARTAMD_SPLIT_KS).
// a launch of a channel count the kernels are not compiled for runs in groups of a compiled width (fir_dispatch.hip, fir_in_groups): what
// its buffers are sized for is the widest group
static ArtFirArgs widest_group (const ArtFirArgs *a)
{
    ArtFirArgs b = *a;
    if (a->C > 32 || (a->C & (a->C - 1))) {
        int wp = 1; while (wp < (a->C > 32 ? 32 : a->C)) wp <<= 1;
        b.stream_C = a->stream_C > a->C ? a->stream_C : a->C;
        b.C = wp; b.in = nullptr; b.hist = nullptr;        // (the groups' own buffers are 256-byte aligned)
    }
    return b;
}

static int matrix_split_parts (const ArtFirArgs *a, const MfmaGeom &g, unsigned int outputs, int kernel_pref)
{
    if (kernel_pref == 5 || ART_PREF_PINS_F32 (kernel_pref) || kernel_pref == 7) return 1;
    static const bool off = [] { const char *e = getenv ("ARTAMD_NO_SPLIT"); return e && *e && *e != '0'; } ();
    if (off) return 1;
    const int C = a->stream_C > a->C ? a->stream_C : a->C;
    const double periods = ceil ((double) outputs / g.P);
    const double cols = C >= 2 ? 128.0 : 64.0;                                   // (mono: 64 periods per tile, half the columns idle)
    const double groups = ceil (periods * C / cols), tiles = g.slot_tiles * groups;
    const int nchunks = g.ktot / MF_KC;
    // As many parts as keep an XCD's items within ONE round of its 32 CUs (a tile walked alone is a chain of ~0.6 us a chunk whatever else the chip does;
    // a second round of items costs more than the parts save), at most four, four chunks a part at least.  Re-fitted in round 5 (tools/micro/split_sweep.sh,
    // profiles/r5_split_rule.txt): the rule had been "two parts from 50 to 110 tiles" — which left 8 ch x 988 taps at 12,288 - 16,384 frames (ART's block) and
    // 32 ch at 4,096 unsplit (22.9 / 25.4 us a call where three parts take 19.2 / 19.9) and split 40,960 - 49,152 frames (9 - 11 period groups: two per XCD,
    // 40 items on 32 CUs) into two rounds (29.5 - 30.1 us where the uncut launch takes 23.6).
    // Long filters only: at 14 chunks a tile (380 taps) three parts LOSE 12 % (16.1 against 14.3 us), at 18 (512 taps) 3 %; at 33 (988 taps) they win 16 - 26 %.
    int ks = 1;
    if (nchunks >= 24) {
        const int gpx = (int) ceil (groups / 8.0);
        int k = 32 / (gpx * g.slot_tiles);
        if (k > 4) k = 4;
        if (k >= 2) ks = k;
    }
    if (kernel_pref == 8) {                                   // (forced: the library's own parts where it splits, else as many as fill the chip)
        if (ks == 1) ks = tiles * 8 <= 768 ? 8 : tiles * 4 <= 768 ? 4 : 2;
        static const int k_env = [] { const char *e = getenv ("ARTAMD_SPLIT_KS"); return e && *e ? atoi (e) : 0; } ();
        if (k_env > 0) ks = k_env;
    }
    {   static const int force = [] { const char *e = getenv ("ARTAMD_SPLIT_FORCE_KS"); return e && *e ? atoi (e) : 0; } ();      // (A/B runs)
        if (force > 0 && kernel_pref != 8) ks = force;         // (clamped below like any other count: four chunks a part at least)
    }
    while (ks > 1 && nchunks / ks < 4) --ks;
    return ks;
}

size_t artfir_split_bytes (const ArtFirArgs *a_, unsigned int outputs, int kernel_pref)
{
    const ArtFirArgs wg_ = widest_group (a_), *a = &wg_;
    if (!a->period_out || a->mode != ART_MODE_FAST) return 0;
    ArtFirArgs b = *a;
    b.n_begin = 0; b.n_end = outputs + (unsigned int) a->period_out * 64u;       // (any launch of the call: at most this many outputs)
    MfmaGeom g;
    if (!matrix_geometry (&b, g)) return 0;
    const int ks = matrix_split_parts (a, g, outputs, kernel_pref);
    const size_t tiles = (size_t) 8 * g.groups_per_xcd * g.slot_tiles;
    if (ks < 2 || tiles * 16 > ART_SPLIT_HEAD_BYTES) return 0;
    return ART_SPLIT_HEAD_BYTES + tiles * ks * 16 * MF_THREADS * sizeof (double);
}

// see arthip_fir_spans_segments (art_internal.h): the launch would run on a streaming kernel (the conditions of `regular` below)
bool artfir_matrix_spans_segments (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref)
{
    if (kernel_pref == 5 || !artfir_takes_matrix_path (a, segs, kernel_pref)) return false;
    MfmaGeom g;
    return matrix_geometry (a, g) != 0 && (size_t) a->n_end * a->C * 4 < 0xffff0000ull && mfma_launch_is_regular (a, segs);
}

// bytes of digit planes the fixed-point kernel wants for a call of this shape (the host sizes a->planes with it before the launch)
size_t artfir_planes_bytes (const ArtFirArgs *a_, unsigned int outputs, int kernel_pref)
{
    const ArtFirArgs wg_ = widest_group (a_), *a = &wg_;
    if (!a->period_out || a->mode != ART_MODE_FAST || kernel_pref == 5 || ART_PREF_PINS_F32 (kernel_pref) || kernel_pref == 8) return 0;
    // Where it pays (MI355X, tools/bench_shapes.py with and without ARTAMD_NO_FIXED, profiles/r2_fixed_point_shapes.txt): the
    // integer kernel gains in proportion to outputs x channels x taps, its staging pass costs in proportion to the input
    // and its extra launch ~4 us: long filters and big calls win (8 ch x 988 taps: from ~90k frames per call, +27 % at 1M;
    // 4 and 32 channels alike), 380-tap and shorter filters lose at every size (13 chunks per tile: the f32 kernel is not
    // matrix-bound there).  kernel_pref 7 takes the fixed-point kernel wherever it can run.
    // (one- and two-channel streams: the register-staged integer kernel plus its staging passes lose to the f32 kernel at every size —
    // 2 ch x 988 taps, 1M frames: 28.4 against 30.9 Gsamples/s, mono 16.8 / 17.5; re-measured at the end of round 4)
    // (medium filters, 256-511 taps: only calls of ~1M frames of an 8-channel stream win — 8 ch x 380 taps 101.7 -> 109.0 Gsamples/s, 8 ch x 448
    // 86.2 -> 98.3, 8 ch x 256 +2 %; 16 ch x 380 loses 4 %, 32 ch x 256 9 %, 4 ch x 380 7 %, and 8 ch x 380 at 524k frames 4 %)
    {
        const int Cs = a->stream_C > a->C ? a->stream_C : a->C;
        const bool long_rows = a->T >= 512 && (double) outputs * Cs * a->T >= 8.5e8;
        const bool medium_rows = a->T >= 256 && a->T < 512 && Cs >= 8 && Cs < 16 && outputs >= 900000u;
        if (kernel_pref != 7 && (Cs <= 2 || !(long_rows || medium_rows))) return 0;
    }
    static const bool off = [] { const char *e = getenv ("ARTAMD_NO_FIXED"); return e && *e && *e != '0'; } ();
    if (off) return 0;
    MfmaGeom g;
    const int cgt = matrix_geometry (a, g);
    if (cgt && kernel_pref != 7) {
        // Mid-sized calls (round 5, tools/micro/fixed_crossover_r5.sh, profiles/r5_fixed_crossover.txt): the f32 streaming kernel's time is a staircase — a round of 32
        // tiles per XCD costs the same however full it is — and where its rounds are well filled it beats the fixed-point path, whose time grows smoothly with the
        // samples: 8 ch x 988 taps at 196,608 frames 41.9 against 47.7 us a call, 16 ch at 81,920 - 98,304 40.5 against 49.5, 32 ch at 49,152 44.3 against 54.3,
        // 8 ch x 512 taps at 196,608 28.5 against 40.1.  Two fitted models (us a call; the stream's size, so that a shard decides as its stream would), the f32 kernel
        // taken where it is ahead by 8 % and more; from ~3 M samples a call on the fixed-point path is ahead everywhere.
        const int Cs = a_->stream_C > a_->C ? a_->stream_C : a_->C;      // (the stream's own channel count: not a group's padded width)
        const double periods = ceil ((double) outputs / g.P), groups = ceil (periods * Cs / 128.0);
        const double rounds = ceil (ceil (groups / 8.0) * g.slot_tiles / 32.0), nchunks = g.ktot / MF_KC;
        const double t_f32 = 7.5 + rounds * (0.45 * nchunks + 1.8);
        const double ks = (double) outputs * a->period_in / a->period_out * Cs * 1e-3;       // thousands of input samples of the call
        const double t_fixed = (20.0 + 0.0145 * ks) * (0.68 + 0.32 * a->T / 988.0);
        static const bool model_off = [] { const char *e = getenv ("ARTAMD_FIXED_MODEL"); return e && *e == '0'; } ();
        // (streams of a compiled width only: others run as several group launches, which the f32 model does not describe; the stream's width, so that its shards agree)
        const bool one_launch = Cs == 4 || Cs == 8 || Cs == 16 || Cs == 32;
        // (fitted, and used, up to 3 M samples a call: beyond, the slab kernel's slope is lower than this line's and the product rule above stands — a first
        // version without the bound sent 8 ch x 380 taps and 16 ch x 512 taps at 1M frames to the f32 kernel: 81.6 against 75.7 and 197 against 178 us)
        if (!model_off && one_launch && ks <= 3000.0 && t_f32 < 0.92 * t_fixed) return 0;
    }
    return cgt ? artfir_i8_bytes (a, g, cgt, outputs) : 0;
}

// device bytes of the rows the fixed-point kernel keeps across the calls of a context (0: this call is not for that kernel)
size_t artfir_rows_bytes (const ArtFirArgs *a_, unsigned int outputs, int kernel_pref)
{
    if (!artfir_rows_cache_enabled ()) return 0;
    const ArtFirArgs wg_ = widest_group (a_), *a = &wg_;
    if (!a->period_out || a->mode != ART_MODE_FAST) return 0;
    MfmaGeom g;
    const int cgt = matrix_geometry (a, g);
    if (!cgt) return 0;
    // the f32 streaming kernel's set at the head, the fixed-point kernel's sets (where the call is for that kernel) behind it
    return artfir_f32_set_bytes (g) + (artfir_planes_bytes (a_, outputs, kernel_pref) ? artfir_i8_rows_bytes (a, g, cgt, outputs) : 0);
}

// Upkeep of the stream's canonical period (fir_matrix_i8.hip, "The rows across calls") by EVERY launch of a rational-ratio stream, whichever kernel
// runs it: which launch founds the period, and which one re-founds it when the positions have drifted off it, then depends on the stream's
// positions alone — not on which of its launches happened to take a matrix path — so that two contexts fed the same stream (a shard and an ordinary
// context, a group of another width) hold the same period and build the same rows.
void artfir_rows_touch (const ArtFirArgs *a, const ArtSegTable *segs)
{
    if (!a->rows_cache || !a->period_out || (a->mode & 3) != ART_MODE_FAST || a->n_end <= a->n_begin) return;
    const int mu = artfir_period_multiple (a->period_out, artfir_i8_slab_enabled () ? 64 : 32);
    HostPos pos0; int slot0, w;
    (void) artfir_rows_canonical (a, segs, mu * a->period_out, mu * a->period_in, (ArtRowsCache *) a->rows_cache, &pos0, &slot0, &w, false);
}

// Launch the matrix-core path for this call if it applies: returns ART_KERNEL_MFMA (| ART_FIR_ROLLED), -1 on a launch failure,
// 0 when the call is for the general kernel.
int artfir_matrix (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref, void *stream)
{
    hipStream_t st = (hipStream_t) stream;
    // MFMA path: exact rational ratio, default numeric mode, interleaved buffers, no history floor — and enough work
    // to beat the general kernel.  Cost models fitted to MI355X measurements (tools/bench_small_taps.py,
    // profiles/r1_small_calls.txt), n = output frames of the launch:
    //     general   ~ 5 us + n * k,    k = (0.2 + 0.04 C) + 0.00007 C T  ns per frame
    //     MFMA      ~ max (floor, work-bound),   floor = 13.5 us + 0.55 us per 32-tap chunk (+ 2 us for C <= 2) — since every tile,
    //                                             also those at the history seam, stages through the same loop (was 14 + 1.4)
    // The MFMA path is taken when the general kernel would take longer than the floor.  For channel counts without a
    // compiled column group the older rule stays: outputs x channels x taps of at least 1.2e8.
    const bool mfma_ok = artfir_takes_matrix_path (a, segs, kernel_pref);

    if (mfma_ok) {
        MfmaGeom g;
        const int cgt = matrix_geometry (a, g);
        const bool ws = cgt != 0;                             // (generic channel counts: the non-specialised instantiation on the same tiles)
        const bool wide = false;                              // (64-slot tiles, two m-tiles per X tile, lost: 156 VGPRs => one workgroup per CU, 29 vs 36 Gsamples/s in round 1; the template keeps the parameter, nothing instantiates it)
        {   // carve the per-launch tables out of the scratch buffer
            const size_t rows = (size_t) g.slot_tiles * g.tile_rows, eff_bytes = rows * g.ktot * sizeof (float);
            char *base = (char *) a->scratch;
            g.eff = (float *) base;
            g.canon_frac = (double *)(base + ((eff_bytes + 15) & ~(size_t) 15));
            g.canon_ip = (int *)(g.canon_frac + rows);
            g.canon_fi = g.canon_ip + rows;
            g.tile_w0 = g.canon_fi + rows;
            if (!base || (size_t)((char *)(g.tile_w0 + 3 * g.slot_tiles) - base) > a->scratch_bytes) return 0;
            g.head = nullptr; g.head_frames = 0;
            if (ws && !wide) {
                // the call's head as one contiguous array: everything a tile whose window starts inside the history can read
                // (+ the two chunks the staging runs ahead)
                const size_t used = (((size_t)((char *)(g.tile_w0 + 3 * g.slot_tiles) - base)) + 255) & ~(size_t) 255;
                g.head_frames = g.head_pad + a->H + (g.ppw - 1) * g.Q + g.ktot + 3 * MF_KC;
                if (used + (size_t) g.head_frames * a->C * sizeof (float) > a->scratch_bytes) return 0;
                g.head = (float *)(base + used);
            }
        }
        const unsigned int wg_threads = ws ? 2 * MF_THREADS : MF_THREADS;
        const unsigned int roll_blocks = a->roll_dst ? (unsigned int)((a->H * a->C + wg_threads - 1) / wg_threads) : 0u;
        dim3 grid ((unsigned int)(8 * g.groups_per_xcd * g.slot_tiles) + roll_blocks, (unsigned int)((a->C + g.cg - 1) / g.cg));

        const bool regular = ws && !wide && kernel_pref != 5 && (size_t) a->n_end * a->C * 4 < 0xffff0000ull && mfma_launch_is_regular (a, segs);
        if (a->segs_truncated && !regular) return 0;          // (the tile kernel replays positions from the table: not beyond it)
        // Fixed point on the integer matrix cores (fir_matrix_i8.hip) where the launch has its digit planes: staging pass + main
        // kernel, which carries the f32 tile loop as its own stand-by (a sample the digits cannot hold is only found on the
        // device) and takes the history roll along.  kernel_pref 6 pins the f32 kernel.
        if (regular && !ART_PREF_PINS_F32 (kernel_pref)) {
            const int i8 = artfir_i8_launch (a, segs, g, cgt, roll_blocks, st);       // 1: enqueued, 0: not for this launch, -1: failed part-way
            if (i8) return i8 > 0 && hipGetLastError () == hipSuccess ? (ART_KERNEL_MFMA | (a->roll_dst ? ART_FIR_ROLLED : 0)) : -1;
        }
        // The f32 streaming kernel on rows kept across calls (fir_matrix_i8.hip, "The rows across calls": the same canonical period, one set of
        // eff / canon_* / tile_w0 at the head of a->rows — no block alignment to honour): the launch is anchored on the canonical period
        // (n_skip slots of its first period computed and not stored), the set's linear indices carried w_shift frames on, and all the call
        // still prepares is its head (mfma_head_kernel) — or, once per stream, the set itself, from the canonical period's constants.
        // (Both forms of the streaming kernel, K split or not; the one-tile-per-workgroup kernel keeps building its rows from its own positions.)
        ArtFirArgs a_v = *a;
        ArtRowsTable tb; tb.on = 0; tb.lin = tb.w = 0; tb.n0 = 0u; tb.base = 0.0;
        bool rows_ready = false, on_kept_rows = false;
        {
            ArtRowsCache *rc = regular && g.head && artfir_rows_cache_enabled () && a->rows && artfir_f32_set_bytes (g) <= a->rows_bytes ? (ArtRowsCache *) a->rows_cache : nullptr;
            HostPos pos0; int slot0 = 0, w = 0;
            // (the virtual start's window inside the head's zero frames — head_pad covers a whole period's input: every tile of the launch's first period
            // group is based on the virtual period's start, so it is slot 0's window that must lie inside, not only the first stored slot's)
            if (rc && artfir_rows_canonical (a, segs, g.P, g.Q, rc, &pos0, &slot0, &w) &&
                rc->c_ip [0] + w - a->T / 2 + 1 >= -g.head_pad) {
                ArtFirArgs t = *a;
                t.n_begin = a->n_begin + (unsigned int)(g.P - slot0); t.n_end = a->n_end + (unsigned int) g.P;
                t.out = a->out - (size_t) g.P * a->C; t.n_skip = slot0;
                MfmaGeom g2;
                if (matrix_geometry (&t, g2) == cgt && g2.slot_tiles == g.slot_tiles && g2.ktot == g.ktot && (size_t) t.n_end * t.C * 4 < 0xffff0000ull) {
                    // (the tables in the set; the head stays in the call's scratch)
                    const size_t rows = (size_t) g.slot_tiles * 32, eff_bytes = rows * g.ktot * sizeof (float);
                    char *base = (char *) a->rows;
                    g2.eff = (float *) base;
                    g2.canon_frac = (double *)(base + ((eff_bytes + 15) & ~(size_t) 15));
                    g2.canon_ip = (int *)(g2.canon_frac + rows); g2.canon_fi = g2.canon_ip + rows; g2.tile_w0 = g2.canon_fi + rows;
                    g2.head = g.head; g2.head_frames = g.head_frames;
                    const bool same = rc->f_valid && rc->f_lowpass == a->lowpass && rc->f_slot_tiles == g.slot_tiles && rc->f_ktot == g.ktot;
                    if (same) { g2.w_shift = w - rc->f_w_build; rows_ready = true; }
                    else {
                        rc->f_valid = 1; rc->f_w_build = w; rc->f_lowpass = a->lowpass; rc->f_slot_tiles = g.slot_tiles; rc->f_ktot = g.ktot;
                        g2.w_shift = 0;
                        tb.on = 1; tb.base = rc->c_base; tb.lin = rc->c_lin; tb.n0 = rc->c_n0; tb.w = w;
                    }
                    a_v = t; g = g2; on_kept_rows = true;
                }
            }
        }
        // the cut-invariant policy runs anchored launches only: anything else is handed to the general kernel (0; the host counts it: resampleHipCutInvariantFallbacks)
        if (kernel_pref == ART_KERNEL_INVARIANT && !on_kept_rows) return 0;
        a = &a_v;
        {   static const bool trace = [] { const char *e = getenv ("ARTAMD_ROWS_TRACE"); return e && *e == '1'; } ();
            if (trace) fprintf (stderr, "rows (f32): launch n %u..%u C %d  kept %d  ready %d  n_skip %d  w_shift %d  regular %d split %p\n", a->n_begin, a->n_end, a->C, (int) on_kept_rows, (int) rows_ready, a->n_skip, g.w_shift, (int) regular, a->split);
        }
        if (rows_ready) {
            const unsigned int hb = (unsigned int)(((size_t) g.head_frames * a->C + 255) / 256);
            hipLaunchKernelGGL (mfma_head_kernel, dim3 (hb < 1u ? 1u : hb), dim3 (256), 0, st, *a, g);
        }
        else if (a->interpolate) hipLaunchKernelGGL (mfma_prepare_kernel<true>, dim3 (g.slot_tiles, g.tile_rows), dim3 (256), 0, st, *a, *segs, g, tb);
        else hipLaunchKernelGGL (mfma_prepare_kernel<false>, dim3 (g.slot_tiles, g.tile_rows), dim3 (256), 0, st, *a, *segs, g, tb);
        if (a->ev_start) arthip_event_record (a->ev_start, stream);

        // Regular launches (all but very long calls and nearest-filter phases on a half step) stream their tiles through
        // persistent workgroups: three per CU, each with an equal share of its XCD's tile list.  kernel_pref 5 pins the
        // one-tile-per-workgroup kernel (identical results; comparisons, tests).
        if (regular) {
            const int tiles_per_xcd = g.groups_per_xcd * g.slot_tiles;
            // As many workgroups as stay resident — 32 CUs per XCD x 3 (80 VGPRs, 46 KB of LDS), less two slots for the history-roll
            // workgroups of the same grid — each striding the XCD's tile list: a partly filled last round runs with fewer
            // workgroups per CU and finishes sooner.  (Measured: 4 channels x 1M frames = 139 tiles per XCD: 0.0888 ms, one tile
            // per workgroup 0.0948, two each 0.110; 8 channels = 280 tiles: the same 94 workgroups as equal shares of three.)
            const int resident = 94;
            int wgs_per_xcd = tiles_per_xcd < resident ? tiles_per_xcd : resident;
            { static const int k_env = [] { const char *e = getenv ("ARTAMD_TILES_PER_WG"); return e && *e ? atoi (e) : 0; } (); if (k_env > 0) wgs_per_xcd = (tiles_per_xcd + k_env - 1) / k_env; }
            // launches of few tiles: a tile's K range as several work items (fir_mfma_split_kernel)
            const bool fixup = artfir_pass_fixup_wanted (a);
            // (the rule looks at the launch's own outputs: the slots a launch on kept rows computes in front of its first do not count)
            const int ks = a->split ? matrix_split_parts (a, g, a->n_end - a->n_begin - (unsigned int) a->n_skip, kernel_pref) : 1;
            if (ks > 1 && (size_t) 8 * tiles_per_xcd * 16 <= ART_SPLIT_HEAD_BYTES &&
                ART_SPLIT_HEAD_BYTES + (size_t) 8 * tiles_per_xcd * ks * 16 * MF_THREADS * sizeof (double) <= a->split_bytes) {
                const int items = tiles_per_xcd * ks;
                const int wgs = items < resident ? items : resident;
                const dim3 kgrid ((unsigned int)(8 * wgs) + roll_blocks);
                unsigned int *arrivals = (unsigned int *) a->split;
                double *partials = (double *)((char *) a->split + ART_SPLIT_HEAD_BYTES);
#define MK_GO_(I, CGT, PS) hipLaunchKernelGGL ((fir_mfma_split_kernel<I, CGT, PS>), kgrid, dim3 (2 * MF_THREADS), 0, st, *a, g, wgs, ks, partials, arrivals)
#define MK_GO(I, CGT) do { if (!I && !a->lowpass && !fixup) MK_GO_ (false, CGT, true); else MK_GO_ (I, CGT, false); } while (0)
                if (a->interpolate) switch (cgt) { case 32: MK_GO (true, 32); break; case 16: MK_GO (true, 16); break; case 8: MK_GO (true, 8); break;
                                                    case 4: MK_GO (true, 4); break; case 2: MK_GO (true, 2); break; default: MK_GO (true, 1); }
                else                switch (cgt) { case 32: MK_GO (false, 32); break; case 16: MK_GO (false, 16); break; case 8: MK_GO (false, 8); break;
                                                    case 4: MK_GO (false, 4); break; case 2: MK_GO (false, 2); break; default: MK_GO (false, 1); }
#undef MK_GO
#undef MK_GO_
                if (a->ev_stop) arthip_event_record (a->ev_stop, stream);
                if (fixup && artfir_pass_fixup (a, g, st)) return -1;
                return hipGetLastError () == hipSuccess ? (ART_KERNEL_MFMA | (a->roll_dst ? ART_FIR_ROLLED : 0)) : -1;
            }
            const dim3 sgrid ((unsigned int)(8 * wgs_per_xcd) + roll_blocks);
#define MS_GO_(I, CGT, PS) hipLaunchKernelGGL ((fir_mfma_stream_kernel<I, CGT, PS>), sgrid, dim3 (2 * MF_THREADS), 0, st, *a, g, wgs_per_xcd)
#define MS_GO(I, CGT) do { if (!I && !a->lowpass && !fixup) MS_GO_ (false, CGT, true); else MS_GO_ (I, CGT, false); } while (0)
            if (a->interpolate) switch (cgt) { case 32: MS_GO (true, 32); break; case 16: MS_GO (true, 16); break; case 8: MS_GO (true, 8); break;
                                                case 4: MS_GO (true, 4); break; case 2: MS_GO (true, 2); break; default: MS_GO (true, 1); }
            else                switch (cgt) { case 32: MS_GO (false, 32); break; case 16: MS_GO (false, 16); break; case 8: MS_GO (false, 8); break;
                                                case 4: MS_GO (false, 4); break; case 2: MS_GO (false, 2); break; default: MS_GO (false, 1); }
#undef MS_GO
#undef MS_GO_
            if (a->ev_stop) arthip_event_record (a->ev_stop, stream);
            if (fixup && artfir_pass_fixup (a, g, st)) return -1;
            return hipGetLastError () == hipSuccess ? (ART_KERNEL_MFMA | (a->roll_dst ? ART_FIR_ROLLED : 0)) : -1;
        }
#define MF_GO(I, CGT) do { if (ws && CGT) hipLaunchKernelGGL ((fir_mfma_kernel<I, CGT, (CGT != 0), 1>), grid, dim3 (2 * MF_THREADS), 0, st, *a, *segs, g); \
                           else hipLaunchKernelGGL ((fir_mfma_kernel<I, CGT, false, 1>), grid, dim3 (MF_THREADS), 0, st, *a, *segs, g); } while (0)
        if (a->interpolate) switch (cgt) { case 32: MF_GO (true, 32); break; case 16: MF_GO (true, 16); break; case 8: MF_GO (true, 8); break; case 4: MF_GO (true, 4); break; case 2: MF_GO (true, 2); break;
                                            case 1: MF_GO (true, 1); break; default: MF_GO (true, 0); }
        else                switch (cgt) { case 32: MF_GO (false, 32); break; case 16: MF_GO (false, 16); break; case 8: MF_GO (false, 8); break; case 4: MF_GO (false, 4); break; case 2: MF_GO (false, 2); break;
                                            case 1: MF_GO (false, 1); break; default: MF_GO (false, 0); }
#undef MF_GO
        if (a->ev_stop) arthip_event_record (a->ev_stop, stream);
        // (outputs off the canonical pattern are evaluated inside the kernel, and its extra workgroups roll the history:
        // two launches per call — prepare, main — where there were three)
        return hipGetLastError () == hipSuccess ? (ART_KERNEL_MFMA | (a->roll_dst ? ART_FIR_ROLLED : 0)) : -1;
    }
    return 0;
}

#endif  // !ART_WIDE
