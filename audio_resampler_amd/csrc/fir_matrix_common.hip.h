#pragma once
// fir_matrix_common.hip.h — what the 4-byte-sample matrix-core translation units share (fir_matrix.hip: f32 MFMA kernels and the
// launch rules; fir_matrix_i8.hip: the fixed-point kernel): tile geometry of one launch, raw buffer resources, vector loads.
#include "fir_common.hip.h"

struct MfmaGeom {
    int P, Q;                             // outputs / inputs per period
    int tile_rows;                        // slots per workgroup tile: 32, or 64 (two MFMA m-tiles sharing one X tile)
    int slot_tiles;                       // ceil (P / tile_rows)
    int ppw;                              // periods per workgroup
    int cg;                               // channels per column group
    int ktot;                             // K columns, multiple of MF_KC
    int period_groups;
    int groups_per_xcd;                   // ceil (period_groups / 8)
    int band_lo, band_hi;
    // per-launch tables in device scratch (written by mfma_prepare_kernel)
    float *eff;                           // [slot_tiles*tile_rows][ktot]  blended rows, shifted to the tile's K origin, zero padded
    int *canon_ip, *canon_fi;             // [slot_tiles*tile_rows]        canonical position of each slot (period 0 of the launch)
    double *canon_frac;                 // K columns [band_lo, band_hi) hold every row's central taps
    // the head of the call as ONE contiguous array (history ++ first input frames, head_pad zero frames in front): tiles that
    // reach into the history stage from it exactly as all others stage from `in` (written by mfma_prepare_kernel)
    float *head; int head_frames;
    // zero frames in front of linear frame 0 in the head: MF_HEAD_PAD, more where a period's input is longer than half a window + that — a launch
    // anchored on the canonical period starts its tiles up to Q frames in front of its first output's window (downsampling streams; round 6)
    int head_pad;
    // per slot tile, 3 ints: [0] linear index of K column 0 in period 0 of the launch; [1] the tile's pass-through rows (nearest-
    // filter mode without a low-pass: the slots whose rounded filter index is a whole input sample, which the reference copies),
    // one bit per row — such a row's sample in period 0 is canon_ip + canon_fi / F; [2] unused (streaming kernels)
    int *tile_w0;
    // rows kept across calls (fir_matrix_i8.hip): canon_ip and tile_w0 [3 st] are the BUILDING launch's linear indices; this launch's are
    // w_shift frames further on (0: the tables were written by this launch)
    int w_shift;
};

// fir_matrix_i8.hip: the fixed-point kernel of regular launches
size_t artfir_i8_bytes (const ArtFirArgs *a, const MfmaGeom &g, int cgt, unsigned int outputs);        // device bytes of the digit planes of a call making `outputs` frames (0: not for it)
size_t artfir_i8_rows_bytes (const ArtFirArgs *a, const MfmaGeom &g, int cgt, unsigned int outputs);   // device bytes of the rows kept ACROSS calls (0: none)
bool artfir_i8_slab_enabled ();           // 64-slot tiles of the slab kernel (periods are taken so as to fill those); ARTAMD_I8_SLAB=0: off
// stage + main kernel of one launch (1), 0: not for this path (no planes, shape), -1: a launch of it failed
int artfir_i8_launch (const ArtFirArgs *a, const ArtSegTable *segs, const MfmaGeom &g, int cgt, unsigned int roll_blocks, hipStream_t st);

// Nearest-filter mode without a low-pass: the outputs whose position falls exactly on an input sample are copies of that sample
// (reference resampler.c:1141-1142).  The matrix kernels' PASS instantiations substitute them in their epilogues — and pay for the extra
// live state with 20-40 spilt registers in the tile loop.  The streaming kernels' launches (ARTAMD_PASS_FIXUP_MIN: from that many samples
// on; default all of them) run the plain instantiation and this pass behind it on the same stream: the flagged slots (tile_w0 [3 st + 1], 32-row slot tiles) of every period
// are overwritten with their samples — the same values the PASS epilogues store.  Returns true if the launch is to run that way.
bool artfir_pass_fixup_wanted (const ArtFirArgs *a);
int  artfir_pass_fixup (const ArtFirArgs *a, const MfmaGeom &g, hipStream_t st);        // 0, or -1: not launched

namespace {

typedef float f32x16 __attribute__ ((ext_vector_type (16)));
typedef float f32x4 __attribute__ ((ext_vector_type (4)));

constexpr int MF_THREADS = 256;
constexpr int MF_KC = 32;                 // k's per staged chunk
constexpr int MF_LD = MF_KC + 4;          // LDS row pitch in floats (144 B: 16-B aligned, conflict-free b128)
constexpr int MF_COLS = 128;              // columns per workgroup
constexpr int MF_MAX_PPW = 64;            // periods per workgroup (C = 2)
// A slot's phase may differ from its canonical value by this many filter steps and still use the tile's
// effective row: adjacent rows differ by < 3e-3 per tap, so the row changes by < 6e-9 relative (a tenth of
// half a float ulp).  The reference's own position arithmetic is quantised to ~1e-7 steps after 1M frames.
constexpr double MF_PHASE_TOL = 2e-6;
constexpr int MF_HEAD_PAD = 64;


typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));
typedef unsigned int u32x2 __attribute__ ((ext_vector_type (2)));

// raw buffer descriptor: the hardware range check returns 0 for any access past `bytes`, which is
// exactly the zero padding the tile needs (beyond the valid input, before/after a filter row)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc (const void *base, unsigned int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc (const_cast<void *> (base), 0, (int) bytes, 0x00020000);
}

template <int VEC> struct VecLoad;
template <> struct VecLoad<1> { static __device__ __forceinline__ void load (float *dst, __amdgpu_buffer_rsrc_t r, unsigned int off) {
    dst [0] = __uint_as_float (__builtin_amdgcn_raw_buffer_load_b32 (r, (int) off, 0, 0)); } };
template <> struct VecLoad<2> { static __device__ __forceinline__ void load (float *dst, __amdgpu_buffer_rsrc_t r, unsigned int off) {
    u32x2 v = __builtin_amdgcn_raw_buffer_load_b64 (r, (int) off, 0, 0);
    dst [0] = __uint_as_float (v.x); dst [1] = __uint_as_float (v.y); } };
template <> struct VecLoad<4> { static __device__ __forceinline__ void load (float *dst, __amdgpu_buffer_rsrc_t r, unsigned int off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128 (r, (int) off, 0, 0);
    dst [0] = __uint_as_float (v.x); dst [1] = __uint_as_float (v.y); dst [2] = __uint_as_float (v.z); dst [3] = __uint_as_float (v.w); } };


} // namespace

// ---------------------------------------------------------------------------------------------------
// Filter rows kept across calls (ArtFirArgs.rows on the device, this block on the host: ArtFirArgs.rows_cache) — the scheme is described
// in front of artfir_i8_launch (fir_matrix_i8.hip); the f32 streaming kernel keeps one set of its own tables (eff / canon_* / tile_w0) at the
// head of the same device buffer (fir_matrix.hip).
// ---------------------------------------------------------------------------------------------------
struct ArtRowsCache {
    // the canonical period
    const void *bank; int T, F, interp, P, Q; double ratio;
    int canon_valid, cap;
    int last_slot, next_slot;             // where the last launch looked up started and where a launch that continues it will (tried before the scan)
    double *c_ph; int *c_ip, *c_fi;       // [cap >= P]
    double c_base; int c_lin; unsigned int c_n0;              // ... as the device evaluates it: epoch offset, ring-to-linear shift, first output
    long long c_origin;                   // ArtFirArgs.lin_origin of the founding launch: (w + lin_origin - c_origin) frames is a launch's distance from the period IN THE STREAM
    // the fixed-point kernel's sets built for it
    int lowpass, tr, ktot, tiles, g, slot_tiles, ktot32; size_t set_bytes;
    int nsets, victim, valid [4], w_build [4];
    // the f32 streaming kernel's set
    int f_valid, f_w_build, f_lowpass, f_slot_tiles, f_ktot;
};
// the canonical period's constants as a kernel argument (the row workgroups evaluate slot k themselves)
struct ArtRowsTable { int on; int lin, w; unsigned int n0; double base; };

namespace {

struct HostPos { int ip, fi; double ph; };
// locate () on the host: the reference's position arithmetic (compiled, like the device's, without contraction)
__attribute__ ((unused)) static HostPos host_locate (const ArtFirArgs *a, const ArtSegTable *segs, unsigned int n)
{
    int e = 0;
    while (e + 1 < segs->count && segs->first [e + 1] <= n) ++e;
    const double step = n ? (double) n / a->ratio : 0.0;
    const double off = segs->base [e] + step;
    const double whole = floor (off);
    double fr = off - whole;
    fr = fr * (double) a->F;
    HostPos p;
    p.ph = fr;
    p.fi = a->interpolate ? (int) floor (fr) : (int) floor (fr + 0.5);
    p.ip = (int) whole + segs->lin_base [e];
    return p;
}

__attribute__ ((unused)) static bool artfir_rows_cache_enabled ()
{
    static const bool on = [] { const char *e = getenv ("ARTAMD_ROWS_CACHE"); return !(e && *e == '0'); } ();
    return on;
}

// device bytes of the f32 streaming kernel's set (at the head of ArtFirArgs.rows)
__attribute__ ((unused)) static size_t artfir_f32_set_bytes (const MfmaGeom &g)
{
    const size_t rows = (size_t) g.slot_tiles * 32;
    return ((rows * g.ktot * sizeof (float) + rows * (sizeof (double) + 2 * sizeof (int)) + (size_t) 3 * g.slot_tiles * sizeof (int) + 1024) + 4095) & ~(size_t) 4095;
}

// The launch's place in the stream's canonical period: *pos0 = its first output's position; with a cache, slot0 / w = the canonical slot
// that output is and the whole number of frames it sits further on (a new canonical period — this launch's first — where the stream has
// left the old one: every set is then invalid).  false: no cache for this launch (none given, or out of memory).
// verify_nearest: a nearest-filter launch that is about to USE kept rows checks the canonical filter index of every slot against its own positions
// (the matrix paths); the per-launch upkeep of the canonical period (artfir_rows_touch) does not.
__attribute__ ((unused)) static bool artfir_rows_canonical (const ArtFirArgs *a_in, const ArtSegTable *segs, int P, int Q, ArtRowsCache *rc, HostPos *pos0_out, int *slot0_out, int *w_out,
                                                            bool verify_nearest = true)
{
    const HostPos pos0 = host_locate (a_in, segs, a_in->n_begin);
    *pos0_out = pos0; *slot0_out = 0; *w_out = 0;
    if (!rc) return false;
    const bool same = rc->canon_valid && rc->bank == (const void *) a_in->bank && rc->T == a_in->T && rc->F == a_in->F && rc->interp == a_in->interpolate &&
                      rc->P == P && rc->Q == Q && rc->ratio == a_in->ratio;
    if (!same) rc->canon_valid = 0;
    if (rc->cap < P) {
        free (rc->c_ph); free (rc->c_ip); free (rc->c_fi);
        rc->cap = P; rc->canon_valid = 0;
        rc->c_ph = (double *) malloc (sizeof (double) * (size_t) rc->cap); rc->c_ip = (int *) malloc (sizeof (int) * (size_t) rc->cap); rc->c_fi = (int *) malloc (sizeof (int) * 2 * (size_t) rc->cap);       // (+ cap flags behind the indices: slots verified below)
        if (!rc->c_ph || !rc->c_ip || !rc->c_fi) { free (rc->c_ph); free (rc->c_ip); free (rc->c_fi); rc->c_ph = nullptr; rc->c_ip = rc->c_fi = nullptr; rc->cap = 0; return false; }
    }
    int slot0 = 0, w = 0;
    if (rc->canon_valid) {
        const double tol = 1e-6, F = (double) a_in->F;
        bool found = false;
        // (a stream's launches follow one another: the slot behind the last launch's outputs first, the last launch's own slot second — the
        // upkeep and the matrix path look the same launch up twice — and only then the whole period)
        // (where several periods are taken at a time the period's slots repeat every period_out: the FIRST of the equal slots is the launch's,
        // whichever way it was found — the choice must not depend on a context's history)
        const int P0 = a_in->period_out > 0 && P % a_in->period_out == 0 ? a_in->period_out : P;
        for (int t = -2; t < P0 && !found; ++t) {
            const int s = t == -2 ? rc->next_slot % P0 : t == -1 ? rc->last_slot % P0 : t;
            if (s < 0 || s >= P) continue;
            double d = fabs (rc->c_ph [s] - pos0.ph);
            if (d > 0.5 * F) d = F - d;
            if (d > tol) continue;
            // (positions in frames: the same lattice point up to the tolerance, a whole number of frames apart)
            const double w_exact = ((double) pos0.ip + pos0.ph / F) - ((double) rc->c_ip [s] + rc->c_ph [s] / F);
            const int wr = (int) floor (w_exact + 0.5);
            if (fabs (w_exact - (double) wr) > 1e-6) break;
            found = true; slot0 = s; w = wr;
        }
        // Several periods at a time (P = mu x period_out): slots s, s + period_out, ... are the same phase, and the launch's first output is ONE of them in the
        // stream's own tiling — the one a whole number of P-periods (Q frames) behind its canonical slot.  Round 5 always took the first: right where
        // period_out is a multiple of the 32-row tiles (160: slot s + 160 sits in the same row of a tile five further on, on the same K chunks), wrong where it
        // is not (96k -> 44.1k: 147 x 3 — a launch that starts in its period's second or third part was anchored 147 / 294 slots off the tiling a longer launch
        // walks: other tile rows, other K chunks, the last bits moved with the cut; tests/test_gpu_cut_invariance.py, round 6).  Which part: from the frames
        // between the launch and the canonical period — the stream's positions alone, not a context's history.
        static const bool first_slot = [] { const char *e = getenv ("ARTAMD_ROWS_FIRST_SLOT"); return e && *e == '1'; } ();      // (A/B runs: round 5's choice)
        if (found && P0 < P && Q % (P / P0) == 0 && !first_slot) {
            const int mu = P / P0, Q0 = Q / mu;
            // (w counts linear frames, whose origin moves with every call: the distance in the stream is w + the calls' input in between)
            const long long w_stream = (long long) w + (a_in->lin_origin - rc->c_origin);
            if (w_stream % Q0 == 0) {
                const int k = (int)(((w_stream / Q0) % mu + mu) % mu), sk = slot0 + k * P0;
                if (k && sk < P) {
                    double d = fabs (rc->c_ph [sk] - pos0.ph);
                    if (d > 0.5 * F) d = F - d;
                    const double w_exact = ((double) pos0.ip + pos0.ph / F) - ((double) rc->c_ip [sk] + rc->c_ph [sk] / F);
                    const int wr = (int) floor (w_exact + 0.5);
                    if (d <= tol && fabs (w_exact - (double) wr) <= 1e-6 && ((long long) wr + (a_in->lin_origin - rc->c_origin)) % Q == 0) { slot0 = sk; w = wr; }
                }
            }
        }
        // (nearest filter: the canonical rounded filter index in every slot, from this launch's own positions — P position evaluations on the host, 6 - 13 us at
        // P = 320 on a path of ~20 us a call: done ONCE per start slot of a canonical period.  A later launch that starts on the same slot sits on the same
        // lattice a whole number of frames on, and the launch's regularity test has already excluded slots near a half step: ADVICE r5)
        int *const c_ok = rc->c_fi + rc->cap;
        const bool whole_period = a_in->n_end - a_in->n_begin >= (unsigned int) P;
        if (found && !a_in->interpolate && verify_nearest && !c_ok [slot0]) {
            for (int t = 0; t < P && found; ++t) {                // (its first P outputs are slots s, s + 1, ... of the canonical period, wrapping into the next)
                const unsigned int n = a_in->n_begin + (unsigned int) t;
                if (n >= a_in->n_end) break;
                // (compared as frame x F + filter: a slot ON an input sample is (ip, F) from a position a hair below it and (ip + 1, 0) from one a hair above —
                // the bank's row F is row 0 one frame on, the same taps on the same samples — and the two must not count as a stream that has left its
                // period: an exact-ratio stream whose filters are a multiple of its phases has such slots in every period, and re-founding the period on
                // whichever launch saw the other spelling first moved the tiles, i.e. the bits, with the way the input was cut into calls)
                const int sidx = slot0 + t, sp = sidx % P;
                const HostPos p = host_locate (a_in, segs, n);
                found = (long long) p.ip * a_in->F + p.fi == ((long long) rc->c_ip [sp] + w + (long long) Q * (sidx / P)) * a_in->F + rc->c_fi [sp];
            }
            if (found && whole_period) c_ok [slot0] = 1;
        }
        if (!found) rc->canon_valid = 0;
    }
    if (!rc->canon_valid) {                                   // a new canonical period: this launch's first
        // (the epoch of the launch's first output, carried through the whole period even where the ring rewinds inside it: the same lattice,
        // and constants the row workgroups can evaluate themselves)
        int e = 0;
        while (e + 1 < segs->count && segs->first [e + 1] <= a_in->n_begin) ++e;
        rc->c_base = segs->base [e]; rc->c_lin = segs->lin_base [e]; rc->c_n0 = a_in->n_begin; rc->c_origin = a_in->lin_origin;
        ArtSegTable one; one.count = 1; one.lin_floor = segs->lin_floor; one.first [0] = 0u; one.lin_base [0] = rc->c_lin; one.base [0] = rc->c_base;
        for (int i = 0; i < P; ++i) { const HostPos p = host_locate (a_in, &one, rc->c_n0 + (unsigned int) i); rc->c_ph [i] = p.ph; rc->c_ip [i] = p.ip; rc->c_fi [i] = p.fi; }
        rc->bank = (const void *) a_in->bank; rc->T = a_in->T; rc->F = a_in->F; rc->interp = a_in->interpolate; rc->P = P; rc->Q = Q; rc->ratio = a_in->ratio;
        for (int i = 0; i < P; ++i) (rc->c_fi + rc->cap) [i] = 0;
        rc->canon_valid = 1; slot0 = 0; w = 0;
        for (int k = 0; k < 4; ++k) rc->valid [k] = 0;
        rc->f_valid = 0;
    }
    rc->last_slot = slot0; rc->next_slot = (int)(((unsigned int) slot0 + (a_in->n_end - a_in->n_begin)) % (unsigned int) P);
    *slot0_out = slot0; *w_out = w;
    return true;
}

} // namespace
