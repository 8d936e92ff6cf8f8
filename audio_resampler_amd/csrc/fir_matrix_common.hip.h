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
    // the head of the call as ONE contiguous array (history ++ first input frames, MF_HEAD_PAD zero frames in front): tiles that
    // reach into the history stage from it exactly as all others stage from `in` (written by mfma_prepare_kernel)
    float *head; int head_frames;
    // per slot tile, 3 ints: [0] linear index of K column 0 in period 0 of the launch; [1] the tile's pass-through rows (nearest-
    // filter mode without a low-pass: the slots whose rounded filter index is a whole input sample, which the reference copies),
    // one bit per row — such a row's sample in period 0 is canon_ip + canon_fi / F; [2] unused (streaming kernels)
    int *tile_w0;
    // rows kept across calls (fir_matrix_i8.hip): canon_ip and tile_w0 [3 st] are the BUILDING launch's linear indices; this launch's are
    // w_shift frames further on (0: the tables were written by this launch)
    int w_shift;
};

// fir_matrix_i8.hip: the fixed-point kernel of regular launches
size_t artfir_i8_bytes (const ArtFirArgs *a, const MfmaGeom &g, int cgt, unsigned int outputs);
size_t artfir_i8_rows_bytes (const ArtFirArgs *a, const MfmaGeom &g, int cgt, unsigned int outputs);   // device bytes of the rows kept ACROSS calls (0: none)   // device bytes of the digit planes of a call making `outputs` frames (0: not for it)
bool artfir_i8_slab_enabled ();           // 64-slot tiles of the slab kernel (periods are taken so as to fill those); ARTAMD_I8_SLAB=0: off
// stage + main kernel of one launch (1), or 0: not for this path (no planes, shape)
int artfir_i8_launch (const ArtFirArgs *a, const ArtSegTable *segs, const MfmaGeom &g, int cgt, unsigned int roll_blocks, hipStream_t st);

// Nearest-filter mode without a low-pass: the outputs whose position falls exactly on an input sample are copies of that sample
// (reference resampler.c:1141-1142).  The matrix kernels' PASS instantiations substitute them in their epilogues — and pay for the extra
// live state with 20-40 spilt registers in the tile loop.  The streaming kernels' launches (ARTAMD_PASS_FIXUP_MIN: from that many samples
// on; default all of them) run the plain instantiation and this pass behind it on the same stream: the flagged slots (tile_w0 [3 st + 1], 32-row slot tiles) of every period
// are overwritten with their samples — the same values the PASS epilogues store.  Returns true if the launch is to run that way.
bool artfir_pass_fixup_wanted (const ArtFirArgs *a);
void artfir_pass_fixup (const ArtFirArgs *a, const MfmaGeom &g, hipStream_t st);

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
