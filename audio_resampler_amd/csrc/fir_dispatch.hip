// fir_dispatch.hip — arthip_fir: one FIR call -> the kernel that runs it (strict order, matrix cores, general).
#include "fir_common.hip.h"
#include <atomic>
#include <cstdlib>

extern "C" {

int arthip_fir_takes_matrix_path (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref) { return artfir_takes_matrix_path (a, segs, kernel_pref) ? 1 : 0; }

int artamdPeriodMultiple (int outputsPerPeriod) { return outputsPerPeriod > 0 ? artfir_period_multiple (outputsPerPeriod, 32) : 0; }
int artamdPeriodMultipleRows (int outputsPerPeriod, int rows) { return outputsPerPeriod > 0 && rows > 0 ? artfir_period_multiple (outputsPerPeriod, rows) : 0; }

int arthip_fir_spans_segments (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref) { return artfir_matrix_spans_segments (a, segs, kernel_pref) ? 1 : 0; }

size_t arthip_fir_split_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref) { return artfir_split_bytes (a, outputs, kernel_pref); }

size_t arthip_fir_planes_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref) { return artfir_planes_bytes (a, outputs, kernel_pref); }

size_t arthip_fir_rows_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref) { return artfir_rows_bytes (a, outputs, kernel_pref); }

// ---------------------------------------------------------------------------------------------------
// Channel counts the matrix-core kernels are not compiled for.  Their tile loops index a stream of exactly 1, 2, 4, 8, 16 or 32 channels
// (compile-time pitch, 16-byte vectors); a 6- or 12- or 64-channel stream used to take the generic instantiation, 5-8 x slower per
// sample (6 ch x 988 taps, 262,144 frames: 7.0 Gsamples/s where 8 channels make 39).  Such a launch now runs in GROUPS of up to 32
// channels: a group's history and input are copied into a buffer of the next compiled width (the extra channels zero), the
// ordinary launch runs on that, and its outputs are copied back into the caller's frames.  A channel's arithmetic does not depend on
// its group's width or on its neighbours: the same bits as the same channel in any other context (a shard, a wider stream).
// ---------------------------------------------------------------------------------------------------
static inline bool channels_irregular (int C) { return C > 32 || (C & (C - 1)) != 0; }
// (never narrower than 4: the last group of a 33- or 34-channel stream runs the kernels every other group of the stream runs — the
// 16-byte-vector forms, fixed point where the stream's size asks for it — not the 1- and 2-channel streams' own choice)
static inline int padded_width (int w) { int p = 4; while (p < w) p <<= 1; return p; }
static inline size_t align256 (size_t b) { return (b + 255) & ~(size_t) 255; }
// a group's buffer: [history][input][outputs], each on a 256-byte boundary (the matrix kernels take 16-byte vectors from the input and
// look at its alignment when they choose their instantiation: directly behind a history of 1.5 T frames a 1-wide group's input sat on
// an 8-byte boundary and ran the generic instantiation — other bits than the same channel has anywhere else)
static inline size_t group_hist_bytes (const ArtFirArgs *a, size_t wp) { return align256 ((size_t) a->H * wp * sizeof (art_s)); }
static inline size_t group_in_bytes (const ArtFirArgs *a, size_t wp) { return align256 ((size_t) a->in_frames * wp * sizeof (art_s)); }

__global__ void group_in_kernel (art_s *dst, art_s *dst_in, const art_s *hist, const art_s *in, int H, int in_frames, int C, int c0, int w, int wp)
{
    const size_t total = (size_t)(H + in_frames) * wp, stride = (size_t) gridDim.x * blockDim.x;
    for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t f = e / wp; const int c = (int)(e - f * wp);
        art_s v = 0;
        if (c < w) v = f < (size_t) H ? hist [f * C + c0 + c] : in [(f - H) * C + c0 + c];
        // ([history frames][wp], then — on a 256-byte boundary of its own, as a caller's buffer would be — [input frames][wp])
        if (f < (size_t) H) dst [e] = v; else dst_in [e - (size_t) H * wp] = v;
    }
}

__global__ void group_out_kernel (art_s *out, const art_s *src, unsigned int n_begin, unsigned int n_end, int C, int c0, int w, int wp)
{
    const size_t total = (size_t)(n_end - n_begin) * w, stride = (size_t) gridDim.x * blockDim.x;
    for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t r = e / w; const int c = (int)(e - r * w);
        out [((size_t) n_begin + r) * C + c0 + c] = src [r * wp + c];
    }
}

size_t arthip_fir_pad_bytes (const ArtFirArgs *a, unsigned int outputs)
{
    if (!channels_irregular (a->C) || a->in_pitch || a->out_pitch || (a->mode & 3) != ART_MODE_FAST) return 0;
    const size_t wp = a->C > 32 ? 32 : (size_t) padded_width (a->C);
    return group_hist_bytes (a, wp) + group_in_bytes (a, wp) + (size_t) outputs * wp * sizeof (art_s) + 256;
}

// 0: not for this launch (the caller goes on as before); else artfir_matrix's return value for the whole launch
static int fir_in_groups (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref, hipStream_t st)
{
    static const bool off = [] { const char *e = getenv ("ARTAMD_NO_GROUPS"); return e && *e && *e != '0'; } ();      // (A/B runs: the generic matrix kernel as before)
    if (off || !a->pad || !channels_irregular (a->C) || a->in_pitch || a->out_pitch || (a->mode & 3) != ART_MODE_FAST || (!a->in && a->in_frames > 0)) return 0;
    if (!artfir_takes_matrix_path (a, segs, kernel_pref)) return 0;
    const unsigned int outs = a->n_end - a->n_begin;
    if (arthip_fir_pad_bytes (a, outs) > a->pad_bytes) return 0;
    const size_t wp_max = a->C > 32 ? 32 : (size_t) padded_width (a->C);
    art_s *p_hist = (art_s *) a->pad;
    art_s *p_in = (art_s *)((char *) a->pad + group_hist_bytes (a, wp_max));
    art_s *p_out = (art_s *)((char *) p_in + group_in_bytes (a, wp_max));
    int rc = 0;
    // (timing: ONE pair of events around all the groups — their copies and staging passes are this launch's time)
    if (a->ev_start) arthip_event_record (a->ev_start, (void *) st);
    for (int c0 = 0; c0 < a->C; c0 += 32) {
        const int w = a->C - c0 < 32 ? a->C - c0 : 32, wp = padded_width (w);
        const size_t in_elems = (size_t)(a->H + a->in_frames) * wp;
        hipLaunchKernelGGL (group_in_kernel, dim3 ((unsigned int)((in_elems + 255) / 256 < 4096 ? (in_elems + 255) / 256 : 4096)), dim3 (256), 0, st,
                            p_hist, p_in, a->hist, a->in, a->H, a->in_frames, a->C, c0, w, wp);
        ArtFirArgs b = *a;
        b.C = wp; b.hist = p_hist; b.in = p_in;
        b.ev_start = b.ev_stop = nullptr;
        b.out = p_out - (size_t) a->n_begin * wp;             // (the kernels index outputs from the call's first)
        b.roll_dst = nullptr; b.roll_appended = 0;            // (the history is rolled once, below, in the stream's own layout)
        b.stream_C = a->stream_C > a->C ? a->stream_C : a->C;
        b.pad = nullptr; b.pad_bytes = 0;
        const int r = artfir_matrix (&b, segs, kernel_pref, (void *) st);
        if (r <= 0) { if (c0 == 0 && r == 0) return 0; return -1; }      // (declined before anything ran: the caller's other paths; later: cannot be, same decisions)
        rc = r;
        const size_t out_elems = (size_t) outs * w;
        hipLaunchKernelGGL (group_out_kernel, dim3 ((unsigned int)((out_elems + 255) / 256 < 4096 ? (out_elems + 255) / 256 : 4096)), dim3 (256), 0, st,
                            a->out, p_out, a->n_begin, a->n_end, a->C, c0, w, wp);
    }
    if (a->ev_stop) arthip_event_record (a->ev_stop, (void *) st);
    if (a->roll_dst && arthip_roll_history (a->roll_dst, a->hist, a->in, 0, a->roll_appended, a->H, a->C, (void *) st)) return -1;
    if (hipGetLastError () != hipSuccess) return -1;
    return (rc & ~ART_FIR_ROLLED) | (a->roll_dst ? ART_FIR_ROLLED : 0);
}

int arthip_fir (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref, void *stream)
{
    hipStream_t st = (hipStream_t) stream;

    if (a->n_end <= a->n_begin) return ART_KERNEL_GENERAL;
    {   // (test hook, tests/test_gpu_failure_path.py: ARTAMD_TEST_FAIL_FIR=k makes the k-th FIR launch of the process fail before anything is enqueued —
        // the only way to see the host's failure path without breaking a device)
        static const int fail_at = [] { const char *e = getenv ("ARTAMD_TEST_FAIL_FIR"); return e && *e ? atoi (e) : 0; } ();
        static std::atomic<int> launches { 0 };
        if (fail_at > 0 && ++launches == fail_at) return -1;
    }
    artfir_rows_touch (a, segs);                              // (the canonical period of the rows kept across calls: every launch looks after it)

    if (a->segs_truncated && ((a->mode & 3) == ART_MODE_STRICT || !artfir_matrix_spans_segments (a, segs, kernel_pref))) return -2;

    if ((a->mode & 3) == ART_MODE_STRICT) {
        artfir_strict (*a, *segs, (a->mode & 4) != 0, st);
        if (a->ev_start) { arthip_event_record (a->ev_start, stream); arthip_event_record (a->ev_stop, stream); }
        return hipGetLastError () == hipSuccess ? ART_KERNEL_GENERAL : -1;
    }

    // matrix cores: exact rational ratio, default numeric mode, interleaved buffers, no history floor — and enough work to beat
    // the general kernel (fir_matrix.hip / fir_matrix64.hip)
    { const int grouped = fir_in_groups (a, segs, kernel_pref, st); if (grouped) return grouped; }
    const int matrix = artfir_matrix (a, segs, kernel_pref, stream);
    if (matrix) return matrix;
    if (a->segs_truncated) return -2;                        // (nothing enqueued: the matrix path declines before its first launch)

    if (a->ev_start) arthip_event_record (a->ev_start, stream);
    if (artfir_general (*a, *segs, st)) {
        // The tile's input span does not fit the LDS (ratios below ~1/4000 with long filters: thousands of input frames per
        // output).  The reference accepts any positive ratio, and a caller that loops until its input is consumed must not
        // see "nothing done": one lane per output sample reading HBM directly (the strict-order kernel: reference source
        // order, float or double accumulator as the mode asks) — slow, correct, and only ever reached by such ratios.
        artfir_strict (*a, *segs, (a->mode & 3) == ART_MODE_PRECISE, st);
        if (a->ev_stop) arthip_event_record (a->ev_stop, stream);
        return hipGetLastError () == hipSuccess ? ART_KERNEL_GENERAL : -1;        // (no history roll rode along: the host launches it)
    }
    if (a->ev_stop) arthip_event_record (a->ev_stop, stream);
    return hipGetLastError () == hipSuccess ? (ART_KERNEL_GENERAL | (a->roll_dst ? ART_FIR_ROLLED : 0)) : -1;
}

}
