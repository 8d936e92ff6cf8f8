// fir_dispatch.hip — arthip_fir: one FIR call -> the kernel that runs it (strict order, matrix cores, general).
#include "fir_common.hip.h"

extern "C" {

int arthip_fir_takes_matrix_path (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref) { return artfir_takes_matrix_path (a, segs, kernel_pref) ? 1 : 0; }

int artamdPeriodMultiple (int outputsPerPeriod) { return outputsPerPeriod > 0 ? artfir_period_multiple (outputsPerPeriod, 32) : 0; }
int artamdPeriodMultipleRows (int outputsPerPeriod, int rows) { return outputsPerPeriod > 0 && rows > 0 ? artfir_period_multiple (outputsPerPeriod, rows) : 0; }

int arthip_fir_spans_segments (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref) { return artfir_matrix_spans_segments (a, segs, kernel_pref) ? 1 : 0; }

size_t arthip_fir_split_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref) { return artfir_split_bytes (a, outputs, kernel_pref); }

size_t arthip_fir_planes_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref) { return artfir_planes_bytes (a, outputs, kernel_pref); }

int arthip_fir (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref, void *stream)
{
    hipStream_t st = (hipStream_t) stream;

    if (a->n_end <= a->n_begin) return ART_KERNEL_GENERAL;

    if (a->segs_truncated && ((a->mode & 3) == ART_MODE_STRICT || !artfir_matrix_spans_segments (a, segs, kernel_pref))) return -2;

    if ((a->mode & 3) == ART_MODE_STRICT) {
        artfir_strict (*a, *segs, (a->mode & 4) != 0, st);
        if (a->ev_start) { arthip_event_record (a->ev_start, stream); arthip_event_record (a->ev_stop, stream); }
        return hipGetLastError () == hipSuccess ? ART_KERNEL_GENERAL : -1;
    }

    // matrix cores: exact rational ratio, default numeric mode, interleaved buffers, no history floor — and enough work to beat
    // the general kernel (fir_matrix.hip / fir_matrix64.hip)
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
