/* stretch.h — time-domain harmonic scaler (tempo / pitch changes without resampling artefacts), C API.
 *
 * Drop-in boundary for the reference's stretch.h (reference stretch.h:30-52): same constants, flags, entry points and
 * call semantics, so ART's --pitch / --tempo / --duration path links against libartamd.so without stretch.c.  The
 * context is created by stretchInit and only ever handled through the API (ART never looks inside, art.c:787-1137);
 * the leading fields keep the reference's names and order, `hip` holds the device side.
 *
 * Each stretchProcess / stretchFlush call is ONE launch of a persistent single-workgroup gfx950 kernel that runs the
 * whole buffering / period-search / cross-fade state machine of the call on the device (csrc/stretch_kernels.hip):
 * every float (or double) operation in the reference's order, bit-identical output and per-call frame counts.
 * Mono or stereo only, as in the reference.
 */
#ifndef ARTAMD_STRETCH_H
#define ARTAMD_STRETCH_H

#include <stdint.h>

#ifndef ARTSAMPLE_T_DEFINED
#define ARTSAMPLE_T_DEFINED
#if defined(PATH_WIDTH) && (PATH_WIDTH==64)
typedef double artsample_t;
#else
typedef float artsample_t;
#endif
#endif

#define MIN_PERIOD  24                  /* shortest pitch period accepted, in frames */
#define MAX_PERIOD  2400                /* longest pitch period accepted, in frames */

enum {
    STRETCH_FAST_FLAG = 0x1,            /* 2:1 decimated period search with three-point refinement */
    STRETCH_DUAL_FLAG = 0x2             /* two cascaded stages: usable ratio range 0.25 .. 4 instead of 0.5 .. 2 */
};

struct artamd_stretch;

typedef struct stretch {
    int num_chans;                      /* 1 or 2 */
    int inbuff_samples;                 /* capacity of the input ring, in values (frames x channels) */
    int shortest, longest;              /* period bounds, in values */
    int tail, head;                     /* read mark / fill level of the ring after the last call (mirrors of the device state) */
    int fast_mode;
    artsample_t *inbuff, *calcbuff, *results;   /* unused on the host (the buffers live in HBM); kept for layout */
    double outsamples_error;            /* accumulated output-length error, in values (mirror) */
    struct stretch *next;               /* second stage when STRETCH_DUAL_FLAG */
    artsample_t *intermediate;          /* unused on the host */
    struct artamd_stretch *hip;
} Stretch;

#ifdef __cplusplus
extern "C" {
#endif

/* periods in frames (ART: rate/350 and rate/50, art.c:787); NULL for invalid periods or when no HIP device is present */
Stretch *stretchInit (int shortest_period, int longest_period, int num_channels, int flags);

/* frames (per channel) `output` must hold for calls of at most max_num_samples frames at ratios up to max_ratio */
int stretchGetOutputCapacity (Stretch *cxt, int max_num_samples, double max_ratio);

/* buffer num_samples frames and emit what can be emitted at `ratio` (output/input length; clipped to 0.5..2, or
 * 0.25..4 in dual mode); returns the frames written to `output`.  Host pointers. */
int stretchProcess (Stretch *cxt, const artsample_t *samples, int num_samples, artsample_t *output, double ratio);

/* emit what is still buffered at normal speed; call until it returns 0 */
int stretchFlush (Stretch *cxt, artsample_t *output);

void stretchReset (Stretch *cxt);
void stretchFree (Stretch *cxt);

#ifdef __cplusplus
}
#endif
#endif
