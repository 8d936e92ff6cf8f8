/* resampler.h — MI355X-native sinc resampler, C API.
 *
 * Drop-in boundary for the reference's resampler.h: the same 15 entry points
 * (reference resampler.h:64-78), the same flag values (reference resampler.h:28-38), the same
 * by-value ResampleResult (reference resampler.h:40-42) and a `Resample` whose leading fields
 * have the reference's layout (reference resampler.h:44-48), so that a caller compiled against
 * either header (ART art.c:827-1136, artest.c:385-594) links and runs unmodified.
 *
 * All arithmetic runs in hand-written gfx950 HIP kernels (audio_resampler_amd/csrc/sinc_fir.hip);
 * the host side (audio_resampler_amd/csrc/resampler_host.c) only designs the filter bank and
 * replays the scalar position state machine.  There is NO CPU fallback: every process call
 * needs a GPU and the init functions return NULL (message on stderr) without one.
 *
 * Device-pointer / stream extensions live in art_hip.h.
 */
#ifndef ARTAMD_RESAMPLER_H
#define ARTAMD_RESAMPLER_H

#include <stdint.h>

#if defined(PATH_WIDTH) && (PATH_WIDTH==64)
#error "the MI355X build implements the 32-bit float sample path only (SURVEY.md 8(f) rank 3)"
#endif
#ifndef ARTSAMPLE_T_DEFINED
#define ARTSAMPLE_T_DEFINED
typedef float artsample_t;
#endif

/* flag bits — numerically identical to the reference (ABI) */
#define SUBSAMPLE_INTERPOLATE   0x1
#define BLACKMAN_HARRIS         0x2
#define INCLUDE_LOWPASS         0x4
#define RESAMPLE_MULTITHREADED  0x8      /* accepted, no effect: channels are already parallel on the GPU */
#define NO_FILTER_REDUCTION     0x10
#define RESAMPLE_FIXED_RATIO    0x20     /* internal */
#define EXTRAPOLATE_ENDPOINTS   0x40     /* end-point LPC extrapolation (fit on the host, samples consumed on the GPU) */
#define EXTRAPOLATE_PREFILL     0x80     /* internal */
#define EXTEND_CONVOLUTION_MATH 0x100    /* fp64 accumulation in the FIR */
#define RESAMPLER_FLUSHED       0x200    /* internal */
#define RESAMPLER_SNAP_OFFSET   0x400    /* internal */
/* extension (no reference equivalent; also selectable with env ARTAMD_STRICT=1): evaluate the FIR
 * in the reference's C source order, un-fused — bit-identical to the reference built with
 * `-O2 -ffp-contract=off`.  Slow; meant for parity runs. */
#define RESAMPLE_STRICT_ORDER   0x10000

typedef struct {
    unsigned int input_used, output_generated;
} ResampleResult;

struct artamd_resampler;                 /* private device-side state */

typedef struct resample {
    /* ---- reference-layout prefix (reference resampler.h:45-48); numChannels is read by callers */
    int numChannels, numSamples, numFilters, numTaps, inputIndex, flags;
    double *tempFilter;                  /* always NULL */
    double outputOffset, fixedRatio, lowpassRatio;
    void *subsample;                     /* always NULL: evaluation happens on the device */
    artsample_t **buffers;               /* always NULL: sample history lives in HBM */
    artsample_t **filters;               /* host copy of the (numFilters+1) x numTaps bank rows */
    /* ---- private */
    struct artamd_resampler *hip;
} Resample;

#ifdef __cplusplus
extern "C" {
#endif

Resample *resampleInit (int numChannels, int numTaps, int numFilters, double lowpassRatio, int flags);
Resample *resampleFixedRatioInit (int numChannels, int numTaps, int maxFilters, double sourceRate, double destinRate, int lowpassFreq, int flags);
ResampleResult resampleProcess (Resample *cxt, const artsample_t *const *input, int numInputFrames, artsample_t *const *output, int numOutputFrames, double ratio);
ResampleResult resampleProcessInterleaved (Resample *cxt, const artsample_t *input, int numInputFrames, artsample_t *output, int numOutputFrames, double ratio);
ResampleResult resampleProcessAndFlush (Resample *cxt, const artsample_t *const *input, int numInputFrames, artsample_t *const *output, int numOutputFrames, double ratio);
ResampleResult resampleProcessAndFlushInterleaved (Resample *cxt, const artsample_t *input, int numInputFrames, artsample_t *output, int numOutputFrames, double ratio);
unsigned int resampleGetRequiredSamples (Resample *cxt, int numOutputFrames, double ratio);
unsigned int resampleGetExpectedOutput (Resample *cxt, int numInputFrames, double ratio);
void resampleAdvancePosition (Resample *cxt, double delta);
double resampleGetLowpassRatio (Resample *cxt);
double resampleGetPosition (Resample *cxt);
int resampleGetNumFilters (Resample *cxt);
int resampleInterpolationUsed (Resample *cxt);
void resampleReset (Resample *cxt);
void resampleFree (Resample *cxt);

#ifdef __cplusplus
}
#endif
#endif
