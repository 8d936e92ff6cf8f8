/* resampler.h — MI355X-native sinc resampler, C API.
 *
 * Drop-in boundary for the reference's resampler.h: the same 15 entry points
 * (reference resampler.h:64-78), the same flag values (reference resampler.h:28-38), the same
 * by-value ResampleResult (reference resampler.h:40-42) and a `Resample` whose leading fields
 * have the reference's layout (reference resampler.h:44-48), so that a caller compiled against
 * either header (ART art.c:827-1136, artest.c:385-594) links and runs unmodified.
 *
 * All arithmetic runs in hand-written gfx950 HIP kernels (audio_resampler_amd/csrc/fir_general.hip,
 * fir_matrix.hip, fir_matrix_i8.hip, fir_matrix64.hip; the launch rule is fir_dispatch.hip); the host
 * side (audio_resampler_amd/csrc/resampler_host.c) only designs the filter bank and replays the
 * scalar position state machine.  There is NO CPU fallback: every process call needs a GPU and the
 * init functions return NULL (message on stderr) without one.
 *
 * One deviation from the reference to know about.  The reference's fixed-ratio output (resampler.c:323-335,
 * 533-535) is bitwise independent of how the input is cut into calls.  In the DEFAULT mode of this library the size
 * of a call picks the kernel (general / f32 matrix cores, K split or not / fixed point on the integer matrix
 * cores), each of which rounds differently inside the parity bar: the same stream cut into other blocks gives
 * output within 2^-23 max(1,|y|) of the double-accumulate result either way — two cuts differ by at most twice
 * that — but not the same bits (tests/test_gpu_parity.py: bounded there; what that means in a 16- or 24-bit
 * file: tests/test_gpu_pcm_default_mode.py).  The reference's property is available three ways (art_hip.h):
 * RESAMPLE_STRICT_ORDER (the reference's own summation order, bit for bit, slow); the general kernel alone
 * (resampleHipSetKernel (cxt, 1)); and the cut-invariant stream policy, resampleHipSetCutInvariant (cxt, 1) /
 * ARTAMD_KERNEL=9 — one arithmetic per stream, the f32 matrix-core streaming kernel anchored on the stream's
 * canonical period for every launch of any size: the same bits for ANY cut into calls, host or device buffers
 * (tests/test_gpu_cut_invariance.py; its cost per call: DESIGN.md 4.1).  Counts and positions (input_used,
 * output_generated, resampleGetPosition) do not depend on the cut in any mode.
 *
 * Device-pointer / stream extensions live in art_hip.h.
 */
#ifndef ARTAMD_RESAMPLER_H
#define ARTAMD_RESAMPLER_H

#include <stdint.h>

/* PATH_WIDTH=64 selects the double-precision sample path (reference resampler.h:22-26); link libartamd64.so then */
#ifndef ARTSAMPLE_T_DEFINED
#define ARTSAMPLE_T_DEFINED
#if defined(PATH_WIDTH) && (PATH_WIDTH==64)
typedef double artsample_t;             /* one audio sample */
#else
typedef float artsample_t;              /* one audio sample */
#endif
#endif

/* Behaviour flags for the `flags` argument of the init calls.  Values are ABI (identical to the reference). */
enum {
    /* ---- what the caller asks for ---- */
    SUBSAMPLE_INTERPOLATE   = 0x001,    /* blend the two neighbouring phase filters (needed for arbitrary ratios) */
    BLACKMAN_HARRIS         = 0x002,    /* 4-term Blackman-Harris window; otherwise Hann */
    INCLUDE_LOWPASS         = 0x004,    /* fold a low-pass into the sinc (set automatically when lowpassRatio < 1) */
    RESAMPLE_MULTITHREADED  = 0x008,    /* spread the channels over the visible GPUs inside this one context (art_hip.h:
                                           artamdSetDevices / ARTAMD_DEVICES / ARTAMD_SHARDS); no effect with one device */
    NO_FILTER_REDUCTION     = 0x010,    /* fixed-ratio init: keep the full filter count (allows phase shifts) */
    EXTRAPOLATE_ENDPOINTS   = 0x040,    /* LPC-extrapolate before the first / after the last input sample */
    EXTEND_CONVOLUTION_MATH = 0x100,    /* fp64 accumulation in the FIR */
    /* ---- state bits the library keeps in Resample.flags (do not set) ---- */
    RESAMPLE_FIXED_RATIO    = 0x020,
    EXTRAPOLATE_PREFILL     = 0x080,
    RESAMPLER_FLUSHED       = 0x200,
    RESAMPLER_SNAP_OFFSET   = 0x400,
    /* ---- extension, no reference equivalent (also: environment ARTAMD_STRICT=1) ----
     * evaluate the FIR in the reference's C source order, un-fused: bit-identical to the reference built
     * with `-O2 -ffp-contract=off`.  Slow; meant for parity runs. */
    RESAMPLE_STRICT_ORDER   = 0x10000
};

/* frames consumed / produced by one process call */
typedef struct {
    unsigned int input_used, output_generated;
} ResampleResult;

struct artamd_resampler;                 /* private device-side state */

typedef struct resample {
    /* reference-layout prefix (reference resampler.h:45-48); callers read numChannels */
    int numChannels, numSamples, numFilters, numTaps, inputIndex, flags;
    double *tempFilter;                  /* always NULL */
    double outputOffset, fixedRatio, lowpassRatio;
    void *subsample;                     /* always NULL: evaluation happens on the device */
    artsample_t **buffers;               /* always NULL: sample history lives in HBM */
    artsample_t **filters;               /* host copy of the (numFilters+1) x numTaps bank rows */
    /* private */
    struct artamd_resampler *hip;
} Resample;

#ifdef __cplusplus
extern "C" {
#endif

/* ---- construction / destruction ------------------------------------------------------------- */

/* Arbitrary-ratio context.  taps: 4..1024, multiple of 4;  filters: 1..1024;
 * lowpassRatio: cutoff relative to the source Nyquist, (0,1) enables the low-pass, anything else disables it. */
Resample *resampleInit (int numChannels,
                        int numTaps,
                        int numFilters,
                        double lowpassRatio,
                        int flags);

/* Fixed-ratio context: when destinRate / gcd fits in maxFilters the bank shrinks to exactly the phases in use
 * and interpolation is dropped.  lowpassFreq in Hz, or 0 with INCLUDE_LOWPASS for an automatic cutoff when
 * downsampling.  The `ratio` argument of the process calls is ignored for such contexts. */
Resample *resampleFixedRatioInit (int numChannels,
                                  int numTaps,
                                  int maxFilters,
                                  double sourceRate,
                                  double destinRate,
                                  int lowpassFreq,
                                  int flags);

void resampleReset (Resample *cxt);      /* forget history and position (also re-arms a flushed context) */
void resampleFree (Resample *cxt);       /* NULL is fine */

/* ---- streaming ---------------------------------------------------------------------------------
 * Runs until the input is exhausted or numOutputFrames have been written, whichever comes first.
 * numInputFrames == -1 flushes: half a window of silence (or extrapolation) is appended; input may be NULL. */

/* planar: one pointer per channel */
ResampleResult resampleProcess (Resample *cxt,
                                const artsample_t *const *input, int numInputFrames,
                                artsample_t *const *output, int numOutputFrames,
                                double ratio);

/* interleaved: frame-major */
ResampleResult resampleProcessInterleaved (Resample *cxt,
                                           const artsample_t *input, int numInputFrames,
                                           artsample_t *output, int numOutputFrames,
                                           double ratio);

/* process, then — if all input was taken and room remains — flush into the tail of the same buffer */
ResampleResult resampleProcessAndFlush (Resample *cxt,
                                        const artsample_t *const *input, int numInputFrames,
                                        artsample_t *const *output, int numOutputFrames,
                                        double ratio);
ResampleResult resampleProcessAndFlushInterleaved (Resample *cxt,
                                                   const artsample_t *input, int numInputFrames,
                                                   artsample_t *output, int numOutputFrames,
                                                   double ratio);

/* ---- position and queries ---------------------------------------------------------------------- */

void   resampleAdvancePosition (Resample *cxt, double delta);   /* forward only; fractional only when interpolating */
double resampleGetPosition (Resample *cxt);                     /* in input samples; negative: an output is due */
unsigned int resampleGetRequiredSamples (Resample *cxt, int numOutputFrames, double ratio);   /* dry run */
unsigned int resampleGetExpectedOutput (Resample *cxt, int numInputFrames, double ratio);     /* dry run */
double resampleGetLowpassRatio (Resample *cxt);
int    resampleGetNumFilters (Resample *cxt);
int    resampleInterpolationUsed (Resample *cxt);

#ifdef __cplusplus
}
#endif
#endif
