/* biquad.h — 1st..4th-order direct-form-I IIR sections, C API.
 *
 * Drop-in boundary for the reference's biquad.h (reference biquad.h:27-47): BiquadCoefficients
 * (36 bytes) and Biquad (80 bytes) are caller-allocated PODs whose layout is ABI
 * (ART allocates them itself, art.c:726-729, 869-870).
 *
 * The design functions are host code.  biquad_apply_buffer / biquad_apply_sample run the recurrence in a
 * gfx950 kernel (one lane per section chain, un-fused float ops in the reference's order => bit-identical to
 * the reference built with -ffp-contract=off).  Batched, device-resident forms are in art_hip.h.
 */
#ifndef ARTAMD_BIQUAD_H
#define ARTAMD_BIQUAD_H

#include <stdint.h>

#ifndef ARTSAMPLE_T_DEFINED
#define ARTSAMPLE_T_DEFINED
#if defined(PATH_WIDTH) && (PATH_WIDTH==64)
typedef double artsample_t;
#else
typedef float artsample_t;
#endif
#endif

/* transfer function (a0 + a1 z^-1 + ... + a4 z^-4) / (1 + b1 z^-1 + ... + b4 z^-4) */
typedef struct {
    artsample_t a0, a1, a2, a3, a4;
    artsample_t b1, b2, b3, b4;
} BiquadCoefficients;

/* one running section: coefficients (gain folded into a[]), four-deep circular input/output history */
typedef struct {
    artsample_t a[5], b[5];
    artsample_t x[4], y[4];
    int order, index;
} Biquad;

#ifdef __cplusplus
extern "C" {
#endif

/* design: Butterworth-Q second-order sections, `frequency` as a fraction of the sample rate */
void biquad_lowpass (BiquadCoefficients *filter, double frequency);
void biquad_highpass (BiquadCoefficients *filter, double frequency);

/* clear the state, fold `gain` into the numerator, derive the order from the non-zero coefficients */
void biquad_init (Biquad *f, const BiquadCoefficients *coeffs, double gain);

/* filter num_samples values in place, stepping `stride` samples between them (interleaved channels) */
void biquad_apply_buffer (Biquad *f, artsample_t *buffer, int num_samples, int stride);

/* one value through the section (the association the decimator's noise shaper uses) */
artsample_t biquad_apply_sample (Biquad *f, artsample_t input);

#ifdef __cplusplus
}
#endif
#endif
