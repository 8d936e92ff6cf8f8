/* decimator.h — float -> little-endian integer PCM with TPDF dither and noise shaping, C API.
 *
 * Drop-in boundary for the reference's decimator.h (reference decimator.h:29-71): same five entry
 * points, same flag values, and a `Decimate` whose leading fields have the reference's layout
 * (artest.c:660-666 reads numChannels and outputBytes).
 *
 * The per-sample pipeline (scale, dither, error feedback, round, shape, clip, pack — reference
 * decimator.c:255-283) runs in gfx950 kernels (audio_resampler_amd/csrc/pcm_kernels.hip) and is
 * bit-exact against the reference.  Device-pointer forms are in art_hip.h.
 */
#ifndef ARTAMD_DECIMATOR_H
#define ARTAMD_DECIMATOR_H

#include <stdint.h>
#include "biquad.h"

#define DITHER_HIGHPASS     0x1
#define DITHER_FLAT         0x2
#define DITHER_LOWPASS      0x4
#define DITHER_ENABLED      (DITHER_HIGHPASS | DITHER_FLAT | DITHER_LOWPASS)

#define SHAPING_1ST_ORDER   0x100
#define SHAPING_2ND_ORDER   0x200
#define SHAPING_3RD_ORDER   0x400
#define SHAPING_ATH_CURVE   0x800
#define SHAPING_ENABLED     (SHAPING_1ST_ORDER | SHAPING_2ND_ORDER | SHAPING_3RD_ORDER | SHAPING_ATH_CURVE)

#define DECIMATE_MULTITHREADED  0x1000   /* accepted, no effect */

struct artamd_decimator;

typedef struct {
    /* ---- reference-layout prefix (reference decimator.h:42-47) */
    int numChannels, outputBits, outputBytes, dither_type, flags;
    double outputGain;
    artsample_t *feedback;               /* host mirrors, refreshed after every host-pointer call */
    uint32_t *tpdf_generators;
    Biquad *noise_shapers;
    /* ---- private */
    struct artamd_decimator *hip;
} Decimate;

#ifdef __cplusplus
extern "C" {
#endif

void floatIntegersLE (unsigned char *input, double inputGain, int inputBits, int inputBytes, int inputStride, artsample_t *output, int numSamples);
Decimate *decimateInit (int numChannels, int outputBits, int outputBytes, double outputGain, int sampleRate, int flags);
int decimateProcessLE (Decimate *cxt, const artsample_t *const *input, int numInputFrames, unsigned char *const *output);
int decimateProcessInterleavedLE (Decimate *cxt, const artsample_t *input, int numInputFrames, unsigned char *output);
void decimateFree (Decimate *cxt);

#ifdef __cplusplus
}
#endif
#endif
