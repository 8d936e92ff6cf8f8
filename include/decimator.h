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

/* `flags` of decimateInit: at most one dither shape, at most one shaping choice.  Values are ABI. */
enum {
    DITHER_HIGHPASS        = 0x0001,     /* TPDF with negative inter-sample correlation */
    DITHER_FLAT            = 0x0002,     /* independent TPDF */
    DITHER_LOWPASS         = 0x0004,     /* TPDF with positive inter-sample correlation */
    SHAPING_1ST_ORDER      = 0x0100,
    SHAPING_2ND_ORDER      = 0x0200,
    SHAPING_3RD_ORDER      = 0x0400,
    SHAPING_ATH_CURVE      = 0x0800,     /* threshold-of-hearing curves for 32/44.1/48/88.2/96 kHz, else 1st order */
    DECIMATE_MULTITHREADED = 0x1000      /* the reference: one worker thread per channel (decimator.c:92-93, 119-136); here: the
                                          * channels spread over the devices of artamdSetDevices () / ARTAMD_DEVICES (art_hip.h),
                                          * one ordinary context per contiguous channel slice — same bytes, clip counts and state */
};
#define DITHER_ENABLED   (DITHER_HIGHPASS | DITHER_FLAT | DITHER_LOWPASS)
#define SHAPING_ENABLED  (SHAPING_1ST_ORDER | SHAPING_2ND_ORDER | SHAPING_3RD_ORDER | SHAPING_ATH_CURVE)

struct artamd_decimator;

typedef struct {
    /* reference-layout prefix (reference decimator.h:42-47) */
    int numChannels, outputBits, outputBytes, dither_type, flags;
    double outputGain;
    artsample_t *feedback;               /* host mirrors of the device state, refreshed after every host-pointer call */
    uint32_t *tpdf_generators;
    Biquad *noise_shapers;
    /* private */
    struct artamd_decimator *hip;
} Decimate;

#ifdef __cplusplus
extern "C" {
#endif

/* outputBits 1..24 significant bits, left-justified in outputBytes little-endian bytes (8-bit output is
 * offset-binary); outputGain is applied before quantisation; sampleRate selects the ATH curve */
Decimate *decimateInit (int numChannels,
                        int outputBits,
                        int outputBytes,
                        double outputGain,
                        int sampleRate,
                        int flags);
void decimateFree (Decimate *cxt);

/* both return the number of samples that had to be clipped */
int decimateProcessInterleavedLE (Decimate *cxt,
                                  const artsample_t *input, int numInputFrames,
                                  unsigned char *output);
int decimateProcessLE (Decimate *cxt,
                       const artsample_t *const *input, int numInputFrames,
                       unsigned char *const *output);

/* the inverse direction: little-endian integers (inputBits in inputBytes, inputStride samples apart) to float */
void floatIntegersLE (unsigned char *input,
                      double inputGain,
                      int inputBits, int inputBytes, int inputStride,
                      artsample_t *output, int numSamples);

#ifdef __cplusplus
}
#endif
#endif
