/* pcm_host.c — host (C) side of the biquad and decimator entry points.
 *
 * Host work: coefficient design (reference biquad.c:18-74, decimator.c:28-97, :389-409) and moving
 * caller buffers to/from HBM.  Every sample is processed by the kernels in pcm_kernels.hip.
 */
#define _USE_MATH_DEFINES
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "art_internal.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------
 * Biquad design
 * ---------------------------------------------------------------------------------------- */

static void butterworth (double freq, double *K_out, double *norm_out, double *b1, double *b2)
{
    const double Q = sqrt (0.5), K = tan (M_PI * freq);
    const double norm = 1.0 / (1.0 + K / Q + K * K);
    *K_out = K; *norm_out = norm;
    *b1 = 2.0 * (K * K - 1.0) * norm;
    *b2 = (1.0 - K / Q + K * K) * norm;
}

void biquad_lowpass (BiquadCoefficients *filter, double frequency)
{
    double K, norm, b1, b2;
    butterworth (frequency, &K, &norm, &b1, &b2);
    memset (filter, 0, sizeof (*filter));
    filter->a0 = (art_s)(K * K * norm);
    filter->a1 = (art_s)(2 * filter->a0);       /* doubled AFTER rounding to art_s (reference biquad.c:26) */
    filter->a2 = filter->a0;
    filter->b1 = (art_s) b1;
    filter->b2 = (art_s) b2;
}

void biquad_highpass (BiquadCoefficients *filter, double frequency)
{
    double K, norm, b1, b2;
    butterworth (frequency, &K, &norm, &b1, &b2);
    memset (filter, 0, sizeof (*filter));
    filter->a0 = (art_s) norm;
    filter->a1 = (art_s)(-2.0 * norm);
    filter->a2 = filter->a0;
    filter->b1 = (art_s) b1;
    filter->b2 = (art_s) b2;
}

void biquad_init (Biquad *f, const BiquadCoefficients *c, double gain)
{
    memset (f, 0, sizeof (*f));
    f->a [0] = (art_s)(c->a0 * gain); f->a [1] = (art_s)(c->a1 * gain); f->a [2] = (art_s)(c->a2 * gain);
    f->a [3] = (art_s)(c->a3 * gain); f->a [4] = (art_s)(c->a4 * gain);
    f->b [1] = c->b1; f->b [2] = c->b2; f->b [3] = c->b3; f->b [4] = c->b4;
    f->order = (c->a4 != 0.0F || c->b4 != 0.0F) ? 4 : (c->a3 != 0.0F || c->b3 != 0.0F) ? 3 :
               (c->a2 != 0.0F || c->b2 != 0.0F) ? 2 : 1;
}

/* ------------------------------------------------------------------------------------------
 * Biquad, host-pointer entry points: a process-wide scratch in HBM, one section, one lane.
 * (Serial by nature; the batched device form below is the one meant for throughput.)
 * ---------------------------------------------------------------------------------------- */

static pthread_mutex_t scratch_lock = PTHREAD_MUTEX_INITIALIZER;
static art_s *scratch_buf; static size_t scratch_cap;
static Biquad *scratch_state;

static int scratch_reserve (size_t samples)
{
    if (!scratch_state && !(scratch_state = arthip_malloc (sizeof (Biquad)))) return -1;
    if (samples * sizeof (art_s) > scratch_cap) {
        arthip_free (scratch_buf);
        scratch_cap = samples * sizeof (art_s) * 2;
        if (!(scratch_buf = arthip_malloc (scratch_cap))) { scratch_cap = 0; return -1; }
    }
    return 0;
}

static void biquad_run_host (Biquad *f, art_s *buffer, int n, int stride, int sample_form)
{
    if (n <= 0) return;
    const size_t span = (size_t)(n - 1) * stride + 1;

    pthread_mutex_lock (&scratch_lock);
    if (arthip_device_count () < 1 || scratch_reserve (span)) {
        fprintf (stderr, "artamd: biquad needs a HIP device (no CPU path): %s\n", arthip_last_error ());
        pthread_mutex_unlock (&scratch_lock);
        abort ();
    }
    arthip_h2d (scratch_buf, buffer, span * sizeof (art_s), NULL);
    arthip_h2d (scratch_state, f, sizeof (Biquad), NULL);
    /* second-order section over a long strided run (ART's -p filters, art.c:1011-1016): the feed-forward / pipelined
     * kernel with one channel; everything else: the generic one-lane kernel */
    if (!sample_form && f->order == 2 && n >= 64)
        arthip_biquad_order2 (scratch_state, 1, 1, scratch_buf, n, stride, NULL);
    else
        arthip_biquad_chain (scratch_state, 1, 1, scratch_buf, n, sample_form ? -stride : stride, NULL);
    arthip_d2h (buffer, scratch_buf, span * sizeof (art_s), NULL);
    arthip_d2h (f, scratch_state, sizeof (Biquad), NULL);
    arthip_sync (NULL);
    pthread_mutex_unlock (&scratch_lock);
}

void biquad_apply_buffer (Biquad *f, artsample_t *buffer, int num_samples, int stride)
{
    biquad_run_host (f, buffer, num_samples, stride, 0);
}

artsample_t biquad_apply_sample (Biquad *f, artsample_t input)
{
    biquad_run_host (f, &input, 1, 1, 1);
    return input;
}

/* ---- device-resident bank of section chains ---- */

struct artamd_biquad_bank {
    Biquad *d_sections;
    int C, S;
    int all_order2;                      /* every section is second order: hand-scheduled kernel */
    void *stream;
};

BiquadBank *biquadBankCreate (const Biquad *sections, int numChannels, int numSections)
{
    if (numChannels < 1 || numSections < 1 || numSections > 4 || arthip_device_count () < 1) {
        fprintf (stderr, "artamd: biquadBankCreate: need 1-4 sections, >=1 channel and a HIP device\n");
        return NULL;
    }
    BiquadBank *b = calloc (1, sizeof (*b));
    const size_t bytes = sizeof (Biquad) * (size_t) numChannels * numSections;
    b->C = numChannels; b->S = numSections;
    b->all_order2 = numSections <= 2;
    for (int i = 0; i < numChannels * numSections; ++i)
        if (sections [i].order != 2) b->all_order2 = 0;
    b->d_sections = arthip_malloc (bytes);
    if (!b->d_sections || arthip_h2d (b->d_sections, sections, bytes, NULL) || arthip_sync (NULL)) { biquadBankFree (b); return NULL; }
    return b;
}

void biquadBankSetStream (BiquadBank *b, void *stream) { b->stream = stream; }

void biquadBankApplyInterleavedDevice (BiquadBank *b, artsample_t *d_buffer, int numFrames)
{
    if (b->all_order2 && numFrames >= 64)
        arthip_biquad_order2 (b->d_sections, b->C, b->S, d_buffer, numFrames, b->C, b->stream);
    else
        arthip_biquad_chain (b->d_sections, b->C, b->S, d_buffer, numFrames, b->C, b->stream);
}

void biquadBankRead (BiquadBank *b, Biquad *sections)
{
    arthip_d2h (sections, b->d_sections, sizeof (Biquad) * (size_t) b->C * b->S, b->stream);
    arthip_sync (b->stream);
}

void biquadBankFree (BiquadBank *b)
{
    if (b) { arthip_free (b->d_sections); free (b); }
}

/* ------------------------------------------------------------------------------------------
 * Decimator
 * ---------------------------------------------------------------------------------------- */

struct artamd_decimator {
    void *stream;
    art_s *d_feedback; uint32_t *d_gens, *d_gens_alt; Biquad *d_shapers;
    unsigned long long *d_clipped;
    unsigned long long clipped_seen;
    art_s *d_in; size_t in_cap;
    unsigned char *d_out; size_t out_cap;
};

/* noise-shaping transfer function N(z) (a0 == 1) -> error-feedback filter H(z), reference decimator.c:389-409 */
static void shaper_design (Biquad *f, double a1, double a2, double a3, double a4, double b1, double b2, double b3, double b4)
{
    BiquadCoefficients c;
    memset (&c, 0, sizeof (c));
    c.a0 = (art_s)(b1 - a1); c.a1 = (art_s)(b2 - a2); c.a2 = (art_s)(b3 - a3); c.a3 = (art_s)(b4 - a4);
    c.b1 = (art_s) b1; c.b2 = (art_s) b2; c.b3 = (art_s) b3; c.b4 = (art_s) b4;
    biquad_init (f, &c, 1.0);
}

static void shaper_for (Biquad *f, int flags, int rate)
{
    if (flags & SHAPING_ATH_CURVE) {
        switch (rate) {     /* ATH-curve shapers (coefficients are data of the reference, decimator.c:68-78) */
            case 32000: shaper_design (f, -0.780459, +0.569358, -0.348221, +0.466316, +0.950797, +0.282052, +0.004337, +1.76209e-5); return;
            case 44100: shaper_design (f, -1.1474, 0.5383, -0.3530, 0.3475, 1.0587, 0.0676, -0.6054, -0.2738); return;
            case 48000: shaper_design (f, -1.3344, 0.7455, -0.4602, 0.4363, 0.9030, 0.0116, -0.5853, -0.2571); return;
            case 88200: shaper_design (f, -2.150679, +2.1402057, -1.042712, +0.206838, +0.67433, +1.017047, +0.4028633, +0.098656); return;
            case 96000: shaper_design (f, -2.16994, +2.01986, -0.894857, +0.1557738, +0.517789, +1.1062189, +0.4825786, +0.244994); return;
            default:    shaper_design (f, -1.0, 0, 0, 0, 0, 0, 0, 0); return;
        }
    }
    if (flags & SHAPING_1ST_ORDER) shaper_design (f, -1.0, 0, 0, 0, 0, 0, 0, 0);
    else if (flags & SHAPING_2ND_ORDER) shaper_design (f, -2.0, +1.0, 0, 0, 0, 0, 0, 0);
    else if (flags & SHAPING_3RD_ORDER) shaper_design (f, -3.0, +3.0, -1.0, 0, 0, 0, 0, 0);
}

static uint32_t lcg_step (uint32_t r) { return ((r << 4) - r) ^ 1; }

Decimate *decimateInit (int numChannels, int outputBits, int outputBytes, double outputGain, int sampleRate, int flags)
{
    if (numChannels < 1 || outputBits < 1 || outputBits > 24 || outputBytes < (outputBits + 7) / 8 || outputBytes > 4) {
        fprintf (stderr, "artamd: decimateInit: unsupported channel/bit/byte combination\n");
        return NULL;
    }
    if (arthip_device_count () < 1) {
        fprintf (stderr, "artamd: no usable HIP device (this library has no CPU path): %s\n", arthip_last_error ());
        return NULL;
    }

    Decimate *cxt = calloc (1, sizeof (Decimate));
    struct artamd_decimator *hip = calloc (1, sizeof (*hip));
    const int C = numChannels;

    cxt->hip = hip;
    cxt->numChannels = C; cxt->outputBits = outputBits; cxt->outputBytes = outputBytes;
    cxt->outputGain = outputGain; cxt->flags = flags;
    cxt->feedback = calloc (C, sizeof (art_s));
    hip->d_feedback = arthip_malloc (sizeof (art_s) * C);
    hip->d_clipped = arthip_malloc (sizeof (unsigned long long));
    arthip_zero (hip->d_feedback, sizeof (art_s) * C, NULL);
    arthip_zero (hip->d_clipped, sizeof (unsigned long long), NULL);

    if (flags & DITHER_ENABLED) {
        /* per-channel seeds: little-endian words cut from the byte stream (state >> 24), three steps per byte */
        uint32_t s = 0x31415926;
        cxt->tpdf_generators = calloc (C, sizeof (uint32_t));
        for (int c = 0; c < C; ++c)
            for (int b = 0; b < 4; ++b) {
                cxt->tpdf_generators [c] |= (uint32_t)(s >> 24) << (8 * b);
                s = lcg_step (lcg_step (lcg_step (s)));
            }
        cxt->dither_type = (flags & DITHER_HIGHPASS) ? -1 : (flags & DITHER_LOWPASS) ? 1 : 0;
        hip->d_gens = arthip_malloc (sizeof (uint32_t) * C);
        hip->d_gens_alt = arthip_malloc (sizeof (uint32_t) * C);
        arthip_h2d (hip->d_gens, cxt->tpdf_generators, sizeof (uint32_t) * C, NULL);
    }

    if (flags & SHAPING_ENABLED) {
        cxt->noise_shapers = calloc (C, sizeof (Biquad));
        for (int c = 0; c < C; ++c)
            shaper_for (cxt->noise_shapers + c, flags, sampleRate);
        hip->d_shapers = arthip_malloc (sizeof (Biquad) * C);
        arthip_h2d (hip->d_shapers, cxt->noise_shapers, sizeof (Biquad) * C, NULL);
    }

    if (arthip_sync (NULL) || !hip->d_feedback || !hip->d_clipped) {
        fprintf (stderr, "artamd: decimateInit: device setup failed: %s\n", arthip_last_error ());
        decimateFree (cxt);
        return NULL;
    }

    return cxt;
}

void decimateFree (Decimate *cxt)
{
    if (!cxt) return;
    struct artamd_decimator *hip = cxt->hip;
    if (hip) {
        arthip_sync (hip->stream);
        arthip_free (hip->d_feedback); arthip_free (hip->d_gens); arthip_free (hip->d_gens_alt); arthip_free (hip->d_shapers);
        arthip_free (hip->d_clipped); arthip_free (hip->d_in); arthip_free (hip->d_out);
        free (hip);
    }
    free (cxt->feedback); free (cxt->tpdf_generators); free (cxt->noise_shapers);
    free (cxt);
}

static void dec_args (Decimate *cxt, ArtDecArgs *a)
{
    struct artamd_decimator *hip = cxt->hip;
    a->C = cxt->numChannels; a->bits = cxt->outputBits; a->bytes = cxt->outputBytes;
    a->dither_type = cxt->dither_type;
    a->dither_on = (cxt->flags & DITHER_ENABLED) != 0;
    a->shaping_on = (cxt->flags & SHAPING_ENABLED) != 0;
    a->shaping_order = (a->shaping_on && cxt->noise_shapers) ? cxt->noise_shapers [0].order : 0;
    a->scale = (art_s)((1 << cxt->outputBits) / 2.0 * cxt->outputGain);
    a->feedback = hip->d_feedback; a->gens = hip->d_gens; a->gens_next = hip->d_gens_alt; a->shapers = hip->d_shapers; a->clipped = hip->d_clipped;
}

void decimateHipSetStream (Decimate *cxt, void *stream) { cxt->hip->stream = stream; }

static void dec_swap_if (Decimate *cxt, int rc)
{
    if (rc == 1) { uint32_t *t = cxt->hip->d_gens; cxt->hip->d_gens = cxt->hip->d_gens_alt; cxt->hip->d_gens_alt = t; }
}

void decimateProcessInterleavedLEDevice (Decimate *cxt, const artsample_t *d_input, int numInputFrames, unsigned char *d_output)
{
    ArtDecArgs a;
    dec_args (cxt, &a);
    dec_swap_if (cxt, arthip_decimate (&a, d_input, numInputFrames, d_output, cxt->hip->stream));
}

long decimateHipClipped (Decimate *cxt)
{
    unsigned long long total = 0;
    arthip_d2h (&total, cxt->hip->d_clipped, sizeof (total), cxt->hip->stream);
    arthip_sync (cxt->hip->stream);
    return (long) total;
}

static int dec_reserve (Decimate *cxt, size_t in_bytes, size_t out_bytes)
{
    struct artamd_decimator *hip = cxt->hip;
    if (in_bytes > hip->in_cap) { arthip_free (hip->d_in); hip->in_cap = in_bytes * 2; if (!(hip->d_in = arthip_malloc (hip->in_cap))) { hip->in_cap = 0; return -1; } }
    if (out_bytes > hip->out_cap) { arthip_free (hip->d_out); hip->out_cap = out_bytes * 2; if (!(hip->d_out = arthip_malloc (hip->out_cap))) { hip->out_cap = 0; return -1; } }
    return 0;
}

/* after a host-pointer call: return this call's clip count and refresh the host-visible state mirrors */
static int dec_finish (Decimate *cxt)
{
    struct artamd_decimator *hip = cxt->hip;
    const int C = cxt->numChannels;
    unsigned long long total = 0;

    arthip_d2h (&total, hip->d_clipped, sizeof (total), hip->stream);
    arthip_d2h (cxt->feedback, hip->d_feedback, sizeof (art_s) * C, hip->stream);
    if (cxt->tpdf_generators) arthip_d2h (cxt->tpdf_generators, hip->d_gens, sizeof (uint32_t) * C, hip->stream);
    if (cxt->noise_shapers) arthip_d2h (cxt->noise_shapers, hip->d_shapers, sizeof (Biquad) * C, hip->stream);
    arthip_sync (hip->stream);

    int delta = (int)(total - hip->clipped_seen);
    hip->clipped_seen = total;
    return delta;
}

int decimateProcessInterleavedLE (Decimate *cxt, const artsample_t *input, int numInputFrames, unsigned char *output)
{
    struct artamd_decimator *hip = cxt->hip;
    if (numInputFrames <= 0) return 0;
    const size_t samples = (size_t) numInputFrames * cxt->numChannels;
    ArtDecArgs a;

    if (dec_reserve (cxt, samples * sizeof (art_s), samples * cxt->outputBytes)) {
        fprintf (stderr, "artamd: decimator device allocation failed: %s\n", arthip_last_error ());
        return 0;
    }
    dec_args (cxt, &a);
    arthip_h2d (hip->d_in, input, samples * sizeof (art_s), hip->stream);
    dec_swap_if (cxt, arthip_decimate (&a, hip->d_in, numInputFrames, hip->d_out, hip->stream));
    arthip_d2h (output, hip->d_out, samples * cxt->outputBytes, hip->stream);
    return dec_finish (cxt);
}

int decimateProcessLE (Decimate *cxt, const artsample_t *const *input, int numInputFrames, unsigned char *const *output)
{
    struct artamd_decimator *hip = cxt->hip;
    if (numInputFrames <= 0) return 0;
    const int C = cxt->numChannels;
    const size_t n = (size_t) numInputFrames, plane_bytes = n * cxt->outputBytes;
    ArtDecArgs a;

    if (dec_reserve (cxt, n * C * sizeof (art_s), plane_bytes * C)) {
        fprintf (stderr, "artamd: decimator device allocation failed: %s\n", arthip_last_error ());
        return 0;
    }
    dec_args (cxt, &a);
    for (int c = 0; c < C; ++c)
        arthip_h2d (hip->d_in + n * c, input [c], n * sizeof (art_s), hip->stream);
    arthip_decimate_planar (&a, hip->d_in, (long) n, numInputFrames, hip->d_out, (long) plane_bytes, hip->stream);
    for (int c = 0; c < C; ++c)
        arthip_d2h (output [c], hip->d_out + plane_bytes * c, plane_bytes, hip->stream);
    return dec_finish (cxt);
}

/* ------------------------------------------------------------------------------------------
 * Integer -> float ingest
 * ---------------------------------------------------------------------------------------- */

static art_s ingest_gain (double gain, int bits)
{
    return bits <= 8 ? (art_s)(gain / 128.0) : bits <= 16 ? (art_s)(gain / 32768.0) : (art_s)(gain / 8388608.0);
}

void floatIntegersLEDevice (const unsigned char *d_input, double inputGain, int inputBits, int inputBytes, int inputStride,
                            artsample_t *d_output, int numSamples, void *stream)
{
    if (inputBits > 24) return;
    arthip_ingest (d_input, ingest_gain (inputGain, inputBits), inputBits, inputBytes, inputStride, d_output, numSamples, stream);
}

void floatIntegersLE (unsigned char *input, double inputGain, int inputBits, int inputBytes, int inputStride, artsample_t *output, int numSamples)
{
    if (numSamples <= 0 || inputBits > 24) return;
    const size_t in_bytes = (size_t) numSamples * inputStride * inputBytes;
    unsigned char *d_in = arthip_malloc (in_bytes);
    art_s *d_out = arthip_malloc (sizeof (art_s) * (size_t) numSamples);

    if (!d_in || !d_out) {
        fprintf (stderr, "artamd: floatIntegersLE needs a HIP device (no CPU path): %s\n", arthip_last_error ());
        abort ();
    }
    /* the last sample's trailing stride bytes may not exist in the caller's buffer */
    const size_t valid = in_bytes - (size_t)(inputStride - 1) * inputBytes;
    arthip_h2d (d_in, input, valid, NULL);
    floatIntegersLEDevice (d_in, inputGain, inputBits, inputBytes, inputStride, d_out, numSamples, NULL);
    arthip_d2h (output, d_out, sizeof (art_s) * (size_t) numSamples, NULL);
    arthip_sync (NULL);
    arthip_free (d_in); arthip_free (d_out);
}
