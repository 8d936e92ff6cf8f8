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

/* The void entry points of the reference (biquad_apply_*, floatIntegersLE) cannot report a failure, and there is no CPU path to
 * fall back to: a failure is printed, counted (artamdErrorCount / artamdLastError: tools check them before they trust the audio
 * they write) and, with ARTAMD_ABORT_ON_ERROR=1, fatal on the spot. */
static int pcm_errors;
static char pcm_last_error [256], pcm_last_error_out [256];
static pthread_mutex_t pcm_error_lock = PTHREAD_MUTEX_INITIALIZER;      /* (callers hold different locks: the message has its own) */
static void pcm_fail (const char *what)
{
    static int abort_on_error = -1;
    pthread_mutex_lock (&pcm_error_lock);
    if (abort_on_error < 0) { const char *e = getenv ("ARTAMD_ABORT_ON_ERROR"); abort_on_error = e && *e && *e != '0'; }
    snprintf (pcm_last_error, sizeof (pcm_last_error), "%s: %s", what, arthip_last_error ());
    fprintf (stderr, "artamd: %s\n", pcm_last_error);
    __atomic_add_fetch (&pcm_errors, 1, __ATOMIC_RELAXED);
    const int fatal = abort_on_error;
    pthread_mutex_unlock (&pcm_error_lock);
    if (fatal) abort ();
}
/* (the resampler's entry points report a failed launch the same way: resampler_host.c) */
void artamd_note_failure (const char *what) { pcm_fail (what); }
int artamdErrorCount (void) { return __atomic_load_n (&pcm_errors, __ATOMIC_RELAXED); }
const char *artamdLastError (void)
{
    if (!artamdErrorCount ()) return NULL;
    pthread_mutex_lock (&pcm_error_lock);               /* (a whole message, never a torn one) */
    memcpy (pcm_last_error_out, pcm_last_error, sizeof (pcm_last_error_out));
    pthread_mutex_unlock (&pcm_error_lock);
    return pcm_last_error_out;
}

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
 * How the cascades run.  A biquad is a serial recurrence through float rounding; the bit-exact form that is parallel over
 * TIME (pcm_kernels.hip, biquad_spec_kernel) needs to know how fast the recursive part forgets its state: the warm-up W
 * is the number of frames after which every unit initial state has decayed below 2^-40 (2^-70 for 8-byte samples; + margin).  Filters that do not
 * forget within SPEC_MAX_WARMUP frames (poles within ~0.97 of the unit circle) and short runs take the serial kernels.
 * Both forms produce the reference's bits (biquad.c:106-163); ARTAMD_BIQUAD_SERIAL=1 forces the serial one.
 * ---------------------------------------------------------------------------------------- */
#define SPEC_MAX_WARMUP 1024

static int forget_length (const Biquad *f)
{
    const int order = f->order;
    int worst = 0;

    if (order < 1 || order > 4) return 0;
    for (int j = 0; j < order; ++j) {
        double y [4] = { 0.0, 0.0, 0.0, 0.0 };
        int last = 0;
        y [j] = 1.0;
        for (int n = 1; n <= SPEC_MAX_WARMUP + 64; ++n) {
            double v = 0.0;
            for (int k = 1; k <= order; ++k) v -= (double) f->b [k] * y [k - 1];
            y [3] = y [2]; y [2] = y [1]; y [1] = y [0]; y [0] = v;
            if (!(fabs (v) <= (ART_WIDE ? 0x1p-70 : 0x1p-40))) last = n;        /* (also catches a blow-up: inf / NaN) */
        }
        if (last > worst) worst = last;
    }
    /* ARTAMD_BIQUAD_WARMUP=n forces the warm-up (tests: a warm-up far too short makes nearly every chunk boundary mismatch,
     * which exercises the verification and repair path; the results must not change) */
    { const char *e = getenv ("ARTAMD_BIQUAD_WARMUP"); if (e && *e) return atoi (e); }
    return worst > SPEC_MAX_WARMUP ? 0 : worst + 16;
}

/* chunk length for `sections` sections with warm-up W each: the warm-up is recomputed work, keep it to about a quarter */
static int spec_chunk (int sections, int W)
{
    int L = 4 * sections * W;
    L = (L + 7) & ~7;
    if (L < 128) L = 128;
    if (L > 8192) L = 8192;
    return L;
}

static int spec_disabled (void)
{
    static int cached = -1;
    if (cached < 0) { const char *e = getenv ("ARTAMD_BIQUAD_SERIAL"); cached = e && *e && *e != '0'; }
    return cached;
}

/* ------------------------------------------------------------------------------------------
 * Biquad, host-pointer entry points (what ART's -p filters call, art.c:1011-1017, :1052-1058: one section of one channel
 * of an interleaved buffer per call).  Only the channel's own samples travel: gathered by the CPU into page-locked
 * staging, one DMA each way, the time-parallel kernel on a dense run, scattered back.  Process-wide scratch.
 * ---------------------------------------------------------------------------------------- */

static pthread_mutex_t scratch_lock = PTHREAD_MUTEX_INITIALIZER;
static struct {
    art_s *d_in, *d_out; size_t cap;                   /* samples */
    art_s *h_buf; size_t h_cap;                        /* page-locked, samples */
    Biquad *d_state, *h_state;
    void *d_spec; size_t spec_cap;
    unsigned int *d_repairs;
    int *d_first_bad;
    int device;
} host_scratch = { .device = -1 };

static int scratch_reserve (size_t samples)
{
    const int device = arthip_current_device ();
    if (host_scratch.device != device) {                /* first use (or the caller moved to another GPU): start over there */
        arthip_free (host_scratch.d_in); arthip_free (host_scratch.d_out); arthip_free (host_scratch.d_state);
        arthip_free (host_scratch.d_spec); arthip_free (host_scratch.d_repairs); arthip_free (host_scratch.d_first_bad);
        arthip_host_free (host_scratch.h_buf); arthip_host_free (host_scratch.h_state);
        memset (&host_scratch, 0, sizeof (host_scratch));
        host_scratch.device = device;
    }
    if (!host_scratch.d_state && !(host_scratch.d_state = arthip_malloc (sizeof (Biquad)))) return -1;
    if (!host_scratch.h_state && !(host_scratch.h_state = arthip_host_alloc (sizeof (Biquad)))) return -1;
    if (!host_scratch.d_repairs) {
        if (!(host_scratch.d_repairs = arthip_malloc (sizeof (unsigned int)))) return -1;
        arthip_zero (host_scratch.d_repairs, sizeof (unsigned int), NULL);
    }
    if (!host_scratch.d_first_bad) {
        if (!(host_scratch.d_first_bad = arthip_malloc (sizeof (int)))) return -1;
        arthip_biquad_spec_arm (host_scratch.d_first_bad, 1, NULL);
    }
    if (samples > host_scratch.cap) {
        arthip_free (host_scratch.d_in); arthip_free (host_scratch.d_out);
        host_scratch.cap = samples + samples / 2 + 1024;
        host_scratch.d_in = arthip_malloc (host_scratch.cap * sizeof (art_s));
        host_scratch.d_out = arthip_malloc (host_scratch.cap * sizeof (art_s));
        if (!host_scratch.d_in || !host_scratch.d_out) { host_scratch.cap = 0; return -1; }
    }
    if (samples > host_scratch.h_cap) {
        arthip_host_free (host_scratch.h_buf);
        host_scratch.h_cap = samples + samples / 2 + 1024;
        if (!(host_scratch.h_buf = arthip_host_alloc (host_scratch.h_cap * sizeof (art_s)))) { host_scratch.h_cap = 0; return -1; }
    }
    return 0;
}

static void biquad_run_host (Biquad *f, art_s *buffer, int n, int stride, int sample_form)
{
    if (n <= 0) return;

    pthread_mutex_lock (&scratch_lock);
    if (arthip_device_count () < 1 || scratch_reserve ((size_t) n)) {
        /* no CPU evaluation path exists: say so (counted: artamdErrorCount) and leave the caller's samples and filter state untouched */
        pcm_fail ("biquad needs a HIP device and scratch memory (no CPU path)");
        pthread_mutex_unlock (&scratch_lock);
        return;
    }

    art_s *h = host_scratch.h_buf;
    if (stride == 1) memcpy (h, buffer, sizeof (art_s) * (size_t) n);
    else for (int i = 0; i < n; ++i) h [i] = buffer [(size_t) i * stride];
    *host_scratch.h_state = *f;
    /* samples and filter state travel in one launch each way (a copy kernel over the page-locked buffers; beyond a megabyte
     * the copy engines) */
    const int by_kernel = sizeof (art_s) * (size_t) n <= ((size_t) 1 << 20);
    if (by_kernel)
        arthip_copy2_by_kernel (host_scratch.d_in, h, sizeof (art_s) * (size_t) n, host_scratch.d_state, host_scratch.h_state, sizeof (Biquad), NULL);
    else {
        arthip_h2d (host_scratch.d_in, h, sizeof (art_s) * (size_t) n, NULL);
        arthip_h2d (host_scratch.d_state, host_scratch.h_state, sizeof (Biquad), NULL);
    }

    const int W = (sample_form || spec_disabled ()) ? 0 : forget_length (f);
    const int L = W ? spec_chunk (1, W) : 0;
    const art_s *d_result = host_scratch.d_in;
    int rc;
    if (W && n >= 2 * L) {
        const size_t need = arthip_biquad_spec_scratch (1, 1, n, L);
        if (need > host_scratch.spec_cap) {
            arthip_free (host_scratch.d_spec);
            host_scratch.d_spec = arthip_malloc (need + need / 2);
            host_scratch.spec_cap = host_scratch.d_spec ? need + need / 2 : 0;
        }
        rc = host_scratch.d_spec ? arthip_biquad_spec (host_scratch.d_state, 1, 1, host_scratch.d_in, 1, host_scratch.d_out, 1, n, L, W,
                                                       host_scratch.d_spec, host_scratch.d_first_bad, host_scratch.d_repairs, NULL) : -1;
        d_result = host_scratch.d_out;
    }
    else if (!sample_form && f->order == 2 && n >= 64)       /* long run of a narrow filter: the pipelined serial kernel, one channel */
        rc = arthip_biquad_order2 (host_scratch.d_state, 1, 1, host_scratch.d_in, n, 1, NULL);
    else
        rc = arthip_biquad_chain (host_scratch.d_state, 1, 1, host_scratch.d_in, n, sample_form ? -1 : 1, NULL);

    if (rc) pcm_fail ("biquad launch failed (samples left unfiltered)");
    else {
        if (by_kernel)
            arthip_copy2_by_kernel (h, d_result, sizeof (art_s) * (size_t) n, host_scratch.h_state, host_scratch.d_state, sizeof (Biquad), NULL);
        else {
            arthip_d2h (h, d_result, sizeof (art_s) * (size_t) n, NULL);
            arthip_d2h (host_scratch.h_state, host_scratch.d_state, sizeof (Biquad), NULL);
        }
        if (!arthip_sync (NULL)) {
            if (stride == 1) memcpy (buffer, h, sizeof (art_s) * (size_t) n);
            else for (int i = 0; i < n; ++i) buffer [(size_t) i * stride] = h [i];      /* only this channel's samples are written */
            *f = *host_scratch.h_state;
        }
        else pcm_fail ("biquad kernel failed (samples left unfiltered)");
    }
    pthread_mutex_unlock (&scratch_lock);
}

void biquad_apply_buffer (Biquad *f, artsample_t *buffer, int num_samples, int stride)
{
    biquad_run_host (f, buffer, num_samples, stride, 0);
}

/* One value through the section: the reference's per-sample association (biquad.c:78-102 — the highest delay first, each delay's
 * feed-forward and feedback products subtracted from one another before they join the sum), evaluated where the state lives: the
 * caller's struct, in host memory.  (Rounds 1-3 sent the struct and the sample to the GPU and back: one launch and 46 us per
 * sample for a dozen flops; runs of samples — biquad_apply_buffer, the decimator's noise shapers — stay on the device.  Compiled
 * -ffp-contract=off: the bits of the device kernels' sample form, tests/test_gpu_parity.py.) */
artsample_t biquad_apply_sample (Biquad *f, artsample_t input)
{
    art_s sum = input * f->a [0];
    int i = f->index & 3;

    for (int k = f->order > 4 ? 4 : f->order; k >= 1; --k) {
        const int d = (i - k + 1) & 3;
        sum += (f->x [d] * f->a [k]) - (f->b [k] * f->y [d]);
    }
    f->index = i = (i + 1) & 3;
    f->x [i] = input;
    f->y [i] = sum;
    return sum;
}

/* chunks the time-parallel biquad had to recompute since the process started (host-pointer calls; diagnostics) */
unsigned int artamdBiquadRepairs (void)
{
    unsigned int n = 0;
    pthread_mutex_lock (&scratch_lock);
    if (host_scratch.d_repairs) { arthip_d2h (&n, host_scratch.d_repairs, sizeof (n), NULL); arthip_sync (NULL); }
    pthread_mutex_unlock (&scratch_lock);
    return n;
}

/* ---- device-resident bank of section chains ---- */

struct artamd_biquad_bank {
    Biquad *d_sections;
    int C, S;
    int all_order2;                      /* every section is second order: hand-scheduled serial kernel */
    int warmup;                          /* time-parallel form: warm-up frames per section (0: serial kernels only) */
    art_s *d_tmp; size_t tmp_cap;        /* the call's input, moved aside (the time-parallel form is not in-place) */
    void *d_spec; size_t spec_cap;
    unsigned int *d_repairs;
    int *d_first_bad;
    void *stream;
    int device;
    /* a bank spread over several devices (biquadBankCreateMulti): ordinary banks with contiguous channel slices, each on its own
     * device and stream, + the slice of the caller's buffer each works on */
    int nshards; BiquadBank **shards; int *shard_first; void **ev_shard; void *ev_parent;
    art_s *d_slice; size_t slice_cap;
};
#define BANK_ENTER(b) const int prev_device_ = arthip_current_device (); \
                      if (prev_device_ != (b)->device) arthip_set_device ((b)->device)
#define BANK_LEAVE(b) do { if (prev_device_ != (b)->device && prev_device_ >= 0) arthip_set_device (prev_device_); } while (0)

BiquadBank *biquadBankCreate (const Biquad *sections, int numChannels, int numSections)
{
    if (numChannels < 1 || numSections < 1 || numSections > 4 || arthip_device_count () < 1) {
        fprintf (stderr, "artamd: biquadBankCreate: need 1-4 sections, >=1 channel and a HIP device\n");
        return NULL;
    }
    BiquadBank *b = calloc (1, sizeof (*b));
    if (!b) return NULL;
    const size_t bytes = sizeof (Biquad) * (size_t) numChannels * numSections;
    b->C = numChannels; b->S = numSections;
    b->device = arthip_current_device ();
    b->all_order2 = numSections <= 2;
    b->warmup = spec_disabled () ? 0 : 1;
    for (int i = 0; i < numChannels * numSections; ++i) {
        if (sections [i].order != 2) b->all_order2 = 0;
        if (b->warmup) {
            /* (channels of one stream share their coefficients: the search runs once per distinct section) */
            int w = (i >= numSections && !memcmp (sections [i].b, sections [i - numSections].b, sizeof (sections [i].b)) &&
                     sections [i].order == sections [i - numSections].order) ? b->warmup : forget_length (sections + i);
            b->warmup = !w ? 0 : w > b->warmup ? w : b->warmup;
        }
    }
    b->d_sections = arthip_malloc (bytes);
    b->d_repairs = arthip_malloc (sizeof (unsigned int));
    b->d_first_bad = arthip_malloc (sizeof (int) * (size_t) numChannels);
    if (!b->d_sections || !b->d_repairs || !b->d_first_bad || arthip_h2d (b->d_sections, sections, bytes, NULL) ||
        arthip_zero (b->d_repairs, sizeof (unsigned int), NULL) || arthip_biquad_spec_arm (b->d_first_bad, numChannels, NULL) ||
        arthip_sync (NULL)) { biquadBankFree (b); return NULL; }
    return b;
}

/* The same bank spread over the devices of artamdSetDevices () / ARTAMD_DEVICES (ARTAMD_SHARDS forces the count), the way a
 * RESAMPLE_MULTITHREADED resampler and a DECIMATE_MULTITHREADED decimator spread (channels are independent: art.c:1011-1017
 * filters them one by one): an ordinary bank when there is one device.  Results are those of the ordinary bank, bit for bit. */
BiquadBank *biquadBankCreateMulti (const Biquad *sections, int numChannels, int numSections)
{
    int devices [ART_MAX_DEVICES];
    const int home = arthip_current_device ();
    const int count = numChannels > 1 && arthip_device_count () >= 1 ? artamd_shard_plan (numChannels, home, devices) : 0;
    if (count <= 1) return biquadBankCreate (sections, numChannels, numSections);

    BiquadBank *b = calloc (1, sizeof (*b));
    if (!b) return NULL;
    b->C = numChannels; b->S = numSections; b->device = home;
    b->shards = calloc ((size_t) count, sizeof (BiquadBank *));
    b->shard_first = calloc ((size_t) count + 1, sizeof (int));
    b->ev_shard = calloc ((size_t) count, sizeof (void *));
    b->ev_parent = arthip_order_event_create ();
    int ok = b->shards && b->shard_first && b->ev_shard && b->ev_parent;
    const int base = numChannels / count, extra = numChannels % count;
    for (int s = 0; ok && s < count; ++s) {
        const int width = base + (s < extra ? 1 : 0);
        b->shard_first [s + 1] = b->shard_first [s] + width;
        arthip_set_device (devices [s]);
        b->shards [s] = biquadBankCreate (sections + (size_t) b->shard_first [s] * numSections, width, numSections);
        b->ev_shard [s] = arthip_order_event_create ();
        b->nshards = s + 1;
        ok = b->shards [s] && b->ev_shard [s] && (b->shards [s]->stream = arthip_stream_create ()) != NULL;
    }
    if (home >= 0) arthip_set_device (home);
    if (!ok) { fprintf (stderr, "artamd: biquadBankCreateMulti: allocation failed: %s\n", arthip_last_error ()); biquadBankFree (b); return NULL; }
    return b;
}

int biquadBankShardCount (BiquadBank *b) { return b->nshards; }

/* (work already enqueued on the old stream uses the bank's state and scratch: drained before the switch) */
void biquadBankSetStream (BiquadBank *b, void *stream)
{
    if (b->stream == stream) return;
    BANK_ENTER (b);
    arthip_sync (b->stream);
    BANK_LEAVE (b);
    b->stream = stream;
}

void biquadBankApplyInterleavedDevice (BiquadBank *b, artsample_t *d_buffer, int numFrames)
{
    if (numFrames <= 0) return;
    if (b->nshards) {
        /* every shard waits for the bank's stream, pulls its channel slice of the caller's buffer (peer-to-peer when it sits on
         * another device), filters it in its own HBM and pushes it back; the bank's stream then waits for all of them */
        const int prev = arthip_current_device (), wps = (int)(sizeof (art_s) / 4);
        arthip_set_device (b->device);
        arthip_event_record (b->ev_parent, b->stream);
        for (int k = 0; k < b->nshards; ++k) {
            BiquadBank *sh = b->shards [k];
            const int first = b->shard_first [k], width = b->shard_first [k + 1] - first;
            const size_t need = sizeof (art_s) * (size_t) numFrames * width;
            arthip_set_device (sh->device);
            arthip_stream_wait_event (sh->stream, b->ev_parent);
            if (need > sh->slice_cap) {
                arthip_sync (sh->stream);
                arthip_free (sh->d_slice);
                sh->slice_cap = need + need / 2;
                if (!(sh->d_slice = arthip_malloc (sh->slice_cap))) {        /* (counted: this shard's channels stay unfiltered, as an ordinary bank's would) */
                    sh->slice_cap = 0;
                    pcm_fail ("sharded biquad bank: device allocation failed (a shard's channels left unfiltered)");
                    arthip_event_record (b->ev_shard [k], sh->stream);
                    continue;
                }
            }
            arthip_slice_copy (sh->d_slice, (size_t) width * wps, d_buffer + first, (size_t) b->C * wps, width * wps, (size_t) numFrames, sh->stream);
            biquadBankApplyInterleavedDevice (sh, sh->d_slice, numFrames);
            arthip_slice_copy (d_buffer + first, (size_t) b->C * wps, sh->d_slice, (size_t) width * wps, width * wps, (size_t) numFrames, sh->stream);
            arthip_event_record (b->ev_shard [k], sh->stream);
        }
        arthip_set_device (b->device);
        for (int k = 0; k < b->nshards; ++k) arthip_stream_wait_event (b->stream, b->ev_shard [k]);
        if (prev >= 0) arthip_set_device (prev);
        return;
    }
    BANK_ENTER (b);
    const int L = b->warmup ? spec_chunk (b->S, b->warmup) : 0;

    if (L && numFrames >= 2 * L) {
        const size_t samples = (size_t) numFrames * b->C, need = arthip_biquad_spec_scratch (b->C, b->S, numFrames, L);
        if (samples * sizeof (art_s) > b->tmp_cap) {
            arthip_free (b->d_tmp);
            b->tmp_cap = (samples + samples / 2) * sizeof (art_s);
            if (!(b->d_tmp = arthip_malloc (b->tmp_cap))) b->tmp_cap = 0;
        }
        if (need > b->spec_cap) {
            arthip_free (b->d_spec);
            b->spec_cap = need + need / 2;
            if (!(b->d_spec = arthip_malloc (b->spec_cap))) b->spec_cap = 0;
        }
        if (b->d_tmp && b->d_spec) {
            arthip_d2d (b->d_tmp, d_buffer, samples * sizeof (art_s), b->stream);
            if (!arthip_biquad_spec (b->d_sections, b->C, b->S, b->d_tmp, b->C, d_buffer, b->C, numFrames, L, b->warmup, b->d_spec, b->d_first_bad, b->d_repairs, b->stream)) {
                BANK_LEAVE (b);
                return;
            }
        }
        fprintf (stderr, "artamd: time-parallel biquad unavailable (%s): serial kernel\n", arthip_last_error ());
    }
    if (b->all_order2 && numFrames >= 64)
        arthip_biquad_order2 (b->d_sections, b->C, b->S, d_buffer, numFrames, b->C, b->stream);
    else
        arthip_biquad_chain (b->d_sections, b->C, b->S, d_buffer, numFrames, b->C, b->stream);
    BANK_LEAVE (b);
}

void biquadBankRead (BiquadBank *b, Biquad *sections)
{
    BANK_ENTER (b);
    if (b->nshards) {
        arthip_sync (b->stream);
        for (int k = 0; k < b->nshards; ++k) biquadBankRead (b->shards [k], sections + (size_t) b->shard_first [k] * b->S);
    }
    else {
        arthip_d2h (sections, b->d_sections, sizeof (Biquad) * (size_t) b->C * b->S, b->stream);
        arthip_sync (b->stream);
    }
    BANK_LEAVE (b);
}

/* chunks the time-parallel form had to recompute for this bank so far (synchronises; diagnostics) */
unsigned int biquadBankRepairs (BiquadBank *b)
{
    unsigned int n = 0;
    BANK_ENTER (b);
    if (b->nshards) {
        arthip_sync (b->stream);
        for (int k = 0; k < b->nshards; ++k) n += biquadBankRepairs (b->shards [k]);
    }
    else {
        arthip_d2h (&n, b->d_repairs, sizeof (n), b->stream);
        arthip_sync (b->stream);
    }
    BANK_LEAVE (b);
    return n;
}

void biquadBankFree (BiquadBank *b)
{
    if (!b) return;
    BANK_ENTER (b);
    arthip_sync (b->stream);
    for (int k = 0; k < b->nshards; ++k) {
        if (b->shards [k]) { void *st = b->shards [k]->stream; biquadBankFree (b->shards [k]); arthip_stream_destroy (st); }
        if (b->ev_shard && b->ev_shard [k]) arthip_event_destroy (b->ev_shard [k]);
    }
    if (b->ev_parent) arthip_event_destroy (b->ev_parent);
    free (b->shards); free (b->shard_first); free (b->ev_shard);
    arthip_free (b->d_sections); arthip_free (b->d_tmp); arthip_free (b->d_spec); arthip_free (b->d_repairs); arthip_free (b->d_first_bad); arthip_free (b->d_slice);
    BANK_LEAVE (b);
    free (b);
}

/* ------------------------------------------------------------------------------------------
 * Decimator
 * ---------------------------------------------------------------------------------------- */

struct artamd_decimator {
    void *stream;
    /* the state the kernels carry — clip counter, error feedback, dither generators (two copies: the parallel kernel leaves
     * the advanced state in the other one), noise shapers — lives in ONE device block, so that a host-pointer call brings
     * it back in the same launch as its output; the pointers below point into it */
    unsigned char *d_state, *h_state; size_t state_bytes;          /* h_state: page-locked mirror */
    art_s *d_feedback; uint32_t *d_gens, *d_gens_alt; Biquad *d_shapers;
    unsigned long long *d_clipped;
    unsigned long long clipped_seen;
    art_s *d_in; size_t in_cap;
    unsigned char *d_out; size_t out_cap;
    unsigned char *h_in, *h_out; size_t h_in_cap, h_out_cap;       /* page-locked staging of small host-pointer calls */
    int device;                                                    /* where the state lives (and where device-pointer buffers are expected) */
    /* DECIMATE_MULTITHREADED over several devices (or ARTAMD_SHARDS): the reference's one-worker-per-channel fan-out
     * (decimator.c:92-93, 119-136) with GPUs for threads — `nshards` ordinary contexts with contiguous channel slices, each on
     * its own device and stream; this context then owns no kernel state itself, only the host mirrors of all channels */
    int nshards; Decimate **shards; int *shard_first; void **ev_shard; void *ev_parent;
};
#define DEC_KERNEL_COPY_LIMIT ((size_t) 1 << 20)
#define DEC_ENTER(hip) const int prev_device_ = arthip_current_device (); \
                       if (prev_device_ != (hip)->device) arthip_set_device ((hip)->device)
#define DEC_LEAVE(hip) do { if (prev_device_ != (hip)->device && prev_device_ >= 0) arthip_set_device (prev_device_); } while (0)

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

/* an ordinary context on the current device; `firstChannel`: its channels are channels firstChannel.. of a wider stream (the
 * dither generators of a stream are seeded channel after channel from one byte stream, decimator.c:40-52) */
static Decimate *dec_init_leaf (int numChannels, int outputBits, int outputBytes, double outputGain, int sampleRate, int flags, int firstChannel)
{
    Decimate *cxt = calloc (1, sizeof (Decimate));
    struct artamd_decimator *hip = calloc (1, sizeof (*hip));
    const int C = numChannels;

    if (!cxt || !hip) { free (cxt); free (hip); return NULL; }
    cxt->hip = hip;
    hip->device = arthip_current_device ();
    cxt->numChannels = C; cxt->outputBits = outputBits; cxt->outputBytes = outputBytes;
    cxt->outputGain = outputGain; cxt->flags = flags;
    cxt->feedback = calloc (C, sizeof (art_s));
    {   /* layout of the state block (every part 16-byte aligned) */
        size_t off = 16;                                                    /* [0] the clip counter */
        const size_t o_fb = off;     off += (sizeof (art_s) * C + 15) & ~(size_t) 15;
        const size_t o_g0 = off;     off += (sizeof (uint32_t) * C + 15) & ~(size_t) 15;
        const size_t o_g1 = off;     off += (sizeof (uint32_t) * C + 15) & ~(size_t) 15;
        const size_t o_sh = off;     off += (sizeof (Biquad) * C + 15) & ~(size_t) 15;
        hip->state_bytes = off;
        hip->d_state = arthip_malloc (off);
        hip->h_state = arthip_host_alloc (off);
        if (hip->d_state && hip->h_state) {
            arthip_zero (hip->d_state, off, NULL);
            hip->d_clipped = (unsigned long long *) hip->d_state;
            hip->d_feedback = (art_s *)(hip->d_state + o_fb);
            if (flags & DITHER_ENABLED) { hip->d_gens = (uint32_t *)(hip->d_state + o_g0); hip->d_gens_alt = (uint32_t *)(hip->d_state + o_g1); }
            if (flags & SHAPING_ENABLED) hip->d_shapers = (Biquad *)(hip->d_state + o_sh);
        }
    }

    if ((flags & DITHER_ENABLED) && hip->d_state) {
        /* per-channel seeds: little-endian words cut from the byte stream (state >> 24), three steps per byte */
        uint32_t s = 0x31415926;
        cxt->tpdf_generators = calloc (C, sizeof (uint32_t));
        for (int c = -firstChannel; c < C; ++c)
            for (int b = 0; b < 4; ++b) {
                if (c >= 0) cxt->tpdf_generators [c] |= (uint32_t)(s >> 24) << (8 * b);
                s = lcg_step (lcg_step (lcg_step (s)));
            }
        cxt->dither_type = (flags & DITHER_HIGHPASS) ? -1 : (flags & DITHER_LOWPASS) ? 1 : 0;
        arthip_h2d (hip->d_gens, cxt->tpdf_generators, sizeof (uint32_t) * C, NULL);
    }

    if ((flags & SHAPING_ENABLED) && hip->d_state) {
        cxt->noise_shapers = calloc (C, sizeof (Biquad));
        for (int c = 0; c < C; ++c)
            shaper_for (cxt->noise_shapers + c, flags, sampleRate);
        arthip_h2d (hip->d_shapers, cxt->noise_shapers, sizeof (Biquad) * C, NULL);
    }

    if (arthip_sync (NULL) || !hip->d_feedback || !hip->d_clipped) {
        fprintf (stderr, "artamd: decimateInit: device setup failed: %s\n", arthip_last_error ());
        decimateFree (cxt);
        return NULL;
    }

    return cxt;
}

static Decimate *dec_init_sharded (int numChannels, int outputBits, int outputBytes, double outputGain, int sampleRate, int flags, int count, const int *devices)
{
    Decimate *cxt = calloc (1, sizeof (Decimate));
    struct artamd_decimator *hip = calloc (1, sizeof (*hip));
    const int prev = arthip_current_device (), C = numChannels;

    if (!cxt || !hip) { free (cxt); free (hip); return NULL; }
    cxt->hip = hip;
    hip->device = prev;
    cxt->numChannels = C; cxt->outputBits = outputBits; cxt->outputBytes = outputBytes;
    cxt->outputGain = outputGain; cxt->flags = flags;
    cxt->dither_type = (flags & DITHER_HIGHPASS) ? -1 : (flags & DITHER_LOWPASS) ? 1 : 0;
    cxt->feedback = calloc (C, sizeof (art_s));
    if (flags & DITHER_ENABLED) cxt->tpdf_generators = calloc (C, sizeof (uint32_t));
    if (flags & SHAPING_ENABLED) cxt->noise_shapers = calloc (C, sizeof (Biquad));
    hip->shards = calloc ((size_t) count, sizeof (Decimate *));
    hip->shard_first = calloc ((size_t) count + 1, sizeof (int));
    hip->ev_shard = calloc ((size_t) count, sizeof (void *));
    hip->ev_parent = arthip_order_event_create ();
    int ok = cxt->feedback && hip->shards && hip->shard_first && hip->ev_shard && hip->ev_parent;

    /* contiguous, balanced channel slices: the first (channels % count) shards get one channel more */
    const int base = C / count, extra = C % count;
    for (int s = 0; ok && s < count; ++s) {
        const int width = base + (s < extra ? 1 : 0);
        hip->shard_first [s + 1] = hip->shard_first [s] + width;
        arthip_set_device (devices [s]);                  /* (artamd_shard_plan has made sure it can address `prev`'s memory) */
        Decimate *leaf = dec_init_leaf (width, outputBits, outputBytes, outputGain, sampleRate, flags & ~DECIMATE_MULTITHREADED, hip->shard_first [s]);
        hip->shards [s] = leaf;
        hip->ev_shard [s] = arthip_order_event_create ();
        hip->nshards = s + 1;
        ok = leaf && hip->ev_shard [s] && (leaf->hip->stream = arthip_stream_create ()) != NULL;
        if (ok) {       /* the host-visible state of the whole stream, as every context shows it */
            const int first = hip->shard_first [s];
            if (cxt->tpdf_generators && leaf->tpdf_generators) memcpy (cxt->tpdf_generators + first, leaf->tpdf_generators, sizeof (uint32_t) * width);
            if (cxt->noise_shapers && leaf->noise_shapers) memcpy (cxt->noise_shapers + first, leaf->noise_shapers, sizeof (Biquad) * width);
        }
    }
    if (prev >= 0) arthip_set_device (prev);
    if (!ok) {
        fprintf (stderr, "artamd: sharded decimator: allocation failed: %s\n", arthip_last_error ());
        decimateFree (cxt);
        return NULL;
    }
    return cxt;
}

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
    if ((flags & DECIMATE_MULTITHREADED) && numChannels > 1) {
        int devices [ART_MAX_DEVICES];
        const int count = artamd_shard_plan (numChannels, arthip_current_device (), devices);
        if (count > 1)
            return dec_init_sharded (numChannels, outputBits, outputBytes, outputGain, sampleRate, flags, count, devices);
    }
    return dec_init_leaf (numChannels, outputBits, outputBytes, outputGain, sampleRate, flags, 0);
}

void decimateFree (Decimate *cxt)
{
    if (!cxt) return;
    struct artamd_decimator *hip = cxt->hip;
    if (hip) {
        DEC_ENTER (hip);
        arthip_sync (hip->stream);
        arthip_free (hip->d_state); arthip_host_free (hip->h_state);
        arthip_free (hip->d_in); arthip_free (hip->d_out); arthip_host_free (hip->h_in); arthip_host_free (hip->h_out);
        for (int s = 0; s < hip->nshards; ++s) {
            if (hip->shards [s]) {
                void *st = hip->shards [s]->hip ? hip->shards [s]->hip->stream : NULL;
                decimateFree (hip->shards [s]);
                arthip_stream_destroy (st);
            }
            if (hip->ev_shard && hip->ev_shard [s]) arthip_event_destroy (hip->ev_shard [s]);
        }
        if (hip->ev_parent) arthip_event_destroy (hip->ev_parent);
        free (hip->shards); free (hip->shard_first); free (hip->ev_shard);
        DEC_LEAVE (hip);
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

/* (the shaper / dither state is shared between calls: the old stream is drained before the switch) */
void decimateHipSetStream (Decimate *cxt, void *stream)
{
    if (cxt->hip->stream == stream) return;
    DEC_ENTER (cxt->hip);
    arthip_sync (cxt->hip->stream);
    DEC_LEAVE (cxt->hip);
    cxt->hip->stream = stream;
}

/* shards of a DECIMATE_MULTITHREADED context (0: an ordinary context) */
int decimateHipShardCount (Decimate *cxt) { return cxt->hip->nshards; }

static void dec_swap_if (Decimate *cxt, int rc)
{
    if (rc == 1) { uint32_t *t = cxt->hip->d_gens; cxt->hip->d_gens = cxt->hip->d_gens_alt; cxt->hip->d_gens_alt = t; }
}

static int dec_reserve (Decimate *cxt, size_t in_bytes, size_t out_bytes);

/* a sharded context's call on device buffers (the caller's, or this context's own staging of a host-pointer call): every shard
 * waits for the context's stream, pulls its channel slice with a slice kernel (peer-to-peer when it sits on another device),
 * decimates it with its own state and pushes its packed bytes into the interleaved output; the context's stream then waits for
 * all of them */
static void dec_sharded_device_call (Decimate *cxt, const art_s *d_input, int frames, unsigned char *d_output)
{
    struct artamd_decimator *hip = cxt->hip;
    const int C = cxt->numChannels, B = cxt->outputBytes, prev = arthip_current_device ();
    const int wps = (int)(sizeof (art_s) / 4);

    /* every shard's buffers first: a shard that cannot get them would leave its channels of the interleaved output unwritten, so
     * the whole call then leaves silence behind, and the failure is counted (art_hip.h: the error contract) */
    for (int k = 0; k < hip->nshards; ++k) {
        Decimate *leaf = hip->shards [k];
        const int width = hip->shard_first [k + 1] - hip->shard_first [k];
        arthip_set_device (leaf->hip->device);
        if (dec_reserve (leaf, (size_t) frames * width * sizeof (art_s), (size_t) frames * width * B)) {
            pcm_fail ("sharded decimator: device allocation failed (output zeroed)");
            arthip_set_device (hip->device);
            arthip_zero (d_output, (size_t) frames * C * B, hip->stream);
            if (prev >= 0) arthip_set_device (prev);
            return;
        }
    }
    arthip_set_device (hip->device);
    arthip_event_record (hip->ev_parent, hip->stream);
    for (int k = 0; k < hip->nshards; ++k) {
        Decimate *leaf = hip->shards [k];
        struct artamd_decimator *sp = leaf->hip;
        const int first = hip->shard_first [k], width = hip->shard_first [k + 1] - first;
        ArtDecArgs a;
        arthip_set_device (sp->device);
        arthip_stream_wait_event (sp->stream, hip->ev_parent);
        arthip_slice_copy (sp->d_in, (size_t) width * wps, d_input + first, (size_t) C * wps, width * wps, (size_t) frames, sp->stream);
        dec_args (leaf, &a);
        dec_swap_if (leaf, arthip_decimate (&a, sp->d_in, frames, sp->d_out, sp->stream));
        arthip_slice_copy_bytes (d_output + (size_t) first * B, (size_t) C * B, sp->d_out, (size_t) width * B, width * B, (size_t) frames, sp->stream);
        arthip_event_record (hip->ev_shard [k], sp->stream);
    }
    arthip_set_device (hip->device);
    for (int k = 0; k < hip->nshards; ++k)
        arthip_stream_wait_event (hip->stream, hip->ev_shard [k]);
    if (prev >= 0) arthip_set_device (prev);
}

void decimateProcessInterleavedLEDevice (Decimate *cxt, const artsample_t *d_input, int numInputFrames, unsigned char *d_output)
{
    if (numInputFrames <= 0) return;
    if (cxt->hip->nshards) { dec_sharded_device_call (cxt, d_input, numInputFrames, d_output); return; }
    ArtDecArgs a;
    DEC_ENTER (cxt->hip);
    dec_args (cxt, &a);
    dec_swap_if (cxt, arthip_decimate (&a, d_input, numInputFrames, d_output, cxt->hip->stream));
    DEC_LEAVE (cxt->hip);
}

long decimateHipClipped (Decimate *cxt)
{
    unsigned long long total = 0;
    if (cxt->hip->nshards) {
        long sum = 0;
        {
            DEC_ENTER (cxt->hip);
            arthip_sync (cxt->hip->stream);
            DEC_LEAVE (cxt->hip);
        }
        for (int k = 0; k < cxt->hip->nshards; ++k) sum += decimateHipClipped (cxt->hip->shards [k]);
        return sum;
    }
    DEC_ENTER (cxt->hip);
    arthip_d2h (&total, cxt->hip->d_clipped, sizeof (total), cxt->hip->stream);
    arthip_sync (cxt->hip->stream);
    DEC_LEAVE (cxt->hip);
    return (long) total;
}

static int dec_reserve (Decimate *cxt, size_t in_bytes, size_t out_bytes)
{
    struct artamd_decimator *hip = cxt->hip;
    in_bytes += 16; out_bytes += 16;                    /* (copy kernels move whole words) */
    if (in_bytes > hip->in_cap) { arthip_free (hip->d_in); hip->in_cap = in_bytes * 2; if (!(hip->d_in = arthip_malloc (hip->in_cap))) { hip->in_cap = 0; return -1; } }
    if (out_bytes > hip->out_cap) { arthip_free (hip->d_out); hip->out_cap = out_bytes * 2; if (!(hip->d_out = arthip_malloc (hip->out_cap))) { hip->out_cap = 0; return -1; } }
    if (in_bytes <= DEC_KERNEL_COPY_LIMIT && in_bytes > hip->h_in_cap) {
        arthip_host_free (hip->h_in); hip->h_in_cap = in_bytes * 2;
        if (!(hip->h_in = arthip_host_alloc (hip->h_in_cap))) { hip->h_in_cap = 0; return -1; }
    }
    if (out_bytes <= DEC_KERNEL_COPY_LIMIT && out_bytes > hip->h_out_cap) {
        arthip_host_free (hip->h_out); hip->h_out_cap = out_bytes * 2;
        if (!(hip->h_out = arthip_host_alloc (hip->h_out_cap))) { hip->h_out_cap = 0; return -1; }
    }
    return 0;
}

/* after a host-pointer call: bring back the output (out_bytes from d_out; NULL destination: the caller fetched it itself)
 * together with the state block, return this call's clip count and refresh the host-visible state mirrors */
static int dec_finish (Decimate *cxt, unsigned char *output, size_t out_bytes)
{
    struct artamd_decimator *hip = cxt->hip;
    const int C = cxt->numChannels;
    const int staged = output && out_bytes + 16 <= DEC_KERNEL_COPY_LIMIT;

    if (staged)         /* output and state in ONE launch, no copy-engine command */
        arthip_copy2_by_kernel (hip->h_out, hip->d_out, (out_bytes + 3) & ~(size_t) 3, hip->h_state, hip->d_state, hip->state_bytes, hip->stream);
    else {
        if (output) arthip_d2h (output, hip->d_out, out_bytes, hip->stream);
        arthip_d2h (hip->h_state, hip->d_state, hip->state_bytes, hip->stream);
    }
    if (arthip_sync (hip->stream)) { fprintf (stderr, "artamd: decimator: %s\n", arthip_last_error ()); return 0; }
    if (staged) memcpy (output, hip->h_out, out_bytes);

    const unsigned char *m = hip->h_state;
    const unsigned long long total = *(const unsigned long long *) m;
    memcpy (cxt->feedback, m + ((unsigned char *) hip->d_feedback - hip->d_state), sizeof (art_s) * C);
    if (cxt->tpdf_generators) memcpy (cxt->tpdf_generators, m + ((unsigned char *) hip->d_gens - hip->d_state), sizeof (uint32_t) * C);
    if (cxt->noise_shapers) memcpy (cxt->noise_shapers, m + ((unsigned char *) hip->d_shapers - hip->d_state), sizeof (Biquad) * C);

    int delta = (int)(total - hip->clipped_seen);
    hip->clipped_seen = total;
    return delta;
}

/* a sharded context after a host-pointer call: every shard's state block comes back on the shard's own stream; clip counts add
 * up, the host mirrors are assembled channel slice by channel slice */
static int dec_finish_shards (Decimate *cxt)
{
    struct artamd_decimator *hip = cxt->hip;
    const int prev = arthip_current_device ();
    int clipped = 0;
    for (int k = 0; k < hip->nshards; ++k) {
        struct artamd_decimator *sp = hip->shards [k]->hip;
        arthip_set_device (sp->device);
        arthip_d2h (sp->h_state, sp->d_state, sp->state_bytes, sp->stream);
    }
    for (int k = 0; k < hip->nshards; ++k) {
        Decimate *leaf = hip->shards [k];
        struct artamd_decimator *sp = leaf->hip;
        const int first = hip->shard_first [k], width = hip->shard_first [k + 1] - first;
        arthip_set_device (sp->device);
        if (arthip_sync (sp->stream)) { pcm_fail ("sharded decimator: a shard's stream failed (its channels are not to be trusted)"); continue; }
        const unsigned char *m = sp->h_state;
        const unsigned long long total = *(const unsigned long long *) m;
        memcpy (cxt->feedback + first, m + ((unsigned char *) sp->d_feedback - sp->d_state), sizeof (art_s) * width);
        if (cxt->tpdf_generators) memcpy (cxt->tpdf_generators + first, m + ((unsigned char *) sp->d_gens - sp->d_state), sizeof (uint32_t) * width);
        if (cxt->noise_shapers) memcpy (cxt->noise_shapers + first, m + ((unsigned char *) sp->d_shapers - sp->d_state), sizeof (Biquad) * width);
        clipped += (int)(total - sp->clipped_seen);
        sp->clipped_seen = total;
    }
    if (prev >= 0) arthip_set_device (prev);
    return clipped;
}

int decimateProcessInterleavedLE (Decimate *cxt, const artsample_t *input, int numInputFrames, unsigned char *output)
{
    struct artamd_decimator *hip = cxt->hip;
    if (numInputFrames <= 0) return 0;
    const size_t samples = (size_t) numInputFrames * cxt->numChannels, in_bytes = samples * sizeof (art_s);
    ArtDecArgs a;
    DEC_ENTER (hip);

    if (dec_reserve (cxt, in_bytes, samples * cxt->outputBytes)) {
        fprintf (stderr, "artamd: decimator device allocation failed: %s\n", arthip_last_error ());
        DEC_LEAVE (hip);
        return 0;
    }
    if (in_bytes + 16 <= DEC_KERNEL_COPY_LIMIT) {
        memcpy (hip->h_in, input, in_bytes);
        arthip_copy_by_kernel (hip->d_in, hip->h_in, in_bytes, hip->stream);
    }
    else arthip_h2d (hip->d_in, input, in_bytes, hip->stream);
    int clipped;
    if (hip->nshards) {
        /* the whole interleaved buffer is staged on this context's device (one dense transfer each way), the shards work on it */
        const size_t out_bytes = samples * cxt->outputBytes;
        dec_sharded_device_call (cxt, hip->d_in, numInputFrames, hip->d_out);
        if (out_bytes + 16 <= DEC_KERNEL_COPY_LIMIT) {
            arthip_copy_by_kernel (hip->h_out, hip->d_out, (out_bytes + 3) & ~(size_t) 3, hip->stream);
            arthip_sync (hip->stream);
            memcpy (output, hip->h_out, out_bytes);
        }
        else { arthip_d2h (output, hip->d_out, out_bytes, hip->stream); arthip_sync (hip->stream); }
        clipped = dec_finish_shards (cxt);
    }
    else {
        dec_args (cxt, &a);
        dec_swap_if (cxt, arthip_decimate (&a, hip->d_in, numInputFrames, hip->d_out, hip->stream));
        clipped = dec_finish (cxt, output, samples * cxt->outputBytes);
    }
    DEC_LEAVE (hip);
    return clipped;
}

/* planar call of an ordinary context, enqueued only (dec_finish / dec_finish_shards waits) */
static int dec_planar_begin (Decimate *cxt, const artsample_t *const *input, int numInputFrames, unsigned char *const *output)
{
    struct artamd_decimator *hip = cxt->hip;
    const int C = cxt->numChannels;
    const size_t n = (size_t) numInputFrames, plane_bytes = n * cxt->outputBytes;
    ArtDecArgs a;

    if (dec_reserve (cxt, n * C * sizeof (art_s), plane_bytes * C)) {
        fprintf (stderr, "artamd: decimator device allocation failed: %s\n", arthip_last_error ());
        return -1;
    }
    dec_args (cxt, &a);
    for (int c = 0; c < C; ++c)
        arthip_h2d (hip->d_in + n * c, input [c], n * sizeof (art_s), hip->stream);
    arthip_decimate_planar (&a, hip->d_in, (long) n, numInputFrames, hip->d_out, (long) plane_bytes, hip->stream);
    for (int c = 0; c < C; ++c)
        arthip_d2h (output [c], hip->d_out + plane_bytes * c, plane_bytes, hip->stream);
    return 0;
}

int decimateProcessLE (Decimate *cxt, const artsample_t *const *input, int numInputFrames, unsigned char *const *output)
{
    struct artamd_decimator *hip = cxt->hip;
    if (numInputFrames <= 0) return 0;
    if (hip->nshards) {
        /* planes need no slicing: a shard's channels are a run of the caller's planes; all shards are enqueued before any is waited for */
        const int prev = arthip_current_device ();
        arthip_set_device (hip->device);                /* (the context's stream on the context's device) */
        arthip_sync (hip->stream);
        for (int k = 0; k < hip->nshards; ++k) {
            arthip_set_device (hip->shards [k]->hip->device);
            if (dec_planar_begin (hip->shards [k], input + hip->shard_first [k], numInputFrames, output + hip->shard_first [k])) {
                /* (counted; the shard's planes are the caller's host memory: silence instead of whatever was there) */
                pcm_fail ("sharded decimator: device allocation failed (a shard's planes zeroed)");
                for (int c = hip->shard_first [k]; c < hip->shard_first [k + 1]; ++c) memset (output [c], 0, (size_t) numInputFrames * cxt->outputBytes);
            }
        }
        if (prev >= 0) arthip_set_device (prev);
        return dec_finish_shards (cxt);
    }
    DEC_ENTER (hip);
    const int clipped = dec_planar_begin (cxt, input, numInputFrames, output) ? 0 : dec_finish (cxt, NULL, 0);
    DEC_LEAVE (hip);
    return clipped;
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

/* host-pointer form: process-wide scratch (device + page-locked), small calls by copy kernels */
static struct { unsigned char *d_in, *h_in; art_s *d_out, *h_out; size_t in_cap, out_cap; int device; } ingest_scratch = { .device = -1 };
static pthread_mutex_t ingest_lock = PTHREAD_MUTEX_INITIALIZER;

void floatIntegersLE (unsigned char *input, double inputGain, int inputBits, int inputBytes, int inputStride, artsample_t *output, int numSamples)
{
    if (numSamples <= 0 || inputBits > 24) return;
    const size_t in_bytes = (size_t) numSamples * inputStride * inputBytes, out_bytes = sizeof (art_s) * (size_t) numSamples;
    /* the last sample's trailing stride bytes may not exist in the caller's buffer */
    const size_t valid = in_bytes - (size_t)(inputStride - 1) * inputBytes;

    pthread_mutex_lock (&ingest_lock);
    const int device = arthip_current_device ();
    if (ingest_scratch.device != device) {
        arthip_free (ingest_scratch.d_in); arthip_free (ingest_scratch.d_out); arthip_host_free (ingest_scratch.h_in); arthip_host_free (ingest_scratch.h_out);
        memset (&ingest_scratch, 0, sizeof (ingest_scratch));
        ingest_scratch.device = device;
    }
    if (in_bytes + 16 > ingest_scratch.in_cap) {
        arthip_free (ingest_scratch.d_in); arthip_host_free (ingest_scratch.h_in);
        ingest_scratch.in_cap = (in_bytes + 16) * 2;
        ingest_scratch.d_in = arthip_malloc (ingest_scratch.in_cap); ingest_scratch.h_in = arthip_host_alloc (ingest_scratch.in_cap);
    }
    if (out_bytes + 16 > ingest_scratch.out_cap) {
        arthip_free (ingest_scratch.d_out); arthip_host_free (ingest_scratch.h_out);
        ingest_scratch.out_cap = (out_bytes + 16) * 2;
        ingest_scratch.d_out = arthip_malloc (ingest_scratch.out_cap); ingest_scratch.h_out = arthip_host_alloc (ingest_scratch.out_cap);
    }
    if (arthip_device_count () < 1 || !ingest_scratch.d_in || !ingest_scratch.d_out || !ingest_scratch.h_in || !ingest_scratch.h_out) {
        /* no CPU evaluation path exists: say so (counted: artamdErrorCount) and hand back silence rather than uninitialised memory */
        pcm_fail ("floatIntegersLE needs a HIP device and scratch memory (no CPU path; output zeroed)");
        memset (output, 0, out_bytes);
        ingest_scratch.in_cap = ingest_scratch.out_cap = 0;
        pthread_mutex_unlock (&ingest_lock);
        return;
    }
    const int small = in_bytes + out_bytes <= ((size_t) 1 << 20);
    if (small) {
        memcpy (ingest_scratch.h_in, input, valid);
        arthip_copy_by_kernel (ingest_scratch.d_in, ingest_scratch.h_in, (valid + 3) & ~(size_t) 3, NULL);
    }
    else arthip_h2d (ingest_scratch.d_in, input, valid, NULL);
    floatIntegersLEDevice (ingest_scratch.d_in, inputGain, inputBits, inputBytes, inputStride, ingest_scratch.d_out, numSamples, NULL);
    if (small) arthip_copy_by_kernel (ingest_scratch.h_out, ingest_scratch.d_out, out_bytes, NULL);
    else arthip_d2h (output, ingest_scratch.d_out, out_bytes, NULL);
    if (arthip_sync (NULL)) { pcm_fail ("floatIntegersLE kernel failed (output zeroed)"); memset (output, 0, out_bytes); }
    else if (small) memcpy (output, ingest_scratch.h_out, out_bytes);
    pthread_mutex_unlock (&ingest_lock);
}
