/* stretch_host.c — host side of the time stretcher: the reference's stretch.h API (reference stretch.c:50-143,
 * :335-373 for init / reset / capacity / flush / free; the per-call state machine itself runs on the device,
 * stretch_kernels.hip).  No CPU path: without a HIP device stretchInit fails loudly. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "art_internal.h"
#include "stretch.h"

typedef struct { int mark, fill, cur, pad; double drift; } DevState;   /* must match stretch_kernels.hip */

struct artamd_stretch {
    ArtStretchArgs args;
    art_s *d_in, *d_out; size_t in_cap, out_cap;      /* staging for host-pointer calls (bytes) */
    int *d_result;
    void *stream;
    int blocks;                                        /* ring size in longest periods: 3, or 4 in fast mode */
    void *d_batch; size_t batch_cap;                   /* batched calls led by this context: items + results */
    unsigned long batch_stamp;                         /* last batched call this context took part in (duplicate check) */
};

static void *regrow (void *dev, size_t *cap, size_t need)
{
    if (need <= *cap) return dev;
    arthip_free (dev);
    dev = arthip_malloc (need + need / 2 + 4096);
    *cap = dev ? need + need / 2 + 4096 : 0;
    return dev;
}

static int push_state (Stretch *cxt)                   /* host mirrors -> device state (init / reset) */
{
    DevState st [2];
    memset (st, 0, sizeof (st));
    st [0].mark = cxt->tail; st [0].fill = cxt->head; st [0].drift = cxt->outsamples_error;
    if (cxt->next) { st [1].mark = cxt->next->tail; st [1].fill = cxt->next->head; st [1].drift = cxt->next->outsamples_error; }
    return arthip_h2d (cxt->hip->args.state, st, sizeof (st), cxt->hip->stream) || arthip_sync (cxt->hip->stream);
}

static void pull_state (Stretch *cxt)                  /* device state -> host mirrors (after every call) */
{
    DevState st [2];
    if (arthip_d2h (st, cxt->hip->args.state, sizeof (st), cxt->hip->stream) || arthip_sync (cxt->hip->stream)) return;
    cxt->tail = st [0].mark; cxt->head = st [0].fill; cxt->outsamples_error = st [0].drift;
    if (cxt->next) { cxt->next->tail = st [1].mark; cxt->next->head = st [1].fill; cxt->next->outsamples_error = st [1].drift; }
}

static Stretch *make_stage (int shortest, int longest, int channels, int fast)
{
    Stretch *s = calloc (1, sizeof (*s));
    if (!s) return NULL;
    s->num_chans = channels;
    s->inbuff_samples = longest * channels * (fast ? 4 : 3);
    s->head = s->tail = s->longest = longest * channels;
    s->shortest = shortest * channels;
    s->fast_mode = fast;
    return s;
}

Stretch *stretchInit (int shortest_period, int longest_period, int num_channels, int flags)
{
    const int fast = (flags & STRETCH_FAST_FLAG) != 0, dual = (flags & STRETCH_DUAL_FLAG) != 0;

    if (fast) {
        longest_period = (longest_period + 1) & ~1;
        shortest_period &= ~1;
    }

    if (longest_period <= shortest_period || shortest_period < MIN_PERIOD || longest_period > MAX_PERIOD) {
        fprintf (stderr, "stretchInit(): invalid periods!\n");
        return NULL;
    }

    if (num_channels < 1 || num_channels > 2) {
        fprintf (stderr, "stretchInit(): mono or stereo only!\n");
        return NULL;
    }

    if (artamdDeviceCount () <= 0) {
        fprintf (stderr, "artamd: stretchInit needs a HIP device (no CPU path): %s\n", arthip_last_error ());
        return NULL;
    }

    Stretch *cxt = make_stage (shortest_period, longest_period, num_channels, fast);
    struct artamd_stretch *hip = calloc (1, sizeof (*hip));
    if (!cxt || !hip) { free (cxt); free (hip); fprintf (stderr, "stretchInit(): out of memory!\n"); return NULL; }
    cxt->hip = hip;
    if (dual && !(cxt->next = make_stage (shortest_period, longest_period, num_channels, fast))) { stretchFree (cxt); return NULL; }

    ArtStretchArgs *a = &hip->args;
    const size_t ring_bytes = sizeof (art_s) * (size_t) cxt->inbuff_samples;
    int failed = 0;
    hip->blocks = fast ? 4 : 3;
    a->channels = num_channels; a->room = cxt->inbuff_samples; a->lo = cxt->shortest; a->hi = cxt->longest;
    a->quick = fast; a->paired = dual;
    for (int s = 0; s < (dual ? 2 : 1); ++s)
        for (int b = 0; b < 2; ++b) {
            a->ring [s][b] = arthip_malloc (ring_bytes);
            failed |= !a->ring [s][b] || arthip_zero (a->ring [s][b], ring_bytes, NULL);
        }
    if (dual) { a->between = arthip_malloc (ring_bytes); failed |= !a->between; }
    a->total = arthip_malloc (sizeof (art_s) * (MAX_PERIOD + 8));
    a->score = arthip_malloc (sizeof (art_s) * (MAX_PERIOD + 8));
    a->state = arthip_malloc (2 * sizeof (DevState));
    hip->d_result = arthip_malloc (sizeof (int));
    failed |= !a->total || !a->score || !a->state || !hip->d_result;

    if (failed || push_state (cxt)) {
        fprintf (stderr, "artamd: stretchInit: device allocation failed: %s\n", arthip_last_error ());
        stretchFree (cxt);
        return NULL;
    }

    return cxt;
}

void stretchFree (Stretch *cxt)
{
    if (!cxt) return;
    if (cxt->hip) {
        ArtStretchArgs *a = &cxt->hip->args;
        for (int s = 0; s < 2; ++s) for (int b = 0; b < 2; ++b) arthip_free (a->ring [s][b]);
        arthip_free (a->between); arthip_free (a->total); arthip_free (a->score); arthip_free (a->state);
        arthip_free (cxt->hip->d_result); arthip_free (cxt->hip->d_in); arthip_free (cxt->hip->d_out); arthip_free (cxt->hip->d_batch);
        free (cxt->hip);
    }
    free (cxt->next);
    free (cxt);
}

void stretchReset (Stretch *cxt)
{
    ArtStretchArgs *a = &cxt->hip->args;
    /* the reference keeps outsamples_error across a reset (stretch.c:102-110) — so do we */
    for (Stretch *s = cxt; s; s = s->next) s->head = s->tail = s->longest;
    for (int s = 0; s < (a->paired ? 2 : 1); ++s)
        arthip_zero (a->ring [s][0], sizeof (art_s) * (size_t) cxt->longest, cxt->hip->stream);
    push_state (cxt);                                   /* ring 0 current again */
}

int stretchGetOutputCapacity (Stretch *cxt, int max_num_samples, double max_ratio)
{
    int frames = max_num_samples;

    for (Stretch *s = cxt; s; s = s->next) {            /* stage by stage, as the reference recurses (stretch.c:117-143) */
        double here = max_ratio, rest = 1.0;
        if (s->next) {
            if (here < 0.5) { rest = here / 0.5; here = 0.5; }
            else if (here > 2.0) { rest = here / 2.0; here = 2.0; }
        }
        frames = (int) ceil (frames * ceil (here * 2.0) / 2.0) + (s->longest / s->num_chans) * (s->fast_mode ? 4 : 3);
        max_ratio = rest;
    }

    return frames;
}

/* (rings and state are shared between calls: the old stream is drained before the switch) */
void stretchHipSetStream (Stretch *cxt, void *hipStream)
{
    if (cxt->hip->stream == hipStream) return;
    arthip_sync (cxt->hip->stream);
    cxt->hip->stream = hipStream;
}

static int device_call (Stretch *cxt, const art_s *d_in, int frames, art_s *d_out, double ratio, int flush)
{
    struct artamd_stretch *hip = cxt->hip;
    int made = 0;

    if (arthip_stretch_call (&hip->args, d_in, frames, d_out, ratio, flush, hip->d_result, hip->stream) ||
        arthip_d2h (&made, hip->d_result, sizeof (int), hip->stream) || arthip_sync (hip->stream)) {
        fprintf (stderr, "artamd: stretch launch failed: %s\n", arthip_last_error ());
        return 0;
    }

    pull_state (cxt);
    return made;
}

int stretchProcessDevice (Stretch *cxt, const artsample_t *d_samples, int num_samples, artsample_t *d_output, double ratio)
{
    return num_samples > 0 ? device_call (cxt, d_samples, num_samples, d_output, ratio, 0) : 0;
}

int stretchFlushDevice (Stretch *cxt, artsample_t *d_output)
{
    return device_call (cxt, NULL, 0, d_output, 1.0, 1);
}

/* n independent streams, one launch: item i is exactly the call stretchProcessDevice (cxts [i], ...) / stretchFlushDevice
 * would make — same device code, one workgroup per stream — so results are identical to n separate calls.  The launch
 * goes to the stream of cxts [0], whose scratch also carries the item table. */
static int batch_call (Stretch *const *cxts, int n, const artsample_t *const *d_samples, const int *num_samples,
                       artsample_t *const *d_outputs, const double *ratios, int flush, int *produced)
{
    if (n <= 0) return 0;
    struct artamd_stretch *lead = cxts [0]->hip;
    const size_t items_bytes = ((size_t) n * sizeof (ArtStretchItem) + 63) & ~(size_t) 63, done_bytes = (size_t) n * sizeof (ArtStretchDone);
    ArtStretchItem *items = malloc (items_bytes);
    ArtStretchDone *done = malloc (done_bytes);
    int rc = -1;

    lead->d_batch = regrow (lead->d_batch, &lead->batch_cap, items_bytes + done_bytes);
    if (!items || !done || !lead->d_batch) goto out;

    static unsigned long calls;
    const unsigned long stamp = __atomic_add_fetch (&calls, 1, __ATOMIC_RELAXED);
    for (int i = 0; i < n; ++i) {
        if (cxts [i]->hip->batch_stamp == stamp) { fprintf (stderr, "artamd: stretch batch: a context appears twice\n"); goto out; }
        cxts [i]->hip->batch_stamp = stamp;
        items [i].args = cxts [i]->hip->args;
        items [i].in = flush ? NULL : d_samples [i];
        items [i].out = d_outputs [i];
        items [i].ratio = flush ? 1.0 : ratios [i];
        items [i].frames = flush ? 0 : num_samples [i];
        /* a stream with nothing to process this round still takes part: its workgroup returns at once */
        items [i].flush = flush;
    }

    ArtStretchDone *d_done = (ArtStretchDone *)((char *) lead->d_batch + items_bytes);
    if (arthip_h2d (lead->d_batch, items, (size_t) n * sizeof (ArtStretchItem), lead->stream) ||
        arthip_stretch_batch ((const ArtStretchItem *) lead->d_batch, d_done, n, lead->stream) ||
        arthip_d2h (done, d_done, done_bytes, lead->stream) || arthip_sync (lead->stream)) {
        fprintf (stderr, "artamd: stretch batch launch failed: %s\n", arthip_last_error ());
        goto out;
    }

    for (int i = 0; i < n; ++i) {
        Stretch *c = cxts [i];
        produced [i] = done [i].made;
        c->tail = done [i].state [0].mark; c->head = done [i].state [0].fill; c->outsamples_error = done [i].state [0].drift;
        if (c->next) { c->next->tail = done [i].state [1].mark; c->next->head = done [i].state [1].fill; c->next->outsamples_error = done [i].state [1].drift; }
    }
    rc = 0;
out:
    free (items); free (done);
    return rc;
}

int stretchProcessBatchDevice (Stretch *const *cxts, int n, const artsample_t *const *d_samples, const int *num_samples,
                               artsample_t *const *d_outputs, const double *ratios, int *produced)
{
    return batch_call (cxts, n, d_samples, num_samples, d_outputs, ratios, 0, produced);
}

int stretchFlushBatchDevice (Stretch *const *cxts, int n, artsample_t *const *d_outputs, int *produced)
{
    return batch_call (cxts, n, NULL, NULL, d_outputs, NULL, 1, produced);
}

/* frames a call can emit at most: what is buffered plus what comes in, at the largest stage ratios, plus slack */
static size_t worst_case_frames (Stretch *cxt, int num_samples)
{
    size_t frames = (size_t) num_samples + (size_t) cxt->inbuff_samples / cxt->num_chans;
    for (Stretch *s = cxt; s; s = s->next)
        frames = frames * 2 + (size_t)(s->inbuff_samples / s->num_chans) * 2;
    return frames;
}

int stretchProcess (Stretch *cxt, const artsample_t *samples, int num_samples, artsample_t *output, double ratio)
{
    struct artamd_stretch *hip = cxt->hip;
    const int C = cxt->num_chans;

    if (num_samples <= 0) return 0;
    hip->d_in = regrow (hip->d_in, &hip->in_cap, sizeof (art_s) * (size_t) num_samples * C);
    hip->d_out = regrow (hip->d_out, &hip->out_cap, sizeof (art_s) * worst_case_frames (cxt, num_samples) * C);
    if (!hip->d_in || !hip->d_out || arthip_h2d (hip->d_in, samples, sizeof (art_s) * (size_t) num_samples * C, hip->stream)) {
        fprintf (stderr, "artamd: stretchProcess: %s\n", arthip_last_error ());
        return 0;
    }

    const int made = device_call (cxt, hip->d_in, num_samples, hip->d_out, ratio, 0);
    if (made > 0) { arthip_d2h (output, hip->d_out, sizeof (art_s) * (size_t) made * C, hip->stream); arthip_sync (hip->stream); }
    return made;
}

int stretchFlush (Stretch *cxt, artsample_t *output)
{
    struct artamd_stretch *hip = cxt->hip;
    const int C = cxt->num_chans;

    hip->d_out = regrow (hip->d_out, &hip->out_cap, sizeof (art_s) * worst_case_frames (cxt, 0) * C);
    if (!hip->d_out) return 0;
    const int made = device_call (cxt, NULL, 0, hip->d_out, 1.0, 1);
    if (made > 0) { arthip_d2h (output, hip->d_out, sizeof (art_s) * (size_t) made * C, hip->stream); arthip_sync (hip->stream); }
    return made;
}
