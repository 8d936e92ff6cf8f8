/* resampler_host.c — host (C) side of the MI355X sinc resampler.
 *
 * What runs here, on the CPU, is only what is inherently scalar and O(1)..O(log n) per call:
 *   - filter-bank design in fp64 (once per context)           reference resampler.c:1090-1133, :149-168
 *   - fixed-ratio resolution (gcd / auto low-pass)            reference resampler.c:310-356
 *   - the position state machine, replayed in CLOSED FORM     reference resampler.c:487-537 (loop form)
 *   - getters / dry runs                                       reference resampler.c:365-397, :853-968
 * Every output sample is computed on the GPU (sinc_fir.hip).  There is no CPU evaluation path:
 * without a device the init functions fail loudly.
 */
#define _USE_MATH_DEFINES
#define _POSIX_C_SOURCE 200809L
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "art_internal.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define HIST_FRAMES(T) ((T) + (T) / 2)      /* frames of history kept in HBM between calls */

/* host-pointer calls up to this many bytes (in + out) go through page-locked staging buffers — the input by a copy kernel, the
 * output written there by the FIR kernels themselves, no copy-engine command at all —; larger ones are copied straight
 * from / to the caller's memory */
#define KERNEL_COPY_LIMIT ((size_t) 1 << 20)     /* staged transfers up to this size are made by a copy kernel, not a copy-engine command */
#define DIRECT_OUT_LIMIT ((size_t) 1 << 20)      /* staged outputs up to this size are written by the FIR kernels straight into the page-locked buffer */
#define STAGE_LIMIT ((size_t) 3 << 19)          /* 1.5 MB: measured on MI355X hosts, 8 ch x 988 taps: 16,384-frame calls 107 -> 92 us staged, 65,536-frame
                                                 * calls 176 us direct vs 235-378 staged (the CPU's own copies into and out of the staging cost more than
                                                 * the runtime's pipelined pageable path saves).  Environment ARTAMD_STAGE_LIMIT=bytes overrides (tests) */

struct BankEntry;
struct shard_pool;
struct artamd_resampler {
    int device;                             /* HIP device this context lives on (made current around every call) */
    void *stream;
    int own_stream;                         /* the stream was created by the library (shards of a sharded context) */
    /* a sharded context (RESAMPLE_MULTITHREADED with several devices / ARTAMD_SHARDS): no device state of its own, its
     * channels are spread over `nshards` ordinary contexts that run side by side (reference resampler.c:442-470 fans the
     * channels of ONE context out to worker threads; here to devices) */
    int nshards;
    Resample **shards;
    int *shard_first;                       /* first channel of each shard (nshards + 1 entries) */
    void *ev_parent, **ev_shard;            /* ordering events of the device-pointer calls */
    struct shard_pool *pool;                /* one worker thread per shard: the shards' launch sequences are enqueued side by side (NULL: by the caller, one after the other) */
    art_s *h_in, *h_out; size_t h_in_cap, h_out_cap;     /* page-locked staging of the host-pointer calls */
    struct BankEntry *bank;                 /* shared filter bank (bank_acquire / bank_release) */
    art_s *d_bank;                          /* = bank->dev */
    art_s *d_hist [2];                      /* ping-pong history, HIST x C interleaved */
    int cur;
    art_s *d_in;  size_t in_cap;            /* staging for host-pointer calls (bytes) */
    art_s *d_out; size_t out_cap;
    art_s *d_tmp; size_t tmp_cap;           /* planar <-> interleaved scratch */
    ArtamdSegment *segs; int seg_cap;
    int floor_active;                       /* ring index 0 is a hard history floor (after a flush-time rewind) */
    int kernel_pref, last_kernel;
    long long lin_origin;                    /* input frames appended since init / reset (ArtFirArgs.lin_origin) */
    unsigned int invariant_fallbacks;        /* launches the cut-invariant policy had to give to the general kernel (resampleHipCutInvariantFallbacks) */
    int stream_channels;                     /* a shard: channels of the whole stream (kernel choice); 0 otherwise */
    /* cached rational structure of the current ratio */
    double period_ratio; int period_out, period_in;
    /* optional HIP-event timing of FIR launches */
    int timing; void **ev; int ev_count, ev_cap; double prep_ms;
    art_s *d_patch; size_t patch_cap;        /* end-point extrapolation: samples computed on the host */
    void *d_scratch; size_t scratch_cap;     /* MFMA path: effective rows + canonical positions of one launch */
    void *d_pad; size_t pad_cap;             /* matrix path of a channel count the kernels are not compiled for: the groups' padded buffers (arthip_fir_pad_bytes) */
    void *d_planes; size_t planes_cap;       /* fixed-point matrix kernel: digit planes of one launch (flag word first) */
    int rows_off; void *d_rows; size_t rows_cap; void *rows_cache; void *last_masks;      /* ... its filter rows, kept across calls (art_internal.h), and where the last launch's row masks live */
    void *d_split; size_t split_cap;         /* K-split streaming kernel: arrival counters (zero at rest) + partial sums of one launch */
    int last_fixed [4];                      /* its last launch of the last call: flag value (0: none), mask words, chunks per tile, kernel form (art_hip.h) */
    unsigned int *d_fix; size_t fix_cap;    /* [0] per-launch, [1] running count of outputs the matrix kernels evaluated off-pattern */
    void *d_batch; size_t batch_cap;         /* argument table of the batched calls led by this context */
    unsigned long batch_stamp;               /* last batched call this context took part in (duplicate check) */
};

static struct shard_pool *shard_pool_create (Resample *cxt, int n);
static void shard_pool_destroy (struct shard_pool *pool);


/* ------------------------------------------------------------------------------------------
 * Filter bank
 * ---------------------------------------------------------------------------------------- */

/* One polyphase row: windowed sinc centred (T/2 - 1 + phase) taps in, DC-normalised, rounded to
 * float from the centre outwards with the rounding error carried along. */
static void design_row (art_s *row, double *work, int T, double phase, double lowpass, int use_bh)
{
    const int mid = T / 2;
    double dc = 0.0;

    for (int k = 0; k < T; ++k) {
        double radians = fabs ((mid - 1) + phase - k) * M_PI;
        double wphase = radians / mid;
        double tap = 1.0;

        if (radians != 0.0) {
            tap = sin (radians * lowpass) / (radians * lowpass);
            tap *= use_bh ? 0.35875 + 0.48829 * cos (wphase) + 0.14128 * cos (2 * wphase) + 0.01168 * cos (3 * wphase)
                          : 0.5 * (1.0 + cos (wphase));
        }

        dc += work [k] = tap;
    }

    const double unity = 1.0 / dc;
    double residue = 0.0;

    /* visiting order mid, mid-1, mid+1, mid-2, ..., 0 */
    for (int step = 0, k = mid; step < T; ++step, k = (k >= mid) ? T - k - 1 : T - k) {
        work [k] *= unity;
        row [k] = (art_s)(work [k] - residue);
        residue += row [k] - work [k];
    }
}

void artamdBuildFilterBank (int T, int F, double lowpass, int flags, artsample_t *bank)
{
    double *work = malloc (sizeof (double) * (size_t) T);

    memset (bank, 0, sizeof (art_s) * (size_t)(F + 1) * T);

    for (int f = 0; f < F; ++f)
        design_row (bank + (size_t) f * T, work, T, (double) f / F, lowpass, flags & BLACKMAN_HARRIS);

    for (int k = 0; k < T; ++k)                         /* row F: row 0 one tap later */
        bank [(size_t) F * T + (k + 1) % T] = bank [k];

    bank [T - 1] = 0.0f;                                /* clear the two window outliers */
    bank [(size_t) F * T] = 0.0f;
    free (work);
}

/* ------------------------------------------------------------------------------------------
 * Position state machine in closed form
 *
 * The reference alternates "consume one input frame" / "emit one output frame" in a scalar loop
 * (resampler.c:494-529).  Output j (counted from the start of the call) sits at ring position
 * base + j/ratio and can be emitted once inputIndex > position + T/2.  Because fl(base + fl(j/ratio))
 * is monotone in j, the number of outputs reachable with a given inputIndex is found by bisection,
 * and the ring rewinds (every 15T consumed frames) split the call into segments with their own
 * `base`.  The arithmetic that decides emit-vs-consume is the reference's own comparison, so the
 * counts and the carried position are bit-identical to the loop.
 * ---------------------------------------------------------------------------------------- */

static int plan_call (ArtamdPosition *p, int nIn, int cap, double ratio, ResampleResult *result,
                      ArtamdSegment *segs, int max_segs, int *lin_floor_out, int keep_offset);

int artamdPlanCall (ArtamdPosition *p, int nIn, int cap, double ratio, ResampleResult *result,
                    ArtamdSegment *segs, int max_segs, int *lin_floor_out)
{
    return plan_call (p, nIn, cap, ratio, result, segs, max_segs, lin_floor_out, 0);
}

/* keep_offset: the call is the silent first part of a caller's call that the library split in two (consume_silently): the
 * end-of-call re-quantisation of the position (resampler.c:533-535) belongs to the second part only */
static int plan_call (ArtamdPosition *p, int nIn, int cap, double ratio, ResampleResult *result,
                      ArtamdSegment *segs, int max_segs, int *lin_floor_out, int keep_offset)
{
    const int T = p->numTaps, half = T / 2, ring = 16 * T, drop = 15 * T;
    double base = p->outputOffset;
    int wp = p->inputIndex, flags = p->flags;
    int nseg = 0;

    if (flags & RESAMPLE_FIXED_RATIO) ratio = p->fixedRatio;
    if (flags & RESAMPLER_FLUSHED) nIn = 0;

    int lin_base = HIST_FRAMES (T) - wp;
    int lin_floor = p->floorActive ? lin_base : INT_MIN;

    if (nIn < 0) {                                      /* flush: half a window of silence is appended */
        if (ring - wp < half) {                         /* resampler.c:667-672; see DESIGN.md "reference bugs" */
            base -= drop; wp -= drop; lin_base += drop;
            p->floorActive = 1;
            lin_floor = lin_base;
        }
        flags |= RESAMPLER_FLUSHED;
        wp += half;
        nIn = 0;
    }

    unsigned int made = 0, used = 0;
    const unsigned int ucap = cap > 0 ? (unsigned int) cap : 0;
    long left = nIn;

#define PUSH_SEGMENT() do { if (nseg < max_segs) { segs [nseg].first_output = made; segs [nseg].lin_base = lin_base; \
                                                   segs [nseg].base_offset = base; } nseg++; } while (0)
    PUSH_SEGMENT ();

    if (!(ratio > 0.0))                                 /* the reference never terminates sensibly here */
        cap = 0;

    while (made < ucap && cap > 0) {
        long reach = (long) wp + left;
        int top = reach < ring ? (int) reach : ring;    /* highest inputIndex reachable in this ring epoch */
        double limit = (double)(top - half);
        unsigned int lo = made, hi = ucap;

        /* two bisection steps placed AT the arithmetic estimate (any probe inside [lo, hi) is a valid step of the same
         * monotone search, so the answer is unchanged): the answer is the estimate rounded up unless the roundings of the
         * reference's own expression disagree with it by one, so the two probes e, e + 1 normally close the interval — two
         * divisions per ring epoch instead of 21 (a 1M-frame call has 70-180 epochs with long filters, 1,456 at 48 taps, where
         * the planner, not the GPU, set the pace of a call) */
        {
            const double est = (limit - base) * ratio;
            if (est > 2.0 && est < 4.0e9) {
                const unsigned int e = (unsigned int) est;
                for (int probe = 0; probe < 2; ++probe) {
                    const unsigned int mid = probe ? e + 1 : e;
                    if (mid >= lo && mid < hi) { if (base + (double) mid / ratio < limit) lo = mid + 1; else hi = mid; }
                }
            }
        }

        while (lo < hi) {                               /* first j with position(j) >= limit */
            unsigned int mid = lo + (hi - lo) / 2;
            if (base + (double) mid / ratio < limit) lo = mid + 1; else hi = mid;
        }

        if (lo > made) {                                /* inputs consumed on the way to output lo-1 */
            long need = (long) floor (base + (double)(lo - 1) / ratio) + half + 1;
            if (need > wp) { used += (unsigned int)(need - wp); left -= need - wp; wp = (int) need; }
            made = lo;
        }

        if (made == ucap) break;

        used += (unsigned int)(top - wp); left -= top - wp; wp = top;

        if (left <= 0) break;                           /* output `made` needs input we do not have */

        base -= drop; wp -= drop; lin_base += drop;     /* ring full: rewind, then take the frame that forced it */
        wp++; used++; left--;
        PUSH_SEGMENT ();
    }
#undef PUSH_SEGMENT

    base += made ? (double) made / ratio : 0.0;

    if ((flags & RESAMPLER_SNAP_OFFSET) && !keep_offset)
        base = floor (base) + floor ((base - floor (base)) * p->numFilters + 0.5) / p->numFilters;

    p->outputOffset = base; p->inputIndex = wp; p->flags = flags;
    result->input_used = used; result->output_generated = made;
    if (lin_floor_out) *lin_floor_out = lin_floor;
    return nseg;
}

/* smallest p/q (q <= 4096) whose double quotient equals `ratio` bit for bit; 0 if none */
static void find_period (double ratio, int *p_out, int *q_out)
{
    *p_out = *q_out = 0;
    if (!(ratio > 1.0 / 4096 && ratio < 4096)) return;

    double x = ratio;
    long h0 = 0, h1 = 1, k0 = 1, k1 = 0;               /* continued-fraction convergents h/k */

    for (int it = 0; it < 32; ++it) {
        double a = floor (x);
        long h2 = (long) a * h1 + h0, k2 = (long) a * k1 + k0;
        if (k2 > 4096 || h2 > 4096 * 4096L) return;
        if ((double) h2 / (double) k2 == ratio) { if (h2 <= 4096) { *p_out = (int) h2; *q_out = (int) k2; } return; }
        h0 = h1; h1 = h2; k0 = k1; k1 = k2;
        double frac = x - a;
        if (frac < 1e-12) return;
        x = 1.0 / frac;
    }
}

/* ------------------------------------------------------------------------------------------
 * Contexts
 * ---------------------------------------------------------------------------------------- */

static void *grow (void *dev, size_t *cap, size_t need)
{
    if (need <= *cap) return dev;
    arthip_free (dev);
    size_t want = need + need / 2 + 4096;
    dev = arthip_malloc (want);
    *cap = dev ? want : 0;
    return dev;
}

static void *grow_pinned (void *host, size_t *cap, size_t need)
{
    if (need <= *cap) return host;
    arthip_host_free (host);
    size_t want = need + need / 2 + 4096;
    host = arthip_host_alloc (want);
    *cap = host ? want : 0;
    return host;
}

/* ------------------------------------------------------------------------------------------
 * Devices.  An ordinary context lives on the HIP device that is current when it is created and makes that device current
 * around every call it serves, whatever the calling thread had selected.  A context created with RESAMPLE_MULTITHREADED
 * spreads its channels over the devices of this list (artamdSetDevices, or environment ARTAMD_DEVICES="0,1,2,3"; default:
 * every visible device) — the reference's one-worker-per-channel fan-out (resampler.c:185-186, :442-470) with GPUs for
 * threads.  ARTAMD_SHARDS=n forces the number of shards (several shards per device, or sharding on a single device: the
 * way the path is tested on a one-GPU box).
 * ---------------------------------------------------------------------------------------- */
#define MAX_DEVICES ART_MAX_DEVICES
static int dev_list [MAX_DEVICES], dev_count = -1;        /* -1: not resolved yet */
static pthread_mutex_t dev_lock = PTHREAD_MUTEX_INITIALIZER;

static void resolve_devices (void)
{
    const int visible = arthip_device_count ();
    const char *env = getenv ("ARTAMD_DEVICES");

    dev_count = 0;
    if (env && *env) {
        while (*env && dev_count < MAX_DEVICES) {
            char *end;
            long d = strtol (env, &end, 10);
            if (end == env) break;
            if (d >= 0 && d < visible) dev_list [dev_count++] = (int) d;
            else fprintf (stderr, "artamd: ARTAMD_DEVICES names device %ld, %d visible: ignored\n", d, visible);
            env = (*end == ',') ? end + 1 : end;
        }
    }
    if (!dev_count)
        for (int d = 0; d < visible && d < MAX_DEVICES; ++d) dev_list [dev_count++] = d;
}

int artamdSetDevices (const int *devices, int count)
{
    const int visible = arthip_device_count ();
    int rc = 0;

    pthread_mutex_lock (&dev_lock);
    if (count <= 0 || !devices) resolve_devices ();
    else {
        for (int i = 0; i < count; ++i)
            if (devices [i] < 0 || devices [i] >= visible) rc = -1;
        if (!rc) {
            dev_count = count < MAX_DEVICES ? count : MAX_DEVICES;
            for (int i = 0; i < dev_count; ++i) dev_list [i] = devices [i];
        }
    }
    pthread_mutex_unlock (&dev_lock);
    return rc;
}

/* how many shards a MULTITHREADED context of `channels` channels gets (0 or 1: an ordinary context) and on which device
 * shard s lives.  Left to itself a context spreads over the listed devices with at least two channels per shard (a stereo
 * stream is not worth two devices' launches: ARTAMD_MIN_SHARD_CHANNELS); ARTAMD_SHARDS forces the count.  The slice kernels
 * of a shard read and write the caller's buffers on `home` in place: a device with no peer route to `home` (IOMMU,
 * containers, mixed topology — a page fault, not an error code, if it were used) is replaced by `home` itself. */
int artamd_shard_plan (int channels, int home, int *devices_out)
{
    pthread_mutex_lock (&dev_lock);
    if (dev_count < 0) resolve_devices ();
    int n = dev_count > 1 ? dev_count : 0;
    const char *env = getenv ("ARTAMD_SHARDS"), *min_env = getenv ("ARTAMD_MIN_SHARD_CHANNELS");
    if (env && *env) n = atoi (env);
    else {
        const int per = min_env && *min_env && atoi (min_env) > 0 ? atoi (min_env) : 2;
        if (n > channels / per) n = channels / per;
    }
    if (n > channels) n = channels;
    if (n > MAX_DEVICES) n = MAX_DEVICES;
    for (int s = 0; s < n; ++s) devices_out [s] = dev_count ? dev_list [s % dev_count] : 0;
    pthread_mutex_unlock (&dev_lock);
    if (home >= 0)
        for (int s = 0; s < n; ++s)
            if (devices_out [s] != home && !arthip_enable_peer (devices_out [s], home)) {
                static int warned;
                if (!warned++) fprintf (stderr, "artamd: device %d cannot address device %d's memory (no peer access): its shards stay on device %d\n", devices_out [s], home, home);
                devices_out [s] = home;
            }
    return n;
}

/* make a context's device current for the duration of a call; restored afterwards */
#define ENTER_DEVICE(hip) const int prev_device_ = arthip_current_device (); \
                          if (prev_device_ != (hip)->device) arthip_set_device ((hip)->device)
#define LEAVE_DEVICE(hip) do { if (prev_device_ != (hip)->device && prev_device_ >= 0) arthip_set_device (prev_device_); } while (0)

/* ------------------------------------------------------------------------------------------
 * Filter banks are shared: a service opens thousands of contexts with a handful of presets, a bank is up to 4 MB of HBM
 * and milliseconds of double-precision design work.  Contexts with the same (taps, filters, low-pass ratio, window) on the
 * same device reference ONE device bank (read-only to every kernel) and copy the designed rows for their own host table
 * (`filters` stays a private, writable array as in the reference).  Freed with its last context.
 * ---------------------------------------------------------------------------------------- */
typedef struct BankEntry {
    int T, F, bh, device, refs;
    double lowpass;
    art_s *host, *dev;
    struct BankEntry *next;
} BankEntry;

static BankEntry *bank_list;
static pthread_mutex_t bank_lock = PTHREAD_MUTEX_INITIALIZER;

static BankEntry *bank_acquire (int T, int F, double lowpass, int flags)
{
    const int bh = (flags & BLACKMAN_HARRIS) != 0, device = arthip_current_device ();
    const size_t bytes = sizeof (art_s) * (size_t)(F + 1) * T;
    BankEntry *e;

    pthread_mutex_lock (&bank_lock);
    for (e = bank_list; e; e = e->next)
        if (e->T == T && e->F == F && e->bh == bh && e->device == device && e->lowpass == lowpass) { e->refs++; break; }
    if (!e && (e = calloc (1, sizeof (*e)))) {
        e->T = T; e->F = F; e->bh = bh; e->device = device; e->lowpass = lowpass; e->refs = 1;
        e->host = malloc (bytes);
        e->dev = arthip_malloc (bytes);
        if (e->host) artamdBuildFilterBank (T, F, lowpass, flags, e->host);
        if (!e->host || !e->dev || arthip_h2d (e->dev, e->host, bytes, NULL) || arthip_sync (NULL)) {
            arthip_free (e->dev); free (e->host); free (e); e = NULL;
        }
        else { e->next = bank_list; bank_list = e; }
    }
    pthread_mutex_unlock (&bank_lock);
    return e;
}

static void bank_release (BankEntry *e)
{
    if (!e) return;
    pthread_mutex_lock (&bank_lock);
    if (--e->refs == 0) {
        for (BankEntry **p = &bank_list; *p; p = &(*p)->next)
            if (*p == e) { *p = e->next; break; }
        arthip_free (e->dev); free (e->host); free (e);
    }
    pthread_mutex_unlock (&bank_lock);
}

/* an ordinary context on the current device (parameters already validated, lowpassRatio normalised) */
static Resample *init_leaf (int numChannels, int numTaps, int numFilters, double lowpassRatio, int flags, int private_stream)
{
    Resample *cxt = calloc (1, sizeof (Resample));
    struct artamd_resampler *hip = calloc (1, sizeof (*hip));
    const size_t bank_count = (size_t)(numFilters + 1) * numTaps;
    const size_t hist_bytes = sizeof (art_s) * (size_t) HIST_FRAMES (numTaps) * numChannels;

    if (!cxt || !hip) {
        fprintf (stderr, "artamd: out of memory\n");
        free (cxt); free (hip);
        return NULL;
    }

    cxt->hip = hip;
    cxt->numChannels = numChannels;
    cxt->numSamples = numTaps * 16;
    cxt->numFilters = numFilters;
    cxt->numTaps = numTaps;
    cxt->flags = flags;
    cxt->lowpassRatio = lowpassRatio;
    cxt->outputOffset = numTaps / 2;
    cxt->inputIndex = numTaps;
    hip->device = arthip_current_device ();
    // (A/B runs and the PCM-level tests of programs that cannot call resampleHipSetKernel — the reference's own art / artest binaries:
    // ARTAMD_KERNEL=<n> is the kernel preference every new context starts with, see include/art_hip.h)
    { const char *env = getenv ("ARTAMD_KERNEL"); if (env && *env) hip->kernel_pref = atoi (env); }
    if (private_stream) { hip->stream = arthip_stream_create (); hip->own_stream = hip->stream != NULL; }

    /* the bank: shared on the device, a private host copy exposed through the reference's `filters` row-pointer table */
    hip->bank = bank_acquire (numTaps, numFilters, lowpassRatio, flags);
    art_s *bank = malloc (sizeof (art_s) * bank_count);
    cxt->filters = malloc (sizeof (art_s *) * (size_t)(numFilters + 1));
    if (!hip->bank || !bank || !cxt->filters) {
        fprintf (stderr, "artamd: filter bank allocation failed: %s\n", arthip_last_error ());
        free (bank); free (cxt->filters); cxt->filters = NULL;
        resampleFree (cxt);
        return NULL;
    }
    memcpy (bank, hip->bank->host, sizeof (art_s) * bank_count);
    for (int f = 0; f <= numFilters; ++f)
        cxt->filters [f] = bank + (size_t) f * numTaps;

    hip->d_bank = hip->bank->dev;
    hip->d_hist [0] = arthip_malloc (hist_bytes);
    hip->d_hist [1] = arthip_malloc (hist_bytes);
    hip->seg_cap = 64;
    hip->segs = malloc (sizeof (ArtamdSegment) * hip->seg_cap);

    if (!hip->d_hist [0] || !hip->d_hist [1] || !hip->segs || (private_stream && !hip->stream) ||
        arthip_zero (hip->d_hist [0], hist_bytes, hip->stream) || arthip_zero (hip->d_hist [1], hist_bytes, hip->stream) ||
        arthip_sync (hip->stream)) {
        fprintf (stderr, "artamd: device allocation failed: %s\n", arthip_last_error ());
        resampleFree (cxt);
        return NULL;
    }

    if (flags & EXTRAPOLATE_ENDPOINTS)
        cxt->flags |= EXTRAPOLATE_PREFILL;

    return cxt;
}

/* a sharded context: `count` ordinary contexts on devices [s], each with a contiguous channel slice and its own stream */
static Resample *init_sharded (int numChannels, int numTaps, int numFilters, double lowpassRatio, int flags, int count, const int *devices)
{
    Resample *cxt = calloc (1, sizeof (Resample));
    struct artamd_resampler *hip = calloc (1, sizeof (*hip));
    const int prev = arthip_current_device ();

    if (!cxt || !hip) { free (cxt); free (hip); return NULL; }
    cxt->hip = hip;
    cxt->numChannels = numChannels;
    cxt->numSamples = numTaps * 16;
    cxt->numFilters = numFilters;
    cxt->numTaps = numTaps;
    cxt->flags = flags;
    cxt->lowpassRatio = lowpassRatio;
    cxt->outputOffset = numTaps / 2;
    cxt->inputIndex = numTaps;
    hip->device = prev;                                  /* device-pointer calls: where the caller's buffers are expected */
    hip->shards = calloc ((size_t) count, sizeof (Resample *));
    hip->shard_first = calloc ((size_t) count + 1, sizeof (int));
    hip->ev_shard = calloc ((size_t) count, sizeof (void *));
    hip->ev_parent = arthip_order_event_create ();
    int ok = hip->shards && hip->shard_first && hip->ev_shard && hip->ev_parent;

    /* Contiguous channel slices.  A shard decides its kernels as its whole stream would (stream_channels), and the matrix-core
     * kernels are compiled for 1, 2, 4, 8, 16 and 32 channels: with every slice one of those widths all shards of a stream run the
     * same kernels — the ordinary context's bits.  So the channels are written as a sum of `count` such widths where that is possible
     * (binary digits of the channel count, the largest part halved until there are enough: 12 over 5 = 4 2 2 2 2, 8 over 3 = 4 2 2);
     * where it is not (7 channels on 2 devices) the slices are balanced.  A slice — or a stream — of any other width runs its matrix-path
     * launches in groups of a compiled width (fir_dispatch.hip, fir_in_groups): a channel's arithmetic depends neither on its group's
     * width nor on its neighbours, so channels of one stream never get different arithmetic whatever the slices are. */
    int widths [MAX_DEVICES], parts = 0;
    {
        int left = numChannels;
        while (left > 0 && parts < count) {              /* the channel count's binary digits, 32 at most per part */
            int w = 32;
            while (w > left) w >>= 1;
            widths [parts++] = w; left -= w;
        }
        if (left) parts = 0;                             /* (more parts than shards) */
    }
    while (parts && parts < count) {                     /* halve the largest part until every shard has one */
        int big = 0;
        for (int i = 1; i < parts; ++i) if (widths [i] >= widths [big]) big = i;      /* (the last of the widest: wide slices first) */
        if (widths [big] == 1) break;
        widths [big] >>= 1;
        for (int i = parts; i > big + 1; --i) widths [i] = widths [i - 1];
        widths [big + 1] = widths [big];
        ++parts;
    }
    if (parts != count) {
        const int base = numChannels / count, extra = numChannels % count;
        for (int s = 0; s < count; ++s) widths [s] = base + (s < extra ? 1 : 0);
    }
    for (int s = 0; ok && s < count; ++s) {
        const int width = widths [s];
        hip->shard_first [s + 1] = hip->shard_first [s] + width;
        arthip_set_device (devices [s]);                  /* (artamd_shard_plan has made sure it can address `prev`'s memory) */
        hip->shards [s] = init_leaf (width, numTaps, numFilters, lowpassRatio, flags & ~RESAMPLE_MULTITHREADED, 1);
        hip->ev_shard [s] = arthip_order_event_create ();
        hip->nshards = s + 1;
        ok = hip->shards [s] && hip->ev_shard [s];
        if (ok) hip->shards [s]->hip->stream_channels = numChannels;
    }
    if (prev >= 0) arthip_set_device (prev);

    if (ok) {       /* `filters`: a private copy of the rows, as in every context */
        const size_t bank_count = (size_t)(numFilters + 1) * numTaps;
        art_s *bank = malloc (sizeof (art_s) * bank_count);
        cxt->filters = malloc (sizeof (art_s *) * (size_t)(numFilters + 1));
        if (bank && cxt->filters) {
            memcpy (bank, hip->shards [0]->filters [0], sizeof (art_s) * bank_count);
            for (int f = 0; f <= numFilters; ++f) cxt->filters [f] = bank + (size_t) f * numTaps;
        }
        else { free (bank); free (cxt->filters); cxt->filters = NULL; ok = 0; }
    }
    if (!ok) {
        fprintf (stderr, "artamd: sharded context: allocation failed: %s\n", arthip_last_error ());
        resampleFree (cxt);
        return NULL;
    }
    cxt->flags = hip->shards [0]->flags | RESAMPLE_MULTITHREADED;
    hip->pool = shard_pool_create (cxt, hip->nshards);       /* (NULL: the calling thread enqueues the shards one after the other) */
    return cxt;
}

Resample *resampleInit (int numChannels, int numTaps, int numFilters, double lowpassRatio, int flags)
{
    if (lowpassRatio > 0.0 && lowpassRatio < 1.0)
        flags |= INCLUDE_LOWPASS;
    else {
        flags &= ~INCLUDE_LOWPASS;
        lowpassRatio = 1.0;
    }

    if ((numTaps & 3) || numTaps <= 0 || numTaps > 1024) {
        fprintf (stderr, "must 4-1024 filter taps, and a multiple of 4!\n");
        return NULL;
    }

    if (numFilters < 1 || numFilters > 1024) {
        fprintf (stderr, "must be 1-1024 filters!\n");
        return NULL;
    }

    if (numChannels < 1) {
        fprintf (stderr, "must have at least one channel!\n");
        return NULL;
    }

    if (arthip_device_count () < 1) {
        fprintf (stderr, "artamd: no usable HIP device (this library has no CPU path): %s\n", arthip_last_error ());
        return NULL;
    }

    { const char *env = getenv ("ARTAMD_STRICT"); if (env && *env && *env != '0') flags |= RESAMPLE_STRICT_ORDER; }

    if ((flags & RESAMPLE_MULTITHREADED) && numChannels > 1) {
        int devices [MAX_DEVICES];
        const int count = artamd_shard_plan (numChannels, arthip_current_device (), devices);
        if (count > 1)
            return init_sharded (numChannels, numTaps, numFilters, lowpassRatio, flags, count, devices);
    }

    return init_leaf (numChannels, numTaps, numFilters, lowpassRatio, flags, 0);
}

static unsigned long gcd_of (unsigned long a, unsigned long b)
{
    while (b) { unsigned long r = a % b; a = b; b = r; }
    return a;
}

Resample *resampleFixedRatioInit (int numChannels, int numTaps, int maxFilters, double sourceRate, double destinRate, int lowpassFreq, int flags)
{
    double lowpass = lowpassFreq / (destinRate / 2.0);
    const double ratio = destinRate / sourceRate;

    if (lowpassFreq > destinRate / 2.0) {
        fprintf (stderr, "lowpass frequency must be lower than destination Nyquist!\n");
        return NULL;
    }

    /* integer rates whose reduced numerator fits the filter budget need no interpolation at all */
    if (sourceRate == floor (sourceRate) && destinRate == floor (destinRate) && !(flags & NO_FILTER_REDUCTION)) {
        unsigned long phases = (unsigned long) destinRate / gcd_of ((unsigned long) sourceRate, (unsigned long) destinRate);

        if (phases <= (unsigned long) maxFilters) {
            flags &= ~SUBSAMPLE_INTERPOLATE;
            maxFilters = (int) phases;

            if (maxFilters & (maxFilters - 1))          /* phases not a power of two: re-quantise per call */
                flags |= RESAMPLER_SNAP_OFFSET;
        }
    }

    if (!lowpassFreq && (flags & INCLUDE_LOWPASS) && destinRate < sourceRate) {
        lowpass = 1.0 - (7.5 / numTaps / ratio);
        if (lowpass < 0.8) lowpass = 0.8;
        if (lowpass < ratio) lowpass = ratio;
    }

    Resample *cxt = resampleInit (numChannels, numTaps, maxFilters, lowpass * ratio, flags | RESAMPLE_FIXED_RATIO);

    if (cxt) {
        cxt->fixedRatio = destinRate / sourceRate;
        for (int k = 0; k < cxt->hip->nshards; ++k)
            cxt->hip->shards [k]->fixedRatio = cxt->fixedRatio;
    }

    return cxt;
}

static int trace_on = -1;
static void trace_report (void);

void resampleFree (Resample *cxt)
{
    if (!cxt) return;
    trace_report ();                        /* (ARTAMD_HOST_TRACE: per-context figures) */

    struct artamd_resampler *hip = cxt->hip;

    if (hip) {
        shard_pool_destroy (hip->pool);
        for (int k = 0; k < hip->nshards; ++k) {
            resampleFree (hip->shards [k]);
            arthip_event_destroy (hip->ev_shard [k]);
        }
        ENTER_DEVICE (hip);
        if (!hip->shards) arthip_sync (hip->stream);
        arthip_event_destroy (hip->ev_parent);
        free (hip->shards); free (hip->shard_first); free (hip->ev_shard);
        bank_release (hip->bank); arthip_free (hip->d_hist [0]); arthip_free (hip->d_hist [1]);
        {   /* every device buffer the context may have grown (NULL where it never did) */
            void *const device_buffers [] = { hip->d_in, hip->d_out, hip->d_tmp, hip->d_fix, hip->d_scratch, hip->d_pad, hip->d_planes, hip->d_rows,
                                              hip->d_split, hip->d_patch, hip->d_batch };
            for (size_t i = 0; i < sizeof (device_buffers) / sizeof (device_buffers [0]); ++i) arthip_free (device_buffers [i]);
        }
        if (hip->rows_cache) { arthip_fir_rows_cache_free (hip->rows_cache); free (hip->rows_cache); }
        arthip_host_free (hip->h_in); arthip_host_free (hip->h_out);
        for (int i = 0; i < hip->ev_cap; ++i) arthip_event_destroy (hip->ev [i]);
        if (hip->own_stream) arthip_stream_destroy (hip->stream);
        LEAVE_DEVICE (hip);
        free (hip->ev);
        free (hip->segs);
        free (hip);
    }

    if (cxt->filters) { free (cxt->filters [0]); free (cxt->filters); }
    free (cxt);
}

void resampleReset (Resample *cxt)
{
    struct artamd_resampler *hip = cxt->hip;

    for (int k = 0; k < hip->nshards; ++k)
        resampleReset (hip->shards [k]);

    if (!hip->nshards) {
        const size_t hist_bytes = sizeof (art_s) * (size_t) HIST_FRAMES (cxt->numTaps) * cxt->numChannels;
        ENTER_DEVICE (hip);
        arthip_zero (hip->d_hist [0], hist_bytes, hip->stream);
        arthip_zero (hip->d_hist [1], hist_bytes, hip->stream);
        LEAVE_DEVICE (hip);
    }
    hip->floor_active = 0; hip->lin_origin = 0;
    cxt->outputOffset = cxt->numTaps / 2;
    cxt->inputIndex = cxt->numTaps;

    if (cxt->flags & EXTRAPOLATE_ENDPOINTS)
        cxt->flags |= EXTRAPOLATE_PREFILL;

    cxt->flags &= ~RESAMPLER_FLUSHED;
}

double resampleGetLowpassRatio (Resample *cxt) { return cxt->lowpassRatio; }
int resampleGetNumFilters (Resample *cxt) { return cxt->numFilters; }
int resampleInterpolationUsed (Resample *cxt) { return cxt->flags & SUBSAMPLE_INTERPOLATE; }

double resampleGetPosition (Resample *cxt)
{
    return cxt->outputOffset + (cxt->numTaps / 2.0) - cxt->inputIndex;
}

void resampleAdvancePosition (Resample *cxt, double delta)
{
    if (delta < 0.0)
        fprintf (stderr, "resampleAdvancePosition() can only advance forward!\n");
    else if (!(cxt->flags & SUBSAMPLE_INTERPOLATE) && floor (delta) != delta)
        fprintf (stderr, "resampleAdvancePosition() cannot advance partial samples without interpolation!\n");
    else {
        cxt->outputOffset += delta;
        for (int k = 0; k < cxt->hip->nshards; ++k)      /* the same addition on the same value: the shards stay in step */
            cxt->hip->shards [k]->outputOffset += delta;
    }
}

/* Dry runs.  These step the position by repeated addition of 1/ratio (reference resampler.c:874, :912)
 * — deliberately NOT the division form the real run uses — so they are replayed as loops. */
unsigned int resampleGetRequiredSamples (Resample *cxt, int numOutputFrames, double ratio)
{
    const int half = cxt->numTaps / 2, drop = cxt->numSamples - cxt->numTaps;
    int wp = cxt->inputIndex;
    double pos = cxt->outputOffset;
    unsigned int used = 0;

    if (cxt->flags & RESAMPLE_FIXED_RATIO) ratio = cxt->fixedRatio;
    if (!(ratio > 0.0)) return 0;

    while (numOutputFrames > 0)
        if (pos >= wp - half) {
            if (wp == cxt->numSamples) { pos -= drop; wp -= drop; }
            wp++; used++;
        }
        else { pos += 1.0 / ratio; numOutputFrames--; }

    return used;
}

unsigned int resampleGetExpectedOutput (Resample *cxt, int numInputFrames, double ratio)
{
    const int half = cxt->numTaps / 2, drop = cxt->numSamples - cxt->numTaps;
    int wp = cxt->inputIndex;
    double pos = cxt->outputOffset;
    unsigned int made = 0;

    if (cxt->flags & RESAMPLE_FIXED_RATIO) ratio = cxt->fixedRatio;
    if (!(ratio > 0.0)) return 0;

    if (cxt->flags & RESAMPLER_FLUSHED) numInputFrames = 0;
    else if (numInputFrames < 0) wp += half;

    for (;;)
        if (pos >= wp - half) {
            if (numInputFrames <= 0) break;
            if (wp == cxt->numSamples) { pos -= drop; wp -= drop; }
            wp++; numInputFrames--;
        }
        else { pos += 1.0 / ratio; made++; }

    return made;
}

/* ------------------------------------------------------------------------------------------
 * Processing
 * ---------------------------------------------------------------------------------------- */

/* Work already enqueued on the old stream (history ping-pong, matrix-path scratch, staging) must not race with what the
 * next call puts on the new one: the old stream is drained before the swap. */
void resampleHipSetStream (Resample *cxt, void *stream)
{
    struct artamd_resampler *hip = cxt->hip;
    if (hip->stream == stream) return;
    ENTER_DEVICE (hip);
    arthip_sync (hip->stream);          /* (a sharded context's own stream carries its staging copies and the shards' completion events) */
    if (hip->own_stream) { arthip_stream_destroy (hip->stream); hip->own_stream = 0; }
    hip->stream = stream;
    LEAVE_DEVICE (hip);
}

void resampleHipSynchronize (Resample *cxt)
{
    struct artamd_resampler *hip = cxt->hip;
    for (int k = 0; k < hip->nshards; ++k) resampleHipSynchronize (hip->shards [k]);
    ENTER_DEVICE (hip);
    arthip_sync (hip->stream);
    LEAVE_DEVICE (hip);
}

void resampleHipSetCutInvariant (Resample *cxt, int on)
{
    resampleHipSetKernel (cxt, on ? ART_KERNEL_INVARIANT : ART_KERNEL_AUTO);
}

unsigned int resampleHipCutInvariantFallbacks (Resample *cxt)
{
    unsigned int n = cxt->hip->invariant_fallbacks;
    for (int k = 0; k < cxt->hip->nshards; ++k) n += resampleHipCutInvariantFallbacks (cxt->hip->shards [k]);
    return n;
}

void resampleHipSetKernel (Resample *cxt, int which)
{
    cxt->hip->kernel_pref = which;
    for (int k = 0; k < cxt->hip->nshards; ++k) resampleHipSetKernel (cxt->hip->shards [k], which);
}

void resampleHipKeepRows (Resample *cxt, int on)
{
    cxt->hip->rows_off = !on;
    for (int k = 0; k < cxt->hip->nshards; ++k) resampleHipKeepRows (cxt->hip->shards [k], on);
}

int resampleHipGetDevice (Resample *cxt) { return cxt->hip->device; }
int resampleHipNumShards (Resample *cxt) { return cxt->hip->nshards; }

int resampleHipShardInfo (Resample *cxt, int shard, int *device, int *firstChannel, int *numChannels)
{
    struct artamd_resampler *hip = cxt->hip;
    if (shard < 0 || shard >= hip->nshards) return -1;
    if (device) *device = hip->shards [shard]->hip->device;
    if (firstChannel) *firstChannel = hip->shard_first [shard];
    if (numChannels) *numChannels = hip->shard_first [shard + 1] - hip->shard_first [shard];
    return 0;
}

void resampleHipSetTiming (Resample *cxt, int enable)
{
    struct artamd_resampler *hip = cxt->hip;
    hip->timing = enable;
    hip->ev_count = 0;
    for (int k = 0; k < hip->nshards; ++k) resampleHipSetTiming (hip->shards [k], enable);
}

/* a sharded context reports its slowest shard (the shards run side by side) */
double resampleHipReadTiming (Resample *cxt, int *numLaunches)
{
    struct artamd_resampler *hip = cxt->hip;
    double total = 0.0;
    if (hip->nshards) {
        int launches = 0;
        for (int k = 0; k < hip->nshards; ++k) {
            int n = 0;
            const double ms = resampleHipReadTiming (hip->shards [k], &n);
            if (ms > total) total = ms;
            if (n > launches) launches = n;
        }
        if (numLaunches) *numLaunches = launches;
        return total;
    }
    ENTER_DEVICE (hip);
    arthip_sync (hip->stream);
    hip->prep_ms = 0.0;
    for (int i = 0; i + 2 < hip->ev_count; i += 3) {         /* (before the launch's first kernel, before its dominant kernel, after it) */
        hip->prep_ms += arthip_event_elapsed_ms (hip->ev [i], hip->ev [i + 1]);
        total += arthip_event_elapsed_ms (hip->ev [i + 1], hip->ev [i + 2]);
    }
    LEAVE_DEVICE (hip);
    if (numLaunches) *numLaunches = hip->ev_count / 3;
    hip->ev_count = 0;
    return total;
}

/* what the launches of the last resampleHipReadTiming spent BEFORE their dominant kernel: the table / staging passes of the
 * matrix-core paths (for the fixed-point kernel: peak pass + digit-plane pass) and the gaps between them */
double resampleHipReadPrepTiming (Resample *cxt)
{
    struct artamd_resampler *hip = cxt->hip;
    double worst = hip->prep_ms;
    for (int k = 0; k < hip->nshards; ++k) {
        const double ms = resampleHipReadPrepTiming (hip->shards [k]);
        if (ms > worst) worst = ms;
    }
    return worst;
}

static void *timing_event (struct artamd_resampler *hip)
{
    if (hip->ev_count == hip->ev_cap) {
        const int cap = hip->ev_cap ? hip->ev_cap * 2 : 96;
        void **grown = realloc (hip->ev, sizeof (void *) * cap);
        if (!grown) return NULL;
        hip->ev = grown;
        for (int i = hip->ev_cap; i < cap; ++i) hip->ev [i] = arthip_event_create ();
        hip->ev_cap = cap;
    }
    return hip->ev [hip->ev_count++];
}

/* Did the last call's FIR run on the fixed-point matrix kernel?  0: no; 1: yes; 2: it was enqueued and stood down (a sample
 * outside (-1.98, 1.98) or not finite: the f32 kernel behind it produced the call).  *pairsPerChunk (optional): digit-pair
 * products issued per 32-tap chunk and 32 x 32 outputs, averaged over the tile families (5 .. 13: the products with a digit plane of
 * the rows that is all zero in a chunk — the first away from the rows' centres, the second in the window's tails — are not issued).  Synchronises. */
int resampleHipLastFixedPoint (Resample *cxt, double *pairsPerChunk)
{
    struct artamd_resampler *hip = cxt->hip->nshards ? cxt->hip->shards [0]->hip : cxt->hip;
    if (pairsPerChunk) *pairsPerChunk = 0.0;
    if (!hip->last_fixed [0] || !hip->d_planes) return 0;
    ENTER_DEVICE (hip);
    int flag = 0;
    const int words = hip->last_fixed [1], chunks = hip->last_fixed [2];
    /* (the rows' masks of the first two digit planes, one after the other) */
    unsigned long long *masks = malloc (sizeof (unsigned long long) * (size_t)(words > 0 ? 2 * words : 1));
    arthip_d2h (&flag, hip->d_planes, sizeof (flag), hip->stream);
    if (masks && words > 0) arthip_d2h (masks, hip->last_masks ? hip->last_masks : (void *)((char *) hip->d_planes + ART_I8_HEAD_BYTES), sizeof (unsigned long long) * (size_t)(2 * words), hip->stream);
    arthip_sync (hip->stream);
    if (pairsPerChunk && masks && words > 0 && chunks > 0) {
        double full [2] = { 0.0, 0.0 };
        for (int pl = 0; pl < 2; ++pl)
            for (int v = 0; v < words; v += 32) {             /* a tile family's mask = OR over its 32 rows */
                unsigned long long m = 0;
                for (int r = 0; r < 32; ++r) m |= masks [pl * words + v + r];
                for (; m; m &= m - 1) full [pl] += 1.0;
            }
        /* 5 products with the rows' two lower digit planes always, 4 more per chunk whose second plane is not all zero, 4 more where the first is not */
        *pairsPerChunk = 5.0 + 4.0 * (full [0] + full [1]) / ((double)(words / 32) * chunks);
    }
    free (masks);
    LEAVE_DEVICE (hip);
    return flag == hip->last_fixed [0] ? 2 : 1;
}

/* which form of the fixed-point kernel the last call's last launch was given to (art_hip.h); 0: none */
int resampleHipLastFixedPointKernel (Resample *cxt)
{
    struct artamd_resampler *hip = cxt->hip->nshards ? cxt->hip->shards [0]->hip : cxt->hip;
    return hip->last_fixed [0] ? hip->last_fixed [3] : 0;
}

int  resampleHipLastKernel (Resample *cxt) { return cxt->hip->nshards ? cxt->hip->shards [0]->hip->last_kernel : cxt->hip->last_kernel; }

/* outputs the matrix kernels have evaluated off their canonical pattern so far (synchronises) */
unsigned int resampleHipLastHandedBack (Resample *cxt)
{
    struct artamd_resampler *hip = cxt->hip;
    unsigned int n = 0;
    if (hip->nshards) {
        for (int k = 0; k < hip->nshards; ++k) n += resampleHipLastHandedBack (hip->shards [k]);
        return n;
    }
    if (!hip->d_fix) return 0;
    ENTER_DEVICE (hip);
    arthip_d2h (&n, hip->d_fix + 1, sizeof (n), hip->stream);      /* running total since context creation */
    arthip_sync (hip->stream);
    LEAVE_DEVICE (hip);
    return n;
}


/* ------------------------------------------------------------------------------------------
 * End-point extrapolation (EXTRAPOLATE_ENDPOINTS; reference resampler.c:677-680, :691-698, :812-819).
 * The LPC fit is scalar host work on a few hundred samples, at most twice per stream; the samples it
 * produces are written into HBM and consumed by the kernels like ordinary history / input.
 * ---------------------------------------------------------------------------------------- */

/* fetch `count` frames starting at linear index `lin` of (history ++ input) into planes[c][0..count) */
static void gather_linear (Resample *cxt, const art_s *d_in, long in_pitch, int lin, int count, art_s *planes)
{
    struct artamd_resampler *hip = cxt->hip;
    const int C = cxt->numChannels, H = HIST_FRAMES (cxt->numTaps);
    art_s *tmp = malloc (sizeof (art_s) * (size_t) count * C);
    const int from_hist = lin < H ? (H - lin < count ? H - lin : count) : 0;

    if (from_hist)
        arthip_d2h (tmp, hip->d_hist [hip->cur] + (size_t) lin * C, sizeof (art_s) * (size_t) from_hist * C, hip->stream);
    if (count > from_hist) {
        const int first = lin + from_hist - H, n = count - from_hist;
        if (in_pitch)
            for (int c = 0; c < C; ++c)      /* planar: land directly in the plane */
                arthip_d2h (planes + (size_t) c * count + from_hist, d_in + (size_t) c * in_pitch + first, sizeof (art_s) * (size_t) n, hip->stream);
        else
            arthip_d2h (tmp + (size_t) from_hist * C, d_in + (size_t) first * C, sizeof (art_s) * (size_t) n * C, hip->stream);
    }
    arthip_sync (hip->stream);

    const int inter = in_pitch ? from_hist : count;     /* frames that arrived interleaved in tmp */
    for (int f = 0; f < inter; ++f)
        for (int c = 0; c < C; ++c)
            planes [(size_t) c * count + f] = tmp [(size_t) f * C + c];
    free (tmp);
}

/* Backward extrapolation into the silent pre-history, just before the first output of a stream. */
static void prefill_history (Resample *cxt, const art_s *d_in, long in_pitch)
{
    struct artamd_resampler *hip = cxt->hip;
    const int T = cxt->numTaps, C = cxt->numChannels, H = HIST_FRAMES (T), half = T / 2;
    long first_emit = (long) floor (cxt->outputOffset) + half + 1;     /* inputIndex when output 0 becomes possible */
    if (first_emit < cxt->inputIndex) first_emit = cxt->inputIndex;
    const int known = (int)(first_emit - T), extra = T - known;

    if (known < 8 || extra <= 0) return;                                /* reference resampler.c:695 / :815 */

    const int lin_known = T + H - cxt->inputIndex;                      /* ring index T in linear terms */
    art_s *planes = malloc (sizeof (art_s) * (size_t) known * C);
    art_s *older = malloc (sizeof (art_s) * (size_t) extra);
    art_s *patch = malloc (sizeof (art_s) * (size_t) extra * C);

    gather_linear (cxt, d_in, in_pitch, lin_known, known, planes);

    for (int c = 0; c < C; ++c) {
        art_extrapolate_backward (planes + (size_t) c * known, known, older, extra);
        for (int e = 0; e < extra; ++e)                                 /* older[e] is ring index T-1-e */
            patch [(size_t)(extra - 1 - e) * C + c] = older [e];
    }

    /* ring [known, T) = linear [lin_known - extra, lin_known): inside the history buffer by construction */
    arthip_h2d (hip->d_hist [hip->cur] + (size_t)(lin_known - extra) * C, patch, sizeof (art_s) * (size_t) extra * C, hip->stream);
    arthip_sync (hip->stream);
    free (planes); free (older); free (patch);
}

/* Forward extrapolation of half a window at flush time; returns a device buffer of T/2 frames x C.  *tail_out (optional)
 * receives the same samples on the host, planar [c][T/2] (caller frees). */
static const art_s *flush_tail (Resample *cxt, art_s **tail_out)
{
    struct artamd_resampler *hip = cxt->hip;
    const int T = cxt->numTaps, C = cxt->numChannels, H = HIST_FRAMES (T), half = T / 2;
    art_s *planes = malloc (sizeof (art_s) * (size_t) half * C);
    art_s *work = malloc (sizeof (art_s) * (size_t) T);
    art_s *patch = malloc (sizeof (art_s) * (size_t) half * C);
    art_s *tail = tail_out ? malloc (sizeof (art_s) * (size_t) half * C) : NULL;

    if (tail_out) *tail_out = NULL;
    if (!planes || !work || !patch || (tail_out && !tail)) {
        fprintf (stderr, "artamd: out of memory (end-point extrapolation)\n");
        free (planes); free (work); free (patch); free (tail);
        return NULL;
    }

    gather_linear (cxt, NULL, 0, H - half, half, planes);

    for (int c = 0; c < C; ++c) {
        memcpy (work, planes + (size_t) c * half, sizeof (art_s) * (size_t) half);
        art_extrapolate_forward (work, half, half);
        for (int f = 0; f < half; ++f)
            patch [(size_t) f * C + c] = work [half + f];
        if (tail) memcpy (tail + (size_t) c * half, work + half, sizeof (art_s) * (size_t) half);
    }

    hip->d_patch = grow (hip->d_patch, &hip->patch_cap, sizeof (art_s) * (size_t) half * C);
    if (hip->d_patch) {
        arthip_h2d (hip->d_patch, patch, sizeof (art_s) * (size_t) half * C, hip->stream);
        arthip_sync (hip->stream);
    }
    free (planes); free (work); free (patch);
    if (tail_out) *tail_out = tail;
    return hip->d_patch;
}

/* The stream's FIRST output is produced by the flush call itself (fewer than T/2 frames ever arrived): the reference's
 * prefill then runs after the postfill (resampler.c:775-791 then :812-819) over the real samples ++ the flush tail.
 * inputIndex is the value BEFORE the flush; `tail` = flush_tail's host copy, planar [c][T/2]. */
static void prefill_at_flush (Resample *cxt, const art_s *tail)
{
    struct artamd_resampler *hip = cxt->hip;
    const int T = cxt->numTaps, C = cxt->numChannels, H = HIST_FRAMES (T), half = T / 2;
    const int real = cxt->inputIndex - T, known = real + half, extra = T - known;

    if (real < 0 || known < 8 || extra <= 0) return;                     /* reference resampler.c:695 / :815 */

    art_s *samples = malloc (sizeof (art_s) * (size_t) known * C);       /* planar [c][known], oldest first */
    art_s *recent = malloc (sizeof (art_s) * (size_t)(real ? real : 1) * C);
    art_s *older = malloc (sizeof (art_s) * (size_t) extra);
    art_s *patch = malloc (sizeof (art_s) * (size_t) extra * C);
    if (!samples || !recent || !older || !patch) {
        fprintf (stderr, "artamd: out of memory (end-point extrapolation)\n");
        free (samples); free (recent); free (older); free (patch);
        return;
    }

    if (real) gather_linear (cxt, NULL, 0, H - real, real, recent);      /* ring [T, inputIndex) = the newest `real` history frames */

    for (int c = 0; c < C; ++c) {
        memcpy (samples + (size_t) c * known, recent + (size_t) c * real, sizeof (art_s) * (size_t) real);
        memcpy (samples + (size_t) c * known + real, tail + (size_t) c * half, sizeof (art_s) * (size_t) half);
        art_extrapolate_backward (samples + (size_t) c * known, known, older, extra);
        for (int e = 0; e < extra; ++e)                                  /* older[e] is ring index T-1-e */
            patch [(size_t)(extra - 1 - e) * C + c] = older [e];
    }

    /* ring [known, T) = linear [H - inputIndex + known, H - inputIndex + T) = [T, H - real): inside the history */
    arthip_h2d (hip->d_hist [hip->cur] + (size_t)(H - cxt->inputIndex + known) * C, patch, sizeof (art_s) * (size_t) extra * C, hip->stream);
    arthip_sync (hip->stream);
    free (samples); free (recent); free (older); free (patch);
}

static ResampleResult enqueue_call_layouts (Resample *cxt, const art_s *d_in, long in_pitch, int nIn, art_s *d_out, long out_pitch, int cap, double ratio);
static ResampleResult enqueue_call (Resample *cxt, const art_s *d_in, long in_pitch, int nIn,
                                    art_s *d_out, long out_pitch, int cap, double ratio);

/* The first `frames` input frames of a call go into the history without any output being due (the caller established
 * that): position and ring epoch advance exactly as the reference's loop would have advanced them. */
static int consume_silently (Resample *cxt, const art_s *d_in, long in_pitch, int frames, double ratio)
{
    struct artamd_resampler *hip = cxt->hip;
    const int T = cxt->numTaps, C = cxt->numChannels, H = HIST_FRAMES (T);
    ArtamdPosition pos;
    ResampleResult res;
    int lin_floor;

    pos.numTaps = T; pos.numFilters = cxt->numFilters; pos.flags = cxt->flags; pos.inputIndex = cxt->inputIndex;
    pos.floorActive = hip->floor_active; pos.outputOffset = cxt->outputOffset; pos.fixedRatio = cxt->fixedRatio;
    plan_call (&pos, frames, 1, ratio, &res, NULL, 0, &lin_floor, 1);
    if (res.output_generated || (int) res.input_used != frames) return -1;

    if (arthip_roll_history (hip->d_hist [hip->cur ^ 1], hip->d_hist [hip->cur], d_in, in_pitch, frames, H, C, hip->stream)) return -1;
    hip->cur ^= 1; hip->lin_origin += frames;
    cxt->outputOffset = pos.outputOffset; cxt->inputIndex = pos.inputIndex;
    hip->floor_active = pos.floorActive;
    return 0;
}

/* Plan one call, enqueue the FIR launches and the history roll.  `d_in` holds the call's input on
 * the device (interleaved, or planar with `in_pitch`); `d_out` receives the output likewise. */
static ResampleResult enqueue_call (Resample *cxt, const art_s *d_in, long in_pitch, int nIn,
                                    art_s *d_out, long out_pitch, int cap, double ratio)
{
    struct artamd_resampler *hip = cxt->hip;
    const int T = cxt->numTaps, C = cxt->numChannels, H = HIST_FRAMES (T);
    ResampleResult res = { 0, 0 };
    ArtamdPosition pos, trial;
    int lin_floor, nseg;

    pos.numTaps = T; pos.numFilters = cxt->numFilters; pos.flags = cxt->flags; pos.inputIndex = cxt->inputIndex;
    pos.floorActive = hip->floor_active; pos.outputOffset = cxt->outputOffset; pos.fixedRatio = cxt->fixedRatio;

    const int is_flush = nIn < 0 && !(cxt->flags & RESAMPLER_FLUSHED);
    const double eff_ratio = (cxt->flags & RESAMPLE_FIXED_RATIO) ? cxt->fixedRatio : ratio;

    /* EXTRAPOLATE_ENDPOINTS, first output of the stream only after the ring has rewound (the position was advanced by more
     * than 15 T): the reference extrapolates backwards from the samples that arrived SINCE the rewind, over the history
     * (resampler.c:812-819 with the ring's inputIndex).  Everything before the frame that makes output 0 possible is
     * consumed silently first; the rest of the call then starts inside the right ring epoch and prefills as usual. */
    if ((cxt->flags & EXTRAPOLATE_PREFILL) && !is_flush && nIn > 1 && cap > 0 && !(cxt->flags & RESAMPLER_FLUSHED)) {
        ResampleResult one;
        trial = pos;
        if (plan_call (&trial, nIn, 1, ratio, &one, NULL, 0, &lin_floor, 1) >= 2 && one.output_generated == 1 && one.input_used >= 2) {
            const int lead = (int) one.input_used - 1;
            if (consume_silently (cxt, d_in, in_pitch, lead, ratio)) {
                fprintf (stderr, "artamd: end-point extrapolation: could not advance to the first output: %s\n", arthip_last_error ());
                return res;
            }
            res = enqueue_call (cxt, in_pitch ? d_in + lead : d_in + (size_t) lead * C, in_pitch, nIn - lead, d_out, out_pitch, cap, ratio);
            res.input_used += (unsigned int) lead;
            return res;
        }
    }

    for (;;) {
        trial = pos;
        nseg = artamdPlanCall (&trial, nIn, cap, ratio, &res, hip->segs, hip->seg_cap, &lin_floor);
        if (nseg <= hip->seg_cap) break;
        ArtamdSegment *grown = realloc (hip->segs, sizeof (ArtamdSegment) * (size_t)(nseg + 16));
        if (!grown) { artamd_note_failure ("resampler: out of memory (segment table)"); res.input_used = res.output_generated = 0; return res; }
        hip->segs = grown; hip->seg_cap = nseg + 16;
    }

    if (eff_ratio != hip->period_ratio) {
        hip->period_ratio = eff_ratio;
        find_period (eff_ratio, &hip->period_out, &hip->period_in);
    }

    const int appended = is_flush ? T / 2 : (int) res.input_used;
    const art_s *flush_in = NULL;
    int rolled = 0;

    if (cxt->flags & EXTRAPOLATE_ENDPOINTS) {
        /* prefill just before the first output of the stream (resampler.c:812-819), whichever call produces it: an
         * ordinary call (a rewind right in front of output 0 leaves one known sample: nothing to extrapolate from), a
         * flush continued after it was cut short, or — below — the flush call itself */
        const int first_now = (cxt->flags & EXTRAPOLATE_PREFILL) && res.output_generated;
        if (first_now && !is_flush && (nseg == 1 || hip->segs [1].first_output > 0))
            prefill_history (cxt, nIn > 0 ? d_in : NULL, in_pitch);
        if (is_flush) {
            art_s *tail = NULL;
            flush_in = flush_tail (cxt, first_now ? &tail : NULL);
            /* (a flush that had to rewind the ring first leaves more than T known samples: nothing to prefill) */
            if (first_now && tail && trial.inputIndex == cxt->inputIndex + T / 2)
                prefill_at_flush (cxt, tail);
            free (tail);
        }
    }

    if (res.output_generated) {
        ArtFirArgs a;
        memset (&a, 0, sizeof (a));
        a.bank = hip->d_bank; a.hist = hip->d_hist [hip->cur];
        a.in = is_flush ? flush_in : d_in; a.in_pitch = is_flush ? 0 : in_pitch;
        a.in_frames = is_flush ? (flush_in ? T / 2 : 0) : (int) res.input_used;
        a.out = d_out; a.out_pitch = out_pitch;
        a.C = C; a.T = T; a.F = cxt->numFilters; a.H = H;
        a.stream_C = hip->stream_channels;
        a.lin_origin = hip->lin_origin;
        a.interpolate = (cxt->flags & SUBSAMPLE_INTERPOLATE) != 0;
        a.lowpass = (cxt->flags & INCLUDE_LOWPASS) != 0;
        /* the double-precision build has one arithmetic: EXTEND_CONVOLUTION_MATH only matters for 4-byte samples
         * (reference resampler.c:191) */
        const int extend = !ART_WIDE && (cxt->flags & EXTEND_CONVOLUTION_MATH);
        a.mode = (cxt->flags & RESAMPLE_STRICT_ORDER) ? ART_MODE_STRICT : extend ? ART_MODE_PRECISE : ART_MODE_FAST;
        if ((cxt->flags & RESAMPLE_STRICT_ORDER) && extend) a.mode |= 4;
        a.ratio = eff_ratio;
        a.period_out = hip->period_out; a.period_in = hip->period_in;
        /* the matrix-core path needs its counters and 8 MB of scratch: allocated only once a call of this context is
         * actually big enough for it (asked with stand-ins first — a service with thousands of small-block contexts never
         * pays for them) */
        int matrix_sized = 0;
        hip->last_fixed [0] = 0;
        /* the canonical period of the rows the matrix kernels keep across calls: looked after by every launch of a rational-ratio stream,
         * whichever kernel runs it (fir_matrix.hip, artfir_rows_touch) */
        if (a.period_out && a.mode == ART_MODE_FAST && !hip->rows_off && !is_flush) {
            if (!hip->rows_cache && arthip_fir_rows_cache_bytes ()) hip->rows_cache = calloc (1, arthip_fir_rows_cache_bytes ());
            a.rows_cache = hip->rows_cache;
        }
        if (a.period_out && a.mode == ART_MODE_FAST && hip->kernel_pref != ART_KERNEL_GENERAL && !is_flush) {
            ArtSegTable probe;
            probe.count = 1; probe.lin_floor = lin_floor;
            a.fix_count = a.fix_list = (unsigned int *) hip; a.scratch = hip; a.scratch_bytes = (size_t) 8 << 20; a.pad = hip;
            a.n_begin = 0; a.n_end = res.output_generated;
            matrix_sized = arthip_fir_takes_matrix_path (&a, &probe, hip->kernel_pref);
            a.fix_count = a.fix_list = NULL; a.scratch = NULL; a.scratch_bytes = 0; a.n_begin = a.n_end = 0; a.pad = NULL;
        }
        if (matrix_sized) {
            /* [0] per-launch count, [1] running total of outputs the matrix kernels evaluated off their canonical pattern
             * (diagnostics only: they are computed inside the kernel) */
            if (!hip->d_fix) {
                hip->d_fix = grow (hip->d_fix, &hip->fix_cap, 64);
                if (hip->d_fix) arthip_zero (hip->d_fix, 2 * sizeof (unsigned int), hip->stream);
            }
            if (hip->d_fix) { a.fix_count = hip->d_fix; a.fix_list = hip->d_fix + 2; a.fix_cap = 0; }
            if (!hip->d_scratch) hip->d_scratch = grow (hip->d_scratch, &hip->scratch_cap, (size_t) 8 << 20);
            a.scratch = hip->d_scratch; a.scratch_bytes = hip->d_scratch ? hip->scratch_cap : 0;
            /* digit planes for the fixed-point kernel (about the size of the call's input; without them the f32 kernels run) */
            const size_t want = arthip_fir_planes_bytes (&a, res.output_generated, hip->kernel_pref);
            if (want > hip->planes_cap) {
                hip->d_planes = grow (hip->d_planes, &hip->planes_cap, want);
                if (hip->d_planes) arthip_zero (hip->d_planes, ART_I8_HEAD_BYTES, hip->stream);
            }
            a.planes = want ? hip->d_planes : NULL; a.planes_bytes = hip->d_planes ? hip->planes_cap : 0;
            /* ... and the matrix kernels' filter rows, which outlive the call: built by the first launch of a stream, looked up by the others */
            const size_t rows_want = hip->rows_off ? 0 : arthip_fir_rows_bytes (&a, res.output_generated, hip->kernel_pref);
            if (rows_want > hip->rows_cap && hip->rows_cache) {
                hip->d_rows = grow (hip->d_rows, &hip->rows_cap, rows_want);
                arthip_fir_rows_cache_reset (hip->rows_cache);
            }
            if (rows_want && hip->d_rows && hip->rows_cache) { a.rows = hip->d_rows; a.rows_bytes = hip->rows_cap; }
            hip->last_masks = NULL; a.rows_masks_out = &hip->last_masks;
            /* calls of few tiles: room for the K-split kernel's partial sums (a grown buffer starts with its counters zeroed; the
             * old one is released behind the launches that used it: stream order) */
            const size_t split_want = want ? 0 : arthip_fir_split_bytes (&a, res.output_generated, hip->kernel_pref);
            if (split_want > hip->split_cap) {
                hip->d_split = grow (hip->d_split, &hip->split_cap, split_want);
                if (hip->d_split) arthip_zero (hip->d_split, ART_SPLIT_HEAD_BYTES, hip->stream);
            }
            a.split = split_want ? hip->d_split : NULL; a.split_bytes = hip->d_split ? hip->split_cap : 0;
            a.fixed_out = hip->last_fixed;
            /* a channel count the matrix kernels are not compiled for: room for its groups' padded copies */
            const size_t pad_want = arthip_fir_pad_bytes (&a, res.output_generated);
            if (pad_want > hip->pad_cap) hip->d_pad = grow (hip->d_pad, &hip->pad_cap, pad_want);
            a.pad = pad_want ? hip->d_pad : NULL; a.pad_bytes = hip->d_pad ? hip->pad_cap : 0;
        }

        /* A call of more ring epochs than a table holds (short filters: an epoch is a few hundred frames) is cut into launches of
         * ART_MAX_SEGS segments — unless it runs on a streaming matrix-core kernel, which follows the lattice from its first period
         * and needs the table for that period only: then the whole call is ONE launch with the first table (asked first; a launch
         * that declines after all enqueues nothing and the cut launches follow) */
        int whole = 0;
        if (matrix_sized && nseg > ART_MAX_SEGS) {
            ArtSegTable tab;
            tab.count = ART_MAX_SEGS; tab.lin_floor = lin_floor;
            for (int s = 0; s < ART_MAX_SEGS; ++s) {
                tab.first [s] = hip->segs [s].first_output; tab.lin_base [s] = hip->segs [s].lin_base; tab.base [s] = hip->segs [s].base_offset;
            }
            a.n_begin = hip->segs [0].first_output; a.n_end = res.output_generated;
            whole = arthip_fir_spans_segments (&a, &tab, hip->kernel_pref);
        }
        for (int s0 = 0; s0 < nseg; s0 += ART_MAX_SEGS) {
            const int s_tab = s0 + ART_MAX_SEGS < nseg ? s0 + ART_MAX_SEGS : nseg;     /* segments in this launch's table ... */
            const int s1 = whole ? nseg : s_tab;                                       /* ... and those it produces */
            ArtSegTable tab;

            tab.count = s_tab - s0; tab.lin_floor = lin_floor;
            for (int s = s0; s < s_tab; ++s) {
                tab.first [s - s0] = hip->segs [s].first_output;
                tab.lin_base [s - s0] = hip->segs [s].lin_base;
                tab.base [s - s0] = hip->segs [s].base_offset;
            }
            a.n_begin = hip->segs [s0].first_output;
            a.n_end = s1 < nseg ? hip->segs [s1].first_output : res.output_generated;
            if (a.n_end > a.n_begin) {
                void *ev_pre = hip->timing ? timing_event (hip) : NULL;
                a.ev_start = hip->timing ? timing_event (hip) : NULL;
                a.ev_stop = hip->timing ? timing_event (hip) : NULL;
                if (ev_pre) arthip_event_record (ev_pre, hip->stream);
                /* the last FIR launch of the call may take the history roll along (one launch less on the stream) */
                a.roll_dst = (s1 == nseg && appended > 0) ? hip->d_hist [hip->cur ^ 1] : NULL;
                a.roll_appended = appended;
                a.segs_truncated = whole;
                int k = arthip_fir (&a, &tab, hip->kernel_pref, hip->stream);
                a.segs_truncated = 0;
                if (k == -2 && whole) {                /* (declined: nothing enqueued — again, cut) */
                    if (hip->timing) hip->ev_count -= 3;        /* (the three events go back: only ev_pre was recorded, none is read) */
                    whole = 0; s0 = -ART_MAX_SEGS; continue;
                }
                if (k >= 0 && (k & ART_FIR_ROLLED)) { rolled = 1; k &= ~ART_FIR_ROLLED; }
                if (k < 0) {
                    /* nothing of the stream has moved: the position and the history ring are committed below, behind the call's last
                     * launch — a caller sees { 0, 0 }, the count in artamdErrorCount, and with ARTAMD_ABORT_ON_ERROR=1 the process stops here */
                    artamd_note_failure ("resampler: FIR launch failed");
                    res.input_used = res.output_generated = 0; return res;
                }
                hip->last_kernel = k;
                /* the cut-invariant policy: a launch of a rational-ratio stream that could not run anchored on the matrix cores went to the general
                 * kernel — still independent of the cut by itself, but another arithmetic than the stream's other outputs: counted (art_hip.h) */
                if (hip->kernel_pref == ART_KERNEL_INVARIANT && k == ART_KERNEL_GENERAL && a.period_out && a.mode == ART_MODE_FAST && !is_flush)
                    hip->invariant_fallbacks++;
            }
            if (whole) break;
        }
    }

    if (appended > 0) {
        if (!rolled)
            arthip_roll_history (hip->d_hist [hip->cur ^ 1], hip->d_hist [hip->cur], is_flush ? flush_in : d_in, is_flush ? 0 : in_pitch, appended, H, C, hip->stream);
        hip->cur ^= 1; hip->lin_origin += appended;
    }

    cxt->outputOffset = trial.outputOffset; cxt->inputIndex = trial.inputIndex;
    cxt->flags = (cxt->flags & ~(RESAMPLER_FLUSHED | EXTRAPOLATE_PREFILL)) | (trial.flags & RESAMPLER_FLUSHED) |
                 ((res.output_generated == 0) ? (cxt->flags & EXTRAPOLATE_PREFILL) : 0);
    hip->floor_active = trial.floorActive;
    return res;
}

/* ---- many independent streams, one launch -------------------------------------------------------------------------
 * A service that resamples hundreds of streams in small blocks is launch-bound one call at a time.  This entry point
 * plans every context's call on the host exactly as the single call does, gathers those the general kernel would run
 * (any ratio per stream, default or EXTEND mode, ordinary call, on the stream of cxts [0]) into one launch per kernel
 * variant — each stream cut into the tiles its own launch would use, so the samples are identical — and simply makes
 * the remaining calls (flushes, strict mode, endpoint extrapolation, calls big enough for the matrix-core path, other
 * streams) one by one.  results [i] is what resampleProcessInterleavedDevice (cxts [i], ...) would have returned. */
static int batch_plan (Resample *cxt, const art_s *d_in, int nIn, art_s *d_out, int cap, double ratio, void *lead_stream,
                       ArtFirArgs *a, ArtSegTable *tab, ResampleResult *res, ArtamdPosition *trial)
{
    struct artamd_resampler *hip = cxt->hip;
    const int T = cxt->numTaps, C = cxt->numChannels, H = HIST_FRAMES (T);
    ArtamdPosition pos;
    int lin_floor, nseg;

    if (nIn < 0 || hip->stream != lead_stream || hip->timing || hip->nshards || hip->device != arthip_current_device () ||
        (cxt->flags & (EXTRAPOLATE_ENDPOINTS | RESAMPLE_STRICT_ORDER | RESAMPLER_FLUSHED))) return 0;

    pos.numTaps = T; pos.numFilters = cxt->numFilters; pos.flags = cxt->flags; pos.inputIndex = cxt->inputIndex;
    pos.floorActive = hip->floor_active; pos.outputOffset = cxt->outputOffset; pos.fixedRatio = cxt->fixedRatio;
    const double eff_ratio = (cxt->flags & RESAMPLE_FIXED_RATIO) ? cxt->fixedRatio : ratio;

    for (;;) {
        *trial = pos;
        nseg = artamdPlanCall (trial, nIn, cap, ratio, res, hip->segs, hip->seg_cap, &lin_floor);
        if (nseg <= hip->seg_cap) break;
        ArtamdSegment *grown = realloc (hip->segs, sizeof (ArtamdSegment) * (size_t)(nseg + 16));
        if (!grown) return 0;                                 /* (the one-by-one path reports it) */
        hip->segs = grown; hip->seg_cap = nseg + 16;
    }
    if (nseg > arthip_fir_batch_max_segments () || res->output_generated == 0) return 0;

    if (eff_ratio != hip->period_ratio) {
        hip->period_ratio = eff_ratio;
        find_period (eff_ratio, &hip->period_out, &hip->period_in);
    }

    memset (a, 0, sizeof (*a));
    a->bank = hip->d_bank; a->hist = hip->d_hist [hip->cur];
    a->in = d_in; a->in_pitch = 0; a->in_frames = (int) res->input_used;
    a->out = d_out; a->out_pitch = 0;
    a->C = C; a->T = T; a->F = cxt->numFilters; a->H = H;
    a->stream_C = hip->stream_channels;
    a->interpolate = (cxt->flags & SUBSAMPLE_INTERPOLATE) != 0;
    a->lowpass = (cxt->flags & INCLUDE_LOWPASS) != 0;
    a->mode = (!ART_WIDE && (cxt->flags & EXTEND_CONVOLUTION_MATH)) ? ART_MODE_PRECISE : ART_MODE_FAST;
    a->ratio = eff_ratio;
    a->period_out = hip->period_out; a->period_in = hip->period_in;
    a->lin_origin = hip->lin_origin;

    tab->count = nseg; tab->lin_floor = lin_floor;
    for (int s = 0; s < nseg; ++s) {
        tab->first [s] = hip->segs [s].first_output;
        tab->lin_base [s] = hip->segs [s].lin_base;
        tab->base [s] = hip->segs [s].base_offset;
    }
    a->n_begin = hip->segs [0].first_output; a->n_end = res->output_generated;

    /* would the single call take the matrix-core path?  (it has its counters and scratch whenever the ratio is
     * rational, the mode default and the kernel not pinned: stand-ins suffice for the question) */
    if (a->period_out && a->mode == ART_MODE_FAST && hip->kernel_pref != ART_KERNEL_GENERAL) {
        a->fix_count = (unsigned int *) hip; a->fix_list = (unsigned int *) hip; a->scratch = hip; a->scratch_bytes = (size_t) 8 << 20;
        if (arthip_fir_takes_matrix_path (a, tab, hip->kernel_pref)) return 0;
        a->fix_count = a->fix_list = NULL; a->scratch = NULL; a->scratch_bytes = 0;
    }

    const int appended = (int) res->input_used;
    a->roll_dst = appended > 0 ? hip->d_hist [hip->cur ^ 1] : NULL;      /* the launch takes the history roll along */
    a->roll_appended = appended;
    return 1;
}

int resampleProcessBatchInterleavedDevice (Resample *const *cxts, int n, const artsample_t *const *d_inputs, const int *numInputFrames,
                                           artsample_t *const *d_outputs, const int *numOutputFrames, const double *ratios,
                                           ResampleResult *results)
{
    if (n <= 0) return 0;
    struct artamd_resampler *lead = cxts [0]->hip;
    ENTER_DEVICE (lead);
    ArtFirArgs *args = malloc (sizeof (ArtFirArgs) * (size_t) n);
    ArtSegTable *tabs = malloc (sizeof (ArtSegTable) * (size_t) n);
    ArtamdPosition *trials = malloc (sizeof (ArtamdPosition) * (size_t) n);
    int *owner = malloc (sizeof (int) * (size_t) n);
    int gathered = 0, rc = -1;

    if (!args || !tabs || !trials || !owner) goto out;
    {   /* a context may appear only once: stamp each with this call's number (one pass) */
        static unsigned long calls;
        const unsigned long stamp = __atomic_add_fetch (&calls, 1, __ATOMIC_RELAXED);
        for (int i = 0; i < n; ++i) {
            if (cxts [i]->hip->batch_stamp == stamp) { fprintf (stderr, "artamd: resample batch: a context appears twice\n"); goto out; }
            cxts [i]->hip->batch_stamp = stamp;
        }
    }

    for (int i = 0; i < n; ++i) {
        if (batch_plan (cxts [i], d_inputs [i], numInputFrames [i], d_outputs [i], numOutputFrames [i], ratios [i], lead->stream,
                        &args [gathered], &tabs [gathered], &results [i], &trials [gathered]))
            owner [gathered++] = i;
        else
            results [i] = resampleProcessInterleavedDevice (cxts [i], d_inputs [i], numInputFrames [i], d_outputs [i], numOutputFrames [i], ratios [i]);
    }

    if (gathered) {
        lead->d_batch = grow (lead->d_batch, &lead->batch_cap, arthip_fir_batch_item_bytes () * (size_t) gathered);
        if (!lead->d_batch || arthip_fir_batch (args, tabs, gathered, lead->d_batch, lead->stream)) {
            fprintf (stderr, "artamd: resample batch launch failed: %s\n", arthip_last_error ());
            for (int k = 0; k < gathered; ++k) results [owner [k]].input_used = results [owner [k]].output_generated = 0;
            goto out;
        }
        for (int k = 0; k < gathered; ++k) {
            Resample *cxt = cxts [owner [k]];
            if (args [k].roll_dst) { cxt->hip->cur ^= 1; cxt->hip->lin_origin += args [k].roll_appended; }
            cxt->outputOffset = trials [k].outputOffset; cxt->inputIndex = trials [k].inputIndex;
            cxt->flags = (cxt->flags & ~(RESAMPLER_FLUSHED | EXTRAPOLATE_PREFILL)) | (trials [k].flags & RESAMPLER_FLUSHED);
            cxt->hip->floor_active = trials [k].floorActive;
            cxt->hip->last_kernel = ART_KERNEL_GENERAL;
        }
    }
    rc = 0;
out:
    free (args); free (tabs); free (trials); free (owner);
    LEAVE_DEVICE (lead);
    return rc;
}

/* what a call would consume / produce, without touching the context */
static ResampleResult peek_call (Resample *cxt, int nIn, int cap, double ratio)
{
    ResampleResult peek;
    ArtamdPosition pos;
    int dummy_floor;

    pos.numTaps = cxt->numTaps; pos.numFilters = cxt->numFilters; pos.flags = cxt->flags; pos.inputIndex = cxt->inputIndex;
    pos.floorActive = cxt->hip->floor_active; pos.outputOffset = cxt->outputOffset; pos.fixedRatio = cxt->fixedRatio;
    artamdPlanCall (&pos, nIn, cap, ratio, &peek, NULL, 0, &dummy_floor);
    return peek;
}

/* a sharded context mirrors the position of its shards (they all hold the same one) */
static ResampleResult shards_agree (Resample *cxt, const ResampleResult *per_shard)
{
    struct artamd_resampler *hip = cxt->hip;
    const Resample *first = hip->shards [0];
    ResampleResult res = per_shard [0];

    for (int k = 1; k < hip->nshards; ++k)
        if (per_shard [k].input_used != res.input_used || per_shard [k].output_generated != res.output_generated ||
            hip->shards [k]->inputIndex != first->inputIndex || hip->shards [k]->outputOffset != first->outputOffset) {
            fprintf (stderr, "artamd: sharded context: shard %d disagrees with shard 0 (a launch failed?)\n", k);
            res.input_used = res.output_generated = 0;
        }
    cxt->outputOffset = first->outputOffset; cxt->inputIndex = first->inputIndex;
    cxt->flags = first->flags | RESAMPLE_MULTITHREADED;
    return res;
}

/* Device-pointer call on a sharded context: the caller's buffers live on the context's own device; every shard pulls its
 * channel slice (strided rows, peer-to-peer over xGMI when the shard sits on another GPU), runs, and pushes its slice of
 * the output back.  Ordered after the context's stream, and the context's stream continues only when all shards are done.
 * Planar buffers need no copies at all: a shard's channels are a contiguous run of planes.
 * A shard's part is six or seven enqueues (wait, slice in, staging / prepare, FIR, slice out, record): ~20 us of host time.  One
 * after the other from the calling thread that is 160 us per call on eight devices — more than a 4-channel shard's 1M-frame call
 * takes on its GPU — so every shard has a WORKER THREAD (round 3's verdict): the caller posts the call's arguments, the workers
 * enqueue their shards side by side (each on its own device and stream; the HIP runtime is entered from several threads at once,
 * which it allows), the caller waits for the enqueues — not the GPUs — and lets the context's stream wait for the shards' events.
 * (Where the shards sit on one device the calling thread does it all, as in rounds 2-3: shard_pool_create.) */
struct shard_job {
    const art_s *d_in; long in_pitch; int nIn; art_s *d_out; long out_pitch; int cap; double ratio;
    ResampleResult peek;
};

static ResampleResult shard_part (Resample *cxt, int k, const struct shard_job *j, int *failed)
{
    struct artamd_resampler *hip = cxt->hip;
    Resample *sh = hip->shards [k];
    struct artamd_resampler *sp = sh->hip;
    const int C = cxt->numChannels, first = hip->shard_first [k], width = sh->numChannels;
    ResampleResult res = { 0, 0 };

    arthip_set_device (sp->device);
    arthip_stream_wait_event (sp->stream, hip->ev_parent);
    if (j->in_pitch && j->out_pitch)
        res = enqueue_call_layouts (sh, j->d_in ? j->d_in + (size_t) first * j->in_pitch : NULL, j->in_pitch, j->nIn,
                                    j->d_out + (size_t) first * j->out_pitch, j->out_pitch, j->cap, j->ratio);
    else if (j->in_pitch || j->out_pitch) {
        /* one side planar, the other interleaved (a pitch of 0): the planar side is the shard's own run of planes; the interleaved side
         * is a strided slice of the caller's stream-wide frames and goes through the shard's slice staging, as in the branch below */
        const int wpw = (int)(sizeof (art_s) / 4);
        if (j->out_pitch) {                                     /* interleaved in, planar out */
            sp->d_in = grow (sp->d_in, &sp->in_cap, sizeof (art_s) * (size_t) j->peek.input_used * width);
            if (j->peek.input_used && !sp->d_in) { *failed = 1; arthip_event_record (hip->ev_shard [k], sp->stream); return res; }
            if (j->peek.input_used && j->d_in)
                arthip_slice_copy (sp->d_in, (size_t) width * wpw, j->d_in + first, (size_t) C * wpw, width * wpw, j->peek.input_used, sp->stream);
            res = enqueue_call_layouts (sh, j->d_in ? sp->d_in : NULL, 0, j->nIn, j->d_out + (size_t) first * j->out_pitch, j->out_pitch, j->cap, j->ratio);
        }
        else {                                                  /* planar in, interleaved out */
            sp->d_out = grow (sp->d_out, &sp->out_cap, sizeof (art_s) * ((size_t) j->peek.output_generated + 16) * width);
            if (j->peek.output_generated && !sp->d_out) { *failed = 1; arthip_event_record (hip->ev_shard [k], sp->stream); return res; }
            res = enqueue_call_layouts (sh, j->d_in ? j->d_in + (size_t) first * j->in_pitch : NULL, j->in_pitch, j->nIn, sp->d_out, 0, j->cap, j->ratio);
            if (res.output_generated)
                arthip_slice_copy (j->d_out + first, (size_t) C * wpw, sp->d_out, (size_t) width * wpw, width * wpw, res.output_generated, sp->stream);
        }
    }
    else {
        sp->d_in = grow (sp->d_in, &sp->in_cap, sizeof (art_s) * (size_t) j->peek.input_used * width);
        sp->d_out = grow (sp->d_out, &sp->out_cap, sizeof (art_s) * (size_t) j->peek.output_generated * width);
        if ((j->peek.input_used && !sp->d_in) || (j->peek.output_generated && !sp->d_out)) { *failed = 1; arthip_event_record (hip->ev_shard [k], sp->stream); return res; }
        const int wpw = (int)(sizeof (art_s) / 4);              /* 4-byte words per sample */
        if (j->peek.input_used && j->d_in)
            arthip_slice_copy (sp->d_in, (size_t) width * wpw, j->d_in + first, (size_t) C * wpw, width * wpw, j->peek.input_used, sp->stream);
        res = enqueue_call (sh, sp->d_in, 0, j->nIn, sp->d_out, 0, j->cap, j->ratio);
        arthip_slice_copy (j->d_out + first, (size_t) C * wpw, sp->d_out, (size_t) width * wpw, width * wpw, res.output_generated, sp->stream);
    }
    arthip_event_record (hip->ev_shard [k], sp->stream);
    return res;
}

struct shard_pool {
    Resample *cxt;
    int n, quit;
    unsigned long call;                      /* number of the call the workers are to make (posted under the lock) */
    int pending;                             /* workers that have not finished it yet */
    struct shard_job job;
    ResampleResult res [MAX_DEVICES];
    int failed [MAX_DEVICES];
    char err [MAX_DEVICES] [256];            /* a worker's own error text (arthip_last_error is per thread) for the caller to report */
    pthread_t thread [MAX_DEVICES];
    int started;
    pthread_mutex_t lock;
    pthread_cond_t go, done;
};

struct shard_worker_arg { struct shard_pool *pool; int k; };

static void *shard_worker (void *p)
{
    struct shard_worker_arg *wa = p;
    struct shard_pool *pool = wa->pool;
    const int k = wa->k;
    unsigned long seen = 0;
    free (wa);
    pthread_mutex_lock (&pool->lock);
    for (;;) {
        while (!pool->quit && pool->call == seen) pthread_cond_wait (&pool->go, &pool->lock);
        if (pool->quit) break;
        seen = pool->call;
        pthread_mutex_unlock (&pool->lock);
        pool->failed [k] = 0;
        arthip_set_last_error (NULL);
        pool->res [k] = shard_part (pool->cxt, k, &pool->job, &pool->failed [k]);
        snprintf (pool->err [k], sizeof (pool->err [k]), "%s", arthip_last_error ());
        pthread_mutex_lock (&pool->lock);
        if (--pool->pending == 0) pthread_cond_signal (&pool->done);
    }
    pthread_mutex_unlock (&pool->lock);
    return NULL;
}

static struct shard_pool *shard_pool_create (Resample *cxt, int n)
{
    /* threads where the shards sit on DIFFERENT devices (each device's queue is fed by its own thread); several shards on one device
     * — ARTAMD_SHARDS on a one-GPU box — are enqueued by the caller: eight threads entering the runtime for one device contend for
     * it (measured, tools/bench_sharded_context.py, 16,384-frame calls: 231 us with threads, 205 without).  ARTAMD_SHARD_THREADS=1 /
     * =0 forces either. */
    const char *e = getenv ("ARTAMD_SHARD_THREADS");
    int distinct = 0;
    for (int k = 0; k < n; ++k) {
        int seen = 0;
        for (int i = 0; i < k; ++i) seen |= cxt->hip->shards [i]->hip->device == cxt->hip->shards [k]->hip->device;
        distinct += !seen;
    }
    if (n < 2 || (e && *e == '0') || (distinct < 2 && !(e && *e == '1'))) return NULL;
    struct shard_pool *pool = calloc (1, sizeof (*pool));
    if (!pool) return NULL;
    pool->cxt = cxt; pool->n = n;
    pthread_mutex_init (&pool->lock, NULL); pthread_cond_init (&pool->go, NULL); pthread_cond_init (&pool->done, NULL);
    for (int k = 0; k < n; ++k) {
        struct shard_worker_arg *wa = malloc (sizeof (*wa));
        if (!wa) break;
        wa->pool = pool; wa->k = k;
        if (pthread_create (&pool->thread [k], NULL, shard_worker, wa)) { free (wa); break; }
        pool->started = k + 1;
    }
    if (pool->started != n) { shard_pool_destroy (pool); return NULL; }
    return pool;
}

static void shard_pool_destroy (struct shard_pool *pool)
{
    if (!pool) return;
    pthread_mutex_lock (&pool->lock);
    pool->quit = 1;
    pthread_cond_broadcast (&pool->go);
    pthread_mutex_unlock (&pool->lock);
    for (int k = 0; k < pool->started; ++k) pthread_join (pool->thread [k], NULL);
    pthread_mutex_destroy (&pool->lock); pthread_cond_destroy (&pool->go); pthread_cond_destroy (&pool->done);
    free (pool);
}

static ResampleResult sharded_device_call (Resample *cxt, const art_s *d_in, long in_pitch, int nIn, art_s *d_out, long out_pitch,
                                           int cap, double ratio)
{
    struct artamd_resampler *hip = cxt->hip;
    const int prev = arthip_current_device ();
    ResampleResult per_shard [MAX_DEVICES], res = { 0, 0 };
    int failed = 0;
    struct shard_job job = { d_in, in_pitch, nIn, d_out, out_pitch, cap, ratio, peek_call (cxt, nIn, cap, ratio) };

    arthip_set_device (hip->device);
    arthip_event_record (hip->ev_parent, hip->stream);

    struct shard_pool *pool = hip->pool;
    if (pool) {
        pthread_mutex_lock (&pool->lock);
        pool->job = job; pool->pending = pool->n; ++pool->call;
        pthread_cond_broadcast (&pool->go);
        while (pool->pending) pthread_cond_wait (&pool->done, &pool->lock);
        pthread_mutex_unlock (&pool->lock);
        for (int k = 0; k < hip->nshards; ++k) {
            per_shard [k] = pool->res [k]; failed |= pool->failed [k];
            /* (a worker that failed, or whose launch made nothing where shard 0's made something: its thread's error text becomes this thread's) */
            if (pool->failed [k] || per_shard [k].output_generated != pool->res [0].output_generated) arthip_set_last_error (pool->err [k]);
        }
    }
    else
        for (int k = 0; k < hip->nshards; ++k) { int f = 0; per_shard [k] = shard_part (cxt, k, &job, &f); failed |= f; }

    arthip_set_device (hip->device);
    for (int k = 0; k < hip->nshards; ++k)
        arthip_stream_wait_event (hip->stream, hip->ev_shard [k]);
    if (prev >= 0) arthip_set_device (prev);

    res = shards_agree (cxt, per_shard);
    if (failed) { artamd_note_failure ("resampler: sharded context: device allocation failed"); res.input_used = res.output_generated = 0; }
    return res;
}

/* enqueue_call for device buffers of either layout (the context's device is current) */
static ResampleResult enqueue_call_layouts (Resample *cxt, const art_s *d_in, long in_pitch, int nIn, art_s *d_out, long out_pitch, int cap, double ratio)
{
    struct artamd_resampler *hip = cxt->hip;
    /* Planar device buffers.  The matrix-core path reads and writes interleaved frames only, and a big planar call on the general
     * kernel is 4-7 x slower than the same call interleaved (8 ch x 988 taps, 1M frames: 960 against 140 us).  Such a call goes
     * through the context's interleaved staging buffers — two transposing copies on the device, ~4 % of the call — and so does a
     * call with only one planar side.  (Small calls stay as they are: the general kernel takes planes as they come.) */
    static int planar_off = -1;
    if (planar_off < 0) { const char *e = getenv ("ARTAMD_PLANAR_DIRECT"); planar_off = e && *e && *e != '0'; }
    if ((in_pitch || out_pitch) && !planar_off && nIn > 0 && cap > 0 && d_in && d_out &&
        ((double) nIn * (hip->stream_channels > cxt->numChannels ? hip->stream_channels : cxt->numChannels) * cxt->numTaps >= 2.0e8 ||       /* (a shard: its whole stream's size) */
         hip->kernel_pref == ART_KERNEL_INVARIANT)) {            /* (the cut-invariant policy: every call, whatever its size, on the same kernel) */
        const int C = cxt->numChannels;
        const art_s *in_i = d_in; art_s *out_i = d_out;
        int ok = 1;
        if (in_pitch) {
            hip->d_in = grow (hip->d_in, &hip->in_cap, sizeof (art_s) * (size_t) nIn * C);
            ok = hip->d_in && !arthip_interleave (hip->d_in, d_in, in_pitch, nIn, C, hip->stream);
            in_i = hip->d_in;
        }
        if (ok && out_pitch) {
            /* (room for the frames the call will make, not for the caller's whole capacity) */
            const ResampleResult pk = peek_call (cxt, nIn, cap, ratio);
            hip->d_out = grow (hip->d_out, &hip->out_cap, sizeof (art_s) * ((size_t) pk.output_generated + 16) * C);
            ok = hip->d_out != NULL;
            out_i = hip->d_out;
        }
        if (ok) {
            const ResampleResult res = enqueue_call (cxt, in_i, 0, nIn, out_i, 0, cap, ratio);
            if (out_pitch && res.output_generated) arthip_deinterleave (d_out, out_pitch, out_i, (int) res.output_generated, C, hip->stream);
            return res;
        }
    }
    return enqueue_call (cxt, d_in, in_pitch, nIn, d_out, out_pitch, cap, ratio);
}

static ResampleResult device_call (Resample *cxt, const art_s *d_in, long in_pitch, int nIn, art_s *d_out, long out_pitch, int cap, double ratio)
{
    struct artamd_resampler *hip = cxt->hip;
    if (hip->nshards) return sharded_device_call (cxt, d_in, in_pitch, nIn, d_out, out_pitch, cap, ratio);
    ENTER_DEVICE (hip);
    const ResampleResult res = enqueue_call_layouts (cxt, d_in, in_pitch, nIn, d_out, out_pitch, cap, ratio);
    LEAVE_DEVICE (hip);
    return res;
}

ResampleResult resampleProcessInterleavedDevice (Resample *cxt, const artsample_t *d_input, int numInputFrames,
                                                 artsample_t *d_output, int numOutputFrames, double ratio)
{
    return device_call (cxt, d_input, 0, numInputFrames, d_output, 0, numOutputFrames, ratio);
}

ResampleResult resampleProcessPlanarDevice (Resample *cxt, const artsample_t *d_input, long inputPitch, int numInputFrames,
                                            artsample_t *d_output, long outputPitch, int numOutputFrames, double ratio)
{
    return device_call (cxt, d_input, inputPitch, numInputFrames, d_output, outputPitch, numOutputFrames, ratio);
}

ResampleResult resampleProcessAndFlushInterleavedDevice (Resample *cxt, const artsample_t *d_input, int numInputFrames,
                                                         artsample_t *d_output, int numOutputFrames, double ratio)
{
    ResampleResult res = resampleProcessInterleavedDevice (cxt, d_input, numInputFrames, d_output, numOutputFrames, ratio);

    if ((numInputFrames -= res.input_used) != 0 || (numOutputFrames -= res.output_generated) == 0)
        return res;

    ResampleResult tail = resampleProcessInterleavedDevice (cxt, NULL, -1, d_output + (size_t) res.output_generated * cxt->numChannels,
                                                            numOutputFrames, ratio);
    res.output_generated += tail.output_generated;
    return res;
}

/* ------------------------------------------------------------------------------------------
 * Host-pointer calls (what ART and artest use).  host_begin stages the call's input in HBM, enqueues the call and starts
 * the copy back; host_end waits and delivers.  A sharded context begins on all its shards before it ends any, so the
 * devices work side by side.  The caller's frames are `in_stride` / `out_stride` samples apart (a shard sees its channel
 * slice of a wider stream: the de-interleaving happens on the way into its HBM).
 *   up to STAGE_LIMIT bytes: packed by the CPU into page-locked staging, ONE dense DMA each way;
 *   beyond: straight from / to the caller's memory (strided rows: a 2-D copy), the runtime pipelines the pages.
 * ---------------------------------------------------------------------------------------- */
typedef struct { ResampleResult res; int staged_out, failed; } HostPending;

/* ARTAMD_HOST_TRACE=1: where a host-pointer call spends its time (host clock, accumulated, printed by resampleFree) */
static double trace_acc [8]; static long trace_calls;
static inline double trace_now (void) { struct timespec ts; clock_gettime (CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
#define TRACE_MARK(slot) do { if (trace_on > 0) { const double now_ = trace_now (); trace_acc [slot] += now_ - trace_t_; trace_t_ = now_; } } while (0)
static void trace_report (void)
{
    if (trace_on > 0 && trace_calls) {
        fprintf (stderr, "artamd host trace, us per call over %ld calls: plan+grow %.1f | pack %.1f | H2D enqueue %.1f | plan+launch %.1f | D2H enqueue %.1f | "
                 "wait %.1f | unpack %.1f\n", trace_calls, trace_acc [0] / trace_calls, trace_acc [1] / trace_calls, trace_acc [2] / trace_calls,
                 trace_acc [3] / trace_calls, trace_acc [4] / trace_calls, trace_acc [5] / trace_calls, trace_acc [6] / trace_calls);
        memset (trace_acc, 0, sizeof (trace_acc)); trace_calls = 0;
    }
}

static void host_begin (Resample *cxt, const art_s *input, int in_stride, const art_s *const *planes, int nIn,
                        art_s *output, int out_stride, art_s *const *out_planes, int cap, double ratio, HostPending *pend)
{
    struct artamd_resampler *hip = cxt->hip;
    const int C = cxt->numChannels;
    const ResampleResult peek = peek_call (cxt, nIn, cap, ratio);
    const size_t in_samples = (size_t) peek.input_used * C, out_samples = (size_t) peek.output_generated * C;
    const char *limit_env = getenv ("ARTAMD_STAGE_LIMIT");
    const int staged = sizeof (art_s) * (in_samples + out_samples) <= (limit_env && *limit_env ? (size_t) strtoull (limit_env, NULL, 10) : STAGE_LIMIT);

    pend->res.input_used = pend->res.output_generated = 0; pend->staged_out = 0; pend->failed = 1;
    if (trace_on < 0) { const char *e = getenv ("ARTAMD_HOST_TRACE"); trace_on = e && *e && *e != '0'; if (trace_on) atexit (trace_report); }
    double trace_t_ = trace_on > 0 ? trace_now () : 0.0;
    trace_calls += trace_on > 0;

    hip->d_in = grow (hip->d_in, &hip->in_cap, sizeof (art_s) * in_samples);
    hip->d_out = grow (hip->d_out, &hip->out_cap, sizeof (art_s) * out_samples);
    if (staged) {
        hip->h_in = grow_pinned (hip->h_in, &hip->h_in_cap, sizeof (art_s) * in_samples);
        hip->h_out = grow_pinned (hip->h_out, &hip->h_out_cap, sizeof (art_s) * out_samples);
    }
    else if (planes || out_planes) {
        const size_t big = in_samples > out_samples ? in_samples : out_samples;
        hip->d_tmp = grow (hip->d_tmp, &hip->tmp_cap, sizeof (art_s) * big);
    }
    if ((in_samples && !hip->d_in) || (out_samples && !hip->d_out) || (staged && ((in_samples && !hip->h_in) || (out_samples && !hip->h_out))) ||
        (!staged && (planes || out_planes) && !hip->d_tmp)) {
        fprintf (stderr, "artamd: staging allocation failed: %s\n", arthip_last_error ());
        return;
    }

    TRACE_MARK (0);
    if (in_samples && staged) {
        art_s *dst = hip->h_in;
        if (planes)
            for (int c = 0; c < C; ++c) {
                const art_s *src = planes [c];
                for (unsigned int f = 0; f < peek.input_used; ++f) dst [(size_t) f * C + c] = src [f];
            }
        else if (in_stride == C)
            memcpy (dst, input, sizeof (art_s) * in_samples);
        else
            for (unsigned int f = 0; f < peek.input_used; ++f) {
                const art_s *row = input + (size_t) f * in_stride;
                for (int c = 0; c < C; ++c) dst [(size_t) f * C + c] = row [c];
            }
        TRACE_MARK (1);
        if (sizeof (art_s) * in_samples <= KERNEL_COPY_LIMIT) arthip_copy_by_kernel (hip->d_in, hip->h_in, sizeof (art_s) * in_samples, hip->stream);
        else arthip_h2d (hip->d_in, hip->h_in, sizeof (art_s) * in_samples, hip->stream);
    }
    else if (in_samples) {
        if (planes) {
            for (int c = 0; c < C; ++c)
                arthip_h2d (hip->d_tmp + (size_t) c * peek.input_used, planes [c], sizeof (art_s) * peek.input_used, hip->stream);
            arthip_interleave (hip->d_in, hip->d_tmp, peek.input_used, (int) peek.input_used, C, hip->stream);
        }
        else if (in_stride == C)
            arthip_h2d (hip->d_in, input, sizeof (art_s) * in_samples, hip->stream);
        else
            arthip_copy2d (hip->d_in, sizeof (art_s) * C, input, sizeof (art_s) * in_stride, sizeof (art_s) * C, peek.input_used, hip->stream);
    }

    TRACE_MARK (2);
    /* small staged calls: the FIR kernels write their output straight into the page-locked buffer (it is mapped into the
     * device's address space; one launch and its dependency gap less than copying it out afterwards) */
    static long direct_limit = -1;                             /* (ARTAMD_DIRECT_OUT_LIMIT=bytes: A/B runs) */
    if (direct_limit < 0) { const char *e = getenv ("ARTAMD_DIRECT_OUT_LIMIT"); direct_limit = e && *e ? atol (e) : (long) DIRECT_OUT_LIMIT; }
    const int direct_out = staged && sizeof (art_s) * out_samples <= (size_t) direct_limit && !hip->nshards;
    pend->res = hip->nshards ? sharded_device_call (cxt, hip->d_in, 0, nIn, hip->d_out, 0, cap, ratio)
                             : enqueue_call (cxt, hip->d_in, 0, nIn, direct_out ? hip->h_out : hip->d_out, 0, cap, ratio);
    pend->failed = 0;
    TRACE_MARK (3);

    const unsigned int made = pend->res.output_generated;
    if (!made) return;
    if (staged) {
        if (!direct_out) {
            if (sizeof (art_s) * (size_t) made * C <= KERNEL_COPY_LIMIT) arthip_copy_by_kernel (hip->h_out, hip->d_out, sizeof (art_s) * (size_t) made * C, hip->stream);
            else arthip_d2h (hip->h_out, hip->d_out, sizeof (art_s) * (size_t) made * C, hip->stream);
        }
        pend->staged_out = 1;
    }
    else if (out_planes) {
        arthip_deinterleave (hip->d_tmp, made, hip->d_out, (int) made, C, hip->stream);
        for (int c = 0; c < C; ++c)
            arthip_d2h (out_planes [c], hip->d_tmp + (size_t) c * made, sizeof (art_s) * made, hip->stream);
    }
    else if (out_stride == C)
        arthip_d2h (output, hip->d_out, sizeof (art_s) * (size_t) made * C, hip->stream);
    else
        arthip_copy2d (output, sizeof (art_s) * out_stride, hip->d_out, sizeof (art_s) * C, sizeof (art_s) * C, made, hip->stream);
    TRACE_MARK (4);
}

static void host_end (Resample *cxt, art_s *output, int out_stride, art_s *const *out_planes, const HostPending *pend)
{
    struct artamd_resampler *hip = cxt->hip;
    const int C = cxt->numChannels;
    const unsigned int made = pend->res.output_generated;

    double trace_t_ = trace_on > 0 ? trace_now () : 0.0;
    arthip_sync (hip->stream);
    TRACE_MARK (5);
    if (!pend->staged_out || !made) return;

    const art_s *src = hip->h_out;
    if (out_planes)
        for (int c = 0; c < C; ++c) {
            art_s *dst = out_planes [c];
            for (unsigned int f = 0; f < made; ++f) dst [f] = src [(size_t) f * C + c];
        }
    else if (out_stride == C)
        memcpy (output, src, sizeof (art_s) * (size_t) made * C);
    else
        for (unsigned int f = 0; f < made; ++f) {
            art_s *row = output + (size_t) f * out_stride;
            for (int c = 0; c < C; ++c) row [c] = src [(size_t) f * C + c];
        }
    TRACE_MARK (6);
}

/* A sharded context stages the WHOLE interleaved buffer on its own device exactly as an ordinary context does (one dense
 * transfer each way at full PCIe rate) and hands it to the device-pointer path, whose shards pull and push their channel
 * slices with slice kernels (peer-to-peer over xGMI when they sit on other GPUs).  (Tried first: every shard packing and
 * uploading its own slice — the 2-D copy command moves 16-byte rows one by one (32 ch x 262,144 frames: 16 ms against 1.5 ms
 * for the dense copy), and CPU packing reads the whole interleaved buffer once per shard.) */
static ResampleResult host_call (Resample *cxt, const art_s *input, const art_s *const *planes, int nIn,
                                 art_s *output, art_s *const *out_planes, int cap, double ratio)
{
    struct artamd_resampler *hip = cxt->hip;
    const int C = cxt->numChannels;
    HostPending pend;

    ENTER_DEVICE (hip);
    host_begin (cxt, input, C, planes, nIn, output, C, out_planes, cap, ratio, &pend);
    host_end (cxt, output, C, out_planes, &pend);
    LEAVE_DEVICE (hip);
    return pend.res;
}

ResampleResult resampleProcessInterleaved (Resample *cxt, const artsample_t *input, int numInputFrames, artsample_t *output, int numOutputFrames, double ratio)
{
    return host_call (cxt, input, NULL, numInputFrames, output, NULL, numOutputFrames, ratio);
}

ResampleResult resampleProcess (Resample *cxt, const artsample_t *const *input, int numInputFrames, artsample_t *const *output, int numOutputFrames, double ratio)
{
    return host_call (cxt, NULL, input, numInputFrames, NULL, output, numOutputFrames, ratio);
}

ResampleResult resampleProcessAndFlushInterleaved (Resample *cxt, const artsample_t *input, int numInputFrames, artsample_t *output, int numOutputFrames, double ratio)
{
    ResampleResult res = resampleProcessInterleaved (cxt, input, numInputFrames, output, numOutputFrames, ratio);

    /* unconsumed input or no room left: the caller made a mistake, report what happened */
    if ((numInputFrames -= res.input_used) != 0 || (numOutputFrames -= res.output_generated) == 0)
        return res;

    ResampleResult tail = resampleProcessInterleaved (cxt, NULL, -1, output + (size_t) res.output_generated * cxt->numChannels, numOutputFrames, ratio);
    res.output_generated += tail.output_generated;
    return res;
}

ResampleResult resampleProcessAndFlush (Resample *cxt, const artsample_t *const *input, int numInputFrames, artsample_t *const *output, int numOutputFrames, double ratio)
{
    ResampleResult res = resampleProcess (cxt, input, numInputFrames, output, numOutputFrames, ratio);

    if ((numInputFrames -= res.input_used) != 0 || (numOutputFrames -= res.output_generated) == 0)
        return res;

    art_s **shifted = malloc (sizeof (art_s *) * (size_t) cxt->numChannels);
    for (int c = 0; c < cxt->numChannels; ++c)
        shifted [c] = output [c] + res.output_generated;

    ResampleResult tail = resampleProcess (cxt, NULL, -1, shifted, numOutputFrames, ratio);
    res.output_generated += tail.output_generated;
    free (shifted);
    return res;
}
