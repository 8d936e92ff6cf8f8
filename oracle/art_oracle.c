/* art_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see art_oracle.h).
 *
 * Scalar restatement of the reference hot path.  Parity status: PINNED — checked
 * bit-for-bit against the real reference built with C source-order semantics
 * (oracle/_ref/libartref_strict.so) by tests/test_oracle_vs_ref.py in the build container,
 * and against the committed golden vectors (tests/golden/) everywhere.
 *
 * Build "strict" (-O2 -ffp-contract=off) for parity, "fast" (reference Makefile:10 flags)
 * for the CPU-baseline timing.
 */
#define _GNU_SOURCE
#define _USE_MATH_DEFINES
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "art_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------
 * Filter bank  (reference resampler.c:1090-1133 init_filter, :149-168 bank assembly)
 * ---------------------------------------------------------------------------------------- */

static void build_phase_row (ora_s *row, double *scratch, int taps, double phase, double lowpass, int blackman_harris)
{
    const int half = taps / 2;
    double total = 0.0;

    for (int j = 0; j < taps; ++j) {
        /* distance (radians) of tap j from the sinc peak; window argument reaches pi at the edges */
        double dist = fabs ((half - 1) + phase - j) * M_PI;
        double warg = dist / half;
        double v = 1.0;

        if (dist != 0.0) {
            v = sin (dist * lowpass) / (dist * lowpass);

            if (blackman_harris)
                v *= 0.35875 + 0.48829 * cos (warg) + 0.14128 * cos (2 * warg) + 0.01168 * cos (3 * warg);
            else
                v *= 0.5 * (1.0 + cos (warg));
        }

        scratch [j] = v;
        total += v;
    }

    /* unity DC gain, then round to the sample type walking centre-outwards carrying the rounding error
     * (resampler.c:1126-1132): visit half, half-1, half+1, half-2, ... 0 */
    const double norm = 1.0 / total;
    double carried = 0.0;
    int j = half;

    while (j < taps) {
        scratch [j] *= norm;
        row [j] = (ora_s)(scratch [j] - carried);
        carried += row [j] - scratch [j];
        j = (j >= half) ? taps - j - 1 : taps - j;
    }
}

static ora_s *build_bank (int taps, int filters, double lowpass, int flags)
{
    ora_s *bank = calloc ((size_t)(filters + 1) * taps, sizeof (ora_s));
    double *scratch = malloc (sizeof (double) * taps);

    for (int f = 0; f < filters; ++f)
        build_phase_row (bank + (size_t) f * taps, scratch, taps, (double) f / filters, lowpass, flags & ORA_BLACKMAN_HARRIS);

    /* row F = row 0 delayed by one tap (with wrap), resampler.c:156-159 */
    for (int j = 0; j < taps; ++j)
        bank [(size_t) filters * taps + (j + 1) % taps] = bank [j];

    /* the two outermost taps are cleared, resampler.c:167-168 */
    bank [taps - 1] = 0.0f;
    bank [(size_t) filters * taps] = 0.0f;

    free (scratch);
    return bank;
}

/* ------------------------------------------------------------------------------------------
 * Context  (resampler.c:115-199, :310-356, :383-397, :927-935, :965-968)
 * ---------------------------------------------------------------------------------------- */

OraResampler *ora_resample_init (int channels, int taps, int filters, double lowpass_ratio, int flags)
{
    if (lowpass_ratio > 0.0 && lowpass_ratio < 1.0)
        flags |= ORA_LOWPASS;
    else {
        flags &= ~ORA_LOWPASS;
        lowpass_ratio = 1.0;
    }

    if ((taps & 3) || taps <= 0 || taps > 1024 || filters < 1 || filters > 1024)
        return NULL;

    OraResampler *r = calloc (1, sizeof (*r));
    r->channels = channels;
    r->taps = taps;
    r->filters = filters;
    r->ring_len = taps * 16;
    r->flags = flags;
    r->lowpass_ratio = lowpass_ratio;
    r->bank = build_bank (taps, filters, lowpass_ratio, flags);
    /* Each channel ring is preceded by `taps` guard zeros.  REFERENCE BUG (documented in DESIGN.md):
     * when a flush arrives with inputIndex within half a window of the ring end, the flush-time rewind
     * (resampler.c:667-672) keeps only `taps` samples while windows still reach up to taps/2 further
     * back, so the reference reads before buffers[c][0] (heap UB; zeros for ch>=1 with glibc, garbage
     * for ch 0).  The oracle and the product define those pre-history samples as silence. */
    r->ring_store = calloc ((size_t) channels * (r->ring_len + taps), sizeof (ora_s));
    r->ring = r->ring_store + taps;
    r->read_pos = taps / 2;
    r->write_pos = taps;

    if (flags & ORA_EXTRAPOLATE)
        r->flags |= ORA_PREFILL;

    return r;
}

static unsigned long gcd_ul (unsigned long a, unsigned long b)
{
    while (b) { unsigned long t = a % b; a = b; b = t; }
    return a;
}

OraResampler *ora_resample_fixed_init (int channels, int taps, int max_filters, double src_rate, double dst_rate, int lowpass_freq, int flags)
{
    double lowpass = lowpass_freq / (dst_rate / 2.0);
    double ratio = dst_rate / src_rate;

    if (lowpass_freq > dst_rate / 2.0)
        return NULL;

    /* exact phase set small enough => no interpolation needed (resampler.c:323-335) */
    if (src_rate == floor (src_rate) && dst_rate == floor (dst_rate) && !(flags & ORA_NO_REDUCTION)) {
        unsigned long phases = (unsigned long) dst_rate / gcd_ul ((unsigned long) src_rate, (unsigned long) dst_rate);

        if (phases <= (unsigned long) max_filters) {
            flags &= ~ORA_INTERPOLATE;
            max_filters = (int) phases;

            if (max_filters & (max_filters - 1))
                flags |= ORA_SNAP;
        }
    }

    /* automatic low-pass for downsampling (resampler.c:340-348) */
    if (!lowpass_freq && (flags & ORA_LOWPASS) && dst_rate < src_rate) {
        lowpass = 1.0 - (7.5 / taps / ratio);
        if (lowpass < 0.8) lowpass = 0.8;
        if (lowpass < ratio) lowpass = ratio;
    }

    OraResampler *r = ora_resample_init (channels, taps, max_filters, lowpass * ratio, flags | ORA_FIXED_RATIO);

    if (r)
        r->fixed_ratio = dst_rate / src_rate;

    return r;
}

void ora_resample_free (OraResampler *r)
{
    if (r) { free (r->bank); free (r->ring_store); free (r); }
}

void ora_resample_reset (OraResampler *r)
{
    memset (r->ring_store, 0, sizeof (ora_s) * (size_t) r->channels * (r->ring_len + r->taps));
    r->read_pos = r->taps / 2;
    r->write_pos = r->taps;
    if (r->flags & ORA_EXTRAPOLATE) r->flags |= ORA_PREFILL;
    r->flags &= ~ORA_FLUSHED;
}

void ora_resample_advance (OraResampler *r, double delta)
{
    if (delta < 0.0) return;                                              /* resampler.c:929 */
    if (!(r->flags & ORA_INTERPOLATE) && floor (delta) != delta) return;  /* :931 */
    r->read_pos += delta;
}

double ora_resample_position (const OraResampler *r)
{
    return r->read_pos + (r->taps / 2.0) - r->write_pos;                  /* resampler.c:967 */
}

/* dry runs (resampler.c:853-918): note these accumulate 1/ratio instead of dividing */
unsigned ora_resample_required_input (const OraResampler *r, int n_out, double ratio)
{
    int half = r->taps / 2, wp = r->write_pos;
    double pos = r->read_pos;
    unsigned used = 0;

    if (r->flags & ORA_FIXED_RATIO) ratio = r->fixed_ratio;

    while (n_out > 0) {
        if (pos >= wp - half) {
            if (wp == r->ring_len) { pos -= r->ring_len - r->taps; wp -= r->ring_len - r->taps; }
            wp++; used++;
        }
        else { pos += 1.0 / ratio; n_out--; }
    }
    return used;
}

unsigned ora_resample_expected_output (const OraResampler *r, int n_in, double ratio)
{
    int half = r->taps / 2, wp = r->write_pos;
    double pos = r->read_pos;
    unsigned made = 0;

    if (r->flags & ORA_FIXED_RATIO) ratio = r->fixed_ratio;
    if (r->flags & ORA_FLUSHED) n_in = 0;
    else if (n_in < 0) wp += half;

    for (;;) {
        if (pos >= wp - half) {
            if (n_in <= 0) break;
            if (wp == r->ring_len) { pos -= r->ring_len - r->taps; wp -= r->ring_len - r->taps; }
            wp++; n_in--;
        }
        else { pos += 1.0 / ratio; made++; }
    }
    return made;
}

/* ------------------------------------------------------------------------------------------
 * Dot products  (resampler.c:1033-1057)
 * ---------------------------------------------------------------------------------------- */

/* sample-type accumulator, pairs taken from both ends towards the middle (resampler.c:1033-1044) */
double ora_dot_outside_in (const ora_s *h, const ora_s *x, int taps)
{
    ora_s acc = 0.0f;

    for (int lo = 0, hi = taps - 1; lo < hi; ++lo, --hi)
        acc += (h [lo] * x [lo]) + (h [hi] * x [hi]);

    return acc;
}

/* double accumulator, taps in order (resampler.c:1049-1057) */
double ora_dot_precise (const ora_s *h, const ora_s *x, int taps)
{
    double acc = 0.0;

    for (int k = 0; k < taps; ++k)
        acc += (double) h [k] * x [k];

    return acc;
}

/* one output value at fractional ring position `pos` (resampler.c:1135-1181) */
static double evaluate_at (const OraResampler *r, const ora_s *ring, double pos)
{
    const int T = r->taps, F = r->filters;
    const double whole = floor (pos);
    /* "extended math" exists for 4-byte samples only (resampler.c:191) */
    double (*dot)(const ora_s *, const ora_s *, int) = (sizeof (ora_s) == 4 && (r->flags & ORA_PRECISE)) ? ora_dot_precise : ora_dot_outside_in;

    if (r->flags & ORA_INTERPOLATE) {
        double frac = (pos - whole) * F;
        int fi = (int) floor (frac);
        const ora_s *win = ring + (int) whole - T / 2 + 1;

        frac -= fi;
        return (dot (r->bank + (size_t) fi * T, win, T) * (1.0 - frac)) +
               (dot (r->bank + (size_t)(fi + 1) * T, win, T) * frac);
    }
    else {
        int fi = (int) floor ((pos - whole) * F + 0.5);
        const ora_s *centre = ring + (int) whole;

        if (!(r->flags & ORA_LOWPASS) && !(fi % F))     /* exact sample hit: pass it through */
            return centre [fi / F];

        return dot (r->bank + (size_t) fi * T, centre - T / 2 + 1, T);
    }
}


/* ------------------------------------------------------------------------------------------
 * End-point extrapolation  (extrapolator.c:22-277): 4-tap LPC by coordinate descent, PARCOR clamp
 * ---------------------------------------------------------------------------------------- */
#define LPC_N 4

static void lpc_to_refl (const double *lpc, double *k)
{
    double t [LPC_N], u [LPC_N];
    for (int i = 0; i < LPC_N; ++i) t [i] = lpc [i];
    for (int m = LPC_N - 1; m >= 0; --m) {
        k [m] = t [m];
        double den = 1.0 - (k [m] * k [m]);
        if (fabs (den) < 1e-6) { k [m] = k [m] < 0.0 ? -0.9999995 : 0.9999995; den = 1.0 - (k [m] * k [m]); }
        for (int i = 0; i < m; ++i) u [i] = (t [i] - k [m] * t [m - i - 1]) / den;
        for (int i = 0; i < m; ++i) t [i] = u [i];
    }
}

static void refl_to_lpc (const double *k, double *lpc)
{
    for (int i = 0; i < LPC_N; i++) {
        lpc [i] = k [i];
        for (int j = 0; j < i / 2; j++) { double t = lpc [j]; lpc [j] += k [i] * lpc [i - 1 - j]; lpc [i - 1 - j] += k [i] * t; }
        if (i & 1) lpc [i >> 1] += lpc [i >> 1] * k [i];
    }
}

static void lpc_fit (const ora_s *v, int n, float *co)
{
    int ne = n - LPC_N, loops = 0, changes = 0;
    double vrms = 0.0, drms = 0.0, err, step = 3.0 / (1 << 4);
    double *sums = malloc (sizeof (double) * (ne > 0 ? ne : 1));

    for (int i = 0; i < LPC_N; ++i) co [i] = 0.0f;
    for (int i = 0; i < ne; ++i) {
        drms += (v [i + LPC_N] - v [i + LPC_N - 1]) * (v [i + LPC_N] - v [i + LPC_N - 1]);
        vrms += v [i + LPC_N] * v [i + LPC_N];
    }
    if (vrms == 0.0) { free (sums); return; }
    err = vrms;

    while (err > 0.0 && loops < 100000) {
        int tc;
        for (int k = 0; k < ne; ++k) {
            double z = 0.0;
            for (int c = 0; c < LPC_N; ++c) z += co [LPC_N - c - 1] * v [k + c];
            sums [k] = z + v [k + LPC_N];
        }
        for (tc = 0; loops++, tc < LPC_N; tc++) {
            double lo = 0.0, hi = 0.0;
            for (int k = 0; k < ne; ++k) {
                double d = v [k + LPC_N - tc - 1] * step;
                lo += (sums [k] - d) * (sums [k] - d);
                hi += (sums [k] + d) * (sums [k] + d);
            }
            if (lo < err || hi < err) {
                if (lo < hi) { err = lo; co [tc] -= step; } else { err = hi; co [tc] += step; }
                changes++;
                break;
            }
        }
        if (tc == LPC_N) { if (step > 3.0 / (1 << 22)) step *= 0.5; else break; }
    }
    free (sums);

    if (changes) {
        double d [LPC_N], k [LPC_N];
        int out = 0;
        for (int i = 0; i < LPC_N; ++i) d [i] = co [i];
        lpc_to_refl (d, k);
        for (int i = 0; i < LPC_N; ++i) if (fabs (k [i]) > 0.9999) { k [i] = k [i] < 0.0 ? -0.9999 : 0.9999; out++; }
        if (out) { refl_to_lpc (k, d); for (int i = 0; i < LPC_N; ++i) co [i] = d [i]; }
    }

    err = 0.0;
    for (int k = 0; k < ne; ++k) {
        double z = 0.0;
        for (int c = 0; c < LPC_N; ++c) z += co [LPC_N - c - 1] * v [k + c];
        err += (z + v [k + LPC_N]) * (z + v [k + LPC_N]);
    }
    if (drms < err && drms < vrms) { for (int i = 0; i < LPC_N; ++i) co [i] = 0.0f; co [0] = -1.0; }
    else if (vrms <= err) for (int i = 0; i < LPC_N; ++i) co [i] = 0.0f;
}

static void lpc_forward (ora_s *v, int n, int extra)
{
    float co [LPC_N];
    memset (v + n, 0, sizeof (ora_s) * extra);
    lpc_fit (v, n, co);
    for (int i = 0; i < extra; ++i) {
        double z = 0.0;
        for (int c = 0; c < LPC_N; ++c) z += v [n - LPC_N + i + c] * co [LPC_N - c - 1];
        v [n + i] = -z;
    }
}

static void lpc_reverse (ora_s *past_end, int n, int extra)      /* past_end[-1] is the newest known sample */
{
    ora_s *r = calloc (n + extra, sizeof (ora_s));
    for (int i = 0; i < n; ++i) r [i] = past_end [-1 - i];
    lpc_forward (r, n, extra);
    for (int i = n; i < n + extra; ++i) past_end [-1 - i] = r [i];
    free (r);
}

/* ------------------------------------------------------------------------------------------
 * Streaming state machine  (resampler.c:487-537 == 604-654 == 766-834; flush :663-685)
 * One channel at a time; every channel replays the identical position sequence.
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    OraResampler *r;
    ora_s *ring;
    const ora_s *in; int in_stride, n_in;
    ora_s *out; int out_stride, out_cap;
    double ratio;
    /* results */
    double read_pos; int write_pos, flags;
    OraResult res;
} ChannelRun;

static void ring_rewind (const OraResampler *r, ora_s *ring, double *pos, int *wp)
{
    const int keep = r->taps, drop = r->ring_len - r->taps;
    memmove (ring, ring + drop, sizeof (ora_s) * keep);
    *pos -= drop;
    *wp -= drop;
}

static void *run_channel (void *arg)
{
    ChannelRun *c = arg;
    const OraResampler *r = c->r;
    const int half = r->taps / 2;
    double pos = r->read_pos, step = 0.0;
    int wp = r->write_pos, flags = r->flags;
    int n_in = c->n_in, cap = c->out_cap;
    const ora_s *in = c->in;
    ora_s *out = c->out;
    unsigned used = 0, made = 0;

    if (n_in < 0) {                                     /* flush: append half a window of silence */
        if (r->ring_len - wp < half)
            ring_rewind (r, c->ring, &pos, &wp);

        memset (c->ring + wp, 0, sizeof (ora_s) * (r->ring_len - wp));
        if (flags & ORA_EXTRAPOLATE)                    /* resampler.c:677-680 */
            lpc_forward (c->ring + wp - half, half, half);
        flags |= ORA_FLUSHED;
        wp += half;
    }

    while (cap > 0) {
        if (pos + step >= wp - half) {                  /* need another input frame */
            if (n_in <= 0)
                break;

            if (wp == r->ring_len)
                ring_rewind (r, c->ring, &pos, &wp);

            c->ring [wp++] = *in;
            in += c->in_stride;
            used++; n_in--;
        }
        else {
            if (flags & ORA_PREFILL) {                  /* resampler.c:812-819: once, before the first output */
                int known = wp - r->taps;
                if (known >= 8)
                    lpc_reverse (c->ring + wp, known, r->taps - known);
                flags &= ~ORA_PREFILL;
            }
            *out = (ora_s) evaluate_at (r, c->ring, pos + step);
            out += c->out_stride;
            step = (double)(++made) / c->ratio;         /* division, not accumulation (resampler.c:526) */
            cap--;
        }
    }

    pos += step;

    if (flags & ORA_SNAP)                               /* resampler.c:533-535 */
        pos = floor (pos) + floor ((pos - floor (pos)) * r->filters + 0.5) / r->filters;

    c->read_pos = pos; c->write_pos = wp; c->flags = flags;
    c->res.used = used; c->res.generated = made;
    return NULL;
}

static OraResult run_all (OraResampler *r, const ora_s *const *in, int in_stride, int n_in,
                          ora_s *const *out, int out_stride, int out_cap, double ratio, int threads)
{
    const int C = r->channels;
    ChannelRun *runs = calloc (C, sizeof (ChannelRun));
    pthread_t *tids = calloc (C, sizeof (pthread_t));
    OraResult res;

    if (r->flags & ORA_FIXED_RATIO) ratio = r->fixed_ratio;     /* resampler.c:435-436 */
    if (r->flags & ORA_FLUSHED) n_in = 0;                       /* :438-439 */

    for (int c = 0; c < C; ++c) {
        ChannelRun *cr = runs + c;
        cr->r = r; cr->ring = r->ring + (size_t) c * (r->ring_len + r->taps);
        cr->in = (in && n_in > 0) ? in [c] : NULL; cr->in_stride = in_stride; cr->n_in = n_in;
        cr->out = out [c]; cr->out_stride = out_stride; cr->out_cap = out_cap;
        cr->ratio = ratio;
    }

    if (threads > 1 && C > 1) {         /* workers.c model: C-1 helpers, last channel on the caller */
        for (int c = 0; c < C - 1; ++c)
            pthread_create (tids + c, NULL, run_channel, runs + c);
        run_channel (runs + C - 1);
        for (int c = 0; c < C - 1; ++c)
            pthread_join (tids [c], NULL);
    }
    else
        for (int c = 0; c < C; ++c)
            run_channel (runs + c);

    r->read_pos = runs [0].read_pos;
    r->write_pos = runs [0].write_pos;
    r->flags = runs [0].flags;
    res = runs [0].res;
    free (runs); free (tids);
    return res;
}

OraResult ora_resample_interleaved (OraResampler *r, const ora_s *in, int n_in, ora_s *out, int out_cap, double ratio, int threads)
{
    const int C = r->channels;
    const ora_s **ip = malloc (sizeof (ora_s *) * C);
    ora_s **op = malloc (sizeof (ora_s *) * C);
    for (int c = 0; c < C; ++c) { ip [c] = in ? in + c : NULL; op [c] = out + c; }
    OraResult res = run_all (r, in ? ip : NULL, C, n_in, op, C, out_cap, ratio, threads);
    free (ip); free (op);
    return res;
}

OraResult ora_resample_planar (OraResampler *r, const ora_s *const *in, int n_in, ora_s *const *out, int out_cap, double ratio, int threads)
{
    return run_all (r, in, 1, n_in, out, 1, out_cap, ratio, threads);
}

/* process then flush into the tail (resampler.c:741-758) */
OraResult ora_resample_interleaved_flush (OraResampler *r, const ora_s *in, int n_in, ora_s *out, int out_cap, double ratio, int threads)
{
    OraResult res = ora_resample_interleaved (r, in, n_in, out, out_cap, ratio, threads);

    if ((n_in - (int) res.used) != 0 || (out_cap - (int) res.generated) == 0)
        return res;

    OraResult tail = ora_resample_interleaved (r, NULL, -1, out + (size_t) res.generated * r->channels,
                                               out_cap - (int) res.generated, ratio, threads);
    res.generated += tail.generated;
    return res;
}

/* ------------------------------------------------------------------------------------------
 * Biquad  (biquad.c:18-163)
 * ---------------------------------------------------------------------------------------- */

void ora_biquad_lowpass (OraBiquadCoeffs *c, double freq)
{
    double Q = sqrt (0.5), K = tan (M_PI * freq);
    double norm = 1.0 / (1.0 + K / Q + K * K);

    memset (c, 0, sizeof (*c));
    c->a0 = (ora_s)(K * K * norm);
    c->a1 = (ora_s)(2 * c->a0);           /* uses the already-rounded a0 (biquad.c:26) */
    c->a2 = c->a0;
    c->b1 = (ora_s)(2.0 * (K * K - 1.0) * norm);
    c->b2 = (ora_s)((1.0 - K / Q + K * K) * norm);
}

void ora_biquad_highpass (OraBiquadCoeffs *c, double freq)
{
    double Q = sqrt (0.5), K = tan (M_PI * freq);
    double norm = 1.0 / (1.0 + K / Q + K * K);

    memset (c, 0, sizeof (*c));
    c->a0 = (ora_s) norm;
    c->a1 = (ora_s)(-2.0 * norm);
    c->a2 = c->a0;
    c->b1 = (ora_s)(2.0 * (K * K - 1.0) * norm);
    c->b2 = (ora_s)((1.0 - K / Q + K * K) * norm);
}

void ora_biquad_init (OraBiquad *f, const OraBiquadCoeffs *c, double gain)
{
    memset (f, 0, sizeof (*f));
    f->a [0] = (ora_s)(c->a0 * gain); f->a [1] = (ora_s)(c->a1 * gain); f->a [2] = (ora_s)(c->a2 * gain);
    f->a [3] = (ora_s)(c->a3 * gain); f->a [4] = (ora_s)(c->a4 * gain);
    f->b [1] = c->b1; f->b [2] = c->b2; f->b [3] = c->b3; f->b [4] = c->b4;

    f->order = (c->a4 != 0.0f || c->b4 != 0.0f) ? 4 :
               (c->a3 != 0.0f || c->b3 != 0.0f) ? 3 :
               (c->a2 != 0.0f || c->b2 != 0.0f) ? 2 : 1;
}

/* per-sample form: highest order term first (biquad.c:78-102) */
ora_s ora_biquad_sample (OraBiquad *f, ora_s in)
{
    ora_s acc = in * f->a [0];
    int i = f->index & 3;

    for (int k = f->order; k >= 1; --k)
        acc += (f->x [(i - k + 1) & 3] * f->a [k]) - (f->b [k] * f->y [(i - k + 1) & 3]);

    i = (i + 1) & 3;
    f->index = i;
    f->x [i] = in;
    f->y [i] = acc;
    return acc;
}

/* buffer form: lowest order term first, strictly left to right (biquad.c:106-163) */
void ora_biquad_buffer (OraBiquad *f, ora_s *buf, int n, int stride)
{
    int i = f->index;

    while (n-- > 0) {
        ora_s acc = *buf * f->a [0];

        for (int k = 1; k <= f->order; ++k) {
            acc = acc + (f->x [(i - k + 1) & 3] * f->a [k]);
            acc = acc - (f->b [k] * f->y [(i - k + 1) & 3]);
        }

        ++i;
        f->x [i & 3] = *buf;
        f->y [i & 3] = acc;
        *buf = acc;
        buf += stride;
    }

    f->index = i;
}

/* ------------------------------------------------------------------------------------------
 * Decimator  (decimator.c:28-97 init, :245-283 loop, :370-382 dither, :389-409 shaper, :416-450 ingest)
 * ---------------------------------------------------------------------------------------- */

static inline uint32_t lcg32 (uint32_t r) { return ((r << 4) - r) ^ 1; }

static void shaper_from_ntf (OraBiquad *f, double a1, double a2, double a3, double a4, double b1, double b2, double b3, double b4)
{
    OraBiquadCoeffs c;
    memset (&c, 0, sizeof (c));
    c.a0 = (ora_s)(b1 - a1); c.a1 = (ora_s)(b2 - a2); c.a2 = (ora_s)(b3 - a3); c.a3 = (ora_s)(b4 - a4);
    c.b1 = (ora_s) b1; c.b2 = (ora_s) b2; c.b3 = (ora_s) b3; c.b4 = (ora_s) b4;
    ora_biquad_init (f, &c, 1.0);
}

OraDecimator *ora_decimate_init (int channels, int bits, int bytes, double gain, int rate, int flags)
{
    OraDecimator *d = calloc (1, sizeof (*d));
    d->channels = channels; d->bits = bits; d->bytes = bytes; d->gain = gain; d->flags = flags;
    d->feedback = calloc (channels, sizeof (ora_s));

    if (flags & ORA_DITHER_ANY) {
        /* seeds = consecutive bytes of (state >> 24), three LCG steps per byte (decimator.c:40-52) */
        uint32_t s = 0x31415926;
        unsigned char *raw = malloc (4 * (size_t) channels);
        for (int i = 0; i < 4 * channels; ++i) { raw [i] = s >> 24; s = lcg32 (lcg32 (lcg32 (s))); }
        d->gens = malloc (sizeof (uint32_t) * channels);
        for (int c = 0; c < channels; ++c)
            d->gens [c] = (uint32_t) raw [4*c] | ((uint32_t) raw [4*c+1] << 8) | ((uint32_t) raw [4*c+2] << 16) | ((uint32_t) raw [4*c+3] << 24);
        free (raw);
        d->dither_type = (flags & ORA_DITHER_HIGHPASS) ? -1 : (flags & ORA_DITHER_LOWPASS) ? 1 : 0;
    }

    if (flags & ORA_SHAPE_ANY) {
        d->shapers = calloc (channels, sizeof (OraBiquad));
        for (int c = 0; c < channels; ++c) {
            OraBiquad *f = d->shapers + c;
            if (flags & ORA_SHAPE_ATH) {
                switch (rate) {
                    case 32000: shaper_from_ntf (f, -0.780459, +0.569358, -0.348221, +0.466316, +0.950797, +0.282052, +0.004337, +1.76209e-5); break;
                    case 44100: shaper_from_ntf (f, -1.1474, 0.5383, -0.3530, 0.3475, 1.0587, 0.0676, -0.6054, -0.2738); break;
                    case 48000: shaper_from_ntf (f, -1.3344, 0.7455, -0.4602, 0.4363, 0.9030, 0.0116, -0.5853, -0.2571); break;
                    case 88200: shaper_from_ntf (f, -2.150679, +2.1402057, -1.042712, +0.206838, +0.67433, +1.017047, +0.4028633, +0.098656); break;
                    case 96000: shaper_from_ntf (f, -2.16994, +2.01986, -0.894857, +0.1557738, +0.517789, +1.1062189, +0.4825786, +0.244994); break;
                    default:    shaper_from_ntf (f, -1.0, 0, 0, 0, 0, 0, 0, 0); break;
                }
            }
            else if (flags & ORA_SHAPE_1ST) shaper_from_ntf (f, -1.0, 0, 0, 0, 0, 0, 0, 0);
            else if (flags & ORA_SHAPE_2ND) shaper_from_ntf (f, -2.0, +1.0, 0, 0, 0, 0, 0, 0);
            else if (flags & ORA_SHAPE_3RD) shaper_from_ntf (f, -3.0, +3.0, -1.0, 0, 0, 0, 0, 0);
        }
    }

    return d;
}

void ora_decimate_free (OraDecimator *d)
{
    if (d) { free (d->feedback); free (d->gens); free (d->shapers); free (d); }
}

/* triangular dither in [-1,1): five LCG steps per value (decimator.c:370-382) */
static double tpdf_value (uint32_t *gen, int type)
{
    uint32_t start = *gen, r = lcg32 (lcg32 (start));
    uint32_t first = (type < 0) ? ~start : (type > 0) ? start : ~r;
    r = lcg32 (lcg32 (lcg32 (r)));
    *gen = r;
    return (((first >> 1) + (r >> 1)) / 2147483648.0) - 1.0;
}

static int decimate_one (OraDecimator *d, int ch, ora_s in, unsigned char *out)
{
    const ora_s scale = (ora_s)((1 << d->bits) / 2.0 * d->gain);
    const int pad = d->bytes - ((d->bits + 7) / 8);
    const int32_t hi = (1 << (d->bits - 1)) - 1, lo = ~hi;
    const int shift = (24 - d->bits) % 8;
    int clipped = 0;

    ora_s dither = (d->flags & ORA_DITHER_ANY) ? (ora_s) tpdf_value (d->gens + ch, d->dither_type) : 0.0f;
    ora_s code = (in * scale) - d->feedback [ch];
    ora_s dithered = code + dither;
    int32_t q = (int32_t) floor ((double) dithered + 0.5);

    if (d->flags & ORA_SHAPE_ANY)
        d->feedback [ch] = ora_biquad_sample (d->shapers + ch, (ora_s) q - code);

    if (q > hi) { q = hi; clipped = 1; }
    else if (q < lo) { q = lo; clipped = 1; }

    uint32_t v = ((uint32_t) q << shift) + ((d->bits <= 8) ? 128 : 0);

    for (int j = 0; j < pad; ++j) *out++ = 0;
    *out++ = (unsigned char) v;
    if (d->bits > 8) { *out++ = (unsigned char)(v >> 8); if (d->bits > 16) *out++ = (unsigned char)(v >> 16); }
    return clipped;
}

int ora_decimate_interleaved (OraDecimator *d, const ora_s *in, int frames, unsigned char *out)
{
    int clips = 0;
    for (int i = 0; i < frames; ++i)
        for (int c = 0; c < d->channels; ++c, out += d->bytes)
            clips += decimate_one (d, c, *in++, out);
    return clips;
}

int ora_decimate_planar (OraDecimator *d, const ora_s *const *in, int frames, unsigned char *const *out)
{
    int clips = 0;
    for (int i = 0; i < frames; ++i)
        for (int c = 0; c < d->channels; ++c)
            clips += decimate_one (d, c, in [c][i], out [c] + (size_t) i * d->bytes);
    return clips;
}

void ora_float_integers_le (const unsigned char *in, double gain, int bits, int bytes, int stride, ora_s *out, int n)
{
    const int width = (bits + 7) / 8;
    const size_t hop = (size_t) stride * bytes;
    in += bytes - width;

    if (bits <= 8) {
        ora_s g = (ora_s)(gain / 128.0);
        for (int i = 0; i < n; ++i, in += hop) out [i] = ((int) in [0] - 128) * g;
    }
    else if (bits <= 16) {
        ora_s g = (ora_s)(gain / 32768.0);
        for (int i = 0; i < n; ++i, in += hop) out [i] = (int16_t)(in [0] | (in [1] << 8)) * g;
    }
    else if (bits <= 24) {
        ora_s g = (ora_s)(gain / 8388608.0);
        for (int i = 0; i < n; ++i, in += hop) {
            int32_t v = (int32_t)((uint32_t) in [0] | ((uint32_t) in [1] << 8) | ((uint32_t)(int32_t)(signed char) in [2] << 16));
            out [i] = v * g;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * artest's synthetic input and checksums  (artest.c:744-798, :90-104, :587-588)
 * ---------------------------------------------------------------------------------------- */

uint64_t ora_noise_fill (ora_s *dst, long count, uint64_t s)
{
    while (count-- > 0) {
        s = ((s << 4) - s) ^ 1; s = ((s << 4) - s) ^ 1; s = ((s << 4) - s) ^ 1;
        *dst++ = (ora_s)((int32_t)(s >> 32) / 4294967296.0);
    }
    return s;
}

void ora_fade_in (ora_s *data, int count)
{
    int zeros = count / 4, ramp = count - zeros;
    for (int i = 0; i < zeros; ++i) *data++ = 0.0f;
    for (int i = 0; i < ramp; ++i, ++data) *data = (ora_s)(*data * ((cos ((ramp - i) * M_PI / ramp) + 1.0) / 2.0));
}

void ora_fade_out (ora_s *data, int count)
{
    int zeros = count / 4, ramp = count - zeros;
    for (int i = 0; i < ramp; ++i, ++data) *data = (ora_s)(*data * ((cos (i * M_PI / ramp) + 1.0) / 2.0));
    for (int i = 0; i < zeros; ++i) *data++ = 0.0f;
}

uint64_t ora_checksum_words (uint64_t c, const void *words, long n)
{
    const uint32_t *w = words;
    while (n-- > 0) c = c * 3 + *w++;
    return c;
}

uint64_t ora_checksum_bytes (uint64_t c, const unsigned char *b, long n)
{
    while (n-- > 0) c = c * 3 + *b++;
    return c;
}
