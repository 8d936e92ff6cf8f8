/* stretch_oracle.c — CPU ORACLE of the time-domain harmonic scaler.  TEST INFRASTRUCTURE ONLY.
 *
 * Own restatement of the behaviour of the reference's stretcher (reference stretch.c:50-566, API stretch.h:47-52):
 * pitch-period search by the ratio  sum |x| / sum |x[i] - x[i+p]|  over two consecutive blocks, and the four
 * period-synchronous transformations (1:2, 1:1, 3:2, 2:1) steered by an accumulated length error.  Pinned bit for bit
 * against the real reference (oracle/_ref/libartref*_strict.so, tests/test_stretch.py) and against the committed
 * vectors tests/golden/stretch.npz.
 *
 * Arithmetic notes that matter for bit-exactness (all follow from the C types in the reference):
 *   - fabs() is the double function: "acc += fabs (v)" adds in double and rounds back to the sample type;
 *   - "(a + b) / 2.0" adds in the sample type, halves in double (exact), stores in the sample type;
 *   - the cross-fade is ((a * (n - i)) + (b * i)) / n in the sample type, left to right, true division.
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "art_oracle.h"

#define ORA_STRETCH_QUICK 0x1
#define ORA_STRETCH_TWIN  0x2
#define ORA_PERIOD_MIN 24
#define ORA_PERIOD_MAX 2400

typedef struct OraStretch {
    int channels, room, lo, hi, mark, fill, quick;     /* lo/hi: shortest/longest period in VALUES (x channels) */
    ora_s *ring, *mono, *score;
    double drift;                                      /* output length error so far, in values */
    struct OraStretch *twin;                           /* second stage of a cascaded pair */
    ora_s *between;
} OraStretch;

/* stretch.c:50-95 */
OraStretch *ora_stretch_init (int shortest, int longest, int channels, int flags)
{
    int blocks = 3;

    if (flags & ORA_STRETCH_QUICK) {
        longest = (longest + 1) & ~1;
        shortest &= ~1;
        blocks = 4;
    }

    if (longest <= shortest || shortest < ORA_PERIOD_MIN || longest > ORA_PERIOD_MAX)
        return NULL;

    OraStretch *s = calloc (1, sizeof (*s));
    s->room = longest * channels * blocks;
    s->ring = calloc (s->room, sizeof (ora_s));
    s->mono = calloc ((size_t) longest * channels, sizeof (ora_s));
    s->score = calloc (longest + 2, sizeof (ora_s));
    s->fill = s->mark = s->hi = longest * channels;
    s->lo = shortest * channels;
    s->quick = (flags & ORA_STRETCH_QUICK) != 0;
    s->channels = channels;

    if (flags & ORA_STRETCH_TWIN) {
        s->twin = ora_stretch_init (shortest, longest, channels, flags & ~ORA_STRETCH_TWIN);
        s->between = calloc ((size_t) longest * channels * blocks, sizeof (ora_s));
    }

    return s;
}

void ora_stretch_free (OraStretch *s)
{
    if (!s) return;
    if (s->twin) { ora_stretch_free (s->twin); free (s->between); }
    free (s->ring); free (s->mono); free (s->score); free (s);
}

/* stretch.c:102-110 */
void ora_stretch_reset (OraStretch *s)
{
    s->fill = s->mark = s->hi;
    memset (s->ring, 0, sizeof (ora_s) * s->mark);
    if (s->twin) ora_stretch_reset (s->twin);
}

static void split_ratio (const OraStretch *s, double *ratio, double *rest)
{
    *rest = 1.0;
    if (!s->twin) return;
    if (*ratio < 0.5) { *rest = *ratio / 0.5; *ratio = 0.5; }
    else if (*ratio > 2.0) { *rest = *ratio / 2.0; *ratio = 2.0; }
}

/* stretch.c:117-143 */
int ora_stretch_capacity (const OraStretch *s, int max_frames, double max_ratio)
{
    double rest;
    split_ratio (s, &max_ratio, &rest);
    int most = (int) ceil (max_frames * ceil (max_ratio * 2.0) / 2.0) + (s->hi / s->channels) * (s->quick ? 4 : 3);
    return s->twin ? ora_stretch_capacity (s->twin, most, rest) : most;
}

/* stretch.c:560-566 */
static void crossfade (ora_s *out, const ora_s *from, const ora_s *to, int n)
{
    for (int i = 0; i < n; ++i)
        out [i] = ((from [i]) * (n - i) + to [i] * i) / n;
}

/* stretch.c:391-470: exhaustive search; returns the period in values */
static int pick_period (const OraStretch *s, ora_s *x)
{
    ora_s total, best = 0;
    ora_s *m = x;
    int p = s->lo / s->channels, pick = p;

    if (s->channels == 2) {
        m = s->mono;
        total = 0;
        for (int i = 0, j = 0; i < s->hi * 2; i += 2)
            total += fabs (m [j++] = (x [i] + x [i + 1]) / 2.0);
    }
    else {
        total = 0;
        for (int i = 0; i < s->hi; ++i)
            total += fabs (m [i]) + fabs (m [i + s->hi]);
    }

    if (!total)                                         /* silence */
        return s->hi;

    total = 0;
    for (int i = 0; i < p; ++i)
        total += fabs (m [i]) + fabs (m [i + p]);

    for (;;) {
        ora_s miss = 0;
        for (int i = p - 1; i >= 0; --i)
            miss += fabs (m [i] - m [i + p]);

        ora_s q = (miss == 0.0) ? FLT_MAX : total / miss;
        if (q >= best) { best = q; pick = p; }

        if (p * s->channels == s->hi)
            break;

        total += fabs (m [p * 2]) + fabs (m [p * 2 + 1]);
        p++;
    }

    return pick * s->channels;
}

/* stretch.c:472-552: 2:1 decimated search with a three-point refinement */
static int pick_period_quick (const OraStretch *s, const ora_s *x)
{
    ora_s total = 0, best = 0;
    ora_s *m = s->mono, *score = s->score;
    int p = s->lo / (s->channels * 2), pick = p;

    if (s->channels == 2)
        for (int i = 0, j = 0; i < s->hi * 2; i += 4)
            total += fabs (m [j++] = (x [i] + x [i + 1] + x [i + 2] + x [i + 3]) / 2.0);
    else
        for (int i = 0, j = 0; i < s->hi * 2; i += 2)
            total += fabs (m [j++] = (x [i] + x [i + 1]) / 2.0);

    if (!total)
        return s->hi;

    total = 0;
    for (int i = 0; i < p; ++i)
        total += fabs (m [i]) + fabs (m [i + p]);

    for (;;) {
        ora_s miss = 0.0;
        for (int i = p - 1; i >= 0; --i)
            miss += fabs (m [i] - m [i + p]);

        score [p] = miss == 0.0 ? FLT_MAX : total / miss;
        if (score [p] >= best) { best = score [p]; pick = p; }

        if (p * s->channels * 2 == s->hi)
            break;

        total += fabs (m [p * 2]) + fabs (m [p * 2 + 1]);
        p++;
    }

    if (pick * s->channels * 2 != s->lo && pick * s->channels * 2 != s->hi) {
        ora_s above = score [pick] - score [pick + 1];
        ora_s below = score [pick] - score [pick - 1];

        if (below > above * M_E) pick = pick * 2 + 1;
        else if (above > below * M_E) pick = pick * 2 - 1;
        else pick *= 2;
    }
    else
        pick *= 2;

    return pick * s->channels;
}

/* stretch.c:161-333 */
int ora_stretch_feed (OraStretch *s, const ora_s *in, int frames, ora_s *out, double ratio)
{
    ora_s *dst = s->twin ? s->between : out;
    int made = 0, made_twin = 0;
    double rest;

    split_ratio (s, &ratio, &rest);
    if (ratio < 0.5) ratio = 0.5; else if (ratio > 2.0) ratio = 2.0;

    int left = frames * s->channels;

    while (left) {
        int take = left < s->room - s->fill ? left : s->room - s->fill;

        memcpy (s->ring + s->fill, in, take * sizeof (ora_s));
        left -= take; in += take; s->fill += take;

        while (s->mark >= s->hi && s->fill - s->mark >= s->hi * (s->quick ? 3 : 2)) {
            ora_s *at = s->ring + s->mark;
            int p = (ratio != 1.0 || s->drift) ? (s->quick ? pick_period_quick (s, at) : pick_period (s, at)) : s->hi;
            double step;

            /* half-integer ratio of this step: nearest when on target, else the one that pulls the error back */
            if (s->drift == 0.0) step = floor (ratio * 2.0 + 0.5) / 2.0;
            else if (s->drift > 0.0) step = floor (ratio * 2.0) / 2.0;
            else step = ceil (ratio * 2.0) / 2.0;

            if (step == 0.5) {                          /* two periods -> one */
                crossfade (dst + made, at, at + p, p);
                s->drift += p - (p * 2.0 * ratio);
                made += p; s->mark += p * 2;
            }
            else if (step == 1.0) {                     /* verbatim */
                memcpy (dst + made, at, p * 2 * sizeof (ora_s));
                if (ratio != 1.0) s->drift += (p * 2.0) - (p * 2.0 * ratio);
                else s->drift = 0;
                made += p * 2; s->mark += p * 2;
            }
            else if (step == 1.5) {                     /* two periods -> three */
                memcpy (dst + made, at, p * sizeof (ora_s));
                crossfade (dst + made + p, at + p, at, p);
                memcpy (dst + made + p * 2, at + p, p * sizeof (ora_s));
                s->drift += (p * 3.0) - (p * 2.0 * ratio);
                made += p * 3; s->mark += p * 2;
            }
            else if (step == 2.0) {                     /* one period -> two (twice in quick mode) */
                for (int rep = 0; rep < (s->quick ? 2 : 1); ++rep) {
                    crossfade (dst + made, s->ring + s->mark, s->ring + s->mark - p, p * 2);
                    s->drift += (p * 2.0) - (p * ratio);
                    made += p * 2; s->mark += p;
                }
            }

            if (s->twin) {
                made_twin += ora_stretch_feed (s->twin, dst, made / s->channels, out + made_twin * s->channels, rest);
                made = 0;
            }

            /* keep one longest period of history in front of the mark */
            memmove (s->ring, s->ring + s->mark - s->hi, (s->room - s->mark + s->hi) * sizeof (ora_s));
            s->fill -= s->mark - s->hi;
            s->mark = s->hi;
        }
    }

    /* nothing to stretch and no error outstanding: pass everything pending straight through (stretch.c:314-330) */
    if (ratio == 1.0 && !s->drift && s->fill != s->mark) {
        int pending = s->fill - s->mark;

        if (s->twin)
            made_twin += ora_stretch_feed (s->twin, s->ring + s->mark, pending / s->channels, out + made_twin * s->channels, rest);
        else {
            memcpy (dst + made, s->ring + s->mark, pending * sizeof (ora_s));
            made += pending;
        }

        memmove (s->ring, s->ring + s->fill - s->hi, s->hi * sizeof (ora_s));
        s->fill = s->mark = s->hi;
    }

    return s->twin ? made_twin : made / s->channels;
}

/* stretch.c:335-356 */
int ora_stretch_drain (OraStretch *s, ora_s *out)
{
    int pending = s->fill - s->mark, frames = 0;

    if (s->twin) {
        if (pending)
            frames = ora_stretch_feed (s->twin, s->ring + s->mark, pending / s->channels, out, 1.0);
        if (!frames)
            frames = ora_stretch_drain (s->twin, out);
    }
    else {
        memcpy (out, s->ring + s->mark, pending * sizeof (ora_s));
        frames = pending / s->channels;
    }

    s->mark = s->fill;
    memset (s->ring, 0, s->mark * sizeof (ora_s));
    return frames;
}
