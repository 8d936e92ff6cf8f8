/* extrapolate_host.c — LPC end-point extrapolation (host side, scalar, at most twice per stream).
 *
 * Behaviour restated from reference extrapolator.c:22-277: a 4-coefficient linear predictor is fitted to the
 * available samples by coordinate descent with a halving step (bounded at 100,000 probes), its reflection
 * (PARCOR) coefficients are clamped to +-0.9999 for stability, degenerate fits fall back to "repeat the last
 * sample" or "silence", and the predictor is then run forward (flush) or backward (prefill).
 * Float/double mixing follows the reference expression by expression, so results are bit-identical to the
 * reference built with C source-order semantics.  Runs on the CPU because it is a serial search over a few
 * hundred samples; the samples it produces are uploaded and consumed by the GPU kernels like any input.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "art_internal.h"

#define ORDER 4
#define PROBE_LIMIT 100000

static void reflection_from_predictor (const double *lpc, double *refl)
{
    double cur [ORDER], next [ORDER];

    memcpy (cur, lpc, sizeof (cur));

    for (int m = ORDER - 1; m >= 0; --m) {
        refl [m] = cur [m];
        double den = 1.0 - (refl [m] * refl [m]);

        if (fabs (den) < 1e-6) {                       /* |k| ~ 1: keep the step-down recursion finite */
            refl [m] = refl [m] < 0.0 ? -0.9999995 : 0.9999995;
            den = 1.0 - (refl [m] * refl [m]);
        }

        for (int i = 0; i < m; ++i)
            next [i] = (cur [i] - refl [m] * cur [m - i - 1]) / den;
        for (int i = 0; i < m; ++i)
            cur [i] = next [i];
    }
}

static void predictor_from_reflection (const double *refl, double *lpc)
{
    for (int i = 0; i < ORDER; ++i) {
        lpc [i] = refl [i];

        for (int j = 0; j < i / 2; ++j) {
            double held = lpc [j];
            lpc [j] += refl [i] * lpc [i - 1 - j];
            lpc [i - 1 - j] += refl [i] * held;
        }

        if (i & 1)
            lpc [i >> 1] += lpc [i >> 1] * refl [i];
    }
}

/* fit coeffs[ORDER] so that x[n] ~ -(sum_c coeffs[ORDER-1-c] * x[n-ORDER+c]) */
static void fit_predictor (const art_s *x, int count, float *coeffs)
{
    const int evals = count - ORDER;
    double energy = 0.0, delta_energy = 0.0, best, step = 3.0 / (1 << 4);
    double *resid = malloc (sizeof (double) * (size_t)(evals > 0 ? evals : 1));
    int probes = 0, accepted = 0;

    memset (coeffs, 0, sizeof (float) * ORDER);

    for (int i = 0; i < evals; ++i) {
        art_s d = x [i + ORDER] - x [i + ORDER - 1];
        delta_energy += d * d;
        energy += x [i + ORDER] * x [i + ORDER];
    }

    if (energy == 0.0) { free (resid); return; }

    best = energy;

    while (best > 0.0 && probes < PROBE_LIMIT) {
        int which;

        for (int k = 0; k < evals; ++k) {                      /* residual of the current predictor */
            double acc = 0.0;
            for (int c = 0; c < ORDER; ++c)
                acc += coeffs [ORDER - c - 1] * x [k + c];
            resid [k] = acc + x [k + ORDER];
        }

        for (which = 0; probes++, which < ORDER; which++) {    /* first coefficient whose +-step helps */
            double down = 0.0, up = 0.0;

            for (int k = 0; k < evals; ++k) {
                double d = x [k + ORDER - which - 1] * step;
                down += (resid [k] - d) * (resid [k] - d);
                up += (resid [k] + d) * (resid [k] + d);
            }

            if (down < best || up < best) {
                if (down < up) { best = down; coeffs [which] -= step; }
                else           { best = up;   coeffs [which] += step; }
                accepted++;
                break;
            }
        }

        if (which == ORDER) {                                  /* nothing helped at this step size */
            if (step > 3.0 / (1 << 22)) step *= 0.5;
            else break;
        }
    }

    free (resid);

    if (accepted) {                                            /* stabilise through reflection coefficients */
        double lpc [ORDER], refl [ORDER];
        int clamped = 0;

        for (int i = 0; i < ORDER; ++i) lpc [i] = coeffs [i];
        reflection_from_predictor (lpc, refl);

        for (int i = 0; i < ORDER; ++i)
            if (fabs (refl [i]) > 0.9999) { refl [i] = refl [i] < 0.0 ? -0.9999 : 0.9999; clamped++; }

        if (clamped) {
            predictor_from_reflection (refl, lpc);
            for (int i = 0; i < ORDER; ++i) coeffs [i] = lpc [i];
        }
    }

    best = 0.0;                                                /* how good is what we ended up with */
    for (int k = 0; k < evals; ++k) {
        double acc = 0.0;
        for (int c = 0; c < ORDER; ++c)
            acc += coeffs [ORDER - c - 1] * x [k + c];
        best += (acc + x [k + ORDER]) * (acc + x [k + ORDER]);
    }

    if (delta_energy < best && delta_energy < energy) {        /* repeating the last sample predicts better */
        memset (coeffs, 0, sizeof (float) * ORDER);
        coeffs [0] = -1.0;
    }
    else if (energy <= best)                                   /* predicting silence is at least as good */
        memset (coeffs, 0, sizeof (float) * ORDER);
}

/* x[0..count) known; writes x[count .. count+extra) */
void art_extrapolate_forward (art_s *x, int count, int extra)
{
    float coeffs [ORDER];

    memset (x + count, 0, sizeof (art_s) * (size_t) extra);
    fit_predictor (x, count, coeffs);

    for (int i = 0; i < extra; ++i) {
        const art_s *tail = x + count - ORDER + i;
        double acc = 0.0;
        for (int c = 0; c < ORDER; ++c)
            acc += tail [c] * coeffs [ORDER - c - 1];
        x [count + i] = -acc;
    }
}

/* newest-first view: known[0] is the most recent of `count` known samples going back in time;
 * fills older[0..extra) with the samples preceding them (older[0] closest in time) */
void art_extrapolate_backward (const art_s *known_newest_last, int count, art_s *older_nearest_first, int extra)
{
    art_s *rev = calloc ((size_t) count + extra, sizeof (art_s));

    for (int i = 0; i < count; ++i)                            /* time-reverse: earliest known sample last */
        rev [i] = known_newest_last [count - 1 - i];
    art_extrapolate_forward (rev, count, extra);
    memcpy (older_nearest_first, rev + count, sizeof (art_s) * (size_t) extra);
    free (rev);
}
