// stretch_kernels.hip — gfx950 kernel of the time-domain harmonic scaler.
//
// Behaviour restated (not translated) from reference stretch.c:161-356 (call state machine), :391-470 / :472-552
// (period search), :560-566 (cross-fade).  One call of stretchProcess / stretchFlush = ONE launch of one persistent
// workgroup of 1024 threads that runs the whole call: the stretcher is a chain of decisions — every step's period
// decides where the next step starts — so there is no parallelism across steps of a stream, only inside a step:
//
//   * period search: a candidate period p needs  miss(p) = sum_{i=p-1..0} |m[i] - m[i+p]|  accumulated in the
//     reference's order (a rounding chain per candidate) — one lane per candidate, ~1000 lanes busy at once; the
//     numerator total(p) is a running sum over candidates (another rounding chain), carried by one lane of the last
//     wave while the others work on the misses; the winner is the LAST candidate reaching the maximum quotient;
//   * the four period-synchronous transformations and the ring compaction are element-parallel over 1024 threads.
//
// All control flow is workgroup-uniform: every thread carries its own copy of the scalar state (mark, fill, drift,
// counts) and updates it with the same arithmetic; only the chosen period travels through LDS.  Streams are
// independent, so throughput scales with the number of concurrent contexts (one workgroup each).
//
// Arithmetic (see oracle/stretch_oracle.c for the derivation from the reference's C types): fabs() in double, sums
// rounded back to the sample type after each addition, (a + b) / 2.0 exact, true division; -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cfloat>
#include "art_internal.h"

namespace {

constexpr int ST_WG = 1024;
constexpr int ST_MONO_MAX = 2 * 2400;              // values of the search buffer (two longest periods)

struct StretchStage {                              // one stage (a cascaded pair has two)
    art_s *ring [2];                               // ping-pong input rings, `room` values each
    art_s *between;                                // stage 1 of a pair: hand-over buffer to stage 2
    int channels, room, lo, hi, quick;
};

struct StretchState { int mark, fill, cur; int pad; double drift; };   // per stage, lives in device memory

struct Scratch {                                   // LDS
    art_s mono [ST_MONO_MAX];
    double addend [2400 + 8];                      // what the numerator chain adds, formed in parallel ahead of it
    art_s red_q [ST_WG / 64]; int red_p [ST_WG / 64];
    int any, pick;
    art_s *score;                                  // device memory: quick mode, quotient per decimated period
    art_s *total;                                  // device memory: numerator per candidate
};

// acc <- (sample type) ((double) acc + v), the "+= fabs (...)" of the reference
__device__ __forceinline__ art_s add_abs (art_s acc, double v) { return (art_s)((double) acc + v); }

template <bool QUICK>
__device__ int pick_period (const StretchStage &S, Scratch &L, const art_s *x)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = S.channels;
    const int span = QUICK ? S.hi / C : (C == 2 ? S.hi : S.hi * 2);      // values of mono[] in use
    // ---- the search signal: channel average (stereo) and / or 2:1 decimation (quick); plain copy for mono
    if (tid == 0) L.any = 0;
    __syncthreads ();
    int nonzero = 0;
    for (int j = tid; j < span; j += ST_WG) {
        art_s v;
        if (QUICK) {
            if (C == 2) { const art_s s = ((x [4 * j] + x [4 * j + 1]) + x [4 * j + 2]) + x [4 * j + 3]; v = (art_s)((double) s / 2.0); }
            else { const art_s s = x [2 * j] + x [2 * j + 1]; v = (art_s)((double) s / 2.0); }
        }
        else if (C == 2) { const art_s s = x [2 * j] + x [2 * j + 1]; v = (art_s)((double) s / 2.0); }
        else v = x [j];
        L.mono [j] = v;
        nonzero |= (v != (art_s) 0);
    }
    if (nonzero) L.any = 1;                        // benign race: everybody writes 1
    __syncthreads ();
    if (!L.any) return S.hi;                       // silence (the reference's energy sum is zero iff every value is)

    const art_s *m = L.mono;
    const int p0 = QUICK ? S.lo / (C * 2) : S.lo / C;
    const int p1 = QUICK ? S.hi / (C * 2) : S.hi / C;                   // inclusive
    // ---- numerators: one rounding chain over the candidates.  Its addends (|a| + |b| in double) are formed by all
    // threads first; the chain itself (last wave, lane 0, concurrent with the misses) then reads them eight at a time, so
    // no LDS latency sits between two of its steps.
    // addend [j], j < p0: builds total (p0);  addend [p], p0 <= p < p1: takes total (p) to total (p + 1)
    for (int j = tid; j < p1; j += ST_WG)
        L.addend [j] = j < p0 ? fabs ((double) m [j]) + fabs ((double) m [j + p0])
                              : fabs ((double) m [2 * j]) + fabs ((double) m [2 * j + 1]);
    __syncthreads ();
    if (wave == ST_WG / 64 - 1) {
        if (lane == 0) {
            art_s total = 0;
            int j = 0;
            for (; j + 8 <= p0; j += 8) {
                double a [8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a [u] = L.addend [j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) total = add_abs (total, a [u]);
            }
            for (; j < p0; ++j) total = add_abs (total, L.addend [j]);
            // total is now total (p0); addend [p0 + k] takes total (p0 + k) to total (p0 + k + 1)
            const int steps = p1 - p0;
            int k = 0;
            for (; k + 8 <= steps; k += 8) {
                double a [8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a [u] = L.addend [p0 + k + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) { L.total [k + u] = total; total = add_abs (total, a [u]); }
            }
            for (; k < steps; ++k) { L.total [k] = total; total = add_abs (total, L.addend [p0 + k]); }
            L.total [steps] = total;
        }
    }
    // ---- misses: one lane per candidate, the reference's descending order
    art_s my_q = -1; int my_p = -1;                // this thread's best (quotients are >= 0)
    art_s miss_of [3]; int cand_of [3]; int ncand = 0;
    if (wave < ST_WG / 64 - 1) {
        for (int p = p0 + tid; p <= p1; p += ST_WG - 64) {
            // "miss += fabs (d)" adds in double and rounds back to the sample type.  With both operands sample-type
            // values that equals the plain sample-type addition (double rounding is innocuous for + when the wide format
            // has >= 2p + 2 bits: 53 >= 50).
            art_s miss = 0;
            int i = p - 1;
            for (; i >= 7; i -= 8) {               // eight elements per trip: the sixteen LDS reads are issued together
                art_s d [8];
#pragma unroll
                for (int u = 0; u < 8; ++u) d [u] = m [i - u] - m [i - u + p];
#pragma unroll
                for (int u = 0; u < 8; ++u) miss = miss + (d [u] < (art_s) 0 ? -d [u] : d [u]);
            }
            for (; i >= 0; --i) {
                const art_s d = m [i] - m [i + p];
                miss = miss + (d < (art_s) 0 ? -d : d);
            }
            miss_of [ncand] = miss; cand_of [ncand] = p; ++ncand;       // <= 3 candidates per thread (2377 / 960)
        }
    }
    __syncthreads ();                              // numerators complete
    for (int k = 0; k < ncand; ++k) {
        const int p = cand_of [k];
        const art_s q = (miss_of [k] == (art_s) 0) ? (art_s) FLT_MAX : L.total [p - p0] / miss_of [k];
        if (QUICK) L.score [p - p0 + 1] = q;       // (+1: the refinement looks one below p0's slot never, but keeps indices >= 0)
        if (q >= my_q) { my_q = q; my_p = p; }     // candidates of a thread ascend: ">=" keeps the later one
    }
    // ---- winner: maximum quotient, the LARGEST period among equals (the reference scans upwards with ">=")
    for (int o = 32; o >= 1; o >>= 1) {
        const art_s oq = __shfl_xor (my_q, o); const int op = __shfl_xor (my_p, o);
        if (oq > my_q || (oq == my_q && op > my_p)) { my_q = oq; my_p = op; }
    }
    if (lane == 0) { L.red_q [wave] = my_q; L.red_p [wave] = my_p; }
    __syncthreads ();
    if (tid == 0) {
        art_s bq = L.red_q [0]; int bp = L.red_p [0];
        for (int w = 1; w < ST_WG / 64; ++w)
            if (L.red_q [w] > bq || (L.red_q [w] == bq && L.red_p [w] > bp)) { bq = L.red_q [w]; bp = L.red_p [w]; }
        int pick = bp;
        if (QUICK) {
            if (pick * C * 2 != S.lo && pick * C * 2 != S.hi) {
                const art_s here = L.score [pick - p0 + 1];
                const art_s above = here - L.score [pick - p0 + 2];
                const art_s below = here - L.score [pick - p0];
                if ((double) below > (double) above * M_E) pick = pick * 2 + 1;
                else if ((double) above > (double) below * M_E) pick = pick * 2 - 1;
                else pick *= 2;
            }
            else pick *= 2;
        }
        L.pick = pick * C;
    }
    __syncthreads ();
    return L.pick;
}

__device__ __forceinline__ void crossfade (art_s *out, const art_s *from, const art_s *to, int n)
{
    for (int i = threadIdx.x; i < n; i += ST_WG) {
        const art_s a = from [i] * (art_s)(n - i);
        const art_s b = to [i] * (art_s) i;
        const art_s s = a + b;
        out [i] = s / (art_s) n;
    }
}

__device__ __forceinline__ void copy_values (art_s *dst, const art_s *src, int n)
{
    for (int i = threadIdx.x; i < n; i += ST_WG) dst [i] = src [i];
}

__device__ __forceinline__ void split_ratio (bool paired, double &ratio, double &rest)
{
    rest = 1.0;
    if (!paired) return;
    if (ratio < 0.5) { rest = ratio / 0.5; ratio = 0.5; }
    else if (ratio > 2.0) { rest = ratio / 2.0; ratio = 2.0; }
}

// One stage consuming `values` input values.  PAIRED: this is stage 1 of a cascade and hands every step's output to
// stage 2 (the same function, unpaired).  Returns FRAMES written to `out` (by the last stage).
template <bool PAIRED>
__device__ int feed (const StretchStage *stages, StretchState *states, Scratch &L, const art_s *in, int values,
                     art_s *out, double ratio)
{
    const StretchStage &S = stages [0];
    StretchState st = states [0];                  // every thread reads the same words; thread 0 writes them back
    art_s *dst = PAIRED ? S.between : out;
    int made = 0, made_next = 0;
    double rest;

    split_ratio (PAIRED, ratio, rest);
    if (ratio < 0.5) ratio = 0.5; else if (ratio > 2.0) ratio = 2.0;
    __syncthreads ();                              // (state read above before anybody writes it back)

    int left = values;
    while (left) {
        const int take = left < S.room - st.fill ? left : S.room - st.fill;
        // Full ring and nothing processable: only reachable by feeding after a flush without a reset, where the reference's
        // loop (stretch.c:195-212: nothing copied, nothing processed, num_samples unchanged) never ends.  A spinning workgroup would take the device with it: stop, keep
        // what the call has produced, drop the rest of its input.
        if (take == 0 && !(st.mark >= S.hi && st.fill - st.mark >= S.hi * (S.quick ? 3 : 2))) break;
        copy_values (S.ring [st.cur] + st.fill, in, take);
        left -= take; in += take; st.fill += take;
        __syncthreads ();

        while (st.mark >= S.hi && st.fill - st.mark >= S.hi * (S.quick ? 3 : 2)) {
            art_s *ring = S.ring [st.cur];
            art_s *at = ring + st.mark;
            int p = S.hi;
            if (ratio != 1.0 || st.drift != 0.0)
                p = S.quick ? pick_period<true> (S, L, at) : pick_period<false> (S, L, at);
            double step;
            if (st.drift == 0.0) step = floor (ratio * 2.0 + 0.5) / 2.0;
            else if (st.drift > 0.0) step = floor (ratio * 2.0) / 2.0;
            else step = ceil (ratio * 2.0) / 2.0;

            if (step == 0.5) {
                crossfade (dst + made, at, at + p, p);
                st.drift += p - (p * 2.0 * ratio);
                made += p; st.mark += p * 2;
            }
            else if (step == 1.0) {
                copy_values (dst + made, at, p * 2);
                if (ratio != 1.0) st.drift += (p * 2.0) - (p * 2.0 * ratio);
                else st.drift = 0;
                made += p * 2; st.mark += p * 2;
            }
            else if (step == 1.5) {
                copy_values (dst + made, at, p);
                crossfade (dst + made + p, at + p, at, p);
                copy_values (dst + made + p * 2, at + p, p);
                st.drift += (p * 3.0) - (p * 2.0 * ratio);
                made += p * 3; st.mark += p * 2;
            }
            else {                                 // 2.0
                for (int rep = 0; rep < (S.quick ? 2 : 1); ++rep) {
                    crossfade (dst + made, ring + st.mark, ring + st.mark - p, p * 2);
                    st.drift += (p * 2.0) - (p * ratio);
                    made += p * 2; st.mark += p;
                }
            }
            __syncthreads ();                      // step output complete (stage 2 / compaction read it)

            if (PAIRED) {
                made_next += feed<false> (stages + 1, states + 1, L, dst, made, out + made_next * S.channels, rest);
                made = 0;
            }

            // keep one longest period of history in front of the mark: compact into the other ring
            const int keep = S.room - st.mark + S.hi;
            copy_values (S.ring [st.cur ^ 1], ring + st.mark - S.hi, keep);
            st.cur ^= 1;
            st.fill -= st.mark - S.hi;
            st.mark = S.hi;
            __syncthreads ();
        }
    }

    if (ratio == 1.0 && st.drift == 0.0 && st.fill != st.mark) {        // nothing to stretch: pass the pending values on
        art_s *ring = S.ring [st.cur];
        const int pending = st.fill - st.mark;
        if (PAIRED)
            made_next += feed<false> (stages + 1, states + 1, L, ring + st.mark, pending, out + made_next * S.channels, rest);
        else {
            copy_values (dst + made, ring + st.mark, pending);
            made += pending;
        }
        __syncthreads ();
        copy_values (S.ring [st.cur ^ 1], ring + st.fill - S.hi, S.hi);
        st.cur ^= 1;
        st.fill = st.mark = S.hi;
        __syncthreads ();
    }

    if (threadIdx.x == 0) states [0] = st;
    __syncthreads ();
    return PAIRED ? made_next : made / S.channels;
}

// everything still buffered, at normal speed (stretch.c:335-356)
template <bool PAIRED>
__device__ int drain (const StretchStage *stages, StretchState *states, Scratch &L, art_s *out)
{
    const StretchStage &S = stages [0];
    StretchState st = states [0];
    __syncthreads ();
    const int pending = st.fill - st.mark;
    int frames = 0;
    if (PAIRED) {
        if (pending) frames = feed<false> (stages + 1, states + 1, L, S.ring [st.cur] + st.mark, pending, out, 1.0);
        if (!frames) frames = drain<false> (stages + 1, states + 1, L, out);
    }
    else {
        copy_values (out, S.ring [st.cur] + st.mark, pending);
        frames = pending / S.channels;
    }
    __syncthreads ();
    st.mark = st.fill;
    for (int i = threadIdx.x; i < st.mark; i += ST_WG) S.ring [st.cur][i] = 0;
    if (threadIdx.x == 0) states [0] = st;
    __syncthreads ();
    return frames;
}

struct StretchLaunch { StretchStage stage [2]; StretchState *state; art_s *total, *score; int paired; };

__global__ __launch_bounds__ (ST_WG)
void stretch_call_kernel (StretchLaunch a, const art_s *in, int frames, art_s *out, double ratio, int flush, int *result)
{
    __shared__ Scratch L;
    if (threadIdx.x == 0) { L.total = a.total; L.score = a.score; }
    __syncthreads ();
    int made;
    if (flush) made = a.paired ? drain<true> (a.stage, a.state, L, out) : drain<false> (a.stage, a.state, L, out);
    else {
        const int values = frames * a.stage [0].channels;
        made = a.paired ? feed<true> (a.stage, a.state, L, in, values, out, ratio) : feed<false> (a.stage, a.state, L, in, values, out, ratio);
    }
    if (threadIdx.x == 0) *result = made;
}

// one workgroup per stream: the single-stream kernel's body on item blockIdx.x; the stream's state after the call is
// copied beside the frame count so that the host reads everything back in one transfer
__global__ __launch_bounds__ (ST_WG)
void stretch_batch_kernel (const ArtStretchItem *items, ArtStretchDone *done)
{
    __shared__ Scratch L;
    __shared__ StretchStage stage [2];
    const ArtStretchItem &it = items [blockIdx.x];
    if (threadIdx.x == 0) {
        for (int s = 0; s < 2; ++s) {
            stage [s].ring [0] = (art_s *) it.args.ring [s][0]; stage [s].ring [1] = (art_s *) it.args.ring [s][1];
            stage [s].between = (art_s *) it.args.between;
            stage [s].channels = it.args.channels; stage [s].room = it.args.room; stage [s].lo = it.args.lo; stage [s].hi = it.args.hi;
            stage [s].quick = it.args.quick;
        }
        L.total = (art_s *) it.args.total; L.score = (art_s *) it.args.score;
    }
    StretchState *state = (StretchState *) it.args.state;
    __syncthreads ();
    int made;
    if (it.flush) made = it.args.paired ? drain<true> (stage, state, L, it.out) : drain<false> (stage, state, L, it.out);
    else if (it.frames <= 0) made = 0;             // as stretchProcessDevice: nothing to do for this stream this round
    else {
        const int values = it.frames * it.args.channels;
        made = it.args.paired ? feed<true> (stage, state, L, it.in, values, it.out, it.ratio) : feed<false> (stage, state, L, it.in, values, it.out, it.ratio);
    }
    __syncthreads ();
    if (threadIdx.x == 0) {
        ArtStretchDone &d = done [blockIdx.x];
        d.made = made; d.pad = 0;
        for (int s = 0; s < 2; ++s) {
            d.state [s].mark = state [s].mark; d.state [s].fill = state [s].fill; d.state [s].cur = state [s].cur;
            d.state [s].pad = 0; d.state [s].drift = state [s].drift;
        }
    }
}

} // namespace

extern "C" int arthip_stretch_batch (const ArtStretchItem *d_items, ArtStretchDone *d_done, int n, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL (stretch_batch_kernel, dim3 (n), dim3 (ST_WG), 0, (hipStream_t) stream, d_items, d_done);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

extern "C" int arthip_stretch_call (const ArtStretchArgs *h, const art_s *d_in, int frames, art_s *d_out, double ratio, int flush,
                                   int *d_result, void *stream)
{
    StretchLaunch a;
    for (int s = 0; s < 2; ++s) {
        a.stage [s].ring [0] = h->ring [s][0]; a.stage [s].ring [1] = h->ring [s][1];
        a.stage [s].between = h->between;
        a.stage [s].channels = h->channels; a.stage [s].room = h->room; a.stage [s].lo = h->lo; a.stage [s].hi = h->hi;
        a.stage [s].quick = h->quick;
    }
    a.state = (StretchState *) h->state;
    a.total = h->total; a.score = h->score;
    a.paired = h->paired;
    hipLaunchKernelGGL (stretch_call_kernel, dim3 (1), dim3 (ST_WG), 0, (hipStream_t) stream, a, d_in, frames, d_out, ratio, flush, d_result);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}
