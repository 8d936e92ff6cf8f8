// pcm_kernels.hip — gfx950 kernels for the serial-recurrence parts of the path:
//   biquad section chains      reference biquad.c:106-163 (apply_buffer), :78-102 (apply_sample)
//   float -> integer decimator reference decimator.c:255-283, dither :370-382
//   integer -> float ingest    reference decimator.c:416-450
//
// Both recurrences feed each output back through float rounding (and, in the decimator, through
// floor()), so they cannot be re-associated or scanned in parallel without changing bits.  The
// bit-exact GPU form is one lane per channel walking time serially; channels run side by side in a
// wave.  Compiled with -ffp-contract=off: every multiply and add rounds separately, as in the
// reference.
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdlib>
#include <type_traits>
#include "art_internal.h"

namespace {

struct SectionRegs {
    art_s a [5], b [5];
    art_s x [4], y [4];       // x[0] = most recent input, x[1] the one before, ...
    int order;
};

__device__ __forceinline__ void load_section (SectionRegs &r, const Biquad &f)
{
    const int i = f.index;
#pragma unroll
    for (int k = 0; k < 5; ++k) { r.a [k] = f.a [k]; r.b [k] = f.b [k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.x [k] = f.x [(i - k) & 3]; r.y [k] = f.y [(i - k) & 3]; }
    r.order = f.order;
}

__device__ __forceinline__ void store_section (Biquad &f, const SectionRegs &r, int steps, bool mask_index)
{
    int i = f.index;
    if (mask_index) i &= 3;
    i += steps;
    if (mask_index) i &= 3;
#pragma unroll
    for (int k = 0; k < 4; ++k) { f.x [(i - k) & 3] = r.x [k]; f.y [(i - k) & 3] = r.y [k]; }
    f.index = i;
}

__device__ __forceinline__ void push (SectionRegs &r, art_s in, art_s out)
{
    r.x [3] = r.x [2]; r.x [2] = r.x [1]; r.x [1] = r.x [0]; r.x [0] = in;
    r.y [3] = r.y [2]; r.y [2] = r.y [1]; r.y [1] = r.y [0]; r.y [0] = out;
}

// buffer form: in*a0, then for k = 1..order: + x_k*a_k, - b_k*y_k, strictly left to right
__device__ __forceinline__ art_s step_buffer_order (SectionRegs &r, art_s in)
{
    art_s acc = in * r.a [0];
#pragma unroll
    for (int k = 1; k <= 4; ++k)
        if (k <= r.order) {
            art_s fwd = r.x [k - 1] * r.a [k];
            acc = acc + fwd;
            art_s back = r.b [k] * r.y [k - 1];
            acc = acc - back;
        }
    push (r, in, acc);
    return acc;
}

// per-sample form: in*a0, then for k = order..1: += (x_k*a_k - b_k*y_k)
__device__ __forceinline__ art_s step_sample_order (SectionRegs &r, art_s in)
{
    art_s acc = in * r.a [0];
#pragma unroll
    for (int k = 4; k >= 1; --k)
        if (k <= r.order) {
            art_s fwd = r.x [k - 1] * r.a [k];
            art_s back = r.b [k] * r.y [k - 1];
            art_s term = fwd - back;
            acc = acc + term;
        }
    push (r, in, acc);
    return acc;
}

constexpr int MAX_CHAIN = 4;

__global__ void biquad_chain_kernel (Biquad *sections, int C, int S, art_s *buf, int frames, int stride, int sample_form)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;

    SectionRegs r [MAX_CHAIN];
#pragma unroll
    for (int s = 0; s < MAX_CHAIN; ++s)
        if (s < S) load_section (r [s], sections [(size_t) c * S + s]);

    art_s *p = buf + c;
    for (int i = 0; i < frames; ++i, p += stride) {
        art_s v = *p;
#pragma unroll
        for (int s = 0; s < MAX_CHAIN; ++s)
            if (s < S) v = sample_form ? step_sample_order (r [s], v) : step_buffer_order (r [s], v);
        *p = v;
    }

#pragma unroll
    for (int s = 0; s < MAX_CHAIN; ++s)
        if (s < S) store_section (sections [(size_t) c * S + s], r [s], frames, sample_form != 0);
}

// ---------------------------------------------------------------------------------------------------
// Bit-exact biquad cascade, parallel over TIME (biquad_spec_kernel + biquad_commit_kernel).
//
// The recurrence rounds after every operation (reference biquad.c:138-146), so it cannot be re-associated; but a
// STABLE filter forgets its state: two runs over the same input that start from different states converge
// geometrically, and once their rounded states coincide at one sample they coincide for ever.  So every chunk of L
// frames is computed by its own lane in the reference's exact operation order, started WARM-UP frames early from a
// zero state (section s starts (S - s) W frames early, so that it is fed converged outputs of section s - 1), and
// records the state it reached at its chunk's first frame and the state it left at its last.  Chunk 0 starts from the
// carried-in state and is exact; chunk k is exact iff chunk k - 1 is exact and the state chunk k reached after its
// warm-up equals, bit for bit, the state chunk k - 1 left.  The commit kernel checks every boundary in parallel; in the
// (rare) case of a mismatch one lane recomputes from the exact state until it rejoins a speculative trajectory — in the
// worst case everything, serially, so the result is exact whatever the filter.  W comes from the decay of the
// recursive part (host side); filters too narrow to forget within the cap take the serial kernels instead.
// Input and output are separate buffers (a chunk's warm-up reads frames its predecessor writes).
// ---------------------------------------------------------------------------------------------------
struct SpecState { art_s x [4], y [4]; };           // one section's delay lines, newest first

template <int S>
__device__ __forceinline__ bool same_state (const SpecState *a, const SpecState *b)
{
    bool same = true;
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // bit patterns, not values: -0.0 and 0.0 carry on differently through a multiply by a negative coefficient
            same = same && __builtin_bit_cast (typename std::conditional<sizeof (art_s) == 4, uint32_t, uint64_t>::type, a [s].x [k]) ==
                           __builtin_bit_cast (typename std::conditional<sizeof (art_s) == 4, uint32_t, uint64_t>::type, b [s].x [k]);
            same = same && __builtin_bit_cast (typename std::conditional<sizeof (art_s) == 4, uint32_t, uint64_t>::type, a [s].y [k]) ==
                           __builtin_bit_cast (typename std::conditional<sizeof (art_s) == 4, uint32_t, uint64_t>::type, b [s].y [k]);
        }
    return same;
}

template <int S>
__device__ __forceinline__ void get_state (SpecState *dst, const SectionRegs *r)
{
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) { dst [s].x [k] = r [s].x [k]; dst [s].y [k] = r [s].y [k]; }
}

template <int S>
__device__ __forceinline__ void put_state (SectionRegs *r, const SpecState *src)
{
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) { r [s].x [k] = src [s].x [k]; r [s].y [k] = src [s].y [k]; }
}

// frames [from, to) of channel c through sections 0 .. ACTIVE-1, exact order; STORE: the results go to `out`.
// The recurrence is latency-bound and a lane's loads are independent of it: two batches of U frames are kept in flight
// (the next batch's loads are issued before the current batch's dependent chain starts).
template <int S, int ACTIVE, bool STORE>
__device__ __forceinline__ void spec_run (SectionRegs *r, const art_s *in, int stride, art_s *out, int out_stride, int c, int from, int to)
{
    constexpr int U = 8;
    auto fetch = [&] (art_s (&v) [U], int n) {
#pragma unroll
        for (int u = 0; u < U; ++u) v [u] = in [(size_t)(n + u) * stride + c];
    };
    auto work = [&] (art_s (&v) [U], int n) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int s = 0; s < ACTIVE; ++s) v [u] = step_buffer_order (r [s], v [u]);
        }
        if (STORE) {
#pragma unroll
            for (int u = 0; u < U; ++u) out [(size_t)(n + u) * out_stride + c] = v [u];
        }
    };
    int n = from;
    const int batches = (to - from) / U;
    if (batches > 0) {
        art_s cur [U], nxt [U];
        fetch (cur, n);
        for (int b = 0; b < batches; ++b, n += U) {
            const bool more = b + 1 < batches;
            if (more) fetch (nxt, n + U);                  // in flight while this batch's dependent chain runs
            work (cur, n);
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) cur [u] = nxt [u];
            }
        }
    }
    for (; n < to; ++n) {
        art_s v = in [(size_t) n * stride + c];
#pragma unroll
        for (int s = 0; s < ACTIVE; ++s) v = step_buffer_order (r [s], v);
        if (STORE) out [(size_t) n * out_stride + c] = v;
    }
}

// the warm-up of a speculative chunk: W frames with section 0 alone, W more with sections 0-1, ... (section s joins (S - s) W
// frames before the chunk's first frame, fed by sections that have already converged)
template <int S, int J = 0>
__device__ __forceinline__ void spec_warm_up (SectionRegs *r, const art_s *in, int stride, int c, int begin, int W)
{
    if constexpr (J < S) {
        spec_run<S, J + 1, false> (r, in, stride, nullptr, 0, c, begin + J * W, begin + (J + 1) * W);
        spec_warm_up<S, J + 1> (r, in, stride, c, begin, W);
    }
}

// task = (chunk k, channel c), c fastest: the lanes of a wave read neighbouring channels of a few chunks
template <int S>
__global__ __launch_bounds__ (256)
void biquad_spec_kernel (const Biquad *sections, int C, int K, int L, int W, const art_s *in, int stride, art_s *out, int out_stride,
                         int frames, SpecState *starts, SpecState *ends)
{
    const long task = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (task >= (long) C * K) return;
    const int c = (int)(task % C), k = (int)(task / C);
    const int first = k * L, last = min (first + L, frames);

    SectionRegs r [S];
#pragma unroll
    for (int s = 0; s < S; ++s) load_section (r [s], sections [(size_t) c * S + s]);

    const int begin = first - S * W;
    if (begin > 0) {
        // speculative start: silence behind every section
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) { r [s].x [q] = 0; r [s].y [q] = 0; }
        spec_warm_up<S> (r, in, stride, c, begin, W);
    }
    else if (first > 0)
        // close to the start of the call: from the carried-in state through frames [0, first) — exact, nothing stored
        spec_run<S, S, false> (r, in, stride, nullptr, 0, c, 0, first);

    SpecState st [S];
    get_state<S> (st, r);
#pragma unroll
    for (int s = 0; s < S; ++s) starts [((size_t) c * K + k) * S + s] = st [s];

    spec_run<S, S, true> (r, in, stride, out, out_stride, c, first, last);

    get_state<S> (st, r);
#pragma unroll
    for (int s = 0; s < S; ++s) ends [((size_t) c * K + k) * S + s] = st [s];
}

// Every chunk boundary checked, one thread each: flags [c][k] = chunk k did not start from the state chunk k-1 left;
// first_bad [c] = the first such k of the channel (stays at its armed value, beyond any chunk count, when there is none).
template <int S>
__global__ __launch_bounds__ (256)
void biquad_check_kernel (int C, int K, const SpecState *starts, const SpecState *ends, unsigned char *bad, int *first_bad)
{
    const long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long) C * K) return;
    const int c = (int)(t / K), k = (int)(t % K);
    if (k == 0) return;
    const bool ok = same_state<S> (starts + ((size_t) c * K + k) * S, ends + ((size_t) c * K + k - 1) * S);
    bad [(size_t) c * K + k] = ok ? 0 : 1;
    if (!ok) atomicMin (first_bad + c, k);
}

// One thread per channel: repairs from the first mismatch (rare: recomputes from the exact state until it rejoins a
// speculative trajectory — in the worst case everything, serially), then the channel's final state goes back into
// `sections`.  repairs: running count of chunks recomputed (diagnostics).  first_bad is re-armed for the next call.
template <int S>
__global__ __launch_bounds__ (64)
void biquad_commit_kernel (Biquad *sections, int C, int K, int L, const art_s *in, int stride, art_s *out, int out_stride, int frames,
                           const SpecState *starts, SpecState *ends, const unsigned char *bad, int *first_bad, unsigned int *repairs)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const SpecState *st = starts + (size_t) c * K * S;
    SpecState *en = ends + (size_t) c * K * S;
    const unsigned char *flags = bad + (size_t) c * K;

    SectionRegs r [S];
#pragma unroll
    for (int s = 0; s < S; ++s) load_section (r [s], sections [(size_t) c * S + s]);

    int k = first_bad [c];
    first_bad [c] = INT_MAX;
    unsigned int redone = 0;
    while (k < K) {
        // chunk k again, from the exact state its predecessor left
        put_state<S> (r, en + (size_t)(k - 1) * S);
        const int first = k * L, last = min (first + L, frames);
        spec_run<S, S, true> (r, in, stride, out, out_stride, c, first, last);
        SpecState now [S];
        get_state<S> (now, r);
#pragma unroll
        for (int s = 0; s < S; ++s) en [(size_t) k * S + s] = now [s];
        ++redone;
        if (k + 1 >= K) break;
        if (same_state<S> (now, st + (size_t)(k + 1) * S)) {
            // rejoined the speculative trajectory: everything up to the next recorded mismatch stands
            int next = k + 2;
            while (next < K && !flags [next]) ++next;
            k = next;
        }
        else ++k;
    }
    if (redone) atomicAdd (repairs, redone);

    put_state<S> (r, en + (size_t)(K - 1) * S);
#pragma unroll
    for (int s = 0; s < S; ++s) store_section (sections [(size_t) c * S + s], r [s], frames, false);
}

__device__ __forceinline__ uint32_t lcg (uint32_t r) { return ((r << 4) - r) ^ 1u; }

__global__ void decimate_kernel (ArtDecArgs a, const art_s *in, long in_pitch, int frames, unsigned char *out, long out_pitch)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.C) return;

    art_s fb = a.feedback [c];
    uint32_t gen = a.dither_on ? a.gens [c] : 0u;
    SectionRegs sh;
    if (a.shaping_on) load_section (sh, a.shapers [c]);

    const int pad = a.bytes - ((a.bits + 7) / 8);
    const int hi = (1 << (a.bits - 1)) - 1, lo = ~hi;
    const int shift = (24 - a.bits) % 8;
    const uint32_t bias = a.bits <= 8 ? 128u : 0u;
    unsigned long long clips = 0;

    for (int i = 0; i < frames; ++i) {
        const art_s s = in_pitch ? in [(size_t) c * in_pitch + i] : in [(size_t) i * a.C + c];
        art_s dither = 0.0f;

        if (a.dither_on) {
            const uint32_t start = gen;
            uint32_t r = lcg (lcg (start));
            const uint32_t first = a.dither_type < 0 ? ~start : a.dither_type > 0 ? start : ~r;
            r = lcg (lcg (lcg (r)));
            gen = r;
            const double tri = ((double)((first >> 1) + (r >> 1)) / 2147483648.0) - 1.0;
            dither = (art_s) tri;
        }

        const art_s scaled = s * a.scale;
        const art_s code = scaled - fb;
        const art_s dithered = code + dither;
        int q = (int) floor ((double) dithered + 0.5);

        if (a.shaping_on) {
            const art_s err = (art_s) q - code;
            fb = step_sample_order (sh, err);
        }

        if (q > hi) { q = hi; clips++; }
        else if (q < lo) { q = lo; clips++; }

        const uint32_t v = ((uint32_t) q << shift) + bias;
        unsigned char *o = out_pitch ? out + (size_t) c * out_pitch + (size_t) i * a.bytes
                                     : out + ((size_t) i * a.C + c) * a.bytes;
        for (int j = 0; j < pad; ++j) *o++ = 0;
        *o++ = (unsigned char) v;
        if (a.bits > 8) { *o++ = (unsigned char)(v >> 8); if (a.bits > 16) *o++ = (unsigned char)(v >> 16); }
    }

    a.feedback [c] = fb;
    if (a.dither_on) a.gens [c] = gen;
    if (a.shaping_on) store_section (a.shapers [c], sh, frames, true);
    if (clips) atomicAdd (a.clipped, clips);
}


// ---------------------------------------------------------------------------------------------------
// LDS-staged forms (interleaved frames, stride == channel count).  The recurrences stay one lane per
// channel — that is what bit-exactness costs — but memory traffic is taken off the serial path: the whole
// workgroup moves a chunk of frames HBM <-> LDS with coalesced 16-byte accesses, then lanes 0..Cg-1 of
// wave 0 run the chunk out of LDS (inputs are known ahead of the recurrence, so the LDS reads pipeline).
// Algorithmic HBM bytes per sample: biquad 8 (in-place), decimator 4 + output bytes.
// ---------------------------------------------------------------------------------------------------
constexpr int ST_THREADS = 256;
constexpr int ST_CHUNK_FLOATS = ART_WIDE ? 4096 : 8192;   // 32 KiB of samples per chunk

__global__ __launch_bounds__ (ST_THREADS)
void biquad_chain_lds_kernel (Biquad *sections, int C, int S, art_s *buf, int frames)
{
    __shared__ __attribute__ ((aligned (16))) art_s tile [ST_CHUNK_FLOATS];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * 64, Cg = min (64, C - c0);          // this block's channel group
    const int chunk_frames = ST_CHUNK_FLOATS / Cg;

    SectionRegs r [MAX_CHAIN];
    if (tid < Cg) {
#pragma unroll
        for (int s = 0; s < MAX_CHAIN; ++s)
            if (s < S) load_section (r [s], sections [(size_t)(c0 + tid) * S + s]);
    }

    for (int f0 = 0; f0 < frames; f0 += chunk_frames) {
        const int nf = min (chunk_frames, frames - f0);
        // HBM -> LDS (coalesced when the group is the whole frame)
        for (int e = tid; e < nf * Cg; e += ST_THREADS) {
            const int f = e / Cg, c = e - f * Cg;
            tile [e] = buf [(size_t)(f0 + f) * C + c0 + c];
        }
        __syncthreads ();
        if (tid < Cg) {
            art_s *p = tile + tid;
            for (int f = 0; f < nf; ++f, p += Cg) {
                art_s v = *p;
#pragma unroll
                for (int s = 0; s < MAX_CHAIN; ++s)
                    if (s < S) v = step_buffer_order (r [s], v);
                *p = v;
            }
        }
        __syncthreads ();
        for (int e = tid; e < nf * Cg; e += ST_THREADS) {
            const int f = e / Cg, c = e - f * Cg;
            buf [(size_t)(f0 + f) * C + c0 + c] = tile [e];
        }
        __syncthreads ();
    }

    if (tid < Cg) {
#pragma unroll
        for (int s = 0; s < MAX_CHAIN; ++s)
            if (s < S) store_section (sections [(size_t)(c0 + tid) * S + s], r [s], frames, false);
    }
}


// ---- order-2 cascade, feed-forward split + section pipeline -----------------------------------------
// A lone wave issues one instruction every ~4 cycles whatever the number of active lanes, so with 8 channels
// the serial lanes are issue- and latency-bound: what counts is instructions (and dependent operations) per
// sample in the recurrence.  Of the nine operations of a section only five depend on earlier outputs; the rest is
// feed-forward and is done for a whole chunk at once by helper waves (same operations, same order, same
// roundings):
//
//     helpers       u[n]  = (x[n]*a0) + (x[n-1]*a1)          p[n] = x[n-2]*a2
//     serial lane   y[n]  = ((u[n] - (b1*y[n-1])) + p[n]) - (b2*y[n-2])
//
// u/p/y live in LDS per channel (time-contiguous: one ds_read_b128 feeds four samples, fetched one block of
// eight samples ahead of the recurrence; row pitch = 4 mod 32 words keeps the channels of a wave on different
// banks).  The workgroup is a four-stage pipeline over chunks, one barrier per step `it`, all stages of a step
// running concurrently on different waves (different SIMDs of the CU):
//
//     wave 2      feed-forward 1 of chunk it+1 (inputs fetched from HBM one step earlier) | fetch chunk it+2
//     wave 3      store chunk it-3 | feed-forward 2 of chunk it-1
//     wave 0      section 1 of chunk it
//     wave 1      section 2 of chunk it-2
//
// The first chunk takes the remainder, every later chunk has the same length (a multiple of 4).
constexpr int FF_CAP = ART_WIDE ? 1536 : 3072;      // samples per LDS array (12 KiB); 9 arrays
constexpr int FF_HELPERS = 64;                     // threads per helper role (one wave each)
constexpr int FF_RUN = 48;                         // frames per helper lane and chunk (upper bound)

// Raw buffer accesses with hardware bounds checking (word 3 = 0x00020000: raw, 32-bit): an out-of-range load
// returns 0 and an out-of-range store is dropped, so the helper loops carry no per-element predicates.
typedef unsigned int ffu2 __attribute__ ((ext_vector_type (2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ff_rsrc (const void *base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc (const_cast<void *> (base), 0, (int) (bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
}
__attribute__ ((unused)) __device__ __forceinline__ float ff_load (__amdgpu_buffer_rsrc_t r, int off, float) { return __uint_as_float (__builtin_amdgcn_raw_buffer_load_b32 (r, off, 0, 0)); }
__attribute__ ((unused)) __device__ __forceinline__ double ff_load (__amdgpu_buffer_rsrc_t r, int off, double)
{
    const ffu2 v = __builtin_amdgcn_raw_buffer_load_b64 (r, off, 0, 0);
    return __hiloint2double ((int) v.y, (int) v.x);
}
__attribute__ ((unused)) __device__ __forceinline__ void ff_store (__amdgpu_buffer_rsrc_t r, int off, float v) { __builtin_amdgcn_raw_buffer_store_b32 (__float_as_uint (v), r, off, 0, 0); }
__attribute__ ((unused)) __device__ __forceinline__ void ff_store (__amdgpu_buffer_rsrc_t r, int off, double v)
{
    ffu2 w; w.x = (unsigned int) __double2loint (v); w.y = (unsigned int) __double2hiint (v);
    __builtin_amdgcn_raw_buffer_store_b64 (w, r, off, 0, 0);
}

__device__ __forceinline__ void ff_serial (art_s *row, const art_s *prow, int len, art_s b1, art_s b2, art_s &y1, art_s &y2)
{
    typedef art_s vec4 __attribute__ ((ext_vector_type (4)));
    auto four = [&] (const vec4 u, const vec4 p) -> vec4 {
        vec4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const art_s m = b1 * y1;
            const art_s t2 = u [j] - m;
            const art_s t3 = t2 + p [j];
            const art_s q = b2 * y2;
            const art_s v = t3 - q;
            y [j] = v; y2 = y1; y1 = v;
        }
        return y;
    };
    // blocks of eight samples, two register sets: the reads of the next block are issued before the recurrence of
    // the current one starts (sched_barrier keeps the compiler from sinking them), so LDS latency never sits on
    // the serial path.  The look-ahead may read up to one block past `len` (inside the LDS allocation, unused).
    const int nblk = len >> 3;
    if (nblk > 0) {
        vec4 ua0 = *(const vec4 *)(row), ua1 = *(const vec4 *)(row + 4), pa0 = *(const vec4 *)(prow), pa1 = *(const vec4 *)(prow + 4);
        int blk = 0;
        for (; blk + 2 <= nblk; blk += 2) {
            art_s *r = row + 8 * blk; const art_s *pr = prow + 8 * blk;
            const vec4 ub0 = *(const vec4 *)(r + 8), ub1 = *(const vec4 *)(r + 12), pb0 = *(const vec4 *)(pr + 8), pb1 = *(const vec4 *)(pr + 12);
            __builtin_amdgcn_sched_barrier (0);
            const vec4 ya0 = four (ua0, pa0), ya1 = four (ua1, pa1);
            *(vec4 *)(r) = ya0; *(vec4 *)(r + 4) = ya1;
            __builtin_amdgcn_sched_barrier (0);
            ua0 = *(const vec4 *)(r + 16); ua1 = *(const vec4 *)(r + 20); pa0 = *(const vec4 *)(pr + 16); pa1 = *(const vec4 *)(pr + 20);
            __builtin_amdgcn_sched_barrier (0);
            const vec4 yb0 = four (ub0, pb0), yb1 = four (ub1, pb1);
            *(vec4 *)(r + 8) = yb0; *(vec4 *)(r + 12) = yb1;
            __builtin_amdgcn_sched_barrier (0);
        }
        if (blk < nblk) {
            const vec4 ya0 = four (ua0, pa0), ya1 = four (ua1, pa1);
            *(vec4 *)(row + 8 * blk) = ya0; *(vec4 *)(row + 8 * blk + 4) = ya1;
        }
    }
    int f = 8 * nblk;
    for (; f < len; ++f) {
        const art_s m = b1 * y1;
        const art_s t2 = row [f] - m;
        const art_s t3 = t2 + prow [f];
        const art_s q = b2 * y2;
        const art_s v = t3 - q;
        row [f] = v; y2 = y1; y1 = v;
    }
}

template <int S>                                   // S = 1 or 2 order-2 sections per channel
__global__ __launch_bounds__ (ST_THREADS)
void biquad_order2_ff_kernel (Biquad *sections, int C, int stride, art_s *buf, int frames, int cpw)   // stride: values between frames (>= C); cpw: channels per workgroup (<= 64)
{
    extern __shared__ __attribute__ ((aligned (32))) unsigned char ff_lds [];
    art_s *const A1 = (art_s *) ff_lds;            // [3][FF_CAP]  u1 -> y1, by chunk % 3
    art_s *const B1 = A1 + 3 * FF_CAP;             // [2][FF_CAP]  p1, by chunk parity
    art_s *const A2 = B1 + 2 * FF_CAP;             // [2][FF_CAP]  u2 -> y2
    art_s *const B2 = A2 + 2 * FF_CAP;             // [2][FF_CAP]  p2
    __shared__ art_s ffc [2][64][3];               // a0, a1, a2 per section and channel
    __shared__ art_s xtail [2][64][2];             // the two inputs preceding a chunk, by chunk parity: [0] nearest
    __shared__ art_s mtail [3][64][2];             // the two section-1 outputs preceding a chunk (chunk % 3)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * cpw, Cg = min (cpw, C - c0);

    // Chunk geometry.  A helper lane owns `run` consecutive frames of one channel (hc) in every chunk, `runs` lanes
    // per channel; a chunk is exactly runs*run frames (run a multiple of 4), the first chunk takes the remainder.
    int pitch = FF_CAP / Cg;
    pitch = pitch >= 36 ? ((pitch - 4) / 32) * 32 + 4 : pitch & ~3;
    const int runs = FF_HELPERS / Cg;
    const int hc = lane % Cg, hr = lane / Cg;
    int run = min (FF_RUN, ((pitch >= 36 ? pitch - 4 : pitch) / runs) & ~3);
    {   // no more chunk than the call has frames (short calls: several short chunks keep all four stages busy)
        const int want = (((frames + 3) / 4 + runs - 1) / runs + 3) & ~3;
        run = max (4, min (run, want));
    }
    const int len = runs * run;                                     // every chunk but the first
    const int nchunks = (frames + len - 1) / len;
    const int len0 = frames - (nchunks - 1) * len;                  // 1 .. len
    auto chunk_start = [&] (int k) { return k == 0 ? 0 : len0 + (k - 1) * len; };
    auto chunk_len = [&] (int k) { return k == 0 ? len0 : len; };
    const int f0 = hr * run;

    // per-lane recurrence state: wave 0 owns section 1, wave 1 section 2
    art_s b1 = 0, b2 = 0, y1 = 0, y2 = 0;
    art_s xt [4] = { 0, 0, 0, 0 };                 // the call's last four inputs (section 1's x history afterwards)
    if (tid < Cg) {
        const Biquad &f1 = sections [(size_t)(c0 + tid) * S];
        ffc [0][tid][0] = f1.a [0]; ffc [0][tid][1] = f1.a [1]; ffc [0][tid][2] = f1.a [2];
        xtail [0][tid][0] = f1.x [f1.index & 3]; xtail [0][tid][1] = f1.x [(f1.index - 1) & 3];
        b1 = f1.b [1]; b2 = f1.b [2]; y1 = f1.y [f1.index & 3]; y2 = f1.y [(f1.index - 1) & 3];
#pragma unroll
        for (int k = 0; k < 4; ++k) xt [k] = buf [(size_t)(frames - 1 - k) * stride + c0 + tid];     // frames >= 4 (launcher)
        if (S == 2) {
            const Biquad &f2 = sections [(size_t)(c0 + tid) * S + 1];
            ffc [1][tid][0] = f2.a [0]; ffc [1][tid][1] = f2.a [1]; ffc [1][tid][2] = f2.a [2];
            mtail [0][tid][0] = f2.x [f2.index & 3]; mtail [0][tid][1] = f2.x [(f2.index - 1) & 3];
        }
    }
    else if (S == 2 && wave == 1 && lane < Cg) {
        const Biquad &f2 = sections [(size_t)(c0 + lane) * S + 1];
        b1 = f2.b [1]; b2 = f2.b [2]; y1 = f2.y [f2.index & 3]; y2 = f2.y [(f2.index - 1) & 3];
    }
    __syncthreads ();

    const int store_lag = S == 2 ? 3 : 1;          // chunk k leaves the last section in step k + store_lag - 1
    const int esz = (int) sizeof (art_s);
    art_s xr [FF_RUN + 2];                         // wave 2: its run of the next chunk's inputs (+ the two before it)
    art_s xl0 = 0, xl1 = 0;                        //         and that chunk's last two inputs (hr == 0 lanes)
    auto fetch = [&] (int k) {
        if (wave != 2 || hr >= runs || k >= nchunks) return;
        const int L = chunk_len (k);
        const __amdgpu_buffer_rsrc_t rs = ff_rsrc (buf + (size_t) chunk_start (k) * stride + c0, ((size_t) L * stride - c0) * esz);
        // the two frames before the chunk (f0 == 0: patched from xtail later) are forced out of range by a select — a
        // negative offset is not left to wrap, the hardware's range check does not wrap register + immediate to 32 bits
#pragma unroll
        for (int j = 0; j < FF_RUN + 2; ++j) {
            if (j >= run + 2) break;
            const int f = f0 + j - 2;
            xr [j] = ff_load (rs, f >= 0 ? (f * stride + hc) * esz : (int) 0xfffffff0u, art_s ());
        }
        xl0 = ff_load (rs, ((L - 1) * stride + hc) * esz, art_s ());
        xl1 = ff_load (rs, L >= 2 ? ((L - 2) * stride + hc) * esz : (int) 0xfffffff0u, art_s ());
    };
    fetch (0);

    for (int it = -1; it < nchunks + store_lag; ++it) {
        // Wave 2 only ever has loads in flight and wave 3 only stores, so neither waits for the other's memory
        // latency; the same (channel, frame) always belongs to the same lane, so wave 3 may store a buffer and
        // refill it without a barrier in between.
        if (wave == 2) {
            if (hr < runs) {
                if (it + 1 < nchunks) {                    // section 1's feed-forward part of chunk it+1, from the
                    const int k = it + 1, L = chunk_len (k), par = k & 1;      // inputs fetched during the previous step
                    art_s *ud = A1 + (k % 3) * FF_CAP + hc * pitch + f0, *pd = B1 + par * FF_CAP + hc * pitch + f0;
                    // the two frames before the chunk may already hold outputs (in-place): they come from xtail
                    if (hr == 0) {
                        xr [0] = xtail [par][hc][1]; xr [1] = xtail [par][hc][0];
                        xtail [par ^ 1][hc][0] = xl0; xtail [par ^ 1][hc][1] = L >= 2 ? xl1 : xr [1];
                    }
                    const art_s a0 = ffc [0][hc][0], a1 = ffc [0][hc][1], a2 = ffc [0][hc][2];
#pragma unroll
                    for (int j = 0; j < FF_RUN; ++j) {     // frames past a short first chunk land in row slack
                        if (j >= run) break;
                        const art_s p0 = xr [j + 2] * a0, p1 = xr [j + 1] * a1;
                        ud [j] = p0 + p1;
                        pd [j] = xr [j] * a2;
                    }
                }
                fetch (it + 2);                            // in flight across the barrier
            }
        }
        else if (wave == 3) {
            if (hr < runs) {
                {   // store the chunk that left the last section
                    const int k = it - store_lag;
                    if (k >= 0) {
                        const int L = chunk_len (k);
                        const art_s *src = (S == 2 ? A2 + (k & 1) * FF_CAP : A1 + (k % 3) * FF_CAP) + hc * pitch + f0;
                        const __amdgpu_buffer_rsrc_t rs = ff_rsrc (buf + (size_t) chunk_start (k) * stride + c0, ((size_t) L * stride - c0) * esz);
                        const int base = (f0 * stride + hc) * esz;
                        art_s v [FF_RUN];
#pragma unroll
                        for (int j = 0; j < FF_RUN; ++j) { if (j >= run) break; v [j] = src [j]; }
#pragma unroll
                        for (int j = 0; j < FF_RUN; ++j) { if (j >= run) break; ff_store (rs, base + j * stride * esz, v [j]); }
                    }
                }
                if (S == 2 && it >= 1 && it <= nchunks) {  // section 2's feed-forward part from section 1's outputs
                    const int k = it - 1;
                    const art_s *row = A1 + (k % 3) * FF_CAP + hc * pitch + f0;
                    art_s *ud = A2 + (k & 1) * FF_CAP + hc * pitch + f0, *pd = B2 + (k & 1) * FF_CAP + hc * pitch + f0;
                    art_s x [FF_RUN + 2];
#pragma unroll
                    for (int j = 0; j < FF_RUN + 2; ++j) {
                        if (j >= run + 2) break;
                        x [j] = row [hr == 0 && j < 2 ? 0 : j - 2];
                    }
                    if (hr == 0) { x [0] = mtail [k % 3][hc][1]; x [1] = mtail [k % 3][hc][0]; }
                    const art_s a0 = ffc [1][hc][0], a1 = ffc [1][hc][1], a2 = ffc [1][hc][2];
#pragma unroll
                    for (int j = 0; j < FF_RUN; ++j) {
                        if (j >= run) break;
                        const art_s p0 = x [j + 2] * a0, p1 = x [j + 1] * a1;
                        ud [j] = p0 + p1;
                        pd [j] = x [j] * a2;
                    }
                }
            }
        }
        else if (wave == 0) {
            if (lane < Cg && it >= 0 && it < nchunks) {
                ff_serial (A1 + (it % 3) * FF_CAP + lane * pitch, B1 + (it & 1) * FF_CAP + lane * pitch, chunk_len (it), b1, b2, y1, y2);
                if (S == 2) { mtail [(it + 1) % 3][lane][0] = y1; mtail [(it + 1) % 3][lane][1] = y2; }
            }
        }
        else if (S == 2) {
            const int k = it - 2;
            if (lane < Cg && k >= 0 && k < nchunks)
                ff_serial (A2 + (k & 1) * FF_CAP + lane * pitch, B2 + (k & 1) * FF_CAP + lane * pitch, chunk_len (k), b1, b2, y1, y2);
        }
        // LDS-only barrier: wave 2's global loads (consumed next step) and wave 3's stores stay in flight across it;
        // no thread reads global memory another thread of this launch wrote
        asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __syncthreads ();

    // ---- state write-back: the four most recent inputs / outputs of each section ----------------------
    if (tid < Cg) {
        // the last chunk (>= 4 frames whenever frames >= 4) is still in LDS as section outputs
        const int kl = nchunks - 1, L = chunk_len (kl);
        const art_s *r1 = A1 + (kl % 3) * FF_CAP + tid * pitch + L, *r2 = A2 + (kl & 1) * FF_CAP + tid * pitch + L;
        Biquad &f1 = sections [(size_t)(c0 + tid) * S];
        {
            const int i = f1.index + frames;
#pragma unroll
            for (int k = 0; k < 4; ++k) { f1.x [(i - k) & 3] = xt [k]; f1.y [(i - k) & 3] = r1 [-1 - k]; }
            f1.index = i;
        }
        if (S == 2) {
            Biquad &f2 = sections [(size_t)(c0 + tid) * S + 1];
            const int i = f2.index + frames;
#pragma unroll
            for (int k = 0; k < 4; ++k) { f2.x [(i - k) & 3] = r1 [-1 - k]; f2.y [(i - k) & 3] = r2 [-1 - k]; }
            f2.index = i;
        }
    }
}

// error-feedback filter with a compile-time order (per-sample association, reference biquad.c:83-95):
//     acc = in*a0;  for k = ORDER..1:  acc += (x_k*a_k) - (b_k*y_k)
// x_k / y_k for k >= 2 do not depend on the newest output, so those terms are formed ahead of the chain.
template <int ORDER>
__device__ __forceinline__ art_s shaper_step (SectionRegs &r, art_s in)
{
    art_s term [4];
#pragma unroll
    for (int k = 1; k <= 4; ++k)
        if (k <= ORDER) { const art_s fwd = r.x [k - 1] * r.a [k]; const art_s back = r.b [k] * r.y [k - 1]; term [k - 1] = fwd - back; }
    art_s acc = in * r.a [0];
#pragma unroll
    for (int k = 4; k >= 1; --k)
        if (k <= ORDER) acc = acc + term [k - 1];
    push (r, in, acc);
    return acc;
}

// The dither generator (decimator.c:370-382) steps r <- 15 r ^ 1 five times per sample.  15 r keeps the
// parity of r and "^ 1" flips it, so a step is 15 r + 1 on even r and 15 r - 1 on odd r and the parity
// alternates every step: five steps are one of two affine maps (chosen by the parity of the start state),
// the parity alternates every SAMPLE, and two samples are a single fixed affine map.  That gives an
// O(log n) jump-ahead, so a chunk's dither can be produced by all threads at once.
struct Affine { uint32_t a, b; };                  // r -> a r + b  (mod 2^32)
__device__ __forceinline__ Affine compose (Affine second, Affine first) { return { second.a * first.a, second.a * first.b + second.b }; }

__device__ __forceinline__ Affine five_steps (bool even_start)
{
    Affine m = { 1u, 0u };
    bool even = even_start;
#pragma unroll
    for (int i = 0; i < 5; ++i) { m = compose (Affine { 15u, even ? 1u : 0xffffffffu }, m); even = !even; }
    return m;
}

__device__ __forceinline__ uint32_t jump_pairs (uint32_t g, unsigned int pairs)     // advance by 2*pairs samples
{
    const bool even = (g & 1u) == 0;
    Affine two = compose (five_steps (!even), five_steps (even));                   // parity returns after two samples
    Affine acc = { 1u, 0u };
    while (pairs) {
        if (pairs & 1u) acc = compose (two, acc);
        two = compose (two, two);
        pairs >>= 1;
    }
    return acc.a * g + acc.b;
}

constexpr int DEC_CHUNK = ART_WIDE ? 2048 : 4096;                    // samples per chunk (16 KiB each for data and dither)
// floor (d + 0.5) as the reference evaluates it (decimator.c:262).  4-byte samples: the reference widens d to double
// first, so the sum is exact; v_cvt_rpi_i32_f32 ("round to nearest, ties towards +infinity") is exactly that function,
// computed without an intermediate rounding — verified on the hardware against floor ((double) d + 0.5) over 5M random
// and adversarial inputs (tools/micro/rpi_check.hip) — and puts two dependent operations on the serial path instead of
// four (floorf, subtract, compare, select).  Beyond +-2^31 it saturates where the reference's conversion is undefined.
// 8-byte samples: the reference's own sum rounds (e.g. d = 0.5 - 2^-54 gives 1), so it is evaluated literally.
__device__ __forceinline__ art_s round_half_up (art_s d)
{
#if ART_WIDE
    return floor (d + 0.5);
#else
    int q;
    asm ("v_cvt_rpi_i32_f32 %0, %1" : "=v" (q) : "v" (d));
    return (float) q;
#endif
}

constexpr int DEC_SEG = 32;                        // consecutive samples of one channel per dither task (even)

template <int ORDER, bool DITHER>                  // ORDER 0 = no noise shaping
__global__ __launch_bounds__ (ST_THREADS)
void decimate_lds_kernel (ArtDecArgs a, const art_s *in, int frames, unsigned char *out, int cpw)     // cpw: channels per workgroup (<= 64)
{
    __shared__ __attribute__ ((aligned (16))) art_s tile [DEC_CHUNK];          // input, then the rounded code values
    __shared__ __attribute__ ((aligned (16))) art_s dth [DITHER ? DEC_CHUNK : 1];
    __shared__ uint32_t s_gen [64], s_next [64];   // generator state at the start of this / the next chunk
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * cpw, Cg = min (cpw, a.C - c0);
    const int chunk_frames = (DEC_CHUNK / Cg) & ~1;                            // even: chunk boundaries keep generator parity

    art_s fb = 0.0f; SectionRegs sh; unsigned long long clips = 0;
    if (tid < Cg) {
        fb = a.feedback [c0 + tid];
        if (DITHER) s_gen [tid] = a.gens [c0 + tid];
        if (ORDER) load_section (sh, a.shapers [c0 + tid]);
    }
    const int nbytes = a.bytes, width = (a.bits + 7) / 8, pad = nbytes - width;
    const int hi = (1 << (a.bits - 1)) - 1, lo = ~hi;
    const int shift = (24 - a.bits) % 8;
    const uint32_t bias = a.bits <= 8 ? 128u : 0u;
    const int dtype = a.dither_type;
    const art_s scale = a.scale;
    __syncthreads ();

    for (int f0 = 0; f0 < frames; f0 += chunk_frames) {
        const int nf = min (chunk_frames, frames - f0);

        // ---- phase A (all threads): load the chunk; produce its dither by jump-ahead
        for (int e = tid; e < nf * Cg; e += ST_THREADS) {
            const int f = e / Cg, c = e - f * Cg;
            tile [e] = in [(size_t)(f0 + f) * a.C + c0 + c];
        }
        if (DITHER) {
            const int segs_per_ch = (nf + DEC_SEG - 1) / DEC_SEG;
            for (int task = tid; task < segs_per_ch * Cg; task += ST_THREADS) {
                const int c = task % Cg, k = task / Cg, n0 = k * DEC_SEG;
                uint32_t g = jump_pairs (s_gen [c], (unsigned int)(n0 / 2));
                const int cnt = min (DEC_SEG, nf - n0);
                for (int i = 0; i < cnt; ++i) {
                    const uint32_t start = g;
                    uint32_t r = lcg (lcg (start));
                    const uint32_t first = dtype < 0 ? ~start : dtype > 0 ? start : ~r;
                    r = lcg (lcg (lcg (r)));
                    g = r;
                    // ((first>>1)+(r>>1))/2^31 - 1.0, converted to the sample type, is exactly this (power-of-two scale)
                    const uint32_t u = (first >> 1) + (r >> 1);
                    dth [(n0 + i) * Cg + c] = (art_s)(int)(u ^ 0x80000000u) * (art_s) 4.656612873077392578125e-10;
                }
                if (n0 + cnt == nf) s_next [c] = g;          // the channel's last task publishes the next chunk's state
            }
        }
        __syncthreads ();

        // ---- phase B: rounding (round_half_up).  Without noise shaping the feedback term never changes, so there is
        // no recurrence and every thread rounds its own samples; with shaping one lane per channel walks time.
        if (!ORDER) {
            for (int e = tid; e < nf * Cg; e += ST_THREADS) {
                const int c = e % Cg;
                const art_s code = tile [e] * scale - a.feedback [c0 + c];
                const art_s dithered = code + (DITHER ? dth [e] : 0.0f);
                tile [e] = round_half_up (dithered);
            }
        }
        else if (tid < Cg) {
            if (DITHER) s_gen [tid] = s_next [tid];          // phase A of the next chunk is two barriers away
            auto one = [&] (art_s smp, art_s dither) -> art_s {
                const art_s scaled = smp * scale;
                const art_s code = scaled - fb;
                const art_s dithered = code + dither;
                const art_s qf = round_half_up (dithered);
                const art_s err = qf - code;
                fb = shaper_step<ORDER> (sh, err);
                return qf;
            };
            constexpr int UB = 8;
            int f = 0;
            for (; f + UB <= nf; f += UB) {
                art_s x [UB], d [UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) { x [u] = tile [(f + u) * Cg + tid]; d [u] = DITHER ? dth [(f + u) * Cg + tid] : 0.0f; }
#pragma unroll
                for (int u = 0; u < UB; ++u) x [u] = one (x [u], d [u]);
#pragma unroll
                for (int u = 0; u < UB; ++u) tile [(f + u) * Cg + tid] = x [u];
            }
            for (; f < nf; ++f) tile [f * Cg + tid] = one (tile [f * Cg + tid], DITHER ? dth [f * Cg + tid] : 0.0f);
        }
        if (!ORDER && DITHER && tid < Cg) s_gen [tid] = s_next [tid];   // (s_next was published before the barrier above)
        __syncthreads ();

        // ---- phase C (all threads): clip, pack little-endian, store
        for (int e = tid; e < nf * Cg; e += ST_THREADS) {
            const int f = e / Cg, c = e - f * Cg;
            int q = (int) tile [e];
            if (q > hi) { q = hi; clips++; }
            else if (q < lo) { q = lo; clips++; }
            const uint32_t v = ((uint32_t) q << shift) + bias;
            unsigned char *o = out + ((size_t)(f0 + f) * a.C + c0 + c) * nbytes;
            for (int j = 0; j < pad; ++j) *o++ = 0;
            *o++ = (unsigned char) v;
            if (width > 1) { *o++ = (unsigned char)(v >> 8); if (width > 2) *o++ = (unsigned char)(v >> 16); }
        }
        __syncthreads ();
    }

    if (tid < Cg) {
        a.feedback [c0 + tid] = fb;
        if (DITHER) a.gens [c0 + tid] = s_gen [tid];
        if (ORDER) store_section (a.shapers [c0 + tid], sh, frames, true);
    }
    if (clips) atomicAdd (a.clipped, clips);
}


// Noise-shaped decimator as a three-stage pipeline over chunks (the serial lane is latency-bound: a 10-deep dependent
// chain per sample — so everything that is not that chain runs beside it, on the other waves):
//     waves 1-3   phase A of chunk it+1 (load, dither by jump-ahead)   |   phase C of chunk it-1 (clip, pack, store)
//     wave 0      phase B of chunk it: one lane per channel through the error-feedback recurrence
// one LDS-only barrier per step; three sample tiles (by chunk % 3), two dither tiles and two generator-state rows.
template <int ORDER, bool DITHER>                  // ORDER >= 1
__global__ __launch_bounds__ (ST_THREADS)
void decimate_pipe_kernel (ArtDecArgs a, const art_s *in, int frames, unsigned char *out, int cpw)
{
    extern __shared__ __attribute__ ((aligned (16))) unsigned char dec_lds [];
    art_s *const tiles = (art_s *) dec_lds;                               // [3][DEC_CHUNK]
    art_s *const dths = tiles + 3 * DEC_CHUNK;                            // [2][DEC_CHUNK]
    __shared__ uint32_t s_gen [2][64];             // generator state at the start of a chunk, by chunk parity
    const int tid = threadIdx.x, wave = tid >> 6;
    const int c0 = blockIdx.x * cpw, Cg = min (cpw, a.C - c0);
    // LDS tiles are CHANNEL-major, [channel][frame] with a pitch of chunk_frames + 4 (the serial lane then moves four frames per
    // ds_read_b128 / ds_write_b128 instead of one per instruction — a lone wave's budget is instructions; the +4 keeps the
    // channels' rows on different banks)
    const int chunk_frames = ((DEC_CHUNK / Cg) - 4) & ~3;                 // multiple of 4 (vector alignment; even: chunk boundaries keep generator parity)
    const int pitch = chunk_frames + 4;
    const int nchunks = (frames + chunk_frames - 1) / chunk_frames;

    art_s fb = 0.0f; SectionRegs sh; unsigned long long clips = 0;
    if (tid < Cg) {
        fb = a.feedback [c0 + tid];
        if (DITHER) s_gen [0][tid] = a.gens [c0 + tid];
        load_section (sh, a.shapers [c0 + tid]);
    }
    const int nbytes = a.bytes, width = (a.bits + 7) / 8, pad = nbytes - width;
    const int hi = (1 << (a.bits - 1)) - 1, lo = ~hi;
    const int shift = (24 - a.bits) % 8;
    const uint32_t bias = a.bits <= 8 ? 128u : 0u;
    const int dtype = a.dither_type;
    const art_s scale = a.scale;
    constexpr int HELPERS = ST_THREADS - 64;
    __syncthreads ();

    for (int it = -1; it <= nchunks; ++it) {
        if (wave >= 1) {
            const int ht = tid - 64;
            if (it + 1 < nchunks) {                // ---- phase A of chunk it+1
                const int k = it + 1, f0 = k * chunk_frames, nf = min (chunk_frames, frames - f0);
                art_s *tile = tiles + (k % 3) * DEC_CHUNK, *dth = dths + (k & 1) * DEC_CHUNK;
                for (int e = ht; e < nf * Cg; e += HELPERS) {
                    const int f = e / Cg, c = e - f * Cg;
                    tile [c * pitch + f] = in [(size_t)(f0 + f) * a.C + c0 + c] * scale;     // (the serial wave's first operation, done here: same product)
                }
                if (DITHER) {
                    const int segs_per_ch = (nf + DEC_SEG - 1) / DEC_SEG;
                    for (int task = ht; task < segs_per_ch * Cg; task += HELPERS) {
                        const int c = task % Cg, sgm = task / Cg, n0 = sgm * DEC_SEG;
                        uint32_t g = jump_pairs (s_gen [k & 1][c], (unsigned int)(n0 / 2));
                        const int cnt = min (DEC_SEG, nf - n0);
                        for (int i = 0; i < cnt; ++i) {
                            const uint32_t start = g;
                            uint32_t r = lcg (lcg (start));
                            const uint32_t first = dtype < 0 ? ~start : dtype > 0 ? start : ~r;
                            r = lcg (lcg (lcg (r)));
                            g = r;
                            const uint32_t u = (first >> 1) + (r >> 1);
                            dth [c * pitch + n0 + i] = (art_s)(int)(u ^ 0x80000000u) * (art_s) 4.656612873077392578125e-10;
                        }
                        if (n0 + cnt == nf) s_gen [(k & 1) ^ 1][c] = g;      // start state of chunk k+1
                    }
                }
            }
            if (it >= 1) {                         // ---- phase C of chunk it-1
                const int k = it - 1, f0 = k * chunk_frames, nf = min (chunk_frames, frames - f0);
                const art_s *tile = tiles + (k % 3) * DEC_CHUNK;
                for (int e = ht; e < nf * Cg; e += HELPERS) {
                    const int f = e / Cg, c = e - f * Cg;
                    int q = (int) tile [c * pitch + f];
                    if (q > hi) { q = hi; clips++; }
                    else if (q < lo) { q = lo; clips++; }
                    const uint32_t v = ((uint32_t) q << shift) + bias;
                    unsigned char *o = out + ((size_t)(f0 + f) * a.C + c0 + c) * nbytes;
                    for (int j = 0; j < pad; ++j) *o++ = 0;
                    *o++ = (unsigned char) v;
                    if (width > 1) { *o++ = (unsigned char)(v >> 8); if (width > 2) *o++ = (unsigned char)(v >> 16); }
                }
            }
        }
        else if (tid < Cg && it >= 0 && it < nchunks) {      // ---- phase B of chunk it
            const int f0 = it * chunk_frames, nf = min (chunk_frames, frames - f0);
            art_s *tile = tiles + (it % 3) * DEC_CHUNK;
            const art_s *dth = dths + (it & 1) * DEC_CHUNK;
            auto one = [&] (art_s smp, art_s dither) -> art_s {
                const art_s scaled = smp;                       // already times `scale` (phase A)
                const art_s code = scaled - fb;
                const art_s dithered = code + dither;
                const art_s qf = round_half_up (dithered);
                const art_s err = qf - code;
                fb = shaper_step<ORDER> (sh, err);
                return qf;
            };
            typedef art_s vec4 __attribute__ ((ext_vector_type (4)));
            art_s *mine = tile + tid * pitch;
            const art_s *my_dither = dth + tid * pitch;
            int f = 0;
            for (; f + 8 <= nf; f += 8) {
                vec4 xa = *reinterpret_cast<const vec4 *> (mine + f), xb = *reinterpret_cast<const vec4 *> (mine + f + 4), da, db;
                if (DITHER) { da = *reinterpret_cast<const vec4 *> (my_dither + f); db = *reinterpret_cast<const vec4 *> (my_dither + f + 4); }
                else { da = (art_s) 0; db = (art_s) 0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) xa [u] = one (xa [u], da [u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) xb [u] = one (xb [u], db [u]);
                *reinterpret_cast<vec4 *> (mine + f) = xa; *reinterpret_cast<vec4 *> (mine + f + 4) = xb;
            }
            for (; f < nf; ++f) mine [f] = one (mine [f], DITHER ? my_dither [f] : (art_s) 0);
        }
        // LDS-only barrier: the helpers' stores (and loads already consumed) stay in flight; nobody reads global memory
        // that this launch writes
        asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __syncthreads ();

    if (tid < Cg) {
        a.feedback [c0 + tid] = fb;
        if (DITHER) a.gens [c0 + tid] = s_gen [nchunks & 1][tid];
        store_section (a.shapers [c0 + tid], sh, frames, true);
    }
    if (clips) atomicAdd (a.clipped, clips);
}

// No noise shaping => no recurrence at all: the time axis is cut into segments of DEC_SEG frames, each thread
// jumps its channel's dither generator to its segment and converts it.  Adjacent threads are adjacent
// channels of the same frames.  The generator state after the call is written to a second array (the first
// is still being read by other threads); the host swaps them.
template <bool DITHER>
__global__ __launch_bounds__ (256)
void decimate_parallel_kernel (ArtDecArgs a, const art_s *in, int frames, unsigned char *out, uint32_t *gens_out)
{
    const long task = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(task % a.C);
    const long seg = task / a.C;
    const long n0 = seg * DEC_SEG;
    if (n0 >= frames) return;
    const int cnt = (int) min ((long) DEC_SEG, frames - n0);

    const int nbytes = a.bytes, width = (a.bits + 7) / 8, pad = nbytes - width;
    const int hi = (1 << (a.bits - 1)) - 1, lo = ~hi;
    const int shift = (24 - a.bits) % 8;
    const uint32_t bias = a.bits <= 8 ? 128u : 0u;
    const art_s fb = a.feedback [c];                      // constant without shaping (decimator.c:264-265)
    uint32_t g = DITHER ? jump_pairs (a.gens [c], (unsigned int)(n0 / 2)) : 0u;
    unsigned int clips = 0;

    for (int i = 0; i < cnt; ++i) {
        art_s dither = 0.0f;
        if (DITHER) {
            const uint32_t start = g;
            uint32_t r = lcg (lcg (start));
            const uint32_t first = a.dither_type < 0 ? ~start : a.dither_type > 0 ? start : ~r;
            r = lcg (lcg (lcg (r)));
            g = r;
            const uint32_t u = (first >> 1) + (r >> 1);
            dither = (art_s)(int)(u ^ 0x80000000u) * (art_s) 4.656612873077392578125e-10;
        }
        const size_t e = (size_t)(n0 + i) * a.C + c;
        const art_s scaled = in [e] * a.scale;
        const art_s code = scaled - fb;
        const art_s dithered = code + dither;
        int q = (int) round_half_up (dithered);
        if (q > hi) { q = hi; clips++; }
        else if (q < lo) { q = lo; clips++; }
        const uint32_t v = ((uint32_t) q << shift) + bias;
        unsigned char *o = out + e * nbytes;
        for (int j = 0; j < pad; ++j) *o++ = 0;
        *o++ = (unsigned char) v;
        if (width > 1) { *o++ = (unsigned char)(v >> 8); if (width > 2) *o++ = (unsigned char)(v >> 16); }
    }
    if (DITHER && n0 + cnt == frames) gens_out [c] = g;
    if (clips) atomicAdd (a.clipped, (unsigned long long) clips);
}

__global__ void ingest_kernel (const unsigned char *in, art_s g, int bits, int bytes, int stride, art_s *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int width = (bits + 7) / 8;
    const unsigned char *p = in + (size_t) i * stride * bytes + (bytes - width);
    art_s v;
    if (bits <= 8) v = (art_s)((int) p [0] - 128) * g;
    else if (bits <= 16) v = (art_s)(int)(short)(p [0] | (p [1] << 8)) * g;
    else v = (art_s)(int)((uint32_t) p [0] | ((uint32_t) p [1] << 8) | ((uint32_t)(int)(signed char) p [2] << 16)) * g;
    out [i] = v;
}

} // namespace

// The serial stages are latency-bound per lane, so a workgroup gains nothing from more channels — but its LDS chunk (and
// with it the work per barrier) shrinks in proportion.  Many-channel calls are therefore spread over MORE workgroups of
// 8 channels (one per CU and beyond) rather than packed 64 to a workgroup; only beyond 512 workgroups do the groups grow.
static int channels_per_workgroup (int C)
{
    int cpw = 8;
    while (cpw < 64 && (C + cpw - 1) / cpw > 512) cpw += 8;
    return cpw;
}

extern "C" {

// The time-parallel bit-exact cascade (biquad_spec_kernel): `d_in` -> `d_out` (distinct buffers), frames x C with the given
// strides; W = warm-up frames per section (host: decay of the recursive part), L = chunk length.  d_states: scratch of
// arthip_biquad_spec_scratch (C, S, frames, L) bytes.  d_repairs: device counter (chunks that had to be recomputed).
size_t arthip_biquad_spec_scratch (int C, int S, int frames, int L)
{
    const size_t K = (size_t)((frames + L - 1) / L);
    return (size_t) C * K * S * sizeof (SpecState) * 2 + (((size_t) C * K + 255) & ~(size_t) 255);
}

// d_first_bad: C ints of device memory that hold a value beyond any chunk count between calls (armed once, here)
int arthip_biquad_spec_arm (int *d_first_bad, int C, void *stream)
{
    return hipMemsetAsync (d_first_bad, 0x7f, sizeof (int) * (size_t) C, (hipStream_t) stream) == hipSuccess ? 0 : -1;     // 0x7f7f7f7f
}

int arthip_biquad_spec (Biquad *d_sections, int C, int S, const art_s *d_in, int in_stride, art_s *d_out, int out_stride, int frames,
                        int L, int W, void *d_states, int *d_first_bad, unsigned int *d_repairs, void *stream)
{
    if (frames <= 0) return 0;
    if (S < 1 || S > MAX_CHAIN || L < 1) return -1;
    const int K = (frames + L - 1) / L;
    SpecState *starts = (SpecState *) d_states, *ends = starts + (size_t) C * K * S;
    unsigned char *bad = (unsigned char *)(ends + (size_t) C * K * S);
    const long tasks = (long) C * K;
    const dim3 grid ((unsigned int)((tasks + 255) / 256)), block (256);
    hipStream_t st = (hipStream_t) stream;
#define SPEC_GO(SS) do { \
        hipLaunchKernelGGL (biquad_spec_kernel<SS>, grid, block, 0, st, (const Biquad *) d_sections, C, K, L, W, d_in, in_stride, d_out, out_stride, frames, starts, ends); \
        hipLaunchKernelGGL (biquad_check_kernel<SS>, grid, block, 0, st, C, K, (const SpecState *) starts, (const SpecState *) ends, bad, d_first_bad); \
        hipLaunchKernelGGL (biquad_commit_kernel<SS>, dim3 ((C + 63) / 64), dim3 (64), 0, st, d_sections, C, K, L, d_in, in_stride, d_out, out_stride, frames, \
                            (const SpecState *) starts, ends, (const unsigned char *) bad, d_first_bad, d_repairs); } while (0)
    switch (S) { case 1: SPEC_GO (1); break; case 2: SPEC_GO (2); break; case 3: SPEC_GO (3); break; default: SPEC_GO (4); }
#undef SPEC_GO
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

int arthip_biquad_order2 (Biquad *d_sections, int C, int S, art_s *d_buf, int frames, int stride, void *stream)
{
    if (frames <= 0) return 0;
    if (S == 1 || S == 2) {
        const size_t lds = (size_t) 9 * FF_CAP * sizeof (art_s) + 256;           // 108 KiB of the CU's 160 (+ look-ahead slack)
        static bool once = false;
        if (!once) {
            (void) hipFuncSetAttribute ((const void *) biquad_order2_ff_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            (void) hipFuncSetAttribute ((const void *) biquad_order2_ff_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            once = true;
        }
        const int cpw = channels_per_workgroup (C);
        const dim3 grid ((C + cpw - 1) / cpw);
        if (S == 1) hipLaunchKernelGGL (biquad_order2_ff_kernel<1>, grid, dim3 (ST_THREADS), lds, (hipStream_t) stream, d_sections, C, stride, d_buf, frames, cpw);
        else hipLaunchKernelGGL (biquad_order2_ff_kernel<2>, grid, dim3 (ST_THREADS), lds, (hipStream_t) stream, d_sections, C, stride, d_buf, frames, cpw);
    }
    else return -1;
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

int arthip_biquad_chain (Biquad *d_sections, int C, int S, art_s *d_buf, int frames, int stride, void *stream)
{
    if (S < 1 || S > MAX_CHAIN || frames <= 0) return S < 1 || S > MAX_CHAIN ? -1 : 0;
    const int sample_form = stride < 0;                  // negative stride selects the per-sample association
    if (sample_form) stride = -stride;
    if (!sample_form && stride == C && frames >= 64) {   // interleaved frames: LDS-staged form
        hipLaunchKernelGGL (biquad_chain_lds_kernel, dim3 ((C + 63) / 64), dim3 (ST_THREADS), 0, (hipStream_t) stream, d_sections, C, S, d_buf, frames);
        return hipGetLastError () == hipSuccess ? 0 : -1;
    }
    hipLaunchKernelGGL (biquad_chain_kernel, dim3 ((C + 63) / 64), dim3 (64), 0, (hipStream_t) stream, d_sections, C, S, d_buf, frames, stride, sample_form);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

static int decimate_launch (const ArtDecArgs *a, const art_s *d_in, long in_pitch, int frames, unsigned char *d_out, long out_pitch, void *stream)
{
    if (frames <= 0) return 0;
    hipLaunchKernelGGL (decimate_kernel, dim3 ((a->C + 63) / 64), dim3 (64), 0, (hipStream_t) stream, *a, d_in, in_pitch, frames, d_out, out_pitch);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

int arthip_decimate (const ArtDecArgs *a, const art_s *d_in, int frames, unsigned char *d_out, void *stream)
{
    if (frames >= 64 && !a->shaping_on && (!a->dither_on || a->gens_next) && (DEC_SEG % 2) == 0) {
        const long tasks = (long) a->C * ((frames + DEC_SEG - 1) / DEC_SEG);
        const dim3 grid ((unsigned int)((tasks + 255) / 256)), block (256);
        if (a->dither_on) hipLaunchKernelGGL (decimate_parallel_kernel<true>, grid, block, 0, (hipStream_t) stream, *a, d_in, frames, d_out, a->gens_next);
        else hipLaunchKernelGGL (decimate_parallel_kernel<false>, grid, block, 0, (hipStream_t) stream, *a, d_in, frames, d_out, a->gens_next);
        return hipGetLastError () == hipSuccess ? 1 : -1;      // 1: generator state now lives in gens_next
    }
    if (frames >= 64) {
        const int cpw = channels_per_workgroup (a->C);
        const dim3 grid ((a->C + cpw - 1) / cpw), block (ST_THREADS);
        hipStream_t st = (hipStream_t) stream;
        const int order = a->shaping_on ? a->shaping_order : 0;
#define DEC_GO(O) do { if (a->dither_on) hipLaunchKernelGGL ((decimate_lds_kernel<O, true>), grid, block, 0, st, *a, d_in, frames, d_out, cpw); \
                       else hipLaunchKernelGGL ((decimate_lds_kernel<O, false>), grid, block, 0, st, *a, d_in, frames, d_out, cpw); } while (0)
#define DEC_PIPE(O) do { auto kd = decimate_pipe_kernel<O, true>; auto kn = decimate_pipe_kernel<O, false>; \
                         static bool once = false; \
                         if (!once) { (void) hipFuncSetAttribute ((const void *) kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int) pipe_lds); \
                                      (void) hipFuncSetAttribute ((const void *) kn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) pipe_lds); once = true; } \
                         if (a->dither_on) hipLaunchKernelGGL (kd, grid, block, pipe_lds, st, *a, d_in, frames, d_out, cpw); \
                         else hipLaunchKernelGGL (kn, grid, block, pipe_lds, st, *a, d_in, frames, d_out, cpw); } while (0)
        const size_t pipe_lds = (size_t) 5 * DEC_CHUNK * sizeof (art_s);
        // with more workgroups than CUs the chip is busy anyway and the smaller LDS footprint of the unpipelined form
        // (more workgroups per CU) wins: 4,096 channels 49 vs 36 Gsamples/s
        if (order >= 1 && grid.x <= 256) {
            switch (order) { case 1: DEC_PIPE (1); break; case 2: DEC_PIPE (2); break; case 3: DEC_PIPE (3); break; default: DEC_PIPE (4); }
            return hipGetLastError () == hipSuccess ? 0 : -1;
        }
        switch (order) { case 0: DEC_GO (0); break; case 1: DEC_GO (1); break; case 2: DEC_GO (2); break; case 3: DEC_GO (3); break; default: DEC_GO (4); }
#undef DEC_PIPE
#undef DEC_GO
        return hipGetLastError () == hipSuccess ? 0 : -1;
    }
    return decimate_launch (a, d_in, 0, frames, d_out, 0, stream);
}

int arthip_decimate_planar (const ArtDecArgs *a, const art_s *d_in, long in_pitch, int frames, unsigned char *d_out, long out_pitch, void *stream)
{
    return decimate_launch (a, d_in, in_pitch, frames, d_out, out_pitch, stream);
}

int arthip_ingest (const unsigned char *d_in, art_s g, int bits, int bytes, int stride, art_s *d_out, int n, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL (ingest_kernel, dim3 ((n + 255) / 256), dim3 (256), 0, (hipStream_t) stream, d_in, g, bits, bytes, stride, d_out, n);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

}
