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
#include "art_internal.h"

namespace {

struct SectionRegs {
    float a [5], b [5];
    float x [4], y [4];       // x[0] = most recent input, x[1] the one before, ...
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

__device__ __forceinline__ void push (SectionRegs &r, float in, float out)
{
    r.x [3] = r.x [2]; r.x [2] = r.x [1]; r.x [1] = r.x [0]; r.x [0] = in;
    r.y [3] = r.y [2]; r.y [2] = r.y [1]; r.y [1] = r.y [0]; r.y [0] = out;
}

// buffer form: in*a0, then for k = 1..order: + x_k*a_k, - b_k*y_k, strictly left to right
__device__ __forceinline__ float step_buffer_order (SectionRegs &r, float in)
{
    float acc = in * r.a [0];
#pragma unroll
    for (int k = 1; k <= 4; ++k)
        if (k <= r.order) {
            float fwd = r.x [k - 1] * r.a [k];
            acc = acc + fwd;
            float back = r.b [k] * r.y [k - 1];
            acc = acc - back;
        }
    push (r, in, acc);
    return acc;
}

// per-sample form: in*a0, then for k = order..1: += (x_k*a_k - b_k*y_k)
__device__ __forceinline__ float step_sample_order (SectionRegs &r, float in)
{
    float acc = in * r.a [0];
#pragma unroll
    for (int k = 4; k >= 1; --k)
        if (k <= r.order) {
            float fwd = r.x [k - 1] * r.a [k];
            float back = r.b [k] * r.y [k - 1];
            float term = fwd - back;
            acc = acc + term;
        }
    push (r, in, acc);
    return acc;
}

constexpr int MAX_CHAIN = 4;

__global__ void biquad_chain_kernel (Biquad *sections, int C, int S, float *buf, int frames, int stride, int sample_form)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;

    SectionRegs r [MAX_CHAIN];
#pragma unroll
    for (int s = 0; s < MAX_CHAIN; ++s)
        if (s < S) load_section (r [s], sections [(size_t) c * S + s]);

    float *p = buf + c;
    for (int i = 0; i < frames; ++i, p += stride) {
        float v = *p;
#pragma unroll
        for (int s = 0; s < MAX_CHAIN; ++s)
            if (s < S) v = sample_form ? step_sample_order (r [s], v) : step_buffer_order (r [s], v);
        *p = v;
    }

#pragma unroll
    for (int s = 0; s < MAX_CHAIN; ++s)
        if (s < S) store_section (sections [(size_t) c * S + s], r [s], frames, sample_form != 0);
}

__device__ __forceinline__ uint32_t lcg (uint32_t r) { return ((r << 4) - r) ^ 1u; }

__global__ void decimate_kernel (ArtDecArgs a, const float *in, long in_pitch, int frames, unsigned char *out, long out_pitch)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.C) return;

    float fb = a.feedback [c];
    uint32_t gen = a.dither_on ? a.gens [c] : 0u;
    SectionRegs sh;
    if (a.shaping_on) load_section (sh, a.shapers [c]);

    const int pad = a.bytes - ((a.bits + 7) / 8);
    const int hi = (1 << (a.bits - 1)) - 1, lo = ~hi;
    const int shift = (24 - a.bits) % 8;
    const uint32_t bias = a.bits <= 8 ? 128u : 0u;
    unsigned long long clips = 0;

    for (int i = 0; i < frames; ++i) {
        const float s = in_pitch ? in [(size_t) c * in_pitch + i] : in [(size_t) i * a.C + c];
        float dither = 0.0f;

        if (a.dither_on) {
            const uint32_t start = gen;
            uint32_t r = lcg (lcg (start));
            const uint32_t first = a.dither_type < 0 ? ~start : a.dither_type > 0 ? start : ~r;
            r = lcg (lcg (lcg (r)));
            gen = r;
            const double tri = ((double)((first >> 1) + (r >> 1)) / 2147483648.0) - 1.0;
            dither = (float) tri;
        }

        const float scaled = s * a.scale;
        const float code = scaled - fb;
        const float dithered = code + dither;
        int q = (int) floor ((double) dithered + 0.5);

        if (a.shaping_on) {
            const float err = (float) q - code;
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

__global__ void ingest_kernel (const unsigned char *in, float g, int bits, int bytes, int stride, float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int width = (bits + 7) / 8;
    const unsigned char *p = in + (size_t) i * stride * bytes + (bytes - width);
    float v;
    if (bits <= 8) v = (float)((int) p [0] - 128) * g;
    else if (bits <= 16) v = (float)(int)(short)(p [0] | (p [1] << 8)) * g;
    else v = (float)(int)((uint32_t) p [0] | ((uint32_t) p [1] << 8) | ((uint32_t)(int)(signed char) p [2] << 16)) * g;
    out [i] = v;
}

} // namespace

extern "C" {

int arthip_biquad_chain (Biquad *d_sections, int C, int S, float *d_buf, int frames, int stride, void *stream)
{
    if (S < 1 || S > MAX_CHAIN || frames <= 0) return S < 1 || S > MAX_CHAIN ? -1 : 0;
    const int sample_form = stride < 0;                  // negative stride selects the per-sample association
    if (sample_form) stride = -stride;
    hipLaunchKernelGGL (biquad_chain_kernel, dim3 ((C + 63) / 64), dim3 (64), 0, (hipStream_t) stream, d_sections, C, S, d_buf, frames, stride, sample_form);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

static int decimate_launch (const ArtDecArgs *a, const float *d_in, long in_pitch, int frames, unsigned char *d_out, long out_pitch, void *stream)
{
    if (frames <= 0) return 0;
    hipLaunchKernelGGL (decimate_kernel, dim3 ((a->C + 63) / 64), dim3 (64), 0, (hipStream_t) stream, *a, d_in, in_pitch, frames, d_out, out_pitch);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

int arthip_decimate (const ArtDecArgs *a, const float *d_in, int frames, unsigned char *d_out, void *stream)
{
    return decimate_launch (a, d_in, 0, frames, d_out, 0, stream);
}

int arthip_decimate_planar (const ArtDecArgs *a, const float *d_in, long in_pitch, int frames, unsigned char *d_out, long out_pitch, void *stream)
{
    return decimate_launch (a, d_in, in_pitch, frames, d_out, out_pitch, stream);
}

int arthip_ingest (const unsigned char *d_in, float g, int bits, int bytes, int stride, float *d_out, int n, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL (ingest_kernel, dim3 ((n + 255) / 256), dim3 (256), 0, (hipStream_t) stream, d_in, g, bits, bytes, stride, d_out, n);
    return hipGetLastError () == hipSuccess ? 0 : -1;
}

}
