// peak_probe.hip — what one pass over a call's input costs (MI355X): per-channel |x| maximum over 33.5 MB of interleaved float32
// with different launch shapes, cache states (a 70 MB write in between, as the digit-plane pass and the main kernel leave it)
// and a pure read for the floor.   hipcc --offload-arch=gfx950 -O3 -o peak_probe peak_probe.hip && ./peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));

template <int UNROLL>
__global__ void peak_k (const u32x4 *sv, size_t nv, int C, unsigned int *peak)
{
    __shared__ unsigned int s_peak [32];
    const int tid = threadIdx.x;
    if (tid < 32) s_peak [tid] = 0u;
    __syncthreads ();
    const size_t stride = (size_t) gridDim.x * blockDim.x, v0 = (size_t) blockIdx.x * blockDim.x + tid;
    unsigned int m [4] = { 0u, 0u, 0u, 0u };
    auto take = [&] (const u32x4 &v) { m [0] = max (m [0], v.x & 0x7fffffffu); m [1] = max (m [1], v.y & 0x7fffffffu); m [2] = max (m [2], v.z & 0x7fffffffu); m [3] = max (m [3], v.w & 0x7fffffffu); };
    size_t v = v0;
    for (; v + (UNROLL - 1) * stride < nv; v += UNROLL * stride) {
        u32x4 x [UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x [u] = __builtin_nontemporal_load (&sv [v + u * stride]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) take (x [u]);
    }
    for (; v < nv; v += stride) take (sv [v]);
    const int lanes_per_set = C >= 4 ? C / 4 : 1;
    for (int off = 32; off >= lanes_per_set; off >>= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) m [e] = max (m [e], (unsigned int) __shfl_xor ((int) m [e], off));
    if ((tid & 63) < lanes_per_set)
#pragma unroll
        for (int e = 0; e < 4; ++e) if (m [e]) atomicMax (&s_peak [(int)((v0 * 4 + e) % C)], m [e]);
    __syncthreads ();
    if (tid < C && s_peak [tid] > __builtin_nontemporal_load (&peak [tid])) atomicMax (&peak [tid], s_peak [tid]);
}
__global__ void fill_k (u32x4 *dst, size_t nv) { for (size_t v = (size_t) blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (size_t) gridDim.x * blockDim.x) dst [v] = u32x4 { 1u, 2u, 3u, (unsigned int) v }; }
__global__ void copy_k (const u32x4 *src, u32x4 *dst, size_t nv) { for (size_t v = (size_t) blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (size_t) gridDim.x * blockDim.x) dst [v] = src [v]; }

int main ()
{
    const size_t frames = 1 << 20; const int C = 8; const size_t nv = frames * C / 4;
    u32x4 *in, *junk, *junk2; unsigned int *peak;
    hipMalloc (&in, nv * 16); hipMalloc (&junk, 160u << 20); hipMalloc (&junk2, 160u << 20); hipMalloc (&peak, 128); hipMemset (peak, 0, 128);
    std::vector<float> h (frames * C); for (size_t i = 0; i < h.size (); ++i) h [i] = (float)((i * 2654435761u) % 1000003u) / 1000003.0f - 0.5f;
    hipMemcpy (in, h.data (), nv * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    auto run = [&] (const char *name, int dirty_mb, auto launch) {
        float best = 1e9f, sum = 0.0f; const int reps = 30;
        for (int r = 0; r < reps + 5; ++r) {
            if (dirty_mb) { const size_t jn = (size_t) dirty_mb * (1u << 20) / 16; hipLaunchKernelGGL (fill_k, dim3 (2048), dim3 (256), 0, 0, junk, jn); hipLaunchKernelGGL (copy_k, dim3 (2048), dim3 (256), 0, 0, junk, junk2, jn); }
            hipEventRecord (e0, 0); launch (); hipEventRecord (e1, 0); hipEventSynchronize (e1);
            float ms; hipEventElapsedTime (&ms, e0, e1);
            if (r >= 5) { sum += ms; if (ms < best) best = ms; }
        }
        printf ("%-44s dirty %3d MB: avg %.2f us  best %.2f us  (%.2f TB/s avg)\n", name, dirty_mb, sum / reps * 1e3, best * 1e3, nv * 16 / (sum / reps * 1e-3) / 1e12);
    };
    for (int dirty : { 0, 70, 150 }) {
        run ("peak 512 x 1024, 4 in flight", dirty, [&] { hipLaunchKernelGGL (peak_k<4>, dim3 (512), dim3 (1024), 0, 0, in, nv, C, peak); });
        run ("peak 256 x 1024, 8 in flight", dirty, [&] { hipLaunchKernelGGL (peak_k<8>, dim3 (256), dim3 (1024), 0, 0, in, nv, C, peak); });
        run ("peak 1024 x 512, 4 in flight", dirty, [&] { hipLaunchKernelGGL (peak_k<4>, dim3 (1024), dim3 (512), 0, 0, in, nv, C, peak); });
        run ("peak 2048 x 256, 4 in flight", dirty, [&] { hipLaunchKernelGGL (peak_k<4>, dim3 (2048), dim3 (256), 0, 0, in, nv, C, peak); });
        run ("peak 1024 x 256, 8 in flight", dirty, [&] { hipLaunchKernelGGL (peak_k<8>, dim3 (1024), dim3 (256), 0, 0, in, nv, C, peak); });
        run ("peak 4096 x 256, 2 in flight", dirty, [&] { hipLaunchKernelGGL (peak_k<2>, dim3 (4096), dim3 (256), 0, 0, in, nv, C, peak); });
        run ("peak 8192 x 256, 1 in flight", dirty, [&] { hipLaunchKernelGGL (peak_k<1>, dim3 (8192), dim3 (256), 0, 0, in, nv, C, peak); });
        run ("copy 2048 x 256 (33.5 MB -> 33.5 MB)", dirty, [&] { hipLaunchKernelGGL (copy_k, dim3 (2048), dim3 (256), 0, 0, in, junk2, nv); });
    }
    run ("empty kernel pair (event overhead)", 0, [&] { hipLaunchKernelGGL (fill_k, dim3 (1), dim3 (64), 0, 0, junk, (size_t) 0); });
    return 0;
}
