// Micro-benchmark: sustained rate of the f32 / f64 MFMA forms used by the FIR kernels (no memory traffic).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__ ((ext_vector_type (16)));
typedef float f32x4 __attribute__ ((ext_vector_type (4)));
typedef double f64x4 __attribute__ ((ext_vector_type (4)));

template <int CHAINS>
__global__ __launch_bounds__ (256) void k_f32_32x32x2 (float *out, int iters, float a, float b)
{
    f32x16 acc [CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc [c][r] = 0.0f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc [c] = __builtin_amdgcn_mfma_f32_32x32x2f32 (a, b, acc [c], 0, 0, 0);
    float s = 0; for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc [c][r];
    out [blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
__global__ __launch_bounds__ (256) void k_f32_16x16x4 (float *out, int iters, float a, float b)
{
    f32x4 acc [CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) acc [c][r] = 0.0f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc [c] = __builtin_amdgcn_mfma_f32_16x16x4f32 (a, b, acc [c], 0, 0, 0);
    float s = 0; for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) s += acc [c][r];
    out [blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
__global__ __launch_bounds__ (256) void k_f64_16x16x4 (float *out, int iters, double a, double b)
{
    f64x4 acc [CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) acc [c][r] = 0.0;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc [c] = __builtin_amdgcn_mfma_f64_16x16x4f64 (a, b, acc [c], 0, 0, 0);
    double s = 0; for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) s += acc [c][r];
    out [blockIdx.x * blockDim.x + threadIdx.x] = (float) s;
}
// f32 MFMA chain with the kernel's fp64 flush (16 cvt + 16 add per 16 MFMAs) interleaved
__global__ __launch_bounds__ (256) void k_f32_flush (float *out, int iters, float a, float b)
{
    double sum [16]; for (int r = 0; r < 16; ++r) sum [r] = 0.0;
    for (int i = 0; i < iters; ++i) {
        f32x16 acc; for (int r = 0; r < 16; ++r) acc [r] = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32 (a, b, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) sum [r] += (double) acc [r];
    }
    double s = 0; for (int r = 0; r < 16; ++r) s += sum [r];
    out [blockIdx.x * blockDim.x + threadIdx.x] = (float) s;
}

template <typename F> double timeit (F launch)
{
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    launch (); hipDeviceSynchronize ();
    hipEventRecord (e0); launch (); hipEventRecord (e1); hipEventSynchronize (e1);
    float ms; hipEventElapsedTime (&ms, e0, e1); return ms;
}

int main ()
{
    float *out; hipMalloc (&out, 4 << 20);
    const int iters = 20000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu;
        const double waves = (double) grid * 4;
        double ms;
        ms = timeit ([&] { hipLaunchKernelGGL (k_f32_32x32x2<1>, dim3 (grid), dim3 (256), 0, 0, out, iters, 1.0f, 2.0f); });
        printf ("wg/cu %d  f32 32x32x2 1 chain : %7.1f TFLOP/s\n", wgs_per_cu, waves * iters * 1 * 4096.0 / ms / 1e9);
        ms = timeit ([&] { hipLaunchKernelGGL (k_f32_32x32x2<2>, dim3 (grid), dim3 (256), 0, 0, out, iters, 1.0f, 2.0f); });
        printf ("wg/cu %d  f32 32x32x2 2 chains: %7.1f TFLOP/s\n", wgs_per_cu, waves * iters * 2 * 4096.0 / ms / 1e9);
        ms = timeit ([&] { hipLaunchKernelGGL (k_f32_16x16x4<4>, dim3 (grid), dim3 (256), 0, 0, out, iters, 1.0f, 2.0f); });
        printf ("wg/cu %d  f32 16x16x4 4 chains: %7.1f TFLOP/s\n", wgs_per_cu, waves * iters * 4 * 2048.0 / ms / 1e9);
        ms = timeit ([&] { hipLaunchKernelGGL (k_f64_16x16x4<4>, dim3 (grid), dim3 (256), 0, 0, out, iters, 1.0, 2.0); });
        printf ("wg/cu %d  f64 16x16x4 4 chains: %7.1f TFLOP/s\n", wgs_per_cu, waves * iters * 4 * 2048.0 / ms / 1e9);
        ms = timeit ([&] { hipLaunchKernelGGL (k_f32_flush, dim3 (grid), dim3 (256), 0, 0, out, iters / 16, 1.0f, 2.0f); });
        printf ("wg/cu %d  f32 32x32x2 + fp64 flush per 16: %7.1f TFLOP/s\n", wgs_per_cu, waves * (iters / 16) * 16 * 4096.0 / ms / 1e9);
    }
    return 0;
}
