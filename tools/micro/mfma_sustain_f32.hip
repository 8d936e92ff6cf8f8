// Micro-benchmark: what v_mfma_f32_32x32x2_f32 SUSTAINS on MI355X — over hundreds of milliseconds, with operands that toggle like real digit
// planes (pseudo-random bytes) or not at all (zeros), from registers only (no LDS, no memory).  The FIR kernels' launches sit behind a 200 ms
// pre-roll: if the chip's power management holds the matrix cores well below the 2.4 GHz x 2048 ops/clk/SIMD figure once the load is sustained
// and the data is live, that — not the kernels' schedules — is what their "fraction of the dense int8 peak" measures.
// Prints, per 25 ms window: TFLOP/s, and the shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz) seen by one wave.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_sustain_f32.hip -o tools/micro/mfma_sustain_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef float f32x16 __attribute__ ((ext_vector_type (16)));


template <int CHAINS>
__global__ __launch_bounds__ (256) void k_mfma (int *out, long long *clk, int iters, int random)
{
    const int tid = threadIdx.x + blockIdx.x * 256;
    float a [4], b [4];
    for (int p = 0; p < 4; ++p) {
        a [p] = random ? (float)(int)((unsigned)(tid * 4 + 16 * p) * 2654435761u) * 4.6566e-10f : 0.0f;
        b [p] = random ? (float)(int)((unsigned)(tid * 4 + 16 * p + 7) * 40503u * 2246822519u) * 4.6566e-10f : 0.0f;
    }
    f32x16 acc [CHAINS];
    for (int s = 0; s < CHAINS; ++s) for (int r = 0; r < 16; ++r) acc [s] [r] = 0;
    const long long t0 = (long long) __builtin_readcyclecounter (), r0 = (long long) __builtin_amdgcn_s_memrealtime ();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < CHAINS; ++s) acc [s] = __builtin_amdgcn_mfma_f32_32x32x2f32 (a [s & 3], b [(s + i) & 3], acc [s], 0, 0, 0);
    }
    const long long t1 = (long long) __builtin_readcyclecounter (), r1 = (long long) __builtin_amdgcn_s_memrealtime ();
    float sum = 0;
    for (int s = 0; s < CHAINS; ++s) for (int r = 0; r < 16; ++r) sum += acc [s] [r];
    out [tid] = (int) sum;
    if (tid == 0) { clk [0] = t1 - t0; clk [1] = r1 - r0; }
}

int main ()
{
    int *out; long long *clk;
    (void) hipMalloc (&out, 256 * 1024 * 8 * sizeof (int)); (void) hipMalloc (&clk, 16);
    const int iters = 4000, CH = 5;
    for (int random = 1; random >= 0; --random)
        for (int wpc = 1; wpc <= 2; ++wpc) {
            printf ("---- %s operands, %d workgroup(s) of 4 waves per CU, %d independent chains per wave\n", random ? "pseudo-random" : "zero", wpc, CH);
            const auto start = std::chrono::steady_clock::now ();
            double window_ops = 0; auto wstart = start; int launches = 0;
            while (std::chrono::duration<double> (std::chrono::steady_clock::now () - start).count () < 0.6) {
                hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
                (void) hipEventRecord (e0);
                for (int k = 0; k < 8; ++k) hipLaunchKernelGGL ((k_mfma<CH>), dim3 (256 * wpc), dim3 (256), 0, 0, out, clk, iters, random);
                (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
                float ms = 0; (void) hipEventElapsedTime (&ms, e0, e1);
                (void) hipEventDestroy (e0); (void) hipEventDestroy (e1);
                const double ops = 8.0 * 256 * wpc * 4 * (double) iters * CH * 4096.0;
                window_ops += ops; ++launches;
                const double wt = std::chrono::duration<double> (std::chrono::steady_clock::now () - wstart).count ();
                if (wt >= 0.05) {
                    long long h [2]; (void) hipMemcpy (h, clk, 16, hipMemcpyDeviceToHost);
                    printf ("   t = %5.0f ms: %7.0f TFLOP/s (this batch %7.0f), shader clock %.2f GHz\n",
                            std::chrono::duration<double> (std::chrono::steady_clock::now () - start).count () * 1e3, window_ops / wt / 1e12, ops / (ms * 1e-3) / 1e12,
                            (double) h [0] / (double) h [1] * 0.1);
                    window_ops = 0; wstart = std::chrono::steady_clock::now ();
                }
            }
        }
    return 0;
}
