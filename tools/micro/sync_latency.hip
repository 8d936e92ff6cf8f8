// Micro-benchmark: how long the host waits for a ~10 us kernel by (a) hipStreamSynchronize, (b) an event polled with hipEventQuery,
// (c) a 32-bit value the stream writes to page-locked memory (hipStreamWriteValue32) and the host spins on, (d) a value the kernel's
// last workgroup writes itself.  What the host-pointer API of small calls could save.   build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void busy (int n, float *out, volatile unsigned int *flag, unsigned int *count, unsigned int seq)
{
    float a = threadIdx.x;
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) out [0] = a;
    if (flag) {
        __syncthreads ();
        if (threadIdx.x == 0) {
            __threadfence ();
            if (atomicAdd (count, 1u) == gridDim.x - 1) { *count = 0; __threadfence_system (); *flag = seq; }
        }
    }
}
static double now () { return std::chrono::duration<double, std::micro> (std::chrono::steady_clock::now ().time_since_epoch ()).count (); }
int main ()
{
    hipStream_t st; hipStreamCreateWithFlags (&st, hipStreamNonBlocking);
    float *out; hipMalloc (&out, 4096);
    unsigned int *flag, *count; hipHostMalloc ((void **) &flag, 64, hipHostMallocDefault); *flag = 0; hipMalloc (&count, 4); hipMemset (count, 0, 4);
    hipEvent_t ev; hipEventCreateWithFlags (&ev, hipEventDisableTiming);
    const int N = 2000, iters = 300;
    for (int mode = 0; mode < 5; ++mode) {
        double total = 0; unsigned int seq = 0;
        for (int it = 0; it < iters + 20; ++it) {
            const double t0 = now ();
            ++seq;
            if (mode == 3) hipLaunchKernelGGL (busy, dim3 (512), dim3 (256), 0, st, N, out, flag, count, seq);
            else hipLaunchKernelGGL (busy, dim3 (512), dim3 (256), 0, st, N, out, (volatile unsigned int *) nullptr, count, seq);
            if (mode == 0) hipStreamSynchronize (st);
            else if (mode == 1) { hipEventRecord (ev, st); while (hipEventQuery (ev) == hipErrorNotReady) ; }
            else if (mode == 2) { hipStreamWriteValue32 (st, flag, seq, 0); while (*(volatile unsigned int *) flag != seq) ; }
            else if (mode == 3) { while (*(volatile unsigned int *) flag != seq) ; }
            else { hipEventRecord (ev, st); hipEventSynchronize (ev); }
            if (it >= 20) total += now () - t0;
        }
        hipStreamSynchronize (st);
        const char *names [] = { "hipStreamSynchronize", "hipEventRecord + hipEventQuery spin", "hipStreamWriteValue32 + spin on pinned word", "kernel's last workgroup writes pinned word, spin", "hipEventRecord + hipEventSynchronize" };
        printf ("%-52s %7.2f us per launch+wait\n", names [mode], total / iters);
    }
    // the kernel alone
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    hipEventRecord (e0, st);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL (busy, dim3 (512), dim3 (256), 0, st, N, out, (volatile unsigned int *) nullptr, count, 0u);
    hipEventRecord (e1, st); hipEventSynchronize (e1);
    float ms; hipEventElapsedTime (&ms, e0, e1);
    printf ("kernel alone, back to back: %.2f us\n", ms * 10);
    return 0;
}
