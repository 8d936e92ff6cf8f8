// fetch_calib.hip — what rocprofv3's FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count, per access width (gfx950): streaming reads
// of 512 MiB (beyond the 256 MiB Infinity Cache) with 4-, 8- and 16-byte loads per lane, and a 512 MiB streaming write.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d out -o calib -- ./fetch_calib
// Expected: 524,288 KB per kernel; the ratio reported / expected is the correction for that access width.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2_t __attribute__ ((ext_vector_type (2)));
typedef unsigned int u32x4_t __attribute__ ((ext_vector_type (4)));
template <typename T> __global__ void read_k (const T *src, size_t n, T *sink)
{
    T acc = {};
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        const T v = src [i];
        const unsigned int *w = reinterpret_cast<const unsigned int *> (&v);
        unsigned int *a = reinterpret_cast<unsigned int *> (&acc);
        for (unsigned int k = 0; k < sizeof (T) / 4; ++k) a [k] ^= w [k];
    }
    const unsigned int *a = reinterpret_cast<const unsigned int *> (&acc);
    unsigned int any = 0; for (unsigned int k = 0; k < sizeof (T) / 4; ++k) any |= a [k];
    if (any == 0x12345678u) sink [0] = acc;                  // (never true: keeps the loads alive)
}
__global__ void write_k (uint4 *dst, size_t n) { for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) dst [i] = uint4 { 1u, 2u, 3u, (unsigned int) i }; }
int main ()
{
    const size_t bytes = (size_t) 512 << 20;
    void *a, *b; hipMalloc (&a, bytes); hipMalloc (&b, bytes); hipMemset (a, 1, bytes); hipMemset (b, 2, bytes);
    hipDeviceSynchronize ();
    hipLaunchKernelGGL (read_k<unsigned int>, dim3 (4096), dim3 (256), 0, 0, (const unsigned int *) a, bytes / 4, (unsigned int *) b);
    hipLaunchKernelGGL (read_k<u32x2_t>, dim3 (4096), dim3 (256), 0, 0, (const u32x2_t *) b, bytes / 8, (u32x2_t *) a);
    hipLaunchKernelGGL (read_k<u32x4_t>, dim3 (4096), dim3 (256), 0, 0, (const u32x4_t *) a, bytes / 16, (u32x4_t *) b);
    hipLaunchKernelGGL (read_k<double>, dim3 (4096), dim3 (256), 0, 0, (const double *) b, bytes / 8, (double *) a);
    hipLaunchKernelGGL (write_k, dim3 (4096), dim3 (256), 0, 0, (uint4 *) a, bytes / 16);
    hipDeviceSynchronize ();
    printf ("expected per kernel: %zu KB read (read_k) / written (write_k)\n", bytes / 1024);
    return 0;
}
