// Probe: where does the dispatcher put the workgroups of a grid shaped like the f32 streaming kernels' (512 threads, 46 KB of LDS, 80 VGPRs: three fit a CU)?
// Every workgroup records its XCC, SE, CU and start time, then spins ~25 us so that the whole grid is resident together.  Prints, per XCD, how many
// workgroups each CU got and which ranks (blockIdx >> 3) they were: breadth first (rank r, r + 32, r + 64 on a CU), depth first (3k .. 3k + 2), or neither.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dispatch_probe.hip -o tools/micro/dispatch_probe ; run: tools/micro/dispatch_probe [workgroups per XCD = 96]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
__global__ __launch_bounds__ (512) void probe (unsigned int *rec, long long spin)
{
    __shared__ float lds [46080 / 4];
    unsigned int hw, xcc;
    asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s" (hw));
    asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s" (xcc));
    const long long t0 = (long long) __builtin_amdgcn_s_memrealtime ();
    lds [threadIdx.x] = (float) hw;
    if (threadIdx.x == 0) { rec [blockIdx.x * 4 + 0] = hw; rec [blockIdx.x * 4 + 1] = xcc; rec [blockIdx.x * 4 + 2] = (unsigned int) t0; }
    while ((long long) __builtin_amdgcn_s_memrealtime () - t0 < spin) { }
    __syncthreads ();
    if (threadIdx.x == 0) rec [blockIdx.x * 4 + 3] = (unsigned int) __builtin_amdgcn_s_memrealtime () + (unsigned int) lds [1];
}
int main (int argc, char **argv)
{
    const int per_xcd = argc > 1 ? atoi (argv [1]) : 96, blocks = 8 * per_xcd;
    unsigned int *d; (void) hipMalloc (&d, blocks * 16); std::vector<unsigned int> h (blocks * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL (probe, dim3 (blocks), dim3 (512), 0, 0, d, (long long) 2500);        // 100 MHz ticks: 25 us
        (void) hipDeviceSynchronize ();
    }
    (void) hipMemcpy (h.data (), d, blocks * 16, hipMemcpyDeviceToHost);
    unsigned int tmin = 0xffffffffu; for (int b = 0; b < blocks; ++b) tmin = std::min (tmin, h [b * 4 + 2]);
    for (int x = 0; x < 8; ++x) {
        std::map<unsigned int, std::vector<int>> cus;
        int xcc_of = -1;
        for (int b = x; b < blocks; b += 8) {
            const unsigned int hw = h [b * 4], cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cus [(se << 8) | (sh << 4) | cu].push_back (b >> 3);
            xcc_of = (int)(h [b * 4 + 1] & 15);
        }
        int hist [8] = { 0 };
        for (auto &kv : cus) hist [std::min ((int) kv.second.size (), 7)]++;
        printf ("blockIdx & 7 = %d -> XCC %d: %zu CUs used; CUs with 1/2/3/4 workgroups: %d %d %d %d\n", x, xcc_of, cus.size (), hist [1], hist [2], hist [3], hist [4]);
        if (x == 0) for (auto &kv : cus) {
            printf ("   se %u sh %u cu %2u: ranks", kv.first >> 8, (kv.first >> 4) & 1, kv.first & 15);
            for (int r : kv.second) printf (" %3d (t+%.1f us)", r, (h [(r * 8 + x) * 4 + 2] - tmin) / 100.0);
            printf ("\n");
        }
    }
    return 0;
}
