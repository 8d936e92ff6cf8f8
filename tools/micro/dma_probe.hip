// Micro-benchmark: what the L2 -> LDS path (buffer_load_dwordx4 ... lds, 1 KB per wave-instruction) delivers per CU as a function of the
// ADDRESS PATTERN of a piece, from an L2-resident working set, all CUs busy — the question behind the fixed-point FIR kernels' staging:
//   pattern 0: a piece = 1 KB contiguous (8 whole 128-byte lines)                      — the filter rows' pieces
//   pattern 1: a piece = 16 segments of 64 bytes, 18,816 bytes apart                    — the X digit planes' pieces of rounds 2-3
//              ([plane][4-frame block][channel] planes of an 8-channel stream: 2 blocks x 32 bytes of 16 periods 588 frames apart)
//   pattern 2: a piece = 32 segments of 32 bytes, 18,816 bytes apart
//   pattern 3: a piece = 4 segments of 256 bytes
//   pattern 4: a piece = 2 segments of 512 bytes
// Each workgroup (8 waves, one per CU: 160 KB of LDS would be the kernel's) walks its own window of the buffer; every wave issues
// `pieces` pieces per step, waits for them (vmcnt(0)) and meets the others at a barrier, `steps` times; 2 steps are in flight.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dma_probe.hip -o tools/micro/dma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__ ((address_space (3))) void *lds_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc (const void *base, unsigned int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc (const_cast<void *> (base), 0, (int) bytes, 0x00020000);
}

template <int PATTERN, int PIECES>
__global__ __launch_bounds__ (512) void k_dma (const unsigned char *src, unsigned int bytes_per_xcd, int steps, int *sink)
{
    __shared__ __attribute__ ((aligned (16))) unsigned char lds [2 * 8 * PIECES * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;
    const __amdgpu_buffer_rsrc_t r = make_rsrc (src + (size_t) xcd * bytes_per_xcd, bytes_per_xcd);
    // the lane's offset inside a piece
    unsigned int loff;
    if (PATTERN == 0) loff = lane * 16;
    else if (PATTERN == 1) loff = (lane >> 2) * 18816u + (lane & 3) * 16;
    else if (PATTERN == 2) loff = (lane >> 1) * 18816u + (lane & 1) * 16;
    else if (PATTERN == 3) loff = (lane >> 4) * 18816u + (lane & 15) * 16;
    else loff = (lane >> 5) * 18816u + (lane & 31) * 16;
    // the workgroup's window and this wave's pieces in it; a step advances every piece by its own contiguous length
    const unsigned int seg = PATTERN == 0 ? 1024u : PATTERN == 1 ? 64u : PATTERN == 2 ? 32u : PATTERN == 3 ? 256u : 512u;
    unsigned int base = (unsigned int) rank * 37u * 4096u + (unsigned int) wave * (PATTERN == 0 ? 65536u : 1u << 20) % (bytes_per_xcd / 2);
    auto issue = [&] (int buf, int step) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            unsigned int off = base + loff + (unsigned int) p * (PATTERN == 0 ? 8192u : 301056u) + (unsigned int) step * seg;
            off %= (bytes_per_xcd - 1024u);
            off &= ~15u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds (r, (lds_ptr_t)(lds + ((buf * 8 + wave) * PIECES + p) * 1024), 16, (int) off, 0, 0, 0);
        }
    };
    issue (0, 0);
    for (int s = 0; s < steps; ++s) {
        issue ((s + 1) & 1, s + 1);
        if (PIECES == 1) asm volatile ("s_waitcnt vmcnt(1)" ::: "memory");
        else if (PIECES == 2) asm volatile ("s_waitcnt vmcnt(2)" ::: "memory");
        else if (PIECES == 4) asm volatile ("s_waitcnt vmcnt(4)" ::: "memory");
        else if (PIECES == 5) asm volatile ("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile ("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_barrier ();
    }
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads ();
    if (sink && threadIdx.x == 0) sink [blockIdx.x] = lds [blockIdx.x & 1023];
}

template <int PATTERN, int PIECES>
static void run (const unsigned char *d, unsigned int bytes_per_xcd, int *sink, const char *what)
{
    const int steps = 2000;
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    hipLaunchKernelGGL ((k_dma<PATTERN, PIECES>), dim3 (256), dim3 (512), 0, 0, d, bytes_per_xcd, 200, sink);
    hipDeviceSynchronize ();
    hipEventRecord (e0);
    hipLaunchKernelGGL ((k_dma<PATTERN, PIECES>), dim3 (256), dim3 (512), 0, 0, d, bytes_per_xcd, steps, sink);
    hipEventRecord (e1); hipEventSynchronize (e1);
    float ms = 0; hipEventElapsedTime (&ms, e0, e1);
    const double bytes = 256.0 * 8 * PIECES * 1024.0 * steps;
    printf ("pattern %d (%s), %2d pieces per wave and step, window %u MB per XCD: %7.1f us  %7.2f TB/s  %6.1f GB/s per CU  %5.1f B/clk/CU at 2.1 GHz\n",
            PATTERN, what, PIECES, bytes_per_xcd >> 20, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256.0, bytes / ms / 1e6 / 256.0 / 2.1);
}

int main ()
{
    const unsigned int per_xcd = 3u << 20;                    // L2-resident: 3 MB per XCD
    unsigned char *d; int *sink;
    hipMalloc (&d, (size_t) per_xcd * 8 + 4096); hipMemset (d, 1, (size_t) per_xcd * 8 + 4096);
    hipMalloc (&sink, 4096 * sizeof (int));
    run<0, 5> (d, per_xcd, sink, "1 KB contiguous");
    run<1, 5> (d, per_xcd, sink, "16 x 64 B");
    run<2, 5> (d, per_xcd, sink, "32 x 32 B");
    run<3, 5> (d, per_xcd, sink, "4 x 256 B");
    run<4, 5> (d, per_xcd, sink, "2 x 512 B");
    run<0, 10> (d, per_xcd, sink, "1 KB contiguous");
    run<1, 10> (d, per_xcd, sink, "16 x 64 B");
    run<3, 10> (d, per_xcd, sink, "4 x 256 B");
    run<0, 2> (d, per_xcd, sink, "1 KB contiguous");
    run<1, 2> (d, per_xcd, sink, "16 x 64 B");
    return 0;
}
