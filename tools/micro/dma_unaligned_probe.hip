// Probe: does buffer_load_dwordx4 ... lds (LDS-DMA) take a global address that is not a multiple of 4 bytes — i.e. could ONE copy of the
// fixed-point kernel's filter rows (K-contiguous bytes) serve every residue r = 0..3 of a tile family by a byte offset in the address,
// instead of one shifted copy per residue in memory?  Loads 64 lanes x 16 bytes from base + lane * 16 + shift for shift = 0..3 and prints
// the first bytes that landed in the LDS against the expected ones.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dma_unaligned_probe.hip -o tools/micro/dma_unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__ ((address_space (3))) void *lds_ptr_t;
__global__ void k (const unsigned char *src, unsigned char *dst, int shift)
{
    __shared__ __attribute__ ((aligned (16))) unsigned char lds [1024];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc (const_cast<unsigned char *> (src), 0, 4096, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds (r, (lds_ptr_t) lds, 16, (int)(threadIdx.x * 16 + shift), 0, 0, 0);
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads ();
    for (int i = threadIdx.x; i < 1024; i += 64) dst [i] = lds [i];
}
int main ()
{
    unsigned char h [4096], out [1024], *d, *o;
    for (int i = 0; i < 4096; ++i) h [i] = (unsigned char)(i * 7 + 3);
    (void) hipMalloc (&d, 4096); (void) hipMalloc (&o, 1024); (void) hipMemcpy (d, h, 4096, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL (k, dim3 (1), dim3 (64), 0, 0, d, o, shift);
        (void) hipMemcpy (out, o, 1024, hipMemcpyDeviceToHost);
        int exact = 0, truncated = 0;
        for (int i = 0; i < 1024; ++i) { exact += out [i] == h [i + shift]; truncated += out [i] == h [i]; }
        printf ("shift %d: %4d of 1024 bytes = source[i + shift], %4d = source[i] (address rounded down); first bytes", shift, exact, truncated);
        for (int i = 0; i < 8; ++i) printf (" %02x", out [i]);
        printf (" (want"); for (int i = 0; i < 8; ++i) printf (" %02x", h [i + shift]); printf (")\n");
    }
    return 0;
}
