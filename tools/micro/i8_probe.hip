// Micro-benchmark / layout check for v_mfma_i32_32x32x32_i8 (gfx950) as a 32-bit fixed-point dot-product engine:
//  (1) operand / result layout against a CPU product;  (2) sustained rate of the 13-products-per-32-taps pattern (four signed
//  8-bit digits per operand, digit pairs i + j <= 4, five i32 accumulators) with operands in registers, read from LDS, and read
//  from LDS while other waves of the workgroup write it at the staging rate.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/i8_probe.hip -o tools/micro/i8_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int i32x4 __attribute__ ((ext_vector_type (4)));
typedef int i32x16 __attribute__ ((ext_vector_type (16)));

__global__ void k_layout (const i32x4 *a, const i32x4 *b, i32x16 *d)
{
    i32x16 acc; for (int r = 0; r < 16; ++r) acc [r] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8 (a [threadIdx.x], b [threadIdx.x], acc, 0, 0, 0);
    d [threadIdx.x] = acc;
}

constexpr int PITCH = 48;                 // bytes per (row, plane) of 32 taps in LDS: 32 + 16 pad => conflict-free b128 reads
// MODE 0: operands in registers; 1: operands from LDS (8 b128 reads per 13 MFMAs); 2: as 1 with 4 more waves writing the other
// LDS buffer at the staging rate (20 KB per chunk) and one barrier per chunk
template <int MODE, int CLASSES>
__global__ __launch_bounds__ (512) void k_rate (int *out, int chunks)
{
    __shared__ __attribute__ ((aligned (16))) unsigned char As [2] [4] [32 * PITCH];
    __shared__ __attribute__ ((aligned (16))) unsigned char Bs [2] [4] [128 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < (int) sizeof (As) / 4; e += blockDim.x) reinterpret_cast<int *> (As) [e] = e * 2654435761u;
    for (int e = tid; e < (int) sizeof (Bs) / 4; e += blockDim.x) reinterpret_cast<int *> (Bs) [e] = e * 40503u;
    __syncthreads ();
    if (wave >= 4) {
        if (MODE < 2) return;
        if (MODE == 3) { for (int c = 0; c < chunks; ++c) __syncthreads (); return; }
        // staging stand-in: 16 b32 writes + 1 b128 write per thread and chunk (X: 16 KB, A: 4 KB), then the barrier
        const int pt = tid & 255;
        int v = pt;
        for (int c = 0; c < chunks; ++c) {
            const int buf = (c & 1) ^ 1;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    *reinterpret_cast<int *> (&Bs [buf] [u] [((pt >> 4) * 8 + (pt & 1) * 4 + e) * PITCH + ((pt >> 1) & 7) * 4]) = v + e;
            i32x4 w = { v, v + 1, v + 2, v + 3 };
            *reinterpret_cast<i32x4 *> (&As [buf] [pt >> 6] [((pt >> 1) & 31) * PITCH + (pt & 1) * 16]) = w;
            v += 7;
            __syncthreads ();
        }
        return;
    }
    i32x16 acc [5];
    for (int s = 0; s < 5; ++s) for (int r = 0; r < 16; ++r) acc [s] [r] = 0;
    const int aoff = (lane & 31) * PITCH + (lane >> 5) * 16, boff = (wave * 32 + (lane & 31)) * PITCH + (lane >> 5) * 16;
    i32x4 a [4], b [4];
    for (int p = 0; p < 4; ++p) { a [p] = *reinterpret_cast<const i32x4 *> (&As [0] [p] [aoff]); b [p] = *reinterpret_cast<const i32x4 *> (&Bs [0] [p] [boff]); }
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        if (MODE >= 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) { a [p] = *reinterpret_cast<const i32x4 *> (&As [buf] [p] [aoff]); b [p] = *reinterpret_cast<const i32x4 *> (&Bs [buf] [p] [boff]); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i + j < CLASSES) acc [i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (a [i], b [j], acc [i + j], 0, 0, 0);
        if (MODE >= 2) __syncthreads ();
    }
    int s = 0; for (int q = 0; q < 5; ++q) for (int r = 0; r < 16; ++r) s += acc [q] [r];
    out [blockIdx.x * 256 + tid] = s;
}

// The same work organised as ONE 12-wave workgroup per CU: waves 0-7 multiply two independent tiles (waves 0-3 / 4-7), each with
// TWO operand register sets (the reads of chunk c + 1 land while the products of chunk c issue), waves 8-11 stage both tiles.
// NINE products per chunk (the common case of the kernel: the top digit plane of the rows is zero).  TWIN = false: the
// eight-wave form above with nine products, for comparison (run at two workgroups per CU).
template <bool TWIN>
__global__ __launch_bounds__ (TWIN ? 768 : 512) void k_shape (int *out, int chunks)
{
    constexpr int TILES = TWIN ? 2 : 1;
    __shared__ __attribute__ ((aligned (16))) unsigned char As [TILES] [2] [4] [32 * PITCH];
    __shared__ __attribute__ ((aligned (16))) unsigned char Bs [TILES] [2] [4] [128 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < (int) sizeof (As) / 4; e += blockDim.x) reinterpret_cast<int *> (As) [e] = e * 2654435761u;
    for (int e = tid; e < (int) sizeof (Bs) / 4; e += blockDim.x) reinterpret_cast<int *> (Bs) [e] = e * 40503u;
    __syncthreads ();
    if (wave >= 4 * TILES) {
        const int pt = tid & 255;
        int v = pt;
        for (int c = 0; c < chunks; ++c) {
            const int buf = (c & 1) ^ 1;
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        *reinterpret_cast<int *> (&Bs [t] [buf] [u] [((pt >> 4) * 8 + (pt & 1) * 4 + e) * PITCH + ((pt >> 1) & 7) * 4]) = v + e;
                i32x4 w = { v, v + 1, v + 2, v + 3 };
                *reinterpret_cast<i32x4 *> (&As [t] [buf] [pt >> 6] [((pt >> 1) & 31) * PITCH + (pt & 1) * 16]) = w;
            }
            v += 7;
            __syncthreads ();
        }
        return;
    }
    const int tile = wave >> 2, w4 = wave & 3;
    i32x16 acc [5];
    for (int s = 0; s < 5; ++s) for (int r = 0; r < 16; ++r) acc [s] [r] = 0;
    const int aoff = (lane & 31) * PITCH + (lane >> 5) * 16, boff = (w4 * 32 + (lane & 31)) * PITCH + (lane >> 5) * 16;
    auto rd = [&] (i32x4 (&a) [4], i32x4 (&b) [4], int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { a [p] = *reinterpret_cast<const i32x4 *> (&As [tile] [buf] [p] [aoff]); b [p] = *reinterpret_cast<const i32x4 *> (&Bs [tile] [buf] [p] [boff]); }
    };
    auto mm = [&] (const i32x4 (&a) [4], const i32x4 (&b) [4]) {
#pragma unroll
        for (int i = 1; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i + j < 5) acc [i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8 (a [i], b [j], acc [i + j], 0, 0, 0);
    };
    i32x4 a0 [4], b0 [4], a1 [4], b1 [4];
    if (TWIN) {
        rd (a0, b0, 0);
        for (int c = 0; c < chunks; c += 2) {
            rd (a1, b1, 1); mm (a0, b0); __syncthreads ();
            rd (a0, b0, 0); mm (a1, b1); __syncthreads ();
        }
    }
    else {
        for (int c = 0; c < chunks; ++c) { rd (a0, b0, c & 1); __syncthreads (); mm (a0, b0); }
    }
    int s = 0; for (int q = 0; q < 5; ++q) for (int r = 0; r < 16; ++r) s += acc [q] [r];
    out [blockIdx.x * 256 + (tid & 255)] = s;
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
    // ---- layout: A[m][k], B[k][n] random int8; lane l holds A[l & 31][16 * (l >> 5) + 0..15] (byte q of dword d = k offset 4 d + q)
    std::vector<signed char> A (32 * 32), B (32 * 32);
    srand (3); for (auto &v : A) v = (signed char)(rand () % 256 - 128); for (auto &v : B) v = (signed char)(rand () % 256 - 128);
    std::vector<int> ha (64 * 4), hb (64 * 4), hd (64 * 16);
    for (int l = 0; l < 64; ++l) for (int d = 0; d < 4; ++d) {
        unsigned int wa = 0, wb = 0;
        for (int q = 0; q < 4; ++q) { int k = 16 * (l >> 5) + 4 * d + q; wa |= (unsigned int)(unsigned char) A [(l & 31) * 32 + k] << (8 * q); wb |= (unsigned int)(unsigned char) B [k * 32 + (l & 31)] << (8 * q); }
        ha [l * 4 + d] = (int) wa; hb [l * 4 + d] = (int) wb;
    }
    int *da, *db, *dd; hipMalloc (&da, 1024); hipMalloc (&db, 1024); hipMalloc (&dd, 4096);
    hipMemcpy (da, ha.data (), 1024, hipMemcpyHostToDevice); hipMemcpy (db, hb.data (), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL (k_layout, dim3 (1), dim3 (64), 0, 0, (const i32x4 *) da, (const i32x4 *) db, (i32x16 *) dd);
    hipMemcpy (hd.data (), dd, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = l & 31;
        int ref = 0; for (int k = 0; k < 32; ++k) ref += (int) A [m * 32 + k] * (int) B [k * 32 + n];
        bad += ref != hd [l * 16 + r];
    }
    printf ("layout check (A row = lane & 31, B col = lane & 31, D row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = lane & 31): %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);

    int *out; hipMalloc (&out, 64 << 20);
    const int chunks = 20000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu;
#define RUN(MODE, CLASSES, NM) do { double ms = timeit ([&] { hipLaunchKernelGGL ((k_rate<MODE, CLASSES>), dim3 (grid), dim3 (512), 0, 0, out, chunks); }); \
        printf ("wg/cu %d  mode %d (%s) classes %d: %7.3f ms  %8.1f Tops/s  (%.0f cycles per 32-tap chunk at 2.4 GHz per workgroup slot)\n", wgs_per_cu, MODE, NM, CLASSES, ms, \
                (double) grid * 4 * chunks * (CLASSES == 5 ? 13 : CLASSES == 4 ? 10 : 1) * 65536.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / chunks / wgs_per_cu); } while (0)
        RUN (0, 5, "registers"); RUN (1, 5, "LDS reads"); RUN (2, 5, "LDS reads + staging writes + barrier");
        RUN (0, 4, "registers"); RUN (1, 4, "LDS reads"); RUN (2, 4, "LDS reads + staging writes + barrier");
        RUN (1, 1, "LDS reads, one product"); RUN (2, 1, "LDS reads + staging writes + barrier, one product"); RUN (3, 1, "LDS reads + barrier (8 waves), one product"); RUN (3, 5, "LDS reads + barrier (8 waves)");
    }
    {
        double ms = timeit ([&] { hipLaunchKernelGGL ((k_shape<false>), dim3 (512), dim3 (512), 0, 0, out, chunks); });
        printf ("two 8-wave workgroups per CU, 9 products per chunk, one operand set : %7.3f ms  %6.0f cycles per chunk and tile at 2.4 GHz\n", ms, ms * 1e-3 * 2.4e9 / chunks / 2);
        ms = timeit ([&] { hipLaunchKernelGGL ((k_shape<true>), dim3 (256), dim3 (768), 0, 0, out, chunks); });
        printf ("one 12-wave workgroup per CU (two tiles), 9 products, two operand sets : %7.3f ms  %6.0f cycles per chunk and tile at 2.4 GHz\n", ms, ms * 1e-3 * 2.4e9 / chunks / 2);
    }
    return 0;
}
