// device_rt.hip — thin HIP runtime glue behind the C host layer (art_internal.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "art_internal.h"

static thread_local char g_err[256] = "no error";

// Small transfers between page-locked host memory and HBM as a KERNEL (the GPU reads / writes the host pages over PCIe
// itself): for the few hundred kilobytes of an ART-sized block the copy engines' fixed cost per command (~10 us each way on
// this stack) is most of a call, a launch in the same stream costs 2-4 us.
__global__ void copy_words_kernel (uint4 *dst, const uint4 *src, size_t quads, unsigned int *dst_tail, const unsigned int *src_tail, int tail_words,
                                   unsigned int *dst2, const unsigned int *src2, int words2)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride) dst [i] = src [i];
    if (blockIdx.x == 0 && (int) threadIdx.x < tail_words) dst_tail [threadIdx.x] = src_tail [threadIdx.x];
    if (blockIdx.x == gridDim.x - 1)                      // a second, short range (a filter's state beside its samples)
        for (int i = threadIdx.x; i < words2; i += blockDim.x) dst2 [i] = src2 [i];
}

static int fail (hipError_t e, const char *what)
{
    if (e == hipSuccess) return 0;
    snprintf (g_err, sizeof (g_err), "%s: %s", what, hipGetErrorString (e));
    return -1;
}

extern "C" {

const char *arthip_last_error (void) { return g_err; }
// (the error string is per thread; a worker thread's is handed to the thread that reports it)
void arthip_set_last_error (const char *text) { snprintf (g_err, sizeof (g_err), "%s", text ? text : "no error"); }
const char *artamdVersion (void) { return ART_WIDE ? "artamd 0.1 (gfx950, 64-bit samples)" : "artamd 0.1 (gfx950)"; }

int arthip_device_count (void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount (&n);
    if (e != hipSuccess) { fail (e, "hipGetDeviceCount"); return 0; }
    if (n < 1) snprintf (g_err, sizeof (g_err), "hipGetDeviceCount returned 0 devices");
    return n;
}

int artamdDeviceCount (void) { return arthip_device_count (); }
// device memory for callers of the device-pointer entry points that bring no runtime of their own (tools/art_gpu.py: no
// torch import, a fraction of a second less start-up per file) — thin names over the runtime; asynchronous on `stream`
void *artamdDeviceAlloc (size_t bytes) { return arthip_malloc (bytes); }
void artamdDeviceFree (void *p) { arthip_free (p); }
int artamdUpload (void *d_dst, const void *h_src, size_t bytes, void *stream) { return arthip_h2d (d_dst, h_src, bytes, stream); }
int artamdDownload (void *h_dst, const void *d_src, size_t bytes, void *stream) { return arthip_d2h (h_dst, d_src, bytes, stream); }
int artamdDeviceZero (void *d_dst, size_t bytes, void *stream) { return arthip_zero (d_dst, bytes, stream); }
int artamdStreamSynchronize (void *stream) { return arthip_sync (stream); }
int arthip_current_device (void) { int d = 0; return hipGetDevice (&d) == hipSuccess ? d : -1; }

void *arthip_malloc (size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    if (fail (hipMalloc (&p, bytes), "hipMalloc")) return nullptr;
    return p;
}

void arthip_free (void *p) { if (p) (void) hipFree (p); }

int arthip_h2d (void *d, const void *s, size_t n, void *st) { return n ? fail (hipMemcpyAsync (d, s, n, hipMemcpyHostToDevice, (hipStream_t) st), "H2D") : 0; }
int arthip_d2h (void *d, const void *s, size_t n, void *st) { return n ? fail (hipMemcpyAsync (d, s, n, hipMemcpyDeviceToHost, (hipStream_t) st), "D2H") : 0; }
int arthip_d2d (void *d, const void *s, size_t n, void *st) { return n ? fail (hipMemcpyAsync (d, s, n, hipMemcpyDeviceToDevice, (hipStream_t) st), "D2D") : 0; }
int arthip_zero (void *d, size_t n, void *st) { return n ? fail (hipMemsetAsync (d, 0, n, (hipStream_t) st), "memset") : 0; }
int arthip_sync (void *st) { return fail (hipStreamSynchronize ((hipStream_t) st), "sync"); }

// rows of `width` 4-byte words, row starts dpitch / spitch words apart: a channel slice of an interleaved stream, moved by the
// GPU (the 2-D copy command is an order of magnitude slower on 16-byte rows); either side may live on a peer device
__global__ void slice_words_kernel (unsigned int *dst, size_t dpitch, const unsigned int *src, size_t spitch, int width, size_t rows)
{
    const size_t total = rows * (size_t) width, stride = (size_t) gridDim.x * blockDim.x;
    for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t r = e / width; const int w = (int)(e - r * width);
        dst [r * dpitch + w] = src [r * spitch + w];
    }
}

// the same for rows of `width` BYTES (packed PCM of a channel slice: 2 or 3 bytes per sample need not make whole words)
__global__ void slice_bytes_kernel (unsigned char *dst, size_t dpitch, const unsigned char *src, size_t spitch, int width, size_t rows)
{
    const size_t total = rows * (size_t) width, stride = (size_t) gridDim.x * blockDim.x;
    for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t r = e / width; const int w = (int)(e - r * width);
        dst [r * dpitch + w] = src [r * spitch + w];
    }
}

// dst / src: one of them page-locked host memory (hipHostMalloc: mapped, same address on the device), both 16-byte aligned,
// bytes a multiple of 4
int arthip_copy_by_kernel (void *dst, const void *src, size_t bytes, void *st) { return arthip_copy2_by_kernel (dst, src, bytes, nullptr, nullptr, 0, st); }

// ... plus a second short range (dst2 / src2: 4-byte aligned, bytes2 a multiple of 4) in the same launch
int arthip_copy2_by_kernel (void *dst, const void *src, size_t bytes, void *dst2, const void *src2, size_t bytes2, void *st)
{
    if (!bytes && !bytes2) return 0;
    const size_t quads = bytes / 16; const int tail = (int)((bytes % 16) / 4);
    unsigned int blocks = (unsigned int)((quads + 255) / 256);
    if (blocks > 512) blocks = 512;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL (copy_words_kernel, dim3 (blocks), dim3 (256), 0, (hipStream_t) st, (uint4 *) dst, (const uint4 *) src, quads,
                        (unsigned int *) dst + quads * 4, (const unsigned int *) src + quads * 4, tail,
                        (unsigned int *) dst2, (const unsigned int *) src2, (int)(bytes2 / 4));
    return fail (hipGetLastError (), "copy kernel");
}

// width_words 4-byte words per row, pitches in words
int arthip_slice_copy (void *dst, size_t dpitch_words, const void *src, size_t spitch_words, int width_words, size_t rows, void *st)
{
    if (!rows || width_words <= 0) return 0;
    const size_t total = rows * (size_t) width_words;
    unsigned int blocks = (unsigned int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL (slice_words_kernel, dim3 (blocks), dim3 (256), 0, (hipStream_t) st, (unsigned int *) dst, dpitch_words,
                        (const unsigned int *) src, spitch_words, width_words, rows);
    return fail (hipGetLastError (), "slice kernel");
}

// rows of `width` bytes, row starts dpitch / spitch bytes apart (whole words where everything is word-aligned)
int arthip_slice_copy_bytes (void *dst, size_t dpitch, const void *src, size_t spitch, int width, size_t rows, void *st)
{
    if (!rows || width <= 0) return 0;
    if (!((dpitch | spitch | (size_t) width | (size_t)(uintptr_t) dst | (size_t)(uintptr_t) src) & 3))
        return arthip_slice_copy (dst, dpitch / 4, src, spitch / 4, width / 4, rows, st);
    const size_t total = rows * (size_t) width;
    unsigned int blocks = (unsigned int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL (slice_bytes_kernel, dim3 (blocks), dim3 (256), 0, (hipStream_t) st, (unsigned char *) dst, dpitch,
                        (const unsigned char *) src, spitch, width, rows);
    return fail (hipGetLastError (), "slice kernel");
}

int arthip_set_device (int device) { return fail (hipSetDevice (device), "hipSetDevice"); }

void *arthip_stream_create (void)
{
    // shards of one context run side by side: their streams must not serialise against the null stream
    hipStream_t s = nullptr;
    return fail (hipStreamCreateWithFlags (&s, hipStreamNonBlocking), "hipStreamCreate") ? nullptr : (void *) s;
}
void arthip_stream_destroy (void *s) { if (s) (void) hipStreamDestroy ((hipStream_t) s); }

// strided rows between any two address spaces (host <-> device, device <-> device, across devices under UVA):
// `rows` rows of `width` bytes, row starts `dpitch` / `spitch` bytes apart
int arthip_copy2d (void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, void *st)
{
    if (!width || !rows) return 0;
    return fail (hipMemcpy2DAsync (dst, dpitch, src, spitch, width, rows, hipMemcpyDefault, (hipStream_t) st), "copy2D");
}
int arthip_copy (void *dst, const void *src, size_t n, void *st) { return n ? fail (hipMemcpyAsync (dst, src, n, hipMemcpyDefault, (hipStream_t) st), "copy") : 0; }

// page-locked host memory for the staging buffers of the host-pointer entry points
void *arthip_host_alloc (size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    return fail (hipHostMalloc (&p, bytes, hipHostMallocDefault), "hipHostMalloc") ? nullptr : p;
}
void arthip_host_free (void *p) { if (p) (void) hipHostFree (p); }

// ordering events (no timing): `waiter` stream continues only after everything recorded so far on `signaller`
void *arthip_order_event_create (void)
{
    hipEvent_t e = nullptr;
    return fail (hipEventCreateWithFlags (&e, hipEventDisableTiming), "hipEventCreate") ? nullptr : (void *) e;
}
int arthip_stream_wait_event (void *st, void *ev) { return fail (hipStreamWaitEvent ((hipStream_t) st, (hipEvent_t) ev, 0), "hipStreamWaitEvent"); }
int arthip_event_sync (void *ev) { return fail (hipEventSynchronize ((hipEvent_t) ev), "hipEventSynchronize"); }

// let `device` read and write memory that lives on `peer` (xGMI); returns 1 when it can (same device, or peer access enabled now
// or before), 0 when the platform refuses (no P2P route: IOMMU, containers, mixed topology) — the caller must then keep the two
// sides on one device: a kernel dereferencing unreachable peer memory is a GPU page fault, not an error code
int arthip_enable_peer (int device, int peer)
{
    int can = 0, prev = 0;
    if (device == peer) return 1;
    if (hipDeviceCanAccessPeer (&can, device, peer) != hipSuccess || !can) { (void) hipGetLastError (); return 0; }
    if (hipGetDevice (&prev) != hipSuccess) return 0;
    int ok = 0;
    if (hipSetDevice (device) == hipSuccess) {
        const hipError_t e = hipDeviceEnablePeerAccess (peer, 0);
        ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
        (void) hipGetLastError ();
    }
    (void) hipSetDevice (prev);
    return ok;
}

void *arthip_event_create (void)
{
    // timing events: no system-scope fence when they fire (nothing on the host reads device memory off them), which
    // keeps the dispatch gap an event costs on the stream to a few microseconds
    hipEvent_t e = nullptr;
    return fail (hipEventCreateWithFlags (&e, hipEventDisableSystemFence), "hipEventCreate") ? nullptr : (void *) e;
}
void arthip_event_destroy (void *e) { if (e) (void) hipEventDestroy ((hipEvent_t) e); }
int arthip_event_record (void *e, void *st) { return fail (hipEventRecord ((hipEvent_t) e, (hipStream_t) st), "hipEventRecord"); }
float arthip_event_elapsed_ms (void *a, void *b)
{
    float ms = 0.0f;
    if (fail (hipEventSynchronize ((hipEvent_t) b), "hipEventSynchronize")) return -1.0f;
    if (fail (hipEventElapsedTime (&ms, (hipEvent_t) a, (hipEvent_t) b), "hipEventElapsedTime")) return -1.0f;
    return ms;
}

}
