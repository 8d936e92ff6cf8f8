// device_rt.hip — thin HIP runtime glue behind the C host layer (art_internal.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "art_internal.h"

static thread_local char g_err[256] = "no error";

static int fail (hipError_t e, const char *what)
{
    if (e == hipSuccess) return 0;
    snprintf (g_err, sizeof (g_err), "%s: %s", what, hipGetErrorString (e));
    return -1;
}

extern "C" {

const char *arthip_last_error (void) { return g_err; }
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
