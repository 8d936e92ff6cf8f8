// graph_probe.hip — would hipGraph shorten a mid-sized device-resident call?  Such a call is two or three dependent kernels on one stream
// (prepare / staging, FIR): this probe replays that shape with spin kernels of the measured durations — as plain stream launches, as one
// captured graph relaunched, and as the graph with every kernel node's parameters rewritten before each launch (what a resampler call
// would need: its arguments change from call to call) — and reports the host's enqueue time and the device's time per iteration.
//   hipcc --offload-arch=gfx950 -O2 -o graph_probe graph_probe.hip && ./graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Args { long long ticks; int pad [96]; };            // ~400 bytes by value, like ArtFirArgs

__global__ void spin (Args a, unsigned long long *sink)
{
    const long long t0 = (long long) __builtin_amdgcn_s_memrealtime ();
    while ((long long) __builtin_amdgcn_s_memrealtime () - t0 < a.ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = (unsigned long long) t0;
}

static double now () { return std::chrono::duration<double> (std::chrono::steady_clock::now ().time_since_epoch ()).count (); }

int main ()
{
    hipStream_t st; (void) hipStreamCreateWithFlags (&st, hipStreamNonBlocking);
    unsigned long long *sink; (void) hipMalloc (&sink, 8);
    const int N = 400;
    const double us [3] = { 5.0, 4.0, 20.0 };
    for (int kernels = 2; kernels <= 3; ++kernels) {
        Args a [3];
        for (int k = 0; k < 3; ++k) { a [k].ticks = (long long)(us [k + (3 - kernels)] * 100.0); }      // s_memrealtime: 100 MHz
        auto enqueue_plain = [&] () { for (int k = 0; k < kernels; ++k) hipLaunchKernelGGL (spin, dim3 (256), dim3 (256), 0, st, a [k], sink); };
        // warm
        for (int i = 0; i < 50; ++i) enqueue_plain ();
        (void) hipStreamSynchronize (st);
        double t0 = now ();
        for (int i = 0; i < N; ++i) enqueue_plain ();
        double t1 = now ();
        (void) hipStreamSynchronize (st);
        double t2 = now ();
        double sum = 0; for (int k = 0; k < kernels; ++k) sum += us [k + (3 - kernels)];
        printf ("%d kernels (%.0f us of work): stream launches   enqueue %6.2f us/iter   device %6.2f us/iter\n", kernels, sum, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
        // graph
        hipGraph_t graph; hipGraphExec_t exec;
        (void) hipStreamBeginCapture (st, hipStreamCaptureModeThreadLocal);
        enqueue_plain ();
        if (hipStreamEndCapture (st, &graph) != hipSuccess) { printf ("capture failed\n"); return 1; }
        if (hipGraphInstantiate (&exec, graph, nullptr, nullptr, 0) != hipSuccess) { printf ("instantiate failed\n"); return 1; }
        for (int i = 0; i < 50; ++i) (void) hipGraphLaunch (exec, st);
        (void) hipStreamSynchronize (st);
        t0 = now ();
        for (int i = 0; i < N; ++i) (void) hipGraphLaunch (exec, st);
        t1 = now ();
        (void) hipStreamSynchronize (st);
        t2 = now ();
        printf ("%d kernels (%.0f us of work): graph relaunched   enqueue %6.2f us/iter   device %6.2f us/iter\n", kernels, sum, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
        // graph with parameters rewritten per launch
        size_t n_nodes = 0; (void) hipGraphGetNodes (graph, nullptr, &n_nodes);
        std::vector<hipGraphNode_t> nodes (n_nodes); (void) hipGraphGetNodes (graph, nodes.data (), &n_nodes);
        std::vector<hipKernelNodeParams> params (n_nodes);
        bool ok = true;
        for (size_t k = 0; k < n_nodes; ++k) ok = ok && hipGraphKernelNodeGetParams (nodes [k], &params [k]) == hipSuccess;
        if (ok) {
            Args b = a [0]; unsigned long long *s2 = sink;
            void *kp [2] = { &b, &s2 };
            t0 = now ();
            for (int i = 0; i < N; ++i) {
                for (size_t k = 0; k < n_nodes; ++k) { b.ticks = params.size () ? a [k % 3].ticks : 0; hipKernelNodeParams p = params [k]; p.kernelParams = kp; (void) hipGraphExecKernelNodeSetParams (exec, nodes [k], &p); }
                (void) hipGraphLaunch (exec, st);
            }
            t1 = now ();
            (void) hipStreamSynchronize (st);
            t2 = now ();
            printf ("%d kernels (%.0f us of work): graph + new params enqueue %6.2f us/iter   device %6.2f us/iter\n", kernels, sum, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
        }
        (void) hipGraphExecDestroy (exec); (void) hipGraphDestroy (graph);
    }
    return 0;
}
