// Does v_cvt_rpi_i32_f32 compute floor (x + 0.5) EXACTLY (as if in infinite precision), i.e. the same as the decimator's
// (int) floor ((double) x + 0.5) for every float in the range of interest?  Checked here over random and adversarial inputs.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k (const float *x, int *y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { int r; asm volatile ("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x [i])); y [i] = r; }
}
int main () {
    std::vector<float> h;
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < 4000000; ++i) {                       // random magnitudes up to 2^24
        const double m = ldexp ((double)(rnd () >> 11) / 9007199254740992.0, (int)(rnd () % 50) - 25);
        h.push_back ((float)((rnd () & 1) ? m : -m));
    }
    for (int kk = -70000; kk <= 70000; ++kk)                  // around every half-integer of the 16/17-bit range
        for (int u = -3; u <= 3; ++u) { float v = (float) kk + 0.5f; for (int t = 0; t < (u < 0 ? -u : u); ++t) v = nextafterf (v, u < 0 ? -1e30f : 1e30f); h.push_back (v); }
    for (int e = -30; e <= 24; ++e)                            // around +-(2^e) and +-(0.5 - tiny)
        for (int u = -4; u <= 4; ++u) for (int sg = -1; sg <= 1; sg += 2) {
            float v = sg * ldexpf (1.0f, e); for (int t = 0; t < (u < 0 ? -u : u); ++t) v = nextafterf (v, u < 0 ? -1e30f : 1e30f); h.push_back (v);
            float w = sg * 0.5f; for (int t = 0; t < (u < 0 ? -u : u); ++t) w = nextafterf (w, u < 0 ? -1e30f : 1e30f); h.push_back (w);
        }
    const int n = (int) h.size ();
    float *dx; int *dy; hipMalloc (&dx, n * 4); hipMalloc (&dy, n * 4);
    hipMemcpy (dx, h.data (), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL (k, dim3 ((n + 255) / 256), dim3 (256), 0, 0, dx, dy, n);
    std::vector<int> r (n); hipMemcpy (r.data (), dy, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) {
        const double want = floor ((double) h [i] + 0.5);
        if (fabs (want) < 2147483000.0 && (double) r [i] != want) { if (bad++ < 10) printf ("x = %.9g (%a): rpi %d, floor(x+0.5) %.0f\n", h [i], h [i], r [i], want); }
    }
    printf ("%d inputs, %ld mismatches\n", n, bad);
    return bad != 0;
}
