// Micro-benchmark: cycles per DEPENDENT vector instruction for a lone wave (one wave per SIMD, as the decimator's serial wave runs), and per
// independent one — the budget of a serial recurrence on gfx950.  Each kernel runs a chain of N instructions of one kind and reports
// (s_memtime ticks) / N.   build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_chain.hip -o tools/micro/valu_chain
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8 (REP8 (x))

template <int KIND>
__global__ __launch_bounds__ (64) void k_chain (float *out, unsigned long long *ticks, float seed, float k2)
{
    float a = seed + threadIdx.x, b = k2, c = seed * 0.5f, d = 1.0f + seed;
    float e0 = a, e1 = b, e2 = c, e3 = d;
    int i0 = 0;
    unsigned long long t0 = __builtin_readcyclecounter ();
    for (int it = 0; it < 64; ++it) {
        if (KIND == 0) { REP64 (asm volatile ("v_add_f32 %0, %0, %1" : "+v" (a) : "v" (b));) }
        if (KIND == 1) { REP64 (asm volatile ("v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v" (a) : "v" (b));) }   // dpp source independent
        if (KIND == 2) { REP64 (asm volatile ("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v" (a) : "v" (b));) }   // dpp source = chain
        if (KIND == 3) { REP64 (asm volatile ("v_mul_f32 %0, %0, %1" : "+v" (a) : "v" (b));) }
        if (KIND == 4) { REP64 (asm volatile ("v_cvt_rpi_i32_f32 %0, %0\n\tv_cvt_f32_i32 %0, %0" : "+v" (a));) }     // two instructions per rep
        if (KIND == 5) { REP64 (asm volatile ("v_cndmask_b32 %0, %0, %1, vcc" : "+v" (a) : "v" (b) : );) }
        if (KIND == 6) { REP64 (asm volatile ("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v" (e0), "+v" (e1), "+v" (e2), "+v" (e3) : "v" (b));) }   // four independent chains: 4 instructions per rep
        if (KIND == 7) { typedef float f2 __attribute__ ((ext_vector_type (2))); f2 p = { a, c }, q = { b, d };
                         REP64 (asm volatile ("v_pk_mul_f32 %0, %0, %1" : "+v" (p) : "v" (q));) a = p.x + p.y; }
        if (KIND == 8) { REP64 (asm volatile ("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v" (a) : "v" (b));) }   // as the compiler emits it: s_nop 1 before each
        if (KIND == 9) { REP64 (asm volatile ("v_sub_f32 %0, %1, %0\n\tv_add_f32 %0, %0, %2\n\tv_cvt_rpi_i32_f32 %0, %0\n\tv_cvt_f32_i32 %0, %0" : "+v" (a) : "v" (b), "v" (c));) }   // 4 per rep, mixed
        if (KIND == 10) { REP64 (asm volatile ("v_fma_f32 %0, %0, %1, %2" : "+v" (a) : "v" (b), "v" (c));) }
        if (KIND == 11) { REP64 (asm volatile ("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1" : "+v" (i0));) }
    }
    unsigned long long t1 = __builtin_readcyclecounter ();
    out [blockIdx.x * 64 + threadIdx.x] = a + e0 + e1 + e2 + e3 + (float) i0;
    if (threadIdx.x == 0) ticks [blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run (const char *what, int per_rep)
{
    float *out; unsigned long long *ticks, h [1024];
    hipMalloc (&out, 1024 * 64 * 4); hipMalloc (&ticks, 1024 * 8);
    hipLaunchKernelGGL (k_chain<KIND>, dim3 (8), dim3 (64), 0, 0, out, ticks, 1.0f, 1.000001f);
    hipLaunchKernelGGL (k_chain<KIND>, dim3 (8), dim3 (64), 0, 0, out, ticks, 1.0f, 1.000001f);
    hipDeviceSynchronize ();
    hipMemcpy (h, ticks, 8 * 8, hipMemcpyDeviceToHost);
    // s_memtime / readcyclecounter ticks at 100 MHz on this part?  report raw ticks and per instruction; calibrate against wall time below
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    hipEventRecord (e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL (k_chain<KIND>, dim3 (8), dim3 (64), 0, 0, out, ticks, 1.0f, 1.000001f);
    hipEventRecord (e1); hipEventSynchronize (e1);
    float ms; hipEventElapsedTime (&ms, e0, e1);
    const double n = 64.0 * 64 * per_rep;
    printf ("%-64s %8.2f ticks/instr   %7.2f ns/instr by wall clock (20 launches, incl. ~launch overhead)\n", what, (double) h [0] / n, ms * 1e6 / 20 / n);
    hipFree (out); hipFree (ticks);
}

int main ()
{
    run<0> ("v_add_f32, dependent", 1);
    run<6> ("v_add_f32, four independent chains", 4);
    run<10> ("v_fma_f32, dependent", 1);
    run<3> ("v_mul_f32, dependent", 1);
    run<4> ("v_cvt_rpi_i32_f32 + v_cvt_f32_i32, dependent", 2);
    run<5> ("v_cndmask_b32 (vcc), dependent", 1);
    run<7> ("v_pk_mul_f32, dependent", 1);
    run<1> ("v_add_f32_dpp quad_perm, chain in src1 (no nop)", 1);
    run<8> ("s_nop 1 + v_add_f32_dpp, chain in src1", 1);
    run<2> ("s_nop 1 + v_add_f32_dpp, chain in the dpp source", 1);
    run<11> ("v_mov_b32_dpp + s_nop 1, dependent", 1);
    run<9> ("sub, add, cvt_rpi, cvt (the decimator's head), dependent", 4);
    return 0;
}
