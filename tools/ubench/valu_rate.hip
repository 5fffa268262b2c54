// Microbenchmark: VALU issue rate on gfx950 for plain vs packed FP32 FMA, and v_rcp / division sequences.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    const float m = 1.0000001f, c = 1e-9f;
    const f2 pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 8 independent v_fma_f32
            a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
            a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
        } else if (MODE == 1) {  // 8 independent v_pk_fma_f32
            p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
            p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
            p4 = __builtin_elementwise_fma(p4, pm, pc); p5 = __builtin_elementwise_fma(p5, pm, pc);
            p6 = __builtin_elementwise_fma(p6, pm, pc); p7 = __builtin_elementwise_fma(p7, pm, pc);
        } else if (MODE == 2) {  // 8 IEEE divisions
            a0 = c / a0 + m; a1 = c / a1 + m; a2 = c / a2 + m; a3 = c / a3 + m;
            a4 = c / a4 + m; a5 = c / a5 + m; a6 = c / a6 + m; a7 = c / a7 + m;
        } else {  // 8 v_cndmask + v_cmp
            a0 = a0 > m ? a1 : a0 + c; a1 = a1 > m ? a2 : a1 + c; a2 = a2 > m ? a3 : a2 + c; a3 = a3 > m ? a4 : a3 + c;
            a4 = a4 > m ? a5 : a4 + c; a5 = a5 > m ? a6 : a5 + c; a6 = a6 > m ? a7 : a6 + c; a7 = a7 > m ? a0 : a7 + c;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, double ops_per_iter)
{
    float* out;
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 4.0, insts = waves * iters * ops_per_iter;
    // cycles per wave-instruction per SIMD at 2.4 GHz: time * 2.4e9 * 1024 SIMDs / insts
    printf("%-28s %d blk/CU: %8.3f ms  -> %.2f SIMD-cycles per wave-instruction (at 2.4 GHz), %.1f T lane-ops/s\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 * 1024 / insts, insts * 64 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
    for (int b : {1, 2, 4, 8}) {
        if (b == 1) { run<0>("v_fma_f32 x8", 1, 8); run<1>("v_pk_fma_f32 x8", 1, 8); run<2>("IEEE div x8 (+add)", 1, 8); run<3>("cmp+cndmask x8", 1, 8); }
        if (b == 2) { run<0>("v_fma_f32 x8", 2, 8); run<1>("v_pk_fma_f32 x8", 2, 8); }
        if (b == 4) { run<0>("v_fma_f32 x8", 4, 8); run<1>("v_pk_fma_f32 x8", 4, 8); run<2>("IEEE div x8 (+add)", 4, 8); run<3>("cmp+cndmask x8", 4, 8); }
        if (b == 8) { run<0>("v_fma_f32 x8", 8, 8); run<1>("v_pk_fma_f32 x8", 8, 8); }
    }
    return 0;
}
