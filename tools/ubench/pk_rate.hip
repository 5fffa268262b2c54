// Microbenchmark: issue rate of packed FP32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against plain v_fma_f32 on gfx950,
// written with inline asm so that the compiler can neither pack the scalar form (SLP) nor split the packed one.
// (tools/ubench/valu_rate.hip compared "8 fmaf" with "8 packed fma" under plain -O3: the 8 fmaf were SLP-packed into 4
// v_pk_fma_f32, so the packed form looked half rate.  It is not.)
// Build: hipcc -O3 --offload-arch=gfx950 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: 8 independent v_fma_f32 per trip.  1: 8 independent v_pk_fma_f32.  2: 8 v_pk_mul_f32.  3: 8 v_pk_add_f32.
// 4: 8 v_pk_fma_f32 with op_sel broadcast of the low half of src1.  5: 4 v_fma_f32 + 4 v_pk_fma_f32 interleaved.
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed)
{
    float a[8];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; p[i] = f2{a[i], a[i] + 0.5f}; }
    const float m = 1.0000001f, c = 1e-9f;
    const f2 pm = {m, m}, pc = {c, c};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pm), "v"(pc));
                if (MODE == 5) {
                    if (i & 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu)
{
    float* out;
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks * 4.0 / 1024.0, insts = 64.0 * iters;
    printf("%-40s %d blk/CU: %8.3f ms -> %.2f SIMD-cycles per wave-instruction at 2.4 GHz\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / (waves_per_simd * insts));
    (void)hipFree(out);
}

int main()
{
    for (int b : {1, 2, 4, 7}) {
#define ALL(B) run<0>("v_fma_f32", B); run<1>("v_pk_fma_f32", B); run<2>("v_pk_mul_f32", B); run<3>("v_pk_add_f32", B); \
               run<4>("v_pk_fma_f32 op_sel", B); run<5>("v_fma_f32 / v_pk_fma_f32 alternating", B);
        if (b == 1) { ALL(1) } else if (b == 2) { ALL(2) } else if (b == 4) { ALL(4) } else { ALL(7) }
    }
    return 0;
}
