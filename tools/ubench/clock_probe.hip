// Microbenchmark: the shader clock a VALU-bound kernel actually runs at (s_memtime counts shader cycles, s_memrealtime a
// constant 100 MHz), for plain and packed FP32 streams at full occupancy on every CU.
// Build: hipcc -O3 --offload-arch=gfx950 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* t, int iters, float seed)
{
    float a[8];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; p[i] = f2{a[i], a[i] + 0.5f}; }
    const float m = 1.0000001f, c = 1e-9f;
    const f2 pm = {m, m}, pc = {c, c};
    const unsigned long long c0 = __builtin_readcyclecounter();  // s_memtime
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
                if (MODE == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
                if (MODE == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int blocks_per_cu, int iters)
{
    float* out;
    unsigned long long* t;
    const int blocks = 256 * blocks_per_cu;
    (void)hipMalloc(&out, blocks * 256 * 4);
    (void)hipMalloc(&t, blocks * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, t, 100, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, t, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), t, blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; b++) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
    const double mhz = cyc / real * 100.0, insts = 64.0 * iters;
    printf("%-16s %d blk/CU %6d trips: %8.3f ms  shader clock %7.1f MHz  %.2f shader cycles per wave-instruction per SIMD (%.2f at a nominal 2400 MHz)\n", name,
           blocks_per_cu, iters, ms, mhz, cyc / blocks / insts / (blocks_per_cu), ms * 1e-3 * 2.4e9 / (blocks_per_cu * insts));
    (void)hipFree(out); (void)hipFree(t);
}

int main()
{
    for (int iters : {2000, 40000}) {
        run<0>("v_fma_f32", 7, iters); run<1>("v_pk_fma_f32", 7, iters); run<2>("v_cndmask_b32", 7, iters); run<3>("v_add_u32", 7, iters);
        run<0>("v_fma_f32", 2, iters); run<1>("v_pk_fma_f32", 2, iters);
    }
    return 0;
}
