// Microbenchmark: what does moving k_track_gn's accumulation (29 v_fma + the wave reduction, ~46 VALU per pixel) onto the
// matrix pipe cost in VALU issue slots?  Per "pixel" (one loop trip of a wave): NV independent v_fma_f32, then optionally
// an LDS transpose (2 ds_write_b128 + 8 ds_read_b32, wave-private region) feeding 8 v_mfma_f32_16x16x4_f32 on ONE accumulator.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_mix.hip -o mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: NV v_fma only.  MODE 1: NV v_fma + 8 MFMA fed from registers (no LDS).  MODE 2: NV v_fma + LDS transpose + 8 MFMA.
// MODE 3: 8 MFMA only.
template <int MODE, int NV>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[4][2][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x + i;
    const float m = 1.0000001f, c = 1e-9f;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int rd_base = (lane >> 3) * 8 + ((((lane & 7) >> 2) ^ ((lane >> 5) & 1)) << 2) + (lane & 3);
    for (int it = 0; it < iters; it++) {
        if (MODE != 3) {
#pragma unroll
            for (int j = 0; j < NV / 8; j++)
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = fmaf(a[i], m, c);
        }
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[i], acc, 0, 0, 0);
        }
        if (MODE == 2) {
            float* buf = lds[wave][it & 1];
            const int s = (lane >> 2) & 1;
            *reinterpret_cast<f4*>(buf + lane * 8 + 4 * (0 ^ s)) = f4{a[0], a[1], a[2], a[3]};
            *reinterpret_cast<f4*>(buf + lane * 8 + 4 * (1 ^ s)) = f4{a[4], a[5], a[6], a[7]};
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = buf[rd_base + 64 * i];
#pragma unroll
            for (int i = 0; i < 8; i++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i], v[i], acc, 0, 0, 0);
        }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NV>
void run(const char* name, int blocks_per_cu)
{
    float* out;
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // SIMD cycles per loop trip of one wave: time * 2.4e9 / (waves per SIMD * iters)
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    printf("%-44s %d blk/CU: %8.3f ms -> %7.1f SIMD-cycles per trip per wave (= %.1f v_fma equivalents)\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / (waves_per_simd * iters), ms * 1e-3 * 2.4e9 / (waves_per_simd * iters) / 4.0);
    hipFree(out);
}

int main()
{
    for (int b : {1, 4, 7}) {
        if (b == 1) {
            run<0, 208>("208 v_fma", 1); run<0, 160>("160 v_fma", 1); run<1, 160>("160 v_fma + 8 mfma16x16x4 (regs)", 1);
            run<2, 160>("160 v_fma + LDS transpose + 8 mfma", 1); run<3, 0>("8 mfma only", 1);
        } else if (b == 4) {
            run<0, 208>("208 v_fma", 4); run<0, 160>("160 v_fma", 4); run<1, 160>("160 v_fma + 8 mfma16x16x4 (regs)", 4);
            run<2, 160>("160 v_fma + LDS transpose + 8 mfma", 4); run<3, 0>("8 mfma only", 4);
        } else {
            run<0, 208>("208 v_fma", 7); run<0, 160>("160 v_fma", 7); run<1, 160>("160 v_fma + 8 mfma16x16x4 (regs)", 7);
            run<2, 160>("160 v_fma + LDS transpose + 8 mfma", 7); run<3, 0>("8 mfma only", 7);
        }
    }
    return 0;
}
