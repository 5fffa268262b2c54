// Microbenchmark: issue cost of the VALU instruction kinds k_track_gn is made of, one kind per kernel, 8 independent
// register chains per wave, 7 waves per SIMD on every CU (the occupancy of the real kernel).  Reports wall-clock SIMD cycles per
// wave-instruction at a nominal 2.4 GHz, relative to v_fma_f32.
// Build: hipcc -O3 --offload-arch=gfx950 inst_cost.hip -o inst_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define KERNEL(NAME, ASM, ...)                                                                              \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters, float seed)                          \
    {                                                                                                       \
        float a[8], b[8];                                                                                   \
        f2 p[8]; double d[8];                                                                               \
        _Pragma("unroll") for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; b[i] = a[i] * 0.5f; p[i] = f2{a[i], b[i]}; d[i] = a[i]; } \
        const float m = 1.0000001f, c = 1e-9f;                                                              \
        const f2 pm = {m, m}, pc = {c, c}; const double dm = m, dc = c; (void)dm; (void)dc;                                                                 \
        unsigned long long msk = seed > 0.5f ? 0x5555555555555555ull : 0x3333333333333333ull;               \
        int iters2 = iters; (void)m; (void)c; (void)pm; (void)pc; (void)msk; (void)iters2;                                                    \
        for (int it = 0; it < iters; it++) {                                                                \
            _Pragma("unroll") for (int rep = 0; rep < 8; rep++) {                                           \
                _Pragma("unroll") for (int i = 0; i < 8; i++) { asm volatile(ASM : __VA_ARGS__); }          \
            }                                                                                               \
        }                                                                                                   \
        float s = 0;                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; i++) s += a[i] + b[i] + p[i].x + p[i].y + (float)d[i];                   \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                            \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2", "+v"(a[i]) : "v"(m), "v"(c))
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2", "+v"(a[i]) : "v"(m), "v"(b[i]))
KERNEL(k_mul, "v_mul_f32 %0, %0, %1", "+v"(a[i]) : "v"(m))
KERNEL(k_add, "v_add_f32 %0, %0, %1", "+v"(a[i]) : "v"(c))
KERNEL(k_pkfma, "v_pk_fma_f32 %0, %0, %1, %2", "+v"(p[i]) : "v"(pm), "v"(pc))
KERNEL(k_pkmul, "v_pk_mul_f32 %0, %0, %1", "+v"(p[i]) : "v"(pm))
KERNEL(k_cnd_vcc, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(a[i]) : "v"(b[i]))
KERNEL(k_cnd_sgpr, "v_cndmask_b32 %0, %0, %1, %2", "+v"(a[i]) : "v"(b[i]), "s"(msk))
KERNEL(k_cnd_zero, "v_cndmask_b32 %0, 0, %0, %1", "+v"(a[i]) : "s"(msk))
KERNEL(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1", : "v"(a[i]), "v"(b[i]) : "vcc")
KERNEL(k_cmp_sgpr, "v_cmp_lt_f32 %0, %1, %2", "=s"(msk) : "v"(a[i]), "v"(b[i]))
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2", "+v"(a[i]) : "v"(m), "v"(b[i]))
KERNEL(k_mov, "v_mov_b32 %0, %1", "=v"(a[i]) : "v"(b[i]))
KERNEL(k_addu, "v_add_u32 %0, %0, %1", "+v"(a[i]) : "v"(b[i]))
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1", "+v"(a[i]) : "v"(b[i]))
KERNEL(k_cvt_i, "v_cvt_i32_f32 %0, %1", "=v"(a[i]) : "v"(b[i]))
KERNEL(k_cvt_f, "v_cvt_f32_i32 %0, %1", "=v"(a[i]) : "v"(b[i]))
KERNEL(k_rcp, "v_rcp_f32 %0, %1", "=v"(a[i]) : "v"(b[i]))
KERNEL(k_fract, "v_fract_f32 %0, %1", "=v"(a[i]) : "v"(b[i]))
KERNEL(k_dpp, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(a[i]) : )
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1", "+v"(a[i]), "+v"(b[i]) : )
KERNEL(k_salu, "s_add_u32 %0, %0, 1", "+s"(iters2) : )
KERNEL(k_fma_sgpr, "v_fma_f32 %0, %0, %1, %2", "+v"(a[i]) : "s"(m), "v"(c))

KERNEL(k_cnd_e64vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc", "+v"(a[i]) : "v"(b[i]))
KERNEL(k_cmpcnd_vcc, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", "+v"(a[i]) : "v"(b[i]) : "vcc")
KERNEL(k_cmpcnd_sgpr, "v_cmp_lt_f32 %2, %0, %1\n v_cndmask_b32 %0, %0, %1, %2", "+v"(a[i]) : "v"(b[i]), "s"(msk))
KERNEL(k_cmpx2_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_fma_f32 %0, %0, %3, %0\n v_fma_f32 %1, %1, %3, %1\n v_cndmask_b32 %0, %0, %1, vcc", "+v"(a[i]), "+v"(b[i]) : "s"(msk), "v"(c) : "vcc")
KERNEL(k_fma3, "v_fma_f32 %0, %1, %2, %0", "+v"(a[i]) : "v"(b[i]), "v"(m))
KERNEL(k_fma_inl, "v_fma_f32 %0, %0, 2.0, %1", "+v"(a[i]) : "v"(c))
KERNEL(k_fma_neg, "v_fma_f32 %0, -%0, %1, %2", "+v"(a[i]) : "v"(m), "v"(c))
KERNEL(k_sub, "v_sub_f32 %0, %0, %1", "+v"(a[i]) : "v"(c))
KERNEL(k_mul_sgpr, "v_mul_f32 %0, %1, %0", "+v"(a[i]) : "s"(m))
KERNEL(k_pkfma_sgpr, "v_pk_fma_f32 %0, %0, %1, %2", "+v"(p[i]) : "s"(pm), "v"(pc))
KERNEL(k_pkadd, "v_pk_add_f32 %0, %0, %1", "+v"(p[i]) : "v"(pc))
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2", "+v"(a[i]) : "v"(b[i]), "v"(m))
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 2, %1", "+v"(a[i]) : "v"(b[i]))
KERNEL(k_and, "v_and_b32 %0, %0, %1", "+v"(a[i]) : "v"(b[i]))
KERNEL(k_max, "v_max_f32 %0, %0, %1", "+v"(a[i]) : "v"(b[i]))
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2", "+v"(a[i]) : "v"(b[i]), "v"(m))
KERNEL(k_rdfl, "v_readfirstlane_b32 %0, %1", "=s"(iters2) : "v"(a[i]))
KERNEL(k_sqrt, "v_sqrt_f32 %0, %1", "=v"(a[i]) : "v"(b[i]))
KERNEL(k_fma64, "v_fma_f64 %0, %0, %1, %2", "+v"(d[i]) : "v"(dm), "v"(dc))
KERNEL(k_cvtf64, "v_cvt_f64_f32 %0, %1", "=v"(d[i]) : "v"(a[i]))
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1", "+v"(a[i]) : "v"(b[i]))
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1", "+v"(a[i]) : "v"(b[i]))
KERNEL(k_mix_cvt, "v_fma_f32 %0, %0, %2, %3\n v_cvt_i32_f32 %1, %1", "+v"(a[i]), "+v"(b[i]) : "v"(m), "v"(c))
KERNEL(k_mix_min3, "v_fma_f32 %0, %0, %2, %3\n v_min3_f32 %1, %1, %2, %3", "+v"(a[i]), "+v"(b[i]) : "v"(m), "v"(c))
KERNEL(k_mix_cnd, "v_fma_f32 %0, %0, %2, %3\n v_cndmask_b32 %1, %1, %2, %4", "+v"(a[i]), "+v"(b[i]) : "v"(m), "v"(c), "s"(msk))
KERNEL(k_mix_pk, "v_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %4, %5", "+v"(a[i]), "+v"(p[i]) : "v"(m), "v"(c), "v"(pm), "v"(pc))
KERNEL(k_mix_sfma, "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %4, %3", "+v"(a[i]), "+v"(b[i]) : "v"(m), "v"(c), "s"(m))
KERNEL(k_mix3, "v_fma_f32 %0, %0, %2, %3\n v_mul_f32 %1, %1, %2\n v_cvt_i32_f32 %4, %4", "+v"(a[i]), "+v"(b[i]) : "v"(m), "v"(c), "v"(p[i].x))

template <class K>
double run(K kern, const char* name, double ref)
{
    float* out;
    const int blocks = 256 * 7, iters = 3000;
    (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 200, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / (7.0 * iters * 64.0);
    printf("%-34s %8.3f ms  %6.2f cycles per wave-instruction per SIMD  (%.2f x v_fma_f32)\n", name, ms, cyc, ref > 0 ? cyc / ref : 1.0);
    (void)hipFree(out);
    return cyc;
}

int main()
{
    run(k_fma, "warm-up", 0);
    for (int pass = 0; pass < 1; pass++) {
        const double r = run(k_fma, "v_fma_f32", 0);
        run(k_fmac, "v_fmac_f32", r); run(k_mul, "v_mul_f32", r); run(k_add, "v_add_f32", r); run(k_fma_sgpr, "v_fma_f32 (one SGPR source)", r);
        run(k_pkfma, "v_pk_fma_f32", r); run(k_pkmul, "v_pk_mul_f32", r);
        run(k_cnd_vcc, "v_cndmask_b32 (vcc)", r); run(k_cnd_sgpr, "v_cndmask_b32 (SGPR mask)", r); run(k_cnd_zero, "v_cndmask_b32 0, v, SGPR mask", r);
        run(k_cmp_vcc, "v_cmp_lt_f32 -> vcc", r); run(k_cmp_sgpr, "v_cmp_lt_f32 -> SGPR pair", r);
        run(k_min3, "v_min3_f32", r); run(k_mov, "v_mov_b32", r); run(k_addu, "v_add_u32", r); run(k_mul24, "v_mul_u32_u24", r);
        run(k_cvt_i, "v_cvt_i32_f32", r); run(k_cvt_f, "v_cvt_f32_i32", r); run(k_rcp, "v_rcp_f32", r); run(k_fract, "v_fract_f32", r);
        run(k_dpp, "v_add_f32 dpp row_shr:1", r); run(k_swap32, "v_permlane32_swap_b32", r);
        run(k_cnd_e64vcc, "v_cndmask_b32_e64 (vcc)", r); run(k_cmpcnd_vcc, "v_cmp -> vcc + v_cndmask vcc (PAIR)", r);
        run(k_cmpcnd_sgpr, "v_cmp -> SGPR + v_cndmask SGPR (PAIR)", r); run(k_cmpx2_cnd, "v_cmp vcc, 2 fma, v_cndmask vcc (4 inst)", r);
        run(k_fma3, "v_fma_f32 three different VGPRs", r); run(k_fma_inl, "v_fma_f32 inline constant", r); run(k_fma_neg, "v_fma_f32 neg modifier", r);
        run(k_sub, "v_sub_f32", r); run(k_mul_sgpr, "v_mul_f32 SGPR source", r); run(k_pkfma_sgpr, "v_pk_fma_f32 SGPR pair source", r);
        run(k_pkadd, "v_pk_add_f32", r); run(k_mad24, "v_mad_u32_u24", r); run(k_lshladd, "v_lshl_add_u32", r); run(k_and, "v_and_b32", r);
        run(k_max, "v_max_f32", r); run(k_med3, "v_med3_f32", r); run(k_rdfl, "v_readfirstlane_b32", r); run(k_sqrt, "v_sqrt_f32", r);
        run(k_fma64, "v_fma_f64", r); run(k_cvtf64, "v_cvt_f64_f32", r); run(k_ldexp, "v_ldexp_f32", r); run(k_mullo, "v_mul_lo_u32", r);
        run(k_mix_cvt, "PAIR v_fma + v_cvt_i32_f32", r); run(k_mix_min3, "PAIR v_fma + v_min3", r); run(k_mix_cnd, "PAIR v_fma + v_cndmask sgpr", r);
        run(k_mix_pk, "PAIR v_fma + v_pk_fma", r); run(k_mix_sfma, "PAIR v_fma + v_fma sgpr", r); run(k_mix3, "TRIPLE v_fma + v_mul + v_cvt", r);
    }
    return 0;
}
