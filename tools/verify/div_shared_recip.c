/* div_shared_recip.c -- CPU check of the shared-reciprocal division used by dvo_math.h gaussian_fuse:
 *     y  = RN(1 / v)                      (recip_fast: v_rcp_f32 + two Newton steps, equal to the IEEE reciprocal for every float in
 *                                          [2^-100, 2^100]: dvo_selftest_reciprocal walks all 2^32 patterns on the device)
 *     q0 = a * y;  r0 = fma(-v, q0, a);  q1 = fma(r0, y, q0);  r1 = fma(-v, q1, a);  q2 = fma(r1, y, q1)
 * against the IEEE quotient a / v, for a, v with magnitudes in [2^-60, 2^60].  q1 is a faithful quotient, so r1 is exact and q2 is
 * the correctly rounded quotient (Markstein's theorem needs y correctly rounded and q1 faithful).  Walks random pairs, pairs built
 * to land next to representable quotients and rounding midpoints, and significand corner cases; prints the mismatch counts of q1
 * and q2 (q2 must be 0).      gcc -O2 -fopenmp -ffp-contract=off -mfma -o div_shared_recip div_shared_recip.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint64_t rng(uint64_t* s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
static inline float f_from(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t u_from(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float rnd_float(uint64_t* s, int emin, int emax)
{   /* random sign, exponent in [emin, emax], random significand (with a bias towards corner significands) */
    const uint64_t r = rng(s);
    uint32_t man = (uint32_t)(r & 0x7fffff);
    const int sel = (int)((r >> 23) & 15);
    if (sel == 0) man = 0; else if (sel == 1) man = 0x7fffff; else if (sel == 2) man &= 0xff; else if (sel == 3) man |= 0x7fff00;
    const int e = emin + (int)((r >> 27) % (uint64_t)(emax - emin + 1));
    return f_from((uint32_t)((r >> 63) << 31) | ((uint32_t)(e + 127) << 23) | man);
}

static inline void check(float a, float v, unsigned long long* bad1, unsigned long long* bad2)
{
    const float ref = a / v;
    const float y = 1.0f / v;
    const float q0 = a * y;
    const float r0 = fmaf(-v, q0, a);
    const float q1 = fmaf(r0, y, q0);
    const float r1 = fmaf(-v, q1, a);
    const float q2 = fmaf(r1, y, q1);
    if (u_from(q1) != u_from(ref)) (*bad1)++;
    if (u_from(q2) != u_from(ref)) {
        if ((*bad2)++ < 5) printf("MISMATCH a=%a v=%a ref=%a q2=%a\n", a, v, ref, q2);
    }
}

int main(int argc, char** argv)
{
    const unsigned long long n = argc > 1 ? strtoull(argv[1], NULL, 10) : 2000000000ull;
    unsigned long long bad1 = 0, bad2 = 0;
#pragma omp parallel reduction(+ : bad1, bad2)
    {
        uint64_t s = 0x9E3779B97F4A7C15ull ^ (uint64_t)(
#ifdef _OPENMP
            omp_get_thread_num() + 1
#else
            1
#endif
        ) * 0xD1B54A32D192ED03ull;
#pragma omp for schedule(static)
        for (unsigned long long i = 0; i < n; i++) {
            const int kind = (int)(i & 3);
            float a, v;
            v = rnd_float(&s, -60, 60);
            if (kind == 0) {                       /* independent operands */
                a = rnd_float(&s, -60, 60);
            } else {                               /* a = RN(q * v) moved by -2..2 ulps: the quotient sits next to q or to a midpoint */
                const float q = rnd_float(&s, -30, 30);
                a = q * v;
                int k = (int)(rng(&s) % 5) - 2;
                uint32_t u = u_from(a);
                u = (uint32_t)((int32_t)u + k);
                a = f_from(u);
                if (kind == 3) {                   /* ... or exactly at half an ulp of q times v */
                    const float qh = f_from(u_from(q) + 0);  /* keep q */
                    a = fmaf(qh, v, 0.5f * (f_from(u_from(q) + 1) - q) * v);
                }
                const float aa = fabsf(a);
                if (!(aa >= 0x1p-60f && aa <= 0x1p60f)) continue;
            }
            check(a, v, &bad1, &bad2);
        }
    }
    printf("pairs %llu: q1 (one correction) differs from a/v %llu times, q2 (two corrections) %llu times\n", n, bad1, bad2);
    return bad2 ? 1 : 0;
}
