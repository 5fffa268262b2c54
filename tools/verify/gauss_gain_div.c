// Exhaustive check behind gauss_gain() in csrc/dvo_math.h: the reference computes 0.5f + (m / 0.8f) * 0.5f for m < 0.8f
// (gaussian.cpp:20).  An IEEE float division costs the GPU ~10 instructions; for the constant divisor 0.8f the quotient
// q1 = fma(fma(-q0, 0.8f, m), 1.25f, q0), q0 = m * 1.25f (1.25f = RN(1 / 0.8f)) is the correctly rounded m / 0.8f for EVERY float
// with 1e-30 < |m| < 1e30, and for |m| <= 1e-30 (zeros, denormals) both forms give a gain of exactly 0.5f.  This program walks
// all 2^32 bit patterns and compares the gains bit for bit; the kernels take the literal division only for |m| >= 1e30.
//   gcc -O2 -fopenmp -ffp-contract=off -o gauss_gain_div gauss_gain_div.c -lm && ./gauss_gain_div [stride]   (~20 s on 8 cores;
//   stride > 1 visits every stride-th bit pattern: tests/test_cabi_and_host.py runs it with 5)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline float f_of(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t u_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(int argc, char** argv)
{
    const long long stride = argc > 1 && atoll(argv[1]) > 0 ? atoll(argv[1]) : 1;
    unsigned long long bad = 0, tested = 0, bad_q = 0;
#pragma omp parallel for reduction(+ : bad, tested, bad_q) schedule(static)
    for (long long k = 0; k < (1LL << 32); k += stride) {
        const float m = f_of((uint32_t)k);
        if (!(m < 0.8f)) continue;               // the other branch (and NaN) returns 1.0f without dividing
        if (!(fabsf(m) < 1e30f)) continue;       // literal division in the kernels
        tested++;
        volatile float q_ref = m / 0.8f;
        const float gain_ref = 0.5f + q_ref * 0.5f;
        const float q0 = m * 1.25f;
        const float rem = fmaf(-q0, 0.8f, m);
        const float q1 = fmaf(rem, 1.25f, q0);
        const float gain = 0.5f + q1 * 0.5f;
        if (u_of(gain) != u_of(gain_ref)) bad++;
        if (fabsf(m) > 1e-30f && u_of(q1) != u_of(q_ref)) bad_q++;
    }
    printf("tested %llu values: %llu gains differ, %llu quotients differ in the normal range\n", tested, bad, bad_q);
    return bad || bad_q;
}
