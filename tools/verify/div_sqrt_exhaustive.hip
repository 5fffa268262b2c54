// div_sqrt_exhaustive.hip -- which short instruction sequences reproduce the IEEE (correctly rounded) float division and square root on
// gfx950, bit for bit, for EVERY operand?  Decided by enumeration on the device:
//   division: all 2^23 x 2^23 mantissa pairs a, b in [1, 2) (the sequences consist of v_rcp_f32 -- whose relative behaviour was
//             enumerated over all floats, dvo_selftest_reciprocal -- and IEEE multiplies / FMAs, which are scale invariant as long as
//             nothing over- or underflows: the exponents of a and b add nothing to check);
//   sqrt:     all positive floats in [2^-100, 2^100].
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o div_sqrt_exhaustive div_sqrt_exhaustive.hip
//   ./div_sqrt_exhaustive [b_chunks_to_run (of 64; default 64)]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

template <int V>
__device__ __forceinline__ float div_seq(float a, float b)
{
    float y = __builtin_amdgcn_rcpf(b);
    if (V == 1 || V == 2) y = fmaf(fmaf(-b, y, 1.0f), y, y);      // one Newton step
    if (V == 2) y = fmaf(fmaf(-b, y, 1.0f), y, y);                // two: recip_fast(), = RN(1 / b)
    float q = a * y;
    q = fmaf(fmaf(-b, q, a), y, q);
    if (V == 3) q = fmaf(fmaf(-b, q, a), y, q);                   // raw reciprocal, two corrections of the quotient
    return q;
}

template <int V>
__global__ void __launch_bounds__(256) k_div(unsigned mb0, unsigned long long* out)
{
    const unsigned mb = mb0 + blockIdx.x * 256u + threadIdx.x;
    const float b = __uint_as_float(0x3f800000u | mb);
    unsigned long long bad = 0, first = ~0ull;
    for (unsigned ma = 0; ma < (1u << 23); ma++) {
        const float a = __uint_as_float(0x3f800000u | ma);
        const float want = a / b, got = div_seq<V>(a, b);
        if (__float_as_uint(want) != __float_as_uint(got)) { bad++; const unsigned long long k = ((unsigned long long)mb << 23) | ma; if (k < first) first = k; }
    }
    if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], first); }
}

template <int V>
__device__ __forceinline__ float sqrt_seq(float x)
{
    if (V == 3) {
        float s = __builtin_amdgcn_sqrtf(x);
        const float h = 0.5f * __builtin_amdgcn_rcpf(s);
        s = fmaf(fmaf(-s, s, x), h, s);
        return s;
    }
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    const float h = 0.5f * y;
    s = fmaf(fmaf(-s, s, x), h, s);
    if (V == 2) s = fmaf(fmaf(-s, s, x), h, s);
    return s;
}

template <int V>
__global__ void __launch_bounds__(256) k_sqrt(unsigned long long* out)
{
    unsigned long long bad = 0, first = ~0ull, n = 0;
    for (unsigned long long p = blockIdx.x * 256ull + threadIdx.x; p < (1ull << 31); p += (unsigned long long)gridDim.x * 256ull) {
        const float x = __uint_as_float((unsigned)p);
        if (!(x >= 7.888609052210118e-31f && x <= 1.2676506002282294e30f)) continue;
        n++;
        const float want = sqrtf(x), got = sqrt_seq<V>(x);
        if (__float_as_uint(want) != __float_as_uint(got)) { bad++; if (p < first) first = p; }
    }
    atomicAdd(&out[2], n);
    if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], first); }
}

// reciprocal with ONE Newton step (recip_fast() has two): exact for every float in [2^-100, 2^100]?
__global__ void __launch_bounds__(256) k_recip1(unsigned long long* out)
{
    unsigned long long bad = 0, first = ~0ull, n = 0;
    for (unsigned long long p = blockIdx.x * 256ull + threadIdx.x; p < (1ull << 31); p += (unsigned long long)gridDim.x * 256ull) {
        const float x = __uint_as_float((unsigned)p);
        if (!(x >= 7.888609052210118e-31f && x <= 1.2676506002282294e30f)) continue;
        n++;
        float y = __builtin_amdgcn_rcpf(x);
        y = fmaf(fmaf(-x, y, 1.0f), y, y);
        if (__float_as_uint(1.0f / x) != __float_as_uint(y)) { bad++; if (p < first) first = p; }
    }
    atomicAdd(&out[2], n);
    if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], first); }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int V>
static int run_div(int chunks, unsigned long long* d_out)
{
    unsigned long long h[3] = {0, ~0ull, 0};
    CK(hipMemcpy(d_out, h, sizeof h, hipMemcpyHostToDevice));
    for (int c = 0; c < chunks; c++) {   // 2^17 values of b per launch (x 2^23 values of a): well under a second each
        hipLaunchKernelGGL(k_div<V>, dim3(512), dim3(256), 0, nullptr, (unsigned)c << 17, d_out);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
    std::printf("division variant %d: %d/64 of the b mantissas x all a mantissas: %llu mismatches", V, chunks, h[0]);
    if (h[0]) std::printf(" (first: mb = 0x%06llx, ma = 0x%06llx)", h[1] >> 23, h[1] & 0x7fffff);
    std::printf("\n");
    std::fflush(stdout);
    return 0;
}

template <int V>
static int run_sqrt(unsigned long long* d_out)
{
    unsigned long long h[3] = {0, ~0ull, 0};
    CK(hipMemcpy(d_out, h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sqrt<V>, dim3(8192), dim3(256), 0, nullptr, d_out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
    std::printf("sqrt variant %d: %llu inputs in [2^-100, 2^100]: %llu mismatches", V, h[2], h[0]);
    if (h[0]) std::printf(" (first: 0x%08llx)", h[1]);
    std::printf("\n");
    std::fflush(stdout);
    return 0;
}

int main(int argc, char** argv)
{
    const int chunks = argc > 1 ? std::atoi(argv[1]) : 64;
    unsigned long long* d_out = nullptr;
    CK(hipMalloc(&d_out, 3 * sizeof(unsigned long long)));
    {
        unsigned long long h[3] = {0, ~0ull, 0};
        CK(hipMemcpy(d_out, h, sizeof h, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_recip1, dim3(8192), dim3(256), 0, nullptr, d_out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
        std::printf("reciprocal with one Newton step: %llu inputs in [2^-100, 2^100]: %llu mismatches", h[2], h[0]);
        if (h[0]) std::printf(" (first: 0x%08llx)", h[1]);
        std::printf("\n");
    }
    if (run_sqrt<1>(d_out) || run_sqrt<2>(d_out) || run_sqrt<3>(d_out)) return 1;
    if (run_div<1>(chunks, d_out) || run_div<2>(chunks, d_out) || run_div<3>(chunks, d_out)) return 1;
    return 0;
}
