// Microbenchmark: latency of the serial pieces of k_gn_solve (one lane per workgroup, 256 workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../direct-visual-odometry_amd/csrc/dvo_math.h"
using namespace dvo;

template <int MODE>
__global__ void __launch_bounds__(256) k(const double* in, float* out, double* st)
{
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    double tot[32];
    for (int i = 0; i < 29; i++) tot[i] = in[i] * (1.0 + 1e-6 * b);
    float upd[6] = {1e-3f, -2e-3f, 5e-4f, 1e-3f, 2e-3f, -1e-3f};
    if (MODE & 1) solve6(tot, tot + 21, upd);
    float xi[6] = {0.01f, 0.02f, -0.01f, 0.003f, -0.002f, 0.001f};
    Pose p;
    for (int i = 0; i < 12; i++) p.R[i < 9 ? i : 0] = 0;
    if (MODE & 2) {
        double Tc[12];
        for (int i = 0; i < 12; i++) Tc[i] = st[b * 12 + i];
        se3_update_pose(Tc, upd, xi, p);
        for (int i = 0; i < 12; i++) st[b * 12 + i] = Tc[i];
    }
    if (MODE & 4) {  // the old form: concatenate + pose_from_xi
        float nxt[6];
        se3_concatenate_f(xi, upd, nxt);
        pose_from_xi(nxt, -1.0f, p);
        for (int i = 0; i < 6; i++) xi[i] = nxt[i];
    }
    for (int i = 0; i < 6; i++) out[b * 32 + i] = upd[i] + xi[i];
    for (int i = 0; i < 12; i++) out[b * 32 + 6 + i] = i < 9 ? p.R[i] : p.t[i - 9];
}

template <int MODE>
void run(const char* name, const double* in, float* out, double* st)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, in, out, st);
    hipEventRecord(e0);
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, in, out, st);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.2f us per launch\n", name, ms * 10.0f);
}

int main()
{
    double h[32] = {0};
    // a well conditioned SPD 6x6 in upper-triangular packing + g
    int k2 = 0;
    for (int i = 0; i < 6; i++) for (int j = i; j < 6; j++) h[k2++] = (i == j) ? 100.0 + i : 1.0 + 0.1 * (i + j);
    for (int i = 0; i < 6; i++) h[21 + i] = 0.5 - 0.1 * i;
    h[27] = 10; h[28] = 1000;
    double *in, *st; float* out;
    hipMalloc(&in, sizeof h); hipMalloc(&out, 256 * 32 * 4); hipMalloc(&st, 256 * 12 * 8);
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    double id[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    for (int b = 0; b < 256; b++) hipMemcpy(st + b * 12, id, sizeof id, hipMemcpyHostToDevice);
    run<0>("empty (loads + stores only)", in, out, st);
    run<1>("solve6 (LDL^T, unrolled)", in, out, st);
    run<2>("se3_update_pose (2 exp + log)", in, out, st);
    run<4>("se3_concatenate + pose_from_xi (3 exp + log)", in, out, st);
    run<3>("solve6 + se3_update_pose", in, out, st);
    return 0;
}
