// Microbenchmark: duration of the pieces of the serial chain one Gauss-Newton step ends with (solve_finish: solve6, exp(upd), log, exp pair),
// one lane, each piece repeated 64 times in a dependent chain (the launch floor of ~3 us hides a single call).
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../../include -o solve_chain2 solve_chain2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../direct-visual-odometry_amd/csrc/dvo_math.h"
using namespace dvo;

template <int MODE>
__global__ void __launch_bounds__(64) k(const double* in, float* out, int reps)
{
    if (threadIdx.x != 0) return;
    double tot[32];
    for (int i = 0; i < 29; i++) tot[i] = in[i];
    float upd[6] = {1e-3f, -2e-3f, 5e-4f, 1e-3f, 2e-3f, -1e-3f};
    float xi[6] = {0.01f, 0.02f, -0.01f, 0.003f, -0.002f, 0.001f};
    double Tc[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    double acc = 0;
    for (int r = 0; r < reps; r++) {
        if (MODE == 1) {          // solve6
            solve6(tot, tot + 21, upd);
            tot[21] += 1e-9 * (double)upd[0];     // dependent chain
        } else if (MODE == 2) {   // se3_exp_d
            double x[6], R[9], t[3];
            for (int i = 0; i < 6; i++) x[i] = (double)xi[i] + acc;
            se3_exp_d(x, R, t);
            acc = 1e-12 * (R[1] + t[0]);
        } else if (MODE == 3) {   // se3_log_d
            double x[6];
            Tc[9] += acc;
            se3_log_d(Tc, Tc + 9, x);
            acc = 1e-12 * (x[0] + x[3]);
            Tc[1] = 1e-3 + acc; Tc[3] = -1e-3 - acc;
        } else if (MODE == 4) {   // se3_update_pose (exp, product, log, exp pair)
            Pose p;
            se3_update_pose(Tc, upd, xi, p);
            upd[0] = 1e-3f + 1e-6f * p.t[0];
        } else if (MODE == 5) {   // sin + cos in double
            const double th = 0.3 + acc;
            acc = 1e-12 * (sin(th) + cos(th));
        } else if (MODE == 6) {   // atan2 + sqrt
            acc = 1e-12 * atan2(0.3 + acc, 0.9) + 1e-12 * sqrt(2.0 + acc);
        }
    }
    out[0] = upd[0] + xi[0] + (float)acc + (float)Tc[0] + (float)tot[21];
}

template <int MODE>
void run(const char* name, const double* in, float* out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float t[2];
    for (int v = 0; v < 2; v++) {
        const int reps = v ? 264 : 8;
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, in, out, reps);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, in, out, reps);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&t[v], e0, e1);
    }
    printf("%-44s %6.2f us per call\n", name, (t[1] - t[0]) * 1000.0f / 20.0f / 256.0f);
}

int main()
{
    double h[32] = {0};
    int k2 = 0;
    for (int i = 0; i < 6; i++) for (int j = i; j < 6; j++) h[k2++] = (i == j) ? 100.0 + i : 1.0 + 0.1 * (i + j);
    for (int i = 0; i < 6; i++) h[21 + i] = 0.5 - 0.1 * i;
    h[27] = 10; h[28] = 1000;
    double* in; float* out;
    (void)hipMalloc(&in, sizeof h); (void)hipMalloc(&out, 64);
    (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    run<1>("solve6 (LDL^T, 21 fp64 divisions)", in, out);
    run<2>("se3_exp_d (sqrt, sin, cos, 5 divisions)", in, out);
    run<3>("se3_log_d (2 sqrt, atan2, sin, cos, 3 div)", in, out);
    run<4>("se3_update_pose (exp, log, exp pair)", in, out);
    run<5>("sin + cos (double)", in, out);
    run<6>("atan2 + sqrt (double)", in, out);
    return 0;
}
