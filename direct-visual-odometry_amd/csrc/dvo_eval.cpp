// dvo_eval.cpp -- trajectory evaluation and export.  The reference has no implementation of its
// own headline accuracy metric; its only pose utility, Convert::inversePose (src/core/convert.cpp:31-39), is wrong
// (t' = -t, T(3,3) = 0) and is NOT reproduced: dvo_pose_inverse is the correct rigid inverse.  Host only (double).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "dvo_engine.h"

namespace dvo {

// symmetric Jacobi eigen-decomposition, n <= 4
static void jacobi_sym(double* A, double* V, int n)
{
    for (int i = 0; i < n * n; i++) V[i] = (i % (n + 1) == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double tau = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < n; k++) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

// Horn's closed-form absolute orientation (unit quaternion): R, t (and optional scale) minimising sum |gt - (s R est + t)|^2
static void align_horn(int n, const double* est, const double* gt, bool with_scale, double R[9], double t[3], double& s)
{
    double ce[3] = {0, 0, 0}, cg[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { ce[k] += est[3 * i + k]; cg[k] += gt[3 * i + k]; }
    for (int k = 0; k < 3; k++) { ce[k] /= n; cg[k] /= n; }
    double M[9] = {0};  // sum (est - ce)(gt - cg)^T
    double ve = 0;
    for (int i = 0; i < n; i++) {
        double a[3], b[3];
        for (int k = 0; k < 3; k++) { a[k] = est[3 * i + k] - ce[k]; b[k] = gt[3 * i + k] - cg[k]; }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) M[3 * r + c] += a[r] * b[c];
        ve += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    }
    const double Sxx = M[0], Sxy = M[1], Sxz = M[2], Syx = M[3], Syy = M[4], Syz = M[5], Szx = M[6], Szy = M[7], Szz = M[8];
    double N[16] = {Sxx + Syy + Szz, Syz - Szy,        Szx - Sxz,        Sxy - Syx,
                    Syz - Szy,       Sxx - Syy - Szz,  Sxy + Syx,        Szx + Sxz,
                    Szx - Sxz,       Sxy + Syx,        -Sxx + Syy - Szz, Syz + Szy,
                    Sxy - Syx,       Szx + Sxz,        Syz + Szy,        -Sxx - Syy + Szz};
    double V[16];
    jacobi_sym(N, V, 4);
    int best = 0;
    for (int i = 1; i < 4; i++)
        if (N[5 * i] > N[5 * best]) best = i;
    const double qw = V[0 * 4 + best], qx = V[1 * 4 + best], qy = V[2 * 4 + best], qz = V[3 * 4 + best];
    R[0] = qw * qw + qx * qx - qy * qy - qz * qz; R[1] = 2 * (qx * qy - qw * qz); R[2] = 2 * (qx * qz + qw * qy);
    R[3] = 2 * (qy * qx + qw * qz); R[4] = qw * qw - qx * qx + qy * qy - qz * qz; R[5] = 2 * (qy * qz - qw * qx);
    R[6] = 2 * (qz * qx - qw * qy); R[7] = 2 * (qz * qy + qw * qx); R[8] = qw * qw - qx * qx - qy * qy + qz * qz;
    s = 1.0;
    if (with_scale && ve > 0) {
        double num = 0;
        for (int i = 0; i < n; i++) {
            double a[3], b[3];
            for (int k = 0; k < 3; k++) { a[k] = est[3 * i + k] - ce[k]; b[k] = gt[3 * i + k] - cg[k]; }
            for (int r = 0; r < 3; r++) num += b[r] * (R[3 * r] * a[0] + R[3 * r + 1] * a[1] + R[3 * r + 2] * a[2]);
        }
        s = num / ve;
    }
    for (int r = 0; r < 3; r++) t[r] = cg[r] - s * (R[3 * r] * ce[0] + R[3 * r + 1] * ce[1] + R[3 * r + 2] * ce[2]);
}

static void mat4_mul(const double A[16], const double B[16], double C[16])
{
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double v = 0;
            for (int k = 0; k < 4; k++) v += A[4 * r + k] * B[4 * k + c];
            C[4 * r + c] = v;
        }
}

static void mat4_inv_rigid(const double T[16], double I[16])
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) I[4 * r + c] = T[4 * c + r];
    for (int r = 0; r < 3; r++) I[4 * r + 3] = -(I[4 * r] * T[3] + I[4 * r + 1] * T[7] + I[4 * r + 2] * T[11]);
    I[12] = 0; I[13] = 0; I[14] = 0; I[15] = 1;
}

}  // namespace dvo

using namespace dvo;

extern "C" {

// Absolute trajectory error: RMSE of |gt_i - (s R est_i + t)| after the optimal rigid (or similarity) alignment.
int dvo_eval_ate(int n, const float* est_xyz, const float* gt_xyz, int with_scale, double* rmse, double R_out[9], double t_out[3], double* scale_out)
{
    if (n < 3 || !est_xyz || !gt_xyz || !rmse) { set_error("dvo_eval_ate: need >= 3 poses"); return DVO_ERR_BAD_ARGUMENT; }
    std::vector<double> e(3 * (size_t)n), g(3 * (size_t)n);
    for (size_t i = 0; i < 3 * (size_t)n; i++) { e[i] = est_xyz[i]; g[i] = gt_xyz[i]; }
    double R[9], t[3], s;
    align_horn(n, e.data(), g.data(), with_scale != 0, R, t, s);
    double sum = 0;
    for (int i = 0; i < n; i++)
        for (int r = 0; r < 3; r++) {
            const double p = s * (R[3 * r] * e[3 * i] + R[3 * r + 1] * e[3 * i + 1] + R[3 * r + 2] * e[3 * i + 2]) + t[r];
            sum += (g[3 * i + r] - p) * (g[3 * i + r] - p);
        }
    *rmse = std::sqrt(sum / n);
    if (R_out) memcpy(R_out, R, sizeof R);
    if (t_out) memcpy(t_out, t, sizeof t);
    if (scale_out) *scale_out = s;
    return DVO_OK;
}

// Relative pose error over a fixed frame interval `delta`: E_i = (Q_i^-1 Q_{i+delta})^-1 (P_i^-1 P_{i+delta}); RMSE of
// the translational part and of the rotation angle (radians).  Poses are row-major 4x4 (world <- camera).
int dvo_eval_rpe(int n, const float* est_T, const float* gt_T, int delta, double* trans_rmse, double* rot_rmse)
{
    if (n < 2 || delta < 1 || delta >= n || !est_T || !gt_T) { set_error("dvo_eval_rpe: bad arguments"); return DVO_ERR_BAD_ARGUMENT; }
    double st = 0, sr = 0;
    int m = 0;
    for (int i = 0; i + delta < n; i++) {
        double P0[16], P1[16], Q0[16], Q1[16], Pi[16], Qi[16], dP[16], dQ[16], dQi[16], E[16];
        for (int k = 0; k < 16; k++) { P0[k] = est_T[16 * i + k]; P1[k] = est_T[16 * (i + delta) + k]; Q0[k] = gt_T[16 * i + k]; Q1[k] = gt_T[16 * (i + delta) + k]; }
        mat4_inv_rigid(P0, Pi); mat4_mul(Pi, P1, dP);
        mat4_inv_rigid(Q0, Qi); mat4_mul(Qi, Q1, dQ);
        mat4_inv_rigid(dQ, dQi); mat4_mul(dQi, dP, E);
        st += E[3] * E[3] + E[7] * E[7] + E[11] * E[11];
        double ct = 0.5 * (E[0] + E[5] + E[10] - 1.0);
        ct = ct > 1 ? 1 : (ct < -1 ? -1 : ct);
        const double ang = std::acos(ct);
        sr += ang * ang;
        m++;
    }
    if (trans_rmse) *trans_rmse = std::sqrt(st / m);
    if (rot_rmse) *rot_rmse = std::sqrt(sr / m);
    return DVO_OK;
}

// The correct rigid inverse (what Convert::inversePose, src/core/convert.cpp:31-39, meant to be).
int dvo_pose_inverse(const float T[16], float out[16])
{
    if (!T || !out) return DVO_ERR_BAD_ARGUMENT;
    double a[16], b[16];
    for (int i = 0; i < 16; i++) a[i] = T[i];
    mat4_inv_rigid(a, b);
    for (int i = 0; i < 16; i++) out[i] = (float)b[i];
    return DVO_OK;
}

// TUM trajectory format: "timestamp tx ty tz qx qy qz qw" per line.
int dvo_traj_write_tum(const char* path, int n, const double* timestamps, const float* T)
{
    if (!path || n < 0 || !T) return DVO_ERR_BAD_ARGUMENT;
    FILE* f = fopen(path, "w");
    if (!f) { set_error(std::string("cannot write ") + path); return DVO_ERR_BAD_ARGUMENT; }
    fprintf(f, "# timestamp tx ty tz qx qy qz qw\n");
    for (int i = 0; i < n; i++) {
        const float* M = T + 16 * (size_t)i;
        const double m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[4], m11 = M[5], m12 = M[6], m20 = M[8], m21 = M[9], m22 = M[10];
        double qw, qx, qy, qz;
        const double tr = m00 + m11 + m22;
        if (tr > 0) {
            const double S = std::sqrt(tr + 1.0) * 2; qw = 0.25 * S; qx = (m21 - m12) / S; qy = (m02 - m20) / S; qz = (m10 - m01) / S;
        } else if (m00 > m11 && m00 > m22) {
            const double S = std::sqrt(1.0 + m00 - m11 - m22) * 2; qw = (m21 - m12) / S; qx = 0.25 * S; qy = (m01 + m10) / S; qz = (m02 + m20) / S;
        } else if (m11 > m22) {
            const double S = std::sqrt(1.0 + m11 - m00 - m22) * 2; qw = (m02 - m20) / S; qx = (m01 + m10) / S; qy = 0.25 * S; qz = (m12 + m21) / S;
        } else {
            const double S = std::sqrt(1.0 + m22 - m00 - m11) * 2; qw = (m10 - m01) / S; qx = (m02 + m20) / S; qy = (m12 + m21) / S; qz = 0.25 * S;
        }
        fprintf(f, "%.6f %.7f %.7f %.7f %.7f %.7f %.7f %.7f\n", timestamps ? timestamps[i] : (double)i, M[3], M[7], M[11], qx, qy, qz, qw);
    }
    fclose(f);
    return DVO_OK;
}

}  // extern "C"
