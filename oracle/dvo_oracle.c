/*
 * dvo_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see dvo_oracle.h).
 * PARITY STATUS: parity unpinned (the reference holds no golden vectors for this path).
 *
 * Plain C restatement of the hot path of KYabuuchi/direct-visual-odometry.
 * Citations are file:line under /root/reference.  Sequential raster order, scalar.
 * Build: gcc -O2 -std=c11 -ffp-contract=off -mfma (explicit fmaf only; see D8).
 */
#include "dvo_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Row-parallel execution of the forEach bodies (cv::Mat::forEach runs them on OpenCV's thread pool: optimize.cpp:28,
 * transform.cpp:39, convert.cpp:48,61).  Default 1 thread = the sequential raster order of D7, bit for bit; n > 1 is used by
 * bench.py's all-core CPU baseline only (per-thread partial sums combined in thread order: deterministic for a fixed n). */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }
int  orc_get_threads(void) { return g_threads; }

/* Arithmetic mode (test instrumentation for deviations D8 / D2).  0 (default) = the canonical order the product
 * shares (shared reciprocals, explicit fmaf chains, SE(3) in double).  ORC_LIT_ARITH: every per-pixel expression exactly as the
 * source text writes it -- true divisions, no shared reciprocal, no fused multiply-add, left-to-right float evaluation
 * (transform.cpp:20-28, optimize.cpp:67-77, convert.cpp:103-104, gaussian.cpp:28, implement.cpp:85,245-246).  ORC_LIT_SE3: the
 * per-pixel pose and the pose composition from the float-literal restatement of se3.cpp (orc_se3_*_f32lit).  Only the
 * sensitivity tests and bench.py's oracle_self_sensitivity leg set it; never concurrently with running oracle calls. */
static int g_lit = 0;
void orc_set_literal(int mask) { g_lit = mask & (ORC_LIT_ARITH | ORC_LIT_SE3); }
int  orc_get_literal(void) { return g_lit; }
/* Sensitivity probe: orc_track moves the first component of the first xi_update (level 0, iteration 0) by n units in the last place.
 * 0 = off.  A 1-ulp nudge is below anything two correct implementations can be expected to share. */
static int g_nudge = 0;
void orc_set_nudge_ulps(int n) { g_nudge = n; }

/* ------------------------------------------------------------------------ */
/* include/math/util.hpp:6-32                                                */
static inline int is_valid(float v) { return ORC_INVALID < v; }
static inline int is_invalid(float v) { return v <= ORC_INVALID; }
static inline int is_epsilon(float v) { return fabsf(v) < ORC_EPSILON; }
static inline int in_range(int x, int y, int w, int h) { return !(x < 0 || w <= x || y < 0 || h <= y); }
/* D4: coordinates that cannot be converted to int are "out of range" */
static inline int coord_ok(float v) { return fabsf(v) < 1073741824.0f; } /* false for NaN/inf */

/* ======================================================================== */
/* SE(3) in double (D2).  src/math/se3.cpp                                   */
/* ======================================================================== */
static void hat_mul(const double w[3], const double v[3], double out[3])
{ /* w x v  (se3.cpp:8-15 hat) */
    out[0] = w[1] * v[2] - w[2] * v[1];
    out[1] = w[2] * v[0] - w[0] * v[2];
    out[2] = w[0] * v[1] - w[1] * v[0];
}

/* se3.cpp:70-98; so3::exp = cv::Rodrigues (Appendix B of SURVEY.md) */
static void se3_exp_d(const double xi[6], double R[9], double t[3])
{
    const double v[3] = {xi[0], xi[1], xi[2]};
    const double w[3] = {xi[3], xi[4], xi[5]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th = sqrt(th2);
    double c = 1.0, s = 0.0;
    if (th < DBL_EPSILON) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    } else {
        c = cos(th); s = sin(th);
        const double c1 = 1.0 - c, rx = w[0] / th, ry = w[1] / th, rz = w[2] / th;
        R[0] = c + c1 * rx * rx;      R[1] = c1 * rx * ry - s * rz; R[2] = c1 * rx * rz + s * ry;
        R[3] = c1 * rx * ry + s * rz; R[4] = c + c1 * ry * ry;      R[5] = c1 * ry * rz - s * rx;
        R[6] = c1 * rx * rz - s * ry; R[7] = c1 * ry * rz + s * rx; R[8] = c + c1 * rz * rz;
    }
    if ((float)th > 1e-6f) { /* se3.cpp:84 */
        const double A = (1.0 - c) / th2, B = (th - s) / (th2 * th);
        double wv[3], wwv[3];
        hat_mul(w, v, wv);
        hat_mul(w, wv, wwv);
        for (int i = 0; i < 3; i++) t[i] = v[i] + A * wv[i] + B * wwv[i];
    } else {
        t[0] = v[0]; t[1] = v[1]; t[2] = v[2];
    }
}

/* se3.cpp:31-43 (so3::log) and :101-124 (se3::log).  theta = atan2(|a|, (tr-1)/2) is the
 * same angle as acos((tr-1)/2) but cannot produce NaN by rounding (D2). */
static void se3_log_d(const double R[9], const double t[3], double xi[6])
{
    const double a[3] = {0.5 * (R[7] - R[5]), 0.5 * (R[2] - R[6]), 0.5 * (R[3] - R[1])};
    const double s = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const double cth = 0.5 * (R[0] + R[4] + R[8] - 1.0);
    const double th = atan2(s, cth);
    double w[3] = {0, 0, 0};
    if ((float)th > 1e-6f && s > 0.0) { /* se3.cpp:37 */
        const double k = th / s;
        w[0] = a[0] * k; w[1] = a[1] * k; w[2] = a[2] * k;
    }
    const double wl2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double wl = sqrt(wl2);
    double v[3] = {t[0], t[1], t[2]};
    if ((float)wl > 1e-6f) { /* se3.cpp:113-118 */
        const double half = 0.5 * wl;
        const double coef = (1.0 - (wl * cos(half)) / (2.0 * sin(half))) / wl2;
        double wt[3], wwt[3];
        hat_mul(w, t, wt);
        hat_mul(w, wt, wwt);
        for (int i = 0; i < 3; i++) v[i] = t[i] - 0.5 * wt[i] + coef * wwt[i];
    }
    xi[0] = v[0]; xi[1] = v[1]; xi[2] = v[2]; xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

static void mat_to_T(const double R[9], const double t[3], float T[16])
{
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[4 * r + c] = (float)R[3 * r + c];
        T[4 * r + 3] = (float)t[r];
    }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

void orc_se3_exp(const float xi[6], float T[16])
{
    double x[6], R[9], t[3];
    for (int i = 0; i < 6; i++) x[i] = xi[i];
    se3_exp_d(x, R, t);
    mat_to_T(R, t, T);
}

void orc_se3_log(const float T[16], float xi[6])
{
    double R[9], t[3], x[6];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) R[3 * r + c] = T[4 * r + c];
        t[r] = T[4 * r + 3];
    }
    se3_log_d(R, t, x);
    for (int i = 0; i < 6; i++) xi[i] = (float)x[i];
}

/* se3.cpp:127-131: log(exp(a) * exp(b)), all in double, rounded to float once */
void orc_se3_concatenate(const float a[6], const float b[6], float out[6])
{
    double xa[6], xb[6], Ra[9], ta[3], Rb[9], tb[3], R[9], t[3], x[6];
    for (int i = 0; i < 6; i++) { xa[i] = a[i]; xb[i] = b[i]; }
    se3_exp_d(xa, Ra, ta);
    se3_exp_d(xb, Rb, tb);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            R[3 * r + c] = Ra[3 * r] * Rb[c] + Ra[3 * r + 1] * Rb[3 + c] + Ra[3 * r + 2] * Rb[6 + c];
        t[r] = Ra[3 * r] * tb[0] + Ra[3 * r + 1] * tb[1] + Ra[3 * r + 2] * tb[2] + ta[r];
    }
    se3_log_d(R, t, x);
    for (int i = 0; i < 6; i++) out[i] = (float)x[i];
}

void orc_se3_exp_f32lit(const float xi[6], float T[16]);
void orc_pose_from_xi(const float xi[6], float sign, float Rt[12])
{
    if (g_lit & ORC_LIT_SE3) { /* transform.cpp:13-14: transform(se3::exp(cv::Mat1f(-xi)), x), exp in float (se3.cpp:70-98) */
        float xs[6], T[16];
        for (int i = 0; i < 6; i++) xs[i] = sign * xi[i];
        orc_se3_exp_f32lit(xs, T);
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) Rt[3 * r + c] = T[4 * r + c];
            Rt[9 + r] = T[4 * r + 3];
        }
        return;
    }
    double x[6], R[9], t[3];
    for (int i = 0; i < 6; i++) x[i] = (double)sign * (double)xi[i];
    se3_exp_d(x, R, t);
    for (int i = 0; i < 9; i++) Rt[i] = (float)R[i];
    for (int i = 0; i < 3; i++) Rt[9 + i] = (float)t[i];
}

/* ---- float-literal restatement of se3.cpp (for the D2 deviation test only) ---- */
static void mat3_mul_f(const float A[9], const float B[9], float C[9])
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double acc = 0; /* cv::gemm accumulates float products in double */
            for (int k = 0; k < 3; k++) acc += (double)A[3 * r + k] * (double)B[3 * k + c];
            C[3 * r + c] = (float)acc;
        }
}

static void hat_f(const float w[3], float H[9])
{ /* se3.cpp:8-15 */
    H[0] = 0; H[1] = -w[2]; H[2] = w[1]; H[3] = w[2]; H[4] = 0; H[5] = -w[0]; H[6] = -w[1]; H[7] = w[0]; H[8] = 0;
}

static void rodrigues_f(const float w[3], float R[9])
{ /* se3.cpp:21-28 -> cv::Rodrigues computes in double, result cast to float */
    double th = sqrt((double)w[0] * w[0] + (double)w[1] * w[1] + (double)w[2] * w[2]);
    if (th < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
        return;
    }
    double c = cos(th), s = sin(th), c1 = 1.0 - c, rx = w[0] / th, ry = w[1] / th, rz = w[2] / th;
    double Rd[9] = {c + c1 * rx * rx, c1 * rx * ry - s * rz, c1 * rx * rz + s * ry,
                    c1 * rx * ry + s * rz, c + c1 * ry * ry, c1 * ry * rz - s * rx,
                    c1 * rx * rz - s * ry, c1 * ry * rz + s * rx, c + c1 * rz * rz};
    for (int i = 0; i < 9; i++) R[i] = (float)Rd[i];
}

void orc_se3_exp_f32lit(const float xi[6], float T[16])
{ /* se3.cpp:70-98 literally, in float */
    const float v[3] = {xi[0], xi[1], xi[2]}, w[3] = {xi[3], xi[4], xi[5]};
    float W[9], WW[9], R[9], t[3];
    hat_f(w, W);
    const float wl = (float)sqrt((double)w[0] * w[0] + (double)w[1] * w[1] + (double)w[2] * w[2]);
    rodrigues_f(w, R);
    if (wl > 1e-6f) {
        mat3_mul_f(W, W, WW);
        const float ka = (1.0f - cosf(wl)) / (wl * wl);
        const float kb = (wl - sinf(wl)) / (wl * wl * wl);
        float V[9];
        for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + W[i] * ka + WW[i] * kb;
        for (int r = 0; r < 3; r++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += (double)V[3 * r + k] * (double)v[k];
            t[r] = (float)acc;
        }
    } else {
        t[0] = v[0]; t[1] = v[1]; t[2] = v[2];
    }
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[4 * r + c] = R[3 * r + c];
        T[4 * r + 3] = t[r];
    }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

void orc_se3_log_f32lit(const float T[16], float xi[6])
{ /* se3.cpp:31-43, 101-124 literally, in float (acos may yield NaN) */
    float R[9], t[3], w[3] = {0, 0, 0};
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) R[3 * r + c] = T[4 * r + c];
        t[r] = T[4 * r + 3];
    }
    const float tr = (float)((double)R[0] + (double)R[4] + (double)R[8]);
    const float th = acosf((tr - 1.0f) * 0.5f);
    if (th > 1e-6f) {
        const float k = 1.0f / (2.0f * sinf(th));
        w[0] = k * (R[7] - R[5]) * th; w[1] = k * (R[2] - R[6]) * th; w[2] = k * (R[3] - R[1]) * th;
    }
    float W[9], WW[9], Vi[9];
    hat_f(w, W);
    const float wl = (float)sqrt((double)w[0] * w[0] + (double)w[1] * w[1] + (double)w[2] * w[2]);
    for (int i = 0; i < 9; i++) Vi[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    if (wl > 1e-6f) {
        mat3_mul_f(W, W, WW);
        const float k = (1.0f - (wl * cosf(wl * 0.5f)) / (2.0f * sinf(wl * 0.5f)));
        for (int i = 0; i < 9; i++) Vi[i] = Vi[i] - 0.5f * W[i] + k * WW[i] / (wl * wl);
    }
    for (int r = 0; r < 3; r++) {
        double acc = 0;
        for (int k = 0; k < 3; k++) acc += (double)Vi[3 * r + k] * (double)t[k];
        xi[r] = (float)acc;
    }
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

void orc_se3_concatenate_f32lit(const float a[6], const float b[6], float out[6])
{
    float Ta[16], Tb[16], T[16];
    orc_se3_exp_f32lit(a, Ta);
    orc_se3_exp_f32lit(b, Tb);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double acc = 0;
            for (int k = 0; k < 4; k++) acc += (double)Ta[4 * r + k] * (double)Tb[4 * k + c];
            T[4 * r + c] = (float)acc;
        }
    orc_se3_log_f32lit(T, out);
}

/* ======================================================================== */
/* Image primitives.  src/core/convert.cpp                                   */
/* ======================================================================== */
float orc_get_pixel(const float* img, int w, int h, int x, int y)
{ /* convert.cpp:107-125 (the 1x1 loops are vestigial) */
    if (in_range(x, y, w, h)) {
        float g = img[y * w + x];
        if (is_valid(g)) return g;
    }
    return ORC_INVALID;
}

void orc_cull_image(const float* src, int w, int h, int times, float* dst)
{ /* convert.cpp:7-20; cv::Size / int truncates */
    if (times == 0) {
        memcpy(dst, src, sizeof(float) * (size_t)w * h);
        return;
    }
    const int r = 1 << times, dw = w / r, dh = h / r;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) dst[y * dw + x] = orc_get_pixel(src, w, h, x * r, y * r);
}

void orc_cull_intrinsic(const float K[9], int times, float out[9])
{ /* convert.cpp:22-29 */
    if (times == 0) {
        memcpy(out, K, 9 * sizeof(float));
        return;
    }
    const double r = (double)(1 << times);
    for (int i = 0; i < 9; i++) out[i] = (float)((double)K[i] / r);
    out[8] = 1.0f;
}

void orc_gradiate(const float* img, int w, int h, int xdir, float* out)
{ /* convert.cpp:41-75 */
    for (int i = 0; i < w * h; i++) out[i] = ORC_INVALID;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float a, b;
            if (xdir) {
                if (x - 1 <= -1 || x + 1 >= w) continue;
                a = orc_get_pixel(img, w, h, x - 1, y);
                b = orc_get_pixel(img, w, h, x + 1, y);
            } else {
                if (y - 1 <= -1 || y + 1 >= h) continue;
                a = orc_get_pixel(img, w, h, x, y - 1);
                b = orc_get_pixel(img, w, h, x, y + 1);
            }
            if (is_invalid(a) || is_invalid(b)) continue;
            out[y * w + x] = b - a;
        }
}

/* bilinear blend in the fixed order of DESIGN.md §3 (D8) */
static inline float blend4(const float g[4], float hx, float vy)
{
    if (g_lit & ORC_LIT_ARITH) /* convert.cpp:103-104, 175-176 as written */
        return (g[0] * (1.f - hx) + g[1] * hx) * (1.f - vy) + (g[2] * (1.f - hx) + g[3] * hx) * vy;
    const float omh = 1.0f - hx, omv = 1.0f - vy;
    const float top = fmaf(g[1], hx, g[0] * omh);
    const float bot = fmaf(g[3], hx, g[2] * omh);
    return fmaf(bot, vy, top * omv);
}

static inline int load_taps(const float* img, int w, int h, float px, float py, float g[4], float* hx, float* vy)
{ /* convert.cpp:82-101 / :133-153: trunc toward zero, clamp missing taps to g00 */
    if (!coord_ok(px) || !coord_ok(py)) return 0; /* D4 */
    const int x0 = (int)px, y0 = (int)py;
    if (!in_range(x0, y0, w, h)) return 0;
    const int x1 = x0 + 1, y1 = y0 + 1;
    *hx = px - (float)x0;
    *vy = py - (float)y0;
    g[0] = g[1] = g[2] = g[3] = img[y0 * w + x0];
    if (in_range(x1, y0, w, h)) g[1] = img[y0 * w + x1];
    if (in_range(x0, y1, w, h)) g[2] = img[y1 * w + x0];
    if (in_range(x1, y1, w, h)) g[3] = img[y1 * w + x1];
    return 1;
}

float orc_get_subpixel_dense(const float* img, int w, int h, float px, float py)
{ /* convert.cpp:77-105: INVALID neighbours are blended in as numbers */
    float g[4], hx, vy;
    if (!load_taps(img, w, h, px, py, g, &hx, &vy)) return ORC_INVALID;
    return blend4(g, hx, vy);
}

float orc_get_subpixel(const float* img, int w, int h, float px, float py)
{ /* convert.cpp:128-177: the fill loop is replicated literally */
    float g[4], hx, vy;
    if (!load_taps(img, w, h, px, py, g, &hx, &vy)) return ORC_INVALID;
    int valid = 0, id = 0;
    float last = -1.0f;
    for (;;) {
        if (is_valid(g[id])) {
            valid++;
            last = g[id];
        } else if (last > 0) {
            g[id] = last;
            valid++;
        }
        if (valid == 4) break;
        if (id == 3 && valid == 0) return ORC_INVALID;
        id = (id + 1) % 4;
    }
    return blend4(g, hx, vy);
}

/* ======================================================================== */
/* Geometry.  src/core/transform.cpp                                         */
/* ======================================================================== */
void orc_back_project(const float K[9], float px, float py, float d, float X[3])
{ /* transform.cpp:25-28: depth * (p - c) / f.  D8: the division by the per-level constant f is a multiplication
   * by its correctly rounded reciprocal (the reference is built -Ofast, i.e. -freciprocal-math) */
    if (g_lit & ORC_LIT_ARITH) { /* depth * (point.x - K(0,2)) / K(0,0) */
        X[0] = d * (px - K[2]) / K[0];
        X[1] = d * (py - K[5]) / K[4];
        X[2] = d;
        return;
    }
    const float ifx = 1.0f / K[0], ify = 1.0f / K[4];
    X[0] = (d * (px - K[2])) * ifx;
    X[1] = (d * (py - K[5])) * ify;
    X[2] = d;
}

void orc_project(const float K[9], const float X[3], float p[2])
{ /* transform.cpp:20-23; D8: one reciprocal of z shared by both coordinates */
    if (g_lit & ORC_LIT_ARITH) { /* point.x * K(0,0) / point.z + K(0,2) */
        p[0] = X[0] * K[0] / X[2] + K[2];
        p[1] = X[1] * K[4] / X[2] + K[5];
        return;
    }
    const float iz = 1.0f / X[2];
    p[0] = (X[0] * K[0]) * iz + K[2];
    p[1] = (X[1] * K[4]) * iz + K[5];
}

void orc_transform(const float Rt[12], const float X[3], float Y[3])
{ /* transform.cpp:7-12: R x + t as an fmaf chain (D8) */
    if (g_lit & ORC_LIT_ARITH) { /* R * cv::Mat(x) + t: a 3x3 by 3x1 float product, row sums left to right in float, then + t
                                  * (cv::gemm's small-matrix path; Appendix B of SURVEY.md: OpenCV source is not in the tree) */
        for (int i = 0; i < 3; i++) Y[i] = (Rt[3 * i] * X[0] + Rt[3 * i + 1] * X[1] + Rt[3 * i + 2] * X[2]) + Rt[9 + i];
        return;
    }
    for (int i = 0; i < 3; i++)
        Y[i] = fmaf(Rt[3 * i], X[0], fmaf(Rt[3 * i + 1], X[1], fmaf(Rt[3 * i + 2], X[2], Rt[9 + i])));
}

void orc_warp(const float Rt[12], float px, float py, float d, const float K[9], float p[2])
{ /* transform.cpp:30-33 */
    float X[3], Y[3];
    orc_back_project(K, px, py, d, X);
    orc_transform(Rt, X, Y);
    orc_project(K, Y, p);
}

void orc_warp_image(const float xi[6], const float* gray, const float* depth, int w, int h, const float K[9], float* out)
{ /* transform.cpp:35-51; the reference re-derives exp(-xi) per pixel (transform.cpp:13-14) */
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int i = y * w + x;
            out[i] = ORC_INVALID;
            const float d = depth[i];
            if (is_epsilon(d)) continue;
            float Rt[12], p[2];
            orc_pose_from_xi(xi, -1.0f, Rt); /* per pixel, as the reference does */
            orc_warp(Rt, (float)x, (float)y, d, K, p);
            out[i] = orc_get_subpixel(gray, w, h, p[0], p[1]);
        }
}

/* ======================================================================== */
/* Solvers                                                                    */
/* ======================================================================== */
static void sym_from_upper(const double H[21], double A[36])
{
    int k = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 6; j++) {
            A[6 * i + j] = H[k];
            A[6 * j + i] = H[k];
            k++;
        }
}

/* cyclic Jacobi eigen-decomposition of a symmetric 6x6 (fallback pseudo-inverse) */
static void jacobi_eig6(double A[36], double V[36])
{
    for (int i = 0; i < 36; i++) V[i] = (i % 7 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < 6; i++) {
            diag += A[7 * i] * A[7 * i];
            for (int j = i + 1; j < 6; j++) off += A[6 * i + j] * A[6 * i + j];
        }
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int p = 0; p < 5; p++)
            for (int q = p + 1; q < 6; q++) {
                const double apq = A[6 * p + q];
                if (apq == 0.0) continue;
                const double tau = (A[7 * q] - A[7 * p]) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < 6; k++) { /* columns */
                    const double akp = A[6 * k + p], akq = A[6 * k + q];
                    A[6 * k + p] = c * akp - s * akq;
                    A[6 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; k++) { /* rows */
                    const double apk = A[6 * p + k], aqk = A[6 * q + k];
                    A[6 * p + k] = c * apk - s * aqk;
                    A[6 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; k++) {
                    const double vkp = V[6 * k + p], vkq = V[6 * k + q];
                    V[6 * k + p] = c * vkp - s * vkq;
                    V[6 * k + q] = s * vkp + c * vkq;
                }
            }
    }
}

/* x = H^+ g.  Fast path LDL^T (double); any pivot <= 1e-12 * max diag -> eigen pseudo-inverse with the
 * cv::solve(DECOMP_SVD) threshold: singular values of A (= sqrt(lambda)) <= 2*FLT_EPSILON*sum are dropped. */
void orc_solve6(const double H[21], const double g[6], float x[6])
{
    double A[36], L[36], d[6], y[6], z[6];
    sym_from_upper(H, A);
    double maxd = 0;
    for (int i = 0; i < 6; i++) maxd = fmax(maxd, A[7 * i]);
    for (int i = 0; i < 6; i++) x[i] = 0.0f;
    if (!(maxd > 0.0)) return;
    int ok = 1;
    memset(L, 0, sizeof L);
    for (int j = 0; j < 6 && ok; j++) {
        double dj = A[7 * j];
        for (int k = 0; k < j; k++) dj -= L[6 * j + k] * L[6 * j + k] * d[k];
        if (!(dj > 1e-12 * maxd)) { ok = 0; break; }
        d[j] = dj;
        L[7 * j] = 1.0;
        for (int i = j + 1; i < 6; i++) {
            double v = A[6 * i + j];
            for (int k = 0; k < j; k++) v -= L[6 * i + k] * L[6 * j + k] * d[k];
            L[6 * i + j] = v / dj;
        }
    }
    if (ok) {
        for (int i = 0; i < 6; i++) { /* L y = g */
            double v = g[i];
            for (int k = 0; k < i; k++) v -= L[6 * i + k] * y[k];
            y[i] = v;
        }
        for (int i = 0; i < 6; i++) y[i] /= d[i];
        for (int i = 5; i >= 0; i--) { /* L^T z = y */
            double v = y[i];
            for (int k = i + 1; k < 6; k++) v -= L[6 * k + i] * z[k];
            z[i] = v;
        }
        for (int i = 0; i < 6; i++) x[i] = (float)z[i];
        return;
    }
    double V[36], sv[6], sum = 0;
    jacobi_eig6(A, V);
    for (int i = 0; i < 6; i++) { sv[i] = sqrt(fmax(A[7 * i], 0.0)); sum += sv[i]; }
    const double thr = 2.0 * (double)FLT_EPSILON * sum;
    for (int i = 0; i < 6; i++) z[i] = 0;
    for (int i = 0; i < 6; i++) {
        if (!(sv[i] > thr)) continue;
        double proj = 0;
        for (int k = 0; k < 6; k++) proj += V[6 * k + i] * g[k];
        proj /= A[7 * i];
        for (int k = 0; k < 6; k++) z[k] += V[6 * k + i] * proj;
    }
    for (int i = 0; i < 6; i++) x[i] = (float)z[i];
}

/* One-sided (Hestenes) Jacobi SVD least squares on the N x 6 float stack, as cv::solve(A,-B,DECOMP_SVD)
 * does for the reference (optimize.cpp:96-98).  Returns xi_update = -x = (A^T A)^+ A^T B. */
void orc_lsq_svd(const float* A, const float* B, int n, float x_update[6])
{
    float* W = (float*)malloc(sizeof(float) * (size_t)n * 6);
    memcpy(W, A, sizeof(float) * (size_t)n * 6);
    double V[36];
    for (int i = 0; i < 36; i++) V[i] = (i % 7 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        int changed = 0;
        for (int i = 0; i < 5; i++)
            for (int j = i + 1; j < 6; j++) {
                double a = 0, b = 0, p = 0;
                for (int k = 0; k < n; k++) {
                    const double wi = W[6 * k + i], wj = W[6 * k + j];
                    a += wi * wi; b += wj * wj; p += wi * wj;
                }
                if (fabs(p) <= (double)FLT_EPSILON * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot(p, beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                for (int k = 0; k < n; k++) {
                    const double wi = W[6 * k + i], wj = W[6 * k + j];
                    W[6 * k + i] = (float)(c * wi + s * wj);
                    W[6 * k + j] = (float)(-s * wi + c * wj);
                }
                for (int k = 0; k < 6; k++) {
                    const double vi = V[6 * k + i], vj = V[6 * k + j];
                    V[6 * k + i] = c * vi + s * vj;
                    V[6 * k + j] = -s * vi + c * vj;
                }
                changed = 1;
            }
        if (!changed) break;
    }
    double sv2[6], sv[6], wb[6], sum = 0;
    for (int i = 0; i < 6; i++) {
        double a = 0, p = 0;
        for (int k = 0; k < n; k++) { a += (double)W[6 * k + i] * W[6 * k + i]; p += (double)W[6 * k + i] * B[k]; }
        sv2[i] = a; sv[i] = sqrt(a); wb[i] = p; sum += sv[i];
    }
    const double thr = 2.0 * (double)FLT_EPSILON * sum;
    double z[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        if (!(sv[i] > thr)) continue;
        const double proj = wb[i] / sv2[i];
        for (int k = 0; k < 6; k++) z[k] += V[6 * k + i] * proj;
    }
    for (int i = 0; i < 6; i++) x_update[i] = (float)z[i];
    free(W);
}

/* ======================================================================== */
/* Track::optimize.  src/track/optimize.cpp:10-99                            */
/* ======================================================================== */
/* The reference's literals (optimize.cpp:22-26, tracker.cpp:16-17).  orc_set_tracker_params overrides them for bench.py's
 * "converging" side leg only (the product takes the same numbers through dvo_config); NULL / negative = back to the reference's. */
static float g_step_default = 2.0f, g_step_l1 = 1.5f, g_step_l2 = 1.0f, g_min_residual = 5e-3f, g_min_update = 5e-4f;
void orc_set_tracker_params(const float step3[3], float min_residual, float min_update)
{
    g_step_default = step3 ? step3[0] : 2.0f; g_step_l1 = step3 ? step3[1] : 1.5f; g_step_l2 = step3 ? step3[2] : 1.0f;
    g_min_residual = min_residual < 0 ? 5e-3f : min_residual;
    g_min_update = min_update < 0 ? 5e-4f : min_update;
}
static inline float level_step(int level)
{ /* optimize.cpp:22-26 */
    if (level == 1) return g_step_l1;
    if (level == 2) return g_step_l2;
    return g_step_default;
}

/* per-pixel body shared by both variants: returns 1 and fills J[6], r, rw when the pixel contributes */
static int optimize_pixel(const float* obj_gray, const float* gradx, const float* grady, const float* ref_depth,
                          const float* ref_sigma, int w, int h, const float K[9], const float Rt[12],
                          float I2, int x, int y, int level, int crop_enable, float step, float J[6], float* r_out, float* rw_out)
{
    const int i = y * w + x;
    if (crop_enable && level == 2) { /* optimize.cpp:33-36 */
        if (x < 20 || x > 140 || y < 20 || y > 100) return 0;
    }
    const float d = ref_depth[i];
    if ((double)d < 0.20) return 0; /* optimize.cpp:39 (double literal) */
    const float I1 = obj_gray[i];
    if (is_invalid(I1) || is_invalid(I2)) return 0; /* optimize.cpp:44-48 */
    float p[2];
    orc_warp(Rt, (float)x, (float)y, d, K, p); /* optimize.cpp:51 */
    if (p[0] < 0 || p[1] < 0 || (float)w <= p[0] || (float)h <= p[1]) return 0; /* :52-56 */
    const float gx = orc_get_subpixel_dense(gradx, w, h, p[0], p[1]);
    const float gy = orc_get_subpixel_dense(grady, w, h, p[0], p[1]);
    if (is_invalid(gx) || is_invalid(gy)) return 0; /* :61-63 */
    /* Jacobian, optimize.cpp:67-77 (un-warped point) */
    float X[3];
    orc_back_project(K, (float)x, (float)y, d, X);
    const float fx = K[0], fy = K[4];
    const float xx = X[0], yy = X[1], zz = X[2];
    const float fgx = fx * gx, fgy = fy * gy;
    if (g_lit & ORC_LIT_ARITH) { /* optimize.cpp:67-77 as written */
        const float x_ = xx, y_ = yy, z_ = zz;
        const float xz_ = x_ / z_, yz_ = y_ / z_;
        J[0] = fgx / z_;
        J[1] = fgy / z_;
        J[2] = -(fgx * x_ + fgy * y_) / z_ / z_;
        J[3] = -fgx * xz_ * yz_ - fgy * (1.0f + yz_ * yz_);
        J[4] = fgx * (1.0f + xz_ * xz_) + fgy * xz_ * yz_;
        J[5] = (-fgx * yz_ + fgy * xz_);
        goto residual;
    }
    const float iz = 1.0f / zz; /* D8: the six divisions by z of optimize.cpp:70-74 share one reciprocal */
    const float xz = xx * iz, yz = yy * iz;
    J[0] = fgx * iz;
    J[1] = fgy * iz;
    J[2] = ((-fmaf(fgy, yy, fgx * xx)) * iz) * iz;
    J[3] = -(((fgx * xz) * yz) + (fgy * fmaf(yz, yz, 1.0f)));
    J[4] = (fgx * fmaf(xz, xz, 1.0f)) + ((fgy * xz) * yz);
    J[5] = fmaf(fgy, xz, -(fgx * yz));
residual:;
    const float r = I2 - I1; /* :79 */
    float sg = ref_sigma[i]; /* :83 std::clamp(sigma, 0.01, 0.5) */
    sg = sg < 0.01f ? 0.01f : (0.5f < sg ? 0.5f : sg);
    const float wgt = step / sg;
    *r_out = r;
    *rw_out = r * wgt;
    return 1;
}

static void optimize_impl(const float* obj_gray, const float* ref_gray, const float* gradx, const float* grady,
                          const float* ref_depth, const float* ref_sigma, int w, int h, const float K[9],
                          const float xi[6], int level, int crop_enable, int variant, orc_outcome* out, uint8_t* mask)
{
    const float step = level_step(level);
    const int n = w * h;
    memset(out, 0, sizeof *out);
    if (mask) memset(mask, 0, (size_t)n);
    float *A = NULL, *B = NULL, *warped = NULL;
    float Rt[12];
    orc_pose_from_xi(xi, -1.0f, Rt);
    if (variant == 1) { /* faithful: materialise warpImage (optimize.hpp:22,29) and the N x 6 stack (optimize.cpp:17-18) */
        A = (float*)calloc((size_t)n * 6, sizeof(float));
        B = (float*)calloc((size_t)n, sizeof(float));
        warped = (float*)malloc(sizeof(float) * (size_t)n);
        orc_warp_image(xi, ref_gray, ref_depth, w, h, K, warped);
    }
    /* The per-pixel lambda of optimize.cpp:28-90.  One thread (the default): plain raster order.  g_threads > 1: rows are dealt
     * to threads in contiguous blocks, every thread sums into its own partial, partials are added in thread order. */
    const int nthr = g_threads > 1 ? g_threads : 1;
    orc_outcome* part = (orc_outcome*)calloc((size_t)nthr, sizeof(orc_outcome));
#pragma omp parallel num_threads(nthr) if (nthr > 1)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        orc_outcome* acc = &part[tid];
#pragma omp for schedule(static)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int i = y * w + x;
                float I2, J[6], r, rw, Rpix[12];
                const float* P = Rt;
                if (variant == 1) {
                    I2 = warped[i];
                    orc_pose_from_xi(xi, -1.0f, Rpix); /* optimize.cpp:51 -> per-pixel se3::exp */
                    P = Rpix;
                } else {
                    const float d = ref_depth[i];
                    if (is_epsilon(d)) {
                        I2 = ORC_INVALID;
                    } else {
                        float p[2];
                        orc_warp(Rt, (float)x, (float)y, d, K, p);
                        I2 = orc_get_subpixel(ref_gray, w, h, p[0], p[1]);
                    }
                }
                if (!optimize_pixel(obj_gray, gradx, grady, ref_depth, ref_sigma, w, h, K, P, I2, x, y, level, crop_enable, step, J, &r, &rw))
                    continue;
                acc->n_valid++;
                if (mask) mask[i] = 1;
                acc->sum_r2 += (double)r * (double)r;
                int k = 0;
                for (int a = 0; a < 6; a++) {
                    for (int b = a; b < 6; b++) acc->H[k++] += (double)J[a] * (double)J[b];
                    acc->g[a] += (double)J[a] * (double)rw;
                }
                if (variant == 1) {
                    for (int a = 0; a < 6; a++) A[6 * i + a] = J[a];
                    B[i] = rw;
                }
            }
    }
    *out = part[0];  /* (one thread: exactly the sequential sums) */
    for (int t = 1; t < nthr; t++) {
        out->n_valid += part[t].n_valid;
        out->sum_r2 += part[t].sum_r2;
        for (int k = 0; k < 21; k++) out->H[k] += part[t].H[k];
        for (int k = 0; k < 6; k++) out->g[k] += part[t].g[k];
    }
    free(part);
    if (out->n_valid == 0) { /* optimize.cpp:92-93 */
        out->residual = -1.0f;
    } else {
        if (variant == 1)
            orc_lsq_svd(A, B, n, out->xi_update);
        else
            orc_solve6(out->H, out->g, out->xi_update);
        out->residual = (float)out->sum_r2 / (float)out->n_valid; /* optimize.cpp:98 */
    }
    free(A); free(B); free(warped);
}

void orc_optimize(const float* obj_gray, const float* ref_gray, const float* ref_depth, const float* ref_sigma,
                  int w, int h, const float K[9], const float xi[6], int level, int crop_enable, int variant,
                  orc_outcome* out, uint8_t* mask)
{
    float* gx = (float*)malloc(sizeof(float) * (size_t)w * h);
    float* gy = (float*)malloc(sizeof(float) * (size_t)w * h);
    orc_gradiate(ref_gray, w, h, 1, gx);
    orc_gradiate(ref_gray, w, h, 0, gy);
    optimize_impl(obj_gray, ref_gray, gx, gy, ref_depth, ref_sigma, w, h, K, xi, level, crop_enable, variant, out, mask);
    free(gx); free(gy);
}

/* ======================================================================== */
/* Frame / pyramid.  include/system/frame.hpp, src/system/frame.cpp          */
/* ======================================================================== */
static float* falloc(int n) { return (float*)malloc(sizeof(float) * (size_t)n); }

static void redecimate(orc_frame* f, float** maps, const float* top)
{ /* frame.cpp:39-61: level i = cullImage(top, levels-1-i) */
    const int L = f->levels, tw = f->w[L - 1], th = f->h[L - 1];
    for (int i = 0; i < L; i++) orc_cull_image(top, tw, th, L - 1 - i, maps[i]);
}

orc_frame* orc_frame_create(const float* gray, const float* depth, const float* sigma, int w, int h,
                            const float K[9], int levels, int culls, int id)
{ /* frame.hpp:91-117 + frame.cpp:16-37 */
    orc_frame* f = (orc_frame*)calloc(1, sizeof *f);
    f->levels = levels; f->culls = culls; f->id = id; f->ref_index = -1;
    const int bw = w >> culls, bh = h >> culls;
    float Kb[9];
    orc_cull_intrinsic(K, culls, Kb);
    float* bg = falloc(bw * bh);
    float* bd = falloc(bw * bh);
    float* bs = falloc(bw * bh);
    orc_cull_image(gray, w, h, culls, bg);
    if (depth) orc_cull_image(depth, w, h, culls, bd);
    else for (int i = 0; i < bw * bh; i++) bd[i] = 0.0f; /* D6: mono depth is an explicit input */
    if (sigma) orc_cull_image(sigma, w, h, culls, bs);
    else for (int i = 0; i < bw * bh; i++) bs[i] = 0.5f; /* frame.hpp:21 */
    for (int i = 0; i < levels; i++) {
        const int t = levels - 1 - i;
        f->w[i] = bw >> t; f->h[i] = bh >> t;
        orc_cull_intrinsic(Kb, t, f->K[i]);
        f->gray[i] = falloc(f->w[i] * f->h[i]);
        f->depth[i] = falloc(f->w[i] * f->h[i]);
        f->sigma[i] = falloc(f->w[i] * f->h[i]);
        orc_cull_image(bg, bw, bh, t, f->gray[i]);
        orc_cull_image(bd, bw, bh, t, f->depth[i]);
        orc_cull_image(bs, bw, bh, t, f->sigma[i]);
    }
    f->age = (float*)calloc((size_t)bw * bh, sizeof(float)); /* frame.hpp:105 */
    free(bg); free(bd); free(bs);
    return f;
}

void orc_frame_destroy(orc_frame* f)
{
    if (!f) return;
    for (int i = 0; i < f->levels; i++) { free(f->gray[i]); free(f->depth[i]); free(f->sigma[i]); }
    free(f->age);
    free(f);
}

void orc_frame_update_depth_sigma_age(orc_frame* f, const float* d, const float* s, const float* a)
{ /* frame.cpp:47-54 */
    const int L = f->levels;
    memcpy(f->age, a, sizeof(float) * (size_t)f->w[L - 1] * f->h[L - 1]);
    redecimate(f, f->depth, d);
    redecimate(f, f->sigma, s);
}
void orc_frame_update_depth_sigma(orc_frame* f, const float* d, const float* s)
{ /* frame.cpp:39-45 */
    redecimate(f, f->depth, d);
    redecimate(f, f->sigma, s);
}
void orc_frame_update_depth(orc_frame* f, const float* d)
{ /* frame.cpp:56-61 */
    redecimate(f, f->depth, d);
}

/* ======================================================================== */
/* Tracker::track.  src/track/tracker.cpp:22-85                              */
/* ======================================================================== */
void orc_track(const orc_frame* obj, const orc_frame* ref, int crop_enable, int variant, int fixed_iters,
               float xi_out[6], orc_track_log* log)
{
    float xi[6] = {0, 0, 0, 0, 0, 0}; /* tracker.cpp:28 */
    if (log) memset(log, 0, sizeof *log);
    for (int level = 0; level < ref->levels; level++) { /* tracker.cpp:32 */
        const int w = ref->w[level], h = ref->h[level];
        float* gx = falloc(w * h);
        float* gy = falloc(w * h);
        orc_gradiate(ref->gray[level], w, h, 1, gx); /* frame.hpp:52-63 (lazy, once per ref scene) */
        orc_gradiate(ref->gray[level], w, h, 0, gy);
        const int max_it = fixed_iters > 0 ? fixed_iters : ORC_MAX_ITER;
        for (int it = 0; it < max_it && it < ORC_MAX_ITER; it++) { /* tracker.cpp:42 */
            orc_outcome o;
            optimize_impl(obj->gray[level], ref->gray[level], gx, gy, ref->depth[level], ref->sigma[level],
                          w, h, ref->K[level], xi, level, crop_enable, variant, &o, NULL);
            float upd[6];
            if (g_nudge && level == 0 && it == 0) {
                for (int k = 0; k < (g_nudge < 0 ? -g_nudge : g_nudge); k++)
                    o.xi_update[0] = nextafterf(o.xi_update[0], g_nudge > 0 ? INFINITY : -INFINITY);
            }
            if (g_lit & ORC_LIT_SE3) orc_se3_concatenate_f32lit(xi, o.xi_update, upd);
            else orc_se3_concatenate(xi, o.xi_update, upd); /* tracker.cpp:46 */
            int ok = 1;
            for (int i = 0; i < 6; i++) if (isnan(upd[i])) ok = 0; /* tracker.cpp:47-51 testXi */
            if (ok) memcpy(xi, upd, sizeof xi);
            double nrm = 0;
            for (int i = 0; i < 6; i++) nrm += (double)o.xi_update[i] * (double)o.xi_update[i];
            nrm = sqrt(nrm);
            if (log) {
                log->n_iter[level] = it + 1;
                log->residual[level][it] = o.residual;
                log->upd_norm[level][it] = (float)nrm;
                log->n_valid[level][it] = o.n_valid;
                memcpy(log->xi_after[level][it], xi, sizeof xi);
            }
            if (fixed_iters > 0) continue;
            if (nrm < (double)g_min_update || o.residual < g_min_residual) break; /* tracker.cpp:68-69 (time stop disabled, D1) */
        }
        free(gx); free(gy);
    }
    memcpy(xi_out, xi, sizeof xi);
}

/* ======================================================================== */
/* Gaussian.  src/math/gaussian.cpp                                          */
/* ======================================================================== */
static inline uint32_t mix32(uint32_t x)
{ /* lowbias32 */
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

float orc_rng_depth(uint32_t seed, uint32_t frame_id, uint32_t pixel)
{ /* D3: uniform_real_distribution<float>(2.0, 0.5) -> a + (b-a)*u, then min(.,4) (gaussian.cpp:9,22) */
    const uint32_t hsh = mix32(mix32(seed ^ (frame_id * 0x9E3779B9U)) ^ (pixel * 0x85EBCA6BU) ^ 0x68E31DA4U);
    const float u = (float)(hsh >> 8) * (1.0f / 16777216.0f);
    const float v = fmaf(u, -1.5f, 2.0f);
    return v < 4.0f ? v : 4.0f;
}

static inline float gauss_gain(float d, float diff)
{ /* gaussian.cpp:20 */
    const float m = d < diff ? d : diff; /* std::min(d, diff) */
    if ((double)m < 0.8) return 0.5f + (m / 0.8f) * 0.5f;
    return 1.0f;
}

int orc_gaussian_update(float* depth, float* sigma, float d, float s, float reset_depth)
{ /* gaussian.cpp:12-31 */
    const float v1 = (*sigma) * (*sigma), v2 = s * s, v = v1 + v2;
    const float diff = fabsf(d - *depth);
    const float gain = gauss_gain(d, diff);
    const float ms = *sigma < s ? s : *sigma; /* std::max(sigma, s) */
    if (diff > gain * ms) {
        *depth = reset_depth;
        *sigma = 0.5f;
        return 0;
    }
    *depth = (g_lit & ORC_LIT_ARITH) ? (v2 * (*depth) + v1 * d) / v : fmaf(v1, d, v2 * (*depth)) / v;
    *sigma = sqrtf((v1 * v2) / v);
    return 1;
}

int orc_gaussian_fuse(float* depth, float* sigma, float d, float s)
{ /* gaussian.cpp:33-50 */
    const float v1 = (*sigma) * (*sigma), v2 = s * s, v = v1 + v2;
    const float diff = fabsf(d - *depth);
    const float gain = gauss_gain(d, diff);
    const float ms = *sigma < s ? s : *sigma;
    if (diff > gain * ms) return 0;
    *depth = (g_lit & ORC_LIT_ARITH) ? (v2 * (*depth) + v1 * d) / v : fmaf(v1, d, v2 * (*depth)) / v;
    *sigma = sqrtf((v1 * v2) / v);
    return 1;
}

/* ======================================================================== */
/* Map::Implement.  src/map/implement.cpp                                    */
/* ======================================================================== */
/* cvRound (round half to even) of a coordinate, with D4 */
static inline int round_coord(float v, int* out)
{
    if (!coord_ok(v)) return 0;
    *out = (int)lrintf(v);
    return 1;
}

void orc_propagate(const float* ref_depth, const float* ref_sigma, const float* ref_age, int w, int h,
                   const float xi[6], const float K[9], float* depth, float* sigma, float* age)
{ /* implement.cpp:217-256; raster order => last writer wins (D7) */
    const float tz = xi[2]; /* implement.cpp:224: the twist component, not the matrix translation */
    const float pv = 0.06f * 0.06f; /* implement.cpp:17-18 */
    float Rt[12];
    orc_pose_from_xi(xi, 1.0f, Rt);
    for (int i = 0; i < w * h; i++) { depth[i] = 1.0f; sigma[i] = 1.0f; age[i] = 0.0f; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int i = y * w + x;
            const float rd = ref_depth[i];
            if (is_epsilon(rd)) continue;
            float p[2];
            orc_warp(Rt, (float)x, (float)y, rd, K, p);
            int qx, qy;
            if (!round_coord(p[0], &qx) || !round_coord(p[1], &qy)) continue;
            if (!in_range(qx, qy, w, h)) continue;
            float s = ref_sigma[i];
            const float d0 = rd < 0.01f ? 0.01f : rd; /* std::max(rd, 0.01f) */
            const float d1 = d0 + tz;
            const float q = d1 / d0;
            const float q4 = q * (q * (q * q)); /* math::pow(q,4): util.hpp:19-27 */
            s = (g_lit & ORC_LIT_ARITH) ? sqrtf(q4 * (s * s) + pv) : sqrtf(fmaf(q4, s * s, pv));
            const int o = qy * w + qx;
            depth[o] = d1 < 0.0f ? 0.0f : d1; /* std::max(d1, 0) */
            sigma[o] = s;
            age[o] = ref_age[i] + 1.0f;
        }
}

void orc_regularize(const float* depth, const float* sigma, int w, int h, float* out)
{ /* implement.cpp:156-180: reads the OLD maps (Jacobi style), order L, R, D, U */
    static const int off[4][2] = {{-1, 0}, {1, 0}, {0, 1}, {0, -1}}; /* (dx, dy) of offsets :160 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float gd = depth[y * w + x], gs = sigma[y * w + x];
            for (int k = 0; k < 4; k++) {
                const int nx = x + off[k][0], ny = y + off[k][1];
                if (!in_range(nx, ny, w, h)) continue;
                orc_gaussian_fuse(&gd, &gs, depth[ny * w + nx], sigma[ny * w + nx]);
            }
            out[y * w + x] = gd < 6.0f ? gd : 6.0f; /* implement.cpp:178 cv::min(.,6.0) */
        }
}

void orc_implement_update(const float* obj_gray, const float* born_gray, const float* born_gx, const float* born_gy,
                          int w, int h, const float r_xi[6], const float K[9], int qx, int qy,
                          float depth, float sigma, float* new_depth, float* new_sigma)
{ /* implement.cpp:182-214 */
    *new_depth = -1.0f; *new_sigma = -1.0f;
    /* EpipolarSegment(-relative_xi, x_i, K, depth, sigma), implement.cpp:23-47 */
    float Rt[12];
    orc_pose_from_xi(r_xi, -1.0f, Rt);
    const float dmin = (depth - sigma) < 0.10f ? 0.10f : (depth - sigma);
    const float dmax = depth + sigma;
    float start[2], end[2];
    orc_warp(Rt, (float)qx, (float)qy, dmax, K, start);
    orc_warp(Rt, (float)qx, (float)qy, dmin, K, end);
    const float sex = start[0] - end[0], sey = start[1] - end[1];
    const float length = (float)sqrt((double)sex * sex + (double)sey * sey); /* cv::norm(Point2f) in double */
    /* doMatching, implement.cpp:106-152 */
    const float og = obj_gray[qy * w + qx];
    const float dirx = (end[0] - start[0]) / length, diry = (end[1] - start[1]) / length;
    float ptx = start[0], pty = start[1];
    float bestx = ptx, besty = pty;
    float min_ssd = 6.0f; /* 2.0f * N */
    int count = 0;
    for (;;) {
        const float ddx = ptx - start[0], ddy = pty - start[1];
        if (!(sqrt((double)ddx * ddx + (double)ddy * ddy) < (double)length)) break;
        float ssd = 0;
        ptx += dirx; pty += diry;
        for (int i = 0; i < 3; i++) {
            const float k = (float)(i - 1);
            const float tx = ptx + dirx * k, ty = pty + diry * k;
            const float sg = orc_get_subpixel_dense(born_gray, w, h, tx, ty);
            if (is_invalid(sg)) { ssd = 6.0f; break; }
            const float diff = sg - og;
            const int aw = 3 - abs(i - 2); /* N - |i - center|, center = (N+1)/2 = 2 */
            ssd = (float)((double)ssd + 1.0 * aw / 3 * (double)(diff * diff));
        }
        if (ssd < min_ssd) { bestx = ptx; besty = pty; min_ssd = ssd; }
        if (count++ > 100) break;
    }
    if ((double)min_ssd > 3 * 0.1) return; /* N * MATCHING_THRESHOLD_RATIO */
    /* implement.cpp:196-200 */
    if (bestx < 0 || besty < 0 || bestx > (float)w || besty > (float)h) return;
    /* depthEstimate, implement.cpp:49-71 -- evaluated in double from float inputs */
    {
        /* x_q = backProject(K, q, 1) in float first, as the reference does */
        float Xq[3];
        orc_back_project(K, (float)qx, (float)qy, 1.0f, Xq);
        const double q0 = Xq[0], q1 = Xq[1], q2 = Xq[2];
        const double t[3] = {-(double)r_xi[0], -(double)r_xi[1], -(double)r_xi[2]}; /* twist part, implement.cpp:56 */
        const double xi3[3] = {bestx, besty, 1.0};
        double Rq[3], KRq[3], Kt[3];
        for (int r = 0; r < 3; r++) Rq[r] = (double)Rt[3 * r] * q0 + (double)Rt[3 * r + 1] * q1 + (double)Rt[3 * r + 2] * q2;
        for (int r = 0; r < 3; r++) {
            KRq[r] = (double)K[3 * r] * Rq[0] + (double)K[3 * r + 1] * Rq[1] + (double)K[3 * r + 2] * Rq[2];
            Kt[r] = (double)K[3 * r] * t[0] + (double)K[3 * r + 1] * t[1] + (double)K[3 * r + 2] * t[2];
        }
        const double r3q = Rq[2];
        double aa = 0, ab = 0;
        for (int r = 0; r < 3; r++) {
            const double a = r3q * xi3[r] - KRq[r];
            const double b = t[2] * xi3[r] - Kt[r];
            aa += a * a; ab += a * b;
        }
        *new_depth = -(float)(ab / aa);
    }
    /* sigmaEstimate, implement.cpp:73-104 */
    {
        const float l = length;
        const float lx = sex / l, ly = sey / l;
        const float alpha = (dmax - dmin) / l;
        int mx = 0, my = 0;
        round_coord(bestx, &mx); round_coord(besty, &my); /* Mat_(Point2f) -> cvRound */
        mx = mx < 0 ? 0 : (mx > w - 1 ? w - 1 : mx); /* D5 clamp */
        my = my < 0 ? 0 : (my > h - 1 ? h - 1 : my);
        const float gx = born_gx[my * w + mx], gy = born_gy[my * w + mx];
        if (is_invalid(gx) || is_invalid(gy)) { *new_sigma = -1.0f; return; }
        const float gl = (g_lit & ORC_LIT_ARITH) ? fabsf(gx * lx + gy * ly) : fabsf(fmaf(gy, ly, gx * lx));
        const float gl2 = gl * gl;
        const float gp2 = gl / l;
        const float epi = 0.25f / (gl2 < ORC_EPSILON ? ORC_EPSILON : gl2);
        const float lum = 0.5f / (gp2 < ORC_EPSILON ? ORC_EPSILON : gp2);
        *new_sigma = alpha * sqrtf(epi + lum);
    }
}

/* ======================================================================== */
/* Map::Mapper + System::VisualOdometry                                      */
/* ======================================================================== */
struct orc_vo {
    float K[9];
    int w, h;
    uint32_t seed;
    int crop_enable, variant;
    orc_frame** hist; int n_hist, cap_hist;
    orc_frame* depth_ref;    /* m_ref_frame of odometrizeUsingDepth, system.hpp:103 */
    orc_frame* last;         /* last non-keyframe obj frame (kept for inspection) */
    int latest_id;           /* frame.cpp:5 */
    float* init_depth; float* init_sigma;
    int last_valid_updates;
};

orc_vo* orc_vo_create(const float K[9], int w, int h, uint32_t rng_seed, int crop_enable, int variant)
{
    orc_vo* vo = (orc_vo*)calloc(1, sizeof *vo);
    memcpy(vo->K, K, sizeof vo->K);
    vo->w = w; vo->h = h; vo->seed = rng_seed; vo->crop_enable = crop_enable; vo->variant = variant;
    vo->latest_id = -1;
    return vo;
}

void orc_vo_destroy(orc_vo* vo)
{
    if (!vo) return;
    for (int i = 0; i < vo->n_hist; i++) orc_frame_destroy(vo->hist[i]);
    free(vo->hist);
    if (vo->last) orc_frame_destroy(vo->last);
    if (vo->depth_ref) orc_frame_destroy(vo->depth_ref);
    free(vo->init_depth); free(vo->init_sigma);
    free(vo);
}

static void hist_push(orc_vo* vo, orc_frame* f)
{ /* frame.hpp:151-157 */
    if (vo->n_hist == vo->cap_hist) {
        vo->cap_hist = vo->cap_hist ? vo->cap_hist * 2 : 16;
        vo->hist = (orc_frame**)realloc(vo->hist, sizeof(orc_frame*) * (size_t)vo->cap_hist);
    }
    vo->hist[vo->n_hist++] = f;
}

void orc_vo_set_initial_depth(orc_vo* vo, const float* depth, const float* sigma)
{
    const int n = (vo->w >> 2) * (vo->h >> 2); /* culls = 2, system.hpp:47 */
    free(vo->init_depth); free(vo->init_sigma);
    vo->init_depth = falloc(n); vo->init_sigma = falloc(n);
    memcpy(vo->init_depth, depth, sizeof(float) * (size_t)n);
    memcpy(vo->init_sigma, sigma, sizeof(float) * (size_t)n);
}

void orc_vo_init_keyframe(orc_vo* vo, const float* gray, const float* depth, const float* sigma)
{ /* system.hpp:24-32, with the mono pyramid geometry (3,2) instead of (4,1): D9 */
    orc_frame* f = orc_frame_create(gray, depth, sigma, vo->w, vo->h, vo->K, 3, 2, ++vo->latest_id);
    hist_push(vo, f);
}

int orc_need_new_frame(const float rel_xi[6], int id, int ref_id)
{ /* mapper.cpp:45-60 */
    const double n = sqrt((double)rel_xi[0] * rel_xi[0] + (double)rel_xi[1] * rel_xi[1] + (double)rel_xi[2] * rel_xi[2]);
    if (n > (double)0.02f) return 1;
    if (id - ref_id >= 6) return 1;
    return 0;
}

static void frame_top_gradients(const orc_frame* f, float** gx, float** gy)
{
    const int L = f->levels, w = f->w[L - 1], h = f->h[L - 1];
    *gx = falloc(w * h); *gy = falloc(w * h);
    orc_gradiate(f->gray[L - 1], w, h, 1, *gx);
    orc_gradiate(f->gray[L - 1], w, h, 0, *gy);
}

int orc_mapper_update(orc_frame** history, int n_hist, const orc_frame* obj, uint32_t rng_seed)
{ /* mapper.cpp:76-137 */
    orc_frame* ref = history[n_hist - 1];
    const int L = ref->levels, w = ref->w[L - 1], h = ref->h[L - 1];
    const float* K = obj->K[obj->levels - 1];
    float* rd = ref->depth[L - 1];
    float* rs = ref->sigma[L - 1];
    float Rt[12];
    orc_pose_from_xi(obj->rel_xi, 1.0f, Rt);
    float** gxs = (float**)calloc((size_t)n_hist, sizeof(float*));
    float** gys = (float**)calloc((size_t)n_hist, sizeof(float*));
    int valid_update = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (x < 16 || x > 144 || y < 12 || y > 108) continue; /* mapper.cpp:90 */
            const int i = y * w + x;
            const float d = rd[i];
            float p[2];
            orc_warp(Rt, (float)x, (float)y, d, K, p); /* mapper.cpp:94 */
            int qx, qy;
            if (!round_coord(p[0], &qx) || !round_coord(p[1], &qy)) continue;
            if (!in_range(qx, qy, w, h)) continue;
            const int age = (int)ref->age[i]; /* mapper.cpp:99 */
            const int bi = n_hist - 1 - age;  /* frame.hpp:176 */
            if (bi < 0 || bi >= n_hist) continue; /* .at() would throw in the reference */
            orc_frame* born = history[bi];
            const float depth = d - obj->rel_xi[2]; /* mapper.cpp:104 */
            const float sigma = rs[i];
            float nb[6], r_xi[6];
            for (int k = 0; k < 6; k++) nb[k] = -born->xi[k];
            orc_se3_concatenate(obj->xi, nb, r_xi); /* mapper.cpp:107 */
            if (!gxs[bi]) frame_top_gradients(born, &gxs[bi], &gys[bi]);
            float nd, ns;
            orc_implement_update(obj->gray[obj->levels - 1], born->gray[born->levels - 1], gxs[bi], gys[bi],
                                 w, h, r_xi, K, qx, qy, depth, sigma, &nd, &ns);
            if (nd > 0.2f && nd < 6.0f && ns > 0.0f && ns < 0.5f) { /* mapper.cpp:122 */
                float gd = depth, gs = sigma;
                const float reset = orc_rng_depth(rng_seed, (uint32_t)obj->id, (uint32_t)i);
                if (!orc_gaussian_update(&gd, &gs, nd, ns, reset)) ref->age[i] = 0.0f; /* mapper.cpp:124-127 */
                else valid_update++;
                rd[i] = gd; rs[i] = gs; /* mapper.cpp:130-131 */
            }
        }
    for (int k = 0; k < n_hist; k++) { free(gxs[k]); free(gys[k]); }
    free(gxs); free(gys);
    /* mapper.cpp:135: ref->updateDepthSigma(ref->depth(), ref->sigma()) */
    {
        float* d = falloc(w * h); float* s = falloc(w * h);
        memcpy(d, rd, sizeof(float) * (size_t)w * h);
        memcpy(s, rs, sizeof(float) * (size_t)w * h);
        orc_frame_update_depth_sigma(ref, d, s);
        free(d); free(s);
    }
    return valid_update;
}

static void mapper_regularize(orc_frame* f)
{ /* mapper.cpp:139-144 */
    const int L = f->levels, w = f->w[L - 1], h = f->h[L - 1];
    float* nd = falloc(w * h);
    orc_regularize(f->depth[L - 1], f->sigma[L - 1], w, h, nd);
    orc_frame_update_depth(f, nd);
    free(nd);
}

int orc_vo_odometrize(orc_vo* vo, const float* gray, float T_world[16])
{ /* system.hpp:44-74 */
    orc_frame* frame = orc_frame_create(gray, NULL, NULL, vo->w, vo->h, vo->K, 3, 2, ++vo->latest_id);
    if (vo->n_hist == 0) {
        if (vo->init_depth) orc_frame_update_depth_sigma(frame, vo->init_depth, vo->init_sigma);
        hist_push(vo, frame);
        const float z[6] = {0, 0, 0, 0, 0, 0};
        orc_se3_exp(z, T_world);
        return 1;
    }
    orc_frame* ref = vo->hist[vo->n_hist - 1];
    float rel[6];
    orc_track(frame, ref, vo->crop_enable, vo->variant, 0, rel, NULL); /* system.hpp:57 */
    memcpy(frame->rel_xi, rel, sizeof rel);                             /* frame.cpp:7-14 */
    frame->ref_index = vo->n_hist - 1;
    orc_se3_concatenate(ref->xi, rel, frame->xi);
    int is_key = 0;
    /* Mapper::estimate, mapper.cpp:16-33 */
    if (orc_need_new_frame(rel, frame->id, ref->id)) {
        const int L = ref->levels, w = ref->w[L - 1], h = ref->h[L - 1];
        float* d = falloc(w * h); float* s = falloc(w * h); float* a = falloc(w * h);
        orc_propagate(ref->depth[L - 1], ref->sigma[L - 1], ref->age, w, h, rel, frame->K[frame->levels - 1], d, s, a); /* mapper.cpp:62-74 */
        orc_frame_update_depth_sigma_age(frame, d, s, a);
        free(d); free(s); free(a);
        hist_push(vo, frame);
        is_key = 1;
    } else {
        vo->last_valid_updates = orc_mapper_update(vo->hist, vo->n_hist, frame, vo->seed);
    }
    mapper_regularize(vo->hist[vo->n_hist - 1]); /* mapper.cpp:26,30 */
    orc_se3_exp(frame->xi, T_world);              /* system.hpp:73 */
    if (!is_key) {
        if (vo->last) orc_frame_destroy(vo->last);
        vo->last = frame;
    }
    return is_key;
}

void orc_vo_odometrize_depth(orc_vo* vo, const float* gray, const float* depth, const float* sigma, float T_rel[16])
{ /* system.hpp:77-93 */
    orc_frame* frame = orc_frame_create(gray, depth, sigma, vo->w, vo->h, vo->K, 4, 1, ++vo->latest_id);
    const float z[6] = {0, 0, 0, 0, 0, 0};
    if (!vo->depth_ref) {
        vo->depth_ref = frame;
        orc_se3_exp(z, T_rel);
        return;
    }
    float rel[6];
    orc_track(frame, vo->depth_ref, vo->crop_enable, vo->variant, 0, rel, NULL);
    memcpy(frame->rel_xi, rel, sizeof rel);
    orc_se3_concatenate(vo->depth_ref->xi, rel, frame->xi);
    orc_frame_destroy(vo->depth_ref);
    vo->depth_ref = frame;
    orc_se3_exp(rel, T_rel);
}

int orc_vo_keyframe_count(const orc_vo* vo) { return vo->n_hist; }
const orc_frame* orc_vo_keyframe(const orc_vo* vo, int i) { return (i >= 0 && i < vo->n_hist) ? vo->hist[i] : NULL; }
const orc_frame* orc_vo_last_frame(const orc_vo* vo) { return vo->last; }
int orc_vo_last_valid_updates(const orc_vo* vo) { return vo->last_valid_updates; }
