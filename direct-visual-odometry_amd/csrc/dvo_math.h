// dvo_math.h -- per-pixel arithmetic and SE(3) algebra shared by the HIP kernels and the C++ host code.
//
// The float operation order written here IS the contract of DESIGN.md §3: every multi-term sum is an
// explicit fmaf chain and the translation unit is compiled with -ffp-contract=off, so the CDNA4 VALU
// and an x86 host evaluate bit-identical per-pixel values.  SE(3) exp/log run in double and are rounded
// to float once (deviation D2).  Reference citations are file:line under the reference tree.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DVO_HD __host__ __device__ __forceinline__
#else
#define DVO_HD inline
#endif

namespace dvo {

constexpr float kInvalid = -2.0f;  // include/math/util.hpp:7
constexpr float kEpsilon = 1e-6f;  // include/math/util.hpp:6

struct Intr {  // fx, fy, cx, cy of a pyramid level (Convert::cullIntrinsic, src/core/convert.cpp:22-29)
    float fx, fy, cx, cy;
    float ifx, ify;  // 1.0f / fx, 1.0f / fy: correctly rounded once per level
};
DVO_HD Intr make_intr(const float K[9])
{
    Intr k;
    k.fx = K[0]; k.fy = K[4]; k.cx = K[2]; k.cy = K[5];
    k.ifx = 1.0f / k.fx; k.ify = 1.0f / k.fy;
    return k;
}

struct Pose {  // R (row major) and t of exp(+-xi), rounded to float once
    float R[9];
    float t[3];
};

DVO_HD bool is_valid(float v) { return kInvalid < v; }      // util.hpp:9
DVO_HD bool is_invalid(float v) { return v <= kInvalid; }   // util.hpp:10
DVO_HD bool is_epsilon(float v) { return fabsf(v) < kEpsilon; }  // util.hpp:29-32
DVO_HD bool coord_ok(float v) { return fabsf(v) < 1073741824.0f; }  // D4: false for NaN/inf/huge

// ---------------------------------------------------------------- geometry, src/core/transform.cpp:20-33
DVO_HD void back_project(const Intr& k, float px, float py, float d, float& X, float& Y, float& Z)
{
    // depth * (p - c) / f (transform.cpp:27): the division by the per-level constant f is a multiplication by its
    // correctly rounded reciprocal -- the reference is built -Ofast (-freciprocal-math), see DESIGN.md §3 D8
    X = (d * (px - k.cx)) * k.ifx;
    Y = (d * (py - k.cy)) * k.ify;
    Z = d;
}

DVO_HD void transform(const Pose& p, float X, float Y, float Z, float& Xo, float& Yo, float& Zo)
{
    Xo = fmaf(p.R[0], X, fmaf(p.R[1], Y, fmaf(p.R[2], Z, p.t[0])));
    Yo = fmaf(p.R[3], X, fmaf(p.R[4], Y, fmaf(p.R[5], Z, p.t[1])));
    Zo = fmaf(p.R[6], X, fmaf(p.R[7], Y, fmaf(p.R[8], Z, p.t[2])));
}

// Correctly rounded 1/z.  On the device: v_rcp_f32 (1 ulp) + one Newton step (two FMAs) for |z| inside
// [2^-100, 2^100] -- 3 instructions instead of the 11 of the IEEE division sequence -- and the IEEE division
// outside that range or for non-finite z (never the case for a real depth).  The fast branch equals 1.0f / z for EVERY
// float in its range: checked exhaustively on the device by dvo_selftest_reciprocal (tests/test_gpu_parity.py), so the
// result is the same bits as the oracle's x86 division.
#define DVO_RECIP_FAST_MIN 7.888609052210118e-31f   /* 2^-100 */
#define DVO_RECIP_FAST_MAX 1.2676506002282294e30f   /* 2^100  */
#if defined(__HIPCC__)   // (both passes of hipcc: the host pass parses the kernels that call it)
__device__ __forceinline__ float recip_fast(float z)  // valid for DVO_RECIP_FAST_MIN <= |z| <= DVO_RECIP_FAST_MAX
{
    float r = __builtin_amdgcn_rcpf(z);
    r = fmaf(fmaf(-z, r, 1.0f), r, r);   // ONE Newton step is already the correctly rounded reciprocal of every float in range on gfx950
    return r;                            // (enumerated: dvo_selftest_reciprocal; rounds 1-2 carried a second, redundant step)
}
#endif
DVO_HD float recip_rn(float z)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r = recip_fast(z);
    const float az = fabsf(z);
    const bool out_of_range = !(az >= DVO_RECIP_FAST_MIN && az <= DVO_RECIP_FAST_MAX);  // (true for NaN)
    if (__builtin_amdgcn_ballot_w64(out_of_range) != 0ull) {  // wave-uniform; never taken for real depths
        asm volatile("; recip_rn: IEEE division fallback");   // (keeps this a branch: the compiler would otherwise
        if (out_of_range) r = 1.0f / z;                       //  evaluate both forms and select)
    }
    return r;
#else
    return 1.0f / z;
#endif
}

DVO_HD void project(const Intr& k, float X, float Y, float Z, float& u, float& v)
{
    const float iz = recip_rn(Z);  // one correctly rounded reciprocal shared by both coordinates (D8)
    u = (X * k.fx) * iz + k.cx;
    v = (Y * k.fy) * iz + k.cy;
}

DVO_HD void warp(const Pose& p, const Intr& k, float px, float py, float d, float& u, float& v)
{
    float X, Y, Z, Xw, Yw, Zw;
    back_project(k, px, py, d, X, Y, Z);
    transform(p, X, Y, Z, Xw, Yw, Zw);
    project(k, Xw, Yw, Zw, u, v);
}

// ---------------------------------------------------------------- sampling, src/core/convert.cpp
DVO_HD float blend4(float g0, float g1, float g2, float g3, float hx, float vy)
{
    const float omh = 1.0f - hx, omv = 1.0f - vy;
    const float top = fmaf(g1, hx, g0 * omh);
    const float bot = fmaf(g3, hx, g2 * omh);
    return fmaf(bot, vy, top * omv);
}

// Convert::getPixel semantics (convert.cpp:107-125) for an in-range index: invalid (or NaN) -> INVALID
DVO_HD float pass_valid(float g) { return is_valid(g) ? g : kInvalid; }

// The fill loop of Convert::getSubpixel (convert.cpp:155-173), literally.  Returns false for "all invalid".
DVO_HD bool fill_quirk(float g[4])
{
    if (is_valid(g[0]) && is_valid(g[1]) && is_valid(g[2]) && is_valid(g[3])) return true;  // common case
    int valid = 0, id = 0;
    float last = -1.0f;
    for (int guard = 0; guard < 16; ++guard) {
        if (is_valid(g[id])) {
            valid++;
            last = g[id];
        } else if (last > 0.0f) {
            g[id] = last;
            valid++;
        }
        if (valid == 4) return true;
        if (id == 3 && valid == 0) return false;
        id = (id + 1) & 3;
    }
    return true;  // unreachable: every pass adds at least one
}

// Generic image accessor used by the samplers: Img must provide  float at(int x, int y) const, int w, h.
struct GlobalImg {
    const float* p;
    int w, h;
    DVO_HD float at(int x, int y) const { return p[y * w + x]; }
};

template <class Img>
DVO_HD bool load_taps(const Img& img, float px, float py, float g[4], float& hx, float& vy)
{  // convert.cpp:82-101 / 133-153: truncation toward zero, missing taps clamp to g00
    if (!coord_ok(px) || !coord_ok(py)) return false;
    const int x0 = (int)px, y0 = (int)py;
    if (x0 < 0 || img.w <= x0 || y0 < 0 || img.h <= y0) return false;
    const int x1 = x0 + 1, y1 = y0 + 1;
    hx = px - (float)x0;
    vy = py - (float)y0;
    const float g00 = img.at(x0, y0);
    g[0] = g00;
    g[1] = (x1 < img.w) ? img.at(x1, y0) : g00;
    g[2] = (y1 < img.h) ? img.at(x0, y1) : g00;
    g[3] = (x1 < img.w && y1 < img.h) ? img.at(x1, y1) : g00;
    return true;
}

template <class Img>
DVO_HD float get_subpixel(const Img& img, float px, float py)
{  // convert.cpp:128-177
    float g[4], hx, vy;
    if (!load_taps(img, px, py, g, hx, vy)) return kInvalid;
    if (!fill_quirk(g)) return kInvalid;
    return blend4(g[0], g[1], g[2], g[3], hx, vy);
}

template <class Img>
DVO_HD float get_subpixel_dense(const Img& img, float px, float py)
{  // convert.cpp:77-105
    float g[4], hx, vy;
    if (!load_taps(img, px, py, g, hx, vy)) return kInvalid;
    return blend4(g[0], g[1], g[2], g[3], hx, vy);
}

// Convert::gradiate at one pixel (convert.cpp:41-75), derived from the gray image on the fly.
template <class Img>
DVO_HD float grad_x_at(const Img& img, int x, int y)
{
    if (x - 1 <= -1 || x + 1 >= img.w) return kInvalid;
    const float a = img.at(x - 1, y), b = img.at(x + 1, y);
    if (!is_valid(a) || !is_valid(b)) return kInvalid;
    return b - a;
}
template <class Img>
DVO_HD float grad_y_at(const Img& img, int x, int y)
{
    if (y - 1 <= -1 || y + 1 >= img.h) return kInvalid;
    const float a = img.at(x, y - 1), b = img.at(x, y + 1);
    if (!is_valid(a) || !is_valid(b)) return kInvalid;
    return b - a;
}

// getSubpixelFromDense(grad_x / grad_y, p) with the gradient maps derived on the fly.  p is already known to
// be inside [0,w) x [0,h) (optimize.cpp:52-56), so (x0,y0) is in range.
template <class Img>
DVO_HD void grad_subpixel(const Img& img, float px, float py, float& gx, float& gy)
{
    const int x0 = (int)px, y0 = (int)py;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float hx = px - (float)x0, vy = py - (float)y0;
    const bool inx = x1 < img.w, iny = y1 < img.h;
    const float gx00 = grad_x_at(img, x0, y0), gy00 = grad_y_at(img, x0, y0);
    const float gx10 = inx ? grad_x_at(img, x1, y0) : gx00, gy10 = inx ? grad_y_at(img, x1, y0) : gy00;
    const float gx01 = iny ? grad_x_at(img, x0, y1) : gx00, gy01 = iny ? grad_y_at(img, x0, y1) : gy00;
    const float gx11 = (inx && iny) ? grad_x_at(img, x1, y1) : gx00, gy11 = (inx && iny) ? grad_y_at(img, x1, y1) : gy00;
    gx = blend4(gx00, gx10, gx01, gx11, hx, vy);
    gy = blend4(gy00, gy10, gy01, gy11, hx, vy);
}

// ---------------------------------------------------------------- one pixel of Track::optimize (optimize.cpp:28-90)
struct GnParams {
    float step;        // optimize.cpp:22-26
    float sigma_min, sigma_max;  // optimize.cpp:83
    float min_depth;   // optimize.cpp:39
    int   crop;        // 1 when this level is cropped (optimize.cpp:33-36)
};

// weight of reliability, optimize.cpp:83-84: step / clamp(sigma, 0.01, 0.5)
DVO_HD float gn_weight(float step, float sigma_min, float sigma_max, float sigma)
{
    const float sc = sigma < sigma_min ? sigma_min : (sigma_max < sigma ? sigma_max : sigma);
    return step / sc;
}

// Jacobian row, residual and weighted residual of one contributing pixel (optimize.cpp:67-89), given the
// per-pixel constants iz = 1.0f / depth (the six divisions by z of optimize.cpp:70-74 share this reciprocal, D8)
// and wgt = gn_weight(...).
DVO_HD void gn_jacobian_pre(const Intr& k, int x, int y, float d, float iz, float wgt, float gx, float gy, float I1,
                            float I2, float J[6], float& r, float& rw)
{
    float X, Y, Z;
    back_project(k, (float)x, (float)y, d, X, Y, Z);
    const float fgx = k.fx * gx, fgy = k.fy * gy;
    const float xz = X * iz, yz = Y * iz;
    J[0] = fgx * iz;
    J[1] = fgy * iz;
    J[2] = ((-fmaf(fgy, Y, fgx * X)) * iz) * iz;
    J[3] = -(((fgx * xz) * yz) + (fgy * fmaf(yz, yz, 1.0f)));
    J[4] = (fgx * fmaf(xz, xz, 1.0f)) + ((fgy * xz) * yz);
    J[5] = fmaf(fgy, xz, -(fgx * yz));
    r = I2 - I1;
    rw = r * wgt;
}

DVO_HD void gn_jacobian(const Intr& k, const GnParams& prm, int x, int y, float d, float gx, float gy, float I1, float I2,
                        float sigma, float J[6], float& r, float& rw)
{
    gn_jacobian_pre(k, x, y, d, 1.0f / d, gn_weight(prm.step, prm.sigma_min, prm.sigma_max, sigma), gx, gy, I1, I2, J, r, rw);
}

// The gates of optimize.cpp:33-48 that do not need the warp.  (double)d < 0.20  <=>  d < 0.2f for float d.
DVO_HD bool gn_gate(const GnParams& prm, int x, int y, float d, float I1)
{
    if (prm.crop && (x < 20 || x > 140 || y < 20 || y > 100)) return false;
    if (d < prm.min_depth) return false;
    return !is_invalid(I1);
}

// Generic sampling part (any position, any validity pattern): warped gray I2 and the gradient at (u, v).
template <class Img>
DVO_HD bool gn_sample(const Img& ref_gray, float d, float u, float v, float& I2, float& gx, float& gy)
{
    // warped_gray(x) = getSubpixel(ref_gray, warp(-xi, x, d)) (transform.cpp:35-51)
    I2 = is_epsilon(d) ? kInvalid : get_subpixel(ref_gray, u, v);
    if (is_invalid(I2)) return false;
    if (u < 0.0f || v < 0.0f || (float)ref_gray.w <= u || (float)ref_gray.h <= v) return false;
    if (!coord_ok(u) || !coord_ok(v)) return false;  // D4 (NaN passes the comparisons above)
    grad_subpixel(ref_gray, u, v, gx, gy);
    return !(is_invalid(gx) || is_invalid(gy));
}

// Returns true when the pixel contributes; J[6], r (residual) and rw (weighted residual) are then set.
template <class Img>
DVO_HD bool gn_pixel(const Img& ref_gray, const Intr& k, const Pose& pose, const GnParams& prm, int x, int y,
                     float d, float I1, float sigma, float J[6], float& r, float& rw)
{
    if (!gn_gate(prm, x, y, d, I1)) return false;
    float u, v, I2, gx, gy;
    warp(pose, k, (float)x, (float)y, d, u, v);
    if (!gn_sample(ref_gray, d, u, v, I2, gx, gy)) return false;
    gn_jacobian(k, prm, x, y, d, gx, gy, I1, I2, sigma, J, r, rw);
    return true;
}

// ---------------------------------------------------------------- SE(3) in double, src/math/se3.cpp
DVO_HD void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// sin / cos / atan2 in double for the SE(3) chain ON THE DEVICE.  One Gauss-Newton step ends in a serial chain on one lane
// (6x6 solve, exp, log, exp: solve_finish), and the math library's sin + cos pair costs ~150 instructions of it three times over.
// The classical kernels (odd / even minimax polynomials on [-pi/4, pi/4], two-constant Cody-Waite reduction done with FMAs, arctangent by
// four break points) give the same accuracy -- below 1 ulp, compared with the library's results over the whole domain by
// dvo_selftest_trig -- in ~45.  Host code keeps libm (as before, host and device differ in the last bit of such calls; every pose is
// rounded to float before anything else sees it).  Arguments outside the fast domain take the library functions.
#if defined(__HIPCC__)
__device__ __forceinline__ double ksin_d(double x)   // |x| <= pi/4
{
    const double z = x * x, v = z * x;
    const double r = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    return fma(v, fma(z, r, -1.66666666666666324348e-01), x);
}
__device__ __forceinline__ double kcos_d(double x)   // |x| <= pi/4
{
    const double z = x * x;
    const double r = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                          2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * r);
}
__device__ __forceinline__ void sincos_dev(double x, double& s, double& c)
{
    const double ax = fabs(x);
    if (ax <= 0.78539816339744830962) { s = ksin_d(x); c = kcos_d(x); return; }
    if (ax < 1.0e5) {
        const double k = rint(x * 0.63661977236758134308);       // x = r + k pi/2, |r| <= pi/4 (+ rounding)
        double r = fma(-k, 1.57079632679489655800e+00, x);        // exact: both are multiples of 2^-52 and the difference is below 1
        r = fma(-k, 6.12323399573676603587e-17, r);
        if (fabs(r) >= 1.0e-5) {                                  // (closer to a multiple of pi/2 than that: leave it to the library's reduction)
            const int n = (int)k & 3;
            const double ss = ksin_d(r), cc = kcos_d(r);
            s = (n & 1) ? cc : ss;
            c = (n & 1) ? ss : cc;
            if (n & 2) s = -s;
            if ((n + 1) & 2) c = -c;
            return;
        }
    }
    s = sin(x); c = cos(x);
}
__device__ __forceinline__ double katan_d(double x)   // x >= 0 (inf allowed)
{
    double hi = 0.0, lo = 0.0;
    bool reduced = true;
    if (x < 0.4375) reduced = false;
    else if (x < 0.6875) { hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; x = (2.0 * x - 1.0) / (2.0 + x); }
    else if (x < 1.1875) { hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; x = (x - 1.0) / (x + 1.0); }
    else if (x < 2.4375) { hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; x = (x - 1.5) / (1.0 + 1.5 * x); }
    else { hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; x = -1.0 / x; }
    const double z = x * x, w = z * z;
    const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02), 6.66107313738753120669e-02),
                                            9.09088713343650656196e-02), 1.42857142725034663711e-01), 3.33333333333329318027e-01);
    const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02), -7.69187620504482999495e-02),
                                     -1.11111104054623557880e-01), -1.99999999998764832476e-01);
    if (!reduced) return x - x * (s1 + s2);
    return hi - ((x * (s1 + s2) - lo) - x);
}
__device__ __forceinline__ double atan2_dev(double y, double x)
{
    if (!(y > 0.0) || !(fabs(x) <= 1.0e300) || !(y <= 1.0e300) || x == 0.0) return atan2(y, x);   // zeros, negative y, NaN, infinities: the library
    const double z = katan_d(y / fabs(x));
    return x > 0.0 ? z : 3.14159265358979311600e+00 - (z - 1.22464679914735317720e-16);
}
#endif

DVO_HD void se3_exp_d(const double xi[6], double R[9], double t[3])
{  // se3.cpp:70-98 with so3::exp = cv::Rodrigues
    const double v[3] = {xi[0], xi[1], xi[2]}, w[3] = {xi[3], xi[4], xi[5]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th = sqrt(th2);
    double c = 1.0, s = 0.0;
    if (th < 2.220446049250313e-16) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        sincos_dev(th, s, c);
#else
        c = cos(th);
        s = sin(th);
#endif
        const double ith = 1.0 / th;   // one division for the axis (and A, B below): the chain is serial, a double division ~14 dependent instructions
        const double c1 = 1.0 - c, rx = w[0] * ith, ry = w[1] * ith, rz = w[2] * ith;
        R[0] = c + c1 * rx * rx;      R[1] = c1 * rx * ry - s * rz; R[2] = c1 * rx * rz + s * ry;
        R[3] = c1 * rx * ry + s * rz; R[4] = c + c1 * ry * ry;      R[5] = c1 * ry * rz - s * rx;
        R[6] = c1 * rx * rz - s * ry; R[7] = c1 * ry * rz + s * rx; R[8] = c + c1 * rz * rz;
    }
    if ((float)th > 1e-6f) {
        const double ith = 1.0 / th, ith2 = ith * ith;
        const double A = (1.0 - c) * ith2, B = (th - s) * (ith2 * ith);
        double wv[3], wwv[3];
        cross3(w, v, wv);
        cross3(w, wv, wwv);
        for (int i = 0; i < 3; i++) t[i] = v[i] + A * wv[i] + B * wwv[i];
    } else {
        t[0] = v[0]; t[1] = v[1]; t[2] = v[2];
    }
}

DVO_HD void se3_log_d(const double R[9], const double t[3], double xi[6])
{  // se3.cpp:31-43, 101-124; theta via atan2 (same angle as acos((tr-1)/2), no NaN by rounding)
    const double a[3] = {0.5 * (R[7] - R[5]), 0.5 * (R[2] - R[6]), 0.5 * (R[3] - R[1])};
    const double s = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const double cth = 0.5 * (R[0] + R[4] + R[8] - 1.0);
#if defined(__HIP_DEVICE_COMPILE__)
    const double th = atan2_dev(s, cth);
#else
    const double th = atan2(s, cth);
#endif
    double w[3] = {0, 0, 0};
    if ((float)th > 1e-6f && s > 0.0) {
        const double k = th / s;
        w[0] = a[0] * k; w[1] = a[1] * k; w[2] = a[2] * k;
    }
    const double wl2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double wl = sqrt(wl2);
    double v[3] = {t[0], t[1], t[2]};
    if ((float)wl > 1e-6f) {
        const double half = 0.5 * wl;
#if defined(__HIP_DEVICE_COMPILE__)
        double sh, ch;
        sincos_dev(half, sh, ch);
#else
        const double sh = sin(half), ch = cos(half);
#endif
        const double coef = (1.0 - (wl * ch) / (2.0 * sh)) / wl2;
        double wt[3], wwt[3];
        cross3(w, t, wt);
        cross3(w, wt, wwt);
        for (int i = 0; i < 3; i++) v[i] = t[i] - 0.5 * wt[i] + coef * wwt[i];
    }
    xi[0] = v[0]; xi[1] = v[1]; xi[2] = v[2]; xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

DVO_HD void se3_concatenate_f(const float a[6], const float b[6], float out[6])
{  // se3.cpp:127-131: log(exp(a) exp(b))
    double xa[6], xb[6], Ra[9], ta[3], Rb[9], tb[3], R[9], t[3], x[6];
    for (int i = 0; i < 6; i++) { xa[i] = a[i]; xb[i] = b[i]; }
    se3_exp_d(xa, Ra, ta);
    se3_exp_d(xb, Rb, tb);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) R[3 * r + c] = Ra[3 * r] * Rb[c] + Ra[3 * r + 1] * Rb[3 + c] + Ra[3 * r + 2] * Rb[6 + c];
        t[r] = Ra[3 * r] * tb[0] + Ra[3 * r + 1] * tb[1] + Ra[3 * r + 2] * tb[2] + ta[r];
    }
    se3_log_d(R, t, x);
    for (int i = 0; i < 6; i++) out[i] = (float)x[i];
}

// exp(+xi) AND exp(-xi) from one sqrt/sincos.  Bit-identical to se3_exp_d(xi) and se3_exp_d(-xi): R(-w) = R(w)^T entry
// for entry (products of two negated factors are the same floats), w' x v' = w x v, w' x (w' x v') = -(w x (w x v)),
// and the same A, B, so t(-xi)[i] = -v[i] + A wv[i] + B (-wwv[i]) in the same operation order.
DVO_HD void se3_exp_pair_d(const double xi[6], double Rp[9], double tp[3], double Rm[9], double tm[3])
{
    const double v[3] = {xi[0], xi[1], xi[2]}, w[3] = {xi[3], xi[4], xi[5]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th = sqrt(th2);
    double c = 1.0, s = 0.0;
    if (th < 2.220446049250313e-16) {
        Rp[0] = 1; Rp[1] = 0; Rp[2] = 0; Rp[3] = 0; Rp[4] = 1; Rp[5] = 0; Rp[6] = 0; Rp[7] = 0; Rp[8] = 1;
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        sincos_dev(th, s, c);
#else
        c = cos(th);
        s = sin(th);
#endif
        const double ith = 1.0 / th;
        const double c1 = 1.0 - c, rx = w[0] * ith, ry = w[1] * ith, rz = w[2] * ith;
        Rp[0] = c + c1 * rx * rx;      Rp[1] = c1 * rx * ry - s * rz; Rp[2] = c1 * rx * rz + s * ry;
        Rp[3] = c1 * rx * ry + s * rz; Rp[4] = c + c1 * ry * ry;      Rp[5] = c1 * ry * rz - s * rx;
        Rp[6] = c1 * rx * rz - s * ry; Rp[7] = c1 * ry * rz + s * rx; Rp[8] = c + c1 * rz * rz;
    }
    Rm[0] = Rp[0]; Rm[1] = Rp[3]; Rm[2] = Rp[6];
    Rm[3] = Rp[1]; Rm[4] = Rp[4]; Rm[5] = Rp[7];
    Rm[6] = Rp[2]; Rm[7] = Rp[5]; Rm[8] = Rp[8];
    if ((float)th > 1e-6f) {
        const double ith = 1.0 / th, ith2 = ith * ith;
        const double A = (1.0 - c) * ith2, B = (th - s) * (ith2 * ith);
        double wv[3], wwv[3];
        cross3(w, v, wv);
        cross3(w, wv, wwv);
        for (int i = 0; i < 3; i++) {
            tp[i] = v[i] + A * wv[i] + B * wwv[i];
            tm[i] = -v[i] + A * wv[i] + B * (-wwv[i]);
        }
    } else {
        for (int i = 0; i < 3; i++) { tp[i] = v[i]; tm[i] = -v[i]; }
    }
}

// One pose update of Tracker::track (tracker.cpp:46-52) with exp(xi) carried from the previous iteration:
//   xi' = float(log(exp(xi) exp(upd)));  if no NaN: Tc <- exp(xi'), pose <- float(exp(-xi')).  Returns false on NaN.
// Tc = (R, t) of exp(xi) for the CURRENT float xi, exactly what se3_exp_d(xi) returns (so the result equals
// se3_concatenate_f(xi, upd) followed by pose_from_xi(xi', -1) bit for bit).
DVO_HD bool se3_update_pose(double Tc[12], const float upd[6], float xi[6], Pose& pose)
{
    double xu[6], Ru[9], tu[3], R[9], t[3], x[6];
    for (int i = 0; i < 6; i++) xu[i] = upd[i];
    se3_exp_d(xu, Ru, tu);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) R[3 * r + c] = Tc[3 * r] * Ru[c] + Tc[3 * r + 1] * Ru[3 + c] + Tc[3 * r + 2] * Ru[6 + c];
        t[r] = Tc[3 * r] * tu[0] + Tc[3 * r + 1] * tu[1] + Tc[3 * r + 2] * tu[2] + Tc[9 + r];
    }
    se3_log_d(R, t, x);
    float nxt[6];
    bool ok = true;
    for (int i = 0; i < 6; i++) {
        nxt[i] = (float)x[i];
        ok = ok && !(nxt[i] != nxt[i]);  // testXi, util.hpp:34-44
    }
    if (!ok) return false;
    double xn[6], Rp[9], tp[3], Rm[9], tm[3];
    for (int i = 0; i < 6; i++) { xi[i] = nxt[i]; xn[i] = nxt[i]; }
    se3_exp_pair_d(xn, Rp, tp, Rm, tm);
    for (int i = 0; i < 9; i++) { Tc[i] = Rp[i]; pose.R[i] = (float)Rm[i]; }
    for (int i = 0; i < 3; i++) { Tc[9 + i] = tp[i]; pose.t[i] = (float)tm[i]; }
    return true;
}

DVO_HD void pose_from_xi(const float xi[6], float sign, Pose& p)
{
    double x[6], R[9], t[3];
    for (int i = 0; i < 6; i++) x[i] = (double)sign * (double)xi[i];
    se3_exp_d(x, R, t);
    for (int i = 0; i < 9; i++) p.R[i] = (float)R[i];
    for (int i = 0; i < 3; i++) p.t[i] = (float)t[i];
}

DVO_HD void se3_exp_f(const float xi[6], float T[16])
{
    Pose p;
    pose_from_xi(xi, 1.0f, p);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[4 * r + c] = p.R[3 * r + c];
        T[4 * r + 3] = p.t[r];
    }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

DVO_HD void se3_log_f(const float T[16], float xi[6])
{
    double R[9], t[3], x[6];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) R[3 * r + c] = T[4 * r + c];
        t[r] = T[4 * r + 3];
    }
    se3_log_d(R, t, x);
    for (int i = 0; i < 6; i++) xi[i] = (float)x[i];
}

// ---------------------------------------------------------------- 6x6 solve: x = H^+ g  (replaces cv::solve(A,-B,SVD), optimize.cpp:96-98)
DVO_HD void jacobi_eig6(double A[36], double V[36])
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
                const double tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + tt * tt), s = tt * c;
                for (int k = 0; k < 6; k++) {
                    const double akp = A[6 * k + p], akq = A[6 * k + q];
                    A[6 * k + p] = c * akp - s * akq;
                    A[6 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; k++) {
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

// H: 21 upper-triangle entries row major.  Fast path LDL^T; a pivot <= 1e-12 * max diag switches to the
// eigen pseudo-inverse with cv::solve(DECOMP_SVD)'s cut (sqrt(lambda) <= 2 FLT_EPSILON sum sqrt(lambda) dropped).
#if defined(__HIPCC__)
#define DVO_HD_NOINLINE __host__ __device__ __attribute__((noinline)) inline
#else
#define DVO_HD_NOINLINE inline
#endif

// index of (i, j), i <= j, in the 21-entry row-major upper triangle
DVO_HD constexpr int tri(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

DVO_HD_NOINLINE void solve6_pinv(const double H[21], const double g[6], float x[6]);

// Fast path: LDL^T with every loop fully unrolled (compile-time indices only, so the factors live in registers --
// the dynamically indexed version ran out of scratch memory and dominated k_gn_solve).  Same operation order as
// before.  Returns through solve6_pinv when a pivot is <= 1e-12 * max diag.
DVO_HD void solve6(const double H[21], const double g[6], float x[6])
{
    double maxd = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) maxd = H[tri(i, i)] > maxd ? H[tri(i, i)] : maxd;
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = 0.0f;
    if (!(maxd > 0.0)) return;
    double L[6][6], d[6], id[6], y[6], z[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double dj = H[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) dj -= L[j][k] * L[j][k] * d[k];
        ok = ok && (dj > 1e-12 * maxd);
        d[j] = dj;
        const double inv = 1.0 / dj;   // ONE division per pivot (6 instead of 21): this runs on one lane, in series, after every iteration
        id[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double v = H[tri(j, i)];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * d[k];
            L[i][j] = v * inv;
        }
    }
    if (ok) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double v = g[i];
#pragma unroll
            for (int k = 0; k < i; k++) v -= L[i][k] * y[k];
            y[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) y[i] *= id[i];
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            double v = y[i];
#pragma unroll
            for (int k = i + 1; k < 6; k++) v -= L[k][i] * z[k];
            z[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = (float)z[i];
        return;
    }
    solve6_pinv(H, g, x);
}

// Rank-deficient systems: eigen pseudo-inverse (rare; kept out of line so its arrays do not burden the fast path)
DVO_HD_NOINLINE void solve6_pinv(const double H[21], const double g[6], float x[6])
{
    double A[36], z[6];
    {
        int k = 0;
        for (int i = 0; i < 6; i++)
            for (int j = i; j < 6; j++) {
                A[6 * i + j] = H[k];
                A[6 * j + i] = H[k];
                k++;
            }
    }
    double V[36], sv[6], sum = 0;
    jacobi_eig6(A, V);
    for (int i = 0; i < 6; i++) {
        sv[i] = sqrt(A[7 * i] > 0.0 ? A[7 * i] : 0.0);
        sum += sv[i];
    }
    const double thr = 2.0 * 1.1920928955078125e-07 * sum;
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

// ---------------------------------------------------------------- Gaussian fusion, src/math/gaussian.cpp
DVO_HD uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// D3: replaces dist(engine) of gaussian.cpp:8-9,22: uniform_real_distribution<float>(2.0, 0.5) then min(.,4)
DVO_HD float rng_depth(uint32_t seed, uint32_t frame_id, uint32_t pixel)
{
    const uint32_t h = mix32(mix32(seed ^ (frame_id * 0x9E3779B9U)) ^ (pixel * 0x85EBCA6BU) ^ 0x68E31DA4U);
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    const float v = fmaf(u, -1.5f, 2.0f);
    return v < 4.0f ? v : 4.0f;
}

DVO_HD float gauss_gain(float d, float diff)
{  // gaussian.cpp:20: (m < 0.8) ? 0.5 + (m / 0.8f) * 0.5 : 1.0   [(double)m < 0.8  <=>  m < 0.8f (0.8f > 0.8)]
    const float m = d < diff ? d : diff;
    // m / 0.8f without the ~10-instruction IEEE division: q0 = m * RN(1 / 0.8f), one residual step.  For every float with
    // |m| < 1e30 the resulting gain equals the reference's bit for bit (all 2.96e9 such values with m < 0.8f walked by
    // tools/verify/gauss_gain_div.c; the quotient itself is the correctly rounded one for 1e-30 < |m| < 1e30, and below that both
    // gains are exactly 0.5f); the literal division remains for |m| >= 1e30 and infinities.
    float q;
    if (__builtin_expect(fabsf(m) < 1e30f, 1)) {
        const float q0 = m * 1.25f;
        q = fmaf(fmaf(-q0, 0.8f, m), 1.25f, q0);
    } else {
        q = m / 0.8f;
    }
    return (m < 0.8f) ? 0.5f + q * 0.5f : 1.0f;
}

DVO_HD bool gaussian_fuse(float& depth, float& sigma, float d, float s)
{  // gaussian.cpp:33-50
    const float v1 = sigma * sigma, v2 = s * s, v = v1 + v2;
    const float diff = fabsf(d - depth);
    const float gain = gauss_gain(d, diff);
    const float ms = sigma < s ? s : sigma;
    if (diff > gain * ms) return false;
    depth = fmaf(v1, d, v2 * depth) / v;
    sigma = sqrtf((v1 * v2) / v);
    return true;
}

#if defined(__HIPCC__)
// The same fusion for operands in a KNOWN range -- every depth and sigma a positive normal float in [2^-20, 2^20], which the caller
// has checked (regularize_fuse4) -- without the ~11-instruction IEEE division and ~12-instruction IEEE square root sequences:
//   a / v   = q + (a - v q) y  with  y = recip_fast(v) = RN(1 / v), q = a y            (6 instructions for the first quotient, 3 for
//             the second: they share y);  bit-identical to the IEEE quotient for ALL 2^23 x 2^23 mantissa pairs, enumerated on the
//             device (tools/verify/div_sqrt_exhaustive.hip, "division variant 2"; dvo_selftest_division re-checks a slice) -- and
//             multiplies / FMAs are scale invariant while nothing leaves the normal range: here v in [2^-40, 2^41], numerators in
//             [2^-80, 2^82], quotients in [2^-42, 2^21], residuals above 2^-110;
//   sqrt(x) = s + (x - s s) (y / 2)  with  y = v_rsq_f32(x), s = x y                   (5 instructions);  bit-identical to sqrtf for
//             EVERY float in [2^-100, 2^100] (same tool, "sqrt variant 1"; dvo_selftest_sqrt).
// The fused sigma lies in [min / sqrt 2, min] and the fused depth between its inputs, so four fusions in a row stay in range.
__device__ __forceinline__ float sqrt_fast(float x)   // == sqrtf(x) for 2^-100 <= x <= 2^100
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y;
    return fmaf(fmaf(-s, s, x), 0.5f * y, s);
}
__device__ __forceinline__ float div_by_recip(float a, float v, float y)   // == a / v for y = recip_fast(v), operands as above
{
    const float q = a * y;
    return fmaf(fmaf(-v, q, a), y, q);
}
__device__ __forceinline__ bool gaussian_fuse_ranged(float& depth, float& sigma, float d, float s)
{  // gaussian.cpp:33-50, the operations of gaussian_fuse() in the same order
    const float v1 = sigma * sigma, v2 = s * s, v = v1 + v2;
    const float diff = fabsf(d - depth);
    const float gain = gauss_gain(d, diff);
    const float ms = sigma < s ? s : sigma;
    if (diff > gain * ms) return false;
    const float y = recip_fast(v);
    depth = div_by_recip(fmaf(v1, d, v2 * depth), v, y);
    sigma = sqrt_fast(div_by_recip(v1 * v2, v, y));
    return true;
}
#define DVO_FUSE_RANGE_LO 0x35800000u   /* 2^-20 */
#define DVO_FUSE_RANGE_HI 0x49800000u   /* 2^20  */
#endif

DVO_HD bool gaussian_update(float& depth, float& sigma, float d, float s, float reset_depth)
{  // gaussian.cpp:12-31
    if (gaussian_fuse(depth, sigma, d, s)) return true;
    depth = reset_depth;
    sigma = 0.5f;
    return false;
}

DVO_HD bool round_coord(float v, int& out)
{  // cv::Point2f -> cv::Point2i = cvRound (round half to even), with D4
    if (!coord_ok(v)) return false;
    out = (int)rintf(v);
    return true;
}

}  // namespace dvo
