/*
 * dvo_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A dependency-free, scalar, sequential (raster order) restatement in plain C
 * of the semi-dense direct-VO hot path of KYabuuchi/direct-visual-odometry.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libdvo.so) never links, loads or calls it.
 *
 * PARITY STATUS: **parity unpinned**.  The reference has no assertions, golden
 * vectors or known-answer tests for this path (all of its test/ programs are
 * interactive viewers), it cannot be compiled here (OpenCV/Eigen/GLFW absent)
 * and there is no Python to import.  This restatement follows the reference
 * sources line by line (each function cites file:line under /root/reference)
 * and is cross-checked by analytic properties in tests/ (identity warp,
 * finite-difference Jacobian, LSQ vs normal equations, exp/log round trip).
 *
 * Deliberate, documented deviations from the literal reference:
 *  D1 wall-clock stop (tracker.cpp:18,68-73) disabled -> iteration counts are
 *     machine independent.
 *  D2 SE(3) exp/log/concatenate evaluated in double and rounded to float once
 *     (the reference mixes float MatExpr algebra with cv::Rodrigues in double);
 *     a float-literal restatement is kept as orc_se3_*_f32lit for comparison.
 *  D3 global std::mt19937 in Gaussian::update (gaussian.cpp:8-9,22) replaced
 *     by a counter-based hash keyed on (seed, frame id, pixel index); same
 *     distribution min(2.0 - 1.5u, 4).
 *  D4 (int)/cvRound of non-finite or |v| >= 2^30 coordinates (UB in the
 *     reference) is defined as "out of range".
 *  D5 sigmaEstimate's gradient lookup index is clamped into the image (the
 *     reference can read one past the edge, implement.cpp:196-200,230).
 *  D6 initial mono depth (cv::randn, frame.hpp:17-21) is an explicit input.
 *  D7 forEach bodies are executed sequentially in raster order (the reference
 *     races on shared state, optimize.cpp:8,64,80; implement.cpp:250-252).
 *  D8 multi-term float sums are evaluated as explicit fmaf chains, and divisions
 *     that share a divisor as one correctly rounded reciprocal times the
 *     numerators (1/fx, 1/fy, 1/Z' in project, 1/z in optimize.cpp:70-74): the
 *     reference is built -Ofast -march=native, i.e. -ffp-contract=fast and
 *     -freciprocal-math.  The order is fixed in DESIGN.md §3 so CPU and GPU
 *     agree bit for bit per pixel.
 *  D10 depthEstimate (implement.cpp:49-71) is evaluated in double from float inputs.
 *  D9 VisualOdometry(gray,depth,sigma,K) builds its keyframe as Frame(...,4,1)
 *     (system.hpp:30) while odometrize() builds Frame(...,3,2) (system.hpp:47), so
 *     the reference throws at m_scenes.at(3) on the first tracked frame; here that
 *     ctor uses the mono geometry (3,2) so the combination works.
 */
#ifndef DVO_ORACLE_H
#define DVO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INVALID (-2.0f)
#define ORC_EPSILON (1e-6f)
#define ORC_MAX_LEVELS 8
#define ORC_MAX_ITER 15

/* Threads used by the forEach bodies (gradiate, warpImage, the optimize lambda): 1 (default) = sequential raster order, the
 * oracle proper; n > 1 = row-parallel as cv::Mat::forEach, for bench.py's all-core CPU baseline. */
void orc_set_threads(int n);
int  orc_get_threads(void);

/* Arithmetic mode -- instrumentation for the D8 / D2 sensitivity measurements only (tests/test_oracle_literal.py, bench.py's
 * accuracy.oracle_self_sensitivity).  0 = canonical (what every parity test compares the GPU against). */
#define ORC_LIT_ARITH 1  /* per-pixel expressions exactly as the source writes them: true divisions, no fmaf, no shared reciprocal */
#define ORC_LIT_SE3   2  /* per-pixel pose and pose composition from the float-literal se3.cpp restatement (orc_se3_*_f32lit) */
void orc_set_literal(int mask);
int  orc_get_literal(void);
/* step[3] = {default, level 1, level 2} (optimize.cpp:22-26), stop thresholds (tracker.cpp:16-17); NULL / negative = the reference's */
void orc_set_tracker_params(const float step3[3], float min_residual, float min_update);
void orc_set_nudge_ulps(int n);   /* orc_track: first xi_update component of level 0 / iteration 0 moved by n ulps (0 = off) */

/* ---- math/se3.cpp ---------------------------------------------------- */
void orc_se3_exp(const float xi[6], float T[16]);               /* se3.cpp:70-98  (double inside, D2) */
void orc_se3_log(const float T[16], float xi[6]);               /* se3.cpp:101-124 */
void orc_se3_concatenate(const float a[6], const float b[6], float out[6]); /* se3.cpp:127-131 */
void orc_se3_exp_f32lit(const float xi[6], float T[16]);        /* float-literal restatement */
void orc_se3_log_f32lit(const float T[16], float xi[6]);
void orc_se3_concatenate_f32lit(const float a[6], const float b[6], float out[6]);
/* R (9, row major) and t (3) of exp(sign*xi) rounded to float: the 12 numbers every per-pixel warp uses */
void orc_pose_from_xi(const float xi[6], float sign, float Rt[12]);

/* ---- core/convert.cpp ------------------------------------------------- */
void  orc_cull_image(const float* src, int w, int h, int times, float* dst);  /* convert.cpp:7-20; dst is (w>>times)x(h>>times) */
void  orc_cull_intrinsic(const float K[9], int times, float out[9]);          /* convert.cpp:22-29 */
void  orc_gradiate(const float* img, int w, int h, int xdir, float* out);     /* convert.cpp:41-75 */
float orc_get_pixel(const float* img, int w, int h, int x, int y);            /* convert.cpp:107-125 */
float orc_get_subpixel(const float* img, int w, int h, float px, float py);   /* convert.cpp:128-177 (fill quirk) */
float orc_get_subpixel_dense(const float* img, int w, int h, float px, float py); /* convert.cpp:77-105 */

/* ---- core/transform.cpp ----------------------------------------------- */
void orc_back_project(const float K[9], float px, float py, float d, float X[3]); /* transform.cpp:25-28 */
void orc_project(const float K[9], const float X[3], float p[2]);                 /* transform.cpp:20-23 */
void orc_transform(const float Rt[12], const float X[3], float Y[3]);             /* transform.cpp:7-18 */
void orc_warp(const float Rt[12], float px, float py, float d, const float K[9], float p[2]); /* transform.cpp:30-33 */
void orc_warp_image(const float xi[6], const float* gray, const float* depth,
                    int w, int h, const float K[9], float* out);                  /* transform.cpp:35-51 */

/* ---- track/optimize.cpp ------------------------------------------------ */
typedef struct {
    double H[21];       /* upper triangle of sum J^T J, row major (00,01,..05,11,..55) */
    double g[6];        /* sum J^T (w r)                                            */
    double sum_r2;      /* sum r^2 (unweighted)                                     */
    int    n_valid;
    float  xi_update[6];/* optimize.cpp:96-98: (A^T A)^+ A^T B                      */
    float  residual;    /* sum_r2 / n_valid, or -1 when n_valid == 0                */
} orc_outcome;

/* One Gauss-Newton step = Track::optimize (optimize.cpp:10-99) on one level.
 * `level` is the level INDEX (Stuff::levels): selects step size and the level-2 crop.
 * crop_enable=1 reproduces optimize.cpp:33-36; 0 = "roofline preset".
 * mask (optional, w*h bytes) receives 1 for every pixel that contributed.
 * variant: 0 = hoisted (pose once per call, fused warp, 6x6 normal equations in double),
 *          1 = faithful (per-pixel se3 exp, materialised warpImage, Nx6 stack + SVD least squares). */
void orc_optimize(const float* obj_gray, const float* ref_gray, const float* ref_depth,
                  const float* ref_sigma, int w, int h, const float K[9], const float xi[6],
                  int level, int crop_enable, int variant, orc_outcome* out, uint8_t* mask);

/* dense least squares used by the faithful variant: x = argmin |A x + B| (min norm), returns -x. */
void orc_lsq_svd(const float* A, const float* B, int n, float x_update[6]);
/* 6x6 pseudo-inverse solve used by the hoisted variant */
void orc_solve6(const double H[21], const double g[6], float x[6]);

/* ---- system/frame.cpp: pyramid ------------------------------------------ */
typedef struct {
    int levels, culls;
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    float K[ORC_MAX_LEVELS][9];
    float* gray[ORC_MAX_LEVELS];
    float* depth[ORC_MAX_LEVELS];
    float* sigma[ORC_MAX_LEVELS];
    float* age;                 /* top level size */
    int   id;
    int   ref_index;            /* index into history of m_ref_frame (-1 none) */
    float xi[6], rel_xi[6];
} orc_frame;

orc_frame* orc_frame_create(const float* gray, const float* depth, const float* sigma,
                            int w, int h, const float K[9], int levels, int culls, int id); /* frame.hpp:91-117, frame.cpp:30-37 */
void orc_frame_destroy(orc_frame* f);
void orc_frame_update_depth_sigma_age(orc_frame* f, const float* d, const float* s, const float* a); /* frame.cpp:47-54 */
void orc_frame_update_depth_sigma(orc_frame* f, const float* d, const float* s);                     /* frame.cpp:39-45 */
void orc_frame_update_depth(orc_frame* f, const float* d);                                           /* frame.cpp:56-61 */

/* ---- track/tracker.cpp --------------------------------------------------- */
typedef struct {
    int   n_iter[ORC_MAX_LEVELS];
    float residual[ORC_MAX_LEVELS][ORC_MAX_ITER];
    float upd_norm[ORC_MAX_LEVELS][ORC_MAX_ITER];
    int   n_valid[ORC_MAX_LEVELS][ORC_MAX_ITER];
    float xi_after[ORC_MAX_LEVELS][ORC_MAX_ITER][6];
} orc_track_log;

/* Tracker::track (tracker.cpp:22-85) with the wall-clock stop disabled (D1).
 * fixed_iters > 0 runs exactly that many iterations per level with no early exit (roofline preset). */
void orc_track(const orc_frame* obj, const orc_frame* ref, int crop_enable, int variant,
               int fixed_iters, float xi_out[6], orc_track_log* log);

/* ---- math/gaussian.cpp ---------------------------------------------------- */
float orc_rng_depth(uint32_t seed, uint32_t frame_id, uint32_t pixel);                  /* D3 */
int   orc_gaussian_update(float* depth, float* sigma, float d, float s, float reset_depth); /* gaussian.cpp:12-31 */
int   orc_gaussian_fuse(float* depth, float* sigma, float d, float s);                  /* gaussian.cpp:33-50 */

/* ---- map/implement.cpp ----------------------------------------------------- */
void orc_propagate(const float* ref_depth, const float* ref_sigma, const float* ref_age,
                   int w, int h, const float xi[6], const float K[9],
                   float* depth, float* sigma, float* age);                             /* implement.cpp:217-256 */
void orc_regularize(const float* depth, const float* sigma, int w, int h, float* out);  /* implement.cpp:156-180 */
/* Implement::update (implement.cpp:182-214) for ONE pixel; returns new depth/sigma or (-1,-1) */
void orc_implement_update(const float* obj_gray, const float* born_gray, const float* born_gx,
                          const float* born_gy, int w, int h, const float r_xi[6], const float K[9],
                          int qx, int qy, float depth, float sigma, float* new_depth, float* new_sigma);

/* ---- map/mapper.cpp + system/system.hpp ------------------------------------ */
typedef struct orc_vo orc_vo;
orc_vo* orc_vo_create(const float K[9], int w, int h, uint32_t rng_seed, int crop_enable, int variant);
void    orc_vo_destroy(orc_vo* vo);
/* explicit initial depth/sigma for the first mono keyframe at the culled base resolution (D6) */
void    orc_vo_set_initial_depth(orc_vo* vo, const float* depth, const float* sigma);
/* VisualOdometry(gray,depth,sigma,K) ctor, system.hpp:24-32 */
void    orc_vo_init_keyframe(orc_vo* vo, const float* gray, const float* depth, const float* sigma);
/* odometrize, system.hpp:44-74.  Returns 1 if the frame became a keyframe. */
int     orc_vo_odometrize(orc_vo* vo, const float* gray, float T_world[16]);
/* odometrizeUsingDepth, system.hpp:77-93 */
void    orc_vo_odometrize_depth(orc_vo* vo, const float* gray, const float* depth, const float* sigma, float T_rel[16]);
int     orc_vo_keyframe_count(const orc_vo* vo);
const orc_frame* orc_vo_keyframe(const orc_vo* vo, int index_from_oldest);
const orc_frame* orc_vo_last_frame(const orc_vo* vo);
int     orc_vo_last_valid_updates(const orc_vo* vo);
/* Mapper pieces exposed for operator-level parity tests */
int     orc_need_new_frame(const float rel_xi[6], int id, int ref_id);                  /* mapper.cpp:45-60 */
/* Mapper::update (mapper.cpp:76-137): history = array of n_hist keyframes, oldest first; ref = history[n_hist-1] */
int     orc_mapper_update(orc_frame** history, int n_hist, const orc_frame* obj, uint32_t rng_seed);

#ifdef __cplusplus
}
#endif
#endif
