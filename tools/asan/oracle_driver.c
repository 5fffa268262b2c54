/* Host sanitizer driver for the CPU oracle: every entry point of oracle/dvo_oracle.h once, on a small synthetic
 * scene, under -fsanitize=address,undefined (make -C oracle asan).  Checks nothing numerically: the parity tests do that. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/dvo_oracle.h"

#define W 160
#define H 120

static float frand(uint32_t* s)
{
    *s = *s * 1664525u + 1013904223u;
    return (float)(*s >> 8) * (1.0f / 16777216.0f);
}

static void scene(float* gray, float* depth, float* sigma, float shift, uint32_t seed)
{
    uint32_t s = seed;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float fx = (float)x + shift, fy = (float)y;
            float g = 0.5f + 0.25f * sinf(0.11f * fx) * cosf(0.07f * fy) + 0.15f * sinf(0.031f * fx * fy * 0.05f) + 0.02f * frand(&s);
            float d = 1.5f + 0.3f * sinf(0.02f * fx) + 0.2f * cosf(0.03f * fy);
            if ((x * 7 + y * 13) % 97 == 0) { d = 0.0f; g = ORC_INVALID; }  /* sensor holes */
            gray[y * W + x] = g;
            depth[y * W + x] = d;
            sigma[y * W + x] = 0.3f;
        }
}

int main(void)
{
    const float K[9] = {130.0f, 0, 80.0f, 0, 130.0f, 60.0f, 0, 0, 1};
    float* g0 = malloc(sizeof(float) * W * H), *d0 = malloc(sizeof(float) * W * H), *s0 = malloc(sizeof(float) * W * H);
    float* g1 = malloc(sizeof(float) * W * H), *d1 = malloc(sizeof(float) * W * H), *s1 = malloc(sizeof(float) * W * H);
    float* tmp = malloc(sizeof(float) * W * H), *tmp2 = malloc(sizeof(float) * W * H), *tmp3 = malloc(sizeof(float) * W * H);
    scene(g0, d0, s0, 0.0f, 1u);
    scene(g1, d1, s1, 0.6f, 2u);

    /* SE(3) */
    const float xi[6] = {0.01f, -0.02f, 0.005f, 0.003f, -0.004f, 0.002f}, zero[6] = {0, 0, 0, 0, 0, 0};
    float T[16], back[6], cat[6], Rt[12];
    orc_se3_exp(xi, T); orc_se3_log(T, back); orc_se3_concatenate(xi, back, cat);
    orc_se3_exp_f32lit(xi, T); orc_se3_log_f32lit(T, back); orc_se3_concatenate_f32lit(xi, zero, cat);
    orc_se3_exp(zero, T); orc_se3_log(T, back);
    orc_pose_from_xi(xi, -1.0f, Rt);

    /* image operators */
    float Kc[9];
    orc_cull_image(g0, W, H, 1, tmp);
    orc_cull_image(g0, W, H, 3, tmp);
    orc_cull_intrinsic(K, 2, Kc);
    orc_gradiate(g0, W, H, 1, tmp);
    orc_gradiate(g0, W, H, 0, tmp2);
    float acc = 0;
    for (int i = -2; i < 6; i++) {
        acc += orc_get_pixel(g0, W, H, i * 40, i * 30);
        acc += orc_get_subpixel(g0, W, H, (float)i * 39.7f, (float)i * 29.9f);
        acc += orc_get_subpixel_dense(g0, W, H, (float)i * 39.7f + 0.5f, (float)i * 29.9f);
    }
    acc += orc_get_subpixel(g0, W, H, NAN, 1.0f) + orc_get_subpixel(g0, W, H, 1e30f, -1e30f);
    acc += orc_get_subpixel(g0, W, H, (float)W - 0.5f, (float)H - 0.5f);
    float X[3], Y[3], p[2];
    orc_back_project(K, 10.0f, 20.0f, 1.5f, X); orc_transform(Rt, X, Y); orc_project(K, Y, p); orc_warp(Rt, 10.0f, 20.0f, 1.5f, K, p);
    orc_warp_image(xi, g0, d0, W, H, K, tmp);

    /* one Gauss-Newton step, both variants, with and without crop / mask; all threads settings */
    orc_outcome out;
    uint8_t* mask = calloc(W * H, 1);
    for (int variant = 0; variant < 2; variant++)
        for (int crop = 0; crop < 2; crop++) {
            orc_optimize(g1, g0, d0, s0, W, H, K, xi, 2, crop, variant, &out, mask);
            orc_optimize(g1, g0, d0, s0, W, H, K, zero, 3, crop, variant, &out, NULL);
        }
    orc_set_threads(4);
    orc_optimize(g1, g0, d0, s0, W, H, K, xi, 2, 1, 1, &out, mask);
    orc_set_threads(1);
    if (orc_get_threads() != 1) return 2;
    /* empty system (everything gated out) and a rank-deficient one */
    for (int i = 0; i < W * H; i++) tmp3[i] = 0.0f;
    orc_optimize(g1, g0, tmp3, s0, W, H, K, xi, 2, 1, 0, &out, NULL);
    orc_optimize(g1, g0, tmp3, s0, W, H, K, xi, 2, 1, 1, &out, NULL);
    {
        float A[12 * 6], B[12], x[6];
        for (int i = 0; i < 12; i++) {
            for (int j = 0; j < 6; j++) A[i * 6 + j] = (j == 3) ? 0.0f : sinf((float)(i * 7 + j));
            B[i] = cosf((float)i);
        }
        orc_lsq_svd(A, B, 12, x);
        double Hm[21] = {0}, gv[6] = {1, 2, 3, 0, 5, 6};
        int k = 0;
        for (int i = 0; i < 6; i++)
            for (int j = i; j < 6; j++) Hm[k++] = (i == j && i != 3) ? 2.0 + i : 0.0;
        orc_solve6(Hm, gv, x);
    }

    /* pyramid + tracker */
    orc_frame* ref = orc_frame_create(g0, d0, s0, W, H, K, 3, 0, 0);
    orc_frame* obj = orc_frame_create(g1, d1, s1, W, H, K, 3, 0, 1);
    orc_track_log log;
    float xo[6];
    orc_track(obj, ref, 1, 0, 0, xo, &log);
    orc_track(obj, ref, 0, 1, 2, xo, &log);
    orc_frame_update_depth_sigma_age(obj, d0, s0, s0);
    orc_frame_update_depth_sigma(obj, d1, s1);
    orc_frame_update_depth(obj, d0);

    /* Gaussian + mapping operators */
    float dd = 1.5f, ss = 0.3f;
    acc += orc_rng_depth(3u, 4u, 5u);
    orc_gaussian_update(&dd, &ss, 1.6f, 0.2f, 2.0f);
    orc_gaussian_fuse(&dd, &ss, 9.0f, 0.1f);
    for (int i = 0; i < W * H; i++) tmp3[i] = (float)(i % 5);
    orc_propagate(d0, s0, tmp3, W, H, xi, K, tmp, tmp2, g1);
    scene(g1, d1, s1, 0.6f, 2u);
    orc_regularize(d0, s0, W, H, tmp);
    orc_gradiate(g0, W, H, 1, tmp);
    orc_gradiate(g0, W, H, 0, tmp2);
    for (int q = 0; q < 40; q++) {
        float nd, ns;
        orc_implement_update(g1, g0, tmp, tmp2, W, H, xi, K, 3 + q * 3, 2 + q * 2, 1.5f, 0.3f, &nd, &ns);
    }
    acc += (float)orc_need_new_frame(xi, 5, 1);
    orc_frame* hist[1] = {ref};
    orc_mapper_update(hist, 1, obj, 7u);
    orc_frame_destroy(ref);
    orc_frame_destroy(obj);

    /* the whole pipeline: mono (track + map) over a few frames, and sensor depth */
    orc_vo* vo = orc_vo_create(K, W, H, 11u, 1, 0);
    orc_cull_image(d0, W, H, 0, tmp);
    orc_vo_set_initial_depth(vo, tmp, s0);
    for (int f = 0; f < 6; f++) {
        scene(g1, d1, s1, 0.4f * (float)f, 2u + (uint32_t)f);
        for (int i = 0; i < W * H; i++)
            if (g1[i] <= ORC_INVALID) g1[i] = 0.5f;
        orc_vo_odometrize(vo, g1, T);
    }
    acc += (float)orc_vo_keyframe_count(vo) + (float)orc_vo_last_valid_updates(vo);
    acc += (orc_vo_keyframe(vo, 0) != NULL) + (orc_vo_last_frame(vo) != NULL);
    orc_vo_destroy(vo);
    vo = orc_vo_create(K, W, H, 11u, 1, 1);
    orc_vo_init_keyframe(vo, g0, d0, s0);
    scene(g1, d1, s1, 0.6f, 2u);
    orc_vo_odometrize_depth(vo, g1, d1, s1, T);
    orc_vo_destroy(vo);

    free(g0); free(d0); free(s0); free(g1); free(d1); free(s1); free(tmp); free(tmp2); free(tmp3); free(mask);
    printf("asan driver ok (%g)\n", (double)acc);
    return 0;
}
