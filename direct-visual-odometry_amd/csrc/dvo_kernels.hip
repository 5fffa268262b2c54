// dvo_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the direct-VO hot path.
//
// wave64, 256-thread workgroups, no MFMA (image warping bound by the vector ALUs and the L1 gather path).
// Reductions are fixed-order (DPP row_shr / row_bcast inside a wave, LDS across the 4 waves, double
// across workgroups) so every run is bit-reproducible.  Built with -ffp-contract=off: only the fmaf()
// calls written in dvo_math.h fuse.
#include <hip/hip_runtime.h>

#include "dvo_kernels.h"

namespace dvo {

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
// XCD-aware bijective remap of the linear workgroup id: workgroups b and b+8 share an XCD (round-robin
// dispatch), so give each XCD one contiguous range of tiles -> neighbouring tiles of an image (which share
// gather rows of ref_gray) hit the same 4 MiB L2.  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg)
{
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + (bid >> 3);
}

// wave64 sum, result valid in lane 63.  Classic GCN scan: row_shr 1,2,4,8 inside each row of 16 lanes,
// then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3.  Fixed order => deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}

__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_step<0x118, 0xf>(v);  // row_shr:8
    v = dpp_step<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v = dpp_step<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return v;
}

// ------------------------------------------------------------------------------------------------
// Packed wave64 reduction of 29 per-lane accumulators (fixed order, deterministic).
// A plain butterfly spends 6 DPP adds per value (174 VALU per wave) although half of the lanes carry useful data
// after every step.  Here every step also PACKS two registers into one, so the work halves each time (68 VALU):
//   32-lane step: v_permlane32_swap (gfx950) of a register pair + 1 add  -> low half = sum of A, high half = sum of B
//   16-lane step: v_permlane16_swap of a pair + 1 add                    -> one value per 16-lane row
//    8-lane step: row_shl:8 / row_shr:8 DPP adds of a pair + bank-masked merge -> one value per 8 lanes
//    4-lane step: row_shl:4 / row_shr:4, same                            -> one value per quad (2 registers left)
//    quad steps : quad_perm adds                                         -> every lane of a quad holds its value's total
// Lane L of output register q then holds value  16 q + {0,8,4,12}[(L>>2)&3] + {0,2,1,3}[L>>4].
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float swap32_add(float a, float b)
{
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float swap16_add(float a, float b)
{
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// lanes selected by BANK_MASK (4-lane banks inside each 16-lane row) take `hi`, the others keep `lo`
template <int BANK_MASK>
__device__ __forceinline__ float bank_merge(float lo, float hi)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lo), __float_as_int(hi), 0xE4 /*quad_perm identity*/, 0xf, BANK_MASK, false));
}

__device__ __forceinline__ int packed_slot_index(int lane, int q)
{
    const int quad = (lane >> 2) & 3, row = lane >> 4;
    const int qm = (quad == 0) ? 0 : (quad == 1) ? 8 : (quad == 2) ? 4 : 12;
    const int rm = (row == 0) ? 0 : (row == 1) ? 2 : (row == 2) ? 1 : 3;
    return 16 * q + qm + rm;
}

// v[0..28] in, out0/out1 as described above
__device__ __forceinline__ void wave_reduce29_packed(const float* v, float& out0, float& out1)
{
    float A[15];
#pragma unroll
    for (int j = 0; j < 14; j++) A[j] = swap32_add(v[2 * j], v[2 * j + 1]);
    A[14] = swap32_add(v[28], 0.0f);
    float B[8];
#pragma unroll
    for (int m = 0; m < 7; m++) B[m] = swap16_add(A[2 * m], A[2 * m + 1]);
    B[7] = swap16_add(A[14], 0.0f);
    float Cc[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const float lo = B[2 * n] + dpp_mov<0x108>(B[2 * n]);          // row_shl:8 -> lanes 0-7 of each row
        const float hi = B[2 * n + 1] + dpp_mov<0x118>(B[2 * n + 1]);  // row_shr:8 -> lanes 8-15
        Cc[n] = bank_merge<0xC>(lo, hi);
    }
    float D[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const float lo = Cc[2 * q] + dpp_mov<0x104>(Cc[2 * q]);          // row_shl:4 -> lanes 0-3, 8-11
        const float hi = Cc[2 * q + 1] + dpp_mov<0x114>(Cc[2 * q + 1]);  // row_shr:4 -> lanes 4-7, 12-15
        D[q] = bank_merge<0xA>(lo, hi);
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        D[q] = D[q] + dpp_mov<0xB1>(D[q]);  // quad_perm [1,0,3,2]
        D[q] = D[q] + dpp_mov<0x4E>(D[q]);  // quad_perm [2,3,0,1]
    }
    out0 = D[0];
    out1 = D[1];
}

__device__ __forceinline__ void split_index(int i, int w, float inv_w, int& x, int& y)
{  // i < 2^24: y = i / w without an integer divide; the +-1 fix-up is branch free
    y = (int)((float)i * inv_w);
    x = i - y * w;
    const int lo = x < 0 ? 1 : 0, hi = x >= w ? 1 : 0;
    y += hi - lo;
    x += (lo - hi) * w;
}

// ------------------------------------------------------------------------------------------------
// k_pyramid: System::Frame pyramid (frame.hpp:91-117, frame.cpp:16-37) = nearest-neighbour decimation
// (Convert::cullImage, convert.cpp:7-20) of up to three maps, all levels in ONE launch.
// One thread per TOP-level pixel reads src(x<<culls, y<<culls) once and writes every level it lands on
// (level t shifts below the top keep pixels whose coordinates are multiples of 2^t).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pyramid(PyramidArgs a)
{
    const int tw = a.w[a.levels - 1], th = a.h[a.levels - 1];
    const int seq = (int)(blockIdx.z * DVO_GRID_SEQ_Y + blockIdx.y);   // seq_grid() (dvo_kernels.h): no division, no 65535 limit
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= tw * th || seq >= a.n_seq) return;
    int x, y;
    split_index(i, tw, a.inv_tw, x, y);
    const size_t src_off = (size_t)seq * a.src_w * a.src_img_rows + (size_t)(y << a.src_row_shift) * a.src_w + (x << a.culls);
    float raw[3] = {0.0f, 0.0f, 0.0f};
    bool have[3];
    if (a.raw_rgb != nullptr) {  // raw sensor frame: convert exactly as k_ingest does (same float operations), only the kept pixels
        unsigned g8;
        if (a.raw_channels == 1) {
            g8 = __builtin_nontemporal_load(a.raw_rgb + src_off);
        } else {
            const uint8_t* p = a.raw_rgb + src_off * (size_t)a.raw_channels;
            g8 = ((unsigned)p[0] * 4899u + (unsigned)p[1] * 9617u + (unsigned)p[2] * 1868u + 8192u) >> 14;
        }
        raw[0] = (float)g8 * a.raw_gray_scale;
        have[0] = true; have[1] = have[2] = a.raw_depth != nullptr;
        if (a.raw_depth != nullptr) {
            const unsigned d = __builtin_nontemporal_load(a.raw_depth + src_off);
            raw[1] = (float)d * a.raw_depth_scale;
            raw[2] = d > 0 ? a.raw_sigma_valid : a.raw_sigma_invalid;
            if (a.raw_invalidate_gray && d == 0) raw[0] = kInvalid;
        }
    } else {
#pragma unroll
        for (int m = 0; m < 3; m++) {
            have[m] = a.src[m] != nullptr;
            if (have[m]) raw[m] = __builtin_nontemporal_load(a.src[m] + src_off);  // read once: streamed past the caches; the three loads are in flight together
        }
    }
    // the reference-frame constants of k_prep_ref, written while depth and sigma are in registers (needs both maps)
    const bool prep = a.wgt[0] != nullptr && have[1] && have[2];
    for (int t = 0; t < a.levels; t++) {
        const int msk = (1 << t) - 1;
        if ((x & msk) | (y & msk)) break;  // level t below the top keeps pixels whose coordinates are multiples of 2^t
        const int l = a.levels - 1 - t, lx = x >> t, ly = y >> t;
        if (lx >= a.w[l] || ly >= a.h[l]) continue;
        const size_t o = (size_t)seq * a.w[l] * a.h[l] + (size_t)ly * a.w[l] + lx;
        float val[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            // top level: cullImage(src, culls); culls == 0 aliases the input (convert.cpp:9-10), no pass_valid
            val[m] = (t == 0 && a.culls == 0) ? raw[m] : pass_valid(raw[m]);
            if (have[m] && a.dst[m][l] != nullptr) __builtin_nontemporal_store(val[m], a.dst[m][l] + o);  // (a map may be consumed without being kept; next read: a whole tracking step later)
        }
        if (prep) __builtin_nontemporal_store(gn_weight(a.step[l], a.sigma_min, a.sigma_max, val[2]), a.wgt[l] + o);
    }
}

// k_pyramid_raw4: the raw-sensor form of k_pyramid for 1-channel u8 gray (+ u16 depth) with CULLS = 1 or 2, four kept pixels
// per thread.  The scalar form spends one byte / short load per lane (a full vector-memory instruction for 1-2 useful bytes and
// half-empty sectors); here a thread owns 4 consecutive top-level pixels = 4 << CULLS source pixels of one row, fetched by one or
// two dwordx2/x4 loads, and writes its four top-level values per map as one dwordx4 store.  Same conversions, same float
// operations as k_ingest + k_pyramid: bit-identical (tests/test_frontend_and_eval.py).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int CULLS>
__global__ void __launch_bounds__(256) k_pyramid_raw4(PyramidArgs a)
{
    const int tw = a.w[a.levels - 1], th = a.h[a.levels - 1], gw = tw >> 2;
    const int seq = (int)(blockIdx.z * DVO_GRID_SEQ_Y + blockIdx.y);
    const int gi = (int)blockIdx.x * 256 + threadIdx.x;
    if (gi >= gw * th || seq >= a.n_seq) return;
    int y, xg;
    split_index(gi, gw, a.inv_tw * 4.0f, xg, y);   // (4 / tw = 1 / gw up to an ulp: split_index corrects +-1)
    const int x0 = xg << 2;
    const size_t src_off = (size_t)seq * a.src_w * a.src_img_rows + (size_t)(y << a.src_row_shift) * a.src_w + ((size_t)x0 << CULLS);
    constexpr int GW = 1 << CULLS;       // 32-bit words of gray bytes this thread reads (2 or 4)
    unsigned gwords[GW], dwords[2 * GW];
    const bool dep = a.raw_depth != nullptr;
    {
        const unsigned* gp = reinterpret_cast<const unsigned*>(a.raw_rgb + src_off);
        if constexpr (CULLS == 1) { const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(gp)); gwords[0] = v.x; gwords[1] = v.y; }
        else { const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gp)); gwords[0] = v.x; gwords[1] = v.y; gwords[2] = v.z; gwords[3] = v.w; }
        if (dep) {
            const u32x4* dp = reinterpret_cast<const u32x4*>(a.raw_depth + src_off);
#pragma unroll
            for (int q = 0; q < GW / 2; q++) {
                const u32x4 v = __builtin_nontemporal_load(dp + q);
                dwords[4 * q] = v.x; dwords[4 * q + 1] = v.y; dwords[4 * q + 2] = v.z; dwords[4 * q + 3] = v.w;
            }
        }
    }
    float raw[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int si = k << CULLS;                                            // source pixel of kept pixel k
        const unsigned g8 = (gwords[si >> 2] >> ((si & 3) * 8)) & 0xffu;
        raw[k][0] = (float)g8 * a.raw_gray_scale;
        raw[k][1] = 0.0f; raw[k][2] = 0.0f;
        if (dep) {
            const unsigned d = (dwords[si >> 1] >> ((si & 1) * 16)) & 0xffffu;
            raw[k][1] = (float)d * a.raw_depth_scale;
            raw[k][2] = d > 0 ? a.raw_sigma_valid : a.raw_sigma_invalid;
            if (a.raw_invalidate_gray && d == 0) raw[k][0] = kInvalid;
        }
    }
    const bool prep = a.wgt[0] != nullptr && dep;
    // top level (t = 0): cullImage(src, CULLS >= 1) -> pass_valid; four values per map, one 16-byte store
    {
        const int l = a.levels - 1;
        const size_t o = (size_t)seq * tw * th + (size_t)y * tw + x0;
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v[3], wgv;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float g = pass_valid(raw[k][0]), d = pass_valid(raw[k][1]), sg = pass_valid(raw[k][2]);
            v[0][k] = g; v[1][k] = d; v[2][k] = sg;
            wgv[k] = gn_weight(a.step[l], a.sigma_min, a.sigma_max, sg);
        }
        __builtin_nontemporal_store(v[0], reinterpret_cast<f4*>(a.dst[0][l] + o));
        if (dep && a.dst[1][l]) __builtin_nontemporal_store(v[1], reinterpret_cast<f4*>(a.dst[1][l] + o));
        if (dep && a.dst[2][l]) __builtin_nontemporal_store(v[2], reinterpret_cast<f4*>(a.dst[2][l] + o));
        if (prep) __builtin_nontemporal_store(wgv, reinterpret_cast<f4*>(a.wgt[l] + o));
    }
    for (int t = 1; t < a.levels; t++) {   // lower levels: pixels whose coordinates are multiples of 2^t
        const int msk = (1 << t) - 1;
        if (y & msk) break;
        const int l = a.levels - 1 - t, ly = y >> t;
        if (ly >= a.h[l]) continue;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = x0 + k;
            if (x & msk) continue;
            const int lx = x >> t;
            if (lx >= a.w[l]) continue;
            const size_t o = (size_t)seq * a.w[l] * a.h[l] + (size_t)ly * a.w[l] + lx;
            const float g = pass_valid(raw[k][0]), d = pass_valid(raw[k][1]), sg = pass_valid(raw[k][2]);
            __builtin_nontemporal_store(g, a.dst[0][l] + o);
            if (dep && a.dst[1][l]) __builtin_nontemporal_store(d, a.dst[1][l] + o);
            if (dep && a.dst[2][l]) __builtin_nontemporal_store(sg, a.dst[2][l] + o);
            if (prep) __builtin_nontemporal_store(gn_weight(a.step[l], a.sigma_min, a.sigma_max, sg), a.wgt[l] + o);
        }
    }
}

// k_cull: a single Convert::cullImage (operator-level parity)
__global__ void __launch_bounds__(256) k_cull(const float* __restrict__ src, int w, int h, int times, float* __restrict__ dst)
{
    const int dw = w >> times, dh = h >> times;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dw * dh) return;
    const int y = i / dw, x = i - y * dw;
    const float raw = src[(size_t)(y << times) * w + (x << times)];
    dst[i] = times > 0 ? pass_valid(raw) : raw;
}

// k_gradient: Convert::gradiate (convert.cpp:41-75), standalone parity op
__global__ void __launch_bounds__(256) k_gradient(const float* __restrict__ img, int w, int h, int xdir, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    GlobalImg g{img, w, h};
    out[i] = xdir ? grad_x_at(g, x, y) : grad_y_at(g, x, y);
}

// k_warp_image: Transform::warpImage (transform.cpp:35-51), standalone parity/visual op.  The fused tracker
// never materialises this image.
__global__ void __launch_bounds__(256) k_warp_image(const float* __restrict__ gray, const float* __restrict__ depth,
                                                    int w, int h, Intr k, Pose pose, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    const float d = depth[i];
    float v = kInvalid;
    if (!is_epsilon(d)) {
        float pu, pv;
        warp(pose, k, (float)x, (float)y, d, pu, pv);
        GlobalImg g{gray, w, h};
        v = get_subpixel(g, pu, pv);
    }
    out[i] = v;
}

// ------------------------------------------------------------------------------------------------
// k_track_gn: the fused Gauss-Newton accumulation = Transform::warpImage + Track::optimize's per-pixel
// lambda (optimize.cpp:28-90) + the A^T A / A^T B products that cv::solve's SVD stands for.
//
// HBM traffic per evaluated pixel, by contract: obj_gray 4 + ref_depth 4 + ref_sigma 4 (coalesced rows) + ref_gray 4
// (12-tap plus-shaped gather around the warped position, served by L1/L2 or the LDS patch) = 16 B.  Read in fact: the weight map
// step / clamp(sigma) in place of sigma -- or nothing at all for raw sensor frames, whose contributing pixels share one weight
// (GnArgs::wgt_const) -- and 1 / depth recomputed (recip_gated): 12-13 B.  Through the L1 it is 56 B per lane and pixel (8 B of rows,
// 48 B of taps), and that path, 78 % busy, limits the kernel together with the vector ALUs.
// Each thread owns PPT pixels (stride 256 => coalesced), keeps 29 accumulators (21 upper-tri J^T J, 6 J^T wr,
// sum r^2, count), then: DPP wave sum -> LDS across the 4 waves -> one 32-float partial per workgroup.
// ------------------------------------------------------------------------------------------------
// 1.0f / d for the lanes that pass the gates of optimize.cpp:33-48 (d >= min_depth > 0), the IEEE quotient bit for bit: the
// v_rcp_f32 + two-FMA form of dvo_math.h wherever it is proven equal to the division (|d| in [2^-100, 2^100], all floats compared
// on the device), the division itself for the wave when a gated lane lies outside.  Lanes that fail the gate get an unspecified
// finite-or-not value: their Jacobian row is replaced by zeros (a select, not a product) before it is used.
__device__ __forceinline__ float recip_gated(float d, bool gate)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r = recip_fast(d);
    const float ad = fabsf(d);
    const bool odd = gate & !((ad >= DVO_RECIP_FAST_MIN) & (ad <= DVO_RECIP_FAST_MAX));   // (true for NaN; min_depth is configurable)
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull) {       // wave-uniform; never taken for a real depth map
        asm volatile("; recip_gated: IEEE division fallback");
        if (odd) r = 1.0f / d;
    }
    return r;
#else
    (void)gate;
    return 1.0f / d;
#endif
}

struct Acc29 {
    float a[29];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int i = 0; i < 29; i++) a[i] = 0.0f;
    }
    __device__ __forceinline__ void add(const float J[6], float r, float rw, float one)
    {
        int idx = 0;
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = p; q < 6; q++) {
                a[idx] = fmaf(J[p], J[q], a[idx]);
                idx++;
            }
#pragma unroll
        for (int p = 0; p < 6; p++) a[21 + p] = fmaf(J[p], rw, a[21 + p]);
        a[27] = fmaf(r, r, a[27]);
        a[28] += one;
    }
};

// dword-aligned wide loads (gfx950 global memory takes unaligned dwordx2/x4)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

// the generic sampler as a real function: rare (border / INVALID taps), keeps the hot loop's code small
struct SlowSample {  // returned in registers (an out-pointer would put the caller's locals in scratch memory)
    float I2, gx, gy;
    int ok;
};
__device__ __noinline__ SlowSample gn_sample_slow(const float* img, int w, int h, float d, float u, float v)
{
    const GlobalImg ref{img, w, h};
    SlowSample s;
    s.ok = gn_sample(ref, d, u, v, s.I2, s.gx, s.gy) ? 1 : 0;
    return s;
}

// The 12 plus-shaped taps around (x0, y0) that warped gray (4 taps) and its central-difference gradient need.
struct Taps {
    f2u ra;  // row y0-1: cols x0, x0+1
    f4u rb;  // row y0  : cols x0-1 .. x0+2
    f4u rc;  // row y0+1: cols x0-1 .. x0+2
    f2u rd;  // row y0+2: cols x0, x0+1
};

// Interior fast path.  When the 4x4 neighbourhood of the warped position lies inside the image and its 12 taps are
// all valid, getSubpixel's fill loop is a no-op and Convert::gradiate has no border/invalid case, so warped gray
// and gradient reduce to straight differences and three blend4() calls -- the SAME float operations the generic
// path (dvo_math.h gn_sample) performs, without its per-tap branches.  Returns 1 = sampled, 0 = pixel rejected,
// -1 = not decidable here (INVALID or NaN tap): take the generic path.
// v_min3_f32 without the canonicalising v_max that fminf() adds to every loaded operand.  NaN operands are either
// skipped (caught by the probe below) or returned (fails the > test): both end in the generic path.
__device__ __forceinline__ float min3_raw(float a, float b, float c)
{
    float m;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    return m;
}

__device__ __forceinline__ void gn_sample_fast(const Taps& t, float u, float v, int x0, int y0, float& I2, float& gx, float& gy,
                                               bool& decidable, bool& valid)
{  // branch free: everything is computed; decidable = no INVALID / NaN tap (else: generic path), valid = pixel contributes
    const float mn = min3_raw(min3_raw(min3_raw(t.ra.x, t.ra.y, t.rd.x), min3_raw(t.rb.x, t.rb.y, t.rb.z), min3_raw(t.rc.x, t.rc.y, t.rc.z)),
                              min3_raw(t.rd.y, t.rb.w, t.rc.w), t.rd.y);
    const float hx = u - (float)x0, vy = v - (float)y0;   // (v_fract_f32 gives the same bits for u >= 1 and saves two conversions: measured, no gain)
    I2 = blend4(t.rb.y, t.rb.z, t.rc.y, t.rc.z, hx, vy);
    gx = blend4(t.rb.z - t.rb.x, t.rb.w - t.rb.y, t.rc.z - t.rc.x, t.rc.w - t.rc.y, hx, vy);
    gy = blend4(t.rc.y - t.ra.x, t.rc.z - t.ra.y, t.rd.x - t.rb.y, t.rd.y - t.rb.z, hx, vy);
    const float probe = (I2 + gx) + gy;  // a NaN tap (fminf skips NaN) poisons at least one of the three
    decidable = (mn > kInvalid) & (probe == probe);
    valid = ((int)is_invalid(I2) | (int)is_invalid(gx) | (int)is_invalid(gy)) == 0;  // bitwise on purpose: no short-circuit branches
}

// The generic sampler (dvo_math.h gn_sample: any position, any validity pattern -- the fill quirk of getSubpixel, INVALID gradients at the
// image border and next to INVALID taps, clamping of missing taps to g00) with every tap it can touch requested UP FRONT from clamped
// coordinates: one memory round trip instead of the five or six dependent ones of the branchy version, its semantics then evaluated from
// registers with selects.  Written for k_track_persist, where every step waits for its slowest tile and the slowest tiles are those with
// border pixels (5.3-6.1 us against 2.85 us for an interior tile: 6.1 -> 5.0 us, profiles/r03_single_cpp_final.txt); then measured in
// the throughput kernels as well: +3.0 % frames/s on the 16 384-sequence batch (311.5 k -> 320.8 k, profiles/r03_patch_sampler_ab.txt) --
// a round-1 attempt of the same idea had been 20 % slower on the probe and was the reason the branchy version stayed this long.
// The same float operations on the same values: bit-identical results (tests/test_gpu_parity.py: operator-level masks against the
// oracle, the schedules against each other, frames full of INVALID blocks, black pixels and depth holes).
__device__ __forceinline__ bool gn_sample_patch(const float* __restrict__ img, const int w, const int h, const float d, const float u, const float v,
                                                float& I2, float& gx, float& gy)
{
    I2 = kInvalid; gx = kInvalid; gy = kInvalid;
    if (is_epsilon(d) || !coord_ok(u) || !coord_ok(v)) return false;   // transform.cpp:44; load_taps' first gate
    const int x0 = (int)u, y0 = (int)v;
    if (x0 < 0 || w <= x0 || y0 < 0 || h <= y0) return false;
    const bool hasXm = x0 >= 1, inx = x0 + 1 < w, hasX2 = x0 + 2 < w, hasYm = y0 >= 1, iny = y0 + 1 < h, hasY2 = y0 + 2 < h;
    const int xm = hasXm ? x0 - 1 : x0, x1 = inx ? x0 + 1 : x0, x2 = hasX2 ? x0 + 2 : x0;
    const int rm = (hasYm ? y0 - 1 : y0) * w, r0 = y0 * w, r1 = (iny ? y0 + 1 : y0) * w, r2 = (hasY2 ? y0 + 2 : y0) * w;
    const float Tm0 = img[rm + x0], Tm1 = img[rm + x1];
    const float T0m = img[r0 + xm], T00 = img[r0 + x0], T01 = img[r0 + x1], T02 = img[r0 + x2];
    const float T1m = img[r1 + xm], T10 = img[r1 + x0], T11 = img[r1 + x1], T12 = img[r1 + x2];
    const float T20 = img[r2 + x0], T21 = img[r2 + x1];
    const float hx = u - (float)x0, vy = v - (float)y0;
    float g[4] = {T00, inx ? T01 : T00, iny ? T10 : T00, (inx && iny) ? T11 : T00};   // load_taps: a missing tap is g00
    if (!fill_quirk(g)) return false;
    I2 = blend4(g[0], g[1], g[2], g[3], hx, vy);
    if (is_invalid(I2)) return false;
    if (u < 0.0f || v < 0.0f || (float)w <= u || (float)h <= v) return false;       // optimize.cpp:52-56
    // Convert::gradiate at the four corners (grad_x_at / grad_y_at): INVALID unless both neighbours exist and are valid
    auto diff = [](bool exists, float a, float b) { return (exists && is_valid(a) && is_valid(b)) ? b - a : kInvalid; };
    const float gx00 = diff(hasXm && inx, T0m, T01);
    const float gx10 = inx ? diff(hasX2, T00, T02) : gx00;
    const float gx01 = iny ? diff(hasXm && inx, T1m, T11) : gx00;
    const float gx11 = (inx && iny) ? diff(hasX2, T10, T12) : gx00;
    const float gy00 = diff(hasYm && iny, Tm0, T10);
    const float gy10 = inx ? diff(hasYm && iny, Tm1, T11) : gy00;
    const float gy01 = iny ? diff(hasY2, T00, T20) : gy00;
    const float gy11 = (inx && iny) ? diff(hasY2, T01, T21) : gy00;
    gx = blend4(gx00, gx10, gx01, gx11, hx, vy);
    gy = blend4(gy00, gy10, gy01, gy11, hx, vy);
    return !(is_invalid(gx) || is_invalid(gy));
}

// k_prep_ref: the per-pixel constant of a reference frame (all levels, one launch): wgt = step(level) / clamp(sigma) -- the
// division of optimize.cpp:83-84, which does not depend on the pose, evaluated once per frame instead of once per Gauss-Newton
// iteration.  (1 / depth, optimize.cpp:70-74, was a second such map in round 1: 4 B per pixel and iteration through HBM and the L1
// for five instructions; the kernels now recompute it -- recip_rn, the IEEE quotient bit for bit.)
__global__ void __launch_bounds__(256) k_prep_ref(PrepArgs a)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.level_end[a.levels - 1]) return;
    float step = a.step[0];
#pragma unroll
    for (int l = 1; l < DVO_MAX_LEVELS; l++)
        if (l < a.levels && i >= a.level_end[l - 1]) step = a.step[l];
    a.wgt[i] = gn_weight(step, a.sigma_min, a.sigma_max, a.sigma[i]);
}

// LDS of one gn_tile() evaluation (the caller owns it: k_track_gn once, k_track_level once per tile and iteration)
template <int PPT>
struct GnTileLds {
    float red[4][32];
    int slow_q[4][PPT * 64];  // deferred pixels (generic sampler), one queue per wave
    int slow_cnt[4];
};

// gn_tile: one 256 x PPT pixel tile (`blk`) of sequence `seq` at pose `pose` -> its 32-float partial row `out_row`
// (global or LDS).  Called by all 256 threads of a workgroup; contains two barriers.
// T2D: the tile is 64 columns x 4*PPT rows (lane = column, wave w owns rows w*PPT .. w*PPT+PPT-1) instead of 256*PPT
// consecutive raster pixels.  Used when the level width is a multiple of 64: only tiles on the image border then hold
// deferred (border) pixels, a thread's pixels share their column (one int->float conversion and one (x - cx) for PPT
// pixels) and there is no row-wrap arithmetic.
template <int PPT, int G, bool MASK, bool T2D>
__device__ __forceinline__ void gn_tile(const GnArgs& a, const Pose& pose, const int seq, const int blk, GnTileLds<PPT>& lds,
                                        float* out_row);

template <int PPT, int G, bool MASK, bool T2D = false>
#if !defined(DVO_GN_WAVES)
#define DVO_GN_WAVES 6   /* waves per SIMD the hot variants are compiled for: 6 = up to 84 VGPRs (78 used, no scratch); at 7 (72 VGPRs) the border sampler spills 24 bytes per lane: 54 MB of extra HBM writes per full-batch launch for the same speed (profiles/r03_patch_sampler_ab.txt) */
#endif
__global__ void __launch_bounds__(256, (PPT <= 4 && G <= 2) ? DVO_GN_WAVES : 1) k_track_gn(GnArgs a)
{
    __shared__ GnTileLds<PPT> lds;
    // Workgroup 0 clears the counter of the list the following k_gn_solve appends to.  The store comes last on every path:
    // a store ahead of the list / pose loads would make the compiler fetch those through the vector memory path.
    auto clear_next = [&]() {
        if (__builtin_amdgcn_readfirstlane((int)blockIdx.x) == 0 && a.next_count) {  // scalar test first (no VGPR kept for it)
            if (threadIdx.x == 0) *a.next_count = 0;
        }
    };
    // The grid covers the tiles of all sequences, but only those of the ACTIVE ones exist: the workgroups of XCD x
    // (blockIdx % 8 == x) take the x-th contiguous eighth of the n_tiles live tiles (neighbouring tiles share an L2) and
    // every other workgroup leaves after this one scalar load -- converged sequences cost (almost) nothing.
    // Tiles whose rows all lie outside the crop window (optimize.cpp:33-36) are not launched either: only tiles
    // [blk_first, blk_first + blk_count) of a sequence exist, and k_gn_solve reads the others as zero.
    const int n_tiles = (a.list ? a.list[0] : a.n_seq) * a.blk_count;
    const int t8 = (n_tiles + 7) >> 3, xcd = blockIdx.x & 7, tile_in_xcd = (int)(blockIdx.x >> 3);
    const int tile_id = xcd * t8 + tile_in_xcd;
    if (tile_in_xcd >= t8 || tile_id >= n_tiles) {
        clear_next();
        return;
    }
    const int slot = tile_id / a.blk_count, blk = a.blk_first + (tile_id - slot * a.blk_count);
    const int seq = a.list ? a.list[4 + slot] : slot;
    const Pose pose = a.state[seq].pose;              // wave-uniform -> scalar loads
    gn_tile<PPT, G, MASK, T2D>(a, pose, seq, blk, lds, a.partials + ((size_t)seq * a.nblk + blk) * 32);
    clear_next();
}

template <int PPT, int G, bool MASK, bool T2D>
__device__ __forceinline__ void gn_tile(const GnArgs& a, const Pose& pose, const int seq, const int blk, GnTileLds<PPT>& lds,
                                        float* out_row)
{
    float (&red)[4][32] = lds.red;
    int (&slow_q)[4][PPT * 64] = lds.slow_q;
    int (&slow_cnt)[4] = lds.slow_cnt;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: lives in an SGPR
    const int w = a.w, h = a.h, npix = w * h;
    int nslow = 0;                                    // wave-uniform
    const size_t img_off = (size_t)seq * a.w * a.h;
    const float* __restrict__ obj = a.obj_gray + img_off;
    const float* __restrict__ dep = a.ref_depth + img_off;
    const float* __restrict__ wgp = a.ref_wgt ? a.ref_wgt + img_off : nullptr;   // nullptr: one weight for every contributing pixel
    const float* __restrict__ refp = a.ref_gray + img_off;
    const float wlim = (float)(w - 2), hlim = (float)(h - 2);

    Acc29 acc;
    acc.zero();
    // G pixels per thread have their gathers in flight together (memory-level parallelism hides the L2/HBM latency)
    static_assert(PPT % G == 0, "PPT must be a multiple of G");
    // pixel k of this thread: coordinates (xA, yA), linear index iA (clamped into the image), inA = it exists
    int xA[PPT], yA[PPT], iA[PPT];
    bool inA[PPT];
    if constexpr (T2D) {  // GnTiling (dvo_kernels.h): TW = 2^t_shift columns, 64 / TW rows per wave-load
        const int sh = a.t_shift, rw = 64 >> sh;
        const int ty = blk / a.tiles_x, tx = blk - ty * a.tiles_x;
        const int x = a.x_org + (tx << sh) + (lane & ((1 << sh) - 1));
        const int y0 = a.y_org + (ty * (4 * PPT) + wave * PPT) * rw + (lane >> sh);
        const int xc = x < w ? x : w - 1;
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            xA[k] = x; yA[k] = y0 + k * rw;
            inA[k] = (x < w) & (yA[k] < h);
            iA[k] = (int)__umul24((unsigned)(yA[k] < h ? yA[k] : h - 1), (unsigned)w) + xc;
        }
    } else {
        const int base = blk * (256 * PPT) + threadIdx.x;
        // (x, y) of this thread's first pixel by one index split; every further pixel is 256 later in raster order
        split_index(base < npix ? base : npix - 1, w, a.inv_w, xA[0], yA[0]);
#pragma unroll
        for (int k = 1; k < PPT; k++) {
            const int xn = xA[k - 1] + a.r256;
            const int wrap = xn >= w ? 1 : 0;
            xA[k] = xn - wrap * w;
            yA[k] = yA[k - 1] + a.q256 + wrap;
        }
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int i = base + k * 256;
            inA[k] = i < npix;  // (coordinates past the end of the image are garbage: gated out)
            iA[k] = inA[k] ? i : npix - 1;
        }
    }
    // every coalesced row load of this thread's PPT pixels is issued up front (independent of the pose): one exposed
    // memory round trip per wave instead of one per group
    float dA[PPT], I1A[PPT], wgA[PPT];
#pragma unroll
    for (int k = 0; k < PPT; k++) {  // ref_depth, obj_gray, weight
        const unsigned ic = (unsigned)iA[k] * 4u;  // byte offset: SGPR base + 32-bit VGPR offset
        dA[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(dep) + ic);
        I1A[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(obj) + ic);
        wgA[k] = a.wgt_const;
        if (wgp) wgA[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(wgp) + ic);   // (wave-uniform)
    }
#pragma unroll
    for (int g0 = 0; g0 < PPT; g0 += G) {
        float d[G], I1[G], iz[G], wg[G], u[G], v[G];
        int xs[G], ys[G], x0[G], y0[G];
        bool gate[G], inter[G];
        Taps t[G];
#pragma unroll
        for (int k = 0; k < G; k++) {
            d[k] = dA[g0 + k]; I1[k] = I1A[g0 + k]; wg[k] = wgA[g0 + k];
        }
#pragma unroll
        for (int k = 0; k < G; k++) {  // gates, warp, issue the gathers (always from a safe address)
            xs[k] = xA[g0 + k]; ys[k] = yA[g0 + k];
            // gn_gate (optimize.cpp:33-48) written with bitwise ops so it stays a predicate, not a branch
            bool crop_ok = true;
            if (a.prm.crop) crop_ok = (xs[k] >= 20) & (xs[k] <= 140) & (ys[k] >= 20) & (ys[k] <= 100);  // wave-uniform branch
            gate[k] = inA[g0 + k] & crop_ok & !(d[k] < a.prm.min_depth) & !is_invalid(I1[k]);
            iz[k] = recip_gated(d[k], gate[k]);   // 1.0f / depth (optimize.cpp:70-74) of the pixels that can contribute
            warp(pose, a.k, (float)xs[k], (float)ys[k], d[k], u[k], v[k]);
            inter[k] = gate[k] & (u[k] >= 1.0f) & (v[k] >= 1.0f) & (u[k] < wlim) & (v[k] < hlim);  // false for NaN
            x0[k] = inter[k] ? (int)u[k] : 1;
            y0[k] = inter[k] ? (int)v[k] : 1;
            // unsigned 32-bit element offsets from the wave-uniform base: SGPR-base + VGPR-offset addressing, no 64-bit VALU math
            // (24-bit multiply: full rate, v_mul_lo_u32 is quarter rate; y0 < h and w are far below 2^24)
            const unsigned c = (__umul24((unsigned)y0[k], (unsigned)w) + (unsigned)x0[k]) * 4u, uw = (unsigned)w * 4u;  // BYTE offsets (< 2^26)
            const char* rb8 = reinterpret_cast<const char*>(refp);
            t[k].ra = *reinterpret_cast<const f2u*>(rb8 + (c - uw));
            t[k].rb = *reinterpret_cast<const f4u*>(rb8 + (c - 4u));
            t[k].rc = *reinterpret_cast<const f4u*>(rb8 + (c + uw - 4u));
            t[k].rd = *reinterpret_cast<const f2u*>(rb8 + (c + 2u * uw));
        }
        // The G pixels are sampled in ONE straight-line block (independent chains interleave: ILP), the rare generic
        // sampler runs in a single separate region, then Jacobians and sums again in one straight-line block.
        float I2[G], gx[G], gy[G];
        bool okf[G];
#pragma unroll
        for (int k = 0; k < G; k++) {
            bool decidable, valid;
            gn_sample_fast(t[k], u[k], v[k], x0[k], y0[k], I2[k], gx[k], gy[k], decidable, valid);
            const bool fast = inter[k] & decidable;  // (inter implies gate)
            okf[k] = fast & valid;
            // Everything else that passed the gate -- the band along the border, positions outside the image (rejected by
            // optimize.cpp:52-56), INVALID or NaN taps -- is left to the generic sampler.  A few lanes per wave need it, so
            // running it here would cost the whole wave ~350 instructions and several dependent round trips per occurrence;
            // instead the pixel index is queued (per wave, lane order: deterministic) and the workgroup evaluates all of
            // its deferred pixels densely after the main loop.
            const bool cand = gate[k] & !fast;
            if (__ballot(cand) == 0ull) continue;  // wave-uniform; the common case in the interior of the image
            // (a position outside [0,w) x [0,h), or NaN, is rejected by optimize.cpp:52-56 whatever the samplers say)
            const bool inside = (u[k] >= 0.0f) & (v[k] >= 0.0f) & (u[k] < (float)w) & (v[k] < (float)h);
            const bool slow = cand & inside;
            const unsigned long long bal = __ballot(slow);
            if (bal != 0ull) {  // wave-uniform
                if (slow) {
                    const int pos = nslow + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    slow_q[wave][pos] = (int)__umul24((unsigned)ys[k], (unsigned)w) + xs[k];  // (recomputed: iA is dead after the loads)
                }
                nslow += __popcll(bal);
            }
        }
#pragma unroll
        for (int k = 0; k < G; k++) {
            // predicated accumulation: rejected pixels add exact zeros, so no control flow merges the 29 accumulators
            const bool ok = okf[k];
            float J[6], r, rw;
            gn_jacobian_pre(a.k, xs[k], ys[k], d[k], iz[k], wg[k], gx[k], gy[k], I1[k], I2[k], J, r, rw);
#pragma unroll
            for (int q = 0; q < 6; q++) J[q] = ok ? J[q] : 0.0f;
            acc.add(J, ok ? r : 0.0f, ok ? rw : 0.0f, ok ? 1.0f : 0.0f);
            if (MASK && ok) a.mask[img_off + (size_t)(ys[k] * w + xs[k])] = 1;
        }
    }
    // deferred pixels: the four wave queues, concatenated in wave order, are spread densely over the workgroup's threads
    if (lane == 0) slow_cnt[wave] = nslow;
    __syncthreads();
    {
        const int c0 = slow_cnt[0], c1 = c0 + slow_cnt[1], c2 = c1 + slow_cnt[2], total = c2 + slow_cnt[3];
        for (int e = threadIdx.x; e < total; e += 256) {
            const int qw = (e >= c0) + (e >= c1) + (e >= c2);
            const int qoff = qw == 0 ? 0 : (qw == 1 ? c0 : (qw == 2 ? c1 : c2));
            const int i = slow_q[qw][e - qoff];  // < npix (only gated pixels are queued)
            int x, y;
            split_index(i, w, a.inv_w, x, y);
            // (carrying d and I1 through LDS with the index instead of re-loading them: measured in the LAT form, no change)
            const float d = dep[i], I1 = obj[i], iz = recip_gated(dep[i], true), wg = wgp ? wgp[i] : a.wgt_const;
            float u, v;
            warp(pose, a.k, (float)x, (float)y, d, u, v);  // same operations on the same inputs as in the main loop
            // inlined (single site): a call here would pin the 29 live accumulators to callee-saved registers and
            // raise the kernel's VGPR allocation.  (Tried: the sampler on a register patch of 12 taps loaded up front from
            // clamped coordinates -- one round trip, no divergent loads.  Slower by 20 % on the probe: the branches of the
            // memory version skip most of the work for most border pixels.)
            SlowSample ss;
            ss.ok = gn_sample_patch(refp, w, h, d, u, v, ss.I2, ss.gx, ss.gy) ? 1 : 0;
            const bool ok = ss.ok != 0;
            float J[6], r, rw;
            gn_jacobian_pre(a.k, x, y, d, iz, wg, ss.gx, ss.gy, I1, ss.I2, J, r, rw);
#pragma unroll
            for (int q = 0; q < 6; q++) J[q] = ok ? J[q] : 0.0f;
            acc.add(J, ok ? r : 0.0f, ok ? rw : 0.0f, ok ? 1.0f : 0.0f);
            if (MASK && ok) a.mask[img_off + i] = 1;
        }
    }
    // wave reduction (DPP), then 4 waves through LDS in fixed order
    {
        float o0, o1;
        wave_reduce29_packed(acc.a, o0, o1);
        if ((lane & 3) == 0) {  // one lane per quad publishes its value (slots 29..31 are zero padding)
            red[wave][packed_slot_index(lane, 0)] = o0;
            red[wave][packed_slot_index(lane, 1)] = o1;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int c = threadIdx.x;
        float s = 0.0f;
        if (c < 29) s = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
        out_row[c] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// k_track_gn_tile: same arithmetic as k_track_gn, different data movement.  A workgroup owns a 64 x (4*PPT)
// pixel tile; it first issues ALL of its global loads in one burst -- the coalesced rows of its own pixels and
// the reference-gray patch (tile grown by `margin` pixels plus the 4x4 tap footprint) into LDS -- then, after one
// barrier, every warped sample is a short-latency LDS read instead of an L1/L2 gather.  Pixels whose footprint
// leaves the staged patch (large motion) gather from global memory; borders / INVALID taps take the generic
// sampler.  All three sources hold the same floats, so results are bit-identical.
// ------------------------------------------------------------------------------------------------
template <int PPT, bool MASK>
__global__ void __launch_bounds__(256) k_track_gn_tile(GnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [0,128): reduction scratch, then the patch
    float* red = smem;
    float* patch = smem + 128;
    const unsigned id = xcd_remap(blockIdx.x, gridDim.x);
    const int seq = id / a.nblk, tile = id - seq * a.nblk;
    const SeqState& st = a.state[seq];
    if (!a.ignore_active && st.active == 0) return;  // converged sequences cost nothing
    const Pose pose = st.pose;
    const int w = a.w, h = a.h;
    constexpr int TH = 4 * PPT;
    const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
    const int tx0 = txi * 64, ty0 = tyi * TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t img_off = (size_t)seq * w * h;
    const float* __restrict__ obj = a.obj_gray + img_off;
    const float* __restrict__ dep = a.ref_depth + img_off;
    const float* __restrict__ wgp = a.ref_wgt ? a.ref_wgt + img_off : nullptr;
    const float* __restrict__ refp = a.ref_gray + img_off;

    // ---- one burst of global loads: own pixels (registers) + reference patch (LDS) ----
    const int x = tx0 + lane;
    float d[PPT], I1[PPT], iz[PPT], wg[PPT];
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int y = ty0 + wave + 4 * k;
        const int xc = x < w ? x : w - 1, yc = y < h ? y : h - 1;  // clamped: no branch, gated out below
        const int i = yc * w + xc;
        d[k] = dep[i];
        I1[k] = obj[i];
        iz[k] = 0.0f;
        wg[k] = wgp ? wgp[i] : a.wgt_const;
    }
    const int M = a.margin, PW = 64 + 2 * M + 3;
    const int px0 = max(tx0 - M - 1, 0), px1 = min(tx0 + 64 + M + 2, w);  // staged columns [px0, px1)
    const int py0 = max(ty0 - M - 1, 0), py1 = min(ty0 + TH + M + 2, h);  // staged rows    [py0, py1)
    for (int r = py0 + wave; r < py1; r += 4) {
        const float* src = refp + r * w;
        float* dst = patch + (r - py0) * PW;
        for (int c = px0 + lane; c < px1; c += 64) dst[c - px0] = src[c];
    }
    __syncthreads();

    const float wlim = (float)(w - 2), hlim = (float)(h - 2);
    Acc29 acc;
    acc.zero();
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int y = ty0 + wave + 4 * k;
        const int crop_ok = (a.prm.crop == 0) | ((x >= 20) & (x <= 140) & (y >= 20) & (y <= 100));
        const bool gate = (x < w) & (y < h) & (crop_ok != 0) & !(d[k] < a.prm.min_depth) & !is_invalid(I1[k]);
        iz[k] = recip_gated(d[k], gate);
        float u, v;
        warp(pose, a.k, (float)x, (float)y, d[k], u, v);
        const bool inter = gate & (u >= 1.0f) & (v >= 1.0f) & (u < wlim) & (v < hlim);  // false for NaN
        const int x0 = inter ? (int)u : 1, y0 = inter ? (int)v : 1;
        const bool inpatch = inter & (x0 - 1 >= px0) & (x0 + 2 < px1) & (y0 - 1 >= py0) & (y0 + 2 < py1);
        Taps t;
        {  // LDS taps (address forced inside the patch for lanes that will not use them)
            const int lx = inpatch ? x0 - px0 : 1, ly = inpatch ? y0 - py0 : 1;
            const float* q = patch + ly * PW + lx;
            t.ra.x = q[-PW]; t.ra.y = q[-PW + 1];
            t.rb.x = q[-1]; t.rb.y = q[0]; t.rb.z = q[1]; t.rb.w = q[2];
            t.rc.x = q[PW - 1]; t.rc.y = q[PW]; t.rc.z = q[PW + 1]; t.rc.w = q[PW + 2];
            t.rd.x = q[2 * PW]; t.rd.y = q[2 * PW + 1];
        }
        if (inter & !inpatch) {  // footprint left the staged patch (large motion): gather from global memory
            const float* p = refp + (y0 * w + x0);
            t.ra = *reinterpret_cast<const f2u*>(p - w);
            t.rb = *reinterpret_cast<const f4u*>(p - 1);
            t.rc = *reinterpret_cast<const f4u*>(p + w - 1);
            t.rd = *reinterpret_cast<const f2u*>(p + 2 * w);
        }
        float I2 = 0.0f, gx = 0.0f, gy = 0.0f;
        bool decidable, valid;
        gn_sample_fast(t, u, v, x0, y0, I2, gx, gy, decidable, valid);
        int s = decidable ? (valid ? 1 : 0) : -1;
        const bool inside = (u >= 0.0f) & (v >= 0.0f) & (u < (float)w) & (v < (float)h);  // outside: rejected (optimize.cpp:52-56)
        s = (gate & inside) ? (inter ? s : -1) : 0;
        if (s < 0) {  // border, INVALID or NaN taps: the generic sampler decides (rare; a real function call)
            const SlowSample ss = gn_sample_slow(refp, w, h, d[k], u, v);
            s = ss.ok; I2 = ss.I2; gx = ss.gx; gy = ss.gy;
        }
        const bool ok = s > 0;
        float J[6], r, rw;
        gn_jacobian_pre(a.k, x, y, d[k], iz[k], wg[k], gx, gy, I1[k], I2, J, r, rw);
#pragma unroll
        for (int q = 0; q < 6; q++) J[q] = ok ? J[q] : 0.0f;
        acc.add(J, ok ? r : 0.0f, ok ? rw : 0.0f, ok ? 1.0f : 0.0f);
        if (MASK && ok) a.mask[img_off + y * w + x] = 1;
    }
    // wave reduction (DPP, step-major), then the 4 waves through LDS in fixed order
    {
        float o0, o1;
        wave_reduce29_packed(acc.a, o0, o1);
        if ((lane & 3) == 0) {
            red[wave * 32 + packed_slot_index(lane, 0)] = o0;
            red[wave * 32 + packed_slot_index(lane, 1)] = o1;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int c = threadIdx.x;
        float s = 0.0f;
        if (c < 29) s = ((red[c] + red[32 + c]) + red[64 + c]) + red[96 + c];
        a.partials[((size_t)seq * a.nblk + tile) * 32 + c] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// k_gn_solve: second reduction stage + the rest of one Tracker::track iteration (tracker.cpp:44-73),
// one 64-thread workgroup per sequence, entirely on the device:
//   partial sums -> double, fixed order;  xi_update = H^+ g (LDL^T / eigen pseudo-inverse, double);
//   xi <- log(exp(xi) exp(xi_update)) unless NaN (testXi);  pose <- exp(-xi);  stop tests.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int solve_finish(const SolveArgs& a, const int seq, SeqState& st, const double* tot, const int first,
                                            const int it_prev, float xi[6], double Tc[12], Pose& np);

// DVO_SOLVE_SEQ sequences per 256-thread workgroup.  The serial chain below (6x6 solve, exp / log in double: ~4.4 us,
// ~250 VGPRs) is the same instruction stream for every sequence, so lanes 0..7 of wave 0 run it for 8 sequences in lockstep
// at the price of one; with one sequence per workgroup a 4096-sequence batch needed several rounds of workgroups, each
// waiting out a full chain.  Stage one: team t (32 lanes = the 32 columns of a partial row) sums the rows of sequence t.
#define DVO_SOLVE_SEQ 8
#define DVO_SOLVE_GROUPS 4   /* row classes of the fixed summation order: rows b = g (mod 4) in batches of 8, then (s0+s1)+(s2+s3) */
// Second reduction stage for one column of one sequence: the workgroup partials p[b * 32], b = 0 .. nblk-1, summed in double in a
// FIXED order -- four row classes (b mod 4), 32 rows per batch with all 32 loads in flight, then (s0 + s1) + (s2 + s3) -- shared by
// k_gn_solve, k_track_gn_fused and (re-stated on LDS rows) k_track_level, so every schedule gives the same bits.
// Rows outside [blk_first, blk_first + blk_count) were not written (crop window): they count as exact zeros in the same slots.
__device__ __forceinline__ double sum_partial_rows(const float* p, const int nblk, const int blk_first, const int blk_count)
{
    const int live0 = blk_count < 0 ? 0 : blk_first, live1 = blk_count < 0 ? nblk : blk_first + blk_count;
    double sg[DVO_SOLVE_GROUPS] = {0.0, 0.0, 0.0, 0.0};
    for (int b0 = 0; b0 < nblk; b0 += 8 * DVO_SOLVE_GROUPS) {  // 32 loads in flight per lane
        float v[DVO_SOLVE_GROUPS][8];
#pragma unroll
        for (int g = 0; g < DVO_SOLVE_GROUPS; g++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int b = b0 + g + DVO_SOLVE_GROUPS * j;
                const bool live = (b >= live0) & (b < live1);
                const float x = p[(size_t)(live ? b : 0) * 32];  // (always a valid address: no branch around the load)
                v[g][j] = live ? x : 0.0f;
            }
#pragma unroll
        for (int g = 0; g < DVO_SOLVE_GROUPS; g++)
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (b0 + g < nblk) sg[g] += (double)v[g][j];  // (a class with no row left adds nothing, as before)
    }
    return (sg[0] + sg[1]) + (sg[2] + sg[3]);
}

// One row class g of sum_partial_rows() -- the rows b = g (mod 4) in increasing b, i.e. exactly the additions that function performs
// for sg[g], in its order -- with EVERY load of the class issued before the first addition (up to DVO_WIDE_BATCHES x 8 rows per lane:
// nblk <= 32 * DVO_WIDE_BATCHES).  One sequence gives a 32-lane team per class instead of one team for all four: a dvo_vo handle's
// finest level (300 partial rows) took ten dependent memory round trips here, ~15 us of its 20 us solve kernel; now one.
// (s0 + s1) + (s2 + s3) of the four results are the bits sum_partial_rows() returns.
#define DVO_WIDE_BATCHES 10
__device__ __forceinline__ double sum_partial_class(const float* p, const int nblk, const int blk_first, const int blk_count, const int g)
{
    const int live0 = blk_count < 0 ? 0 : blk_first, live1 = blk_count < 0 ? nblk : blk_first + blk_count;
    float v[DVO_WIDE_BATCHES][8];
#pragma unroll
    for (int bi = 0; bi < DVO_WIDE_BATCHES; bi++) {
        // (wave-uniform: batches past the last row issue no loads -- a wave holds at most 63 loads in flight, so the 80 of a full
        //  table are two memory round trips and the 24 of a 75-row level one: 4.0 -> 2 us of every single-stream iteration)
        if (32 * bi < nblk) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int b = 32 * bi + g + DVO_SOLVE_GROUPS * j;
                const bool live = (b >= live0) & (b < live1);
                const float x = p[(size_t)(live ? b : 0) * 32];
                v[bi][j] = live ? x : 0.0f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[bi][j] = 0.0f;
        }
    }
    double sg = 0.0;
#pragma unroll
    for (int bi = 0; bi < DVO_WIDE_BATCHES; bi++)
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (32 * bi < nblk && 32 * bi + g < nblk) sg += (double)v[bi][j];   // (the same zeros in the same slots as sum_partial_rows)
    return sg;
}

__global__ void __launch_bounds__(32 * DVO_SOLVE_SEQ) k_gn_solve(SolveArgs a)
{
    __shared__ double tot[DVO_SOLVE_SEQ][32];
    __shared__ double part[2][DVO_SOLVE_GROUPS][32];
    const int n_in = a.list_in ? a.list_in[0] : a.n_seq;  // sequences this launch handles
    // progress word in mapped host memory (adaptive schedule, Tracker::track): "iteration reached, n sequences were active".
    // Kept out of k_track_gn on purpose: that kernel sits exactly at its 72-VGPR budget and one more live value makes it spill.
    // Fine-grained host memory + a system-scope atomic store (sc0 sc1: written through, never parked in L2).  Relaxed on purpose:
    // the word is a hint for the host's launch schedule, no data is read on its strength, and a system-scope RELEASE here would
    // write back every dirty L2 line (the partial rows k_track_gn has just produced) once per launch.
    if (a.progress && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.progress, n_in + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)blockIdx.x * DVO_SOLVE_SEQ >= n_in) return;
    // Everything this kernel needs from memory is requested up front (a fresh kernel starts with cold caches: each
    // dependent round trip costs ~2 us): lanes 0..7 the state of their sequence, every team its partial rows.
    const int my_slot = (int)blockIdx.x * DVO_SOLVE_SEQ + (int)threadIdx.x;   // (serial stage: threads 0..7)
    const bool serial = threadIdx.x < DVO_SOLVE_SEQ && my_slot < n_in;
    int my_seq = 0;
    if (serial) my_seq = a.list_in ? a.list_in[4 + my_slot] : my_slot;
    SeqState& st = a.state[my_seq];
    int was_active = 0, it_prev = 0;
    float xi[6] = {0, 0, 0, 0, 0, 0};
    double Tc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (serial) {
        was_active = st.active; it_prev = st.iter;
#pragma unroll
        for (int i = 0; i < 6; i++) xi[i] = st.xi[i];
#pragma unroll
        for (int i = 0; i < 12; i++) Tc[i] = st.Tc[i];
    }
    // stage one: second reduction of the workgroup partials, fixed order (bit-reproducible; k_track_level mirrors it)
    const int c = threadIdx.x & 31, team = threadIdx.x >> 5;
    const int t_slot = (int)blockIdx.x * DVO_SOLVE_SEQ + team;
    if (n_in <= 2 && a.nblk <= 32 * DVO_WIDE_BATCHES) {   // (workgroup-uniform) one or two sequences: a team per row class, one round trip
        const int ws = team >> 2, wg = team & 3;
        if (ws < n_in && c < 29) {
            const int t_seq = a.list_in ? a.list_in[4 + ws] : ws;
            part[ws][wg][c] = sum_partial_class(a.partials + (size_t)t_seq * a.nblk * 32 + c, a.nblk, a.blk_first, a.blk_count, wg);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int ts = threadIdx.x >> 5;
            tot[ts][c] = (ts < n_in && c < 29) ? (part[ts][0][c] + part[ts][1][c]) + (part[ts][2][c] + part[ts][3][c]) : 0.0;
        }
    } else if (t_slot < n_in && c < 29) {
        const int t_seq = a.list_in ? a.list_in[4 + t_slot] : t_slot;
        tot[team][c] = sum_partial_rows(a.partials + (size_t)t_seq * a.nblk * 32 + c, a.nblk, a.blk_first, a.blk_count);
    } else if (c >= 29) {
        tot[team][c] = 0.0;
    }
    __syncthreads();
    if (!serial) return;
    if (!a.ignore_active && was_active == 0) return;  // converged sequence: nothing to do
    Pose np;
    (void)solve_finish(a, my_seq, st, tot[threadIdx.x], a.ignore_active, it_prev, xi, Tc, np);
}

// The serial part of one Tracker::track iteration (tracker.cpp:44-73) for one sequence, run by ONE thread: 6x6 solve,
// pose update, stop tests, log.  `tot` = the 29 sums in double; xi / Tc = the sequence's twist and exp(+xi) on entry,
// updated on return (and written to `st`); np = exp(-xi) after the update.  Returns the new active flag.
__device__ __forceinline__ int solve_finish(const SolveArgs& a, const int seq, SeqState& st, const double* tot, const int first,
                                            const int it_prev, float xi[6], double Tc[12], Pose& np)
{
    const int n_valid = (int)tot[28];
    const double sum_r2 = tot[27];
    float upd[6] = {0, 0, 0, 0, 0, 0};
    float residual = -1.0f;  // optimize.cpp:92-93
    if (n_valid > 0) {
        solve6(tot, tot + 21, upd);
        residual = (float)sum_r2 / (float)n_valid;  // optimize.cpp:98
    }
    if (a.dbg_stamp) a.dbg_stamp[0] = wall_clock64();
    // xi <- log(exp(xi) exp(upd)) unless NaN (tracker.cpp:46-51); pose <- exp(-xi) for Stuff::update (optimize.hpp:26-30)
    const bool updated = se3_update_pose(Tc, upd, xi, np);
    if (a.dbg_stamp) a.dbg_stamp[1] = wall_clock64();
    if (updated) {
#pragma unroll
        for (int i = 0; i < 6; i++) st.xi[i] = xi[i];
#pragma unroll
        for (int i = 0; i < 12; i++) st.Tc[i] = Tc[i];
        st.pose = np;
    }
    double nrm = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) nrm += (double)upd[i] * (double)upd[i];
    nrm = sqrt(nrm);

    const int it = first ? 0 : it_prev;
    if (a.log && it < DVO_MAX_ITERATIONS) {
        dvo_track_log& lg = a.log[seq];
        lg.n_iter[a.level] = it + 1;
        lg.residual[a.level][it] = residual;
        lg.update_norm[a.level][it] = (float)nrm;
        lg.n_valid[a.level][it] = n_valid;
#pragma unroll
        for (int i = 0; i < 6; i++) lg.xi_after[a.level][it][i] = xi[i];
#pragma unroll
        for (int i = 0; i < 6; i++) lg.xi_update[a.level][it][i] = upd[i];
    }
    if (a.result) {  // operator-level output (dvo_op_gn_step)
        dvo_gn_result& r = a.result[seq];
        for (int i = 0; i < 21; i++) r.H[i] = tot[i];
        for (int i = 0; i < 6; i++) { r.g[i] = tot[21 + i]; r.xi_update[i] = upd[i]; r.xi_next[i] = xi[i]; }
        r.sum_r2 = sum_r2;
        r.n_valid = n_valid;
        r.residual = residual;
    }
    st.iter = it + 1;
    int active = 1;
    if (a.fixed_iterations > 0) {
        active = (it + 1 < a.fixed_iterations) ? 1 : 0;
    } else if (nrm < (double)a.min_update || residual < a.min_residual || it + 1 >= a.max_iterations) {
        active = 0;  // tracker.cpp:68-73 (the wall-clock term is disabled, D1)
    }
    st.active = active;
    if (active && a.list_out) a.list_out[4 + atomicAdd(&a.list_out[0], 1)] = seq;
    if (a.counters) {  // profile: evaluated pixels / sequence-iterations
        atomicAdd(&a.counters[0], (unsigned long long)a.level_pixels);
        atomicAdd(&a.counters[1], 1ull);
    }
    return active | (updated ? 2 : 0);
}

// ------------------------------------------------------------------------------------------------
// k_track_gn_fused: one Tracker::track iteration in ONE launch, for a handful of sequences (a dvo_vo handle: one).  Every
// workgroup evaluates its tile exactly as k_track_gn does; the workgroup that finishes a sequence's LAST tile (an arrival ticket
// per sequence) then runs what k_gn_solve would: the second reduction stage in the same fixed order and solve_finish().  With one
// sequence the two-kernel form is nothing but latency -- two launches, two kernel boundaries and a cold-cache solve kernel per
// iteration; here it is one launch and the solve starts the moment the last partial row lands.  The price is the solve's register
// footprint (248 VGPRs, two waves per SIMD) for the whole kernel, which is irrelevant for the <= 1024 workgroups this path is used
// for and ruinous for a batch (measured in round 1): the host picks it per level (Tracker::track).
// Hand-off: partial rows are plain stores; each workgroup drains them (vmcnt(0)), barrier, lane 0 agent-scope release fence, then the
// ticket (agent-scope atomic add); the last arriver takes an agent-scope acquire fence (invalidates this CU's L1) before anyone
// in its workgroup loads the rows.  No workgroup ever waits for another one: nothing can hang.
// ------------------------------------------------------------------------------------------------
struct FusedArgs {
    int* ticket;        // [n_seq] arrival counters, zero between launches
    int* report;        // 2 ints of THIS (level, iteration): [0] sequences reported, [1] sequences still active afterwards
    int* progress;      // optional, fine-grained HOST memory: last reporter stores (active afterwards + 1)
    int n_seq;
};

template <int PPT, int G, bool T2D>
__global__ void __launch_bounds__(256) k_track_gn_fused(GnArgs a, SolveArgs sa, FusedArgs f)
{
    __shared__ GnTileLds<PPT> lds;
    __shared__ double tot[32];
    __shared__ double part[DVO_SOLVE_GROUPS][32];
    __shared__ int last_s;
    const int tile_id = (int)blockIdx.x;                       // grid = n_seq * blk_count exactly
    const int seq = tile_id / a.blk_count, blk = a.blk_first + (tile_id - seq * a.blk_count);
    SeqState& st = sa.state[seq];
    auto report = [&](int active) {                              // one thread per sequence and launch
        if (active) __hip_atomic_fetch_add(&f.report[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int r = __hip_atomic_fetch_add(&f.report[0], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (r == f.n_seq - 1 && f.progress) {
            const int act = __hip_atomic_load(&f.report[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(f.progress, act + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    if (!sa.ignore_active && st.active == 0) {                   // converged sequence: its first tile reports, nobody works
        if (blk == a.blk_first && threadIdx.x == 0) report(0);
        return;
    }
    const Pose pose = st.pose;                                   // wave-uniform -> scalar loads
    gn_tile<PPT, G, false, T2D>(a, pose, seq, blk, lds, a.partials + ((size_t)seq * a.nblk + blk) * 32);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's row stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(&f.ticket[seq], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == a.blk_count - 1) ? 1 : 0;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&f.ticket[seq], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        last_s = last;
    }
    __syncthreads();
    if (!last_s) return;
    if (a.nblk <= 32 * DVO_WIDE_BATCHES) {                        // a team per row class: every partial row requested at once
        const int c = threadIdx.x & 31, wg = (int)(threadIdx.x >> 5);
        if (wg < DVO_SOLVE_GROUPS && c < 29)
            part[wg][c] = sum_partial_class(a.partials + (size_t)seq * a.nblk * 32 + c, a.nblk, sa.blk_first, sa.blk_count, wg);
        __syncthreads();
        if (threadIdx.x < 32) tot[c] = c < 29 ? (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]) : 0.0;
    } else if (threadIdx.x < 32) {
        const int c = threadIdx.x;
        tot[c] = c < 29 ? sum_partial_rows(a.partials + (size_t)seq * a.nblk * 32 + c, a.nblk, sa.blk_first, sa.blk_count) : 0.0;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float xi[6];
    double Tc[12];
#pragma unroll
    for (int i = 0; i < 6; i++) xi[i] = st.xi[i];
#pragma unroll
    for (int i = 0; i < 12; i++) Tc[i] = st.Tc[i];
    Pose np;
    const int r = solve_finish(sa, seq, st, tot, sa.ignore_active, st.iter, xi, Tc, np);
    report(r & 1);
}

// ------------------------------------------------------------------------------------------------
// k_track_persist: ALL of Tracker::track (tracker.cpp:22-85: every level, every iteration) for ONE sequence in ONE launch -- what a
// dvo_vo handle runs per frame.  A fixed grid of co-resident workgroups strides over the tiles of the current level (gn_tile, the
// same code and tile size as k_track_gn: the same partial rows), every workgroup then takes an arrival ticket, and the LAST arriver
// does what k_gn_solve does (second reduction stage in the fixed order, solve_finish: 6x6 solve, pose composition, stop tests,
// log), decides what comes next (same level / next level / done) and publishes it with an epoch word; the other workgroups poll
// that word (s_sleep between polls) and go on.  Per iteration that replaces a kernel boundary (~4 us of launch floor + cold caches
// per launch, profiles/r03_single_*_trace.txt) by one ticket and one epoch hand-over.  Bit-identical to the launch-per-iteration
// schedules: same tiles, same summation order, same solve_finish().
// Safety: the grid is sized by the occupancy query (every workgroup resident), and EVERY wait is bounded -- a workgroup that polls
// `spin_limit` times without seeing its epoch marks the launch as given up and leaves; the others then run into the same limit.
// The host sees the mark instead of the result tag and re-runs the frame with the launch-per-iteration schedule.
// ------------------------------------------------------------------------------------------------
template <int PPT, int G, bool MONO>   // MONO: with the mono handle's keyframe decision as the tail (its own instance: the tail's two
__global__ void __launch_bounds__(256) k_track_persist(PersistArgs p)   // exponentials and logarithm otherwise weigh on the sensor-depth handle's solver)
{
    __shared__ GnTileLds<PPT> lds;
    __shared__ double tot[32];
    __shared__ double part[DVO_SOLVE_GROUPS][32];
    __shared__ int line_s[16];   // the control line as this workgroup last read / is about to write it
    __shared__ float mono_fx_s[6];   // (MONO) the frame's pose and keyframe flag, from the solver's thread to the age-table tail
    __shared__ int mono_need_s;
    __shared__ int gave_up_s;
    // Workgroup 0 only solves; workgroups 1 .. grid-1 evaluate tiles.  Nobody takes a contended atomic: a worker announces its tiles of
    // step s by storing (epoch0 + s + 1) into its OWN slot of `arrive`, the solver polls the slots (one lane each); the solver
    // publishes step s's outcome -- next level, status, and the new pose itself -- as ONE 64-byte line whose first word is
    // (epoch0 + s + 1), which the workers poll.  The sequence's serial state (xi, exp(xi), iteration count) never leaves the
    // solver's registers.
    int* const ctl = p.ctl;                 // [0] epoch, [1] next level, [2] status (0 go on, 1 done), [3..14] pose, [15] epoch again
    int* const arrive = p.ctl + 16;         // [grid]
    const int epoch0 = p.host_tag << 10;    // epoch numbers of this launch: never equal to a value an earlier launch left behind
    const int n_work = (int)gridDim.x - 1;
    int level = 0, step = 0;
    if (blockIdx.x != 0) {
        // ---------------------------------------------------------------------------------------------- tile workers
        const int me = (int)blockIdx.x - 1;
        for (;;) {
            if (threadIdx.x < 16) {   // the control line of this step (the first step is the identity pose on level 0: tracker.cpp:28)
                int v = 0;
                if (step == 0) {
                    v = (threadIdx.x == 3 || threadIdx.x == 7 || threadIdx.x == 11) ? __float_as_int(1.0f) : 0;   // (pose words 3..14: R row major, t)
                } else {
                    int polls = 0;
                    for (;;) {
                        v = __hip_atomic_load(&ctl[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // The line is complete when its FIRST word (the epoch, stored last, after a release) and its LAST word (a copy
                        // of the epoch, stored with the payload) both carry this step's number: whatever order the 64 bytes were
                        // fetched in, a current first word means every store before the release had landed when it was read, and a
                        // current last word means the payload store itself had landed when the other end was read.
                        const int e0 = __builtin_amdgcn_readlane(v, 0), e15 = __builtin_amdgcn_readlane(v, 15);
                        if (e0 == epoch0 + step && e15 == epoch0 + step) break;
                        if (++polls > p.spin_limit) { v = (threadIdx.x == 2) ? 2 : v; break; }   // bounded: give up (status 2)
                    }
                }
                line_s[threadIdx.x] = v;
            }
            __syncthreads();
            const bool stamp = p.dbg && me == p.dbg_worker && threadIdx.x == 0 && step < 64;
            if (stamp) p.dbg[(64 + step) * 8 + 0] = wall_clock64();   // line seen
            const int status = line_s[2];
            if (status != 0) {
                if (status == 2 && threadIdx.x == 0 && p.host_result)
                    __hip_atomic_store(reinterpret_cast<int*>(p.host_result + 23), p.host_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            level = line_s[1];
            Pose pose;
#pragma unroll
            for (int i = 0; i < 9; i++) pose.R[i] = __int_as_float(__builtin_amdgcn_readfirstlane(line_s[3 + i]));   // wave-uniform: SGPRs
#pragma unroll
            for (int i = 0; i < 3; i++) pose.t[i] = __int_as_float(__builtin_amdgcn_readfirstlane(line_s[12 + i]));
            const PersistLevel& L = p.lv[level];
            GnArgs a;
            a.obj_gray = L.obj_gray; a.ref_gray = L.ref_gray; a.ref_depth = L.ref_depth; a.ref_wgt = L.ref_wgt; a.wgt_const = L.wgt_const;
            a.state = p.state; a.partials = p.partials; a.mask = nullptr;
            a.w = L.w; a.h = L.h; a.nblk = L.nblk; a.inv_w = L.inv_w; a.q256 = L.q256; a.r256 = L.r256; a.k = L.k; a.prm = L.prm;
            a.ignore_active = 1; a.list = nullptr; a.next_count = nullptr; a.n_seq = 1;
            a.blk_first = L.blk_first; a.blk_count = L.blk_count; a.t_shift = L.t_shift; a.x_org = L.x_org; a.y_org = L.y_org;
            a.tiles_x = L.tiles_x; a.tiles_y = 0; a.margin = 0;
            if (stamp) p.dbg[(64 + step) * 8 + 3] = wall_clock64();   // arguments and pose in registers
            for (int t = me; t < L.blk_count; t += n_work) {
                float* row = p.partials + (size_t)(L.blk_first + t) * 32;
                if constexpr (PPT == 4) {
                    if (L.t2d) gn_tile<PPT, G, false, true>(a, pose, 0, L.blk_first + t, lds, row);
                    else gn_tile<PPT, G, false, false>(a, pose, 0, L.blk_first + t, lds, row);
                } else {
                    gn_tile<PPT, G, false, false>(a, pose, 0, L.blk_first + t, lds, row);
                }
                __syncthreads();   // (the next tile reuses the LDS scratch)
            }
            if (stamp) p.dbg[(64 + step) * 8 + 4] = wall_clock64();   // gn_tile returned
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's row stores have left
            __syncthreads();
            if (stamp) p.dbg[(64 + step) * 8 + 1] = wall_clock64();   // tiles done
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // the rows, before the announcement
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&arrive[me], epoch0 + step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (stamp) p.dbg[(64 + step) * 8 + 2] = wall_clock64();   // announced
            step++;
            __syncthreads();   // (line_s is rewritten in the next round)
        }
        return;
    }
    // -------------------------------------------------------------------------------------------------- the solver (workgroup 0)
    SeqState& st = p.state[0];
    float xi[6] = {0, 0, 0, 0, 0, 0};                       // thread 0's: the sequence's serial state (tracker.cpp:28: xi = 0)
    double Tc[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    int it_prev = 0, first = 1;
    if (threadIdx.x == 0 && p.log) {
        p.log[0].levels = p.levels;
        for (int l = 0; l < DVO_MAX_LEVELS; l++) p.log[0].n_iter[l] = 0;
    }
    for (;;) {
        const PersistLevel& L = p.lv[level];
        // every worker has announced step `step` (bounded polls: one lane per worker slot)
        if (threadIdx.x == 0) gave_up_s = 0;
        __syncthreads();
        const bool stamp = p.dbg && threadIdx.x == 0 && step < 64;
        if (stamp) p.dbg[step * 8 + 0] = wall_clock64();   // starts waiting
        for (int wk = (int)threadIdx.x; wk < n_work; wk += 256) {
            int polls = 0;
            while (__hip_atomic_load(&arrive[wk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch0 + step + 1) {
                if (++polls > p.spin_limit) { gave_up_s = 1; break; }
            }
        }
        if (stamp) p.dbg[step * 8 + 1] = wall_clock64();   // own slots seen
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the workers' rows, after their announcements
        __syncthreads();
        if (stamp) p.dbg[step * 8 + 2] = wall_clock64();   // everybody arrived, fence done
        int status = 0, nl = level;
        if (gave_up_s) {
            status = 2;
        } else {
            const int c = threadIdx.x & 31, wg = (int)(threadIdx.x >> 5);
            if (wg < DVO_SOLVE_GROUPS && c < 29) part[wg][c] = sum_partial_class(p.partials + c, L.nblk, L.blk_first, L.blk_count, wg);
            __syncthreads();
            if (threadIdx.x < 32) tot[c] = c < 29 ? (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]) : 0.0;
            __syncthreads();
        }
        if (stamp) p.dbg[step * 8 + 3] = wall_clock64();   // rows summed
        if (threadIdx.x == 0) {
            Pose np;
            if (status == 0) {
                SolveArgs sa;
                sa.state = p.state; sa.partials = p.partials; sa.log = p.log; sa.result = nullptr; sa.counters = nullptr;
                sa.nblk = L.nblk; sa.level = level; sa.level_pixels = L.level_pixels;
                sa.max_iterations = p.max_iterations; sa.fixed_iterations = p.fixed_iterations;
                sa.min_update = p.min_update; sa.min_residual = p.min_residual;
                sa.ignore_active = first; sa.list_in = nullptr; sa.list_out = nullptr; sa.progress = nullptr; sa.n_seq = 1;
                sa.blk_first = L.blk_first; sa.blk_count = L.blk_count;
                if (stamp) sa.dbg_stamp = &p.dbg[(64 + step) * 8 + 5];   // (two free slots of worker 0's row)
                const int r = solve_finish(sa, 0, st, tot, first, it_prev, xi, Tc, np);
                it_prev = first ? 1 : it_prev + 1;       // (= st.iter)
                if (!(r & 2)) pose_from_xi(xi, -1.0f, np);   // update rejected (NaN, tracker.cpp:47-51): the pose stays exp(-xi) of the unchanged xi
                if (!(r & 1)) {            // the level stopped (tracker.cpp:68-73): next level, or done
                    nl = level + 1;
                    if (nl >= p.levels) {  // k_export_poses' work: twist + exp(xi) (system.hpp:92), also into the mapped host block
                        status = 1;
                        float T[16];
                        for (int i = 0; i < 6; i++) { st.xi[i] = xi[i]; p.xi_out[i] = xi[i]; }
                        se3_exp_f(xi, T);
                        for (int i = 0; i < 16; i++) p.T_out[i] = T[i];
                        float fxw[6], Tw[16];
                        int need = 0;
                        if (MONO && p.mono.enabled) {   // a mono handle: Frame::updateXi, needNewFrame and exp(xi) here, not in a launch of their own
                            MonoSeq& m = *p.mono.meta;
                            for (int i = 0; i < 6; i++) m.ref_xi[i] = p.mono.ref_xi[i];
                            m.ref_id = p.mono.ref_id; m.n_total = p.mono.n_total;
                            need = mono_decide_one(m, xi, p.mono.frame_id, p.mono.min_translation, p.mono.max_frames, fxw, Tw);
                            for (int i = 0; i < 6; i++) mono_fx_s[i] = fxw[i];
                            mono_need_s = need;
                        }
                        if (p.host_result) {
                            for (int i = 0; i < 6; i++) p.host_result[i] = xi[i];
                            for (int i = 0; i < 16; i++) p.host_result[6 + i] = T[i];
                            if (MONO && p.mono.enabled) {
                                for (int i = 0; i < 6; i++) p.host_result[24 + i] = fxw[i];
                                for (int i = 0; i < 16; i++) p.host_result[30 + i] = Tw[i];
                                reinterpret_cast<int*>(p.host_result)[46] = need;
                            }
                            __hip_atomic_store(reinterpret_cast<int*>(p.host_result + 22), p.host_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                }
            } else if (p.host_result) {
                __hip_atomic_store(reinterpret_cast<int*>(p.host_result + 23), p.host_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (stamp) { p.dbg[step * 8 + 4] = wall_clock64(); p.dbg[step * 8 + 6] = level; }   // solved
            line_s[0] = epoch0 + step + 1; line_s[1] = nl; line_s[2] = status; line_s[15] = epoch0 + step + 1;
            for (int i = 0; i < 9; i++) line_s[3 + i] = __float_as_int(np.R[i]);
            for (int i = 0; i < 3; i++) line_s[12 + i] = __float_as_int(np.t[i]);
        }
        __syncthreads();
        // publish: words 1..15 first, then (release) the epoch word the workers poll
        if (threadIdx.x >= 1 && threadIdx.x < 16) __hip_atomic_store(&ctl[threadIdx.x], line_s[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) __hip_atomic_store(&ctl[0], line_s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (stamp) p.dbg[step * 8 + 5] = wall_clock64();   // published
        const int st_now = line_s[2], nl_now = line_s[1];
        __syncthreads();
        if (st_now != 0) {
            if constexpr (MONO) {
                // The frame is tracked and (thread 0, above) decided.  Not a keyframe: Mapper::update follows, and its per-keyframe
                // relative poses are this workgroup's last job -- one thread per keyframe of FrameHistory -- instead of a launch of their own.
                if (st_now == 1 && p.mono.enabled && p.mono.hist_xi && mono_need_s == 0) {
                    float fxi[6];
                    for (int k = 0; k < 6; k++) fxi[k] = mono_fx_s[k];   // (written by thread 0 before the barriers above)
                    for (int i = (int)threadIdx.x; i < p.mono.n_hist; i += 256) {
                        AgeEntry e;
                        age_entry_one(fxi, p.mono.hist_xi + (size_t)i * 6, i, e);
                        p.mono.ages[i] = e;
                    }
                    if (threadIdx.x == 0 && p.mono.zero_word) *p.mono.zero_word = 0;
                }
            }
            break;
        }
        first = (nl_now != level) ? 1 : 0;
        level = nl_now;
        step++;
    }
}

// ------------------------------------------------------------------------------------------------
// k_track_level: ALL iterations of one (coarse) pyramid level for one sequence in one launch -- the loop of
// tracker.cpp:42-74 on the device.  One workgroup per sequence: every iteration evaluates the level's live tiles with
// gn_tile() into LDS rows, sums them exactly as k_gn_solve does (same grouping, same order: bit-identical to the
// k_track_gn + k_gn_solve pair at the same PPT), and thread 0 runs solve_finish(); the workgroup leaves when its
// sequence stops.  A dependent kernel boundary costs ~8 us on this GPU, whatever the kernel does: for levels of a few
// tiles the 2 x max_iterations launches of the unfused schedule were all boundary; here there is one.
// ------------------------------------------------------------------------------------------------
template <int PPT, int G>
__global__ void __launch_bounds__(256) k_track_level(GnArgs ga, SolveArgs sa)
{
    __shared__ GnTileLds<PPT> lds;
    __shared__ float rows[DVO_FUSED_MAX_TILES][32];
    __shared__ double part[8][32];
    __shared__ double tot[32];
    __shared__ float pose_s[12];
    __shared__ int flag_s;
    const int seq = blockIdx.x;
    SeqState& st = sa.state[seq];
    // thread 0 owns the serial state of the sequence across iterations
    float xi[6];
    double Tc[12];
    int it_prev = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) xi[i] = st.xi[i];
#pragma unroll
        for (int i = 0; i < 12; i++) Tc[i] = st.Tc[i];
#pragma unroll
        for (int i = 0; i < 9; i++) pose_s[i] = st.pose.R[i];
#pragma unroll
        for (int i = 0; i < 3; i++) pose_s[9 + i] = st.pose.t[i];
    }
    __syncthreads();
    const int max_it = sa.fixed_iterations > 0 ? sa.fixed_iterations : sa.max_iterations;
    const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int live0 = ga.blk_first, live1 = ga.blk_first + ga.blk_count;
    for (int it = 0; it < max_it; it++) {
        Pose pose;  // wave-uniform: broadcast from LDS into SGPRs
#pragma unroll
        for (int i = 0; i < 9; i++) pose.R[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pose_s[i])));
#pragma unroll
        for (int i = 0; i < 3; i++) pose.t[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pose_s[9 + i])));
        for (int blk = live0; blk < live1; blk++) gn_tile<PPT, G, false, false>(ga, pose, seq, blk, lds, rows[blk]);
        __syncthreads();
        {  // second reduction stage, as in k_gn_solve (rows outside the live range are exact zeros)
            double s = 0.0;
            if (c < 29 && grp < DVO_SOLVE_GROUPS) {  // the grouping of k_gn_solve (its 128 threads)
                for (int b0 = grp; b0 < ga.nblk; b0 += 8 * DVO_SOLVE_GROUPS) {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int b = b0 + DVO_SOLVE_GROUPS * j;
                        const bool live = (b >= live0) & (b < live1);
                        const float x = rows[live ? b : live0][c];  // (index always inside the array)
                        s += live ? (double)x : 0.0;
                    }
                }
            }
            part[grp][c] = s;
        }
        __syncthreads();
        if (threadIdx.x < 32)
            tot[c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        __syncthreads();
        if (threadIdx.x == 0) {
            Pose np;
            const int r = solve_finish(sa, seq, st, tot, it == 0 ? 1 : 0, it_prev, xi, Tc, np);
            it_prev = it + 1;
            if (r & 2) {
#pragma unroll
                for (int i = 0; i < 9; i++) pose_s[i] = np.R[i];
#pragma unroll
                for (int i = 0; i < 3; i++) pose_s[9 + i] = np.t[i];
            }
            flag_s = r & 1;
        }
        __syncthreads();
        if (flag_s == 0) break;
    }
}

// k_track_begin: Tracker::track line 28 (xi = 0) for every sequence
__global__ void __launch_bounds__(256) k_track_begin(SeqState* state, dvo_track_log* log, int n_seq, int levels)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_seq) return;
    SeqState& st = state[s];
    for (int i = 0; i < 6; i++) st.xi[i] = 0.0f;
    for (int i = 0; i < 9; i++) st.pose.R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    for (int i = 0; i < 3; i++) st.pose.t[i] = 0.0f;
    for (int i = 0; i < 12; i++) st.Tc[i] = (i < 9 && i % 4 == 0) ? 1.0 : 0.0;
    st.active = 1;
    st.iter = 0;
    if (log) {
        log[s].levels = levels;
        for (int l = 0; l < DVO_MAX_LEVELS; l++) log[s].n_iter[l] = 0;
    }
}

// k_set_pose: load a caller-supplied twist (operator-level gn_step / probes)
__global__ void k_set_pose(SeqState* state, const float* xi, int n_seq)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seq) return;
    float x[6];
    double xd[6], R[9], t[3];
    for (int i = 0; i < 6; i++) { x[i] = xi[s * 6 + i]; state[s].xi[i] = x[i]; xd[i] = x[i]; }
    pose_from_xi(x, -1.0f, state[s].pose);
    se3_exp_d(xd, R, t);
    for (int i = 0; i < 9; i++) state[s].Tc[i] = R[i];
    for (int i = 0; i < 3; i++) state[s].Tc[9 + i] = t[i];
    state[s].active = 1;
    state[s].iter = 0;
}

// k_export_poses: relative twist + exp(xi) 4x4 (system.hpp:92) per sequence
// host_result (optional, fine-grained mapped HOST memory, one sequence): the same 22 floats, then a sequence word written with a
// system-scope release store -- the caller's thread polls it instead of queueing a device-to-host copy and waiting for the stream
// (a dvo_vo handle returns one pose per call: three small copies + a stream synchronisation were ~55 us of every frame).
__global__ void k_export_poses(const SeqState* state, float* xi_out, float* T_out, int n_seq, float* host_result, int host_tag)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seq) return;
    float x[6], T[16];
    for (int i = 0; i < 6; i++) { x[i] = state[s].xi[i]; xi_out[s * 6 + i] = x[i]; }
    se3_exp_f(x, T);
    for (int i = 0; i < 16; i++) T_out[s * 16 + i] = T[i];
    if (host_result && s == 0) {
        for (int i = 0; i < 6; i++) host_result[i] = x[i];
        for (int i = 0; i < 16; i++) host_result[6 + i] = T[i];
        __hip_atomic_store(reinterpret_cast<int*>(host_result + 22), host_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// k_se3: device evaluation of the double-precision pose algebra (parity op): op 0 exp, 1 log, 2 concatenate
__global__ void k_se3(int op, const float* in_a, const float* in_b, float* out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (op == 0) {
        float xi[6], T[16];
        for (int i = 0; i < 6; i++) xi[i] = in_a[i];
        se3_exp_f(xi, T);
        for (int i = 0; i < 16; i++) out[i] = T[i];
    } else if (op == 1) {
        float T[16], xi[6];
        for (int i = 0; i < 16; i++) T[i] = in_a[i];
        se3_log_f(T, xi);
        for (int i = 0; i < 6; i++) out[i] = xi[i];
    } else {
        float x[6], y[6], o[6];
        for (int i = 0; i < 6; i++) { x[i] = in_a[i]; y[i] = in_b[i]; }
        se3_concatenate_f(x, y, o);
        for (int i = 0; i < 6; i++) out[i] = o[i];
    }
}

// (the mapping kernels -- propagate, regularize, depth update, keyframe promotion -- live in dvo_map_kernels.hip)

// ------------------------------------------------------------------------------------------------
// Dataset front-end kernels (what src/core/loader.cpp does on the CPU with OpenCV)
// ------------------------------------------------------------------------------------------------
// k_ingest: raw sensor frames -> the float maps of the ABI, on the device (uploads 1.5 MB/frame instead of 3.7 MB).
//   gray  = BGR2GRAY(u8) / 255 (loader.cpp:55-60; OpenCV's fixed-point luma R 4899, G 9617, B 1868, >> 14), PNG channel
//           order is R,G,B[,A]; 1-channel input is taken as gray.
//   depth = u16 * depth_scale (loader.cpp:145: 1/5000), sigma = sigma_valid where depth > 0 else sigma_invalid and,
//   with invalidate_gray, gray = INVALID where depth == 0 -- what Transform::mapDepthtoGray leaves behind
//   (src/core/transform.cpp:60-76: pixels without depth stay INVALID with sigma 1, the others get sigma 0.1).
__global__ void __launch_bounds__(256) k_ingest(const uint8_t* __restrict__ rgb, int channels, const uint16_t* __restrict__ depth16,
                                                int n, float gray_scale, float depth_scale, float sigma_valid, float sigma_invalid,
                                                int invalidate_gray, float* __restrict__ gray, float* __restrict__ depth, float* __restrict__ sigma)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned g8;
    if (channels == 1) {
        g8 = rgb[i];
    } else {
        const uint8_t* p = rgb + (size_t)i * channels;
        g8 = ((unsigned)p[0] * 4899u + (unsigned)p[1] * 9617u + (unsigned)p[2] * 1868u + 8192u) >> 14;
    }
    float gv = (float)g8 * gray_scale;
    if (depth16) {
        const unsigned d = depth16[i];
        depth[i] = (float)d * depth_scale;
        sigma[i] = d > 0 ? sigma_valid : sigma_invalid;
        if (invalidate_gray && d == 0) gv = kInvalid;
    }
    gray[i] = gv;
}

// k_undistort: Loader::getNormalizedUndistortedImages (loader.cpp:15-42): cv::initUndistortRectifyMap(K, D, I, K) +
// cv::remap(INTER_NEAREST, BORDER_CONSTANT = INVALID), restated from the radial-tangential model definition.
__global__ void __launch_bounds__(256) k_undistort(const float* __restrict__ src, int w, int h, Intr k, float k1, float k2, float p1,
                                                   float p2, float k3, float border, float* __restrict__ dst)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int v = i / w, u = i - v * w;
    const double x = ((double)u - k.cx) / k.fx, y = ((double)v - k.cy) / k.fy;
    const double r2 = x * x + y * y, radial = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
    const double xd = x * radial + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
    const double yd = y * radial + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
    const float mx = (float)(xd * k.fx + k.cx), my = (float)(yd * k.fy + k.cy);
    int sx, sy;
    float out = border;
    if (round_coord(mx, sx) && round_coord(my, sy) && sx >= 0 && sx < w && sy >= 0 && sy < h) out = src[sy * w + sx];
    dst[i] = out;
}

// ------------------------------------------------------------------------------------------------
// k_visualize: the false-colour views of src/core/draw.cpp:7-100 as plain RGB bytes (no GUI).
//   mode 0 gray      (Draw::visualizeGray: value*255, INVALID pixels blue)
//   mode 1 depth     (Draw::visualizeDepth(depth, sigma): hue from depth, value from sigma; b may be null -> full value)
//   mode 2 sigma     (Draw::visualizeSigma: 255 - 500 sigma)
//   mode 3 age       (Draw::visualizeAge: 10 * age)
//   mode 4 gradient  (Draw::visualizeGradient: green positive, red negative, INVALID blue)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned char sat_u8(float v)
{
    v = rintf(v);
    return (unsigned char)(v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v));
}

__global__ void __launch_bounds__(256) k_visualize(int mode, const float* __restrict__ a, const float* __restrict__ b, int n, uint8_t* __restrict__ rgb)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = a[i];
    unsigned char R = 0, G = 0, B = 0;
    if (mode == 0) {
        R = G = B = sat_u8(v * 255.0f);
        if (!is_valid(v)) B = 255;
    } else if (mode == 1) {
        if (!(v < kEpsilon)) {
            const float hue = fminf(fmaxf((v - 0.70f) * 70.0f, 0.0f), 180.0f);  // OpenCV 8-bit hue: degrees / 2
            const float val = b ? (float)(unsigned char)(-500.0f * fminf(b[i], 0.5f) + 255.0f) / 255.0f : 1.0f;
            const float hh = (float)(unsigned char)hue * 2.0f / 60.0f;
            const float c = val, x = c * (1.0f - fabsf(fmodf(hh, 2.0f) - 1.0f));
            float r = 0, g = 0, bl = 0;
            if (hh < 1) { r = c; g = x; } else if (hh < 2) { r = x; g = c; } else if (hh < 3) { g = c; bl = x; }
            else if (hh < 4) { g = x; bl = c; } else if (hh < 5) { r = x; bl = c; } else { r = c; bl = x; }
            R = sat_u8(r * 255.0f); G = sat_u8(g * 255.0f); B = sat_u8(bl * 255.0f);
        }
    } else if (mode == 2) {
        R = G = B = sat_u8(v * -500.0f + 255.0f);
    } else if (mode == 3) {
        R = G = B = sat_u8(v * 10.0f);
    } else {
        if (is_invalid(v)) B = 255;
        else { G = (unsigned char)(255.0f * fminf(fmaxf(v, 0.0f), 1.0f)); R = (unsigned char)(255.0f * fminf(fmaxf(-v, 0.0f), 1.0f)); }
    }
    rgb[3 * (size_t)i] = R; rgb[3 * (size_t)i + 1] = G; rgb[3 * (size_t)i + 2] = B;
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (host)
// ------------------------------------------------------------------------------------------------
// k_selftest_reciprocal: recip_rn() (dvo_math.h) against the IEEE division for ALL 2^32 float bit patterns.
// out[0] = patterns that took the rcp + 2 FMA branch, out[1] = mismatching patterns (must be 0), out[2] = smallest bad pattern.
__global__ void __launch_bounds__(256) k_selftest_reciprocal(unsigned long long* out)
{
    unsigned long long fast = 0, bad = 0, first = ~0ull;
    for (unsigned long long p = blockIdx.x * 256ull + threadIdx.x; p < (1ull << 32); p += (unsigned long long)gridDim.x * 256ull) {
        const float z = __uint_as_float((unsigned)p);
        const float az = fabsf(z);
        const float want = 1.0f / z, got = recip_rn(z);
        const bool same = (__float_as_uint(got) == __float_as_uint(want)) || (got != got && want != want);
        if (az >= DVO_RECIP_FAST_MIN && az <= DVO_RECIP_FAST_MAX) fast++;
        if (!same) { bad++; if (p < first) first = p; }
    }
    atomicAdd(&out[0], fast);
    atomicAdd(&out[1], bad);
    atomicMin(&out[2], first);
}

// k_selftest_sqrt: sqrt_fast() (dvo_math.h; the regularize kernels' square root) against sqrtf for every float in [2^-100, 2^100].
// k_selftest_division: div_by_recip(a, b, recip_fast(b)) against a / b for b = 1.mb, mb = b_first + i * b_stride (i < b_count), and ALL
// 2^23 mantissas of a in [1, 2) -- the operations after v_rcp_f32 (whose every input recip_fast() was enumerated on:
// k_selftest_reciprocal) are IEEE multiplies and FMAs, scale invariant inside the normal range, so mantissa pairs are all there is to check.
// out = {operands checked, mismatches, first bad (sqrt: bit pattern; division: mb << 23 | ma)}, pre-set {0, 0, ~0}.
__global__ void __launch_bounds__(256) k_selftest_sqrt(unsigned long long* out)
{
    unsigned long long n = 0, bad = 0, first = ~0ull;
    for (unsigned long long p = blockIdx.x * 256ull + threadIdx.x; p < (1ull << 31); p += (unsigned long long)gridDim.x * 256ull) {
        const float x = __uint_as_float((unsigned)p);
        if (!(x >= DVO_RECIP_FAST_MIN && x <= DVO_RECIP_FAST_MAX)) continue;
        n++;
        if (__float_as_uint(sqrt_fast(x)) != __float_as_uint(sqrtf(x))) { bad++; if (p < first) first = p; }
    }
    atomicAdd(&out[0], n);
    if (bad) { atomicAdd(&out[1], bad); atomicMin(&out[2], first); }
}

__global__ void __launch_bounds__(256) k_selftest_division(unsigned b_first, unsigned b_stride, unsigned b_count, unsigned long long* out)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= b_count) return;
    const unsigned mb = (b_first + i * b_stride) & 0x7fffffu;
    const float b = __uint_as_float(0x3f800000u | mb);
    const float y = recip_fast(b);
    unsigned long long bad = 0, first = ~0ull;
    for (unsigned ma = 0; ma < (1u << 23); ma++) {
        const float a = __uint_as_float(0x3f800000u | ma);
        if (__float_as_uint(div_by_recip(a, b, y)) != __float_as_uint(a / b)) { bad++; const unsigned long long k = ((unsigned long long)mb << 23) | ma; if (k < first) first = k; }
    }
    atomicAdd(&out[0], 1ull << 23);
    if (bad) { atomicAdd(&out[1], bad); atomicMin(&out[2], first); }
}

// k_selftest_trig: the SE(3) chain's sin / cos / atan2 (dvo_math.h: sincos_dev, atan2_dev) against the math library over their domain:
// sample i of n -> x log-spaced in [1e-12, 1e5) (even i) or uniform in [0, 16) (odd i) for sin / cos; (s, c) = (u, 2v - 1), u in (0, 1],
// v in [0, 1] on a sqrt(n) x sqrt(n) grid plus the same scaled by 1e-9 for atan2 (the log map feeds it sqrt(...) and a trace).
// out[0..2] = largest relative difference (as the bits of a non-negative double) of sin, cos, atan2.
__global__ void __launch_bounds__(256) k_selftest_trig(unsigned n, unsigned side, unsigned long long* out)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const double f = ((double)i + 0.5) / (double)n;
    const double x = (i & 1) ? 16.0 * f : exp(log(1e-12) + f * (log(1e5) - log(1e-12)));
    double sd, cd;
    sincos_dev(x, sd, cd);
    const double sl = sin(x), cl = cos(x);
    const double es = fabs(sd - sl) / fabs(sl), ec = fabs(cd - cl) / fabs(cl);
    const unsigned gy = i / side, gx = i - gy * side;
    double y = ((double)gy + 1.0) / (double)side, c = 2.0 * ((double)gx / (double)(side - 1)) - 1.0;
    if (gy & 1) { y *= 1e-9; }
    const double ad = atan2_dev(y, c), al = atan2(y, c);
    const double ea = fabs(ad - al) / fabs(al);
    atomicMax(&out[0], (unsigned long long)__double_as_longlong(es));
    atomicMax(&out[1], (unsigned long long)__double_as_longlong(ec));
    atomicMax(&out[2], (unsigned long long)__double_as_longlong(ea));
}
void launch_selftest_trig(unsigned n, unsigned side, unsigned long long* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_trig, dim3((n + 255) / 256), dim3(256), 0, s, n, side, out);
}

void launch_selftest_sqrt(unsigned long long* out, hipStream_t s) { hipLaunchKernelGGL(k_selftest_sqrt, dim3(8192), dim3(256), 0, s, out); }
void launch_selftest_division(unsigned b_first, unsigned b_stride, unsigned b_count, unsigned long long* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_division, dim3((b_count + 255) / 256), dim3(256), 0, s, b_first, b_stride, b_count, out);
}

void launch_selftest_reciprocal(unsigned long long* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_reciprocal, dim3(8192), dim3(256), 0, s, out);
}

void launch_visualize(int mode, const float* a, const float* b, int n, uint8_t* rgb, hipStream_t s)
{
    hipLaunchKernelGGL(k_visualize, dim3((n + 255) / 256), dim3(256), 0, s, mode, a, b, n, rgb);
}

void launch_ingest(const uint8_t* rgb, int channels, const uint16_t* depth16, int n, float depth_scale, float sigma_valid,
                   float sigma_invalid, int invalidate_gray, float* gray, float* depth, float* sigma, hipStream_t s)
{
    hipLaunchKernelGGL(k_ingest, dim3((n + 255) / 256), dim3(256), 0, s, rgb, channels, depth16, n, (float)(1.0 / 255.0), depth_scale,
                       sigma_valid, sigma_invalid, invalidate_gray, gray, depth, sigma);
}

void launch_undistort(const float* src, int w, int h, const Intr& k, const float D[5], float border, float* dst, hipStream_t s)
{
    hipLaunchKernelGGL(k_undistort, dim3((w * h + 255) / 256), dim3(256), 0, s, src, w, h, k, D[0], D[1], D[2], D[3], D[4], border, dst);
}

static inline unsigned cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }

void launch_pyramid(const PyramidArgs& a0, int n_seq, hipStream_t s)
{
    PyramidArgs a = a0;
    a.n_seq = n_seq;
    const int tw = a.w[a.levels - 1], th = a.h[a.levels - 1];
    // raw 1-channel frames with the usual alignment: four kept pixels per thread, wide loads and stores
    const bool vec = a.raw_rgb != nullptr && a.raw_channels == 1 && (a.culls == 1 || a.culls == 2) && (tw % 4) == 0 && (a.src_w % (4 << a.culls)) == 0 &&
                     (reinterpret_cast<uintptr_t>(a.raw_rgb) % 16) == 0 && (reinterpret_cast<uintptr_t>(a.raw_depth) % 16) == 0;
    bool aligned = true;   // the 16-byte top-level stores
    {
        const int T = a.levels - 1;
        const void* tops[4] = {a.dst[0][T], a.dst[1][T], a.dst[2][T], a.wgt[T]};
        for (const void* p : tops) aligned = aligned && (reinterpret_cast<uintptr_t>(p) % 16) == 0;
        aligned = aligned && ((size_t)tw * th % 4) == 0;
    }
    if (vec && aligned) {
        const dim3 grid = seq_grid(cdiv((tw >> 2) * th, 256), (unsigned)n_seq);
        if (a.culls == 1) hipLaunchKernelGGL(k_pyramid_raw4<1>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_pyramid_raw4<2>, grid, dim3(256), 0, s, a);
        return;
    }
    hipLaunchKernelGGL(k_pyramid, seq_grid(cdiv(tw * th, 256), (unsigned)n_seq), dim3(256), 0, s, a);
}

void launch_cull(const float* src, int w, int h, int times, float* dst, hipStream_t s)
{
    const int n = (w >> times) * (h >> times);
    hipLaunchKernelGGL(k_cull, dim3(cdiv(n, 256)), dim3(256), 0, s, src, w, h, times, dst);
}

void launch_gradient(const float* img, int w, int h, int xdir, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_gradient, dim3(cdiv(w * h, 256)), dim3(256), 0, s, img, w, h, xdir, out);
}

void launch_warp_image(const float* gray, const float* depth, int w, int h, const Intr& k, const Pose& pose, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_warp_image, dim3(cdiv(w * h, 256)), dim3(256), 0, s, gray, depth, w, h, k, pose, out);
}

int gn_blocks_per_seq(int w, int h, int ppt, int crop) { return gn_tiling(w, h, ppt, crop).count; }

template <int PPT, int G>
static void launch_track_gn_t(const GnArgs& a, unsigned tiles, hipStream_t s)
{
    const unsigned g = (tiles + 7u) & ~7u;  // a multiple of 8: blockIdx % 8 is the XCD
    if constexpr (PPT == 4) {
        if (gn_tiling(a.w, a.h, PPT, a.prm.crop).t2d) {  // 2-D tiles
            if (a.mask) hipLaunchKernelGGL((k_track_gn<PPT, G, true, true>), dim3(g), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_track_gn<PPT, G, false, true>), dim3(g), dim3(256), 0, s, a);
            return;
        }
    }
    if (a.mask) hipLaunchKernelGGL((k_track_gn<PPT, G, true>), dim3(g), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_track_gn<PPT, G, false>), dim3(g), dim3(256), 0, s, a);
}

void launch_track_gn(const GnArgs& a0, int n_seq, int ppt, int group, hipStream_t s, int grid_seqs)
{
    GnArgs a = a0;
    a.n_seq = n_seq;
    const GnTiling tl = gn_tiling(a.w, a.h, ppt, a.prm.crop);
    a.blk_first = tl.live_first; a.blk_count = tl.live_count;
    a.t_shift = tl.shift; a.x_org = tl.x_org; a.y_org = tl.y_org;
    if (tl.t2d) a.tiles_x = tl.tiles_x;
    // grid_seqs: an upper bound of the sequences on the active list (the host knows one from the progress words): the grid then only
    // holds workgroups that can find a tile -- at 16 384 sequences x 75 tiles an all-empty grid alone costs ~0.35 ms to dispatch
    const int gs = (grid_seqs > 0 && grid_seqs < n_seq && a.list != nullptr) ? grid_seqs : n_seq;
    unsigned grid = (unsigned)a.blk_count * (unsigned)gs;
    if (grid == 0) grid = 8;  // (nothing live: the workgroups only clear the next list counter)
    switch (ppt * 10 + group) {
        case 11: launch_track_gn_t<1, 1>(a, grid, s); break;
        case 21: launch_track_gn_t<2, 1>(a, grid, s); break;
        case 22: launch_track_gn_t<2, 2>(a, grid, s); break;
        case 41: launch_track_gn_t<4, 1>(a, grid, s); break;
        case 42: launch_track_gn_t<4, 2>(a, grid, s); break;
        case 44: launch_track_gn_t<4, 4>(a, grid, s); break;
        case 81: launch_track_gn_t<8, 1>(a, grid, s); break;
        case 82: launch_track_gn_t<8, 2>(a, grid, s); break;
        default: launch_track_gn_t<8, 4>(a, grid, s); break;
    }
}

// k_track_gn_fused for the (ppt, group) pairs the small-batch tiling picks; returns false when there is no such instance
bool launch_track_gn_fused(const GnArgs& a0, const SolveArgs& sa0, int n_seq, int ppt, int group, int* ticket, int* report, int* progress,
                           hipStream_t s)
{
    GnArgs a = a0;
    SolveArgs sa = sa0;
    a.n_seq = n_seq; a.list = nullptr; a.next_count = nullptr; a.mask = nullptr;
    const GnTiling tl = gn_tiling(a.w, a.h, ppt, a.prm.crop);
    a.blk_first = tl.live_first; a.blk_count = tl.live_count;
    a.t_shift = tl.shift; a.x_org = tl.x_org; a.y_org = tl.y_org;
    if (tl.t2d) a.tiles_x = tl.tiles_x;
    sa.n_seq = n_seq; sa.list_in = nullptr; sa.list_out = nullptr; sa.progress = nullptr;
    sa.blk_first = tl.live_first; sa.blk_count = tl.live_count;
    if (a.blk_count <= 0) return false;
    FusedArgs f{ticket, report, progress, n_seq};
    const dim3 grid((unsigned)a.blk_count * (unsigned)n_seq);
    const int key = ppt * 10 + group;
    if (key == 11) hipLaunchKernelGGL((k_track_gn_fused<1, 1, false>), grid, dim3(256), 0, s, a, sa, f);
    else if (key == 22) hipLaunchKernelGGL((k_track_gn_fused<2, 2, false>), grid, dim3(256), 0, s, a, sa, f);
    else if (key == 42 && tl.t2d) hipLaunchKernelGGL((k_track_gn_fused<4, 2, true>), grid, dim3(256), 0, s, a, sa, f);
    else if (key == 42) hipLaunchKernelGGL((k_track_gn_fused<4, 2, false>), grid, dim3(256), 0, s, a, sa, f);
    else return false;
    return true;
}

bool track_persist_available(int ppt, int group)
{
    const int key = ppt * 10 + group;
    return key == 11 || key == 22 || key == 42;
}

int track_persist_max_grid(int ppt, int group, int* out)
{
    int per_cu = 0, dev = 0;
    const int key = ppt * 10 + group;
    hipError_t e = hipErrorInvalidValue;
    if (key == 11) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_track_persist<1, 1, true>, 256, 0);   // (the instance with more registers)
    else if (key == 22) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_track_persist<2, 2, true>, 256, 0);
    else if (key == 42) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_track_persist<4, 4, true>, 256, 0);
    if (e != hipSuccess) return DVO_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return DVO_ERR_HIP;
    *out = per_cu * prop.multiProcessorCount;
    return DVO_OK;
}

bool launch_track_persist(const PersistArgs& p, int ppt, int group, int grid, hipStream_t s)
{
    const int key = ppt * 10 + group;
    if (grid < 1) return false;
    // (the gather group only sets how many pixels' gathers are in flight together, never a result: this kernel runs one wave per SIMD
    //  whatever it does, so it takes all of a thread's pixels at once)
    if (p.mono.enabled) {
        if (key == 11) hipLaunchKernelGGL((k_track_persist<1, 1, true>), dim3(grid), dim3(256), 0, s, p);
        else if (key == 22) hipLaunchKernelGGL((k_track_persist<2, 2, true>), dim3(grid), dim3(256), 0, s, p);
        else if (key == 42) hipLaunchKernelGGL((k_track_persist<4, 4, true>), dim3(grid), dim3(256), 0, s, p);
        else return false;
        return true;
    }
    if (key == 11) hipLaunchKernelGGL((k_track_persist<1, 1, false>), dim3(grid), dim3(256), 0, s, p);
    else if (key == 22) hipLaunchKernelGGL((k_track_persist<2, 2, false>), dim3(grid), dim3(256), 0, s, p);
    else if (key == 42) hipLaunchKernelGGL((k_track_persist<4, 4, false>), dim3(grid), dim3(256), 0, s, p);
    else return false;
    return true;
}

void launch_track_level(const GnArgs& ga0, const SolveArgs& sa0, int n_seq, hipStream_t s)
{
    GnArgs ga = ga0;
    SolveArgs sa = sa0;
    ga.n_seq = n_seq;
    ga.list = nullptr; ga.next_count = nullptr; ga.mask = nullptr;
    const GnTiling tl = gn_tiling(ga.w, ga.h, 4, ga.prm.crop);  // (the caller made sure these are raster tiles: !tl.t2d)
    ga.blk_first = tl.live_first; ga.blk_count = tl.live_count;
    sa.list_in = nullptr; sa.list_out = nullptr; sa.result = nullptr;
    hipLaunchKernelGGL((k_track_level<4, 2>), dim3((unsigned)n_seq), dim3(256), 0, s, ga, sa);
}

template <int PPT>
static void launch_track_gn_tile_t(const GnArgs& a, const dim3& grid, hipStream_t s)
{
    const size_t lds = (128 + (size_t)(4 * PPT + 2 * a.margin + 3) * (64 + 2 * a.margin + 3)) * sizeof(float);
    if (a.mask) hipLaunchKernelGGL((k_track_gn_tile<PPT, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((k_track_gn_tile<PPT, false>), grid, dim3(256), lds, s, a);
}

void launch_track_gn_tile(const GnArgs& a, int n_seq, int ppt, hipStream_t s)
{
    const dim3 grid((unsigned)a.nblk * (unsigned)n_seq);
    switch (ppt) {
        case 1: launch_track_gn_tile_t<1>(a, grid, s); break;
        case 2: launch_track_gn_tile_t<2>(a, grid, s); break;
        case 4: launch_track_gn_tile_t<4>(a, grid, s); break;
        default: launch_track_gn_tile_t<8>(a, grid, s); break;
    }
}

void launch_prep_ref(const PrepArgs& a, hipStream_t s)
{
    const size_t n = a.level_end[a.levels - 1];
    if (n == 0) return;
    hipLaunchKernelGGL(k_prep_ref, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
}

void launch_gn_solve(const SolveArgs& a, int n_seq, hipStream_t s)
{
    SolveArgs b = a;
    b.n_seq = n_seq;
    hipLaunchKernelGGL(k_gn_solve, dim3((unsigned)((n_seq + DVO_SOLVE_SEQ - 1) / DVO_SOLVE_SEQ)), dim3(32 * DVO_SOLVE_SEQ), 0, s, b);
}

void launch_track_begin(SeqState* state, dvo_track_log* log, int n_seq, int levels, hipStream_t s)
{
    hipLaunchKernelGGL(k_track_begin, dim3(cdiv(n_seq, 256)), dim3(256), 0, s, state, log, n_seq, levels);
}

void launch_set_pose(SeqState* state, const float* xi_dev, int n_seq, hipStream_t s)
{
    hipLaunchKernelGGL(k_set_pose, dim3(cdiv(n_seq, 64)), dim3(64), 0, s, state, xi_dev, n_seq);
}

void launch_export_poses(const SeqState* state, float* xi_out, float* T_out, int n_seq, hipStream_t s, float* host_result, int host_tag)
{
    hipLaunchKernelGGL(k_export_poses, dim3(cdiv(n_seq, 64)), dim3(64), 0, s, state, xi_out, T_out, n_seq, host_result, host_tag);
}

void launch_se3(int op, const float* a, const float* b, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_se3, dim3(1), dim3(64), 0, s, op, a, b, out);
}

}  // namespace dvo
