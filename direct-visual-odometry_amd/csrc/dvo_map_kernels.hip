// dvo_map_kernels.hip -- the mapping half of the hot path (src/map/mapper.cpp, src/map/implement.cpp) as gfx950 kernels, for
// n_seq sequences per launch.  Maps are [n_seq][h][w] blocks of the top pyramid level; the workgroup index carries the sequence
// (grid.x = workgroups per sequence * n_seq) and every kernel that belongs to one branch of Mapper::estimate (mapper.cpp:16-33)
// starts by reading its sequence's need flag -- one scalar load -- and leaves if the sequence took the other branch.
// The same kernels serve the single-sequence dvo_vo handle (n_seq = 1) and the operator-level entry points.
// Built with -ffp-contract=off: only the fmaf() calls written here and in dvo_math.h fuse.
#include <hip/hip_runtime.h>

#include "dvo_kernels.h"

namespace dvo {

static inline unsigned cdiv_u(unsigned a, unsigned b) { return (a + b - 1) / b; }

// (sequence, pixel) of a thread; false when the thread has no pixel.  Grid = seq_grid() (dvo_kernels.h): the sequence is
// (blockIdx.z, blockIdx.y), no division.
__device__ __forceinline__ int grid_seq() { return (int)(blockIdx.z * DVO_GRID_SEQ_Y + blockIdx.y); }
__device__ __forceinline__ bool seq_pixel(int npix, int n_seq, int& seq, int& i)
{
    seq = grid_seq();
    i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    return (i < npix) & (seq < n_seq);
}
// y = i / w, x = i % w for 0 <= i < 2^24 through the reciprocal (the +-1 fix-up is branch free)
__device__ __forceinline__ void split_row(int i, int w, float inv_w, int& x, int& y)
{
    y = (int)((float)i * inv_w);
    x = i - y * w;
    const int lo = x < 0 ? 1 : 0, hi = x >= w ? 1 : 0;
    y += hi - lo;
    x += (lo - hi) * w;
}

// ------------------------------------------------------------------------------------------------
// k_mono_decide: what System::VisualOdometry::odometrize does between Tracker::track and Mapper::estimate
// (system.hpp:57-73, frame.cpp:7-14, mapper.cpp:45-60), one thread per sequence, in the double-precision pose algebra of
// dvo_math.h: rel_xi <- the tracker's twist, frame_xi <- concatenate(ref_xi, rel_xi), need <- needNewFrame,
// rel_pose <- exp(+rel_xi), T_world <- exp(frame_xi).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_mono_decide(MonoSeq* meta, const SeqState* state, int n_seq, int frame_id, float min_translation,
                                                    int max_frames, float* xi_world, float* T_world, int* is_key, MonoRef host_ref, int* need_list)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_seq) return;
    MonoSeq& m = meta[s];
    if (host_ref.valid) {   // a dvo_vo handle keeps FrameHistory on the host: the reference keyframe's pose / id arrive as kernel arguments
        for (int i = 0; i < 6; i++) m.ref_xi[i] = host_ref.ref_xi[i];
        m.ref_id = host_ref.ref_id; m.n_total = host_ref.n_total;
    }
    float rel[6], fx[6], T[16];
    for (int i = 0; i < 6; i++) rel[i] = state[s].xi[i];
    const int need = mono_decide_one(m, rel, frame_id, min_translation, max_frames, fx, T);
    if (need && need_list) need_list[4 + atomicAdd(&need_list[0], 1)] = s;   // (the order of the list changes no result: sequences are independent)
    if (xi_world) for (int i = 0; i < 6; i++) xi_world[s * 6 + i] = fx[i];
    if (T_world) for (int i = 0; i < 16; i++) T_world[s * 16 + i] = T[i];
    if (is_key) is_key[s] = need;
}

// k_mono_commit: FrameHistory::setRefFrame / push (frame.hpp:151-157) for the sequences that created a keyframe: the frame's
// pose becomes the reference pose and enters the ring.  all = 1: first frame (identity pose, every sequence).
__global__ void __launch_bounds__(64) k_mono_commit(MonoSeq* meta, float* hist_xi, int n_seq, int R, int all, int frame_id, float* xi_world,
                                                    float* T_world, int* is_key)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_seq) return;
    MonoSeq& m = meta[s];
    if (all) {
        for (int i = 0; i < 6; i++) { m.ref_xi[i] = 0.0f; m.frame_xi[i] = 0.0f; m.rel_xi[i] = 0.0f; }
        for (int i = 0; i < 9; i++) m.rel_pose.R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
        for (int i = 0; i < 3; i++) m.rel_pose.t[i] = 0.0f;
        for (int i = 0; i < 16; i++) m.T_world[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        m.ref_id = frame_id; m.frame_id = frame_id; m.n_total = 1; m.need = 1; m.valid_updates = 0;
        for (int i = 0; i < 6; i++) hist_xi[((size_t)s * R) * 6 + i] = 0.0f;
        if (xi_world) for (int i = 0; i < 6; i++) xi_world[s * 6 + i] = 0.0f;
        if (T_world) for (int i = 0; i < 16; i++) T_world[s * 16 + i] = m.T_world[i];
        if (is_key) is_key[s] = 1;
        return;
    }
    if (!m.need) return;
    const int slot = m.n_total % R;
    for (int i = 0; i < 6; i++) { m.ref_xi[i] = m.frame_xi[i]; hist_xi[((size_t)s * R + slot) * 6 + i] = m.frame_xi[i]; }
    m.ref_id = m.frame_id;
    m.n_total += 1;
}

// k_age_table: the per-keyframe part of Mapper::update (mapper.cpp:99-107) hoisted out of the pixel loop: for every retained
// keyframe, r_xi = concatenate(obj.xi, -born.xi), the pose exp(-r_xi) that the epipolar search warps with and -r_xi's
// translation (implement.cpp:56).  One thread per (sequence, history index).
__global__ void __launch_bounds__(64) k_age_table(AgeTableArgs a)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t == 0 && a.zero_word) *a.zero_word = 0;
    if (t >= a.n_seq * a.R) return;
    const int seq = t / a.R, i = t - seq * a.R;
    const MonoSeq& m = a.meta[seq];
    int n_hist, slot;
    if (a.n_hist >= 0) {          // single handle: the table is the whole FrameHistory, slot = history index
        n_hist = a.n_hist; slot = i;
    } else {                      // ring: the newest min(n_total, R) keyframes
        if (m.need) return;       // (sequences that create a keyframe do not search)
        n_hist = m.n_total < a.R ? m.n_total : a.R;
        slot = (m.n_total - n_hist + i) % a.R;
    }
    if (i >= n_hist) return;
    AgeEntry e;
    age_entry_one(m.frame_xi, a.hist_xi + ((size_t)seq * a.R + slot) * 6, slot, e);
    a.ages[(size_t)seq * a.R + i] = e;
}

// ------------------------------------------------------------------------------------------------
// Implement::propagate (implement.cpp:217-256).  The reference's forEach scatter races; its sequential semantics are "last
// writer in raster order wins" (D7).  Three passes reproduce that exactly:
//   pass 0: outputs <- (1, 1, 0), owner <- -1;  pass 1: owner[target] = max(source index);
//   pass 2: every target pulls depth/sigma/age from its owning source pixel.
// ------------------------------------------------------------------------------------------------
// (four pixels per thread: on most frames most sequences take the other branch, and a workgroup that only finds that out costs
//  as much to dispatch as one that works -- a quarter of the workgroups)
#define DVO_PROP_PER_THREAD 4
#define DVO_LIST_SLOTS 1024   /* sequence slots of a launch that works through a compact list (PropArgs::need_list) */
__device__ __forceinline__ int prop_chunk(int& seq)   // first pixel of this thread (stride 256), sequence of the workgroup
{
    seq = grid_seq();
    return (int)blockIdx.x * (256 * DVO_PROP_PER_THREAD) + (int)threadIdx.x;
}

// The sequences a workgroup of the propagate / promote kernels works through: with a list, slot j takes entries j, j + n_slots, ...;
// without one, its own sequence (when the `need` flag says so).  Returns false when there is nothing (more) to do.
__device__ __forceinline__ bool next_listed_seq(const int* list, int n_slots, int n_seq, const MonoSeq* meta, int& cursor, int& seq)
{
    if (list) {
        if (cursor < 0) cursor = grid_seq(); else cursor += n_slots;
        if (cursor >= list[0]) return false;
        seq = list[4 + cursor];
        return true;
    }
    if (cursor >= 0) return false;
    cursor = 0;
    seq = grid_seq();
    return seq < n_seq && !(meta && !meta[seq].need);
}

__global__ void __launch_bounds__(256) k_propagate_init(PropArgs a)
{
    int seq, cursor = -1;
    const int n = a.w * a.h, i0 = prop_chunk(seq);
    while (next_listed_seq(a.need_list, a.n_slots, a.n_seq, a.meta, cursor, seq)) {
#pragma unroll
        for (int k = 0; k < DVO_PROP_PER_THREAD; k++) {
            const int i = i0 + k * 256;
            if (i >= n) break;
            const size_t o = (size_t)seq * n + i;
            a.depth[o] = 1.0f; a.sigma[o] = 1.0f; a.age[o] = 0.0f; a.owner[o] = -1;
        }
    }
}

__global__ void __launch_bounds__(256) k_propagate_owner(PropArgs a)
{
    int seq, cursor = -1;
    const int w = a.w, h = a.h, n = w * h, i0 = prop_chunk(seq);
    while (next_listed_seq(a.need_list, a.n_slots, a.n_seq, a.meta, cursor, seq)) {
        const Pose pose = a.meta ? a.meta[seq].rel_pose : a.pose;   // wave-uniform
#pragma unroll
        for (int k = 0; k < DVO_PROP_PER_THREAD; k++) {
            const int i = i0 + k * 256;
            if (i >= n) break;
            int x, y;
            split_row(i, w, a.inv_w, x, y);
            const float rd = a.ref_depth[(size_t)seq * n + i];
            if (is_epsilon(rd)) continue;
            float pu, pv;
            warp(pose, a.k, (float)x, (float)y, rd, pu, pv);
            int qx, qy;
            if (!round_coord(pu, qx) || !round_coord(pv, qy)) continue;
            if (qx < 0 || w <= qx || qy < 0 || h <= qy) continue;
            atomicMax(&a.owner[(size_t)seq * n + qy * w + qx], i);
        }
    }
}

__global__ void __launch_bounds__(256) k_propagate_pull(PropArgs a)
{
    int seq, cursor = -1;
    const int n = a.w * a.h, o0 = prop_chunk(seq);
    while (next_listed_seq(a.need_list, a.n_slots, a.n_seq, a.meta, cursor, seq)) {
        const float tz = a.meta ? a.meta[seq].rel_xi[2] : a.tz;
        const size_t base = (size_t)seq * n;
#pragma unroll
        for (int k = 0; k < DVO_PROP_PER_THREAD; k++) {
            const int o = o0 + k * 256;
            if (o >= n) break;
            const int i = a.owner[base + o];
            if (i < 0) continue;
            const float rd = a.ref_depth[base + i];
            float s = a.ref_sigma[base + i];
            const float d0 = rd < 0.01f ? 0.01f : rd;
            const float d1 = d0 + tz;
            const float q = d1 / d0;
            const float q4 = q * (q * (q * q));          // math::pow(q, 4), util.hpp:19-27
            s = sqrtf(fmaf(q4, s * s, 0.06f * 0.06f));   // implement.cpp:246-247
            a.depth[base + o] = d1 < 0.0f ? 0.0f : d1;
            a.sigma[base + o] = s;
            a.age[base + o] = a.ref_age[base + i] + 1.0f;
        }
    }
}

// The four neighbour fusions of Implement::regularize (implement.cpp:166-177), order L, R, D, U.  All eight neighbour loads are
// issued before the first fusion (from the centre's address where a neighbour does not exist; that value is not used): the fusions
// are a dependent chain, and with each load inside its own `if` the kernel paid four exposed memory round trips per pixel.
__device__ __forceinline__ void regularize_fuse4(const float* __restrict__ depth, const float* __restrict__ sigma, int i, int x, int y, int w, int h,
                                                 float& gd, float& gs)
{
    const bool hasL = x - 1 >= 0, hasR = x + 1 < w, hasD = y + 1 < h, hasU = y - 1 >= 0;
    const int iL = hasL ? i - 1 : i, iR = hasR ? i + 1 : i, iD = hasD ? i + w : i, iU = hasU ? i - w : i;
    const float dL = depth[iL], sL = sigma[iL], dR = depth[iR], sR = sigma[iR];
    const float dD = depth[iD], sD = sigma[iD], dU = depth[iU], sU = sigma[iU];
    // Are all ten operands positive normal floats in [2^-20, 2^20]?  As unsigned integers their bit patterns order like the floats,
    // negative values, infinities and NaNs lie above every positive finite one, zeros and subnormals below 2^-20: one min / max tree.
    // (A neighbour that does not exist was loaded from the centre: in range if the centre is.)  Then the eight divisions and four
    // square roots of the fusions take their verified short forms (dvo_math.h); otherwise -- a wave-uniform branch no real map takes --
    // the IEEE sequences.  Same bits either way.
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned b0 = __float_as_uint(gd), b1 = __float_as_uint(gs), b2 = __float_as_uint(dL), b3 = __float_as_uint(sL), b4 = __float_as_uint(dR),
                   b5 = __float_as_uint(sR), b6 = __float_as_uint(dD), b7 = __float_as_uint(sD), b8 = __float_as_uint(dU), b9 = __float_as_uint(sU);
    const unsigned lo = min(min(min(b0, b1), min(b2, b3)), min(min(b4, b5), min(min(b6, b7), min(b8, b9))));
    const unsigned hi = max(max(max(b0, b1), max(b2, b3)), max(max(b4, b5), max(max(b6, b7), max(b8, b9))));
    const bool ranged = (lo >= DVO_FUSE_RANGE_LO) & (hi <= DVO_FUSE_RANGE_HI);
    if (__builtin_amdgcn_ballot_w64(!ranged) == 0ull) {
        if (hasL) gaussian_fuse_ranged(gd, gs, dL, sL);
        if (hasR) gaussian_fuse_ranged(gd, gs, dR, sR);
        if (hasD) gaussian_fuse_ranged(gd, gs, dD, sD);
        if (hasU) gaussian_fuse_ranged(gd, gs, dU, sU);
        return;
    }
    asm volatile("; regularize_fuse4: IEEE division / square root");
#endif
    if (hasL) gaussian_fuse(gd, gs, dL, sL);
    if (hasR) gaussian_fuse(gd, gs, dR, sR);
    if (hasD) gaussian_fuse(gd, gs, dD, sD);
    if (hasU) gaussian_fuse(gd, gs, dU, sU);
}

// Implement::regularize (implement.cpp:156-180): reads the old maps, fuses L, R, D, U in that order.
__global__ void __launch_bounds__(256) k_regularize(const float* __restrict__ depth_all, const float* __restrict__ sigma_all, int w, int h,
                                                    int n_seq, float inv_w, float* __restrict__ out_all)
{
    int seq, i;
    if (!seq_pixel(w * h, n_seq, seq, i)) return;
    const size_t base = (size_t)seq * w * h;
    const float* __restrict__ depth = depth_all + base;
    const float* __restrict__ sigma = sigma_all + base;
    int x, y;
    split_row(i, w, inv_w, x, y);
    float gd = depth[i], gs = sigma[i];
    regularize_fuse4(depth, sigma, i, x, y, w, h, gd, gs);
    out_all[base + i] = gd < 6.0f ? gd : 6.0f;
}

// k_regularize_redecimate: k_regularize and the re-decimation that follows it in the mono pipeline, one pass.  Same operations per
// pixel as k_regularize + k_pyramid(culls = 0, depth and sigma): the regularized value is written to a second top-level buffer (the
// 5-point stencil of the neighbours still reads the old one) and decimated on the spot into every level it lands on, with sigma,
// and the Gauss-Newton weight -- saves one launch and the 8 B/px round trip of the intermediate map.
__global__ void __launch_bounds__(256) k_regularize_redecimate(RegDecArgs a)
{
    const int T = a.levels - 1, w = a.w[T], h = a.h[T];
    int seq, i;
    if (!seq_pixel(w * h, a.n_seq, seq, i)) return;
    const size_t base = (size_t)seq * w * h;
    const float* __restrict__ depth = a.depth + base;
    const float* __restrict__ sigma = a.sigma + base;
    int x, y;
    split_row(i, w, a.inv_w, x, y);
    const float s0 = sigma[i];
    float gd = depth[i], gs = s0;
    regularize_fuse4(depth, sigma, i, x, y, w, h, gd, gs);
    const float nd = gd < 6.0f ? gd : 6.0f;                       // implement.cpp:178
    a.depth_top_out[base + i] = nd;                               // top level: the map itself (cullImage(src, 0) aliases, convert.cpp:9-10)
    __builtin_nontemporal_store(gn_weight(a.step[T], a.sigma_min, a.sigma_max, s0), a.wgt[T] + base + i);
    const float vd = pass_valid(nd), vs = pass_valid(s0);
    for (int t = 1; t < a.levels; t++) {                          // lower levels keep pixels whose coordinates are multiples of 2^t
        const int msk = (1 << t) - 1;
        if ((x & msk) | (y & msk)) break;
        const int l = T - t, lx = x >> t, ly = y >> t;
        if (lx >= a.w[l] || ly >= a.h[l]) continue;
        const size_t o = (size_t)seq * a.w[l] * a.h[l] + (size_t)ly * a.w[l] + lx;
        a.depth_lv[l][o] = vd;
        a.sigma_lv[l][o] = vs;
        a.wgt[l][o] = gn_weight(a.step[l], a.sigma_min, a.sigma_max, vs);
    }
}

// ------------------------------------------------------------------------------------------------
// One reference pixel of Mapper::update + Implement::update (mapper.cpp:76-137, implement.cpp:23-152,182-214).  The per-keyframe
// relative poses come from k_age_table (never a per-pixel exp/log).  Split in a head (up to the epipolar segment, which fixes the
// number of search steps) and a tail (the search and what follows), so that k_depth_update can reorder pixels between the two.
// GlobalImg (dvo_math.h) with the address space stated: a keyframe image reached through a pointer table is otherwise a flat pointer
struct GlobalImgG {
    const __attribute__((address_space(1))) float* p;
    int w, h;
    __device__ __forceinline__ float at(int x, int y) const { return p[y * w + x]; }
};

struct UpdHead {   // what the head of the per-pixel computation hands to the search (kept in LDS between the two phases of k_depth_update)
    float sx, sy, ex, ey, length, depth, sigma, dmin, dmax;
    int qx, qy, bi;
};

// Head: mapper.cpp:90-107 + EpipolarSegment (implement.cpp:23-47).  Returns the number of search steps the pixel will take
// (an upper bound within one step; 0 = the pixel leaves before the search: nothing to do).
__device__ __forceinline__ int depth_update_head(const UpdateArgs& a, const int seq, const int x, const int y, UpdHead& hd)
{
    const int w = a.w, h = a.h, npix = w * h;
    const MonoSeq* m = a.meta ? a.meta + seq : nullptr;
    const int i = y * w + x;
    const size_t base = (size_t)seq * npix;
    const Pose rel_pose = m ? m->rel_pose : a.rel_pose;
    const float rel_tz = m ? m->rel_xi[2] : a.rel_tz;
    int n_hist = a.n_hist;
    if (m && a.ring_gray) n_hist = m->n_total < a.R ? m->n_total : a.R;
    const float d = a.ref_depth[base + i];
    float pu, pv;
    warp(rel_pose, a.k, (float)x, (float)y, d, pu, pv);               // mapper.cpp:94
    int qx, qy;
    if (!round_coord(pu, qx) || !round_coord(pv, qy)) return 0;
    if (qx < 0 || w <= qx || qy < 0 || h <= qy) return 0;
    const int age = (int)a.ref_age[base + i];                          // mapper.cpp:99
    int bi = n_hist - 1 - age;                                         // frame.hpp:176
    if (bi < 0 && a.clamp_age) {   // (rare: only once a sequence has created more than R keyframes and a pixel survived them all)
        bi = 0;
        if (m) atomicAdd(const_cast<int*>(&m->clamped), 1);
    }
    if (bi < 0 || bi >= n_hist) return 0;
    const AgeEntry& born = a.ages[(size_t)seq * a.R + bi];
    const float depth = d - rel_tz;                                    // mapper.cpp:104
    const float sigma = a.ref_sigma[base + i];
    // EpipolarSegment, implement.cpp:23-47
    const float dmin = (depth - sigma) < 0.10f ? 0.10f : (depth - sigma);
    const float dmax = depth + sigma;
    float sx, sy, ex, ey;
    warp(born.pose, a.k, (float)qx, (float)qy, dmax, sx, sy);
    warp(born.pose, a.k, (float)qx, (float)qy, dmin, ex, ey);
    const float sex = sx - ex, sey = sy - ey;
    const float length = (float)sqrt((double)sex * (double)sex + (double)sey * (double)sey);
    hd.sx = sx; hd.sy = sy; hd.ex = ex; hd.ey = ey; hd.length = length; hd.depth = depth; hd.sigma = sigma; hd.dmin = dmin; hd.dmax = dmax;
    hd.qx = qx; hd.qy = qy; hd.bi = bi;
    // steps of the search loop: it runs while |pt - start| < length, pt advancing one pixel per step, at most 102 times
    if (!(length > 0.0f)) return 1;                       // (NaN or zero length: the loop test fails at once; the pixel still has its tail)
    return length >= 102.0f ? 103 : (int)length + 2;
}

// Tail: doMatching, depthEstimate, sigmaEstimate and the fusion (implement.cpp:49-152,182-214, mapper.cpp:122-131) of one pixel.
__device__ __forceinline__ void depth_update_tail(const UpdateArgs& a, const int seq, const int x, const int y, const UpdHead& hd)
{
    const int w = a.w, h = a.h, npix = w * h;
    const MonoSeq* m = a.meta ? a.meta + seq : nullptr;
    const int i = y * w + x;
    const size_t base = (size_t)seq * npix;
    const int obj_id = m ? m->frame_id : a.obj_id;
    const AgeEntry& born = a.ages[(size_t)seq * a.R + hd.bi];
    const float* born_gray = a.ring_gray ? a.ring_gray + ((size_t)seq * a.R + born.slot) * npix : a.gray_table[born.slot];
    const GlobalImgG bg{(const __attribute__((address_space(1))) float*)born_gray, w, h};
    const float sx = hd.sx, sy = hd.sy, ex = hd.ex, ey = hd.ey, length = hd.length, depth = hd.depth, sigma = hd.sigma, dmin = hd.dmin, dmax = hd.dmax;
    const int qx = hd.qx, qy = hd.qy;
    const float sex = sx - ex, sey = sy - ey;
    // doMatching, implement.cpp:106-152
    const float og = a.obj_gray[base + qy * w + qx];
    const float dirx = (ex - sx) / length, diry = (ey - sy) / length;
    float ptx = sx, pty = sy, bestx = sx, besty = sy, min_ssd = 6.0f;
    float s_c = 0.0f, s_pc = 0.0f, pcx = 0.0f, pcy = 0.0f;   // sample reuse between steps (see the loop)
    int count = 0;
    // The loop test of implement.cpp:113 is sqrt(dx^2 + dy^2) < length in double.  With L2 = length^2 (exact in double), that is
    // decided without the square root whenever dx^2 + dy^2 is not within a relative 1e-12 of L2 (sqrt is monotonic and correctly
    // rounded: its result can only differ from the exact root's side of `length` inside that band); inside the band the literal
    // test runs.  Same decisions, one fp64 sqrt per pixel instead of one per step.
    const double Ld = (double)length, L2 = Ld * Ld, L2lo = L2 * (1.0 - 1e-12), L2hi = L2 * (1.0 + 1e-12);
    // ... and a float estimate of dx^2 + dy^2 (relative error < 2e-7) decides every step that is not within 1e-5 of the end of the
    // segment without any fp64 instruction; the double test above only runs in that last sliver (and for NaN / overflow).
    const float L2f = length * length, L2f_lo = L2f * (1.0f - 1e-5f), L2f_hi = L2f * (1.0f + 1e-5f);
    for (;;) {
        const float ddx = ptx - sx, ddy = pty - sy;
        const float q2f = fmaf(ddx, ddx, ddy * ddy);
        bool go;
        if (q2f < L2f_lo) go = true;
        else if (q2f > L2f_hi) go = false;
        else {
            const double q2 = (double)ddx * (double)ddx + (double)ddy * (double)ddy;
            if (q2 < L2lo) go = true;
            else if (q2 > L2hi) go = false;
            else go = sqrt(q2) < Ld;              // (also the NaN case: every comparison above is false)
        }
        if (!go) break;
        float ssd = 0.0f;
        ptx += dirx;
        pty += diry;
        // A step samples at pt - dir, pt and pt + dir (implement.cpp:124-126: pt + (i - N/2) * dir, i = 0, 1, 2).  pt + dir is, bit for
        // bit, the NEXT step's pt (the same float addition), so that sample is computed once and is the next step's centre; and
        // pt - dir equals the PREVIOUS step's pt except where the addition rounded across a binade (checked per step and lane, in
        // both coordinates): then the previous centre sample is this step's first.  Same coordinates -> same taps, fractions and
        // blend -> the same sample values as three evaluations per step, for ~1.3 evaluations (one new sample, plus the first one
        // again for the whole wave when any lane's round trip failed).  The first step of a pixel evaluates all three.
        // The reference's early exit -- the first INVALID sample sets ssd = 6 and leaves (implement.cpp:128-131) -- is applied to
        // the values: same result.
        float sgv[3];
        const float nx = ptx + dirx * -1.0f, ny = pty + diry * -1.0f;   // (the literal expressions of the three targets)
        const float cx = ptx + dirx * 0.0f, cy = pty + diry * 0.0f;
        const float fx = ptx + dirx * 1.0f, fy = pty + diry * 1.0f;
        const bool first = count == 0;                                     // wave-uniform: the lanes of a wave enter the loop together
        // (cx, cy) == the previous step's (fx, fy) holds by construction from the second step on (cx = pt + 0*dir = pt for finite dir;
        //  for a non-finite dir every sample of every step is INVALID either way)
        const bool need_n = first | !((nx == pcx) & (ny == pcy));
        const bool any_n = __ballot(need_n) != 0ull;
        // Interior fast path: pt in [2, w-3) x [2, h-3) and |dir| <= 1 (+ an ulp) put all three samples' 2 x 2 footprints inside the
        // image, where Convert::getSubpixelFromDense (convert.cpp:77-105) has no clamp, no range test and no INVALID exit: the same
        // truncation, fractions and blend4() without ~20 instructions of bounds logic per sample.  Wave-uniform choice.
        const bool inner = (ptx >= 2.0f) & (ptx < (float)(w - 3)) & (pty >= 2.0f) & (pty < (float)(h - 3));
        float s_n = s_pc, s_f;
        if (__ballot(!inner) == 0ull) {
            auto fast = [&](float px, float py) {
                const int x0 = (int)px, y0 = (int)py;
                // (global address space stated: born_gray may come from a pointer table, which makes it a flat pointer -- and flat loads
                //  also count against lgkmcnt and take the aperture check)
                const __attribute__((address_space(1))) float* q = (const __attribute__((address_space(1))) float*)born_gray + (y0 * w + x0);
                return blend4(q[0], q[1], q[w], q[w + 1], px - (float)x0, py - (float)y0);
            };
            s_f = fast(fx, fy);
            if (first) s_c = fast(cx, cy);
            if (any_n) { const float t = fast(nx, ny); s_n = need_n ? t : s_pc; }
        } else {
            s_f = get_subpixel_dense(bg, fx, fy);
            if (first) s_c = get_subpixel_dense(bg, cx, cy);
            if (any_n) { const float t = get_subpixel_dense(bg, nx, ny); s_n = need_n ? t : s_pc; }
        }
        sgv[0] = s_n; sgv[1] = s_c; sgv[2] = s_f;
        s_pc = s_c; pcx = cx; pcy = cy;      // this step's centre: the next step's first sample where the round trip holds
        s_c = s_f;                           // this step's far sample: the next step's centre
        bool any_invalid = false;
#pragma unroll
        for (int jj = 0; jj < 3; jj++) {
            any_invalid = any_invalid | is_invalid(sgv[jj]);
            const float diff = sgv[jj] - og;
            const int aw = 3 - abs(jj - 2);
            ssd = (float)((double)ssd + 1.0 * aw / 3 * (double)(diff * diff));  // implement.cpp:134 (discarded when a sample is INVALID)
        }
        if (any_invalid) ssd = 6.0f;
        if (ssd < min_ssd) { bestx = ptx; besty = pty; min_ssd = ssd; }
        if (count++ > 100) break;
    }
    if ((double)min_ssd > 3 * 0.1) return;                             // implement.cpp:145
    if (bestx < 0.0f || besty < 0.0f || bestx > (float)w || besty > (float)h) return;  // implement.cpp:196-200
    // depthEstimate, implement.cpp:49-71 (double from float inputs)
    float nd;
    {
        float q0f, q1f, q2f;
        back_project(a.k, (float)qx, (float)qy, 1.0f, q0f, q1f, q2f);
        const double q0 = q0f, q1 = q1f, q2 = q2f;
        const double t[3] = {(double)born.tneg[0], (double)born.tneg[1], (double)born.tneg[2]};
        const double xi3[3] = {(double)bestx, (double)besty, 1.0};
        double Rq[3], KRq[3], Kt[3];
        for (int r = 0; r < 3; r++)
            Rq[r] = (double)born.pose.R[3 * r] * q0 + (double)born.pose.R[3 * r + 1] * q1 + (double)born.pose.R[3 * r + 2] * q2;
        if (a.k_sparse) {
            // K = [fx 0 cx; 0 fy cy; 0 0 1]: a product with an exact zero adds an exact zero, so (fx x + 0 y) + cx z is fx x + cx z and
            // (0 x + 0 y) + 1 z is z -- the same doubles as the three-term rows below (for finite x, y, z: they are), 12 instead of 30
            // fp64 operations per pixel for K R q and K t.
            const double fx = (double)a.K9[0], cx = (double)a.K9[2], fy = (double)a.K9[4], cy = (double)a.K9[5];
            KRq[0] = fx * Rq[0] + cx * Rq[2]; KRq[1] = fy * Rq[1] + cy * Rq[2]; KRq[2] = Rq[2];
            Kt[0] = fx * t[0] + cx * t[2]; Kt[1] = fy * t[1] + cy * t[2]; Kt[2] = t[2];
        } else {
            for (int r = 0; r < 3; r++) {
                KRq[r] = (double)a.K9[3 * r] * Rq[0] + (double)a.K9[3 * r + 1] * Rq[1] + (double)a.K9[3 * r + 2] * Rq[2];
                Kt[r] = (double)a.K9[3 * r] * t[0] + (double)a.K9[3 * r + 1] * t[1] + (double)a.K9[3 * r + 2] * t[2];
            }
        }
        double aa = 0.0, ab = 0.0;
        // (sparse K: the third components are Rq[2] * 1 - Rq[2] and t[2] * 1 - t[2], exact zeros whose products add exact zeros)
        const int nr = a.k_sparse ? 2 : 3;
        for (int r = 0; r < nr; r++) {
            const double va = Rq[2] * xi3[r] - KRq[r];
            const double vb = t[2] * xi3[r] - Kt[r];
            aa += va * va;
            ab += va * vb;
        }
        nd = -(float)(ab / aa);
    }
    // sigmaEstimate, implement.cpp:73-104
    float ns;
    {
        const float l = length;
        const float lx = sex / l, ly = sey / l;
        const float alpha = (dmax - dmin) / l;
        int mx = 0, my = 0;
        round_coord(bestx, mx);
        round_coord(besty, my);
        mx = mx < 0 ? 0 : (mx > w - 1 ? w - 1 : mx);  // D5 clamp
        my = my < 0 ? 0 : (my > h - 1 ? h - 1 : my);
        const float gx = grad_x_at(bg, mx, my), gy = grad_y_at(bg, mx, my);
        if (is_invalid(gx) || is_invalid(gy)) return;  // new_sigma = -1 fails the gate of mapper.cpp:122
        const float gl = fabsf(fmaf(gy, ly, gx * lx));
        const float gl2 = gl * gl, gp2 = gl / l;
        const float epi = 0.25f / (gl2 < kEpsilon ? kEpsilon : gl2);
        const float lum = 0.5f / (gp2 < kEpsilon ? kEpsilon : gp2);
        ns = alpha * sqrtf(epi + lum);
    }
    if (nd > 0.2f && nd < 6.0f && ns > 0.0f && ns < 0.5f) {            // mapper.cpp:122
        float gd = depth, gs = sigma;
        const float reset = rng_depth(a.seed, (uint32_t)obj_id, (uint32_t)i);
        if (!gaussian_update(gd, gs, nd, ns, reset)) a.ref_age[base + i] = 0.0f;  // mapper.cpp:124-127
        else atomicAdd(a.valid_updates ? a.valid_updates : const_cast<int*>(&m->valid_updates), 1);   // (an explicit counter wins)
        a.ref_depth[base + i] = gd;                                    // mapper.cpp:130-131
        a.ref_sigma[base + i] = gs;
    }
}

// Mapper::update + Implement::update (mapper.cpp:76-137, implement.cpp:23-152,182-214): one thread per reference pixel of the
// window mapper.cpp:90 keeps.  FP32-VALU bound (<= 102 search steps x 3 bilinear samples per pixel), not HBM bound.
// The search length varies per pixel by two orders of magnitude (it is the epipolar segment of depth +- sigma: a few pixels for most,
// ~100 where depth - sigma clamps at 0.10 m) and a wave runs as long as its longest lane.  So the workgroup first computes every
// pixel's step count (the cheap head of the computation), sorts its 256 pixels by it (counting sort in LDS, longest first) and
// hands them out in that order: each wave then holds pixels of similar length and finishes together.  Pixels that leave before
// the search are dropped from the list.  Pixels are independent and the valid-update count is an integer sum, so the order changes
// no result (bit-exact vs the oracle and the one-pass form: tests/test_gpu_parity.py, tests/test_real_data.py).
__global__ void __launch_bounds__(256) k_depth_update(UpdateArgs a)
{
    __shared__ int bucket_cnt[128];     // pixels per step count, index 127 - steps (longest first)
    __shared__ int bucket_off[128];
    __shared__ int order[256];          // producer thread of the pixel each thread searches
    __shared__ int n_listed;
    __shared__ float head_f[9][256];    // UpdHead of every listed pixel, by producer thread (the head is computed once)
    __shared__ int head_i[3][256];
    const int w = a.w, h = a.h;
    // Only the window of mapper.cpp:90 (x in [16,144], y in [12,108]) is launched when the crop is on.
    const int x_lo = a.crop ? 16 : 0, y_lo = a.crop ? 12 : 0;
    const int ww = a.crop ? (min(144, w - 1) - 16 + 1) : w, wh = a.crop ? (min(108, h - 1) - 12 + 1) : h;
    if (ww <= 0 || wh <= 0) return;
    int seq, j;
    const bool mine = seq_pixel(ww * wh, a.n_seq, seq, j);       // (seq is block-uniform; only the last block of a sequence has idle threads)
    if (seq >= a.n_seq) return;
    if (a.meta && a.ring_gray && a.meta[seq].need) return;   // this sequence created a keyframe instead (mapper.cpp:23-27); block-uniform
    const int j0 = j - (int)threadIdx.x;                 // window index of this workgroup's first thread
    if (threadIdx.x < 128) bucket_cnt[threadIdx.x] = 0;
    __syncthreads();
    int steps = 0, rank = 0;
    if (mine) {
        int wx, wy;
        split_row(j, ww, a.inv_ww, wx, wy);
        UpdHead hd;
        steps = depth_update_head(a, seq, x_lo + wx, y_lo + wy, hd);
        if (steps > 0) {
            rank = atomicAdd(&bucket_cnt[127 - steps], 1);
            const int t = (int)threadIdx.x;
            head_f[0][t] = hd.sx; head_f[1][t] = hd.sy; head_f[2][t] = hd.ex; head_f[3][t] = hd.ey; head_f[4][t] = hd.length;
            head_f[5][t] = hd.depth; head_f[6][t] = hd.sigma; head_f[7][t] = hd.dmin; head_f[8][t] = hd.dmax;
            head_i[0][t] = hd.qx; head_i[1][t] = hd.qy; head_i[2][t] = hd.bi;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // exclusive prefix sum over the 128 buckets, two per lane of wave 0
        const int b0 = 2 * (int)threadIdx.x;
        const int c0 = bucket_cnt[b0], c1 = bucket_cnt[b0 + 1];
        int incl = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if ((int)threadIdx.x >= d) incl += up;
        }
        const int excl = incl - (c0 + c1);
        bucket_off[b0] = excl;
        bucket_off[b0 + 1] = excl + c0;
        if (threadIdx.x == 63) n_listed = incl;
    }
    __syncthreads();
    if (steps > 0) order[bucket_off[127 - steps] + rank] = (int)threadIdx.x;
    __syncthreads();
    if ((int)threadIdx.x < n_listed) {
        const int p = order[threadIdx.x];
        UpdHead hd;
        hd.sx = head_f[0][p]; hd.sy = head_f[1][p]; hd.ex = head_f[2][p]; hd.ey = head_f[3][p]; hd.length = head_f[4][p];
        hd.depth = head_f[5][p]; hd.sigma = head_f[6][p]; hd.dmin = head_f[7][p]; hd.dmax = head_f[8][p];
        hd.qx = head_i[0][p]; hd.qy = head_i[1][p]; hd.bi = head_i[2][p];
        const int jq = j0 + p;
        int wx, wy;
        split_row(jq, ww, a.inv_ww, wx, wy);
        depth_update_tail(a, seq, x_lo + wx, y_lo + wy, hd);
    }
}

// ------------------------------------------------------------------------------------------------
// k_promote: the tracked frame becomes the newest keyframe (Mapper::estimate's needNewFrame branch, mapper.cpp:23-27 +
// FrameHistory::push): its gray pyramid, the propagated top-level depth / sigma and the age map are copied over the reference
// set's, and its top-level gray enters the keyframe ring (slot n_total % R).  One launch, every segment of every flagged sequence.
// ------------------------------------------------------------------------------------------------
#define DVO_PROMOTE_PER_THREAD 8
// VEC = 4: every segment length is a multiple of 4 floats (the usual geometries), copied as 16-byte accesses; VEC = 1: any length
template <int VEC>
__global__ void __launch_bounds__(256) k_promote(PromoteArgs a)
{
    typedef float fvec __attribute__((ext_vector_type(VEC)));
    int total = a.npix;  // + the ring segment
    for (int g = 0; g < a.n_seg; g++) total += a.count[g];
    total /= VEC;        // in units of VEC floats
    // 2048 units per workgroup: most sequences do not create a keyframe on a given frame, and a workgroup that only finds that
    // out costs as much to dispatch as one that copies -- fewer, fatter workgroups
    int seq, cursor = -1;
    const int i0 = (int)blockIdx.x * (256 * DVO_PROMOTE_PER_THREAD) + (int)threadIdx.x;
    while (next_listed_seq(a.all ? nullptr : a.need_list, a.n_slots, a.n_seq, a.all ? nullptr : a.meta, cursor, seq)) {
        const MonoSeq& m = a.meta[seq];
        const int slot = a.all ? 0 : m.n_total % a.R;   // (k_mono_commit increments n_total AFTER this kernel)
#pragma unroll
        for (int k = 0; k < DVO_PROMOTE_PER_THREAD; k++) {
            int i = i0 + k * 256;
            if (i >= total) break;
            bool done = false;
            for (int g = 0; g < a.n_seg; g++) {
                const int cnt = a.count[g] / VEC;
                if (i < cnt) {
                    const size_t o = (size_t)seq * cnt + i;
                    reinterpret_cast<fvec*>(a.dst[g])[o] = reinterpret_cast<const fvec*>(a.src[g])[o];
                    done = true;
                    break;
                }
                i -= cnt;
            }
            if (!done) {
                const int cnt = a.npix / VEC;
                reinterpret_cast<fvec*>(a.ring_gray)[((size_t)seq * a.R + slot) * cnt + i] = reinterpret_cast<const fvec*>(a.gray_top)[(size_t)seq * cnt + i];
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_broadcast(const float* __restrict__ src, float* __restrict__ dst, int count, int n_seq)
{
    int seq, i;
    if (!seq_pixel(count, n_seq, seq, i)) return;
    dst[(size_t)seq * count + i] = src[i];
}

// ------------------------------------------------------------------------------------------------ launch wrappers
void launch_mono_decide(MonoSeq* meta, const SeqState* state, int n_seq, int frame_id, float min_translation, int max_frames,
                        float* xi_world, float* T_world, int* is_key, const MonoRef* host_ref, hipStream_t s, int* need_list)
{
    MonoRef r;
    if (host_ref) r = *host_ref; else { for (int i = 0; i < 6; i++) r.ref_xi[i] = 0.0f; r.ref_id = 0; r.n_total = 0; r.valid = 0; }
    hipLaunchKernelGGL(k_mono_decide, dim3(cdiv_u(n_seq, 64)), dim3(64), 0, s, meta, state, n_seq, frame_id, min_translation, max_frames,
                       xi_world, T_world, is_key, r, need_list);
}

void launch_mono_commit(MonoSeq* meta, float* hist_xi, int n_seq, int R, int all, int frame_id, float* xi_world, float* T_world, int* is_key,
                        hipStream_t s)
{
    hipLaunchKernelGGL(k_mono_commit, dim3(cdiv_u(n_seq, 64)), dim3(64), 0, s, meta, hist_xi, n_seq, R, all, frame_id, xi_world, T_world, is_key);
}

void launch_age_table(const AgeTableArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_age_table, dim3(cdiv_u((unsigned)a.n_seq * (unsigned)a.R, 64)), dim3(64), 0, s, a);
}

void launch_promote(const PromoteArgs& a, hipStream_t s)
{
    int total = a.npix;
    bool vec = (a.npix % 4) == 0 && (reinterpret_cast<uintptr_t>(a.gray_top) % 16) == 0 && (reinterpret_cast<uintptr_t>(a.ring_gray) % 16) == 0;
    for (int g = 0; g < a.n_seg; g++) {
        total += a.count[g];
        vec = vec && (a.count[g] % 4) == 0 && (reinterpret_cast<uintptr_t>(a.src[g]) % 16) == 0 && (reinterpret_cast<uintptr_t>(a.dst[g]) % 16) == 0;
    }
    PromoteArgs b = a;
    const bool listed = !a.all && a.need_list != nullptr;
    b.n_slots = listed ? (a.n_seq < DVO_LIST_SLOTS ? a.n_seq : DVO_LIST_SLOTS) : 0;
    const unsigned gs = listed ? (unsigned)b.n_slots : (unsigned)a.n_seq;
    if (vec) hipLaunchKernelGGL(k_promote<4>, seq_grid(cdiv_u(total / 4, 256 * DVO_PROMOTE_PER_THREAD), gs), dim3(256), 0, s, b);
    else hipLaunchKernelGGL(k_promote<1>, seq_grid(cdiv_u(total, 256 * DVO_PROMOTE_PER_THREAD), gs), dim3(256), 0, s, b);
}

void launch_broadcast(const float* src, float* dst, int count, int n_seq, hipStream_t s)
{
    hipLaunchKernelGGL(k_broadcast, seq_grid(cdiv_u(count, 256), (unsigned)n_seq), dim3(256), 0, s, src, dst, count, n_seq);
}

void launch_propagate_batch(const PropArgs& a0, hipStream_t s)
{
    PropArgs a = a0;
    a.inv_w = 1.0f / (float)a.w;
    a.n_slots = a.need_list ? (a.n_seq < DVO_LIST_SLOTS ? a.n_seq : DVO_LIST_SLOTS) : 0;
    const dim3 grid = seq_grid(cdiv_u(a.w * a.h, 256 * DVO_PROP_PER_THREAD), a.need_list ? (unsigned)a.n_slots : (unsigned)a.n_seq);
    hipLaunchKernelGGL(k_propagate_init, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_propagate_owner, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_propagate_pull, grid, dim3(256), 0, s, a);
}

void launch_propagate(const float* ref_depth, const float* ref_sigma, const float* ref_age, int w, int h, const Intr& k,
                      const Pose& pose, float tz, int* owner, float* depth, float* sigma, float* age, hipStream_t s)
{
    PropArgs a;
    a.ref_depth = ref_depth; a.ref_sigma = ref_sigma; a.ref_age = ref_age;
    a.depth = depth; a.sigma = sigma; a.age = age; a.owner = owner;
    a.w = w; a.h = h; a.n_seq = 1; a.k = k; a.meta = nullptr; a.pose = pose; a.tz = tz;
    launch_propagate_batch(a, s);
}

void launch_regularize_batch(const float* depth, const float* sigma, int w, int h, int n_seq, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_regularize, seq_grid(cdiv_u(w * h, 256), (unsigned)n_seq), dim3(256), 0, s, depth, sigma, w, h, n_seq, 1.0f / (float)w, out);
}

void launch_regularize_redecimate(const RegDecArgs& a0, hipStream_t s)
{
    RegDecArgs a = a0;
    const int T = a.levels - 1;
    a.inv_w = 1.0f / (float)a.w[T];
    hipLaunchKernelGGL(k_regularize_redecimate, seq_grid(cdiv_u(a.w[T] * a.h[T], 256), (unsigned)a.n_seq), dim3(256), 0, s, a);
}

void launch_regularize(const float* depth, const float* sigma, int w, int h, float* out, hipStream_t s)
{
    launch_regularize_batch(depth, sigma, w, h, 1, out, s);
}

void launch_depth_update(const UpdateArgs& a0, hipStream_t s)
{
    UpdateArgs a = a0;
    const int ww = a.crop ? ((a.w - 1 < 144 ? a.w - 1 : 144) - 16 + 1) : a.w, wh = a.crop ? ((a.h - 1 < 108 ? a.h - 1 : 108) - 12 + 1) : a.h;
    if (ww <= 0 || wh <= 0) return;
    a.inv_ww = 1.0f / (float)ww;
    a.k_sparse = (a.K9[1] == 0.0f && a.K9[3] == 0.0f && a.K9[6] == 0.0f && a.K9[7] == 0.0f && a.K9[8] == 1.0f) ? 1 : 0;
    hipLaunchKernelGGL(k_depth_update, seq_grid(cdiv_u(ww * wh, 256), (unsigned)a.n_seq), dim3(256), 0, s, a);
}

}  // namespace dvo
