// dvo_kernels.h -- argument blocks and launch wrappers of the HIP kernels (dvo_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/dvo.h"
#include "dvo_math.h"

namespace dvo {

// Per-sequence tracker state, resident on the device for the whole coarse-to-fine loop.
struct SeqState {
    float xi[6];   // current relative twist (tracker.cpp:28,48)
    Pose  pose;    // exp(-xi) rounded to float: what every warp of the level uses
    int   active;  // 0 once the level's stop test fired (tracker.cpp:68-73)
    int   iter;    // iterations done on the current level
    double Tc[12]; // exp(+xi) in double (R row major, then t): carried so each iteration evaluates 2 exps instead of 4
};

struct PyramidArgs {
    const float* src[3];               // gray, depth, sigma at input resolution [n_seq][src_h][src_w] (nullptr = skip)
    float* dst[3][DVO_MAX_LEVELS];     // per map, per level [n_seq][h][w]
    int src_w, src_h, culls, levels;
    int w[DVO_MAX_LEVELS], h[DVO_MAX_LEVELS];
    float inv_tw;                      // 1 / top-level width
    int n_seq = 0;                     // set by launch_pyramid
    // optional (wgt[0] != nullptr, depth and sigma present): also write the k_prep_ref map of every level
    float* wgt[DVO_MAX_LEVELS];
    float step[DVO_MAX_LEVELS];
    float sigma_min, sigma_max;
    // optional raw sensor input (raw_rgb != nullptr; src[] is then ignored): the conversion of k_ingest (loader.cpp:55-60,137-147,
    // transform.cpp:60-76) is applied to the 1/4^culls of the pixels the pyramid keeps, while they are loaded -- the float maps
    // of the full frame are never materialised.  raw_depth == nullptr: gray only (mono).
    const uint8_t* raw_rgb;      // [n_seq][src_h][src_w][raw_channels], channels 1 (gray), 3 (R,G,B) or 4 (R,G,B,A)
    const uint16_t* raw_depth;   // [n_seq][src_h][src_w]
    int raw_channels, raw_invalidate_gray;
    // the input buffers (raw or float) hold src_img_rows rows per image and top-level row y comes from stored row y << src_row_shift:
    // (src_h, culls) for a whole frame, (src_h >> culls, 0) when the host uploaded only the rows the pyramid keeps (upload_rows)
    int src_img_rows, src_row_shift;
    float raw_gray_scale, raw_depth_scale, raw_sigma_valid, raw_sigma_invalid;
};

// Grids of the per-(sequence, pixel) kernels: x = workgroups of one sequence, (y, z) = the sequence -- seq = z * 32768 + y, so the
// sequence index costs no division and is not capped by the 65 535 limit of one grid dimension.  Kernels return for seq >= n_seq.
#define DVO_GRID_SEQ_Y 32768u
#ifdef __HIPCC__
inline dim3 seq_grid(unsigned blocks_per_seq, unsigned n_seq)
{
    const unsigned gy = n_seq < DVO_GRID_SEQ_Y ? (n_seq ? n_seq : 1u) : DVO_GRID_SEQ_Y;
    return dim3(blocks_per_seq ? blocks_per_seq : 1u, gy, (n_seq + gy - 1u) / gy);
}
#endif

struct GnArgs {
    const float* obj_gray;   // level buffers [n_seq][h][w]
    const float* ref_gray;
    const float* ref_depth;
    const float* ref_wgt;    // step / clamp(ref_sigma)     (k_prep_ref); 1 / ref_depth is recomputed per pixel (recip_rn: the IEEE quotient)
    float wgt_const = 0.0f;  // ref_wgt == nullptr: the weight of every pixel that can contribute (FrameSet::sigma_by_validity)
    const SeqState* state;
    float* partials;         // [n_seq][nblk][32]
    uint8_t* mask;           // optional [n_seq][h][w], pre-zeroed
    int w, h, nblk;
    float inv_w;
    int q256, r256;          // 256 / w and 256 % w: a thread's next pixel is 256 further in raster order
    Intr k;
    GnParams prm;
    int ignore_active;       // 1 on the first iteration of a level / probes (k_track_gn_tile only; k_track_gn uses `list`)
    // k_track_gn is launched with a fixed, resident-sized grid whose workgroups stride over the tiles of the ACTIVE
    // sequences only.  list = nullptr: all n_seq sequences; else list[0] = count and list[4..] = sequence ids
    // (written by the previous k_gn_solve).  next_count, if set, is zeroed for the k_gn_solve that follows.
    const int* list = nullptr;
    int* next_count = nullptr;
    int n_seq = 1;
    int blk_first = 0, blk_count = 0;  // live tiles of a sequence; set by launch_track_gn from gn_tiling()
    int t_shift = 6, x_org = 0, y_org = 0;  // 2-D tiles (with tiles_x below): see GnTiling
    // k_track_gn_tile only: 64 x (4*PPT) pixel tiles with the reference patch staged in LDS
    int tiles_x, tiles_y;    // nblk = tiles_x * tiles_y
    int margin;              // patch = tile grown by margin+1 (left/top) and margin+2 (right/bottom) pixels
};

struct PrepArgs {  // per-pixel constants of a reference frame, all levels in one launch
    const float* depth;      // level buffers are contiguous: [level][n_seq][h][w]
    const float* sigma;
    float* wgt;
    size_t level_end[DVO_MAX_LEVELS];  // cumulative element count after each level
    float step[DVO_MAX_LEVELS];
    float sigma_min, sigma_max;
    int levels;
};

struct SolveArgs {
    SeqState* state;
    const float* partials;
    dvo_track_log* log;        // optional [n_seq]
    dvo_gn_result* result;     // optional [n_seq]
    unsigned long long* counters;  // optional: [0] += level_pixels, [1] += 1 per solved sequence-iteration
    int nblk, level, level_pixels;
    int max_iterations, fixed_iterations;
    float min_update, min_residual;
    int ignore_active;         // 1 on the first iteration of a level: every sequence restarts (iter = 0)
    // active-sequence lists ([0] = count, [4..] = ids): workgroup b handles list_in[4 + b] (nullptr: sequence b) and
    // appends its sequence to list_out while it stays active (order is irrelevant to the results)
    const int* list_in = nullptr;
    int* list_out = nullptr;
    int* progress = nullptr;   // optional, mapped HOST memory: workgroup 0 stores (sequences this iteration evaluated + 1)
    int n_seq = 1;             // sequences of the launch (set by launch_gn_solve)
    int blk_first = 0, blk_count = -1;  // partial rows outside [blk_first, blk_first + blk_count) count as zero (-1: all rows)
    long long* dbg_stamp = nullptr;     // diagnostic (k_track_persist's timeline): [0] after the 6x6 solve, [1] after the pose update
};

// k_track_persist: the whole of Tracker::track for ONE sequence in one launch (a dvo_vo handle).
struct PersistLevel {
    const float* obj_gray; const float* ref_gray; const float* ref_depth; const float* ref_wgt;
    float wgt_const, inv_w;
    int w, h, nblk, q256, r256;
    Intr k;
    GnParams prm;
    int blk_first, blk_count, t_shift, x_org, y_org, tiles_x, t2d, level_pixels;
};
struct MonoSeq;
struct AgeEntry;
struct PersistMono {       // optional tail of k_track_persist for a mono dvo_vo handle: what k_mono_decide does, by the solver's thread, in the
    MonoSeq* meta;         // same launch -- the pose, the keyframe decision and the world pose reach the host with the tracker's tag
    float ref_xi[6];       // the reference keyframe (the host keeps FrameHistory): MonoRef's fields
    int ref_id, n_total;
    int frame_id, max_frames;
    float min_translation;
    int enabled;
    // ... and k_age_table's (the per-keyframe relative poses of Mapper::update) when the frame is not a keyframe, by the solver's workgroup:
    const float* hist_xi;  // [n_hist][6], device copy of FrameHistory's poses (current: the host refreshed it before the launch); nullptr: not here
    AgeEntry* ages;        // [n_hist]
    int n_hist;
    int* zero_word;        // the valid-update counter of the depth update that follows (cleared here)
};
struct PersistArgs {
    PersistLevel lv[DVO_MAX_LEVELS];
    int levels;
    SeqState* state;       // [1]
    float* partials;       // [max nblk][32]
    dvo_track_log* log;    // [1]
    int* ctl;              // device memory, 64-byte aligned: the control line [0] epoch [1] next level [2] status [4..15] pose, then one arrival slot per workgroup
    int max_iterations, fixed_iterations;
    float min_update, min_residual;
    float* xi_out; float* T_out;
    float* host_result;    // fine-grained mapped host memory: [0..5] xi, [6..21] T, [22] tag, [23] tag of a launch that gave up;
                           // with `mono`: [24..29] frame_xi, [30..45] T_world, [46] need (written before the tag)
    PersistMono mono;
    int host_tag;          // unique per launch: also the base of this launch's epoch numbers
    int spin_limit;        // polls of the epoch word before a workgroup gives up (every wait in the kernel is bounded)
    int dbg_worker;        // which tile worker leaves the stamps (DVO_PERSIST_TIMELINE=<index>; 0 owns a corner tile of every level)
    long long* dbg;        // optional [2][64][8] wall-clock stamps (100 MHz) of the solver and of worker 0 per step (tools/persist_timeline.py)
};
bool track_persist_available(int ppt, int group);
int  track_persist_max_grid(int ppt, int group, int* out);   // workgroups that are co-resident on the current device
bool launch_track_persist(const PersistArgs& p, int ppt, int group, int grid, hipStream_t s);

// Tile geometry of k_track_gn: the single source of tile counts for host and device code.
//  * raster tiles: 256 * ppt consecutive pixels (any size; tiles outside the crop rows are not live);
//  * 2-D tiles (ppt = 4): TW = 2^shift columns x (64 / TW) * 16 rows, lane = (column, row-in-wave), wave w owns pixel rows
//    w*4 .. w*4+3 of the lane's row set.  Used when the width is a multiple of 64, 32 or 16 (64 x 16, 32 x 32, 16 x 64 tiles),
//    the level has no crop window and is at least two tiles tall: only tiles on the image border hold border pixels.
struct GnTiling {
    int t2d = 0, shift = 6, tiles_x = 1, x_org = 0, y_org = 0;
    int count = 1;                        // tiles (= partial rows) per sequence
    int live_first = 0, live_count = 0;   // tiles that are launched and summed
    long long live_pixels = 0;            // image pixels they cover (profile counter)
};
inline GnTiling gn_tiling(int w, int h, int ppt, int crop)
{
    GnTiling t;
    const int npix = w * h;
    // (Tried: 32 x 32 tiles laid over the crop window [20,140] x [20,100] of the crop level, so that no tile touches the image
    // border and nothing outside the window is evaluated.  No measurable change -- that level's launches are latency chains,
    // not work -- so the crop level keeps raster tiles; x_org / y_org / shift stay general for it.)
    if (ppt == 4 && !crop && w >= 16 && (w % 16) == 0) {
        // the widest power-of-two tile width (<= 64) that divides the level width; rows per tile = (64 / TW) * 16, and the
        // tile must not be taller than about half the image (else every tile touches the top and bottom border anyway)
        const int shift = (w % 64) == 0 ? 6 : ((w % 32) == 0 ? 5 : 4);
        const int rows = (64 >> shift) * 16;
        if (rows * 2 <= h) {
            t.t2d = 1; t.shift = shift; t.tiles_x = w >> shift;
            t.count = t.tiles_x * ((h + rows - 1) / rows);
            t.live_first = 0; t.live_count = t.count; t.live_pixels = npix;
            return t;
        }
    }
    const int T = 256 * ppt;
    t.count = (npix + T - 1) / T;
    t.live_first = 0; t.live_count = t.count; t.live_pixels = npix;
    if (crop) {
        int lo = t.count, hi = -1;
        for (int b = 0; b < t.count; b++) {
            const int row0 = (b * T) / w;
            int last = b * T + T - 1;
            if (last > npix - 1) last = npix - 1;
            if (last / w >= 20 && row0 <= 100) { if (b < lo) lo = b; if (b > hi) hi = b; }
        }
        if (hi < lo) { t.live_count = 0; t.live_pixels = 0; return t; }
        long long p1 = (long long)(hi + 1) * T;
        if (p1 > npix) p1 = npix;
        t.live_first = lo; t.live_count = hi - lo + 1; t.live_pixels = p1 - (long long)lo * T;
    }
    return t;
}

// ------------------------------------------------------------------------------------------------
// Mapping (src/map/mapper.cpp, src/map/implement.cpp) for n_seq sequences at once.  Every map is [n_seq][h][w] (top pyramid level);
// a launch covers all sequences and each sequence takes part or not according to its own Mapper::needNewFrame flag, which
// lives on the device (MonoSeq::need): no host round trip between tracking and mapping.
// ------------------------------------------------------------------------------------------------
struct MonoSeq {          // Mapper + FrameHistory state of one sequence (mapper.cpp:16-60, frame.hpp:146-188)
    float ref_xi[6];      // Frame::m_xi of the newest keyframe           } the first 32 bytes are what a single dvo_vo handle
    int   ref_id;         // its Frame::id                                } uploads before k_mono_decide (its FrameHistory
    int   n_total;        // keyframes created so far                     } lives on the host)
    float frame_xi[6];    // m_xi of the frame being processed = concatenate(ref_xi, rel_xi), frame.cpp:7-14
    float rel_xi[6];      // m_relative_xi = Tracker::track's result
    Pose  rel_pose;       // exp(+rel_xi): the warp of Mapper::propagate (mapper.cpp:66) and Mapper::update (mapper.cpp:94)
    float T_world[16];    // exp(frame_xi), system.hpp:73
    int   frame_id;
    int   need;           // Mapper::needNewFrame (mapper.cpp:45-60) of this frame
    int   valid_updates;  // "valid update: N pixel", mapper.cpp:136
    int   clamped;        // cumulative: pixels whose age pointed past the keyframe ring and were searched against the oldest retained
                          // keyframe instead (UpdateArgs::clamp_age; 0 for a history that holds every keyframe, as the reference's)
};

// What System::VisualOdometry::odometrize does between Tracker::track and Mapper::estimate (system.hpp:57-73, frame.cpp:7-14,
// mapper.cpp:45-60) for one sequence, in the double-precision pose algebra of dvo_math.h: rel_xi <- the tracker's twist, frame_xi <-
// concatenate(ref_xi, rel_xi), need <- needNewFrame, rel_pose <- exp(+rel_xi), T_world <- exp(frame_xi).  Shared by k_mono_decide and
// k_track_persist's mono tail: the same operations, the same bits.
__device__ __forceinline__ int mono_decide_one(MonoSeq& m, const float rel[6], int frame_id, float min_translation, int max_frames, float fx[6],
                                               float T[16])
{
    float ref[6];
    for (int i = 0; i < 6; i++) ref[i] = m.ref_xi[i];
    se3_concatenate_f(ref, rel, fx);
    const double tn2 = (double)rel[0] * rel[0] + (double)rel[1] * rel[1] + (double)rel[2] * rel[2];
    const int need = (sqrt(tn2) > (double)min_translation || (frame_id - m.ref_id >= max_frames)) ? 1 : 0;  // mapper.cpp:45-60
    se3_exp_f(fx, T);
    Pose rp;
    pose_from_xi(rel, 1.0f, rp);
    for (int i = 0; i < 6; i++) { m.rel_xi[i] = rel[i]; m.frame_xi[i] = fx[i]; }
    m.rel_pose = rp;
    for (int i = 0; i < 16; i++) m.T_world[i] = T[i];
    m.frame_id = frame_id;
    m.need = need;
    m.valid_updates = 0;
    return need;
}

struct AgeEntry {      // one keyframe as seen from the current frame (Mapper::update, mapper.cpp:99-107)
    Pose  pose;        // exp(-r_xi), r_xi = concatenate(obj.xi, -born.xi)
    float tneg[3];     // -r_xi[0:3] (twist part; implement.cpp:56)
    int   slot;        // where the born keyframe's top-level gray lives: ring slot (batch) or history index (single handle)
};

// One entry of the age table (Mapper::update, mapper.cpp:99-107, hoisted out of the pixel loop): r_xi = concatenate(obj.xi, -born.xi), the
// pose exp(-r_xi) the epipolar search warps with and -r_xi's translation (implement.cpp:56).  Shared by k_age_table and k_track_persist's tail.
__device__ __forceinline__ void age_entry_one(const float frame_xi[6], const float* born_xi, int slot, AgeEntry& e)
{
    float ox[6], nb[6], r_xi[6];
    for (int k = 0; k < 6; k++) { ox[k] = frame_xi[k]; nb[k] = -born_xi[k]; }
    se3_concatenate_f(ox, nb, r_xi);
    pose_from_xi(r_xi, -1.0f, e.pose);
    for (int k = 0; k < 3; k++) e.tneg[k] = -r_xi[k];
    e.slot = slot;
}

struct AgeTableArgs {  // k_age_table: AgeEntry of every retained keyframe, once per frame and sequence (never per pixel)
    const MonoSeq* meta;
    const float* hist_xi;    // [n_seq][R][6], indexed by slot
    AgeEntry* ages;          // [n_seq][R], indexed by HISTORY index (0 = oldest retained keyframe)
    int n_seq, R;
    int n_hist;              // >= 0: explicit history length and slot = index (single handle); < 0: ring, length min(n_total, R)
    int* zero_word = nullptr; // optional: cleared by this launch (the single handle's valid-update counter, UpdateArgs::valid_updates)
};

struct UpdateArgs {
    float* ref_depth; float* ref_sigma; float* ref_age;   // [n_seq][h][w], in place (top level of the reference keyframes)
    const float* obj_gray;                                 // [n_seq][h][w]
    const AgeEntry* ages;            // [n_seq][R], index = history index (oldest first)
    const float* ring_gray;          // [n_seq][R][h][w]: top-level gray of the retained keyframes (batch), or nullptr
    const float* const* gray_table;  // [n_hist] device pointers (single handle / operator level), used when ring_gray == nullptr
    const MonoSeq* meta;             // rel_pose, rel_xi[2], n_total, need, valid_updates per sequence; nullptr: the explicit fields
    int n_seq, R, n_hist, w, h, crop, obj_id;
    int clamp_age;           // bounded history: a pixel born in a dropped keyframe searches the oldest retained one
    float inv_ww = 0.0f;     // 1 / width of the launched window (set by launch_depth_update)
    uint32_t seed;
    Intr k;
    float K9[9];
    int k_sparse = 0;        // K9 = [fx 0 cx; 0 fy cy; 0 0 1] exactly (set by launch_depth_update): depthEstimate skips the products with the zeros
    Pose rel_pose;           // exp(+rel_xi)          } operator level only (meta == nullptr)
    float rel_tz;            // rel_xi[2]             }
    int* valid_updates;      //                       }
};

struct PropArgs {      // Implement::propagate (implement.cpp:217-256)
    const float* ref_depth; const float* ref_sigma; const float* ref_age;   // [n_seq][h][w]
    float* depth; float* sigma; float* age;                                  // [n_seq][h][w]
    int* owner;                                                              // [n_seq][h][w] scratch
    int w, h, n_seq;
    Intr k;
    const MonoSeq* meta;     // per-sequence pose + need flag; nullptr: `pose` / `tz` below, unconditional
    Pose pose; float tz;
    float inv_w = 0.0f;      // 1 / w (set by launch_propagate_batch)
    // Compact list of the sequences that take this branch ([0] = count, [4..] = ids, written by k_mono_decide): the grid then holds
    // min(n_seq, n_slots) sequence slots and slot j works through list entries j, j + n_slots, ... -- on a typical frame a sixth of
    // the sequences create a keyframe, and a workgroup that only finds out it has nothing to do costs as much to dispatch as one
    // that works.  nullptr: every sequence (or its `meta` flag).
    const int* need_list = nullptr;
    int n_slots = 0;
};

#define DVO_PROMOTE_MAX_SEG 8
struct PromoteArgs {   // a tracked frame becomes the newest keyframe of the sequences whose need flag is set (mapper.cpp:23-27)
    const float* src[DVO_PROMOTE_MAX_SEG];   // [n_seq][count] blocks of the frame set ...
    float* dst[DVO_PROMOTE_MAX_SEG];         // ... copied over the same blocks of the reference set
    int count[DVO_PROMOTE_MAX_SEG];
    int n_seg, n_seq;
    const float* gray_top;   // [n_seq][npix] also pushed into the keyframe ring:
    float* ring_gray;        // [n_seq][R][npix], slot n_total % R
    int npix, R;
    const MonoSeq* meta;
    int all;                 // 1: every sequence (first frame), 0: need flag
    const int* need_list = nullptr;   // as PropArgs::need_list (all == 0 only)
    int n_slots = 0;
};

struct MonoRef { float ref_xi[6]; int ref_id, n_total, valid; };   // reference keyframe of a single dvo_vo handle (kernel argument)
void launch_mono_decide(MonoSeq* meta, const SeqState* state, int n_seq, int frame_id, float min_translation, int max_frames,
                        float* xi_world, float* T_world, int* is_key, const MonoRef* host_ref, hipStream_t s, int* need_list = nullptr);
void launch_mono_commit(MonoSeq* meta, float* hist_xi, int n_seq, int R, int all, int frame_id, float* xi_world, float* T_world, int* is_key,
                        hipStream_t s);
void launch_age_table(const AgeTableArgs& a, hipStream_t s);
void launch_promote(const PromoteArgs& a, hipStream_t s);
void launch_broadcast(const float* src, float* dst, int count, int n_seq, hipStream_t s);  // dst[seq][i] = src[i]
void launch_propagate_batch(const PropArgs& a, hipStream_t s);
void launch_regularize_batch(const float* depth, const float* sigma, int w, int h, int n_seq, float* out, hipStream_t s);
// Mapper::regularize + Frame::updateDepth / updateDepthSigma (mapper.cpp:139-144, frame.cpp:39-61) in one pass over the top level:
// the regularized depth goes to `depth_top_out` (a second top-level buffer: the stencil reads the old one) and, with the current
// sigma, to every lower level together with 1/depth and the weight.
struct RegDecArgs {
    const float* depth; const float* sigma;   // top level [n_seq][h][w], read only
    float* depth_top_out;                     // [n_seq][h][w]
    float* depth_lv[DVO_MAX_LEVELS]; float* sigma_lv[DVO_MAX_LEVELS]; float* wgt[DVO_MAX_LEVELS];  // per level (top: wgt only)
    int w[DVO_MAX_LEVELS], h[DVO_MAX_LEVELS], levels, n_seq;
    float step[DVO_MAX_LEVELS], sigma_min, sigma_max;
    float inv_w = 0.0f;                       // 1 / top-level width (set by the launch wrapper)
};
void launch_regularize_redecimate(const RegDecArgs& a, hipStream_t s);

void launch_pyramid(const PyramidArgs& a, int n_seq, hipStream_t s);
void launch_cull(const float* src, int w, int h, int times, float* dst, hipStream_t s);
void launch_gradient(const float* img, int w, int h, int xdir, float* out, hipStream_t s);
void launch_warp_image(const float* gray, const float* depth, int w, int h, const Intr& k, const Pose& pose, float* out, hipStream_t s);
int  gn_blocks_per_seq(int w, int h, int ppt, int crop);  // = gn_tiling(...).count
void launch_track_gn(const GnArgs& a, int n_seq, int ppt, int group, hipStream_t s, int grid_seqs = 0);
void launch_prep_ref(const PrepArgs& a, hipStream_t s);
// LDS-tiled variant: a.tiles_x/tiles_y/margin/nblk must be set (see gn_tile_geometry)
void launch_track_gn_tile(const GnArgs& a, int n_seq, int ppt, hipStream_t s);
inline void gn_tile_geometry(int w, int h, int ppt, int& tiles_x, int& tiles_y)
{
    tiles_x = (w + 63) / 64;
    tiles_y = (h + 4 * ppt - 1) / (4 * ppt);
}
void launch_gn_solve(const SolveArgs& a, int n_seq, hipStream_t s);
// one Tracker::track iteration in one launch for a few sequences (k_track_gn_fused); false = no instance for (ppt, group)
bool launch_track_gn_fused(const GnArgs& a, const SolveArgs& sa, int n_seq, int ppt, int group, int* ticket, int* report, int* progress,
                           hipStream_t s);
inline bool gn_fused_available(int ppt, int group) { return (ppt == 1 && group == 1) || (ppt == 2 && group == 2) || (ppt == 4 && group == 2); }
// k_track_level: every iteration of one level in one launch (one workgroup per sequence); the level must have been
// tiled with 4 pixels per thread (ga.nblk = gn_blocks_per_seq(w, h, 4)) and have at most DVO_FUSED_MAX_TILES tiles
#define DVO_FUSED_MAX_TILES 8
void launch_track_level(const GnArgs& ga, const SolveArgs& sa, int n_seq, hipStream_t s);
void launch_track_begin(SeqState* state, dvo_track_log* log, int n_seq, int levels, hipStream_t s);
void launch_set_pose(SeqState* state, const float* xi_dev, int n_seq, hipStream_t s);
void launch_export_poses(const SeqState* state, float* xi_out, float* T_out, int n_seq, hipStream_t s, float* host_result = nullptr, int host_tag = 0);
void launch_se3(int op, const float* a, const float* b, float* out, hipStream_t s);
void launch_propagate(const float* ref_depth, const float* ref_sigma, const float* ref_age, int w, int h, const Intr& k,
                      const Pose& pose, float tz, int* owner, float* depth, float* sigma, float* age, hipStream_t s);
void launch_regularize(const float* depth, const float* sigma, int w, int h, float* out, hipStream_t s);
void launch_depth_update(const UpdateArgs& a, hipStream_t s);
void launch_ingest(const uint8_t* rgb, int channels, const uint16_t* depth16, int n, float depth_scale, float sigma_valid,
                   float sigma_invalid, int invalidate_gray, float* gray, float* depth, float* sigma, hipStream_t s);
void launch_selftest_trig(unsigned n, unsigned side, unsigned long long* out3, hipStream_t s);   // out3 = bits of the largest relative differences (sin, cos, atan2), pre-set 0
void launch_selftest_sqrt(unsigned long long* out3, hipStream_t s);          // out3 = {inputs, mismatches, first bad pattern}, pre-set {0, 0, ~0}
void launch_selftest_division(unsigned b_first, unsigned b_stride, unsigned b_count, unsigned long long* out3, hipStream_t s);   // (at most 2^17 values of b per launch)
void launch_selftest_reciprocal(unsigned long long* out3, hipStream_t s);  // out3 = {fast-path inputs, mismatches, first bad pattern}, pre-set {0, 0, ~0}
void launch_visualize(int mode, const float* a, const float* b, int n, uint8_t* rgb, hipStream_t s);
void launch_undistort(const float* src, int w, int h, const Intr& k, const float D[5], float border, float* dst, hipStream_t s);

}  // namespace dvo
