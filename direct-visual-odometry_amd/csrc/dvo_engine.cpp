// dvo_engine.cpp -- host side of libdvo.so (see dvo_engine.h).  Reference citations: file:line under the reference tree.
#include "dvo_engine.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#include <dlfcn.h>

namespace dvo {

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_error;
void set_error(const std::string& s) { g_error = s; }
const char* last_error() { return g_error.c_str(); }

int check_hip(hipError_t e, const char* what)
{
    if (e == hipSuccess) return DVO_OK;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return DVO_ERR_NO_DEVICE;
    if (e == hipErrorOutOfMemory) return DVO_ERR_OUT_OF_MEMORY;
    return DVO_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------ tracing (roctx, optional)
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char* e = getenv("DVO_TRACE");
        if (!e || e[0] == '0') return;
        // rocprofv3 (--marker-trace) intercepts the rocprofiler-sdk flavour of roctx; the legacy libroctx64 serves roctracer tools
        void* h = nullptr;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void trace_push(const char* name) { if (roctx().push) (void)roctx().push(name); }
void trace_pop() { if (roctx().pop) (void)roctx().pop(); }

bool host_buffer_is_pinned(const void* p)
{
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();   // (an ordinary malloc'ed pointer is "invalid value" to HIP: not an error of ours)
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

int select_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible: libdvo has no CPU fallback");
        return DVO_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device ordinal out of range");
        return DVO_ERR_BAD_ARGUMENT;
    }
    DVO_HIP(hipSetDevice(device));
    return DVO_OK;
}

int DevBuf::alloc(size_t n)
{
    release();
    if (n == 0) n = 4;
    DVO_HIP(hipMalloc(&p, n));
    bytes = n;
    return DVO_OK;
}
void DevBuf::release()
{
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    owned = true;
}

int KeyframePool::take(size_t bytes, void** out)
{
    if (block_bytes == 0) block_bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > block_bytes) { set_error("keyframe pool: block size changed"); return DVO_ERR_BAD_ARGUMENT; }
    if (free_blocks.empty()) {
        const int per_slab = 16;
        auto slab = std::make_unique<DevBuf>();
        DVO_TRY(slab->alloc(block_bytes * per_slab));
        for (int i = per_slab - 1; i >= 0; i--) free_blocks.push_back(static_cast<char*>(slab->p) + block_bytes * (size_t)i);
        slabs.push_back(std::move(slab));
    }
    *out = free_blocks.back();
    free_blocks.pop_back();
    return DVO_OK;
}

// ------------------------------------------------------------------------------------------------ geometry
static void cull_intrinsic(const float K[9], int times, float out[9])
{  // Convert::cullIntrinsic, convert.cpp:22-29
    if (times == 0) {
        memcpy(out, K, 9 * sizeof(float));
        return;
    }
    const double r = (double)(1 << times);
    for (int i = 0; i < 9; i++) out[i] = (float)((double)K[i] / r);
    out[8] = 1.0f;
}

int make_geometry(const float K[9], int w, int h, int levels, int culls, Geometry& g)
{
    if (levels < 1 || levels > DVO_MAX_LEVELS || culls < 0 || culls > 8 || w <= 0 || h <= 0) {
        set_error("bad pyramid geometry");
        return DVO_ERR_BAD_ARGUMENT;
    }
    g.src_w = w; g.src_h = h; g.levels = levels; g.culls = culls;
    const int bw = w >> culls, bh = h >> culls;
    float Kb[9];
    cull_intrinsic(K, culls, Kb);
    g.px_total = 0;
    for (int i = 0; i < levels; i++) {
        const int t = levels - 1 - i;  // frame.cpp:33-35
        g.w[i] = bw >> t; g.h[i] = bh >> t;
        if (g.w[i] < 4 || g.h[i] < 4) {  // k_track_gn's lanes without an interior position gather around (1, 1): rows / columns 0..3
            set_error("image too small for this many pyramid levels (every level must be at least 4 x 4)");
            return DVO_ERR_BAD_ARGUMENT;
        }
        if ((size_t)g.w[i] * g.h[i] >= (1u << 24)) {
            set_error("pyramid level larger than 2^24 pixels");
            return DVO_ERR_BAD_ARGUMENT;
        }
        cull_intrinsic(Kb, t, g.K9[i]);
        g.k[i] = make_intr(g.K9[i]);
        g.px_total += (size_t)g.w[i] * g.h[i];
    }
    return DVO_OK;
}

static float level_step(const dvo_config& c, int level)
{  // optimize.cpp:22-26
    if (level == 1) return c.step_level1;
    if (level == 2) return c.step_level2;
    return c.step_default;
}

int FrameSet::alloc(const Geometry& geo, int n, const dvo_config& cfg, void* mem)
{
    g = geo;
    n_seq = n;
    if (mem) arena.adopt(mem, arena_bytes(g, n));
    else DVO_TRY(arena.alloc(arena_bytes(g, n)));
    float* p = arena.as<float>();
    for (int m = 0; m < 4; m++)
        for (int l = 0; l < g.levels; l++) {
            (m == 0 ? gray : m == 1 ? depth : m == 2 ? sigma : wgt)[l] = p;
            p += (size_t)g.w[l] * g.h[l] * n;
        }
    for (int l = 0; l < g.levels; l++) step[l] = level_step(cfg, l);
    sigma_min = cfg.sigma_min;
    sigma_max = cfg.sigma_max;
    allow_const_weight = cfg.min_depth > 0.0f && getenv("DVO_WEIGHT_MAPS") == nullptr;   // (the variable forces the maps: A/B runs, tests)
    return DVO_OK;
}

// wgt of every level from the current sigma pyramid (k_prep_ref)
static void prep_reference(FrameSet& fs, hipStream_t s)
{
    PrepArgs a;
    memset(&a, 0, sizeof a);
    a.depth = fs.depth[0]; a.sigma = fs.sigma[0]; a.wgt = fs.wgt[0];  // levels are contiguous
    size_t end = 0;
    for (int l = 0; l < fs.g.levels; l++) {
        end += (size_t)fs.g.w[l] * fs.g.h[l] * fs.n_seq;
        a.level_end[l] = end;
        a.step[l] = fs.step[l];
    }
    a.sigma_min = fs.sigma_min; a.sigma_max = fs.sigma_max;
    a.levels = fs.g.levels;
    launch_prep_ref(a, s);
}

static void fuse_prep(PyramidArgs& a, const FrameSet& fs)
{
    for (int l = 0; l < fs.g.levels; l++) {
        a.wgt[l] = fs.wgt[l];
        a.step[l] = fs.step[l];
    }
    a.sigma_min = fs.sigma_min; a.sigma_max = fs.sigma_max;
}

void build_pyramid(FrameSet& fs, const float* gray_dev, const float* depth_dev, const float* sigma_dev, hipStream_t s, bool keep_sigma,
                   bool rows_decimated)
{
    PyramidArgs a;
    memset(&a, 0, sizeof a);
    a.src[0] = gray_dev; a.src[1] = depth_dev; a.src[2] = sigma_dev;
    a.src_w = fs.g.src_w; a.src_h = fs.g.src_h; a.culls = fs.g.culls; a.levels = fs.g.levels;
    a.src_img_rows = rows_decimated ? fs.g.src_h >> fs.g.culls : fs.g.src_h;
    a.src_row_shift = rows_decimated ? 0 : fs.g.culls;
    for (int l = 0; l < fs.g.levels; l++) {
        a.w[l] = fs.g.w[l]; a.h[l] = fs.g.h[l];
        a.dst[0][l] = fs.gray[l]; a.dst[1][l] = fs.depth[l];
        a.dst[2][l] = (keep_sigma || !(depth_dev && sigma_dev)) ? fs.sigma[l] : nullptr;
    }
    a.inv_tw = 1.0f / (float)fs.g.w[fs.g.top()];
    fs.sigma_by_validity = false;
    if (depth_dev && sigma_dev) fuse_prep(a, fs);  // wgt written by the same launch (no k_prep_ref pass)
    launch_pyramid(a, fs.n_seq, s);
}

void build_pyramid(FrameSet& fs, const FrameInput& in, hipStream_t s, bool keep_sigma)
{
    if (!in.raw()) { build_pyramid(fs, in.gray, in.depth, in.sigma, s, keep_sigma, in.rows_decimated); return; }
    PyramidArgs a;
    memset(&a, 0, sizeof a);
    a.raw_rgb = in.rgb; a.raw_channels = in.channels; a.raw_depth = in.depth16;
    a.raw_gray_scale = (float)(1.0 / 255.0); a.raw_depth_scale = in.depth_scale;
    a.raw_sigma_valid = 0.1f; a.raw_sigma_invalid = 1.0f; a.raw_invalidate_gray = 1;   // transform.cpp:60-76
    a.src_w = fs.g.src_w; a.src_h = fs.g.src_h; a.culls = fs.g.culls; a.levels = fs.g.levels;
    a.src_img_rows = in.rows_decimated ? fs.g.src_h >> fs.g.culls : fs.g.src_h;
    a.src_row_shift = in.rows_decimated ? 0 : fs.g.culls;
    const bool dep = in.depth16 != nullptr;
    for (int l = 0; l < fs.g.levels; l++) {
        a.w[l] = fs.g.w[l]; a.h[l] = fs.g.h[l];
        a.dst[0][l] = fs.gray[l];
        a.dst[1][l] = dep ? fs.depth[l] : nullptr;
        a.dst[2][l] = (dep && keep_sigma) ? fs.sigma[l] : nullptr;
    }
    a.inv_tw = 1.0f / (float)fs.g.w[fs.g.top()];
    fs.sigma_by_validity = dep && !keep_sigma && fs.allow_const_weight;
    if (fs.sigma_by_validity) {  // no wgt maps: the weight of every pixel that can contribute is a constant of the level
        for (int l = 0; l < fs.g.levels; l++) fs.wgt_valid[l] = gn_weight(fs.step[l], fs.sigma_min, fs.sigma_max, a.raw_sigma_valid);
    } else if (dep) {
        fuse_prep(a, fs);
    }
    launch_pyramid(a, fs.n_seq, s);
}

int upload_rows(void* dst, const void* src, size_t row_bytes, int img_rows, size_t n_img, int culls, bool decimate, hipStream_t s,
                    size_t* stored)
{
    if (!decimate || culls <= 0) {
        const size_t n = row_bytes * (size_t)img_rows * n_img;
        if (stored) *stored = n;
        DVO_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s));
        return DVO_OK;
    }
    // img_rows is a multiple of 2^culls (can_decimate_rows), so the kept rows of ALL images are the rows r = 0 (mod 2^culls) of the
    // n_img * img_rows rows of the whole buffer: one 2-D copy with a source pitch of 2^culls rows
    const size_t rows = ((size_t)img_rows >> culls) * n_img;
    if (stored) *stored = row_bytes * rows;
    DVO_HIP(hipMemcpy2DAsync(dst, row_bytes, src, row_bytes << culls, row_bytes, rows, hipMemcpyHostToDevice, s));
    return DVO_OK;
}

void redecimate(FrameSet& fs, const float* depth_top, const float* sigma_top, hipStream_t s)
{  // level i = cullImage(top, levels-1-i); the top level itself is the map handed in (frame.cpp:39-61)
    PyramidArgs a;
    memset(&a, 0, sizeof a);
    const int T = fs.g.top();
    a.src[1] = depth_top; a.src[2] = sigma_top;
    a.src_w = fs.g.w[T]; a.src_h = fs.g.h[T]; a.culls = 0; a.levels = fs.g.levels;
    a.src_img_rows = a.src_h; a.src_row_shift = 0;
    fs.sigma_by_validity = false;
    for (int l = 0; l < fs.g.levels; l++) {
        a.w[l] = fs.g.w[l]; a.h[l] = fs.g.h[l];
        a.dst[1][l] = fs.depth[l]; a.dst[2][l] = fs.sigma[l];
    }
    a.inv_tw = 1.0f / (float)fs.g.w[T];
    if (depth_top && sigma_top) fuse_prep(a, fs);
    launch_pyramid(a, fs.n_seq, s);
    if (!(depth_top && sigma_top)) prep_reference(fs, s);  // one map only: the other comes from the stored pyramid
}

// ------------------------------------------------------------------------------------------------ tracker
Tracker::~Tracker()
{
    if (h_state) (void)hipHostFree(h_state);
    if (h_progress) (void)hipHostFree(h_progress);
    if (h_result) (void)hipHostFree(h_result);
    for (auto st : sub_streams) (void)hipStreamDestroy(st);
    for (auto e : ev_join) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (auto& e : ev_pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
}

int Tracker::init(const Geometry& geo, int n, const dvo_config& c)
{
    g = geo; n_seq = n; cfg = c;
    if (cfg.max_iterations < 1 || cfg.max_iterations > DVO_MAX_ITERATIONS || cfg.fixed_iterations > DVO_MAX_ITERATIONS) {
        set_error("iteration counts must be within [1, DVO_MAX_ITERATIONS]");
        return DVO_ERR_BAD_ARGUMENT;
    }
    for (int l = 0; l < g.levels; l++)
        if (g.w[l] < 4 || g.h[l] < 4) {  // k_track_gn's parked lanes gather the 4 x 4 taps around (1, 1): every level must hold them
            set_error("pyramid level smaller than 4 x 4 pixels");
            return DVO_ERR_BAD_ARGUMENT;
        }
    if ((unsigned long long)n_seq * (unsigned long long)g.w[g.top()] * g.h[g.top()] / 256ull >= (1ull << 31)) {
        set_error("too many sequences for one launch grid");
        return DVO_ERR_BAD_ARGUMENT;
    }
    size_t max_part = 0;
    // gn_use_lds_patch: -1 = auto, 0 = global gathers, N > 0 = LDS patch with margin N.  Auto is the global-gather
    // kernel: measured on MI355X (profiles/r01_gn_variants.md) the LDS-staged variant is 10-15 % slower.
    tile_margin = cfg.gn_use_lds_patch < 0 ? 0 : cfg.gn_use_lds_patch;
    if (tile_margin > 24) tile_margin = 24;
    for (int l = 0; l < g.levels; l++) {
        int p = cfg.gn_pixels_per_thread;
        const bool auto_p = (p != 1 && p != 2 && p != 4 && p != 8);
        if (auto_p) p = 4;  // auto: biggest tile that still gives >= 4 workgroups per CU
        fused[l] = false;
        if (tile_margin > 0) {
            gn_tile_geometry(g.w[l], g.h[l], p, tiles_x[l], tiles_y[l]);
            while (auto_p && p > 1 && (size_t)n_seq * tiles_x[l] * tiles_y[l] < 1024) {
                p >>= 1;
                gn_tile_geometry(g.w[l], g.h[l], p, tiles_x[l], tiles_y[l]);
            }
            nblk[l] = tiles_x[l] * tiles_y[l];
        } else {
            // small levels: every iteration inside one k_track_level launch (tiles of 256 x 4 pixels)
            // Off by default: measured on MI355X (512 sequences) the one-workgroup-per-sequence form runs the two coarse
            // levels in ~1.0 ms against ~0.6 ms for the batched launches -- 2 waves per SIMD cannot hide the latency chain
            // of a tile, while the batched kernels share the whole chip among the sequences that are still active.
            const int fuse_max = cfg.track_fused_tiles <= 0 ? 0
                                                            : (cfg.track_fused_tiles > DVO_FUSED_MAX_TILES ? DVO_FUSED_MAX_TILES : cfg.track_fused_tiles);
            const int crop_l = level_params(l).crop;
            const GnTiling t4 = gn_tiling(g.w[l], g.h[l], 4, crop_l);
            fused[l] = fuse_max > 0 && !t4.t2d && t4.count <= fuse_max;  // (k_track_level: raster tiles)
            if (fused[l]) p = 4;
            // (one sequence on the one-launch-per-call schedule keeps 4 pixels per thread: k_track_persist's workgroups wait for each
            //  other, and 75 of them hand over faster than 300 -- 406 against 451 us per 640x480 frame, profiles/r03_single_ab.txt)
            //  A mono handle's levels (at most 160 x 120) show no such difference -- 132-140 us per frame at 1, 2 and 4 pixels per
            //  thread -- so it keeps the tile size every other schedule picks for one sequence (persist_ppt < 0): the same bits.)
            const bool single_p4 = prefer_persist && n_seq == 1 && cfg.track_single_launch == 0 && !cfg.profile && persist_ppt >= 0;
            if (single_p4 && auto_p && persist_ppt > 0) p = persist_ppt;
            while (!fused[l] && auto_p && !single_p4 && p > 1 && (size_t)n_seq * gn_blocks_per_seq(g.w[l], g.h[l], p, crop_l) < 1024) p >>= 1;
            nblk[l] = gn_blocks_per_seq(g.w[l], g.h[l], p, crop_l);
            tiles_x[l] = tiles_y[l] = 0;
        }
        ppt[l] = p;
        int gg = cfg.gn_gather_group;
        if (gg != 1 && gg != 2 && gg != 4) gg = 2;
        while (gg > p || p % gg) gg >>= 1;
        group[l] = gg < 1 ? 1 : gg;
        if ((size_t)nblk[l] > max_part) max_part = nblk[l];
    }
    DVO_TRY(state.alloc(sizeof(SeqState) * (size_t)n_seq));
    DVO_TRY(partials.alloc(sizeof(float) * 32 * max_part * (size_t)n_seq));
    DVO_TRY(log.alloc(sizeof(dvo_track_log) * (size_t)n_seq));
    DVO_TRY(counters.alloc(2 * sizeof(unsigned long long)));
    // sub-batches on concurrent streams: only for the sync-free schedule of big batches (the small-batch poll needs one chain)
    n_sub = cfg.track_streams;
    if (n_sub <= 0) n_sub = 1;
    if (n_sub > 8) n_sub = 8;
    if (n_seq <= 8 || tile_margin > 0) n_sub = 1;
    while (n_sub > 1 && n_seq / n_sub < 8) n_sub--;
    for (int k = 1; k < n_sub; k++) {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        DVO_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        sub_streams.push_back(st);
        DVO_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ev_join.push_back(ev);
    }
    if (n_sub > 1) DVO_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    DVO_TRY(work.alloc(2 * sizeof(int) * (size_t)n_sub * (size_t)(n_seq + 4)));
    DVO_HIP(hipMemset(work.p, 0, work.bytes));
    // adaptive schedule (see dvo_engine.h): one launch chain only, global-gather kernel only, reference stop tests only
    adaptive = cfg.track_adaptive >= 0 && n_sub == 1 && tile_margin == 0 && cfg.fixed_iterations <= 0;
    if (adaptive) {
        const size_t words = 2 * (size_t)DVO_MAX_LEVELS * DVO_MAX_ITERATIONS;
        // fine-grained (coherent) host memory: the device publishes a word with a system-scope release store and the host
        // sees it without waiting for a kernel boundary (coarse-grained memory only guarantees that after a host sync)
        DVO_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_progress), words * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
        memset(h_progress, 0, words * sizeof(int));
        DVO_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_progress), h_progress, 0));
    }
    // one launch per iteration (k_track_gn_fused) for handles of a few sequences, per level: the kernel carries the solve's 248
    // VGPRs (two waves per SIMD), so only where the level's grid is a few dozen workgroups -- measured on one 640x480 stream:
    // 14-16 us per iteration against 17-18 us for the launch pair up to 52 workgroups, but 31 us against 18 us at 300
    for (int l = 0; l < g.levels; l++) {
        const GnTiling tl = gn_tiling(g.w[l], g.h[l], ppt[l], level_params(l).crop);
        single_launch[l] = cfg.track_single_launch >= 0 && n_seq <= 8 && tile_margin == 0 && !fused[l] && !cfg.profile && n_sub == 1 &&
                           gn_fused_available(ppt[l], group[l]) && tl.live_count > 0 && (long long)n_seq * tl.live_count <= 64;
    }
    // k_track_persist (one launch per track() call): one sequence, the global-gather kernel, one (ppt, group) pair on every level,
    // every level within the wide reduction's row limit, no profiling; track_single_launch: < 0 = launch pairs only, 1 = one launch per
    // iteration at most (k_track_gn_fused), 0 (default) = one launch per call where the result goes through enable_host_result()
    persist_ok = prefer_persist && n_seq == 1 && tile_margin == 0 && !cfg.profile && cfg.track_single_launch == 0 && n_sub == 1 && track_persist_available(ppt[0], group[0]);
    int max_tiles = 0;
    for (int l = 0; l < g.levels && persist_ok; l++) {
        const GnTiling tl = gn_tiling(g.w[l], g.h[l], ppt[l], level_params(l).crop);
        if (ppt[l] != ppt[0] || group[l] != group[0] || fused[l] || nblk[l] > 320 || tl.live_count <= 0) persist_ok = false;
        if (tl.live_count > max_tiles) max_tiles = tl.live_count;
    }
    // ~0.2 s of polling per wait: far beyond any iteration, short enough that a wedged launch ends.  DVO_PERSIST_SPIN_LIMIT (tests): a
    // limit of 0 makes every launch give up at once, so the fallback path runs; DVO_PERSIST_TIMELINE: in-kernel stamps (diagnostic).
    persist_spin_limit = 1 << 18;
    if (const char* e = getenv("DVO_PERSIST_SPIN_LIMIT")) persist_spin_limit = atoi(e);
    persist_timeline = getenv("DVO_PERSIST_TIMELINE") != nullptr;
    if (persist_ok) {
        int cap = 0;
        if (track_persist_max_grid(ppt[0], group[0], &cap) != DVO_OK || cap < 1) persist_ok = false;
        else {
            persist_grid = 1 + (max_tiles < cap - 1 ? max_tiles : cap - 1);   // the solver + one worker per tile of the largest level, all resident at once (they wait for each other)
            DVO_TRY(persist_ctl.alloc((16 + (size_t)persist_grid) * sizeof(int)));   // control line (64 B) + one arrival slot per workgroup
            DVO_HIP(hipMemset(persist_ctl.p, 0, persist_ctl.bytes));
        }
    }
    DVO_TRY(ticket.alloc(sizeof(int) * (size_t)n_seq));
    DVO_HIP(hipMemset(ticket.p, 0, ticket.bytes));
    DVO_TRY(freport.alloc(sizeof(int) * 2 * 2 * (size_t)DVO_MAX_LEVELS * DVO_MAX_ITERATIONS));
    DVO_HIP(hipMemset(freport.p, 0, freport.bytes));
    DVO_TRY(xi_out.alloc(sizeof(float) * 6 * (size_t)n_seq));
    DVO_TRY(T_out.alloc(sizeof(float) * 16 * (size_t)n_seq));
    DVO_HIP(hipMemset(counters.p, 0, 2 * sizeof(unsigned long long)));
    DVO_HIP(hipMemset(log.p, 0, log.bytes));
    return DVO_OK;
}

GnParams Tracker::level_params(int level) const
{
    GnParams p;
    p.step = cfg.step_default;  // optimize.cpp:22-26
    if (level == 1) p.step = cfg.step_level1;
    if (level == 2) p.step = cfg.step_level2;
    p.sigma_min = cfg.sigma_min; p.sigma_max = cfg.sigma_max;
    p.min_depth = cfg.min_depth;
    p.crop = (cfg.crop_enable && level == 2) ? 1 : 0;  // optimize.cpp:33-36
    return p;
}

GnArgs Tracker::gn_args(const FrameSet& obj, const FrameSet& ref, int level, uint8_t* mask, int ignore_active) const
{
    GnArgs a;
    a.obj_gray = obj.gray[level];
    a.ref_gray = ref.gray[level];
    a.ref_depth = ref.depth[level];
    a.ref_wgt = ref.sigma_by_validity ? nullptr : ref.wgt[level];
    a.wgt_const = ref.sigma_by_validity ? ref.wgt_valid[level] : 0.0f;
    a.state = state.as<SeqState>();
    a.partials = partials.as<float>();
    a.mask = mask;
    a.w = g.w[level]; a.h = g.h[level]; a.nblk = nblk[level];
    a.inv_w = 1.0f / (float)g.w[level];
    a.q256 = 256 / g.w[level]; a.r256 = 256 % g.w[level];
    a.k = g.k[level];
    a.prm = level_params(level);
    a.ignore_active = ignore_active;
    a.tiles_x = tiles_x[level]; a.tiles_y = tiles_y[level]; a.margin = tile_margin;
    return a;
}

void Tracker::launch_gn(const GnArgs& a, int level, int count, hipStream_t s, int grid_seqs) const
{
    if (tile_margin > 0) launch_track_gn_tile(a, count, ppt[level], s);
    else launch_track_gn(a, count, ppt[level], group[level], s, grid_seqs);
}

int Tracker::track(const FrameSet& obj, const FrameSet& ref, hipStream_t s)
{
    last_obj = &obj; last_ref = &ref;
    persist_used = false;
    if (persist_ok && !persist_failed && h_result) {   // the whole call in one launch (k_track_persist)
        PersistArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.levels = g.levels;
        for (int l = 0; l < g.levels; l++) {
            const GnArgs ga = gn_args(obj, ref, l, nullptr, 0);
            const GnTiling tl = gn_tiling(ga.w, ga.h, ppt[l], ga.prm.crop);
            PersistLevel& L = pa.lv[l];
            L.obj_gray = ga.obj_gray; L.ref_gray = ga.ref_gray; L.ref_depth = ga.ref_depth; L.ref_wgt = ga.ref_wgt; L.wgt_const = ga.wgt_const;
            L.inv_w = ga.inv_w; L.w = ga.w; L.h = ga.h; L.nblk = ga.nblk; L.q256 = ga.q256; L.r256 = ga.r256; L.k = ga.k; L.prm = ga.prm;
            L.blk_first = tl.live_first; L.blk_count = tl.live_count; L.t_shift = tl.shift; L.x_org = tl.x_org; L.y_org = tl.y_org;
            L.tiles_x = tl.t2d ? tl.tiles_x : 0; L.t2d = tl.t2d; L.level_pixels = (int)tl.live_pixels;
        }
        pa.state = state.as<SeqState>(); pa.partials = partials.as<float>(); pa.log = log.as<dvo_track_log>();
        pa.ctl = persist_ctl.as<int>();
        pa.max_iterations = cfg.max_iterations; pa.fixed_iterations = cfg.fixed_iterations;
        pa.min_update = cfg.min_update; pa.min_residual = cfg.min_residual;
        pa.xi_out = xi_out.as<float>(); pa.T_out = T_out.as<float>(); pa.host_result = d_result;
        result_tag = (result_tag + 1) & 0x1fffff;
        if (result_tag == 0) result_tag = 1;
        pa.host_tag = result_tag;
        pa.spin_limit = persist_spin_limit;
        pa.mono = mono_tail;
        if (persist_timeline) {   // diagnostic: stamps of the solver and of worker 0 (tools/persist_timeline.py reads them back)
            if (!persist_dbg.p) { DVO_TRY(persist_dbg.alloc(2 * 64 * 8 * sizeof(long long))); }
            DVO_HIP(hipMemsetAsync(persist_dbg.p, 0, persist_dbg.bytes, s));
            pa.dbg = persist_dbg.as<long long>();
            pa.dbg_worker = atoi(getenv("DVO_PERSIST_TIMELINE"));
        }
        if (launch_track_persist(pa, ppt[0], group[0], persist_grid, s)) {
            persist_used = true;
            DVO_HIP(hipGetLastError());
            return DVO_OK;
        }
    }
    launch_track_begin(state.as<SeqState>(), log.as<dvo_track_log>(), n_seq, g.levels, s);
    const int max_it = cfg.fixed_iterations > 0 ? cfg.fixed_iterations : cfg.max_iterations;
    // Small batches: every few iterations ask the device whether anything is still active, so a converged
    // level does not pay for its remaining (empty) launches.  Big batches run the fixed schedule sync-free.
    const bool poll = (cfg.fixed_iterations <= 0) && n_seq <= 8 && !adaptive;
    if (poll && !h_state) DVO_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_state), sizeof(SeqState) * (size_t)n_seq, hipHostMallocDefault));
    SeqState* host_state = h_state;  // pinned: the read-back is one async copy + one stream sync, no staging
    // fork: the sub-batch streams start once everything queued on `s` so far (pyramids, k_track_begin) is done
    const int subs = poll ? 1 : n_sub;
    if (subs > 1) {
        DVO_HIP(hipEventRecord(ev_fork, s));
        for (int k = 1; k < subs; k++) DVO_HIP(hipStreamWaitEvent(sub_streams[k - 1], ev_fork, 0));
    }
    // adaptive: this call's progress words (the other set may still be written by the tail of the previous call)
    volatile int* prog_h = nullptr;
    int* prog_d = nullptr;
    if (adaptive) {
        progress_set ^= 1;
        const size_t off = (size_t)progress_set * DVO_MAX_LEVELS * DVO_MAX_ITERATIONS;
        prog_h = h_progress + off; prog_d = d_progress + off;
        for (int i = 0; i < DVO_MAX_LEVELS * DVO_MAX_ITERATIONS; i++) prog_h[i] = 0;
    }
    bool any_single = false;
    for (int l = 0; l < g.levels; l++) any_single = any_single || single_launch[l];
    int* rep_set = freport.as<int>() + (size_t)(progress_set & 1) * 2 * DVO_MAX_LEVELS * DVO_MAX_ITERATIONS;
    if (any_single) DVO_HIP(hipMemsetAsync(rep_set, 0, sizeof(int) * 2 * DVO_MAX_LEVELS * DVO_MAX_ITERATIONS, s));
    static const char* const kLevelName[DVO_MAX_LEVELS] = {"track level 0", "track level 1", "track level 2", "track level 3", "track level 4",
                                                           "track level 5", "track level 6", "track level 7"};
    for (int level = 0; level < g.levels; level++) {  // tracker.cpp:32
        TraceRange tr(kLevelName[level]);
        const bool lists = tile_margin == 0 && !single_launch[level];  // (k_track_gn_tile keeps the per-sequence active flag test)
        const size_t level_px = (size_t)g.w[level] * g.h[level];
        const int host_its = fused[level] ? 1 : max_it;  // a fused level iterates on the device (k_track_level)
        for (int it = 0; it < host_its; it++) {        // tracker.cpp:42
            const int first = (it == 0) ? 1 : 0;
            // Stay `ahead` iterations ahead of the GPU: wait until launch it-ahead of this level has reported, and stop the level
            // if it had no active sequence -- every later launch of the level would be empty.  Skipping empty launches
            // changes no result.  Two iterations ahead for every batch size: a batch iteration takes >= 20 us on the GPU, the host
            // needs ~10 us to see a progress word and queue the next pair, and every iteration queued beyond the last useful one
            // is an empty launch pair (~15 us): measured +1.6 % (mono, 1 iteration per level) / +0.7 % (sensor depth) against 4.
            const int ahead = 2;
            int active_ub = 0;   // sequences that entered iteration it - ahead: the active set only shrinks within a level, so this bounds the list of `it`
            if (adaptive && !fused[level] && it >= ahead) {
                volatile int* pw = prog_h + level * DVO_MAX_ITERATIONS + (it - ahead);
                long spins = 0;
                while (*pw == 0) {
                    if (++spins > 2000000000L) {
                        (void)hipStreamSynchronize(s);  // nothing may be left writing the progress words / state when we return
                        set_error("adaptive schedule: the GPU made no progress");
                        return DVO_ERR_HIP;
                    }
                    __builtin_ia32_pause();
                }
                if (*pw - 1 == 0) break;
                if (!single_launch[level]) active_ub = *pw - 1;
            }
            const GnArgs ga0 = gn_args(obj, ref, level, nullptr, first);
            for (int k = 0; k < subs; k++) {  // launches of the sub-batches interleave on their streams
                const int q0 = subs > 1 ? sub_first(k) : 0, q1 = subs > 1 ? sub_first(k + 1) : n_seq, nq = q1 - q0;
                hipStream_t sk = k == 0 ? s : sub_streams[k - 1];
                GnArgs ga = ga0;  // view of sequences [q0, q1)
                ga.obj_gray += q0 * level_px; ga.ref_gray += q0 * level_px; ga.ref_depth += q0 * level_px;
                if (ga.ref_wgt) ga.ref_wgt += q0 * level_px;
                ga.state += q0;
                ga.partials += (size_t)q0 * nblk[level] * 32;
                if (single_launch[level]) {   // GN accumulation + solve of this iteration in one launch (k_track_gn_fused)
                    SolveArgs fa;
                    fa.state = state.as<SeqState>() + q0;
                    fa.partials = ga.partials;
                    fa.log = log.as<dvo_track_log>() + q0;
                    fa.result = nullptr;
                    fa.counters = nullptr;
                    fa.nblk = nblk[level]; fa.level = level; fa.level_pixels = (int)level_px;
                    fa.max_iterations = cfg.max_iterations; fa.fixed_iterations = cfg.fixed_iterations;
                    fa.min_update = cfg.min_update; fa.min_residual = cfg.min_residual;
                    fa.ignore_active = first;
                    if (launch_track_gn_fused(ga, fa, nq, ppt[level], group[level], ticket.as<int>(), rep_set + 2 * (level * DVO_MAX_ITERATIONS + it),
                                              adaptive ? prog_d + level * DVO_MAX_ITERATIONS + it : nullptr, sk))
                        continue;
                }
                if (fused[level]) {
                    SolveArgs fa;
                    fa.state = state.as<SeqState>() + q0;
                    fa.partials = nullptr;
                    fa.log = log.as<dvo_track_log>() + q0;
                    fa.result = nullptr;
                    fa.counters = nullptr;  // (the profile counters describe k_track_gn launches only)
                    fa.nblk = nblk[level]; fa.level = level; fa.level_pixels = (int)level_px;
                    fa.max_iterations = cfg.max_iterations; fa.fixed_iterations = cfg.fixed_iterations;
                    fa.min_update = cfg.min_update; fa.min_residual = cfg.min_residual;
                    fa.ignore_active = 1;
                    launch_track_level(ga, fa, nq, sk);
                    continue;
                }
                // iteration `it` evaluates the sequences k_gn_solve(it - 1) left active (all of them when it == 0) and
                // clears the list k_gn_solve(it) appends to
                const int* list_prev = (first || !lists) ? nullptr : work_list(k, it - 1);
                ga.list = list_prev;
                ga.next_count = lists ? work_list(k, it) : nullptr;
                if (cfg.profile) {
                    if (ev_used == ev_pool.size()) {
                        hipEvent_t e0, e1;
                        DVO_HIP(hipEventCreate(&e0));
                        DVO_HIP(hipEventCreate(&e1));
                        ev_pool.emplace_back(e0, e1);
                    }
                    DVO_HIP(hipEventRecord(ev_pool[ev_used].first, sk));
                    launch_gn(ga, level, nq, sk, active_ub);
                    DVO_HIP(hipEventRecord(ev_pool[ev_used].second, sk));
                    ev_used++;
                } else {
                    launch_gn(ga, level, nq, sk, active_ub);
                }
                SolveArgs sa;
                sa.state = state.as<SeqState>() + q0;
                sa.partials = ga.partials;
                sa.log = log.as<dvo_track_log>() + q0;
                sa.result = nullptr;
                sa.counters = cfg.profile ? counters.as<unsigned long long>() : nullptr;
                sa.nblk = nblk[level]; sa.level = level; sa.level_pixels = (int)level_px;
                sa.max_iterations = cfg.max_iterations; sa.fixed_iterations = cfg.fixed_iterations;
                sa.min_update = cfg.min_update; sa.min_residual = cfg.min_residual;
                sa.ignore_active = first;
                sa.list_in = list_prev;
                sa.list_out = lists ? work_list(k, it) : nullptr;
                if (adaptive) sa.progress = prog_d + level * DVO_MAX_ITERATIONS + it;
                if (lists) {
                    // (profile counter: the pixels k_track_gn actually reads -- tiles outside the crop rows are never launched)
                    const GnTiling tl = gn_tiling(ga.w, ga.h, ppt[level], ga.prm.crop);
                    sa.blk_first = tl.live_first; sa.blk_count = tl.live_count;
                    const long long live_px = tl.live_pixels;
                    sa.level_pixels = (int)live_px;
                }
                launch_gn_solve(sa, nq, sk);
            }
            if (poll && !fused[level] && it + 1 < max_it) {
                DVO_HIP(hipMemcpyAsync(host_state, state.p, sizeof(SeqState) * (size_t)n_seq, hipMemcpyDeviceToHost, s));
                DVO_HIP(hipStreamSynchronize(s));
                bool any = false;
                for (int q = 0; q < n_seq; q++) any = any || host_state[q].active != 0;
                if (!any) break;
            }
        }
    }
    // join: `s` continues (pose export, the caller's next frame) only after every sub-batch chain has finished
    for (int k = 1; k < subs; k++) {
        DVO_HIP(hipEventRecord(ev_join[k - 1], sub_streams[k - 1]));
        DVO_HIP(hipStreamWaitEvent(s, ev_join[k - 1], 0));
    }
    if (h_result) { result_tag = (result_tag + 1) & 0x1fffff; if (result_tag == 0) result_tag = 1; }
    launch_export_poses(state.as<SeqState>(), xi_out.as<float>(), T_out.as<float>(), n_seq, s, d_result, result_tag);
    DVO_HIP(hipGetLastError());
    return DVO_OK;
}

int Tracker::read_persist_timeline(long long* out)   // [2][64][8]
{
    if (!persist_dbg.p) return DVO_ERR_NOT_READY;
    DVO_HIP(hipMemcpy(out, persist_dbg.p, persist_dbg.bytes, hipMemcpyDeviceToHost));
    return DVO_OK;
}

int Tracker::enable_host_result()
{
    if (h_result) return DVO_OK;
    DVO_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_result), 64 * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
    memset(h_result, 0, 64 * sizeof(float));
    DVO_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_result), h_result, 0));
    return DVO_OK;
}

int Tracker::wait_host_result(hipStream_t s, float xi[6], float T[16])
{
    if (!h_result) { set_error("host result not enabled"); return DVO_ERR_NOT_READY; }
    volatile int* tag = reinterpret_cast<volatile int*>(h_result + 22);
    volatile int* gave_up = reinterpret_cast<volatile int*>(h_result + 23);
    long spins = 0;
    while (*tag != result_tag) {
        __builtin_ia32_pause();
        if (persist_used && *gave_up == result_tag) {
            // k_track_persist ran into its polling limit (its workgroups were not all resident: the GPU is shared with something
            // that fills it).  Let it drain, reset its words and run this frame -- and every later one -- launch by launch.
            DVO_HIP(hipStreamSynchronize(s));
            DVO_HIP(hipMemset(persist_ctl.p, 0, persist_ctl.bytes));
            persist_failed = true;
            if (!last_obj || !last_ref) { set_error("k_track_persist gave up and the frame sets are gone"); return DVO_ERR_HIP; }
            DVO_TRY(track(*last_obj, *last_ref, s));
            spins = 0;
            continue;
        }
        if ((++spins & 0xfffff) == 0) {   // every ~1 M polls: has the stream finished (or failed) without the tag appearing?
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) {
                if (*tag == result_tag) break;
                set_error("track(): the stream drained but the result tag never arrived");
                return DVO_ERR_HIP;
            }
            if (q != hipErrorNotReady) { set_error(hipGetErrorString(q)); return DVO_ERR_HIP; }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int i = 0; i < 6; i++) xi[i] = h_result[i];
    for (int i = 0; i < 16; i++) T[i] = h_result[6 + i];
    return DVO_OK;
}

int Tracker::collect_profile(hipStream_t s)
{
    DVO_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < ev_used; i++) {
        float ms = 0;
        DVO_HIP(hipEventElapsedTime(&ms, ev_pool[i].first, ev_pool[i].second));
        prof_ms += ms;
        prof_launches++;
    }
    ev_used = 0;
    return DVO_OK;
}

// ------------------------------------------------------------------------------------------------ keyframes
int Keyframe::alloc(const Geometry& g, const dvo_config& cfg, KeyframePool* from)
{
    const size_t top = sizeof(float) * (size_t)g.w[g.top()] * g.h[g.top()];
    if (from) {   // one block of the pool: [arena | age | depth_alt], each 256-byte aligned
        const size_t a0 = (FrameSet::arena_bytes(g, 1) + 255) & ~(size_t)255, a1 = (top + 255) & ~(size_t)255;
        void* blk = nullptr;
        DVO_TRY(from->take(a0 + 2 * a1, &blk));
        pool = from; block = blk;
        DVO_TRY(fs.alloc(g, 1, cfg, blk));
        age.adopt(static_cast<char*>(blk) + a0, top);
        depth_alt.adopt(static_cast<char*>(blk) + a0 + a1, top);
    } else {
        DVO_TRY(fs.alloc(g, 1, cfg));
        DVO_TRY(age.alloc(top));
        DVO_TRY(depth_alt.alloc(top));
    }
    depth_spare = depth_alt.as<float>();
    return DVO_OK;
}

// ------------------------------------------------------------------------------------------------ VisualOdometry
int VisualOdometry::fetch_valid_updates()
{
    if (!valid_updates_pending) return DVO_OK;
    DVO_TRY(select_device(device));
    DVO_HIP(hipMemcpyAsync(&last_valid_updates, valid_dev.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    DVO_HIP(hipStreamSynchronize(stream));
    valid_updates_pending = false;
    return DVO_OK;
}

int VisualOdometry::fetch_log()
{
    if (!log_src) return DVO_OK;
    DVO_TRY(select_device(device));
    DVO_HIP(hipMemcpyAsync(&last_log, log_src->log.p, sizeof last_log, hipMemcpyDeviceToHost, stream));
    DVO_HIP(hipStreamSynchronize(stream));
    log_src = nullptr;
    return DVO_OK;
}

int VisualOdometry::upload_streams()
{
    if (ustream[0]) return DVO_OK;
    for (int i = 0; i < 2; i++) DVO_HIP(hipStreamCreateWithFlags(&ustream[i], hipStreamNonBlocking));
    for (int i = 0; i < 3; i++) DVO_HIP(hipEventCreateWithFlags(&uevent[i], hipEventDisableTiming));   // [2]: the mono frame's upload
    return DVO_OK;
}

VisualOdometry::~VisualOdometry()
{
    for (int i = 0; i < 2; i++) {
        if (ustream[i]) { (void)hipStreamSynchronize(ustream[i]); (void)hipStreamDestroy(ustream[i]); }
    }
    for (int i = 0; i < 3; i++)
        if (uevent[i]) (void)hipEventDestroy(uevent[i]);
    if (h_pin) (void)hipHostFree(h_pin);
    if (h_stage) { if (stream) (void)hipStreamSynchronize(stream); (void)hipHostFree(h_stage); }
    if (h_tables) { if (stream) (void)hipStreamSynchronize(stream); (void)hipHostFree(h_tables); }
    if (own_stream && stream) (void)hipStreamDestroy(stream);
}

int VisualOdometry::init(const float K9[9], int width, int height, const dvo_config* c)
{
    if (!K9 || width < 64 || height < 64) {
        set_error("dvo_vo_create: bad K or frame size");
        return DVO_ERR_BAD_ARGUMENT;
    }
    if (c) cfg = *c; else dvo_config_default(&cfg);
    memcpy(K, K9, sizeof K);
    w = width; h = height; device = cfg.device;
    DVO_TRY(select_device(device));
    if (cfg.stream) stream = (hipStream_t)cfg.stream;
    else { DVO_HIP(hipStreamCreate(&stream)); own_stream = true; }
    stage_mono_rows = getenv("DVO_MONO_STAGE") == nullptr || atoi(getenv("DVO_MONO_STAGE")) != 0;   // (read per handle: tests compare both paths)
    stage_raw_rows = getenv("DVO_RAW_STAGE") == nullptr || atoi(getenv("DVO_RAW_STAGE")) != 0;
    DVO_TRY(make_geometry(K, w, h, 3, 2, geoM));  // system.hpp:47
    DVO_TRY(make_geometry(K, w, h, 4, 1, geoD));  // system.hpp:82
    const size_t n = (size_t)w * h * sizeof(float);
    DVO_TRY(in_gray.alloc(n)); DVO_TRY(in_depth.alloc(n)); DVO_TRY(in_sigma.alloc(n));
    const size_t tn = (size_t)geoM.w[2] * geoM.h[2];
    DVO_TRY(tmp_a.alloc(tn * 4)); DVO_TRY(tmp_b.alloc(tn * 4)); DVO_TRY(tmp_c.alloc(tn * 4));
    DVO_TRY(owner.alloc(tn * 4));
    DVO_TRY(valid_dev.alloc(sizeof(int)));
    DVO_TRY(meta_dev.alloc(sizeof(MonoSeq)));
    DVO_HIP(hipMemset(meta_dev.p, 0, sizeof(MonoSeq)));
    DVO_HIP(hipHostMalloc(&h_pin, sizeof(MonoSeq) + sizeof(dvo_track_log), hipHostMallocDefault));  // pinned: the per-frame read-back is one DMA
    memset(&h_meta, 0, sizeof h_meta);
    memset(&last_log, 0, sizeof last_log);
    return DVO_OK;
}

void default_initial_depth(int n, uint32_t seed, std::vector<float>& d, std::vector<float>& s)
{  // stands in for cv::randn(depth, 1.5, 0.5); max(depth, 0.5); sigma = 0.5 (frame.hpp:17-21), deviation D6
    d.resize(n); s.assign(n, 0.5f);
    for (int i = 0; i < n; i++) {
        const uint32_t a = mix32(seed ^ (uint32_t)(2 * i + 1) * 0x9E3779B9U), b = mix32(a ^ 0x85EBCA6BU);
        const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777217.0f), u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
        const float z = std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2);
        const float v = 1.5f + 0.5f * z;
        d[i] = v < 0.5f ? 0.5f : v;
    }
}

int VisualOdometry::init_keyframe(const float* gray, const float* depth, const float* sigma)
{  // system.hpp:24-32 with the mono geometry (deviation D9)
    if (!gray || !depth || !sigma) { set_error("null image"); return DVO_ERR_BAD_ARGUMENT; }
    DVO_TRY(select_device(device));
    const size_t n = (size_t)w * h * sizeof(float);
    DVO_HIP(hipMemcpyAsync(in_gray.p, gray, n, hipMemcpyHostToDevice, stream));
    DVO_HIP(hipMemcpyAsync(in_depth.p, depth, n, hipMemcpyHostToDevice, stream));
    DVO_HIP(hipMemcpyAsync(in_sigma.p, sigma, n, hipMemcpyHostToDevice, stream));
    auto kf = std::make_unique<Keyframe>();
    DVO_TRY(kf->alloc(geoM, cfg, &kf_pool));
    kf->id = ++latest_id;
    build_pyramid(kf->fs, in_gray.as<float>(), in_depth.as<float>(), in_sigma.as<float>(), stream);
    DVO_HIP(hipMemsetAsync(kf->age.p, 0, kf->age.bytes, stream));
    DVO_HIP(hipStreamSynchronize(stream));
    hist.push_back(std::move(kf));
    hist_version++;
    return DVO_OK;
}

int VisualOdometry::map_propagate(Keyframe& frame, const Keyframe& ref)
{  // Mapper::propagate, mapper.cpp:62-74; the pose exp(+rel_xi) is the one k_mono_decide left in meta_dev
    const int T = geoM.top(), tw = geoM.w[T], th = geoM.h[T];
    PropArgs a;
    a.ref_depth = ref.fs.depth[T]; a.ref_sigma = ref.fs.sigma[T]; a.ref_age = ref.age.as<float>();
    a.depth = frame.fs.depth[T]; a.sigma = frame.fs.sigma[T]; a.age = frame.age.as<float>();
    a.owner = owner.as<int>();
    a.w = tw; a.h = th; a.n_seq = 1; a.k = geoM.k[T];
    a.meta = meta_dev.as<MonoSeq>();
    memset(&a.pose, 0, sizeof a.pose); a.tz = 0.0f;
    launch_propagate_batch(a, stream);
    // (Frame::updateDepthSigmaAge, frame.cpp:47-54, re-decimates both maps here; Mapper::regularize and Frame::updateDepth follow at
    //  once, mapper.cpp:26, and decimate the depth again: map_regularize() derives every level of both pyramids in its one pass)
    return DVO_OK;
}

static const size_t kStageLimitBytes = 512 * 1024;

int VisualOdometry::alloc_stage()
{  // pinned, device-mapped: [colour / gray rows, up to 4 bytes per pixel][16-bit depth rows]
    if (h_stage) return DVO_OK;
    DVO_HIP(hipHostMalloc(&h_stage, (size_t)w * h * 6, hipHostMallocMapped));
    DVO_HIP(hipHostGetDevicePointer(&d_stage, h_stage, 0));
    return DVO_OK;
}

static void stage_rows_host(void* dst, const void* src, size_t row_bytes, int rows, int culls, bool decimate)
{  // the rows the pyramid keeps (every 2^culls-th) -- or all of them -- packed
    if (!decimate || culls <= 0) { memcpy(dst, src, row_bytes * (size_t)rows); return; }
    const int kept = rows >> culls;
    for (int r = 0; r < kept; r++)
        memcpy(static_cast<char*>(dst) + (size_t)r * row_bytes, static_cast<const char*>(src) + ((size_t)r << culls) * row_bytes, row_bytes);
}

int VisualOdometry::refresh_history_tables()
{  // device copies of FrameHistory's poses and top-level gray pointers (+ room for the age table): current after this call
    const int T = geoM.top();
    const int n_hist = (int)hist.size();
    if (hist_table_version == hist_version && hist_table_n == n_hist) return DVO_OK;
    // Staged in PINNED host memory, copied without a synchronisation: the staging block is rewritten only by a later call of this
    // function, i.e. in a later frame, and every frame waits for its tracking result, which is stream-ordered after these copies.
    const size_t xi_bytes = sizeof(float) * 6 * (size_t)n_hist, gt_bytes = sizeof(float*) * (size_t)n_hist;
    if (h_tables_bytes < xi_bytes + gt_bytes) {
        DVO_HIP(hipStreamSynchronize(stream));   // (the old tables / staging block may still be read by queued work)
        if (h_tables) DVO_HIP(hipHostFree(h_tables));
        h_tables = nullptr;
        h_tables_bytes = 2 * (xi_bytes + gt_bytes);
        DVO_HIP(hipHostMalloc(&h_tables, h_tables_bytes, hipHostMallocDefault));
        DVO_TRY(ages.alloc(sizeof(AgeEntry) * (size_t)n_hist * 2));
        DVO_TRY(hist_xi_dev.alloc(2 * xi_bytes));
        DVO_TRY(gray_tab_dev.alloc(2 * gt_bytes));
    }
    float* hx = static_cast<float*>(h_tables);
    const float** gt = reinterpret_cast<const float**>(static_cast<char*>(h_tables) + xi_bytes);   // (xi_bytes is a multiple of 8)
    for (int i = 0; i < n_hist; i++) {
        memcpy(hx + (size_t)i * 6, hist[i]->xi, 6 * sizeof(float));
        gt[i] = hist[i]->fs.gray[T];
    }
    DVO_HIP(hipMemcpyAsync(hist_xi_dev.p, hx, xi_bytes, hipMemcpyHostToDevice, stream));
    DVO_HIP(hipMemcpyAsync(gray_tab_dev.p, gt, gt_bytes, hipMemcpyHostToDevice, stream));
    hist_table_n = n_hist; hist_table_version = hist_version;
    return DVO_OK;
}

int VisualOdometry::map_update(Keyframe& obj)
{  // Mapper::update, mapper.cpp:76-137
    Keyframe& ref = *hist.back();
    const int T = geoM.top(), tw = geoM.w[T], th = geoM.h[T];
    const int n_hist = (int)hist.size();
    // mapper.cpp:107: r_xi = concatenate(obj.xi, -born.xi), once per keyframe, on the device (k_age_table).  The keyframes' poses and
    // top-level gray pointers only change when FrameHistory does (a keyframe pushed, dropped or loaded): the device copies are
    // refreshed then (hist_version), not on every frame -- two uploads and a stream synchronisation less per tracked frame.
    DVO_TRY(refresh_history_tables());
    AgeTableArgs ta;
    ta.meta = meta_dev.as<MonoSeq>(); ta.hist_xi = hist_xi_dev.as<float>(); ta.ages = ages.as<AgeEntry>();
    ta.n_seq = 1; ta.R = n_hist; ta.n_hist = n_hist;
    ta.zero_word = valid_dev.as<int>();   // mapper.cpp:136's count of this update (read back when dvo_vo_last_valid_updates asks)
    if (!age_table_done) launch_age_table(ta, stream);   // (done: the tail of k_track_persist computed it, odometrize())
    age_table_done = false;
    UpdateArgs a;
    memset(&a, 0, sizeof a);
    a.ref_depth = ref.fs.depth[T]; a.ref_sigma = ref.fs.sigma[T]; a.ref_age = ref.age.as<float>();
    a.obj_gray = obj.fs.gray[T];
    a.ages = ages.as<AgeEntry>();
    a.ring_gray = nullptr;
    a.gray_table = gray_tab_dev.as<const float*>();
    a.meta = meta_dev.as<MonoSeq>();
    a.n_seq = 1; a.R = n_hist; a.n_hist = n_hist; a.w = tw; a.h = th; a.crop = cfg.crop_enable; a.obj_id = obj.id;
    a.clamp_age = history_limit > 0 ? 1 : 0;
    a.seed = cfg.rng_seed;
    a.k = geoM.k[T];
    memcpy(a.K9, geoM.K9[T], sizeof a.K9);
    a.valid_updates = valid_dev.as<int>();
    launch_depth_update(a, stream);
    valid_updates_pending = true;
    // (mapper.cpp:135's Frame::updateDepthSigma: folded into map_regularize(), as in map_propagate())
    return DVO_OK;
}

int VisualOdometry::map_regularize(Keyframe& kf)
{  // Mapper::regularize (mapper.cpp:139-144) + Frame::updateDepth (frame.cpp:56-61), together with the re-decimation of sigma that the
   // preceding propagate / update left pending (frame.cpp:39-54): every level of depth and sigma is a decimation of the top maps, so one
   // pass (k_regularize_redecimate, the batched pipeline's kernel) leaves the values the reference's three re-decimations leave
    const int T = geoM.top();
    RegDecArgs ra;
    memset(&ra, 0, sizeof ra);
    ra.depth = kf.fs.depth[T]; ra.sigma = kf.fs.sigma[T];
    ra.depth_top_out = kf.depth_spare;
    for (int l = 0; l < geoM.levels; l++) {
        ra.w[l] = geoM.w[l]; ra.h[l] = geoM.h[l];
        ra.depth_lv[l] = kf.fs.depth[l]; ra.sigma_lv[l] = kf.fs.sigma[l]; ra.wgt[l] = kf.fs.wgt[l];
        ra.step[l] = kf.fs.step[l];
    }
    ra.levels = geoM.levels; ra.n_seq = 1; ra.sigma_min = kf.fs.sigma_min; ra.sigma_max = kf.fs.sigma_max;
    kf.fs.sigma_by_validity = false;
    launch_regularize_redecimate(ra, stream);
    std::swap(kf.fs.depth[T], kf.depth_spare);   // the top-level depth map alternates between the arena block and depth_alt
    return DVO_OK;
}

int VisualOdometry::odometrize(const float* gray, float T_world[16], int* is_key, const uint8_t* raw, int raw_channels)
{  // system.hpp:44-74; `raw` != nullptr: the frame arrives as u8 gray / RGB(A) and is converted while the pyramid is built
    if ((!gray && !raw) || !T_world) { set_error("null argument"); return DVO_ERR_BAD_ARGUMENT; }
    if (raw && raw_channels != 1 && raw_channels != 3 && raw_channels != 4) { set_error("bad channel count"); return DVO_ERR_BAD_ARGUMENT; }
    DVO_TRY(select_device(device));
    if (!trkM_ready) {   // one launch per track() call (k_track_persist), as the sensor-depth tracker
        trkM.prefer_persist = true;
        if (const char* e = getenv("DVO_MONO_PERSIST")) trkM.prefer_persist = atoi(e) != 0;
        trkM.persist_ppt = -1;   // the tile size of the other schedules
        if (const char* e = getenv("DVO_MONO_PERSIST_PPT")) trkM.persist_ppt = atoi(e);
        DVO_TRY(trkM.init(geoM, 1, cfg));
        if (trkM.persist_ok) DVO_TRY(trkM.enable_host_result());
        trkM_ready = true;
    }
    // The frame goes up on a stream of its own: the previous call returned with its mapping kernels (update / propagate, regularize)
    // still queued on `stream`, and a copy queued behind them would wait for them although it touches nothing they do -- the staging
    // buffers were last read by the previous frame's pyramid, which that call waited for (the pose read-back).  The pyramid waits
    // for the copy (event), then everything is in stream order again.
    // (r3, later) ... or no copy at all: the mono pyramid keeps one row in four (cull 2), 77-307 KB of the frame.  The caller's thread
    // copies those rows into a pinned, device-mapped staging block -- what the runtime's own path for pageable memory starts with -- and
    // k_pyramid reads them from there over the host link; the runtime's copy (API + DMA + 22-37 us until the dependent kernel starts,
    // profiles/r03_mono_single_trace_final.txt) drops out.  The block was last read by the previous frame's pyramid, which that call waited
    // for.  DVO_MONO_STAGE=0: the copy on the side stream, as before.
    FrameInput fin;
    fin.rows_decimated = decimate_host_rows && can_decimate_rows(geoM);   // only the rows the pyramid keeps are copied / cross PCIe
    const size_t row_bytes = raw ? (size_t)w * raw_channels : (size_t)w * sizeof(float);
    // (a thread's memcpy beats the runtime's copy path up to about half a megabyte -- measured at 77-460 KB; above that the DMA engine wins)
    const bool stage_rows = stage_mono_rows && row_bytes * (size_t)(fin.rows_decimated ? h >> geoM.culls : h) <= kStageLimitBytes;
    if (stage_rows) {
        if (!h_stage) {
            DVO_TRY(alloc_stage());
        }
        stage_rows_host(h_stage, raw ? static_cast<const void*>(raw) : static_cast<const void*>(gray), row_bytes, h, geoM.culls, fin.rows_decimated);
        if (raw) { fin.rgb = static_cast<const uint8_t*>(d_stage); fin.channels = raw_channels; }
        else fin.gray = static_cast<const float*>(d_stage);
    } else {
        DVO_TRY(upload_streams());
        hipStream_t up = ustream[1];
        if (raw) {
            const size_t px = (size_t)w * h;
            if (raw_rgb.bytes < px * 4) {
                DVO_HIP(hipStreamSynchronize(stream));
                DVO_TRY(raw_rgb.alloc(px * 4)); DVO_TRY(raw_depth.alloc(px * 2));
            }
            DVO_TRY(upload_rows(raw_rgb.p, raw, row_bytes, h, 1, geoM.culls, fin.rows_decimated, up, nullptr));
            fin.rgb = raw_rgb.as<uint8_t>(); fin.channels = raw_channels;
        } else {
            DVO_TRY(upload_rows(in_gray.p, gray, row_bytes, h, 1, geoM.culls, fin.rows_decimated, up, nullptr));
            fin.gray = in_gray.as<float>();
        }
        DVO_HIP(hipEventRecord(uevent[2], up));
        DVO_HIP(hipStreamWaitEvent(stream, uevent[2], 0));
    }
    if (!scratch) { scratch = std::make_unique<Keyframe>(); DVO_TRY(scratch->alloc(geoM, cfg, &kf_pool)); }
    Keyframe& frame = *scratch;
    frame.id = ++latest_id;
    for (int i = 0; i < 6; i++) { frame.xi[i] = 0; frame.rel_xi[i] = 0; }
    build_pyramid(frame.fs, fin, stream);
    if (is_key) *is_key = 0;
    const int T = geoM.top();
    const size_t tn = (size_t)geoM.w[T] * geoM.h[T];
    if (hist.empty()) {  // system.hpp:49-54
        if (init_depth.empty()) default_initial_depth((int)tn, cfg.rng_seed, init_depth, init_sigma);
        DVO_HIP(hipMemcpyAsync(frame.fs.depth[T], init_depth.data(), tn * 4, hipMemcpyHostToDevice, stream));
        DVO_HIP(hipMemcpyAsync(frame.fs.sigma[T], init_sigma.data(), tn * 4, hipMemcpyHostToDevice, stream));
        DVO_HIP(hipMemsetAsync(frame.age.p, 0, frame.age.bytes, stream));
        redecimate(frame.fs, frame.fs.depth[T], frame.fs.sigma[T], stream);
        DVO_HIP(hipStreamSynchronize(stream));
        hist.push_back(std::move(scratch));
        hist_version++;
        const float z[6] = {0, 0, 0, 0, 0, 0};
        se3_exp_f(z, T_world);
        if (is_key) *is_key = 1;
        return DVO_OK;
    }
    Keyframe& ref = *hist.back();
    // Frame::updateXi (frame.cpp:7-14), Mapper::needNewFrame (mapper.cpp:45-60) and exp(xi) (system.hpp:73) on the device; the host keeps
    // FrameHistory, so the reference keyframe's pose and id go along as kernel arguments.  On the one-launch schedule they are the tail
    // of k_track_persist and everything the host needs arrives in the mapped block with the tracker's tag: no further launch, no copy,
    // no stream synchronisation.  Otherwise: k_mono_decide (the batched pipeline's kernel) + one copy.
    MonoRef hdr;
    memcpy(hdr.ref_xi, ref.xi, sizeof hdr.ref_xi);
    hdr.ref_id = ref.id; hdr.n_total = (int)hist.size(); hdr.valid = 1;
    memset(&trkM.mono_tail, 0, sizeof trkM.mono_tail);
    if (trkM.persist_ok && !trkM.persist_failed) {
        PersistMono& pm = trkM.mono_tail;
        pm.meta = meta_dev.as<MonoSeq>();
        memcpy(pm.ref_xi, ref.xi, sizeof pm.ref_xi);
        pm.ref_id = ref.id; pm.n_total = (int)hist.size();
        pm.frame_id = frame.id; pm.max_frames = cfg.keyframe_max_frames; pm.min_translation = cfg.keyframe_min_translation;
        pm.enabled = 1;
        // ... and Mapper::update's per-keyframe relative poses (k_age_table), which only need the frame's pose and FrameHistory's
        DVO_TRY(refresh_history_tables());
        pm.hist_xi = hist_xi_dev.as<float>(); pm.ages = ages.as<AgeEntry>(); pm.n_hist = (int)hist.size();
        pm.zero_word = valid_dev.as<int>();
    }
    DVO_TRY(trkM.track(frame.fs, ref.fs, stream));  // system.hpp:57
    bool decided = false;
    if (trkM.persist_used) {
        float rel[6], Trel[16];
        DVO_TRY(trkM.wait_host_result(stream, rel, Trel));   // (a launch that gave up is re-run launch by launch in there: persist_used is false then)
        if (trkM.persist_used) {
            memcpy(h_meta.rel_xi, rel, sizeof h_meta.rel_xi);
            memcpy(h_meta.frame_xi, trkM.h_result + 24, sizeof h_meta.frame_xi);
            memcpy(h_meta.T_world, trkM.h_result + 30, sizeof h_meta.T_world);
            h_meta.need = reinterpret_cast<const int*>(trkM.h_result)[46];
            decided = true;
            age_table_done = h_meta.need == 0;   // (the same launch computed the age table for the update that follows)
        }
    }
    if (!decided) {
        launch_mono_decide(meta_dev.as<MonoSeq>(), trkM.state.as<SeqState>(), 1, frame.id, cfg.keyframe_min_translation, cfg.keyframe_max_frames,
                           nullptr, nullptr, nullptr, &hdr, stream);
        DVO_HIP(hipMemcpyAsync(h_pin, meta_dev.p, sizeof(MonoSeq), hipMemcpyDeviceToHost, stream));
        DVO_HIP(hipStreamSynchronize(stream));
        memcpy(&h_meta, h_pin, sizeof h_meta);
    }
    log_src = &trkM;   // (the 15 KB per-iteration log is read back when dvo_vo_last_track_log asks for it)
    memcpy(frame.rel_xi, h_meta.rel_xi, sizeof frame.rel_xi);
    memcpy(frame.xi, h_meta.frame_xi, sizeof frame.xi);
    frame.ref_id = ref.id;
    const bool need = h_meta.need != 0;
    memcpy(last_xi, frame.xi, sizeof last_xi);
    memcpy(last_rel, frame.rel_xi, sizeof last_rel);
    last_id = frame.id;
    memcpy(T_world, h_meta.T_world, 16 * sizeof(float));
    if (need) {
        DVO_TRY(map_propagate(frame, ref));
        hist.push_back(std::move(scratch));
        hist_version++;
        if (is_key) *is_key = 1;
        if (history_limit > 0 && (int)hist.size() > history_limit) {  // bounded store: drop the oldest keyframes
            DVO_HIP(hipStreamSynchronize(stream));                    // their buffers may still be read by queued kernels
            hist.erase(hist.begin(), hist.begin() + ((int)hist.size() - history_limit));
            hist_version++;
        }
    } else {
        DVO_TRY(map_update(frame));
    }
    DVO_TRY(map_regularize(*hist.back()));  // mapper.cpp:26,30
    DVO_HIP(hipGetLastError());
    return DVO_OK;
}

int VisualOdometry::odometrize_depth(const float* gray, const float* depth, const float* sigma, float T_rel[16])
{  // system.hpp:77-93
    if (!gray || !depth || !sigma || !T_rel) { set_error("null argument"); return DVO_ERR_BAD_ARGUMENT; }
    DVO_TRY(select_device(device));
    if (!trkD_ready) { trkD.prefer_persist = true; DVO_TRY(trkD.init(geoD, 1, cfg)); trkD_ready = true; }
    FrameInput in;   // float maps: only the rows the pyramid keeps cross PCIe (upload_rows)
    in.rows_decimated = decimate_host_rows && can_decimate_rows(geoD);
    const size_t rb = (size_t)w * sizeof(float);
    // Tracking this frame needs its GRAY pyramid only (obj) -- depth, sigma and the weight come from the reference, the previous frame
    // (tracker.cpp:22-41).  So only the gray map's copy + pyramid sit on the critical path; the depth and sigma maps go up on a side
    // stream and their pyramid is built there WHILE this frame is tracked (three strided copies in a row were 98 us between one frame's
    // tracking and the next, profiles/r03_single_hip_trace.txt; one is ~30).  The next call makes the tracking stream wait for that
    // build (this frame is then the reference); this call returns only after the side copies have read the caller's buffers.
    DVO_TRY(upload_streams());
    if (!depth_cur) { depth_cur = std::make_unique<Keyframe>(); DVO_TRY(depth_cur->alloc(geoD, cfg)); }
    if (side_built) DVO_HIP(hipStreamWaitEvent(stream, uevent[1], 0));   // the reference's depth / sigma / weight pyramid (built during the previous call)
    DVO_TRY(upload_rows(in_gray.p, gray, rb, h, 1, geoD.culls, in.rows_decimated, stream, nullptr));
    Keyframe* const target = depth_cur.get();
    const bool dec = in.rows_decimated;
    const std::function<int()> side = [&]() -> int {   // queued after this frame's pyramid + tracking launches: those are the critical path
        DVO_TRY(upload_rows(in_depth.p, depth, rb, h, 1, geoD.culls, dec, ustream[0], nullptr));
        DVO_TRY(upload_rows(in_sigma.p, sigma, rb, h, 1, geoD.culls, dec, ustream[0], nullptr));
        DVO_HIP(hipEventRecord(uevent[0], ustream[0]));
        build_pyramid(target->fs, nullptr, in_depth.as<float>(), in_sigma.as<float>(), ustream[0], true, dec);
        DVO_HIP(hipEventRecord(uevent[1], ustream[0]));
        side_built = true;
        return DVO_OK;
    };
    in.gray = in_gray.as<float>(); in.depth = nullptr; in.sigma = nullptr;
    const int rc = odometrize_depth_staged(T_rel, &in, &side);
    if (side_built) DVO_HIP(hipEventSynchronize(uevent[0]));   // (long done: the copies take ~50 us, the tracking ~270)
    return rc;
}

int VisualOdometry::odometrize_depth_raw(const uint8_t* rgb, int channels, const uint16_t* depth16, float depth_scale, float T_rel[16])
{  // the same call fed with raw sensor frames: u8 gray/RGB(A) + u16 depth, converted on the device while the pyramid is built
    if (!rgb || !depth16 || !T_rel || (channels != 1 && channels != 3 && channels != 4)) { set_error("bad raw frame"); return DVO_ERR_BAD_ARGUMENT; }
    DVO_TRY(select_device(device));
    const size_t px = (size_t)w * h;
    if (raw_rgb.bytes < px * 4) { DVO_TRY(raw_rgb.alloc(px * 4)); DVO_TRY(raw_depth.alloc(px * 2)); }
    FrameInput in;
    in.rows_decimated = decimate_host_rows && can_decimate_rows(geoD);   // only the rows the pyramid keeps cross PCIe
    DVO_TRY(upload_streams());
    if (side_built) { DVO_HIP(hipStreamWaitEvent(stream, uevent[1], 0)); side_built = false; }   // a float-map frame's depth pyramid may still be building: it is this call's reference
    // The kept rows of both frames (0.46 MB of 0.92 at cull 1) are staged by the caller's thread in pinned, device-mapped memory and read
    // from there by k_pyramid_raw4 -- no runtime copy (as in the mono loop; DVO_RAW_STAGE=0: two copies, the depth on the side stream).
    if (stage_raw_rows && ((size_t)w * channels + (size_t)w * 2) * (size_t)(in.rows_decimated ? h >> geoD.culls : h) <= kStageLimitBytes) {
        DVO_TRY(alloc_stage());
        char* hs = static_cast<char*>(h_stage);
        stage_rows_host(hs, rgb, (size_t)w * channels, h, geoD.culls, in.rows_decimated);
        stage_rows_host(hs + (size_t)w * h * 4, depth16, (size_t)w * 2, h, geoD.culls, in.rows_decimated);
        in.rgb = static_cast<const uint8_t*>(d_stage); in.channels = channels;
        in.depth16 = reinterpret_cast<const uint16_t*>(static_cast<const char*>(d_stage) + (size_t)w * h * 4); in.depth_scale = depth_scale;
        return odometrize_depth_staged(T_rel, &in);
    }
    DVO_TRY(upload_rows(raw_depth.p, depth16, (size_t)w * 2, h, 1, geoD.culls, in.rows_decimated, ustream[0], nullptr));
    DVO_HIP(hipEventRecord(uevent[0], ustream[0]));
    DVO_TRY(upload_rows(raw_rgb.p, rgb, (size_t)w * channels, h, 1, geoD.culls, in.rows_decimated, stream, nullptr));
    DVO_HIP(hipStreamWaitEvent(stream, uevent[0], 0));
    in.rgb = raw_rgb.as<uint8_t>(); in.channels = channels; in.depth16 = raw_depth.as<uint16_t>(); in.depth_scale = depth_scale;
    return odometrize_depth_staged(T_rel, &in);
}

int VisualOdometry::odometrize_depth_staged(float T_rel[16], const FrameInput* raw, const std::function<int()>* after_launch)
{
    if (!trkD_ready) { trkD.prefer_persist = true; DVO_TRY(trkD.init(geoD, 1, cfg)); trkD_ready = true; }
    if (!depth_cur) { depth_cur = std::make_unique<Keyframe>(); DVO_TRY(depth_cur->alloc(geoD, cfg)); }
    Keyframe& frame = *depth_cur;
    frame.id = ++latest_id;
    if (raw) build_pyramid(frame.fs, *raw, stream);   // (raw sensor frame or float maps staged by the caller)
    else build_pyramid(frame.fs, in_gray.as<float>(), in_depth.as<float>(), in_sigma.as<float>(), stream);
    const float z[6] = {0, 0, 0, 0, 0, 0};
    if (!depth_ref) {  // system.hpp:83-86
        for (int i = 0; i < 6; i++) { frame.xi[i] = 0; frame.rel_xi[i] = 0; }
        if (after_launch) DVO_TRY((*after_launch)());
        DVO_HIP(hipStreamSynchronize(stream));
        depth_ref = std::move(depth_cur);
        se3_exp_f(z, T_rel);
        return DVO_OK;
    }
    DVO_TRY(trkD.enable_host_result());
    DVO_TRY(trkD.track(frame.fs, depth_ref->fs, stream));
    if (after_launch) DVO_TRY((*after_launch)());   // (work that is not on this frame's critical path is queued once the tracking is)
    // The pose comes back through mapped host memory (k_export_poses' last store), not through a copy + stream synchronisation; the
    // caller's input buffers were consumed by copies that are stream-ordered before the kernels whose result this waits for.
    float rel[6];
    DVO_TRY(trkD.wait_host_result(stream, rel, T_rel));
    log_src = &trkD;   // the per-iteration log stays on the device until dvo_vo_last_track_log asks for it
    memcpy(frame.rel_xi, rel, sizeof rel);
    frame.ref_id = depth_ref->id;
    se3_concatenate_f(depth_ref->xi, rel, frame.xi);
    memcpy(last_xi, frame.xi, sizeof last_xi);
    memcpy(last_rel, rel, sizeof last_rel);
    last_id = frame.id;
    std::swap(depth_ref, depth_cur);  // m_ref_frame = frame, system.hpp:91
    return DVO_OK;
}

// ------------------------------------------------------------------------------------------------ batch
Batch::~Batch()
{
    if (pstream) { (void)hipStreamSynchronize(pstream); (void)hipStreamDestroy(pstream); }
    if (cstream) { (void)hipStreamSynchronize(cstream); (void)hipStreamDestroy(cstream); }
    for (auto& st : stage) {
        if (st.copied) (void)hipEventDestroy(st.copied);
        if (st.consumed) (void)hipEventDestroy(st.consumed);
    }
    if (ev_last_track) (void)hipEventDestroy(ev_last_track);
    for (int i = 0; i < 3; i++) if (ev_built[i]) (void)hipEventDestroy(ev_built[i]);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
}

int Batch::init(int n, const float K9[9], int w, int h, int levels, int culls, const dvo_config* c)
{
    if (n < 1 || !K9) { set_error("dvo_batch_create: bad arguments"); return DVO_ERR_BAD_ARGUMENT; }
    if (c) cfg = *c; else dvo_config_default(&cfg);
    n_seq = n; device = cfg.device;
    DVO_TRY(select_device(device));
    if (cfg.stream) stream = (hipStream_t)cfg.stream;
    else { DVO_HIP(hipStreamCreate(&stream)); own_stream = true; }
    DVO_TRY(make_geometry(K9, w, h, levels, culls, g));
    for (int i = 0; i < 3; i++) DVO_TRY(fs[i].alloc(g, n, cfg));
    DVO_TRY(trk.init(g, n, cfg));
    {   // lowest priority: the pyramid build should fill what the tracker leaves idle, not compete with it
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        DVO_HIP(hipStreamCreateWithPriority(&pstream, hipStreamNonBlocking, lo));
    }
    DVO_HIP(hipEventCreateWithFlags(&ev_last_track, hipEventDisableTiming));
    for (int i = 0; i < 3; i++) DVO_HIP(hipEventCreateWithFlags(&ev_built[i], hipEventDisableTiming));
    return DVO_OK;
}

// Builds the pyramids of a frame that will be handed to push_device later (call order per step: prefetch(k+1); push(k)) on the
// side stream.  The set it builds into may be the reference of the tracking queued last: the build waits for that tracking --
// not for anything queued afterwards -- and then runs beside the tracking of frame k.
// One frame of every sequence from HOST memory: up to three buffers (float gray / depth / sigma, or raw rgb / depth16) go to the
// staging slot of this push on the copy stream; the tracking stream waits for that copy only.  Slot k & 1 is reused by push k + 2,
// whose copy waits until push k (pyramid build + tracking) is done with it.
int Batch::push_host_frame(const void* p0, size_t n0, const void* p1, size_t n1, const void* p2, size_t n2, FrameInput in)
{
    DVO_TRY(select_device(device));
    if (!cstream) {
        DVO_HIP(hipStreamCreateWithFlags(&cstream, hipStreamNonBlocking));
        for (auto& st : stage) {
            DVO_HIP(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming));
            DVO_HIP(hipEventCreateWithFlags(&st.consumed, hipEventDisableTiming));
        }
    }
    // The adaptive schedule keeps the host inside track() until the GPU is nearly done with the frame; a host-fed batch wants the
    // next frame's transfer queued meanwhile, so it runs the fixed schedule (bit-identical results, tested).
    trk.adaptive = false;
    Stage& st = stage[n_host_push & 1];
    n_host_push++;
    if (st.a.bytes < n0) DVO_TRY(st.a.alloc(n0));
    if (p1 && st.b.bytes < n1) DVO_TRY(st.b.alloc(n1));
    if (p2 && st.c.bytes < n2) DVO_TRY(st.c.alloc(n2));
    if (st.used) DVO_HIP(hipStreamWaitEvent(cstream, st.consumed, 0));
    if (in.raw()) {  // only the rows the pyramid keeps cross PCIe (the staging buffers are sized for whole frames)
        in.rows_decimated = decimate_host_rows && can_decimate_rows(g);
        DVO_TRY(upload_rows(st.a.p, p0, (size_t)g.src_w * in.channels, g.src_h, (size_t)n_seq, g.culls, in.rows_decimated, cstream, nullptr));
        DVO_TRY(upload_rows(st.b.p, p1, (size_t)g.src_w * 2, g.src_h, (size_t)n_seq, g.culls, in.rows_decimated, cstream, nullptr));
    } else {
        in.rows_decimated = decimate_host_rows && can_decimate_rows(g);
        const size_t rb = (size_t)g.src_w * sizeof(float);
        DVO_TRY(upload_rows(st.a.p, p0, rb, g.src_h, (size_t)n_seq, g.culls, in.rows_decimated, cstream, nullptr));
        if (p1) DVO_TRY(upload_rows(st.b.p, p1, rb, g.src_h, (size_t)n_seq, g.culls, in.rows_decimated, cstream, nullptr));
        if (p2) DVO_TRY(upload_rows(st.c.p, p2, rb, g.src_h, (size_t)n_seq, g.culls, in.rows_decimated, cstream, nullptr));
    }
    DVO_HIP(hipEventRecord(st.copied, cstream));
    // Pageable memory: the runtime may pin it in place and return while the DMA is still reading it, and the caller is free to
    // release the buffer as soon as this call returns -- so wait for the copy (only the copy: the tracking of the previous frame
    // keeps running on `stream`).  Pinned buffers stay asynchronous, as the header says.
    if (!host_buffer_is_pinned(p0) || (p1 && !host_buffer_is_pinned(p1)) || (p2 && !host_buffer_is_pinned(p2))) DVO_HIP(hipStreamSynchronize(cstream));
    DVO_HIP(hipStreamWaitEvent(stream, st.copied, 0));
    if (in.raw()) { in.rgb = st.a.as<uint8_t>(); in.depth16 = st.b.as<uint16_t>(); }
    else { in.gray = st.a.as<float>(); in.depth = st.b.as<float>(); in.sigma = st.c.as<float>(); }
    const int rc = push(in);
    DVO_HIP(hipEventRecord(st.consumed, stream));
    st.used = true;
    return rc;
}

int Batch::prefetch(const FrameInput& in)
{
    if (!in.key0() || !in.has_depth()) { set_error("null device pointer"); return DVO_ERR_BAD_ARGUMENT; }
    DVO_TRY(select_device(device));
    const int slot = free_slot();
    if (npre >= 2 || slot < 0) { set_error("dvo_batch_prefetch_device: two prefetched frames are already waiting for their push"); return DVO_ERR_NOT_READY; }
    if (tracked_once) DVO_HIP(hipStreamWaitEvent(pstream, ev_last_track, 0));
    build_pyramid(fs[slot], in, pstream, /*keep_sigma=*/false);
    DVO_HIP(hipEventRecord(ev_built[slot], pstream));
    preq[npre] = slot;
    pre_key[npre][0] = in.key0(); pre_key[npre][1] = in.key1();
    npre++;
    DVO_HIP(hipGetLastError());
    return DVO_OK;
}

int Batch::push(const FrameInput& in)
{
    if (!in.key0() || !in.has_depth()) { set_error("null device pointer"); return DVO_ERR_BAD_ARGUMENT; }
    DVO_TRY(select_device(device));
    int target;
    if (npre > 0 && pre_key[0][0] == in.key0() && pre_key[0][1] == in.key1()) {
        target = preq[0];                                   // built by prefetch: the tracker waits for that build
        DVO_HIP(hipStreamWaitEvent(stream, ev_built[target], 0));
        preq[0] = preq[1];
        for (int i = 0; i < 2; i++) pre_key[0][i] = pre_key[1][i];
        npre--;
    } else {
        target = (cur < 0 && npre == 0) ? 0 : free_slot();
        if (target < 0) { set_error("dvo_batch_push_device: the frames prefetched must be pushed first, in order"); return DVO_ERR_BAD_ARGUMENT; }
        build_pyramid(fs[target], in, stream, /*keep_sigma=*/false);  // Frame(gray,depth,sigma,K,levels,culls)
    }
    if (cur >= 0) {
        DVO_TRY(trk.track(fs[target], fs[cur], stream));    // system.hpp:88
        DVO_HIP(hipEventRecord(ev_last_track, stream));
        tracked_once = true;
        have_poses = true;
    }
    prev = cur;
    cur = target;                                           // system.hpp:91
    DVO_HIP(hipGetLastError());
    return DVO_OK;
}

}  // namespace dvo
