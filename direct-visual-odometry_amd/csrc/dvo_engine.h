// dvo_engine.h -- C++ host side of libdvo.so: device memory, pyramids, the batched tracker and the
// single-sequence VisualOdometry (tracking + mapping).  Mirrors the reference's classes:
//   System::Frame / Scene  -> FrameSet (n_seq frames, one buffer per pyramid level)   include/system/frame.hpp
//   Track::Tracker         -> Tracker                                                 src/track/tracker.cpp
//   Map::Mapper            -> Mapper functions on Keyframe                             src/map/mapper.cpp
//   System::VisualOdometry -> VisualOdometry                                           include/system/system.hpp
// There is NO CPU fallback: every entry point fails with DVO_ERR_NO_DEVICE / DVO_ERR_HIP without a GPU.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime_api.h>

#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "dvo_kernels.h"

namespace dvo {

void set_error(const std::string& s);
const char* last_error();
int check_hip(hipError_t e, const char* what);
#define DVO_HIP(call)                                              \
    do {                                                           \
        int _st = ::dvo::check_hip((call), #call);                 \
        if (_st != DVO_OK) return _st;                             \
    } while (0)
#define DVO_TRY(call)                                              \
    do {                                                           \
        int _st = (call);                                          \
        if (_st != DVO_OK) return _st;                             \
    } while (0)

struct DevBuf {  // RAII hipMalloc (or a view of memory somebody else owns: adopt())
    void* p = nullptr;
    size_t bytes = 0;
    bool owned = true;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n);
    void adopt(void* mem, size_t n) { release(); p = mem; bytes = n; owned = false; }
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Device memory for the keyframes of one dvo_vo handle: FrameHistory grows by one Frame per keyframe (frame.hpp:151-157), and a hipMalloc
// per map of every new keyframe costs more than tracking a frame.  Blocks of one keyframe's size are cut from slabs of 16 and recycled
// when a keyframe is dropped (dvo_vo_set_history_limit).
struct KeyframePool {
    size_t block_bytes = 0;
    std::vector<std::unique_ptr<DevBuf>> slabs;
    std::vector<void*> free_blocks;
    int take(size_t bytes, void** out);
    void give(void* blk) { free_blocks.push_back(blk); }
};

struct Geometry {  // pyramid shape of Frame(gray, K, levels, culls): frame.hpp:91-117, frame.cpp:30-37
    int src_w = 0, src_h = 0, levels = 0, culls = 0;
    int w[DVO_MAX_LEVELS] = {0}, h[DVO_MAX_LEVELS] = {0};
    float K9[DVO_MAX_LEVELS][9];
    Intr k[DVO_MAX_LEVELS];
    size_t px_total = 0;  // sum over levels of w*h
    int top() const { return levels - 1; }
};
int make_geometry(const float K[9], int w, int h, int levels, int culls, Geometry& g);

struct FrameSet {  // n_seq frames: gray/depth/sigma pyramids, level l stored as [n_seq][h_l][w_l]
    Geometry g;
    int n_seq = 0;
    DevBuf arena;
    float* gray[DVO_MAX_LEVELS] = {nullptr};
    float* depth[DVO_MAX_LEVELS] = {nullptr};
    float* sigma[DVO_MAX_LEVELS] = {nullptr};
    // Per-pixel constant of a REFERENCE frame, derived once per frame instead of once per GN iteration:
    // wgt = step(level) / clamp(sigma) (optimize.cpp:83-84).  Same float operations, hoisted out of the iteration loop.
    float* wgt[DVO_MAX_LEVELS] = {nullptr};
    // Frames that came through the raw sensor conversion without a stored sigma pyramid carry sigma = 0.1 where depth > 0 and 1.0
    // elsewhere (transform.cpp:75): a pixel can only contribute with depth >= min_depth > 0, so its weight is the one constant
    // wgt_valid[level] and the wgt maps are neither written nor read (allow_const_weight: min_depth > 0).
    bool allow_const_weight = false, sigma_by_validity = false;
    float wgt_valid[DVO_MAX_LEVELS] = {0};
    float step[DVO_MAX_LEVELS] = {0};
    float sigma_min = 0.01f, sigma_max = 0.5f;
    int alloc(const Geometry& geo, int n, const dvo_config& cfg, void* mem = nullptr);   // mem: caller-owned block of arena_bytes()
    static size_t arena_bytes(const Geometry& geo, int n) { return 4 * geo.px_total * (size_t)n * sizeof(float); }
};

// Builds pyramids of (gray, depth, sigma) device inputs [n_seq][src_h][src_w]; depth/sigma may be null.
// keep_sigma = false: the sigma pyramid is only folded into `wgt`, not stored (frame-to-frame tracking never reads it again)
void build_pyramid(FrameSet& fs, const float* gray_dev, const float* depth_dev, const float* sigma_dev, hipStream_t s, bool keep_sigma = true,
                   bool rows_decimated = false);
// One frame of every sequence as handed over by the caller: float maps (gray [+ depth + sigma]) or raw sensor frames
// (u8 gray / RGB / RGBA [+ u16 depth], converted while the pyramid is built: loader.cpp:55-60,137-147, transform.cpp:60-76).
struct FrameInput {
    const float* gray = nullptr; const float* depth = nullptr; const float* sigma = nullptr;
    const uint8_t* rgb = nullptr; int channels = 0; const uint16_t* depth16 = nullptr; float depth_scale = 1.0f / 5000.0f;
    bool rows_decimated = false;  // the buffers hold only the rows the pyramid keeps (every 2^culls-th), see upload_rows
    bool raw() const { return rgb != nullptr; }
    const void* key0() const { return raw() ? (const void*)rgb : (const void*)gray; }
    const void* key1() const { return raw() ? (const void*)depth16 : (const void*)depth; }
    bool has_depth() const { return raw() ? depth16 != nullptr : (depth != nullptr && sigma != nullptr); }
};
void build_pyramid(FrameSet& fs, const FrameInput& in, hipStream_t s, bool keep_sigma = true);
// Host -> device copy of n_img images (raw or float; rows of row_bytes bytes).  With culls > 0 and decimate set only every
// 2^culls-th row of each image is transferred (one strided DMA): the pyramid never reads the others (Convert::cullImage keeps
// pixels whose coordinates are multiples of 2^culls), so 1 - 2^-culls of the PCIe traffic carries nothing.  Returns the bytes
// the device buffer holds through *stored.
int upload_rows(void* dst, const void* src, size_t row_bytes, int img_rows, size_t n_img, int culls, bool decimate, hipStream_t s,
                    size_t* stored);
inline bool can_decimate_rows(const Geometry& g) { return g.culls > 0 && (g.src_h % (1 << g.culls)) == 0; }
// Frame::updateDepthSigma / updateDepth (frame.cpp:39-61): re-decimate from a top-level map (may alias the top level)
void redecimate(FrameSet& fs, const float* depth_top, const float* sigma_top, hipStream_t s);

struct Tracker {  // Track::Tracker for n_seq sequences at once
    Geometry g;
    int n_seq = 0;
    dvo_config cfg;
    DevBuf state, partials, log, counters, xi_out, T_out;
    // Two active-sequence lists per sub-batch ([0] = count, [4..] = ids local to the sub-batch), written by k_gn_solve,
    // read by the next k_track_gn.
    DevBuf work;
    int* work_list(int sub, int i) const { return work.as<int>() + (size_t)(2 * sub + (i & 1)) * (size_t)(n_seq + 4); }
    // Sub-batches: the sequences are split into n_sub contiguous groups whose launch chains (k_track_gn -> k_gn_solve ->
    // ...) run on concurrent HIP streams, so one group's latency-bound solve / launch gaps are covered by another
    // group's work.  Group 0 uses the caller's stream; fork/join events order the groups against it.
    int n_sub = 1;
    std::vector<hipStream_t> sub_streams;  // n_sub - 1 library-owned streams
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    int sub_first(int k) const { return (int)(((long long)n_seq * k) / n_sub); }
    int ppt[DVO_MAX_LEVELS], nblk[DVO_MAX_LEVELS], group[DVO_MAX_LEVELS];
    int tiles_x[DVO_MAX_LEVELS], tiles_y[DVO_MAX_LEVELS];
    bool fused[DVO_MAX_LEVELS];  // level runs as ONE k_track_level launch (all iterations on the device)
    bool single_launch[DVO_MAX_LEVELS];  // level runs one k_track_gn_fused launch per iteration (small handles: see dvo_kernels.hip)
    DevBuf ticket, freport;      // k_track_gn_fused: arrival tickets [n_seq]; (reported, active) per (set, level, iteration)
    // Adaptive schedule: progress words in mapped host memory, one per (level, iteration), two sets used alternately.
    // k_gn_solve's workgroup 0 stores (active sequences + 1); the host reads them to stay ~2 iterations ahead of the GPU and
    // to stop enqueuing a level once a launch reported zero active sequences (its remaining launches would be empty).
    int* h_progress = nullptr;   // host view
    int* d_progress = nullptr;   // device view of the same memory
    int progress_set = 0;
    bool adaptive = false;
    SeqState* h_state = nullptr;  // pinned host mirror of `state` for the small-batch convergence poll
    int tile_margin = 0;  // > 0: k_track_gn_tile (LDS-staged reference patch); 0: k_track_gn (global gathers)
    void launch_gn(const GnArgs& a, int level, int count, hipStream_t s, int grid_seqs = 0) const;  // `a` views `count` sequences
    // Result hand-over for a handle that returns one pose per call (dvo_vo): k_export_poses also writes xi, T and a tag into
    // fine-grained mapped host memory, and wait_host_result() polls the tag -- no device-to-host copy, no stream synchronisation.
    // One launch per track() call (k_track_persist) for a single sequence whose result is handed over through h_result: eligible
    // when every level fits the kernel's wide reduction and shares one tile size; `persist_failed` = a launch gave up waiting
    // (GPU oversubscribed): the handle then stays on the launch-per-iteration schedule.
    bool prefer_persist = false;   // set before init() by the owner whose results go through enable_host_result() (VisualOdometry's sensor-depth tracker)
    bool persist_ok = false, persist_failed = false, persist_used = false;
    int persist_grid = 0, persist_spin_limit = 1 << 18;
    bool persist_timeline = false;
    DevBuf persist_ctl, persist_dbg;
    int read_persist_timeline(long long* out);
    const FrameSet* last_obj = nullptr; const FrameSet* last_ref = nullptr;
    float* h_result = nullptr;   // host view: [0..5] xi, [6..21] T, [22] tag (int), [23] tag of a persistent launch that gave up
    float* d_result = nullptr;   // device view of the same memory
    int result_tag = 0;
    PersistMono mono_tail = {};    // armed by a mono dvo_vo handle before track(): k_track_persist also does k_mono_decide's work
    int persist_ppt = 0;           // pixels per thread of the one-launch schedule: 0 = 4, > 0 = that many, < 0 = what the other schedules pick
    int enable_host_result();
    int wait_host_result(hipStream_t s, float xi[6], float T[16]);   // of the last track() call
    // profiling (cfg.profile)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    double prof_ms = 0;
    uint64_t prof_launches = 0;
    ~Tracker();
    int init(const Geometry& geo, int n, const dvo_config& c);
    GnParams level_params(int level) const;
    GnArgs gn_args(const FrameSet& obj, const FrameSet& ref, int level, uint8_t* mask, int ignore_active) const;
    // Tracker::track (tracker.cpp:22-85): enqueue the whole coarse-to-fine loop; poses land in xi_out / T_out
    int track(const FrameSet& obj, const FrameSet& ref, hipStream_t s);
    int collect_profile(hipStream_t s);
};

struct Keyframe {  // System::Frame of one sequence, plus the age map and pose (frame.hpp:72-144)
    FrameSet fs;
    DevBuf age;    // top-level [h][w]
    DevBuf depth_alt;              // second top-level depth map: k_regularize_redecimate writes the regularized map beside the one it reads
    float* depth_spare = nullptr;  // whichever of the arena's top-level block and depth_alt fs.depth[top] does not point to
    float xi[6] = {0, 0, 0, 0, 0, 0}, rel_xi[6] = {0, 0, 0, 0, 0, 0};
    int id = -1, ref_id = -1;
    KeyframePool* pool = nullptr;  // where `block` (the memory of all three buffers) goes back to
    void* block = nullptr;
    int alloc(const Geometry& g, const dvo_config& cfg, KeyframePool* from = nullptr);
    ~Keyframe() { if (pool && block) pool->give(block); }
};

struct VisualOdometry {  // System::VisualOdometry, system.hpp:12-104
    float K[9];
    int w = 0, h = 0, device = 0;
    dvo_config cfg;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    Geometry geoM, geoD;  // Frame(gray,K,3,2) (system.hpp:47) and Frame(g,d,s,K,4,1) (system.hpp:82)
    Tracker trkM, trkD;
    bool trkM_ready = false, trkD_ready = false;
    KeyframePool kf_pool;                              // (declared before the keyframes: destroyed after them)
    std::vector<std::unique_ptr<Keyframe>> hist;       // FrameHistory, oldest first
    std::unique_ptr<Keyframe> scratch;                 // the frame being processed (promoted on keyframe)
    std::unique_ptr<Keyframe> depth_ref, depth_cur;    // m_ref_frame of odometrizeUsingDepth
    DevBuf in_gray, in_depth, in_sigma;                // full-resolution staging
    DevBuf tmp_a, tmp_b, tmp_c, owner, ages, valid_dev;
    // Mapper state of this sequence on the device (the same kernels as the batched mono pipeline: k_mono_decide, k_age_table),
    // so that a dvo_vo handle and a sequence of a mono dvo_batch produce identical bits.
    DevBuf meta_dev, hist_xi_dev, gray_tab_dev;
    MonoSeq h_meta;
    void* h_pin = nullptr;   // pinned staging for the per-frame read-back (MonoSeq + track log)
    std::vector<float> init_depth, init_sigma;
    int latest_id = -1;                                // Frame::latest_id, frame.cpp:5
    int history_limit = 0;                             // 0 = keep every keyframe (the reference); N = keep the newest N
    int last_id = -1, last_valid_updates = 0;
    bool valid_updates_pending = false;                // last_valid_updates is stale: the count of the last map_update is still on the device
    int fetch_valid_updates();
    // device copies of FrameHistory's poses / gray pointers (k_age_table, k_depth_update): refreshed when the history changes

    int hist_table_n = -1;
    unsigned long long hist_version = 0, hist_table_version = ~0ull;   // hist_version: bumped wherever `hist` changes
    float last_xi[6] = {0}, last_rel[6] = {0};
    dvo_track_log last_log;
    Tracker* log_src = nullptr;                        // the tracker whose device log is newer than last_log (read back on demand only)
    int fetch_log();                                   // dvo_vo_last_track_log: the 15 KB per-iteration record is copied when asked for
    ~VisualOdometry();
    int init(const float K9[9], int width, int height, const dvo_config* c);
    int odometrize(const float* gray, float T_world[16], int* is_key, const uint8_t* raw = nullptr, int raw_channels = 0);
    int odometrize_depth(const float* gray, const float* depth, const float* sigma, float T_rel[16]);
    int odometrize_depth_raw(const uint8_t* rgb, int channels, const uint16_t* depth16, float depth_scale, float T_rel[16]);
    int odometrize_depth_staged(float T_rel[16], const struct FrameInput* raw = nullptr,   // frame already staged on the device
                                const std::function<int()>* after_launch = nullptr);
    DevBuf raw_rgb, raw_depth;
    // the maps of one frame go up on separate streams: three strided copies queued on one stream run one after the other with
    // ~9 us between them (98 us from the end of one frame's tracking to the next pyramid, profiles/r03_single_hip_trace.txt)
    hipStream_t ustream[2] = {nullptr, nullptr};
    hipEvent_t uevent[3] = {nullptr, nullptr};
    int upload_streams();
    bool side_built = false;   // uevent[1] marks a depth / sigma pyramid built on the side stream (odometrize_depth)
    bool decimate_host_rows = getenv("DVO_UPLOAD_FULL_FRAMES") == nullptr;  // as Batch::decimate_host_rows
    int init_keyframe(const float* gray, const float* depth, const float* sigma);
    int map_propagate(Keyframe& frame, const Keyframe& ref);
    int map_update(Keyframe& obj);
    int refresh_history_tables();
    int alloc_stage();
    bool stage_mono_rows = true, stage_raw_rows = true;   // frames reach the pyramid kernel through the staging block (DVO_MONO_STAGE / DVO_RAW_STAGE = 0: runtime copies)
    void* h_stage = nullptr;       // pinned, device-mapped staging of a mono frame's kept rows (k_pyramid reads it through d_stage)
    void* d_stage = nullptr;
    void* h_tables = nullptr;      // pinned staging of the history tables
    size_t h_tables_bytes = 0;
    bool age_table_done = false;   // this frame's age table came out of k_track_persist's tail
    int map_regularize(Keyframe& kf);
};

struct Batch {  // n_seq independent sequences, frame-to-frame tracking with sensor depth
    int n_seq = 0, device = 0;
    dvo_config cfg;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    Geometry g;
    Tracker trk;
    // Three frame sets: the reference (cur), the frame being tracked against it, and one that the pyramid of a LATER frame can
    // be built into on a low-priority side stream while tracking runs (prefetch_device): k_pyramid is HBM bound, the tracker
    // mostly VALU / latency bound.
    FrameSet fs[3];
    int cur = -1;           // frame set of the newest tracked frame = reference of the next (-1: none yet)
    int prev = -1;          // its own reference (probes)
    // prefetched frame sets waiting for their push_device, oldest first.  In steady state two are outstanding at the moment
    // prefetch_device is called: "prefetch(k+1); push(k)" finds frame k (prefetched one step earlier) still waiting.
    int preq[2] = {-1, -1};
    int npre = 0;
    const void* pre_key[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    hipStream_t pstream = nullptr;
    hipEvent_t ev_last_track = nullptr, ev_built[3] = {nullptr, nullptr, nullptr};
    bool tracked_once = false;
    bool have_poses = false;
    // Host input (push_host / push_raw_host): two staging slots filled on a copy stream, so the H2D transfer of frame k+1 runs
    // beside the tracking of frame k (PCIe is the bound of a host-fed batch: 0.92 MB per raw 640x480 frame).
    struct Stage { DevBuf a, b, c; hipEvent_t copied = nullptr, consumed = nullptr; bool used = false; } stage[2];
    hipStream_t cstream = nullptr;
    int n_host_push = 0;
    bool decimate_host_rows = getenv("DVO_UPLOAD_FULL_FRAMES") == nullptr;  // raw host frames: transfer only the rows the pyramid keeps
    int push_host_frame(const void* p0, size_t n0, const void* p1, size_t n1, const void* p2, size_t n2, FrameInput in);
    ~Batch();
    int init(int n, const float K9[9], int w, int h, int levels, int culls, const dvo_config* c);
    int free_slot() const  // a frame set that is neither the reference nor waiting prefetched (-1: none)
    {
        for (int i = 0; i < 3; i++)
            if (i != cur && !(npre > 0 && preq[0] == i) && !(npre > 1 && preq[1] == i)) return i;
        return -1;
    }
    int prefetch(const FrameInput& in);
    int push(const FrameInput& in);
};

int select_device(int device);
// true when `p` is page-locked host memory known to HIP (hipHostMalloc / hipHostRegister): only then may an asynchronous copy
// still read the buffer after the call that queued it has returned
bool host_buffer_is_pinned(const void* p);

// Optional roctx ranges (the reference prints a Timer line per Gauss-Newton iteration and mapping stage: tracker.cpp:43,54-61,
// mapper.cpp:18,27,32): with DVO_TRACE=1 in the environment every tracking level and mapping stage of a frame is a named range in a
// `rocprofv3 --marker-trace` timeline.  libroctx64 is loaded lazily with dlopen; without it, or without DVO_TRACE, these are no-ops.
void trace_push(const char* name);
void trace_pop();
struct TraceRange { explicit TraceRange(const char* n) { trace_push(n); } ~TraceRange() { trace_pop(); } };

// n_seq independent MONO sequences on one GPU: System::VisualOdometry::odometrize (system.hpp:44-74) -- track against the newest
// keyframe, then Mapper::estimate (propagate + new keyframe, or stereo update) and regularize -- for every sequence per call, with
// the keyframe decision taken on the device per sequence.  FrameHistory is a ring of the newest R keyframes per sequence
// (their top-level gray + pose; the reference keeps every frame forever, frame.hpp:146-188): with R >= the number of keyframes
// a run creates the results equal the reference's unbounded history, beyond that a dvo_vo handle with history limit R.
struct MonoBatch {
    int n_seq = 0, device = 0, R = 8;
    dvo_config cfg;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    Geometry g;              // Frame(gray, K, 3, 2), system.hpp:47
    Tracker trk;
    FrameSet ref, frm;       // newest keyframe of every sequence; the frame being processed
    DevBuf ref_age, frm_age, owner, tmp, ring_gray, hist_xi, ages, meta, init_depth, init_sigma;
    DevBuf xi_world, T_world, is_key;
    DevBuf need_list;        // [0] = number of sequences that create a keyframe on this frame, [4..] their ids (k_mono_decide)
    float* depth_alt = nullptr;  // the second top-level depth buffer of `ref` (k_regularize_redecimate ping-pongs between the two)
    int latest_id = -1;      // Frame::latest_id, frame.cpp:5 (all sequences advance in lockstep)
    bool have_init = false;
    int init(int n, const float K9[9], int w, int h, int ring, const dvo_config* c);
    ~MonoBatch();
    int set_initial_depth(const float* depth_host, const float* sigma_host);              // one map, broadcast to every sequence
    int set_initial_depth_device(const float* depth_dev, const float* sigma_dev);         // [n_seq][th][tw]
    int odometrize(const FrameInput& in);                                                  // gray [n_seq][h][w] float, or raw u8
    int odometrize_host(const void* frames, size_t bytes, FrameInput in);                  // the same from host memory (copy stream, 2 slots)
    struct Stage { DevBuf buf; hipEvent_t copied = nullptr, consumed = nullptr; bool used = false; } stage[2];
    hipStream_t cstream = nullptr;
    int n_host = 0;
    bool decimate_host_rows = getenv("DVO_UPLOAD_FULL_FRAMES") == nullptr;  // as Batch::decimate_host_rows
    int top_pixels() const { return g.w[g.top()] * g.h[g.top()]; }
    // profiling of the mapping stages (cfg.profile): hipEvent pairs on `stream` around k_depth_update (+ k_age_table),
    // k_regularize_redecimate and the three k_propagate_* passes of every frame
    struct MapEv { hipEvent_t e[6]; };
    std::vector<MapEv> map_ev;
    size_t map_ev_used = 0;
    double prof_update_ms = 0, prof_regularize_ms = 0, prof_propagate_ms = 0;
    uint64_t prof_frames = 0;
    int collect_map_profile();
};

void default_initial_depth(int n, uint32_t seed, std::vector<float>& d, std::vector<float>& s);

}  // namespace dvo

// the C handle behind dvo_batch*: a sensor-depth batch (impl) or, when `mono` is set, a mono track + map batch
struct dvo_batch {
    dvo::Batch impl;
    std::unique_ptr<dvo::MonoBatch> mono;
};
