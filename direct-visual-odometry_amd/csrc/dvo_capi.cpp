// dvo_capi.cpp -- the extern "C" surface declared in include/dvo.h.
#include <hip/hip_runtime.h>

#include <cstring>
#include <dlfcn.h>
#include <new>
#include <vector>

#include "dvo_engine.h"

using namespace dvo;

struct dvo_vo { VisualOdometry impl; };

extern "C" {

void dvo_config_default(dvo_config* c)
{
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->max_iterations = 15;              // tracker.cpp:19
    c->min_update = 5e-4f;               // tracker.cpp:17
    c->min_residual = 5e-3f;             // tracker.cpp:16
    c->fixed_iterations = 0;
    c->crop_enable = 1;
    c->step_default = 2.0f;              // optimize.cpp:22-26
    c->step_level1 = 1.5f;
    c->step_level2 = 1.0f;
    c->sigma_min = 0.01f;                // optimize.cpp:83
    c->sigma_max = 0.5f;
    c->min_depth = 0.20f;                // optimize.cpp:39
    c->keyframe_min_translation = 0.02f; // mapper.cpp:12
    c->keyframe_max_frames = 6;          // mapper.cpp:13
    c->rng_seed = 0;
    c->device = 0;
    c->stream = nullptr;
    c->profile = 0;
    c->gn_pixels_per_thread = 0;
    c->gn_use_lds_patch = -1;
    c->gn_gather_group = 0;
    c->track_streams = 0;
    c->track_adaptive = 0;
    c->track_fused_tiles = 0;
    c->track_single_launch = 0;
}

const char* dvo_version(void) { return "dvo-mi355x 0.1 (gfx950)"; }

const char* dvo_status_string(int s)
{
    switch (s) {
        case DVO_OK: return "ok";
        case DVO_ERR_BAD_ARGUMENT: return "bad argument";
        case DVO_ERR_HIP: return "HIP runtime error";
        case DVO_ERR_NO_DEVICE: return "no HIP device (libdvo has no CPU fallback)";
        case DVO_ERR_NO_VALID_PIXELS: return "no valid pixels";
        case DVO_ERR_NOT_READY: return "not ready";
        case DVO_ERR_OUT_OF_MEMORY: return "out of device memory";
        default: return "unknown status";
    }
}

const char* dvo_last_error(void) { return last_error(); }

int dvo_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------------ VisualOdometry
int dvo_vo_create(const float K[9], int width, int height, const dvo_config* cfg, dvo_vo** out)
{
    if (!out) return DVO_ERR_BAD_ARGUMENT;
    *out = nullptr;
    dvo_vo* vo = new (std::nothrow) dvo_vo();
    if (!vo) return DVO_ERR_OUT_OF_MEMORY;
    const int st = vo->impl.init(K, width, height, cfg);
    if (st != DVO_OK) { delete vo; return st; }
    *out = vo;
    return DVO_OK;
}

int dvo_vo_destroy(dvo_vo* vo)
{
    if (!vo) return DVO_OK;
    (void)select_device(vo->impl.device);
    if (vo->impl.stream) (void)hipStreamSynchronize(vo->impl.stream);
    delete vo;
    return DVO_OK;
}

int dvo_vo_set_initial_depth(dvo_vo* vo, const float* depth, const float* sigma)
{
    if (!vo || !depth || !sigma) return DVO_ERR_BAD_ARGUMENT;
    const Geometry& g = vo->impl.geoM;
    const size_t n = (size_t)g.w[g.top()] * g.h[g.top()];
    vo->impl.init_depth.assign(depth, depth + n);
    vo->impl.init_sigma.assign(sigma, sigma + n);
    return DVO_OK;
}

int dvo_vo_init_keyframe(dvo_vo* vo, const float* gray, const float* depth, const float* sigma)
{
    if (!vo) return DVO_ERR_BAD_ARGUMENT;
    return vo->impl.init_keyframe(gray, depth, sigma);
}

int dvo_vo_odometrize(dvo_vo* vo, const float* gray, float T_world[16], int* is_keyframe)
{
    if (!vo) return DVO_ERR_BAD_ARGUMENT;
    return vo->impl.odometrize(gray, T_world, is_keyframe);
}

int dvo_vo_odometrize_raw(dvo_vo* vo, const uint8_t* rgb, int channels, float T_world[16], int* is_keyframe)
{
    if (!vo || !rgb) return DVO_ERR_BAD_ARGUMENT;
    return vo->impl.odometrize(nullptr, T_world, is_keyframe, rgb, channels);
}

int dvo_vo_odometrize_depth(dvo_vo* vo, const float* gray, const float* depth, const float* sigma, float T_rel[16])
{
    if (!vo) return DVO_ERR_BAD_ARGUMENT;
    return vo->impl.odometrize_depth(gray, depth, sigma, T_rel);
}

int dvo_vo_odometrize_depth_raw(dvo_vo* vo, const uint8_t* rgb, int channels, const uint16_t* depth16, float depth_scale, float T_rel[16])
{
    if (!vo) return DVO_ERR_BAD_ARGUMENT;
    return vo->impl.odometrize_depth_raw(rgb, channels, depth16, depth_scale > 0.0f ? depth_scale : 1.0f / 5000.0f, T_rel);   // (0 = the TUM scale, as the batch entries)
}

int dvo_vo_keyframe_count(const dvo_vo* vo) { return vo ? (int)vo->impl.hist.size() : 0; }

int dvo_vo_keyframe_info(const dvo_vo* vo, int index, int* id, int* levels, int* tw, int* th, float xi[6], float rel_xi[6])
{
    if (!vo || index < 0 || index >= (int)vo->impl.hist.size()) return DVO_ERR_BAD_ARGUMENT;  // frame.hpp:176 .at()
    const Keyframe& k = *vo->impl.hist[index];
    if (id) *id = k.id;
    if (levels) *levels = k.fs.g.levels;
    if (tw) *tw = k.fs.g.w[k.fs.g.top()];
    if (th) *th = k.fs.g.h[k.fs.g.top()];
    if (xi) memcpy(xi, k.xi, 6 * sizeof(float));
    if (rel_xi) memcpy(rel_xi, k.rel_xi, 6 * sizeof(float));
    return DVO_OK;
}

int dvo_vo_keyframe_get(const dvo_vo* vo, int index, int level, float* gray, float* depth, float* sigma, float* age, float K[9])
{
    if (!vo || index < 0 || index >= (int)vo->impl.hist.size()) return DVO_ERR_BAD_ARGUMENT;
    const Keyframe& k = *vo->impl.hist[index];
    if (level < 0 || level >= k.fs.g.levels) return DVO_ERR_BAD_ARGUMENT;  // frame.hpp:125 .at()
    DVO_TRY(select_device(vo->impl.device));
    hipStream_t s = vo->impl.stream;
    const size_t n = (size_t)k.fs.g.w[level] * k.fs.g.h[level] * sizeof(float);
    if (gray) DVO_HIP(hipMemcpyAsync(gray, k.fs.gray[level], n, hipMemcpyDeviceToHost, s));
    if (depth) DVO_HIP(hipMemcpyAsync(depth, k.fs.depth[level], n, hipMemcpyDeviceToHost, s));
    if (sigma) DVO_HIP(hipMemcpyAsync(sigma, k.fs.sigma[level], n, hipMemcpyDeviceToHost, s));
    if (age) {
        if (level != k.fs.g.top()) { set_error("age is stored for the top level only"); return DVO_ERR_BAD_ARGUMENT; }
        DVO_HIP(hipMemcpyAsync(age, k.age.p, n, hipMemcpyDeviceToHost, s));
    }
    if (K) memcpy(K, k.fs.g.K9[level], 9 * sizeof(float));
    DVO_HIP(hipStreamSynchronize(s));
    return DVO_OK;
}

int dvo_vo_last_frame_pose(const dvo_vo* vo, int* id, float xi[6], float rel_xi[6])
{
    if (!vo) return DVO_ERR_BAD_ARGUMENT;
    if (vo->impl.last_id < 0) return DVO_ERR_NOT_READY;
    if (id) *id = vo->impl.last_id;
    if (xi) memcpy(xi, vo->impl.last_xi, 6 * sizeof(float));
    if (rel_xi) memcpy(rel_xi, vo->impl.last_rel, 6 * sizeof(float));
    return DVO_OK;
}

// diagnostic (DVO_PERSIST_TIMELINE=1): wall-clock stamps [2][64][8] of the last k_track_persist launch of the sensor-depth tracker
int dvo_debug_persist_timeline(dvo_vo* vo, long long* out)
{
    if (!vo || !out) return DVO_ERR_BAD_ARGUMENT;
    return vo->impl.trkD.read_persist_timeline(out);
}

int dvo_vo_last_valid_updates(const dvo_vo* vo)
{
    if (!vo) return 0;
    (void)const_cast<dvo_vo*>(vo)->impl.fetch_valid_updates();   // (read back on demand)
    return vo->impl.last_valid_updates;
}

int dvo_vo_last_track_log(const dvo_vo* vo, dvo_track_log* log)
{
    if (!vo || !log) return DVO_ERR_BAD_ARGUMENT;
    DVO_TRY(const_cast<dvo_vo*>(vo)->impl.fetch_log());   // (read back on demand: the record is not part of the per-frame hand-over)
    *log = vo->impl.last_log;
    return DVO_OK;
}

// ------------------------------------------------------------------------------------------------ batch
int dvo_batch_create(int n_seq, const float K[9], int width, int height, int levels, int culls, const dvo_config* cfg, dvo_batch** out)
{
    if (!out) return DVO_ERR_BAD_ARGUMENT;
    *out = nullptr;
    dvo_batch* b = new (std::nothrow) dvo_batch();
    if (!b) return DVO_ERR_OUT_OF_MEMORY;
    const int st = b->impl.init(n_seq, K, width, height, levels, culls, cfg);
    if (st != DVO_OK) { delete b; return st; }
    *out = b;
    return DVO_OK;
}

int dvo_batch_destroy(dvo_batch* b)
{
    if (!b) return DVO_OK;
    if (b->mono) {
        (void)select_device(b->mono->device);
        if (b->mono->stream) (void)hipStreamSynchronize(b->mono->stream);
    } else {
        (void)select_device(b->impl.device);
        if (b->impl.stream) (void)hipStreamSynchronize(b->impl.stream);
    }
    delete b;
    return DVO_OK;
}

// a mono batch (dvo_batch_create_mono) has no sensor-depth entry points
#define DVO_NOT_MONO(b)                                                                                   \
    do {                                                                                                  \
        if ((b)->mono) { set_error("this entry point needs a sensor-depth batch (dvo_batch_create)"); return DVO_ERR_BAD_ARGUMENT; } \
    } while (0)

int dvo_batch_push_device(dvo_batch* b, const float* gray, const float* depth, const float* sigma)
{
    if (!b) return DVO_ERR_BAD_ARGUMENT;
    DVO_NOT_MONO(b);
    FrameInput in;
    in.gray = gray; in.depth = depth; in.sigma = sigma;
    return b->impl.push(in);
}

int dvo_batch_prefetch_device(dvo_batch* b, const float* gray, const float* depth, const float* sigma)
{
    if (!b) return DVO_ERR_BAD_ARGUMENT;
    DVO_NOT_MONO(b);
    FrameInput in;
    in.gray = gray; in.depth = depth; in.sigma = sigma;
    return b->impl.prefetch(in);
}

static int raw_input(dvo_batch* b, const uint8_t* rgb, int channels, const uint16_t* depth16, float depth_scale, FrameInput& in)
{
    if (!b || !rgb || !depth16 || (channels != 1 && channels != 3 && channels != 4)) { set_error("bad raw frame"); return DVO_ERR_BAD_ARGUMENT; }
    in.rgb = rgb; in.channels = channels; in.depth16 = depth16; in.depth_scale = depth_scale > 0.0f ? depth_scale : 1.0f / 5000.0f;
    return DVO_OK;
}

int dvo_batch_push_raw_device(dvo_batch* b, const uint8_t* rgb_dev, int channels, const uint16_t* depth16_dev, float depth_scale)
{
    if (!b) return DVO_ERR_BAD_ARGUMENT;
    DVO_NOT_MONO(b);
    FrameInput in;
    DVO_TRY(raw_input(b, rgb_dev, channels, depth16_dev, depth_scale, in));
    return b->impl.push(in);
}

int dvo_batch_prefetch_raw_device(dvo_batch* b, const uint8_t* rgb_dev, int channels, const uint16_t* depth16_dev, float depth_scale)
{
    if (!b) return DVO_ERR_BAD_ARGUMENT;
    DVO_NOT_MONO(b);
    FrameInput in;
    DVO_TRY(raw_input(b, rgb_dev, channels, depth16_dev, depth_scale, in));
    return b->impl.prefetch(in);
}

int dvo_batch_push_raw_host(dvo_batch* b, const uint8_t* rgb, int channels, const uint16_t* depth16, float depth_scale)
{
    if (!b) return DVO_ERR_BAD_ARGUMENT;
    DVO_NOT_MONO(b);
    FrameInput in;
    DVO_TRY(raw_input(b, rgb, channels, depth16, depth_scale, in));
    Batch& B = b->impl;
    const size_t px = (size_t)B.n_seq * B.g.src_w * B.g.src_h;
    return B.push_host_frame(rgb, px * (size_t)channels, depth16, px * 2, nullptr, 0, in);
}

int dvo_batch_push_host(dvo_batch* b, const float* gray, const float* depth, const float* sigma)
{
    if (!b || !gray || !depth || !sigma) return DVO_ERR_BAD_ARGUMENT;
    DVO_NOT_MONO(b);
    Batch& B = b->impl;
    const size_t n = (size_t)B.n_seq * B.g.src_w * B.g.src_h * sizeof(float);
    FrameInput in;
    in.gray = gray; in.depth = depth; in.sigma = sigma;   // (replaced by the staging pointers)
    return B.push_host_frame(gray, n, depth, n, sigma, n, in);
}

int dvo_batch_last_poses(dvo_batch* b, float* xi_rel, float* T_rel)
{
    if (!b) return DVO_ERR_BAD_ARGUMENT;
    DVO_NOT_MONO(b);
    Batch& B = b->impl;
    if (!B.have_poses) return DVO_ERR_NOT_READY;
    DVO_TRY(select_device(B.device));
    if (xi_rel) DVO_HIP(hipMemcpyAsync(xi_rel, B.trk.xi_out.p, sizeof(float) * 6 * (size_t)B.n_seq, hipMemcpyDeviceToHost, B.stream));
    if (T_rel) DVO_HIP(hipMemcpyAsync(T_rel, B.trk.T_out.p, sizeof(float) * 16 * (size_t)B.n_seq, hipMemcpyDeviceToHost, B.stream));
    DVO_HIP(hipStreamSynchronize(B.stream));
    return DVO_OK;
}

int dvo_batch_copy_poses_device(dvo_batch* b, float* xi_dst_dev, float* T_dst_dev)
{
    if (!b) return DVO_ERR_BAD_ARGUMENT;
    DVO_NOT_MONO(b);
    Batch& B = b->impl;
    if (!B.have_poses) return DVO_ERR_NOT_READY;
    DVO_TRY(select_device(B.device));
    if (xi_dst_dev) DVO_HIP(hipMemcpyAsync(xi_dst_dev, B.trk.xi_out.p, sizeof(float) * 6 * (size_t)B.n_seq, hipMemcpyDeviceToDevice, B.stream));
    if (T_dst_dev) DVO_HIP(hipMemcpyAsync(T_dst_dev, B.trk.T_out.p, sizeof(float) * 16 * (size_t)B.n_seq, hipMemcpyDeviceToDevice, B.stream));
    return DVO_OK;
}

int dvo_batch_last_track_log(dvo_batch* b, int seq, dvo_track_log* log)
{
    if (!b || !log || seq < 0) return DVO_ERR_BAD_ARGUMENT;
    if (b->mono) {  // the log of the last tracked frame of a mono batch
        MonoBatch& M = *b->mono;
        if (seq >= M.n_seq) return DVO_ERR_BAD_ARGUMENT;
        if (M.latest_id < 1) return DVO_ERR_NOT_READY;
        DVO_TRY(select_device(M.device));
        DVO_HIP(hipMemcpyAsync(log, M.trk.log.as<dvo_track_log>() + seq, sizeof *log, hipMemcpyDeviceToHost, M.stream));
        DVO_HIP(hipStreamSynchronize(M.stream));
        return DVO_OK;
    }
    if (seq >= b->impl.n_seq) return DVO_ERR_BAD_ARGUMENT;
    Batch& B = b->impl;
    if (!B.have_poses) return DVO_ERR_NOT_READY;
    DVO_TRY(select_device(B.device));
    DVO_HIP(hipMemcpyAsync(log, B.trk.log.as<dvo_track_log>() + seq, sizeof *log, hipMemcpyDeviceToHost, B.stream));
    DVO_HIP(hipStreamSynchronize(B.stream));
    return DVO_OK;
}

int dvo_batch_synchronize(dvo_batch* b)
{
    if (!b) return DVO_ERR_BAD_ARGUMENT;
    if (b->mono) {
        DVO_TRY(select_device(b->mono->device));
        DVO_HIP(hipStreamSynchronize(b->mono->stream));
        return DVO_OK;
    }
    DVO_TRY(select_device(b->impl.device));
    DVO_HIP(hipStreamSynchronize(b->impl.stream));
    return DVO_OK;
}

int dvo_shard_range(int n_sequences, int world_size, int rank, int* first, int* count)
{
    if (n_sequences < 0 || world_size < 1 || rank < 0 || rank >= world_size || !first || !count) return DVO_ERR_BAD_ARGUMENT;
    const int base = n_sequences / world_size, extra = n_sequences % world_size;
    *count = base + (rank < extra ? 1 : 0);
    *first = rank * base + (rank < extra ? rank : extra);
    return DVO_OK;
}

int dvo_batch_gather_poses_rccl(dvo_batch* b, void* rccl_comm, int world_size, float* xi_all_dev)
{
    if (!b || !rccl_comm || world_size < 1 || !xi_all_dev) return DVO_ERR_BAD_ARGUMENT;
    // ncclAllGather(sendbuff, recvbuff, sendcount, datatype, comm, stream); ncclFloat = 7 (rccl.h).  Resolved at run time so that
    // libdvo.so has no link-time dependency on a collective library it needs on multi-GPU hosts only.
    typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
    static all_gather_fn all_gather = nullptr;
    if (!all_gather) {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (h) all_gather = reinterpret_cast<all_gather_fn>(dlsym(h, "ncclAllGather"));
        if (!all_gather) { set_error("librccl.so (ncclAllGather) could not be loaded"); return DVO_ERR_NOT_READY; }
    }
    const float* src; size_t n; hipStream_t st; int dev;
    if (b->mono) {
        if (b->mono->latest_id < 0) return DVO_ERR_NOT_READY;
        src = b->mono->xi_world.as<float>(); n = (size_t)b->mono->n_seq * 6; st = b->mono->stream; dev = b->mono->device;
    } else {
        if (!b->impl.have_poses) return DVO_ERR_NOT_READY;
        src = b->impl.trk.xi_out.as<float>(); n = (size_t)b->impl.n_seq * 6; st = b->impl.stream; dev = b->impl.device;
    }
    DVO_TRY(select_device(dev));
    const int rc = all_gather(src, xi_all_dev, n, 7 /* ncclFloat */, rccl_comm, st);
    if (rc != 0) { set_error("ncclAllGather failed"); return DVO_ERR_HIP; }
    return DVO_OK;
}

int dvo_batch_profile(dvo_batch* b, dvo_gn_profile* out, int reset)
{
    if (!b || !out) return DVO_ERR_BAD_ARGUMENT;
    Tracker& trk = b->mono ? b->mono->trk : b->impl.trk;
    hipStream_t st = b->mono ? b->mono->stream : b->impl.stream;
    DVO_TRY(select_device(b->mono ? b->mono->device : b->impl.device));
    DVO_TRY(trk.collect_profile(st));
    unsigned long long c[2] = {0, 0};
    DVO_HIP(hipMemcpy(c, trk.counters.p, sizeof c, hipMemcpyDeviceToHost));
    out->gn_ms = trk.prof_ms;
    out->gn_launches = trk.prof_launches;
    out->gn_pixels = c[0];
    out->gn_iterations = c[1];
    if (reset) {
        trk.prof_ms = 0; trk.prof_launches = 0;
        DVO_HIP(hipMemset(trk.counters.p, 0, sizeof c));
    }
    return DVO_OK;
}

int dvo_batch_probe_gn(dvo_batch* b, int level, int n_launches, float* avg_ms, uint64_t* pixels_per_launch)
{
    if (!b || !avg_ms || n_launches < 1) return DVO_ERR_BAD_ARGUMENT;
    DVO_NOT_MONO(b);
    Batch& B = b->impl;
    if (level < 0 || level >= B.g.levels) return DVO_ERR_BAD_ARGUMENT;
    if (B.cur < 0 || !B.have_poses) return DVO_ERR_NOT_READY;
    DVO_TRY(select_device(B.device));
    // obj = the newest frame set, ref = the one before it: exactly the operands of the last track() call
    const GnArgs ga = B.trk.gn_args(B.fs[B.cur], B.fs[B.prev], level, nullptr, 1);
    hipEvent_t e0, e1;
    DVO_HIP(hipEventCreate(&e0));
    DVO_HIP(hipEventCreate(&e1));
    B.trk.launch_gn(ga, level, B.trk.n_seq, B.stream);  // warm
    DVO_HIP(hipEventRecord(e0, B.stream));
    for (int i = 0; i < n_launches; i++) B.trk.launch_gn(ga, level, B.trk.n_seq, B.stream);
    DVO_HIP(hipEventRecord(e1, B.stream));
    DVO_HIP(hipEventSynchronize(e1));
    float ms = 0;
    DVO_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / (float)n_launches;
    if (pixels_per_launch) *pixels_per_launch = (uint64_t)B.n_seq * B.g.w[level] * B.g.h[level];
    return DVO_OK;
}

// ------------------------------------------------------------------------------------------------ operator level
namespace {
struct OpCtx {  // device selection + a private stream for one operator call
    hipStream_t s = nullptr;
    int open(int dev)
    {
        DVO_TRY(select_device(dev));
        DVO_HIP(hipStreamCreate(&s));
        return DVO_OK;
    }
    ~OpCtx() { if (s) (void)hipStreamDestroy(s); }
};
int upload(DevBuf& b, const float* host, size_t count, hipStream_t s)
{
    DVO_TRY(b.alloc(count * sizeof(float)));
    DVO_HIP(hipMemcpyAsync(b.p, host, count * sizeof(float), hipMemcpyHostToDevice, s));
    return DVO_OK;
}
int download(float* host, const void* dev, size_t count, hipStream_t s)
{
    DVO_HIP(hipMemcpyAsync(host, dev, count * sizeof(float), hipMemcpyDeviceToHost, s));
    return DVO_OK;
}
Intr intr_of(const float K[9]) { return make_intr(K); }
}  // namespace

int dvo_op_cull_image(int dev, const float* src, int w, int h, int times, float* dst)
{
    if (!src || !dst || w < 1 || h < 1 || times < 0 || times > 8) return DVO_ERR_BAD_ARGUMENT;
    OpCtx c; DVO_TRY(c.open(dev));
    const size_t n = (size_t)w * h, dn = (size_t)(w >> times) * (h >> times);
    if (dn == 0) return DVO_OK;  // empty output, as cv::Mat::zeros(0 x 0)
    DevBuf a, b;
    DVO_TRY(upload(a, src, n, c.s));
    DVO_TRY(b.alloc(dn * 4));
    launch_cull(a.as<float>(), w, h, times, b.as<float>(), c.s);
    DVO_TRY(download(dst, b.p, dn, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

int dvo_op_gradient(int dev, const float* img, int w, int h, int xdir, float* out)
{
    if (!img || !out || w < 1 || h < 1) return DVO_ERR_BAD_ARGUMENT;
    OpCtx c; DVO_TRY(c.open(dev));
    const size_t n = (size_t)w * h;
    DevBuf a, b;
    DVO_TRY(upload(a, img, n, c.s));
    DVO_TRY(b.alloc(n * 4));
    launch_gradient(a.as<float>(), w, h, xdir, b.as<float>(), c.s);
    DVO_TRY(download(out, b.p, n, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

int dvo_op_warp_image(int dev, const float xi[6], const float* gray, const float* depth, int w, int h, const float K[9], float* out)
{
    if (!xi || !gray || !depth || !K || !out || w < 1 || h < 1) return DVO_ERR_BAD_ARGUMENT;
    OpCtx c; DVO_TRY(c.open(dev));
    const size_t n = (size_t)w * h;
    DevBuf g, d, o, xin, T;
    DVO_TRY(upload(g, gray, n, c.s));
    DVO_TRY(upload(d, depth, n, c.s));
    DVO_TRY(o.alloc(n * 4));
    // the pose comes from the DEVICE exp (that is what the tracker uses): exp(-xi)
    float neg[6];
    for (int i = 0; i < 6; i++) neg[i] = -xi[i];
    DVO_TRY(upload(xin, neg, 6, c.s));
    DVO_TRY(T.alloc(16 * 4));
    launch_se3(0, xin.as<float>(), nullptr, T.as<float>(), c.s);
    float Th[16];
    DVO_TRY(download(Th, T.p, 16, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    Pose pose;
    for (int r = 0; r < 3; r++) {
        for (int q = 0; q < 3; q++) pose.R[3 * r + q] = Th[4 * r + q];
        pose.t[r] = Th[4 * r + 3];
    }
    launch_warp_image(g.as<float>(), d.as<float>(), w, h, intr_of(K), pose, o.as<float>(), c.s);
    DVO_TRY(download(out, o.p, n, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

int dvo_op_pyramid(int dev, const float* gray, const float* depth, const float* sigma, int w, int h, int levels, int culls,
                   float* const gray_out[], float* const depth_out[], float* const sigma_out[])
{
    if (!gray || !gray_out) return DVO_ERR_BAD_ARGUMENT;
    OpCtx c; DVO_TRY(c.open(dev));
    const float Kid[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    Geometry g;
    DVO_TRY(make_geometry(Kid, w, h, levels, culls, g));
    FrameSet fs;
    dvo_config dc;
    dvo_config_default(&dc);
    DVO_TRY(fs.alloc(g, 1, dc));
    const size_t n = (size_t)w * h;
    DevBuf a, b, s3;
    DVO_TRY(upload(a, gray, n, c.s));
    if (depth) DVO_TRY(upload(b, depth, n, c.s));
    if (sigma) DVO_TRY(upload(s3, sigma, n, c.s));
    build_pyramid(fs, a.as<float>(), depth ? b.as<float>() : nullptr, sigma ? s3.as<float>() : nullptr, c.s);
    for (int l = 0; l < levels; l++) {
        const size_t ln = (size_t)g.w[l] * g.h[l];
        if (gray_out[l]) DVO_TRY(download(gray_out[l], fs.gray[l], ln, c.s));
        if (depth && depth_out && depth_out[l]) DVO_TRY(download(depth_out[l], fs.depth[l], ln, c.s));
        if (sigma && sigma_out && sigma_out[l]) DVO_TRY(download(sigma_out[l], fs.sigma[l], ln, c.s));
    }
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

int dvo_op_gn_step(int dev, const dvo_config* cfg, const float* obj_gray, const float* ref_gray, const float* ref_depth,
                   const float* ref_sigma, int w, int h, const float K[9], const float xi[6], int level,
                   dvo_gn_result* out, uint8_t* mask)
{
    if (!obj_gray || !ref_gray || !ref_depth || !ref_sigma || !K || !xi || !out || w < 1 || h < 1 || level < 0 || level >= DVO_MAX_LEVELS)
        return DVO_ERR_BAD_ARGUMENT;
    dvo_config cf;
    if (cfg) cf = *cfg; else dvo_config_default(&cf);
    OpCtx c; DVO_TRY(c.open(dev));
    // a one-level "pyramid" whose only level carries index `level` semantics (step size, crop)
    Geometry g;
    DVO_TRY(make_geometry(K, w, h, 1, 0, g));
    // place the level at index `level` so Tracker::level_params / logs see the right index
    Geometry gl = g;
    gl.levels = level + 1;
    for (int l = 0; l <= level; l++) { gl.w[l] = w; gl.h[l] = h; memcpy(gl.K9[l], g.K9[0], sizeof g.K9[0]); gl.k[l] = g.k[0]; }
    Tracker trk;
    DVO_TRY(trk.init(gl, 1, cf));
    const size_t n = (size_t)w * h;
    DevBuf og, rg, rd, rs, mk, xin, res;
    DVO_TRY(upload(og, obj_gray, n, c.s));
    DVO_TRY(upload(rg, ref_gray, n, c.s));
    DVO_TRY(upload(rd, ref_depth, n, c.s));
    DVO_TRY(upload(rs, ref_sigma, n, c.s));
    DVO_TRY(upload(xin, xi, 6, c.s));
    DVO_TRY(res.alloc(sizeof(dvo_gn_result)));
    if (mask) { DVO_TRY(mk.alloc(n)); DVO_HIP(hipMemsetAsync(mk.p, 0, n, c.s)); }
    launch_set_pose(trk.state.as<SeqState>(), xin.as<float>(), 1, c.s);
    GnArgs ga;
    // per-pixel constants of the reference level (what build_pyramid does for whole frames)
    DevBuf wgb;
    DVO_TRY(wgb.alloc(n * 4));
    {
        PrepArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.depth = rd.as<float>(); pa.sigma = rs.as<float>(); pa.wgt = wgb.as<float>();
        pa.level_end[0] = n;
        pa.step[0] = trk.level_params(level).step;
        pa.sigma_min = cf.sigma_min; pa.sigma_max = cf.sigma_max;
        pa.levels = 1;
        launch_prep_ref(pa, c.s);
    }
    ga.obj_gray = og.as<float>(); ga.ref_gray = rg.as<float>(); ga.ref_depth = rd.as<float>();
    ga.ref_wgt = wgb.as<float>();
    ga.state = trk.state.as<SeqState>();
    ga.partials = trk.partials.as<float>();
    ga.mask = mask ? mk.as<uint8_t>() : nullptr;
    ga.w = w; ga.h = h; ga.nblk = trk.nblk[level]; ga.inv_w = 1.0f / (float)w;
    ga.q256 = 256 / w; ga.r256 = 256 % w;
    ga.k = gl.k[level];
    ga.prm = trk.level_params(level);
    ga.ignore_active = 1;
    ga.tiles_x = trk.tiles_x[level]; ga.tiles_y = trk.tiles_y[level]; ga.margin = trk.tile_margin;
    trk.launch_gn(ga, level, 1, c.s);
    SolveArgs sa;
    sa.state = trk.state.as<SeqState>(); sa.partials = trk.partials.as<float>();
    sa.log = nullptr; sa.result = res.as<dvo_gn_result>(); sa.counters = nullptr;
    sa.nblk = trk.nblk[level]; sa.level = level; sa.level_pixels = w * h;
    sa.max_iterations = cf.max_iterations; sa.fixed_iterations = cf.fixed_iterations;
    sa.min_update = cf.min_update; sa.min_residual = cf.min_residual; sa.ignore_active = 1;
    if (trk.tile_margin == 0) {
        const GnTiling tl = gn_tiling(w, h, trk.ppt[level], ga.prm.crop);
        sa.blk_first = tl.live_first; sa.blk_count = tl.live_count;
    }
    launch_gn_solve(sa, 1, c.s);
    DVO_HIP(hipMemcpyAsync(out, res.p, sizeof *out, hipMemcpyDeviceToHost, c.s));
    if (mask) DVO_HIP(hipMemcpyAsync(mask, mk.p, n, hipMemcpyDeviceToHost, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    DVO_HIP(hipGetLastError());
    return DVO_OK;
}

int dvo_op_track(int dev, const dvo_config* cfg, const float* obj_gray, const float* ref_gray, const float* ref_depth,
                 const float* ref_sigma, int w, int h, const float K[9], int levels, int culls, float xi_out[6], dvo_track_log* log)
{
    if (!obj_gray || !ref_gray || !ref_depth || !ref_sigma || !K || !xi_out) return DVO_ERR_BAD_ARGUMENT;
    dvo_config cf;
    if (cfg) cf = *cfg; else dvo_config_default(&cf);
    OpCtx c; DVO_TRY(c.open(dev));
    Geometry g;
    DVO_TRY(make_geometry(K, w, h, levels, culls, g));
    FrameSet obj, ref;
    DVO_TRY(obj.alloc(g, 1, cf));
    DVO_TRY(ref.alloc(g, 1, cf));
    Tracker trk;
    DVO_TRY(trk.init(g, 1, cf));
    const size_t n = (size_t)w * h;
    DevBuf og, rg, rd, rs;
    DVO_TRY(upload(og, obj_gray, n, c.s));
    DVO_TRY(upload(rg, ref_gray, n, c.s));
    DVO_TRY(upload(rd, ref_depth, n, c.s));
    DVO_TRY(upload(rs, ref_sigma, n, c.s));
    build_pyramid(obj, og.as<float>(), nullptr, nullptr, c.s);
    build_pyramid(ref, rg.as<float>(), rd.as<float>(), rs.as<float>(), c.s);
    DVO_TRY(trk.track(obj, ref, c.s));
    DVO_TRY(download(xi_out, trk.xi_out.p, 6, c.s));
    if (log) DVO_HIP(hipMemcpyAsync(log, trk.log.p, sizeof *log, hipMemcpyDeviceToHost, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

int dvo_op_propagate(int dev, const float* ref_depth, const float* ref_sigma, const float* ref_age, int w, int h,
                     const float xi[6], const float K[9], float* depth, float* sigma, float* age)
{
    if (!ref_depth || !ref_sigma || !ref_age || !xi || !K || !depth || !sigma || !age || w < 1 || h < 1) return DVO_ERR_BAD_ARGUMENT;
    OpCtx c; DVO_TRY(c.open(dev));
    const size_t n = (size_t)w * h;
    DevBuf a, b, g, od, os, oa, ow;
    DVO_TRY(upload(a, ref_depth, n, c.s));
    DVO_TRY(upload(b, ref_sigma, n, c.s));
    DVO_TRY(upload(g, ref_age, n, c.s));
    DVO_TRY(od.alloc(n * 4)); DVO_TRY(os.alloc(n * 4)); DVO_TRY(oa.alloc(n * 4)); DVO_TRY(ow.alloc(n * 4));
    Pose pose;
    pose_from_xi(xi, 1.0f, pose);  // host double exp, as VisualOdometry::map_propagate does
    launch_propagate(a.as<float>(), b.as<float>(), g.as<float>(), w, h, intr_of(K), pose, xi[2], ow.as<int>(),
                     od.as<float>(), os.as<float>(), oa.as<float>(), c.s);
    DVO_TRY(download(depth, od.p, n, c.s));
    DVO_TRY(download(sigma, os.p, n, c.s));
    DVO_TRY(download(age, oa.p, n, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

int dvo_op_regularize(int dev, const float* depth, const float* sigma, int w, int h, float* out)
{
    if (!depth || !sigma || !out || w < 1 || h < 1) return DVO_ERR_BAD_ARGUMENT;
    OpCtx c; DVO_TRY(c.open(dev));
    const size_t n = (size_t)w * h;
    DevBuf a, b, o;
    DVO_TRY(upload(a, depth, n, c.s));
    DVO_TRY(upload(b, sigma, n, c.s));
    DVO_TRY(o.alloc(n * 4));
    launch_regularize(a.as<float>(), b.as<float>(), w, h, o.as<float>(), c.s);
    DVO_TRY(download(out, o.p, n, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

int dvo_op_depth_update(int dev, const dvo_config* cfg, int n_hist, const float* const hist_gray[], const float* hist_xi,
                        const float* obj_gray, const float obj_xi[6], const float obj_rel_xi[6], int obj_id,
                        const float K[9], int w, int h, float* ref_depth, float* ref_sigma, float* ref_age, int* valid_updates)
{
    if (n_hist < 1 || !hist_gray || !hist_xi || !obj_gray || !obj_xi || !obj_rel_xi || !K || !ref_depth || !ref_sigma || !ref_age)
        return DVO_ERR_BAD_ARGUMENT;
    dvo_config cf;
    if (cfg) cf = *cfg; else dvo_config_default(&cf);
    OpCtx c; DVO_TRY(c.open(dev));
    const size_t n = (size_t)w * h;
    std::vector<DevBuf> grays(n_hist);
    std::vector<AgeEntry> tab(n_hist);
    std::vector<const float*> gptr(n_hist);
    for (int i = 0; i < n_hist; i++) {
        DVO_TRY(upload(grays[i], hist_gray[i], n, c.s));
        float nb[6], r_xi[6];
        for (int k = 0; k < 6; k++) nb[k] = -hist_xi[6 * i + k];
        se3_concatenate_f(obj_xi, nb, r_xi);
        pose_from_xi(r_xi, -1.0f, tab[i].pose);
        for (int k = 0; k < 3; k++) tab[i].tneg[k] = -r_xi[k];
        tab[i].slot = i;
        gptr[i] = grays[i].as<float>();
    }
    DevBuf og, rd, rs, ra, ages, vd, gtab;
    DVO_TRY(upload(og, obj_gray, n, c.s));
    DVO_TRY(upload(rd, ref_depth, n, c.s));
    DVO_TRY(upload(rs, ref_sigma, n, c.s));
    DVO_TRY(upload(ra, ref_age, n, c.s));
    DVO_TRY(ages.alloc(sizeof(AgeEntry) * (size_t)n_hist));
    DVO_HIP(hipMemcpyAsync(ages.p, tab.data(), sizeof(AgeEntry) * (size_t)n_hist, hipMemcpyHostToDevice, c.s));
    DVO_TRY(gtab.alloc(sizeof(float*) * (size_t)n_hist));
    DVO_HIP(hipMemcpyAsync(gtab.p, gptr.data(), sizeof(float*) * (size_t)n_hist, hipMemcpyHostToDevice, c.s));
    DVO_TRY(vd.alloc(sizeof(int)));
    DVO_HIP(hipMemsetAsync(vd.p, 0, sizeof(int), c.s));
    UpdateArgs a;
    memset(&a, 0, sizeof a);
    a.ref_depth = rd.as<float>(); a.ref_sigma = rs.as<float>(); a.ref_age = ra.as<float>();
    a.obj_gray = og.as<float>(); a.ages = ages.as<AgeEntry>();
    a.ring_gray = nullptr; a.gray_table = gtab.as<const float*>(); a.meta = nullptr;
    a.n_seq = 1; a.R = n_hist;
    a.n_hist = n_hist; a.w = w; a.h = h; a.crop = cf.crop_enable; a.obj_id = obj_id; a.seed = cf.rng_seed;
    a.clamp_age = 0;
    a.k = intr_of(K);
    memcpy(a.K9, K, sizeof a.K9);
    pose_from_xi(obj_rel_xi, 1.0f, a.rel_pose);
    a.rel_tz = obj_rel_xi[2];
    a.valid_updates = vd.as<int>();
    launch_depth_update(a, c.s);
    DVO_TRY(download(ref_depth, rd.p, n, c.s));
    DVO_TRY(download(ref_sigma, rs.p, n, c.s));
    DVO_TRY(download(ref_age, ra.p, n, c.s));
    int v = 0;
    DVO_HIP(hipMemcpyAsync(&v, vd.p, sizeof v, hipMemcpyDeviceToHost, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    if (valid_updates) *valid_updates = v;
    return DVO_OK;
}

int dvo_op_ingest(int dev, const uint8_t* rgb, int channels, const uint16_t* depth16, int w, int h, float depth_scale,
                  float sigma_valid, float sigma_invalid, int invalidate_gray, float* gray, float* depth, float* sigma)
{
    if (!rgb || !gray || w < 1 || h < 1 || (channels != 1 && channels != 3 && channels != 4)) return DVO_ERR_BAD_ARGUMENT;
    if (depth16 && (!depth || !sigma)) return DVO_ERR_BAD_ARGUMENT;
    OpCtx c; DVO_TRY(c.open(dev));
    const size_t n = (size_t)w * h;
    DevBuf r, d16, g, d, s;
    DVO_TRY(r.alloc(n * channels));
    DVO_HIP(hipMemcpyAsync(r.p, rgb, n * channels, hipMemcpyHostToDevice, c.s));
    if (depth16) {
        DVO_TRY(d16.alloc(n * 2));
        DVO_HIP(hipMemcpyAsync(d16.p, depth16, n * 2, hipMemcpyHostToDevice, c.s));
        DVO_TRY(d.alloc(n * 4)); DVO_TRY(s.alloc(n * 4));
    }
    DVO_TRY(g.alloc(n * 4));
    launch_ingest(r.as<uint8_t>(), channels, depth16 ? d16.as<uint16_t>() : nullptr, (int)n, depth_scale, sigma_valid, sigma_invalid,
                  invalidate_gray, g.as<float>(), d.as<float>(), s.as<float>(), c.s);
    DVO_TRY(download(gray, g.p, n, c.s));
    if (depth16) { DVO_TRY(download(depth, d.p, n, c.s)); DVO_TRY(download(sigma, s.p, n, c.s)); }
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

int dvo_op_visualize(int dev, int mode, const float* a, const float* b, int w, int h, uint8_t* rgb)
{
    if (!a || !rgb || w < 1 || h < 1 || mode < 0 || mode > 4) return DVO_ERR_BAD_ARGUMENT;
    OpCtx c; DVO_TRY(c.open(dev));
    const size_t n = (size_t)w * h;
    DevBuf da, db, out;
    DVO_TRY(upload(da, a, n, c.s));
    if (b) DVO_TRY(upload(db, b, n, c.s));
    DVO_TRY(out.alloc(n * 3));
    launch_visualize(mode, da.as<float>(), b ? db.as<float>() : nullptr, (int)n, out.as<uint8_t>(), c.s);
    DVO_HIP(hipMemcpyAsync(rgb, out.p, n * 3, hipMemcpyDeviceToHost, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

int dvo_op_undistort(int dev, const float* src, int w, int h, const float K[9], const float D[5], float* dst)
{
    if (!src || !dst || !K || !D || w < 1 || h < 1) return DVO_ERR_BAD_ARGUMENT;
    OpCtx c; DVO_TRY(c.open(dev));
    const size_t n = (size_t)w * h;
    DevBuf a, b;
    DVO_TRY(upload(a, src, n, c.s));
    DVO_TRY(b.alloc(n * 4));
    launch_undistort(a.as<float>(), w, h, intr_of(K), D, DVO_INVALID, b.as<float>(), c.s);
    DVO_TRY(download(dst, b.p, n, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

static int se3_op(int dev, int op, const float* a, int na, const float* b, float* out, int nout)
{
    OpCtx c; DVO_TRY(c.open(dev));
    DevBuf da, db, dout;
    DVO_TRY(upload(da, a, na, c.s));
    if (b) DVO_TRY(upload(db, b, 6, c.s));
    DVO_TRY(dout.alloc(16 * 4));
    launch_se3(op, da.as<float>(), b ? db.as<float>() : nullptr, dout.as<float>(), c.s);
    DVO_TRY(download(out, dout.p, nout, c.s));
    DVO_HIP(hipStreamSynchronize(c.s));
    return DVO_OK;
}

int dvo_op_se3_exp(int dev, const float xi[6], float T[16]) { return (xi && T) ? se3_op(dev, 0, xi, 6, nullptr, T, 16) : DVO_ERR_BAD_ARGUMENT; }
int dvo_op_se3_log(int dev, const float T[16], float xi[6]) { return (xi && T) ? se3_op(dev, 1, T, 16, nullptr, xi, 6) : DVO_ERR_BAD_ARGUMENT; }
int dvo_op_se3_concatenate(int dev, const float a[6], const float b[6], float out[6])
{
    return (a && b && out) ? se3_op(dev, 2, a, 6, b, out, 6) : DVO_ERR_BAD_ARGUMENT;
}

int dvo_selftest_reciprocal(int dev, uint64_t* fast_path_inputs, uint64_t* mismatches, uint32_t* first_bad_bits)
{
    if (!mismatches) return DVO_ERR_BAD_ARGUMENT;
    DVO_TRY(select_device(dev));
    DevBuf out;
    DVO_TRY(out.alloc(3 * sizeof(unsigned long long)));
    unsigned long long h[3] = {0ull, 0ull, ~0ull};
    DVO_HIP(hipMemcpy(out.p, h, sizeof h, hipMemcpyHostToDevice));
    launch_selftest_reciprocal(out.as<unsigned long long>(), nullptr);
    DVO_HIP(hipDeviceSynchronize());
    DVO_HIP(hipMemcpy(h, out.p, sizeof h, hipMemcpyDeviceToHost));
    if (fast_path_inputs) *fast_path_inputs = h[0];
    *mismatches = h[1];
    if (first_bad_bits) *first_bad_bits = (uint32_t)h[2];
    return DVO_OK;
}

int dvo_selftest_trig(int dev, double* max_rel_sin, double* max_rel_cos, double* max_rel_atan2, uint64_t* samples)
{
    if (!max_rel_sin || !max_rel_cos || !max_rel_atan2) return DVO_ERR_BAD_ARGUMENT;
    DVO_TRY(select_device(dev));
    DevBuf out;
    DVO_TRY(out.alloc(3 * sizeof(unsigned long long)));
    DVO_HIP(hipMemset(out.p, 0, out.bytes));
    const unsigned side = 4096, n = side * side;
    launch_selftest_trig(n, side, out.as<unsigned long long>(), nullptr);
    DVO_HIP(hipDeviceSynchronize());
    unsigned long long h[3];
    DVO_HIP(hipMemcpy(h, out.p, sizeof h, hipMemcpyDeviceToHost));
    memcpy(max_rel_sin, &h[0], 8); memcpy(max_rel_cos, &h[1], 8); memcpy(max_rel_atan2, &h[2], 8);
    if (samples) *samples = n;
    return DVO_OK;
}

int dvo_selftest_sqrt(int dev, uint64_t* inputs, uint64_t* mismatches, uint32_t* first_bad_bits)
{
    if (!mismatches) return DVO_ERR_BAD_ARGUMENT;
    DVO_TRY(select_device(dev));
    DevBuf out;
    DVO_TRY(out.alloc(3 * sizeof(unsigned long long)));
    unsigned long long h[3] = {0ull, 0ull, ~0ull};
    DVO_HIP(hipMemcpy(out.p, h, sizeof h, hipMemcpyHostToDevice));
    launch_selftest_sqrt(out.as<unsigned long long>(), nullptr);
    DVO_HIP(hipDeviceSynchronize());
    DVO_HIP(hipMemcpy(h, out.p, sizeof h, hipMemcpyDeviceToHost));
    if (inputs) *inputs = h[0];
    *mismatches = h[1];
    if (first_bad_bits) *first_bad_bits = (uint32_t)h[2];
    return DVO_OK;
}

int dvo_selftest_division(int dev, uint32_t b_first, uint32_t b_stride, uint32_t b_count, uint64_t* pairs, uint64_t* mismatches, uint64_t* first_bad_pair)
{
    if (!mismatches || b_count == 0 || b_count > (1u << 23)) return DVO_ERR_BAD_ARGUMENT;
    DVO_TRY(select_device(dev));
    DevBuf out;
    DVO_TRY(out.alloc(3 * sizeof(unsigned long long)));
    unsigned long long h[3] = {0ull, 0ull, ~0ull};
    DVO_HIP(hipMemcpy(out.p, h, sizeof h, hipMemcpyHostToDevice));
    for (uint32_t done = 0; done < b_count; done += 1u << 17) {   // 2^17 values of b (x 2^23 of a) per launch: each well under a second
        const uint32_t n = b_count - done < (1u << 17) ? b_count - done : (1u << 17);
        launch_selftest_division(b_first + done * b_stride, b_stride, n, out.as<unsigned long long>(), nullptr);
        DVO_HIP(hipDeviceSynchronize());
    }
    DVO_HIP(hipMemcpy(h, out.p, sizeof h, hipMemcpyDeviceToHost));
    if (pairs) *pairs = h[0];
    *mismatches = h[1];
    if (first_bad_pair) *first_bad_pair = h[2];
    return DVO_OK;
}

}  // extern "C"
