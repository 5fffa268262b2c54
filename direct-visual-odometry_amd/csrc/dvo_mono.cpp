// dvo_mono.cpp -- the batched MONO pipeline (BASELINE config "tracking + inverse-depth filter"): System::VisualOdometry::odometrize
// (include/system/system.hpp:44-74) and Map::Mapper (src/map/mapper.cpp:16-144) for n_seq sequences per call.  One fixed launch
// sequence per frame, no host round trip: every sequence's keyframe decision (Mapper::needNewFrame) is a flag in device memory that
// the mapping kernels test per sequence.  Reference citations: file:line under the reference tree.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <utility>

#include "dvo_engine.h"

namespace dvo {

MonoBatch::~MonoBatch()
{
    if (cstream) { (void)hipStreamSynchronize(cstream); (void)hipStreamDestroy(cstream); }
    for (auto& st : stage) {
        if (st.copied) (void)hipEventDestroy(st.copied);
        if (st.consumed) (void)hipEventDestroy(st.consumed);
    }
    for (auto& m : map_ev)
        for (hipEvent_t e : m.e) (void)hipEventDestroy(e);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
}

int MonoBatch::collect_map_profile()
{
    DVO_HIP(hipStreamSynchronize(stream));
    for (size_t i = 0; i < map_ev_used; i++) {
        float p = 0, u = 0, r = 0;
        DVO_HIP(hipEventElapsedTime(&p, map_ev[i].e[0], map_ev[i].e[1]));
        DVO_HIP(hipEventElapsedTime(&u, map_ev[i].e[2], map_ev[i].e[3]));
        DVO_HIP(hipEventElapsedTime(&r, map_ev[i].e[4], map_ev[i].e[5]));
        prof_propagate_ms += p; prof_update_ms += u; prof_regularize_ms += r;
        prof_frames++;
    }
    map_ev_used = 0;
    return DVO_OK;
}

int MonoBatch::init(int n, const float K9[9], int w, int h, int ring, const dvo_config* c)
{
    if (n < 1 || !K9 || w < 64 || h < 64 || ring < 1 || ring > 64) { set_error("dvo_batch_create_mono: bad arguments"); return DVO_ERR_BAD_ARGUMENT; }
    if (c) cfg = *c; else dvo_config_default(&cfg);
    n_seq = n; R = ring; device = cfg.device;
    DVO_TRY(select_device(device));
    if (cfg.stream) stream = (hipStream_t)cfg.stream;
    else { DVO_HIP(hipStreamCreate(&stream)); own_stream = true; }
    DVO_TRY(make_geometry(K9, w, h, 3, 2, g));  // Frame(gray, K, 3, 2), system.hpp:47
    DVO_TRY(ref.alloc(g, n, cfg));
    DVO_TRY(frm.alloc(g, n, cfg));
    DVO_TRY(trk.init(g, n, cfg));
    const size_t np = (size_t)top_pixels(), all = np * (size_t)n * sizeof(float);
    DVO_TRY(ref_age.alloc(all)); DVO_TRY(frm_age.alloc(all)); DVO_TRY(owner.alloc(all)); DVO_TRY(tmp.alloc(all));
    DVO_TRY(ring_gray.alloc(all * (size_t)R));
    DVO_TRY(hist_xi.alloc(sizeof(float) * 6 * (size_t)R * n));
    DVO_TRY(ages.alloc(sizeof(AgeEntry) * (size_t)R * n));
    DVO_TRY(meta.alloc(sizeof(MonoSeq) * (size_t)n));
    DVO_TRY(init_depth.alloc(np * sizeof(float))); DVO_TRY(init_sigma.alloc(np * sizeof(float)));
    DVO_TRY(xi_world.alloc(sizeof(float) * 6 * (size_t)n));
    DVO_TRY(T_world.alloc(sizeof(float) * 16 * (size_t)n));
    DVO_TRY(is_key.alloc(sizeof(int) * (size_t)n));
    DVO_TRY(need_list.alloc(sizeof(int) * ((size_t)n + 4)));
    DVO_HIP(hipMemset(meta.p, 0, meta.bytes));
    DVO_HIP(hipMemset(hist_xi.p, 0, hist_xi.bytes));
    return DVO_OK;
}

int MonoBatch::set_initial_depth(const float* depth_host, const float* sigma_host)
{  // replaces cv::randn(depth, 1.5, 0.5), max(depth, 0.5), sigma = 0.5 of the first mono keyframe (frame.hpp:17-21, D6)
    if (!depth_host || !sigma_host) { set_error("null map"); return DVO_ERR_BAD_ARGUMENT; }
    DVO_TRY(select_device(device));
    const size_t np = (size_t)top_pixels();
    DVO_HIP(hipMemcpy(init_depth.p, depth_host, np * sizeof(float), hipMemcpyHostToDevice));
    DVO_HIP(hipMemcpy(init_sigma.p, sigma_host, np * sizeof(float), hipMemcpyHostToDevice));
    const int T = g.top();
    launch_broadcast(init_depth.as<float>(), ref.depth[T], (int)np, n_seq, stream);
    launch_broadcast(init_sigma.as<float>(), ref.sigma[T], (int)np, n_seq, stream);
    have_init = true;
    DVO_HIP(hipGetLastError());
    return DVO_OK;
}

int MonoBatch::set_initial_depth_device(const float* depth_dev, const float* sigma_dev)
{
    if (!depth_dev || !sigma_dev) { set_error("null map"); return DVO_ERR_BAD_ARGUMENT; }
    DVO_TRY(select_device(device));
    const size_t all = (size_t)top_pixels() * n_seq * sizeof(float);
    const int T = g.top();
    DVO_HIP(hipMemcpyAsync(ref.depth[T], depth_dev, all, hipMemcpyDeviceToDevice, stream));
    DVO_HIP(hipMemcpyAsync(ref.sigma[T], sigma_dev, all, hipMemcpyDeviceToDevice, stream));
    have_init = true;
    return DVO_OK;
}

int MonoBatch::odometrize_host(const void* frames, size_t bytes, FrameInput in)
{  // frames of every sequence from host memory: H2D on a copy stream into one of two staging slots (as Batch::push_host_frame)
    if (!frames) { set_error("null host pointer"); return DVO_ERR_BAD_ARGUMENT; }
    DVO_TRY(select_device(device));
    if (!cstream) {
        DVO_HIP(hipStreamCreateWithFlags(&cstream, hipStreamNonBlocking));
        for (auto& st : stage) {
            DVO_HIP(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming));
            DVO_HIP(hipEventCreateWithFlags(&st.consumed, hipEventDisableTiming));
        }
    }
    trk.adaptive = false;   // (the host must not be held inside track(): the next frame's transfer is queued meanwhile)
    Stage& st = stage[n_host & 1];
    if (st.buf.bytes < bytes) DVO_TRY(st.buf.alloc(bytes));
    if (st.used) DVO_HIP(hipStreamWaitEvent(cstream, st.consumed, 0));
    if (in.raw()) {  // only the rows the pyramid keeps cross PCIe (Batch::push_host_frame)
        in.rows_decimated = decimate_host_rows && can_decimate_rows(g);
        DVO_TRY(upload_rows(st.buf.p, frames, (size_t)g.src_w * in.channels, g.src_h, (size_t)n_seq, g.culls, in.rows_decimated, cstream, nullptr));
    } else {
        in.rows_decimated = decimate_host_rows && can_decimate_rows(g);
        DVO_TRY(upload_rows(st.buf.p, frames, (size_t)g.src_w * sizeof(float), g.src_h, (size_t)n_seq, g.culls, in.rows_decimated, cstream, nullptr));
    }
    DVO_HIP(hipEventRecord(st.copied, cstream));
    if (!host_buffer_is_pinned(frames)) DVO_HIP(hipStreamSynchronize(cstream));   // pageable source: see Batch::push_host_frame
    DVO_HIP(hipStreamWaitEvent(stream, st.copied, 0));
    if (in.raw()) in.rgb = st.buf.as<uint8_t>(); else in.gray = st.buf.as<float>();
    const int rc = odometrize(in);
    if (rc != DVO_OK) {   // the frame was not consumed: same staging slot next time, once whatever was queued has drained
        (void)hipStreamSynchronize(stream);
        return rc;
    }
    DVO_HIP(hipEventRecord(st.consumed, stream));
    st.used = true;
    n_host++;
    return DVO_OK;
}

int MonoBatch::odometrize(const FrameInput& in)
{  // system.hpp:44-74 for every sequence
    if (!in.key0() || (in.raw() && in.channels != 1 && in.channels != 3 && in.channels != 4)) { set_error("null device pointer / bad channel count"); return DVO_ERR_BAD_ARGUMENT; }
    FrameInput gin = in;          // mono: gray only
    gin.depth = nullptr; gin.sigma = nullptr; gin.depth16 = nullptr;
    DVO_TRY(select_device(device));
    const int T = g.top(), tw = g.w[T], th = g.h[T], np = tw * th;
    // Frame::latest_id (frame.cpp:5) advances only once the frame's launches were queued: a call that fails leaves the batch where it
    // was (a failed first frame leaves it not started), so frame ids -- and with them keyframe_max_frames -- never shift
    const int frame_id = latest_id + 1;
    MonoSeq* m = meta.as<MonoSeq>();
    if (frame_id == 0) {  // system.hpp:49-54: the first frame is the first keyframe of every sequence
        if (!have_init) {
            std::vector<float> d, s;
            default_initial_depth(np, cfg.rng_seed, d, s);
            DVO_TRY(set_initial_depth(d.data(), s.data()));
        }
        build_pyramid(ref, gin, stream);
        DVO_HIP(hipMemsetAsync(ref_age.p, 0, ref_age.bytes, stream));
        redecimate(ref, ref.depth[T], ref.sigma[T], stream);
        PromoteArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.n_seg = 0; pa.n_seq = n_seq; pa.gray_top = ref.gray[T]; pa.ring_gray = ring_gray.as<float>(); pa.npix = np; pa.R = R;
        pa.meta = m; pa.all = 1;
        launch_promote(pa, stream);
        launch_mono_commit(m, hist_xi.as<float>(), n_seq, R, 1, frame_id, xi_world.as<float>(), T_world.as<float>(), is_key.as<int>(), stream);
        DVO_HIP(hipGetLastError());
        latest_id = frame_id;
        return DVO_OK;
    }
    MapEv* pe = nullptr;
    if (cfg.profile) {
        if (map_ev_used == map_ev.size()) {
            MapEv m6;
            for (hipEvent_t& e : m6.e) DVO_HIP(hipEventCreate(&e));
            map_ev.push_back(m6);
        }
        pe = &map_ev[map_ev_used];
    }
    { TraceRange tr("mono pyramid"); build_pyramid(frm, gin, stream); }           // Frame(gray, K, 3, 2)
    { TraceRange tr("mono track"); DVO_TRY(trk.track(frm, ref, stream)); }           // system.hpp:57
    TraceRange tr_map("mono map (decide, propagate | update, promote, regularize)");
    DVO_HIP(hipMemsetAsync(need_list.p, 0, 4 * sizeof(int), stream));
    launch_mono_decide(m, trk.state.as<SeqState>(), n_seq, frame_id, cfg.keyframe_min_translation, cfg.keyframe_max_frames,
                       xi_world.as<float>(), T_world.as<float>(), is_key.as<int>(), nullptr, stream, need_list.as<int>());
    // ---- Mapper::estimate (mapper.cpp:16-33), both branches launched, each sequence takes its own ----
    {   // need: propagate the reference maps into the frame (mapper.cpp:62-74) ...
        PropArgs a;
        a.ref_depth = ref.depth[T]; a.ref_sigma = ref.sigma[T]; a.ref_age = ref_age.as<float>();
        a.depth = frm.depth[T]; a.sigma = frm.sigma[T]; a.age = frm_age.as<float>();
        a.owner = owner.as<int>();
        a.w = tw; a.h = th; a.n_seq = n_seq; a.k = g.k[T]; a.meta = m;
        memset(&a.pose, 0, sizeof a.pose); a.tz = 0.0f;
        a.need_list = need_list.as<int>();
        if (pe) DVO_HIP(hipEventRecord(pe->e[0], stream));
        launch_propagate_batch(a, stream);
        if (pe) DVO_HIP(hipEventRecord(pe->e[1], stream));
    }
    {   // !need: stereo update of the reference maps against the keyframe each pixel was born in (mapper.cpp:76-137)
        AgeTableArgs ta;
        ta.meta = m; ta.hist_xi = hist_xi.as<float>(); ta.ages = ages.as<AgeEntry>(); ta.n_seq = n_seq; ta.R = R; ta.n_hist = -1;
        if (pe) DVO_HIP(hipEventRecord(pe->e[2], stream));
        launch_age_table(ta, stream);
        UpdateArgs a;
        memset(&a, 0, sizeof a);
        a.ref_depth = ref.depth[T]; a.ref_sigma = ref.sigma[T]; a.ref_age = ref_age.as<float>();
        a.obj_gray = frm.gray[T];
        a.ages = ages.as<AgeEntry>();
        a.ring_gray = ring_gray.as<float>(); a.gray_table = nullptr; a.meta = m;
        a.n_seq = n_seq; a.R = R; a.n_hist = 0; a.w = tw; a.h = th; a.crop = cfg.crop_enable; a.obj_id = frame_id;
        a.clamp_age = 1;
        a.seed = cfg.rng_seed;
        a.k = g.k[T];
        memcpy(a.K9, g.K9[T], sizeof a.K9);
        launch_depth_update(a, stream);
        if (pe) DVO_HIP(hipEventRecord(pe->e[3], stream));
    }
    {   // ... need: the frame becomes the newest keyframe (FrameHistory::push, frame.hpp:151-157)
        PromoteArgs pa;
        memset(&pa, 0, sizeof pa);
        int sgi = 0;
        for (int l = 0; l < g.levels; l++) { pa.src[sgi] = frm.gray[l]; pa.dst[sgi] = ref.gray[l]; pa.count[sgi] = g.w[l] * g.h[l]; sgi++; }
        pa.src[sgi] = frm.depth[T]; pa.dst[sgi] = ref.depth[T]; pa.count[sgi] = np; sgi++;
        pa.src[sgi] = frm.sigma[T]; pa.dst[sgi] = ref.sigma[T]; pa.count[sgi] = np; sgi++;
        pa.src[sgi] = frm_age.as<float>(); pa.dst[sgi] = ref_age.as<float>(); pa.count[sgi] = np; sgi++;
        pa.n_seg = sgi; pa.n_seq = n_seq; pa.gray_top = frm.gray[T]; pa.ring_gray = ring_gray.as<float>(); pa.npix = np; pa.R = R;
        pa.meta = m; pa.all = 0;
        pa.need_list = need_list.as<int>();
        launch_promote(pa, stream);
        launch_mono_commit(m, hist_xi.as<float>(), n_seq, R, 0, frame_id, nullptr, nullptr, nullptr, stream);
    }
    // Mapper::regularize (mapper.cpp:139-144) of the newest keyframe, then Frame::updateDepthSigma / updateDepth (frame.cpp:39-61):
    // every level of depth and sigma is a decimation of the top maps, so ONE pass re-derives both pyramids (and the
    // weight) from the regularized depth and the current sigma -- the same values the reference's two re-decimations leave.
    {
        RegDecArgs ra;
        memset(&ra, 0, sizeof ra);
        ra.depth = ref.depth[T]; ra.sigma = ref.sigma[T];
        if (!depth_alt) depth_alt = tmp.as<float>();
        ra.depth_top_out = depth_alt;
        for (int l = 0; l < g.levels; l++) {
            ra.w[l] = g.w[l]; ra.h[l] = g.h[l];
            ra.depth_lv[l] = ref.depth[l]; ra.sigma_lv[l] = ref.sigma[l]; ra.wgt[l] = ref.wgt[l];
            ra.step[l] = ref.step[l];
        }
        ra.levels = g.levels; ra.n_seq = n_seq; ra.sigma_min = ref.sigma_min; ra.sigma_max = ref.sigma_max;
        if (pe) DVO_HIP(hipEventRecord(pe->e[4], stream));
        launch_regularize_redecimate(ra, stream);
        if (pe) { DVO_HIP(hipEventRecord(pe->e[5], stream)); map_ev_used++; }
        std::swap(ref.depth[T], depth_alt);   // the top-level depth map alternates between the arena block and `tmp`
    }
    DVO_HIP(hipGetLastError());
    latest_id = frame_id;
    return DVO_OK;
}

}  // namespace dvo

using namespace dvo;

extern "C" {

int dvo_batch_create_mono(int n_seq, const float K[9], int width, int height, int ring_keyframes, const dvo_config* cfg, dvo_batch** out)
{
    if (!out) return DVO_ERR_BAD_ARGUMENT;
    *out = nullptr;
    dvo_batch* b = new (std::nothrow) dvo_batch();
    if (!b) return DVO_ERR_OUT_OF_MEMORY;
    b->mono.reset(new (std::nothrow) MonoBatch());
    if (!b->mono) { delete b; return DVO_ERR_OUT_OF_MEMORY; }
    const int st = b->mono->init(n_seq, K, width, height, ring_keyframes > 0 ? ring_keyframes : 8, cfg);
    if (st != DVO_OK) { delete b; return st; }
    *out = b;
    return DVO_OK;
}

#define DVO_NEED_MONO(b)                                                                                      \
    do {                                                                                                      \
        if (!(b) || !(b)->mono) { set_error("this entry point needs a mono batch (dvo_batch_create_mono)"); return DVO_ERR_BAD_ARGUMENT; } \
    } while (0)

int dvo_batch_set_initial_depth(dvo_batch* b, const float* depth, const float* sigma)
{
    DVO_NEED_MONO(b);
    return b->mono->set_initial_depth(depth, sigma);
}

int dvo_batch_set_initial_depth_device(dvo_batch* b, const float* depth_dev, const float* sigma_dev)
{
    DVO_NEED_MONO(b);
    return b->mono->set_initial_depth_device(depth_dev, sigma_dev);
}

int dvo_batch_odometrize_device(dvo_batch* b, const float* gray_dev)
{
    DVO_NEED_MONO(b);
    FrameInput in;
    in.gray = gray_dev;
    return b->mono->odometrize(in);
}

int dvo_batch_odometrize_raw_device(dvo_batch* b, const uint8_t* rgb_dev, int channels)
{
    DVO_NEED_MONO(b);
    FrameInput in;
    in.rgb = rgb_dev; in.channels = channels;
    return b->mono->odometrize(in);
}

int dvo_batch_odometrize_host(dvo_batch* b, const float* gray)
{
    DVO_NEED_MONO(b);
    FrameInput in;
    in.gray = gray;
    return b->mono->odometrize_host(gray, sizeof(float) * (size_t)b->mono->n_seq * b->mono->g.src_w * b->mono->g.src_h, in);
}

int dvo_batch_odometrize_raw_host(dvo_batch* b, const uint8_t* rgb, int channels)
{
    DVO_NEED_MONO(b);
    if (channels != 1 && channels != 3 && channels != 4) { set_error("bad channel count"); return DVO_ERR_BAD_ARGUMENT; }
    FrameInput in;
    in.rgb = rgb; in.channels = channels;
    return b->mono->odometrize_host(rgb, (size_t)channels * b->mono->n_seq * b->mono->g.src_w * b->mono->g.src_h, in);
}

int dvo_batch_world_poses(dvo_batch* b, float* xi_world, float* T_world, int* is_keyframe)
{
    DVO_NEED_MONO(b);
    MonoBatch& M = *b->mono;
    if (M.latest_id < 0) return DVO_ERR_NOT_READY;
    DVO_TRY(select_device(M.device));
    if (xi_world) DVO_HIP(hipMemcpyAsync(xi_world, M.xi_world.p, sizeof(float) * 6 * (size_t)M.n_seq, hipMemcpyDeviceToHost, M.stream));
    if (T_world) DVO_HIP(hipMemcpyAsync(T_world, M.T_world.p, sizeof(float) * 16 * (size_t)M.n_seq, hipMemcpyDeviceToHost, M.stream));
    if (is_keyframe) DVO_HIP(hipMemcpyAsync(is_keyframe, M.is_key.p, sizeof(int) * (size_t)M.n_seq, hipMemcpyDeviceToHost, M.stream));
    DVO_HIP(hipStreamSynchronize(M.stream));
    return DVO_OK;
}

int dvo_batch_copy_world_poses_device(dvo_batch* b, float* xi_dst_dev, float* T_dst_dev, int* key_dst_dev)
{
    DVO_NEED_MONO(b);
    MonoBatch& M = *b->mono;
    if (M.latest_id < 0) return DVO_ERR_NOT_READY;
    DVO_TRY(select_device(M.device));
    if (xi_dst_dev) DVO_HIP(hipMemcpyAsync(xi_dst_dev, M.xi_world.p, sizeof(float) * 6 * (size_t)M.n_seq, hipMemcpyDeviceToDevice, M.stream));
    if (T_dst_dev) DVO_HIP(hipMemcpyAsync(T_dst_dev, M.T_world.p, sizeof(float) * 16 * (size_t)M.n_seq, hipMemcpyDeviceToDevice, M.stream));
    if (key_dst_dev) DVO_HIP(hipMemcpyAsync(key_dst_dev, M.is_key.p, sizeof(int) * (size_t)M.n_seq, hipMemcpyDeviceToDevice, M.stream));
    return DVO_OK;
}

int dvo_batch_keyframe_get(dvo_batch* b, int seq, int level, float* gray, float* depth, float* sigma, float* age, float xi[6], int* id,
                           int* n_keyframes, int* valid_updates)
{
    DVO_NEED_MONO(b);
    MonoBatch& M = *b->mono;
    if (seq < 0 || seq >= M.n_seq || level < 0 || level >= M.g.levels) return DVO_ERR_BAD_ARGUMENT;
    if (M.latest_id < 0) return DVO_ERR_NOT_READY;
    DVO_TRY(select_device(M.device));
    hipStream_t s = M.stream;
    const size_t n = (size_t)M.g.w[level] * M.g.h[level], off = n * (size_t)seq;
    if (gray) DVO_HIP(hipMemcpyAsync(gray, M.ref.gray[level] + off, n * 4, hipMemcpyDeviceToHost, s));
    if (depth) DVO_HIP(hipMemcpyAsync(depth, M.ref.depth[level] + off, n * 4, hipMemcpyDeviceToHost, s));
    if (sigma) DVO_HIP(hipMemcpyAsync(sigma, M.ref.sigma[level] + off, n * 4, hipMemcpyDeviceToHost, s));
    if (age) {
        if (level != M.g.top()) { set_error("age is stored for the top level only"); return DVO_ERR_BAD_ARGUMENT; }
        DVO_HIP(hipMemcpyAsync(age, M.ref_age.as<float>() + off, n * 4, hipMemcpyDeviceToHost, s));
    }
    MonoSeq m;
    DVO_HIP(hipMemcpyAsync(&m, M.meta.as<MonoSeq>() + seq, sizeof m, hipMemcpyDeviceToHost, s));
    DVO_HIP(hipStreamSynchronize(s));
    if (xi) memcpy(xi, m.ref_xi, 6 * sizeof(float));
    if (id) *id = m.ref_id;
    if (n_keyframes) *n_keyframes = m.n_total;
    if (valid_updates) *valid_updates = m.valid_updates;
    return DVO_OK;
}

int dvo_batch_mono_stats(dvo_batch* b, int seq, dvo_mono_stats* out)
{
    DVO_NEED_MONO(b);
    MonoBatch& M = *b->mono;
    if (!out || seq < 0 || seq >= M.n_seq) return DVO_ERR_BAD_ARGUMENT;
    if (M.latest_id < 0) return DVO_ERR_NOT_READY;
    DVO_TRY(select_device(M.device));
    MonoSeq m;
    DVO_HIP(hipMemcpyAsync(&m, M.meta.as<MonoSeq>() + seq, sizeof m, hipMemcpyDeviceToHost, M.stream));
    DVO_HIP(hipStreamSynchronize(M.stream));
    out->frames = M.latest_id + 1;
    out->keyframes_created = m.n_total;
    out->ring_keyframes = M.R;
    out->valid_updates_last_frame = m.valid_updates;
    out->clamped_pixels = m.clamped;
    return DVO_OK;
}

int dvo_batch_profile_mapping(dvo_batch* b, dvo_map_profile* out, int reset)
{
    DVO_NEED_MONO(b);
    MonoBatch& M = *b->mono;
    if (!out) return DVO_ERR_BAD_ARGUMENT;
    DVO_TRY(select_device(M.device));
    DVO_TRY(M.collect_map_profile());
    out->frames = M.prof_frames;
    out->depth_update_ms = M.prof_update_ms;
    out->regularize_ms = M.prof_regularize_ms;
    out->propagate_ms = M.prof_propagate_ms;
    const int T = M.g.top(), w = M.g.w[T], h = M.g.h[T];
    // the window of mapper.cpp:90 (cols 16..144, rows 12..108 at 160x120) when crop_enable, else the whole map
    const int cx = M.cfg.crop_enable ? ((w - 1 < 144 ? w - 1 : 144) - 16 + 1) : w, cy = M.cfg.crop_enable ? ((h - 1 < 108 ? h - 1 : 108) - 12 + 1) : h;
    const int wx = cx > 0 ? cx : 0, wy = cy > 0 ? cy : 0;   // (as launch_depth_update)
    out->update_window_pixels = (uint64_t)wx * wy * (uint64_t)M.n_seq;
    out->map_pixels = (uint64_t)w * h * (uint64_t)M.n_seq;
    if (reset) { M.prof_frames = 0; M.prof_update_ms = M.prof_regularize_ms = M.prof_propagate_ms = 0; }
    return DVO_OK;
}

}  // extern "C"
