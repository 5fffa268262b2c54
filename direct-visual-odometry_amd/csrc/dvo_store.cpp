// dvo_store.cpp -- keyframe store / checkpoint and debug image export (row 4).
// The reference keeps every keyframe forever in FrameHistory (include/system/frame.hpp:146-188; reduceHistory is
// private and unused) and has no checkpointing.  Here the history can be dumped to / restored from one binary file and,
// optionally, bounded (dvo_vo_set_history_limit; 0 = unbounded = the reference's behaviour).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "dvo_engine.h"

using namespace dvo;

struct dvo_vo { VisualOdometry impl; };  // same layout as in dvo_capi.cpp

namespace {

struct FileHeader {
    char magic[8];
    int32_t version, width, height, levels, culls, n_keyframes, latest_id, reserved;
    float K[9];
};
const char kMagic[8] = {'D', 'V', 'O', 'K', 'F', '0', '1', 0};

bool put(FILE* f, const void* p, size_t n) { return fwrite(p, 1, n, f) == n; }
bool get(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

}  // namespace

extern "C" {

int dvo_vo_save(const dvo_vo* vo, const char* path)
{
    if (!vo || !path) return DVO_ERR_BAD_ARGUMENT;
    const VisualOdometry& v = vo->impl;
    DVO_TRY(select_device(v.device));
    DVO_HIP(hipStreamSynchronize(v.stream));
    FILE* f = fopen(path, "wb");
    if (!f) { set_error(std::string("cannot write ") + path); return DVO_ERR_BAD_ARGUMENT; }
    FileHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, kMagic, 8);
    h.version = 1; h.width = v.w; h.height = v.h; h.levels = v.geoM.levels; h.culls = v.geoM.culls;
    h.n_keyframes = (int)v.hist.size(); h.latest_id = v.latest_id;
    memcpy(h.K, v.K, sizeof h.K);
    bool ok = put(f, &h, sizeof h);
    std::vector<float> buf;
    for (const auto& kp : v.hist) {
        const Keyframe& k = *kp;
        const int32_t ids[2] = {k.id, k.ref_id};
        ok = ok && put(f, ids, sizeof ids) && put(f, k.xi, sizeof k.xi) && put(f, k.rel_xi, sizeof k.rel_xi);
        const Geometry& g = k.fs.g;
        for (int l = 0; l < g.levels && ok; l++) {
            buf.resize((size_t)g.w[l] * g.h[l]);
            DVO_HIP(hipMemcpy(buf.data(), k.fs.gray[l], buf.size() * 4, hipMemcpyDeviceToHost));
            ok = put(f, buf.data(), buf.size() * 4);
        }
        const int T = g.top();
        buf.resize((size_t)g.w[T] * g.h[T]);
        const float* maps[3] = {k.fs.depth[T], k.fs.sigma[T], k.age.as<float>()};
        for (int m = 0; m < 3 && ok; m++) {
            DVO_HIP(hipMemcpy(buf.data(), maps[m], buf.size() * 4, hipMemcpyDeviceToHost));
            ok = put(f, buf.data(), buf.size() * 4);
        }
    }
    fclose(f);
    if (!ok) { set_error("short write"); return DVO_ERR_BAD_ARGUMENT; }
    return DVO_OK;
}

int dvo_vo_load(dvo_vo* vo, const char* path)
{
    if (!vo || !path) return DVO_ERR_BAD_ARGUMENT;
    VisualOdometry& v = vo->impl;
    DVO_TRY(select_device(v.device));
    FILE* f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot read ") + path); return DVO_ERR_BAD_ARGUMENT; }
    FileHeader h;
    if (!get(f, &h, sizeof h) || memcmp(h.magic, kMagic, 8) != 0 || h.version != 1) {
        fclose(f);
        set_error("not a dvo keyframe store");
        return DVO_ERR_BAD_ARGUMENT;
    }
    if (h.width != v.w || h.height != v.h || h.levels != v.geoM.levels || h.culls != v.geoM.culls || memcmp(h.K, v.K, sizeof h.K) != 0) {
        fclose(f);
        set_error("keyframe store was written for a different camera / geometry");
        return DVO_ERR_BAD_ARGUMENT;
    }
    std::vector<std::unique_ptr<Keyframe>> hist;
    std::vector<float> buf;
    bool ok = true;
    for (int i = 0; i < h.n_keyframes && ok; i++) {
        auto k = std::make_unique<Keyframe>();
        int st = k->alloc(v.geoM, v.cfg, &v.kf_pool);
        if (st != DVO_OK) { fclose(f); return st; }
        int32_t ids[2];
        ok = get(f, ids, sizeof ids) && get(f, k->xi, sizeof k->xi) && get(f, k->rel_xi, sizeof k->rel_xi);
        k->id = ids[0]; k->ref_id = ids[1];
        const Geometry& g = v.geoM;
        for (int l = 0; l < g.levels && ok; l++) {
            buf.resize((size_t)g.w[l] * g.h[l]);
            ok = get(f, buf.data(), buf.size() * 4);
            if (ok) { const int hs = check_hip(hipMemcpy(k->fs.gray[l], buf.data(), buf.size() * 4, hipMemcpyHostToDevice), "hipMemcpy(keyframe gray)"); if (hs != DVO_OK) { fclose(f); return hs; } }
        }
        const int T = g.top();
        buf.resize((size_t)g.w[T] * g.h[T]);
        float* maps[3] = {k->fs.depth[T], k->fs.sigma[T], k->age.as<float>()};
        for (int m = 0; m < 3 && ok; m++) {
            ok = get(f, buf.data(), buf.size() * 4);
            if (ok) { const int hs = check_hip(hipMemcpy(maps[m], buf.data(), buf.size() * 4, hipMemcpyHostToDevice), "hipMemcpy(keyframe map)"); if (hs != DVO_OK) { fclose(f); return hs; } }
        }
        if (ok) {
            redecimate(k->fs, k->fs.depth[T], k->fs.sigma[T], v.stream);  // lower levels + iz / wgt, as Frame::updateDepthSigma
            hist.push_back(std::move(k));
        }
    }
    fclose(f);
    if (!ok) { set_error("truncated keyframe store"); return DVO_ERR_BAD_ARGUMENT; }
    DVO_HIP(hipStreamSynchronize(v.stream));
    v.hist = std::move(hist);
    v.hist_version++;   // (the device copies of the history's poses / gray pointers are rebuilt by the next map_update)
    v.latest_id = h.latest_id;
    return DVO_OK;
}

int dvo_vo_set_history_limit(dvo_vo* vo, int max_keyframes)
{
    if (!vo || max_keyframes < 0) return DVO_ERR_BAD_ARGUMENT;
    vo->impl.history_limit = max_keyframes;
    return DVO_OK;
}

// binary PPM (P6), rgb = [h][w][3] bytes
int dvo_ppm_write(const char* path, const uint8_t* rgb, int w, int h)
{
    if (!path || !rgb || w < 1 || h < 1) return DVO_ERR_BAD_ARGUMENT;
    FILE* f = fopen(path, "wb");
    if (!f) { set_error(std::string("cannot write ") + path); return DVO_ERR_BAD_ARGUMENT; }
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    const bool ok = fwrite(rgb, 1, (size_t)w * h * 3, f) == (size_t)w * h * 3;
    fclose(f);
    return ok ? DVO_OK : DVO_ERR_BAD_ARGUMENT;
}

}  // extern "C"
