// single_stream_bench.cpp -- the reference's own loop (test/sequence.cpp:10-23: `T = vo.odometrizeUsingDepth(gray, depth, sigma)` per
// frame, one sequence, synchronous) timed from C++ through include/dvo.hpp, with no Python in the measurement: float maps from
// pageable and from pinned host memory, and raw u8 + u16 sensor frames.  Frames come from a raw float32 file ([n][3][h][w]: gray,
// depth, sigma) written by the caller (tools/bench_single_cpp.py).
// Then the mono loop (main.cpp:49).
//   single_stream_bench <frames.f32> <n> <w> <h> <fx> <fy> <cx> <cy> <frames_to_time>
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dvo.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    if (argc < 10) { std::fprintf(stderr, "usage: %s frames.f32 n w h fx fy cx cy frames_to_time\n", argv[0]); return 2; }
    const int n = std::atoi(argv[2]), w = std::atoi(argv[3]), h = std::atoi(argv[4]), N = std::atoi(argv[9]);
    const dvo::Mat3 K = {(float)std::atof(argv[5]), 0, (float)std::atof(argv[7]), 0, (float)std::atof(argv[6]), (float)std::atof(argv[8]), 0, 0, 1};
    const size_t px = (size_t)w * h;
    std::vector<float> frames((size_t)n * 3 * px);
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(frames.data(), sizeof(float), frames.size(), f) != frames.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::fclose(f);
    auto idx = [&](int k) { const int period = 2 * (n - 1), r = k % period; return r < n ? r : period - r; };   // ping-pong: neighbours only
    try {
        for (int mode = std::getenv("DVO_SKIP_DEPTH") ? 3 : 0; mode < 3; mode++) {
            // mode 0: float maps in pageable memory (what cv::Mat1f buffers are); 1: the same in pinned memory; 2: raw u8 gray + u16 depth
            float* pin = nullptr;
            std::vector<uint8_t> g8; std::vector<uint16_t> d16;
            if (mode == 1) {
                if (hipHostMalloc((void**)&pin, frames.size() * sizeof(float), hipHostMallocDefault) != hipSuccess) return 3;
                std::memcpy(pin, frames.data(), frames.size() * sizeof(float));
            }
            if (mode == 2) {
                g8.resize((size_t)n * px); d16.resize((size_t)n * px);
                for (int i = 0; i < n; i++)
                    for (size_t p = 0; p < px; p++) {
                        g8[i * px + p] = (uint8_t)std::lrintf(std::fmin(std::fmax(frames[(size_t)i * 3 * px + p] * 255.0f, 0.0f), 255.0f));
                        d16[i * px + p] = (uint16_t)std::lrintf(std::fmin(std::fmax(frames[(size_t)i * 3 * px + px + p] * 5000.0f, 0.0f), 65535.0f));
                    }
            }
            const float* base = mode == 1 ? pin : frames.data();
            long its = 0;
            double dt = 0;
            for (int pass = 0; pass < 2; pass++) {   // pass 0 is timed; pass 1 repeats it (same frames, same results) and reads the iteration counts
                dvo_config cfg = dvo::default_config();
                if (std::getenv("DVO_PPT")) cfg.gn_pixels_per_thread = std::atoi(std::getenv("DVO_PPT"));          // A/B knobs: tile size ...
                if (std::getenv("DVO_SINGLE_LAUNCH")) cfg.track_single_launch = std::atoi(std::getenv("DVO_SINGLE_LAUNCH"));   // ... and schedule
                dvo::VisualOdometry vo(K, w, h, &cfg);
                double t0 = 0;
                for (int k = 0; k < N + 3; k++) {
                    if (k == 3) t0 = now();
                    const int i = idx(k);
                    dvo::Mat4 T;
                    if (mode == 2) dvo::check(dvo_vo_odometrize_depth_raw(vo.handle(), g8.data() + i * px, 1, d16.data() + i * px, 0.0f, T.data()));
                    else { const float* g = base + (size_t)i * 3 * px; T = vo.odometrizeUsingDepth(g, g + px, g + 2 * px); }
                    if (pass == 1 && k >= 3) {
                        const dvo_track_log lg = vo.lastTrackLog();
                        for (int l = 0; l < lg.levels; l++) its += lg.n_iter[l];
                    }
                }
                if (pass == 0) dt = now() - t0;
            }
            std::printf("%-52s %8.1f frames/s  %7.1f us/frame  (%.1f GN iterations per frame)\n",
                        mode == 0 ? "odometrizeUsingDepth, float maps, pageable host" : mode == 1 ? "odometrizeUsingDepth, float maps, pinned host"
                                                                                                   : "odometrizeUsingDepthRaw, u8 gray + u16 depth",
                        N / dt, dt / N * 1e6, (double)its / N);
            if (pin) (void)hipHostFree(pin);
        }
        // mono: the reference's main loop (main.cpp:49: `T = vo.odometrize(gray)`), tracking + Mapper::estimate + regularize per frame;
        // initial map = the first frame's depth decimated by 4 with sigma 0.5 (a map the stereo updates succeed on), float gray from
        // pageable host memory, then raw u8 gray
        if (!std::getenv("DVO_SKIP_MONO")) {
            const int tw = w / 4, th = h / 4;
            std::vector<float> d0((size_t)tw * th), s0((size_t)tw * th, 0.5f);
            for (int y = 0; y < th; y++)
                for (int x = 0; x < tw; x++) d0[(size_t)y * tw + x] = frames[px + (size_t)(4 * y) * w + 4 * x];
            std::vector<uint8_t> g8((size_t)n * px);
            for (int i = 0; i < n; i++)
                for (size_t p = 0; p < px; p++) g8[i * px + p] = (uint8_t)std::lrintf(std::fmin(std::fmax(frames[(size_t)i * 3 * px + p] * 255.0f, 0.0f), 255.0f));
            for (int mode = 0; mode < 2; mode++) {
                dvo_config cfg = dvo::default_config();
                cfg.rng_seed = 1;
                dvo::VisualOdometry vo(K, w, h, &cfg);
                vo.setInitialDepth(d0.data(), s0.data());
                double t0 = 0;
                int keys = 0;
                long its = 0;
                for (int k = 0; k < N + 3; k++) {
                    if (k == 3) t0 = now();
                    const int i = idx(k);
                    bool key = false;
                    if (mode == 0) (void)vo.odometrize(frames.data() + (size_t)i * 3 * px, &key);
                    else (void)vo.odometrizeRaw(g8.data() + i * px, 1, &key);
                    if (k >= 3) keys += key ? 1 : 0;
                }
                const double dt = now() - t0;
                {   // (outside the timed loop) iterations of the last frame
                    const dvo_track_log lg = vo.lastTrackLog();
                    for (int l = 0; l < lg.levels; l++) its += lg.n_iter[l];
                }
                std::printf("%-52s %8.1f frames/s  %7.1f us/frame  (%d keyframes in %d frames; last frame: %ld GN iterations)\n",
                            mode == 0 ? "odometrize (mono: track + map), float gray, pageable" : "odometrizeRaw (mono: track + map), u8 gray", N / dt,
                            dt / N * 1e6, keys, N, its);
            }
        }
    } catch (const dvo::Error& e) {
        std::fprintf(stderr, "dvo error %d: %s\n", e.status, e.what());
        return 1;
    }
    return 0;
}
