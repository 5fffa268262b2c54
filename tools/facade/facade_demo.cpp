// facade_demo.cpp -- a C++ consumer of include/dvo.hpp, shaped like the reference's main programs after INTEGRATION.md's adaptor:
// `System::VisualOdometry vo(K); for (frame) T = vo.odometrize(gray)` (main.cpp:33,49) and
// `T = vo.odometrizeUsingDepth(gray, depth, sigma)` (test/sequence.cpp:10-23).  No OpenCV: frames come from a raw float32 file
// ([n][3][h][w]: gray, depth, sigma per frame) and the 4x4 poses go to a raw float32 file, so that tests/test_facade_cpp.py can
// compare them bit for bit with the ctypes path.  Built by `make -C direct-visual-odometry_amd facade` (g++ + libdvo.so).
//   facade_demo <frames.f32> <n> <w> <h> <fx> <fy> <cx> <cy> <mono|depth> <poses_out.f32> [rng_seed]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dvo.hpp"

int main(int argc, char** argv)
{
    if (argc < 11) { std::fprintf(stderr, "usage: %s frames.f32 n w h fx fy cx cy mono|depth poses.f32 [seed]\n", argv[0]); return 2; }
    const int n = std::atoi(argv[2]), w = std::atoi(argv[3]), h = std::atoi(argv[4]);
    const dvo::Mat3 K = {(float)std::atof(argv[5]), 0, (float)std::atof(argv[7]), 0, (float)std::atof(argv[6]), (float)std::atof(argv[8]), 0, 0, 1};
    const bool mono = std::strcmp(argv[9], "mono") == 0;
    const size_t px = (size_t)w * h;
    std::vector<float> frames((size_t)n * 3 * px);
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(frames.data(), sizeof(float), frames.size(), f) != frames.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::fclose(f);
    std::vector<float> poses;
    try {
        dvo_config cfg = dvo::default_config();            // the reference's constants
        if (argc > 11) cfg.rng_seed = (uint32_t)std::atoi(argv[11]);
        dvo::VisualOdometry vo(K, w, h, &cfg);             // System::VisualOdometry vo(K), main.cpp:33
        if (mono) {                                        // explicit initial depth instead of cv::randn (frame.hpp:17-21): frame 0's, culled by 4
            std::vector<float> d((px / 16)), s(px / 16, 0.5f);
            for (int y = 0; y < h / 4; y++)
                for (int x = 0; x < w / 4; x++) d[(size_t)y * (w / 4) + x] = frames[px + (size_t)(4 * y) * w + 4 * x];
            vo.setInitialDepth(d.data(), s.data());
        }
        for (int i = 0; i < n; i++) {
            const float* g = frames.data() + (size_t)i * 3 * px;
            bool key = false;
            const dvo::Mat4 T = mono ? vo.odometrize(g, &key)                               // main.cpp:49
                                     : vo.odometrizeUsingDepth(g, g + px, g + 2 * px);      // test/sequence.cpp:20
            poses.insert(poses.end(), T.begin(), T.end());
            const dvo_track_log lg = vo.lastTrackLog();
            std::printf("frame %d: t = (%+.5f %+.5f %+.5f)%s  iterations/level", i, T[3], T[7], T[11], key ? "  [keyframe]" : "");
            for (int l = 0; l < lg.levels; l++) std::printf(" %d", lg.n_iter[l]);
            std::printf("\n");
        }
        if (mono) {
            const dvo::Keyframe kf = vo.keyframe(vo.keyframeCount() - 1);                  // FrameHistory::operator[], frame.hpp:176
            std::printf("keyframes: %d, newest id %d, top level %dx%d\n", vo.keyframeCount(), kf.id, kf.width, kf.height);
        }
    } catch (const dvo::Error& e) {
        std::fprintf(stderr, "dvo error %d: %s\n", e.status, e.what());
        return 1;
    }
    f = std::fopen(argv[10], "wb");
    if (!f || std::fwrite(poses.data(), sizeof(float), poses.size(), f) != poses.size()) return 2;
    std::fclose(f);
    return 0;
}
