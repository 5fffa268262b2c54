// Host sanitizer driver for libdvo's host-only translation units: the PNG / dataset front-end (dvo_io.cpp)
// and the trajectory evaluation (dvo_eval.cpp) under -fsanitize=address,undefined, built with g++ (make -C
// direct-visual-odometry_amd asan).  Valid PNGs of every supported layout, crafted headers, a byte-mutation fuzz, TUM / list
// directories with ragged lines, ATE / RPE / pose inverse / TUM export.  argv[1] = scratch directory.
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dvo.h"

namespace dvo {
static std::string g_err;
void set_error(const std::string& s) { g_err = s; }  // (the library's lives in dvo_capi.cpp, which needs the HIP runtime)
}  // namespace dvo

static void put32(std::vector<unsigned char>& v, uint32_t x)
{
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
static void chunk(std::vector<unsigned char>& png, const char* type, const std::vector<unsigned char>& data)
{
    put32(png, (uint32_t)data.size());
    const size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    put32(png, (uint32_t)crc32(0, png.data() + at, (uInt)(png.size() - at)));
}
// colour_type: 0 gray, 2 RGB, 4 gray+alpha, 6 RGBA; every scanline filter type in turn
static std::vector<unsigned char> make_png(int w, int h, int depth, int colour_type, uint32_t seed)
{
    const int ch = colour_type == 0 ? 1 : colour_type == 2 ? 3 : colour_type == 4 ? 2 : 4, bpp = ch * depth / 8;
    std::vector<unsigned char> raw;
    std::vector<unsigned char> prev((size_t)w * bpp, 0), cur((size_t)w * bpp);
    for (int y = 0; y < h; y++) {
        for (auto& b : cur) { seed = seed * 1664525u + 1013904223u; b = (unsigned char)(seed >> 24); }
        const int ft = y % 5;
        raw.push_back((unsigned char)ft);
        for (int i = 0; i < w * bpp; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) / 2;
            else if (ft == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            raw.push_back((unsigned char)(cur[i] - pred));
        }
        prev = cur;
    }
    uLongf zl = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zl);
    compress2(z.data(), &zl, raw.data(), (uLong)raw.size(), 6);
    z.resize(zl);
    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a}, ihdr;
    put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
    ihdr.push_back((unsigned char)depth); ihdr.push_back((unsigned char)colour_type); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(png, "IHDR", ihdr);
    // two IDAT chunks: the reader has to concatenate them
    const size_t half = z.size() / 2;
    chunk(png, "IDAT", std::vector<unsigned char>(z.begin(), z.begin() + half));
    chunk(png, "IDAT", std::vector<unsigned char>(z.begin() + half, z.end()));
    chunk(png, "IEND", {});
    return png;
}
static void write_file(const std::string& path, const void* p, size_t n)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    if (n) fwrite(p, 1, n, f);
    fclose(f);
}
static int read_png(const std::string& path)
{
    int w = 0, h = 0, c = 0, b = 0;
    const int rc = dvo_png_info(path.c_str(), &w, &h, &c, &b);
    if (rc != 0) return rc;
    std::vector<unsigned char> px((size_t)w * h * c * (b / 8));
    const int rc2 = dvo_png_read(path.c_str(), px.data(), px.size());
    if (rc2 == 0 && px.size() > 4) dvo_png_read(path.c_str(), px.data(), px.size() - 1);  // capacity too small: must refuse
    return rc2;
}

int main(int argc, char** argv)
{
    const std::string dir = std::string(argc > 1 ? argv[1] : "/tmp") + "/dvo_asan_host";
    (void)!system(("mkdir -p " + dir + "/tum/rgb " + dir + "/tum/depth").c_str());
    const std::string f = dir + "/t.png";
    int ok_files = 0;
    // every supported layout
    for (int ct : {0, 2, 4, 6})
        for (int depth : {8, 16}) {
            const auto png = make_png(37, 11, depth, ct, 17u * (uint32_t)ct + (uint32_t)depth);
            write_file(f, png.data(), png.size());
            if (read_png(f) != 0) { fprintf(stderr, "valid PNG refused (colour type %d, depth %d): %s\n", ct, depth, dvo::g_err.c_str()); return 1; }
            ok_files++;
        }
    // crafted headers: huge / zero sizes, wrong IHDR length, missing IHDR, truncated file, bad colour type, interlace, empty file
    {
        auto png = make_png(8, 8, 8, 0, 5u);
        auto bad = png; bad[16] = 0x7f; bad[17] = 0xff; write_file(f, bad.data(), bad.size()); read_png(f);      // width 2^31
        bad = png; memset(&bad[16], 0, 4); write_file(f, bad.data(), bad.size()); read_png(f);                    // width 0
        bad = png; bad[11] = 12; write_file(f, bad.data(), bad.size()); read_png(f);                              // IHDR length 12
        bad = png; memcpy(&bad[12], "IDAT", 4); write_file(f, bad.data(), bad.size()); read_png(f);               // no IHDR first
        bad = png; bad[25] = 3; write_file(f, bad.data(), bad.size()); read_png(f);                               // palette
        bad = png; bad[28] = 1; write_file(f, bad.data(), bad.size()); read_png(f);                               // interlaced
        bad = png; bad[24] = 4; write_file(f, bad.data(), bad.size()); read_png(f);                               // 4-bit
        for (size_t cut : {size_t(0), size_t(7), size_t(20), size_t(33), size_t(40), png.size() - 13, png.size() - 1}) {
            write_file(f, png.data(), cut);
            read_png(f);
        }
        // declared size larger than the pixel data the stream holds
        bad = png; bad[19] = 64; bad[23] = 64; write_file(f, bad.data(), bad.size()); read_png(f);
        read_png(dir + "/does_not_exist.png");
    }
    // mutation fuzz: random byte flips / overwrites in valid files (CRC is not checked by the reader, so these reach the decoder)
    {
        uint32_t s = 12345u;
        int accepted = 0;
        for (int it = 0; it < 3000; it++) {
            auto png = make_png(5 + it % 13, 3 + it % 7, (it & 1) ? 16 : 8, (it % 4) * 2, (uint32_t)it);
            const int nmut = 1 + it % 4;
            for (int m = 0; m < nmut; m++) {
                s = s * 1664525u + 1013904223u;
                const size_t pos = 8 + (s >> 8) % (png.size() - 8);
                s = s * 1664525u + 1013904223u;
                png[pos] = (it & 2) ? (unsigned char)(s >> 24) : (unsigned char)(png[pos] ^ (1u << ((s >> 24) & 7)));
            }
            write_file(f, png.data(), png.size());
            accepted += read_png(f) == 0;
        }
        printf("fuzz: %d of 3000 mutated files still decode\n", accepted);
    }
    // TUM directory with comments, ragged lines, unmatched stamps, ground truth; the reference's list files
    {
        const auto png = make_png(16, 12, 8, 2, 1u), dpng = make_png(16, 12, 16, 0, 2u);
        write_file(dir + "/tum/rgb/1.png", png.data(), png.size());
        write_file(dir + "/tum/depth/1.png", dpng.data(), dpng.size());
        const char* rgb = "# color images\n# timestamp filename\n1.000 rgb/1.png\n1.033 rgb/2.png\nnot_a_number x\n\n2.000 rgb/3.png\n3.5\n";
        const char* dep = "# depth\n1.005 depth/1.png\n1.040 depth/2.png\n9.0 depth/9.png\n";
        const char* gt = "# gt\n1.001 0 0 0 0 0 0 1\n1.030 0.1 0 0 0 0 0\n";  // second line one value short
        write_file(dir + "/tum/rgb.txt", rgb, strlen(rgb));
        write_file(dir + "/tum/depth.txt", dep, strlen(dep));
        write_file(dir + "/tum/groundtruth.txt", gt, strlen(gt));
        dvo_dataset* ds = nullptr;
        if (dvo_dataset_open_tum((dir + "/tum").c_str(), 0.02, &ds) != 0 || dvo_dataset_size(ds) != 2) { fprintf(stderr, "TUM association failed\n"); return 1; }
        for (int i = -1; i <= dvo_dataset_size(ds); i++) {
            double t; char a[256], b[256]; float g7[7];
            dvo_dataset_entry(ds, i, &t, a, b, 256, g7);
            dvo_dataset_entry(ds, i, &t, a, b, 8, g7);   // path capacity too small
            dvo_dataset_entry(ds, i, nullptr, nullptr, nullptr, 0, nullptr);
            dvo_dataset_entry(ds, i, &t, a, b, 0, g7);   // no room at all: refused
        }
        dvo_dataset_close(ds);
        ds = nullptr;
        dvo_dataset_open_tum((dir + "/nowhere").c_str(), 0.02, &ds);
        write_file(dir + "/tum/depth.txt", "", 0);
        if (dvo_dataset_open_tum((dir + "/tum").c_str(), 0.02, &ds) == 0) dvo_dataset_close(ds);
        const char* list = "a.png\nb.png  b_depth.png\n\n   \nc.png d.png extra\n";
        write_file(dir + "/info.txt", list, strlen(list));
        if (dvo_dataset_open_list(dir.c_str(), nullptr, &ds) != 0 || dvo_dataset_size(ds) != 3) { fprintf(stderr, "list file failed\n"); return 1; }
        dvo_dataset_close(ds);
        if (dvo_dataset_open_list(dir.c_str(), (dir + "/info.txt").c_str(), &ds) == 0) dvo_dataset_close(ds);
        dvo_dataset_open_list(dir.c_str(), (dir + "/missing.txt").c_str(), &ds);
        dvo_dataset_close(nullptr);
    }
    // trajectory evaluation
    {
        const int n = 40;
        std::vector<float> est(n * 3), gtp(n * 3), Te(n * 16), Tg(n * 16);
        for (int i = 0; i < n; i++) {
            const float a = 0.1f * i;
            const float p[3] = {cosf(a), sinf(a), 0.05f * i};
            for (int k = 0; k < 3; k++) { gtp[i * 3 + k] = p[k]; est[i * 3 + k] = 1.1f * p[(k + 1) % 3] + 0.3f + 0.001f * sinf(7.0f * i + k); }
            const float c = cosf(a), s = sinf(a);
            const float T[16] = {c, -s, 0, p[0], s, c, 0, p[1], 0, 0, 1, p[2], 0, 0, 0, 1};
            memcpy(&Tg[i * 16], T, sizeof(T));
            memcpy(&Te[i * 16], T, sizeof(T));
            Te[i * 16 + 3] += 0.002f * sinf(3.0f * i);
        }
        double rmse, R[9], t[3], sc, tr, rr;
        for (int ws = 0; ws < 2; ws++)
            if (dvo_eval_ate(n, est.data(), gtp.data(), ws, &rmse, R, t, &sc) != 0) { fprintf(stderr, "ATE failed\n"); return 1; }
        dvo_eval_ate(n, est.data(), gtp.data(), 1, &rmse, nullptr, nullptr, nullptr);
        dvo_eval_ate(2, est.data(), gtp.data(), 0, &rmse, R, t, &sc);   // degenerate: too few / collinear points
        dvo_eval_ate(0, est.data(), gtp.data(), 0, &rmse, R, t, &sc);
        std::vector<float> same(n * 3, 1.0f);
        dvo_eval_ate(n, same.data(), same.data(), 1, &rmse, R, t, &sc);  // all points equal
        for (int delta : {1, 5, n - 1, n, 0}) dvo_eval_rpe(n, Te.data(), Tg.data(), delta, &tr, &rr);
        float inv[16];
        dvo_pose_inverse(&Tg[16], inv);
        std::vector<double> stamps(n);
        for (int i = 0; i < n; i++) stamps[i] = 0.033 * i;
        dvo_traj_write_tum((dir + "/traj.txt").c_str(), n, stamps.data(), Te.data());
        dvo_traj_write_tum((dir + "/traj2.txt").c_str(), n, nullptr, Te.data());
        dvo_traj_write_tum((dir + "/no/such/dir/traj.txt").c_str(), n, nullptr, Te.data());
    }
    (void)!system(("rm -rf " + dir).c_str());
    printf("asan driver ok (%d valid PNG layouts)\n", ok_files);
    return 0;
}
