// dvo_io.cpp -- dataset front-end: what src/core/loader.cpp does for the reference, without
// OpenCV: a small zlib-based PNG reader (8/16-bit gray, RGB, RGBA, non-interlaced), the reference's list-file
// format (include/core/loader.hpp:28-52,77-105: one filename per line, or "rgb depth" per line) and the TUM RGB-D
// directory layout (rgb.txt / depth.txt / groundtruth.txt, associated by timestamp).  Host only: no GPU needed.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <stdexcept>
#include <sstream>
#include <string>
#include <vector>

#include "dvo_engine.h"

namespace dvo {

// ------------------------------------------------------------------------------------------------ PNG
static uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

struct PngImage {
    int w = 0, h = 0, channels = 0, bit_depth = 0;
    std::vector<unsigned char> data;  // 8-bit: bytes; 16-bit: host-endian uint16
};

static int png_decode_impl(const std::string& path, PngImage& img, bool header_only);

// The decoder trusts nothing in the file (a crafted PNG in a dataset directory must not take the process down): sizes are
// bounded before any allocation and no C++ exception crosses the extern "C" boundary.
static int png_decode(const std::string& path, PngImage& img, bool header_only)
{
    try {
        return png_decode_impl(path, img, header_only);
    } catch (const std::bad_alloc&) {
        set_error(path + ": out of memory while decoding");
        return DVO_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& e) {
        set_error(path + ": " + e.what());
        return DVO_ERR_BAD_ARGUMENT;
    }
}

static int png_decode_impl(const std::string& path, PngImage& img, bool header_only)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { set_error("cannot open " + path); return DVO_ERR_BAD_ARGUMENT; }
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (buf.size() < 33 || memcmp(buf.data(), sig, 8) != 0) { set_error(path + ": not a PNG file"); return DVO_ERR_BAD_ARGUMENT; }
    size_t pos = 8;
    std::vector<unsigned char> idat;
    int color_type = -1, interlace = 0;
    bool first = true;
    while (pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        const char* type = reinterpret_cast<const char*>(&buf[pos + 4]);
        if (pos + 12 + (size_t)len > buf.size()) { set_error(path + ": truncated chunk"); return DVO_ERR_BAD_ARGUMENT; }
        const unsigned char* d = &buf[pos + 8];
        if (first != (memcmp(type, "IHDR", 4) == 0)) { set_error(path + ": IHDR must be the first chunk, once"); return DVO_ERR_BAD_ARGUMENT; }
        first = false;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) { set_error(path + ": bad IHDR length"); return DVO_ERR_BAD_ARGUMENT; }
            const uint32_t uw = be32(d), uh = be32(d + 4);
            if (uw == 0 || uh == 0 || uw > 16384u || uh > 16384u) { set_error(path + ": image size outside [1, 16384]"); return DVO_ERR_BAD_ARGUMENT; }
            img.w = (int)uw; img.h = (int)uh;
            img.bit_depth = d[8]; color_type = d[9]; interlace = d[12];
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    switch (color_type) {
        case 0: img.channels = 1; break;
        case 2: img.channels = 3; break;
        case 4: img.channels = 2; break;
        case 6: img.channels = 4; break;
        default: set_error(path + ": unsupported PNG colour type (palette?)"); return DVO_ERR_BAD_ARGUMENT;
    }
    if ((img.bit_depth != 8 && img.bit_depth != 16) || interlace != 0 || img.w <= 0 || img.h <= 0) {
        set_error(path + ": only non-interlaced 8/16-bit PNGs are supported");
        return DVO_ERR_BAD_ARGUMENT;
    }
    if (header_only) return DVO_OK;
    const int bpp = img.channels * img.bit_depth / 8;  // bytes per pixel
    const size_t stride = (size_t)img.w * bpp;
    std::vector<unsigned char> raw((stride + 1) * (size_t)img.h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) {
        set_error(path + ": zlib inflate failed");
        return DVO_ERR_BAD_ARGUMENT;
    }
    img.data.assign(stride * (size_t)img.h, 0);
    for (int y = 0; y < img.h; y++) {
        const unsigned char* in = &raw[(stride + 1) * (size_t)y];
        const int ft = in[0];
        unsigned char* cur = &img.data[stride * (size_t)y];
        const unsigned char* up = y ? cur - stride : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            int v = in[i + 1];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: set_error(path + ": bad PNG filter"); return DVO_ERR_BAD_ARGUMENT;
            }
            cur[i] = (unsigned char)v;
        }
    }
    if (img.bit_depth == 16) {  // big endian on disk -> host uint16
        uint16_t* p = reinterpret_cast<uint16_t*>(img.data.data());
        const size_t n = img.data.size() / 2;
        for (size_t i = 0; i < n; i++) {
            const unsigned char* q = &img.data[2 * i];
            p[i] = (uint16_t)((q[0] << 8) | q[1]);
        }
    }
    return DVO_OK;
}

// ------------------------------------------------------------------------------------------------ datasets
struct DatasetEntry {
    double t = 0;
    std::string rgb, depth;
    float gt[7] = {NAN, NAN, NAN, NAN, NAN, NAN, NAN};
};

struct Dataset {
    std::vector<DatasetEntry> e;
};

static bool read_stamped(const std::string& file, std::vector<std::pair<double, std::string>>& out)
{  // TUM "timestamp payload..." lines, '#' comments
    std::ifstream f(file);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        double t;
        if (!(ss >> t)) continue;
        std::string rest;
        std::getline(ss, rest);
        const size_t b = rest.find_first_not_of(" \t");
        out.emplace_back(t, b == std::string::npos ? std::string() : rest.substr(b));
    }
    return true;
}

static int open_tum(const std::string& dir, double max_dt, Dataset& ds)
{
    std::vector<std::pair<double, std::string>> rgb, depth, gt;
    if (!read_stamped(dir + "/rgb.txt", rgb) || !read_stamped(dir + "/depth.txt", depth)) {
        set_error("TUM directory needs rgb.txt and depth.txt: " + dir);
        return DVO_ERR_BAD_ARGUMENT;
    }
    read_stamped(dir + "/groundtruth.txt", gt);  // optional
    size_t j = 0, g = 0;
    for (const auto& r : rgb) {  // nearest depth stamp (both lists are sorted)
        while (j + 1 < depth.size() && std::fabs(depth[j + 1].first - r.first) <= std::fabs(depth[j].first - r.first)) j++;
        if (depth.empty() || std::fabs(depth[j].first - r.first) > max_dt) continue;
        DatasetEntry e;
        e.t = r.first;
        e.rgb = dir + "/" + r.second;
        e.depth = dir + "/" + depth[j].second;
        if (!gt.empty()) {
            while (g + 1 < gt.size() && std::fabs(gt[g + 1].first - r.first) <= std::fabs(gt[g].first - r.first)) g++;
            if (std::fabs(gt[g].first - r.first) <= max_dt) {
                std::istringstream ss(gt[g].second);
                for (int k = 0; k < 7; k++) ss >> e.gt[k];
            }
        }
        ds.e.push_back(e);
    }
    return DVO_OK;
}

static int open_list(const std::string& dir, const std::string& list_file, Dataset& ds)
{  // include/core/loader.hpp:38-47,87-98
    std::ifstream f(list_file.empty() ? dir + "/info.txt" : list_file);
    if (!f) { set_error("cannot open the list file of " + dir); return DVO_ERR_BAD_ARGUMENT; }  // the reference abort()s here
    std::string line;
    double t = 0;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string a, b;
        if (!(ss >> a)) continue;
        DatasetEntry e;
        e.t = t;
        t += 1.0;
        e.rgb = dir + "/" + a;
        if (ss >> b) e.depth = dir + "/" + b;
        ds.e.push_back(e);
    }
    return DVO_OK;
}

}  // namespace dvo

using namespace dvo;

struct dvo_dataset { Dataset impl; };

extern "C" {

int dvo_png_info(const char* path, int* width, int* height, int* channels, int* bit_depth)
{
    if (!path) return DVO_ERR_BAD_ARGUMENT;
    PngImage img;
    DVO_TRY(png_decode(path, img, true));
    if (width) *width = img.w;
    if (height) *height = img.h;
    if (channels) *channels = img.channels;
    if (bit_depth) *bit_depth = img.bit_depth;
    return DVO_OK;
}

int dvo_png_read(const char* path, void* pixels, size_t capacity_bytes)
{
    if (!path || !pixels) return DVO_ERR_BAD_ARGUMENT;
    PngImage img;
    DVO_TRY(png_decode(path, img, false));
    if (img.data.size() > capacity_bytes) { set_error("pixel buffer too small"); return DVO_ERR_BAD_ARGUMENT; }
    memcpy(pixels, img.data.data(), img.data.size());
    return DVO_OK;
}

int dvo_dataset_open_tum(const char* dir, double max_dt, dvo_dataset** out)
{
    if (!dir || !out) return DVO_ERR_BAD_ARGUMENT;
    dvo_dataset* d = new dvo_dataset();
    const int st = open_tum(dir, max_dt > 0 ? max_dt : 0.02, d->impl);
    if (st != DVO_OK) { delete d; return st; }
    *out = d;
    return DVO_OK;
}

int dvo_dataset_open_list(const char* dir, const char* list_file, dvo_dataset** out)
{
    if (!dir || !out) return DVO_ERR_BAD_ARGUMENT;
    dvo_dataset* d = new dvo_dataset();
    const int st = open_list(dir, list_file ? list_file : "", d->impl);
    if (st != DVO_OK) { delete d; return st; }
    *out = d;
    return DVO_OK;
}

int dvo_dataset_size(const dvo_dataset* d) { return d ? (int)d->impl.e.size() : 0; }

int dvo_dataset_entry(const dvo_dataset* d, int i, double* timestamp, char* rgb_path, char* depth_path, int path_capacity, float gt_pose7[7])
{
    if (!d || i < 0 || i >= (int)d->impl.e.size()) return DVO_ERR_BAD_ARGUMENT;
    if ((rgb_path || depth_path) && path_capacity <= 0) return DVO_ERR_BAD_ARGUMENT;
    const DatasetEntry& e = d->impl.e[i];
    if (timestamp) *timestamp = e.t;
    if (rgb_path) snprintf(rgb_path, (size_t)path_capacity, "%s", e.rgb.c_str());
    if (depth_path) snprintf(depth_path, (size_t)path_capacity, "%s", e.depth.c_str());
    if (gt_pose7) memcpy(gt_pose7, e.gt, sizeof e.gt);
    return DVO_OK;
}

int dvo_dataset_close(dvo_dataset* d)
{
    delete d;
    return DVO_OK;
}

}  // extern "C"
