// dvo.hpp -- header-only C++17 facade over the C ABI of libdvo.so (include/dvo.h).
//
// Mirrors the public surface of the reference's System::VisualOdometry (include/system/system.hpp:12-104):
// the same three entry points (constructor from K, odometrize, odometrizeUsingDepth) plus keyframe / depth-map
// access (include/system/frame.hpp:125-139,146-188).  Images are row-major float buffers instead of cv::Mat1f;
// INTEGRATION.md shows the cv::Mat adaptor a maintainer of the reference would write on top of this.
// Errors surface as dvo::Error (the reference abort()s or throws std::out_of_range).
#pragma once
#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "dvo.h"

namespace dvo {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

inline void check(int st)
{
    if (st != DVO_OK) throw Error(st, std::string(dvo_status_string(st)) + ": " + dvo_last_error());
}

using Mat4 = std::array<float, 16>;  // row-major 4x4 pose
using Vec6 = std::array<float, 6>;   // twist (vx, vy, vz, wx, wy, wz), src/math/se3.cpp:74-75
using Mat3 = std::array<float, 9>;   // row-major intrinsics

inline dvo_config default_config()
{
    dvo_config c;
    dvo_config_default(&c);
    return c;
}

struct Keyframe {  // one level of a System::Frame (include/system/frame.hpp:72-144)
    int id = -1, levels = 0, level = 0, width = 0, height = 0;
    Vec6 xi{}, relative_xi{};
    Mat3 K{};
    std::vector<float> gray, depth, sigma, age;  // age only on the top level
};

class VisualOdometry {
public:
    // VisualOdometry(const cv::Mat1f& K), system.hpp:15
    VisualOdometry(const Mat3& K, int width, int height, const dvo_config* cfg = nullptr) : w_(width), h_(height)
    {
        check(dvo_vo_create(K.data(), width, height, cfg, &vo_));
    }
    // VisualOdometry(gray, depth, sigma, K), system.hpp:24-32
    VisualOdometry(const float* gray, const float* depth, const float* sigma, const Mat3& K, int width, int height,
                   const dvo_config* cfg = nullptr)
        : VisualOdometry(K, width, height, cfg)
    {
        check(dvo_vo_init_keyframe(vo_, gray, depth, sigma));
    }
    ~VisualOdometry() { dvo_vo_destroy(vo_); }
    VisualOdometry(const VisualOdometry&) = delete;
    VisualOdometry& operator=(const VisualOdometry&) = delete;

    // replaces the cv::randn initial depth (include/system/frame.hpp:17-21): maps at width/4 x height/4
    void setInitialDepth(const float* depth, const float* sigma) { check(dvo_vo_set_initial_depth(vo_, depth, sigma)); }

    // cv::Mat1f odometrize(const cv::Mat1f& gray), system.hpp:44-74: 4x4 world pose exp(m_xi)
    Mat4 odometrize(const float* gray, bool* is_keyframe = nullptr)
    {
        Mat4 T;
        int key = 0;
        check(dvo_vo_odometrize(vo_, gray, T.data(), &key));
        if (is_keyframe) *is_keyframe = key != 0;
        return T;
    }
    // the same from a raw u8 frame (gray / R,G,B / R,G,B,A), converted on the device
    Mat4 odometrizeRaw(const uint8_t* rgb, int channels, bool* is_keyframe = nullptr)
    {
        Mat4 T;
        int key = 0;
        check(dvo_vo_odometrize_raw(vo_, rgb, channels, T.data(), &key));
        if (is_keyframe) *is_keyframe = key != 0;
        return T;
    }
    // cv::Mat1f odometrizeUsingDepth(gray, depth, sigma), system.hpp:77-93: 4x4 relative pose
    Mat4 odometrizeUsingDepth(const float* gray, const float* depth, const float* sigma)
    {
        Mat4 T;
        check(dvo_vo_odometrize_depth(vo_, gray, depth, sigma, T.data()));
        return T;
    }

    // FrameHistory::size / operator[] (frame.hpp:174-176); index 0 = oldest keyframe
    int keyframeCount() const { return dvo_vo_keyframe_count(vo_); }
    Keyframe keyframe(int index, int level = -1) const
    {
        Keyframe k;
        int tw = 0, th = 0;
        check(dvo_vo_keyframe_info(vo_, index, &k.id, &k.levels, &tw, &th, k.xi.data(), k.relative_xi.data()));
        k.level = level < 0 ? k.levels - 1 : level;
        const int shift = k.levels - 1 - k.level;
        k.width = tw >> shift;
        k.height = th >> shift;
        const size_t n = (size_t)k.width * k.height;
        k.gray.resize(n); k.depth.resize(n); k.sigma.resize(n);
        float* age = nullptr;
        if (shift == 0) { k.age.resize(n); age = k.age.data(); }
        check(dvo_vo_keyframe_get(vo_, index, k.level, k.gray.data(), k.depth.data(), k.sigma.data(), age, k.K.data()));
        return k;
    }
    dvo_track_log lastTrackLog() const
    {
        dvo_track_log log;
        check(dvo_vo_last_track_log(vo_, &log));
        return log;
    }
    int width() const { return w_; }
    int height() const { return h_; }
    dvo_vo* handle() { return vo_; }

private:
    dvo_vo* vo_ = nullptr;
    int w_, h_;
};

// n_seq independent sequences on one GPU (frame-to-frame tracking with sensor depth)
class BatchTracker {
public:
    BatchTracker(int n_seq, const Mat3& K, int width, int height, int levels = 4, int culls = 1, const dvo_config* cfg = nullptr)
        : n_(n_seq)
    {
        check(dvo_batch_create(n_seq, K.data(), width, height, levels, culls, cfg, &b_));
    }
    ~BatchTracker() { dvo_batch_destroy(b_); }
    BatchTracker(const BatchTracker&) = delete;
    BatchTracker& operator=(const BatchTracker&) = delete;
    void pushDevice(const float* gray, const float* depth, const float* sigma) { check(dvo_batch_push_device(b_, gray, depth, sigma)); }
    void pushHost(const float* gray, const float* depth, const float* sigma) { check(dvo_batch_push_host(b_, gray, depth, sigma)); }
    // raw sensor frames as cv::imread delivers them (u8 gray / RGB(A), u16 depth): converted inside the pyramid kernel
    void pushRawDevice(const uint8_t* rgb, int channels, const uint16_t* depth16, float depth_scale = 0.0f)
    { check(dvo_batch_push_raw_device(b_, rgb, channels, depth16, depth_scale)); }
    void pushRawHost(const uint8_t* rgb, int channels, const uint16_t* depth16, float depth_scale = 0.0f)
    { check(dvo_batch_push_raw_host(b_, rgb, channels, depth16, depth_scale)); }
    std::vector<Vec6> lastTwists()
    {
        std::vector<Vec6> out(n_);
        check(dvo_batch_last_poses(b_, out[0].data(), nullptr));
        return out;
    }
    std::vector<Mat4> lastPoses()
    {
        std::vector<Mat4> out(n_);
        check(dvo_batch_last_poses(b_, nullptr, out[0].data()));
        return out;
    }
    dvo_batch* handle() { return b_; }

private:
    dvo_batch* b_ = nullptr;
    int n_;
};

// n_seq independent MONO sequences on one GPU: System::VisualOdometry::odometrize (system.hpp:44-74: track + Mapper::estimate +
// regularize) for every sequence per call; Mapper::needNewFrame (src/map/mapper.cpp:45-60) is decided per sequence on the device
class BatchMono {
public:
    BatchMono(int n_seq, const Mat3& K, int width, int height, int ring_keyframes = 8, const dvo_config* cfg = nullptr) : n_(n_seq)
    {
        check(dvo_batch_create_mono(n_seq, K.data(), width, height, ring_keyframes, cfg, &b_));
    }
    ~BatchMono() { dvo_batch_destroy(b_); }
    BatchMono(const BatchMono&) = delete;
    BatchMono& operator=(const BatchMono&) = delete;
    void setInitialDepth(const float* depth, const float* sigma) { check(dvo_batch_set_initial_depth(b_, depth, sigma)); }
    void odometrizeDevice(const float* gray) { check(dvo_batch_odometrize_device(b_, gray)); }
    void odometrizeRawDevice(const uint8_t* rgb, int channels) { check(dvo_batch_odometrize_raw_device(b_, rgb, channels)); }
    std::vector<Mat4> worldPoses(std::vector<int>* is_keyframe = nullptr)
    {
        std::vector<Mat4> out(n_);
        if (is_keyframe) is_keyframe->resize(n_);
        check(dvo_batch_world_poses(b_, nullptr, out[0].data(), is_keyframe ? is_keyframe->data() : nullptr));
        return out;
    }
    dvo_batch* handle() { return b_; }

private:
    dvo_batch* b_ = nullptr;
    int n_;
};

}  // namespace dvo
