/*
 * dvo.h -- C ABI of libdvo.so: MI355X-native semi-dense direct visual odometry hot path.
 *
 * Drop-in boundary for the tracking/mapping path of KYabuuchi/direct-visual-odometry.  The reference has
 * no FFI/plugin layer (plain C++ classes over cv::Mat), so each entry point below names the C++ interface
 * it replaces (file:line under the reference tree).  All images are row-major contiguous float32, one
 * channel; gray in [0,1]; DVO_INVALID (-2.0f) marks unusable pixels (include/math/util.hpp:7-10); depth
 * and sigma in metres, depth 0 = none; K is a row-major 3x3; a twist xi is (vx,vy,vz,wx,wy,wz)
 * (src/math/se3.cpp:74-75); poses are row-major 4x4.
 *
 * Ownership: the caller owns every buffer it passes; the library copies on entry and never keeps a host
 * pointer.  `*_device` entry points take HIP device pointers (resident inputs, zero copy).
 * Errors: every call returns a dvo_status; nothing aborts or throws (the reference abort()s or throws
 * std::out_of_range: src/core/transform.cpp:16-17, include/system/frame.hpp:125).  Data sentinels are the
 * reference's: residual -1 and a zero update when no pixel is valid (src/track/optimize.cpp:92-93).
 * Threading: one handle = one HIP stream, used by one thread at a time; distinct handles are independent.
 */
#ifndef DVO_H
#define DVO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DVO_INVALID (-2.0f)
#define DVO_MAX_LEVELS 8
#define DVO_MAX_ITERATIONS 32

typedef enum {
    DVO_OK = 0,
    DVO_ERR_BAD_ARGUMENT = 1,
    DVO_ERR_HIP = 2,          /* a HIP runtime call failed; dvo_last_error() has the text            */
    DVO_ERR_NO_DEVICE = 3,    /* no gfx950 device visible: the library has NO CPU fallback            */
    DVO_ERR_NO_VALID_PIXELS = 4, /* informational: an optimize step had n_valid == 0                  */
    DVO_ERR_NOT_READY = 5,    /* e.g. track requested before a reference frame exists                 */
    DVO_ERR_OUT_OF_MEMORY = 6
} dvo_status;

/* Constants of the reference, runtime-configurable.  dvo_config_default() fills the reference's literals. */
typedef struct dvo_config {
    int      max_iterations;        /* 15    src/track/tracker.cpp:19                                   */
    float    min_update;            /* 5e-4  src/track/tracker.cpp:17                                   */
    float    min_residual;          /* 5e-3  src/track/tracker.cpp:16                                   */
    int      fixed_iterations;      /* 0 = early exit as the reference; N>0 = exactly N per level       */
    int      crop_enable;           /* 1 = level-2 crop of optimize.cpp:33-36 and mapper.cpp:90 crop    */
    float    step_default;          /* 2.0   src/track/optimize.cpp:22                                  */
    float    step_level1;           /* 1.5   src/track/optimize.cpp:23-24                               */
    float    step_level2;           /* 1.0   src/track/optimize.cpp:25-26                               */
    float    sigma_min, sigma_max;  /* 0.01, 0.5  src/track/optimize.cpp:83                             */
    float    min_depth;             /* 0.20  src/track/optimize.cpp:39                                  */
    float    keyframe_min_translation; /* 0.02 src/map/mapper.cpp:12                                    */
    int      keyframe_max_frames;   /* 6     src/map/mapper.cpp:13                                      */
    uint32_t rng_seed;              /* seed of the counter-based reset depth (replaces gaussian.cpp:8-9) */
    int      device;                /* HIP device ordinal                                                */
    void*    stream;                /* hipStream_t to launch on, or NULL for a library-owned stream      */
    int      profile;               /* 1 = bracket every k_track_gn launch with hipEvents (bench only)   */
    int      gn_pixels_per_thread;  /* 0 = choose from the problem size                                  */
    int      gn_use_lds_patch;      /* -1 = auto, 0 = global gathers, 1 = LDS-staged reference patch     */
    int      gn_gather_group;       /* 0 = auto; pixels per thread whose gathers are in flight together  */
    int      track_streams;         /* 0 = auto; sub-batches of a dvo_batch tracked on concurrent HIP streams */
    int      track_adaptive;        /* 0 = auto (on), -1 = off: the host stays two iterations ahead of the GPU and stops a level's launches once no sequence is active */
    int      track_fused_tiles;     /* N > 0: levels of at most N (<= 8) 1024-px tiles run all iterations in ONE launch; 0 = off (default) */
    int      track_single_launch;   /* 0 = auto: a dvo_vo handle's sensor-depth tracking runs in ONE launch per call (k_track_persist: co-resident workgroups,
                                       every wait bounded, falls back by itself); other handles of <= 8 sequences run one launch per Gauss-Newton
                                       iteration (the workgroup that finishes a sequence's last tile solves); 1 = one launch per iteration at most;
                                       -1 = launch pairs only.  Every setting gives the same bits at the same tile size */
} dvo_config;

void        dvo_config_default(dvo_config* cfg);
const char* dvo_version(void);
const char* dvo_status_string(int status);
const char* dvo_last_error(void);          /* thread-local text of the last failure */
int         dvo_device_count(void);        /* number of visible HIP devices (0 = none) */

/* ------------------------------------------------------------------------------------------------
 * System::VisualOdometry (include/system/system.hpp:12-104): frame in, pose out, one sequence.
 * ------------------------------------------------------------------------------------------------ */
typedef struct dvo_vo dvo_vo;

/* VisualOdometry(const cv::Mat1f& K), system.hpp:15.  width/height = size of the frames that will be fed. */
int dvo_vo_create(const float K[9], int width, int height, const dvo_config* cfg, dvo_vo** out);
int dvo_vo_destroy(dvo_vo* vo);
/* Replaces the cv::randn initial depth of the first mono keyframe (include/system/frame.hpp:17-21):
 * depth/sigma at the culled base resolution (width/4 x height/4).  Optional; default = hash-based
 * N(1.5,0.5) clamped >= 0.5 with sigma 0.5. */
int dvo_vo_set_initial_depth(dvo_vo* vo, const float* depth, const float* sigma);
/* VisualOdometry(gray, depth, sigma, K), system.hpp:24-32: first keyframe with a given depth map. */
int dvo_vo_init_keyframe(dvo_vo* vo, const float* gray, const float* depth, const float* sigma);
/* cv::Mat1f odometrize(const cv::Mat1f& gray), system.hpp:44-74 -> 4x4 world pose exp(m_xi).
 * is_keyframe (optional) receives 1 when the frame was promoted to keyframe. */
int dvo_vo_odometrize(dvo_vo* vo, const float* gray, float T_world[16], int* is_keyframe);
/* the same fed with a raw u8 frame [height][width][channels] (channels 1 gray / 3 R,G,B / 4 R,G,B,A): cv::imread's output before
 * Loader::getNormalizedImages (src/core/loader.cpp:55-62); converted on the device, bit-identical to the float entry point */
int dvo_vo_odometrize_raw(dvo_vo* vo, const uint8_t* rgb, int channels, float T_world[16], int* is_keyframe);
/* cv::Mat1f odometrizeUsingDepth(gray, depth, sigma), system.hpp:77-93 -> 4x4 RELATIVE pose. */
int dvo_vo_odometrize_depth(dvo_vo* vo, const float* gray, const float* depth, const float* sigma, float T_rel[16]);

/* FrameHistory (include/system/frame.hpp:146-188): keyframe / depth-map access. index 0 = oldest. */
int dvo_vo_keyframe_count(const dvo_vo* vo);
int dvo_vo_keyframe_info(const dvo_vo* vo, int index, int* id, int* levels, int* top_width, int* top_height,
                         float xi[6], float rel_xi[6]);
/* Frame::gray/depth/sigma/age/K at a pyramid level (frame.hpp:125-139); any output pointer may be NULL.
 * age is only stored for the top level. */
int dvo_vo_keyframe_get(const dvo_vo* vo, int index, int level, float* gray, float* depth, float* sigma,
                        float* age, float K[9]);
/* the most recent non-keyframe frame's pose (Frame::m_xi, m_relative_xi) */
int dvo_vo_last_frame_pose(const dvo_vo* vo, int* id, float xi[6], float rel_xi[6]);
int dvo_vo_last_valid_updates(const dvo_vo* vo); /* "valid update: N pixel", src/map/mapper.cpp:136 */

/* Per-iteration record the reference prints (src/track/tracker.cpp:56-61). */
typedef struct dvo_track_log {
    int   levels;
    int   n_iter[DVO_MAX_LEVELS];
    float residual[DVO_MAX_LEVELS][DVO_MAX_ITERATIONS];
    float update_norm[DVO_MAX_LEVELS][DVO_MAX_ITERATIONS];
    int   n_valid[DVO_MAX_LEVELS][DVO_MAX_ITERATIONS];
    float xi_after[DVO_MAX_LEVELS][DVO_MAX_ITERATIONS][6];
    float xi_update[DVO_MAX_LEVELS][DVO_MAX_ITERATIONS][6];  /* Outcome::xi_update of the iteration (optimize.cpp:98), before the composition */
} dvo_track_log;
int dvo_vo_last_track_log(const dvo_vo* vo, dvo_track_log* log);
/* Diagnostic (environment DVO_PERSIST_TIMELINE=1): wall-clock stamps (100 MHz) [2][64][8] the solver workgroup and worker 0 of the last
 * one-launch-per-call tracking kernel left per Gauss-Newton step (tools/persist_timeline.py prints them).  DVO_ERR_NOT_READY otherwise. */
int dvo_debug_persist_timeline(dvo_vo* vo, long long* out);

/* ------------------------------------------------------------------------------------------------
 * Batched tracking: n_seq independent sequences on one GPU, frame-to-frame with sensor depth
 * (the odometrizeUsingDepth loop of test/sequence.cpp:10-23, n_seq at a time).  Inputs are device
 * pointers to [n_seq][height][width] float32 arrays already resident in HBM.
 * ------------------------------------------------------------------------------------------------ */
typedef struct dvo_batch dvo_batch;

int dvo_batch_create(int n_seq, const float K[9], int width, int height, int levels, int culls,
                     const dvo_config* cfg, dvo_batch** out);
int dvo_batch_destroy(dvo_batch* b);
/* Frame(gray,depth,sigma,K,levels,culls) for every sequence (frame.hpp:91-106): builds the pyramids of the
 * new frames, tracks them against the previous frames (Tracker::track, tracker.cpp:22-85) and makes them the
 * new reference.  The first call only stores the reference.  Asynchronous on the handle's stream. */
int dvo_batch_push_device(dvo_batch* b, const float* gray_dev, const float* depth_dev, const float* sigma_dev);
/* Optional look-ahead: build the pyramids of a frame that will be pushed later on a library-owned, low-priority side
 * stream, so that this HBM-bound pass runs beside the tracking of the frame pushed in between.  Call order per step:
 *   dvo_batch_prefetch_device(frame k+1); dvo_batch_push_device(frame k);
 * The same three pointers must later be handed to dvo_batch_push_device, in the order the frames were prefetched; at most
 * two may be waiting.  The buffers must be complete when this is called: the read is NOT ordered against the caller's
 * stream. */
int dvo_batch_prefetch_device(dvo_batch* b, const float* gray_dev, const float* depth_dev, const float* sigma_dev);
/* same, from host memory: the H2D copies run on a library-owned copy stream into one of two staging slots, so the transfer of
 * frame k+1 overlaps the tracking of frame k when the caller pushes without synchronising in between.  With PINNED host
 * buffers (hipHostMalloc / hipHostRegister) the copies are asynchronous: leave a buffer unchanged until a later
 * dvo_batch_synchronize / dvo_batch_last_poses.  A pageable buffer has been copied when the call returns (the call waits for the
 * copy, not for the tracking) and may be reused or freed at once. */
int dvo_batch_push_host(dvo_batch* b, const float* gray, const float* depth, const float* sigma);
/* The same three calls fed with RAW sensor frames, [n_seq][height][width] u8 gray / R,G,B / R,G,B,A (channels 1 / 3 / 4) + u16
 * depth: what cv::imread delivers before Loader::getNormalizedImages converts it (src/core/loader.cpp:137-147).  The conversion
 * (BGR2GRAY fixed-point luma, 1/255, depth * depth_scale [0 = 1/5000], sigma 0.1 / 1.0 and INVALID gray where depth == 0:
 * src/core/transform.cpp:60-76) runs inside the pyramid kernel on the pixels the pyramid keeps: 3 B/px (gray) instead of 12 B/px
 * cross PCIe and are read from HBM, and the results are bit-identical to dvo_op_ingest + the float entry points.  The HOST forms
 * (here, dvo_batch_odometrize_raw_host, dvo_vo_odometrize_raw, dvo_vo_odometrize_depth_raw) transfer only the image rows the pyramid
 * keeps -- every 2^culls-th, Convert::cullImage (src/core/convert.cpp:7-20) -- by one strided copy per buffer: half of the bytes
 * for Frame(g,d,s,K,4,1), a quarter for Frame(gray,K,3,2).  DVO_UPLOAD_FULL_FRAMES=1 in the environment restores whole frames. */
int dvo_batch_push_raw_device(dvo_batch* b, const uint8_t* rgb_dev, int channels, const uint16_t* depth16_dev, float depth_scale);
int dvo_batch_prefetch_raw_device(dvo_batch* b, const uint8_t* rgb_dev, int channels, const uint16_t* depth16_dev, float depth_scale);
int dvo_batch_push_raw_host(dvo_batch* b, const uint8_t* rgb, int channels, const uint16_t* depth16, float depth_scale);
/* relative twists [n_seq][6] and 4x4 relative poses [n_seq][16] of the last push (synchronises). NULL = skip */
int dvo_batch_last_poses(dvo_batch* b, float* xi_rel, float* T_rel);
/* asynchronous device-to-device copy of the last push's poses (on the handle's stream): pose-out without a
 * host round trip.  xi_dst_dev [n_seq][6], T_dst_dev [n_seq][16]; either may be NULL. */
int dvo_batch_copy_poses_device(dvo_batch* b, float* xi_dst_dev, float* T_dst_dev);
int dvo_batch_last_track_log(dvo_batch* b, int seq, dvo_track_log* log);
int dvo_batch_synchronize(dvo_batch* b);
/* ---- one block of sequences per GPU (BASELINE config 5) -------------------------------------------
 * The path shards across sequences only (frame t of a sequence tracks against state from frames < t: system.hpp:48,57,67): every
 * rank -- one process per GPU -- owns a contiguous block of the sequences and tracks it with no communication.
 * dvo_shard_range: the block of `rank` (counts differ by at most one; the split of dvo_amd/shard.py and bench.py --gpus N).
 * dvo_batch_gather_poses_rccl: the one collective of the path, from C++: an all-gather of the last push's twists over RCCL / xGMI,
 * queued on the handle's stream.  `rccl_comm` is the caller's ncclComm_t (ncclCommInitRank over its own bootstrap -- MPI,
 * torch.distributed's store, a shared file); every rank must hold the same n_seq (pad the last block).  xi_all_dev: device memory
 * [world_size][n_seq][6].  librccl.so is loaded on first use (dlopen): DVO_ERR_NOT_READY with an explanation when it is absent. */
int dvo_shard_range(int n_sequences, int world_size, int rank, int* first, int* count);
int dvo_batch_gather_poses_rccl(dvo_batch* b, void* rccl_comm, int world_size, float* xi_all_dev);
/* Profile counters (cfg.profile = 1): accumulated over every k_track_gn launch since the last reset. */
typedef struct dvo_gn_profile {
    double   gn_ms;             /* sum of hipEvent-bracketed k_track_gn durations                         */
    uint64_t gn_launches;
    uint64_t gn_pixels;         /* pixels evaluated (active sequences x level pixels), summed              */
    uint64_t gn_iterations;     /* sequence-iterations executed                                            */
} dvo_gn_profile;
int dvo_batch_profile(dvo_batch* b, dvo_gn_profile* out, int reset);
/* Roofline probe: launch k_track_gn `n_launches` times back to back on level `level` with every sequence
 * active (poses unchanged, partials discarded) and return the average duration from two hipEvents. */
int dvo_batch_probe_gn(dvo_batch* b, int level, int n_launches, float* avg_ms, uint64_t* pixels_per_launch);

/* ------------------------------------------------------------------------------------------------
 * Batched MONO pipeline: System::VisualOdometry::odometrize (system.hpp:44-74) = Tracker::track against the newest keyframe +
 * Map::Mapper::estimate / regularize (src/map/mapper.cpp:16-144), n_seq sequences per call.  The keyframe decision
 * (Mapper::needNewFrame, mapper.cpp:45-60) is taken per sequence ON THE DEVICE; one call enqueues a fixed launch sequence and
 * returns.  FrameHistory (include/system/frame.hpp:146-188) is a ring of the newest `ring_keyframes` keyframes per sequence.
 * Every sequence gives the bits a dvo_vo handle with the same config gives (and, past `ring_keyframes` keyframes, one with
 * dvo_vo_set_history_limit(ring_keyframes)).
 * ------------------------------------------------------------------------------------------------ */
int dvo_batch_create_mono(int n_seq, const float K[9], int width, int height, int ring_keyframes, const dvo_config* cfg, dvo_batch** out);
/* Initial depth / sigma of the first keyframes (replaces cv::randn, frame.hpp:17-21) at width/4 x height/4: one host map for every
 * sequence, or device maps [n_seq][height/4][width/4].  Optional; default as dvo_vo. Call before the first frame. */
int dvo_batch_set_initial_depth(dvo_batch* b, const float* depth, const float* sigma);
int dvo_batch_set_initial_depth_device(dvo_batch* b, const float* depth_dev, const float* sigma_dev);
/* odometrize(gray) for every sequence: gray_dev = [n_seq][height][width] float32 in HBM.  Asynchronous on the handle's stream. */
int dvo_batch_odometrize_device(dvo_batch* b, const float* gray_dev);
/* same from raw u8 frames [n_seq][height][width][channels] (channels 1 / 3 / 4), converted inside the pyramid kernel */
int dvo_batch_odometrize_raw_device(dvo_batch* b, const uint8_t* rgb_dev, int channels);
/* the same two from host memory (copy stream + two staging slots, as dvo_batch_push_host) */
int dvo_batch_odometrize_host(dvo_batch* b, const float* gray);
int dvo_batch_odometrize_raw_host(dvo_batch* b, const uint8_t* rgb, int channels);
/* world twists [n_seq][6], world poses exp(xi) [n_seq][16] (system.hpp:73) and keyframe flags [n_seq] of the last frame
 * (synchronises); any pointer may be NULL.  _device: asynchronous device-to-device copies on the handle's stream. */
int dvo_batch_world_poses(dvo_batch* b, float* xi_world, float* T_world, int* is_keyframe);
int dvo_batch_copy_world_poses_device(dvo_batch* b, float* xi_dst_dev, float* T_dst_dev, int* key_dst_dev);
/* the newest keyframe of sequence `seq` (FrameHistory::getRefFrame, frame.hpp:159-166): maps of one pyramid level (age: top level
 * only), its world twist, id, the number of keyframes the sequence has created and the valid-update count of the last frame */
int dvo_batch_keyframe_get(dvo_batch* b, int seq, int level, float* gray, float* depth, float* sigma, float* age, float xi[6], int* id,
                           int* n_keyframes, int* valid_updates);
/* Per-sequence counters of a mono batch.  `clamped_pixels` makes the one deviation of the ring from the reference's unbounded
 * FrameHistory (frame.hpp:146-188) visible: a pixel older than the `ring_keyframes` retained keyframes is searched against the
 * oldest retained one instead of the keyframe it was born in (mapper.cpp:99-101, frame_history[age]); the count is cumulative and
 * stays 0 until a sequence has created more than `ring_keyframes` keyframes AND a pixel has survived all of them. */
typedef struct dvo_mono_stats {
    int frames;                    /* frames consumed (Frame::latest_id + 1) */
    int keyframes_created;
    int ring_keyframes;
    int valid_updates_last_frame;  /* mapper.cpp:136 */
    int clamped_pixels;
} dvo_mono_stats;
int dvo_batch_mono_stats(dvo_batch* b, int seq, dvo_mono_stats* out);
/* Profile of the mapping stages (cfg.profile = 1): hipEvent-bracketed durations on the handle's stream, summed over the frames
 * since the last reset.  depth_update = k_age_table + k_depth_update (Mapper::update), regularize = k_regularize_redecimate
 * (Mapper::regularize + Frame::updateDepth*), propagate = the three k_propagate_* passes (Mapper::propagate). */
typedef struct dvo_map_profile {
    uint64_t frames;
    double   depth_update_ms, regularize_ms, propagate_ms;
    uint64_t update_window_pixels;   /* pixels one k_depth_update launch covers (window of mapper.cpp:90 x n_seq) */
    uint64_t map_pixels;             /* top-level pixels x n_seq (one k_regularize_redecimate launch) */
} dvo_map_profile;
int dvo_batch_profile_mapping(dvo_batch* b, dvo_map_profile* out, int reset);

/* ------------------------------------------------------------------------------------------------
 * Operator level (host pointers): each runs the corresponding HIP kernel once.  Used by the parity
 * tests and reusable on their own.  `dev` is the HIP device ordinal.
 * ------------------------------------------------------------------------------------------------ */
/* Convert::cullImage, src/core/convert.cpp:7-20.  dst is (w>>times) x (h>>times). */
int dvo_op_cull_image(int dev, const float* src, int w, int h, int times, float* dst);
/* Convert::gradiate, src/core/convert.cpp:41-75 */
int dvo_op_gradient(int dev, const float* img, int w, int h, int xdir, float* out);
/* Transform::warpImage, src/core/transform.cpp:35-51 */
int dvo_op_warp_image(int dev, const float xi[6], const float* gray, const float* depth, int w, int h,
                      const float K[9], float* out);
/* Frame pyramid, src/system/frame.cpp:16-37: fills gray/depth/sigma level buffers (coarsest first, each
 * (w>>culls>>(levels-1-i)) x (h>>culls>>(levels-1-i))), any of depth/sigma may be NULL. */
int dvo_op_pyramid(int dev, const float* gray, const float* depth, const float* sigma, int w, int h,
                   int levels, int culls, float* const gray_out[], float* const depth_out[], float* const sigma_out[]);
/* Track::optimize, src/track/optimize.cpp:10-99: one Gauss-Newton step on one level.
 * H = upper triangle of sum J^T J (21), g = sum J^T (w r) (6).  mask (optional, w*h bytes). */
typedef struct dvo_gn_result {
    double H[21];
    double g[6];
    double sum_r2;
    int    n_valid;
    float  xi_update[6];
    float  residual;
    float  xi_next[6];   /* se3::concatenate(xi, xi_update), tracker.cpp:46 */
} dvo_gn_result;
int dvo_op_gn_step(int dev, const dvo_config* cfg, const float* obj_gray, const float* ref_gray,
                   const float* ref_depth, const float* ref_sigma, int w, int h, const float K[9],
                   const float xi[6], int level, dvo_gn_result* out, uint8_t* mask);
/* Tracker::track, src/track/tracker.cpp:22-85, on full-resolution frames (pyramids built on device). */
int dvo_op_track(int dev, const dvo_config* cfg, const float* obj_gray, const float* ref_gray,
                 const float* ref_depth, const float* ref_sigma, int w, int h, const float K[9],
                 int levels, int culls, float xi_out[6], dvo_track_log* log);
/* Map::Implement::propagate, src/map/implement.cpp:217-256 */
int dvo_op_propagate(int dev, const float* ref_depth, const float* ref_sigma, const float* ref_age, int w, int h,
                     const float xi[6], const float K[9], float* depth, float* sigma, float* age);
/* Map::Implement::regularize, src/map/implement.cpp:156-180 */
int dvo_op_regularize(int dev, const float* depth, const float* sigma, int w, int h, float* out);
/* Map::Mapper::update, src/map/mapper.cpp:76-137, against n_hist keyframes (oldest first, the last one is the
 * reference keyframe whose depth/sigma/age are updated in place).  All maps are top-level w x h.
 * hist_gray[i], hist_xi[i] (6 floats each) describe keyframe i. */
int dvo_op_depth_update(int dev, const dvo_config* cfg, int n_hist, const float* const hist_gray[],
                        const float* hist_xi, const float* obj_gray, const float obj_xi[6],
                        const float obj_rel_xi[6], int obj_id, const float K[9], int w, int h,
                        float* ref_depth, float* ref_sigma, float* ref_age, int* valid_updates);
/* math::se3 (src/math/se3.cpp:70-131) evaluated ON THE DEVICE (the tracker's pose chain runs there) */
int dvo_op_se3_exp(int dev, const float xi[6], float T[16]);
int dvo_op_se3_log(int dev, const float T[16], float xi[6]);
int dvo_op_se3_concatenate(int dev, const float a[6], const float b[6], float out[6]);

/* ------------------------------------------------------------------------------------------------
 * Dataset front-end: what src/core/loader.cpp + include/core/loader.hpp do with OpenCV.
 * ------------------------------------------------------------------------------------------------ */
/* PNG reader (cv::imread(IMREAD_UNCHANGED), loader.cpp:61-73,149-160): non-interlaced 8/16-bit gray, gray+alpha, RGB, RGBA.
 * pixels: row-major interleaved channels in FILE order (R,G,B[,A]); 16-bit samples as host-endian uint16. */
int dvo_png_info(const char* path, int* width, int* height, int* channels, int* bit_depth);
int dvo_png_read(const char* path, void* pixels, size_t capacity_bytes);
typedef struct dvo_dataset dvo_dataset;
/* TUM RGB-D directory: rgb.txt + depth.txt associated by nearest timestamp (|dt| <= max_dt, default 0.02 s),
 * optional groundtruth.txt (tx ty tz qx qy qz qw). */
int dvo_dataset_open_tum(const char* dir, double max_dt, dvo_dataset** out);
/* The reference's list files (include/core/loader.hpp:38-47,87-98): "file" or "rgb depth" per line; NULL = dir/info.txt */
int dvo_dataset_open_list(const char* dir, const char* list_file, dvo_dataset** out);
int dvo_dataset_size(const dvo_dataset* d);
int dvo_dataset_entry(const dvo_dataset* d, int i, double* timestamp, char* rgb_path, char* depth_path, int path_capacity,
                      float gt_pose7[7]);
int dvo_dataset_close(dvo_dataset* d);
/* Device-side conversion of raw sensor frames (k_ingest): gray = BGR2GRAY(u8)/255 (loader.cpp:55-60,137-147), depth =
 * u16 * depth_scale (1/5000), sigma = sigma_valid where depth > 0 else sigma_invalid, gray = INVALID where depth == 0 when
 * invalidate_gray (what Transform::mapDepthtoGray leaves, src/core/transform.cpp:60-76).  depth16 may be NULL (gray only). */
int dvo_op_ingest(int dev, const uint8_t* rgb, int channels, const uint16_t* depth16, int w, int h, float depth_scale,
                  float sigma_valid, float sigma_invalid, int invalidate_gray, float* gray, float* depth, float* sigma);
/* odometrizeUsingDepth fed with raw frames (u8 gray/RGB/RGBA + u16 depth): converted on the device, 1.5 instead of 3.7 MB/frame */
int dvo_vo_odometrize_depth_raw(dvo_vo* vo, const uint8_t* rgb, int channels, const uint16_t* depth16, float depth_scale,
                                float T_rel[16]);
/* Loader::getNormalizedUndistortedImages (loader.cpp:15-42): radial-tangential undistortion D = (k1,k2,p1,p2,k3), nearest
 * remap, INVALID border. */
int dvo_op_undistort(int dev, const float* src, int w, int h, const float K[9], const float D[5], float* dst);

/* ------------------------------------------------------------------------------------------------
 * Keyframe store / checkpoint and debug views (row 4).
 * ------------------------------------------------------------------------------------------------ */
/* Dump / restore the whole FrameHistory (include/system/frame.hpp:146-188): per keyframe id, poses, the gray pyramid and the
 * top-level depth / sigma / age.  dvo_vo_load needs a handle created with the same K and frame size; it replaces the history. */
int dvo_vo_save(const dvo_vo* vo, const char* path);
int dvo_vo_load(dvo_vo* vo, const char* path);
/* 0 = keep every keyframe (the reference); N > 0 = keep the newest N (a pixel born in a dropped keyframe then matches
 * against the oldest retained one -- the reference's commented-out `age = std::min(age, 2)`, src/map/mapper.cpp:100). */
int dvo_vo_set_history_limit(dvo_vo* vo, int max_keyframes);
/* False-colour views of src/core/draw.cpp:7-100 as RGB bytes [h][w][3]: mode 0 gray (INVALID blue), 1 depth (hue) with
 * optional sigma (value) in b, 2 sigma, 3 age, 4 gradient.  No GUI: write them with dvo_ppm_write. */
int dvo_op_visualize(int dev, int mode, const float* a, const float* b, int w, int h, uint8_t* rgb);
int dvo_ppm_write(const char* path, const uint8_t* rgb, int w, int h);

/* Device self-test: the kernels' correctly rounded reciprocal (v_rcp_f32 + two FMA corrections inside [2^-100, 2^100], IEEE
 * division elsewhere) against the IEEE division for all 2^32 float bit patterns.  mismatches must come back 0. */
int dvo_selftest_reciprocal(int device, uint64_t* fast_path_inputs, uint64_t* mismatches, uint32_t* first_bad_bits);
/* The regularize kernels' short square root (v_rsq_f32 + 4 operations) against sqrtf for every float in [2^-100, 2^100], and their
 * short division (the reciprocal above + 3 operations) against the IEEE quotient for b = 1.mb, mb = b_first + i * b_stride
 * (i < b_count), times ALL 2^23 mantissas of a in [1, 2); b_first = 0, b_stride = 1, b_count = 2^23 is every mantissa pair (minutes).
 * first_bad_pair = mb << 23 | ma.  mismatches must come back 0. */
int dvo_selftest_sqrt(int device, uint64_t* inputs, uint64_t* mismatches, uint32_t* first_bad_bits);
/* The double-precision sin / cos / atan2 kernels the device's SE(3) chain uses (polynomial kernels instead of the math library's
 * general-purpose routines) against that library on 2^24 arguments of their domain: largest relative differences (expected < 1e-15). */
int dvo_selftest_trig(int device, double* max_rel_sin, double* max_rel_cos, double* max_rel_atan2, uint64_t* samples);
int dvo_selftest_division(int device, uint32_t b_first, uint32_t b_stride, uint32_t b_count, uint64_t* pairs, uint64_t* mismatches,
                          uint64_t* first_bad_pair);

/* ------------------------------------------------------------------------------------------------
 * Trajectory evaluation / export.  Host side, double precision.
 * ------------------------------------------------------------------------------------------------ */
/* ATE: RMSE of |gt_i - (s R est_i + t)| after the optimal rigid (with_scale: similarity) alignment (Horn). xyz: [n][3] */
int dvo_eval_ate(int n, const float* est_xyz, const float* gt_xyz, int with_scale, double* rmse, double R_out[9],
                 double t_out[3], double* scale_out);
/* RPE over `delta` frames: RMSE of the translational part [m] and of the rotation angle [rad]; poses [n][16] row major */
int dvo_eval_rpe(int n, const float* est_T, const float* gt_T, int delta, double* trans_rmse, double* rot_rmse);
/* the correct rigid inverse (Convert::inversePose, src/core/convert.cpp:31-39, is wrong in the reference) */
int dvo_pose_inverse(const float T[16], float out[16]);
/* "timestamp tx ty tz qx qy qz qw" per line; timestamps may be NULL (frame index is written) */
int dvo_traj_write_tum(const char* path, int n, const double* timestamps, const float* T);

#ifdef __cplusplus
}
#endif
#endif
