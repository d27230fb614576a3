// engine.h -- the device-side engine behind the C ABI: owns the stream, the packed weights, the NHWC
// activation buffers and the per-batch-size hipGraphs.  Replaces TrtRetinaFaceNet (trtretinafacenet.cpp:
// allocateMemory :146-210, doInference :48-102, blob_by_name :104-114) + the decode/NMS half of
// RetinaFace::detect (RetinaFace.cpp:666-726).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/retinaface_amd.h"
#include "kernels.h"
#include "plan.h"

namespace rf {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
struct ArgError : std::runtime_error { using std::runtime_error::runtime_error; };

struct EngineOptions {
    int precision = RF_PRECISION_FP16;
    int net_h = 0, net_w = 0;
    int max_batch = 8;
    int device = -1;
    int max_candidates = 4096;
    int max_detections = 256;
    bool use_graph = true;
    int lanes = 3;                   // launches in flight (each lane has its own stream + buffers + graphs)
    int coalesce = 0;                // enqueued batches merged into one launch (max_batch * coalesce images; 0 = about 256 images of 448 x 448 per launch): the kernels are
                                     // persistent and pipeline tile t+1's loads under tile t's compute, which pays off once
                                     // a launch holds several tiles per resident workgroup (measured: 4 -> 16 = +8 %)
    bool keep_outputs = false;
    int copy_threads = 0;            // threads (caller's included) that stage host frames into pinned memory; 0 = min(8, cores / 4): measured best on a 256-thread host, 46 GB/s; more threads contend
    bool resize_bilinear = false;    // oversize frames: false = area average (the NPP build), true = cv::resize bilinear (the build without NPP)
    bool plan_cache = true;          // read / write <model_dir>/<stem>.<precision>.rfplan (packed weight image, weights.h)
    std::string plan_cache_path;     // explicit cache file instead (tests)
    std::vector<int> devices;        // more than one entry: one engine per device, batches sharded by image (multi.cpp)
    std::string model_stem = "mnet-deconv-0517";
};

struct ActInfo { void *ptr; int h, w, c; std::vector<float> scale; };      // scale: empty, or one per channel (int8)

struct OpInfo {
    std::string name;            // reference layer name(s) the launch covers
    std::string kernel;          // which kernel instance runs it, e.g. "dwpw<128,128,s1>" (joins profiles/*.json)
    double alg_elems_in = 0;     // per image: layer-wise input elements (SURVEY.md 8d "E")
    double alg_elems_out = 0;    // per image: layer-wise output elements
    double alg_u8_in = 0;        // per image: bytes read as u8 (the frame), not scaled by the element size
    double macs = 0;             // per image
    // per image: what the launch has to move through HBM at the very least GIVEN its fusion -- every tensor it reads from HBM once,
    // every tensor it writes once (weights: < 1 MB per launch, ignored).  `useful` HBM fraction = these bytes / time / peak.
    double hbm_elems_in = 0, hbm_elems_out = 0;
    std::function<void(hipStream_t, int)> launch;   // (stream, n_images)
};

// the constructor's `network` presets (RetinaFace.cpp:209-271): anchor ratios of a preset (empty = no anchors), and the base
// anchors of FPN level 0 / 1 / 2 (strides 32 / 16 / 8) for those ratios, 2 per ratio
bool network_preset(const std::string &network, std::vector<float> *ratios);
void preset_base_anchors(const std::vector<float> &ratios, int level, float out[][4]);

// host-only test hook: runs the host half of engine start-up (plan cache or model -> packed image); 1 = served from the cache
int plan_cache_probe(const std::string &model_dir, const EngineOptions &opt, size_t *arena_bytes);

class Engine {
public:
    // opt.devices.size() > 1 gives the image-sharding multi-device engine (multi.cpp), otherwise one single-device engine
    static std::unique_ptr<Engine> create(const std::string &model_dir, const std::string &network, float nms,
                                          const EngineOptions &opt);
    static std::unique_ptr<Engine> create_single(const std::string &model_dir, const std::string &network, float nms,
                                                 const EngineOptions &opt);
    virtual ~Engine() {}

    // frames on host / device; synchronous
    virtual void detect(const uint8_t *const *frames, const int *rows, const int *cols, const int *steps, int n,
                        bool on_device, float threshold, rf_face *out, int cap_per_image, int *counts,
                        bool *truncated) = 0;
    // asynchronous: frames on host (staged through pinned memory before the call returns, unless the caller registered
    // them with host_register) or on the device
    virtual int enqueue(const void *const *frames, const int *rows, const int *cols, const int *steps, int n, bool on_device,
                        float threshold) = 0;
    virtual void wait(int ticket, rf_face *out, int cap_per_image, int *counts, bool *truncated) = 0;
    virtual int num_slots() const = 0;
    // pinned caller memory: register pins the range (hipHostRegisterPortable) and records it; adopt / forget only record /
    // drop a range another engine of the same handle pinned (multi-device handles pin once)
    virtual void host_register(const void *ptr, size_t bytes) = 0;
    virtual void host_unregister(const void *ptr) = 0;
    virtual void host_adopt(const void *ptr, size_t bytes) = 0;
    virtual void host_forget(const void *ptr) = 0;
    // drop what the engine remembers about where device frame pointers live (callers that free / re-home frame buffers)
    virtual void invalidate_residency() = 0;
    // device frames that arrived from another device since the handle was built, and the peer copies that carried them
    virtual void scatter_stats(long long *frames, long long *copies) const = 0;

    virtual int last_anchor_indices(int image, int32_t *out, int cap) const = 0;
    virtual int last_candidate_counts(int *counts, int n) const = 0;
    virtual void last_timings(float *pre, float *infer, float *post, float *total) const = 0;
    virtual long get_output(const std::string &blob, int image, float *dst, size_t cap) = 0;
    virtual long debug_activation(const std::string &blob, int image, float *dst, size_t cap, int dims[3]) = 0;
    virtual int profile(const void *const *d_frames, int n, int iters, int cap, const char **names, const char **kernels,
                        float *avg_ms, double *alg_bytes, double *macs) = 0;
    // compulsory HBM bytes of every launch for n images (OpInfo::hbm_elems_*), in launch order; returns the number of launches
    virtual int compulsory_bytes(int n, int cap, double *bytes) = 0;

    int net_h() const { return net_h_; }
    int net_w() const { return net_w_; }
    int max_batch() const { return opt_.max_batch; }
    virtual int num_devices() const { return 1; }

protected:
    EngineOptions opt_;
    int net_h_ = 0, net_w_ = 0;
    float nms_threshold_ = 0.4f;
};

}  // namespace rf
