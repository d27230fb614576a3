// engine.cpp -- see engine.h.
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <thread>

#include "copier.h"
#include "knobs.h"
#include "pack.h"
#include "weights.h"

namespace rf {

#define RF_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            throw HipError(std::string(#expr) + " failed: " + hipGetErrorString(e_));                  \
    } while (0)

namespace {

// RF_HOST_TRACE=1 (probe): where the HOST time of a call goes -- per-stage wall-clock sums, printed when the engine is destroyed
struct HostTrace {
    bool on = knob(K_HOST_TRACE) != 0;
    double sum[8] = {};
    long n[8] = {};
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void add(int k, double t0) { if (on) { sum[k] += now() - t0; n[k]++; } }
    void report(const char *tag) const {
        if (!on) return;
        static const char *names[8] = {"detect() total", "submit: checks + table fill", "launch: table H2D enqueue", "launch: graph / kernel launches",
                                       "launch: event record", "harvest: event synchronize", "harvest: copy records", "wait: copy out"};
        for (int k = 0; k < 8; k++)
            if (n[k]) fprintf(stderr, "[rf host trace %s] %-32s %9.2f us avg over %ld\n", tag, names[k], sum[k] / n[k], n[k]);
    }
};

// The HIP current device is per host thread and defaults to 0: every entry point binds the calling thread to the engine's
// device for the duration of the call and puts the caller's device back (rf_options.device may differ from it, and the
// caller may use another thread per call -- the header only promises one caller thread AT A TIME).
class DeviceGuard {
public:
    explicit DeviceGuard(int dev) : dev_(dev) {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = dev;
        if (prev_ != dev_) (void)hipSetDevice(dev_);
        bind_launch_device(dev_);
    }
    ~DeviceGuard() {
        if (prev_ != dev_) { (void)hipSetDevice(prev_); bind_launch_device(prev_); }
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
private:
    int dev_, prev_ = 0;
};

// one base anchor of generate_anchors(base_size 16, ratios {.., ratio, ..}, scales {.., scale, ..}) -- RetinaFace.cpp:9-103, with
// the reference's float / double rounding points (`0.5 * (w - 1)` is double arithmetic stored to float; sqrt / round on floats)
void base_anchor(int scale, float ratio, float out[4]) {
    float w = 15.f - 0.f + 1, h = 15.f - 0.f + 1;                                   // _whctrs of (0, 0, 15, 15)
    float xc = 0.f + 0.5 * (w - 1), yc = 0.f + 0.5 * (h - 1);
    float size = w * h, sc = size / ratio;                                          // _ratio_enum, :34-51
    float w2 = std::round(std::sqrt(sc)), h2 = std::round(w2 * ratio);
    float rx1 = xc - 0.5 * (w2 - 1), ry1 = yc - 0.5 * (h2 - 1), rx2 = xc + 0.5 * (w2 - 1), ry2 = yc + 0.5 * (h2 - 1);
    w = rx2 - rx1 + 1; h = ry2 - ry1 + 1;                                           // _scale_enum, :53-68
    xc = rx1 + 0.5 * (w - 1); yc = ry1 + 0.5 * (h - 1);
    w = w * scale; h = h * scale;
    out[0] = xc - 0.5 * (w - 1); out[1] = yc - 0.5 * (h - 1);
    out[2] = xc + 0.5 * (w - 1); out[3] = yc + 0.5 * (h - 1);
}

}  // namespace

// The reference constructor's `network` presets (RetinaFace.cpp:209-271).  Only fmc == 3 presets get an anchor configuration
// (strides 32 / 16 / 8, scales {32,16} {8,4} {2,1}, base 16) and only "net3" / "net3a" give it ratios: "ssh" / "vgg" set pixel
// means that are never applied and leave `_ratio` empty, net4 / net5 / net5a / net6 print "please reconfig anchor_cfg", any
// other name prints "network setting error" -- all of those construct fine and then find no faces, because post-processing
// loops over zero anchors.  Returns false for names the reference does not know (its behaviour is the same as for "ssh").
bool network_preset(const std::string &network, std::vector<float> *ratios) {
    ratios->clear();
    if (network == "net3") { *ratios = {1.0f}; return true; }
    if (network == "net3a") { *ratios = {1.0f, 1.5f}; return true; }
    for (const char *k : {"ssh", "vgg", "net4", "net5", "net5a", "net6"})
        if (network == k) return true;
    return false;
}

void preset_base_anchors(const std::vector<float> &ratios, int level, float out[][4]) {
    static const int scales[3][2] = {{32, 16}, {8, 4}, {2, 1}};
    int a = 0;
    for (float r : ratios)                              // generate_anchors: ratios outer, scales inner (:86-91)
        for (int s = 0; s < 2; s++) base_anchor(scales[level][s], r, out[a++]);
}

namespace {

// One lane = one independent copy of everything a batch touches while in flight: stream(s), activation buffers,
// candidate buffers, pinned host descriptor / result blocks and the captured hipGraphs.  Batches on different
// lanes overlap on the GPU (most kernels of a batch-8 pass fill only a fraction of the 256 CUs); weights are shared.
constexpr int kMaxCoalesce = 256;       // enqueues merged into one launch: a batch-1 caller still fills a 256-image super-batch

template <typename T>
class EngineImpl final : public Engine, private WeightPack<T> {
    typedef WeightPack<T> WP;
    using typename WP::GemmW;
    using typename WP::DwW;
    using WP::kNone;
    using WP::arena_; using WP::c0_w_; using WP::c0_b_; using WP::c0_hi_; using WP::stem_dw_; using WP::stem2_dw_; using WP::stem_pw_;
    using WP::stem2_pw_; using WP::aggr_a_lat_; using WP::aggr_a_up_; using WP::act_scale_; using WP::dw_w_; using WP::pw_w_; using WP::lat_w_;
    using WP::aggr_w_; using WP::ssh_w_; using WP::head_a_; using WP::mult_ptr;
public:
    // `plan` may be a skeleton (no weights): everything weight-related comes packed in `pack` (built from the model or read from
    // the plan cache, weights.h)
    EngineImpl(const Plan &plan, WeightPack<T> &&pack, float nms, const EngineOptions &opt, const std::vector<float> &ratios)
        : WeightPack<T>(std::move(pack)) {
        opt_ = opt;
        nms_threshold_ = nms;
        ratios_ = ratios;
        na_ = 2 * (int)ratios.size();              // anchors per cell: 2 scales x the preset's ratios; 0 = nothing to decode
        net_h_ = opt.net_h > 0 ? opt.net_h : plan.net_h;
        net_w_ = opt.net_w > 0 ? opt.net_w : plan.net_w;
        if (net_h_ <= 0 || net_w_ <= 0 || net_h_ % 32 || net_w_ % 32)
            throw ArgError("network input size must be a positive multiple of 32 (got " + std::to_string(net_h_) + "x" +
                           std::to_string(net_w_) + ")");
        if (opt_.max_batch < 1) throw ArgError("max_batch must be >= 1");
        int mc = opt_.max_candidates;
        if (mc < 64 || mc > 4096 || (mc & (mc - 1))) throw ArgError("max_candidates must be a power of two in [64, 4096]");
        if (opt_.max_detections < 1 || opt_.max_detections > 4096) throw ArgError("max_detections must be in [1, 4096]");
        if (opt_.lanes < 1 || opt_.lanes > 16) throw ArgError("lanes must be in [1, 16]");
        // default super-batch: ~256 images of 448 x 448 worth of pixels per launch whatever the batch size (measured: 8 x 32 and 32 x 8 beat 8 x 16 / 32 x 4 by
        // 2-4 %; every persistent kernel then has several tiles per resident workgroup)
        if (opt_.coalesce == 0)
            opt_.coalesce = (int)std::max<long>(1, std::min<long>(kMaxCoalesce, 256L * 448 * 448 / ((long)std::max(opt_.max_batch, 1) * net_h_ * net_w_)));
        if (opt_.coalesce < 1 || opt_.coalesce > kMaxCoalesce) throw ArgError("coalesce must be in [1, 256]");
        if (opt_.copy_threads < 0 || opt_.copy_threads > 64) throw ArgError("copy_threads must be in [0, 64]");
        cap_images_ = opt_.max_batch * opt_.coalesce;
        // every lane owns activation buffers for a full super-batch (~13 MB per 448 x 448 image in fp16): refuse sizes that can only
        // end in an out-of-memory error three allocations later
        if ((long)cap_images_ * net_h_ * net_w_ > 2048L * 448 * 448)
            throw ArgError("max_batch x coalesce = " + std::to_string(cap_images_) + " images per launch: more than 2048 images of 448 x 448 worth of pixels");
        tickets_.resize(4 * opt_.lanes * opt_.coalesce + 8);
        if (opt_.device >= 0) device_ = opt_.device;
        else RF_HIP(hipGetDevice(&device_));
        int ndev = 0;
        RF_HIP(hipGetDeviceCount(&ndev));
        if (device_ < 0 || device_ >= ndev) throw ArgError("device ordinal " + std::to_string(device_) + " out of range (" + std::to_string(ndev) + " devices)");
        if (device_ >= kMaxDevices) throw ArgError("device ordinal " + std::to_string(device_) + ": at most " + std::to_string(kMaxDevices) + " devices per process");
        // device frames are checked for residency whenever another device exists they could live on (RF_FORCE_SCATTER: test knob,
        // treats every device frame as foreign so the scatter path runs on a one-GPU box)
        force_scatter_ = knob(K_FORCE_SCATTER) != 0;
        scatter_per_frame_ = knob(K_SCATTER_PER_FRAME) != 0;
        if (knob(K_SYNC_SPLIT) == 0) stage_pieces_ = 1;
        else if (const char *e = getenv("RF_SYNC_PIECES")) { const int v = atoi(e); if (v >= 1 && v <= 64) stage_pieces_ = v; }      // measurement knob of the A/B
        head_start_max_ = knob(K_HEAD_START) ? opt_.max_batch : 0;      // launches of up to max_batch images = synchronous calls and un-coalesced tickets
        copy_streams_ = knob(K_COPY_STREAMS) > 1 ? 2 : 1;       // probe knob RF_COPY_STREAMS (tools/probes/host_rate.py)
        check_residency_ = ndev > 1 || force_scatter_;
        DeviceGuard guard(device_);                  // the caller's current device is put back when construction ends
        if (sizeof(T) == 1) {
            const int chk = cvt_pk_u8_selfcheck();
            if (chk < 0) throw HipError("int8: the v_cvt_pk_u8_f32 rounding probe could not run on this device (allocation, launch or copy failed)");
            if (chk > 0) throw Unsupported("int8: v_cvt_pk_u8_f32 on this device does not round to nearest even / saturate as the requantising epilogues assume");
        }
        try { arena_.upload(); } catch (const std::exception &e) { throw HipError(e.what()); }
        // Lanes are built on first use: a caller that only makes synchronous calls of <= max_batch images keeps re-using lane 0 and
        // never pays for the other lanes' activation buffers (~13 MB per 448 x 448 image in fp16 x the super-batch size).  Lane 0
        // is built now; if its buffers do not fit the device (a small or shared GPU) the super-batch is halved until they do.
        plan_ = plan;
        for (auto *f : {&plan_.conv0, &plan_.lateral[0], &plan_.lateral[1], &plan_.lateral[2], &plan_.aggr[0], &plan_.aggr[1]}) { f->w.clear(); f->w.shrink_to_fit(); }
        lanes_.resize(opt_.lanes);
        for (;;) {
            try { ensure_lane(0); break; }
            catch (const HipError &) {
                if (opt_.coalesce == 1) throw;
                opt_.coalesce = std::max(1, opt_.coalesce / 2);
                cap_images_ = opt_.max_batch * opt_.coalesce;
            }
        }
        // RF_PREBUILD_LANES=1: build every lane now (allocation, first eager run and graph capture happen at create time instead of
        // in the middle of the asynchronous steady state: one latency spike per lane less for latency-sensitive callers).  A lane
        // that does not fit is dropped, as on first use.  num_slots() may therefore be smaller than lanes x coalesce: callers read it
        // after rf_create, and again if an enqueue ever fails with RF_ERR_HIP.
        if (knob(K_PREBUILD_LANES)) {
            for (int l = 1; l < (int)lanes_.size(); l++) {
                try { ensure_lane(l); }
                catch (const HipError &) { lanes_.resize(l); break; }
            }
        }
        const int hw = (int)std::thread::hardware_concurrency();
        const int helpers = opt_.copy_threads > 0 ? opt_.copy_threads - 1 : std::max(0, std::min(8, hw / 4) - 1);
        copier_.reset(new ParallelCopier(helpers, knob(K_NT_COPY) != 0));       // RF_NT_COPY=0 (probe knob): plain memcpy into the staging block
    }

    ~EngineImpl() override {
        DeviceGuard guard(device_);
        trace_.report(sizeof(T) == 1 ? "int8" : sizeof(T) == 2 ? "fp16" : "fp32");
        for (auto &r : registered_) if (r.owned) (void)hipHostUnregister((void *)r.base);
        for (auto &l : lanes_) free_lane(l);
        for (void *p : dev_allocs_) (void)hipFree(p);
        for (void *p : host_allocs_) (void)hipHostFree(p);
        arena_.release();
        for (auto &e : prof_ev_) (void)hipEventDestroy(e);
    }

    // ------------------------------------------------------------------------------------------ API
    void detect(const uint8_t *const *frames, const int *rows, const int *cols, const int *steps, int n, bool on_device,
                float threshold, rf_face *out, int cap_per_image, int *counts, bool *truncated) override {
        if (n < 0 || (n > 0 && (!frames || !rows || !cols || !counts))) throw ArgError("null argument");
        if (cap_per_image < 0 || (cap_per_image > 0 && !out)) throw ArgError("out is null");
        DeviceGuard guard(device_);
        const double t_detect = trace_.on ? HostTrace::now() : 0.0;
        struct AtExit { HostTrace &t; double t0; ~AtExit() { t.add(0, t0); } } at_exit{trace_, t_detect};
        *truncated = false;
        // every frame is checked before the first chunk is launched: a bad frame in a later chunk must not leave earlier
        // chunks in flight with nobody waiting for them
        for (int i = 0; i < n; i++) check_frame(frames[i], rows[i], cols[i], steps ? steps[i] : cols[i] * 3);
        std::vector<int> all_cand;
        std::vector<std::vector<int32_t>> all_anchor;
        // a synchronous call keeps up to `lanes` chunks of max_batch images in flight and collects them in order
        std::vector<std::pair<int, int>> inflight;      // (ticket, base)
        auto collect = [&]() {
            int ticket = inflight.front().first, base = inflight.front().second;
            inflight.erase(inflight.begin());
            bool tr = false;
            wait_impl(ticket, out ? out + (size_t)base * cap_per_image : nullptr, cap_per_image, counts + base, &tr);
            *truncated = *truncated || tr;
            all_cand.insert(all_cand.end(), last_cand_counts_.begin(), last_cand_counts_.end());
            for (int i = 0; i < last_n_; i++) all_anchor.push_back(std::move(last_anchor_[i]));
        };
        const bool timed = !opt_.use_graph;             // the eager engine measures the pre / infer / post split
        // A call of MORE than max_batch images (round 5): its chunks are not launched one by one (eight dependent 32-image launch sequences for
        // a 256-image call: each latency-bound, 1.07 ms) but join super-batches like enqueued tickets do -- one launch sequence per
        // max_batch x coalesce images (0.8 ms for the same call); the last, partial one is launched by the first wait.
        const bool coalesce_chunks = !timed && n > opt_.max_batch && opt_.coalesce > 1;
        const int max_inflight = coalesce_chunks ? (int)lanes_.size() * opt_.coalesce : (int)lanes_.size();
        try {
            for (int base = 0; base < n; base += opt_.max_batch) {
                int m = std::min(opt_.max_batch, n - base);
                std::vector<int> st(m);
                for (int i = 0; i < m; i++) st[i] = steps ? steps[base + i] : cols[base + i] * 3;
                if ((int)inflight.size() >= max_inflight || (timed && !inflight.empty())) collect();      // (>=: pick_lane() may have dropped lanes that did not fit)
                int ticket = submit(frames + base, rows + base, cols + base, st.data(), m, on_device, threshold, !coalesce_chunks);
                inflight.emplace_back(ticket, base);
            }
            while (!inflight.empty()) collect();
        } catch (...) {
            // hand every ticket of this call back to the pool (results dropped) before the error leaves
            for (auto &tb : inflight) {
                try { bool tr = false; wait_impl(tb.first, nullptr, 0, nullptr, &tr); } catch (...) { tickets_[tb.first].state = Ticket::FREE; }
            }
            throw;
        }
        if (n > opt_.max_batch) last_first_image_ = -1000000;   // blob accessors are per-launch: not valid for chunked calls
        last_n_ = n;                         // the "most recent completed batch" of a chunked call is the whole call
        last_cand_counts_.swap(all_cand);
        last_anchor_.swap(all_anchor);
    }

    int enqueue(const void *const *frames, const int *rows, const int *cols, const int *steps, int n, bool on_device,
                float threshold) override {
        if (n < 1 || n > opt_.max_batch) throw ArgError("enqueue: n must be in [1, max_batch]");
        if (!frames || !rows || !cols) throw ArgError("null argument");
        DeviceGuard guard(device_);
        std::vector<int> st(n);
        for (int i = 0; i < n; i++) {
            st[i] = steps ? steps[i] : cols[i] * 3;
            check_frame((const uint8_t *)frames[i], rows[i], cols[i], st[i]);
        }
        return submit((const uint8_t *const *)frames, rows, cols, st.data(), n, on_device, threshold, false);
    }

    void wait(int ticket, rf_face *out, int cap_per_image, int *counts, bool *truncated) override {
        if (cap_per_image < 0 || (cap_per_image > 0 && !out)) throw ArgError("wait: out is null");
        DeviceGuard guard(device_);
        wait_impl(ticket, out, cap_per_image, counts, truncated);
    }

    void host_register(const void *ptr, size_t bytes) override {
        if (!ptr || !bytes) throw ArgError("host_register: null / empty range");
        DeviceGuard guard(device_);
        RF_HIP(hipHostRegister((void *)ptr, bytes, hipHostRegisterPortable));
        registered_.push_back(HostRange{(uintptr_t)ptr, bytes, true});
    }
    void host_adopt(const void *ptr, size_t bytes) override {
        if (!ptr || !bytes) throw ArgError("host_register: null / empty range");
        registered_.push_back(HostRange{(uintptr_t)ptr, bytes, false});
    }
    void host_unregister(const void *ptr) override { drop_range(ptr, true); }
    void host_forget(const void *ptr) override { drop_range(ptr, false); }
    void invalidate_residency() override { residency_.clear(); }
    void scatter_stats(long long *frames, long long *copies) const override { *frames += scattered_frames_; *copies += peer_copies_; }

    // tickets the caller may keep outstanding before it has to wait: every lane can hold a full super-batch
    int num_slots() const override { return (int)lanes_.size() * opt_.coalesce; }

    int last_anchor_indices(int image, int32_t *out, int cap) const override {
        if (image < 0 || image >= last_n_) throw ArgError("image index out of range");
        if (cap > 0 && !out) throw ArgError("null argument");
        int n = (int)last_anchor_[image].size();
        for (int i = 0; i < std::min(n, cap); i++) out[i] = last_anchor_[image][i];
        return n;
    }
    int last_candidate_counts(int *counts, int n) const override {
        if (n > 0 && !counts) throw ArgError("null argument");
        for (int i = 0; i < std::min(n, last_n_); i++) counts[i] = last_cand_counts_[i];
        return last_n_;
    }
    void last_timings(float *pre, float *infer, float *post, float *total) const override {
        if (pre) *pre = have_split_ ? t_pre_ : -1.f;
        if (infer) *infer = have_split_ ? t_infer_ : -1.f;
        if (post) *post = have_split_ ? t_post_ : -1.f;
        if (total) *total = have_split_ ? t_total_ : -1.f;
    }

    long get_output(const std::string &blob, int image, float *dst, size_t cap) override {
        if (!opt_.keep_outputs) throw ArgError("rf_get_output needs options.keep_outputs = 1");
        DeviceGuard guard(device_);
        static const char *kinds[3] = {"face_rpn_cls_prob_reshape_stride", "face_rpn_bbox_pred_stride",
                                       "face_rpn_landmark_pred_stride"};
        const int chans[3] = {2 * head_a_, 4 * head_a_, 10 * head_a_};
        Lane &l = lanes_[last_lane_];
        for (int si = 0; si < 3; si++)
            for (int k = 0; k < 3; k++) {
                if (blob != std::string(kinds[k]) + std::to_string(strides_[si])) continue;
                if (image < 0 || image >= last_n_ || last_first_image_ < 0) throw ArgError("image index out of range");
                size_t hw = (size_t)(net_h_ / strides_[si]) * (net_w_ / strides_[si]);
                size_t cnt = hw * chans[k];
                if (!dst) return (long)cnt;
                if (cap < cnt) throw ArgError("destination too small");
                RF_HIP(hipStreamSynchronize(l.stream));
                RF_HIP(hipMemcpy(dst, l.d_dump[si][k] + (size_t)(last_first_image_ + image) * cnt, cnt * sizeof(float),
                                 hipMemcpyDeviceToHost));
                return (long)cnt;
            }
        throw ArgError("unknown output blob '" + blob + "'");
    }

    long debug_activation(const std::string &blob, int image, float *dst, size_t cap, int dims[3]) override {
        DeviceGuard guard(device_);
        Lane &l = lanes_[last_lane_];
        if (blob == "input_canvas") {
            // the net-sized u8 BGR canvas an OVERSIZE frame was scaled onto (valid when the last launch had one)
            if (image < 0 || image >= last_n_ || last_first_image_ < 0) throw ArgError("image index out of range");
            const size_t cnt = (size_t)net_h_ * net_w_ * 3;
            if (dims) { dims[0] = net_h_; dims[1] = net_w_; dims[2] = 3; }
            if (!dst) return (long)cnt;
            if (cap < cnt) throw ArgError("destination too small");
            RF_HIP(hipStreamSynchronize(l.stream));
            std::vector<uint8_t> tmp(cnt);
            RF_HIP(hipMemcpy(tmp.data(), l.d_canvas + (size_t)(last_first_image_ + image) * cnt, cnt, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < cnt; i++) dst[i] = (float)tmp[i];
            return (long)cnt;
        }
        // "<blob>#raw": the stored values themselves (int8 engine: the quanta, not multiplied by the tensor's scale)
        bool raw = false;
        std::string key = blob;
        if (key.size() > 4 && key.compare(key.size() - 4, 4, "#raw") == 0) { raw = true; key.resize(key.size() - 4); }
        auto it = l.acts.find(key);
        if (it == l.acts.end()) throw ArgError("unknown activation '" + key + "'");
        const ActInfo &ai = it->second;
        if (image < 0 || image >= last_n_ || last_first_image_ < 0) throw ArgError("image index out of range");
        size_t cnt = (size_t)ai.h * ai.w * ai.c;
        if (dims) { dims[0] = ai.h; dims[1] = ai.w; dims[2] = ai.c; }
        if (!dst) return (long)cnt;
        if (cap < cnt) throw ArgError("destination too small");
        RF_HIP(hipStreamSynchronize(l.stream));
        std::vector<T> tmp(cnt);
        RF_HIP(hipMemcpy(tmp.data(), (const T *)ai.ptr + (size_t)(last_first_image_ + image) * cnt, cnt * sizeof(T),
                         hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; i++) dst[i] = Cast<T>::to(tmp[i]) * (raw || ai.scale.empty() ? 1.f : ai.scale[i % ai.c]);      // NHWC: channel = i % c
        return (long)cnt;
    }

    int profile(const void *const *d_frames, int n, int iters, int cap, const char **names, const char **kernels,
                float *avg_ms, double *alg_bytes, double *macs) override {
        if (n < 1 || n > cap_images_ || iters < 1) throw ArgError("profile: bad n / iters");
        DeviceGuard guard(device_);
        launch_pending();
        Lane &l = lanes_[0];
        harvest(l);
        RF_HIP(hipStreamSynchronize(l.stream));
        const int mb = cap_images_;
        for (int i = 0; i < n; i++) l.h_frames[mb + i] = FrameDesc{(const uint8_t *)d_frames[i], net_h_, net_w_, net_w_ * 3, 0};
        *l.h_params = RunParams{0.5f, nms_threshold_, n, 0};
        RF_HIP(hipMemcpyAsync(l.d_frames, l.h_frames, l.table_bytes, hipMemcpyHostToDevice, l.stream));
        size_t nops = l.ops.size();
        while (prof_ev_.size() < 2 * nops) { hipEvent_t e; RF_HIP(hipEventCreate(&e)); prof_ev_.push_back(e); }
        // Every launch is timed on the engine's own stream as `kReps` back-to-back repetitions between one HIP event
        // pair (a single launch between two events would mostly measure the ~6 us event/launch overhead); the result is
        // the average duration per launch including its launch boundary.  A full pass runs first in launch order so
        // every kernel sees the real producer's data.
        constexpr int kReps = 8;
        const size_t kh = nops - 2, kn = nops - 1;     // heads, NMS: producer / consumer of the candidate counters
        std::vector<double> sum(nops, 0.0);
        for (int it = -1; it < iters; it++) {          // one untimed warm-up pass
            for (size_t k = 0; k < kh; k++) {
                l.ops[k].launch(l.stream, n);           // real predecessor state for the ops that follow
                RF_HIP(hipEventRecord(prof_ev_[2 * k], l.stream));
                for (int r = 0; r < kReps; r++) l.ops[k].launch(l.stream, n);
                RF_HIP(hipEventRecord(prof_ev_[2 * k + 1], l.stream));
            }
            // NMS re-arms the counters the heads fill, so the two are repeated as a pair; the heads are also repeated
            // alone (candidates pile up, one NMS drains them afterwards) and NMS = pair - heads.
            RF_HIP(hipEventRecord(prof_ev_[2 * kh], l.stream));
            for (int r = 0; r < kReps; r++) { l.ops[kh].launch(l.stream, n); l.ops[kn].launch(l.stream, n); }
            RF_HIP(hipEventRecord(prof_ev_[2 * kh + 1], l.stream));
            RF_HIP(hipEventRecord(prof_ev_[2 * kn], l.stream));
            for (int r = 0; r < kReps; r++) l.ops[kh].launch(l.stream, n);
            RF_HIP(hipEventRecord(prof_ev_[2 * kn + 1], l.stream));
            l.ops[kn].launch(l.stream, n);
            RF_HIP(hipStreamSynchronize(l.stream));
            RF_HIP(hipGetLastError());
            if (it < 0) continue;
            for (size_t k = 0; k < nops; k++) {
                float ms = 0;
                RF_HIP(hipEventElapsedTime(&ms, prof_ev_[2 * k], prof_ev_[2 * k + 1]));
                if (k < kh) sum[k] += ms / kReps;
                else if (k == kh) sum[kn] += ms / kReps;      // pair, fixed up below
                else sum[kh] += ms / kReps;                   // heads alone
            }
        }
        sum[kn] = std::max(sum[kn] - sum[kh], 0.0);
        for (size_t k = 0; k < nops && (int)k < cap; k++) {
            if (names) names[k] = l.ops[k].name.c_str();
            if (kernels) kernels[k] = l.ops[k].kernel.c_str();
            if (avg_ms) avg_ms[k] = (float)(sum[k] / iters);
            if (alg_bytes) alg_bytes[k] = n * (l.ops[k].alg_u8_in + sizeof(T) * (l.ops[k].alg_elems_in + l.ops[k].alg_elems_out));
            if (macs) macs[k] = n * l.ops[k].macs;
        }
        return (int)nops;
    }

    int compulsory_bytes(int n, int cap, double *bytes) override {
        if (n < 0 || (cap > 0 && !bytes)) throw ArgError("null argument");
        const Lane &l = lanes_[0];
        for (size_t k = 0; k < l.ops.size() && (int)k < cap; k++)
            bytes[k] = n * (l.ops[k].alg_u8_in + sizeof(T) * (l.ops[k].hbm_elems_in + l.ops[k].hbm_elems_out));
        return (int)l.ops.size();
    }

private:
    struct Lane {
        bool built = false;
        std::vector<void *> dev_allocs, host_allocs;      // what build_lane allocated for this lane
        hipStream_t stream = nullptr;
        int cus = 0;                          // CUs the stream may use (0 = all): persistent grids of this lane's launches are sized by it
        unsigned long long launch_seq = 0;    // order of this lane's last launch among all launches of the engine (pick_lane)
        // host-frame uploads from the pinned staging block alternate between the lane's stream and a second one (a second SDMA
        // engine); the launch waits for both.  RF_COPY_STREAMS=1 (probe knob) switches the second stream off
        hipStream_t copy2 = nullptr;
        hipEvent_t copy2_done = nullptr;
        bool copy2_used = false;
        unsigned uploads = 0;
        hipEvent_t time_ev[4] = {nullptr, nullptr, nullptr, nullptr};
        hipEvent_t done = nullptr;
        std::map<int, hipGraphExec_t> graphs;
        std::set<int> warmed;
        std::map<std::string, ActInfo> acts;
        std::vector<OpInfo> ops;              // launch order
        size_t first_post = 0;                // index of the first post-processing launch (heads)
        // pinned host staging of the per-launch table: [0,mb) source frames, [mb,2mb) what the stem reads, then RunParams;
        // ONE small H2D copy per launch puts it in HBM (a launch covers up to max_batch*coalesce images, so the copy is
        // amortised; reading it from the kernels over PCIe instead cost every workgroup a PCIe round trip)
        FrameDesc *h_frames = nullptr;
        RunParams *h_params = nullptr;
        FrameDesc *d_frames = nullptr;
        size_t table_bytes = 0;
        int *h_counts = nullptr;              // [2*max_batch]: kept counts, candidate counts
        Candidate *h_out = nullptr;           // [max_batch*max_det]
        // device
        RunParams *d_params = nullptr;
        int *d_cand_count = nullptr;
        Candidate *d_cand = nullptr;
        uint8_t *d_canvas = nullptr;
        // host frames: pinned staging block + its device twin (allocated on first use); a super-batch's frames are packed into
        // it back to back and cross PCIe as one DMA per enqueue
        uint8_t *h_stage = nullptr, *d_stage = nullptr;
        size_t stage_cap = 0, stage_used = 0;
        float *d_dump[3][3] = {};
        bool busy = false;                    // a launched super-batch whose results have not been harvested yet
        int n_images = 0;                     // images of the super-batch being assembled / in flight on this lane
        float threshold = 0.f;
        bool need_resize = false, timed = false;
        std::vector<int> tickets;             // tickets riding on that super-batch
        std::vector<char> empty;              // per image: img.empty() (count forced to 0)
    };

    // One enqueue() / detect chunk.  Several tickets are coalesced into one launch ("super-batch") of up to
    // max_batch * coalesce images: the launch count per image -- what bounds throughput at batch 8 -- drops accordingly.
    struct Ticket {
        enum State { FREE, PENDING, LAUNCHED, DONE };
        State state = FREE;
        int lane = 0, first_image = 0, n = 0;
        bool timed = false;
        std::vector<int> kept, ncand, first_record;
        std::vector<Candidate> records;
    };

    // ------------------------------------------------------------------------------------------ build
    template <typename U> U *dalloc(size_t count) {
        void *p = nullptr;
        RF_HIP(hipMalloc(&p, std::max<size_t>(count * sizeof(U), 256)));
        (building_ ? building_->dev_allocs : dev_allocs_).push_back(p);
        return (U *)p;
    }
    template <typename U> U *halloc(size_t count) {
        void *p = nullptr;
        RF_HIP(hipHostMalloc(&p, std::max<size_t>(count * sizeof(U), 256), hipHostMallocDefault));
        (building_ ? building_->host_allocs : host_allocs_).push_back(p);
        memset(p, 0, std::max<size_t>(count * sizeof(U), 256));
        return (U *)p;
    }

    void free_lane(Lane &l) {
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        for (auto &kv : l.graphs) (void)hipGraphExecDestroy(kv.second);
        for (hipEvent_t &e : l.time_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.copy2_done) (void)hipEventDestroy(l.copy2_done);
        if (l.copy2) { (void)hipStreamSynchronize(l.copy2); (void)hipStreamDestroy(l.copy2); }
        if (l.d_stage) (void)hipFree(l.d_stage);
        if (l.h_stage) (void)hipHostFree(l.h_stage);
        if (l.stream) (void)hipStreamDestroy(l.stream);
        for (void *p : l.dev_allocs) (void)hipFree(p);
        for (void *p : l.host_allocs) (void)hipHostFree(p);
        l = Lane();
    }

    // build lane k if it has not been used yet; a failed build releases what it had allocated and rethrows
    void ensure_lane(int k) {
        Lane &l = lanes_[k];
        if (l.built) return;
        building_ = &l;
        try { build_lane(l, plan_); }
        catch (...) { building_ = nullptr; (void)hipGetLastError(); free_lane(l); throw; }
        building_ = nullptr;
        l.built = true;
    }

    // Which lane opens the next super-batch: an idle lane that exists (a synchronous caller therefore stays on lane 0), else a
    // lane that has not been built yet (if its buffers do not fit the device any more the engine simply runs with fewer lanes),
    // else the oldest busy one (its results are harvested first).
    int pick_lane() {
        const int L = (int)lanes_.size();
        for (int k = 0; k < L; k++) {
            const int l = (next_lane_ + k) % L;
            if (lanes_[l].built && !lanes_[l].busy) return l;
        }
        for (int l = 0; l < L; l++)
            if (!lanes_[l].built) {
                try { ensure_lane(l); return l; }
                catch (const HipError &) { if (l == 0) throw; lanes_.resize(l); next_lane_ %= l; break; }      // lanes l.. were never built: drop them
            }
        // every lane is busy: the one launched longest ago (its results are harvested first; next_lane_ is not it once idle lanes
        // have been preferred out of order)
        int oldest = 0;
        for (int l = 1; l < (int)lanes_.size(); l++)
            if (lanes_[l].launch_seq < lanes_[oldest].launch_seq) oldest = l;
        return oldest;
    }

    static constexpr bool kInt8 = sizeof(T) == 1;
    typedef typename DwWeightT<T>::type DWT;

    void build_lane(Lane &L, const Plan &plan) {
        const int mb = cap_images_;           // images per launch: max_batch * coalesce
        const int H = net_h_, W = net_w_;
        const double P = (double)H * W;
        // RF_CU_SPLIT=1 (probe knob): lane l runs on one half of every XCD's CUs (the mask's low / high 128 bits: measured with
        // tools/probes/cu_mask.cpp -- 128 distinct CUs, all 8 XCDs, honoured by graph replays), halves alternating by lane, so that two lanes'
        // DIFFERENT kernels (a VALU-bound stem beside an HBM-bound block) share the chip spatially instead of queueing for each other's slots
        const int lane_index = (int)(&L - lanes_.data());
        if (knob(K_CU_SPLIT) == 1 && lanes_.size() > 1) {
            uint32_t mask[8];
            for (int i = 0; i < 8; i++) mask[i] = ((lane_index & 1) == 0) == (i < 4) ? 0xffffffffu : 0u;
            RF_HIP(hipExtStreamCreateWithCUMask(&L.stream, 8, mask));
            L.cus = 128;
        } else {
            RF_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        }
        for (auto &e : L.time_ev) RF_HIP(hipEventCreate(&e));
        RF_HIP(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        if (copy_streams_ > 1) {
            RF_HIP(hipStreamCreateWithFlags(&L.copy2, hipStreamNonBlocking));
            RF_HIP(hipEventCreateWithFlags(&L.copy2_done, hipEventDisableTiming));
        }

        L.table_bytes = 2 * mb * sizeof(FrameDesc) + sizeof(RunParams);
        unsigned char *htab = halloc<unsigned char>(L.table_bytes);
        unsigned char *dtab = dalloc<unsigned char>(L.table_bytes);
        L.h_frames = (FrameDesc *)htab;
        L.h_params = (RunParams *)(htab + 2 * mb * sizeof(FrameDesc));
        L.d_frames = (FrameDesc *)dtab;
        L.d_params = (RunParams *)(dtab + 2 * mb * sizeof(FrameDesc));
        const size_t hdr = ((2 * mb * sizeof(int) + 255) / 256) * 256;
        unsigned char *res = halloc<unsigned char>(hdr + (size_t)mb * opt_.max_detections * sizeof(Candidate));
        L.h_counts = (int *)res;
        L.h_out = (Candidate *)(res + hdr);
        L.d_cand_count = dalloc<int>(mb);
        RF_HIP(hipMemset(L.d_cand_count, 0, mb * sizeof(int)));
        L.d_cand = dalloc<Candidate>((size_t)mb * opt_.max_candidates);
        L.d_canvas = dalloc<uint8_t>((size_t)mb * H * W * 3);

        auto act = [&](const std::string &name, int h, int w, int c) {
            T *p = dalloc<T>((size_t)mb * h * w * c);
            auto sc = act_scale_.find(name);
            L.acts[name] = ActInfo{p, h, w, c, sc == act_scale_.end() ? std::vector<float>() : sc->second};
            return p;
        };
        // activations: one buffer per reference blob that survives fusion (288 GB of HBM: nothing is recycled)
        int h = H / 2, w = W / 2;
        T *cur = nullptr;
        size_t first_block = 0;
        int c = 8;
        bool fused2 = false;
        if constexpr (std::is_same<T, half_t>::value) fused2 = stem2_variant() != 0;
        if (fused2) {
            if constexpr (std::is_same<T, half_t>::value) {
                // fp16 engine: preprocess + conv0 + blocks 0 and 1 (conv1..conv4) are ONE launch (stem2_kernel): the 224^2 x 16 map of
                // the net never exists in HBM
                const auto &b0 = plan.blocks[0], &b1 = plan.blocks[1];
                const int h4 = H / 4, w4 = W / 4;
                T *out = act(b1.pw.out_blob, h4, w4, b1.pw.cout);
                Stem2Params sp;
                sp.frames = L.d_frames + mb; sp.out = out;
                sp.w0 = arena_.template ptr<half_t>(c0_hi_); sp.b0 = arena_.template ptr<float>(WP::c0_b_mma_);
                sp.w0_raw = arena_.template ptr<half_t>(WP::c0_raw_);
                sp.c0_tab = arena_.template ptr<uint32_t>(WP::stem2_c0tab_); sp.dw1_mma4 = arena_.template ptr<uint32_t>(WP::stem2_dw4_);
                sp.dw0_w = arena_.template ptr<float>(stem_dw_.w); sp.dw0_b = arena_.template ptr<float>(stem_dw_.b);
                sp.pw0_w = arena_.template ptr<half_t>(stem_pw_.w); sp.pw0_b = arena_.template ptr<float>(WP::stem2_c2_b_);
                sp.c2_floor = arena_.template ptr<uint32_t>(WP::stem2_c2_floor_); sp.c3_floor = arena_.template ptr<uint32_t>(WP::stem2_c3_floor_);
                sp.dw1_mma = arena_.template ptr<uint32_t>(stem2_dw_.mma); sp.dw1_b = arena_.template ptr<float>(stem2_dw_.b);
                sp.pw1_w = arena_.template ptr<half_t>(stem2_pw_.w); sp.pw1_b = arena_.template ptr<float>(stem2_pw_.b);
                sp.n = 0; sp.net_h = H; sp.net_w = W;
                OpInfo op;
                op.name = "pre+" + plan.conv0.name + "+" + b0.dw.name + "+" + b0.pw.name + "+" + b1.dw.name + "+" + b1.pw.name;
                op.kernel = "stem2";
                op.alg_u8_in = 3.0 * P;
                // layer-wise accounting: every covered layer's input + output elements, fused or not
                op.alg_elems_in = 8.0 * h * w + 8.0 * h * w + 16.0 * h * w + 16.0 * h4 * w4;                       // conv1, conv2, conv3, conv4 inputs
                op.alg_elems_out = 8.0 * h * w + 8.0 * h * w + 16.0 * h * w + 16.0 * h4 * w4 + 32.0 * h4 * w4;     // conv0 .. conv4 outputs
                op.macs = (plan.conv0.macs_per_out_pixel() + b0.dw.macs_per_out_pixel() + b0.pw.macs_per_out_pixel()) * h * w +
                          (b1.dw.macs_per_out_pixel() + b1.pw.macs_per_out_pixel()) * h4 * w4;
                op.hbm_elems_out = 32.0 * h4 * w4;                        // the frame in (alg_u8_in), the 32-channel net/4 map out
                op.launch = [sp](hipStream_t s, int n) { Stem2Params q = sp; q.n = n; launch_stem2(s, q); };
                L.ops.push_back(op);
                cur = out; c = b1.pw.cout; first_block = 2; h = h4; w = w4;
                if (dwpw2_variant() && plan.blocks.size() > 3 && plan.blocks[2].dw.cout == 32 && plan.blocks[2].pw.cout == 32 &&
                    plan.blocks[3].dw.stride == 2 && plan.blocks[3].pw.cout == 64) {
                    // blocks 2 and 3 (conv5..conv8) as ONE launch (dwpw2_kernel): the 112^2 x 32 map between them stays in LDS
                    const auto &ba = plan.blocks[2], &bb = plan.blocks[3];
                    T *out2 = act(bb.pw.out_blob, h / 2, w / 2, bb.pw.cout);
                    DwPw2Params dp;
                    dp.in = cur; dp.out = out2;
                    dp.dwa_mma = arena_.template ptr<uint32_t>(dw_w_[2].mma); dp.dwa_b = arena_.template ptr<float>(dw_w_[2].b);
                    dp.pwa_w = arena_.template ptr<half_t>(pw_w_[2].w); dp.pwa_b = arena_.template ptr<float>(pw_w_[2].b);
                    dp.dwb_mma = arena_.template ptr<uint32_t>(dw_w_[3].mma); dp.dwb_b = arena_.template ptr<float>(dw_w_[3].b);
                    dp.pwb_w = arena_.template ptr<half_t>(pw_w_[3].w); dp.pwb_b = arena_.template ptr<float>(pw_w_[3].b);
                    dp.n = 0; dp.hin = h; dp.win = w;
                    OpInfo o2;
                    o2.name = ba.dw.name + "+" + ba.pw.name + "+" + bb.dw.name + "+" + bb.pw.name;
                    o2.kernel = "dwpw2<32,32,64>";
                    const double pa = (double)h * w, pb = (double)(h / 2) * (w / 2);
                    o2.alg_elems_in = 32.0 * pa + 32.0 * pa + 32.0 * pa + 32.0 * pb;          // dw A, pw A, dw B, pw B inputs (layer-wise)
                    o2.alg_elems_out = 32.0 * pa + 32.0 * pa + 32.0 * pb + 64.0 * pb;
                    o2.macs = (ba.dw.macs_per_out_pixel() + ba.pw.macs_per_out_pixel()) * pa + (bb.dw.macs_per_out_pixel() + bb.pw.macs_per_out_pixel()) * pb;
                    o2.hbm_elems_in = 32.0 * pa; o2.hbm_elems_out = 64.0 * pb;
                    o2.launch = [dp](hipStream_t s, int n) { DwPw2Params q = dp; q.n = n; launch_dwpw2(s, q); };
                    L.ops.push_back(o2);
                    cur = out2; c = bb.pw.cout; first_block = 4; h /= 2; w /= 2;
                }
            }
        } else if constexpr (sizeof(T) == 1 || (sizeof(T) == 2 && kProbeBuild)) {
            // int8 engine (and the fp16 engine of the probe build with RF_STEM2=0): preprocess + conv0 + the first depthwise/pointwise
            // block are ONE launch (stem_kernel); it computes in fp16 and stores its 16-channel output in the engine's storage type
            const auto &blk = plan.blocks[0];
            T *out = act(blk.pw.out_blob, h, w, blk.pw.cout);
            StemParams<T> sp;
            sp.frames = L.d_frames + mb; sp.out = out;
            sp.w0 = arena_.template ptr<half_t>(c0_hi_); sp.b0 = arena_.template ptr<float>(WP::c0_b_mma_);
            sp.w0_raw = knob(K_STEM_RAW) ? arena_.template ptr<half_t>(WP::c0_raw_) : nullptr;
            sp.c0_tab = knob(K_STEM_RAW) == 2 ? arena_.template ptr<uint32_t>(WP::stem_c0tab_) : nullptr;
            sp.dw_w = arena_.template ptr<float>(stem_dw_.w); sp.dw_b = arena_.template ptr<float>(stem_dw_.b);
            sp.pw_w = arena_.template ptr<half_t>(stem_pw_.w); sp.pw_b = arena_.template ptr<float>(stem_pw_.b);
            sp.pw_m = mult_ptr(stem_pw_);
            sp.n = 0; sp.net_h = H; sp.net_w = W;
            OpInfo op;
            op.name = "pre+" + plan.conv0.name + "+" + blk.dw.name + "+" + blk.pw.name;
            op.kernel = "stem";
            op.alg_u8_in = 3.0 * P;
            op.alg_elems_in = 8.0 * h * w + 8.0 * h * w;                       // dw input, pw input
            op.alg_elems_out = 8.0 * h * w + 8.0 * h * w + 16.0 * h * w;       // conv0, dw, pw outputs
            op.macs = (plan.conv0.macs_per_out_pixel() + blk.dw.macs_per_out_pixel() + blk.pw.macs_per_out_pixel()) * h * w;
            op.hbm_elems_out = 16.0 * h * w;
            op.launch = [sp](hipStream_t s, int n) { StemParams<T> q = sp; q.n = n; launch_stem<T>(s, q); };
            L.ops.push_back(op);
            cur = out; c = blk.pw.cout; first_block = 1;
        } else if constexpr (sizeof(T) == 2) {
            throw Unsupported("fp16 engine without stem2: a probe-build configuration (RF_STEM2=0)");
        } else {
            cur = act(plan.conv0.out_blob, h, w, 8);
            OpInfo op;
            op.name = "pre+" + plan.conv0.name;
            op.kernel = "conv0";
            op.alg_u8_in = 3.0 * P;
            op.alg_elems_out = 8.0 * h * w;
            op.macs = plan.conv0.macs_per_out_pixel() * h * w;
            op.hbm_elems_out = 8.0 * h * w;
            const float *wp = arena_.template ptr<float>(c0_w_), *bp = arena_.template ptr<float>(c0_b_);
            const FrameDesc *fr = L.d_frames + mb;
            T *o = cur;
            op.launch = [=](hipStream_t s, int n) { launch_conv0<T>(s, fr, o, wp, bp, n, H, W); };
            L.ops.push_back(op);
        }
        // FPN tap -> lateral index: block 4 (stride 8) -> lateral[2], block 10 (stride 16) -> [1], block 12 (stride 32) -> [0]
        T *lat[3] = {nullptr, nullptr, nullptr};
        for (size_t i = first_block; i < plan.blocks.size(); i++) {
            const auto &blk = plan.blocks[i];
            int ho = h / blk.dw.stride, wo = w / blk.dw.stride;
            T *out = act(blk.pw.out_blob, ho, wo, blk.pw.cout);
            if (dwpw_tile_info<T>(c, blk.pw.cout, blk.dw.stride, true, ho, wo).th == 0)
                throw ModelError("no kernel instance for depthwise/pointwise block " + blk.dw.name);
            DwPwParams<T> p;
            p.in = cur; p.out = out;
            p.dw_w = arena_.template ptr<DWT>(dw_w_[i].w); p.dw_b = arena_.template ptr<float>(dw_w_[i].b);
            if (dw_w_[i].mma) p.dw_mma = arena_.template ptr<uint32_t>(dw_w_[i].mma);
            if (dw_w_[i].m != kNone) p.dw_m = arena_.template ptr<float>(dw_w_[i].m);
            p.pw_w = arena_.template ptr<T>(pw_w_[i].w); p.pw_b = arena_.template ptr<float>(pw_w_[i].b); p.pw_m = mult_ptr(pw_w_[i]);
            p.n = 0; p.hin = h; p.win = w; p.hout = ho; p.wout = wo;
            p.cin = c; p.cout = blk.pw.cout; p.stride = blk.dw.stride; p.has_dw = true;
            OpInfo op;
            op.name = blk.dw.name + "+" + blk.pw.name;
            op.kernel = "dwpw<" + std::to_string(c) + "," + std::to_string(blk.pw.cout) + ",s" + std::to_string(blk.dw.stride) + ">";
            op.alg_elems_in = (double)c * h * w + (double)c * ho * wo;
            op.alg_elems_out = (double)c * ho * wo + (double)blk.pw.cout * ho * wo;
            op.macs = (blk.dw.macs_per_out_pixel() + blk.pw.macs_per_out_pixel()) * ho * wo;
            op.hbm_elems_in = (double)c * h * w; op.hbm_elems_out = (double)blk.pw.cout * ho * wo;
            const int li = i == 4 ? 2 : i == 10 ? 1 : i == 12 ? 0 : -1;
            if (li >= 0) {          // the lateral 1x1 is computed from this block's output tile while it is in LDS
                const FoldedConv &lf = plan.lateral[li];
                lat[li] = act(lf.out_blob, ho, wo, 64);
                p.lat_w = arena_.template ptr<T>(lat_w_[li].w); p.lat_b = arena_.template ptr<float>(lat_w_[li].b); p.lat_out = lat[li];
                p.lat_m = mult_ptr(lat_w_[li]);
                op.name += "+" + lf.name;
                op.kernel.insert(op.kernel.size() - 1, ",lat");
                op.alg_elems_in += (double)blk.pw.cout * ho * wo;
                op.alg_elems_out += 64.0 * ho * wo;
                op.hbm_elems_out += 64.0 * ho * wo;
                op.macs += lf.macs_per_out_pixel() * ho * wo;
            }
            op.launch = [p](hipStream_t s, int n) { DwPwParams<T> q = p; q.n = n; launch_dwpw<T>(s, q); };
            L.ops.push_back(op);
            cur = out; h = ho; w = wo; c = blk.pw.cout;
        }
        // FPN: P3 = c3 lateral; P2 / P1 = aggr conv with "lateral + bilinear x2 upsample(coarser)" fused into its input staging
        T *feat[3];
        feat[0] = lat[0];
        for (int i = 0; i < 2; i++) {
            int fh = H / strides_[i + 1], fw = W / strides_[i + 1];
            feat[i + 1] = act(plan.aggr[i].out_blob, fh, fw, 64);
            Conv3Params<T> p;
            p.in = lat[i + 1]; p.in_ld = 64; p.in_off = 0; p.up = feat[i];
            p.w = arena_.template ptr<T>(aggr_w_[i].w); p.b = arena_.template ptr<float>(aggr_w_[i].b); p.m = mult_ptr(aggr_w_[i]);
            p.a_lat = aggr_a_lat_[i]; p.a_up = aggr_a_up_[i];
            p.blend_fp32 = knob(K_BLEND_FP32) != 0;            // read once per lane (not per launch)
            p.out0 = feat[i + 1]; p.ld0 = 64; p.off0 = 0; p.n0 = 64; p.out1 = nullptr; p.ld1 = 0; p.off1 = 0;
            p.n = 0; p.h = fh; p.w_ = fw; p.cin = 64; p.cout = 64;
            OpInfo op;
            op.name = std::string(i == 0 ? "rf_c3_upsampling" : "rf_c2_upsampling") + "+" + plan.aggr[i].name;
            {
                TileInfo ti = conv3x3_tile_info<T>(64, 64, fh, fw);
                op.kernel = "conv3x3<64,64," + std::to_string(ti.th) + "x" + std::to_string(ti.tw) + ",up>";
            }
            op.alg_elems_in = 64.0 * (fh / 2) * (fw / 2) + 64.0 * fh * fw;     // deconv input + conv input
            op.alg_elems_out = 64.0 * fh * fw + 64.0 * fh * fw;               // deconv output + conv output
            op.macs = plan.aggr[i].macs_per_out_pixel() * fh * fw + 16.0 * 64 * (fh / 2) * (fw / 2);
            op.hbm_elems_in = 64.0 * (fh / 2) * (fw / 2) + 64.0 * fh * fw; op.hbm_elems_out = 64.0 * fh * fw;
            op.launch = [p](hipStream_t s, int n) { Conv3Params<T> q = p; q.n = n; launch_conv3x3<T>(s, &q, 1); };
            L.ops.push_back(op);
        }
        // SSH context modules: each of the three merged convs is ONE launch covering strides 32, 16 and 8
        struct Level3 { Conv3Params<T> p[3]; };
        Level3 lv_a, lv_b, lv_c;
        OpInfo op_a, op_b, op_c, op_h;
        // fp16 / int8: conv_b and conv_c of the context module are ONE launch (ssh_tail_kernel); context_conv3_1 stays in LDS
        bool fuse_tail = false;
        if constexpr (sizeof(T) <= 2) fuse_tail = ssh_tail_variant() != 0;
        struct Tail3 { SshTailParams<T> p[3]; } tl;
        struct HeadLevels { HeadParams<T> p[3]; } hl;
        int anchor_off = 0;
        for (int i = 0; i < 3; i++) {
            const SshModule &m = plan.ssh[i];
            int fh = H / strides_[i], fw = W / strides_[i];
            std::string pre = "rf_c" + std::to_string(3 - i) + "_det_";
            T *cat = act(pre + "concat_relu", fh, fw, 64);
            T *ctx1 = act(pre + "context_conv1_relu", fh, fw, 16);
            T *ctx31 = fuse_tail ? nullptr : act(pre + "context_conv3_1_relu", fh, fw, 16);
            auto fill = [&](Conv3Params<T> &p, OpInfo &op, const FoldedConv &f, const GemmW &gw, const T *in, int cin, T *o0,
                            int ld0, int off0, int n0, T *o1, int ld1, int off1, int nlayers) {
                p.in = in; p.in_ld = cin; p.in_off = 0; p.up = nullptr;
                p.w = arena_.template ptr<T>(gw.w); p.b = arena_.template ptr<float>(gw.b); p.m = mult_ptr(gw);
                p.out0 = o0; p.ld0 = ld0; p.off0 = off0; p.n0 = n0; p.out1 = o1; p.ld1 = ld1; p.off1 = off1;
                p.n = 0; p.h = fh; p.w_ = fw; p.cin = cin; p.cout = f.cout;
                op.name += (op.name.empty() ? "" : " | ") + f.name;
                op.alg_elems_in += (double)nlayers * cin * fh * fw;   // each merged sibling reads the input once, layer-wise
                op.alg_elems_out += (double)f.cout * fh * fw;
                op.hbm_elems_in += (double)cin * fh * fw;              // merged siblings share ONE read of the input
                op.hbm_elems_out += (double)f.cout * fh * fw;
                op.macs += f.macs_per_out_pixel() * fh * fw;
            };
            fill(lv_a.p[i], op_a, m.conv_a, ssh_w_[i][0], feat[i], 64, cat, 64, 0, 32, ctx1, 16, 0, 2);
            fill(lv_b.p[i], op_b, m.conv_b, ssh_w_[i][1], ctx1, 16, cat, 64, 32, 16, ctx31, 16, 0, 2);
            fill(lv_c.p[i], op_c, m.conv_c, ssh_w_[i][2], ctx31, 16, cat, 64, 48, 16, nullptr, 0, 0, 1);
            if constexpr (sizeof(T) <= 2) {
                SshTailParams<T> &tp = tl.p[i];
                tp.in = ctx1; tp.cat = cat; tp.n = 0; tp.h = fh; tp.w_ = fw;
                tp.wb = arena_.template ptr<T>(ssh_w_[i][1].w); tp.bb = arena_.template ptr<float>(ssh_w_[i][1].b); tp.mb = mult_ptr(ssh_w_[i][1]);
                tp.wc = arena_.template ptr<T>(ssh_w_[i][2].w); tp.bc = arena_.template ptr<float>(ssh_w_[i][2].b); tp.mc = mult_ptr(ssh_w_[i][2]);
            }
            HeadParams<T> &hp = hl.p[i];
            hp.in = cat; hp.w = arena_.template ptr<T>(ssh_w_[i][3].w); hp.b = arena_.template ptr<float>(ssh_w_[i][3].b);
            hp.m = mult_ptr(ssh_w_[i][3]);
            hp.n = 0; hp.h = fh; hp.w_ = fw; hp.stride = strides_[i]; hp.anchor_offset = anchor_off;
            hp.num_anchors = na_;
            memset(hp.base, 0, sizeof(hp.base));
            preset_base_anchors(ratios_, i, hp.base);
            hp.net_h = H; hp.net_w = W; hp.params = L.d_params;
            hp.cand = L.d_cand; hp.cand_count = L.d_cand_count; hp.cap = opt_.max_candidates;
            hp.dump_prob = hp.dump_bbox = hp.dump_lmk = nullptr;
            if (opt_.keep_outputs) {
                const int chans[3] = {2 * head_a_, 4 * head_a_, 10 * head_a_};
                for (int k = 0; k < 3; k++) L.d_dump[i][k] = dalloc<float>((size_t)mb * chans[k] * fh * fw);
                hp.dump_prob = L.d_dump[i][0]; hp.dump_bbox = L.d_dump[i][1]; hp.dump_lmk = L.d_dump[i][2];
            }
            op_h.name += (op_h.name.empty() ? "" : " | ") + m.head.name;
            op_h.alg_elems_in += 3.0 * 64 * fh * fw;
            op_h.alg_elems_out += 16.0 * head_a_ * fh * fw;
            op_h.hbm_elems_in += 64.0 * fh * fw;                       // the concat tensor in; candidates out: a few KB
            op_h.macs += m.head.macs_per_out_pixel() * fh * fw;
            anchor_off += na_ * fh * fw;
        }
        op_a.launch = [lv_a](hipStream_t s, int n) { Level3 q = lv_a; for (auto &p : q.p) p.n = n; launch_conv3x3<T>(s, q.p, 3); };
        op_b.launch = [lv_b](hipStream_t s, int n) { Level3 q = lv_b; for (auto &p : q.p) p.n = n; launch_conv3x3<T>(s, q.p, 3); };
        op_c.launch = [lv_c](hipStream_t s, int n) { Level3 q = lv_c; for (auto &p : q.p) p.n = n; launch_conv3x3<T>(s, q.p, 3); };
        op_a.kernel = "conv3x3<64,48,8x8>";
        op_b.kernel = "conv3x3<16,32,8x8>";
        op_c.kernel = "conv3x3<16,16,8x8>";
        op_h.kernel = "head";
        L.ops.push_back(op_a);
        if (fuse_tail) {
            if constexpr (sizeof(T) <= 2) {
                OpInfo op_t;
                op_t.name = op_b.name + " | " + op_c.name;
                op_t.kernel = "ssh_tail<16,32,16>";
                op_t.alg_elems_in = op_b.alg_elems_in + op_c.alg_elems_in;       // layer-wise accounting, fused or not
                op_t.alg_elems_out = op_b.alg_elems_out + op_c.alg_elems_out;
                op_t.macs = op_b.macs + op_c.macs;
                op_t.hbm_elems_in = op_b.hbm_elems_in;                 // context_conv1 in; context_conv3_1 stays in LDS
                op_t.hbm_elems_out = op_b.hbm_elems_out + op_c.hbm_elems_out - op_c.hbm_elems_in;      // concat[32:64]: 16 + 16 channels
                op_t.launch = [tl](hipStream_t s, int n) { Tail3 q = tl; for (auto &p : q.p) p.n = n; launch_ssh_tail<T>(s, q.p, 3); };
                L.ops.push_back(op_t);
            }
        } else {
            L.ops.push_back(op_b);
            L.ops.push_back(op_c);
        }
        L.first_post = L.ops.size();
        op_h.name += " +softmax+decode";
        op_h.launch = [hl](hipStream_t s, int n) { HeadLevels q = hl; for (auto &p : q.p) p.n = n; launch_head<T>(s, q.p, 3); };
        L.ops.push_back(op_h);
        {
            NmsParams np;
            np.cand = L.d_cand; np.cand_count = L.d_cand_count; np.cap = opt_.max_candidates; np.params = L.d_params;
            np.out = L.h_out; np.out_count = L.h_counts; np.out_cand_count = L.h_counts + mb;
            np.max_det = opt_.max_detections; np.n = 0;
            OpInfo op;
            op.name = "sort+nms";
            op.kernel = "nms";
            op.launch = [np](hipStream_t s, int n) { NmsParams q = np; q.n = n; launch_nms(s, q); };
            L.ops.push_back(op);
        }
    }

    // ------------------------------------------------------------------------------------------ run
    static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

    // img.empty() (RetinaFace.cpp:578-580) is legal and yields count 0; everything else must be a sane CV_8UC3 view
    void check_frame(const uint8_t *ptr, int rows, int cols, int step) const {
        if (!ptr || rows <= 0 || cols <= 0) return;
        if (rows > 4096 * 3072 / std::max(cols, 1)) throw ArgError("frame larger than 4096x3072 (RetinaFace.cpp:325)");
        if (step < cols * 3) throw ArgError("row step smaller than cols*3");
    }

    // Staging capacity of a lane: room for a full super-batch of net-sized frames, or for what one enqueue needs if that is more
    // (sized by the frames actually staged, not by cap_images x the largest frame ever seen).
    void ensure_stage(Lane &L, size_t need, bool device_only = false) {
        if (need <= L.stage_cap && (device_only || L.h_stage)) return;
        const size_t want = std::max(std::max(need, L.stage_cap), (size_t)cap_images_ * align256((size_t)net_h_ * net_w_ * 3));
        RF_HIP(hipStreamSynchronize(L.stream));
        if (L.d_stage) RF_HIP(hipFree(L.d_stage));
        if (L.h_stage) RF_HIP(hipHostFree(L.h_stage));
        L.d_stage = nullptr; L.h_stage = nullptr; L.stage_cap = 0;
        // frames scattered from other GPUs (peer copies) never touch host memory: no pinned mirror for a handle that only stages those
        if (!device_only) RF_HIP(hipHostMalloc((void **)&L.h_stage, want, hipHostMallocDefault));
        RF_HIP(hipMalloc((void **)&L.d_stage, want));
        L.stage_cap = want;
    }

    // Where a caller's "device" frame lives (rf_detect_batch_device / rf_enqueue_batch_device).  -1: readable in place by this
    // engine's kernels (its own device's memory, or pinned / managed host memory); >= 0: the ordinal of ANOTHER device of the
    // node -- the frame is then scattered to this device over xGMI before the launch (submit()).  Anything the runtime does not
    // know as device-accessible memory is refused here instead of faulting inside a kernel.
    // The answer is cached per ALLOCATION (hipMemGetAddressRange: base + size of the allocation the pointer lies in), so a camera
    // ring or a frame tensor costs one runtime lookup, not one per frame per call (hipPointerGetAttributes takes the runtime's
    // allocation-map lock: ~1 us x 256 frames per super-batch).  Round 4 cached per 2 MiB page and never looked again: a buffer
    // freed and re-allocated under the same address on another device kept its old answer.  Now (1) an entry covers
    // exactly one allocation, so two small allocations sharing a page cannot alias; (2) an entry is re-validated against the
    // runtime once it is older than kResidencyTtlUs -- base, size and owner must still match or the entry is dropped and the
    // pointer looked up afresh -- which bounds the lifetime of a stale answer to 2 ms of wall clock and costs one runtime call
    // per 2 ms per allocation; (3) rf_invalidate_residency() drops the cache at once for callers that free or re-home frame
    // buffers while the handle lives (documented in include/retinaface_amd.h).  Host / managed memory and pointers the runtime
    // rejects are never cached.
    static constexpr double kResidencyTtlUs = 2000.0;
    struct Residency { uintptr_t base; size_t bytes; int where; int owner; double checked_us; };
    static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // one runtime lookup; returns false for host / managed memory (answer in *where, not cacheable)
    bool lookup_residency(const uint8_t *p, Residency *out) {
        hipPointerAttribute_t attr;
        memset(&attr, 0, sizeof(attr));
        if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
            (void)hipGetLastError();
            throw ArgError("device frame pointer is not known to the HIP runtime (host memory passed to a *_device entry point?)");
        }
        if (attr.type == hipMemoryTypeUnregistered) throw ArgError("device frame pointer is unregistered host memory");
        out->owner = attr.device;
        out->checked_us = now_us();
        if (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeManaged) { out->where = -1; out->base = 0; out->bytes = 0; return false; }
        out->where = attr.device == device_ ? (force_scatter_ ? device_ : -1) : attr.device;
        hipDeviceptr_t base = nullptr;
        size_t bytes = 0;
        if (hipMemGetAddressRange(&base, &bytes, (hipDeviceptr_t)p) != hipSuccess || !base || !bytes) {
            (void)hipGetLastError();
            out->base = 0; out->bytes = 0;
            return false;                       // a device pointer whose allocation cannot be delimited is looked up every time
        }
        out->base = (uintptr_t)base; out->bytes = bytes;
        return true;
    }

    int foreign_device_of(const uint8_t *p) {
        const uintptr_t a = (uintptr_t)p;
        auto it = residency_.upper_bound(a);
        if (it != residency_.begin()) {
            --it;
            Residency &r = it->second;
            if (a >= r.base && a < r.base + r.bytes) {
                if (now_us() - r.checked_us <= kResidencyTtlUs) return r.where;
                Residency fresh;                                  // stale: the allocation must still be the one that was cached
                const bool cacheable = lookup_residency(p, &fresh);
                if (cacheable && fresh.base == r.base && fresh.bytes == r.bytes && fresh.owner == r.owner) { r.checked_us = fresh.checked_us; return r.where; }
                residency_.erase(it);
                residency_revalidation_misses_++;
                if (cacheable) insert_residency(fresh);
                return fresh.where;
            }
        }
        Residency fresh;
        if (lookup_residency(p, &fresh)) insert_residency(fresh);
        return fresh.where;
    }
    // a new allocation may overlap stale entries of freed ones: drop every entry that intersects it (both the miss path and the
    // stale-revalidation path come through here)
    void insert_residency(const Residency &fresh) {
        if (residency_.size() > 4096) residency_.clear();
        auto lo = residency_.lower_bound(fresh.base);
        if (lo != residency_.begin()) { auto pv = std::prev(lo); if (pv->second.base + pv->second.bytes > fresh.base) lo = pv; }
        while (lo != residency_.end() && lo->second.base < fresh.base + fresh.bytes) lo = residency_.erase(lo);
        residency_[fresh.base] = fresh;
    }

    bool is_registered(const uint8_t *p, size_t bytes) const {
        for (const auto &r : registered_)
            if ((uintptr_t)p >= r.base && (uintptr_t)p + bytes <= r.base + r.bytes) return true;
        return false;
    }
    void drop_range(const void *ptr, bool unpin) {
        DeviceGuard guard(device_);
        for (size_t i = 0; i < registered_.size(); i++)
            if (registered_[i].base == (uintptr_t)ptr) {
                for (auto &l : lanes_) if (l.stream) RF_HIP(hipStreamSynchronize(l.stream));     // no DMA may still read it
                if (unpin && registered_[i].owned) RF_HIP(hipHostUnregister((void *)ptr));
                registered_.erase(registered_.begin() + i);
                return;
            }
        throw ArgError("host_unregister: range was not registered on this handle");
    }

    void wait_impl(int ticket, rf_face *out, int cap_per_image, int *counts, bool *truncated) {
        if (ticket < 0 || ticket >= (int)tickets_.size() || tickets_[ticket].state == Ticket::FREE)
            throw ArgError("wait: invalid ticket");
        Ticket &t = tickets_[ticket];
        if (t.state == Ticket::PENDING) launch_pending();          // its super-batch has not been launched yet
        if (t.state == Ticket::LAUNCHED) {
            // About to block on a running super-batch: a partial one queued behind it starts now and runs on its own lane
            // meanwhile (the tail of a burst overlaps instead of serialising).  When the waited batch has already finished
            // nothing is flushed -- in a steady pipeline the pending batch keeps filling up to max_batch x coalesce images.
            Lane &L = lanes_[t.lane];
            if (L.busy && hipEventQuery(L.done) == hipErrorNotReady) launch_pending();
            harvest(L);                                             // first waiter of a super-batch collects all of it
        }
        bool tr = false;
        last_lane_ = t.lane;
        last_first_image_ = t.first_image;
        last_n_ = t.n;
        last_cand_counts_.assign(t.n, 0);
        if ((int)last_anchor_.size() < t.n) last_anchor_.resize(t.n);      // per-image vectors keep their capacity across waits
        for (int i = 0; i < t.n; i++) {
            int kept = t.kept[i], ncand = t.ncand[i];
            last_cand_counts_[i] = ncand;
            if (ncand > opt_.max_candidates) tr = true;
            int avail = std::min(kept, opt_.max_detections);
            if (kept > opt_.max_detections) tr = true;
            if (counts) counts[i] = kept;
            int ncopy = out ? std::min(avail, cap_per_image) : 0;
            if (out && avail > cap_per_image) tr = true;
            const Candidate *src = t.records.data() + t.first_record[i];
            last_anchor_[i].resize(avail);
            for (int k = 0; k < avail; k++) last_anchor_[i][k] = src[k].anchor;
            for (int k = 0; k < ncopy; k++) memcpy(&out[(size_t)i * cap_per_image + k], &src[k], sizeof(rf_face));
        }
        if (t.timed) {
            Lane &s = lanes_[t.lane];
            float a = 0, b = 0, c = 0;
            (void)hipEventElapsedTime(&a, s.time_ev[0], s.time_ev[1]);
            (void)hipEventElapsedTime(&b, s.time_ev[1], s.time_ev[2]);
            (void)hipEventElapsedTime(&c, s.time_ev[2], s.time_ev[3]);
            t_pre_ = a; t_infer_ = b; t_post_ = c; t_total_ = a + b + c;
            have_split_ = true;
        }
        t.state = Ticket::FREE;
        if (truncated) *truncated = tr;
    }

    int alloc_ticket() {
        for (size_t k = 0; k < tickets_.size(); k++) {
            int id = (int)((next_ticket_ + k) % tickets_.size());
            if (tickets_[id].state == Ticket::FREE) { next_ticket_ = (id + 1) % (int)tickets_.size(); return id; }
        }
        throw ArgError("too many outstanding tickets: call rf_wait before enqueueing more");
    }

    // copy a finished super-batch's results out of the lane's pinned block into its tickets, so the lane can be reused
    void harvest(Lane &L) {
        if (!L.busy) return;
        double tt = trace_.on ? HostTrace::now() : 0.0;
        RF_HIP(hipEventSynchronize(L.done));
        trace_.add(5, tt);
        tt = trace_.on ? HostTrace::now() : 0.0;
        struct AtExit { HostTrace &t; double t0; ~AtExit() { t.add(6, t0); } } at_exit{trace_, tt};
        const int mb = cap_images_;
        for (int id : L.tickets) {
            Ticket &t = tickets_[id];
            t.kept.assign(t.n, 0); t.ncand.assign(t.n, 0); t.first_record.assign(t.n, 0);
            t.records.clear();
            for (int i = 0; i < t.n; i++) {
                const int img = t.first_image + i;
                if (L.empty[img]) continue;
                t.kept[i] = L.h_counts[img];
                t.ncand[i] = L.h_counts[mb + img];
                t.first_record[i] = (int)t.records.size();
                const int avail = std::min(t.kept[i], opt_.max_detections);
                const Candidate *src = L.h_out + (size_t)img * opt_.max_detections;
                t.records.insert(t.records.end(), src, src + avail);
            }
            t.state = Ticket::DONE;
        }
        L.tickets.clear();
        L.busy = false;
        L.n_images = 0;
    }

    void launch_pending() {
        if (pending_lane_ < 0) return;
        Lane &s = lanes_[pending_lane_];
        pending_lane_ = -1;
        const int n = s.n_images;
        if (n == 0) return;
        *s.h_params = RunParams{s.threshold, nms_threshold_, n, 0};
        if (s.need_resize) {
            const int mb = cap_images_;
            for (int i = 0; i < n; i++)
                s.h_frames[mb + i] = FrameDesc{s.d_canvas + (size_t)i * net_h_ * net_w_ * 3, net_h_, net_w_, net_w_ * 3, 0};
        }
        if (s.copy2_used) {             // uploads that went through the second copy stream: the launch waits for them
            RF_HIP(hipEventRecord(s.copy2_done, s.copy2));
            RF_HIP(hipStreamWaitEvent(s.stream, s.copy2_done, 0));
            s.copy2_used = false;
        }
        double tt = trace_.on ? HostTrace::now() : 0.0;
        RF_HIP(hipMemcpyAsync(s.d_frames, s.h_frames, s.table_bytes, hipMemcpyHostToDevice, s.stream));
        trace_.add(2, tt);
        tt = trace_.on ? HostTrace::now() : 0.0;
        if (s.need_resize) {
            if (opt_.resize_bilinear) launch_resize_bilinear(s.stream, s.d_frames, s.d_canvas, n, net_h_, net_w_);
            else launch_resize_area(s.stream, s.d_frames, s.d_canvas, n, net_h_, net_w_);
        }
        const bool eager_timed = s.timed;
        if (eager_timed) RF_HIP(hipEventRecord(s.time_ev[1], s.stream));
        bind_launch_cus(s.cus);
        struct Unbind { ~Unbind() { bind_launch_cus(0); } } unbind;
        if (opt_.use_graph && s.warmed.count(n) && n <= head_start_max_ && s.ops.size() > 2) {
            // Head start for small launches (round 5, RF_HEAD_START): hipGraphLaunch costs the host ~16 us before the first kernel can start, a
            // tenth of a synchronous batch-8 call.  The first kernel (the stem: the longest of a small launch, 13 us at batch 8) is launched
            // eagerly -- ~3 us of host time -- and the graph of the remaining launches is submitted while it runs.
            auto it = s.graphs.find(-n);
            if (it == s.graphs.end()) it = s.graphs.emplace(-n, capture(s, n, 1)).first;
            s.ops[0].launch(s.stream, n);
            RF_HIP(hipGraphLaunch(it->second, s.stream));
        } else if (opt_.use_graph && s.warmed.count(n)) {
            auto it = s.graphs.find(n);
            if (it == s.graphs.end()) it = s.graphs.emplace(n, capture(s, n)).first;
            RF_HIP(hipGraphLaunch(it->second, s.stream));
        } else {
            for (size_t k = 0; k < s.ops.size(); k++) {
                if (eager_timed && k == s.first_post) RF_HIP(hipEventRecord(s.time_ev[2], s.stream));
                s.ops[k].launch(s.stream, n);
            }
            s.warmed.insert(n);    // first run of a batch size is always eager: function attributes get set outside capture
        }
        RF_HIP(hipGetLastError());
        trace_.add(3, tt);
        tt = trace_.on ? HostTrace::now() : 0.0;
        if (eager_timed) RF_HIP(hipEventRecord(s.time_ev[3], s.stream));
        RF_HIP(hipEventRecord(s.done, s.stream));
        trace_.add(4, tt);
        s.busy = true;
        s.launch_seq = ++launch_counter_;
        for (int id : s.tickets) tickets_[id].state = Ticket::LAUNCHED;
    }

    // One enqueue / one chunk of a synchronous call joins the super-batch being assembled (or opens the next lane).  Everything
    // that can fail for a caller-side reason runs BEFORE any state changes: a thrown error leaves no half-open lane or ticket.
    int submit(const uint8_t *const *frames, const int *rows, const int *cols, const int *steps, int n, bool on_device,
               float threshold, bool sync_call) {
        const int mb = cap_images_;
        bool need_resize = false, all_registered = true;
        size_t stage_need = 0;
        std::vector<char> empty(n, 0);
        std::vector<size_t> off(n, 0);
        std::vector<int> src_dev(n, -1);          // device frames: >= 0 = resident on that OTHER device, staged by a peer copy
        for (int i = 0; i < n; i++) {
            check_frame(frames[i], rows[i], cols[i], steps[i]);
            empty[i] = !frames[i] || rows[i] <= 0 || cols[i] <= 0;      // img.empty(), RetinaFace.cpp:578-580
            if (empty[i]) continue;
            if (rows[i] > net_h_ || cols[i] > net_w_) need_resize = true;
            if (!on_device) {
                off[i] = stage_need;
                stage_need += align256((size_t)rows[i] * cols[i] * 3);
                all_registered = all_registered && is_registered(frames[i], (size_t)(rows[i] - 1) * steps[i] + (size_t)cols[i] * 3);
            } else if (check_residency_) {
                src_dev[i] = foreign_device_of(frames[i]);
                if (src_dev[i] >= 0) {            // the rows keep the caller's step: one contiguous span, one peer copy
                    off[i] = stage_need;
                    stage_need += align256((size_t)(rows[i] - 1) * steps[i] + (size_t)cols[i] * 3);
                }
            }
        }
        const bool eager_timed = sync_call && !opt_.use_graph;
        // a pending super-batch is closed when this chunk does not fit (images or staging bytes) or must not be mixed with it
        // (different threshold, timed eager run)
        if (pending_lane_ >= 0) {
            Lane &p = lanes_[pending_lane_];
            if (p.n_images + n > mb || p.threshold != threshold || eager_timed || p.timed ||
                (stage_need && p.stage_used + stage_need > p.stage_cap) ||
                (stage_need && !on_device && !p.h_stage))             // the open lane only has a device-side staging block (peer copies so far)
                launch_pending();
        }
        const int id = alloc_ticket();
        if (pending_lane_ < 0) {
            const int lane = pick_lane();
            Lane &s = lanes_[lane];
            harvest(s);                       // waits for the previous super-batch on this lane, if any
            if (stage_need) ensure_stage(s, stage_need, on_device);
            next_lane_ = (lane + 1) % (int)lanes_.size();
            s.n_images = 0;
            s.stage_used = 0;
            s.threshold = threshold;
            s.need_resize = false;
            s.timed = eager_timed;
            s.empty.assign(mb, 0);
            s.tickets.clear();
            pending_lane_ = lane;
            if (eager_timed) RF_HIP(hipEventRecord(s.time_ev[0], s.stream));
        }
        Lane &s = lanes_[pending_lane_];
        Ticket &t = tickets_[id];
        t.state = Ticket::PENDING; t.lane = pending_lane_; t.first_image = s.n_images; t.n = n; t.timed = eager_timed;
        // the launch's frame table (nothing below reads it before it is complete; a throw leaves n_images where it was)
        for (int i = 0; i < n; i++) {
            const int img = s.n_images + i;
            FrameDesc src{nullptr, 0, 0, 0, 0};
            s.empty[img] = empty[i];
            if (!empty[i]) {
                if (on_device && src_dev[i] < 0) src = FrameDesc{frames[i], rows[i], cols[i], steps[i], 0};
                else if (on_device) src = FrameDesc{s.d_stage + s.stage_used + off[i], rows[i], cols[i], steps[i], 0};
                else src = FrameDesc{s.d_stage + s.stage_used + off[i], rows[i], cols[i], cols[i] * 3, 0};
            }
            s.h_frames[img] = src;
            s.h_frames[mb + img] = src;       // replaced by the canvas in launch_pending() when a resize is needed
        }
        try {
            if (stage_need) {
                uint8_t *hbase = s.h_stage + s.stage_used, *dbase = s.d_stage + s.stage_used;
                hipStream_t up = s.stream;
                // staged (pageable) frames: measured 69-74 k -> 76-81 k images/s at 448 x 448 with the second stream (48.6 GB/s = 0.9 of
                // the box's pinned-copy rate); frames in rf_host_register'ed memory got SLOWER with it (75.6 k -> 69.4 k) and stay on one
                if (s.copy2 && !on_device && !all_registered && (s.uploads++ & 1)) { up = s.copy2; s.copy2_used = true; }
                if (on_device) {
                    // the batch split of a multi-GPU node: frames resident on another device cross xGMI as one peer copy each
                    // (SDMA, on this lane's stream: it overlaps the compute of the super-batches in flight on the other lanes)
                    // Frames that follow each other in the source allocation AND in the staging block (dense rows, a span that is a
                    // multiple of the 256-byte staging alignment: every BASELINE shape) travel as ONE copy per run -- a contiguous slice of a
                    // sharded batch is one SDMA descriptor instead of 32 (round 6; RF_SCATTER_PER_FRAME=1 keeps one copy per frame for the A/B).
                    for (int i = 0; i < n; i++) {
                        if (empty[i] || src_dev[i] < 0) continue;
                        size_t run = (size_t)(rows[i] - 1) * steps[i] + (size_t)cols[i] * 3;
                        int j = i + 1;
                        if (!scatter_per_frame_ && steps[i] == cols[i] * 3)
                            while (j < n && !empty[j] && src_dev[j] == src_dev[i] && steps[j] == cols[j] * 3 && frames[j] == frames[i] + run &&
                                   off[j] == off[i] + run)
                                run += (size_t)rows[j] * cols[j] * 3, j++;
                        RF_HIP(hipMemcpyPeerAsync(dbase + off[i], device_, frames[i], src_dev[i], run, s.stream));
                        scattered_frames_ += j - i;
                        peer_copies_++;
                        i = j - 1;
                    }
                } else if (all_registered) {
                    // caller buffers pinned with rf_host_register: the DMA engine reads them in place.  Frames with dense rows
                    // that follow each other in memory (a ring of camera buffers) and in the staging block go as ONE copy.
                    upload_registered(frames, rows, cols, steps, empty, off, 0, n, dbase, up);
                } else {
                    // PIPELINED STAGING of a synchronous call (round 6): the reference's calling convention is ONE call
                    // that uploads, preprocesses and infers (RetinaFace.cpp:749-846), and through round 5 such a call ran staging copy, PCIe
                    // transfer and compute strictly one after the other (batch 8 at 448 x 448, C ABI: 0.285 ms = 0.06 + 0.10 + 0.13).  The frames
                    // are now staged and sent in pieces of ~kStagePieceBytes: while piece k crosses the bus the host stages piece k + 1.  Same
                    // stream, same bytes, same launch afterwards: results are byte-identical (RF_SYNC_SPLIT=0 = one piece, the A/B).
                    // Measured and rejected in the same round (profiles/r06_sync_host_call_ab.txt): sending the pieces on the copy stream and
                    // starting the STEM of each piece behind its event -- 4 events + 4 cross-stream waits + 4 eager launches + a second graph
                    // cost more host time (0.309 / 0.278 ms pageable / registered) than the 13 us stem they hid (0.285 / 0.224 ms unsplit).
                    const int pieces = (sync_call && stage_pieces_ > 1) ? (int)std::min<size_t>((size_t)stage_pieces_, std::max<size_t>(1, stage_need / kStagePieceBytes)) : 1;
                    if (pieces <= 1) {
                        copy_jobs_.clear();
                        for (int i = 0; i < n; i++)
                            if (!empty[i])
                                copy_jobs_.push_back(ParallelCopier::Job{hbase + off[i], frames[i], (size_t)cols[i] * 3, (size_t)rows[i], (size_t)steps[i]});
                        copier_->run(copy_jobs_);          // the caller's buffers are free again when this returns
                        RF_HIP(hipMemcpyAsync(dbase, hbase, stage_need, hipMemcpyHostToDevice, up));
                    } else {
                        // piece boundaries are byte positions of the staging block cut at row granularity: a piece is whole frames and / or a
                        // row range of a frame, so ONE large frame (1280 x 896 = 3.4 MB) is pipelined as well
                        size_t sent = 0;
                        int i = 0, r = 0;                          // next frame / next row of it to stage
                        for (int pc = 0; pc < pieces; pc++) {
                            const size_t goal = pc == pieces - 1 ? stage_need : stage_need * (pc + 1) / pieces;
                            copy_jobs_.clear();
                            size_t end = sent;
                            while (i < n && end < goal) {
                                if (empty[i]) { i++; r = 0; continue; }
                                const size_t rb = (size_t)cols[i] * 3, at = off[i] + (size_t)r * rb;
                                int take = rows[i] - r;
                                if (at + (size_t)take * rb > goal) take = (int)std::max<size_t>(1, (goal - std::min(goal, at) + rb - 1) / rb);
                                take = std::min(take, rows[i] - r);
                                copy_jobs_.push_back(ParallelCopier::Job{hbase + at, frames[i] + (size_t)r * steps[i], rb, (size_t)take, (size_t)steps[i]});
                                end = at + (size_t)take * rb;
                                r += take;
                                if (r == rows[i]) { i++; r = 0; }
                            }
                            if (copy_jobs_.empty()) continue;
                            copier_->run(copy_jobs_);      // the caller's rows of this piece are free again when this returns
                            RF_HIP(hipMemcpyAsync(dbase + sent, hbase + sent, end - sent, hipMemcpyHostToDevice, up));
                            sent = end;
                        }
                        staged_pieces_ += pieces;
                    }
                }
            }
        } catch (...) {
            t.state = Ticket::FREE;
            throw;
        }
        s.stage_used += stage_need;
        s.need_resize = s.need_resize || need_resize;
        s.n_images += n;
        s.tickets.push_back(id);
        // launch now when full, when the caller is synchronous, or when coalescing is off
        if (s.n_images + 1 > mb || sync_call || opt_.coalesce == 1) launch_pending();
        return id;
    }

    // frames [a, b) of a chunk, in pinned caller memory, to their places in the device staging block: one DMA per run of dense frames that
    // follow each other in memory and in the block, a 2-D copy for frames with a row pitch
    void upload_registered(const uint8_t *const *frames, const int *rows, const int *cols, const int *steps, const std::vector<char> &empty,
                           const std::vector<size_t> &off, int a, int b, uint8_t *dbase, hipStream_t up) {
        for (int i = a; i < b; i++) {
            if (empty[i]) continue;
            const size_t fb = (size_t)rows[i] * cols[i] * 3;
            if (steps[i] != cols[i] * 3) {
                RF_HIP(hipMemcpy2DAsync(dbase + off[i], (size_t)cols[i] * 3, frames[i], (size_t)steps[i], (size_t)cols[i] * 3,
                                        (size_t)rows[i], hipMemcpyHostToDevice, up));
                continue;
            }
            size_t run = fb;
            int j = i + 1;
            while (j < b && !empty[j] && steps[j] == cols[j] * 3 && frames[j] == frames[i] + run && off[j] == off[i] + run)
                run += (size_t)rows[j] * cols[j] * 3, j++;
            RF_HIP(hipMemcpyAsync(dbase + off[i], frames[i], run, hipMemcpyHostToDevice, up));
            i = j - 1;
        }
    }

    hipGraphExec_t capture(Lane &L, int n, size_t first_op = 0) {
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        RF_HIP(hipStreamBeginCapture(L.stream, hipStreamCaptureModeThreadLocal));
        std::string err;
        try {
            for (size_t k = first_op; k < L.ops.size(); k++) L.ops[k].launch(L.stream, n);
        } catch (const std::exception &e) { err = e.what(); }
        hipError_t end = hipStreamEndCapture(L.stream, &g);
        if (!err.empty() || end != hipSuccess || !g)
            throw HipError("hipGraph capture failed: " + (err.empty() ? std::string(hipGetErrorString(end)) : err));
        hipError_t inst = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (inst != hipSuccess) throw HipError(std::string("hipGraphInstantiate failed: ") + hipGetErrorString(inst));
        return ge;
    }

    // ------------------------------------------------------------------------------------------ state
    int device_ = 0;
    HostTrace trace_;
    bool check_residency_ = false, force_scatter_ = false, scatter_per_frame_ = false;
    int stage_pieces_ = 4;                    // most pieces a synchronous host-frame call is staged + sent in (RF_SYNC_SPLIT=0: one, as rounds 1-5; RF_SYNC_PIECES=n)
    static constexpr size_t kStagePieceBytes = 1200 << 10;      // ~2 frames of 448 x 448: 24 us on the bus, ~15 us of staging
    long staged_pieces_ = 0;
    int copy_streams_ = 2;
    int head_start_max_ = 0;                   // launches of at most this many images start their first kernel eagerly ahead of the graph
    long scattered_frames_ = 0;               // device frames that arrived from another device
    long peer_copies_ = 0;                    // ... in this many hipMemcpyPeerAsync calls (one per contiguous run of frames)
    std::vector<float> ratios_;                // the network preset's anchor ratios (empty: a preset without anchors)
    int na_ = 2;                               // anchors per cell the preset decodes (head_a_: what the model's heads carry)
    std::unique_ptr<ParallelCopier> copier_;
    std::vector<ParallelCopier::Job> copy_jobs_;
    struct HostRange { uintptr_t base; size_t bytes; bool owned; };
    std::vector<HostRange> registered_;       // rf_host_register ranges (pinned caller memory; owned = pinned by this engine)
    std::vector<hipEvent_t> prof_ev_;
    std::vector<void *> dev_allocs_, host_allocs_;
    const int strides_[3] = {32, 16, 8};

    Plan plan_;                               // skeleton (shapes, names): what build_lane needs when a lane is built later
    Lane *building_ = nullptr;
    std::vector<Lane> lanes_;
    int next_lane_ = 0, last_lane_ = 0, last_first_image_ = 0, pending_lane_ = -1;
    unsigned long long launch_counter_ = 0;
    std::map<uintptr_t, Residency> residency_;          // allocation base -> where it lives (foreign_device_of)
    long residency_revalidation_misses_ = 0;            // cached entries whose allocation had changed when they were re-validated
    int cap_images_ = 0;                      // images per launch = max_batch * coalesce
    std::vector<Ticket> tickets_;
    int next_ticket_ = 0;

    int last_n_ = 0;
    std::vector<int> last_cand_counts_;
    std::vector<std::vector<int32_t>> last_anchor_;
    float t_pre_ = 0, t_infer_ = 0, t_post_ = 0, t_total_ = 0;
    bool have_split_ = false;
};

}  // namespace

std::string plan_cache_path(const std::string &model_dir, const std::string &stem, int precision) {
    static const char *names[3] = {"fp32", "fp16", "int8"};
    return model_dir + "/" + stem + "." + names[precision < 0 || precision > 2 ? 1 : precision] + ".rfplan";
}

// Host-only half of engine start-up: the packed weight image and the plan skeleton, from the plan cache when it matches the
// model files, otherwise from the model (and then the cache is (re)written, best effort: a read-only model directory is fine).
template <typename T>
void prepare_pack(const std::string &model_dir, const EngineOptions &opt, Plan *plan, WeightPack<T> *pack, bool *from_cache) {
    PlanCacheKey key;
    key.build = 1469598103934665603ull;
    for (const char *c = __DATE__ " " __TIME__ " " __FILE__; *c; c++) { key.build ^= (unsigned char)*c; key.build *= 1099511628211ull; }
    key.precision = opt.precision;
    key.stem2 = std::is_same<T, half_t>::value ? stem2_variant() : 0;
    if (knob(K_STEM2_DC) == 0) key.stem2 |= 0x100;     // probe knob RF_STEM2_DC: another packed image
    const std::string path = opt.plan_cache_path.empty() ? plan_cache_path(model_dir, opt.model_stem, opt.precision) : opt.plan_cache_path;
    *from_cache = false;
    if (opt.plan_cache) {
        key.source_hash = model_source_hash(model_dir, opt.model_stem);        // throws IoError when the model is missing
        std::string bytes;
        if (read_file_if_exists(path, &bytes)) {
            try {
                if (load_plan_cache<T>(bytes, key, plan, pack)) { *from_cache = true; return; }
            } catch (const IoError &) { /* damaged cache file: rebuild below and overwrite it */ }
            *plan = Plan();
            *pack = WeightPack<T>();
        }
    }
    Model model = load_model_dir(model_dir, opt.model_stem);
    *plan = compile_plan(model);
    if (opt.precision == RF_PRECISION_INT8 && plan->int8_scales.empty())
        throw Unsupported("int8 precision needs a calibration table (<stem>.table.int8, or scales inside the .rfw)");
    pack->pack(*plan);
    if (opt.plan_cache) write_file_best_effort(path, save_plan_cache<T>(key, *plan, *pack));
}

template <typename T>
static std::unique_ptr<Engine> make_engine(const std::string &model_dir, const std::string &network, float nms, const EngineOptions &opt,
                                           const std::vector<float> &ratios) {
    Plan plan;
    WeightPack<T> pack;
    bool from_cache = false;
    prepare_pack<T>(model_dir, opt, &plan, &pack, &from_cache);       // host-only steps first: their errors do not need a GPU
    if (!ratios.empty() && 2 * (int)ratios.size() != plan.anchors_per_cell)
        // the reference would index past the end of the score blob here (RetinaFace.cpp:669-694 with _num_anchors != blob channels / 2)
        throw ModelError("network preset '" + network + "' decodes " + std::to_string(2 * ratios.size()) + " anchors per cell but the model's "
                         "heads carry " + std::to_string(plan.anchors_per_cell));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw HipError("no HIP device available");
    return std::unique_ptr<Engine>(new EngineImpl<T>(plan, std::move(pack), nms, opt, ratios));
}

// test hook (host only): build / validate the plan cache of one model without creating an engine
int plan_cache_probe(const std::string &model_dir, const EngineOptions &opt, size_t *arena_bytes) {
    Plan plan;
    bool hit = false;
    switch (opt.precision) {
        case RF_PRECISION_FP16: { WeightPack<half_t> p; prepare_pack<half_t>(model_dir, opt, &plan, &p, &hit); if (arena_bytes) *arena_bytes = p.arena_.bytes(); break; }
        case RF_PRECISION_FP32: { WeightPack<float> p; prepare_pack<float>(model_dir, opt, &plan, &p, &hit); if (arena_bytes) *arena_bytes = p.arena_.bytes(); break; }
        case RF_PRECISION_INT8: { WeightPack<int8_t> p; prepare_pack<int8_t>(model_dir, opt, &plan, &p, &hit); if (arena_bytes) *arena_bytes = p.arena_.bytes(); break; }
        default: throw ArgError("unknown precision");
    }
    return hit ? 1 : 0;
}

std::unique_ptr<Engine> Engine::create_single(const std::string &model_dir, const std::string &network, float nms,
                                              const EngineOptions &opt) {
    std::vector<float> ratios;
    (void)network_preset(network, &ratios);       // unknown names behave like "ssh": constructed, no anchors (RetinaFace.cpp:237-239)
    switch (opt.precision) {
        case RF_PRECISION_FP16: return make_engine<half_t>(model_dir, network, nms, opt, ratios);
        case RF_PRECISION_FP32: return make_engine<float>(model_dir, network, nms, opt, ratios);
        case RF_PRECISION_INT8: return make_engine<int8_t>(model_dir, network, nms, opt, ratios);
        default: throw ArgError("unknown precision");
    }
}

}  // namespace rf
