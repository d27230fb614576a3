// multi.cpp -- image sharding of detectBatchImages() over the GPUs of one node, inside one handle.
//
// The call being sharded is RetinaFace::detectBatchImages (RetinaFace.cpp:749-940): images are independent end to end --
// preprocess and forward are per image, NMS runs per image (:916-918) -- and the weights are 0.9 MB, so every device gets a full
// engine and a contiguous slice of ceil(n / G) images; there is no data-path collective.  One host thread per
// device drives its engine, so the H2D staging, the launches and the result harvest of the G slices overlap; results land
// directly in the caller's arrays at the slice's offset.  rf_options.devices may name the same ordinal more than once (two
// engines sharing a GPU): that is how the sharding logic is tested on a one-GPU box.
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>

#include "engine.h"
#include "pack.h"

namespace rf {

namespace {

// one persistent host thread per device: run(job) hands it a closure, join() waits for it and rethrows what it threw
class Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        th_.join();
    }
    void run(std::function<void()> job) {
        std::lock_guard<std::mutex> lk(mu_);
        job_ = std::move(job);
        pending_ = true;
        err_ = nullptr;
        cv_.notify_all();
    }
    void join() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !pending_; });
        if (err_) { std::exception_ptr e = err_; err_ = nullptr; std::rethrow_exception(e); }
    }
private:
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || (pending_ && job_); });
                if (stop_) return;
                job = std::move(job_);
                job_ = nullptr;
            }
            std::exception_ptr err;
            try { job(); } catch (...) { err = std::current_exception(); }
            std::lock_guard<std::mutex> lk(mu_);
            err_ = err;
            pending_ = false;
            cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool pending_ = false, stop_ = false;
    std::exception_ptr err_;
    std::thread th_;
};

class MultiEngine final : public Engine {
public:
    MultiEngine(const std::string &model_dir, const std::string &network, float nms, const EngineOptions &opt) {
        opt_ = opt;
        nms_threshold_ = nms;
        for (int dev : opt.devices) {
            EngineOptions eo = opt;
            eo.devices.clear();
            eo.device = dev;
            eng_.push_back(Engine::create_single(model_dir, network, nms, eo));
        }
        net_h_ = eng_[0]->net_h();
        net_w_ = eng_[0]->net_w();
        workers_.reserve(eng_.size());
        for (size_t g = 0; g < eng_.size(); g++) workers_.emplace_back(new Worker);
        lo_.assign(eng_.size() + 1, 0);
        last_shard_ = lo_;
        issued_.resize(eng_.size());
    }
    ~MultiEngine() override {
        workers_.clear();          // threads first: nothing may be running when the engines go
        eng_.clear();
    }

    int num_devices() const override { return (int)eng_.size(); }

    // contiguous slices of ceil(n / G) images; trailing devices may get fewer or none (shard_range in retinaface_amd/shard.py)
    static void shard(int n, int G, std::vector<int> &lo) {
        lo.resize(G + 1);
        for (int g = 0; g <= G; g++) lo[g] = shard_begin(n, G, g);
    }

    void detect(const uint8_t *const *frames, const int *rows, const int *cols, const int *steps, int n, bool on_device,
                float threshold, rf_face *out, int cap_per_image, int *counts, bool *truncated) override {
        if (n < 0 || (n > 0 && (!frames || !rows || !cols || !counts))) throw ArgError("null argument");
        if (cap_per_image < 0 || (cap_per_image > 0 && !out)) throw ArgError("out is null");
        const int G = (int)eng_.size();
        shard(n, G, lo_);
        std::vector<char> tr(G, 0);
        std::vector<std::vector<int>> cand(G);
        std::vector<std::vector<std::vector<int32_t>>> anchors(G);
        for (int g = 0; g < G; g++) {
            const int lo = lo_[g], m = lo_[g + 1] - lo_[g];
            if (m == 0) continue;
            workers_[g]->run([=, &tr, &cand, &anchors] {
                bool t = false;
                Engine &e = *eng_[g];
                e.detect(frames + lo, rows + lo, cols + lo, steps ? steps + lo : nullptr, m, on_device, threshold,
                         out ? out + (size_t)lo * cap_per_image : nullptr, cap_per_image, counts + lo, &t);
                tr[g] = t;
                cand[g].resize(m);
                e.last_candidate_counts(cand[g].data(), m);
                anchors[g].resize(m);
                for (int i = 0; i < m; i++) {
                    std::vector<int32_t> &a = anchors[g][i];
                    a.resize(opt_.max_detections);
                    const int k = e.last_anchor_indices(i, a.data(), (int)a.size());
                    a.resize(std::min(k, (int)a.size()));
                }
            });
        }
        std::exception_ptr first;
        for (int g = 0; g < G; g++) {
            if (lo_[g + 1] == lo_[g]) continue;
            try { workers_[g]->join(); } catch (...) { if (!first) first = std::current_exception(); }
        }
        if (first) std::rethrow_exception(first);
        *truncated = false;
        last_cand_.clear();
        last_anchor_.clear();
        for (int g = 0; g < G; g++) {
            *truncated = *truncated || tr[g];
            last_cand_.insert(last_cand_.end(), cand[g].begin(), cand[g].end());
            for (auto &a : anchors[g]) last_anchor_.push_back(std::move(a));
        }
        last_n_ = n;
        last_from_wait_ = -1;
        last_shard_ = lo_;                                   // what get_output / debug_activation locate images by: lo_ is scratch of the next call
        for (int g = 0; g < G; g++)
            if (lo_[g + 1] > lo_[g]) last_engine_ = g;
    }

    // the asynchronous API spreads whole enqueues round-robin over the devices (an enqueue is at most max_batch images: too
    // small to split); ticket = device slot + G * the engine's own ticket
    int enqueue(const void *const *fr, const int *rows, const int *cols, const int *steps, int n, bool on_device,
                float threshold) override {
        const int G = (int)eng_.size();
        const int g = next_;
        const int t = eng_[g]->enqueue(fr, rows, cols, steps, n, on_device, threshold);     // throws before any state changes
        next_ = (next_ + 1) % G;
        if ((int)issued_[g].size() <= t) issued_[g].resize(t + 1, 0);
        issued_[g][t] = 1;
        return g + G * t;
    }
    void wait(int ticket, rf_face *out, int cap_per_image, int *counts, bool *truncated) override {
        const int G = (int)eng_.size();
        // a ticket is only forwarded to the engine that issued it and has not been waited for yet: a stale or foreign number
        // must not reach another engine's slot of the same index
        if (ticket < 0 || ticket / G >= (int)issued_[ticket % G].size() || !issued_[ticket % G][ticket / G])
            throw ArgError("wait: invalid ticket");
        // the engine rejects a caller mistake (cap_per_image > 0 without an output array ...) WITHOUT consuming the ticket: it stays
        // issued here too, so the caller can wait again with good arguments (clearing it first leaked the slot: ADVICE r3)
        eng_[ticket % G]->wait(ticket / G, out, cap_per_image, counts, truncated);
        issued_[ticket % G][ticket / G] = 0;
        last_from_wait_ = ticket % G;
        last_engine_ = ticket % G;
    }
    int num_slots() const override {
        int s = 0;
        for (auto &e : eng_) s += e->num_slots();
        return s;
    }
    void host_register(const void *ptr, size_t bytes) override {
        eng_[0]->host_register(ptr, bytes);                       // pins the range (portable: valid for every device)
        for (size_t g = 1; g < eng_.size(); g++) eng_[g]->host_adopt(ptr, bytes);
    }
    void host_adopt(const void *ptr, size_t bytes) override {
        for (auto &e : eng_) e->host_adopt(ptr, bytes);
    }
    void host_unregister(const void *ptr) override {
        for (size_t g = eng_.size(); g-- > 1;) eng_[g]->host_forget(ptr);
        eng_[0]->host_unregister(ptr);
    }
    void host_forget(const void *ptr) override {
        for (auto &e : eng_) e->host_forget(ptr);
    }
    void invalidate_residency() override {
        for (auto &e : eng_) e->invalidate_residency();
    }
    void scatter_stats(long long *frames, long long *copies) const override {
        for (auto &e : eng_) e->scatter_stats(frames, copies);
    }

    int last_anchor_indices(int image, int32_t *out, int cap) const override {
        if (last_from_wait_ >= 0) return eng_[last_from_wait_]->last_anchor_indices(image, out, cap);
        if (image < 0 || image >= last_n_) throw ArgError("image index out of range");
        const int k = (int)last_anchor_[image].size();
        for (int i = 0; i < std::min(k, cap); i++) out[i] = last_anchor_[image][i];
        return k;
    }
    int last_candidate_counts(int *counts, int n) const override {
        if (last_from_wait_ >= 0) return eng_[last_from_wait_]->last_candidate_counts(counts, n);
        for (int i = 0; i < std::min(n, last_n_); i++) counts[i] = last_cand_[i];
        return last_n_;
    }
    // the engine that served the most recent call (a sharded detect: the one whose slice held the call's last image)
    void last_timings(float *pre, float *infer, float *post, float *total) const override {
        eng_[last_engine_]->last_timings(pre, infer, post, total);
    }
    // per-launch accessors: image i of the last sharded call lives on the device whose slice holds it
    long get_output(const std::string &blob, int image, float *dst, size_t cap) override {
        int g = 0, local = image;
        locate(image, &g, &local);
        return eng_[g]->get_output(blob, local, dst, cap);
    }
    long debug_activation(const std::string &blob, int image, float *dst, size_t cap, int dims[3]) override {
        int g = 0, local = image;
        locate(image, &g, &local);
        return eng_[g]->debug_activation(blob, local, dst, cap, dims);
    }
    int profile(const void *const *d_frames, int n, int iters, int cap, const char **names, const char **kernels,
                float *avg_ms, double *alg_bytes, double *macs) override {
        return eng_[0]->profile(d_frames, n, iters, cap, names, kernels, avg_ms, alg_bytes, macs);
    }
    int compulsory_bytes(int n, int cap, double *bytes) override { return eng_[0]->compulsory_bytes(n, cap, bytes); }

private:
    void locate(int image, int *g, int *local) const {
        if (last_from_wait_ >= 0) { *g = last_from_wait_; *local = image; return; }
        if (image < 0 || image >= last_n_) throw ArgError("image index out of range");
        for (int k = 0; k + 1 < (int)last_shard_.size(); k++)
            if (image >= last_shard_[k] && image < last_shard_[k + 1]) { *g = k; *local = image - last_shard_[k]; return; }
        throw ArgError("image index out of range");
    }

    std::vector<std::unique_ptr<Engine>> eng_;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::vector<int> lo_, last_shard_;
    std::vector<std::vector<char>> issued_;              // per engine: its tickets that are outstanding
    int next_ = 0, last_n_ = 0, last_from_wait_ = -1, last_engine_ = 0;
    std::vector<int> last_cand_;
    std::vector<std::vector<int32_t>> last_anchor_;
};

}  // namespace

std::unique_ptr<Engine> Engine::create(const std::string &model_dir, const std::string &network, float nms,
                                       const EngineOptions &opt) {
    if (opt.devices.size() <= 1) {
        EngineOptions eo = opt;
        if (opt.devices.size() == 1) eo.device = opt.devices[0];
        eo.devices.clear();
        return create_single(model_dir, network, nms, eo);
    }
    if ((int)opt.devices.size() > kMaxDevices) throw ArgError("at most " + std::to_string(kMaxDevices) + " device entries");
    return std::unique_ptr<Engine>(new MultiEngine(model_dir, network, nms, opt));
}

}  // namespace rf
