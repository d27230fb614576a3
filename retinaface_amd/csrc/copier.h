// copier.h -- host-only helper of the engine: rows of host frames -> pinned staging memory on several threads.
// (Header-only so tests/csrc/test_copier.cpp exercises exactly the code the engine runs.)
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace rf {

// Host frames reach the GPU through pinned staging memory; the copy into it is the only per-byte CPU work of the hot path and a
// single core moves ~10 GB/s, a fifth of what PCIe Gen5 takes.  A few helper threads split every enqueue's rows between them
// (the caller's thread works too), so staging runs at memory speed and the DMA engine sees ONE large copy per enqueue.
class ParallelCopier {
public:
    struct Job { uint8_t *dst; const uint8_t *src; size_t row_bytes, rows, src_step; };
    explicit ParallelCopier(int helpers) {
        for (int i = 0; i < helpers; i++) threads_.emplace_back([this] { worker(); });
    }
    ~ParallelCopier() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void run(const std::vector<Job> &jobs) {
        // pieces of ~64 KB (whole rows of one frame): several per thread, so the threads finish together
        std::vector<Job> pcs;
        for (const Job &j : jobs) {
            if (!j.rows || !j.row_bytes) continue;
            const size_t per = std::max<size_t>(1, (64 << 10) / j.row_bytes);
            for (size_t r = 0; r < j.rows; r += per)
                pcs.push_back(Job{j.dst + r * j.row_bytes, j.src + r * j.src_step, j.row_bytes, std::min(per, j.rows - r), j.src_step});
        }
        if (pcs.empty()) return;
        if (threads_.empty() || pcs.size() == 1) { for (const Job &p : pcs) copy(p); return; }
        {
            std::unique_lock<std::mutex> lk(mu_);
            idle_cv_.wait(lk, [this] { return busy_ == 0; });      // a helper that woke late for the previous round has left drain()
            pieces_.swap(pcs);
            next_.store(0);
            left_ = pieces_.size();
            gen_++;
            gen_pub_.store(gen_, std::memory_order_release);
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return left_ == 0; });
    }
private:
    static void copy(const Job &p) {
        if (p.src_step == p.row_bytes) { memcpy(p.dst, p.src, p.row_bytes * p.rows); return; }
        for (size_t r = 0; r < p.rows; r++) memcpy(p.dst + r * p.row_bytes, p.src + r * p.src_step, p.row_bytes);
    }
    void drain() {                       // pieces_ only changes while no helper is in here (busy_ == 0, under mu_)
        size_t done = 0;
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= pieces_.size()) break;
            copy(pieces_[i]);
            done++;
        }
        if (done) {
            std::lock_guard<std::mutex> lk(mu_);
            left_ -= done;
            if (left_ == 0) done_cv_.notify_all();
        }
    }
    void worker() {
        unsigned long seen = 0;
        for (;;) {
            // a serving loop enqueues every ~100 us: spin briefly for the next round before sleeping (a condvar wake-up alone
            // costs tens of microseconds, a third of a 4.8 MB staging copy)
            for (int spin = 0; spin < 4000 && gen_pub_.load(std::memory_order_acquire) == seen; spin++) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
            }
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                busy_++;
            }
            drain();
            std::lock_guard<std::mutex> lk(mu_);
            if (--busy_ == 0) idle_cv_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::vector<Job> pieces_;
    std::atomic<size_t> next_{0};
    size_t left_ = 0;
    int busy_ = 0;
    unsigned long gen_ = 0;
    std::atomic<unsigned long> gen_pub_{0};      // copy of gen_ the helpers may poll without the lock
    bool stop_ = false;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_, idle_cv_;
};

}  // namespace rf
