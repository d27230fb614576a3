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
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace rf {

// Staging copy with NON-TEMPORAL stores (round 5).  The destination is a pinned block the CPU never reads back -- the DMA engine does -- so
// a cached store is wasted twice: every destination line is first READ into the cache (write-allocate: memory traffic 3 bytes per byte copied
// instead of 2) and then evicts something the caller's next frame wanted.  glibc's memcpy switches to streaming stores only above ~3/4 of the
// shared cache size (tens of MB on the EPYC hosts), far above the 64 KB pieces the helper threads copy.  32-byte AVX2 streams, destination
// aligned by a short head copy, source unaligned loads; the sfence at the end orders the stores before the DMA that follows.  Falls back to
// memcpy where AVX2 is missing.
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void stream_copy_avx2(uint8_t *dst, const uint8_t *src, size_t n) {
    const size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
    if (head) { const size_t h = head < n ? head : n; memcpy(dst, src, h); dst += h; src += h; n -= h; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a);
        _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c);
        _mm256_stream_si256((__m256i *)(dst + i + 96), d);
    }
    for (; i + 32 <= n; i += 32) _mm256_stream_si256((__m256i *)(dst + i), _mm256_loadu_si256((const __m256i *)(src + i)));
    if (i < n) memcpy(dst + i, src + i, n - i);
}
#endif
inline bool stream_copy_available() {
#if defined(__x86_64__)
    static const bool ok = __builtin_cpu_supports("avx2");
    return ok;
#else
    return false;
#endif
}

// Host frames reach the GPU through pinned staging memory; the copy into it is the only per-byte CPU work of the hot path and a
// single core moves ~10 GB/s, a fifth of what PCIe Gen5 takes.  A few helper threads split every enqueue's rows between them
// (the caller's thread works too), so staging runs at memory speed and the DMA engine sees ONE large copy per enqueue.
class ParallelCopier {
public:
    struct Job { uint8_t *dst; const uint8_t *src; size_t row_bytes, rows, src_step; };
    explicit ParallelCopier(int helpers, bool streaming = true) : streaming_(streaming && stream_copy_available()) {
        for (int i = 0; i < helpers; i++) threads_.emplace_back([this] { worker(); });
    }
    bool streaming() const { return streaming_; }
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
        if (threads_.empty() || pcs.size() == 1) { for (const Job &p : pcs) copy(p); fence(); return; }
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
    void copy(const Job &p) const {
#if defined(__x86_64__)
        if (streaming_) {
            if (p.src_step == p.row_bytes) { stream_copy_avx2(p.dst, p.src, p.row_bytes * p.rows); return; }
            for (size_t r = 0; r < p.rows; r++) stream_copy_avx2(p.dst + r * p.row_bytes, p.src + r * p.src_step, p.row_bytes);
            return;
        }
#endif
        if (p.src_step == p.row_bytes) { memcpy(p.dst, p.src, p.row_bytes * p.rows); return; }
        for (size_t r = 0; r < p.rows; r++) memcpy(p.dst + r * p.row_bytes, p.src + r * p.src_step, p.row_bytes);
    }
    void fence() const {                 // streaming stores are weakly ordered: make them globally visible before the thread reports its pieces done
#if defined(__x86_64__)
        if (streaming_) _mm_sfence();
#endif
    }
    void drain() {                       // pieces_ only changes while no helper is in here (busy_ == 0, under mu_)
        size_t done = 0;
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= pieces_.size()) break;
            copy(pieces_[i]);
            done++;
        }
        fence();
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
    const bool streaming_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_, idle_cv_;
};

}  // namespace rf
