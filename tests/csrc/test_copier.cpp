// Host unit test of retinaface_amd/csrc/copier.h (the engine's staging copy) and of pack.h's shard rule.
//   g++ -O1 -std=c++17 -pthread -o test_copier tests/csrc/test_copier.cpp && ./test_copier
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../retinaface_amd/csrc/copier.h"
#include "../../retinaface_amd/csrc/pack.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main() {
    std::mt19937 rng(7);
    for (int mode = 0; mode < 2; mode++)
    for (int helpers : {0, 1, 3, 7}) {
        rf::ParallelCopier pc(helpers, mode == 1);            // plain memcpy, then the non-temporal (AVX2 streaming-store) copy
        if (mode == 1 && helpers == 0) std::printf("streaming copy %s\n", pc.streaming() ? "on" : "unavailable on this CPU (memcpy)");
        for (int round = 0; round < 40; round++) {              // many rounds back to back: late-waking helpers must not race the next one
            const int nframes = 1 + (int)(rng() % 8);
            std::vector<std::vector<uint8_t>> src(nframes);
            std::vector<rf::ParallelCopier::Job> jobs;
            std::vector<size_t> off(nframes);
            size_t total = 0;
            struct Shape { size_t rows, cols3, step; };
            std::vector<Shape> sh(nframes);
            for (int i = 0; i < nframes; i++) {
                const size_t rows = (rng() % 5 == 0) ? 0 : 1 + rng() % 300, cols3 = 3 * (1 + rng() % 500), step = cols3 + (rng() % 3 ? 0 : rng() % 64);
                sh[i] = {rows, cols3, step};
                src[i].resize(rows * step + 1);
                for (auto &b : src[i]) b = (uint8_t)rng();
                off[i] = total;
                total += (rows * cols3 + 255) / 256 * 256;
            }
            std::vector<uint8_t> dst(total + 1, 0xEE);
            for (int i = 0; i < nframes; i++)
                jobs.push_back(rf::ParallelCopier::Job{dst.data() + off[i], src[i].data(), sh[i].cols3, sh[i].rows, sh[i].step});
            pc.run(jobs);
            for (int i = 0; i < nframes; i++)
                for (size_t r = 0; r < sh[i].rows; r++)
                    CHECK(std::memcmp(dst.data() + off[i] + r * sh[i].cols3, src[i].data() + r * sh[i].step, sh[i].cols3) == 0);
            CHECK(dst[total] == 0xEE);
        }
        pc.run({});                                             // empty job list is legal
    }
    // shard rule: contiguous, covering, at most ceil(n / G) per device, identical to shard.py's shard_range
    for (int n : {0, 1, 7, 8, 9, 255, 256, 257})
        for (int G : {1, 2, 3, 4, 8}) {
            CHECK(rf::shard_begin(n, G, 0) == 0 && rf::shard_begin(n, G, G) == n);
            for (int g = 0; g < G; g++) {
                const int lo = rf::shard_begin(n, G, g), hi = rf::shard_begin(n, G, g + 1);
                CHECK(lo <= hi && hi - lo <= (n + G - 1) / G);
            }
        }
    std::printf(fails ? "FAILED (%d)\n" : "ok\n", fails);
    return fails ? 1 : 0;
}
