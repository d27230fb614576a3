// Probe: host cost of hipPointerGetAttributes on a device pointer (the per-frame residency check of rf_*_batch_device when the
// process sees more than one GPU).  build: hipcc -O2 -o tools/probes/ptr_attr.bin tools/probes/ptr_attr.cpp
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

int main() {
    void *p = nullptr, *many[64];
    if (hipMalloc(&p, 64 << 20) != hipSuccess) { printf("no device\n"); return 1; }
    for (auto &m : many) (void)hipMalloc(&m, 1 << 20);          // a populated allocation map
    hipPointerAttribute_t a;
    const int n = 1000000;
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        long acc = 0;
        for (int i = 0; i < n; i++) {
            (void)hipPointerGetAttributes(&a, (char *)p + (size_t)(i & 63) * 600000);
            acc += a.device;
        }
        double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("hipPointerGetAttributes: %.0f ns per call (device %ld)\n", ns, acc / n);
    }
    return 0;
}
