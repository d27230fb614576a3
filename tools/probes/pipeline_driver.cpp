// Probe: the bench's enqueue/wait pipeline driven from C++ instead of Python -- separates host-language overhead from the
// engine's own per-step cost.  Reads /tmp/rf_frames.raw (8 frames 448x448x3 u8, written by the python side of the probe).
// build (on the GPU box): hipcc -O2 tools/probes/pipeline_driver.cpp -Iinclude -Lretinaface_amd/lib -lretinaface_amd -Wl,-rpath,$PWD/retinaface_amd/lib -o /tmp/pipeline_driver
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "retinaface_amd.h"
int main(int argc, char **argv) {
    const int B = 8, H = 448, W = 448, steps = argc > 1 ? atoi(argv[1]) : 4000;
    std::vector<uint8_t> raw((size_t)B * H * W * 3);
    FILE *f = fopen("/tmp/rf_frames.raw", "rb");
    if (!f || fread(raw.data(), 1, raw.size(), f) != raw.size()) { printf("no frames\n"); return 1; }
    fclose(f);
    uint8_t *d = nullptr;
    if (hipMalloc(&d, raw.size()) != hipSuccess || hipMemcpy(d, raw.data(), raw.size(), hipMemcpyHostToDevice) != hipSuccess) return 2;
    rf_options o; memset(&o, 0, sizeof(o)); o.struct_size = sizeof(o); o.precision = RF_PRECISION_FP16; o.net_h = H; o.net_w = W; o.max_batch = B;
    o.model_stem = "mnet25";
    rf_handle h = nullptr;
    if (rf_create("assets", "net3", 0.4f, &o, &h) != 0) { printf("create failed: %s\n", rf_last_error(nullptr)); return 3; }
    const void *ptrs[B]; int rows[B], cols[B], st[B];
    for (int i = 0; i < B; i++) { ptrs[i] = d + (size_t)i * H * W * 3; rows[i] = H; cols[i] = W; st[i] = W * 3; }
    const int slots = rf_num_slots(h);
    std::vector<rf_face> out((size_t)B * 256); int counts[B];
    std::vector<int> inflight; size_t head = 0; long faces = 0;
    auto run = [&](int n) {
        for (int s = 0; s < n; s++) {
            if ((int)(inflight.size() - head) == slots) { rf_wait(h, inflight[head++], out.data(), 256, counts); for (int i = 0; i < B; i++) faces += counts[i]; }
            int t = -1; rf_enqueue_batch_device(h, ptrs, rows, cols, st, B, 0.5f, &t); inflight.push_back(t);
        }
        while (head < inflight.size()) { rf_wait(h, inflight[head++], out.data(), 256, counts); for (int i = 0; i < B; i++) faces += counts[i]; }
    };
    run(200); faces = 0;
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    run(steps);
    hipDeviceSynchronize();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("C++ driver: %d steps, %.4f ms/step, %.0f images/s, %.0f faces/s (slots %d)\n", steps, dt / steps * 1e3, steps * B / dt, faces / dt, slots);
    rf_destroy(h);
    return 0;
}
