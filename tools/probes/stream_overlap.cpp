// Probe: do small kernels from different HIP streams overlap on this MI355X / ROCm build?
// Each kernel = 32 workgroups spinning ~20 us.  Prints wall time for 64 launches on 1, 2, 4 streams (eager) and
// for a 4-branch hipGraph.  hipcc --offload-arch=gfx950 -O2 stream_overlap.cpp -o stream_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(long long cycles, int *sink) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}

int main() {
    int *sink;
    hipMalloc(&sink, 4);
    const int NS = 4;
    hipStream_t st[NS];
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    long long cyc = 2000;   // wall_clock64 ticks at 100 MHz -> 20 us
    for (int w = 0; w < 3; w++) { hipLaunchKernelGGL(spin, dim3(32), dim3(256), 0, st[0], cyc, sink); }
    hipDeviceSynchronize();
    for (int ns : {1, 2, 4}) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 64; i++) hipLaunchKernelGGL(spin, dim3(32), dim3(256), 0, st[i % ns], cyc, sink);
        hipDeviceSynchronize();
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("eager: 64 x 20us kernels on %d stream(s): %.1f us (serial would be >= 1280)\n", ns, us);
    }
    // graph with 4 parallel branches of 16 kernels each
    hipGraph_t g; hipGraphExec_t ge;
    hipEvent_t ev[8];
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal);
    for (int b = 1; b < NS; b++) { hipEventRecord(ev[b], st[0]); hipStreamWaitEvent(st[b], ev[b], 0); }
    for (int i = 0; i < 64; i++) hipLaunchKernelGGL(spin, dim3(32), dim3(256), 0, st[i % NS], cyc, sink);
    for (int b = 1; b < NS; b++) { hipEventRecord(ev[4 + b], st[b]); hipStreamWaitEvent(st[0], ev[4 + b], 0); }
    hipStreamEndCapture(st[0], &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        hipGraphLaunch(ge, st[0]);
        hipStreamSynchronize(st[0]);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("graph: 4 branches x 16 x 20us kernels: %.1f us (serial would be >= 1280, ideal 320)\n", us);
    }
    // 4 single-chain graphs launched on 4 streams
    hipGraph_t g1; hipGraphExec_t ge1[NS];
    hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 16; i++) hipLaunchKernelGGL(spin, dim3(32), dim3(256), 0, st[0], cyc, sink);
    hipStreamEndCapture(st[0], &g1);
    for (auto &e : ge1) hipGraphInstantiate(&e, g1, nullptr, nullptr, 0);
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int b = 0; b < NS; b++) hipGraphLaunch(ge1[b], st[b]);
        hipDeviceSynchronize();
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("4 chain-graphs (16 x 20us) on 4 streams: %.1f us (serial >= 1280, ideal 320)\n", us);
    }
    return 0;
}
