// Probe: what ONE dependent kernel of a hipGraph chain costs on this MI355X / ROCm build when it does (a) nothing, (b) one dependent
// global round trip (load a value its predecessor wrote, store one for its successor) -- the floor under the 17 launches of one synchronous
// detect() call (tools/probes/sync_gaps.py measured 4.8-13 us per kernel at batch 8 with 0.4 us gaps between them).
// A chain of 17 kernels with 49 workgroups x 256 threads each (the 128-channel block at batch 8), captured once, replayed 2000 times.
// hipcc --offload-arch=gfx950 -O2 launch_floor.cpp -o launch_floor.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void empty_kernel(int *) {}
__global__ void hop_kernel(const float *in, float *out, int n) {        // reads what the previous kernel wrote, writes for the next
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 1.0001f + 1.f;
}
__global__ void hop3_kernel(const float *in, const float *w, float *out, int n) {     // weights + input -> LDS -> barrier -> compute -> store (a tile body's skeleton)
    __shared__ float s[256];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a = i < n ? in[i] : 0.f, b = w[threadIdx.x];
    s[threadIdx.x] = a * b;
    __syncthreads();
    float c = s[(threadIdx.x + 17) & 255];
    __syncthreads();
    s[threadIdx.x] = c + a;
    __syncthreads();
    if (i < n) out[i] = s[(threadIdx.x + 31) & 255];
}

template <typename F> static double chain_us(hipStream_t st, int links, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < links; k++) launch(k);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 50; w++) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    const int reps = 2000;
    double best = 1e30;
    for (int trial = 0; trial < 3; trial++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }      // one synchronous "call" at a time
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        if (us < best) best = us;
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best;
}

int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int n = 49 * 256;
    float *a, *b, *w; int *d;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&w, 1024); hipMalloc(&d, 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(w, 0, 1024);
    for (int links : {1, 17}) {
        double e = chain_us(st, links, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(49), dim3(256), 0, st, d); });
        double h = chain_us(st, links, [&](int k) { hipLaunchKernelGGL(hop_kernel, dim3(49), dim3(256), 0, st, k & 1 ? b : a, k & 1 ? a : b, n); });
        double h3 = chain_us(st, links, [&](int k) { hipLaunchKernelGGL(hop3_kernel, dim3(49), dim3(256), 0, st, k & 1 ? b : a, w, k & 1 ? a : b, n); });
        printf("graph of %2d dependent kernels (49 x 256 threads), one synchronous replay at a time: empty %.1f us, one global hop each %.1f us, "
               "load+LDS+3 barriers+store each %.1f us\n", links, e, h, h3);
        if (links == 17) printf("per kernel in the chain of 17 (minus the 1-kernel replay): empty %.2f us, hop %.2f us, tile skeleton %.2f us\n",
                                 e / 17, h / 17, h3 / 17);
    }
    return 0;
}
