// Probe: rounding of v_cvt_pk_u8_f32 on gfx950 -- every tie k + 0.5 (k = 0..254), values just below / above, negatives, > 255, against rintf + clamp.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/probes/cvt_pk_u8.bin tools/probes/cvt_pk_u8.cpp
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const float *x, unsigned *y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
}

int main() {
    std::vector<float> h;
    for (int kk = 0; kk < 256; kk++) {
        h.push_back(kk + 0.5f);
        h.push_back(std::nextafterf(kk + 0.5f, 0.f));
        h.push_back(std::nextafterf(kk + 0.5f, 1000.f));
        h.push_back((float)kk);
        h.push_back(kk + 0.25f);
        h.push_back(kk + 0.75f);
    }
    for (float v : {-0.4f, -0.5f, -0.6f, -3.f, 255.4f, 255.5f, 256.f, 300.f, 1e9f}) h.push_back(v);
    const int n = (int)h.size();
    float *dx; unsigned *dy;
    if (hipMalloc(&dx, n * 4) != hipSuccess) { printf("no device\n"); return 1; }
    (void)hipMalloc(&dy, n * 4);
    (void)hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, n);
    std::vector<unsigned> r(n);
    (void)hipMemcpy(r.data(), dy, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        float c = std::fmin(std::fmax(std::rint(h[i]), 0.f), 255.f);
        if (r[i] != (unsigned)c) { if (bad < 10) printf("x = %.9g: hw %u, rint+clamp %u\n", h[i], r[i], (unsigned)c); bad++; }
    }
    printf("v_cvt_pk_u8_f32 vs clamp(rint(x), 0, 255) on %d values incl. all 256 ties: %d mismatches\n", n, bad);
    return 0;
}
