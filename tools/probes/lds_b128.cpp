// Probe: cost of ds_read_b128 with the B-fragment access pattern of the implicit-GEMM kernels (lane -> pixel = lane & 15,
// k-group = lane >> 4; address = pixel_index * stride + kgroup * 16 B) for different pixel strides and tile widths.
// One wave per workgroup, batches of 8 independent reads (throughput, not latency), cycles by s_memtime.  build+run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_b128.cpp -o /tmp/lds_b128 && /tmp/lds_b128
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int *offs, float *sink, long long *cycles, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 16384; i += 64) ((float *)smem)[i] = (float)i;
    __syncthreads();
    const int off = offs[lane];
    f32x4 acc = {0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *(const volatile f32x4 *)(smem + off);     // 8 independent reads in flight
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    int *d_offs; float *d_sink; long long *d_cyc;
    hipMalloc(&d_offs, 64 * 4); hipMalloc(&d_sink, 64 * 4 * 4); hipMalloc(&d_cyc, 8 * 4);
    const int iters = 512;
    printf("cycles per ds_read_b128 (latency ~64 + conflict replays), B-fragment pattern lane -> (pixel = lane & 15, kgroup = lane >> 4)\n");
    printf("stride(B) :  TW=8 step1   TW=16 step1   TW=8 step2 (stride-2 depthwise)   TW=4 step1\n");
    for (int stride = 32; stride <= 336; stride += 16) {
        printf("%6d    :", stride);
        for (int cfg = 0; cfg < 4; cfg++) {
            const int tw = cfg == 1 ? 16 : (cfg == 3 ? 4 : 8), step = cfg == 2 ? 2 : 1;
            const int hc = (tw - 1) * step + 3;
            int h[64];
            for (int l = 0; l < 64; l++) {
                int p = l & 15, kg = l >> 4;
                int py = p / tw, px = p % tw;
                h[l] = ((py * step * hc + px * step) * stride + kg * 16) % 65536;
            }
            hipMemcpy(d_offs, h, sizeof(h), hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, 0, d_offs, d_sink, d_cyc, iters);
            long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
            printf("   %7.2f", (double)c / (iters * 8));
        }
        printf("\n");
    }
    return 0;
}
