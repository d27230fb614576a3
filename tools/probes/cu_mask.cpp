// Probe: what a HIP stream created with hipExtStreamCreateWithCUMask actually gets on MI355X (256 CUs = 8 XCDs x 32): which XCDs / CUs run the
// workgroups of a kernel launched on it, for three 128-bit-set masks: the LOW 128 bits, the EVEN bits, the low 16 bits of every 32.
// Also: does a hipGraph captured on such a stream and replayed on it stay inside the mask, and how long does a fixed spin kernel take on a
// masked vs an unmasked stream (work per CU doubles when the grid is the same).  hipcc --offload-arch=gfx950 -O2 cu_mask.cpp -o cu_mask.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void where_kernel(unsigned *out, long long spin) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));        // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));         // HW_REG_HW_ID: cu [11:8], sh [12], se [15:13]
        out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
    }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}

static void report(const char *tag, const std::vector<unsigned> &v) {
    int per_xcc[16] = {};
    std::vector<char> seen(16 * 65536, 0);
    int distinct = 0;
    for (unsigned x : v) {
        const unsigned xcc = (x >> 16) & 15, cu = (x >> 8) & 15, sh = (x >> 12) & 1, se = (x >> 13) & 7;
        per_xcc[xcc]++;
        const unsigned key = (xcc << 8) | (se << 5) | (sh << 4) | cu;
        if (!seen[key]) { seen[key] = 1; distinct++; }
    }
    printf("%-34s distinct (xcc, se, sh, cu) = %3d | workgroups per XCC:", tag, distinct);
    for (int i = 0; i < 8; i++) printf(" %4d", per_xcc[i]);
    printf("\n");
}

int main() {
    const int nblk = 4096;
    unsigned *d; hipMalloc(&d, nblk * 4);
    std::vector<unsigned> h(nblk);
    struct M { const char *name; uint32_t w[8]; };
    M masks[4] = {{"no mask", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
                  {"low 128 bits", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
                  {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
                  {"low 16 of every 32", {0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu}}};
    for (auto &m : masks) {
        hipStream_t st;
        hipError_t e = strcmp(m.name, "no mask") ? hipExtStreamCreateWithCUMask(&st, 8, m.w) : hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) { printf("%s: stream creation failed: %s\n", m.name, hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0xff, nblk * 4, st);
        hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, st, d, 200);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, nblk * 4, hipMemcpyDeviceToHost);
        char tag[64]; snprintf(tag, sizeof tag, "%s, eager:", m.name);
        report(tag, h);
        // the same through a captured graph
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, st, d, 200);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, nblk * 4, hipMemcpyDeviceToHost);
        snprintf(tag, sizeof tag, "%s, graph replay:", m.name);
        report(tag, h);
        // time: 2048 workgroups x 20 us spin (8 per CU on 256 CUs = one round)
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 20; r++) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        printf("%-34s 20 replays of 4096 x 2 us workgroups: %.1f us each\n", m.name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20);
        hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
    }
    // two masked streams with complementary halves running concurrently
    for (int pat = 1; pat <= 3; pat++) {
        uint32_t a[8], b[8];
        for (int i = 0; i < 8; i++) { a[i] = masks[pat].w[i]; b[i] = ~masks[pat].w[i]; }
        hipStream_t sa, sb;
        if (hipExtStreamCreateWithCUMask(&sa, 8, a) != hipSuccess || hipExtStreamCreateWithCUMask(&sb, 8, b) != hipSuccess) continue;
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 20; r++) {
            hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, sa, d, 200);
            hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, sb, d, 200);
        }
        hipStreamSynchronize(sa); hipStreamSynchronize(sb);
        printf("complementary halves (%s | rest), 20 + 20 launches concurrently: %.1f us per pair\n", masks[pat].name,
               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20);
        hipStreamDestroy(sa); hipStreamDestroy(sb);
    }
    return 0;
}
