// Probe: semantics of `buffer_load_dwordx4 ... offen lds` (LDS-DMA, __builtin_amdgcn_raw_ptr_buffer_load_lds, 16 B per lane) on gfx950 --
// what the chained depthwise/pointwise kernel's staging relies on:
//   (1) lane l of a wave-instruction lands at  M0 base + 16 * l  (lane-linear, 1 KiB per instruction);
//   (2) a lane whose buffer offset is out of range writes ZEROS (hardware range check = zero padding of the halo);
//   (3) a lane that is masked off (exec) leaves its 16 LDS bytes untouched;
//   (4) the data is visible to every wave of the workgroup after s_waitcnt vmcnt(0) + s_barrier;
//   (5) timing: 8 DMA pieces per wave back to back.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/probes/lds_dma.bin tools/probes/lds_dma.cpp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

constexpr int kPieces = 8;                       // per wave
constexpr unsigned kOob = 0x80000000u;

__global__ __launch_bounds__(256) void k(const unsigned char *src, unsigned bytes, u32x4 *out, const int *mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pre-fill LDS with a pattern so untouched bytes are recognisable
    for (int i = tid; i < 4 * kPieces * 64; i += 256) ((u32x4 *)smem)[i] = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
    __syncthreads();
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)bytes, 0x00020000);
    const int m = *mode;
    for (int i = 0; i < kPieces; i++) {
        const int slot = (wave * kPieces + i) * 64 + lane;
        unsigned off = (unsigned)slot * 16u;
        if (m == 1 && (slot % 5) == 0) off = kOob;                          // (2) out-of-range lanes
        if (m == 3 && (slot % 5) == 0) off = (unsigned)(-16 - 16 * (slot % 7));   // negative offsets (rows above the image)
        const bool on = !(m == 2 && (slot % 3) == 0);                       // (3) masked-off lanes
        if (on)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (size_t)(wave * kPieces + i) * 1024), 16, (int)off, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    __syncthreads();
    // every thread reads a slot written by ANOTHER wave
    for (int i = tid; i < 4 * kPieces * 64; i += 256) {
        const int j = (i + 64 * kPieces) % (4 * kPieces * 64);
        out[j] = ((const u32x4 *)smem)[j];
    }
}

int main() {
    const int slots = 4 * kPieces * 64, bytes = slots * 16;
    std::vector<unsigned> h(slots * 4);
    for (int i = 0; i < slots * 4; i++) h[i] = 0x10000000u + (unsigned)i;
    unsigned char *dsrc; u32x4 *dout; int *dmode;
    if (hipMalloc(&dsrc, bytes) != hipSuccess) { printf("no device\n"); return 1; }
    (void)hipMalloc(&dout, bytes);
    (void)hipMalloc(&dmode, 4);
    (void)hipMemcpy(dsrc, h.data(), bytes, hipMemcpyHostToDevice);
    int total_bad = 0;
    for (int mode = 0; mode < 4; mode++) {
        (void)hipMemcpy(dmode, &mode, 4, hipMemcpyHostToDevice);
        (void)hipMemset(dout, 0xff, bytes);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), bytes, 0, dsrc, (unsigned)bytes, dout, dmode);
        if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d: launch failed\n", mode); return 2; }
        std::vector<unsigned> r(slots * 4);
        (void)hipMemcpy(r.data(), dout, bytes, hipMemcpyDeviceToHost);
        int bad = 0, zeros = 0, kept = 0;
        for (int s = 0; s < slots; s++)
            for (int d = 0; d < 4; d++) {
                unsigned want = h[s * 4 + d];
                if ((mode == 1 || mode == 3) && s % 5 == 0) { want = 0u; zeros++; }
                if (mode == 2 && s % 3 == 0) { want = 0xdeadbeefu; kept++; }
                if (r[s * 4 + d] != want) { if (bad < 6) printf("  mode %d slot %d dword %d: got %08x want %08x\n", mode, s, d, r[s * 4 + d], want); bad++; }
            }
        printf("mode %d (%s): %d mismatching dwords of %d (zero-expected %d, untouched-expected %d)\n", mode,
               mode == 0 ? "lane-linear placement + cross-wave visibility" : mode == 1 ? "out-of-range offset -> zeros" :
               mode == 2 ? "masked-off lanes leave LDS untouched" : "negative offsets -> zeros", bad, slots * 4, zeros, kept);
        total_bad += bad;
    }
    printf("LDS-DMA probe: %s\n", total_bad ? "FAILED" : "all semantics as assumed");
    return total_bad ? 3 : 0;
}
