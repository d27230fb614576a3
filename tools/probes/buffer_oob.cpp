// Probe: range-check semantics of raw buffer loads/stores on gfx950 (decides how the kernels may use descriptors for padding).
//   1. a dwordx4 load that straddles num_records: are the in-range dwords returned (per-dword check) or is the whole load 0?
//   2. negative (wrapped) voffset -> 0;  3. is soffset included in the range check?  4. OOB store dropped?
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_oob.cpp -o /tmp/buffer_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned *buf, int nbytes, unsigned *out) {
    auto r = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, nbytes, 0x00020000);
    const int t = threadIdx.x;
    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, nbytes - 8, 0, 0);          // last 2 dwords in range, 2 beyond
    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, -16, 0, 0);                 // wrapped negative offset
    u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(r, 0, nbytes - 8, 0);          // same address as a, via soffset
    u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(r, nbytes - 6, 0, 0);          // unaligned?  (dword-aligned required: expect garbage/0)
    if (t == 0) {
        for (int i = 0; i < 4; i++) { out[i] = a[i]; out[4 + i] = b[i]; out[8 + i] = c[i]; out[12 + i] = d[i]; }
        u32x4 v = {0xdead0001u, 0xdead0002u, 0xdead0003u, 0xdead0004u};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, nbytes - 8, 0, 0);            // straddling store: which dwords land?
        __builtin_amdgcn_raw_buffer_store_b128(v, r, nbytes + 64, 0, 0);           // fully OOB store
    }
}
int main() {
    const int n = 64;                      // dwords in range; the allocation is larger so we can see what lands beyond
    unsigned *buf, *out, h[128], ho[16];
    hipMalloc(&buf, 128 * 4); hipMalloc(&out, 16 * 4);
    for (int i = 0; i < 128; i++) h[i] = 0x1000 + i;
    hipMemcpy(buf, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, n * 4, out);
    hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost); hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
    printf("straddling load   : %x %x %x %x   (in-range values would be %x %x, beyond %x %x)\n", ho[0], ho[1], ho[2], ho[3], 0x1000 + n - 2, 0x1000 + n - 1, 0x1000 + n, 0x1000 + n + 1);
    printf("negative offset   : %x %x %x %x\n", ho[4], ho[5], ho[6], ho[7]);
    printf("same via soffset  : %x %x %x %x\n", ho[8], ho[9], ho[10], ho[11]);
    printf("offset n-6 (unal.): %x %x %x %x\n", ho[12], ho[13], ho[14], ho[15]);
    printf("after stores      : buf[n-2..n+1] = %x %x %x %x ; buf[n+16..n+19] = %x %x %x %x\n", h[n - 2], h[n - 1], h[n], h[n + 1], h[n + 16], h[n + 17], h[n + 18], h[n + 19]);
    return 0;
}
