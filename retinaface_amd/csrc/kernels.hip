// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the RetinaFace detect() hot path.
//
// Layout: activations NHWC (channels innermost: one pixel = one contiguous 16..512-byte run, every global
// access is a 16-byte-per-lane vector), weights pre-packed on the host in MFMA A-fragment order (pack.h).
// Every dense contraction (1x1 pointwise / lateral / head, 3x3 FPN-aggr / SSH) runs on the matrix cores as
// D[cout][pixel] += W[cout][k] * X[k][pixel] with 64-wide wavefronts; depthwise 3x3 and conv0 (Cin = 3) are
// VALU stencils over LDS-staged halo tiles.  BN/Scale/ReLU/bias/concat/eltwise/upsample/softmax/decode are
// fused into the producing or consuming kernel, so one kernel = one row of SURVEY.md App. A "fusion groups".
//
// What each kernel replaces in the reference is cited at its definition.
#include "kernels.h"

#include <atomic>

#include "knobs.h"
#include "pack.h"

// RF_PROBES (make probe -> libretinaface_amd_probe.so): the measured-and-rejected kernel variants DESIGN.md cites and the RF_* probe knobs that
// select them.  The product library is compiled WITHOUT them: every `#ifdef RF_PROBES` block below is a variant that lost its A/B measurement
// (profiles/r0N_*), kept buildable so the evidence stays reproducible (tests/test_gpu_parity.py::test_probe_knob_kernel_variants_stay_correct
// runs them through the probe build).

namespace rf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef signed char i8x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Vec;
template <> struct Vec<half_t> { static constexpr int N = 8; typedef f16x8 type; };
template <> struct Vec<float> { static constexpr int N = 4; typedef f32x4 type; };
template <> struct Vec<int8_t> { static constexpr int N = 16; typedef i8x16 type; };

// storage type -> value: fp16 / fp32 are plain conversions; int8 is round-to-nearest + saturate to [-127, 127]
// (symmetric TensorRT-style quantisation; the scale is folded into the producer's multiplier / bias on the host)
template <typename T> __device__ __forceinline__ T to_T(float v);
template <> __device__ __forceinline__ half_t to_T<half_t>(float v) { return (half_t)v; }
template <> __device__ __forceinline__ float to_T<float>(float v) { return v; }
template <> __device__ __forceinline__ int8_t to_T<int8_t>(float v) { return (int8_t)(int)fminf(fmaxf(rintf(v), -127.f), 127.f); }

// depthwise weights: same type as the activations, except int8 activations use fp32 weights (the stencil runs on the
// VALU in fp32 anyway; only the GEMMs gain from int8)
template <typename T> struct DwWeight { typedef T type; };
template <> struct DwWeight<int8_t> { typedef float type; };

template <typename T> struct Mma;
template <> struct Mma<half_t> {
    static constexpr int K = 32, KPL = 8;
    typedef f16x8 Frag;
    typedef f32x4 Acc;
    static __device__ __forceinline__ Frag zero() { Frag f; for (int e = 0; e < 8; e++) f[e] = (half_t)0; return f; }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int K = 4, KPL = 1;
    typedef float Frag;
    typedef f32x4 Acc;
    static __device__ __forceinline__ Frag zero() { return 0.f; }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

// int8: v_mfma_i32_16x16x64_i8, 16 consecutive k per lane (one 16-byte NHWC run), int32 accumulate
template <> struct Mma<int8_t> {
    static constexpr int K = 64, KPL = 16;
    typedef i32x4 Frag;
    typedef i32x4 Acc;
    static __device__ __forceinline__ Frag zero() { Frag f = {0, 0, 0, 0}; return f; }
    static __device__ __forceinline__ Acc mma(Frag a, Frag b, Acc c) {
        return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    }
};

template <typename V, int N> __device__ __forceinline__ V vzero() {
    V v;
#pragma unroll
    for (int e = 0; e < N; e++) v[e] = 0;
    return v;
}

constexpr int kThreads = 256;   // 4 wavefronts

// Phase timeline of a workgroup (probe builds only: make TRACE=1 -> libretinaface_amd_trace.so, tools/probes/phase_trace.py).
// Wave 0 of every workgroup stamps s_memtime at each phase boundary of the kernel selected by g_trace_kernel.
#ifdef RF_KERNEL_TRACE
// probe build only (make trace; tools/probes/phase_trace.py): thread 0 of every workgroup stamps s_memtime at phase
// boundaries of the launch whose tile count equals g_trace_key (tile counts identify a launch; grids are derived)
constexpr int kTraceSlots = 12, kTraceBlocks = 8192;
__device__ unsigned long long g_trace[kTraceBlocks * kTraceSlots];
__device__ int g_trace_kernel = 0;        // 2 = dwpw, 3 = conv3x3, 4 = stem2
__device__ unsigned g_trace_key = 0;      // 0 = any launch of that kernel family
#define RF_TRACE_KEY(expr) const unsigned rf_trace_key = (unsigned)(expr)
#define RF_TRACE(kid, slot)                                                                                  \
    do {                                                                                                     \
        if (g_trace_kernel == (kid) && threadIdx.x == 0 && blockIdx.x < kTraceBlocks &&                       \
            (g_trace_key == 0 || rf_trace_key == g_trace_key))                                                \
            g_trace[blockIdx.x * kTraceSlots + (slot)] = __builtin_amdgcn_s_memtime();                         \
    } while (0)
#define RF_TRACE_T(kid, slot, thread)                                                                        \
    do {                                                                                                     \
        if (g_trace_kernel == (kid) && threadIdx.x == (thread) && blockIdx.x < kTraceBlocks)                   \
            g_trace[blockIdx.x * kTraceSlots + (slot)] = __builtin_amdgcn_s_memtime();                         \
    } while (0)
#else
#define RF_TRACE_KEY(expr) do { } while (0)
#define RF_TRACE(kid, slot) do { } while (0)
#define RF_TRACE_T(kid, slot, thread) do { } while (0)
#endif

template <typename F>
static void set_max_lds(F func, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute((const void *)func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Per-device launch state.  Function attributes (dynamic-LDS limit) and occupancy are properties of (kernel, device): with more
// than one engine in a process (rf_options.devices) each device sets them up on its first launch.  The engine binds the calling
// thread's device with bind_launch_device() (engine.cpp DeviceGuard); unbound threads ask the runtime.
// (kMaxDevices: kernels.h; the engine refuses ordinals beyond it, so no two devices ever share a slot.  The caches are atomics:
// MultiEngine's per-device host threads -- and two engines on ONE device -- run these first launches concurrently; every
// writer stores the same value.)
static thread_local int t_launch_device = -1;
void bind_launch_device(int device) { t_launch_device = device; }
static int launch_device() {
    int dev = t_launch_device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    return dev >= 0 && dev < kMaxDevices ? dev : 0;
}

// Launch helpers report a shape they have no kernel instance for as rf::Unsupported (the C ABI maps it to RF_ERR_UNSUPPORTED);
// the engine validates the layer table up front, so this is a programming error -- but never an abort().
typedef Unsupported LaunchUnsupported;

// Persistent grid: as many workgroups as the chip keeps resident (CUs x RESIDENT), trimmed so every workgroup walks the
// same number of tiles (no nearly-empty last round).  Launches with fewer tiles than that get one tile per workgroup.
static std::atomic<int> g_num_cus[kMaxDevices] = {};
// a lane whose stream is confined to a subset of the CUs (hipExtStreamCreateWithCUMask, engine.cpp RF_CU_SPLIT) sizes its persistent grids for that subset
static thread_local int t_launch_cus = 0;
void bind_launch_cus(int cus) { t_launch_cus = cus; }
static int num_cus() {
    if (t_launch_cus > 0) return t_launch_cus;
    const int dev = launch_device();
    int cus = g_num_cus[dev].load(std::memory_order_relaxed);
    if (!cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        g_num_cus[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}
// Below `min_rounds` x resident tiles the hardware dispatcher's dynamic one-tile-per-workgroup schedule is at least as
// good as walking two tiles in sequence, so the grid stays one workgroup per tile.
static int persistent_grid(int tiles, int resident_per_cu) {
    const float frac = knob_grid_frac();           // probe knob RF_GRID_FRAC (1 in the product): leave a share of the resident slots to the other lanes' kernels
    if (frac < 1.f) {
        const int slots = num_cus() * (resident_per_cu > 0 ? resident_per_cu : 1);
        const int part = (int)((float)slots * frac);
        return persistent_grid_size(tiles, part > 0 ? part : 1, 0.f);
    }
    return persistent_grid_size(tiles, num_cus() * (resident_per_cu > 0 ? resident_per_cu : 1), knob_persist_min_rounds());      // (probe knob RF_PERSIST_MIN_ROUNDS; 1 measured on MI355X)
}
template <typename F> static int resident_per_cu(F kern, size_t lds_bytes) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, kThreads, lds_bytes) != hipSuccess || nb < 1) nb = 1;
    return nb;
}
// first launch of a kernel instance on the current device: raise its dynamic-LDS limit, query its residency
template <typename F> static int kernel_residency(std::atomic<int> (&cache)[kMaxDevices], F kern, size_t lds_bytes) {
    const int dev = launch_device();
    int r = cache[dev].load(std::memory_order_acquire);
    if (!r) {
        set_max_lds(kern, lds_bytes);
        r = resident_per_cu(kern, lds_bytes);
        cache[dev].store(r, std::memory_order_release);      // published only after the attribute is set
    }
    return r;
}

// =============================================================================================
// K_a  preprocess + conv0
//   reference: cudaMemset + imageROIResize8U3C (factor 1 = top-left copy) + convertBGR2RGBfloatKernel +
//   imageSplitKernel (RetinaFace.cpp:598-608, resizeconvertion.cu:46-63,165-185,279-316) and the first
//   TensorRT layer mobilenet0_conv0_fwd + BN + ReLU (prototxt :11-53).
//   One thread = one output pixel x 8 channels; the 33x33x3 u8 input patch of a 16x16 output tile is staged
//   in LDS; fp32 accumulate on raw 0..255 pixels (pre-BN magnitudes ~1e3, SURVEY.md App. A hazards).
// =============================================================================================
constexpr int C0_T = 16;                       // output tile edge
constexpr int C0_IN = 2 * C0_T + 1;            // 33 input rows / cols
constexpr int C0_LD = 26;                      // dwords fetched per staged row: 99 payload bytes + <= 3 bytes of misalignment
constexpr int C0_ROWD = 27;                    // LDS row stride in dwords (+1: rows land on different banks)

template <typename T>
__global__ __launch_bounds__(kThreads) void conv0_kernel(const FrameDesc *__restrict__ frames, T *__restrict__ out,
                                                         const float *__restrict__ w, const float *__restrict__ b,
                                                         int ho, int wo, int tiles_x, int tiles_y, int nblk) {
    __shared__ uint32_t s_in32[C0_IN * C0_ROWD];
    const uint8_t *s_in = (const uint8_t *)s_in32;
    const int tid = threadIdx.x;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int tx = bid % tiles_x;
    const int ty = (bid / tiles_x) % tiles_y;
    const int img = bid / (tiles_x * tiles_y);
    const FrameDesc fd = frames[img];
    const int iy0 = 2 * ty * C0_T - 1;
    const int bx0 = (2 * tx * C0_T - 1) * 3;          // first byte column of the patch (-3 at the left edge)
    const uintptr_t base = (uintptr_t)fd.ptr;
    const size_t row_bytes = (size_t)fd.cols * 3;
    // stage the 33 x 99-byte u8 patch with aligned 4-byte loads (the row start is arbitrary mod 4: each row keeps its
    // own misalignment, re-derived in the compute phase); bytes outside the frame are the zero canvas / zero padding
    for (int i = tid; i < C0_IN * C0_LD; i += kThreads) {
        const int r = i / C0_LD, d = i % C0_LD;
        const int iy = iy0 + r;
        uint32_t v = 0;
        if (iy >= 0 && iy < fd.rows) {
            const uintptr_t lo = base + (size_t)iy * fd.step, hi = lo + row_bytes;
            const uintptr_t a0 = ((lo + bx0) & ~(uintptr_t)3) + 4 * d;
            if (a0 >= lo && a0 + 4 <= hi) {
                v = *(const uint32_t *)a0;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (a0 + k >= lo && a0 + k < hi) v |= (uint32_t)(*(const uint8_t *)(a0 + k)) << (8 * k);
            }
        }
        s_in32[r * C0_ROWD + d] = v;
    }
    __syncthreads();
    const int py = tid / C0_T, px = tid % C0_T;
    const int oy = ty * C0_T + py, ox = tx * C0_T + px;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; o++) acc[o] = b[o];
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
        const int r = 2 * py + ky;
        const int mis = (int)((base + (size_t)(iy0 + r) * fd.step + bx0) & 3);
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            const uint8_t *p = s_in + r * (C0_ROWD * 4) + mis + (2 * px + kx) * 3;
            // frame is BGR, the network's input channel 0 is R (convertBGR2RGBfloat): net channel c = frame channel 2-c
            float v[3] = {(float)p[2], (float)p[1], (float)p[0]};
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int o = 0; o < 8; o++) acc[o] = fmaf(w[o * 27 + (ky * 3 + kx) * 3 + c], v[c], acc[o]);
        }
    }
    if (oy < ho && ox < wo) {
        T *dst = out + (((size_t)img * ho + oy) * wo + ox) * 8;
        if constexpr (sizeof(T) == 2) {
            f16x8 r;
#pragma unroll
            for (int o = 0; o < 8; o++) r[o] = (half_t)fmaxf(acc[o], 0.f);
            *(f16x8 *)dst = r;
        } else {
            f32x4 r0, r1;
#pragma unroll
            for (int o = 0; o < 4; o++) { r0[o] = fmaxf(acc[o], 0.f); r1[o] = fmaxf(acc[o + 4], 0.f); }
            *(f32x4 *)dst = r0;
            *(f32x4 *)(dst + 4) = r1;
        }
    }
}

template <typename T>
void launch_conv0(hipStream_t s, const FrameDesc *frames, T *out, const float *w, const float *b, int n, int net_h,
                  int net_w) {
    int ho = net_h / 2, wo = net_w / 2;
    int tiles_x = (wo + C0_T - 1) / C0_T, tiles_y = (ho + C0_T - 1) / C0_T;
    int nblk = n * tiles_x * tiles_y;
    hipLaunchKernelGGL(conv0_kernel<T>, dim3(nblk), dim3(kThreads), 0, s, frames, out, w, b, ho, wo, tiles_x, tiles_y, nblk);
}
template void launch_conv0<half_t>(hipStream_t, const FrameDesc *, half_t *, const float *, const float *, int, int, int);
template void launch_conv0<float>(hipStream_t, const FrameDesc *, float *, const float *, const float *, int, int, int);

// Global traffic of the persistent kernels goes through buffer descriptors (one per image, rebuilt from scalars per tile):
// 32-bit per-lane byte offsets that are constants of the thread (tile origin added with one v_add), and the hardware range
// check instead of clamps and masks -- a row above / below the image is a negative / too-large offset and reads as zero
// (stores are dropped); only the column test needs an instruction.  This took ~20 VALU + ~10 SALU per 16-byte item out of
// loops that are a few hundred instructions per tile.  Measured semantics on gfx950 (tools/probes/buffer_oob.cpp): the check is
// per dword of a multi-dword access, an offset that wraps past 2^32 is out of range, soffset is included in the check (but is
// unsigned, so the possibly negative tile origin is added into voffset), unaligned offsets work, out-of-range stores are dropped.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0x80000000u;
template <typename P> __device__ __forceinline__ auto image_rsrc(P *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
template <typename V, typename R> __device__ __forceinline__ V buf_load16(R rsrc, unsigned off) {
#ifndef RF_LOAD_CPOL
#define RF_LOAD_CPOL 0
#endif
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, RF_LOAD_CPOL));
}
template <typename V, typename R> __device__ __forceinline__ void buf_store16(R rsrc, unsigned off, V v) {
    // Cache policy of the activation stores: `nt sc1` (round 5).  An activation tile is written once and read next by ANOTHER kernel, on whichever XCD its tiles
    // land: left dirty in this XCD's write-back L2 it has to be written back at the kernel boundary (the release of a kernel's end is a buffer_wbl2 over everything
    // the kernel dirtied: 1.7 us clean, 6.5 us dirty by the guide's price list, 17 times per launch sequence).  Streaming stores spread that write-back over the
    // kernel.  Measured as whole-library builds inside one call each (-DRF_STORE_CPOL=n, tools/gpu/r5.sh cpol, profiles/r05_store_cache_policy_ab.txt), three-lane
    // pipeline / sum of the kernels repeated back to back / one synchronous batch-8 call: default write-back 312.7 k images/s / 848 us / 0.128 ms; nt 315.3 k / 798 /
    // 0.129; **nt sc1 316.3 k / 798 / 0.129**; sc1 or sc0 sc1 alone 311 k / 830 / 0.124; nt on the activation LOADS as well 310.6 k (the halo re-reads of
    // neighbouring tiles want the L2).  Cache policy only: results are bit-identical.
#ifndef RF_STORE_CPOL
#define RF_STORE_CPOL 18      // bit 1 = nt, bit 4 = sc1 (gfx940+ encoding of the builtin's aux operand)
#endif
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)off, 0, RF_STORE_CPOL);
}

// ---- LDS-DMA and the synchronisation primitives of the warp-specialised kernels (round 4)
typedef __attribute__((address_space(3))) void *lds_void_ptr;
// One LDS-DMA wave-instruction: lane l fetches 16 bytes at buffer offset off[l] and the hardware writes them to lds_base + 16 * l (the
// destination is wave-uniform base + lane-linear; out-of-range offsets land as zeros).  A plain (non-template) device function: inside a
// template the host pass, which does not know the builtin, silently drops the whole kernel instantiation.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, void *lds_base, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)lds_base, 16, (int)off, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that does NOT drain the vector-memory counter (__syncthreads() does when an LDS-DMA is in flight): LDS traffic of this
// wave is complete, then s_barrier.  Whoever needs a DMA to have landed waits for it explicitly (counted vmcnt) before calling this.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// =============================================================================================
// GEMM core shared by K_b / K_c / K_d:  acc[i][j] += W-fragment(ct_i, kc) x X-fragment(pt_j, kc)
// 4 waves split the output-channel tiles first (WN), the pixel tiles second (WP).
// =============================================================================================
template <int NT, int PT> struct WaveSplit {
    static constexpr int WN = NT % 4 == 0 ? 4 : (NT % 2 == 0 ? 2 : 1);
    static constexpr int WP = 4 / WN;
    static constexpr int NI = NT / WN;
    static constexpr int NJ = PT / WP;
    static_assert(NT % WN == 0 && PT % WP == 0 && NJ >= 1, "tile does not split over 4 waves");
};

// Weight-fragment stream of one wave: NI output-channel tiles x KCH K-chunks, read straight from L2 in MFMA A-fragment
// order (one coalesced 16 B-per-lane load per fragment).  fp16 path: software pipelined through registers -- group 0
// (<= 12 fragments) is issued at kernel entry, BEFORE the activation tile is staged, so the L2 round trip of the
// weights overlaps the HBM round trip of the activations; group g+1 is issued before group g's MFMAs.
// fp32 path (parity reference, speed irrelevant): plain loop.
template <typename T, int NI, int NJ, int KCH, int WN, int GFRAGS = 12>
struct GemmPipe {
    typedef Mma<T> M;
    typedef typename M::Frag Frag;
    static constexpr bool PIPE = sizeof(T) == 2;
    static constexpr int GMAX = GFRAGS / NI > 0 ? GFRAGS / NI : 1;      // K-chunks per prefetch group (GFRAGS fragments)
    static constexpr int G = PIPE ? (KCH < GMAX ? KCH : GMAX) : 1;
    static constexpr int NG = (KCH + G - 1) / G;
    Frag buf[PIPE ? 2 : 1][G][NI];
    const Frag *wp;

    __device__ __forceinline__ void load(int g, int slot) {
#pragma unroll
        for (int c = 0; c < G; c++) {
            const int kc = g * G + c;
            if (kc < KCH) {
#pragma unroll
                for (int i = 0; i < NI; i++) buf[slot][c][i] = wp[((i * WN) * KCH + kc) * 64];
            }
        }
    }
    __device__ __forceinline__ void init(const void *w, int wn, int lane) {
        wp = (const Frag *)w + (size_t)wn * KCH * 64 + lane;
        if constexpr (PIPE) load(0, 0);
    }
    // xf(j, kc) returns the activation (B) fragment of pixel tile j for K-chunk kc
    template <typename XF> __device__ __forceinline__ void run(typename M::Acc (&acc)[NI][NJ], XF &&xf) {
        if constexpr (PIPE) {
            Frag x[2][NJ];                 // activation fragments one K-chunk ahead of their MFMAs (see gemm_stationary)
#pragma unroll
            for (int j = 0; j < NJ; j++) x[0][j] = xf(j, 0);
#pragma unroll
            for (int g = 0; g < NG; g++) {
                if (g + 1 < NG) load(g + 1, (g + 1) & 1);
#pragma unroll
                for (int c = 0; c < G; c++) {
                    const int kc = g * G + c;
                    if (kc < KCH) {
                        if (kc + 1 < KCH) {
#pragma unroll
                            for (int j = 0; j < NJ; j++) x[(kc + 1) & 1][j] = xf(j, kc + 1);
                        }
#pragma unroll
                        for (int i = 0; i < NI; i++)
#pragma unroll
                            for (int j = 0; j < NJ; j++) acc[i][j] = M::mma(buf[g & 1][c][i], x[kc & 1][j], acc[i][j]);
                    }
                }
            }
        } else {
#pragma unroll 2
            for (int kc = 0; kc < KCH; kc++) {
                Frag wf[NI], x[NJ];
#pragma unroll
                for (int i = 0; i < NI; i++) wf[i] = wp[((i * WN) * KCH + kc) * 64];
#pragma unroll
                for (int j = 0; j < NJ; j++) x[j] = xf(j, kc);
#pragma unroll
                for (int i = 0; i < NI; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) acc[i][j] = M::mma(wf[i], x[j], acc[i][j]);
            }
        }
    }
};

// GEMM with the weight fragments resident in registers: acc[i][j] += w[i][kc] x xf(j, kc).  The activation (B) fragments are
// read from LDS DEPTH-1 K-chunks ahead of the MFMAs that use them -- written out explicitly, because the compiler keeps
// source order here and would otherwise expose one LDS round trip (~100+ cycles) per K-chunk against 16 cycles per MFMA.
template <typename T, int NI, int NJ, int KCH, int DEPTH = (NJ >= 4 ? 2 : 3), bool PIN = false, typename XF>
__device__ __forceinline__ void gemm_stationary(typename Mma<T>::Acc (&acc)[NI][NJ], const typename Mma<T>::Frag (&w)[NI][KCH], XF &&xf) {
    typedef Mma<T> M;
    typename M::Frag x[DEPTH][NJ];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++)
        if (d < KCH) {
#pragma unroll
            for (int j = 0; j < NJ; j++) x[d][j] = xf(j, d);
        }
#pragma unroll
    for (int kc = 0; kc < KCH; kc++) {
        if (kc + DEPTH - 1 < KCH) {
#pragma unroll
            for (int j = 0; j < NJ; j++) x[(kc + DEPTH - 1) % DEPTH][j] = xf(j, kc + DEPTH - 1);
        }
        // Pin the software pipeline (round 4): left to itself hipcc's scheduler sinks every B-fragment read to just before the MFMA that
        // consumes it (one register set, `ds_read; s_waitcnt lgkmcnt(0); v_mfma` per MFMA: the ISA of round 3's 3x3 convs exposed one LDS
        // round trip per one or two MFMAs).  Nothing may cross these two fences, so the reads of chunk kc + DEPTH - 1 are in flight while the
        // MFMAs of chunk kc issue.  PIN is a per-call-site choice, measured (tools/gpu/rounds_3_4.sh r4_call3): the long K loops of the 64-channel 3x3
        // convs gain 10 %, the 5-chunk loops of ssh_tail and the fused laterals lose (44 -> 50 us, +1.4 us): their reads are better left
        // where the compiler puts them.
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) acc[i][j] = M::mma(w[i][kc], x[kc % DEPTH][j], acc[i][j]);
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    }
}

// int8 ReLU epilogue: v_cvt_pk_u8_f32 IS clamp(rint(x), 0, 255) -- round to nearest even, ties included, saturating (measured on gfx950:
// tools/probes/cvt_pk_u8.cpp, all 256 ties + neighbours + out-of-range values; and the whole bit-exact int8 suite passes on a build
// without the separate v_rndne_f32 that rounds 1 and 2 spent per value).  Only the upper clamp at 127 is left to the VALU.
#define RF_RNDNE(x) (x)
// The claim is about ONE instruction on ONE architecture under the default rounding mode: this file is built for nothing else (and the
// engine re-checks it on the device it runs on: cvt_pk_u8_selfcheck(), called once per device from rf_create).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "kernels.hip relies on gfx950 semantics (v_cvt_pk_u8_f32 rounding, LDS-DMA, MFMA shapes): build with --offload-arch=gfx950"
#endif

// ReLU on the bit pattern: a signed integer max with 0 is max(x, +0) for every float (negative floats, -0 included, have the sign
// bit set = negative integers).  One v_max_i32 / v_pk_max_i16; fmaxf() costs two instructions wherever the compiler cannot prove
// its operand canonical (MFMA results), and these kernels are VALU-issue bound.
__device__ __forceinline__ float relu_f(float x) {
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}
typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef short s16x2_ __attribute__((ext_vector_type(2)));
// two floats -> packed fp16 (one v_cvt_pk_f16_f32, round to nearest even), optionally ReLU'd AFTER the rounding (rounding is
// monotonic and keeps the sign, so max(rn(x), 0) == rn(max(x, 0)))
__device__ __forceinline__ uint32_t pack_f16(float a, float b, bool relu) {
    const f32x2_ f = {a, b};
    s16x2_ h = __builtin_bit_cast(s16x2_, __builtin_convertvector(f, f16x2_));
    const s16x2_ z = {0, 0};
    if (relu) h = __builtin_elementwise_max(h, z);
    return __builtin_bit_cast(uint32_t, h);
}

// The same with a per-channel floor instead of 0 (DC-centred storage, see stem2_kernel): floor2 = two packed fp16 values -mu.
// max(rn(x), -mu) == rn(max(x, -mu)) because -mu is an fp16 number and rounding is monotonic.  One v_cvt_pk + one v_pk_max_f16.
__device__ __forceinline__ uint32_t pack_f16_floor(float a, float b, uint32_t floor2) {
    const f32x2_ f = {a, b};
    const f16x2_ h = __builtin_elementwise_max(__builtin_convertvector(f, f16x2_), __builtin_bit_cast(f16x2_, floor2));
    return __builtin_bit_cast(uint32_t, h);
}

// Accumulator of a GEMM tile before its first MFMA.  fp16 / fp32 engines: the BIAS (it rides through the MFMA chain as the C
// operand, so the epilogue has no add); int8: zero (the epilogue is y = acc * mult + bias on the int32 sum).
template <typename T> __device__ __forceinline__ typename Mma<T>::Acc acc_init(f32x4 bias) {
    if constexpr (sizeof(T) == 1) return vzero<typename Mma<T>::Acc, 4>();
    else return bias;
}

// epilogue: 4 consecutive output channels of one pixel -> LDS tile s_out[pixel][LDO].
// fp16 / fp32 storage: acc already holds sum + bias (acc_init); ReLU, convert.  int8 storage: y = acc * mult + bias with
// mult[c] = w_scale[c] * in_scale / out_scale and bias = b / out_scale, so y is in units of the output quantum (acc is the int32
// sum, or -- the stem -- an fp32 sum WITHOUT bias).
template <typename T, int LDO, typename ACC>
__device__ __forceinline__ void store_acc(T *s_out, f32x4 mult, f32x4 bv, ACC acc, int ct, int pt, int lane, bool relu) {
    const int c0 = acc_cout(ct, lane, 0);
    const int p = acc_pixel(pt, lane);
    if constexpr (sizeof(T) == 2) {
        uint2 h;
        h.x = pack_f16((float)acc[0], (float)acc[1], relu);
        h.y = pack_f16((float)acc[2], (float)acc[3], relu);
        *(uint2 *)(s_out + p * LDO + c0) = h;
    } else if constexpr (sizeof(T) == 4) {
        f32x4 f;
#pragma unroll
        for (int r = 0; r < 4; r++) f[r] = relu ? relu_f((float)acc[r]) : (float)acc[r];
        *(f32x4 *)(s_out + p * LDO + c0) = f;
    } else {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = fmaf((float)acc[r], mult[r], bv[r]);
        uint32_t packed = 0;
        if (relu) {
            // ReLU'd quanta are 0..127: clamp with one v_med3, then v_cvt_pk_u8_f32 rounds to nearest even (as rintf) and drops the value
            // into its byte -- 2 instructions per value after the fma instead of ~7 (clamp pair, round, convert, mask, shift, or); the
            // int8 epilogues were a third of these kernels' VALU work
#pragma unroll
            for (int r = 0; r < 4; r++)
                packed = __builtin_amdgcn_cvt_pk_u8_f32(RF_RNDNE(__builtin_amdgcn_fmed3f(v[r], 0.f, 127.f)), r, packed);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) packed |= ((uint32_t)(uint8_t)to_T<int8_t>(v[r])) << (8 * r);
        }
        *(uint32_t *)(s_out + p * LDO + c0) = packed;
    }
}

// int8 engine, depthwise output ("mid", never leaves LDS, read by the pointwise 1 x 1 only): ReLU'd quanta 0..255 of amax / 255, stored as
// q - 128 for the signed i8 MFMA (the pointwise bias carries 128 * sum(w_q): weights.h kMidU8).  v_cvt_pk_u8_f32 = clamp(rint(x), 0, 255)
// is the whole requantisation -- no v_med3 -- and ONE xor per four values moves them to two's complement.
template <int LDO, typename ACC>
__device__ __forceinline__ void store_mid_u8(int8_t *s_out, f32x4 mult, f32x4 bv, ACC acc, int ct, int pt, int lane) {
    uint32_t packed = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) packed = __builtin_amdgcn_cvt_pk_u8_f32(fmaf((float)acc[r], mult[r], bv[r]), r, packed);
    *(uint32_t *)(s_out + acc_pixel(pt, lane) * LDO + acc_cout(ct, lane, 0)) = packed ^ 0x80808080u;
}

__device__ __forceinline__ f32x4 load_mult(const float *m, int c0) {
    const f32x4 ones = {1.f, 1.f, 1.f, 1.f};
    return m ? *(const f32x4 *)(m + c0) : ones;
}

// =============================================================================================
// K_a' stem (fp16 engine): preprocess + conv0 + BN + ReLU + depthwise conv1 + BN + ReLU + pointwise conv2 + BN + ReLU
//   = K_a fused with the first K_b block, so the 8-channel 224^2 map (the largest activation of the net per byte of
//   useful work) never exists in HBM.  The kernel is instruction-issue bound (few bytes per pixel, 300+ MACs), so the
//   design minimises VALU work:
//   * staging expands BGR (3 B) to BGRX (4 B) pixels: every later read is an aligned dword, and the zero X byte is the
//     K padding of the MFMA;
//   * conv0 runs on the matrix cores with K ordered (ky, kx, c4): 36 -> 64, two MFMAs; a lane's 8 k's are two whole
//     pixels = two aligned ds_read_b32; u8 -> fp16 is exact and takes 1 instruction per byte
//     (v_perm_b32 builds 0x6400|b = 1024 + b, v_pk_add_f16 subtracts 1024);
//   * weights are split hi + lo in fp16 (two more MFMAs) so conv0 keeps fp32-grade weights on raw 0..255 inputs.
//   * everything BETWEEN the frame and the 16-channel output stays fp32-grade although it never leaves LDS: the conv0 tile is
//     fp32, the depthwise taps are fp32, and the depthwise result feeds the pointwise MFMA as an fp16 hi + lo pair against
//     hi / lo weights in the otherwise empty K slots (8 real channels of K = 32) -- one MFMA, no extra LDS.  On raw 0..255
//     pixels these three tensors carry values up to ~1.5e3 whose fp16 rounding was 73 % of the box-error variance of the
//     whole fp16 engine (tools/fp16_error_budget.py); with them wide the engine is inside north_star's 1e-3 IoU.
//   Tile: 8 x 32 outputs of conv2 <- 10 x 34 conv0 pixels (halo recompute 1.33x) <- 21 x 69 input pixels.
// =============================================================================================
constexpr int ST_TH = 8, ST_TW = 32, ST_P = ST_TH * ST_TW;
constexpr int ST_HR = ST_TH + 2, ST_HC = ST_TW + 2;            // conv0 pixels needed: 10 x 34 = 340
constexpr int ST_NPIX = ST_HR * ST_HC;
constexpr int ST_PTILES = (ST_NPIX + 15) / 16;                 // 22 MFMA pixel tiles
constexpr int ST_IR = 2 * ST_HR + 1;                           // 21 input rows
constexpr int ST_IPX = 2 * ST_HC + 1;                          // 69 input pixels per row
constexpr int ST_GRP = (ST_IPX + 3) / 4;                       // 18 staging groups of 4 pixels (12 B in, 16 B out)
constexpr int ST_ROWD = ST_GRP * 4;                            // 72 dwords (BGRX pixels) per staged row

template <typename TO>
struct StemArgs {
    const FrameDesc *frames; TO *out;
    const half_t *w0;                 // conv0 weights: 4 A fragments [hi k<32 | lo k<32 | hi k>=32 | lo k>=32][64 lanes][8]
    const half_t *w0_raw;             // the same for the raw-row staging: [2 window types][4][64][8] (weights.h c0_raw_), or nullptr
    const uint2 *c0_tab;              // conv0 pixel table of the raw path [4 waves][6 tiles][64 lanes] (pack.h stem_conv0_table), or nullptr
    const float *b0;                  // [8]
    const float *dw_w; const float *dw_b;     // depthwise taps [9][8] fp32
    const half_t *pw_w; const float *pw_b;    // pointwise 16 x 8 as ONE A fragment, K slots [hi | hi | lo | 0] (stem_pw_fragment, pack.h)
    const float *pw_m;                // nullptr, or (int8 output) per-channel multiplier 1 / out_scale; pw_b pre-divided
    int ho, wo, tiles_x, tiles_y, nblk;
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// 4 u8 in a dword -> 4 fp16 holding 1024 + b, exact: (0x6400 | b) IS the fp16 number 1024 + b.  The offset is not subtracted
// here (round 2 spent one v_pk_add_f16 per two bytes on it, 6 of the ~30 VALU instructions of a conv0 MFMA tile in kernels that
// are VALU-issue bound): conv0 is linear, so it is folded into the bias on the host -- b0 - 1024 * sum(all 27 taps), using the
// hi + lo fp16 weights the MFMA really multiplies by (weights.h c0_b_mma_).  That constant is right for zero-padding pixels too:
// a padded byte is 0, arrives as 1024, and its 1024 * w is part of the sum that was taken off.  The X byte of a BGRX pixel
// meets a zero weight.  The fp32 accumulator then carries ~1e4-magnitude partial sums (ulp 1e-3) instead of ~1e3: still two
// orders below the fp16 rounding of the first tensor that is stored.
__device__ __forceinline__ void u8x4_to_f16(uint32_t v, f16x8 &dst, int at) {
    union { uint32_t u; f16x2 h; } lo, hi;
    lo.u = __builtin_amdgcn_perm(0x64646464u, v, 0x04010400u);     // {b0, 0x64, b1, 0x64}
    hi.u = __builtin_amdgcn_perm(0x64646464u, v, 0x04030402u);     // {b2, 0x64, b3, 0x64}
    dst[at] = lo.h[0]; dst[at + 1] = lo.h[1]; dst[at + 2] = hi.h[0]; dst[at + 3] = hi.h[1];
}


template <typename TO> struct StemCfg {
    static constexpr int LDA = 16;                    // depthwise result per pixel: 8 x fp16 hi | 8 x fp16 lo (32 B)
    static constexpr int LDO = 16 + Vec<TO>::N;       // output tile row stride in TO elements (16 B of padding)
    static constexpr int IN_BYTES = ST_IR * ST_ROWD * 4, A_BYTES = ST_P * LDA * 2;
    static constexpr int C0_BYTES = ST_PTILES * 16 * 8 * 4, OUT_BYTES = ST_P * LDO * (int)sizeof(TO);
    static constexpr int REGION_A = IN_BYTES > A_BYTES ? IN_BYTES : A_BYTES;
    static constexpr int REGION_B = C0_BYTES > OUT_BYTES ? C0_BYTES : OUT_BYTES;
    static constexpr int LDS_BYTES = REGION_A + REGION_B + 9 * 8 * 4;
    static constexpr int OCC = LDS_BYTES <= 20 * 1024 ? 8 : 7;     // int8 output: 19.7 KB -> the hardware's 32 waves per CU
};

template <typename TO>
__global__ __launch_bounds__(kThreads, (StemCfg<TO>::OCC)) void stem_kernel(StemArgs<TO> a) {
    typedef half_t T;                                 // compute type of the stem; TO = storage type of its output
    typedef Mma<T> M;
    typedef StemCfg<TO> SC;
    constexpr int LDA = SC::LDA, LDO = SC::LDO;
    // LDS (19.7 KB with an int8 output tile -> 8 workgroups per CU; 23.8 KB with fp16 -> 6).  Two regions, each reused once the barrier after its last
    // reader has passed:
    //   region A: staged BGRX patch (phases 1-2)  ->  depthwise result hi/lo (phases 3-4)
    //   region B: fp32 conv0 tile   (phases 2-3)  ->  output tile (phase 4 - store)
    constexpr int REGION_A = SC::REGION_A, REGION_B = SC::REGION_B;
    constexpr int C0_PLANE = ST_PTILES * 16 * 4;      // fp32 conv0 tile as two 4-channel planes (see stem2_kernel)
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[SC::LDS_BYTES];
    uint32_t *s_in = (uint32_t *)s_raw;                                             // BGRX pixels
    T *s_a = (T *)s_raw;
    float *s_c0 = (float *)(s_raw + REGION_A);
    TO *s_out = (TO *)(s_raw + REGION_A);
    float *s_dw = (float *)(s_raw + REGION_A + REGION_B);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, a.nblk);
    const int tx = bid % a.tiles_x;
    const int ty = (bid / a.tiles_x) % a.tiles_y;
    const int img = bid / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    const FrameDesc fd = a.frames[img];

    // RAW staging (round 6, as stem2_kernel): frames whose base and row pitch are multiples of 16 B and that fill the net's width bring
    // their patch rows into LDS as they are (LDS-DMA, 16 B per lane, 16 lanes = 256 B per row: bytes [192 tx - 16, +256) of the row, so
    // patch pixel px starts at LDS byte 7 + 3 px); conv0 reads a pixel's 9 bytes per kernel row from an aligned 12-byte window -- bytes
    // 3..11 of it for even conv0 columns (weights.h c0_raw_ set 1), 1..9 for odd ones (set 0).  Waves 0,1: even columns, 2,3: odd.
    const bool raw = ((((uintptr_t)fd.ptr) | (uintptr_t)(unsigned)fd.step) & 15u) == 0 && fd.cols == 2 * a.wo && a.w0_raw != nullptr;

    // conv0's pixel table (as stem2_kernel, V2 bit 4): which pixel a lane of the wave's k-th tile computes comes from memory, not from the VALU
    static_assert(ST_HR == kStemR0H && ST_HC == kStemR0W && C0_PLANE == kStemC0Plane, "pack.h stem_conv0_table is built for this geometry");
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool tab = raw && a.c0_tab != nullptr;
    const uint2 *c0_tab = a.c0_tab + (wave_u * kStemC0Tiles) * 64;
    uint2 c0_e = {0u, 0u};
    if (tab) c0_e = c0_tab[lane];

    // ---- phase 0: operands that depend only on kernel arguments
    const f16x8 *wf = (raw ? (const f16x8 *)a.w0_raw + (wave < 2 ? 256 : 0) : (const f16x8 *)a.w0) + lane;
    const f16x8 w_hi1 = wf[0], w_lo1 = wf[64], w_hi2 = wf[128], w_lo2 = wf[192];
    const f32x4 b0 = lane < 32 ? *(const f32x4 *)(a.b0 + (lane >> 4) * 4) : vzero<f32x4, 4>();
    // (the operands of phases 3 and 4 are loaded one phase ahead of their use, not here: 20 more live registers across the
    // staging / conv0 phases cost the 8th workgroup per CU)

    // ---- phase 1: stage the u8 patch as BGRX dwords.  One item = 4 pixels = 12 consecutive frame bytes at an arbitrary
    //      alignment, fetched as the 4 aligned dwords that contain them (ONE buffer_load_dwordx4), realigned with
    //      v_alignbyte, expanded 3 -> 4 bytes per pixel, one 16-byte LDS store.  The frame is addressed through a descriptor
    //      whose base is the frame pointer rounded DOWN to a dword and whose size is rounded UP to one: every fetched dword
    //      is an aligned memory dword holding at least one frame byte (never straddles a page); rows above the frame are
    //      negative offsets and read 0 (hardware range check), rows below are poisoned; bytes left / right of a row are
    //      cleared here.  All 32-bit: no 64-bit address arithmetic or compares on the saturated VALU.
    if (raw) {
        const int iy0 = 2 * oy0 - 3, bxa = 6 * ox0 - 16;
        const unsigned fbytes = (unsigned)(fd.rows - 1) * (unsigned)fd.step + (unsigned)fd.cols * 3u;
        const auto rs = image_rsrc(fd.ptr, fbytes);
        constexpr int PIECES = (ST_IR + 3) / 4;                        // 4 rows of 256 B per wave-instruction: 6 for the 21-row patch
        static_assert(PIECES * 1024 <= SC::REGION_A, "raw patch fits region A");
#pragma unroll 1
        for (int k = wave; k < PIECES; k += 4) {
            const int r = 4 * k + (lane >> 4), cb = bxa + 16 * (lane & 15);
            // rows above / below the frame: out of range = zeros; the chunk left of the row start (tile column 0, chunk 0) is forced out
            const unsigned off = cb < 0 ? kOobOffset : (unsigned)((iy0 + r) * fd.step + cb);
            lds_dma16(rs, s_raw + k * 1024, off);
        }
        wait_vmcnt<0>();
    } else {
        const int iy0 = 2 * oy0 - 3;                                  // input row of patch row 0
        const int bx0 = (2 * ox0 - 3) * 3;                            // input byte column of patch pixel 0
        const uintptr_t fp = (uintptr_t)fd.ptr;
        const int delta = (int)(fp & 3);
        // the descriptor ends with the last row's last pixel, not with a whole `step` (an ROI of a larger image has
        // step > cols*3, and nothing past its last pixel may be touched)
        const unsigned fbytes = ((unsigned)delta + (unsigned)(fd.rows - 1) * (unsigned)fd.step + (unsigned)fd.cols * 3u + 3u) & ~3u;
        const auto rs = image_rsrc((const uint8_t *)(fp & ~(uintptr_t)3), fbytes);
        const int row_bytes = fd.cols * 3;
        for (int i = tid; i < ST_IR * ST_GRP; i += kThreads) {
            const int r = i / ST_GRP, g = i % ST_GRP;
            const int iy = iy0 + r;
            const int off = bx0 + 12 * g;                              // byte column of the item's first byte
            // the leftmost item starts 9 bytes (3 pixels of conv padding) before the row: fetch from the row start and shift --
            // in row 0 those bytes would be a negative offset whose 4th dword wraps to 0, which the hardware treats as out of range
            const int A = delta + iy * fd.step + (off < 0 ? 0 : off);
            const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)iy < (unsigned)fd.rows ? (A & ~3) : (int)kOobOffset, 0, 0);
            const uint32_t sh = (uint32_t)(A & 3);
            uint32_t w0 = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
            uint32_t w1 = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
            uint32_t w2 = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
            static_assert((2 * 0 * ST_TW - 3) * 3 == -9, "the left border item starts exactly 9 bytes before the row");
            if (off < 0) { w2 = w0 << 8; w1 = 0u; w0 = 0u; }           // fetched from column 0: move bytes 0..2 to 9..11
            if (off < 0 || off + 12 > row_bytes) {                     // left / right border of the frame: clear outside bytes
                const int lo = off < 0 ? -off : 0, hi = row_bytes - off < 12 ? row_bytes - off : 12;
                auto bmask = [](int nb) -> uint32_t { return nb <= 0 ? 0u : (nb >= 4 ? 0xffffffffu : (1u << (8 * nb)) - 1u); };
                w0 &= bmask(hi) & ~bmask(lo);
                w1 &= bmask(hi - 4) & ~bmask(lo - 4);
                w2 &= bmask(hi - 8) & ~bmask(lo - 8);
            }
            // 12 bytes -> 4 BGRX dwords.  The X byte is left as it falls (the next pixel's B): its K slot meets a zero weight in both the hi
            // and the lo fragment and (0x6400 | X) is a finite fp16, so the product is an exact +0 -- three v_and_b32 per item less (round 5)
            uint4 o4;
            o4.x = w0;
            o4.y = __builtin_amdgcn_alignbyte(w1, w0, 3);
            o4.z = __builtin_amdgcn_alignbyte(w2, w1, 2);
            o4.w = w2 >> 8;
            *(uint4 *)(s_in + r * ST_ROWD + g * 4) = o4;
        }
    }
    if (tid < 18) *(f32x4 *)(s_dw + tid * 4) = *(const f32x4 *)(a.dw_w + tid * 4);
    __syncthreads();

    // ---- phase 2: conv0 on the 10 x 34 halo'd region: D[cout 16 (8 real)][pixel 16] += W[16][64] x patch[64][16]
    //      k = 4*(3*ky + kx) + c4.  Lane group kb supplies window pixels 2kb, 2kb+1 (MFMA 1) and pixel 8 (MFMA 2, kb 0 only).
    const int kb = lane >> 4;
    if (tab) {
        typedef uint32_t u32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
        const int n = (wave_u & 1) == 0 ? kStemC0Tiles : kStemC0Tiles - 1;      // 11 tiles per column parity over two waves
        const unsigned p2d = (unsigned)(2 - (kb >> 1)) * 256u;
#pragma unroll 1
        for (int k = 0; k < n; k++) {
            const uint2 e = c0_e;
            c0_e = c0_tab[(k + 1) * 64 + lane];                                 // (slot n exists: unused, the next wave's first, or the spare row at the table's end)
            const unsigned p1 = e.x & 0xffffu, ob = e.x >> 16;
            const u32x2a4 d1 = *(const u32x2a4 *)(s_raw + p1), d2 = *(const u32x2a4 *)(s_raw + p1 + p2d);
            f16x8 x1, x2;
            u8x4_to_f16(d1[0], x1, 0);
            u8x4_to_f16(d1[1], x1, 4);
            u8x4_to_f16(d2[0], x2, 0);
            u8x4_to_f16(d2[1], x2, 4);
            f32x4 acc = b0;
            acc = M::mma(w_hi1, x1, acc);
            acc = M::mma(w_lo1, x1, acc);
            acc = M::mma(w_hi2, x2, acc);
            acc = M::mma(w_lo2, x2, acc);
            if (e.y >> 16) {
                const int cy = oy0 - 1 + (int)(e.y & 0xffu), cx = ox0 - 1 + (int)((e.y >> 8) & 0xffu);
                const float lim = ((unsigned)cy < (unsigned)a.ho && (unsigned)cx < (unsigned)a.wo) ? __builtin_inff() : 0.f;
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; r++) h[r] = __builtin_amdgcn_fmed3f(acc[r], 0.f, lim);
                *(f32x4 *)((unsigned char *)s_c0 + ob) = h;
            }
        }
    } else if (raw) {
        typedef uint32_t u32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
        const uint32_t *s_rows = (const uint32_t *)s_raw;
        constexpr int wp = ST_HC / 2;                               // 17 even and 17 odd conv0 columns of the 34
        constexpr int tiles = (wp * ST_HR + 15) / 16;               // 11 per parity
        const int par = wave >> 1;
        const int d0 = par ? 3 : 1;                                 // first dword of column 0's / column 1's aligned window
#pragma unroll 1
        for (int t = wave & 1; t < tiles; t += 2) {
            const int qq = t * 16 + (lane & 15);
            const int hy = qq / wp, m = qq - hy * wp;
            const int hx = 2 * m + par;
            const uint32_t *p1 = s_rows + (2 * hy + (kb >> 1)) * 64 + 3 * m + d0 + 2 * (kb & 1);
            const uint32_t *p2 = s_rows + (2 * hy + 2) * 64 + 3 * m + d0 + 2 * (kb & 1);
            const u32x2a4 d1 = *(const u32x2a4 *)p1, d2 = *(const u32x2a4 *)p2;
            f16x8 x1, x2;
            u8x4_to_f16(d1[0], x1, 0);
            u8x4_to_f16(d1[1], x1, 4);
            u8x4_to_f16(d2[0], x2, 0);
            u8x4_to_f16(d2[1], x2, 4);
            f32x4 acc = b0;
            acc = M::mma(w_hi1, x1, acc);
            acc = M::mma(w_lo1, x1, acc);
            acc = M::mma(w_hi2, x2, acc);
            acc = M::mma(w_lo2, x2, acc);
            if (lane < 32 && hy < ST_HR) {
                const int cy = oy0 - 1 + hy, cx = ox0 - 1 + hx;
                const float lim = ((unsigned)cy < (unsigned)a.ho && (unsigned)cx < (unsigned)a.wo) ? __builtin_inff() : 0.f;
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; r++) h[r] = __builtin_amdgcn_fmed3f(acc[r], 0.f, lim);
                *(f32x4 *)(s_c0 + kb * C0_PLANE + (hy * ST_HC + hx) * 4) = h;
            }
        }
    } else {
    const int ppA = 2 * kb, ppB = 2 * kb + 1;
    const int offA = (ppA / 3) * ST_ROWD + ppA % 3, offB = (ppB / 3) * ST_ROWD + ppB % 3, offC = 2 * ST_ROWD + 2;
    for (int t = wave; t < ST_PTILES; t += 4) {
        const int q = t * 16 + (lane & 15);
        const int hy = q / ST_HC, hx = q % ST_HC;
        const uint32_t *pp = s_in + (2 * hy) * ST_ROWD + 2 * hx;
        // no masks on the reads (see stem2_kernel phase 2): pixels past the patch only reach result columns nobody reads, and
        // window pixel 8 in the lane groups kb > 0 meets zero weights
        const uint32_t vA = pp[offA], vB = pp[offB], vC = pp[offC];
        f16x8 x1, x2 = vzero<f16x8, 8>();
        u8x4_to_f16(vA, x1, 0);
        u8x4_to_f16(vB, x1, 4);
        u8x4_to_f16(vC, x2, 0);
        f32x4 acc = b0;                                   // bias rides in the accumulator
        acc = M::mma(w_hi1, x1, acc);
        acc = M::mma(w_lo1, x1, acc);
        acc = M::mma(w_hi2, x2, acc);
        acc = M::mma(w_lo2, x2, acc);
        if (lane < 32) {
            // conv0 pixels outside its own map are the ZERO PADDING of the depthwise conv, not conv0(zero input): ReLU and that
            // mask are one clamp to [0, lim]
            const int cy = oy0 - 1 + hy, cx = ox0 - 1 + hx;
            const float lim = ((unsigned)cy < (unsigned)a.ho && (unsigned)cx < (unsigned)a.wo) ? __builtin_inff() : 0.f;
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; r++) h[r] = __builtin_amdgcn_fmed3f(acc[r], 0.f, lim);
            *(f32x4 *)(s_c0 + kb * C0_PLANE + q * 4) = h;
        }
    }
    }
    __syncthreads();

    // ---- phase 3: depthwise 3x3 (conv1), one output pixel x 8 channels per thread, fp32 in / fp32 taps / fp32 accumulate;
    //      the result is stored as an fp16 hi + lo pair (hi = RN(v), lo = RN(v - hi): 22 significant bits)
    {
        const int py = tid / ST_TW, px = tid % ST_TW;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = a.dw_b[e];
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const int q = (py + ky) * ST_HC + px + kx;
                const f32x4 x0 = *(const f32x4 *)(s_c0 + q * 4);
                const f32x4 x1 = *(const f32x4 *)(s_c0 + C0_PLANE + q * 4);
                const f32x4 w0 = *(const f32x4 *)(s_dw + (ky * 3 + kx) * 8), w1 = *(const f32x4 *)(s_dw + (ky * 3 + kx) * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; e++) { acc[e] = fmaf(x0[e], w0[e], acc[e]); acc[e + 4] = fmaf(x1[e], w1[e], acc[e + 4]); }
            }
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float v = relu_f(acc[e]);
            hi[e] = (half_t)v;
            lo[e] = (half_t)(v - (float)hi[e]);
        }
        *(f16x8 *)(s_a + tid * LDA) = hi;        // region A: the staged patch is dead since the barrier after phase 2
        *(f16x8 *)(s_a + tid * LDA + 8) = lo;
    }
    GemmPipe<T, 1, 4, 1, 1> pipe;                 // phase 4's operands: requested before the barrier
    pipe.init(a.pw_w, 0, lane);
    const f32x4 pw_bias = *(const f32x4 *)(a.pw_b + acc_cout(0, lane, 0));
    const f32x4 pw_mult = load_mult(a.pw_m, acc_cout(0, lane, 0));
    __syncthreads();

    // ---- phase 4: pointwise 8 -> 16 (conv2) on MFMA, K padded to 32
    f32x4 acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[0][j] = pw_bias;      // (int8 output: weights and bias are already divided by the output scale, weights.h)
    // K slots of the one MFMA: [W_hi x_hi | W_hi x_lo | W_lo x_hi | 0]
    pipe.run(acc, [&](int j, int) -> M::Frag {
        return kb < 3 ? *(const M::Frag *)(s_a + acc_pixel(wave + j * 4, lane) * LDA + (kb & 1) * 8) : M::zero();
    });
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if constexpr (sizeof(TO) == 1) {
            // the accumulator is the value in output quanta: clamp, round to nearest even + pack (v_cvt_pk_u8_f32), 2 instructions per value
            uint32_t packed = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) packed = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(acc[0][j][r], 0.f, 127.f), r, packed);
            *(uint32_t *)(s_out + acc_pixel(wave + j * 4, lane) * LDO + acc_cout(0, lane, 0)) = packed;
        } else {
            store_acc<TO, LDO>(s_out, pw_mult, pw_bias, acc[0][j], 0, wave + j * 4, lane, true);
        }
    }
    __syncthreads();
    typedef typename Vec<TO>::type VO;
    constexpr int OPV = 16 / Vec<TO>::N;                                  // 16-byte chunks per output pixel
    const auto ro = image_rsrc(a.out + (size_t)img * a.ho * a.wo * 16, (unsigned)(a.ho * a.wo * 16) * (unsigned)sizeof(TO));
    const int obase = (oy0 * a.wo + ox0) * 16 * (int)sizeof(TO);
    for (int i = tid; i < ST_P * OPV; i += kThreads) {
        const int p = i / OPV, cv = i % OPV;
        const int py = p / ST_TW, px = p % ST_TW;
        // rows below the map fall outside the descriptor and are dropped; columns need the explicit test
        const unsigned off = ox0 + px < a.wo ? (unsigned)(((py * a.wo + px) * 16 + cv * Vec<TO>::N) * (int)sizeof(TO) + obase) : kOobOffset;
        buf_store16(ro, off, *(const VO *)(s_out + p * LDO + cv * Vec<TO>::N));
    }
}

template <typename TO> void launch_stem(hipStream_t s, const StemParams<TO> &p) {
    const int ho = p.net_h / 2, wo = p.net_w / 2;      // conv2 output = conv0 output size (stride-1 block)
    StemArgs<TO> a;
    a.frames = p.frames; a.out = p.out; a.w0 = p.w0; a.b0 = p.b0; a.w0_raw = p.w0_raw; a.c0_tab = (const uint2 *)p.c0_tab;
    a.dw_w = p.dw_w; a.dw_b = p.dw_b; a.pw_w = p.pw_w; a.pw_b = p.pw_b; a.pw_m = p.pw_m;
    a.ho = ho; a.wo = wo;
    a.tiles_x = (wo + ST_TW - 1) / ST_TW; a.tiles_y = (ho + ST_TH - 1) / ST_TH;
    a.nblk = p.n * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL(stem_kernel<TO>, dim3(a.nblk), dim3(kThreads), 0, s, a);
}
#ifdef RF_PROBES
template void launch_stem<half_t>(hipStream_t, const StemParams<half_t> &);       // fp16 engine with RF_STEM2=0 (K_a' + separate blocks)
#endif
template void launch_stem<int8_t>(hipStream_t, const StemParams<int8_t> &);

// =============================================================================================
// K_a'' stem2 (fp16 engine): K_a' + the first stride-2 block (conv3 depthwise s2 + conv4 pointwise 16 -> 32) in ONE launch.
//   The 224^2 x 16 map between them was the largest tensor of the net (1.6 MB per 448^2 image, written once and read once:
//   414 MB of the ~1.96 GB a 128-image launch moved through HBM) and its consumer dwpw<16,32,s2> ran at the measured HBM copy
//   rate -- the only way to make that faster is not to move the bytes.  Here the map lives in LDS only.
//   Tile = 7 x TW outputs of conv4 (112^2 map)  <-  (15 x (2TW+1)) conv2 pixels  <-  (17 x (2TW+3)) conv0 pixels
//        <-  (35 x (4TW+7)) input pixels.  TW = 8: 255 conv2 pixels = one pass of the 256 threads, 20 KB of LDS.
//   Phases (one barrier between each):  1 stage BGRX patch | 2 conv0 on MFMA -> fp32 tile | 3 depthwise conv1 (fp32, hi/lo
//   out) | 4 pointwise conv2 on MFMA -> fp16 tile (zero outside the map: it is conv3's padding) | 5 depthwise conv3 stride 2
//   | 6 pointwise conv4 on MFMA | 7 coalesced NHWC store.  Numerics of phases 1-4 are exactly K_a' (same rounding points).
// =============================================================================================
template <int TW_, bool F16P, int PADKB = 0> struct Stem2Cfg {     // F16P: the staged patch holds fp16 (converted once) instead of u8; PADKB: unused LDS (occupancy probe)
    static constexpr int TH = 7, TW = TW_, P4 = TH * TW;                // conv4 output tile
    static constexpr int R2H = 2 * TH + 1, R2W = 2 * TW + 1, N2 = R2H * R2W;     // conv2 pixels the tile needs
    static constexpr int R0H = R2H + 2, R0W = R2W + 2, N0 = R0H * R0W;           // conv0 / conv1-input pixels
    static constexpr int T0 = (N0 + 15) / 16, T2 = (N2 + 15) / 16, T4 = (P4 + 15) / 16;   // MFMA pixel tiles of conv0 / conv2 / conv4
    static constexpr int IR = 2 * R0H + 1, IPX = 2 * R0W + 1;                    // input rows / pixels per row
    static constexpr int GRP = (IPX + 3) / 4, ROWD = GRP * 4;                    // staging groups of 4 pixels, dwords per staged row
    static constexpr int LDA1 = 24, LDO = 40;                                    // conv3 result / output tile row pitch (fp16 elements)
    static constexpr int IN_BYTES = IR * ROWD * (F16P ? 8 : 4), A_BYTES = T2 * 16 * 32;       // region A: patch (4 x fp16 per pixel) -> conv1 result (hi|lo, 32 B / pixel)
    static constexpr int A1_BYTES = T4 * 16 * LDA1 * 2, OUT_BYTES = T4 * 16 * LDO * 2;   //           -> conv3 result + output tile
    static constexpr int C0_BYTES = T0 * 16 * 32, C2_BYTES = T2 * 16 * 32;       // region B: fp32 conv0 tile -> fp16 conv2 tile
    static constexpr int MAX3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
    static constexpr int REGION_A = MAX3(IN_BYTES, A_BYTES, A1_BYTES + OUT_BYTES);
    static constexpr int REGION_B = C0_BYTES > C2_BYTES ? C0_BYTES : C2_BYTES;
    static constexpr int LDS_BYTES = REGION_A + REGION_B + 9 * 8 * 4 + PADKB * 1024;
    // 7 x 8 tiles: 4 waves per workgroup; 7 x 16 tiles: the same work per thread with 8 waves (half the horizontal halo per output,
    // twice the LDS per workgroup, the same 32 waves per CU)
    static constexpr int THREADS = TW_ >= 16 ? 512 : 256;
    static constexpr int OCC = LDS_BYTES <= 20 * 1024 ? 8 : LDS_BYTES <= 23 * 1024 ? 7 : (160 * 1024 / LDS_BYTES);
    static_assert(REGION_A % 16 == 0 && REGION_B % 16 == 0 && A1_BYTES % 16 == 0, "LDS carve must stay 16-byte aligned");
};

struct Stem2Args {
    const FrameDesc *frames; half_t *out;           // out: [n][ho4][wo4][32]
    const half_t *w0; const float *b0;              // conv0 (as StemArgs)
    const half_t *w0_raw;                           // conv0 for the raw-row staging: [2 column parities][4 fragments][64][8] (weights.h c0_raw_)
    const uint2 *c0_tab;                            // ... its pixel table [4 waves][6 tiles][64 lanes] (pack.h stem2_conv0_table)
    const u32x4 *dw1_mma4;                          // conv3's diagonal A fragments expanded to 4 dwords per lane [5][64]
    const float *dw0_w; const float *dw0_b; const half_t *pw0_w; const float *pw0_b;     // conv1 / conv2 (as StemArgs)
    const uint32_t *dw1_mma; const float *dw1_b;    // conv3 taps as diagonal MFMA A fragments [5][64] dwords (pack.h dw_mma_dword), bias [16]
    const half_t *pw1_w; const float *pw1_b;        // conv4: 32 x 16 as hi | lo along K (32 K slots, all used), bias [32]
    const uint32_t *c2_floor, *c3_floor;            // DC-centred tiles: -mu of the conv2 / conv3 tile per channel, packed fp16 pairs [8] each
    int ho, wo, ho4, wo4, tiles_x, tiles_y, nblk;   // ho x wo = conv0 / conv2 map (net / 2), ho4 x wo4 = conv4 map (net / 4)
};

template <int TW_, bool F16P, int PADKB = 0, int V2 = 1>
__global__ __launch_bounds__((Stem2Cfg<TW_, F16P, PADKB>::THREADS), (Stem2Cfg<TW_, F16P, PADKB>::OCC)) void stem2_kernel(Stem2Args a) {
    typedef Stem2Cfg<TW_, F16P, PADKB> C;
    constexpr int NT = C::THREADS, NW = NT / 64;         // threads / waves per workgroup
    typedef half_t T;
    typedef Mma<T> M;
    constexpr int TW = C::TW, P4 = C::P4, R2W = C::R2W, N2 = C::N2, R0W = C::R0W;
    constexpr int ROWD = C::ROWD, GRP = C::GRP, IR = C::IR, LDA1 = C::LDA1, LDO = C::LDO;
    // fp32 conv0 tile as two channel planes [channels 0-3 | channels 4-7][pixel][4]: 16 consecutive pixels of a plane are 256
    // contiguous bytes (every ds_read/write_b128 of a 16-lane group is conflict-free) and a depthwise tap is the thread's base
    // address + a compile-time offset (the XOR-swizzled single plane of K_a' cost ~50 address instructions per pixel here)
    constexpr int C0_PLANE = C::T0 * 16 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[C::LDS_BYTES];
    uint2 *s_in = (uint2 *)s_raw;                                // BGRX patch, 4 x fp16 per pixel  (phases 1-2)
    uint32_t *s_in8 = (uint32_t *)s_raw;                         // ... or (!F16P) 4 x u8 per pixel
    T *s_a = (T *)s_raw;                                         // conv1 result hi|lo              (phases 3-4)
    T *s_a1 = (T *)s_raw;                                        // conv3 result                    (phases 5-6)
    T *s_out = (T *)(s_raw + C::A1_BYTES);                       // conv4 tile                      (phases 6-7)
    float *s_c0 = (float *)(s_raw + C::REGION_A);                // fp32 conv0 tile                 (phases 2-3)
    T *s_c2 = (T *)(s_raw + C::REGION_A);                        // fp16 conv2 tile                 (phases 4-5)
    // V2 (round 4, from the LDS counters: this kernel's LDS pipe is busy 0.79 of the time, 0.31 in conflict cycles): (1) the conv2 tile as two 8-channel
    // PLANES of 16 bytes per pixel instead of 32-byte pixels -- its 8-byte epilogue writes were 4-way bank conflicts at the 32-byte pitch, 2-way now, the
    // stride-2 fragment reads of conv3 are unchanged (2-way); (2) conv3 -> conv4 chained in registers (see K_b2c): conv4's K axis is re-ordered so that a
    // lane's conv3 accumulator IS its conv4 B fragment {d0, d1, d0, d1} (hi | lo weights): no conv3 tile in LDS, one barrier less.
    constexpr bool PLANAR = (V2 & 1) != 0, CHAIN = (V2 & 2) != 0;      // V2: bit 0 = (1), bit 1 = (2)
    constexpr bool ROT = (V2 & 4) != 0 && C::R2W == 17 && NT == 256;   //     bit 2 = the depthwise-1 phase's thread -> pixel map (phase 3)
    constexpr int C2_PLANE = PLANAR ? C::T2 * 16 * 8 + 8 : 0;    // halfs; an odd number of 16-byte slots, so the two planes' lanes never share a slot
    static_assert(!PLANAR || 2 * C2_PLANE * 2 <= C::REGION_B, "planar conv2 tile fits region B");
    float *s_dw0 = (float *)(s_raw + C::REGION_A + C::REGION_B);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, a.nblk);
    const int tx = bid % a.tiles_x;
    const int ty = (bid / a.tiles_x) % a.tiles_y;
    const int img = bid / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * C::TH, ox0 = tx * TW;                   // tile origin in the conv4 map
    const FrameDesc fd = a.frames[img];
    const int kb = lane >> 4;
    // the conv0 region [2 oy0 - 2, +R0H) x [2 ox0 - 2, +R0W) (and with it the conv2 region) lies inside the 224^2-type map: no
    // pixel of this tile is another layer's zero padding
    const bool interior = 2 * oy0 - 2 >= 0 && 2 * oy0 - 2 + C::R0H <= a.ho && 2 * ox0 - 2 >= 0 && 2 * ox0 - 2 + C::R0W <= a.wo;
    RF_TRACE_KEY(a.nblk);
    RF_TRACE(4, 0);

    // RAW staging (round 6, V2 bit 3): frames whose base and row pitch are multiples of 16 B and that fill the net's
    // width (every BASELINE shape: 448 x 3 = 1344 = 16 x 84, 1280 x 3 = 3840) bring their patch rows into LDS as they are -- 3 bytes per
    // pixel, `buffer_load_dwordx4 ... lds`, no VGPR round trip, rows above / below the frame zero-filled by the descriptor's range check --
    // instead of the load / realign / border-mask / BGR -> BGRX repack of the general path (~100 of this kernel's ~584 VALU instructions per
    // wave: profiles/r05_stem2_conv0_loop_isa.txt).  conv0 then reads each pixel's 9 bytes per kernel row from an aligned 12-byte window
    // (weights.h c0_raw_: one fragment set per column parity, waves 0-1 take the even conv0 columns, waves 2-3 the odd ones).  Anything
    // else (odd pointers, ROIs, frames narrower than the net) takes the general path: tests/test_gpu_parity.py
    // test_device_frames_unaligned_pointer_odd_step_and_roi.  Wave-uniform per workgroup.
    constexpr bool RAWCAP = (V2 & 8) != 0 && !F16P && NT == 256 && TW == 8;
    const bool raw = RAWCAP && ((((uintptr_t)fd.ptr) | (uintptr_t)(unsigned)fd.step) & 15u) == 0 && fd.cols == 2 * a.wo && a.w0_raw != nullptr;
    // V2 bits 4 / 5 / 6 (round 6, MEASURED AND REJECTED, probe build only): index arithmetic that is the same for every workgroup taken from memory instead of the
    // VALU -- bit 4: conv0's pixel table on the raw path (pack.h stem2_conv0_table: 30 -> 19 VALU instructions per MFMA tile), bit 5: conv3's diagonal A fragments
    // already expanded (20 v_cndmask per wave less), bit 6: the table path's LDS reads as explicit ds_read2_b32.  All bit-identical, all slower: 234.5 us base ->
    // 239.7 (bit 5), 241.7 (4 + 6), 257.9 (all); and 407 us with bit 4 alone, because the compiler reads the 4-byte-aligned dword pair with ONE ds_read_b64, which
    // this chip executes at a fraction of the rate of ds_read2_b32 when the address is not 8-byte aligned.  The kernel is VALU-bound at 0.86, not VALU-ONLY-bound: a
    // dependent global load per tile costs more than the 11 instructions it replaces (profiles/r06_stem_index_tables_rejected.txt).
    constexpr bool TAB = (V2 & 16) != 0 && RAWCAP;
    constexpr bool TABASM = (V2 & 64) != 0;                      // the table path's two LDS reads as explicit ds_read2_b32 (the compiler picks ds_read_b64 at 4-byte alignment)
    static_assert(!TAB || (C::R0H == kStem2R0H && C::R0W == kStem2R0W && C0_PLANE == kStem2C0Plane), "pack.h stem2_conv0_table is built for this geometry");
    // (the wave index as a scalar: the table pointer of tile k is then base + SGPR arithmetic + one constant per-lane offset, no VALU per tile)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const uint2 *c0_tab = a.c0_tab + (wave_u * kStem2C0Tiles) * 64;
    uint2 c0_e = {0u, 0u};
    if (TAB && raw) c0_e = c0_tab[lane];                         // the first tile's entry: requested before the patch is staged

    // ---- phase 0: operands that depend only on kernel arguments
    const f16x8 *wf = (raw ? (const f16x8 *)a.w0_raw + (wave >> 1) * 256 : (const f16x8 *)a.w0) + lane;
    const f16x8 w_hi1 = wf[0], w_lo1 = wf[64], w_hi2 = wf[128], w_lo2 = wf[192];
    const f32x4 b0 = lane < 32 ? *(const f32x4 *)(a.b0 + kb * 4) : vzero<f32x4, 4>();
    // (the operands of phases 4 and 6 are loaded one phase ahead of their use, not here: 24 more live registers across the
    // conv0 / depthwise phases would push the kernel under 7 workgroups per CU)

    // ---- phase 1: stage the u8 patch as BGRX dwords (K_a' phase 1; the patch now starts 5 pixels left of / above the first
    //      conv0 pixel's centre, so an item can lie entirely or partly left of the frame)
    if (raw) {
        // raw rows: LDS row r (128 B = 8 lanes x 16 B) holds the frame bytes [bxa, bxa + 128) of input row iy0 + r, bxa = 3 (4 ox0 - 5) - 1 =
        // 12 ox0 - 16 (a multiple of 16: ox0 is a multiple of 8), so patch pixel px starts at LDS byte 1 + 3 px.  One wave-instruction = 8 rows.
        const int iy0 = 4 * oy0 - 5, bxa = 12 * ox0 - 16;
        const unsigned fbytes = (unsigned)(fd.rows - 1) * (unsigned)fd.step + (unsigned)fd.cols * 3u;
        const auto rs = image_rsrc(fd.ptr, fbytes);
        constexpr int PIECES = (IR + 7) / 8;                           // 5 for the 35-row patch; rows 35..39 exist in LDS (finite bytes, never used)
        static_assert(PIECES * 1024 <= C::REGION_A, "raw patch fits region A");
#pragma unroll 1
        for (int k = wave; k < PIECES; k += NW) {
            const int r = 8 * k + (lane >> 3), cb = bxa + 16 * (lane & 7);
            // rows above / below the frame: the offset is negative (wraps) or past the last row = out of range = zeros.  Bytes LEFT of the
            // row start belong to the previous row and are in range: the one chunk that can lie there (tile column 0, chunk 0) is forced out.
            const unsigned off = cb < 0 ? kOobOffset : (unsigned)((iy0 + r) * fd.step + cb);
            lds_dma16(rs, s_raw + k * 1024, off);
        }
        wait_vmcnt<0>();
    } else {
        const int iy0 = 4 * oy0 - 5;                                  // input row of patch row 0
        const int bx0 = (4 * ox0 - 5) * 3;                            // input byte column of patch pixel 0
        const uintptr_t fp = (uintptr_t)fd.ptr;
        const int delta = (int)(fp & 3);
        const unsigned fbytes = ((unsigned)delta + (unsigned)(fd.rows - 1) * (unsigned)fd.step + (unsigned)fd.cols * 3u + 3u) & ~3u;
        const auto rs = image_rsrc((const uint8_t *)(fp & ~(uintptr_t)3), fbytes);
        const int row_bytes = fd.cols * 3;
#pragma unroll 1
        for (int i = tid; i < IR * GRP; i += NT) {
            const int r = i / GRP, g = i % GRP;
            const int iy = iy0 + r;
            const int off = bx0 + 12 * g;                              // byte column of the item's first byte (may be negative)
            const int neg = off < 0 ? -off : 0;                        // bytes of the item that lie left of the row: 0, 3, 15 (or more)
            const bool row_ok = (unsigned)iy < (unsigned)fd.rows && neg < 12 && off < row_bytes;
            const int A = delta + iy * fd.step + (off < 0 ? 0 : off);  // a partly-left item is fetched from the row start and shifted
            const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, row_ok ? (A & ~3) : (int)kOobOffset, 0, 0);
            const uint32_t sh = (uint32_t)(A & 3);
            uint32_t w0 = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
            uint32_t w1 = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
            uint32_t w2 = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
            if (neg) {                                                 // move byte b of the fetched 12 to position b + neg (neg = 3 here)
                const uint32_t ls = (uint32_t)(4 - (neg & 3)) & 3u;    // alignbyte shift that moves bytes up by (neg & 3)
                uint32_t n0 = w0, n1 = w1, n2 = w2;
                if (neg & 3) {
                    n2 = __builtin_amdgcn_alignbyte(w2, w1, ls);
                    n1 = __builtin_amdgcn_alignbyte(w1, w0, ls);
                    n0 = __builtin_amdgcn_alignbyte(w0, 0u, ls);
                }
                if (neg >= 8) { n2 = n0; n1 = 0u; n0 = 0u; }
                else if (neg >= 4) { n2 = n1; n1 = n0; n0 = 0u; }
                w0 = n0; w1 = n1; w2 = n2;
            }
            if (off < 0 || off + 12 > row_bytes) {                     // left / right border of the frame: clear outside bytes
                const int lo = neg, hi = row_bytes - off < 12 ? row_bytes - off : 12;
                auto bmask = [](int nb) -> uint32_t { return nb <= 0 ? 0u : (nb >= 4 ? 0xffffffffu : (1u << (8 * nb)) - 1u); };
                w0 &= bmask(hi) & ~bmask(lo);
                w1 &= bmask(hi - 4) & ~bmask(lo - 4);
                w2 &= bmask(hi - 8) & ~bmask(lo - 8);
            }
            // 12 bytes -> 4 BGRX dwords; the X byte is left as it falls (see K_a': zero weight x finite fp16 = exact +0): three v_and_b32 per item less
            uint4 o4;
            o4.x = F16P ? (w0 & 0x00ffffffu) : w0;
            o4.y = F16P ? (((w0 >> 24) | (w1 << 8)) & 0x00ffffffu) : __builtin_amdgcn_alignbyte(w1, w0, 3);
            o4.z = F16P ? (((w1 >> 16) | (w2 << 16)) & 0x00ffffffu) : __builtin_amdgcn_alignbyte(w2, w1, 2);
            o4.w = w2 >> 8;
            // u8 -> fp16 HERE, once per input pixel (conv0 reads every pixel 2.25 times on average: converting in phase 2 was
            // 12 of its 30 VALU instructions per MFMA tile); the X byte becomes the half 0 = the K padding
            if constexpr (F16P) {
                f16x8 h01, h23;
                u8x4_to_f16(o4.x, h01, 0);
                u8x4_to_f16(o4.y, h01, 4);
                u8x4_to_f16(o4.z, h23, 0);
                u8x4_to_f16(o4.w, h23, 4);
                *(f16x8 *)(s_in + r * ROWD + g * 4) = h01;
                *(f16x8 *)(s_in + r * ROWD + g * 4 + 2) = h23;
            } else {
                *(uint4 *)(s_in8 + r * ROWD + g * 4) = o4;
            }
        }
    }
    if (tid < 18) *(f32x4 *)(s_dw0 + tid * 4) = *(const f32x4 *)(a.dw0_w + tid * 4);
    RF_TRACE(4, 1);
    __syncthreads();

    // ---- phase 2: conv0 on the (R0H x R0W) region, K = 4*(3*ky + kx) + c4 (K_a' phase 2)
    if (raw) {
        // raw rows: K = 16 ky + (byte of the pixel's aligned 12-byte window); MFMA 1 = kernel rows 0 and 1 (lane groups kb 0,1 | 2,3),
        // MFMA 2 = kernel row 2 (kb 0,1; kb 2,3 meet zero weights).  A lane reads two consecutive dwords per MFMA (ds_read2_b32).
        typedef uint32_t u32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
        const uint32_t *s_rows = (const uint32_t *)s_raw;
        if constexpr (TAB) {
            const int n = wave_u == 0 ? kStem2C0Tiles : kStem2C0Tiles - 1;    // 11 even-column tiles over waves 0,1; 10 odd-column tiles over waves 2,3
            const unsigned p2d = (unsigned)(2 - (kb >> 1)) * 128u;            // MFMA 2 reads kernel row 2: two rows (kb 0,1) / one row (kb 2,3: zero weights) further down
#pragma unroll 1
            for (int k = 0; k < n; k++) {
                const uint2 e = c0_e;
                c0_e = c0_tab[(k + 1) * 64 + lane];                           // next tile's entry, in flight during this tile's MFMAs (entry n of the
                                                                              // last tile's prefetch exists: an unused slot or the next wave's first)
                const unsigned p1 = e.x & 0xffffu, ob = e.x >> 16;
                u32x2a4 d1, d2;
                if constexpr (TABASM) {
                    typedef uint32_t u32x2r __attribute__((ext_vector_type(2)));
                    u32x2r r1, r2;
                    const unsigned l1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)s_raw + p1;
                    asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r1), "=&v"(r2) : "v"(l1), "v"(l1 + p2d) : "memory");
                    d1[0] = r1[0]; d1[1] = r1[1]; d2[0] = r2[0]; d2[1] = r2[1];
                } else {
                    d1 = *(const u32x2a4 *)(s_raw + p1); d2 = *(const u32x2a4 *)(s_raw + p1 + p2d);
                }
                f16x8 x1, x2;
                u8x4_to_f16(d1[0], x1, 0);
                u8x4_to_f16(d1[1], x1, 4);
                u8x4_to_f16(d2[0], x2, 0);
                u8x4_to_f16(d2[1], x2, 4);
                f32x4 acc = b0;
                acc = M::mma(w_hi1, x1, acc);
                acc = M::mma(w_lo1, x1, acc);
                acc = M::mma(w_hi2, x2, acc);
                acc = M::mma(w_lo2, x2, acc);
                if (e.y >> 16) {
                    float lim = __builtin_inff();
                    if (!interior) {
                        asm volatile("" ::: "memory");
                        const int cy = 2 * oy0 - 2 + (int)(e.y & 0xffu), cx = 2 * ox0 - 2 + (int)((e.y >> 8) & 0xffu);
                        lim = ((unsigned)cy < (unsigned)a.ho && (unsigned)cx < (unsigned)a.wo) ? __builtin_inff() : 0.f;
                    }
                    f32x4 h;
#pragma unroll
                    for (int r = 0; r < 4; r++) h[r] = __builtin_amdgcn_fmed3f(acc[r], 0.f, lim);
                    *(f32x4 *)((unsigned char *)s_c0 + ob) = h;
                }
            }
        } else {
        // waves 0,1: even conv0 columns (10 of the 19), waves 2,3: odd ones (9); the parity is a compile-time constant of each copy of the
        // loop, so the pixel -> (row, column) division is by a constant
        auto conv0_raw = [&](auto parity) {
            constexpr int par = decltype(parity)::value;
            constexpr int wp = par ? R0W / 2 : (R0W + 1) / 2;
            constexpr int tiles = (wp * C::R0H + 15) / 16;
#pragma unroll 1
            for (int t = wave & 1; t < tiles; t += 2) {
                const int qq = t * 16 + (lane & 15);
                const int hy = qq / wp, m = qq - hy * wp;
                const int hx = 2 * m + par;
                const uint32_t *p1 = s_rows + (2 * hy + (kb >> 1)) * 32 + 3 * m + par + 2 * (kb & 1);
                const uint32_t *p2 = s_rows + (2 * hy + 2) * 32 + 3 * m + par + 2 * (kb & 1);
                const u32x2a4 d1 = *(const u32x2a4 *)p1, d2 = *(const u32x2a4 *)p2;
                f16x8 x1, x2;
                u8x4_to_f16(d1[0], x1, 0);
                u8x4_to_f16(d1[1], x1, 4);
                u8x4_to_f16(d2[0], x2, 0);
                u8x4_to_f16(d2[1], x2, 4);
                f32x4 acc = b0;
                acc = M::mma(w_hi1, x1, acc);
                acc = M::mma(w_lo1, x1, acc);
                acc = M::mma(w_hi2, x2, acc);
                acc = M::mma(w_lo2, x2, acc);
                if (lane < 32 && hy < C::R0H) {
                    float lim = __builtin_inff();
                    if (!interior) {
                        asm volatile("" ::: "memory");
                        const int cy = 2 * oy0 - 2 + hy, cx = 2 * ox0 - 2 + hx;
                        lim = ((unsigned)cy < (unsigned)a.ho && (unsigned)cx < (unsigned)a.wo) ? __builtin_inff() : 0.f;
                    }
                    f32x4 h;
#pragma unroll
                    for (int r = 0; r < 4; r++) h[r] = __builtin_amdgcn_fmed3f(acc[r], 0.f, lim);
                    *(f32x4 *)(s_c0 + kb * C0_PLANE + (hy * R0W + hx) * 4) = h;
                }
            }
        };
        if (wave < 2) conv0_raw(std::integral_constant<int, 0>());
        else conv0_raw(std::integral_constant<int, 1>());
        }
    } else {
        const int ppA = 2 * kb, ppB = 2 * kb + 1;
        const int offA = (ppA / 3) * ROWD + ppA % 3, offB = (ppB / 3) * ROWD + ppB % 3, offC = 2 * ROWD + 2;
#pragma unroll 1
        for (int t = wave; t < C::T0; t += NW) {
            const int q = t * 16 + (lane & 15);
            const int hy = q / R0W, hx = q % R0W;
            f16x8 x1, x2;
            if constexpr (F16P) {
            const uint2 *pp = s_in + (2 * hy) * ROWD + 2 * hx;
            // No masks on the reads: the pixels q >= N0 of the last MFMA tile read rows past the patch (still inside this
            // workgroup's LDS; whatever they hold only reaches their own columns of the result, which phase 3 never reads) and
            // every lane group reads window pixel 8 (a finite fp16), whose K slots 36..63 meet zero weights (weights.h)
            const uint2 vA = pp[offA], vB = pp[offB], vC = pp[offC];
            const uint4 u1 = {vA.x, vA.y, vB.x, vB.y}, u2 = {vC.x, vC.y, 0u, 0u};
            x1 = __builtin_bit_cast(f16x8, u1); x2 = __builtin_bit_cast(f16x8, u2);
            } else {
                const uint32_t *pp = s_in8 + (2 * hy) * ROWD + 2 * hx;
                x2 = vzero<f16x8, 8>();
                u8x4_to_f16(pp[offA], x1, 0);
                u8x4_to_f16(pp[offB], x1, 4);
                u8x4_to_f16(pp[offC], x2, 0);
            }
            f32x4 acc = b0;                                   // bias rides in the accumulator (lanes >= 32 hold padding rows)
            acc = M::mma(w_hi1, x1, acc);
            acc = M::mma(w_lo1, x1, acc);
            acc = M::mma(w_hi2, x2, acc);
            acc = M::mma(w_lo2, x2, acc);
            if (lane < 32) {
                // conv0 pixels outside its own map are the ZERO PADDING of the depthwise conv, not conv0(zero input):
                // ReLU and that mask are one clamp to [0, lim]
                float lim = __builtin_inff();
                if (!interior) {                                  // wave-uniform: 3 of 4 tiles of a 448^2 frame skip the border test
                    asm volatile("" ::: "memory");                // (an empty side effect: keeps the compiler from if-converting the branch away)
                    const int cy = 2 * oy0 - 2 + hy, cx = 2 * ox0 - 2 + hx;
                    lim = ((unsigned)cy < (unsigned)a.ho && (unsigned)cx < (unsigned)a.wo) ? __builtin_inff() : 0.f;
                }
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; r++) h[r] = __builtin_amdgcn_fmed3f(acc[r], 0.f, lim);      // one v_med3_f32: clamp to [0, lim]
                *(f32x4 *)(s_c0 + kb * C0_PLANE + q * 4) = h;
            }
        }
    }
    RF_TRACE(4, 2);
    __syncthreads();

    // ---- phase 3: depthwise conv1 on the (R2H x R2W) region, fp32, result as fp16 hi + lo (K_a' phase 3)
    {
        float dw_bias[8];
#pragma unroll
        for (int e = 0; e < 8; e++) dw_bias[e] = a.dw0_b[e];
#pragma unroll 1
        for (int i = tid; i < C::T2 * 16; i += NT) {
            // which region pixel a thread computes.  V2 bit 2 (ROT, 17-wide regions): rows of 16 lanes with the column ROTATED by 3 per row, so that
            // a lane's conv0-tile slot (19 ry + rx) mod 16 is its lane index mod 16 whatever the row -- every 16-lane group of the 18 ds_read_b128 per
            // thread then meets 16 distinct slots; with pixel = thread index the groups straddled a 17-pixel row and every read was 2-way (tools/lds_model.py:
            // 144 -> 72 LDS cycles per wave).  The 17th column goes to threads 240..254.  Same arithmetic per pixel: bit-identical.
            auto pixel_of = [&](int &ry, int &rx) {
                if constexpr (ROT) {
                    if (i < 240) { ry = i >> 4; rx = ((i & 15) - 3 * ry) & 15; }
                    else { ry = i - 240; rx = 16; }
                } else {
                    ry = i / R2W; rx = i % R2W;
                }
            };
            int ry, rx;
            pixel_of(ry, rx);
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] = dw_bias[e];
            if (i < N2) {
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const int q = (ry + ky) * R0W + rx + kx;          // = the thread's base pixel + a compile-time tap offset
                        const f32x4 x0 = *(const f32x4 *)(s_c0 + q * 4);
                        const f32x4 x1 = *(const f32x4 *)(s_c0 + C0_PLANE + q * 4);
                        const f32x4 w0 = *(const f32x4 *)(s_dw0 + (ky * 3 + kx) * 8), w1 = *(const f32x4 *)(s_dw0 + (ky * 3 + kx) * 8 + 4);
#pragma unroll
                        for (int e = 0; e < 4; e++) { acc[e] = fmaf(x0[e], w0[e], acc[e]); acc[e + 4] = fmaf(x1[e], w1[e], acc[e + 4]); }
                    }
            }
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float v = relu_f(acc[e]);
                hi[e] = (half_t)v;
                lo[e] = (half_t)(v - (float)hi[e]);
            }
            int po = i;                               // the pixel's slot in the conv1 tile (recomputed here: one register less across the taps)
            if constexpr (ROT) {
                asm volatile("" ::: "memory");
                int ry2, rx2;
                pixel_of(ry2, rx2);
                po = i < N2 ? ry2 * R2W + rx2 : i;        // (thread 255 owns no pixel: it keeps its old slot)
            }
            *(f16x8 *)(s_a + po * 16) = hi;          // region A: the staged patch is dead since the barrier after phase 2
            *(f16x8 *)(s_a + po * 16 + 8) = lo;
        }
    }
    const f16x8 pw0_frag = ((const f16x8 *)a.pw0_w)[lane];            // phase 4's operands: requested before the barrier
    const f32x4 pw0_bias = *(const f32x4 *)(a.pw0_b + kb * 4);        // bias - mu2 (host)
    const uint2 c2_floor = *(const uint2 *)(a.c2_floor + kb * 2);     // -mu2 of the lane's 4 channels
    RF_TRACE(4, 3);
    __syncthreads();

    // ---- phase 4: pointwise conv2 (8 -> 16) on MFMA, K slots [W_hi x_hi | W_hi x_lo | W_lo x_hi | 0]; the result tile is fp16
    //      (the rounding point the un-fused engine had in HBM).  Pixels outside the 224^2 map are conv3's zero padding.
    //      DC-CENTRED STORAGE: this tile carried ~25 % of what was left of the fp16 engine's box-error variance, because its
    //      values ride on a large per-channel DC level (the response to the frame's mean brightness) and fp16 rounds relative to the
    //      magnitude.  The tile therefore holds relu(y) - mu2[c], mu2 = the layer's response to a flat mid-grey frame (an fp16
    //      number per channel, weights.h): the MFMA's bias is b - mu2, ReLU becomes a max with -mu2, zero padding becomes -mu2,
    //      and conv3 adds mu2 * sum(taps) back through ITS bias -- exact algebra, no extra instruction, rounding error / 4.6.
#pragma unroll 1
    for (int t = wave; t < C::T2; t += NW) {
        const int i = t * 16 + (lane & 15);
        const M::Frag x = kb < 3 ? *(const M::Frag *)(s_a + i * 16 + (kb & 1) * 8) : M::zero();
        const f32x4 acc = M::mma(pw0_frag, x, pw0_bias);
        bool inside = i < N2;
        if (!interior) {
            asm volatile("" ::: "memory");
            const int ry = i / R2W, rx = i % R2W;
            const int y2 = 2 * oy0 - 1 + ry, x2 = 2 * ox0 - 1 + rx;
            inside = inside && (unsigned)y2 < (unsigned)a.ho && (unsigned)x2 < (unsigned)a.wo;
        }
        uint2 h;
        h.x = inside ? pack_f16_floor(acc[0], acc[1], c2_floor.x) : c2_floor.x;
        h.y = inside ? pack_f16_floor(acc[2], acc[3], c2_floor.y) : c2_floor.y;
        if constexpr (PLANAR) *(uint2 *)(s_c2 + (kb >> 1) * C2_PLANE + i * 8 + (kb & 1) * 4) = h;
        else *(uint2 *)(s_c2 + i * 16 + kb * 4) = h;
    }
    constexpr bool DW4 = (V2 & 32) != 0;                     // conv3's A fragments arrive expanded (4 dwords per lane and chunk): no v_cndmask in phase 5
    uint32_t dw1v[kDwMmaChunks];                             // phase 5's operands: requested before the barrier
    u32x4 dw1v4[kDwMmaChunks];
#pragma unroll
    for (int kc = 0; kc < kDwMmaChunks; kc++) {
        if constexpr (DW4) dw1v4[kc] = a.dw1_mma4[kc * 64 + lane];
        else dw1v[kc] = a.dw1_mma[kc * 64 + lane];
    }
    const f32x4 dbias = *(const f32x4 *)(a.dw1_b + kb * 4);           // bias + mu2 * sum(taps) - mu3 (host)
    const uint2 c3_floor = *(const uint2 *)(a.c3_floor + kb * 2);     // the conv3 result is stored centred as well (conv4 is 1x1: its bias takes W mu3)
    // V2: conv4's operands too (its A fragments in the chained K order: halves 0-3 = W_hi of channels 4 kb .., 4-7 = W_lo of the same channels)
    M::Frag pw1c[2];
    f32x4 pw1cb[2];
    if constexpr (CHAIN) {
        const uint2 *w8 = (const uint2 *)a.pw1_w;
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
            const uint2 lo = w8[((ct * 64) + (kb >> 1) * 16 + (lane & 15)) * 2 + (kb & 1)];
            const uint2 hi = w8[((ct * 64) + (2 + (kb >> 1)) * 16 + (lane & 15)) * 2 + (kb & 1)];
            const uint4 u = {lo.x, lo.y, hi.x, hi.y};
            pw1c[ct] = __builtin_bit_cast(M::Frag, u);
            pw1cb[ct] = *(const f32x4 *)(a.pw1_b + ct * 16 + kb * 4);
        }
    }
    RF_TRACE(4, 4);
    __syncthreads();

    // ---- phase 5: depthwise conv3, stride 2, pad 1, on the matrix cores: a dense 3x3 conv 16 -> 16 whose weight matrix is diagonal
    //      (K = 9 taps x 16 channels -> 5 chunks of 32; pack.h dw_mma_dword, as in K_b).  B fragments are 16-byte reads of the fp16
    //      conv2 tile at the tap's offset; one pixel tile per wave.  On the VALU this phase was 54 convert + pk_fma instructions
    //      plus 18 LDS reads per thread; the taps are fp16 here, equalised per channel on the host (weights.h) so that their
    //      rounding stays out of the error budget.
    {
        const int dsel = dw_mma_dword_index(lane);
        const bool hi_tap = lane >= 32;                     // chunk kc carries tap 2 kc (lanes 0..31) and 2 kc + 1 (lanes 32..63)
#pragma unroll 1
        for (int pt = wave; pt < C::T4; pt += NW) {
            const int p = pt * 16 + (lane & 15);
            const int py = p / TW, px = p % TW;             // p >= P4: reads past the tile's last row (inside this workgroup's LDS), result never stored
            constexpr int PXH = PLANAR ? 8 : 16;            // halfs per pixel of the tile (plane)
            const T *src = PLANAR ? s_c2 + (kb & 1) * C2_PLANE + ((2 * py) * R2W + 2 * px) * 8 : s_c2 + ((2 * py) * R2W + 2 * px) * 16 + (kb & 1) * 8;
            M::Frag bf[kDwMmaChunks];
#pragma unroll
            for (int kc = 0; kc < kDwMmaChunks; kc++) {
                const int t0 = 2 * kc, t1 = 2 * kc + 1;
                const int o0 = ((t0 / 3) * R2W + t0 % 3) * PXH, o1 = t1 < 9 ? ((t1 / 3) * R2W + t1 % 3) * PXH : o0;      // tap 9 does not exist: its A columns are zero
                bf[kc] = *(const M::Frag *)(src + (hi_tap ? o1 : o0));
            }
            f32x4 acc = dbias;
#pragma unroll
            for (int kc = 0; kc < kDwMmaChunks; kc++) {
                typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
                u32x4_ wa;
                if constexpr (DW4) {
                    wa = dw1v4[kc];
                } else {
                    const uint32_t wd = dw1v[kc];
#pragma unroll
                    for (int d = 0; d < 4; d++) wa[d] = dsel == d ? wd : 0u;
                }
                acc = M::mma(__builtin_bit_cast(M::Frag, wa), bf[kc], acc);
            }
            uint2 h;
            h.x = pack_f16_floor(acc[0], acc[1], c3_floor.x);
            h.y = pack_f16_floor(acc[2], acc[3], c3_floor.y);
            if constexpr (CHAIN) {
                const uint4 u = {h.x, h.y, h.x, h.y};
                const M::Frag x = __builtin_bit_cast(M::Frag, u);
#pragma unroll
                for (int ct = 0; ct < 2; ct++) {
                    const f32x4 o = M::mma(pw1c[ct], x, pw1cb[ct]);
                    uint2 g;
                    g.x = pack_f16(o[0], o[1], true);
                    g.y = pack_f16(o[2], o[3], true);
                    *(uint2 *)(s_out + p * LDO + ct * 16 + kb * 4) = g;      // region A: the conv1 result is dead since the barrier after phase 4
                }
            } else {
                *(uint2 *)(s_a1 + p * LDA1 + kb * 4) = h;      // region A again: the conv1 result is dead since the barrier after phase 4
            }
        }
    }
    RF_TRACE(4, 5);
    __syncthreads();
    if constexpr (!CHAIN) {
    const f16x8 pw1_frag0 = ((const f16x8 *)a.pw1_w)[lane], pw1_frag1 = ((const f16x8 *)a.pw1_w)[64 + lane];      // phase 6's operands
    const f32x4 pw1_bias0 = *(const f32x4 *)(a.pw1_b + kb * 4), pw1_bias1 = *(const f32x4 *)(a.pw1_b + 16 + kb * 4);

    // ---- phase 6: pointwise conv4 (16 -> 32) on MFMA: 2 channel tiles x T4 pixel tiles.  K = 16, and the MFMA has 32 slots: the
    //      weights ride as an fp16 hi | lo pair along K (both halves read the same 16 inputs), so their rounding costs nothing
#pragma unroll 1
    for (int pr = wave; pr < 2 * C::T4; pr += NW) {
        const int ct = pr & 1, pt = pr >> 1;
        const M::Frag x = *(const M::Frag *)(s_a1 + (pt * 16 + (lane & 15)) * LDA1 + (kb & 1) * 8);
        const f32x4 acc = M::mma(ct ? pw1_frag1 : pw1_frag0, x, ct ? pw1_bias1 : pw1_bias0);
        uint2 h;
        h.x = pack_f16(acc[0], acc[1], true);
        h.y = pack_f16(acc[2], acc[3], true);
        *(uint2 *)(s_out + (pt * 16 + (lane & 15)) * LDO + ct * 16 + kb * 4) = h;
    }
    RF_TRACE(4, 6);
    __syncthreads();
    }

    // ---- phase 7: tile -> HBM, 16 B per lane; rows below the map fall outside the descriptor, columns need the test
    {
        const auto ro = image_rsrc(a.out + (size_t)img * a.ho4 * a.wo4 * 32, (unsigned)(a.ho4 * a.wo4 * 32) * 2u);
        const int obase = (oy0 * a.wo4 + ox0) * 32 * 2;
        for (int i = tid; i < P4 * 4; i += NT) {
            const int p = i >> 2, cv = i & 3;
            const int py = p / TW, px = p % TW;
            const unsigned off = ox0 + px < a.wo4 ? (unsigned)(((py * a.wo4 + px) * 32 + cv * 8) * 2 + obase) : kOobOffset;
            buf_store16(ro, off, *(const f16x8 *)(s_out + p * LDO + cv * 8));
        }
    }
    RF_TRACE(4, 7);
}

int stem2_variant() { return knob(K_STEM2); }      // probe knob RF_STEM2: 1 = 7x8 tiles (the product), 0 = off, 2 = 7x16, 3 = 7x8 with the patch converted to fp16 at staging (7 workgroups / CU: measured slower)

void launch_stem2(hipStream_t s, const Stem2Params &p) {
    Stem2Args a;
    a.frames = p.frames; a.out = p.out; a.w0 = p.w0; a.b0 = p.b0; a.w0_raw = p.w0_raw;
    a.c0_tab = (const uint2 *)p.c0_tab; a.dw1_mma4 = (const u32x4 *)p.dw1_mma4;
    a.dw0_w = p.dw0_w; a.dw0_b = p.dw0_b; a.pw0_w = p.pw0_w; a.pw0_b = p.pw0_b;
    a.dw1_mma = p.dw1_mma; a.dw1_b = p.dw1_b; a.pw1_w = p.pw1_w; a.pw1_b = p.pw1_b;
    a.c2_floor = p.c2_floor; a.c3_floor = p.c3_floor;
    a.ho = p.net_h / 2; a.wo = p.net_w / 2; a.ho4 = p.net_h / 4; a.wo4 = p.net_w / 4;
    int tw = 8;
#ifdef RF_PROBES
    const int v = stem2_variant();
    tw = v == 2 ? 16 : 8;
#endif
    a.tiles_x = (a.wo4 + tw - 1) / tw; a.tiles_y = (a.ho4 + 6) / 7;
    a.nblk = p.n * a.tiles_x * a.tiles_y;
#ifdef RF_PROBES
    if (tw == 16) { hipLaunchKernelGGL((stem2_kernel<16, false, 0, 0>), dim3(a.nblk), dim3(Stem2Cfg<16, false>::THREADS), 0, s, a); return; }
    if (v == 3) { hipLaunchKernelGGL((stem2_kernel<8, true, 0, 0>), dim3(a.nblk), dim3(kThreads), 0, s, a); return; }
    // RF_STEM2_PAD (probe knob): 3 / 7 KB of unused LDS per workgroup = 7 / 6 workgroups per CU instead of 8: stem2 alone gets slower
    // (+3.5 % / +10 %), but at 8 it owns every wave slot of the chip and nothing of another lane can run beside it
    const int pad = knob(K_STEM2_PAD);
    if (pad == 3) { hipLaunchKernelGGL((stem2_kernel<8, false, 3, 0>), dim3(a.nblk), dim3(kThreads), 0, s, a); return; }
    if (pad == 7) { hipLaunchKernelGGL((stem2_kernel<8, false, 7, 0>), dim3(a.nblk), dim3(kThreads), 0, s, a); return; }
    // RF_STEM2_V2 (probe knob): bit 0 = planar conv2 tile, bit 1 = conv3 -> conv4 chained in registers, bit 2 = rotated thread -> pixel map of the
    // depthwise-1 phase; 0 = round 3; 5 = round 4's default (bit-identical to round 3: 250.3 -> 246.5 -> 238.6 us, tools/gpu/rounds_3_4.sh r4_call30, r4_call32)
    switch (knob(K_STEM2_V2)) {
        case 15: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 15>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;    // raw-row staging without the index tables
        case 31: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 31>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;    // + conv0 pixel table
        case 47: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 47>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;    // + expanded conv3 fragments only
        case 95: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 95>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;    // + table with explicit ds_read2_b32
        case 127: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 127>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;  // all
        case 7: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 7>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;      // round 5's product
        case 5: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 5>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;
        case 3: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 3>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;
        case 2: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 2>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;
        case 1: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 1>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;
        case 0: hipLaunchKernelGGL((stem2_kernel<8, false, 0, 0>), dim3(a.nblk), dim3(kThreads), 0, s, a); return;
        default: break;
    }
#endif
    // The product: V2 = 7 -- planar conv2 tile, rotated depthwise-1 map and (round 5) conv3 -> conv4 chained in registers: 239.3 -> 235.1 us
    // (tools/gpu/rounds_3_4.sh r4_call35).  The chain permutes conv4's K order, i.e. re-rolls its fp32 summation: on one of the 208 contract frames the
    // NMS winner moves between twin anchors 300 / 301 whose oracle scores are 0.997809 / 0.997806 -- which the anchor-twin band of the parity
    // tests (tests/anchor_twins.py) admits, and nothing else.
    // V2 = 15 (round 6): + bit 3, the raw-row staging of aligned full-width frames (see the kernel; it re-orders conv0's K axis as well).  Bits 4-6 (index
    // tables from memory instead of index arithmetic, bit-identical) measured slower and live in the probe build: profiles/r06_stem_index_tables_rejected.txt.
    hipLaunchKernelGGL((stem2_kernel<8, false, 0, 15>), dim3(a.nblk), dim3(kThreads), 0, s, a);
}

// =============================================================================================
// K_b  depthwise 3x3 + BN + ReLU  ->  pointwise 1x1 + BN + ReLU   (13 backbone pairs, prototxt :55-1193)
//      HAS_DW = false: plain 1x1 + bias + ReLU.
//      LAT = true: the FPN lateral that taps this block's output (rf_c1_red_conv / rf_c2_lateral / rf_c3_lateral,
//      1x1 -> 64 + BN + ReLU, prototxt :1199-1237, :1513-1551, :1908-1946) is computed from the output tile while it
//      is still in LDS: one launch and one HBM re-read less per tap.
//   phase 0  weight fragments (group 0), biases -> registers: their L2 round trip overlaps phase 1's HBM round trip
//   phase 1  halo tile ((TH-1)*S+3) x ((TW-1)*S+3) x CIN -> LDS, 16 B per lane, zero padding resolved here
//   phase 2  depthwise stencil on the VALU, fp32 accumulate, result (as T) -> LDS tile s_a[pixel][CIN]
//   phase 3  pointwise as MFMA GEMM, activations by ds_read_b128 from s_a
//   phase 4  bias + ReLU, through LDS so the NHWC store is 16 B per lane and fully coalesced
//   phase 5  (LAT) second GEMM 64 x COUT on the LDS-resident output tile, same epilogue
// =============================================================================================
template <typename T, int CIN, int COUT, int STRIDE, bool HAS_DW, int TH, int TW, bool PADROW = false> struct DwPwCfg {
    typedef typename DwWeight<T>::type DW;
    typedef Mma<T> M;
    static constexpr int VEC = Vec<T>::N;
    static constexpr int P = TH * TW;
    static constexpr int HR = HAS_DW ? (TH - 1) * STRIDE + 3 : 0;
    static constexpr int HC = HAS_DW ? (TW - 1) * STRIDE + 3 : 0;
    // B-fragment rows of the pointwise GEMM / result tile (read as B fragments by the fused lateral).  Rows of 256 B and more keep
    // the 16-byte pad: lds_row's 32-byte pad measured slower there (18.1 -> 19.1 us on the 128-channel blocks).
    static constexpr int LDA = CIN * sizeof(T) >= 256 ? CIN + VEC : lds_row<T>(CIN);
    static constexpr int LDO = COUT * sizeof(T) >= 256 ? COUT + VEC : lds_row<T>(COUT);
    // fp16 engine: the depthwise stencil runs on the matrix cores as a diagonal-weight 3x3 conv per 16-channel group (see
    // pack.h dw_mma_dword): these kernels are VALU-issue bound (SQ counters: VALU busy 70-100 % of issue cycles, MFMA < 10 %)
    // and the stencil was ~45 % of their VALU instructions.  Needs the (group, pixel-tile) pairs to split over 4 waves.
    // int8 engine: the same with v_mfma_i32_16x16x64_i8 and 15-bit taps split over two fragments (pack.h dw_mma_dword_i8).
    static constexpr bool DWMMA = HAS_DW && sizeof(T) <= 2 && CIN % 16 == 0 && ((CIN / 16) * (P / 16)) % 4 == 0 &&
                                  (CIN >= 64 || (P / 16) % (4 / (CIN / 16 > 0 ? CIN / 16 : 1)) == 0);
    // Halo tile as the depthwise MFMA's B operand: pixel pitch 32 mod 64 bytes and (TW = 8, stride 1: a 16-pixel MFMA tile spans
    // two halo rows) a row pitch that is a multiple of the 256-byte bank row make every B-fragment ds_read_b128 conflict-free
    // (see Conv3Cfg::ROWP).  The halo tile is written with contiguous 128-byte runs, so unlike s_a / s_out it has no write-side
    // preference.  PADROW = false keeps the round-1 layout (probe knob RF_DWPAD=0).
    static constexpr int LDIN = DWMMA ? (!PADROW && CIN * sizeof(T) >= 256 ? CIN + VEC : lds_row<T>(CIN)) : CIN;
    static constexpr int ROWP = PADROW && DWMMA && TW == 8 && STRIDE == 1 ? (HC * LDIN * (int)sizeof(T) + 255) / 256 * 256 / (int)sizeof(T) : HC * LDIN;
    static constexpr size_t IN_BYTES = sizeof(T) * (size_t)(HR * ROWP);
    static constexpr size_t DW_BYTES = HAS_DW && !DWMMA ? sizeof(DW) * (size_t)(9 * CIN) : 0;
    static constexpr size_t A_BYTES = sizeof(T) * (size_t)(P * LDA);
    static constexpr size_t O_BYTES = sizeof(T) * (size_t)(P * LDO);
    // The kernel is persistent: a workgroup walks tiles t, t+G, t+2G, ... and stages tile t+G while it computes tile t,
    // so the halo region is live during the whole tile and s_out cannot reuse it.
    static constexpr size_t LDS_BYTES = IN_BYTES + DW_BYTES + A_BYTES + O_BYTES;
    static_assert(P % 16 == 0 && CIN % VEC == 0 && COUT % 16 == 0, "bad tile");
    static_assert(kThreads % (CIN / VEC) == 0, "a thread must keep one channel group across its depthwise items");
    static_assert(IN_BYTES % 16 == 0 && DW_BYTES % 16 == 0 && A_BYTES % 16 == 0, "LDS carve must stay 16-byte aligned");
    // GEMM split and weight residency.  fp16 / int8: the wave's whole share of the pointwise matrix (NI x KCH MFMA
    // A-fragments, 4 VGPRs each) is loaded ONCE per workgroup and stays in registers for every tile it walks
    // ("weights stationary"), so the tile loop has no global load except the prefetch of the next tile -- nothing the
    // in-order vmcnt would make that prefetch wait behind.  Only when the share is too big for the register file
    // (256x256) do the fragments stream from L2 per tile through GemmPipe.  fp32 (parity engine): always streamed.
    static constexpr int NT = COUT / 16, PT = P / 16;
    typedef WaveSplit<NT, PT> WS;
    static constexpr int KCH = (CIN + M::K - 1) / M::K;
    static constexpr bool STAT = sizeof(T) <= 2 && WS::NI * KCH <= 16;
    // in flight per thread while a tile is computed: the next tile's halo, 16 B per register group
    static constexpr int STAGE_ITEMS = HAS_DW ? HR * HC * (CIN / VEC) : P * (CIN / VEC);
    static constexpr int NPF = (STAGE_ITEMS + kThreads - 1) / kThreads;
    // Time = (tiles / resident workgroups) x per-tile chain, so the big-map layers (CIN <= 64: thousands of tiles per
    // launch) are compiled for as many workgroups per CU as LDS allows; the small-map layers let the compiler take the
    // registers it wants (weights stationary + deep accumulators).
    static constexpr int LDS_OCC = (int)(160 * 1024 / LDS_BYTES) > 8 ? 8 : (int)(160 * 1024 / LDS_BYTES);
    static constexpr bool BIG_MAP = CIN <= 64 && HAS_DW && sizeof(T) <= 2;
    static constexpr int OCC_CAP = sizeof(T) == 1 ? (P > 128 ? 2 : (CIN >= 64 || P > 64 ? 3 : 4)) : (CIN >= 64 ? 4 : 5);   // 170 / 128 / 102 VGPRs: the largest budgets that compile without spills
    static constexpr int OCC = BIG_MAP ? (LDS_OCC < 1 ? 1 : (LDS_OCC > OCC_CAP ? OCC_CAP : LDS_OCC)) : 1;
    static constexpr int GFRAGS = 12;                      // streamed case only
    // (capping the 128-channel blocks at 128 VGPRs for a 4th workgroup per CU spills 33 registers: 18 -> 37 us, measured)
    // (round 2 re-tried the 128-VGPR cap with a shallower B prefetch: still 20-28 spills)
    template <bool LAT> static constexpr int occ() { return OCC; }
};

template <typename T>
struct DwPwArgs {
    const T *in; T *out; const typename DwWeight<T>::type *dw_w; const float *dw_b; const uint32_t *dw_mma; const T *pw_w; const float *pw_b;
    const T *lat_w; const float *lat_b; T *lat_out;
    const float *pw_m, *lat_m;        // int8: per-output-channel requantisation multipliers (nullptr otherwise)
    const float *dw_m;                // int8 depthwise on MFMA: per-channel tap scale
    int hin, win, hout, wout, tiles_x, tiles_y, nblk;     // nblk = tiles in the launch (the grid may be smaller)
};

template <typename T, int CIN, int COUT, int STRIDE, bool HAS_DW, int TH, int TW, bool LAT, bool PADROW>
__global__ __launch_bounds__(kThreads, (DwPwCfg<T, CIN, COUT, STRIDE, HAS_DW, TH, TW, PADROW>::template occ<LAT>())) void dwpw_kernel(DwPwArgs<T> a) {
    typedef DwPwCfg<T, CIN, COUT, STRIDE, HAS_DW, TH, TW, PADROW> C;
    typedef typename Vec<T>::type V;
    typedef Mma<T> M;
    typedef typename M::Frag Frag;
    typedef typename C::WS WS;
    constexpr int VEC = C::VEC, P = C::P, HC = C::HC, LDA = C::LDA, LDO = C::LDO, LDIN = C::LDIN, ROWP = C::ROWP;
    constexpr int CPV = CIN / VEC, PT = C::PT, KCH = C::KCH, NPF = C::NPF;
    constexpr bool STAT = C::STAT, DWMMA = C::DWMMA;
    typedef typename C::DW DW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *s_in = (T *)smem;
    DW *s_dw = (DW *)(smem + C::IN_BYTES);
    T *s_a = (T *)(smem + C::IN_BYTES + C::DW_BYTES);
    T *s_out = (T *)(smem + C::IN_BYTES + C::DW_BYTES + C::A_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    RF_TRACE_KEY(a.nblk);
    RF_TRACE(2, 0);

    // ---- once per workgroup: weights, biases, depthwise taps
    const int wn = wave % WS::WN, wp = wave / WS::WN;
    Frag wst[STAT ? WS::NI : 1][STAT ? KCH : 1];
    if constexpr (STAT) {
        const Frag *wsrc = (const Frag *)a.pw_w + (size_t)wn * KCH * 64 + lane;
#pragma unroll
        for (int i = 0; i < WS::NI; i++)
#pragma unroll
            for (int kc = 0; kc < KCH; kc++) wst[i][kc] = wsrc[((i * WS::WN) * KCH + kc) * 64];
    }
    f32x4 pw_bias[WS::NI], pw_mult[WS::NI];
#pragma unroll
    for (int i = 0; i < WS::NI; i++) {
        pw_bias[i] = *(const f32x4 *)(a.pw_b + acc_cout(wn + i * WS::WN, lane, 0));
        pw_mult[i] = load_mult(a.pw_m, acc_cout(wn + i * WS::WN, lane, 0));
    }
    // lateral: 64 output channels = 4 tiles, one per wave, all pixel tiles
    constexpr int LKCH = (COUT + M::K - 1) / M::K;
    constexpr bool LSTAT = LAT && sizeof(T) <= 2 && LKCH <= 8;
    Frag lst[1][LSTAT ? LKCH : 1];
    f32x4 lat_bias = vzero<f32x4, 4>(), lat_mult = vzero<f32x4, 4>();
    if constexpr (LAT) {
        if constexpr (LSTAT) {
            const Frag *lsrc = (const Frag *)a.lat_w + (size_t)wave * LKCH * 64 + lane;
#pragma unroll
            for (int kc = 0; kc < LKCH; kc++) lst[0][kc] = lsrc[kc * 64];
        }
        lat_bias = *(const f32x4 *)(a.lat_b + acc_cout(wave, lane, 0));
        lat_mult = load_mult(a.lat_m, acc_cout(wave, lane, 0));
    }
    float dw_bias[VEC];
    if constexpr (HAS_DW && !DWMMA) {
#pragma unroll
        for (int e = 0; e < VEC; e++) dw_bias[e] = a.dw_b[(tid % CPV) * VEC + e];
        for (int i = tid; i < 9 * CIN * (int)sizeof(DW) / 16; i += kThreads)
            ((f32x4 *)s_dw)[i] = ((const f32x4 *)a.dw_w)[i];          // visible after the first barrier below
    }
    // depthwise on MFMA: wave -> (channel group, pixel tile) pairs; per group 5 diagonal A fragments, each kept as ONE dword
    // per lane and expanded to 4 when used
    constexpr bool I8 = sizeof(T) == 1;
    constexpr int NG = CIN / 16 > 0 ? CIN / 16 : 1, DKCH = I8 ? kDwMmaChunksI8 : kDwMmaChunks;
    constexpr int DPARTS = I8 ? 2 : 1;                               // int8: hi and lo fragments per chunk
    constexpr int GW = DWMMA ? (NG >= 4 ? NG / 4 : 1) : 1;          // groups per wave
    constexpr int PW = DWMMA ? (NG * PT / 4) / GW : 1;              // pixel tiles per wave and group
    uint32_t dwv[GW][DKCH][DPARTS];
    f32x4 dwb4[GW], dwm4[GW];
    int dpix[PW], dtap[DKCH];
    const int dsel = I8 ? dw_mma_dword_index_i8(lane) : dw_mma_dword_index(lane);
    if constexpr (DWMMA) {
#pragma unroll
        for (int gi = 0; gi < GW; gi++) {
            const int g = NG >= 4 ? wave + 4 * gi : wave % NG;
#pragma unroll
            for (int kc = 0; kc < DKCH; kc++)
#pragma unroll
                for (int hl = 0; hl < DPARTS; hl++) dwv[gi][kc][hl] = a.dw_mma[((g * DKCH + kc) * DPARTS + hl) * 64 + lane];
            dwb4[gi] = *(const f32x4 *)(a.dw_b + acc_cout(g, lane, 0));
            dwm4[gi] = load_mult(I8 ? a.dw_m : nullptr, acc_cout(g, lane, 0));
        }
#pragma unroll
        for (int pi = 0; pi < PW; pi++) {
            const int pt = NG >= 4 ? pi : wave / NG + pi * (4 / NG);
            const int p = acc_pixel(pt, lane);
            // fp16: a lane's 8 k's are half of a tap's 16 channels; int8: all 16 channels of one tap
            dpix[pi] = (p / TW) * STRIDE * ROWP + (p % TW) * STRIDE * LDIN + (I8 ? 0 : ((lane >> 4) & 1) * 8);
        }
#pragma unroll
        for (int kc = 0; kc < DKCH; kc++) {
            const int tap = I8 ? kc * 4 + (lane >> 4) : kc * 2 + (lane >> 5);      // k = tap*16 + c
            dtap[kc] = tap < 9 ? (tap / 3) * ROWP + (tap % 3) * LDIN : -1;
        }
    }

    // ---- the halo (or, without a depthwise stage, the tile itself) of a tile -> registers: unconditional buffer loads,
    // zero padding by the hardware range check (rows) and a poisoned offset (columns)
    V pre[NPF];
    int koff[NPF], kdx[NPF];
#pragma unroll
    for (int k = 0; k < NPF; k++) {
        int i = tid + k * kThreads;
        i = i < C::STAGE_ITEMS ? i : C::STAGE_ITEMS - 1;
        const int pix = i / CPV, cv = i % CPV;
        const int dy = HAS_DW ? pix / HC : pix / TW, dx = HAS_DW ? pix % HC : pix % TW;
        koff[k] = ((dy * a.win + dx) * CIN + cv * VEC) * (int)sizeof(T);
        kdx[k] = dx;
    }
    const unsigned in_img_bytes = (unsigned)(a.hin * a.win * CIN) * (unsigned)sizeof(T);
    const unsigned out_img_bytes = (unsigned)(a.hout * a.wout * COUT) * (unsigned)sizeof(T);
    auto fetch = [&](int tx, int ty, int img) {
        const auto rs = image_rsrc(a.in + (size_t)img * a.hin * a.win * CIN, in_img_bytes);
        const int iy0 = HAS_DW ? ty * TH * STRIDE - 1 : ty * TH, ix0 = HAS_DW ? tx * TW * STRIDE - 1 : tx * TW;
        const int sbase = (iy0 * a.win + ix0) * CIN * (int)sizeof(T);
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const unsigned off = (unsigned)(ix0 + kdx[k]) < (unsigned)a.win ? (unsigned)(koff[k] + sbase) : kOobOffset;
            pre[k] = buf_load16<V>(rs, off);
        }
    };
    // ---- tile (img, oy0, ox0), finished in LDS, -> HBM: 16 B per lane, fully coalesced NHWC rows
    auto store_tile = [&](int img, int oy0, int ox0) {
        constexpr int OPV = COUT / VEC;
        const auto ro = image_rsrc(a.out + (size_t)img * a.hout * a.wout * COUT, out_img_bytes);
        const int obase = (oy0 * a.wout + ox0) * COUT * (int)sizeof(T);
        for (int i = tid; i < P * OPV; i += kThreads) {
            const int p = i / OPV, cv = i % OPV;
            const int py = p / TW, px = p % TW;
            const unsigned off = ox0 + px < a.wout ? (unsigned)(((py * a.wout + px) * COUT + cv * VEC) * (int)sizeof(T) + obase) : kOobOffset;
            buf_store16(ro, off, *(const V *)(s_out + p * LDO + cv * VEC));
        }
        if constexpr (LAT) {
            constexpr int LDL = 64 + VEC, LPV = 64 / VEC;
            const auto rl = image_rsrc(a.lat_out + (size_t)img * a.hout * a.wout * 64, (unsigned)(a.hout * a.wout * 64) * (unsigned)sizeof(T));
            const int lbase = (oy0 * a.wout + ox0) * 64 * (int)sizeof(T);
            for (int i = tid; i < P * LPV; i += kThreads) {
                const int p = i / LPV, cv = i % LPV;
                const int py = p / TW, px = p % TW;
                const unsigned off = ox0 + px < a.wout ? (unsigned)(((py * a.wout + px) * 64 + cv * VEC) * (int)sizeof(T) + lbase) : kOobOffset;
                buf_store16(rl, off, *(const V *)(s_a + p * LDL + cv * VEC));
            }
        }
    };
    // tile walk t = first, first + G, ...: (tx, ty, img) advance by the decomposed step with carries -- no division per tile
    // (a uniform integer division is ~16 scalar instructions, and these loops are a few hundred instructions per tile)
    const TileStep step(G, a.tiles_x, a.tiles_y);
    TileCoord cur(first, a.tiles_x, a.tiles_y), nxt = cur;
    step.advance(nxt);
    if (first < a.nblk) fetch(cur.tx, cur.ty, cur.img);
    // every once-per-workgroup load has landed before the loop: inside it the only loads in flight are the prefetch
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0), expcnt / lgkmcnt untouched

    // The tile loop is software pipelined around the memory round trips:
    //   stage(t): registers -> LDS | fetch(t+G) issued | store(t-1): LDS -> HBM | barrier | stencil | GEMM | epilogue
    // so tile t's compute covers both the flight of tile t+G's loads and the acknowledgement of tile t-1's stores.
    int p_img = -1, p_oy0 = 0, p_ox0 = 0;
    for (int t = first; t < a.nblk; t += G) {
        const int tx = cur.tx, ty = cur.ty, img = cur.img;
        RF_TRACE(2, 8);

        // ---- phase 1: staged registers -> LDS, next tile's loads issued, previous tile -> HBM
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int i = tid + k * kThreads;
            if (i < C::STAGE_ITEMS) {
                if constexpr (HAS_DW) *(V *)(s_in + ((i / CPV) / HC) * ROWP + ((i / CPV) % HC) * LDIN + (i % CPV) * VEC) = pre[k];
                else *(V *)(s_a + (i / CPV) * LDA + (i % CPV) * VEC) = pre[k];
            }
        }
        RF_TRACE(2, 9);
        // the next tile's loads are issued before the previous tile's stores: on the in-order vmcnt the stores are then
        // younger than the loads, and by the time the loads are waited for (one tile of compute later) both have landed
        if (t + G < a.nblk) fetch(nxt.tx, nxt.ty, nxt.img);
        cur = nxt;
        step.advance(nxt);
        RF_TRACE(2, 10);
        // Two barriers per tile instead of three where there is no fused lateral (round 3): the previous tile is stored AFTER this
        // tile's first barrier, so the barrier that used to close a tile (epilogue -> s_out visible) is this one.  Hazards: s_in is
        // rewritten by the next staging after every thread has passed the depthwise barrier; s_a by the next stencil after the next
        // first barrier, which every thread reaches only after its GEMM has read s_a; s_out is read here (after the first barrier,
        // which follows the epilogue that wrote it) and rewritten by this tile's epilogue after the depthwise barrier.  With a
        // lateral the result tile of the lateral shares s_a with the stencil, so the store has to stay ahead of the barrier.
        // int8 only: per kernel it is 1.5-2 % in both precisions, but the fp16 three-lane pipeline measured 289.5 -> 287.6 k images/s with
        // it (int8: 363.2 -> 366.1 k; tools/gpu/rounds_3_4.sh r3_call14, two interleaved repetitions each)
        constexpr bool LATE_STORE = HAS_DW && !LAT && sizeof(T) == 1;
        if constexpr (!LATE_STORE) {
            if (p_img >= 0) store_tile(p_img, p_oy0, p_ox0);
        }
        RF_TRACE(2, 1);
        __syncthreads();
        RF_TRACE(2, 2);
        if constexpr (LATE_STORE) {
            if (p_img >= 0) store_tile(p_img, p_oy0, p_ox0);
        }
        p_img = img; p_oy0 = ty * TH; p_ox0 = tx * TW;

        if constexpr (DWMMA) {
            // ---- phase 2 (fp16 / int8): depthwise 3x3 as diagonal-weight implicit GEMM, D[c][pixel] per 16-channel group
#pragma unroll
            for (int gi = 0; gi < GW; gi++) {
                const int g = NG >= 4 ? wave + 4 * gi : wave % NG;
                typename M::Acc dacc[DPARTS][PW];
#pragma unroll
                for (int hl = 0; hl < DPARTS; hl++)
#pragma unroll
                    for (int pi = 0; pi < PW; pi++) dacc[hl][pi] = acc_init<T>(dwb4[gi]);
                // B fragments (shifted halo reads) run DDEPTH - 1 reads ahead of their MFMAs, as in gemm_stationary: without it
                // every MFMA waited for its own LDS round trip (21 s_waitcnt for 20 MFMAs in the 128-channel block: this
                // phase was the longest of the tile, 685 of ~3000 ns in the phase trace)
                constexpr int NB = DKCH * PW, DDEPTH = NB < 4 ? NB : 4;
                Frag bq[DDEPTH];
                auto bload = [&](int idx) -> Frag {
                    const int kc = idx / PW, pi = idx % PW;
                    return dtap[kc] >= 0 ? *(const Frag *)(s_in + dpix[pi] + dtap[kc] + g * 16) : M::zero();
                };
#pragma unroll
                for (int d = 0; d < DDEPTH - 1; d++) bq[d] = bload(d);
#pragma unroll
                for (int kc = 0; kc < DKCH; kc++) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    Frag af[DPARTS];
#pragma unroll
                    for (int hl = 0; hl < DPARTS; hl++) {
                        u32x4 wa;
                        uint32_t wd = dwv[gi][kc][hl];
                        asm volatile("" : "+v"(wd));        // opaque: keeps the 4-dword expansion inside the tile loop (1 VGPR, not 4, per fragment)
#pragma unroll
                        for (int d = 0; d < 4; d++) wa[d] = dsel == d ? wd : 0u;
                        af[hl] = __builtin_bit_cast(Frag, wa);
                    }
#pragma unroll
                    for (int pi = 0; pi < PW; pi++) {
                        const int idx = kc * PW + pi;
                        if (idx + DDEPTH - 1 < NB) bq[(idx + DDEPTH - 1) % DDEPTH] = bload(idx + DDEPTH - 1);
#pragma unroll
                        for (int hl = 0; hl < DPARTS; hl++) dacc[hl][pi] = M::mma(af[hl], bq[idx % DDEPTH], dacc[hl][pi]);
                    }
                }
#pragma unroll
                for (int pi = 0; pi < PW; pi++) {
                    const int pt = NG >= 4 ? pi : wave / NG + pi * (4 / NG);
                    if constexpr (I8) {
                        typename M::Acc tot;
#pragma unroll
                        for (int r = 0; r < 4; r++) tot[r] = dacc[0][pi][r] * 128 + dacc[DPARTS - 1][pi][r];      // taps = 128*hi + lo
                        store_mid_u8<LDA>(s_a, dwm4[gi], dwb4[gi], tot, g, pt, lane);
                    } else {
                        store_acc<T, LDA>(s_a, dwm4[gi], dwb4[gi], dacc[0][pi], g, pt, lane, true);
                    }
                }
            }
            RF_TRACE(2, 3);
            __syncthreads();
            RF_TRACE(2, 4);
        } else if constexpr (HAS_DW) {
            // ---- phase 2: depthwise stencil on the VALU (fp32 parity engine, int8, and shapes the MFMA split does not fit)
            const int cv = tid % CPV;                 // kThreads % CPV == 0: the channel group is fixed per thread
            for (int i = tid; i < P * CPV; i += kThreads) {
                int p = i / CPV;
                int py = p / TW, px = p % TW;
                float acc[VEC];
#pragma unroll
                for (int e = 0; e < VEC; e++) acc[e] = dw_bias[e];
                V xs[9];               // all nine taps requested before the first is used: one LDS round trip, not nine
#pragma unroll
                for (int k9 = 0; k9 < 9; k9++)
                    xs[k9] = *(const V *)(s_in + (py * STRIDE + k9 / 3) * ROWP + (px * STRIDE + k9 % 3) * LDIN + cv * VEC);
#pragma unroll
                for (int k9 = 0; k9 < 9; k9++) {
                    const DW *wv = s_dw + k9 * CIN + cv * VEC;
#pragma unroll
                    for (int e = 0; e < VEC; e++) acc[e] = fmaf((float)xs[k9][e], (float)wv[e], acc[e]);
                }
                V r;
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    if constexpr (sizeof(T) == 1) r[e] = (int8_t)((int)fminf(fmaxf(rintf(acc[e]), 0.f), 255.f) - 128);      // mid on 0..255 quanta, stored - 128 (store_mid_u8)
                    else r[e] = to_T<T>(fmaxf(acc[e], 0.f));
                }
                *(V *)(s_a + p * LDA + cv * VEC) = r;
            }
            RF_TRACE(2, 3);
            __syncthreads();
            RF_TRACE(2, 4);
        }

        // ---- phase 3: pointwise GEMM  D[cout][pixel], K = CIN
        typename M::Acc acc[WS::NI][WS::NJ];
#pragma unroll
        for (int i = 0; i < WS::NI; i++)
#pragma unroll
            for (int j = 0; j < WS::NJ; j++) acc[i][j] = acc_init<T>(pw_bias[i]);
        auto xf = [&](int j, int kc) -> Frag {
            const int kb = kc * M::K + (lane >> 4) * M::KPL;
            const int p = acc_pixel(wp + j * WS::WP, lane);
            return kb < CIN ? *(const Frag *)(s_a + p * LDA + kb) : M::zero();
        };
        if constexpr (STAT) {
            gemm_stationary<T, WS::NI, WS::NJ, KCH>(acc, wst, xf);
        } else {
            GemmPipe<T, WS::NI, WS::NJ, KCH, WS::WN, C::GFRAGS> pipe;
            pipe.init(a.pw_w, wn, lane);
            pipe.run(acc, xf);
        }
        // ---- phase 4: bias + ReLU -> LDS (stored to HBM while the next tile is staged)
#pragma unroll
        for (int i = 0; i < WS::NI; i++)
#pragma unroll
            for (int j = 0; j < WS::NJ; j++)
                store_acc<T, LDO>(s_out, pw_mult[i], pw_bias[i], acc[i][j], wn + i * WS::WN, wp + j * WS::WP, lane, true);
        RF_TRACE(2, 5);
        if constexpr (!LATE_STORE) __syncthreads();
        RF_TRACE(2, 6);

        if constexpr (LAT) {
            // ---- phase 5: fused lateral  D2[64][pixel] = Wlat[64][COUT] x out_tile; s_a (dead since the barrier above)
            // takes the result and is read by store_tile before the next tile's first barrier, written again after it
            static_assert(LDA >= 64 + VEC, "lateral result tile must fit the depthwise tile");
            constexpr int LDL = 64 + VEC;
            typename M::Acc acc2[1][PT];
#pragma unroll
            for (int j = 0; j < PT; j++) acc2[0][j] = acc_init<T>(lat_bias);
            auto lf = [&](int j, int kc) -> Frag {
                const int kb = kc * M::K + (lane >> 4) * M::KPL;
                return *(const Frag *)(s_out + acc_pixel(j, lane) * LDO + kb);
            };
            if constexpr (LSTAT) {
                gemm_stationary<T, 1, PT, LKCH>(acc2, lst, lf);
            } else {
                GemmPipe<T, 1, PT, LKCH, 4, C::GFRAGS> lpipe;
                lpipe.init(a.lat_w, wave, lane);
                lpipe.run(acc2, lf);
            }
#pragma unroll
            for (int j = 0; j < PT; j++) store_acc<T, LDL>(s_a, lat_mult, lat_bias, acc2[0][j], wave, j, lane, true);
            __syncthreads();
        }
        RF_TRACE(2, 7);
    }
    if constexpr (HAS_DW && !LAT && sizeof(T) == 1) __syncthreads();      // (LATE_STORE: the last tile's epilogue has no closing barrier)
    if (p_img >= 0) store_tile(p_img, p_oy0, p_ox0);
}

// =============================================================================================
// K_b(8)  the 256-channel block (conv25 + conv26 + rf_c3_lateral) with EIGHT waves per workgroup and every weight stationary (round 5).
//   K_b's 4-wave form cannot hold the 256 x 256 pointwise matrix in registers (128 KB = 128 VGPRs per thread of 256): it streams the A
//   fragments from L2 for every 32-pixel tile (GemmPipe) at ONE workgroup = one wave per SIMD per CU -- nothing hides the stream's latency,
//   and the counters say so: HBM 0.20, VALU 0.13, MFMA 0.13 of the chip for 37-42 us per 256 images (profiles/r04_kernels_and_counters_*).
//   With 512 threads the same matrix is 64 VGPRs per thread: wave w owns output-channel tiles {w, w + 8} for both pixel tiles of a 4 x 8
//   tile (2 x 8 K-chunks = 16 A fragments), the 64 x 256 lateral is one channel tile x one pixel tile per wave (8 fragments), the depthwise
//   stencil two 16-channel groups per wave -- the tile loop has no global load except the halo prefetch, and the CU runs two waves per SIMD.
//   Same arithmetic, same summation order per output as K_b (K-chunks in order, bias as the initial accumulator): bit-identical results.
// =============================================================================================
constexpr int kWideThreads = 512;

template <typename T, int CIN, int COUT, int TH, int TW, bool LAT, bool PADROW, int OCC = 1>
__global__ __launch_bounds__(kWideThreads, OCC) void dwpw_wide_kernel(DwPwArgs<T> a) {
    typedef DwPwCfg<T, CIN, COUT, 1, true, TH, TW, PADROW> C;
    typedef typename Vec<T>::type V;
    typedef Mma<T> M;
    typedef typename M::Frag Frag;
    constexpr int NW = kWideThreads / 64;
    constexpr int VEC = C::VEC, P = C::P, HC = C::HC, LDA = C::LDA, LDO = C::LDO, LDIN = C::LDIN, ROWP = C::ROWP;
    constexpr int CPV = CIN / VEC, PT = C::PT, KCH = C::KCH;
    constexpr int NT = COUT / 16, NI = NT / NW, NJ = PT;
    constexpr bool I8 = sizeof(T) == 1;
    constexpr int NG = CIN / 16, GW = NG / NW, PW = PT, DKCH = I8 ? kDwMmaChunksI8 : kDwMmaChunks;
    constexpr int DPARTS = I8 ? 2 : 1;                               // int8: 15-bit taps as hi and lo fragments per chunk (K_b)
    constexpr int LKCH = (COUT + M::K - 1) / M::K, LNJ = LAT ? PT / (NW / 4) : 1;
    constexpr int NPF = (C::STAGE_ITEMS + kWideThreads - 1) / kWideThreads;
    static_assert(sizeof(T) <= 2 && C::DWMMA && NT % NW == 0 && NG % NW == 0 && (!LAT || PT % (NW / 4) == 0), "shape does not split over 8 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *s_in = (T *)smem;
    T *s_a = (T *)(smem + C::IN_BYTES + C::DW_BYTES);
    T *s_out = (T *)(smem + C::IN_BYTES + C::DW_BYTES + C::A_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);

    // ---- once per workgroup: every weight this wave will ever multiply by
    Frag wst[NI][KCH];
    {
        const Frag *wsrc = (const Frag *)a.pw_w + (size_t)wave * KCH * 64 + lane;
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int kc = 0; kc < KCH; kc++) wst[i][kc] = wsrc[((i * NW) * KCH + kc) * 64];
    }
    f32x4 pw_bias[NI], pw_mult[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        pw_bias[i] = *(const f32x4 *)(a.pw_b + acc_cout(wave + i * NW, lane, 0));
        pw_mult[i] = load_mult(a.pw_m, acc_cout(wave + i * NW, lane, 0));
    }
    const int lct = wave & 3, lp0 = (wave >> 2) * LNJ;      // lateral: channel tile, first pixel tile of this wave
    Frag lst[1][LAT ? LKCH : 1];
    f32x4 lat_bias = vzero<f32x4, 4>(), lat_mult = vzero<f32x4, 4>();
    if constexpr (LAT) {
        const Frag *lsrc = (const Frag *)a.lat_w + (size_t)lct * LKCH * 64 + lane;
#pragma unroll
        for (int kc = 0; kc < LKCH; kc++) lst[0][kc] = lsrc[kc * 64];
        lat_bias = *(const f32x4 *)(a.lat_b + acc_cout(lct, lane, 0));
        lat_mult = load_mult(a.lat_m, acc_cout(lct, lane, 0));
    }
    uint32_t dwv[GW][DKCH][DPARTS];
    f32x4 dwb4[GW], dwm4[GW];
    int dpix[PW], dtap[DKCH];
    const int dsel = I8 ? dw_mma_dword_index_i8(lane) : dw_mma_dword_index(lane);
#pragma unroll
    for (int gi = 0; gi < GW; gi++) {
        const int g = wave + NW * gi;
#pragma unroll
        for (int kc = 0; kc < DKCH; kc++)
#pragma unroll
            for (int hl = 0; hl < DPARTS; hl++) dwv[gi][kc][hl] = a.dw_mma[((g * DKCH + kc) * DPARTS + hl) * 64 + lane];
        dwb4[gi] = *(const f32x4 *)(a.dw_b + acc_cout(g, lane, 0));
        dwm4[gi] = load_mult(I8 ? a.dw_m : nullptr, acc_cout(g, lane, 0));
    }
#pragma unroll
    for (int pi = 0; pi < PW; pi++) {
        const int p = acc_pixel(pi, lane);
        dpix[pi] = (p / TW) * ROWP + (p % TW) * LDIN + (I8 ? 0 : ((lane >> 4) & 1) * 8);      // fp16: a lane's 8 k's are half of a tap's 16 channels; int8: all 16
    }
#pragma unroll
    for (int kc = 0; kc < DKCH; kc++) {
        const int tap = I8 ? kc * 4 + (lane >> 4) : kc * 2 + (lane >> 5);
        dtap[kc] = tap < 9 ? (tap / 3) * ROWP + (tap % 3) * LDIN : -1;
    }

    // ---- halo of a tile -> registers (K_b's scheme: unconditional buffer loads, padding by the range check / a poisoned offset)
    V pre[NPF];
    int koff[NPF], kdx[NPF];
#pragma unroll
    for (int k = 0; k < NPF; k++) {
        int i = tid + k * kWideThreads;
        i = i < C::STAGE_ITEMS ? i : C::STAGE_ITEMS - 1;
        const int pix = i / CPV, cv = i % CPV;
        const int dy = pix / HC, dx = pix % HC;
        koff[k] = ((dy * a.win + dx) * CIN + cv * VEC) * (int)sizeof(T);
        kdx[k] = dx;
    }
    const unsigned in_img_bytes = (unsigned)(a.hin * a.win * CIN) * (unsigned)sizeof(T);
    const unsigned out_img_bytes = (unsigned)(a.hout * a.wout * COUT) * (unsigned)sizeof(T);
    auto fetch = [&](int tx, int ty, int img) {
        const auto rs = image_rsrc(a.in + (size_t)img * a.hin * a.win * CIN, in_img_bytes);
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        const int sbase = (iy0 * a.win + ix0) * CIN * (int)sizeof(T);
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const unsigned off = (unsigned)(ix0 + kdx[k]) < (unsigned)a.win ? (unsigned)(koff[k] + sbase) : kOobOffset;
            pre[k] = buf_load16<V>(rs, off);
        }
    };
    auto store_tile = [&](int img, int oy0, int ox0) {
        constexpr int OPV = COUT / VEC;
        const auto ro = image_rsrc(a.out + (size_t)img * a.hout * a.wout * COUT, out_img_bytes);
        const int obase = (oy0 * a.wout + ox0) * COUT * (int)sizeof(T);
        for (int i = tid; i < P * OPV; i += kWideThreads) {
            const int p = i / OPV, cv = i % OPV;
            const int py = p / TW, px = p % TW;
            const unsigned off = ox0 + px < a.wout ? (unsigned)(((py * a.wout + px) * COUT + cv * VEC) * (int)sizeof(T) + obase) : kOobOffset;
            buf_store16(ro, off, *(const V *)(s_out + p * LDO + cv * VEC));
        }
        if constexpr (LAT) {
            constexpr int LDL = 64 + VEC, LPV = 64 / VEC;
            const auto rl = image_rsrc(a.lat_out + (size_t)img * a.hout * a.wout * 64, (unsigned)(a.hout * a.wout * 64) * (unsigned)sizeof(T));
            const int lbase = (oy0 * a.wout + ox0) * 64 * (int)sizeof(T);
            for (int i = tid; i < P * LPV; i += kWideThreads) {
                const int p = i / LPV, cv = i % LPV;
                const int py = p / TW, px = p % TW;
                const unsigned off = ox0 + px < a.wout ? (unsigned)(((py * a.wout + px) * 64 + cv * VEC) * (int)sizeof(T) + lbase) : kOobOffset;
                buf_store16(rl, off, *(const V *)(s_a + p * LDL + cv * VEC));
            }
        }
    };
    const TileStep step(G, a.tiles_x, a.tiles_y);
    TileCoord cur(first, a.tiles_x, a.tiles_y), nxt = cur;
    step.advance(nxt);
    if (first < a.nblk) fetch(cur.tx, cur.ty, cur.img);
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): every once-per-workgroup load has landed; inside the loop only the prefetch is in flight

    int p_img = -1, p_oy0 = 0, p_ox0 = 0;
    for (int t = first; t < a.nblk; t += G) {
        const int tx = cur.tx, ty = cur.ty, img = cur.img;
        // ---- staged registers -> LDS, next tile's loads issued, previous tile -> HBM (its lateral result shares s_a with the stencil: before the barrier)
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int i = tid + k * kWideThreads;
            if (i < C::STAGE_ITEMS) *(V *)(s_in + ((i / CPV) / HC) * ROWP + ((i / CPV) % HC) * LDIN + (i % CPV) * VEC) = pre[k];
        }
        if (t + G < a.nblk) fetch(nxt.tx, nxt.ty, nxt.img);
        cur = nxt;
        step.advance(nxt);
        if (p_img >= 0) store_tile(p_img, p_oy0, p_ox0);
        __syncthreads();
        p_img = img; p_oy0 = ty * TH; p_ox0 = tx * TW;

        // ---- depthwise 3x3 as a diagonal-weight implicit GEMM per 16-channel group (K_b phase 2), two groups per wave
#pragma unroll
        for (int gi = 0; gi < GW; gi++) {
            const int g = wave + NW * gi;
            typename M::Acc dacc[DPARTS][PW];
#pragma unroll
            for (int hl = 0; hl < DPARTS; hl++)
#pragma unroll
                for (int pi = 0; pi < PW; pi++) dacc[hl][pi] = acc_init<T>(dwb4[gi]);
            constexpr int NB = DKCH * PW, DDEPTH = NB < 4 ? NB : 4;
            Frag bq[DDEPTH];
            auto bload = [&](int idx) -> Frag {
                const int kc = idx / PW, pi = idx % PW;
                return dtap[kc] >= 0 ? *(const Frag *)(s_in + dpix[pi] + dtap[kc] + g * 16) : M::zero();
            };
#pragma unroll
            for (int d = 0; d < DDEPTH - 1; d++) bq[d] = bload(d);
#pragma unroll
            for (int kc = 0; kc < DKCH; kc++) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                Frag af[DPARTS];
#pragma unroll
                for (int hl = 0; hl < DPARTS; hl++) {
                    u32x4 wa;
                    uint32_t wd = dwv[gi][kc][hl];
                    asm volatile("" : "+v"(wd));
#pragma unroll
                    for (int d = 0; d < 4; d++) wa[d] = dsel == d ? wd : 0u;
                    af[hl] = __builtin_bit_cast(Frag, wa);
                }
#pragma unroll
                for (int pi = 0; pi < PW; pi++) {
                    const int idx = kc * PW + pi;
                    if (idx + DDEPTH - 1 < NB) bq[(idx + DDEPTH - 1) % DDEPTH] = bload(idx + DDEPTH - 1);
#pragma unroll
                    for (int hl = 0; hl < DPARTS; hl++) dacc[hl][pi] = M::mma(af[hl], bq[idx % DDEPTH], dacc[hl][pi]);
                }
            }
#pragma unroll
            for (int pi = 0; pi < PW; pi++) {
                if constexpr (I8) {
                    typename M::Acc tot;
#pragma unroll
                    for (int r = 0; r < 4; r++) tot[r] = dacc[0][pi][r] * 128 + dacc[DPARTS - 1][pi][r];      // taps = 128 * hi + lo
                    store_mid_u8<LDA>(s_a, dwm4[gi], dwb4[gi], tot, g, pi, lane);
                } else {
                    store_acc<T, LDA>(s_a, dwm4[gi], dwb4[gi], dacc[0][pi], g, pi, lane, true);
                }
            }
        }
        __syncthreads();

        // ---- pointwise GEMM, weights in registers: D[cout][pixel], K = CIN
        {
            typename M::Acc acc[NI][NJ];
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) acc[i][j] = acc_init<T>(pw_bias[i]);
            auto xf = [&](int j, int kc) -> Frag {
                const int kb = kc * M::K + (lane >> 4) * M::KPL;
                return *(const Frag *)(s_a + acc_pixel(j, lane) * LDA + kb);
            };
            gemm_stationary<T, NI, NJ, KCH>(acc, wst, xf);
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) store_acc<T, LDO>(s_out, pw_mult[i], pw_bias[i], acc[i][j], wave + i * NW, j, lane, true);
        }
        __syncthreads();

        if constexpr (LAT) {
            // ---- fused lateral 64 x COUT on the LDS-resident output tile; s_a (dead since the barrier above) takes the result
            static_assert(LDA >= 64 + VEC, "lateral result tile must fit the depthwise tile");
            constexpr int LDL = 64 + VEC;
            typename M::Acc acc2[1][LNJ];
#pragma unroll
            for (int j = 0; j < LNJ; j++) acc2[0][j] = acc_init<T>(lat_bias);
            auto lf = [&](int j, int kc) -> Frag {
                const int kb = kc * M::K + (lane >> 4) * M::KPL;
                return *(const Frag *)(s_out + acc_pixel(lp0 + j, lane) * LDO + kb);
            };
            gemm_stationary<T, 1, LNJ, LKCH>(acc2, lst, lf);
#pragma unroll
            for (int j = 0; j < LNJ; j++) store_acc<T, LDL>(s_a, lat_mult, lat_bias, acc2[0][j], lct, lp0 + j, lane, true);
            __syncthreads();
        }
    }
    if (p_img >= 0) store_tile(p_img, p_oy0, p_ox0);
}

template <typename T, int CIN, int COUT, int TH, int TW, bool LAT, bool PADROW, int OCC = 1>
static void dwpw_wide_launch(hipStream_t s, const DwPwParams<T> *p, int tiles_x, int tiles_y) {
    typedef DwPwCfg<T, CIN, COUT, 1, true, TH, TW, PADROW> C;
    auto kern = dwpw_wide_kernel<T, CIN, COUT, TH, TW, LAT, PADROW, OCC>;
    static std::atomic<int> resident_cache[kMaxDevices] = {};
    const int dev = launch_device();
    int resident = resident_cache[dev].load(std::memory_order_acquire);
    if (!resident) {
        set_max_lds(kern, C::LDS_BYTES);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, kWideThreads, C::LDS_BYTES) != hipSuccess || nb < 1) nb = 1;
        resident = nb;
        resident_cache[dev].store(resident, std::memory_order_release);
    }
    DwPwArgs<T> a{p->in, p->out, p->dw_w, p->dw_b, p->dw_mma, p->pw_w, p->pw_b, p->lat_w, p->lat_b, p->lat_out, p->pw_m, p->lat_m, p->dw_m,
                  p->hin, p->win, p->hout, p->wout, tiles_x, tiles_y, p->n * tiles_x * tiles_y};
    hipLaunchKernelGGL(kern, dim3(persistent_grid(a.nblk, resident)), dim3(kWideThreads), C::LDS_BYTES, s, a);
}

#ifdef RF_PROBES      // measured and rejected (profiles/r04_rejected_ws_variants.txt): probe build only
// =============================================================================================
// K_b'  depthwise + pointwise (+ lateral) block WAVE-SPECIALISED (round 4): K_b's phases with the memory side taken off the four
//   GEMM waves.  A fifth wave (the producer) owns every global-memory instruction of the tile loop: it stores the previous tile's
//   result(s) from LDS and brings the halo of the tile DIST = NBUF - 1 steps ahead into one of NBUF LDS buffers by LDS-DMA
//   (buffer_load ... lds: no VGPRs, no ds_write pass, zero padding by the descriptor's range check); the consumers run
//   depthwise (diagonal MFMA) -> barrier -> pointwise GEMM -> [barrier -> lateral GEMM] -> barrier on LDS only.  The producer joins the
//   consumers' barriers (s_barrier is workgroup wide) but waits only for the halo the NEXT interval needs, with a counted vmcnt: its own
//   stores and the younger DMA stay in flight across the barrier.  Same tile geometry, LDS pitches, weight packing and arithmetic as
//   K_b: the results are bit-identical (the int8 instance is held to the integer oracle like K_b's).
//   What it took off the critical path in the merged SSH conv (K_c'): 104 -> 56 us.
// =============================================================================================
template <typename T, int CIN, int COUT, int STRIDE, int TH, int TW, bool LAT, int NBUF, bool PADROW, bool PROD = true> struct DwPwWsCfg {
    typedef DwPwCfg<T, CIN, COUT, STRIDE, true, TH, TW, PADROW> B;
    static constexpr int VEC = B::VEC, P = B::P;
    static constexpr int CPP = B::LDIN / VEC;                          // 16-byte chunks per halo pixel, padding included
    static constexpr int CPR = B::ROWP / VEC;                          // chunks per halo row, padding included
    static constexpr int DPP = CIN / VEC;                              // data chunks per pixel
    static constexpr int SLOTS = B::HR * CPR;
    // PROD: a fifth wave owns the memory side.  !PROD (K_b''): no producer wave -- each of the four waves issues a quarter of the halo DMA and of the
    // stores (a 320-thread workgroup at this register budget is resident once per CU; 256-thread ones pack)
    static constexpr int NW = PROD ? 1 : 4;                            // waves that share the memory side
    static constexpr int PIECES = ((SLOTS + 63) / 64 + NW - 1) / NW * NW;      // DMA wave-instructions per halo tile (whole pieces per wave)
    static constexpr int PPW = PIECES / NW;
    static constexpr int LDL = 64 + VEC;
    static constexpr int NST_OUT = P * (COUT / VEC) / 64, NST_LAT = LAT ? P * (64 / VEC) / 64 : 0;
    static constexpr int NSTORE = NST_OUT + NST_LAT;                   // store wave-instructions per tile
    static constexpr int NSTW = NSTORE / NW;
    static constexpr bool SPLIT_OK = NST_OUT % NW == 0 && NST_LAT % NW == 0;      // the same number of vmcnt events in every wave
    static constexpr size_t IN_BYTES = (size_t)PIECES * 1024;
    static constexpr size_t A_BYTES = B::A_BYTES, O_BYTES = B::O_BYTES;
    static constexpr size_t L_BYTES = LAT ? sizeof(T) * (size_t)(P * LDL) : 0;
    static constexpr size_t LDS_BYTES = NBUF * IN_BYTES + A_BYTES + O_BYTES + L_BYTES;
    static constexpr int THREADS = PROD ? 320 : 256;                   // 4 GEMM waves (+ the producer)
    static constexpr int WG_CAP = LAT ? 2 : 3;                         // (the lateral's second GEMM needs ~140 VGPRs: 3 workgroups of 5 waves would spill)
    static constexpr int WG_PER_CU = (int)(160 * 1024 / LDS_BYTES) > WG_CAP ? WG_CAP : (int)(160 * 1024 / LDS_BYTES);
    static constexpr int WAVES_PER_EU = PROD ? (WG_PER_CU * 5 + 3) / 4 : WG_PER_CU;      // what __launch_bounds__ needs for WG_PER_CU resident workgroups
    static_assert(B::DWMMA && B::STAT && sizeof(T) <= 2, "only the shapes whose depthwise stage runs on MFMA with stationary pointwise weights");
    static_assert(B::LDIN % VEC == 0 && B::ROWP % VEC == 0 && IN_BYTES >= B::IN_BYTES, "halo layout in whole 16-byte slots");
    static_assert((P * (COUT / VEC)) % 64 == 0 && (!LAT || (P * (64 / VEC)) % 64 == 0), "whole store instructions per tile");
    static_assert(NBUF == 2 || NBUF == 3, "prefetch distance 1 or 2");
    static_assert(!LAT || (COUT + Mma<T>::K - 1) / Mma<T>::K <= 8, "lateral weights stationary");
};

template <typename T, int CIN, int COUT, int STRIDE, int TH, int TW, bool LAT, int NBUF, bool PADROW, bool PROD = true>
__global__ __launch_bounds__((DwPwWsCfg<T, CIN, COUT, STRIDE, TH, TW, LAT, NBUF, PADROW, PROD>::THREADS), (DwPwWsCfg<T, CIN, COUT, STRIDE, TH, TW, LAT, NBUF, PADROW, PROD>::WAVES_PER_EU))
void dwpw_ws_kernel(DwPwArgs<T> a) {
    typedef DwPwWsCfg<T, CIN, COUT, STRIDE, TH, TW, LAT, NBUF, PADROW, PROD> W;
    typedef typename W::B C;
    typedef typename Vec<T>::type V;
    typedef Mma<T> M;
    typedef typename M::Frag Frag;
    typedef typename C::WS WS;
    constexpr int VEC = C::VEC, P = C::P, HC = C::HC, LDA = C::LDA, LDO = C::LDO, LDIN = C::LDIN, ROWP = C::ROWP;
    constexpr int PT = C::PT, KCH = C::KCH, LDL = W::LDL;
    constexpr int DIST = NBUF - 1, NW = W::NW, PPW = W::PPW;
    constexpr bool I8 = sizeof(T) == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int IN_ELEMS = (int)(W::IN_BYTES / sizeof(T));
    T *s_in = (T *)smem;                                                           // [NBUF][IN_ELEMS]
    T *s_a = (T *)(smem + NBUF * W::IN_BYTES);                                     // depthwise result
    T *s_out = (T *)(smem + NBUF * W::IN_BYTES + W::A_BYTES);                      // block output tile
    T *s_lat = (T *)(smem + NBUF * W::IN_BYTES + W::A_BYTES + W::O_BYTES);         // lateral output tile (LAT)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const int n_my = first < a.nblk ? (a.nblk - 1 - first) / G + 1 : 0;

    // ---- the memory side: the producer wave's (PROD) or this wave's quarter (!PROD) of the halo DMA and of the stores
    const int mw = PROD ? 0 : wave;                                    // index among the NW waves that share it
    int kpack[PPW];                                                    // byte offset inside the image | halo column << 26 (63 = no pixel)
#pragma unroll
    for (int i = 0; i < PPW; i++) {
        const int s = (mw * PPW + i) * 64 + lane;
        const int row = s / W::CPR, rem = s % W::CPR;
        const int px = rem / W::CPP, ch = rem % W::CPP;
        const bool real = row < C::HR && px < HC && ch < W::DPP;
        kpack[i] = real ? ((((row * a.win + px) * CIN + ch * VEC) * (int)sizeof(T)) | (px << 26)) : (int)(63u << 26);
    }
    const unsigned in_img_bytes = (unsigned)(a.hin * a.win * CIN) * (unsigned)sizeof(T);
    const unsigned out_img_bytes = (unsigned)(a.hout * a.wout * COUT) * (unsigned)sizeof(T);
    auto dma = [&](int tx, int ty, int img, T *dst) {
        const auto rs = image_rsrc(a.in + (size_t)img * a.hin * a.win * CIN, in_img_bytes);
        const int iy0 = ty * TH * STRIDE - 1, ix0 = tx * TW * STRIDE - 1;
        const int sbase = (iy0 * a.win + ix0) * CIN * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < PPW; i++) {
            const int dx = (int)((unsigned)kpack[i] >> 26);
            const unsigned off = (dx != 63 && (unsigned)(ix0 + dx) < (unsigned)a.win) ? (unsigned)((kpack[i] & 0x03ffffff) + sbase) : kOobOffset;
            lds_dma16(rs, (unsigned char *)dst + (mw * PPW + i) * 1024, off);
        }
    };
    auto store_tile = [&](int img, int oy0, int ox0) {
        constexpr int OPV = COUT / VEC;
        const auto ro = image_rsrc(a.out + (size_t)img * a.hout * a.wout * COUT, out_img_bytes);
        const int obase = (oy0 * a.wout + ox0) * COUT * (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < W::NST_OUT / NW; j++) {
            const int i = lane + 64 * (mw + NW * j);
            const int p = i / OPV, cv = i % OPV;
            const int py = p / TW, px = p % TW;
            const unsigned off = ox0 + px < a.wout ? (unsigned)(((py * a.wout + px) * COUT + cv * VEC) * (int)sizeof(T) + obase) : kOobOffset;
            buf_store16(ro, off, *(const V *)(s_out + p * LDO + cv * VEC));
        }
        if constexpr (LAT) {
            constexpr int LPV = 64 / VEC;
            const auto rl = image_rsrc(a.lat_out + (size_t)img * a.hout * a.wout * 64, (unsigned)(a.hout * a.wout * 64) * (unsigned)sizeof(T));
            const int lbase = (oy0 * a.wout + ox0) * 64 * (int)sizeof(T);
#pragma unroll
            for (int j = 0; j < W::NST_LAT / NW; j++) {
                const int i = lane + 64 * (mw + NW * j);
                const int p = i / LPV, cv = i % LPV;
                const int py = p / TW, px = p % TW;
                const unsigned off = ox0 + px < a.wout ? (unsigned)(((py * a.wout + px) * 64 + cv * VEC) * (int)sizeof(T) + lbase) : kOobOffset;
                buf_store16(rl, off, *(const V *)(s_lat + p * LDL + cv * VEC));
            }
        }
    };
    const TileStep step(G, a.tiles_x, a.tiles_y);
    TileCoord cur(first, a.tiles_x, a.tiles_y), pf = cur;             // cur: the tile being computed; pf: the next tile to fetch
    int p_img = 0, p_oy0 = 0, p_ox0 = 0;                               // the tile whose result is still in LDS
    // what may stay in flight when tile k + 1 has to have landed: this interval's own stores and DMA (vmcnt is in order)
    auto wait_next_halo = [&](bool st, bool ld) {
        if (DIST == 1) wait_vmcnt<0>();
        else if (st && ld) wait_vmcnt<(W::NSTW + PPW < 63 ? W::NSTW + PPW : 63)>();
        else if (ld) wait_vmcnt<PPW>();
        else if (st) wait_vmcnt<W::NSTW>();
        else wait_vmcnt<0>();
    };

    if constexpr (PROD) {
      if (wave == 4) {
        // ================================================= producer
#pragma unroll
        for (int d = 0; d < DIST; d++)
            if (d < n_my) { dma(pf.tx, pf.ty, pf.img, s_in + d * IN_ELEMS); step.advance(pf); }
        if (DIST == 2 && n_my >= 2) wait_vmcnt<PPW>();
        else wait_vmcnt<0>();
        lds_barrier();
        for (int k = 0; k < n_my; k++) {
            const bool st = k > 0, ld = k + DIST < n_my;
            // the previous tile's result(s): s_out / s_lat are rewritten by this interval's pointwise / lateral phase, i.e. after the
            // barrier below, which this wave passes only with its LDS reads complete
            if (st) store_tile(p_img, p_oy0, p_ox0);
            lds_barrier();                                             // (depthwise -> pointwise)
            if (ld) { dma(pf.tx, pf.ty, pf.img, s_in + ((k + DIST) % NBUF) * IN_ELEMS); step.advance(pf); }      // (issued under the pointwise phase)
            p_img = cur.img; p_oy0 = cur.ty * TH; p_ox0 = cur.tx * TW;
            step.advance(cur);
            if constexpr (LAT) lds_barrier();                          // (pointwise -> lateral)
            // tile k + 1 must have landed before the closing barrier; what was issued in THIS interval may stay in flight (DIST == 2)
            wait_next_halo(st, ld);
            lds_barrier();
        }
        if (n_my > 0) store_tile(p_img, p_oy0, p_ox0);                // (the consumers' last writes precede the closing barrier)
        return;
      }
    } else {
        // K_b'': every wave brings in its quarter of the first DIST halos; the wait below (weights + these) and the barrier publish them
#pragma unroll
        for (int d = 0; d < DIST; d++)
            if (d < n_my) { dma(pf.tx, pf.ty, pf.img, s_in + d * IN_ELEMS); step.advance(pf); }
    }

    // ================================================= consumers (waves 0-3): once per workgroup, weights and per-lane constants as in K_b
    const int wn = wave % WS::WN, wp = wave / WS::WN;
    Frag wst[WS::NI][KCH];
    {
        const Frag *wsrc = (const Frag *)a.pw_w + (size_t)wn * KCH * 64 + lane;
#pragma unroll
        for (int i = 0; i < WS::NI; i++)
#pragma unroll
            for (int kc = 0; kc < KCH; kc++) wst[i][kc] = wsrc[((i * WS::WN) * KCH + kc) * 64];
    }
    f32x4 pw_bias[WS::NI], pw_mult[WS::NI];
#pragma unroll
    for (int i = 0; i < WS::NI; i++) {
        pw_bias[i] = *(const f32x4 *)(a.pw_b + acc_cout(wn + i * WS::WN, lane, 0));
        pw_mult[i] = load_mult(a.pw_m, acc_cout(wn + i * WS::WN, lane, 0));
    }
    constexpr int LKCH = (COUT + M::K - 1) / M::K;
    Frag lst[1][LAT ? LKCH : 1];
    f32x4 lat_bias = vzero<f32x4, 4>(), lat_mult = vzero<f32x4, 4>();
    if constexpr (LAT) {
        const Frag *lsrc = (const Frag *)a.lat_w + (size_t)wave * LKCH * 64 + lane;
#pragma unroll
        for (int kc = 0; kc < LKCH; kc++) lst[0][kc] = lsrc[kc * 64];
        lat_bias = *(const f32x4 *)(a.lat_b + acc_cout(wave, lane, 0));
        lat_mult = load_mult(a.lat_m, acc_cout(wave, lane, 0));
    }
    constexpr int NG = CIN / 16, DKCH = I8 ? kDwMmaChunksI8 : kDwMmaChunks;
    constexpr int DPARTS = I8 ? 2 : 1;
    constexpr int GW = NG >= 4 ? NG / 4 : 1;
    constexpr int PW = (NG * PT / 4) / GW;
    uint32_t dwv[GW][DKCH][DPARTS];
    f32x4 dwb4[GW], dwm4[GW];
    int dpix[PW], dtap[DKCH];
    const int dsel = I8 ? dw_mma_dword_index_i8(lane) : dw_mma_dword_index(lane);
#pragma unroll
    for (int gi = 0; gi < GW; gi++) {
        const int g = NG >= 4 ? wave + 4 * gi : wave % NG;
#pragma unroll
        for (int kc = 0; kc < DKCH; kc++)
#pragma unroll
            for (int hl = 0; hl < DPARTS; hl++) dwv[gi][kc][hl] = a.dw_mma[((g * DKCH + kc) * DPARTS + hl) * 64 + lane];
        dwb4[gi] = *(const f32x4 *)(a.dw_b + acc_cout(g, lane, 0));
        dwm4[gi] = load_mult(I8 ? a.dw_m : nullptr, acc_cout(g, lane, 0));
    }
#pragma unroll
    for (int pi = 0; pi < PW; pi++) {
        const int pt = NG >= 4 ? pi : wave / NG + pi * (4 / NG);
        const int p = acc_pixel(pt, lane);
        dpix[pi] = (p / TW) * STRIDE * ROWP + (p % TW) * STRIDE * LDIN + (I8 ? 0 : ((lane >> 4) & 1) * 8);
    }
#pragma unroll
    for (int kc = 0; kc < DKCH; kc++) {
        const int tap = I8 ? kc * 4 + (lane >> 4) : kc * 2 + (lane >> 5);
        dtap[kc] = tap < 9 ? (tap / 3) * ROWP + (tap % 3) * LDIN : -1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                // everything above is in registers: no global access below this line
    lds_barrier();                                                     // (the producer's prologue: tile 0 has landed)

    for (int k = 0; k < n_my; k++) {
        const T *s_in_b = s_in + (k % NBUF) * IN_ELEMS;
        const bool st = k > 0, ld = k + DIST < n_my;
        if constexpr (!PROD) {
            // this wave's quarter of the previous tile's stores (s_out / s_lat are rewritten after the next barrier, which the wave passes only
            // with these LDS reads complete) and of the halo DIST tiles ahead (its buffer was last read before the previous interval's first barrier)
            if (st) store_tile(p_img, p_oy0, p_ox0);
            if (ld) { dma(pf.tx, pf.ty, pf.img, s_in + ((k + DIST) % NBUF) * IN_ELEMS); step.advance(pf); }
            p_img = cur.img; p_oy0 = cur.ty * TH; p_ox0 = cur.tx * TW;
            step.advance(cur);
        }
        // ---- depthwise 3x3 as diagonal-weight implicit GEMM (K_b phase 2)
#pragma unroll
        for (int gi = 0; gi < GW; gi++) {
            const int g = NG >= 4 ? wave + 4 * gi : wave % NG;
            typename M::Acc dacc[DPARTS][PW];
#pragma unroll
            for (int hl = 0; hl < DPARTS; hl++)
#pragma unroll
                for (int pi = 0; pi < PW; pi++) dacc[hl][pi] = acc_init<T>(dwb4[gi]);
            constexpr int NB = DKCH * PW, DDEPTH = NB < 4 ? NB : 4;
            Frag bq[DDEPTH];
            auto bload = [&](int idx) -> Frag {
                const int kc = idx / PW, pi = idx % PW;
                return dtap[kc] >= 0 ? *(const Frag *)(s_in_b + dpix[pi] + dtap[kc] + g * 16) : M::zero();
            };
#pragma unroll
            for (int d = 0; d < DDEPTH - 1; d++) bq[d] = bload(d);
#pragma unroll
            for (int kc = 0; kc < DKCH; kc++) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                Frag af[DPARTS];
#pragma unroll
                for (int hl = 0; hl < DPARTS; hl++) {
                    u32x4 wa;
                    uint32_t wd = dwv[gi][kc][hl];
                    asm volatile("" : "+v"(wd));
#pragma unroll
                    for (int d = 0; d < 4; d++) wa[d] = dsel == d ? wd : 0u;
                    af[hl] = __builtin_bit_cast(Frag, wa);
                }
#pragma unroll
                for (int pi = 0; pi < PW; pi++) {
                    const int idx = kc * PW + pi;
                    if (idx + DDEPTH - 1 < NB) bq[(idx + DDEPTH - 1) % DDEPTH] = bload(idx + DDEPTH - 1);
#pragma unroll
                    for (int hl = 0; hl < DPARTS; hl++) dacc[hl][pi] = M::mma(af[hl], bq[idx % DDEPTH], dacc[hl][pi]);
                }
            }
#pragma unroll
            for (int pi = 0; pi < PW; pi++) {
                const int pt = NG >= 4 ? pi : wave / NG + pi * (4 / NG);
                if constexpr (I8) {
                    typename M::Acc tot;
#pragma unroll
                    for (int r = 0; r < 4; r++) tot[r] = dacc[0][pi][r] * 128 + dacc[DPARTS - 1][pi][r];
                    store_mid_u8<LDA>(s_a, dwm4[gi], dwb4[gi], tot, g, pt, lane);
                } else {
                    store_acc<T, LDA>(s_a, dwm4[gi], dwb4[gi], dacc[0][pi], g, pt, lane, true);
                }
            }
        }
        lds_barrier();
        // ---- pointwise GEMM + epilogue (K_b phases 3-4)
        {
            typename M::Acc acc[WS::NI][WS::NJ];
#pragma unroll
            for (int i = 0; i < WS::NI; i++)
#pragma unroll
                for (int j = 0; j < WS::NJ; j++) acc[i][j] = acc_init<T>(pw_bias[i]);
            gemm_stationary<T, WS::NI, WS::NJ, KCH>(acc, wst, [&](int j, int kc) -> Frag {
                const int kb = kc * M::K + (lane >> 4) * M::KPL;
                const int p = acc_pixel(wp + j * WS::WP, lane);
                return kb < CIN ? *(const Frag *)(s_a + p * LDA + kb) : M::zero();
            });
#pragma unroll
            for (int i = 0; i < WS::NI; i++)
#pragma unroll
                for (int j = 0; j < WS::NJ; j++)
                    store_acc<T, LDO>(s_out, pw_mult[i], pw_bias[i], acc[i][j], wn + i * WS::WN, wp + j * WS::WP, lane, true);
        }
        if constexpr (LAT) {
            lds_barrier();
            // ---- fused lateral on the LDS-resident output tile (K_b phase 5): wave = output-channel tile, all pixel tiles
            typename M::Acc acc2[1][PT];
#pragma unroll
            for (int j = 0; j < PT; j++) acc2[0][j] = acc_init<T>(lat_bias);
            gemm_stationary<T, 1, PT, LKCH>(acc2, lst, [&](int j, int kc) -> Frag {
                const int kb = kc * M::K + (lane >> 4) * M::KPL;
                return *(const Frag *)(s_out + acc_pixel(j, lane) * LDO + kb);
            });
#pragma unroll
            for (int j = 0; j < PT; j++) store_acc<T, LDL>(s_lat, lat_mult, lat_bias, acc2[0][j], wave, j, lane, true);
        }
        if constexpr (!PROD) wait_next_halo(st, ld);                   // this wave's quarter of tile k + 1; the barrier publishes all four
        lds_barrier();
    }
    if constexpr (!PROD) { if (n_my > 0) store_tile(p_img, p_oy0, p_ox0); }
}

// probe knob RF_DWPWWS: 0 = off (K_b everywhere); 2 / 3 = halo buffers of the warp-specialised blocks
static int dwpw_ws_variant() { return knob(K_DWPWWS); }

template <typename T, int CIN, int COUT, int STRIDE, int TH, int TW, bool LAT, int NBUF, bool PADROW, bool PROD = true>
static void dwpw_ws_launch(hipStream_t s, const DwPwParams<T> *p, int tiles_x, int tiles_y) {
    typedef DwPwWsCfg<T, CIN, COUT, STRIDE, TH, TW, LAT, NBUF, PADROW, PROD> W;
    auto kern = dwpw_ws_kernel<T, CIN, COUT, STRIDE, TH, TW, LAT, NBUF, PADROW, PROD>;
    static std::atomic<int> resident_cache[kMaxDevices] = {};
    const int dev = launch_device();
    int resident = resident_cache[dev].load(std::memory_order_acquire);
    if (!resident) {
        set_max_lds(kern, W::LDS_BYTES);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, W::THREADS, W::LDS_BYTES) != hipSuccess || nb < 1) nb = 1;
        resident = nb;
        resident_cache[dev].store(resident, std::memory_order_release);
    }
    DwPwArgs<T> a{p->in, p->out, p->dw_w, p->dw_b, p->dw_mma, p->pw_w, p->pw_b, p->lat_w, p->lat_b, p->lat_out, p->pw_m, p->lat_m, p->dw_m,
                  p->hin, p->win, p->hout, p->wout, tiles_x, tiles_y, p->n * tiles_x * tiles_y};
    hipLaunchKernelGGL(kern, dim3(persistent_grid(a.nblk, resident)), dim3(W::THREADS), W::LDS_BYTES, s, a);
}
#endif  // RF_PROBES

template <typename T, int CIN, int COUT, int STRIDE, bool HAS_DW, int TH, int TW, bool LAT, bool PADROW>
static void dwpw_launch(hipStream_t s, const DwPwParams<T> *p, int tiles_x, int tiles_y) {
    typedef DwPwCfg<T, CIN, COUT, STRIDE, HAS_DW, TH, TW, PADROW> C;
    auto kern = dwpw_kernel<T, CIN, COUT, STRIDE, HAS_DW, TH, TW, LAT, PADROW>;
    static std::atomic<int> resident_cache[kMaxDevices] = {};
    const int resident = kernel_residency(resident_cache, kern, C::LDS_BYTES);
    DwPwArgs<T> a{p->in, p->out, p->dw_w, p->dw_b, p->dw_mma, p->pw_w, p->pw_b, p->lat_w, p->lat_b, p->lat_out, p->pw_m, p->lat_m, p->dw_m,
                  p->hin, p->win, p->hout, p->wout, tiles_x, tiles_y, p->n * tiles_x * tiles_y};
    const int grid = sizeof(T) <= 2 ? persistent_grid(a.nblk, resident) : a.nblk;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), C::LDS_BYTES, s, a);
}

template <typename T, int CIN, int COUT, int STRIDE, bool HAS_DW, int TH, int TW, bool PADROW = false>
static TileInfo dwpw_dispatch(hipStream_t s, const DwPwParams<T> *p, int hout, int wout) {
    typedef DwPwCfg<T, CIN, COUT, STRIDE, HAS_DW, TH, TW, PADROW> C;
    // halo rows padded to the bank row (PADROW) wherever a 16-pixel MFMA tile spans two halo rows; RF_DWPAD=0 (probe knob): round-1 layout
    constexpr bool WANTS_PADROW = !PADROW && sizeof(T) <= 2 && HAS_DW && STRIDE == 1 && TW == 8 && CIN >= 32;
#ifdef RF_PROBES
    if constexpr (WANTS_PADROW) {
        if (p && knob(K_DWPAD)) return dwpw_dispatch<T, CIN, COUT, STRIDE, HAS_DW, TH, TW, true>(s, p, hout, wout);
    }
#else
    if constexpr (WANTS_PADROW) return dwpw_dispatch<T, CIN, COUT, STRIDE, HAS_DW, TH, TW, true>(s, p, hout, wout);
    else {
#endif
    int tiles_x = (wout + TW - 1) / TW, tiles_y = (hout + TH - 1) / TH;
    TileInfo ti{TH, TW, C::LDS_BYTES, tiles_x * tiles_y};
    if (!p) return ti;
#ifdef RF_PROBES
    // warp-specialised instances (K_b'): the stride-1 blocks with 64 / 128 channels, with or without the fused lateral
    if constexpr (sizeof(T) <= 2 && HAS_DW && STRIDE == 1 && CIN == COUT && (CIN == 64 || CIN == 128) && C::DWMMA && C::STAT) {
        const int v = dwpw_ws_variant();
        if (v == 12 || v == 13) {           // K_b'': the memory side spread over the four GEMM waves (no producer wave)
            if constexpr (DwPwWsCfg<T, CIN, COUT, STRIDE, TH, TW, true, 2, PADROW, false>::SPLIT_OK) {
                if (p->lat_out) {
                    if (v == 12) dwpw_ws_launch<T, CIN, COUT, STRIDE, TH, TW, true, 2, PADROW, false>(s, p, tiles_x, tiles_y);
                    else dwpw_ws_launch<T, CIN, COUT, STRIDE, TH, TW, true, 3, PADROW, false>(s, p, tiles_x, tiles_y);
                    return ti;
                }
            }
            if constexpr (DwPwWsCfg<T, CIN, COUT, STRIDE, TH, TW, false, 2, PADROW, false>::SPLIT_OK) {
                if (!p->lat_out) {
                    if (v == 12) dwpw_ws_launch<T, CIN, COUT, STRIDE, TH, TW, false, 2, PADROW, false>(s, p, tiles_x, tiles_y);
                    else dwpw_ws_launch<T, CIN, COUT, STRIDE, TH, TW, false, 3, PADROW, false>(s, p, tiles_x, tiles_y);
                    return ti;
                }
            }
        }
        if (v == 2 || v == 3) {
            if (p->lat_out) {
                if (v == 2) dwpw_ws_launch<T, CIN, COUT, STRIDE, TH, TW, true, 2, PADROW>(s, p, tiles_x, tiles_y);
                else dwpw_ws_launch<T, CIN, COUT, STRIDE, TH, TW, true, 3, PADROW>(s, p, tiles_x, tiles_y);
            } else {
                if (v == 2) dwpw_ws_launch<T, CIN, COUT, STRIDE, TH, TW, false, 2, PADROW>(s, p, tiles_x, tiles_y);
                else dwpw_ws_launch<T, CIN, COUT, STRIDE, TH, TW, false, 3, PADROW>(s, p, tiles_x, tiles_y);
            }
            return ti;
        }
    }
#endif
    // K_b(8) for the 128-channel blocks, where it measured faster (A/B inside one call each, tools/gpu/r5.sh c4 / c7, profiles/r05_wide_blocks_ab.txt;
    // every variant bit-identical, tools/probes/knob_equal.py):
    //   fp16: the block WITH the fused lateral (conv21 + conv22 + rf_c2_lateral) -- at 126 VGPRs two 8-wave workgroups share a CU (16 waves) where K_b's
    //         182-VGPR build has two 4-wave ones: 34.8 -> 33.0 us; the plain blocks 27.7 -> 28.0 us: they stay on K_b;
    //   int8: the other way round -- the four plain blocks 25.0 -> 23.1 us each (121 VGPRs), the lateral block 29.0 -> 30.6 us (148 VGPRs: one workgroup per CU).
    // Probe knobs: RF_WIDE128 (fp16) 0 = none, 1 = lateral block, 2 = all five; RF_WIDE_I8 bit 0 = 256-channel block, bit 1 = plain blocks, bit 2 = lateral block.
    if constexpr (HAS_DW && STRIDE == 1 && CIN == 128 && COUT == 128 && ((sizeof(T) == 2 && TH == 4 && (TW == 8 || (kProbeBuild && TW == 16))) || (sizeof(T) == 1 && TH == 4 && TW == 16))) {
        constexpr bool I8W = sizeof(T) == 1;
        const int v = I8W ? knob(K_WIDE_I8) : knob(K_WIDE128);
        const bool lat_wide = I8W ? (v & 4) != 0 : v >= 1, plain_wide = I8W ? (v & 2) != 0 : v == 2;
        if constexpr (!I8W || kProbeBuild) {
            if (lat_wide && p->lat_out) { dwpw_wide_launch<T, CIN, COUT, TH, TW, true, PADROW, 2>(s, p, tiles_x, tiles_y); return ti; }
        }
        if constexpr (I8W || kProbeBuild) {
            if (plain_wide && !p->lat_out) { dwpw_wide_launch<T, CIN, COUT, TH, TW, false, PADROW, 2>(s, p, tiles_x, tiles_y); return ti; }
        }
    }
    if (p->lat_out) {
        // laterals tap the outputs of blocks 4 (64ch), 10 (128ch) and 12 (256ch)
        // fp16 256-channel block: eight waves, every weight stationary (K_b(8), round 5); RF_WIDE256=0 (probe knob): K_b with the streamed matrix
        if constexpr (HAS_DW && STRIDE == 1 && CIN == 256 && COUT == 256 && ((sizeof(T) == 2 && TH == 4 && TW == 8) || (sizeof(T) == 1 && TH == 8 && TW == 8))) {
            // fp16 37.3 -> 26.6 us (tools/gpu/r5.sh c2), int8 23.6 -> 20.4 us (c7)
            if (sizeof(T) == 2 ? knob(K_WIDE256) != 0 : (knob(K_WIDE_I8) & 1) != 0) { dwpw_wide_launch<T, CIN, COUT, TH, TW, true, PADROW>(s, p, tiles_x, tiles_y); return ti; }
        }
        if constexpr (HAS_DW && STRIDE == 1 && CIN == COUT && COUT >= 64) dwpw_launch<T, CIN, COUT, STRIDE, HAS_DW, TH, TW, true, PADROW>(s, p, tiles_x, tiles_y);
        else throw LaunchUnsupported("fused lateral: only stride-1 blocks with cin == cout >= 64 have a kernel instance");
    } else {
        dwpw_launch<T, CIN, COUT, STRIDE, HAS_DW, TH, TW, false, PADROW>(s, p, tiles_x, tiles_y);
    }
    return ti;
#ifndef RF_PROBES
    }
#endif
}

// the layer table of SURVEY.md App. A -> tile geometry.  The product build holds ONE tile shape per (precision, layer) -- the winner of the A/B
// measurements cited below -- and only the layers that precision's engine launches through K_b; the probe build (RF_PROBES) adds the shapes that lost,
// selected by the RF_TILE* knobs, the fp16 instances of the front blocks (RF_STEM2=0 / RF_DWPW2=0 un-fuse them) and the plain 1x1 instances.
template <typename T>
static TileInfo dwpw_select(hipStream_t s, const DwPwParams<T> *p, int cin, int cout, int stride, bool has_dw, int hout,
                            int wout) {
#define RF_DWPW(CI, CO, ST, DW, TH_, TW_) \
    if (cin == CI && cout == CO && stride == ST && has_dw == DW) return dwpw_dispatch<T, CI, CO, ST, DW, TH_, TW_>(s, p, hout, wout);
    constexpr bool I8 = sizeof(T) == 1;
#ifdef RF_PROBES
    constexpr bool FRONT = sizeof(T) >= 2;       // fp16 with RF_STEM2=0 / RF_DWPW2=0 runs blocks 0-3 through K_b as well
#else
    constexpr bool FRONT = sizeof(T) == 4;                  // fp16: blocks 0-3 live in stem2 / dwpw2; int8: block 0 lives in the stem, 1-3 have their own shapes below
#endif
    if constexpr (FRONT) { RF_DWPW(8, 16, 1, true, 8, 32) }
    if constexpr (I8) {
        // Tile shapes of the int8 engine's big-map blocks (fp16 runs them inside stem2 / dwpw2).  These kernels spend their time in the requantising
        // epilogue (VALU-active 0.5-0.64 of the chip), which is per element, so the tile shape moves little; measured one by one inside one call
        // (tools/gpu/rounds_3_4.sh r4_call18, us per 256 images, all bit-identical): 16->32 s2 8x8 56.2 | 8x16 53.5 | 16x16 57.2;  32->32 8x8 68.6 | 8x16 67.1 |
        // 16x16 69.6;  32->64 s2 4x8 32.7 | 8x8 30.5 | 8x16 34.4;  64->128 s2 4x8 18.6 | 8x8 22.1.  The product = the best of each.
#ifdef RF_PROBES
        const int va = knob(K_TILE_A), vb = knob(K_TILE_B), vc = knob(K_TILE_C), vd = knob(K_TILE_D);      // 0 = round-3 shapes
        if (va == 2) { RF_DWPW(16, 32, 2, true, 16, 16) }
        if (va == 0) { RF_DWPW(16, 32, 2, true, 8, 8) }
        if (vb == 2) { RF_DWPW(32, 32, 1, true, 16, 16) }
        if (vb == 0) { RF_DWPW(32, 32, 1, true, 8, 8) }
        if (vc == 2) { RF_DWPW(32, 64, 2, true, 8, 16) }
        if (vc == 0) { RF_DWPW(32, 64, 2, true, 4, 8) }
        if (vd == 1) { RF_DWPW(64, 128, 2, true, 8, 8) }
#endif
        RF_DWPW(16, 32, 2, true, 8, 16)
        RF_DWPW(32, 32, 1, true, 8, 16)
        RF_DWPW(32, 64, 2, true, 8, 8)
    }
    if constexpr (FRONT) {
        RF_DWPW(16, 32, 2, true, 8, 8)
        RF_DWPW(32, 32, 1, true, 8, 8)
        RF_DWPW(32, 64, 2, true, 4, 8)
    }
#ifdef RF_PROBES
    if constexpr (sizeof(T) <= 2) {
        const int v64 = knob(K_TILE64);
        if (v64 == 1) { RF_DWPW(64, 64, 1, true, 8, 8) }
        if (v64 == 2) { RF_DWPW(64, 64, 1, true, 8, 16) }
    }
#endif
    RF_DWPW(64, 64, 1, true, 4, 8)
    RF_DWPW(64, 128, 2, true, 4, 8)
    // 128-channel blocks: 4x16 tiles for the int8 engine (26.0 -> 24.9 us each, tools/gpu/rounds_3_4.sh r4_call16, r4_call18); fp16: 27.7 -> 28.3, stays 4x8.
    // RF_TILE128 = 0 / 1 / 2 / 3 forces 4x8 / 8x8 / 4x16 / 8x16 (probe knob; -1 = the per-precision default)
#ifdef RF_PROBES
    if constexpr (sizeof(T) <= 2) {
        const int v128 = knob(K_TILE128) < 0 ? (I8 ? 2 : 0) : knob(K_TILE128);
        if (v128 == 1) { RF_DWPW(128, 128, 1, true, 8, 8) }
        if (v128 == 2) { RF_DWPW(128, 128, 1, true, 4, 16) }
        if (v128 == 3) { RF_DWPW(128, 128, 1, true, 8, 16) }
        RF_DWPW(128, 128, 1, true, 4, 8)
    }
#endif
    if constexpr (I8) { RF_DWPW(128, 128, 1, true, 4, 16) }
    else { RF_DWPW(128, 128, 1, true, 4, 8) }
    RF_DWPW(128, 256, 2, true, 4, 8)
    // 256-channel block, 8x8 tiles: the streamed 256 x 256 weight matrix is read once per 64 pixels instead of 32.  Measured (tools/gpu/rounds_3_4.sh r4_call13):
    // int8 26.4 -> 24.4 us, fp16 35.5 -> 35.6 (its 64 accumulator registers on top of the fragment stream: 2-14 spills with the lateral): int8 only.
    // RF_TILE256 = 0 / 1 forces 4x8 / 8x8 (probe knob; 2 = the per-precision default)
#ifdef RF_PROBES
    if constexpr (sizeof(T) <= 2) {
        const int v256 = knob(K_TILE256);
        if (v256 == 1) { RF_DWPW(256, 256, 1, true, 8, 8) }
        if (v256 == 0) { RF_DWPW(256, 256, 1, true, 4, 8) }
    }
#endif
    if constexpr (I8) { RF_DWPW(256, 256, 1, true, 8, 8) }
    else { RF_DWPW(256, 256, 1, true, 4, 8) }
#ifdef RF_PROBES
    RF_DWPW(256, 64, 1, false, 4, 8)          // plain 1x1 instances (the engines fuse the laterals into the producing block)
    RF_DWPW(128, 64, 1, false, 4, 8)
    RF_DWPW(64, 64, 1, false, 8, 8)
#endif
#undef RF_DWPW
    return TileInfo{0, 0, 0, 0};
}

template <typename T> void launch_dwpw(hipStream_t s, const DwPwParams<T> &p) {
    TileInfo ti = dwpw_select<T>(s, &p, p.cin, p.cout, p.stride, p.has_dw, p.hout, p.wout);
    if (ti.th == 0) throw LaunchUnsupported("no depthwise/pointwise kernel instance for this layer shape");
}
template <typename T> TileInfo dwpw_tile_info(int cin, int cout, int stride, bool has_dw, int hout, int wout) {
    return dwpw_select<T>(nullptr, nullptr, cin, cout, stride, has_dw, hout, wout);
}
template void launch_dwpw<half_t>(hipStream_t, const DwPwParams<half_t> &);
template void launch_dwpw<float>(hipStream_t, const DwPwParams<float> &);
template void launch_dwpw<int8_t>(hipStream_t, const DwPwParams<int8_t> &);
template TileInfo dwpw_tile_info<half_t>(int, int, int, bool, int, int);
template TileInfo dwpw_tile_info<float>(int, int, int, bool, int, int);
template TileInfo dwpw_tile_info<int8_t>(int, int, int, bool, int, int);

// =============================================================================================
// K_b2  two backbone blocks in one launch: [depthwise s1 + pointwise CI -> CA] -> [depthwise s2 + pointwise CA -> CB]
//   (fp16 engine, conv5..conv8: 32 -> 32 at 112^2, then 32 -> 64 down to 56^2).  Both blocks ran at the measured HBM copy rate
//   (208 + 155 MB per 128-image launch, PMC), so the only way to make them faster is to keep the 112^2 x 32 map between them in
//   LDS: this kernel reads the first block's input once (103 MB) and writes the second block's output (51 MB).
//   Tile = 4 x 8 outputs of the second block  <-  9 x 17 = 153 pixels of the first block's output (its stride-2 3x3 window)
//        <-  11 x 19 = 209 input pixels.  The first block is recomputed on the 1-pixel ring (153 / 128 = 1.2x).
//   Phases, one barrier between each:  1 halo -> LDS (buffer loads, hardware zero padding) | 2 depthwise A on MFMA (diagonal
//   fragments, as K_b) | 3 pointwise A, result tile fp16 with ZEROS outside the map (block B's padding) | 4 depthwise B,
//   stride 2 | 5 pointwise B | 6 coalesced store.  LDS 35 KB (regions reused once their last reader passed a barrier).
// =============================================================================================
struct DwPw2Args {
    const half_t *in; half_t *out;                      // in: [n][hin][win][32], out: [n][hout][wout][64]
    const uint32_t *dwa_mma; const float *dwa_b; const half_t *pwa_w; const float *pwa_b;
    const uint32_t *dwb_mma; const float *dwb_b; const half_t *pwb_w; const float *pwb_b;
    int hin, win, hout, wout, tiles_x, tiles_y, nblk;
    int ring;                                           // 1 = depthwise A as one ring pipeline (round 4), 0 = chunk by chunk (probe knob RF_DWPW2_RING)
};

template <bool RINGP, bool HPAD = true, bool LAY2 = true>
__global__ __launch_bounds__(kThreads, 3) void dwpw2_kernel(DwPw2Args a) {
    typedef half_t T;
    typedef Mma<T> M;
    typedef M::Frag Frag;
    typedef f16x8 V;
    constexpr int CI = 32, CB = 64;
    constexpr int TH = 4, TW = 8, P = TH * TW;                         // block-B outputs per tile
    constexpr int RH = 2 * TH + 1, RW = 2 * TW + 1, NR = RH * RW;      // block-A outputs the tile needs: 9 x 17 = 153
    constexpr int HH = RH + 2, HW = RW + 2, NH = HH * HW;              // input halo: 11 x 19 = 209
    constexpr int PTA = (NR + 15) / 16;                                // 10 MFMA pixel tiles of block A
    constexpr int UA = PTA / 2;                                        // units (pixel tiles) of block A per wave: 5
    constexpr int LD = lds_row<T>(32);                                 // 48 halfs = 96 B per pixel: conflict-free B-fragment pitch
    // LAY2 (round 4, from the same counters): every 8-byte epilogue write of 16 pixels at a 96-byte pitch is a 4-way bank conflict (24 banks per pixel:
    // four distinct bank positions), and block B's stride-2 depthwise fragments collide 2-way on the contiguous block-A tile.  The depthwise-A tile
    // therefore sits at 80 bytes per pixel (writes 2-way, the pointwise reads 2-way instead of free: 5 x 8 + 5 x 8 instead of 5 x 16 + 5 x 4 LDS cycles per
    // wave), the block-A tile at 80 bytes per pixel in rows of 88 slots (writes 2-way, block B's stride-2 reads conflict-free) and the output tile at
    // 144 bytes per pixel (writes 2-way).
    constexpr int LDO = LAY2 ? 72 : lds_row<T>(CB);                    // 72 / 80
    constexpr int LDSA = LAY2 ? 40 : LD;                               // depthwise-A result: pixel pitch
    constexpr int LDM = LAY2 ? 40 : LD, MROW = LAY2 ? 88 * 8 : RW * LD;      // block-A tile: pixel pitch, row pitch (halfs)
    // Halo rows are padded by 4 slots of 16 B (HROW = 118 slots = 6 mod 16): a depthwise-A pixel tile is 16 consecutive pixels of the 17-wide region, so
    // nearly every tile wraps from one halo row to the next; with the plain 19-pixel pitch (114 slots = 2 mod 16) the wrap shifted the second part of
    // the tile by 12 slots and its ds_read_b128 lane groups collided 2-way (SQ_LDS_BANK_CONFLICT 0.42 of this kernel's cycles, LDS busy 0.75:
    // tools/gpu/rounds_3_4.sh r4_call22); 118 makes the wrap look like 17 contiguous pixels to the bank function (6 slots per pixel).
    constexpr int HROW = HW * LD + (HPAD ? 32 : 0);                    // halfs per halo row
    constexpr int IN_BYTES = HH * HROW * 2, A_BYTES = PTA * 16 * LDSA * 2, B_BYTES = P * LD * 2, OUT_BYTES = P * LDO * 2;
    constexpr int NPF = (NH * 4 + kThreads - 1) / kThreads;            // halo items (16 B) per thread: 4
    static_assert(PTA % 2 == 0 && (LAY2 ? RH * MROW * 2 : PTA * 16 * LD * 2) <= IN_BYTES && B_BYTES + OUT_BYTES <= A_BYTES && RW * LDM <= MROW, "region reuse");
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[IN_BYTES + A_BYTES];
    T *s_in = (T *)s_raw;                              // halo                       (phases 1-2)
    T *s_mid = (T *)s_raw;                             // block-A output tile        (phases 3-4)
    T *s_a = (T *)(s_raw + IN_BYTES);                  // depthwise-A result         (phases 2-3)
    T *s_b = (T *)(s_raw + IN_BYTES);                  // depthwise-B result         (phases 4-5)
    T *s_out = (T *)(s_raw + IN_BYTES + B_BYTES);      // output tile                (phases 5-6)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const int kb = lane >> 4;
    const int dsel = dw_mma_dword_index(lane);
    const f32x4 ones = {1.f, 1.f, 1.f, 1.f};
    RF_TRACE_KEY(a.nblk);
    RF_TRACE(5, 0);

    // ---- once per workgroup.  Units are split so that a wave always works on ONE channel group / channel tile (wave & 1) and
    //      on pixel tiles (wave >> 1) + 2 i: its weights are a handful of registers, loaded here and never again
    const int g = wave & 1;
    uint32_t dwa[kDwMmaChunks], dwb[kDwMmaChunks];
#pragma unroll
    for (int kc = 0; kc < kDwMmaChunks; kc++) {
        dwa[kc] = a.dwa_mma[(g * kDwMmaChunks + kc) * 64 + lane];
        dwb[kc] = a.dwb_mma[(g * kDwMmaChunks + kc) * 64 + lane];
    }
    const Frag pwa = ((const Frag *)a.pwa_w)[g * 64 + lane];
    const Frag pwb = ((const Frag *)a.pwb_w)[wave * 64 + lane];
    const f32x4 dwa_b = *(const f32x4 *)(a.dwa_b + acc_cout(g, lane, 0)), pwa_b = *(const f32x4 *)(a.pwa_b + acc_cout(g, lane, 0));
    const f32x4 dwb_b = *(const f32x4 *)(a.dwb_b + acc_cout(g, lane, 0)), pwb_b = *(const f32x4 *)(a.pwb_b + acc_cout(wave, lane, 0));
    auto dw_frag = [&](uint32_t wd) -> Frag {          // diagonal depthwise A fragment from its one dword per lane
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        asm volatile("" : "+v"(wd));
        u32x4_ wa;
#pragma unroll
        for (int d = 0; d < 4; d++) wa[d] = dsel == d ? wd : 0u;
        return __builtin_bit_cast(Frag, wa);
    };
    // per-lane constants of the wave's block-A units: region pixel, its halo offset, its coordinates inside the region
    int pa_base[UA], pa_yx[UA];                                        // pa_yx: (ry << 16 | rx), ry poisoned for tail lanes
#pragma unroll
    for (int i = 0; i < UA; i++) {
        const int p = ((wave >> 1) + 2 * i) * 16 + (lane & 15);
        const int pc = p < NR ? p : NR - 1;                            // tail lanes recompute the last pixel; never used past NR
        pa_yx[i] = ((p < NR ? pc / RW : 0x4000) << 16) | (pc % RW);   // poisoned row: fails the inside-the-map test
        pa_base[i] = (pc / RW) * HROW + (pc % RW) * LD + g * 16 + (kb & 1) * 8;
    }
    // tap of chunk kc: 2 kc for lanes 0..31, 2 kc + 1 for lanes 32..63 (k = tap*16 + c): two compile-time offsets per chunk,
    // picked by the lane half when used (this kernel runs at 3 workgroups per CU: the 128-VGPR budget of a 4th spills 37 registers)
    const bool hi = lane >= 32;
    auto tap_a = [&](int kc) -> int { const int t0 = 2 * kc, t1 = 2 * kc + 1; return hi ? (t1 < 9 ? (t1 / 3) * HROW + (t1 % 3) * LD : -1) : (t0 / 3) * HROW + (t0 % 3) * LD; };
    auto tap_b = [&](int kc) -> int { const int t0 = 2 * kc, t1 = 2 * kc + 1; return hi ? (t1 < 9 ? (t1 / 3) * MROW + (t1 % 3) * LDM : -1) : (t0 / 3) * MROW + (t0 % 3) * LDM; };
    const int pb = (wave >> 1) * 16 + (lane & 15);                     // the wave's block-B pixel
    const int pb_base = (pb / TW) * 2 * MROW + (pb % TW) * 2 * LDM + g * 16 + (kb & 1) * 8;

    // halo of a tile -> registers (unconditional buffer loads; rows / columns outside the map read as zero = depthwise A's padding)
    V pre[NPF];
    int koff[NPF], kdx[NPF];                                           // kdx: halo column | LDS offset of the item (halfs) << 8
#pragma unroll
    for (int k = 0; k < NPF; k++) {
        int i = tid + k * kThreads;
        i = i < NH * 4 ? i : NH * 4 - 1;
        const int pix = i >> 2, cv = i & 3;
        koff[k] = (((pix / HW) * a.win + pix % HW) * CI + cv * 8) * 2;
        kdx[k] = (pix % HW) | (((pix / HW) * HROW + (pix % HW) * LD + cv * 8) << 8);
    }
    const unsigned in_img_bytes = (unsigned)(a.hin * a.win * CI) * 2u;
    auto fetch = [&](int tx, int ty, int img) {
        const auto rs = image_rsrc(a.in + (size_t)img * a.hin * a.win * CI, in_img_bytes);
        const int iy0 = 2 * ty * TH - 2, ix0 = 2 * tx * TW - 2;
        const int sbase = (iy0 * a.win + ix0) * CI * 2;
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const unsigned off = (unsigned)(ix0 + (kdx[k] & 0xff)) < (unsigned)a.win ? (unsigned)(koff[k] + sbase) : kOobOffset;
            pre[k] = buf_load16<V>(rs, off);
        }
    };
    const TileStep step(G, a.tiles_x, a.tiles_y);
    TileCoord cur(first, a.tiles_x, a.tiles_y), nxt = cur;
    step.advance(nxt);
    if (first < a.nblk) fetch(cur.tx, cur.ty, cur.img);
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the once-per-workgroup loads have landed; in the loop only the prefetch is in flight

    for (int t = first; t < a.nblk; t += G) {
        const int oy0 = cur.ty * TH, ox0 = cur.tx * TW, img = cur.img;
        // ---- phase 1: prefetched halo -> LDS, next tile's halo requested
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int i = tid + k * kThreads;
            if (i < NH * 4) *(V *)(s_in + (kdx[k] >> 8)) = pre[k];
        }
        if (t + G < a.nblk) fetch(nxt.tx, nxt.ty, nxt.img);
        cur = nxt;
        step.advance(nxt);
        RF_TRACE(5, 1);
        __syncthreads();

        // ---- phase 2: depthwise A (3x3, stride 1) on the 153-pixel region: the wave's 5 pixel tiles advance together, so five
        //      independent accumulators are in flight and each chunk's fragment is expanded once
        {
            M::Acc acc[UA];
#pragma unroll
            for (int i = 0; i < UA; i++) acc[i] = dwa_b;                 // bias rides in the accumulator (acc_init)
            if constexpr (RINGP) {
                // round 4: the 25 (chunk, pixel tile) steps as ONE software pipeline -- B fragments run RING - 1 reads ahead of their
                // MFMAs across chunk boundaries (round 3 read a chunk's five fragments, multiplied, and only then read the next chunk's:
                // one exposed LDS round trip per chunk)
                constexpr int NB = kDwMmaChunks * UA, RING = 4;
                Frag bq[RING];
                auto bload = [&](int idx) -> Frag {
                    const int kc = idx / UA, i = idx % UA;
                    return tap_a(kc) >= 0 ? *(const Frag *)(s_in + pa_base[i] + tap_a(kc)) : M::zero();
                };
#pragma unroll
                for (int d = 0; d < RING - 1; d++) bq[d] = bload(d);
#pragma unroll
                for (int kc = 0; kc < kDwMmaChunks; kc++) {
                    const Frag af = dw_frag(dwa[kc]);
#pragma unroll
                    for (int i = 0; i < UA; i++) {
                        const int idx = kc * UA + i;
                        if (idx + RING - 1 < NB) bq[(idx + RING - 1) % RING] = bload(idx + RING - 1);
                        acc[i] = M::mma(af, bq[idx % RING], acc[i]);
                    }
                }
            } else {
#pragma unroll
                for (int kc = 0; kc < kDwMmaChunks; kc++) {
                    Frag bf[UA];
#pragma unroll
                    for (int i = 0; i < UA; i++) bf[i] = tap_a(kc) >= 0 ? *(const Frag *)(s_in + pa_base[i] + tap_a(kc)) : M::zero();
                    const Frag af = dw_frag(dwa[kc]);
#pragma unroll
                    for (int i = 0; i < UA; i++) acc[i] = M::mma(af, bf[i], acc[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < UA; i++) store_acc<T, LDSA>(s_a, ones, dwa_b, acc[i], g, (wave >> 1) + 2 * i, lane, true);
        }
        RF_TRACE(5, 2);
        __syncthreads();

        // ---- phase 3: pointwise A (CI -> CA, K = 32: one MFMA per pixel tile); outside the map the tile holds block B's zero padding
        {
            Frag bf[UA];
#pragma unroll
            for (int i = 0; i < UA; i++) bf[i] = *(const Frag *)(s_a + (((wave >> 1) + 2 * i) * 16 + (lane & 15)) * LDSA + kb * 8);
#pragma unroll
            for (int i = 0; i < UA; i++) {
                const M::Acc acc = M::mma(pwa, bf[i], pwa_b);
                const int ry = pa_yx[i] >> 16, rx = pa_yx[i] & 0xffff;
                const int y = 2 * oy0 - 1 + ry, x = 2 * ox0 - 1 + rx;
                const bool inside = (unsigned)y < (unsigned)a.hin && (unsigned)x < (unsigned)a.win;
                uint2 h;
                h.x = inside ? pack_f16(acc[0], acc[1], true) : 0u;
                h.y = inside ? pack_f16(acc[2], acc[3], true) : 0u;
                if constexpr (LAY2) {
                    if (ry < 0x4000) *(uint2 *)(s_mid + ry * MROW + rx * LDM + acc_cout(g, lane, 0)) = h;     // (tail lanes own no pixel of the 2-D tile)
                } else {
                    *(uint2 *)(s_mid + (((wave >> 1) + 2 * i) * 16 + (lane & 15)) * LD + acc_cout(g, lane, 0)) = h;     // the halo is dead since the last barrier
                }
            }
        }
        RF_TRACE(5, 3);
        __syncthreads();

        // ---- phase 4: depthwise B (3x3, stride 2) on the 32 output pixels: one (group, pixel tile) unit per wave
        {
            M::Acc acc = dwb_b;
            Frag bf[kDwMmaChunks];
#pragma unroll
            for (int kc = 0; kc < kDwMmaChunks; kc++) bf[kc] = tap_b(kc) >= 0 ? *(const Frag *)(s_mid + pb_base + tap_b(kc)) : M::zero();
#pragma unroll
            for (int kc = 0; kc < kDwMmaChunks; kc++) acc = M::mma(dw_frag(dwb[kc]), bf[kc], acc);
            store_acc<T, LD>(s_b, ones, dwb_b, acc, g, wave >> 1, lane, true);      // the depthwise-A result is dead since the last barrier
        }
        RF_TRACE(5, 4);
        __syncthreads();

        // ---- phase 5: pointwise B (CA -> CB): wave = output-channel tile, both pixel tiles
        {
            Frag bf[2];
#pragma unroll
            for (int pt = 0; pt < 2; pt++) bf[pt] = *(const Frag *)(s_b + (pt * 16 + (lane & 15)) * LD + kb * 8);
#pragma unroll
            for (int pt = 0; pt < 2; pt++) store_acc<T, LDO>(s_out, ones, pwb_b, M::mma(pwb, bf[pt], pwb_b), wave, pt, lane, true);
        }
        RF_TRACE(5, 5);
        __syncthreads();

        // ---- phase 6: 32 px x 64 ch tile -> HBM: one 16-byte item per thread (s_out is next written after two more barriers)
        {
            const auto ro = image_rsrc(a.out + (size_t)img * a.hout * a.wout * CB, (unsigned)(a.hout * a.wout * CB) * 2u);
            const int p = tid >> 3, cv = tid & 7;
            const int py = p / TW, px = p % TW;
            const unsigned off = ox0 + px < a.wout ? (unsigned)((((oy0 + py) * a.wout + ox0 + px) * CB + cv * 8) * 2) : kOobOffset;
            buf_store16(ro, off, *(const V *)(s_out + p * LDO + cv * 8));
        }
        RF_TRACE(5, 6);
    }
}

#ifdef RF_PROBES      // K_b2c: measured and rejected (110.8 vs 90.8 us), probe build only
// =============================================================================================
// K_b2c  dwpw2 with the depthwise -> pointwise hops CHAINED IN REGISTERS (round 4).
//   In the swapped GEMM (D[cout][pixel]) a lane's accumulator registers hold 4 consecutive channels of the SAME pixel (column) the lane feeds
//   as a B operand: D rows 4 kb .. 4 kb + 3 of two 16-channel groups ARE the lane's 8 K slots of the next GEMM, if that GEMM's K axis is
//   ordered  k = 8 kb + e  <->  channel 4 kb + e (e < 4), 16 + 4 kb + (e - 4) (e >= 4).  The permutation is applied to the pointwise weights when
//   their A fragments are loaded (two 8-byte loads instead of one 16-byte load, once per workgroup), so the depthwise result never goes through
//   LDS: no s_a / s_b tiles, 13 -> 8 ds_write_b64 and 37 -> 30 ds_read_b128 per wave and tile, 5 -> 3 barriers.
//   Why it matters: SQ_LDS_IDX_ACTIVE said the LDS pipe of K_b2 was busy 0.75 of the kernel's time, 0.41 in bank-conflict cycles
//   (tools/gpu/rounds_3_4.sh r4_call22): 4-way conflicts of every 8-byte epilogue write at a 96-byte pixel pitch, 2-way conflicts of the depthwise reads on
//   tiles that wrap a halo row.  Layouts here: halo rows padded to 118 slots of 16 B (= 6 mod 16: a wrap looks like contiguous pixels, see K_b2);
//   the block-A tile at an 80-byte pixel pitch and 88 slots per row (its only readers are block B's stride-2 depthwise fragments: conflict-free;
//   writes 2-way instead of 4-way); the output tile at 144 bytes per pixel (2-way writes).
//   Work split: phase A  wave w -> region pixel tiles w, w + 4, w + 8 (both channel groups, both output-channel tiles: 10 + 2 MFMAs per tile);
//   phase B  wave w -> pixel tile w & 1, output-channel tiles 2 (w >> 1), +1 (the depthwise part is computed by both waves of a pixel tile).
// =============================================================================================
__global__ __launch_bounds__(kThreads, 3) void dwpw2c_kernel(DwPw2Args a) {
    typedef half_t T;
    typedef Mma<T> M;
    typedef M::Frag Frag;
    typedef f16x8 V;
    constexpr int CI = 32, CB = 64;
    constexpr int TH = 4, TW = 8, P = TH * TW;
    constexpr int RH = 2 * TH + 1, RW = 2 * TW + 1, NR = RH * RW;      // block-A outputs the tile needs: 9 x 17 = 153
    constexpr int HH = RH + 2, HW = RW + 2, NH = HH * HW;              // input halo: 11 x 19 = 209
    constexpr int PTA = (NR + 15) / 16, UA = (PTA + 3) / 4;            // 10 region pixel tiles, up to 3 per wave
    constexpr int LD = 48, HROW = HW * LD + 32;                        // halo: 96 B per pixel, 118 slots per row
    constexpr int LDM = 40, MROW = 88 * 8;                             // block-A tile: 80 B per pixel, 88 slots per row
    constexpr int LDO = 72;                                            // output tile: 144 B per pixel
    constexpr int IN_BYTES = HH * HROW * 2, MID_BYTES = RH * MROW * 2, OUT_BYTES = P * LDO * 2;
    constexpr int NPF = (NH * 4 + kThreads - 1) / kThreads;            // halo items (16 B) per thread: 4
    // Per-phase constants live in LDS, not in registers (the kernel needs 212 VGPRs with them resident, 168 is the budget of 3 workgroups per CU and at
    // 2 the chain is slower than K_b2: 127 vs 102 us): bias vectors [10][16] fp32, the chained pointwise fragments [2 + 4][64 lanes] and block B's
    // depthwise dwords [2][5][64]; each phase reads its share once per tile (4 + 2 (+ 10 narrow) reads per wave).
    constexpr int BIAS_BYTES = 10 * 16 * 4, PWF_BYTES = 6 * 64 * 16, DWB_BYTES = 2 * 2 * kDwMmaChunks * 64 * 4;
    static_assert(RW * LDM <= MROW && IN_BYTES % 16 == 0 && MID_BYTES % 16 == 0 && OUT_BYTES % 16 == 0 && BIAS_BYTES % 16 == 0, "LDS carve");
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[IN_BYTES + MID_BYTES + OUT_BYTES + BIAS_BYTES + PWF_BYTES + DWB_BYTES];
    T *s_in = (T *)s_raw;                              // halo                (written in phase 1, read in phase A)
    T *s_mid = (T *)(s_raw + IN_BYTES);                // block-A output tile (written in phase A, read in phase B)
    T *s_out = (T *)(s_raw + IN_BYTES + MID_BYTES);    // output tile         (written in phase B, read by the store)
    float *s_bias = (float *)(s_raw + IN_BYTES + MID_BYTES + OUT_BYTES);                       // [dwa 0,1 | pwa 0,1 | dwb 0,1 | pwb 0..3][16]
    Frag *s_pwf = (Frag *)(s_raw + IN_BYTES + MID_BYTES + OUT_BYTES + BIAS_BYTES);             // [pwa 0,1 | pwb 0..3][64]
    uint32_t *s_dwb = (uint32_t *)(s_raw + IN_BYTES + MID_BYTES + OUT_BYTES + BIAS_BYTES + PWF_BYTES);      // [block A | block B][2][5][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const int kb = lane >> 4, m16 = lane & 15;
    const int dsel = dw_mma_dword_index(lane);
    RF_TRACE_KEY(a.nblk);

    // ---- once per workgroup: weights.  Pointwise A fragments in the chained K order: halves 0-3 = channels 4 kb .. of group 0, 4-7 = of group 1
    for (int i = tid; i < 2 * kDwMmaChunks * 64; i += kThreads) { s_dwb[i] = a.dwa_mma[i]; s_dwb[2 * kDwMmaChunks * 64 + i] = a.dwb_mma[i]; }
    auto chained_frag = [&](const half_t *packed, int ct) -> Frag {
        const uint2 *w8 = (const uint2 *)packed;
        const uint2 lo = w8[((ct * 64) + (kb >> 1) * 16 + m16) * 2 + (kb & 1)];
        const uint2 hi = w8[((ct * 64) + (2 + (kb >> 1)) * 16 + m16) * 2 + (kb & 1)];
        const uint4 u = {lo.x, lo.y, hi.x, hi.y};
        return __builtin_bit_cast(Frag, u);
    };
    const int cp = wave >> 1;                                          // phase B: this wave's pair of output-channel tiles
    if (wave < 2) s_pwf[wave * 64 + lane] = chained_frag(a.pwa_w, wave);
    s_pwf[(2 + wave) * 64 + lane] = chained_frag(a.pwb_w, wave);
    if (tid < 160) {
        const int v = tid >> 4, c = tid & 15;
        s_bias[tid] = v < 2 ? a.dwa_b[v * 16 + c] : v < 4 ? a.pwa_b[(v - 2) * 16 + c] : v < 6 ? a.dwb_b[(v - 4) * 16 + c] : a.pwb_b[(v - 6) * 16 + c];
    }
    auto bias_of = [&](int v) -> f32x4 { return *(const f32x4 *)(s_bias + v * 16 + kb * 4); };       // (first read after the loop's first barrier)
    auto dw_frag = [&](uint32_t wd) -> Frag {          // diagonal depthwise A fragment from its one dword per lane
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        asm volatile("" : "+v"(wd));
        u32x4_ wa;
#pragma unroll
        for (int d = 0; d < 4; d++) wa[d] = dsel == d ? wd : 0u;
        return __builtin_bit_cast(Frag, wa);
    };
    auto chain = [&](const M::Acc &g0, const M::Acc &g1) -> Frag {     // two ReLU'd fp16 accumulator tiles = the next GEMM's B fragment
        const uint4 u = {pack_f16(g0[0], g0[1], true), pack_f16(g0[2], g0[3], true), pack_f16(g1[0], g1[1], true), pack_f16(g1[2], g1[3], true)};
        return __builtin_bit_cast(Frag, u);
    };
    // per-lane constants of the wave's phase-A pixel tiles
    int pa_base[UA], pa_yx[UA];                                        // pa_yx: (ry << 16 | rx), ry poisoned for tail lanes (p >= NR)
#pragma unroll
    for (int i = 0; i < UA; i++) {
        const int p = (wave + 4 * i) * 16 + m16;
        const int pc = p < NR ? p : NR - 1;
        const int ry = pc / RW, rx = pc % RW;
        pa_yx[i] = ((p < NR ? ry : 0x4000) << 16) | rx;
        pa_base[i] = ry * HROW + rx * LD + (kb & 1) * 8;
    }
    const bool hi = lane >= 32;                                        // chunk kc carries tap 2 kc (lanes 0..31) and 2 kc + 1 (lanes 32..63)
    auto tap_a = [&](int kc) -> int { const int t0 = 2 * kc, t1 = 2 * kc + 1; return hi ? (t1 < 9 ? (t1 / 3) * HROW + (t1 % 3) * LD : -1) : (t0 / 3) * HROW + (t0 % 3) * LD; };
    auto tap_b = [&](int kc) -> int { const int t0 = 2 * kc, t1 = 2 * kc + 1; return hi ? (t1 < 9 ? (t1 / 3) * MROW + (t1 % 3) * LDM : -1) : (t0 / 3) * MROW + (t0 % 3) * LDM; };
    const int pb = (wave & 1) * 16 + m16;                              // phase B: the lane's output pixel
    const int pb_base = ((pb / TW) * 2) * MROW + ((pb % TW) * 2) * LDM + (kb & 1) * 8;
    const int pb_out = pb * LDO + kb * 4;

    // halo of a tile -> registers (unconditional buffer loads; rows / columns outside the map read as zero = depthwise A's padding)
    V pre[NPF];
    int kdx[NPF];                                                      // halo column | halo row << 5 | LDS offset of the item (halfs) << 9
#pragma unroll
    for (int k = 0; k < NPF; k++) {
        int i = tid + k * kThreads;
        i = i < NH * 4 ? i : NH * 4 - 1;
        const int pix = i >> 2, cv = i & 3;
        kdx[k] = (pix % HW) | ((pix / HW) << 5) | (((pix / HW) * HROW + (pix % HW) * LD + cv * 8) << 9);
    }
    const int cv16 = (tid & 3) * 16;                                   // the item's 16-byte channel chunk (kThreads is a multiple of 4)
    const unsigned in_img_bytes = (unsigned)(a.hin * a.win * CI) * 2u;
    auto fetch = [&](int tx, int ty, int img) {
        const auto rs = image_rsrc(a.in + (size_t)img * a.hin * a.win * CI, in_img_bytes);
        const int iy0 = 2 * ty * TH - 2, ix0 = 2 * tx * TW - 2;
        const int sbase = (iy0 * a.win + ix0) * CI * 2 + cv16;
        const int rowb = a.win * CI * 2;
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int col = kdx[k] & 31, row = (kdx[k] >> 5) & 15;
            const unsigned off = (unsigned)(ix0 + col) < (unsigned)a.win ? (unsigned)(row * rowb + col * (CI * 2) + sbase) : kOobOffset;
            pre[k] = buf_load16<V>(rs, off);
        }
    };
    const TileStep step(G, a.tiles_x, a.tiles_y);
    TileCoord cur(first, a.tiles_x, a.tiles_y), nxt = cur;
    step.advance(nxt);
    if (first < a.nblk) fetch(cur.tx, cur.ty, cur.img);
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the once-per-workgroup loads have landed; in the loop only the prefetch is in flight

    for (int t = first; t < a.nblk; t += G) {
        const int oy0 = cur.ty * TH, ox0 = cur.tx * TW, img = cur.img;
        // ---- phase 1: prefetched halo -> LDS, next tile's halo requested
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int i = tid + k * kThreads;
            if (i < NH * 4) *(V *)(s_in + (kdx[k] >> 9)) = pre[k];
        }
        if (t + G < a.nblk) fetch(nxt.tx, nxt.ty, nxt.img);
        cur = nxt;
        step.advance(nxt);
        __syncthreads();

        // ---- phase A: depthwise A (both channel groups) -> registers -> pointwise A (both channel tiles) -> block-A tile, zero outside the map
        {
            M::Acc acc[UA][2];
            {
                const f32x4 b0 = bias_of(0), b1 = bias_of(1);
#pragma unroll
                for (int i = 0; i < UA; i++) { acc[i][0] = b0; acc[i][1] = b1; }
            }
            // one chunk's six B fragments ahead of the MFMAs, pinned: left alone the scheduler hoists all five chunks' reads (120 registers) and spills
            Frag bf[2][UA][2];
            uint32_t dwd[2][2];
            auto load_chunk = [&](int kc, Frag (&b)[UA][2], uint32_t (&d)[2]) {
#pragma unroll
                for (int i = 0; i < UA; i++)
#pragma unroll
                    for (int g = 0; g < 2; g++) b[i][g] = tap_a(kc) >= 0 ? *(const Frag *)(s_in + pa_base[i] + tap_a(kc) + g * 16) : M::zero();
                d[0] = s_dwb[kc * 64 + lane]; d[1] = s_dwb[(kDwMmaChunks + kc) * 64 + lane];
            };
            load_chunk(0, bf[0], dwd[0]);
#pragma unroll
            for (int kc = 0; kc < kDwMmaChunks; kc++) {
                if (kc + 1 < kDwMmaChunks) load_chunk(kc + 1, bf[(kc + 1) & 1], dwd[(kc + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const Frag af0 = dw_frag(dwd[kc & 1][0]), af1 = dw_frag(dwd[kc & 1][1]);
#pragma unroll
                for (int i = 0; i < UA; i++) {
                    acc[i][0] = M::mma(af0, bf[kc & 1][i][0], acc[i][0]);
                    acc[i][1] = M::mma(af1, bf[kc & 1][i][1], acc[i][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const Frag pwa[2] = {s_pwf[lane], s_pwf[64 + lane]};
            const f32x4 pwa_b[2] = {bias_of(2), bias_of(3)};
#pragma unroll
            for (int i = 0; i < UA; i++) {                 // (pixel tiles 10 and 11 of waves 2 and 3 do not exist: computed on the last pixel, never stored)
                {
                    const Frag x = chain(acc[i][0], acc[i][1]);
                    const int ry = pa_yx[i] >> 16, rx = pa_yx[i] & 0xffff;
                    const int y = 2 * oy0 - 1 + ry, xx = 2 * ox0 - 1 + rx;
                    T *const dst = s_mid + ry * MROW + rx * LDM + kb * 4;
                    const bool inside = (unsigned)y < (unsigned)a.hin && (unsigned)xx < (unsigned)a.win;
#pragma unroll
                    for (int ct = 0; ct < 2; ct++) {
                        const M::Acc o = M::mma(pwa[ct], x, pwa_b[ct]);
                        uint2 h;
                        h.x = inside ? pack_f16(o[0], o[1], true) : 0u;
                        h.y = inside ? pack_f16(o[2], o[3], true) : 0u;
                        if (ry < 0x4000) *(uint2 *)(dst + ct * 16) = h;      // (tail lanes of the last tile own no pixel)
                    }
                }
            }
        }
        __syncthreads();

        // ---- phase B: depthwise B (stride 2, both groups) -> registers -> pointwise B (this wave's two channel tiles) -> output tile
        {
            M::Acc d0 = bias_of(4), d1 = bias_of(5);
            uint32_t dwb[2][kDwMmaChunks];
#pragma unroll
            for (int g = 0; g < 2; g++)
#pragma unroll
                for (int kc = 0; kc < kDwMmaChunks; kc++) dwb[g][kc] = s_dwb[((2 + g) * kDwMmaChunks + kc) * 64 + lane];
            const Frag pwb[2] = {s_pwf[(2 + 2 * cp) * 64 + lane], s_pwf[(3 + 2 * cp) * 64 + lane]};
            const f32x4 pwb_b[2] = {bias_of(6 + 2 * cp), bias_of(7 + 2 * cp)};
            Frag bf[kDwMmaChunks][2];
#pragma unroll
            for (int kc = 0; kc < kDwMmaChunks; kc++)
#pragma unroll
                for (int g = 0; g < 2; g++) bf[kc][g] = tap_b(kc) >= 0 ? *(const Frag *)(s_mid + pb_base + tap_b(kc) + g * 16) : M::zero();
#pragma unroll
            for (int kc = 0; kc < kDwMmaChunks; kc++) {
                d0 = M::mma(dw_frag(dwb[0][kc]), bf[kc][0], d0);
                d1 = M::mma(dw_frag(dwb[1][kc]), bf[kc][1], d1);
            }
            const Frag x = chain(d0, d1);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const M::Acc o = M::mma(pwb[j], x, pwb_b[j]);
                uint2 h;
                h.x = pack_f16(o[0], o[1], true);
                h.y = pack_f16(o[2], o[3], true);
                *(uint2 *)(s_out + pb_out + (2 * cp + j) * 16) = h;
            }
        }
        __syncthreads();

        // ---- store: 32 px x 64 ch tile -> HBM, one 16-byte item per thread (s_out is next written after two more barriers)
        {
            const auto ro = image_rsrc(a.out + (size_t)img * a.hout * a.wout * CB, (unsigned)(a.hout * a.wout * CB) * 2u);
            const int p = tid >> 3, cv = tid & 7;
            const int py = p / TW, px = p % TW;
            const unsigned off = ox0 + px < a.wout ? (unsigned)((((oy0 + py) * a.wout + ox0 + px) * CB + cv * 8) * 2) : kOobOffset;
            buf_store16(ro, off, *(const V *)(s_out + p * LDO + cv * 8));
        }
    }
}
#endif  // RF_PROBES

void launch_dwpw2(hipStream_t s, const DwPw2Params &p) {
    DwPw2Args a;
    a.in = p.in; a.out = p.out;
    a.dwa_mma = p.dwa_mma; a.dwa_b = p.dwa_b; a.pwa_w = p.pwa_w; a.pwa_b = p.pwa_b;
    a.dwb_mma = p.dwb_mma; a.dwb_b = p.dwb_b; a.pwb_w = p.pwb_w; a.pwb_b = p.pwb_b;
    a.hin = p.hin; a.win = p.win; a.hout = p.hin / 2; a.wout = p.win / 2;
    a.tiles_x = (a.wout + 7) / 8; a.tiles_y = (a.hout + 3) / 4;
    a.nblk = p.n * a.tiles_x * a.tiles_y;
    a.ring = 0;
#ifdef RF_PROBES
    // probe knobs (each measured and rejected): RF_DWPW2_RING=1: depthwise A as one ring pipeline (107 -> 232 us: 8 registers more than the
    // 168-VGPR budget of 3 workgroups per CU, spills); RF_DWPW2_CHAIN=1: K_b2c, the register-chained form (LDS cycles halved, 110.8 vs 100.5 us: longer serial
    // chains); RF_DWPW2_HPAD=0: unpadded halo rows (round 3); RF_DWPW2_LAY2=0: 96-byte pitches everywhere (round 3)
    const int ring = knob(K_DWPW2_RING);
    a.ring = ring;
    if (knob(K_DWPW2_CHAIN) && !ring) {
        static std::atomic<int> resident_cachec[kMaxDevices] = {};
        const int resident = kernel_residency(resident_cachec, dwpw2c_kernel, 0);
        hipLaunchKernelGGL(dwpw2c_kernel, dim3(persistent_grid(a.nblk, resident)), dim3(kThreads), 0, s, a);
        return;
    }
    if (ring) {
        static std::atomic<int> resident_cacher[kMaxDevices] = {};
        const int resident = kernel_residency(resident_cacher, dwpw2_kernel<true, true, false>, 0);
        hipLaunchKernelGGL((dwpw2_kernel<true, true, false>), dim3(persistent_grid(a.nblk, resident)), dim3(kThreads), 0, s, a);
        return;
    }
    if (!knob(K_DWPW2_HPAD)) {
        static std::atomic<int> resident_cache1[kMaxDevices] = {};
        const int resident = kernel_residency(resident_cache1, dwpw2_kernel<false, false, false>, 0);
        hipLaunchKernelGGL((dwpw2_kernel<false, false, false>), dim3(persistent_grid(a.nblk, resident)), dim3(kThreads), 0, s, a);
        return;
    }
    if (!knob(K_DWPW2_LAY2)) {
        static std::atomic<int> resident_cache2[kMaxDevices] = {};
        const int resident = kernel_residency(resident_cache2, dwpw2_kernel<false, true, false>, 0);
        hipLaunchKernelGGL((dwpw2_kernel<false, true, false>), dim3(persistent_grid(a.nblk, resident)), dim3(kThreads), 0, s, a);
        return;
    }
#endif
    static std::atomic<int> resident_cache0[kMaxDevices] = {};
    const int resident = kernel_residency(resident_cache0, dwpw2_kernel<false, true, true>, 0);
    hipLaunchKernelGGL((dwpw2_kernel<false, true, true>), dim3(persistent_grid(a.nblk, resident)), dim3(kThreads), 0, s, a);
}

int dwpw2_variant() { return knob(K_DWPW2); }      // probe knob RF_DWPW2: 0 = two separate K_b launches

// Input of the FPN aggregation convs: lateral + bilinear x2 upsample of the coarser level (Deconvolution k4 s2 p1 g64 + Crop + Eltwise SUM,
// prototxt :1553-1592 / :1948-1987, closed form SURVEY.md App. B.6), one 16-byte item: `lat` = the lateral's channels, u0..u3 = the four
// coarse-map taps (weights 9/16, 3/16, 3/16, 1/16: (my, mx), (my, mx2), (my2, mx), (my2, mx2)), ok = the pixel lies inside the map (outside,
// the value is the conv's zero padding).  One function for the lock-step kernel (K_c, operands from HBM) and the warp-specialised one
// (K_c'', operands from LDS): the arithmetic is the same instruction for instruction.
template <typename T>
__device__ __forceinline__ typename Vec<T>::type upadd_blend(typename Vec<T>::type v, typename Vec<T>::type t0, typename Vec<T>::type t1,
                                                            typename Vec<T>::type t2, typename Vec<T>::type t3, bool ok, bool int_blend,
                                                            float a_lat, float a_up) {
    typedef typename Vec<T>::type V;
    constexpr int VEC = Vec<T>::N;
    if constexpr (sizeof(T) == 2) {
        // fp16 engine: the blend in packed fp16 (v_pk_fma_f16, two channels per instruction).  The tap weights 9/16,
        // 3/16, 3/16, 1/16 are exact in fp16; the sum is rounded after every step instead of once (the `plus` tensors
        // carry 0.2 % of the fp16 engine's box-error variance, tools/fp16_error_budget.py) -- and the blend, which was
        // ~40 % of this kernel's VALU instructions in fp32 (8 x (mul + 3 fma_mix + fma + select) + 4 converts per
        // 16-byte item), is 4 x (mul + 3 fma + add + select)
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 w0 = {(half_t)0.5625f, (half_t)0.5625f}, w1 = {(half_t)0.1875f, (half_t)0.1875f}, w3 = {(half_t)0.0625f, (half_t)0.0625f};
        uint4 r;
        uint32_t *rp = (uint32_t *)&r;
        const uint4 lat4 = __builtin_bit_cast(uint4, v), u0 = __builtin_bit_cast(uint4, t0), u1 = __builtin_bit_cast(uint4, t1),
                    u2 = __builtin_bit_cast(uint4, t2), u3 = __builtin_bit_cast(uint4, t3);
        const uint32_t *lp = (const uint32_t *)&lat4, *p0 = (const uint32_t *)&u0, *p1 = (const uint32_t *)&u1, *p2 = (const uint32_t *)&u2, *p3 = (const uint32_t *)&u3;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            h2 acc = __builtin_bit_cast(h2, p0[d]) * w0;
            acc = __builtin_elementwise_fma(__builtin_bit_cast(h2, p1[d]), w1, acc);
            acc = __builtin_elementwise_fma(__builtin_bit_cast(h2, p2[d]), w1, acc);
            acc = __builtin_elementwise_fma(__builtin_bit_cast(h2, p3[d]), w3, acc);
            acc = acc + __builtin_bit_cast(h2, lp[d]);
            rp[d] = ok ? __builtin_bit_cast(uint32_t, acc) : 0u;
        }
        return __builtin_bit_cast(V, r);
    } else if (sizeof(T) == 1 && int_blend) {
        // int8 engine with per-channel scales (the three tensors of the add share one scale per channel, weights.h): the
        // blend in packed 16-bit integer arithmetic on the byte lanes -- q = min(rne(lat + (9a + 3b + 3c + d) / 16), 127),
        // every operand a ReLU output in 0..127, so the sum fits 11 bits.  Bit-identical to the fp32 path below (all its
        // intermediate values are exact, its rounding is rintf's round-half-even) at ~8 instead of ~20 instructions per channel.
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const uint4 lat4 = __builtin_bit_cast(uint4, v), u0 = __builtin_bit_cast(uint4, t0), u1 = __builtin_bit_cast(uint4, t1),
                    u2 = __builtin_bit_cast(uint4, t2), u3 = __builtin_bit_cast(uint4, t3);
        const uint32_t *lp = (const uint32_t *)&lat4, *p0 = (const uint32_t *)&u0, *p1 = (const uint32_t *)&u1, *p2 = (const uint32_t *)&u2, *p3 = (const uint32_t *)&u3;
        uint4 r;
        uint32_t *rp = (uint32_t *)&r;
        const u16x2 c9 = {9, 9}, c3 = {3, 3}, c7 = {7, 7}, c1 = {1, 1}, c127 = {127, 127};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t res[2];
#pragma unroll
            for (int par = 0; par < 2; par++) {                       // even / odd bytes of the dword as two 16-bit lanes
                const uint32_t sel = par ? 0x0c030c01u : 0x0c020c00u;
                auto lanes = [&](uint32_t x) -> u16x2 { return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, x, sel)); };
                u16x2 sum = lanes(p2[d]) * c3 + lanes(p3[d]);
                sum = lanes(p1[d]) * c3 + sum;
                sum = lanes(p0[d]) * c9 + sum;
                const u16x2 lat2 = lanes(lp[d]);
                const u16x2 odd = ((sum >> 4) + lat2) & c1;            // round half to even -- of lat + sum / 16, so the parity is the total's
                sum = (sum + c7 + odd) >> 4;
                sum = __builtin_elementwise_min((u16x2)(sum + lat2), c127);
                res[par] = __builtin_bit_cast(uint32_t, sum);
            }
            const uint32_t packed = __builtin_amdgcn_perm(res[1], res[0], 0x06020400u);
            rp[d] = ok ? packed : 0u;
        }
        return __builtin_bit_cast(V, r);
    } else {
        // the tap weights live in registers (not literals) so that each MAC is one v_fma_mix_f32 on the fp16 tap
        // instead of a convert + fmac pair: this staging blend is ~half of the kernel's VALU instructions
        float wq[4] = {0.5625f, 0.1875f, 0.1875f, 0.0625f};
#pragma unroll
        for (int q = 0; q < 4; q++) asm volatile("" : "+s"(wq[q]));
        const V up[4] = {t0, t1, t2, t3};
        float sacc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) sacc[e] = wq[0] * (float)up[0][e];
#pragma unroll
        for (int q = 1; q < 4; q++)
#pragma unroll
            for (int e = 0; e < VEC; e++) sacc[e] = fmaf(wq[q], (float)up[q][e], sacc[e]);
#pragma unroll
        for (int e = 0; e < VEC; e++) v[e] = ok ? to_T<T>(fmaf((float)v[e], a_lat, sacc[e] * a_up)) : to_T<T>(0.f);
        return v;
    }
}

// =============================================================================================
// K_c  dense 3x3 p1 s1 convolution + BN + ReLU as implicit GEMM (K = 9*CIN) on MFMA
//   rf_c2_aggr / rf_c1_aggr (UPADD: input = lateral + bilinear x2 upsample of the coarser level, i.e.
//   Deconvolution k4 s2 p1 g64 + Crop + Eltwise SUM, prototxt :1553-1592 / :1948-1987, closed form
//   SURVEY.md App. B.6) and the merged SSH convs 64->48, 16->32, 16->16 (prototxt :1239-1432 etc.).
//   The (TH+2)x(TW+2) halo tile is staged once in LDS; the B fragment of tap (ky,kx) is just the same
//   tile read at a shifted pixel offset, so no im2col buffer exists anywhere.
//   One launch can cover up to 3 FPN levels (same conv shape, different maps / weights): the SSH module of
//   strides 32, 16 and 8 is 3 launches, not 9.
// =============================================================================================
template <typename T, int CIN, int COUT, int TH, int TW, bool ALLC = false, bool PADROW = false> struct Conv3Cfg {
    typedef Mma<T> M;
    static constexpr int VEC = Vec<T>::N;
    static constexpr int P = TH * TW;
    static constexpr int HR = TH + 2, HC = TW + 2;
    static constexpr int LDI = lds_row<T>(CIN);
    static constexpr int LDO = lds_row<T>(COUT) > COUT ? lds_row<T>(COUT) : COUT + VEC;
    // Halo-row pitch.  A B-fragment ds_read_b128 covers 16 pixels; with TW = 8 those are 8 pixels of one halo row and 8 of the
    // next.  The read is conflict-free when the 16 pixel offsets are those of 16 CONSECUTIVE pixels modulo the 256-byte bank row
    // (pixel pitch = 32 mod 64 bytes: lds_row): pad the row pitch to a multiple of 256 bytes so that "next row" == "+8 pixels".
    static constexpr int ROWP = PADROW && TW == 8 ? (HC * LDI * (int)sizeof(T) + 255) / 256 * 256 / (int)sizeof(T) : HC * LDI;
    static constexpr size_t IN_BYTES = sizeof(T) * (size_t)(HR * ROWP);
    static constexpr size_t O_BYTES = sizeof(T) * (size_t)(P * LDO);
    // persistent + software pipelined like K_b: tile t+G is staged while tile t's result is still being stored, so the
    // halo tile and the result tile are separate LDS regions (and the GEMM -> epilogue barrier disappears).  DB: BOTH regions
    // double buffered, which leaves ONE barrier per tile (see the tile loop).  Built and measured in round 3, correct (the whole
    // -m gpu suite) and OFF: c1 87.4 -> 86.7 us, c2 29.7 -> 29.3, the SSH 64 -> 48 conv 103.6 -> 105.8 (fp16, A/B inside one call,
    // tools/gpu/rounds_3_4.sh r3_call13; int8 unchanged) for twice the LDS -- these tile loops do not wait at their barriers, they wait for
    // the dependent LDS -> MFMA -> LDS chain inside each wave.
    static constexpr bool DB = false;
    static constexpr size_t LDS_BYTES = (DB ? 2 : 1) * (IN_BYTES + O_BYTES);
    static_assert(IN_BYTES % 16 == 0, "LDS carve must stay 16-byte aligned");
    static constexpr int NT = COUT / 16, PT = P / 16;
    static constexpr int KTOT = 9 * CIN;
    static constexpr int KCH = (KTOT + M::K - 1) / M::K;
    // Wave split.  ALLC ("all channels"): every wave owns ALL output-channel tiles of its own pixel tiles, the whole weight
    // matrix stationary in its registers (64 -> 64: 4 x 18 A fragments = 288 VGPRs, one workgroup per CU).  A B fragment is then
    // read from LDS ONCE per workgroup and feeds NT MFMAs; with the channel tiles split over the waves instead (the round-1
    // layout, kept for the fp32 parity engine) every wave re-read every pixel's fragments: one 1 KB ds_read_b128 per MFMA, which
    // made these kernels LDS-bound at 9-25 % of the MFMA rate (SQ counters, profiles/r01_pmc_sq_n128_448_fp16.json).
    // !ALLC: output-channel tiles first; 3 channel tiles (the merged 64->48 SSH conv) run on 3 of the 4 waves.
    static constexpr bool ODD = !ALLC && NT == 3;
    typedef WaveSplit<(ODD || ALLC) ? 4 : NT, PT> WS;
    static constexpr int WN = ALLC ? 1 : (ODD ? 3 : WS::WN), WP = ALLC ? 4 : (ODD ? 1 : WS::WP);
    static constexpr int NI = ALLC ? NT : (ODD ? 1 : WS::NI), NJ = ALLC ? PT / 4 : (ODD ? PT : WS::NJ);
    static_assert(!ALLC || (PT % 4 == 0 && PT >= 4), "ALLC needs the pixel tiles to split over 4 waves");
    static constexpr int WREGS = NI * KCH * 4;                              // VGPRs of stationary weights per lane
    static constexpr bool STAT = sizeof(T) <= 2 && (ALLC ? WREGS <= 300 : NI * KCH <= 18);
    static constexpr int STAGE_ITEMS = HR * HC * (CIN / VEC);
    static constexpr int NPF = (STAGE_ITEMS + kThreads - 1) / kThreads;
    // register budget -> workgroups per CU (512 / 256 / 168 / 128 / 96 VGPRs per lane at 1 / 2 / 3 / 4 / 5)
    static constexpr int OCC = sizeof(T) > 2 ? 1 : ALLC ? (WREGS > 160 ? 1 : WREGS > 64 ? 2 : 5) : (CIN >= 64 ? 3 : 5);
    static constexpr int GFRAGS = 12;                                      // streamed case only
};

template <typename T>
struct Conv3Level {
    const T *in; const T *up; const T *w; const float *b; const float *m; T *out0; T *out1;
    int in_ld, in_off, ld0, off0, n0, ld1, off1, h, w_, tiles_x, tiles_y;
    int ntiles;             // tiles of this level in the launch
    int gb_begin, gsz;      // the level's share of the grid: workgroups [gb_begin, gb_begin + gsz) walk its tiles
    float a_lat, a_up;      // UPADD: staged value = to_T(lat * a_lat + upsample * a_up); 1, 1 unless int8 (scale ratios)
    int int_blend;          // int8, a_lat == a_up == 1: blend in packed 16-bit integers (bit-identical to the fp32 form)
};
template <typename T>
struct Conv3Args {
    Conv3Level<T> lv[3];
};

template <typename T, int CIN, int COUT, int TH, int TW, bool UPADD, bool ALLC, bool PADROW>
__global__ __launch_bounds__(kThreads, (Conv3Cfg<T, CIN, COUT, TH, TW, ALLC, PADROW>::OCC)) void conv3x3_kernel(Conv3Args<T> a) {
    typedef Conv3Cfg<T, CIN, COUT, TH, TW, ALLC, PADROW> C;
    typedef typename Vec<T>::type V;
    typedef Mma<T> M;
    typedef typename M::Frag Frag;
    constexpr int VEC = C::VEC, P = C::P, HC = C::HC, LDI = C::LDI, LDO = C::LDO, ROWP = C::ROWP;
    constexpr int CPV = CIN / VEC, KTOT = C::KTOT, KCH = C::KCH, NPF = C::NPF;
    constexpr int WN = C::WN, WP = C::WP, NI = C::NI, NJ = C::NJ;
    constexpr bool STAT = C::STAT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool DB = C::DB;
    constexpr int IN_ELEMS = (int)(C::IN_BYTES / sizeof(T)), O_ELEMS = (int)(C::O_BYTES / sizeof(T));
    T *s_in = (T *)smem;                                               // [2][IN_ELEMS] when double buffered
    T *s_out = (T *)(smem + (DB ? 2 : 1) * C::IN_BYTES);               // [2][O_ELEMS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int gbid = xcd_remap(blockIdx.x, gridDim.x);
    const int lvl = (gbid >= a.lv[1].gb_begin ? 1 : 0) + (gbid >= a.lv[2].gb_begin ? 1 : 0);
    const Conv3Level<T> &L = a.lv[lvl];
    const int first = gbid - L.gb_begin, G = L.gsz, ntiles = L.ntiles;
    const int tiles_x = L.tiles_x;
    const int lh = L.h, lw = L.w_;
    RF_TRACE_KEY(a.lv[0].ntiles + COUT);           // tile count of the first level + output channels: identifies one conv3x3 launch
    RF_TRACE(3, 0);

    // ---- once per workgroup: this wave's weight share and biases
    const int wn = C::ODD ? wave : wave % WN, wp = C::ODD ? 0 : wave / WN;      // ALLC: wn = 0, wp = wave
    const bool gemm_wave = !C::ODD || wave < 3;
    const int wnc = gemm_wave ? wn : 0;            // the idle wave reads tile 0's constants and never uses them
    Frag wst[STAT ? NI : 1][STAT ? KCH : 1];
    if constexpr (STAT) {
        const Frag *wsrc = (const Frag *)L.w + (size_t)wnc * KCH * 64 + lane;
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int kc = 0; kc < KCH; kc++) wst[i][kc] = wsrc[((i * WN) * KCH + kc) * 64];
    }
    f32x4 bias[NI], mult[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        bias[i] = *(const f32x4 *)(L.b + acc_cout(wnc + i * WN, lane, 0));
        mult[i] = load_mult(L.m, acc_cout(wnc + i * WN, lane, 0));
    }
    const float a_lat = L.a_lat, a_up = L.a_up;
    const bool int_blend = L.int_blend != 0;
    const T *in = L.in;
    const int in_ld = L.in_ld, in_off = L.in_off;

    // ---- halo tile of a tile -> registers: unconditional buffer loads, padding by range check + poisoned offsets (K_b)
    V pre[NPF];
    V upv[UPADD ? NPF : 1][4];
    int koff[NPF], kdyx[NPF];
#pragma unroll
    for (int k = 0; k < NPF; k++) {
        int i = tid + k * kThreads;
        i = i < C::STAGE_ITEMS ? i : C::STAGE_ITEMS - 1;
        const int pix = i / CPV, cv = i % CPV;
        const int dy = pix / HC, dx = pix % HC;
        koff[k] = ((dy * lw + dx) * in_ld + cv * VEC) * (int)sizeof(T);
        kdyx[k] = dy << 16 | dx;
    }
    const unsigned in_img_bytes = (unsigned)(lh * lw * in_ld - in_off) * (unsigned)sizeof(T);
    const int hh = lh >> 1, wh = lw >> 1;
    const unsigned up_img_bytes = (unsigned)(hh * wh * CIN) * (unsigned)sizeof(T);
    unsigned pre_ok = 0;          // UPADD only: bit k = staged pixel k lies inside the map (outside, the blend must give 0)
    auto fetch = [&](int tx, int ty, int img) {
        const auto rs = image_rsrc(in + (size_t)img * lh * lw * in_ld + in_off, in_img_bytes);
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        const int sbase = (iy0 * lw + ix0) * in_ld * (int)sizeof(T);
        pre_ok = 0;
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int ix = ix0 + (kdyx[k] & 0xffff);
            const unsigned off = (unsigned)ix < (unsigned)lw ? (unsigned)(koff[k] + sbase) : kOobOffset;
            pre[k] = buf_load16<V>(rs, off);
            if constexpr (UPADD) {
                // out[2m] = .75 in[m] + .25 in[m-1];  out[2m+1] = .75 in[m] + .25 in[m+1];  taps outside the coarse map = 0
                const int iy = iy0 + (kdyx[k] >> 16);
                pre_ok |= ((unsigned)iy < (unsigned)lh && (unsigned)ix < (unsigned)lw ? 1u : 0u) << k;
                const auto ru = image_rsrc(L.up + (size_t)img * hh * wh * CIN, up_img_bytes);
                const int cvo = (int)((tid + k * kThreads) % CPV) * VEC * (int)sizeof(T);
                const int my = iy >> 1, mx = ix >> 1;
                const int my2 = (iy & 1) ? my + 1 : my - 1, mx2 = (ix & 1) ? mx + 1 : mx - 1;
                const int ys[4] = {my, my, my2, my2}, xs[4] = {mx, mx2, mx, mx2};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // rows outside [0, hh) land outside the descriptor's range; columns need the explicit test
                    const unsigned uo = (unsigned)xs[q] < (unsigned)wh ? (unsigned)((ys[q] * wh + xs[q]) * CIN * (int)sizeof(T) + cvo) : kOobOffset;
                    upv[k][q] = buf_load16<V>(ru, uo);
                }
            }
        }
    };
    // ---- tile (img, oy0, ox0), finished in LDS, -> HBM (two destinations: the concat slice and the next conv's input)
    T *out0 = L.out0, *out1 = L.out1;
    const int n0 = L.n0, ld0 = L.ld0, off0 = L.off0, ld1 = L.ld1, off1 = L.off1;
    auto store_tile = [&](const T *s_res, int img, int oy0, int ox0) {
        constexpr int OPV = COUT / VEC;
        const auto r0 = image_rsrc(out0 + (size_t)img * lh * lw * ld0 + off0, (unsigned)(lh * lw * ld0 - off0) * (unsigned)sizeof(T));
        const auto r1 = image_rsrc(out1 + (size_t)img * lh * lw * ld1 + off1, (unsigned)(lh * lw * ld1 - off1) * (unsigned)sizeof(T));
        const int pbase = oy0 * lw + ox0;
        for (int i = tid; i < P * OPV; i += kThreads) {
            const int p = i / OPV, cv = i % OPV;
            const int py = p / TW, px = p % TW;
            const int c = cv * VEC;
            const int pix = pbase + py * lw + px;
            const bool okx = ox0 + px < lw;
            const V v = *(const V *)(s_res + p * LDO + c);
            if (c < n0) buf_store16(r0, okx ? (unsigned)((pix * ld0 + c) * (int)sizeof(T)) : kOobOffset, v);
            else buf_store16(r1, okx ? (unsigned)((pix * ld1 + (c - n0)) * (int)sizeof(T)) : kOobOffset, v);
        }
    };
    const TileStep step(G, tiles_x, L.tiles_y);
    TileCoord cur(first, tiles_x, L.tiles_y), nxt = cur;
    step.advance(nxt);
    if (first < ntiles) fetch(cur.tx, cur.ty, cur.img);
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): once-per-workgroup loads have landed before the tile loop

    int pbase[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int p = acc_pixel(wp + j * WP, lane);
        pbase[j] = (p / TW) * ROWP + (p % TW) * LDI;
    }

    // tile loop, fp32 engine:  stage(t) | fetch(t+G) issued | store(t-1) | barrier | GEMM(t) | epilogue(t) -> s_out | barrier
    // fp16 / int8 (double buffered):  stage(t) -> s_in[b] | fetch(t+G) | BARRIER | store(t-1) <- s_out[b^1] | GEMM(t) <- s_in[b] |
    //   epilogue(t) -> s_out[b] | b ^= 1.  One barrier per tile is enough: s_out[b^1] was written by epilogue(t-1) before the barrier
    //   and is next written by epilogue(t+1), after the next barrier; s_in[b] is next written by stage(t+2), two barriers later, and
    //   s_in[b^1] -- written by stage(t+1) right after this tile's GEMM -- was last read by GEMM(t-1), before this tile's barrier.
    int p_img = -1, p_oy0 = 0, p_ox0 = 0;
    int buf = 0;
    for (int t = first; t < ntiles; t += G) {
        const int tx = cur.tx, ty = cur.ty, img = cur.img;
        T *const s_in_b = s_in + (DB ? buf * IN_ELEMS : 0);
        T *const s_out_b = s_out + (DB ? buf * O_ELEMS : 0);
        const T *const s_out_p = s_out + (DB ? (buf ^ 1) * O_ELEMS : 0);
        RF_TRACE(3, 8);
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int i = tid + k * kThreads;
            if (i < C::STAGE_ITEMS) {
                V v = pre[k];
                if constexpr (UPADD) v = upadd_blend<T>(v, upv[k][0], upv[k][1], upv[k][2], upv[k][3], (pre_ok >> k) & 1u, int_blend, a_lat, a_up);
                *(V *)(s_in_b + ((i / CPV) / HC) * ROWP + ((i / CPV) % HC) * LDI + (i % CPV) * VEC) = v;
            }
        }
        RF_TRACE(3, 9);
        if (t + G < ntiles) fetch(nxt.tx, nxt.ty, nxt.img);
        cur = nxt;
        step.advance(nxt);
        RF_TRACE(3, 10);
        if constexpr (!DB) {
            if (p_img >= 0) store_tile(s_out_p, p_img, p_oy0, p_ox0);
        }
        RF_TRACE(3, 1);
        __syncthreads();
        RF_TRACE(3, 2);
        if constexpr (DB) {
            if (p_img >= 0) store_tile(s_out_p, p_img, p_oy0, p_ox0);
        }
        p_img = img; p_oy0 = ty * TH; p_ox0 = tx * TW;

        if (gemm_wave) {
            typename M::Acc acc[NI][NJ];
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) acc[i][j] = acc_init<T>(bias[i]);
            auto xf = [&](int j, int kc) -> Frag {
                const int kb = kc * M::K + (lane >> 4) * M::KPL;      // k = tap*CIN + c, KPL consecutive c of one tap
                const int tap = kb / CIN, c = kb % CIN;
                const int koff = (tap / 3) * ROWP + (tap % 3) * LDI + c;
                return kb < KTOT ? *(const Frag *)(s_in_b + pbase[j] + koff) : M::zero();
            };
            if constexpr (STAT) {
                gemm_stationary<T, NI, NJ, KCH, (NJ >= 4 ? 2 : 3), (KCH >= 9)>(acc, wst, xf);
            } else {
                GemmPipe<T, NI, NJ, KCH, WN, C::GFRAGS> pipe;
                pipe.init(L.w, wn, lane);
                pipe.run(acc, xf);
            }
            RF_TRACE(3, 3);
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    store_acc<T, LDO>(s_out_b, mult[i], bias[i], acc[i][j], wn + i * WN, wp + j * WP, lane, true);
        }
        RF_TRACE(3, 5);
        if constexpr (!DB) __syncthreads();
        else buf ^= 1;
        RF_TRACE(3, 6);
    }
    if constexpr (DB) __syncthreads();                                 // the last tile's epilogue has no closing barrier
    if (p_img >= 0) store_tile(s_out + (DB ? (buf ^ 1) * O_ELEMS : 0), p_img, p_oy0, p_ox0);
    RF_TRACE(3, 7);
}

// =============================================================================================
// K_c'  the merged SSH conv (3x3, 64 -> 48, all three FPN levels) WAVE-SPECIALISED (round 4).
//   48 output channels = three MFMA channel tiles: three waves of the workgroup own one each (18 stationary A fragments) and the
//   fourth had nothing to multiply.  In K_c all four waves walked every tile in lock step -- stage the halo, request the next one,
//   store the previous result, barrier, GEMM (3 of 4 waves), epilogue, barrier -- and the phase timeline (tools/probes/phase_trace.py,
//   profiles/r04_phase_timelines.txt) showed the GEMM to be only half of a tile's 3.9 us: 1.2 us went to the three memory phases that
//   precede it and 0.4 us to the two barriers.  Here the roles are split:
//     wave 3  (producer)   per tile: stores the PREVIOUS tile's result (s_out[b^1] -> HBM) and brings in the NEXT halo tile by
//                          LDS-DMA (`buffer_load_dwordx4 ... lds`: HBM -> s_in[b^1], no VGPRs, no ds_write pass; zero padding by
//                          the descriptor's range check, out-of-range lanes land as zeros: tools/probes/lds_dma.cpp)
//     waves 0-2 (consumers) per tile: GEMM on s_in[b] + epilogue into s_out[b]; they execute no global memory instruction in the loop
//   and everybody meets at ONE barrier per tile.  The halo image in LDS is lane-linear per DMA instruction (1 KiB = 64 slots of
//   16 B), so the padded layout of K_c (pixel pitch 32 mod 64 bytes, row pitch a multiple of 256 bytes: conflict-free B fragments)
//   is kept by giving the pad chunks slots of their own that load nothing.
// =============================================================================================
template <typename T, int NBUF> struct Conv3WsCfg {
    typedef Conv3Cfg<T, 64, 48, 8, 8, false, true> B;                  // tile geometry, LDS pitches, wave split (ODD: 3 GEMM waves)
    static constexpr int VEC = B::VEC;
    static constexpr int CPP = B::LDI / VEC;                           // 16-byte chunks per halo pixel, padding included
    static constexpr int CPR = B::ROWP / VEC;                          // chunks per halo row, padding included
    static constexpr int DPP = 64 / VEC;                               // data chunks per pixel
    static constexpr int SLOTS = B::HR * CPR;
    static constexpr int PIECES = (SLOTS + 63) / 64;                   // DMA wave-instructions per halo tile
    static constexpr int NSTORE = B::P * (48 / VEC) / 64;              // store wave-instructions per result tile (producer wave)
    static constexpr size_t IN_BYTES = (size_t)PIECES * 1024;          // whole KiB: the last piece may overhang the image
    static constexpr size_t O_BYTES = B::O_BYTES;
    static constexpr size_t LDS_BYTES = NBUF * IN_BYTES + 2 * O_BYTES; // NBUF halo tiles (prefetch distance NBUF - 1), result tile double buffered
    static_assert(B::LDI % VEC == 0 && B::ROWP % VEC == 0 && B::ODD && B::STAT, "layout assumptions of the warp-specialised conv");
    static_assert(IN_BYTES >= B::IN_BYTES && O_BYTES % 16 == 0 && (NBUF == 2 || NBUF == 3), "LDS carve");
    static_assert(NSTORE + PIECES < 64, "counted vmcnt");
};

template <typename T, int DEPTH, int NBUF, int SPLIT = 0>
__global__ __launch_bounds__(kThreads, (SPLIT ? 2 : (NBUF == 2 || sizeof(T) == 1 ? 3 : 2))) void conv3x3_ws_kernel(Conv3Args<T> a) {
    typedef Conv3WsCfg<T, NBUF> W;
    typedef typename W::B C;
    typedef typename Vec<T>::type V;
    typedef Mma<T> M;
    typedef typename M::Frag Frag;
    constexpr int CIN = 64, COUT = 48, TH = 8, TW = 8;
    constexpr int VEC = C::VEC, P = C::P, HC = C::HC, LDI = C::LDI, LDO = C::LDO, ROWP = C::ROWP;
    constexpr int KTOT = C::KTOT, KCH = C::KCH, NJ = C::NJ;
    constexpr int DIST = NBUF - 1;                                     // the producer runs DIST tiles ahead of the consumers
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int IN_ELEMS = (int)(W::IN_BYTES / sizeof(T)), O_ELEMS = (int)(W::O_BYTES / sizeof(T));
    T *s_in = (T *)smem;                                               // [NBUF][IN_ELEMS]
    T *s_out = (T *)(smem + NBUF * W::IN_BYTES);                       // [2][O_ELEMS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // scalar: the role branch below is a scalar branch
    const int gbid = xcd_remap(blockIdx.x, gridDim.x);
    const int lvl = (gbid >= a.lv[1].gb_begin ? 1 : 0) + (gbid >= a.lv[2].gb_begin ? 1 : 0);
    const Conv3Level<T> &L = a.lv[lvl];
    const int first = gbid - L.gb_begin, G = L.gsz, ntiles = L.ntiles;
    const int lh = L.h, lw = L.w_;
    const int n_my = first < ntiles ? (ntiles - 1 - first) / G + 1 : 0;            // tiles this workgroup walks
    T *out0 = L.out0, *out1 = L.out1;
    const int ld0 = L.ld0, off0 = L.off0, ld1 = L.ld1, off1 = L.off1;      // (n0 == 32: checked by the launcher)
    // finished tile (img, oy0, ox0) in LDS -> HBM by threads t0, t0 + nthr, ...: channels [0, 32) -> the concat slice, [32, 48) -> context_conv1's
    // own tensor.  Two loops, one per destination: a loop that picked the descriptor per lane would be serialised by the compiler into a
    // "waterfall" over the distinct descriptors
    auto store_tile = [&](const T *s_res, int img, int oy0, int ox0, int t0, int nthr) {
        constexpr int N0 = 32, V0 = N0 / VEC, V1 = (COUT - N0) / VEC;
        const auto r0 = image_rsrc(out0 + (size_t)img * lh * lw * ld0 + off0, (unsigned)(lh * lw * ld0 - off0) * (unsigned)sizeof(T));
        const auto r1 = image_rsrc(out1 + (size_t)img * lh * lw * ld1 + off1, (unsigned)(lh * lw * ld1 - off1) * (unsigned)sizeof(T));
        const int pbase = oy0 * lw + ox0;
        for (int i = t0; i < P * V0; i += nthr) {
            const int p = i / V0, c = (i % V0) * VEC;
            const int py = p / TW, px = p % TW;
            const int pix = pbase + py * lw + px;
            buf_store16(r0, ox0 + px < lw ? (unsigned)((pix * ld0 + c) * (int)sizeof(T)) : kOobOffset, *(const V *)(s_res + p * LDO + c));
        }
        for (int i = t0; i < P * V1; i += nthr) {
            const int p = i / V1, c = (i % V1) * VEC;
            const int py = p / TW, px = p % TW;
            const int pix = pbase + py * lw + px;
            buf_store16(r1, ox0 + px < lw ? (unsigned)((pix * ld1 + c) * (int)sizeof(T)) : kOobOffset, *(const V *)(s_res + p * LDO + N0 + c));
        }
    };

    if (wave == 3) {
        // ================================================= producer
        // slot s of the halo image -> (halo row, halo column, chunk); pad slots and slots past the last row load nothing (they are
        // given an out-of-range offset: the DMA writes zeros there, inside this buffer's whole-KiB region)
        int kpack[W::PIECES];                                          // byte offset inside the image | halo column << 27 (31 = no pixel)
        const int in_ld = L.in_ld;
#pragma unroll
        for (int i = 0; i < W::PIECES; i++) {
            const int s = i * 64 + lane;
            const int row = s / W::CPR, rem = s % W::CPR;
            const int px = rem / W::CPP, ch = rem % W::CPP;
            const bool real = row < C::HR && px < HC && ch < W::DPP;
            kpack[i] = real ? ((((row * lw + px) * in_ld + ch * VEC) * (int)sizeof(T)) | (px << 27)) : (int)(31u << 27);
        }
        const unsigned in_img_bytes = (unsigned)(lh * lw * in_ld - L.in_off) * (unsigned)sizeof(T);
        const T *in = L.in + L.in_off;
        auto dma = [&](int tx, int ty, int img, T *dst) {
            const auto rs = image_rsrc(in + (size_t)img * lh * lw * in_ld, in_img_bytes);
            const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
            const int sbase = (iy0 * lw + ix0) * in_ld * (int)sizeof(T);
#pragma unroll
            for (int i = 0; i < W::PIECES; i++) {
                const int dx = (int)((unsigned)kpack[i] >> 27);
                const unsigned off = (dx != 31 && (unsigned)(ix0 + dx) < (unsigned)lw) ? (unsigned)((kpack[i] & 0x07ffffff) + sbase) : kOobOffset;
                lds_dma16(rs, (unsigned char *)dst + i * 1024, off);
            }
        };
        const TileStep step(G, L.tiles_x, L.tiles_y);
        TileCoord cur(first, L.tiles_x, L.tiles_y), pf = cur;         // cur: the tile the consumers multiply; pf: the next tile to fetch
        // prologue: the first DIST tiles are requested; tile 0 has landed when at most the younger request is outstanding
#pragma unroll
        for (int d = 0; d < DIST; d++)
            if (d < n_my) { dma(pf.tx, pf.ty, pf.img, s_in + d * IN_ELEMS); step.advance(pf); }
        if (DIST == 2 && n_my >= 2) wait_vmcnt<W::PIECES>();
        else wait_vmcnt<0>();
        lds_barrier();
        int p_img = 0, p_oy0 = 0, p_ox0 = 0;
        for (int k = 0; k < n_my; k++) {
            const bool st = k > 0, ld = k + DIST < n_my;
            if (st) store_tile(s_out + ((k - 1) & 1) * O_ELEMS, p_img, p_oy0, p_ox0, lane, 64);
            if (ld) { dma(pf.tx, pf.ty, pf.img, s_in + ((k + DIST) % NBUF) * IN_ELEMS); step.advance(pf); }
            p_img = cur.img; p_oy0 = cur.ty * TH; p_ox0 = cur.tx * TW;
            step.advance(cur);
            // tile k + 1 must have landed before the barrier.  DIST == 1: it is the request just made.  DIST == 2: it was requested one
            // interval ago; everything issued in THIS interval (NSTORE stores, PIECES loads) may stay in flight (the counter is in order)
            if (DIST == 1) wait_vmcnt<0>();
            else if (st && ld) wait_vmcnt<W::NSTORE + W::PIECES>();
            else if (ld) wait_vmcnt<W::PIECES>();
            else if (st) wait_vmcnt<W::NSTORE>();
            else wait_vmcnt<0>();
            lds_barrier();
        }
        if (n_my > 0) store_tile(s_out + ((n_my - 1) & 1) * O_ELEMS, p_img, p_oy0, p_ox0, lane, 64);
    } else if (SPLIT && wave < 2) {
        // ================================================= consumers 0 / 1 (SPLIT): output-channel tiles 0 and 1, pixel tiles 2 wave, 2 wave + 1.
        // With the plain split (wave = channel tile) every B fragment is read from LDS by all three GEMM waves: 216 KB per tile for
        // 72 MFMAs each, and LDS bandwidth is as loaded as the matrix pipe.  Here channels [0, 32) belong to two waves that split the
        // PIXELS (each reads half of the fragments, uses each twice) and channels [32, 48) to the third: 144 KB, the same 72 MFMAs each.
        Frag wst[2][KCH];
        {
            const Frag *wsrc = (const Frag *)L.w + lane;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int kc = 0; kc < KCH; kc++) wst[i][kc] = wsrc[(i * KCH + kc) * 64];
        }
        f32x4 bias[2], mult[2];
#pragma unroll
        for (int i = 0; i < 2; i++) { bias[i] = *(const f32x4 *)(L.b + acc_cout(i, lane, 0)); mult[i] = load_mult(L.m, acc_cout(i, lane, 0)); }
        int pbase[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int p = acc_pixel(2 * wave + j, lane);
            pbase[j] = (p / TW) * ROWP + (p % TW) * LDI;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        lds_barrier();
        for (int k = 0; k < n_my; k++) {
            const T *s_in_b = s_in + (k % NBUF) * IN_ELEMS;
            T *s_out_b = s_out + (k & 1) * O_ELEMS;
            typename M::Acc acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = acc_init<T>(bias[i]);
            gemm_stationary<T, 2, 2, KCH, (DEPTH > 2 ? DEPTH : 3), true>(acc, wst, [&](int j, int kc) -> Frag {
                const int kb = kc * M::K + (lane >> 4) * M::KPL;
                const int tap = kb / CIN, c = kb % CIN;
                const int koff = (tap / 3) * ROWP + (tap % 3) * LDI + c;
                return kb < KTOT ? *(const Frag *)(s_in_b + pbase[j] + koff) : M::zero();
            });
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) store_acc<T, LDO>(s_out_b, mult[i], bias[i], acc[i][j], i, 2 * wave + j, lane, true);
            lds_barrier();
        }
    } else {
        // ================================================= consumers: wave = output-channel tile (SPLIT: only wave 2 = tile 2), all pixel tiles
        const int wn = wave;
        Frag wst[1][KCH];
        {
            const Frag *wsrc = (const Frag *)L.w + (size_t)wn * KCH * 64 + lane;
#pragma unroll
            for (int kc = 0; kc < KCH; kc++) wst[0][kc] = wsrc[kc * 64];
        }
        const f32x4 bias = *(const f32x4 *)(L.b + acc_cout(wn, lane, 0));
        const f32x4 mult = load_mult(L.m, acc_cout(wn, lane, 0));
        int pbase[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int p = acc_pixel(j, lane);
            pbase[j] = (p / TW) * ROWP + (p % TW) * LDI;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                            // weights are in registers: no global access below this line
        lds_barrier();
        for (int k = 0; k < n_my; k++) {
            const T *s_in_b = s_in + (k % NBUF) * IN_ELEMS;
            T *s_out_b = s_out + (k & 1) * O_ELEMS;
            typename M::Acc acc[1][NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) acc[0][j] = acc_init<T>(bias);
            gemm_stationary<T, 1, NJ, KCH, DEPTH, true>(acc, wst, [&](int j, int kc) -> Frag {
                const int kb = kc * M::K + (lane >> 4) * M::KPL;       // k = tap*CIN + c, KPL consecutive c of one tap
                const int tap = kb / CIN, c = kb % CIN;
                const int koff = (tap / 3) * ROWP + (tap % 3) * LDI + c;
                return kb < KTOT ? *(const Frag *)(s_in_b + pbase[j] + koff) : M::zero();
            });
#pragma unroll
            for (int j = 0; j < NJ; j++) store_acc<T, LDO>(s_out_b, mult, bias, acc[0][j], wn, j, lane, true);
            lds_barrier();
        }
    }
}

// =============================================================================================
// K_c''  the FPN aggregation convs (rf_c2_aggr / rf_c1_aggr: 3x3, 64 -> 64, input = lateral + bilinear x2 upsample of the coarser level)
//   WAVE-SPECIALISED like K_c' (round 4).  In K_c the fused upsample + add made the staging phase the longest of the tile: every thread
//   fetched its lateral item AND four coarse-map taps into registers (20 VGPRs per item: why the tile was 4 x 8), blended and wrote LDS,
//   all in lock step with the GEMM.  Here the producer wave brings BOTH operands into LDS by LDS-DMA -- the 10 x 10 lateral halo in K_c's
//   padded layout and the 8 x 8 coarse-map patch the tile's 2x upsample reads -- DIST tiles ahead; the four GEMM waves blend LDS -> LDS in
//   place (same arithmetic: upadd_blend), barrier, multiply (8 x 8 tiles: 72 MFMAs per wave and tile, pinned B-fragment pipeline), write
//   the result over the coarse patch's space, barrier.  One LDS ring of NBUF x [halo | coarse patch / result tile].
// =============================================================================================
template <typename T, int NBUF> struct Conv3UpWsCfg {
    typedef Conv3Cfg<T, 64, 64, 8, 8, false, true> B;                  // 8 x 8 tile geometry and LDS pitches of the halo; wave = channel tile
    static constexpr int VEC = B::VEC, P = B::P;
    static constexpr int CPP = B::LDI / VEC, CPR = B::ROWP / VEC, DPP = 64 / VEC;
    static constexpr int SLOTS = B::HR * CPR, PIECES = (SLOTS + 63) / 64;
    static constexpr int CH = 8 / 2 + 4, CW = 8 / 2 + 4;               // coarse-map patch of an 8 x 8 tile: rows oy0/2 - 2 .. oy0/2 + 5
    static constexpr int CSLOTS = CH * CW * DPP, CPIECES = (CSLOTS + 63) / 64;
    static constexpr int LDO = 64 + VEC;                               // result tile pitch (elements)
    static constexpr int NSTORE = P * (64 / VEC) / 64;
    static constexpr size_t L_BYTES = (size_t)SLOTS * 16;              // exact: the last DMA piece is masked past the last slot
    static constexpr size_t C_BYTES = (size_t)CSLOTS * 16, O_BYTES = sizeof(T) * (size_t)(P * LDO);
    static constexpr size_t X_BYTES = C_BYTES > O_BYTES ? C_BYTES : O_BYTES;       // coarse patch, then (after the blend) the result tile
    static constexpr size_t BUF_BYTES = L_BYTES + X_BYTES;
    static constexpr size_t LDS_BYTES = NBUF * BUF_BYTES;
    static constexpr int THREADS = 320;
    static constexpr int WG_CAP = sizeof(T) == 2 ? 2 : 3;              // fp16: 18 stationary A fragments + the blend's operands need ~165 VGPRs
    static constexpr int WG_PER_CU = (int)(160 * 1024 / LDS_BYTES) > WG_CAP ? WG_CAP : (int)(160 * 1024 / LDS_BYTES);
    static constexpr int WAVES_PER_EU = (WG_PER_CU * 5 + 3) / 4;
    static constexpr int NBL = (B::HR * B::HC * DPP + 255) / 256;      // blend items per consumer thread
    static_assert(B::NI == 1 && B::NJ == 4 && B::WN == 4 && B::STAT && !B::ODD, "wave = output-channel tile, all four pixel tiles");
    static_assert(L_BYTES % 16 == 0 && X_BYTES % 16 == 0 && CSLOTS % 64 == 0 && PIECES + CPIECES + NSTORE < 63, "LDS carve / counted vmcnt");
};

#ifdef RF_PROBES      // producer-wave form: measured and rejected (c1 78 -> 110 us), probe build only
template <typename T, int DEPTH, int NBUF>
__global__ __launch_bounds__(320, (Conv3UpWsCfg<T, NBUF>::WAVES_PER_EU)) void conv3x3_up_ws_kernel(Conv3Args<T> a) {
    typedef Conv3UpWsCfg<T, NBUF> W;
    typedef typename W::B C;
    typedef typename Vec<T>::type V;
    typedef Mma<T> M;
    typedef typename M::Frag Frag;
    constexpr int CIN = 64, TH = 8, TW = 8;
    constexpr int VEC = C::VEC, P = C::P, HC = C::HC, HR = C::HR, LDI = C::LDI, ROWP = C::ROWP, LDO = W::LDO;
    constexpr int KTOT = C::KTOT, KCH = C::KCH, NJ = C::NJ, DPP = W::DPP, CW = W::CW;
    constexpr int DIST = NBUF - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Conv3Level<T> &L = a.lv[0];                                  // one level per launch (the aggregation convs are separate layers)
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G), ntiles = L.ntiles;
    const int lh = L.h, lw = L.w_, hh = lh >> 1, wh = lw >> 1;
    const int n_my = first < ntiles ? (ntiles - 1 - first) / G + 1 : 0;
    const TileStep step(G, L.tiles_x, L.tiles_y);
    auto halo_of = [&](int b) -> T * { return (T *)(smem + (size_t)b * W::BUF_BYTES); };
    auto aux_of = [&](int b) -> T * { return (T *)(smem + (size_t)b * W::BUF_BYTES + W::L_BYTES); };

    if (wave == 4) {
        // ================================================= producer
        int kl[W::PIECES], kc[W::CPIECES];                             // byte offset inside the image | column << 26 (63 = not a pixel)
        const int in_ld = L.in_ld;
#pragma unroll
        for (int i = 0; i < W::PIECES; i++) {
            const int s = i * 64 + lane;
            const int row = s / W::CPR, rem = s % W::CPR;
            const int px = rem / W::CPP, ch = rem % W::CPP;
            const bool real = row < HR && px < HC && ch < DPP;
            kl[i] = real ? ((((row * lw + px) * in_ld + ch * VEC) * (int)sizeof(T)) | (px << 26)) : (int)(63u << 26);
        }
#pragma unroll
        for (int i = 0; i < W::CPIECES; i++) {
            const int s = i * 64 + lane;
            const int cpix = s / DPP, ch = s % DPP;
            const int cy = cpix / CW, cx = cpix % CW;
            kc[i] = (((cy * wh + cx) * CIN + ch * VEC) * (int)sizeof(T)) | (cx << 26);
        }
        const unsigned lat_bytes = (unsigned)(lh * lw * in_ld - L.in_off) * (unsigned)sizeof(T);
        const unsigned up_bytes = (unsigned)(hh * wh * CIN) * (unsigned)sizeof(T);
        const T *in = L.in + L.in_off;
        auto dma = [&](int tx, int ty, int img, int b) {
            const auto rs = image_rsrc(in + (size_t)img * lh * lw * in_ld, lat_bytes);
            const auto ru = image_rsrc(L.up + (size_t)img * hh * wh * CIN, up_bytes);
            const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
            const int sbase = (iy0 * lw + ix0) * in_ld * (int)sizeof(T);
            unsigned char *hd = (unsigned char *)halo_of(b), *cd = (unsigned char *)aux_of(b);
#pragma unroll
            for (int i = 0; i < W::PIECES; i++) {
                const int dx = (int)((unsigned)kl[i] >> 26);
                const unsigned off = (dx != 63 && (unsigned)(ix0 + dx) < (unsigned)lw) ? (unsigned)((kl[i] & 0x03ffffff) + sbase) : kOobOffset;
                if (i * 64 + lane < W::SLOTS) lds_dma16(rs, hd + i * 1024, off);        // (only the last piece is partial)
            }
            const int cy0 = ty * (TH / 2) - 2, cx0 = tx * (TW / 2) - 2;
            const int cbase = (cy0 * wh + cx0) * CIN * (int)sizeof(T);
#pragma unroll
            for (int i = 0; i < W::CPIECES; i++) {
                const int cx = (int)((unsigned)kc[i] >> 26);
                const unsigned off = (unsigned)(cx0 + cx) < (unsigned)wh ? (unsigned)((kc[i] & 0x03ffffff) + cbase) : kOobOffset;
                lds_dma16(ru, cd + i * 1024, off);
            }
        };
        T *out0 = L.out0;
        const int ld0 = L.ld0, off0 = L.off0;
        auto store_tile = [&](int b, int img, int oy0, int ox0) {
            constexpr int OPV = 64 / VEC;
            const T *s_res = aux_of(b);
            const auto r0 = image_rsrc(out0 + (size_t)img * lh * lw * ld0 + off0, (unsigned)(lh * lw * ld0 - off0) * (unsigned)sizeof(T));
            const int pbase = oy0 * lw + ox0;
#pragma unroll
            for (int i = lane; i < P * OPV; i += 64) {
                const int p = i / OPV, c = (i % OPV) * VEC;
                const int py = p / TW, px = p % TW;
                buf_store16(r0, ox0 + px < lw ? (unsigned)(((pbase + py * lw + px) * ld0 + c) * (int)sizeof(T)) : kOobOffset, *(const V *)(s_res + p * LDO + c));
            }
        };
        TileCoord cur(first, L.tiles_x, L.tiles_y), pf = cur;
#pragma unroll
        for (int d = 0; d < DIST; d++)
            if (d < n_my) { dma(pf.tx, pf.ty, pf.img, d); step.advance(pf); }
        if (DIST == 2 && n_my >= 2) wait_vmcnt<W::PIECES + W::CPIECES>();
        else wait_vmcnt<0>();
        lds_barrier();
        int p_img = 0, p_oy0 = 0, p_ox0 = 0;
        for (int k = 0; k < n_my; k++) {
            const bool st = k > 0, ld = k + DIST < n_my;
            RF_TRACE_T(6, 5, 256);
            // the previous result sits in the aux space of buffer (k - 1) % NBUF == (k + DIST) % NBUF: it is read out (LDS reads complete
            // before the stores that carry the data are issued) BEFORE the coarse patch of tile k + DIST is requested into the same space
            lds_barrier();
            RF_TRACE_T(6, 6, 256);                                             // (blend -> GEMM): joined FIRST -- this wave's issue work (~300 instructions)
                                                                       // then overlaps the consumers' GEMM instead of holding up their short blend pass
            if (st) store_tile((k - 1) % NBUF, p_img, p_oy0, p_ox0);
            RF_TRACE_T(6, 7, 256);
            if (ld) { dma(pf.tx, pf.ty, pf.img, (k + DIST) % NBUF); step.advance(pf); }
            RF_TRACE_T(6, 8, 256);
            p_img = cur.img; p_oy0 = cur.ty * TH; p_ox0 = cur.tx * TW;
            step.advance(cur);
            if (DIST == 1) wait_vmcnt<0>();
            else if (st && ld) wait_vmcnt<W::NSTORE + W::PIECES + W::CPIECES>();
            else if (ld) wait_vmcnt<W::PIECES + W::CPIECES>();
            else if (st) wait_vmcnt<W::NSTORE>();
            else wait_vmcnt<0>();
            RF_TRACE_T(6, 9, 256);
            lds_barrier();
            RF_TRACE_T(6, 10, 256);
        }
        if (n_my > 0) store_tile((n_my - 1) % NBUF, p_img, p_oy0, p_ox0);
        return;
    }

    // ================================================= consumers: wave = output-channel tile
    const int wn = wave;
    Frag wst[1][KCH];
    {
        const Frag *wsrc = (const Frag *)L.w + (size_t)wn * KCH * 64 + lane;
#pragma unroll
        for (int kc2 = 0; kc2 < KCH; kc2++) wst[0][kc2] = wsrc[kc2 * 64];
    }
    const f32x4 bias = *(const f32x4 *)(L.b + acc_cout(wn, lane, 0));
    const f32x4 mult = load_mult(L.m, acc_cout(wn, lane, 0));
    const float a_lat = L.a_lat, a_up = L.a_up;
    const bool int_blend = L.int_blend != 0;
    int pbase[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int p = acc_pixel(j, lane);
        pbase[j] = (p / TW) * ROWP + (p % TW) * LDI;
    }
    // blend items of this thread: halo pixel (dy, dx), 16-byte chunk cv -> element offsets of the lateral item and of the coarse patch's
    // tap (my, mx); the other three taps are one row / one column further in the direction the pixel's parity picks (tile origins are
    // multiples of 8: the parities are thread constants)
    int b_lat[W::NBL], b_c0[W::NBL], b_step[W::NBL], b_yx[W::NBL];
#pragma unroll
    for (int m = 0; m < W::NBL; m++) {
        int i = tid + m * 256;
        const bool real = i < HR * HC * DPP;
        i = real ? i : 0;
        const int pix = i / DPP, cv = i % DPP;
        const int dy = pix / HC, dx = pix % HC;
        const int ly = 2 + ((dy - 1) >> 1), lx = 2 + ((dx - 1) >> 1);
        const int sy = ((dy - 1) & 1) ? 1 : -1, sx = ((dx - 1) & 1) ? 1 : -1;
        b_lat[m] = dy * ROWP + dx * LDI + cv * VEC;
        b_c0[m] = (ly * CW + lx) * CIN + cv * VEC;
        b_step[m] = (sy * CW * CIN) * 65536 + (sx * CIN + 32768);      // row step in the high half, column step (+32768) in the low half
        b_yx[m] = real ? (dy << 8 | dx) : -1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                // weights are in registers: no global access below this line
    lds_barrier();
    TileCoord cur(first, L.tiles_x, L.tiles_y);
    for (int k = 0; k < n_my; k++) {
        const int b = k % NBUF;
        T *s_in_b = halo_of(b);
        T *s_x = aux_of(b);
        const int iy0 = cur.ty * TH - 1, ix0 = cur.tx * TW - 1;
        step.advance(cur);
        RF_TRACE_T(6, 0, 0);
        // ---- blend: lateral + upsample(coarse) -> the halo tile, in place (K_c's staging arithmetic, operands from LDS)
#pragma unroll
        for (int m = 0; m < W::NBL; m++) {
            if (b_yx[m] < 0) continue;
            const int rs_ = b_step[m] >> 16, cs_ = (b_step[m] & 0xffff) - 32768;
            const V lat = *(const V *)(s_in_b + b_lat[m]);
            const V u0 = *(const V *)(s_x + b_c0[m]), u1 = *(const V *)(s_x + b_c0[m] + cs_);
            const V u2 = *(const V *)(s_x + b_c0[m] + rs_), u3 = *(const V *)(s_x + b_c0[m] + rs_ + cs_);
            const bool ok = (unsigned)(iy0 + (b_yx[m] >> 8)) < (unsigned)lh && (unsigned)(ix0 + (b_yx[m] & 0xff)) < (unsigned)lw;
            *(V *)(s_in_b + b_lat[m]) = upadd_blend<T>(lat, u0, u1, u2, u3, ok, int_blend, a_lat, a_up);
        }
        RF_TRACE_T(6, 1, 0);
        lds_barrier();
        RF_TRACE_T(6, 2, 0);
        // ---- GEMM on the blended halo tile, result over the (dead) coarse patch
        typename M::Acc acc[1][NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[0][j] = acc_init<T>(bias);
        gemm_stationary<T, 1, NJ, KCH, DEPTH, true>(acc, wst, [&](int j, int kc2) -> Frag {
            const int kb = kc2 * M::K + (lane >> 4) * M::KPL;
            const int tap = kb / CIN, c = kb % CIN;
            const int koff = (tap / 3) * ROWP + (tap % 3) * LDI + c;
            return kb < KTOT ? *(const Frag *)(s_in_b + pbase[j] + koff) : M::zero();
        });
#pragma unroll
        for (int j = 0; j < NJ; j++) store_acc<T, LDO>(s_x, mult, bias, acc[0][j], wn, j, lane, true);
        RF_TRACE_T(6, 3, 0);
        lds_barrier();
        RF_TRACE_T(6, 4, 0);
    }
}
#endif  // RF_PROBES

// =============================================================================================
// K_c3  the aggregation convs with LDS-DMA staging and NO dedicated producer wave (round 4, after the phase stamps of K_c'': a five-wave
//   workgroup at ~165 VGPRs is admitted only once per CU -- the 3-waves-per-SIMD register limit needs both workgroups' fifth waves on
//   different SIMDs -- and its single producer wave spent 0.7 us per tile reading the result tile back from LDS store by store).
//   Same LDS ring, same blend, same GEMM as K_c''; every one of the four waves (= output-channel tiles) issues a quarter of the
//   tile's memory traffic itself: its share of the previous result's stores at the top of the interval, its share of the DMA for
//   the tile DIST steps ahead right after the blend barrier -- i.e. under its own GEMM -- and waits (counted vmcnt) only for the share
//   of the NEXT tile's halo it requested one interval earlier.
// =============================================================================================
template <typename T, int DEPTH, int NBUF>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_up_dma_kernel(Conv3Args<T> a) {
    typedef Conv3UpWsCfg<T, NBUF> W;
    typedef typename W::B C;
    typedef typename Vec<T>::type V;
    typedef Mma<T> M;
    typedef typename M::Frag Frag;
    constexpr int CIN = 64, TH = 8, TW = 8;
    constexpr int VEC = C::VEC, P = C::P, HC = C::HC, HR = C::HR, LDI = C::LDI, ROWP = C::ROWP, LDO = W::LDO;
    constexpr int KTOT = C::KTOT, KCH = C::KCH, NJ = C::NJ, DPP = W::DPP, CW = W::CW;
    constexpr int DIST = NBUF - 1;
    constexpr int LPW = (W::PIECES + 3) / 4, CPW = (W::CPIECES + 3) / 4;          // DMA pieces per wave: halo, coarse patch
    constexpr int LMIN = W::PIECES / 4, CMIN = W::CPIECES / 4;                     // ... the fewest any wave issues
    constexpr int SPW = P * (64 / VEC) / kThreads;                                 // store instructions per wave and tile
    static_assert(P * (64 / VEC) % kThreads == 0 && SPW + LPW + CPW < 63, "whole store instructions per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Conv3Level<T> &L = a.lv[0];
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G), ntiles = L.ntiles;
    const int lh = L.h, lw = L.w_, hh = lh >> 1, wh = lw >> 1;
    const int n_my = first < ntiles ? (ntiles - 1 - first) / G + 1 : 0;
    const TileStep step(G, L.tiles_x, L.tiles_y);
    auto halo_of = [&](int b) -> T * { return (T *)(smem + (size_t)b * W::BUF_BYTES); };
    auto aux_of = [&](int b) -> T * { return (T *)(smem + (size_t)b * W::BUF_BYTES + W::L_BYTES); };

    // ---- this wave's share of the DMA: halo pieces wave, wave + 4, ...; coarse-patch pieces likewise
    int kl[LPW], kc[CPW];                                              // byte offset inside the image | column << 26 (63 = not a pixel / no piece)
    const int in_ld = L.in_ld;
#pragma unroll
    for (int j = 0; j < LPW; j++) {
        const int s = (wave + 4 * j) * 64 + lane;
        const int row = s / W::CPR, rem = s % W::CPR;
        const int px = rem / W::CPP, ch = rem % W::CPP;
        const bool real = row < HR && px < HC && ch < DPP;
        kl[j] = real ? ((((row * lw + px) * in_ld + ch * VEC) * (int)sizeof(T)) | (px << 26)) : (int)(63u << 26);
    }
#pragma unroll
    for (int j = 0; j < CPW; j++) {
        const int s = (wave + 4 * j) * 64 + lane;
        const int cpix = s / DPP, ch = s % DPP;
        const int cy = cpix / CW, cx = cpix % CW;
        kc[j] = (((cy * wh + cx) * CIN + ch * VEC) * (int)sizeof(T)) | (cx << 26);
    }
    const unsigned lat_bytes = (unsigned)(lh * lw * in_ld - L.in_off) * (unsigned)sizeof(T);
    const unsigned up_bytes = (unsigned)(hh * wh * CIN) * (unsigned)sizeof(T);
    const T *in = L.in + L.in_off;
    auto dma = [&](int tx, int ty, int img, int b) {
        const auto rs = image_rsrc(in + (size_t)img * lh * lw * in_ld, lat_bytes);
        const auto ru = image_rsrc(L.up + (size_t)img * hh * wh * CIN, up_bytes);
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        const int sbase = (iy0 * lw + ix0) * in_ld * (int)sizeof(T);
        unsigned char *hd = (unsigned char *)halo_of(b), *cd = (unsigned char *)aux_of(b);
#pragma unroll
        for (int j = 0; j < LPW; j++) {
            const int piece = wave + 4 * j;                            // (scalar)
            const int dx = (int)((unsigned)kl[j] >> 26);
            const unsigned off = (dx != 63 && (unsigned)(ix0 + dx) < (unsigned)lw) ? (unsigned)((kl[j] & 0x03ffffff) + sbase) : kOobOffset;
            if (piece < W::PIECES && piece * 64 + lane < W::SLOTS) lds_dma16(rs, hd + piece * 1024, off);
        }
        const int cy0 = ty * (TH / 2) - 2, cx0 = tx * (TW / 2) - 2;
        const int cbase = (cy0 * wh + cx0) * CIN * (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < CPW; j++) {
            const int piece = wave + 4 * j;
            const int cx = (int)((unsigned)kc[j] >> 26);
            const unsigned off = (unsigned)(cx0 + cx) < (unsigned)wh ? (unsigned)((kc[j] & 0x03ffffff) + cbase) : kOobOffset;
            if (piece < W::CPIECES) lds_dma16(ru, cd + piece * 1024, off);
        }
    };
    T *out0 = L.out0;
    const int ld0 = L.ld0, off0 = L.off0;
    auto store_tile = [&](int b, int img, int oy0, int ox0) {          // all 256 threads: SPW items each
        constexpr int OPV = 64 / VEC;
        const T *s_res = aux_of(b);
        const auto r0 = image_rsrc(out0 + (size_t)img * lh * lw * ld0 + off0, (unsigned)(lh * lw * ld0 - off0) * (unsigned)sizeof(T));
        const int pbase = oy0 * lw + ox0;
        V v[SPW];
#pragma unroll
        for (int m = 0; m < SPW; m++) { const int i = tid + m * kThreads; v[m] = *(const V *)(s_res + (i / OPV) * LDO + (i % OPV) * VEC); }
#pragma unroll
        for (int m = 0; m < SPW; m++) {
            const int i = tid + m * kThreads;
            const int p = i / OPV, c = (i % OPV) * VEC;
            const int py = p / TW, px = p % TW;
            buf_store16(r0, ox0 + px < lw ? (unsigned)(((pbase + py * lw + px) * ld0 + c) * (int)sizeof(T)) : kOobOffset, v[m]);
        }
    };

    // ---- weights (wave = output-channel tile), per-lane constants of the GEMM and the blend
    const int wn = wave;
    Frag wst[1][KCH];
    {
        const Frag *wsrc = (const Frag *)L.w + (size_t)wn * KCH * 64 + lane;
#pragma unroll
        for (int kc2 = 0; kc2 < KCH; kc2++) wst[0][kc2] = wsrc[kc2 * 64];
    }
    const f32x4 bias = *(const f32x4 *)(L.b + acc_cout(wn, lane, 0));
    const f32x4 mult = load_mult(L.m, acc_cout(wn, lane, 0));
    const float a_lat = L.a_lat, a_up = L.a_up;
    const bool int_blend = L.int_blend != 0;
    int pbase[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int p = acc_pixel(j, lane);
        pbase[j] = (p / TW) * ROWP + (p % TW) * LDI;
    }
    int b_lat[W::NBL], b_c0[W::NBL], b_step[W::NBL], b_yx[W::NBL];
#pragma unroll
    for (int m = 0; m < W::NBL; m++) {
        int i = tid + m * 256;
        const bool real = i < HR * HC * DPP;
        i = real ? i : 0;
        const int pix = i / DPP, cv = i % DPP;
        const int dy = pix / HC, dx = pix % HC;
        const int ly = 2 + ((dy - 1) >> 1), lx = 2 + ((dx - 1) >> 1);
        const int sy = ((dy - 1) & 1) ? 1 : -1, sx = ((dx - 1) & 1) ? 1 : -1;
        b_lat[m] = dy * ROWP + dx * LDI + cv * VEC;
        b_c0[m] = (ly * CW + lx) * CIN + cv * VEC;
        b_step[m] = (sy * CW * CIN) * 65536 + (sx * CIN + 32768);
        b_yx[m] = real ? (dy << 8 | dx) : -1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                // weights are in registers: the counted waits below see only this wave's DMA and stores

    TileCoord cur(first, L.tiles_x, L.tiles_y), pf = cur;
#pragma unroll
    for (int d = 0; d < DIST; d++)
        if (d < n_my) { dma(pf.tx, pf.ty, pf.img, d); step.advance(pf); }
    if (DIST == 2 && n_my >= 2) wait_vmcnt<LMIN + CMIN>();            // tile 0's share has landed (at most the younger request is outstanding)
    else wait_vmcnt<0>();
    lds_barrier();
    int p_img = 0, p_oy0 = 0, p_ox0 = 0;
    for (int k = 0; k < n_my; k++) {
        const int b = k % NBUF;
        const bool st = k > 0, ld = k + DIST < n_my;
        T *s_in_b = halo_of(b);
        T *s_x = aux_of(b);
        const int iy0 = cur.ty * TH - 1, ix0 = cur.tx * TW - 1;
        // ---- the previous result leaves (its space is the coarse patch of tile k + DIST, requested after the barrier below)
        if (st) store_tile((k - 1) % NBUF, p_img, p_oy0, p_ox0);
        p_img = cur.img; p_oy0 = cur.ty * TH; p_ox0 = cur.tx * TW;
        step.advance(cur);
        // ---- blend: lateral + upsample(coarse) -> the halo tile, in place
#pragma unroll
        for (int m = 0; m < W::NBL; m++) {
            if (b_yx[m] < 0) continue;
            const int rs_ = b_step[m] >> 16, cs_ = (b_step[m] & 0xffff) - 32768;
            const V lat = *(const V *)(s_in_b + b_lat[m]);
            const V u0 = *(const V *)(s_x + b_c0[m]), u1 = *(const V *)(s_x + b_c0[m] + cs_);
            const V u2 = *(const V *)(s_x + b_c0[m] + rs_), u3 = *(const V *)(s_x + b_c0[m] + rs_ + cs_);
            const bool ok = (unsigned)(iy0 + (b_yx[m] >> 8)) < (unsigned)lh && (unsigned)(ix0 + (b_yx[m] & 0xff)) < (unsigned)lw;
            *(V *)(s_in_b + b_lat[m]) = upadd_blend<T>(lat, u0, u1, u2, u3, ok, int_blend, a_lat, a_up);
        }
        lds_barrier();
        // ---- this wave's share of the halo + coarse patch of tile k + DIST: in flight under the GEMM
        if (ld) { dma(pf.tx, pf.ty, pf.img, (k + DIST) % NBUF); step.advance(pf); }
        typename M::Acc acc[1][NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[0][j] = acc_init<T>(bias);
        gemm_stationary<T, 1, NJ, KCH, DEPTH, true>(acc, wst, [&](int j, int kc2) -> Frag {
            const int kb = kc2 * M::K + (lane >> 4) * M::KPL;
            const int tap = kb / CIN, c = kb % CIN;
            const int koff = (tap / 3) * ROWP + (tap % 3) * LDI + c;
            return kb < KTOT ? *(const Frag *)(s_in_b + pbase[j] + koff) : M::zero();
        });
#pragma unroll
        for (int j = 0; j < NJ; j++) store_acc<T, LDO>(s_x, mult, bias, acc[0][j], wn, j, lane, true);
        // tile k + 1's share must have landed before the barrier.  DIST == 1: it is the request just made.  DIST == 2: it was made one
        // interval ago; this interval's stores (SPW) and requests (>= LMIN + CMIN) may stay in flight
        if (DIST == 1) wait_vmcnt<0>();
        else if (st && ld) wait_vmcnt<SPW + LMIN + CMIN>();
        else if (ld) wait_vmcnt<LMIN + CMIN>();
        else if (st) wait_vmcnt<SPW>();
        else wait_vmcnt<0>();
        lds_barrier();
    }
    if (n_my > 0) store_tile((n_my - 1) % NBUF, p_img, p_oy0, p_ox0);
}

#ifdef RF_PROBES
static int conv3_ws_variant() { return knob(K_CONV3WS); }        // probe knob RF_CONV3WS: 1 = the product; 0 = K_c for the merged SSH conv as well; 22 / 23 / 32 / 33 / 122 / 132: see conv3_ws_launch
#endif

template <typename T, int DEPTH, int NBUF, int SPLIT = 0>
static void conv3_ws_launch_v(hipStream_t s, Conv3Args<T> &a, int nlv, int total_tiles) {
    typedef Conv3WsCfg<T, NBUF> W;
    auto kern = conv3x3_ws_kernel<T, DEPTH, NBUF, SPLIT>;
    static std::atomic<int> resident_cache[kMaxDevices] = {};
    const int resident = kernel_residency(resident_cache, kern, W::LDS_BYTES);
    const int want = persistent_grid(total_tiles, resident);
    int grid = 0;
    for (int l = 0; l < 3; l++) {
        Conv3Level<T> &L = a.lv[l];
        if (l >= nlv) { L.gb_begin = 0x7fffffff; L.gsz = 1; L.ntiles = 0; continue; }
        long g = want == total_tiles ? L.ntiles : ((long)L.ntiles * want + total_tiles - 1) / total_tiles;
        if (g < 1) g = 1;
        if (g > L.ntiles) g = L.ntiles;
        L.gb_begin = grid;
        L.gsz = (int)g;
        grid += (int)g;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), W::LDS_BYTES, s, a);
}
// RF_CONV3WS = 10 * halo buffers + B-fragment prefetch depth (probe knob): 22, 23, 32, 33; 1 = the default = 32.  Measured per 256 images
// (tools/gpu/rounds_3_4.sh r4_call5, 4 repetitions each, profiles/r04_ws_conv_ab.txt): fp16 lock-step 87-90 us (100-104 before its pipeline was pinned),
// two halo buffers 75-85 (bimodal: the interval is then one memory round trip), THREE 56-58 at either depth; int8 37.2 lock-step, 39-41 /
// 44 warp-specialised (its GEMM is too short to hide one producer wave's issue work): int8 stays on K_c.
template <typename T>
static void conv3_ws_launch(hipStream_t s, Conv3Args<T> &a, int nlv, int total_tiles) {
#ifdef RF_PROBES
    switch (conv3_ws_variant()) {
        case 22: conv3_ws_launch_v<T, 2, 2>(s, a, nlv, total_tiles); return;
        case 23: conv3_ws_launch_v<T, 3, 2>(s, a, nlv, total_tiles); return;
        case 33: conv3_ws_launch_v<T, 3, 3>(s, a, nlv, total_tiles); return;
        case 122: conv3_ws_launch_v<T, 2, 2, 1>(s, a, nlv, total_tiles); return;     // 1xx: the 2 + 2 + 1 role split of the GEMM waves
        case 32: conv3_ws_launch_v<T, 2, 3>(s, a, nlv, total_tiles); return;
        default: break;
    }
#endif
    conv3_ws_launch_v<T, 2, 3, 1>(s, a, nlv, total_tiles);       // = 132: three halo buffers, 2 + 2 + 1 roles (57.1 -> 54.5 us over 32, tools/gpu/rounds_3_4.sh r4_call14)
}

template <typename T, int CIN, int COUT, int TH, int TW, bool UPADD, bool ALLC, bool PADROW>
static void conv3_launch(hipStream_t s, Conv3Args<T> &a, int nlv, int total_tiles) {
    typedef Conv3Cfg<T, CIN, COUT, TH, TW, ALLC, PADROW> C;
    if constexpr (sizeof(T) <= 2 && CIN == 64 && COUT == 48 && TH == 8 && TW == 8 && !UPADD && !ALLC && PADROW) {
        bool split32 = true;
        for (int l = 0; l < nlv; l++) split32 = split32 && a.lv[l].n0 == 32 && a.lv[l].out1 != nullptr;
#ifdef RF_PROBES
        const bool want = conv3_ws_variant() > 1 || (conv3_ws_variant() == 1 && sizeof(T) == 2);       // default: fp16 only (see conv3_ws_launch)
        if (want && split32) { conv3_ws_launch<T>(s, a, nlv, total_tiles); return; }
#else
        if constexpr (sizeof(T) == 2) {
            if (split32) { conv3_ws_launch<T>(s, a, nlv, total_tiles); return; }
        }
#endif
    }
    auto kern = conv3x3_kernel<T, CIN, COUT, TH, TW, UPADD, ALLC, PADROW>;
    static std::atomic<int> resident_cache[kMaxDevices] = {};
    const int resident = kernel_residency(resident_cache, kern, C::LDS_BYTES);
    // grid: persistent size for the whole launch, shared out to the levels in proportion to their tiles
    const int want = sizeof(T) <= 2 ? persistent_grid(total_tiles, resident) : total_tiles;
    int grid = 0;
    for (int l = 0; l < 3; l++) {
        Conv3Level<T> &L = a.lv[l];
        if (l >= nlv) { L.gb_begin = 0x7fffffff; L.gsz = 1; L.ntiles = 0; continue; }
        long g = want == total_tiles ? L.ntiles : ((long)L.ntiles * want + total_tiles - 1) / total_tiles;
        if (g < 1) g = 1;
        if (g > L.ntiles) g = L.ntiles;
        L.gb_begin = grid;
        L.gsz = (int)g;
        grid += (int)g;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), C::LDS_BYTES, s, a);
}

template <typename T, int CIN, int COUT, int TH, int TW, bool ALLC = false, bool PADROW = false>
static TileInfo conv3_dispatch(hipStream_t s, const Conv3Params<T> *p, int nlv, int h, int w) {
    typedef Conv3Cfg<T, CIN, COUT, TH, TW, ALLC, PADROW> C;
    TileInfo ti{TH, TW, C::LDS_BYTES, ((w + TW - 1) / TW) * ((h + TH - 1) / TH)};
    if (!p) return ti;
    Conv3Args<T> a;
    int total = 0;
    for (int l = 0; l < 3; l++) {
        const Conv3Params<T> &q = p[l < nlv ? l : nlv - 1];
        int tiles_x = (q.w_ + TW - 1) / TW, tiles_y = (q.h + TH - 1) / TH;
        a.lv[l] = Conv3Level<T>{q.in, q.up, q.w, q.b, q.m, q.out0, q.out1, q.in_ld, q.in_off, q.ld0, q.off0, q.n0, q.ld1, q.off1,
                                q.h, q.w_, tiles_x, tiles_y, q.n * tiles_x * tiles_y, 0, 1, q.a_lat, q.a_up,
                                (sizeof(T) == 1 && q.a_lat == 1.f && q.a_up == 1.f && !q.blend_fp32) ? 1 : 0};      // blend_fp32: test knob RF_BLEND_FP32, read by the engine
        if (l < nlv) total += q.n * tiles_x * tiles_y;
    }
    if (p[0].up) {
        if constexpr (CIN == 64 && COUT == 64) conv3_launch<T, CIN, COUT, TH, TW, true, ALLC, PADROW>(s, a, nlv, total);
        else throw LaunchUnsupported("fused upsample + add: only the 64 -> 64 aggregation conv has a kernel instance");
    } else {
        conv3_launch<T, CIN, COUT, TH, TW, false, ALLC, PADROW>(s, a, nlv, total);
    }
    return ti;
}

// probe knob (tools/probes): RF_CONV3=0 selects the round-1 wave split (channel tiles over the waves) for A/B measurements; -1 = the product
#ifdef RF_PROBES
static int conv3_variant() { return knob(K_CONV3); }
#endif
// probe knob RF_CONV3UPWS: 1 = auto (see conv3_select); 0 = K_c for the aggregation convs; 2 / 3 = halo buffers of the producer-wave kernel (K_c''); 12 / 13: per-wave DMA
static int conv3_up_ws_variant() { return knob(K_CONV3UPWS); }

template <typename T, int DEPTH, int NBUF, bool PRODUCER_WAVE>
static void conv3_up_ws_launch(hipStream_t s, const Conv3Params<T> &q) {
    typedef Conv3UpWsCfg<T, NBUF> W;
#ifdef RF_PROBES
    auto kern = PRODUCER_WAVE ? conv3x3_up_ws_kernel<T, DEPTH, NBUF> : conv3x3_up_dma_kernel<T, DEPTH, NBUF>;
#else
    static_assert(!PRODUCER_WAVE, "the producer-wave aggregation conv is a probe-build kernel");
    auto kern = conv3x3_up_dma_kernel<T, DEPTH, NBUF>;
#endif
    constexpr int THREADS = PRODUCER_WAVE ? W::THREADS : kThreads;
    static std::atomic<int> resident_cache[kMaxDevices] = {};
    const int dev = launch_device();
    int resident = resident_cache[dev].load(std::memory_order_acquire);
    if (!resident) {
        set_max_lds(kern, W::LDS_BYTES);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, THREADS, W::LDS_BYTES) != hipSuccess || nb < 1) nb = 1;
        resident = nb;
        resident_cache[dev].store(resident, std::memory_order_release);
    }
    const int tiles_x = (q.w_ + 7) / 8, tiles_y = (q.h + 7) / 8;
    Conv3Args<T> a;
    for (int l = 0; l < 3; l++)
        a.lv[l] = Conv3Level<T>{q.in, q.up, q.w, q.b, q.m, q.out0, q.out1, q.in_ld, q.in_off, q.ld0, q.off0, q.n0, q.ld1, q.off1,
                                q.h, q.w_, tiles_x, tiles_y, q.n * tiles_x * tiles_y, 0, 1, q.a_lat, q.a_up,
                                (sizeof(T) == 1 && q.a_lat == 1.f && q.a_up == 1.f && !q.blend_fp32) ? 1 : 0};
    const int total = q.n * tiles_x * tiles_y;
    if (total == 0) return;
    a.lv[0].gsz = persistent_grid(total, resident);
    hipLaunchKernelGGL(kern, dim3(a.lv[0].gsz), dim3(THREADS), W::LDS_BYTES, s, a);
}

template <typename T>
static TileInfo conv3_select(hipStream_t s, const Conv3Params<T> *p, int nlv, int cin, int cout, int h, int w) {
    if constexpr (sizeof(T) <= 2) {
        // the aggregation convs (fused upsample + add), warp-specialised: maps whose sides are even (tile origins must be: the upsample's
        // parities are thread constants) and a single output tensor
        // Default (RF_CONV3UPWS unset = 1): fp16 maps of >= 48 x 48 pixels take the variant where every wave issues its own share of the
        // LDS-DMA (two ring buffers): rf_c1_aggr 77.8 -> 73.6 us per 256 images; the 28 x 28 map of rf_c2_aggr and the int8 engine measured
        // equal or slower and stay on K_c (tools/gpu/rounds_3_4.sh r4_call12, profiles/r04_rejected_ws_variants.txt).
        const int upv = conv3_up_ws_variant() == 1 ? ((sizeof(T) == 2 && p && (long)p[0].h * p[0].w_ >= 48 * 48) ? 12 : 0) : conv3_up_ws_variant();
        if (p && nlv == 1 && p[0].up && cin == 64 && cout == 64 && upv >= 2 && p[0].h % 2 == 0 && p[0].w_ % 2 == 0 &&
            p[0].n0 == 64 && p[0].in_ld == 64) {
#ifdef RF_PROBES
            switch (upv) {                                             // 2 / 3: producer wave, 2 / 3 ring buffers; 12 / 13: every wave its own share
                case 2: conv3_up_ws_launch<T, 2, 2, true>(s, p[0]); break;
                case 3: conv3_up_ws_launch<T, 2, 3, true>(s, p[0]); break;
                case 12: conv3_up_ws_launch<T, 2, 2, false>(s, p[0]); break;
                default: conv3_up_ws_launch<T, 2, 3, false>(s, p[0]); break;
            }
#else
            conv3_up_ws_launch<T, 2, 2, false>(s, p[0]);
#endif
            return TileInfo{8, 8, Conv3UpWsCfg<T, 3>::LDS_BYTES, ((w + 7) / 8) * ((h + 7) / 8)};
        }
#ifdef RF_PROBES
        // fp16 / int8: every wave owns all output channels of its pixels (Conv3Cfg ALLC), 8x8 tiles
        const int v = conv3_variant();
        if (v >= 1) {
            if (cin == 64 && cout == 64) return v == 2 ? conv3_dispatch<T, 64, 64, 4, 16, true>(s, p, nlv, h, w) : conv3_dispatch<T, 64, 64, 8, 8, true>(s, p, nlv, h, w);
            if (cin == 64 && cout == 48) return v == 2 ? conv3_dispatch<T, 64, 48, 8, 16, true>(s, p, nlv, h, w) : conv3_dispatch<T, 64, 48, 8, 8, true>(s, p, nlv, h, w);
            if (cin == 16 && cout == 32) return conv3_dispatch<T, 16, 32, 8, 8, true>(s, p, nlv, h, w);
            if (cin == 16 && cout == 16) return conv3_dispatch<T, 16, 16, 8, 8, true>(s, p, nlv, h, w);
            return TileInfo{0, 0, 0, 0};
        }
        if (v == -1) {          // round-1 wave split, halo rows padded to the bank row (conflict-free B-fragment reads)
            if (cin == 64 && cout == 64) return conv3_dispatch<T, 64, 64, 4, 8, false, true>(s, p, nlv, h, w);
            if (cin == 16 && cout == 32) return conv3_dispatch<T, 16, 32, 8, 8, false, true>(s, p, nlv, h, w);
            if (cin == 64 && cout == 48) return conv3_dispatch<T, 64, 48, 8, 8, false, true>(s, p, nlv, h, w);
            if (cin == 16 && cout == 16) return conv3_dispatch<T, 16, 16, 8, 8, false, true>(s, p, nlv, h, w);
        }
#else
        // the product: round-1 wave split, halo rows padded to the bank row (conflict-free B-fragment reads); the 16-channel convs of the SSH tail
        // live in ssh_tail_kernel (fp16 / int8), so only the aggregation conv and the merged SSH 64 -> 48 conv have instances here
        if (cin == 64 && cout == 64) return conv3_dispatch<T, 64, 64, 4, 8, false, true>(s, p, nlv, h, w);
        if (cin == 64 && cout == 48) return conv3_dispatch<T, 64, 48, 8, 8, false, true>(s, p, nlv, h, w);
        return TileInfo{0, 0, 0, 0};
#endif
    }
#ifndef RF_PROBES
    if constexpr (sizeof(T) > 2)          // (fp16 / int8 returned above; RF_CONV3=0 of the probe build sends them here too)
#endif
    {
        // single-level small maps (stride 32 / 16 at 448^2: 14x14, 28x28) get 4x8 tiles for more workgroups where the
        // channel tiles still split over 4 waves (COUT % 32 == 0); everything else 8x8
        const bool small = nlv == 1 && (size_t)h * w <= 32 * 32;
        if (cin == 64 && cout == 64)
            return conv3_dispatch<T, 64, 64, 4, 8>(s, p, nlv, h, w);
        if (cin == 16 && cout == 32)
            return small ? conv3_dispatch<T, 16, 32, 4, 8>(s, p, nlv, h, w) : conv3_dispatch<T, 16, 32, 8, 8>(s, p, nlv, h, w);
        if (cin == 64 && cout == 48) return conv3_dispatch<T, 64, 48, 8, 8>(s, p, nlv, h, w);
        if (cin == 16 && cout == 16) return conv3_dispatch<T, 16, 16, 8, 8>(s, p, nlv, h, w);
    }
    return TileInfo{0, 0, 0, 0};
}

template <typename T> void launch_conv3x3(hipStream_t s, const Conv3Params<T> *levels, int nlevels) {
    if (nlevels < 1 || nlevels > 3) throw LaunchUnsupported("conv3x3: 1..3 levels per launch");
    for (int l = 1; l < nlevels; l++)
        if (levels[l].cin != levels[0].cin || levels[l].cout != levels[0].cout || levels[l].up)
            throw LaunchUnsupported("conv3x3: the levels of one launch must share (cin, cout) and have no fused upsample");
    TileInfo ti = conv3_select<T>(s, levels, nlevels, levels[0].cin, levels[0].cout, levels[0].h, levels[0].w_);
    if (ti.th == 0) throw LaunchUnsupported("no 3x3 kernel instance for this (cin, cout)");
}
template <typename T> TileInfo conv3x3_tile_info(int cin, int cout, int h, int w) {
    return conv3_select<T>(nullptr, nullptr, 1, cin, cout, h, w);
}
template void launch_conv3x3<half_t>(hipStream_t, const Conv3Params<half_t> *, int);
template void launch_conv3x3<float>(hipStream_t, const Conv3Params<float> *, int);
template void launch_conv3x3<int8_t>(hipStream_t, const Conv3Params<int8_t> *, int);
template TileInfo conv3x3_tile_info<half_t>(int, int, int, int);
template TileInfo conv3x3_tile_info<float>(int, int, int, int);
template TileInfo conv3x3_tile_info<int8_t>(int, int, int, int);

// =============================================================================================
// K_c2  the tail of the SSH context module in ONE launch (fp16 / int8 engines):
//       conv_b = context_conv2 (16 ch, -> concat[32:48]) || context_conv3_1 (16 ch, + ReLU)      3x3, 16 -> 32   (prototxt :1285-1380 etc.)
//       conv_c = context_conv3_2 (16 ch, -> concat[48:64]) on context_conv3_1's output            3x3, 16 -> 16
//   As two conv3x3<16,*> launches these were the only kernels of the path bound by scalar bookkeeping (SQ: SALU / VALU 1.7-2.4:
//   ~95 SALU per 8x8 tile for the tile walk, three per-image descriptors and the level select, against ~30 VALU + 5 MFMA), and the
//   16-channel context_conv3_1 map made an HBM round trip in between.  Here a workgroup computes conv_b on the 10 x 10 region its
//   8 x 8 tile of conv_c needs (recompute 1.56x of a 144-MAC-per-output conv: nothing), keeps context_conv3_1 in LDS (zeros
//   outside the map = conv_c's padding), and writes concat[32:64] as one 32-channel run per pixel.  Per tile: 1 staged halo
//   (12 x 12 x 16 ch), 2 barriers (out tile double buffered), 14 + 4 (pixel tile, channel tile) GEMM units of KCH MFMAs, one store.  All three FPN levels in
//   one grid, persistent + prefetching like K_c.
// =============================================================================================
template <typename T> struct SshTailCfg {
    typedef Mma<T> M;
    static constexpr int VEC = Vec<T>::N;
    static constexpr int TH = 8, TW = 8, P = TH * TW;                  // conv_c outputs per tile
    static constexpr int R1 = TW + 2, N1 = (TH + 2) * R1, PT1 = (N1 + 15) / 16;      // conv_b region 10 x 10 = 100 px -> 7 MFMA pixel tiles
    static constexpr int HC = TW + 4, NH = (TH + 4) * HC;              // context_conv1 halo 12 x 12
    static constexpr int LDI = lds_row<T>(16);                         // pixel pitch of the 16-channel tiles (32 bytes)
    static constexpr int LDO = lds_row<T>(32);                         // out tile: 32 channels per pixel
    static constexpr int KTOT = 9 * 16, KCH = (KTOT + M::K - 1) / M::K;
    static constexpr int CPV = 16 / VEC;                               // 16-byte items per halo pixel
    static constexpr int STAGE_ITEMS = NH * CPV, NPF = (STAGE_ITEMS + kThreads - 1) / kThreads;
    static constexpr size_t IN_BYTES = sizeof(T) * (size_t)(NH * LDI), MID_BYTES = sizeof(T) * (size_t)(PT1 * 16 * LDI),
                            OUT_BYTES = sizeof(T) * (size_t)(P * LDO);
    static constexpr size_t LDS_BYTES = IN_BYTES + MID_BYTES + 2 * OUT_BYTES;      // the out tile is double buffered (2 barriers per tile)
    static_assert(IN_BYTES % 16 == 0 && MID_BYTES % 16 == 0, "LDS carve must stay 16-byte aligned");
};

template <typename T>
struct SshTailLevel {
    const T *in; const T *wb; const float *bb; const float *mb; const T *wc; const float *bc; const float *mc; T *cat;
    int h, w_, tiles_x, tiles_y, ntiles, gb_begin, gsz;
};
template <typename T> struct SshTailArgs { SshTailLevel<T> lv[3]; };

// 4 consecutive output channels of one pixel as packed storage bits (what store_acc writes): fp16 -> 8 bytes, int8 -> 4 bytes in .x
template <typename T, typename ACC>
__device__ __forceinline__ uint2 pack_acc(f32x4 mult, f32x4 bv, ACC acc) {
    uint2 h = {0u, 0u};
    if constexpr (sizeof(T) == 2) {
        h.x = pack_f16((float)acc[0], (float)acc[1], true);
        h.y = pack_f16((float)acc[2], (float)acc[3], true);
    } else {
        static_assert(sizeof(T) == 1, "fp16 / int8 engines only");
#pragma unroll
        for (int r = 0; r < 4; r++)
            h.x = __builtin_amdgcn_cvt_pk_u8_f32(RF_RNDNE(__builtin_amdgcn_fmed3f(fmaf((float)acc[r], mult[r], bv[r]), 0.f, 127.f)), r, h.x);
    }
    return h;
}
template <typename T> __device__ __forceinline__ void store_packed4(T *dst, uint2 h) {
    if constexpr (sizeof(T) == 2) *(uint2 *)dst = h;
    else *(uint32_t *)dst = h.x;
}

template <typename T, int OCC>
__global__ __launch_bounds__(kThreads, OCC) void ssh_tail_kernel(SshTailArgs<T> a) {
    typedef SshTailCfg<T> C;
    typedef typename Vec<T>::type V;
    typedef Mma<T> M;
    typedef typename M::Frag Frag;
    constexpr int VEC = C::VEC, TH = C::TH, TW = C::TW, P = C::P, R1 = C::R1, N1 = C::N1, PT1 = C::PT1, HC = C::HC;
    constexpr int LDI = C::LDI, LDO = C::LDO, KTOT = C::KTOT, KCH = C::KCH, CPV = C::CPV, NPF = C::NPF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *s_in = (T *)smem;                                          // context_conv1 halo, 12 x 12 px
    T *s_mid = (T *)(smem + C::IN_BYTES);                         // context_conv3_1 region, 10 x 10 px (zeros outside the map)
    T *s_out = (T *)(smem + C::IN_BYTES + C::MID_BYTES);          // concat[32:64] of the 8 x 8 tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gbid = xcd_remap(blockIdx.x, gridDim.x);
    const int lvl = (gbid >= a.lv[1].gb_begin ? 1 : 0) + (gbid >= a.lv[2].gb_begin ? 1 : 0);
    const SshTailLevel<T> &L = a.lv[lvl];
    const int first = gbid - L.gb_begin, G = L.gsz, ntiles = L.ntiles;
    const int lh = L.h, lw = L.w_;
    const int kg = lane >> 4;

    // ---- once per workgroup: the wave's weight fragments (conv_b: channel tile wave & 1; conv_c: its one tile), biases
    const int ctb = wave & 1, ptb0 = wave >> 1;                   // conv_b units: channel tile ctb, pixel tiles ptb0 + 2 i
    Frag wb[1][KCH], wc[1][KCH];
    {
        const Frag *sb = (const Frag *)L.wb + (size_t)ctb * KCH * 64 + lane, *sc = (const Frag *)L.wc + lane;
#pragma unroll
        for (int kc = 0; kc < KCH; kc++) { wb[0][kc] = sb[kc * 64]; wc[0][kc] = sc[kc * 64]; }
    }
    const f32x4 bias_b = *(const f32x4 *)(L.bb + acc_cout(ctb, lane, 0)), mult_b = load_mult(L.mb, acc_cout(ctb, lane, 0));
    const f32x4 bias_c = *(const f32x4 *)(L.bc + acc_cout(0, lane, 0)), mult_c = load_mult(L.mc, acc_cout(0, lane, 0));

    // per-lane constants: conv_b pixels (region index, clamped for the tail lanes of the last pixel tile), conv_c pixel
    int pb1[4], ryx1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int pt = ptb0 + 2 * i;
        int p = pt * 16 + (lane & 15);
        const bool real = pt < PT1 && p < N1;
        p = real ? p : N1 - 1;
        const int ry = p / R1, rx = p % R1;
        pb1[i] = (ry * HC + rx) * LDI;
        ryx1[i] = real ? (ry << 8 | rx) : -1;
    }
    const int p2 = wave * 16 + (lane & 15);                       // conv_c: pixel tile = wave
    const int pb2 = ((p2 / TW) * R1 + p2 % TW) * LDI;

    // ---- halo of a tile -> registers (unconditional buffer loads; rows by the range check, columns by a poisoned offset)
    V pre[NPF];
    int koff[NPF], kdx[NPF];
#pragma unroll
    for (int k = 0; k < NPF; k++) {
        int i = tid + k * kThreads;
        i = i < C::STAGE_ITEMS ? i : C::STAGE_ITEMS - 1;
        const int pix = i / CPV, cv = i % CPV;
        koff[k] = (((pix / HC) * lw + pix % HC) * 16 + cv * VEC) * (int)sizeof(T);
        kdx[k] = pix % HC;
    }
    const unsigned in_img_bytes = (unsigned)(lh * lw * 16) * (unsigned)sizeof(T);
    auto fetch = [&](int tx, int ty, int img) {
        const auto rs = image_rsrc(L.in + (size_t)img * lh * lw * 16, in_img_bytes);
        const int iy0 = ty * TH - 2, ix0 = tx * TW - 2;
        const int sbase = (iy0 * lw + ix0) * 16 * (int)sizeof(T);
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const unsigned off = (unsigned)(ix0 + kdx[k]) < (unsigned)lw ? (unsigned)(koff[k] + sbase) : kOobOffset;
            pre[k] = buf_load16<V>(rs, off);
        }
    };
    // ---- finished tile -> concat[32:64]: one 32-channel run per pixel
    auto store_tile = [&](const T *s_res, int img, int oy0, int ox0) {
        constexpr int OPV = 32 / VEC;
        const auto ro = image_rsrc(L.cat + (size_t)img * lh * lw * 64 + 32, (unsigned)(lh * lw * 64 - 32) * (unsigned)sizeof(T));
        const int pbase = oy0 * lw + ox0;
        for (int i = tid; i < P * OPV; i += kThreads) {
            const int p = i / OPV, cv = i % OPV;
            const int py = p / TW, px = p % TW;
            const unsigned off = ox0 + px < lw ? (unsigned)(((pbase + py * lw + px) * 64 + cv * VEC) * (int)sizeof(T)) : kOobOffset;
            buf_store16(ro, off, *(const V *)(s_res + p * LDO + cv * VEC));
        }
    };
    const TileStep step(G, L.tiles_x, L.tiles_y);
    TileCoord cur(first, L.tiles_x, L.tiles_y), nxt = cur;
    step.advance(nxt);
    if (first < ntiles) fetch(cur.tx, cur.ty, cur.img);
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the once-per-workgroup loads have landed before the tile loop

    auto xf_tap = [&](int kc, int rowp) -> int {       // LDS element offset of this lane's K slice in chunk kc, or -1 past the 9 taps
        const int kb = kc * M::K + kg * M::KPL;        // k = tap * 16 + c
        const int tap = kb / 16, c = kb % 16;
        return kb < KTOT ? (tap / 3) * rowp + (tap % 3) * LDI + c : -1;
    };

    // Two barriers per tile: the out tile is double buffered and the previous tile is stored after this tile's first barrier (s_out[b ^ 1] was
    // completed by conv_c(t-1) before that barrier and is next written after the next tile's first barrier; s_in is rewritten by the next
    // staging after conv_b's readers have passed the second barrier; s_mid by conv_b(t+1) after the next first barrier, which follows
    // conv_c(t)'s reads).
    int p_img = -1, p_oy0 = 0, p_ox0 = 0;
    int buf = 0;
    constexpr int O_ELEMS = (int)(C::OUT_BYTES / sizeof(T));
    for (int t = first; t < ntiles; t += G) {
        const int oy0 = cur.ty * TH, ox0 = cur.tx * TW, img = cur.img;
        T *const s_out_b = s_out + buf * O_ELEMS;
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int i = tid + k * kThreads;
            if (i < C::STAGE_ITEMS) *(V *)(s_in + (i / CPV) * LDI + (i % CPV) * VEC) = pre[k];
        }
        if (t + G < ntiles) fetch(nxt.tx, nxt.ty, nxt.img);
        cur = nxt;
        step.advance(nxt);
        __syncthreads();
        if (p_img >= 0) store_tile(s_out + (buf ^ 1) * O_ELEMS, p_img, p_oy0, p_ox0);
        p_img = img; p_oy0 = oy0; p_ox0 = ox0;

        // ---- conv_b on the 10 x 10 region: this wave's channel tile x its (up to) 4 pixel tiles, two at a time (four accumulators
        //      + their B-fragment queue pushed the kernel over the 128-VGPR budget of 4 workgroups per CU)
#pragma unroll
        for (int half = 0; half < 2; half++) {
            typename M::Acc acc[1][2];
#pragma unroll
            for (int j = 0; j < 2; j++) acc[0][j] = acc_init<T>(bias_b);
            gemm_stationary<T, 1, 2, KCH, 2>(acc, wb, [&](int j, int kc) -> Frag {
                const int o = xf_tap(kc, HC * LDI);
                return o >= 0 ? *(const Frag *)(s_in + pb1[2 * half + j] + o) : M::zero();
            });
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const int j = 2 * half + jj;
                if (ryx1[j] < 0) continue;
                const int ry = ryx1[j] >> 8, rx = ryx1[j] & 0xff;
                const uint2 h = pack_acc<T>(mult_b, bias_b, acc[0][jj]);
                if (ctb == 0) {
                    // context_conv2 -> concat[32:48]: only the tile's own 8 x 8 pixels
                    if ((unsigned)(ry - 1) < (unsigned)TH && (unsigned)(rx - 1) < (unsigned)TW)
                        store_packed4<T>(s_out_b + ((ry - 1) * TW + rx - 1) * LDO + kg * 4, h);
                } else {
                    // context_conv3_1 on the whole region; outside the map it is conv_c's ZERO padding
                    const int y = oy0 - 1 + ry, x = ox0 - 1 + rx;
                    const bool inside = (unsigned)y < (unsigned)lh && (unsigned)x < (unsigned)lw;
                    const uint2 z = {0u, 0u};
                    store_packed4<T>(s_mid + (ry * R1 + rx) * LDI + kg * 4, inside ? h : z);
                }
            }
        }
        __syncthreads();

        // ---- conv_c on the 8 x 8 tile: pixel tile = wave
        {
            typename M::Acc acc[1][1];
            acc[0][0] = acc_init<T>(bias_c);
            gemm_stationary<T, 1, 1, KCH, 2>(acc, wc, [&](int, int kc) -> Frag {
                const int o = xf_tap(kc, R1 * LDI);
                return o >= 0 ? *(const Frag *)(s_mid + pb2 + o) : M::zero();
            });
            store_packed4<T>(s_out_b + p2 * LDO + 16 + kg * 4, pack_acc<T>(mult_c, bias_c, acc[0][0]));
        }
        buf ^= 1;
    }
    __syncthreads();                                               // the last tile's conv_c has no closing barrier
    if (p_img >= 0) store_tile(s_out + (buf ^ 1) * O_ELEMS, p_img, p_oy0, p_ox0);
}

int ssh_tail_variant() { return knob(K_SSHTAIL); }      // probe knob RF_SSHTAIL: 1 = the product; 0 = two conv3x3<16,*> launches, 2 = 3 workgroups per CU

template <typename T> void launch_ssh_tail(hipStream_t s, const SshTailParams<T> *levels, int nlevels) {
    typedef SshTailCfg<T> C;
    if (nlevels < 1 || nlevels > 3) throw LaunchUnsupported("ssh tail: 1..3 levels per launch");
    // RF_SSHTAIL=2 (probe knob): the 3-workgroups-per-CU build (155 VGPRs, no spill) instead of the 4-per-CU one (128 VGPRs)
#ifdef RF_PROBES
    const bool occ3 = ssh_tail_variant() == 2;
    auto kern = occ3 ? ssh_tail_kernel<T, 3> : ssh_tail_kernel<T, 4>;
#else
    constexpr bool occ3 = false;
    auto kern = ssh_tail_kernel<T, 4>;
#endif
    static std::atomic<int> resident_cache[2][kMaxDevices] = {};
    const int resident = kernel_residency(resident_cache[occ3 ? 1 : 0], kern, C::LDS_BYTES);
    SshTailArgs<T> a;
    int total = 0;
    for (int l = 0; l < 3; l++) {
        const SshTailParams<T> &q = levels[l < nlevels ? l : nlevels - 1];
        const int tiles_x = (q.w_ + C::TW - 1) / C::TW, tiles_y = (q.h + C::TH - 1) / C::TH;
        a.lv[l] = SshTailLevel<T>{q.in, q.wb, q.bb, q.mb, q.wc, q.bc, q.mc, q.cat, q.h, q.w_, tiles_x, tiles_y, q.n * tiles_x * tiles_y, 0, 1};
        if (l < nlevels) total += q.n * tiles_x * tiles_y;
    }
    if (total == 0) return;
    const int want = persistent_grid(total, resident);
    int grid = 0;
    for (int l = 0; l < 3; l++) {
        SshTailLevel<T> &L = a.lv[l];
        if (l >= nlevels) { L.gb_begin = 0x7fffffff; L.gsz = 1; L.ntiles = 0; continue; }
        long g = want == total ? L.ntiles : ((long)L.ntiles * want + total - 1) / total;
        if (g < 1) g = 1;
        if (g > L.ntiles) g = L.ntiles;
        L.gb_begin = grid; L.gsz = (int)g; grid += (int)g;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), C::LDS_BYTES, s, a);
}
template void launch_ssh_tail<half_t>(hipStream_t, const SshTailParams<half_t> *, int);
template void launch_ssh_tail<int8_t>(hipStream_t, const SshTailParams<int8_t> *, int);

// =============================================================================================
// K_d  heads + softmax + decode + threshold compaction
//   reference: 3 x 1x1 Convolution + Reshape/Softmax/Reshape on the GPU (prototxt :1434-1511), 9 D2H copies
//   (trtretinafacenet.cpp:63-72) and the CPU loop RetinaFace.cpp:666-724 with bbox_pred (:378-398),
//   clip_boxes (:179-199) and landmark_pred (:418-432).  Here only above-threshold candidates leave the chip.
//   The float/double rounding points of the reference are reproduced (fp contraction off in decode_anchor).
//   One launch covers the heads of all three strides.
// =============================================================================================
constexpr int HEAD_P = 64;          // pixels per workgroup (flat, row-major)

__device__ __forceinline__ void decode_anchor(const float *__restrict__ o /*16A head outputs of this pixel*/, int a, int na,
                                              float ax1, float ay1, float ax2, float ay2, int net_w, int net_h,
                                              float conf, int anchor_index, Candidate *dst) {
#pragma clang fp contract(off)
    // bbox_pred, RetinaFace.cpp:378-398: "0.5 * (w - 1.0)" is double arithmetic, results stored as float
    const float width = ax2 - ax1 + 1.f;
    const float height = ay2 - ay1 + 1.f;
    const float ctr_x = (float)((double)ax1 + 0.5 * ((double)width - 1.0));
    const float ctr_y = (float)((double)ay1 + 0.5 * ((double)height - 1.0));
    const float *d = o + 2 * na + a * 4;
    const float pred_ctr_x = d[0] * width + ctr_x;
    const float pred_ctr_y = d[1] * height + ctr_y;
    const float pred_w = expf(d[2]) * width;
    const float pred_h = expf(d[3]) * height;
    float x1 = (float)((double)pred_ctr_x - 0.5 * ((double)pred_w - 1.0));
    float y1 = (float)((double)pred_ctr_y - 0.5 * ((double)pred_h - 1.0));
    float x2 = (float)((double)pred_ctr_x + 0.5 * ((double)pred_w - 1.0));
    float y2 = (float)((double)pred_ctr_y + 0.5 * ((double)pred_h - 1.0));
    // clip_boxes, RetinaFace.cpp:179-199 (one-sided)
    if (x1 < 0) x1 = 0;
    if (y1 < 0) y1 = 0;
    if (x2 > (float)(net_w - 1)) x2 = (float)(net_w - 1);
    if (y2 > (float)(net_h - 1)) y2 = (float)(net_h - 1);
    dst->score = conf;
    dst->x1 = x1; dst->y1 = y1; dst->x2 = x2; dst->y2 = y2;
    // landmark_pred, RetinaFace.cpp:418-432 (not clipped)
    const float *l = o + 6 * na + a * 10;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        dst->px[k] = l[2 * k] * width + ctr_x;
        dst->py[k] = l[2 * k + 1] * height + ctr_y;
    }
    dst->anchor = anchor_index;
}

template <typename T>
struct HeadLevel {
    const T *in; const T *w; const float *b; const float *m;
    float *dump_prob, *dump_bbox, *dump_lmk;
    float base[4][4];
    int hw, w_, stride, anchor_offset, blocks_per_image, blk_begin;
};
template <typename T>
struct HeadArgs {
    HeadLevel<T> lv[3];
    int net_h, net_w;
    const RunParams *params;
    Candidate *cand; int *cand_count; int cap;
    int nblk;
};

template <typename T, int NA>
__global__ __launch_bounds__(kThreads) void head_kernel(HeadArgs<T> a) {
    typedef typename Vec<T>::type V;
    typedef Mma<T> M;
    constexpr int VEC = Vec<T>::N, CIN = 64, COUT = 16 * NA, P = HEAD_P, LDA = lds_row<T>(CIN), LDOH = COUT + 1;
    constexpr int CPV = CIN / VEC;
    __shared__ __attribute__((aligned(16))) T s_a[P * LDA];
    __shared__ float s_o[P * LDOH];               // fp32 result tile, row pitch 16A + 1: conflict-free column reads

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int gbid = xcd_remap(blockIdx.x, a.nblk);
    const int lvl = (gbid >= a.lv[1].blk_begin ? 1 : 0) + (gbid >= a.lv[2].blk_begin ? 1 : 0);
    const HeadLevel<T> &L = a.lv[lvl];
    const int bid = gbid - L.blk_begin;
    const int img = bid / L.blocks_per_image;
    const int p0 = (bid % L.blocks_per_image) * P;
    const int hw = L.hw;

    constexpr int NT = COUT / 16, PT = P / 16;       // A = 2: waves = 2 (cout) x 2 (pixel halves); A = 4: 4 (cout) x 1
    typedef WaveSplit<NT, PT> WS;
    // fp16 engine: the weights are an fp16 hi + lo pair laid out along K (weights.h): K = 128, both halves read the same 64 inputs
    constexpr int KCH1 = CIN / M::K, KCH = sizeof(T) == 2 ? 2 * KCH1 : KCH1;
    const int wn = wave % WS::WN, wp = wave / WS::WN;
    GemmPipe<T, 1, WS::NJ, KCH, WS::WN> pipe;
    pipe.init(L.w, wn, lane);
    const float threshold = a.params->threshold;
    const f32x4 bias = *(const f32x4 *)(L.b + acc_cout(wn, lane, 0));
    const f32x4 mult = load_mult(L.m, acc_cout(wn, lane, 0));          // int8: w_scale * in_scale -> real logits

    const T *inb = L.in + (size_t)img * hw * CIN;
    for (int i = tid; i < P * CPV; i += kThreads) {
        int p = i / CPV, cv = i % CPV;
        V v = vzero<V, VEC>();
        if (p0 + p < hw) v = *(const V *)(inb + (size_t)(p0 + p) * CIN + cv * VEC);
        *(V *)(s_a + p * LDA + cv * VEC) = v;
    }
    __syncthreads();

    typename M::Acc acc[1][WS::NJ];
#pragma unroll
    for (int j = 0; j < WS::NJ; j++) acc[0][j] = vzero<typename M::Acc, 4>();
    pipe.run(acc, [&](int j, int kc) -> typename M::Frag {
        const int kb = (kc % KCH1) * M::K + (lane >> 4) * M::KPL;
        return *(const typename M::Frag *)(s_a + acc_pixel(wp + j * WS::WP, lane) * LDA + kb);
    });
#pragma unroll
    for (int j = 0; j < WS::NJ; j++) {
        const int p = acc_pixel(wp + j * WS::WP, lane);
        const int c0 = acc_cout(wn, lane, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) s_o[p * LDOH + c0 + r] = fmaf((float)acc[0][j][r], mult[r], bias[r]);
    }
    __syncthreads();

    if (tid < NA * P) {
        const int p = tid % P, an = tid / P;          // anchor index a is wave-uniform
        const int gp = p0 + p;
        if (gp < hw) {
            const float *o = s_o + p * LDOH;
            // Softmax over the pair (channel a, channel A+a) -- Reshape (2, -1) puts the A background maps first, then the A
            // foreground maps: Caffe subtracts the max, exponentiates, normalises
            const float s0 = o[an], s1 = o[NA + an];
            const float m = fmaxf(s0, s1);
            const float e0 = expf(s0 - m), e1 = expf(s1 - m);
            const float sum = e0 + e1;
            const float conf = e1 / sum;
            if (L.dump_prob) {
                const size_t shw = (size_t)hw;
                L.dump_prob[((size_t)img * 2 * NA + an) * shw + gp] = e0 / sum;
                L.dump_prob[((size_t)img * 2 * NA + NA + an) * shw + gp] = conf;
#pragma unroll
                for (int c = 0; c < 4; c++) L.dump_bbox[((size_t)img * 4 * NA + an * 4 + c) * shw + gp] = o[2 * NA + an * 4 + c];
#pragma unroll
                for (int c = 0; c < 10; c++) L.dump_lmk[((size_t)img * 10 * NA + an * 10 + c) * shw + gp] = o[6 * NA + an * 10 + c];
            }
            if (conf > threshold) {                    // "if (conf <= threshold) continue", RetinaFace.cpp:693
                const int iy = gp / L.w_, ix = gp % L.w_;
                const float sx = (float)(ix * L.stride), sy = (float)(iy * L.stride);
                const int slot = atomicAdd(&a.cand_count[img], 1);
                if (slot < a.cap)
                    decode_anchor(o, an, NA, L.base[an][0] + sx, L.base[an][1] + sy, L.base[an][2] + sx, L.base[an][3] + sy,
                                  a.net_w, a.net_h, conf, L.anchor_offset + an * hw + gp,
                                  a.cand + (size_t)img * a.cap + slot);
            }
        }
    }
}

template <typename T> void launch_head(hipStream_t s, const HeadParams<T> *levels, int nlevels) {
    if (nlevels < 1 || nlevels > 3) throw LaunchUnsupported("heads: 1..3 strides per launch");
    HeadArgs<T> a;
    int blk = 0;
    for (int l = 0; l < 3; l++) {
        const HeadParams<T> &p = levels[l < nlevels ? l : nlevels - 1];
        HeadLevel<T> &L = a.lv[l];
        L.in = p.in; L.w = p.w; L.b = p.b; L.m = p.m;
        L.dump_prob = p.dump_prob; L.dump_bbox = p.dump_bbox; L.dump_lmk = p.dump_lmk;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) L.base[i][j] = p.base[i][j];
        L.hw = p.h * p.w_; L.w_ = p.w_; L.stride = p.stride; L.anchor_offset = p.anchor_offset;
        L.blocks_per_image = (L.hw + HEAD_P - 1) / HEAD_P;
        L.blk_begin = l < nlevels ? blk : 0x7fffffff;
        if (l < nlevels) blk += p.n * L.blocks_per_image;
    }
    const HeadParams<T> &p0 = levels[0];
    a.net_h = p0.net_h; a.net_w = p0.net_w; a.params = p0.params;
    a.cand = p0.cand; a.cand_count = p0.cand_count; a.cap = p0.cap;
    a.nblk = blk;
    // a preset without an anchor configuration (RetinaFace.cpp:225-271: "ssh", "vgg", net4/5/6 ...) has nothing to decode: the
    // reference's post-processing loops over zero anchors, so no candidate is ever produced
    if (p0.num_anchors == 0 || a.nblk == 0) return;
    if (p0.num_anchors == 4) hipLaunchKernelGGL((head_kernel<T, 4>), dim3(a.nblk), dim3(kThreads), 0, s, a);
    else if (p0.num_anchors == 2) hipLaunchKernelGGL((head_kernel<T, 2>), dim3(a.nblk), dim3(kThreads), 0, s, a);
    else throw LaunchUnsupported("heads: 2 or 4 anchors per cell");
}
template void launch_head<half_t>(hipStream_t, const HeadParams<half_t> *, int);
template void launch_head<float>(hipStream_t, const HeadParams<float> *, int);
template void launch_head<int8_t>(hipStream_t, const HeadParams<int8_t> *, int);

// =============================================================================================
// K_e  per-image NMS: total order (score desc, anchor index asc), greedy suppression with the reference's
//      +1-pixel IoU and strict ">" (RetinaFace.cpp:434-492).  One 256-thread workgroup per image:
//      bitonic sort of 64-bit keys in LDS, then a serial walk over survivors where each survivor's
//      suppression sweep is data-parallel.  fp contraction off: same roundings as the scalar CPU loop.
//      Results are written straight to the caller-visible (pinned host) result block.
// =============================================================================================
constexpr int NMS_THREADS = 256;

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(NmsParams a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int cap = a.cap;
    unsigned long long *s_key = (unsigned long long *)smem;                 // [cap]
    float4 *s_box = (float4 *)(s_key + cap);                                 // [cap]
    int *s_slot = (int *)(s_box + cap);                                      // [cap]
    int *s_kept = s_slot + cap;                                              // [max_det]
    unsigned char *s_alive = (unsigned char *)(s_kept + a.max_det);          // [cap]

    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const Candidate *cand = a.cand + (size_t)img * cap;
    const float thr = a.params->nms_threshold;
    const int n_found = a.cand_count[img];
    int n = n_found;
    if (n > cap) n = cap;
    // Typical images carry a few dozen candidates: then ONE wavefront does everything (the other three retire, so
    // every __syncthreads() below degenerates to a wave-local fence) and the order comes from a rank sort (one pass,
    // no barrier ladder).  Larger sets fall back to a 256-thread bitonic network.
    const bool small = n <= 64;
    if (small && tid >= 64) return;
    const int nthr = small ? 64 : NMS_THREADS;
    if (n <= 256) {
        for (int i = tid; i < n; i += nthr) {
            const unsigned int sbits = __float_as_uint(cand[i].score);      // scores are positive: bit order = value order
            s_key[i] = ((unsigned long long)(~sbits) << 32) | (unsigned int)cand[i].anchor;
        }
        __syncthreads();
        for (int i = tid; i < n; i += nthr) {
            const unsigned long long ki = s_key[i];
            int rank = 0;
            // keys are unique in normal operation (anchor index); the index tie-break keeps the ranks a permutation even
            // if the same anchor was appended twice (rf_profile repeats the head launch)
            for (int j = 0; j < n; j++) { const unsigned long long kj = s_key[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
            s_slot[rank] = i;
        }
        __syncthreads();
    } else {
        int npow = 512;
        while (npow < n) npow <<= 1;
        for (int i = tid; i < npow; i += nthr) {
            unsigned long long key = ~0ull;
            if (i < n) {
                const unsigned int sbits = __float_as_uint(cand[i].score);
                key = ((unsigned long long)(~sbits) << 32) | (unsigned int)cand[i].anchor;
            }
            s_key[i] = key;
            s_slot[i] = i;
        }
        __syncthreads();
        for (int k = 2; k <= npow; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < npow; i += nthr) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long ki = s_key[i], kl = s_key[l];
                        const bool up = (i & k) == 0;
                        if ((ki > kl) == up) {
                            s_key[i] = kl; s_key[l] = ki;
                            const int t = s_slot[i]; s_slot[i] = s_slot[l]; s_slot[l] = t;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    for (int i = tid; i < n; i += nthr) {
        const Candidate *c = cand + s_slot[i];
        s_box[i] = make_float4(c->x1, c->y1, c->x2, c->y2);
        s_alive[i] = 1;
    }
    __syncthreads();

    const int lane = tid & 63;
    int kept = 0;
    for (int base = 0; base < n; base += 64) {
        const int idx = base + lane;
        unsigned long long mask = __ballot(idx < n && s_alive[idx]);
        while (mask) {
            const int sel = base + __ffsll((long long)mask) - 1;
            if (tid == 0 && kept < a.max_det) s_kept[kept] = sel;
            kept++;
            const float4 sb = s_box[sel];
            const float area1 = (sb.z - sb.x + 1) * (sb.w - sb.y + 1);
            for (int i = sel + 1 + tid; i < n; i += nthr) {
                if (!s_alive[i]) continue;
                const float4 bi = s_box[i];
                const float x = fmaxf(sb.x, bi.x);
                const float y = fmaxf(sb.y, bi.y);
                const float w = fminf(sb.z, bi.z) - x + 1;
                const float h = fminf(sb.w, bi.w) - y + 1;
                if (w <= 0 || h <= 0) continue;
                const float area2 = (bi.z - bi.x + 1) * (bi.w - bi.y + 1);
                const float inter = w * h;
                if (inter / (area1 + area2 - inter) > thr) s_alive[i] = 0;
            }
            __syncthreads();
            const unsigned long long later = (sel - base) >= 63 ? 0ull : (~0ull << (sel - base + 1));
            mask = __ballot(idx < n && s_alive[idx]) & later;
        }
    }
    __syncthreads();
    const int nout = kept < a.max_det ? kept : a.max_det;
    for (int i = tid; i < nout * 16; i += nthr) {
        const int k = i >> 4, f = i & 15;
        const uint32_t *src = (const uint32_t *)(cand + s_slot[s_kept[k]]);
        ((uint32_t *)(a.out + (size_t)img * a.max_det + k))[f] = src[f];
    }
    if (tid == 0) {
        a.out_count[img] = kept;
        a.out_cand_count[img] = n_found;
        a.cand_count[img] = 0;        // re-arm the head kernels' counter for the next batch on this lane
    }
}

void launch_nms(hipStream_t s, const NmsParams &p) {
    size_t lds = (size_t)p.cap * (8 + 16 + 4 + 1) + (size_t)p.max_det * 4 + 16;
    static std::atomic<size_t> attr_bytes[kMaxDevices] = {};
    const int dev = launch_device();
    if (lds > attr_bytes[dev].load(std::memory_order_acquire)) { set_max_lds(nms_kernel, lds); attr_bytes[dev].store(lds, std::memory_order_release); }
    hipLaunchKernelGGL(nms_kernel, dim3(p.n), dim3(NMS_THREADS), lds, s, p);
}

// =============================================================================================
// Area-average downscale for frames larger than the net (TRT+NPP variant, factor < 1:
// resizeconvertion.cu:298-311 uses NPPI_INTER_SUPER, a closed-source box filter).  Each destination pixel
// averages the source rectangle it covers with fractional edge weights; aspect preserved, top-left
// anchored, the rest of the canvas stays zero (cudaMemset, RetinaFace.cpp:598).  "parity unpinned".
// =============================================================================================
__global__ __launch_bounds__(kThreads) void resize_area_kernel(const FrameDesc *__restrict__ src, uint8_t *__restrict__ dst,
                                                               int net_h, int net_w) {
    const int img = blockIdx.z;
    const FrameDesc fd = src[img];
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= net_w || y >= net_h) return;
    uint8_t *d = dst + ((size_t)img * net_h * net_w + (size_t)y * net_w + x) * 3;
    float f = fminf((float)net_w / (float)fd.cols, (float)net_h / (float)fd.rows);
    if (f > 1.f) f = 1.f;
    const int dw = (int)((float)fd.cols * f), dh = (int)((float)fd.rows * f);
    if (x >= dw || y >= dh || fd.ptr == nullptr) { d[0] = d[1] = d[2] = 0; return; }
    if (f == 1.f) {
        const uint8_t *sp = fd.ptr + (size_t)y * fd.step + x * 3;
        d[0] = sp[0]; d[1] = sp[1]; d[2] = sp[2];
        return;
    }
    const float inv = 1.f / f;
    const float sx0 = x * inv, sx1 = fminf((x + 1) * inv, (float)fd.cols);
    const float sy0 = y * inv, sy1 = fminf((y + 1) * inv, (float)fd.rows);
    float acc[3] = {0.f, 0.f, 0.f}, wsum = 0.f;
    for (int yy = (int)sy0; yy < fd.rows && (float)yy < sy1; yy++) {
        const float wy = fminf((float)(yy + 1), sy1) - fmaxf((float)yy, sy0);
        for (int xx = (int)sx0; xx < fd.cols && (float)xx < sx1; xx++) {
            const float wgt = wy * (fminf((float)(xx + 1), sx1) - fmaxf((float)xx, sx0));
            const uint8_t *sp = fd.ptr + (size_t)yy * fd.step + xx * 3;
            acc[0] += wgt * sp[0]; acc[1] += wgt * sp[1]; acc[2] += wgt * sp[2];
            wsum += wgt;
        }
    }
    for (int c = 0; c < 3; c++) {
        float v = wsum > 0.f ? acc[c] / wsum : 0.f;
        d[c] = (uint8_t)fminf(fmaxf(rintf(v), 0.f), 255.f);
    }
}

void launch_resize_area(hipStream_t s, const FrameDesc *src, uint8_t *dst, int n, int net_h, int net_w) {
    dim3 grid((net_w + 31) / 32, (net_h + 7) / 8, n);
    hipLaunchKernelGGL(resize_area_kernel, grid, dim3(kThreads), 0, s, src, dst, net_h, net_w);
}

// =============================================================================================
// Bilinear downscale for frames larger than the net: the reference's build WITHOUT NPP (CMake default), RetinaFace.cpp:585-620:
//   scale = max(cols / netW, rows / netH) in float, cv::resize(img, Size(), 1 / scale, 1 / scale)  [OpenCV INTER_LINEAR],
//   zero padding on the short side.  OpenCV is third party and absent from the reference tree; its published 8-bit fixed-point
//   algorithm (imgproc/resize.cpp: 11-bit taps, int32 horizontal pass, ((b0*(H0>>4))>>16 + (b1*(H1>>4))>>16 + 2) >> 2 vertical
//   pass) is implemented here integer for integer, and in double / float exactly where OpenCV uses them (contraction off), so it
//   is bit-identical to oracle/csrc/cv_resize_linear.h.  Frames that fit are copied 1:1.
// =============================================================================================
__device__ __forceinline__ void bilinear_tap(int d, double scale, int n_src, bool clamp, int *s0, int *s1, int *a0, int *a1) {
#pragma clang fp contract(off)
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (clamp) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
    }
    int t0 = (int)rintf((1.f - f) * 2048.f), t1 = (int)rintf(f * 2048.f);
    *a0 = t0 > 32767 ? 32767 : (t0 < -32768 ? -32768 : t0);
    *a1 = t1 > 32767 ? 32767 : (t1 < -32768 ? -32768 : t1);
    const int lo = s < 0 ? 0 : (s < n_src ? s : n_src - 1), hi = s + 1 < 0 ? 0 : (s + 1 < n_src ? s + 1 : n_src - 1);
    *s0 = lo; *s1 = hi;
}

__global__ __launch_bounds__(kThreads) void resize_bilinear_kernel(const FrameDesc *__restrict__ src, uint8_t *__restrict__ dst,
                                                                   int net_h, int net_w) {
#pragma clang fp contract(off)
    const int img = blockIdx.z;
    const FrameDesc fd = src[img];
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= net_w || y >= net_h) return;
    uint8_t *d = dst + ((size_t)img * net_h * net_w + (size_t)y * net_w + x) * 3;
    if (fd.ptr == nullptr) { d[0] = d[1] = d[2] = 0; return; }
    // float sw = 1.0 * img.cols / inputW (double arithmetic, stored to float), RetinaFace.cpp:586-589
    const float sw = (float)(1.0 * (double)fd.cols / (double)net_w), sh = (float)(1.0 * (double)fd.rows / (double)net_h);
    float scale = sw > sh ? sw : sh;
    scale = scale > 1.0 ? scale : 1.0f;
    if (!(scale > 1)) {                                   // fits: copyMakeBorder only (:621-624)
        if (x >= fd.cols || y >= fd.rows) { d[0] = d[1] = d[2] = 0; return; }
        const uint8_t *sp = fd.ptr + (size_t)y * fd.step + x * 3;
        d[0] = sp[0]; d[1] = sp[1]; d[2] = sp[2];
        return;
    }
    const double fx = (double)(1 / scale);                // `1 / scale` is a float division; cv::resize takes it as double
    const int dcols = (int)__builtin_rint((double)fd.cols * fx), drows = (int)__builtin_rint((double)fd.rows * fx);
    if (x >= dcols || y >= drows) { d[0] = d[1] = d[2] = 0; return; }
    const double sc = 1.0 / fx;
    int x0, x1, a0, a1, y0, y1, b0, b1;
    bilinear_tap(x, sc, fd.cols, true, &x0, &x1, &a0, &a1);
    bilinear_tap(y, sc, fd.rows, false, &y0, &y1, &b0, &b1);
    const uint8_t *r0 = fd.ptr + (size_t)y0 * fd.step, *r1 = fd.ptr + (size_t)y1 * fd.step;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int h0 = r0[3 * x0 + c] * a0 + r0[3 * x1 + c] * a1;
        const int h1 = r1[3 * x0 + c] * a0 + r1[3 * x1 + c] * a1;
        d[c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}

void launch_resize_bilinear(hipStream_t s, const FrameDesc *src, uint8_t *dst, int n, int net_h, int net_w) {
    dim3 grid((net_w + 31) / 32, (net_h + 7) / 8, n);
    hipLaunchKernelGGL(resize_bilinear_kernel, grid, dim3(kThreads), 0, s, src, dst, net_h, net_w);
}

// One-off check per device that v_cvt_pk_u8_f32 is clamp(rint(x), 0, 255) with ties to even (the int8 epilogues have no separate
// rounding instruction): 64 ties, their neighbours and out-of-range values.  Returns the number of mismatches (0 = as assumed).
__global__ void cvt_pk_u8_probe_kernel(int *bad) {
    const int i = threadIdx.x;
    const float tie = (float)i + 0.5f;                           // positive: the neighbouring floats are the bit patterns +- 1
    const float vals[4] = {tie, __builtin_bit_cast(float, __builtin_bit_cast(int, tie) - 1), __builtin_bit_cast(float, __builtin_bit_cast(int, tie) + 1),
                           (float)i * 5.f - 30.f};
    int wrong = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned got = __builtin_amdgcn_cvt_pk_u8_f32(vals[k], 0, 0u);
        const float want = fminf(fmaxf(rintf(vals[k]), 0.f), 255.f);
        wrong += got != (unsigned)want;
    }
    if (wrong) atomicAdd(bad, wrong);
}
// 0 = rounds as assumed, 1 = mismatch (an ISA property of the device: cached), -1 = the probe itself could not run (a transient runtime
// error -- out of memory, a failed copy or launch: reported as such by the engine and NOT cached, so a later rf_create tries again).
// Runs on a private non-blocking stream: the legacy null stream would synchronise with -- and could invalidate -- another thread's
// blocking-stream work or graph capture on the same device.
int cvt_pk_u8_selfcheck() {
    static std::atomic<int> result[kMaxDevices] = {};          // 0 = not run, 1 = ok, 2 = mismatch
    const int dev = launch_device();
    int r = result[dev].load(std::memory_order_acquire);
    if (!r) {
        int *d = nullptr, *h = nullptr;
        hipStream_t st = nullptr;
        bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipMalloc((void **)&d, sizeof(int)) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&h, sizeof(int), hipHostMallocDefault) == hipSuccess;
        if (ok) {
            *h = -1;
            ok = hipMemsetAsync(d, 0, sizeof(int), st) == hipSuccess;
            if (ok) {
                hipLaunchKernelGGL(cvt_pk_u8_probe_kernel, dim3(1), dim3(64), 0, st, d);
                ok = hipGetLastError() == hipSuccess;
            }
            ok = ok && hipMemcpyAsync(h, d, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess;
            ok = ok && hipStreamSynchronize(st) == hipSuccess;
            ok = ok && *h >= 0;
        }
        const int mismatches = ok ? *h : -1;
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        if (st) (void)hipStreamDestroy(st);
        if (!ok) { (void)hipGetLastError(); return -1; }
        r = mismatches == 0 ? 1 : 2;
        result[dev].store(r, std::memory_order_release);       // only a completed comparison is cached
    }
    return r == 1 ? 0 : 1;
}

#ifdef RF_KERNEL_TRACE
extern "C" int rf_trace_select(int kernel_id, unsigned grid) {
    static unsigned long long zeros[kTraceBlocks * kTraceSlots];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), zeros, sizeof(zeros));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_key), &grid, sizeof(unsigned));
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_kernel), &kernel_id, sizeof(int));
}
extern "C" int rf_trace_read(unsigned long long *dst, int nblocks) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace), (size_t)nblocks * kTraceSlots * sizeof(unsigned long long));
}
#endif

}  // namespace rf
