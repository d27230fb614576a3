// kernels.h -- launch API of the hand-written gfx950 kernels (kernels.hip).  Everything NHWC.
// T = rf::half_t (fp16 storage, fp32 accumulate, v_mfma_f32_16x16x32_f16) or float (fp32 storage,
// exact-f32 v_mfma_f32_16x16x4_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>

namespace rf {

typedef _Float16 half_t;

// RF_PROBES: the probe build (make probe), which also holds the measured-and-rejected kernel variants and the RF_* probe knobs (knobs.h)
#ifdef RF_PROBES
constexpr bool kProbeBuild = true;
#else
constexpr bool kProbeBuild = false;
#endif

// a request the engine has no kernel instance / configuration for (C ABI: RF_ERR_UNSUPPORTED)
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

// depthwise weights are stored in the activation type, except fp32 for int8 activations
template <typename T> struct DwWeightT { typedef T type; };
template <> struct DwWeightT<int8_t> { typedef float type; };

// The launch helpers keep per-device state (CU count, LDS attribute / occupancy of each kernel instance); the engine tells
// them which device the calling thread is bound to (engine.cpp DeviceGuard).
void bind_launch_device(int device);
// ... and how many CUs the stream it is about to launch on may use (0 = all of the device's): persistent grids are sized by it
void bind_launch_cus(int cus);
// int8 engines: 0 when v_cvt_pk_u8_f32 on the bound device rounds to nearest even and saturates (what the requantising epilogues rely on),
// 1 when it does not (cached per device), -1 when the probe could not run (a runtime error: not cached)
int cvt_pk_u8_selfcheck();
constexpr int kMaxDevices = 64;      // device ordinals a process may use (per-device launch state is sized by it; engine / multi.cpp enforce it)

// One input frame: CV_8UC3 BGR, row y at ptr + y*step (cv::Mat data/step; RetinaFace.cpp:594).
struct FrameDesc {
    const uint8_t *ptr;
    int rows, cols, step, pad_;
};

// Pre-NMS candidate written by the head kernel: the reference's FaceDetectInfo (RetinaFace.h:37-42) + the
// global anchor index that defines the NMS tie order.  64 bytes.
struct Candidate {
    float score;
    float x1, y1, x2, y2;
    float px[5], py[5];
    int32_t anchor;
};

// Per-launch scalars; they live in device memory (copied with the frame table) so a captured hipGraph stays valid.
struct RunParams {
    float threshold;       // keep iff conf > threshold   (RetinaFace.cpp:693)
    float nms_threshold;   // suppress iff IoU > nms      (RetinaFace.cpp:486)
    int32_t n_images;
    int32_t pad_;
};

// ---- K_a: preprocess (BGR u8 HWC -> RGB, top-left placement on a zero canvas; resizeconvertion.cu:46-63,
//      165-185, 279-316 with factor 1) fused with mobilenet0_conv0 (3x3 s2 p1 3->8) + BN + ReLU.
template <typename T>
void launch_conv0(hipStream_t s, const FrameDesc *frames, T *out, const float *w, const float *b, int n, int net_h,
                  int net_w);

// ---- K_a' (fp16 engine): K_a fused with the first depthwise/pointwise block; conv0 on MFMA (hi+lo split weights).
template <typename TO>
struct StemParams {
    const FrameDesc *frames; TO *out;              // out: [n][net_h/2][net_w/2][16], fp16 or int8
    const half_t *w0; const float *b0;             // conv0: 4 A fragments (hi/lo x k<32/k>=32), K = (ky,kx,BGRX) 36 -> 64
    const half_t *w0_raw = nullptr;                // ... for the raw-row staging (weights.h c0_raw_), nullptr = general path only
    const uint32_t *c0_tab = nullptr;              // conv0 pixel table of the raw path (pack.h stem_conv0_table), nullptr = index arithmetic
    const float *dw_w; const float *dw_b;          // depthwise taps [9][8], fp32
    const half_t *pw_w; const float *pw_b;         // pointwise 16 x 8 as one A fragment with K slots [hi | hi | lo | 0] (pack.h)
    const float *pw_m = nullptr;                   // int8 output: 1 / out_scale per channel (pw_b pre-divided)
    int n, net_h, net_w;
};
template <typename TO> void launch_stem(hipStream_t s, const StemParams<TO> &p);

// ---- K_a'' (fp16 engine): K_a' fused with the first stride-2 block (conv3 depthwise + conv4 pointwise): net / 4 map, 32 channels.
struct Stem2Params {
    const FrameDesc *frames; half_t *out;          // out: [n][net_h/4][net_w/4][32]
    const half_t *w0; const float *b0;
    const half_t *w0_raw = nullptr;                // conv0 fragments of the raw-row staging (weights.h c0_raw_): [2 parities][4][64][8]
    const uint32_t *c0_tab = nullptr;              // conv0 pixel table of the raw path (pack.h stem2_conv0_table): [4][6][64] x uint2
    const uint32_t *dw1_mma4 = nullptr;            // conv3's diagonal A fragments expanded: [5][64][4] dwords (weights.h stem2_dw4_)
    const float *dw0_w; const float *dw0_b; const half_t *pw0_w; const float *pw0_b;
    const uint32_t *dw1_mma; const float *dw1_b;   // conv3: taps as diagonal MFMA A fragments [5][64] dwords (pack.h), bias [16]
    const half_t *pw1_w; const float *pw1_b;       // conv4: 32 x 16 as hi | lo along K (k < 16: rn16(w), k >= 16: rn16(w - hi)), MFMA-fragment packed, bias [32]
    const uint32_t *c2_floor, *c3_floor;           // DC-centred conv2 / conv3 tiles: -mu per channel as packed fp16 pairs, [8] dwords each (the
                                                   // biases above are pre-adjusted on the host, weights.h)
    int n, net_h, net_w;
};
void launch_stem2(hipStream_t s, const Stem2Params &p);
int stem2_variant();      // 0 = off (K_a' + a separate dwpw<16,32,s2> launch), 1 = 7x8 tiles, 2 = 7x16, 3 = 7x8 fp16 patch (probe knob RF_STEM2)

// ---- K_b: depthwise 3x3 (+BN+ReLU) -> pointwise 1x1 (+BN+ReLU), the intermediate never leaves LDS.
//      has_dw = false gives a plain 1x1 conv (+bias, +ReLU): the FPN laterals.
template <typename T>
struct DwPwParams {
    const T *in; T *out;
    const typename DwWeightT<T>::type *dw_w;        // [9][cin] (fp32 when T = int8)
    const float *dw_b;    // [cin]
    const uint32_t *dw_mma = nullptr;   // fp16 engine: the depthwise taps as per-lane dwords of DIAGONAL MFMA A fragments,
                                        // [cin/16][5][64] (dw_mma_dword in pack.h): the stencil runs on the matrix cores
    const T *pw_w;        // MFMA-fragment packed (pack.h), k = cin
    const float *pw_b;    // [cout]
    const T *lat_w = nullptr; const float *lat_b = nullptr; T *lat_out = nullptr;   // optional fused FPN lateral (cout -> 64)
    const float *pw_m = nullptr, *lat_m = nullptr;   // int8: requantisation multipliers per output channel
    const float *dw_m = nullptr;                     // int8 depthwise on MFMA: per-channel scale of the 15-bit integer taps
    int n, hin, win, hout, wout;
    int cin, cout, stride;
    bool has_dw;
};
template <typename T> void launch_dwpw(hipStream_t s, const DwPwParams<T> &p);

// ---- K_b2 (fp16 engine): two backbone blocks, 32 -> 32 stride 1 then 32 -> 64 stride 2, in one launch; the map between them stays in LDS
struct DwPw2Params {
    const half_t *in; half_t *out;                 // in [n][hin][win][32], out [n][hin/2][win/2][64]
    const uint32_t *dwa_mma; const float *dwa_b; const half_t *pwa_w; const float *pwa_b;     // block A: diagonal dw fragments (pack.h), packed pw
    const uint32_t *dwb_mma; const float *dwb_b; const half_t *pwb_w; const float *pwb_b;     // block B
    int n, hin, win;
};
void launch_dwpw2(hipStream_t s, const DwPw2Params &p);
int dwpw2_variant();     // probe knob RF_DWPW2: 0 = off

// ---- K_c: dense 3x3 p1 s1 conv as an implicit GEMM on MFMA (+bias +ReLU).  Optional fused input
//      "lateral + bilinear x2 upsample(coarser)" (Deconvolution k4 s2 p1 + Crop + Eltwise SUM,
//      prototxt :1553-1592) and an output split into two NHWC destinations (merged sibling convs).
template <typename T>
struct Conv3Params {
    const T *in; int in_ld, in_off;       // input pixel stride / channel offset (elements)
    const T *up;                          // nullptr, or coarser level [n][h/2][w/2][64]
    const T *w; const float *b;           // packed (k = 9*cin), bias[cout]
    const float *m = nullptr;             // int8: per-output-channel requantisation multiplier
    float a_lat = 1.f, a_up = 1.f;        // fused upsample+add: staged = lat * a_lat + up * a_up (int8 scale ratios)
    bool blend_fp32 = false;              // int8: blend in fp32 even where the packed-integer form applies (test knob RF_BLEND_FP32, read ONCE when the lane is built)
    T *out0; int ld0, off0, n0;           // output channels [0, n0)    -> out0[pixel*ld0 + off0 + c]
    T *out1; int ld1, off1;               // output channels [n0, cout) -> out1[pixel*ld1 + off1 + c - n0]
    int n, h, w_, cin, cout;
};
// `levels`: 1..3 parameter sets of the same (cin, cout) covered by ONE launch (the FPN levels of the SSH module)
template <typename T> void launch_conv3x3(hipStream_t s, const Conv3Params<T> *levels, int nlevels);

// ---- K_c2 (fp16 / int8 engines): the tail of the SSH context module -- conv_b (16 -> 32: context_conv2 || context_conv3_1) and conv_c
//      (16 -> 16: context_conv3_2 on context_conv3_1) -- in one launch; context_conv3_1 never leaves LDS; writes concat[32:64].
template <typename T>
struct SshTailParams {
    const T *in;                                   // context_conv1 output [n][h][w][16]
    const T *wb; const float *bb; const float *mb; // conv_b packed (k = 144), bias [32], int8 multipliers [32] or nullptr
    const T *wc; const float *bc; const float *mc; // conv_c packed, bias [16], multipliers [16] or nullptr
    T *cat;                                        // concat tensor [n][h][w][64]: channels 32..63 are written
    int n, h, w_;
};
template <typename T> void launch_ssh_tail(hipStream_t s, const SshTailParams<T> *levels, int nlevels);   // 1..3 FPN levels per launch
int ssh_tail_variant();     // probe knob RF_SSHTAIL: 0 = off (two conv3x3<16,*> launches)

// ---- K_d: the three 1x1 heads of one stride as one 64->32 GEMM + 2-class softmax + anchor decode +
//      bbox / landmark regression + clip + threshold compaction (RetinaFace.cpp:666-724, 378-432, 179-199).
template <typename T>
struct HeadParams {
    const T *in;                          // [n][h][w][64] = rf_cX_det_concat_relu
    const T *w; const float *b;           // packed 16A x 64, bias[16A]: cls [0, 2A) (background A | foreground A), bbox [2A, 6A), landmark [6A, 16A)
    const float *m = nullptr;             // int8: per-channel dequantisation multiplier (w_scale * in_scale)
    int n, h, w_, stride, anchor_offset;  // anchor_offset = global index of (a=0, iy=0, ix=0) of this stride
    int num_anchors = 2;                  // A: anchors per cell, 2 ("net3") or 4 ("net3a"); 0 = a preset without anchors: no candidates
    float base[4][4];                     // the A base anchors of this stride (RetinaFace.cpp:34-103)
    int net_h, net_w;
    const RunParams *params;
    Candidate *cand; int *cand_count; int cap;
    float *dump_prob, *dump_bbox, *dump_lmk;   // optional NCHW fp32 copies of the 3 blobs (nullptr = off)
};
template <typename T> void launch_head(hipStream_t s, const HeadParams<T> *levels, int nlevels);   // 1..3 strides per launch

// ---- K_e: per-image sort (score desc, anchor index asc) + greedy NMS (RetinaFace.cpp:434-492); one
//      workgroup per image, everything in LDS.
struct NmsParams {
    const Candidate *cand; int *cand_count; int cap;         // cap: power of two <= 4096; counter is reset to 0 at the end
    const RunParams *params;
    Candidate *out; int *out_count; int *out_cand_count;     // out[img*max_det + k]; true kept / candidate counts
    int max_det;                                             // (out* may be pinned host memory: written over PCIe)
    int n;
};
void launch_nms(hipStream_t s, const NmsParams &p);

// Area-average downscale of an over-size frame onto the net-size u8 canvas (NPPI_INTER_SUPER stand-in,
// resizeconvertion.cu:298-311; closed-source NPP semantics -> "parity unpinned").
void launch_resize_area(hipStream_t s, const FrameDesc *src, uint8_t *dst, int n, int net_h, int net_w);
// The same step in the reference's build without NPP: cv::resize bilinear (RetinaFace.cpp:611-620), OpenCV's fixed-point algorithm.
void launch_resize_bilinear(hipStream_t s, const FrameDesc *src, uint8_t *dst, int n, int net_h, int net_w);

// LDS bytes / tile geometry chosen for a layer (exposed for tests and DESIGN.md tables)
struct TileInfo { int th, tw; size_t lds_bytes; int blocks_per_image; };
template <typename T> TileInfo dwpw_tile_info(int cin, int cout, int stride, bool has_dw, int hout, int wout);
template <typename T> TileInfo conv3x3_tile_info(int cin, int cout, int h, int w);

}  // namespace rf
