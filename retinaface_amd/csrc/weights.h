// weights.h -- host-only weight packer of the engine and its on-disk cache.
//
// WeightPack<T> turns a compiled Plan (BN folded, siblings merged: plan.cpp) into the ONE contiguous byte image the kernels read
// (storage type T, MFMA A-fragment order, int8 quantisation with the calibration table) plus the table of offsets into it.
// That image is what TensorRT's serialized engine is to the reference (trtnetbase.cpp:205-243: first run builds and writes
// "<name>.cache", later runs deserialize it): save() / load() write and read "<stem>.<precision>.rfplan" next to the model --
// keyed by a hash of the model files, the precision, the packing version and the fused-stem variant -- so a warm start is
// read file -> hipMemcpy, with no parse, no BN fold and no packing.  The Plan travels in the file as a skeleton (layer names,
// shapes, strides: what build_lane needs), without weights.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "kernels.h"
#include "knobs.h"
#include "model.h"
#include "pack.h"
#include "plan.h"

namespace rf {

template <typename T> struct Cast;
template <> struct Cast<float> { static float from(float v) { return v; } static float to(float v) { return v; } };
template <> struct Cast<half_t> {
    static half_t from(float v) { return (half_t)v; }
    static float to(half_t v) { return (float)v; }
};
template <> struct Cast<int8_t> {            // values are already integers in [-127, 127] (quantised on the host)
    static int8_t from(float v) { return (int8_t)std::lrintf(v); }
    static float to(int8_t v) { return (float)v; }
};

// Device memory arena for the read-only weights: one allocation, 256-byte aligned sub-buffers.
class Arena {
public:
    size_t reserve(size_t bytes) {
        size_t off = host_.size();
        host_.resize(off + ((bytes + 255) / 256) * 256, 0);
        return off;
    }
    template <typename U> size_t put(const std::vector<U> &v) {
        size_t off = reserve(v.size() * sizeof(U));
        memcpy(host_.data() + off, v.data(), v.size() * sizeof(U));
        return off;
    }
    void upload() {            // the whole weight image: one allocation, ONE copy
        if (hipMalloc(&dev_, host_.size() ? host_.size() : 256) != hipSuccess) throw std::runtime_error("hipMalloc of the weight arena failed");
        if (hipMemcpy(dev_, host_.data(), host_.size(), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("upload of the weight arena failed");
    }
    std::vector<unsigned char> &host() { return host_; }
    const std::vector<unsigned char> &host() const { return host_; }
    template <typename U> const U *ptr(size_t off) const { return (const U *)((const char *)dev_ + off); }
    void release() { if (dev_) (void)hipFree(dev_); dev_ = nullptr; }
    size_t bytes() const { return host_.size(); }
private:
    std::vector<unsigned char> host_;
    void *dev_ = nullptr;
};

// GEMM-shaped weights [cout][k_total] -> MFMA A-fragment order (pack.h), zero padded to whole K chunks
template <typename T> std::vector<T> pack_gemm(const std::vector<float> &w, int cout, int k_total, int K, int KPL) {
    int kch = k_chunks_for(k_total, K);
    std::vector<T> out((size_t)(cout / 16) * kch * 64 * KPL, Cast<T>::from(0.f));
    for (int o = 0; o < cout; o++)
        for (int k = 0; k < k_total; k++)
            out[packed_weight_index(o, k, kch, K, KPL)] = Cast<T>::from(w[(size_t)o * k_total + k]);
    return out;
}

template <typename T> constexpr int mma_k() { return sizeof(T) == 1 ? 64 : sizeof(T) == 2 ? 32 : 4; }
template <typename T> constexpr int mma_kpl() { return sizeof(T) == 1 ? 16 : sizeof(T) == 2 ? 8 : 1; }


// ---------------------------------------------------------------------------------------------------- byte archive
constexpr uint32_t kPlanCacheVersion = 11;      // bump when the packing / layout of anything below changes

struct ArOut {
    std::string b;
    static constexpr bool kLoading = false;
    void raw(const void *p, size_t n) { b.append((const char *)p, n); }
    template <typename U> void pod(U &v) { static_assert(std::is_trivially_copyable<U>::value, ""); raw(&v, sizeof(U)); }
    void str(std::string &s) { uint32_t n = (uint32_t)s.size(); pod(n); raw(s.data(), n); }
    template <typename U> void vec(std::vector<U> &v) { uint64_t n = v.size(); pod(n); if (n) raw(v.data(), n * sizeof(U)); }
};
struct ArIn {
    const std::string &b;
    size_t p = 0;
    static constexpr bool kLoading = true;
    explicit ArIn(const std::string &buf) : b(buf) {}
    void raw(void *dst, size_t n) { if (n > b.size() - p) throw IoError("plan cache: truncated"); memcpy(dst, b.data() + p, n); p += n; }
    template <typename U> void pod(U &v) { raw(&v, sizeof(U)); }
    void str(std::string &s) { uint32_t n = 0; pod(n); if (n > b.size() - p) throw IoError("plan cache: truncated"); s.assign(b.data() + p, n); p += n; }
    template <typename U> void vec(std::vector<U> &v) {
        uint64_t n = 0; pod(n);
        if (n > (b.size() - p) / sizeof(U)) throw IoError("plan cache: truncated");
        v.resize((size_t)n);
        if (n) raw(v.data(), (size_t)n * sizeof(U));
    }
};

template <class Ar> void io_conv(Ar &ar, FoldedConv &f) {          // skeleton: everything but the weights
    ar.str(f.name); ar.str(f.out_blob);
    ar.pod(f.cout); ar.pod(f.cin); ar.pod(f.k); ar.pod(f.stride); ar.pod(f.pad); ar.pod(f.group); ar.pod(f.relu);
}
template <class Ar> void io_plan(Ar &ar, Plan &p) {
    ar.pod(p.net_h); ar.pod(p.net_w); ar.pod(p.anchors_per_cell);
    io_conv(ar, p.conv0);
    uint32_t nb = (uint32_t)p.blocks.size();
    ar.pod(nb);
    if (Ar::kLoading) { if (nb > 64) throw IoError("plan cache: bad block count"); p.blocks.resize(nb); }
    for (auto &blk : p.blocks) { io_conv(ar, blk.dw); io_conv(ar, blk.pw); }
    for (auto &f : p.lateral) io_conv(ar, f);
    for (auto &f : p.aggr) io_conv(ar, f);
    for (auto &m : p.ssh) { ar.pod(m.stride); io_conv(ar, m.conv_a); io_conv(ar, m.conv_b); io_conv(ar, m.conv_c); io_conv(ar, m.head); }
}

template <typename T>
struct WeightPack {
    static constexpr bool kInt8 = sizeof(T) == 1;
    typedef typename DwWeightT<T>::type DWT;
    static constexpr size_t kNone = (size_t)-1;
    struct GemmW { size_t w, b, m = kNone; };          // m: int8 requantisation multipliers (absent otherwise)
    struct DwW { size_t w, b, mma = 0, m = kNone; };     // m: int8 depthwise-on-MFMA tap scales

    Arena arena_;
    size_t c0_w_ = 0, c0_b_ = 0, c0_hi_ = 0, c0_b_mma_ = 0;      // c0_b_mma_: conv0 bias of the MFMA stems (offset-folded, see pack())
    size_t c0_raw_ = 0;                                          // conv0 fragments for the raw-row staging of stem2 (two column parities, see pack())
    size_t stem_c0tab_ = 0;                                      // the int8 stem's conv0 pixel table (pack.h stem_conv0_table)
    size_t stem2_dw4_ = 0, stem2_c0tab_ = 0;                     // stem2: conv3's diagonal A fragments expanded to 4 dwords per lane; conv0's pixel table (see pack())
    DwW stem_dw_{0, 0}, stem2_dw_{0, 0};
    GemmW stem_pw_{0, 0}, stem2_pw_{0, 0};
    size_t stem2_c2_b_ = 0, stem2_c2_floor_ = 0, stem2_c3_floor_ = 0;      // stem2's DC-centred tiles (pack())
    float aggr_a_lat_[2] = {1.f, 1.f}, aggr_a_up_[2] = {1.f, 1.f};
    std::map<std::string, std::vector<float>> act_scale_;    // int8: blob -> per-channel scales (debug accessors dequantise)
    std::vector<DwW> dw_w_;
    std::vector<GemmW> pw_w_;
    GemmW lat_w_[3], aggr_w_[2], ssh_w_[3][4];
    int head_a_ = 2;                                    // anchors per cell the model's heads carry


    // per-tensor activation scale of a reference blob (TensorRT calibration cache)
    float scale_of(const Plan &plan, const std::string &blob) const {
        for (const auto &kv : plan.int8_scales)
            if (kv.first == blob) return kv.second;
        throw Unsupported("int8: the calibration table has no scale for tensor '" + blob + "'");
    }
    // Per-channel activation scales (an extension of the TensorRT cache format: besides `tensor: hex` lines, which every reader
    // takes as the per-tensor scale, tools/calibrate_int8.py writes `tensor#<c>: hex` lines).  A GEMM's per-input-channel scale
    // folds into its weights and its per-output-channel scale into the requantisation multiplier (put_gemm), so per-channel
    // activations cost nothing at run time; a per-tensor table (the one the reference ships) is broadcast.
    typedef std::vector<float> Sc;
    Sc scales_of(const Plan &plan, const std::string &blob, int channels) const {
        Sc v(channels, 1.f);
        if constexpr (!kInt8) return v;
        bool per_channel = false;
        for (const auto &kv : plan.int8_scales)
            if (kv.first == blob + "#0") { per_channel = true; break; }
        if (!per_channel) { std::fill(v.begin(), v.end(), scale_of(plan, blob)); return v; }
        for (int c = 0; c < channels; c++) v[c] = scale_of(plan, blob + "#" + std::to_string(c));
        return v;
    }
    static Sc slice(const Sc &v, int lo, int hi) { return Sc(v.begin() + lo, v.begin() + hi); }
    static Sc concat(Sc a, const Sc &b) { a.insert(a.end(), b.begin(), b.end()); return a; }
    static Sc cmax(const Sc &a, const Sc &b, const Sc &c) {
        Sc v(a.size());
        for (size_t i = 0; i < a.size(); i++) v[i] = std::max(a[i], std::max(b[i], c[i]));
        return v;
    }
    const float *mult_ptr(const GemmW &g) const { return g.m == kNone ? nullptr : arena_.ptr<float>(g.m); }

    // int8, round 6: a depthwise stage's output ("mid") never leaves LDS and is read by a 1 x 1 convolution only, so it needs no zero
    // padding and can use all 8 bits: ReLU'd quanta 0..255 of HALF the calibrated quantum (amax / 255 instead of amax / 127), stored as
    // q - 128 so the signed i8 MFMA reads it unchanged; the constant 128 * sum_k w_q[o][k] goes into the pointwise bias (exact integer
    // algebra).  One bit more on the tensors that carried ~60 % of the activation-quantisation noise of the box deltas
    // (tools/probes/int8_mix_sim.py), and one VALU instruction LESS per value (v_cvt_pk_u8_f32 saturates at 255 by itself).
    static constexpr float kMidU8 = (float)(127.0 / 255.0);
    static Sc mid_u8_scale(const Sc &s) {
        Sc v(s);
        for (float &x : v) x *= kMidU8;
        return v;
    }
    const QWeights *qweights_for(const Plan &plan, const FoldedConv &f, int ktot) const {
        for (const auto &qw : plan.int8_qweights)
            if (qw.op == f.name) {
                if (qw.cout != f.cout || qw.ktot != ktot)
                    throw ModelError("int8: calibrated weights of '" + f.name + "' have the wrong shape");
                return &qw;
            }
        return nullptr;
    }
    // what put_gemm decided for one fused op (host-only hook for the calibration tool: rf_plan_int8_gemm)
    struct GemmRecord { std::vector<float> quanta, in_scale, row_scale, out_scale; int cout = 0, ktot = 0, cin = 0, in_u8 = 0; };
    std::map<std::string, GemmRecord> *record_ = nullptr;
    const Plan *plan_ = nullptr;            // the plan being packed (calibrated weights are looked up by fused-op name)

    // fp16 / fp32: weights as they are.  int8: per-output-channel symmetric weight quantisation (w_scale = amax / 127, what
    // TensorRT does with a per-tensor activation table); the epilogue computes acc * mult + bias with
    //   mult[c] = w_scale[c] * in_scale / out_scale[c],  bias[c] = b[c] / out_scale[c]     (out_scale = 1: real output)
    // Rounding: to nearest, or -- when the model carries calibrated weights for this op (Plan::int8_qweights) -- the integers the
    // calibration chose on the SAME grid, with its bias correction.  in_u8: the input tensor is a depthwise mid (see kMidU8).
    GemmW put_gemm(const FoldedConv &f, const Sc &in_scale_table = {}, const Sc &out_scale = {}, bool in_u8 = false) {
        const int cin_g = f.cin / f.group;
        const int ktot = f.k * f.k * cin_g;
        GemmW g;
        if constexpr (!kInt8) {
            g.w = arena_.put(pack_gemm<T>(f.w, f.cout, ktot, mma_k<T>(), mma_kpl<T>()));
            g.b = arena_.put(f.b);
        } else {
            const Sc in_scale = in_u8 ? mid_u8_scale(in_scale_table) : in_scale_table;
            if (!in_scale.empty() && (int)in_scale.size() != cin_g) throw ModelError("int8: input scale count does not match " + f.name);
            if (!out_scale.empty() && (int)out_scale.size() != f.cout) throw ModelError("int8: output scale count does not match " + f.name);
            const QWeights *qw = plan_ ? qweights_for(*plan_, f, ktot) : nullptr;
            std::vector<float> ws_in(f.w), q(f.w.size()), mult(f.cout), bias(f.cout), row(f.cout);
            if (!in_scale.empty())
                for (int o = 0; o < f.cout; o++)
                    for (int k = 0; k < ktot; k++) ws_in[(size_t)o * ktot + k] *= in_scale[k % cin_g];       // k = tap*cin + c
            for (int o = 0; o < f.cout; o++) {
                float amax = 0.f;
                for (int k = 0; k < ktot; k++) amax = std::max(amax, std::fabs(ws_in[(size_t)o * ktot + k]));
                const float ws = amax > 0.f ? amax / 127.f : 1.f;
                row[o] = ws;
                long qsum = 0;
                for (int k = 0; k < ktot; k++) {
                    const float rtn = std::min(127.f, std::max(-127.f, std::nearbyintf(ws_in[(size_t)o * ktot + k] / ws)));
                    const float v = qw ? (float)qw->q[(size_t)o * ktot + k] : rtn;
                    q[(size_t)o * ktot + k] = v;
                    qsum += (long)v;
                }
                const float os = out_scale.empty() ? 1.f : out_scale[o];
                mult[o] = ws / os;
                bias[o] = (qw ? f.b[o] + qw->bias_delta[o] : f.b[o]) / os;
                if (in_u8) bias[o] = std::fmaf(mult[o], (float)(128 * qsum), bias[o]);      // |128 * qsum| < 2^24: exact in fp32
            }
            if (record_) {
                GemmRecord r;
                r.cout = f.cout; r.ktot = ktot; r.cin = cin_g; r.in_u8 = in_u8 ? 1 : 0;
                r.quanta.resize(ws_in.size());
                for (int o = 0; o < f.cout; o++)
                    for (int k = 0; k < ktot; k++) r.quanta[(size_t)o * ktot + k] = ws_in[(size_t)o * ktot + k] / row[o];
                r.in_scale = in_scale.empty() ? Sc(cin_g, 1.f) : in_scale;
                r.row_scale = row;
                r.out_scale = out_scale.empty() ? Sc(f.cout, 1.f) : out_scale;
                (*record_)[f.name] = std::move(r);
            }
            g.w = arena_.put(pack_gemm<T>(q, f.cout, ktot, mma_k<T>(), mma_kpl<T>()));
            g.b = arena_.put(bias);
            g.m = arena_.put(mult);
        }
        return g;
    }

    // fp16 engine: per-channel rescaling of a depthwise + pointwise block that costs nothing at run time and takes the rounding of
    // the nine fp16 taps out of the error budget.  relu(y / t) = relu(y) / t for t > 0, so channel c of the depthwise stage may
    // produce its output divided by t[c] (taps and bias / t[c]) if column c of the pointwise matrix is multiplied by t[c].  The
    // taps' RELATIVE rounding errors depend on where they fall between fp16 grid points, i.e. on t: scanning t over one octave
    // and keeping the best value brings the rms tap error of a channel to 0.2-0.3 of plain rounding (measured on both shipped
    // models); the pointwise column is rounded afterwards as usual (different numbers, same expected error).  The taps were
    // ~24 % of what was left of the fp16 engine's box-error variance (tools/fp16_error_budget.py).
    static void equalize_depthwise(FoldedConv &dw, FoldedConv &pw) {
        const int c = dw.cout;
        if (dw.group != c || dw.k != 3 || pw.k != 1 || pw.cin != c) return;
        auto err2 = [](const float *w9, double t) {
            double e = 0.0;
            for (int k = 0; k < 9; k++) {
                const double v = (double)w9[k] / t;
                const double q = (double)(float)(half_t)(float)v * t;
                e += (q - (double)w9[k]) * (q - (double)w9[k]);
            }
            return e;
        };
        for (int ch = 0; ch < c; ch++) {
            const float *w9 = &dw.w[(size_t)ch * 9];
            double best_t = 1.0, best_e = err2(w9, 1.0);
            if (best_e == 0.0) continue;
            for (int i = 1; i < 2048; i++) {
                const double t = 1.0 + i / 2048.0, e = err2(w9, t);
                if (e < best_e) { best_e = e; best_t = t; }
            }
            const float t = (float)best_t;
            for (int k = 0; k < 9; k++) dw.w[(size_t)ch * 9 + k] = (float)((double)dw.w[(size_t)ch * 9 + k] / best_t);
            dw.b[ch] = (float)((double)dw.b[ch] / best_t);
            for (int o = 0; o < pw.cout; o++) pw.w[(size_t)o * c + ch] *= t;
        }
    }

    // depthwise weights [c][3][3][1] -> [tap][c].  int8: fp32 weights pre-scaled so the stencil maps input quanta straight to
    // output quanta: w * in_scale / mid_scale, b / mid_scale
    DwW put_dw(const FoldedConv &dw, const Sc &in_scale = {}, const Sc &mid_scale_table = {}) {
        const int c = dw.cout;
        const Sc mid_scale = kInt8 ? mid_u8_scale(mid_scale_table) : mid_scale_table;      // int8: 0..255 quanta of amax / 255 (kMidU8)
        std::vector<DWT> w((size_t)9 * c);
        std::vector<float> b(dw.b);
        for (int ch = 0; ch < c; ch++)
            for (int t = 0; t < 9; t++) {
                float v = dw.w[(size_t)ch * 9 + t];
                if constexpr (kInt8) w[(size_t)t * c + ch] = v * in_scale[ch] / mid_scale[ch];
                else w[(size_t)t * c + ch] = Cast<DWT>::from(v);
            }
        if constexpr (kInt8) for (int ch = 0; ch < c; ch++) b[ch] /= mid_scale[ch];
        DwW r{arena_.put(w), arena_.put(b), 0};
        if constexpr (std::is_same<T, half_t>::value) {
            // the same taps as diagonal MFMA A fragments (pack.h dw_mma_dword): [c/16][5][64] dwords
            if (c % 16 == 0) {
                std::vector<uint32_t> mm((size_t)(c / 16) * kDwMmaChunks * 64);
                for (int g = 0; g < c / 16; g++)
                    for (int lane = 0; lane < 64; lane++) {
                        uint16_t w9[9];
                        for (int t = 0; t < 9; t++) {
                            half_t h = w[(size_t)t * c + g * 16 + (lane & 15)];
                            std::memcpy(&w9[t], &h, 2);
                        }
                        for (int kc = 0; kc < kDwMmaChunks; kc++) mm[((size_t)g * kDwMmaChunks + kc) * 64 + lane] = dw_mma_dword(kc, lane, w9);
                    }
                r.mma = arena_.put(mm);
            }
        }
        if constexpr (kInt8) {
            // int8 engine: the taps (already in output quanta per input quantum) as 15-bit integers w = 128*hi + lo with a
            // per-channel scale, hi / lo as diagonal i8 MFMA A fragments (pack.h dw_mma_dword_i8): [c/16][3][hi, lo][64] dwords
            if (c % 16 == 0) {
                std::vector<float> ws(c, 1.f);
                std::vector<int8_t> hi((size_t)9 * c), lo((size_t)9 * c);
                for (int ch = 0; ch < c; ch++) {
                    float amax = 0.f;
                    for (int t = 0; t < 9; t++) amax = std::max(amax, std::fabs((float)w[(size_t)t * c + ch]));
                    ws[ch] = amax > 0.f ? amax / (float)kDwI8Range : 1.f;
                    for (int t = 0; t < 9; t++) {
                        const int wi = (int)std::lrintf((float)w[(size_t)t * c + ch] / ws[ch]);
                        dw_i8_split(std::max(-kDwI8Range, std::min(kDwI8Range, wi)), &hi[(size_t)t * c + ch], &lo[(size_t)t * c + ch]);
                    }
                }
                std::vector<uint32_t> mm((size_t)(c / 16) * kDwMmaChunksI8 * 2 * 64);
                for (int g = 0; g < c / 16; g++)
                    for (int lane = 0; lane < 64; lane++) {
                        int8_t h9[9], l9[9];
                        for (int t = 0; t < 9; t++) { h9[t] = hi[(size_t)t * c + g * 16 + (lane & 15)]; l9[t] = lo[(size_t)t * c + g * 16 + (lane & 15)]; }
                        for (int kc = 0; kc < kDwMmaChunksI8; kc++) {
                            mm[(((size_t)g * kDwMmaChunksI8 + kc) * 2 + 0) * 64 + lane] = dw_mma_dword_i8(kc, lane, h9);
                            mm[(((size_t)g * kDwMmaChunksI8 + kc) * 2 + 1) * 64 + lane] = dw_mma_dword_i8(kc, lane, l9);
                        }
                    }
                r.mma = arena_.put(mm);
                r.m = arena_.put(ws);
            }
        }
        return r;
    }

    // Fold-independent part of engine start-up: BN-folded weights -> storage type -> MFMA fragment order, one contiguous image.
    void pack(const Plan &plan) {
        plan_ = &plan;
        if constexpr (kInt8)
            if (plan.int8_scales.empty()) throw Unsupported("int8 precision needs a calibration table (<stem>.table.int8)");
        c0_w_ = arena_.put(plan.conv0.w);
        c0_b_ = arena_.put(plan.conv0.b);
        size_t first_block = 0;
        if constexpr (sizeof(T) <= 2) {
            // stem kernel (fp16 and int8 engines): conv0 as 16 x 64 A fragments, K = 4*(3*ky + kx) + c4 with c4 = B, G, R, pad
            // (the frame's own byte order: net channel c is frame channel 2-c), fp16 hi + lo so the sum carries ~22 mantissa
            // bits.  Layout: [hi k<32 | lo k<32 | hi k>=32 | lo k>=32][lane 64][8]
            std::vector<half_t> frag(4 * 64 * 8, (half_t)0);
            for (int half = 0; half < 2; half++)
                for (int lane = 0; lane < 64; lane++)
                    for (int el = 0; el < 8; el++) {
                        int row = lane & 15, k = half * 32 + (lane >> 4) * 8 + el;
                        int tap = k / 4, c4 = k % 4;
                        if (row >= 8 || tap >= 9 || c4 == 3) continue;
                        float w = plan.conv0.w[(size_t)row * 27 + tap * 3 + (2 - c4)];
                        half_t h = (half_t)w;
                        frag[((half * 2 + 0) * 64 + lane) * 8 + el] = h;
                        frag[((half * 2 + 1) * 64 + lane) * 8 + el] = (half_t)(w - (float)h);
                    }
            c0_hi_ = arena_.put(frag);
            // The same weights for stem2's RAW staging (round 6): the patch rows sit in LDS as the frame's own bytes (3 per pixel, brought in
            // by LDS-DMA), and a conv0 pixel's three input pixels of one row are 9 consecutive bytes inside an ALIGNED 12-byte window --
            // bytes 1..9 of it when the conv0 column is even, 3..11 when it is odd (the patch starts at byte 1 of its first dword and a conv0
            // column advances 6 bytes).  K = 16 * ky + j, j = byte of the window (12..15: the next dword, zero weights), one fragment set per
            // parity: [parity][hi k<32 | lo k<32 | hi k>=32 | lo k>=32][lane 64][8].  Same hi / lo split, same 27 products per output.
            std::vector<half_t> raw(2 * 4 * 64 * 8, (half_t)0);
            for (int par = 0; par < 2; par++)
                for (int half = 0; half < 2; half++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int el = 0; el < 8; el++) {
                            const int row = lane & 15, k = half * 32 + (lane >> 4) * 8 + el;
                            const int ky = k / 16, j = k % 16, b = j - (par ? 3 : 1);      // b = 3 * kx + frame channel
                            if (row >= 8 || ky >= 3 || b < 0 || b >= 9) continue;
                            const float w = plan.conv0.w[(size_t)row * 27 + (ky * 3 + b / 3) * 3 + (2 - b % 3)];
                            const half_t h = (half_t)w;
                            raw[(((size_t)par * 4 + half * 2 + 0) * 64 + lane) * 8 + el] = h;
                            raw[(((size_t)par * 4 + half * 2 + 1) * 64 + lane) * 8 + el] = (half_t)(w - (float)h);
                        }
            c0_raw_ = arena_.put(raw);
            stem_c0tab_ = arena_.put(stem_conv0_table());
            // the stems feed 1024 + pixel into those fragments (kernels.hip u8x4_to_f16): bias - 1024 * sum of the hi + lo weights
            {
                std::vector<float> bm(plan.conv0.b);
                for (int row = 0; row < 8; row++) {
                    double sw = 0.0;
                    for (int tap = 0; tap < 9; tap++)
                        for (int c = 0; c < 3; c++) {
                            const float w = plan.conv0.w[(size_t)row * 27 + tap * 3 + c];
                            const half_t h = (half_t)w;
                            sw += (double)(float)h + (double)(float)(half_t)(w - (float)h);
                        }
                    bm[row] = (float)((double)plan.conv0.b[row] - 1024.0 * sw);
                }
                c0_b_mma_ = arena_.put(bm);
            }
            // the stem computes its depthwise + pointwise block in fp16 whatever the storage type of its OUTPUT
            const auto &b0 = plan.blocks[0];
            std::vector<float> dw((size_t)9 * 8);                  // taps stay fp32 (see the stem kernel's header)
            for (int ch = 0; ch < 8; ch++)
                for (int t = 0; t < 9; t++) dw[(size_t)t * 8 + ch] = b0.dw.w[(size_t)ch * 9 + t];
            stem_dw_ = DwW{arena_.put(dw), arena_.put(b0.dw.b)};
            // pointwise 16 x 8: one A fragment whose K slots are [w_hi | w_hi | w_lo | 0] (pack.h stem_pw_slot)
            // int8 engine: the stem's output is in quanta of its calibrated scale.  The 1 / scale is folded into the (hi + lo) weights and
            // the bias here, so the kernel's accumulator IS the value to round: no multiply in the epilogue (round 3; it was fma(acc, 1 / s, b / s))
            Sc stem_os(b0.pw.cout, 1.f);
            if constexpr (kInt8) stem_os = scales_of(plan, b0.pw.out_blob, b0.pw.cout);
            std::vector<half_t> pwf((size_t)64 * 8, (half_t)0);
            for (int lane = 0; lane < 64; lane++) {
                int row = 0, use_lo = 0;
                if (!stem_pw_slot(lane, &row, &use_lo)) continue;
                for (int e = 0; e < 8; e++) {
                    const float w = kInt8 ? b0.pw.w[(size_t)row * 8 + e] / stem_os[row] : b0.pw.w[(size_t)row * 8 + e];
                    const half_t hi = (half_t)w;
                    pwf[(size_t)lane * 8 + e] = use_lo ? (half_t)(w - (float)hi) : hi;
                }
            }
            stem_pw_.w = arena_.put(pwf);
            if constexpr (kInt8) {
                std::vector<float> b(b0.pw.b);
                for (int o = 0; o < b0.pw.cout; o++) b[o] /= stem_os[o];
                stem_pw_.b = arena_.put(b);
            } else {
                stem_pw_.b = arena_.put(b0.pw.b);
            }
            first_block = 1;
            dw_w_.push_back(DwW{0, 0});
            pw_w_.push_back(GemmW{0, 0});
            if constexpr (std::is_same<T, half_t>::value) {
                if (stem2_variant()) {
                    // stem2 also runs the first stride-2 block: conv3 taps as diagonal MFMA fragments (equalised per channel like every
                    // other depthwise stage, see equalize_depthwise), conv4 as a standard packed 32 x 16 GEMM
                    FoldedConv dwq = plan.blocks[1].dw, pwq = plan.blocks[1].pw;
                    equalize_depthwise(dwq, pwq);
                    // DC-centred LDS tiles (stem2_kernel phases 4-6).  mu = the layers' response to a flat mid-grey frame (every
                    // pixel 128: the interior of every map is then one constant per channel), rounded to fp16 so that the shift
                    // itself is exact.  Shifting a tensor by a per-channel constant is exact algebra as long as its consumer's bias
                    // takes the constant back: conv3 (depthwise, taps t): + mu2 * sum(t); conv4 (1x1, W): + W mu3.  Both sums use
                    // the weights AS THE KERNEL SEES THEM (fp16-rounded taps, hi + lo pointwise weights).  RF_STEM2_DC=0: mu = 0
                    // (probe / test knob: the tiles are then plain ReLU outputs as in round 2).
                    std::vector<float> mu2(16, 0.f), mu3(16, 0.f);
                    if (knob(K_STEM2_DC) != 0) {
                        double y0[8], y1[8];
                        for (int c = 0; c < 8; c++) {
                            double sw = 0.0;
                            for (int k = 0; k < 27; k++) sw += plan.conv0.w[(size_t)c * 27 + k];
                            y0[c] = std::max(0.0, (double)plan.conv0.b[c] + 128.0 * sw);
                            double st = 0.0;
                            for (int t = 0; t < 9; t++) st += b0.dw.w[(size_t)c * 9 + t];
                            y1[c] = std::max(0.0, (double)b0.dw.b[c] + y0[c] * st);
                        }
                        for (int o = 0; o < 16; o++) {
                            double y2 = b0.pw.b[o];
                            for (int c = 0; c < 8; c++) y2 += (double)b0.pw.w[(size_t)o * 8 + c] * y1[c];
                            mu2[o] = (float)(half_t)(float)std::min(30000.0, std::max(0.0, y2));       // (far inside the fp16 range)
                            double st = 0.0;
                            for (int t = 0; t < 9; t++) st += dwq.w[(size_t)o * 9 + t];
                            mu3[o] = (float)(half_t)(float)std::min(30000.0, std::max(0.0, (double)dwq.b[o] + (double)mu2[o] * st));
                        }
                    }
                    auto packed_floor = [&](const std::vector<float> &mu) {
                        std::vector<uint32_t> f(8);
                        for (int i = 0; i < 8; i++) {
                            half_t lo = (half_t)(mu[2 * i] == 0.f ? 0.f : -mu[2 * i]), hi = (half_t)(mu[2 * i + 1] == 0.f ? 0.f : -mu[2 * i + 1]);
                            uint16_t bl, bh;
                            std::memcpy(&bl, &lo, 2); std::memcpy(&bh, &hi, 2);
                            f[i] = (uint32_t)bl | ((uint32_t)bh << 16);
                        }
                        return f;
                    };
                    {   // conv2: bias - mu2 (its own copy: the un-fused stem kernel shares stem_pw_ and stores plain values)
                        std::vector<float> b2(b0.pw.b);
                        for (int o = 0; o < 16; o++) b2[o] -= mu2[o];
                        stem2_c2_b_ = arena_.put(b2);
                        stem2_c2_floor_ = arena_.put(packed_floor(mu2));
                        stem2_c3_floor_ = arena_.put(packed_floor(mu3));
                    }
                    for (int c = 0; c < 16; c++) {          // conv3: + mu2 * sum(fp16 taps) - mu3
                        double st = 0.0;
                        for (int t = 0; t < 9; t++) st += (double)(float)(half_t)dwq.w[(size_t)c * 9 + t];
                        dwq.b[c] = (float)((double)dwq.b[c] + (double)mu2[c] * st - (double)mu3[c]);
                    }
                    stem2_dw_ = put_dw(dwq);
                    {   // round 6: the same diagonal A fragments EXPANDED to the four dwords a lane feeds the MFMA -- [5][64][4] dwords, one 16-byte load
                        // per chunk instead of one dword + four v_cndmask (20 VALU instructions per wave of stem2's conv3 phase)
                        const uint32_t *mm = (const uint32_t *)(arena_.host().data() + stem2_dw_.mma);
                        std::vector<uint32_t> ex((size_t)kDwMmaChunks * 64 * 4, 0u);
                        for (int kc = 0; kc < kDwMmaChunks; kc++)
                            for (int lane = 0; lane < 64; lane++) ex[((size_t)kc * 64 + lane) * 4 + dw_mma_dword_index(lane)] = mm[kc * 64 + lane];
                        stem2_dw4_ = arena_.put(ex);
                    }
                    stem2_c0tab_ = arena_.put(stem2_conv0_table());
                    // conv4: K = 16 of the MFMA's 32 slots -> the weights ride as hi | lo along K; bias + (hi + lo) mu3
                    std::vector<float> w2((size_t)pwq.cout * 32, 0.f), b4(pwq.b);
                    for (int o = 0; o < pwq.cout; o++) {
                        double corr = 0.0;
                        for (int k = 0; k < 16; k++) {
                            const float w = pwq.w[(size_t)o * 16 + k];
                            const float hi = (float)(half_t)w, lo = (float)(half_t)(w - hi);
                            w2[(size_t)o * 32 + k] = hi;
                            w2[(size_t)o * 32 + 16 + k] = lo;
                            corr += ((double)hi + (double)lo) * (double)mu3[k];
                        }
                        b4[o] = (float)((double)b4[o] + corr);
                    }
                    stem2_pw_.w = arena_.put(pack_gemm<half_t>(w2, pwq.cout, 32, 32, 8));
                    stem2_pw_.b = arena_.put(b4);
                    first_block = 2;
                    dw_w_.push_back(DwW{0, 0});
                    pw_w_.push_back(GemmW{0, 0});
                }
            }
        }
        Sc s_prev = scales_of(plan, plan.blocks[0].pw.out_blob, plan.blocks[0].pw.cout);
        Sc s_tap[3];                                  // scales of the block outputs the laterals tap (blocks 12, 10, 4)
        for (size_t i = first_block; i < plan.blocks.size(); i++) {
            const auto &blk = plan.blocks[i];
            const Sc s_mid = scales_of(plan, blk.dw.out_blob, blk.dw.cout), s_out = scales_of(plan, blk.pw.out_blob, blk.pw.cout);
            if constexpr (std::is_same<T, half_t>::value) {
                FoldedConv dwq = blk.dw, pwq = blk.pw;
                equalize_depthwise(dwq, pwq);
                dw_w_.push_back(put_dw(dwq, s_prev, s_mid));
                pw_w_.push_back(put_gemm(pwq, s_mid, s_out));
            } else {
                dw_w_.push_back(put_dw(blk.dw, s_prev, s_mid));
                pw_w_.push_back(put_gemm(blk.pw, s_mid, s_out, kInt8));
            }
            s_prev = s_out;
            if constexpr (kInt8) act_scale_[blk.pw.out_blob] = s_out;
            if (i == 12) s_tap[0] = s_out;
            if (i == 10) s_tap[1] = s_out;
            if (i == 4) s_tap[2] = s_out;
        }
        if constexpr (kInt8) act_scale_[plan.blocks[0].pw.out_blob] = scales_of(plan, plan.blocks[0].pw.out_blob, plan.blocks[0].pw.cout);
        // FPN.  The fused "lateral + upsample(coarser)" staging adds two int8 tensors and requantises to the `_plus` scale:
        //   q_plus = round(q_lat * s_lat / s_plus + blend(q_up) * s_up / s_plus).
        // With a per-tensor table the two ratios are scalars (a_lat, a_up).  With per-channel scales the three tensors of each
        // add are given ONE common per-channel scale (the largest of their calibrated ones, as for concat inputs), so the ratios
        // are 1 and the kernel needs no per-channel multipliers.
        Sc s_lat[3], s_feat[3], s_plus[2], s_aggr[2];
        bool per_channel = false;
        if constexpr (kInt8)
            for (const auto &kv : plan.int8_scales) per_channel = per_channel || kv.first == "_plus0#0";
        for (int i = 0; i < 3; i++) s_lat[i] = scales_of(plan, plan.lateral[i].out_blob, 64);
        for (int i = 0; i < 2; i++) {
            s_plus[i] = scales_of(plan, i == 0 ? "_plus0" : "_plus1", 64);
            s_aggr[i] = scales_of(plan, plan.aggr[i].out_blob, 64);
        }
        if (per_channel) {
            // add 0: {c3 lateral (= P3), c2 lateral, _plus0};  add 1: {c2 aggr (= P2), c1 lateral, _plus1}
            s_lat[0] = s_lat[1] = s_plus[0] = cmax(s_lat[0], s_lat[1], s_plus[0]);
            s_aggr[0] = s_lat[2] = s_plus[1] = cmax(s_aggr[0], s_lat[2], s_plus[1]);
        }
        for (int i = 0; i < 3; i++) lat_w_[i] = put_gemm(plan.lateral[i], s_tap[i], s_lat[i]);
        s_feat[0] = s_lat[0];
        for (int i = 0; i < 2; i++) {
            if constexpr (kInt8) {
                aggr_a_lat_[i] = per_channel ? 1.f : s_lat[i + 1][0] / s_plus[i][0];
                aggr_a_up_[i] = per_channel ? 1.f : s_feat[i][0] / s_plus[i][0];
            }
            aggr_w_[i] = put_gemm(plan.aggr[i], s_plus[i], s_aggr[i]);
            s_feat[i + 1] = s_aggr[i];
        }
        for (int i = 0; i < 3; i++) {
            const SshModule &m = plan.ssh[i];
            std::string pre = "rf_c" + std::to_string(3 - i) + "_det_";
            // the three concat inputs are quantised with the concat tensor's scales (per tensor: one shared scale, as in the
            // TensorRT table; per channel: each branch writes its slice with that slice's scales)
            const Sc s_cat = scales_of(plan, pre + "concat_relu", 64);
            const Sc s_c1 = scales_of(plan, pre + "context_conv1_relu", 16), s_c31 = scales_of(plan, pre + "context_conv3_1_relu", 16);
            ssh_w_[i][0] = put_gemm(m.conv_a, s_feat[i], concat(slice(s_cat, 0, 32), s_c1));
            ssh_w_[i][1] = put_gemm(m.conv_b, s_c1, concat(slice(s_cat, 32, 48), s_c31));
            ssh_w_[i][2] = put_gemm(m.conv_c, s_c31, slice(s_cat, 48, 64));
            if constexpr (std::is_same<T, half_t>::value) {
                // fp16 engine: head weights as an fp16 hi + lo pair along K (k < 64: rn16(w), k >= 64: rn16(w - hi); the kernel reads
                // the same 64 activations for both halves).  Box deltas are what the IoU bound is measured on, the heads are
                // 0.1 % of the MACs, and their weight rounding was ~7 % of the remaining box-error variance.
                FoldedConv h2 = m.head;
                const int ci = m.head.cin;
                h2.cin = 2 * ci;
                h2.w.assign((size_t)h2.cout * 2 * ci, 0.f);
                for (int o = 0; o < h2.cout; o++)
                    for (int k = 0; k < ci; k++) {
                        const float w = m.head.w[(size_t)o * ci + k];
                        const float hi = (float)(half_t)w;
                        h2.w[(size_t)o * 2 * ci + k] = hi;
                        h2.w[(size_t)o * 2 * ci + ci + k] = w - hi;
                    }
                ssh_w_[i][3] = put_gemm(h2);
            } else {
                ssh_w_[i][3] = put_gemm(m.head, s_cat, {});          // heads are dequantised to real logits / deltas
            }
            if constexpr (kInt8) {
                act_scale_[pre + "concat_relu"] = s_cat;
                act_scale_[pre + "context_conv1_relu"] = s_c1;
                act_scale_[pre + "context_conv3_1_relu"] = s_c31;
            }
        }
        if constexpr (kInt8) {
            for (int i = 0; i < 3; i++) act_scale_[plan.lateral[i].out_blob] = s_lat[i];
            for (int i = 0; i < 2; i++) act_scale_[plan.aggr[i].out_blob] = s_feat[i + 1];
        }
        head_a_ = plan.anchors_per_cell;
        plan_ = nullptr;
    }


    // ------------------------------------------------------------------------------------------ (de)serialisation
    template <class Ar> void io(Ar &ar) {
        ar.pod(c0_w_); ar.pod(c0_b_); ar.pod(c0_hi_); ar.pod(c0_b_mma_); ar.pod(c0_raw_); ar.pod(stem_c0tab_); ar.pod(stem2_dw4_); ar.pod(stem2_c0tab_);
        ar.pod(stem_dw_); ar.pod(stem2_dw_); ar.pod(stem_pw_); ar.pod(stem2_pw_);
        ar.pod(stem2_c2_b_); ar.pod(stem2_c2_floor_); ar.pod(stem2_c3_floor_);
        ar.pod(aggr_a_lat_); ar.pod(aggr_a_up_); ar.pod(head_a_);
        ar.vec(dw_w_); ar.vec(pw_w_);
        ar.pod(lat_w_); ar.pod(aggr_w_); ar.pod(ssh_w_);
        uint32_t ns = (uint32_t)act_scale_.size();
        ar.pod(ns);
        if (Ar::kLoading) {
            act_scale_.clear();
            for (uint32_t i = 0; i < ns; i++) { std::string k; std::vector<float> v; ar.str(k); ar.vec(v); act_scale_[k] = std::move(v); }
        } else {
            for (auto &kv : act_scale_) { std::string k = kv.first; ar.str(k); ar.vec(kv.second); }
        }
        ar.vec(arena_.host());
    }

    // every offset a launch dereferences must lie inside the image (a cache file whose checksum matches was written by this code,
    // so this only ever fires on a logic error -- but it turns an out-of-bounds device read into a rebuild)
    bool offsets_in_bounds() const {
        const size_t n = arena_.bytes();
        auto ok = [n](size_t off) { return off == kNone || off < n; };
        auto okg = [&](const GemmW &g) { return ok(g.w) && ok(g.b) && ok(g.m); };
        auto okd = [&](const DwW &d) { return ok(d.w) && ok(d.b) && ok(d.mma) && ok(d.m); };
        bool good = ok(c0_w_) && ok(c0_b_) && ok(c0_hi_) && ok(c0_b_mma_) && ok(c0_raw_) && ok(stem_c0tab_) && ok(stem2_dw4_) && ok(stem2_c0tab_) && okd(stem_dw_) && okd(stem2_dw_) && okg(stem_pw_) && okg(stem2_pw_) &&
                    ok(stem2_c2_b_) && ok(stem2_c2_floor_) && ok(stem2_c3_floor_);
        for (const auto &d : dw_w_) good = good && okd(d);
        for (const auto &g : pw_w_) good = good && okg(g);
        for (const auto &g : lat_w_) good = good && okg(g);
        for (const auto &g : aggr_w_) good = good && okg(g);
        for (const auto &lv : ssh_w_) for (const auto &g : lv) good = good && okg(g);
        return good && dw_w_.size() == pw_w_.size();
    }
};

inline uint64_t fnv1a64(const char *p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
    return h;
}

// what a cache file is valid for
struct PlanCacheKey {
    uint64_t source_hash = 0;          // FNV-1a of the model files' bytes (model.cpp model_source_hash)
    uint64_t build = 0;                // fingerprint of the library build that packed it: a rebuilt library never trusts an old image
    int32_t precision = 0, stem2 = 0;
};

// <model_dir>/<stem>.<fp32|fp16|int8>.rfplan
std::string plan_cache_path(const std::string &model_dir, const std::string &stem, int precision);

template <typename T> std::string save_plan_cache(const PlanCacheKey &key, Plan plan, WeightPack<T> &wp) {
    ArOut ar;
    ar.raw("RFP1", 4);
    uint32_t ver = kPlanCacheVersion;
    ar.pod(ver);
    PlanCacheKey k = key;
    ar.pod(k);
    ArOut body;
    io_plan(body, plan);
    wp.io(body);
    uint64_t len = body.b.size(), sum = fnv1a64(body.b.data(), body.b.size());     // payload length + checksum: a torn or bit-rotted file is rebuilt
    ar.pod(len);
    ar.pod(sum);
    ar.raw(body.b.data(), body.b.size());
    return ar.b;
}
// false = not a cache for this key (stale / other precision / other version): rebuild.  Throws IoError on a damaged file.
template <typename T> bool load_plan_cache(const std::string &bytes, const PlanCacheKey &key, Plan *plan, WeightPack<T> *wp) {
    if (bytes.size() < 8 || bytes.compare(0, 4, "RFP1") != 0) return false;
    ArIn ar(bytes);
    ar.p = 4;
    uint32_t ver = 0;
    ar.pod(ver);
    if (ver != kPlanCacheVersion) return false;
    PlanCacheKey k;
    ar.pod(k);
    if (k.source_hash != key.source_hash || k.build != key.build || k.precision != key.precision || k.stem2 != key.stem2) return false;
    uint64_t len = 0, sum = 0;
    ar.pod(len);
    ar.pod(sum);
    if (len != bytes.size() - ar.p) throw IoError("plan cache: payload length mismatch (torn write?)");
    if (fnv1a64(bytes.data() + ar.p, (size_t)len) != sum) throw IoError("plan cache: checksum mismatch");
    io_plan(ar, *plan);
    wp->io(ar);
    if (ar.p != bytes.size()) throw IoError("plan cache: trailing bytes");
    if (!wp->offsets_in_bounds()) throw IoError("plan cache: weight offset outside the image");
    return true;
}

}  // namespace rf
