// plan.cpp -- see plan.h.
#include "plan.h"

#include <cmath>
#include <new>
#include <stdexcept>

namespace rf {
namespace {

void expect(bool cond, const std::string &msg) {
    if (!cond) throw ModelError("unsupported graph: " + msg);
}

// first layer of `type` whose bottom[0] is `blob` (in-place layers keep the blob name)
const Layer *consumer(const Model &m, const std::string &blob, const std::string &type, const Layer *after) {
    bool seen = after == nullptr;
    for (const auto &l : m.layers) {
        if (!seen) { if (&l == after) seen = true; continue; }
        if (l.type == type && !l.bottoms.empty() && l.bottoms[0] == blob) return &l;
    }
    return nullptr;
}

// Convolution [-> BatchNorm -> Scale] [-> ReLU], folded.  BN (use_global_stats):
//   y = gamma * (x - mean/sf) / sqrt(var/sf + eps) + beta      (fp64 here, cast to fp32 once)
FoldedConv fold_conv(const Model &m, const std::string &conv_name, bool allow_relu = true) {
    const Layer &c = m.get(conv_name);
    expect(c.type == "Convolution", conv_name + " is not a Convolution");
    expect(!c.blobs.empty(), conv_name + " has no weights (caffemodel not attached?)");
    const Blob &wb = c.blobs[0];
    expect(wb.dims.size() == 4 && wb.dims[0] == c.num_output && wb.dims[2] == c.kernel &&
               wb.dims[3] == c.kernel, conv_name + ": weight blob shape does not match convolution_param");
    FoldedConv f;
    f.name = conv_name;
    f.cout = c.num_output;
    f.k = c.kernel;
    f.stride = c.stride;
    f.pad = c.pad;
    f.group = c.group;
    const int cin_g = wb.dims[1];
    f.cin = cin_g * c.group;
    std::vector<double> scale(f.cout, 1.0), shift(f.cout, 0.0);
    if (c.bias_term) {
        expect(c.blobs.size() >= 2 && (int)c.blobs[1].count() == f.cout, conv_name + ": bias blob missing");
        for (int o = 0; o < f.cout; o++) shift[o] = c.blobs[1].data[o];
    }
    std::string top = c.tops.at(0);
    const Layer *bn = consumer(m, top, "BatchNorm", &c);
    const Layer *last = &c;
    if (bn) {
        expect(bn->blobs.size() >= 3 && (int)bn->blobs[0].count() == f.cout, bn->name + ": BatchNorm blobs missing");
        const Layer *sc = consumer(m, bn->tops.at(0), "Scale", bn);
        expect(sc != nullptr, bn->name + " is not followed by a Scale layer");
        expect(sc->blobs.size() >= 1 && (int)sc->blobs[0].count() == f.cout, sc->name + ": Scale blobs missing");
        float sf = bn->blobs[2].data[0];
        float inv = sf == 0.f ? 0.f : 1.f / sf;
        for (int o = 0; o < f.cout; o++) {
            double mean = (double)(bn->blobs[0].data[o] * inv);
            double var = (double)(bn->blobs[1].data[o] * inv);
            double gamma = sc->blobs[0].data[o];
            double beta = (sc->scale_bias && sc->blobs.size() >= 2) ? sc->blobs[1].data[o] : 0.0;
            double kk = gamma / std::sqrt(var + (double)bn->eps);
            scale[o] = kk;
            shift[o] = (shift[o] - mean) * kk + beta;
        }
        top = sc->tops.at(0);
        last = sc;
    }
    if (allow_relu) {
        const Layer *relu = consumer(m, top, "ReLU", last);
        if (relu) { f.relu = true; top = relu->tops.at(0); }
    }
    f.out_blob = top;
    f.w.resize((size_t)f.cout * f.k * f.k * cin_g);
    f.b.resize(f.cout);
    for (int o = 0; o < f.cout; o++) {
        f.b[o] = (float)shift[o];
        for (int i = 0; i < cin_g; i++)
            for (int y = 0; y < f.k; y++)
                for (int x = 0; x < f.k; x++)
                    f.w[(((size_t)o * f.k + y) * f.k + x) * cin_g + i] =
                        (float)((double)wb.data[(((size_t)o * cin_g + i) * f.k + y) * f.k + x] * scale[o]);
    }
    return f;
}

// concatenate sibling convs (same input, same geometry) along the output-channel axis
FoldedConv merge(const std::vector<FoldedConv> &parts) {
    FoldedConv r = parts[0];
    for (size_t i = 1; i < parts.size(); i++) {
        const FoldedConv &p = parts[i];
        expect(p.cin == r.cin && p.k == r.k && p.stride == r.stride && p.pad == r.pad && p.group == r.group,
               "sibling convolutions " + r.name + " / " + p.name + " differ in geometry");
        r.name += "+" + p.name;
        r.cout += p.cout;
        r.w.insert(r.w.end(), p.w.begin(), p.w.end());
        r.b.insert(r.b.end(), p.b.begin(), p.b.end());
    }
    return r;
}

void check_bilinear_upsample(const Model &m, const std::string &name, const std::string &expect_bottom) {
    const Layer &d = m.get(name);
    expect(d.type == "Deconvolution" && d.kernel == 4 && d.stride == 2 && d.pad == 1 && d.group == 64 &&
               d.num_output == 64 && !d.bias_term, name + " is not the k4 s2 p1 depthwise upsampling deconvolution");
    expect(!d.bottoms.empty() && d.bottoms[0] == expect_bottom, name + " does not consume " + expect_bottom);
    expect(!d.blobs.empty() && d.blobs[0].count() == 64 * 16, name + ": weight blob missing");
    static const float k1[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    for (int c = 0; c < 64; c++)
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++)
                expect(std::fabs(d.blobs[0].data[(c * 4 + y) * 4 + x] - k1[y] * k1[x]) < 1e-6f,
                       name + ": weights are not the fixed bilinear kernel the fused upsample assumes");
}

}  // namespace

static Plan compile_plan_checked(const Model &m);
// (the walk indexes bottoms / tops with .at(): a layer that lacks the blob the topology needs is a ModelError, not a std::out_of_range)
Plan compile_plan(const Model &m) {
    try { return compile_plan_checked(m); }
    catch (const std::out_of_range &) { throw ModelError("graph: a layer of the expected topology has no bottom / top / blob where the walk needs one"); }
    catch (const std::length_error &) { throw ModelError("graph: a layer's dimensions do not describe a tensor that fits memory"); }
    catch (const std::bad_alloc &) { throw ModelError("graph: a layer's dimensions do not describe a tensor that fits memory"); }
}
static Plan compile_plan_checked(const Model &m) {
    Plan p;
    expect(m.input_shape[1] == 3, "network input must have 3 channels");
    p.net_h = m.input_shape[2];
    p.net_w = m.input_shape[3];

    // ---- backbone: conv0, then 13 x (depthwise 3x3, pointwise 1x1), every conv + BN + ReLU ----
    p.conv0 = fold_conv(m, "mobilenet0_conv0_fwd");
    expect(p.conv0.cin == 3 && p.conv0.cout == 8 && p.conv0.k == 3 && p.conv0.stride == 2 && p.conv0.pad == 1 &&
               p.conv0.group == 1 && p.conv0.relu, "mobilenet0_conv0_fwd is not 3x3 s2 p1 3->8 + BN + ReLU");
    expect(m.get("mobilenet0_conv0_fwd").bottoms.at(0) == m.input_name, "conv0 does not read the input blob");
    std::string prev = p.conv0.out_blob;
    int c = 8;
    static const int couts[13] = {16, 32, 32, 64, 64, 128, 128, 128, 128, 128, 128, 256, 256};
    static const int strides[13] = {1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1};
    for (int i = 0; i < 13; i++) {
        Plan::DwPw blk;
        std::string dn = "mobilenet0_conv" + std::to_string(2 * i + 1) + "_fwd";
        std::string pn = "mobilenet0_conv" + std::to_string(2 * i + 2) + "_fwd";
        blk.dw = fold_conv(m, dn);
        blk.pw = fold_conv(m, pn);
        expect(m.get(dn).bottoms.at(0) == prev, dn + " does not consume " + prev);
        expect(blk.dw.k == 3 && blk.dw.pad == 1 && blk.dw.group == c && blk.dw.cout == c && blk.dw.cin == c &&
                   blk.dw.stride == strides[i] && blk.dw.relu, dn + " is not the expected depthwise 3x3");
        expect(m.get(pn).bottoms.at(0) == blk.dw.out_blob, pn + " does not consume " + blk.dw.out_blob);
        expect(blk.pw.k == 1 && blk.pw.pad == 0 && blk.pw.group == 1 && blk.pw.stride == 1 && blk.pw.cin == c &&
                   blk.pw.cout == couts[i] && blk.pw.relu, pn + " is not the expected pointwise 1x1");
        c = couts[i];
        prev = blk.pw.out_blob;
        p.blocks.push_back(std::move(blk));
    }
    const std::string c1 = p.blocks[4].pw.out_blob;    // relu10, stride 8
    const std::string c2 = p.blocks[10].pw.out_blob;   // relu22, stride 16
    const std::string c3 = p.blocks[12].pw.out_blob;   // relu26, stride 32

    // ---- FPN ----
    p.lateral[0] = fold_conv(m, "rf_c3_lateral");
    p.lateral[1] = fold_conv(m, "rf_c2_lateral");
    p.lateral[2] = fold_conv(m, "rf_c1_red_conv");
    const std::string taps[3] = {c3, c2, c1};
    const int tap_c[3] = {256, 128, 64};
    const char *lat_names[3] = {"rf_c3_lateral", "rf_c2_lateral", "rf_c1_red_conv"};
    for (int i = 0; i < 3; i++) {
        const FoldedConv &l = p.lateral[i];
        expect(m.get(lat_names[i]).bottoms.at(0) == taps[i], std::string(lat_names[i]) + " does not consume " + taps[i]);
        expect(l.k == 1 && l.cin == tap_c[i] && l.cout == 64 && l.relu && l.group == 1 && l.stride == 1,
               std::string(lat_names[i]) + " is not a 1x1 -> 64 + BN + ReLU");
    }
    check_bilinear_upsample(m, "rf_c3_upsampling", p.lateral[0].out_blob);
    p.aggr[0] = fold_conv(m, "rf_c2_aggr");
    check_bilinear_upsample(m, "rf_c2_upsampling", p.aggr[0].out_blob);
    p.aggr[1] = fold_conv(m, "rf_c1_aggr");
    const char *plus_names[2] = {"_plus0", "_plus1"};
    const char *aggr_names[2] = {"rf_c2_aggr", "rf_c1_aggr"};
    for (int i = 0; i < 2; i++) {
        const Layer &pl = m.get(plus_names[i]);
        expect(pl.type == "Eltwise" && pl.eltwise_op == "SUM" && pl.bottoms.size() == 2 &&
                   pl.bottoms[0] == p.lateral[i + 1].out_blob, std::string(plus_names[i]) + " is not lateral + upsample");
        const Layer *crop = m.find(pl.bottoms[1]);
        expect(crop && crop->type == "Crop", std::string(plus_names[i]) + ": second input is not the Crop of the upsample");
        for (int o : crop->crop_offsets) expect(o == 0, crop->name + ": non-zero crop offset");
        expect(m.get(aggr_names[i]).bottoms.at(0) == pl.tops.at(0), std::string(aggr_names[i]) + " does not consume the sum");
        const FoldedConv &a = p.aggr[i];
        expect(a.k == 3 && a.pad == 1 && a.stride == 1 && a.group == 1 && a.cin == 64 && a.cout == 64 && a.relu,
               std::string(aggr_names[i]) + " is not 3x3 64->64 + BN + ReLU");
    }

    // ---- SSH context modules + heads ----
    const std::string feats[3] = {p.lateral[0].out_blob, p.aggr[0].out_blob, p.aggr[1].out_blob};
    const int strides_fpn[3] = {32, 16, 8};
    for (int i = 0; i < 3; i++) {
        SshModule &s = p.ssh[i];
        s.stride = strides_fpn[i];
        std::string pre = "rf_c" + std::to_string(3 - i) + "_det_";
        FoldedConv det1 = fold_conv(m, pre + "conv1");
        FoldedConv ctx1 = fold_conv(m, pre + "context_conv1");
        FoldedConv ctx2 = fold_conv(m, pre + "context_conv2");
        FoldedConv ctx31 = fold_conv(m, pre + "context_conv3_1");
        FoldedConv ctx32 = fold_conv(m, pre + "context_conv3_2");
        auto is3x3 = [](const FoldedConv &f, int cin, int cout) {
            return f.k == 3 && f.pad == 1 && f.stride == 1 && f.group == 1 && f.cin == cin && f.cout == cout;
        };
        expect(is3x3(det1, 64, 32) && !det1.relu, pre + "conv1 is not 3x3 64->32 + BN");
        expect(is3x3(ctx1, 64, 16) && ctx1.relu, pre + "context_conv1 is not 3x3 64->16 + BN + ReLU");
        expect(is3x3(ctx2, 16, 16) && !ctx2.relu, pre + "context_conv2 is not 3x3 16->16 + BN");
        expect(is3x3(ctx31, 16, 16) && ctx31.relu, pre + "context_conv3_1 is not 3x3 16->16 + BN + ReLU");
        expect(is3x3(ctx32, 16, 16) && !ctx32.relu, pre + "context_conv3_2 is not 3x3 16->16 + BN");
        expect(m.get(pre + "conv1").bottoms.at(0) == feats[i] && m.get(pre + "context_conv1").bottoms.at(0) == feats[i],
               pre + "conv1 / context_conv1 do not read " + feats[i]);
        expect(m.get(pre + "context_conv2").bottoms.at(0) == ctx1.out_blob &&
                   m.get(pre + "context_conv3_1").bottoms.at(0) == ctx1.out_blob &&
                   m.get(pre + "context_conv3_2").bottoms.at(0) == ctx31.out_blob, pre + "context chain is mis-wired");
        const Layer &cat = m.get(pre + "concat");
        expect(cat.type == "Concat" && cat.axis == 1 && cat.bottoms.size() == 3 && cat.bottoms[0] == det1.out_blob &&
                   cat.bottoms[1] == ctx2.out_blob && cat.bottoms[2] == ctx32.out_blob, pre + "concat is mis-wired");
        const Layer *crelu = consumer(m, cat.tops.at(0), "ReLU", &cat);
        expect(crelu != nullptr, pre + "concat is not followed by ReLU");
        const std::string cat_out = crelu->tops.at(0);
        // The ReLU after the concat applies to all three branches, so every merged output is rectified.
        det1.relu = ctx2.relu = ctx32.relu = true;
        s.conv_a = merge({det1, ctx1});
        s.conv_a.out_blob = cat_out;          // channels 0..31 of the concat; 32..47 go to ctx1.out_blob
        s.conv_b = merge({ctx2, ctx31});
        s.conv_c = ctx32;
        std::string st = "stride" + std::to_string(s.stride);
        FoldedConv cls = fold_conv(m, "face_rpn_cls_score_" + st, false);
        FoldedConv box = fold_conv(m, "face_rpn_bbox_pred_" + st, false);
        FoldedConv lmk = fold_conv(m, "face_rpn_landmark_pred_" + st, false);
        // A anchors per cell: cls 2A (background A | foreground A), bbox 4A, landmark 10A.  A = 2 for the models the reference
        // ships (network "net3"); "net3a" (ratios {1, 1.5}, RetinaFace.cpp:219-221) needs a model with A = 4.
        const int A = cls.cout / 2;
        expect(cls.k == 1 && cls.cin == 64 && (A == 2 || A == 4) && cls.cout == 2 * A && box.cout == 4 * A && lmk.cout == 10 * A,
               "heads of " + st + " are not 1x1 64 -> 2A / 4A / 10A with A = 2 or 4");
        expect(p.anchors_per_cell == 0 || p.anchors_per_cell == A, "the strides disagree on the number of anchors per cell");
        p.anchors_per_cell = A;
        expect(m.get("face_rpn_cls_score_" + st).bottoms.at(0) == cat_out &&
                   m.get("face_rpn_bbox_pred_" + st).bottoms.at(0) == cat_out &&
                   m.get("face_rpn_landmark_pred_" + st).bottoms.at(0) == cat_out, "heads of " + st + " do not read " + cat_out);
        // Reshape(.,2,-1,.) -> Softmax(axis 1) -> Reshape(.,4,-1,.): 2-class softmax pairing channel a with a+2
        const Layer *r1 = consumer(m, cls.out_blob, "Reshape", nullptr);
        expect(r1 != nullptr, "cls_score of " + st + " is not reshaped");
        const Layer *sm = consumer(m, r1->tops.at(0), "Softmax", r1);
        expect(sm != nullptr && sm->axis == 1, "cls_score of " + st + " is not followed by Softmax(axis 1)");
        s.head = merge({cls, box, lmk});
        s.head.out_blob = "face_rpn_heads_" + st;
        s.conv_b.out_blob = cat_out;
        s.conv_c.out_blob = cat_out;
    }
    p.int8_scales = m.int8_scales;
    p.int8_qweights = m.int8_qweights;
    return p;
}

}  // namespace rf
