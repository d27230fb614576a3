// plan.h -- host "graph compiler": walks the Caffe graph (model.h), checks it is the
// MobileNet-0.25 + FPN + SSH topology of model/mnet-deconv-0517.prototxt / model/mnet25.prototxt
// (SURVEY.md App. A), folds BatchNorm+Scale into the preceding convolution in fp64, merges sibling
// convolutions that share an input, and emits the fused-op list the HIP kernels implement.
// This is the job TensorRT's builder does for the reference (trtnetbase.cpp:252-330).
#pragma once
#include <string>
#include <vector>

#include "model.h"

namespace rf {

// A convolution with BN/Scale folded in.  Weights are O-H-W-I (k = (ky*kw + kx)*cin_g + c is
// contiguous), which is the K order of the implicit GEMMs and the NHWC activation layout.
struct FoldedConv {
    std::string name;        // reference layer name(s), '+'-joined when merged
    std::string out_blob;    // reference blob this op's output corresponds to (debug / parity tests)
    int cout = 0, cin = 0;   // cin = input channels of the whole conv (not per group)
    int k = 1, stride = 1, pad = 0, group = 1;
    bool relu = false;
    std::vector<float> w;    // [cout][k][k][cin/group]
    std::vector<float> b;    // [cout]
    double macs_per_out_pixel() const { return (double)cout * k * k * (cin / group); }
};

struct SshModule {           // one per stride, SURVEY.md App. A "SSH context module"
    int stride = 0;
    FoldedConv conv_a;       // det_conv1 (32, ReLU comes from concat_relu) || context_conv1 (16, ReLU)   64 -> 48
    FoldedConv conv_b;       // context_conv2 (16) || context_conv3_1 (16, ReLU)                            16 -> 32
    FoldedConv conv_c;       // context_conv3_2 (16)                                                        16 -> 16
    FoldedConv head;         // cls_score (2A) || bbox_pred (4A) || landmark_pred (10A), 1x1                64 -> 16A
};

struct Plan {
    int net_h = 0, net_w = 0;             // as written in the prototxt / rfw (may be overridden by options)
    FoldedConv conv0;                     // 3x3 s2 3->8 on raw RGB 0..255
    struct DwPw { FoldedConv dw, pw; };
    std::vector<DwPw> blocks;             // 13 depthwise+pointwise pairs (conv1..conv26)
    FoldedConv lateral[3];                // [0] rf_c3_lateral (256->64), [1] rf_c2_lateral (128->64), [2] rf_c1_red_conv (64->64)
    FoldedConv aggr[2];                   // [0] rf_c2_aggr, [1] rf_c1_aggr  (input = lateral + bilinear x2 upsample of the coarser level)
    SshModule ssh[3];                     // strides 32, 16, 8
    int anchors_per_cell = 0;             // A: head channels are 2A | 4A | 10A (2 for the shipped models)
    // TensorRT calibration cache: tensor (blob) name -> per-tensor activation scale, real ~= q * scale (SURVEY App. B.7;
    // consumed by the int8 engine).  Empty when the model carries no table.
    std::vector<std::pair<std::string, float>> int8_scales;
    // calibrated int8 weights per fused dense conv (model.h QWeights; empty: the int8 engine rounds to nearest)
    std::vector<QWeights> int8_qweights;
};

Plan compile_plan(const Model &m);        // throws ModelError when the graph is not the expected topology

}  // namespace rf
