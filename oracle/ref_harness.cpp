// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.  C entry points around the *reference's own* RetinaFace class
// (retinaface/RetinaFace.cpp compiled unmodified from /root/reference by oracle/build_ref.py, against the stand-in
// third-party headers in oracle/ref_shim/).  Used to pin the oracle's restatements and to generate tests/golden/ref_*.npz.
//
// The reference's detect()/detectBatchImages() return nothing and keep nothing, so results are read the
// way the class itself offers: postProcess() (RetinaFace.cpp:495-574, a private member -- reached with the
// `#define private public` include trick below, which does not change the class layout) decodes + NMSes whatever the
// engine's blobs hold; nms() / anchors_plane() / generate_anchors_fpn() are called directly.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#define private public
#include "RetinaFace.h"
#undef private

// free functions with external linkage in RetinaFace.cpp (:106, :127)
vector<vector<anchor_box>> generate_anchors_fpn(bool dense_anchor, vector<anchor_cfg> cfg);
vector<anchor_box> anchors_plane(int height, int width, int stride, vector<anchor_box> base_anchors);

// ---------------------------------------------------------------------------------------------------------------------
// TrtRetinaFaceNet stand-in
// ---------------------------------------------------------------------------------------------------------------------
RefShimConfig &ref_shim_config() {
    static RefShimConfig c;
    return c;
}
static TrtRetinaFaceNet *g_primary = nullptr;

TrtRetinaFaceNet::TrtRetinaFaceNet(std::string name) : cfg_(ref_shim_config()), name_(name) {
    if (!g_primary) g_primary = this;
}
TrtRetinaFaceNet::~TrtRetinaFaceNet() {
    free(buffers_[0]);
    if (g_primary == this) g_primary = nullptr;
}
TrtRetinaFaceNet *TrtRetinaFaceNet::primary() { return g_primary; }

void TrtRetinaFaceNet::buildTrtContext(const std::string &, const std::string &, bool) {
    free(buffers_[0]);
    buffers_[0] = calloc((size_t)cfg_.max_batch * 3 * cfg_.net_h * cfg_.net_w, sizeof(float));
    blobs_.clear();
    static const int strides[3] = {32, 16, 8};     // output order of model/mnet-deconv-0517.prototxt
    static const struct { const char *stem; int c; } kinds[3] = {
        {"face_rpn_cls_prob_reshape_stride", 2}, {"face_rpn_bbox_pred_stride", 4}, {"face_rpn_landmark_pred_stride", 10}};   // x anchors per cell
    for (int s = 0; s < 3; ++s)
        for (int k = 0; k < 3; ++k) {
            TrtBlob b;
            b.layer_name = std::string(kinds[k].stem) + std::to_string(strides[s]);
            b.layer_index = (int)blobs_.size();
            b.outputDims.d[0] = kinds[k].c * cfg_.head_anchors;
            b.outputDims.d[1] = cfg_.net_h / strides[s];
            b.outputDims.d[2] = cfg_.net_w / strides[s];
            b.outputSize = b.outputDims.c() * b.outputDims.h() * b.outputDims.w();
            b.batchsize = cfg_.max_batch;
            b.result.assign(cfg_.max_batch, std::vector<float>((size_t)b.outputSize, 0.f));
            blobs_.push_back(b);
        }
}

void TrtRetinaFaceNet::doInference(int batchSize, float *) {
    size_t per = (size_t)3 * cfg_.net_h * cfg_.net_w;
    last_batch_ = batchSize;
    last_input_.assign((const float *)buffers_[0], (const float *)buffers_[0] + per * batchSize);
    if (cfg_.forward) cfg_.forward(last_input_.data(), batchSize, cfg_.net_h, cfg_.net_w, cfg_.user);
}

TrtBlob *TrtRetinaFaceNet::blob_by_name(std::string layer_name) {
    for (auto &b : blobs_)
        if (b.layer_name == layer_name) return &b;
    fprintf(stderr, "ref_shim: no blob %s\n", layer_name.c_str());
    abort();
}

std::vector<int> TrtRetinaFaceNet::getOutputWidth() { return {blobs_[0].outputDims.w(), blobs_[3].outputDims.w(), blobs_[6].outputDims.w()}; }
std::vector<int> TrtRetinaFaceNet::getOutputHeight() { return {blobs_[0].outputDims.h(), blobs_[3].outputDims.h(), blobs_[6].outputDims.h()}; }

void TrtRetinaFaceNet::set_output(const std::string &name, int image, const float *data, size_t count) {
    TrtBlob *b = blob_by_name(name);
    if (image < 0 || image >= (int)b->result.size() || count != (size_t)b->outputSize) {
        fprintf(stderr, "ref_shim: set_output(%s, %d, %zu) out of range (size %d)\n", name.c_str(), image, count, b->outputSize);
        abort();
    }
    b->result[image].assign(data, data + count);
}

// ---------------------------------------------------------------------------------------------------------------------
// C entry points (ctypes: oracle/ref_build.py)
// ---------------------------------------------------------------------------------------------------------------------
static RetinaFace *g_rf = nullptr;

static int pack(const std::vector<FaceDetectInfo> &v, float *out, int cap) {
    int n = 0;
    for (const auto &f : v) {
        if (n >= cap) break;
        float *o = out + 15 * n++;
        o[0] = f.score; o[1] = f.rect.x1; o[2] = f.rect.y1; o[3] = f.rect.x2; o[4] = f.rect.y2;
        for (int k = 0; k < 5; ++k) { o[5 + k] = f.pts.x[k]; o[10 + k] = f.pts.y[k]; }   // FacePts order: x[5], y[5]
    }
    return (int)v.size();
}

extern "C" {

typedef void (*rfref_forward_fn)(const float *input, int n, int h, int w, void *user);

// `network`: the constructor's preset name (RetinaFace.cpp:209-243).  head_anchors = A of the stand-in engine's output blobs
// (channels 2A / 4A / 10A; the shipped models have A = 2).
int rfref_create_net(const char *model_dir, const char *network, int head_anchors, int net_h, int net_w, int max_batch, float nms,
                     rfref_forward_fn fwd, void *user) {
    delete g_rf;
    g_rf = nullptr;
    RefShimConfig &c = ref_shim_config();
    c.net_h = net_h; c.net_w = net_w; c.max_batch = max_batch; c.forward = fwd; c.user = user;
    c.head_anchors = head_anchors;
    g_primary = nullptr;
    std::string m(model_dir);
    g_rf = new RetinaFace(m, network, nms);       // RetinaFace.cpp:205
    return 0;
}
int rfref_create(const char *model_dir, int net_h, int net_w, int max_batch, float nms, rfref_forward_fn fwd, void *user) {
    return rfref_create_net(model_dir, "net3", 2, net_h, net_w, max_batch, nms, fwd, user);
}

// what the constructor made of the preset: strides with an anchor configuration, and the base anchors of one of them
int rfref_num_levels() { return (int)g_rf->_feat_stride_fpn.size(); }
int rfref_base_anchors(int stride, float *out4, int cap_boxes) {
    auto it = g_rf->_anchors_fpn.find("stride" + std::to_string(stride));
    if (it == g_rf->_anchors_fpn.end()) return -1;
    const std::vector<anchor_box> &a = it->second;
    for (size_t i = 0; i < a.size() && (int)i < cap_boxes; ++i) { out4[4 * i] = a[i].x1; out4[4 * i + 1] = a[i].y1; out4[4 * i + 2] = a[i].x2; out4[4 * i + 3] = a[i].y2; }
    return (int)a.size();
}

void rfref_destroy() { delete g_rf; g_rf = nullptr; }

void rfref_set_output(const char *name, int image, const float *data, size_t count) {
    g_rf->trtNet->set_output(name, image, data, count);
}

// decode + NMS of image `image`'s blobs through RetinaFace::postProcess (reads result[0]; NMS threshold is the
// literal 0.4 at RetinaFace.cpp:571).  Returns the number of faces (may exceed cap).
int rfref_postprocess(int image, float threshold, float *out15, int cap) {
    static const char *stems[3] = {"face_rpn_cls_prob_reshape_stride", "face_rpn_bbox_pred_stride", "face_rpn_landmark_pred_stride"};
    static const int strides[3] = {32, 16, 8};
    auto swap0 = [&]() {
        if (image == 0) return;
        for (int s = 0; s < 3; ++s)
            for (int k = 0; k < 3; ++k) {
                TrtBlob *b = g_rf->trtNet->blob_by_name(std::string(stems[k]) + std::to_string(strides[s]));
                std::swap(b->result[0], b->result[image]);
            }
    };
    swap0();
    std::vector<FaceDetectInfo> v = g_rf->postProcess(g_rf->trtNet->getNetWidth(), g_rf->trtNet->getNetHeight(), threshold);
    swap0();
    return pack(v, out15, cap);
}

// the real detect() (RetinaFace.cpp:576-746): preprocess -> doInference (callback) -> decode -> NMS, result dropped by
// the reference; then postProcess on the same blobs to hand the faces out.  bgr: rows x cols x 3, `step` bytes per row.
int rfref_detect(const uint8_t *bgr, int rows, int cols, int step, float threshold, float *out15, int cap) {
    cv::Mat img(rows, cols, CV_8UC3, (void *)bgr, (size_t)step);
    if (step != cols * 3) img = img.clone();      // the reference assumes continuous frames
    g_rf->detect(img, threshold);
    return rfref_postprocess(0, threshold, out15, cap);
}

// the real detectBatchImages() (RetinaFace.cpp:749-940); faces are then read per image with rfref_postprocess.
void rfref_detect_batch(const uint8_t *const *bgr, const int *rows, const int *cols, int n, float threshold) {
    std::vector<cv::Mat> imgs;
    for (int i = 0; i < n; ++i) imgs.push_back(cv::Mat(rows[i], cols[i], CV_8UC3, (void *)bgr[i]));
    g_rf->detectBatchImages(imgs, threshold);
}

// the input tensor the reference handed to the engine in the last detect / detectBatchImages: n x 3 x H x W floats
long rfref_last_input(float *dst, size_t cap) {
    const std::vector<float> &v = g_rf->trtNet->last_input();
    if (dst && cap >= v.size()) memcpy(dst, v.data(), v.size() * sizeof(float));
    return (long)v.size();
}

// RetinaFace::nms (RetinaFace.cpp:439-492) on caller-supplied faces (n x 15)
int rfref_nms(const float *in15, int n, float threshold, float *out15, int cap) {
    std::vector<FaceDetectInfo> v(n);
    for (int i = 0; i < n; ++i) {
        const float *o = in15 + 15 * i;
        v[i].score = o[0]; v[i].rect = {o[1], o[2], o[3], o[4]};
        for (int k = 0; k < 5; ++k) { v[i].pts.x[k] = o[5 + k]; v[i].pts.y[k] = o[10 + k]; }
    }
    return pack(g_rf->nms(v, threshold), out15, cap);
}

// anchors of one level as the class built them at construction (RetinaFace.cpp:293-301): out = h*w*2 x 4
long rfref_anchors(int stride, float *out4, size_t cap_boxes) {
    const std::vector<anchor_box> &a = g_rf->_anchors["stride" + std::to_string(stride)];
    if (out4 && cap_boxes >= a.size())
        for (size_t i = 0; i < a.size(); ++i) { out4[4 * i] = a[i].x1; out4[4 * i + 1] = a[i].y1; out4[4 * i + 2] = a[i].x2; out4[4 * i + 3] = a[i].y2; }
    return (long)a.size();
}

// anchors_plane (RetinaFace.cpp:127) over generate_anchors_fpn (:106) for an arbitrary feature-map size
long rfref_anchors_plane(int height, int width, int level, float *out4, size_t cap_boxes) {
    vector<vector<anchor_box>> base = generate_anchors_fpn(false, g_rf->cfg);
    vector<anchor_box> a = anchors_plane(height, width, g_rf->_feat_stride_fpn[level], base[level]);
    if (out4 && cap_boxes >= a.size())
        for (size_t i = 0; i < a.size(); ++i) { out4[4 * i] = a[i].x1; out4[4 * i + 1] = a[i].y1; out4[4 * i + 2] = a[i].x2; out4[4 * i + 3] = a[i].y2; }
    return (long)a.size();
}

// single-box helpers (RetinaFace.cpp:378-398, :418-432)
void rfref_bbox_pred(const float *anchor4, const float *regress4, float *out4) {
    anchor_box a{anchor4[0], anchor4[1], anchor4[2], anchor4[3]};
    anchor_box r = g_rf->bbox_pred(a, cv::Vec4f(regress4[0], regress4[1], regress4[2], regress4[3]));
    out4[0] = r.x1; out4[1] = r.y1; out4[2] = r.x2; out4[3] = r.y2;
}
void rfref_landmark_pred(const float *anchor4, const float *pts10, float *out10) {
    anchor_box a{anchor4[0], anchor4[1], anchor4[2], anchor4[3]};
    FacePts p;
    for (int k = 0; k < 5; ++k) { p.x[k] = pts10[k]; p.y[k] = pts10[5 + k]; }
    FacePts r = g_rf->landmark_pred(a, p);
    for (int k = 0; k < 5; ++k) { out10[k] = r.x[k]; out10[5 + k] = r.y[k]; }
}

}  // extern "C"
