// RetinaFace.cpp -- the reference's class surface (retinaface/RetinaFace.h:63-78) over the C ABI.
#include "../../include/RetinaFace.h"

#include <cstring>
#include <stdexcept>

static_assert(sizeof(FaceDetectInfo) == sizeof(rf_face) && sizeof(rf_face) == 15 * sizeof(float),
              "FaceDetectInfo must stay layout-compatible with the reference (RetinaFace.h:37-42)");

static void check(int status, rf_handle h, const char *what) {
    if (status == RF_OK || status == RF_ERR_TRUNCATED) return;
    throw std::runtime_error(std::string(what) + ": " + rf_last_error(h));
}

void RetinaFace::init(const string &model, const rf_options *options, const string &net, float nms) {
    network = net;
    nms_threshold = nms;
    rf_options o;
    if (options) o = *options; else { memset(&o, 0, sizeof(o)); }
    o.struct_size = sizeof(o);
    if (o.max_detections > 0) maxDet_ = o.max_detections;
    check(rf_create(model.c_str(), net.c_str(), nms, &o, &h_), nullptr, "RetinaFace");
    rf_get_net_size(h_, &netH_, &netW_, nullptr);
}

RetinaFace::RetinaFace(string &model, string net, float nms) { init(model, nullptr, net, nms); }
RetinaFace::RetinaFace(const string &model, const rf_options &options, string net, float nms) { init(model, &options, net, nms); }
RetinaFace::~RetinaFace() { rf_destroy(h_); }

void RetinaFace::detectBatchImages(vector<cv::Mat> imgs, float threshold) {
    const int n = (int)imgs.size();
    lastBatch_.assign(n, vector<FaceDetectInfo>());
    if (n == 0) return;
    vector<const uint8_t *> ptrs(n);
    vector<int> rows(n), cols(n), steps(n), counts(n, 0);
    for (int i = 0; i < n; i++) {
        ptrs[i] = imgs[i].empty() ? nullptr : imgs[i].data;
        rows[i] = imgs[i].rows; cols[i] = imgs[i].cols; steps[i] = (int)(size_t)imgs[i].step;
    }
    vector<rf_face> faces((size_t)n * maxDet_);
    check(rf_detect_batch(h_, ptrs.data(), rows.data(), cols.data(), steps.data(), n, threshold, faces.data(), maxDet_,
                          counts.data()), h_, "RetinaFace::detectBatchImages");
    for (int i = 0; i < n; i++) {
        int k = counts[i] < maxDet_ ? counts[i] : maxDet_;
        lastBatch_[i].resize(k);
        if (k) memcpy(lastBatch_[i].data(), &faces[(size_t)i * maxDet_], (size_t)k * sizeof(rf_face));
    }
}

void RetinaFace::detectPad32(const Mat &img, float threshold) {
    last_.clear();
    if (img.empty()) return;                       // RetinaFace.cpp:945-947
    const uint8_t *ptr = img.data;
    int rows = img.rows, cols = img.cols, step = (int)(size_t)img.step, count = 0;
    vector<rf_face> faces(maxDet_);
    check(rf_detect_batch_pad32(h_, &ptr, &rows, &cols, &step, 1, 0, threshold, faces.data(), maxDet_, &count), h_, "RetinaFace::detectPad32");
    int k = count < maxDet_ ? count : maxDet_;
    last_.resize(k);
    if (k) memcpy(last_.data(), faces.data(), (size_t)k * sizeof(rf_face));
}

void RetinaFace::detect(const Mat &img, float threshold, float /*scales: unused in the reference too*/) {
    last_.clear();
    if (img.empty()) return;                       // RetinaFace.cpp:578-580
    const uint8_t *ptr = img.data;
    int rows = img.rows, cols = img.cols, step = (int)(size_t)img.step, count = 0;
    vector<rf_face> faces(maxDet_);
    check(rf_detect_batch(h_, &ptr, &rows, &cols, &step, 1, threshold, faces.data(), maxDet_, &count), h_, "RetinaFace::detect");
    int k = count < maxDet_ ? count : maxDet_;
    last_.resize(k);
    if (k) memcpy(last_.data(), faces.data(), (size_t)k * sizeof(rf_face));
}
