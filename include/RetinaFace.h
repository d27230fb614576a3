/*
 * RetinaFace.h -- drop-in for the reference's detector class (retinaface/RetinaFace.h:63-78): the same
 * constructor, detect() and detectBatchImages() signatures and the same public result PODs
 * (anchor_box, FacePts, FaceDetectInfo: retinaface/RetinaFace.h:15-42), implemented on the MI355X engine
 * behind include/retinaface_amd.h instead of TensorRT + NPP + CPU loops.
 *
 * Differences a caller can observe, all additive:
 *   - detect()/detectBatchImages() return void in the reference and drop their result
 *     (RetinaFace.cpp:726-747, :916-939); here the detections stay available through lastResult() /
 *     lastBatchResult() until the next call.  Coordinates are network-input pixels, as in the reference.
 *   - a second constructor takes rf_options (precision, net size, batch, model stem) -- the reference bakes
 *     these in at compile time / in prototxt line 7.
 *   - errors throw std::runtime_error instead of abort()/exit(0).
 */
#ifndef RETINAFACE_H
#define RETINAFACE_H

#include <map>
#include <string>
#include <vector>

#if !defined(RF_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define RF_HAVE_OPENCV 1
#endif
#endif
#ifndef RF_HAVE_OPENCV
#include "rf_mat.h"
#endif
#include "retinaface_amd.h"

using namespace cv;
using namespace std;

struct anchor_win { float x_ctr, y_ctr, w, h; };
struct anchor_box { float x1, y1, x2, y2; };
struct FacePts { float x[5]; float y[5]; };
struct FaceDetectInfo { float score; anchor_box rect; FacePts pts; };

struct anchor_cfg {
    int STRIDE = 0;
    vector<int> SCALES;
    int BASE_SIZE = 0;
    vector<float> RATIOS;
    int ALLOWED_BORDER = 0;
};

class RetinaFace {
public:
    RetinaFace(string &model, string network = "net3", float nms = 0.4);
    RetinaFace(const string &model, const rf_options &options, string network = "net3", float nms = 0.4);
    ~RetinaFace();
    RetinaFace(const RetinaFace &) = delete;
    RetinaFace &operator=(const RetinaFace &) = delete;

    void detectBatchImages(vector<cv::Mat> imgs, float threshold = 0.5);
    void detect(const Mat &img, float threshold = 0.5, float scales = 1.0);

    /* the reference's Caffe-build detect (`void detect(Mat img, ...)`, RetinaFace.cpp:943-1075; it cannot share the name with
       the TensorRT-build signature above, RetinaFace.h:70): no resize, pad to x32, run at that size; result in lastResult() */
    void detectPad32(const Mat &img, float threshold = 0.5);

    /* `scale` of RetinaFace.cpp:585-589: multiply lastResult() coordinates by it for source-frame pixels (:732-739, commented) */
    float frameScale(const Mat &img) const { return rf_frame_scale(h_, img.rows, img.cols); }

    /* additive accessors */
    const vector<FaceDetectInfo> &lastResult() const { return last_; }
    const vector<vector<FaceDetectInfo>> &lastBatchResult() const { return lastBatch_; }
    int netWidth() const { return netW_; }
    int netHeight() const { return netH_; }
    rf_handle handle() const { return h_; }

private:
    void init(const string &model, const rf_options *options, const string &network, float nms);
    rf_handle h_ = nullptr;
    int netW_ = 0, netH_ = 0, maxDet_ = 256;
    string network;
    float nms_threshold;
    vector<FaceDetectInfo> last_;
    vector<vector<FaceDetectInfo>> lastBatch_;
};

#endif /* RETINAFACE_H */
