// Minimal stand-in for <opencv2/opencv.hpp> -- TEST INFRASTRUCTURE ONLY (see ../README.md).
// Implements exactly the OpenCV surface the reference's RetinaFace.h / RetinaFace.cpp touch when built
// with -DUSE_TENSORRT and without USE_NPP: Mat (u8 / f32, interleaved), copyMakeBorder (constant),
// convertTo (u8 -> f32), cvtColor (BGR<->RGB), split (into pre-allocated planes), tick counters.
// cv::resize (bilinear, 8UC3) goes through a restatement of OpenCV's published fixed-point algorithm (oracle/csrc/cv_resize_linear.h).
#pragma once
#include "../../csrc/cv_resize_linear.h"
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_BGR2RGB 4

namespace cv {

typedef unsigned char uchar;

struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
};
struct Scalar {
    double val[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {}
};
struct Point2f {
    float x = 0, y = 0;
    Point2f() {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() {}
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};
struct Vec4f {
    float val[4];
    Vec4f() : val{0, 0, 0, 0} {}
    Vec4f(float a, float b, float c, float d) : val{a, b, c, d} {}
    float &operator[](int i) { return val[i]; }
    const float &operator[](int i) const { return val[i]; }
};

enum { BORDER_CONSTANT = 0 };

class Mat {
public:
    int rows = 0, cols = 0;
    uchar *data = nullptr;
    size_t step = 0;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t step_ = 0) : rows(r), cols(c), data((uchar *)ext), type_(type) {
        step = step_ ? step_ : (size_t)c * elemSize();
    }
    void create(int r, int c, int type) {
        type_ = type; rows = r; cols = c;
        step = (size_t)c * elemSize();
        own_ = std::make_shared<std::vector<uchar>>((size_t)r * step);
        data = own_->data();
    }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { return (size_t)channels() * (depth() == CV_32F ? 4 : 1); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    uchar *ptr(int r) { return data + (size_t)r * step; }
    const uchar *ptr(int r) const { return data + (size_t)r * step; }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) memcpy(m.ptr(r), ptr(r), (size_t)cols * elemSize());
        return m;
    }
    void convertTo(Mat &dst, int rtype) const {
        int ddepth = rtype & 7;
        Mat out(rows, cols, CV_MAKETYPE(ddepth, channels()));
        int n = cols * channels();
        for (int r = 0; r < rows; ++r) {
            if (depth() == CV_8U && ddepth == CV_32F) {
                const uchar *s = ptr(r); float *d = (float *)out.ptr(r);
                for (int i = 0; i < n; ++i) d[i] = (float)s[i];
            } else if (depth() == ddepth) {
                memcpy(out.ptr(r), ptr(r), (size_t)n * (ddepth == CV_32F ? 4 : 1));
            } else {
                fprintf(stderr, "ref_shim: convertTo %d -> %d not provided\n", depth(), ddepth); abort();
            }
        }
        dst = out;
    }

private:
    int type_ = 0;
    std::shared_ptr<std::vector<uchar>> own_;
};

inline void copyMakeBorder(const Mat &src, Mat &dst, int top, int bottom, int left, int right, int borderType,
                           const Scalar &value = Scalar()) {
    if (borderType != BORDER_CONSTANT || top < 0 || bottom < 0 || left < 0 || right < 0) {
        fprintf(stderr, "ref_shim: copyMakeBorder(%d,%d,%d,%d,type %d) not provided\n", top, bottom, left, right, borderType);
        abort();
    }
    Mat out(src.rows + top + bottom, src.cols + left + right, src.type());
    int cn = src.channels();
    for (int r = 0; r < out.rows; ++r)
        for (int c = 0; c < out.cols; ++c)
            for (int k = 0; k < cn; ++k) {
                if (src.depth() == CV_32F) ((float *)out.ptr(r))[c * cn + k] = (float)value.val[k];
                else out.ptr(r)[c * cn + k] = (uchar)value.val[k];
            }
    for (int r = 0; r < src.rows; ++r)
        memcpy(out.ptr(r + top) + (size_t)left * src.elemSize(), src.ptr(r), (size_t)src.cols * src.elemSize());
    dst = out;
}

inline void cvtColor(const Mat &src, Mat &dst, int code) {
    if (code != CV_BGR2RGB || src.channels() != 3) { fprintf(stderr, "ref_shim: cvtColor code %d not provided\n", code); abort(); }
    Mat out(src.rows, src.cols, src.type());
    for (int r = 0; r < src.rows; ++r)
        for (int c = 0; c < src.cols; ++c) {
            if (src.depth() == CV_32F) {
                const float *s = (const float *)src.ptr(r) + 3 * c; float *d = (float *)out.ptr(r) + 3 * c;
                d[0] = s[2]; d[1] = s[1]; d[2] = s[0];
            } else {
                const uchar *s = src.ptr(r) + 3 * c; uchar *d = out.ptr(r) + 3 * c;
                d[0] = s[2]; d[1] = s[1]; d[2] = s[0];
            }
        }
    dst = out;
}

// OpenCV's split() into a vector<Mat> reuses planes that already have the right size and type -- the reference relies
// on that to write straight into its input buffer (RetinaFace.cpp:629-641).
inline void split(const Mat &src, std::vector<Mat> &mv) {
    int cn = src.channels();
    mv.resize(cn);
    int ptype = CV_MAKETYPE(src.depth(), 1);
    size_t es = src.depth() == CV_32F ? 4 : 1;
    for (int k = 0; k < cn; ++k) {
        if (mv[k].rows != src.rows || mv[k].cols != src.cols || mv[k].type() != ptype || !mv[k].data) mv[k].create(src.rows, src.cols, ptype);
        for (int r = 0; r < src.rows; ++r)
            for (int c = 0; c < src.cols; ++c) memcpy(mv[k].ptr(r) + c * es, src.ptr(r) + ((size_t)c * cn + k) * es, es);
    }
}

// cv::resize with an empty dsize and INTER_LINEAR on CV_8UC3 -- the only form the reference uses (RetinaFace.cpp:613,:617) --
// through the restatement of OpenCV's published algorithm in oracle/csrc/cv_resize_linear.h (a stand-in like everything here).
inline void resize(const Mat &src, Mat &dst, Size dsize, double fx = 0, double fy = 0, int interpolation = 1) {
    if (src.type() != CV_8UC3 || dsize.width != 0 || dsize.height != 0 || fx <= 0 || fy <= 0 || interpolation != 1) {
        fprintf(stderr, "ref_shim: cv::resize is provided for CV_8UC3, Size(), fx, fy, INTER_LINEAR only\n");
        abort();
    }
    int drows = 0, dcols = 0;
    cv_resize_dsize(src.rows, src.cols, fx, fy, &drows, &dcols);
    Mat out(drows, dcols, CV_8UC3);
    cv_resize_linear_8uc3(src.data, src.rows, src.cols, src.step, out.data, drows, dcols, out.step, fx, fy);
    dst = out;
}

inline long long getTickCount() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline double getTickFrequency() { return 1e9; }

// drawing / display calls only appear in commented-out or non-compiled code; declared so stray uses still link
inline void rectangle(Mat &, Rect, const Scalar &, int = 1) {}
inline void circle(Mat &, Point2f, int, const Scalar &, int = 1) {}

}  // namespace cv
