/*
 * rf_mat.h -- the sliver of cv::Mat the RetinaFace class touches, for builds without OpenCV
 * (OpenCV is a dependency of the reference, CMakeLists.txt:117-125, and is absent in this image).
 * The reference only ever reads img.data / img.rows / img.cols, assumes CV_8UC3 and calls empty()
 * (retinaface/RetinaFace.cpp:578,594-596).  With OpenCV installed include/RetinaFace.h uses the
 * real cv::Mat instead and this header is not seen.
 */
#ifndef RF_MAT_H
#define RF_MAT_H

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>

#ifndef CV_8UC3
#define CV_8UC3 16
#endif

namespace cv {

class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char *data = nullptr;
    size_t step = 0;                       /* bytes per row */

    Mat() {}
    /* owning, zero-initialised */
    Mat(int rows_, int cols_, int type_) : rows(rows_), cols(cols_), step((size_t)cols_ * 3), type_(type_) {
        owner_.reset(new unsigned char[(size_t)rows_ * step](), std::default_delete<unsigned char[]>());
        data = owner_.get();
    }
    /* non-owning view over caller memory (like cv::Mat(rows, cols, type, void* data, size_t step)) */
    Mat(int rows_, int cols_, int type_, void *data_, size_t step_ = 0)
        : rows(rows_), cols(cols_), data((unsigned char *)data_), step(step_ ? step_ : (size_t)cols_ * 3), type_(type_) {}

    bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
    int type() const { return type_; }
    int channels() const { return 3; }
    bool isContinuous() const { return step == (size_t)cols * 3; }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * 3);
        return m;
    }

private:
    int type_ = CV_8UC3;
    std::shared_ptr<unsigned char> owner_;
};

}  // namespace cv

#endif /* RF_MAT_H */
