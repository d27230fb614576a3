// cv_resize_linear.h -- plain restatement of OpenCV's cv::resize(..., INTER_LINEAR) for CV_8UC3 (test infrastructure).
//
// The reference's build without NPP shrinks frames that are larger than the network input with
//     cv::resize(img, resize, cv::Size(), 1 / scale, 1 / scale);                       (retinaface/RetinaFace.cpp:611-620)
// i.e. OpenCV's default bilinear interpolation.  OpenCV is a third-party dependency that is not vendored in /root/reference
// (CMakeLists.txt links the system's opencv, no version pinned) and is not installed in this image, so its PUBLISHED
// algorithm is restated here: imgproc/src/resize.cpp, the legacy fixed-point path every 2.4 / 3.x / 4.x release shares for
// 8-bit images (INTER_LINEAR, not INTER_LINEAR_EXACT; builds that route 8UC3 through IPP may differ in the last bit):
//   * dsize = (cvRound(cols * fx), cvRound(rows * fy)), scale_x = 1 / fx, scale_y = 1 / fy               (cv::resize prologue)
//   * per destination column: fxf = float((dx + 0.5) * scale_x - 0.5), sx = floor(fxf), fxf -= sx; sx < 0 -> (0, 0);
//     sx >= cols - 1 -> (cols - 1, 0); taps ialpha = {saturate_cast<short>((1 - fxf) * 2048), saturate_cast<short>(fxf * 2048)}
//   * per destination row: the same without the clamp of the fraction; the two source rows are clipped to [0, rows - 1]
//   * horizontal pass in int32: H = S[sx] * a0 + S[sx + 1] * a1                                            (HResizeLinear)
//   * vertical pass: dst = uchar(( ((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2 ) >> 2)   (VResizeLinear<uchar,...>)
// PARITY STATUS: unpinned (no OpenCV here to compare with).  Used by oracle/ref_shim (so that the reference's own detect()
// runs its resize branch) and by oracle/csrc/rf_post_ref.c's host tests; the numpy twin is oracle/retinaface_post.py.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline int cvr_round(double v) { return (int)nearbyint(v); }          /* cvRound: round half to even (default FE mode) */
static inline short cvr_sat_short(float v) {
    int i = (int)nearbyintf(v);
    return (short)(i > 32767 ? 32767 : (i < -32768 ? -32768 : i));
}

/* destination size of cv::resize(src, dst, Size(), fx, fy) */
static inline void cv_resize_dsize(int rows, int cols, double fx, double fy, int *drows, int *dcols) {
    *dcols = cvr_round(cols * fx);
    *drows = cvr_round(rows * fy);
}

/* src: rows x cols x 3 u8, row pitch sstep bytes -> dst: drows x dcols x 3, row pitch dstep (sizes from cv_resize_dsize) */
static inline void cv_resize_linear_8uc3(const uint8_t *src, int rows, int cols, size_t sstep, uint8_t *dst, int drows, int dcols,
                                         size_t dstep, double fx, double fy) {
    const double scale_x = 1.0 / fx, scale_y = 1.0 / fy;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dcols);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * (size_t)dcols);
    int *h0 = (int *)malloc(sizeof(int) * 3 * (size_t)dcols), *h1 = (int *)malloc(sizeof(int) * 3 * (size_t)dcols);
    for (int dx = 0; dx < dcols; dx++) {
        float f = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(f);
        f -= (float)sx;
        if (sx < 0) { f = 0.f; sx = 0; }
        if (sx >= cols - 1) { f = 0.f; sx = cols - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = cvr_sat_short((1.f - f) * 2048.f);
        ialpha[2 * dx + 1] = cvr_sat_short(f * 2048.f);
    }
    for (int dy = 0; dy < drows; dy++) {
        float f = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(f);
        f -= (float)sy;
        const short b0 = cvr_sat_short((1.f - f) * 2048.f), b1 = cvr_sat_short(f * 2048.f);
        int y0 = sy < 0 ? 0 : (sy < rows ? sy : rows - 1), y1 = sy + 1 < 0 ? 0 : (sy + 1 < rows ? sy + 1 : rows - 1);
        const uint8_t *r0 = src + (size_t)y0 * sstep, *r1 = src + (size_t)y1 * sstep;
        for (int dx = 0; dx < dcols; dx++) {
            const int sx = xofs[dx], sx1 = sx + 1 < cols ? sx + 1 : cols - 1;     /* tap 1 has weight 0 at the right border */
            const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            for (int c = 0; c < 3; c++) {
                h0[3 * dx + c] = r0[3 * sx + c] * a0 + r0[3 * sx1 + c] * a1;
                h1[3 * dx + c] = r1[3 * sx + c] * a0 + r1[3 * sx1 + c] * a1;
            }
        }
        uint8_t *d = dst + (size_t)dy * dstep;
        for (int i = 0; i < 3 * dcols; i++)
            d[i] = (uint8_t)((((b0 * (h0[i] >> 4)) >> 16) + ((b1 * (h1[i] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(h0); free(h1);
}
