// demo.cpp -- what retinaface/main.cpp:7-58 does (construct RetinaFace, call detect / detectBatchImages), written
// against include/RetinaFace.h.  Reads raw BGR frames (rows x cols x 3 bytes) instead of cv::imread so it builds
// without OpenCV; prints one line per face: image index, score, box, global anchor index is not part of the class API.
//   demo <model_dir> <stem> <net_h> <net_w> <fp16|fp32> <threshold> <rows> <cols> <frame.bgr> [<frame.bgr> ...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "RetinaFace.h"

int main(int argc, char **argv) {
    if (argc < 10) { fprintf(stderr, "usage: demo model_dir stem net_h net_w fp16|fp32 thr rows cols frame.bgr...\n"); return 2; }
    string dir = argv[1];
    rf_options o;
    memset(&o, 0, sizeof(o));
    o.struct_size = sizeof(o);
    o.model_stem = argv[2];
    o.net_h = atoi(argv[3]);
    o.net_w = atoi(argv[4]);
    o.precision = strcmp(argv[5], "fp32") == 0 ? RF_PRECISION_FP32 : RF_PRECISION_FP16;
    float thr = (float)atof(argv[6]);
    int rows = atoi(argv[7]), cols = atoi(argv[8]);
    try {
        RetinaFace rf(dir, o, "net3", 0.4);
        vector<cv::Mat> imgs;
        for (int i = 9; i < argc; i++) {
            cv::Mat m(rows, cols, CV_8UC3);
            std::ifstream f(argv[i], std::ios::binary);
            f.read((char *)m.data, (std::streamsize)rows * cols * 3);
            if (!f) { fprintf(stderr, "cannot read %s\n", argv[i]); return 2; }
            imgs.push_back(m);
        }
        rf.detect(imgs[0], thr);                       // single-image entry point
        for (const FaceDetectInfo &d : rf.lastResult())
            printf("detect 0 %.6f %.3f %.3f %.3f %.3f\n", d.score, d.rect.x1, d.rect.y1, d.rect.x2, d.rect.y2);
        rf.detectBatchImages(imgs, thr);               // batch entry point
        for (size_t i = 0; i < imgs.size(); i++)
            for (const FaceDetectInfo &d : rf.lastBatchResult()[i])
                printf("batch %zu %.6f %.3f %.3f %.3f %.3f %.3f %.3f\n", i, d.score, d.rect.x1, d.rect.y1, d.rect.x2, d.rect.y2,
                       d.pts.x[0], d.pts.y[4]);
        cv::Mat empty;
        rf.detect(empty, thr);
        printf("empty %zu\n", rf.lastResult().size());
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
