/*
 * Plain-C restatement of the reference's host-side anchor generation, threshold scan,
 * box / landmark regression, clipping and greedy NMS.  TEST INFRASTRUCTURE ONLY (see
 * oracle/__init__.py): linked by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg, never by the product.
 *
 * Follows, function by function:
 *   rfo_anchors_plane   retinaface/RetinaFace.cpp:127-154  (base anchors of :34-125, net3 cfg :245-268)
 *   rfo_bbox_pred       retinaface/RetinaFace.cpp:378-398
 *   rfo_clip_box        retinaface/RetinaFace.cpp:179-199
 *   rfo_landmark_pred   retinaface/RetinaFace.cpp:418-432
 *   rfo_decode          retinaface/RetinaFace.cpp:666-724 (== :999-1072)
 *   rfo_nms             retinaface/RetinaFace.cpp:434-492
 *
 * Built WITHOUT -mfma / -ffast-math so every float op rounds where the reference's
 * x86-64 build rounds.  "0.5 * (w - 1.0)" is double arithmetic in the reference; kept.
 * PARITY STATUS: pinned.  The reference has no tests, but its own RetinaFace.cpp compiles here against stand-in
 * third-party headers (oracle/build_ref.py); tests/test_reference_pin.py holds this file bit-exact to that build
 * (live, and through the frozen vectors of tests/golden/ref_pin.npz) and to the numpy restatement.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x1, y1, x2, y2; } rfo_box;
typedef struct { float score; rfo_box rect; float px[5]; float py[5]; } rfo_face;   /* RetinaFace.h:37-42 */

static const int kStrides[3] = {32, 16, 8};
static const int kScales[3][2] = {{32, 16}, {8, 4}, {2, 1}};

/* base anchors: generate_anchors(base 16, ratio {1.0}, scales) -- RetinaFace.cpp:34-103 */
static void base_anchor(int scale, rfo_box *out)
{
    rfo_box b = {0.f, 0.f, 15.f, 15.f};
    /* _ratio_enum with ratio 1.0 */
    float w = b.x2 - b.x1 + 1, h = b.y2 - b.y1 + 1;
    float xc = b.x1 + 0.5 * (w - 1), yc = b.y1 + 0.5 * (h - 1);
    float size = w * h;
    float sc = size / 1.0f;
    float w2 = roundf(sqrtf(sc));
    float h2 = roundf(w2 * 1.0f);
    rfo_box r;
    r.x1 = xc - 0.5 * (w2 - 1); r.y1 = yc - 0.5 * (h2 - 1);
    r.x2 = xc + 0.5 * (w2 - 1); r.y2 = yc + 0.5 * (h2 - 1);
    /* _scale_enum */
    w = r.x2 - r.x1 + 1; h = r.y2 - r.y1 + 1;
    xc = r.x1 + 0.5 * (w - 1); yc = r.y1 + 0.5 * (h - 1);
    w = w * scale; h = h * scale;
    out->x1 = xc - 0.5 * (w - 1); out->y1 = yc - 0.5 * (h - 1);
    out->x2 = xc + 0.5 * (w - 1); out->y2 = yc + 0.5 * (h - 1);
}

/* out: [2*height*width] boxes, k-major then row then col */
void rfo_anchors_plane(int stride_index, int height, int width, rfo_box *out)
{
    int stride = kStrides[stride_index];
    size_t n = 0;
    for (int k = 0; k < 2; k++) {
        rfo_box b;
        base_anchor(kScales[stride_index][k], &b);
        for (int ih = 0; ih < height; ih++) {
            int sh = ih * stride;
            for (int iw = 0; iw < width; iw++) {
                int sw = iw * stride;
                out[n].x1 = b.x1 + sw; out[n].y1 = b.y1 + sh;
                out[n].x2 = b.x2 + sw; out[n].y2 = b.y2 + sh;
                n++;
            }
        }
    }
}

static rfo_box rfo_bbox_pred(rfo_box anchor, const float regress[4])
{
    rfo_box rect;
    float width = anchor.x2 - anchor.x1 + 1;
    float height = anchor.y2 - anchor.y1 + 1;
    float ctr_x = anchor.x1 + 0.5 * (width - 1.0);
    float ctr_y = anchor.y1 + 0.5 * (height - 1.0);
    float pred_ctr_x = regress[0] * width + ctr_x;
    float pred_ctr_y = regress[1] * height + ctr_y;
    float pred_w = expf(regress[2]) * width;
    float pred_h = expf(regress[3]) * height;
    rect.x1 = pred_ctr_x - 0.5 * (pred_w - 1.0);
    rect.y1 = pred_ctr_y - 0.5 * (pred_h - 1.0);
    rect.x2 = pred_ctr_x + 0.5 * (pred_w - 1.0);
    rect.y2 = pred_ctr_y + 0.5 * (pred_h - 1.0);
    return rect;
}

static void rfo_clip_box(rfo_box *b, int width, int height)
{
    if (b->x1 < 0) b->x1 = 0;
    if (b->y1 < 0) b->y1 = 0;
    if (b->x2 > width - 1) b->x2 = width - 1;
    if (b->y2 > height - 1) b->y2 = height - 1;
}

static void rfo_landmark_pred(rfo_box anchor, const float pts[10], float px[5], float py[5])
{
    float width = anchor.x2 - anchor.x1 + 1;
    float height = anchor.y2 - anchor.y1 + 1;
    float ctr_x = anchor.x1 + 0.5 * (width - 1.0);
    float ctr_y = anchor.y1 + 0.5 * (height - 1.0);
    for (int j = 0; j < 5; j++) {
        px[j] = pts[2 * j] * width + ctr_x;
        py[j] = pts[2 * j + 1] * height + ctr_y;
    }
}

/*
 * heads[9]: for stride 32,16,8: cls_prob_reshape (4,h,w), bbox_pred (8,h,w), landmark_pred (20,h,w),
 * NCHW fp32 for ONE image.  Writes up to cap candidates in the reference's visiting order together with
 * their global anchor index; returns the number found (may exceed cap: caller must re-run with more room).
 */
int rfo_decode(const float *const heads[9], int net_h, int net_w, float threshold,
               rfo_face *out, int32_t *anchor_index, int cap)
{
    int n = 0;
    int goff = 0;
    for (int si = 0; si < 3; si++) {
        int stride = kStrides[si];
        int h = net_h / stride, w = net_w / stride;
        size_t count = (size_t)h * w;
        const float *score = heads[si * 3 + 0] + 2 * count;   /* second half of the blob */
        const float *bbox = heads[si * 3 + 1];
        const float *lmk = heads[si * 3 + 2];
        rfo_box *anchors = (rfo_box *)malloc(sizeof(rfo_box) * 2 * count);
        rfo_anchors_plane(si, h, w, anchors);
        for (size_t num = 0; num < 2; num++) {
            for (size_t j = 0; j < count; j++) {
                float conf = score[j + count * num];
                if (conf <= threshold) continue;
                if (n < cap) {
                    float regress[4], pts[10];
                    for (int c = 0; c < 4; c++) regress[c] = bbox[j + count * (c + num * 4)];
                    rfo_box rect = rfo_bbox_pred(anchors[j + count * num], regress);
                    rfo_clip_box(&rect, net_w, net_h);
                    for (size_t k = 0; k < 5; k++) {
                        pts[2 * k] = lmk[j + count * (num * 10 + k * 2)];
                        pts[2 * k + 1] = lmk[j + count * (num * 10 + k * 2 + 1)];
                    }
                    out[n].score = conf;
                    out[n].rect = rect;
                    rfo_landmark_pred(anchors[j + count * num], pts, out[n].px, out[n].py);
                    anchor_index[n] = goff + (int)(num * count + j);
                }
                n++;
            }
        }
        goff += (int)(2 * count);
        free(anchors);
    }
    return n;
}

typedef struct { rfo_face f; int32_t idx; } rfo_item;

static int cmp_item(const void *a, const void *b)
{
    const rfo_item *x = (const rfo_item *)a, *y = (const rfo_item *)b;
    if (x->f.score > y->f.score) return -1;
    if (x->f.score < y->f.score) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);   /* fixed tie order: anchor index ascending */
}

/* in-place: faces/anchor_index are reordered; returns the number kept (first `kept` entries). */
int rfo_nms(rfo_face *faces, int32_t *anchor_index, int n, float threshold)
{
    if (n <= 0) return 0;
    rfo_item *items = (rfo_item *)malloc(sizeof(rfo_item) * n);
    for (int i = 0; i < n; i++) { items[i].f = faces[i]; items[i].idx = anchor_index[i]; }
    qsort(items, n, sizeof(rfo_item), cmp_item);
    unsigned char *merged = (unsigned char *)calloc(n, 1);
    int kept = 0;
    for (int s = 0; s < n; s++) {
        if (merged[s]) continue;
        merged[s] = 1;
        rfo_box sb = items[s].f.rect;
        faces[kept] = items[s].f;
        anchor_index[kept] = items[s].idx;
        kept++;
        float area1 = (sb.x2 - sb.x1 + 1) * (sb.y2 - sb.y1 + 1);
        for (int i = s + 1; i < n; i++) {
            if (merged[i]) continue;
            rfo_box *bi = &items[i].f.rect;
            float x = sb.x1 > bi->x1 ? sb.x1 : bi->x1;
            float y = sb.y1 > bi->y1 ? sb.y1 : bi->y1;
            float w = (sb.x2 < bi->x2 ? sb.x2 : bi->x2) - x + 1;
            float h = (sb.y2 < bi->y2 ? sb.y2 : bi->y2) - y + 1;
            if (w <= 0 || h <= 0) continue;
            float area2 = (bi->x2 - bi->x1 + 1) * (bi->y2 - bi->y1 + 1);
            float area_intersect = w * h;
            if (area_intersect / (area1 + area2 - area_intersect) > threshold) merged[i] = 1;
        }
    }
    free(items);
    free(merged);
    return kept;
}


/* ---- oversize frames: OpenCV's bilinear resize, restated in cv_resize_linear.h (exported for the numpy-vs-C agreement test) */
#include "cv_resize_linear.h"

void rfo_cv_resize_dsize(int rows, int cols, double fx, double fy, int *drows, int *dcols) { cv_resize_dsize(rows, cols, fx, fy, drows, dcols); }
void rfo_cv_resize_linear(const uint8_t *src, int rows, int cols, uint8_t *dst, double fx, double fy) {
    int drows, dcols;
    cv_resize_dsize(rows, cols, fx, fy, &drows, &dcols);
    cv_resize_linear_8uc3(src, rows, cols, (size_t)cols * 3, dst, drows, dcols, (size_t)dcols * 3, fx, fy);
}
