/* serve.c -- a serving loop on the C ABI alone (plain C, no C++ / torch / OpenCV): what a caller of the reference's
 * detectBatchImages (retinaface/RetinaFace.cpp:749-940) looks like once the upload (:760-782) is a pipeline.
 *
 *   rf_serve <model_dir> <stem> <fp16|int8|fp32> <net_h> <net_w> <batch> <seconds> [device ...]
 *
 * A ring of host frame buffers (random noise + the occasional copy of frame.bgr if given via RF_SERVE_FRAME=<file rows cols>)
 * is pinned once with rf_host_register; the loop keeps rf_num_slots() batches in flight with rf_enqueue_batch and collects
 * them in order with rf_wait.  With more than one device ordinal the handle shards every synchronous rf_detect_batch call by
 * image instead (rf_options.devices) -- the asynchronous tickets belong to single-device handles.  Prints images/s and faces. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "retinaface_amd.h"

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int st_ = (call);                                                                \
        if (st_ != RF_OK && st_ != RF_ERR_TRUNCATED) {                                   \
            fprintf(stderr, "%s failed (%d): %s\n", #call, st_, rf_last_error(h));       \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 8) {
        fprintf(stderr, "usage: rf_serve model_dir stem fp16|int8|fp32 net_h net_w batch seconds [device ...]\n");
        return 2;
    }
    rf_handle h = NULL;
    rf_options o;
    memset(&o, 0, sizeof(o));
    o.struct_size = sizeof(o);
    o.model_stem = argv[2];
    o.precision = strcmp(argv[3], "fp32") == 0 ? RF_PRECISION_FP32 : strcmp(argv[3], "int8") == 0 ? RF_PRECISION_INT8 : RF_PRECISION_FP16;
    o.net_h = atoi(argv[4]);
    o.net_w = atoi(argv[5]);
    const int B = atoi(argv[6]);
    const double seconds = atof(argv[7]);
    o.max_batch = B;
    int32_t devs[16];
    int ndev = 0;
    for (int i = 8; i < argc && ndev < 16; i++) devs[ndev++] = atoi(argv[i]);
    if (ndev > 1) { o.n_devices = ndev; o.devices = devs; }
    else if (ndev == 1) o.device = devs[0] + 1;
    CHECK(rf_create(argv[1], "net3", 0.4f, &o, &h));

    const int H = o.net_h, W = o.net_w;
    const size_t frame_bytes = (size_t)H * W * 3;
    const int slots = ndev > 1 ? 1 : rf_num_slots(h);
    const int ring = (slots + 1) * B;                       /* one batch more than may be in flight */
    unsigned char *pool = (unsigned char *)malloc(frame_bytes * (size_t)ring);
    if (!pool) return 1;
    unsigned s = 12345u;
    for (size_t i = 0; i < frame_bytes * (size_t)ring; i++) { s = s * 1664525u + 1013904223u; pool[i] = (unsigned char)(120 + ((s >> 24) & 15)); }
    if (ndev <= 1) CHECK(rf_host_register(h, pool, frame_bytes * (size_t)ring));      /* DMA reads the ring in place */

    const uint8_t **ptrs = (const uint8_t **)malloc(sizeof(*ptrs) * (size_t)B);
    int *rows = (int *)malloc(sizeof(int) * (size_t)B), *cols = (int *)malloc(sizeof(int) * (size_t)B), *steps = (int *)malloc(sizeof(int) * (size_t)B);
    int *counts = (int *)malloc(sizeof(int) * (size_t)B), *tickets = (int *)malloc(sizeof(int) * (size_t)slots);
    rf_face *out = (rf_face *)malloc(sizeof(rf_face) * (size_t)B * 64);
    for (int i = 0; i < B; i++) { rows[i] = H; cols[i] = W; steps[i] = W * 3; }

    long images = 0, faces = 0, batches = 0;
    int head = 0, tail = 0, inflight = 0;                   /* ticket ring */
    const double t0 = now();
    while (now() - t0 < seconds || inflight > 0) {
        if (now() - t0 < seconds && inflight < slots) {
            for (int i = 0; i < B; i++) ptrs[i] = pool + frame_bytes * (size_t)(((batches % (slots + 1)) * B) + i);
            if (ndev > 1) {                                  /* image-sharded synchronous call */
                CHECK(rf_detect_batch(h, ptrs, rows, cols, steps, B, 0.5f, out, 64, counts));
                for (int i = 0; i < B; i++) faces += counts[i];
                images += B;
            } else {
                CHECK(rf_enqueue_batch(h, ptrs, rows, cols, steps, B, 0.5f, &tickets[head]));
                head = (head + 1) % slots;
                inflight++;
            }
            batches++;
            continue;
        }
        if (inflight > 0) {
            CHECK(rf_wait(h, tickets[tail], out, 64, counts));
            tail = (tail + 1) % slots;
            inflight--;
            for (int i = 0; i < B; i++) faces += counts[i];
            images += B;
        }
    }
    const double dt = now() - t0;
    printf("rf_serve: %ld images in %.3f s = %.0f images/s (%d x %d, batch %d, %s, %d device%s, %d tickets in flight), %ld faces\n", images, dt,
           (double)images / dt, W, H, B, argv[3], rf_num_devices(h), rf_num_devices(h) > 1 ? "s" : "", slots, faces);
    if (ndev <= 1) CHECK(rf_host_unregister(h, pool));
    rf_destroy(h);
    free(pool); free(ptrs); free(rows); free(cols); free(steps); free(counts); free(tickets); free(out);
    return 0;
}
