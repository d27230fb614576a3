/*
 * retinaface_amd.h -- C ABI of the MI355X-native RetinaFace detect() engine.
 *
 * The reference (clancylian/retinaface) has no C ABI and no plugin registry: its only public
 * surface for this path is the C++ class in retinaface/RetinaFace.h:63-78.  This header is the
 * boundary a binding for that class (or any FFI: ctypes / cgo / JNI) would sit on; every entry
 * point cites the reference interface it replaces.  include/RetinaFace.h re-creates the C++
 * class verbatim on top of it.  No torch / OpenCV / HIP types appear in any signature.
 *
 * Threading: one handle = one caller thread at a time (the reference is single-threaded and
 * not re-entrant either: shared staging buffers, RetinaFace.cpp:323-336).  The rule is enforced: a
 * call that enters while another thread's call on the same handle is in flight returns
 * RF_ERR_INVALID_ARG at once ("handle in use by another thread" from rf_last_error on the refused
 * thread) and leaves the call in flight undisturbed.  Different handles are independent.
 * Errors: every call returns RF_OK (0) or a negative rf_status; rf_last_error() gives text.
 * (The reference abort()s / exit(0)s / bare-throws instead: trtutility.h:9-16,
 * trtnetbase.cpp:201-204, RetinaFace.cpp:327-335.)
 */
#ifndef RETINAFACE_AMD_H
#define RETINAFACE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_ABI_VERSION 2

typedef enum rf_status {
    RF_OK = 0,
    RF_ERR_INVALID_ARG = -1,
    RF_ERR_IO = -2,             /* model file missing / unreadable */
    RF_ERR_MODEL = -3,          /* graph is not the mnet0.25 + FPN + SSH topology this engine implements */
    RF_ERR_HIP = -4,            /* a HIP runtime call failed (no GPU, OOM, launch failure) */
    RF_ERR_UNSUPPORTED = -5,    /* e.g. int8 requested without a calibration table */
    RF_ERR_TRUNCATED = -6       /* more candidates / detections than the configured caps; counts[] hold the true numbers */
} rf_status;

typedef enum rf_precision {
    RF_PRECISION_FP32 = 0,      /* fp32 storage, exact-f32 MFMA (parity reference path)            */
    RF_PRECISION_FP16 = 1,      /* fp16 storage, fp32 accumulate (reference: kHALF, trtnetbase.cpp:268-274) */
    RF_PRECISION_INT8 = 2       /* int8 storage + v_mfma_i32_16x16x64_i8 with the per-tensor activation scales of the
                                   TensorRT calibration table (trtnetbase.cpp:295-311), per-channel weight scales */
} rf_precision;

/* Result record: byte-identical to the reference's FaceDetectInfo (RetinaFace.h:15-42):
 * score, rect{x1,y1,x2,y2}, pts{x[5], y[5]} = 15 floats, coordinates in network-input pixels. */
typedef struct rf_face {
    float score;
    float x1, y1, x2, y2;
    float px[5];
    float py[5];
} rf_face;

/* Replaces the compile-time / hard-coded configuration of the reference:
 * precision (trtnetbase.cpp:268-274,295), net H x W (prototxt line 7 via trtnetbase.cpp:149-197),
 * maxBatchSize = 8 (trtretinafacenet.cpp:21), model stem (RetinaFace.cpp:276).
 * Zero-initialise, set struct_size = sizeof(rf_options), fill what you need; 0 = default. */
typedef struct rf_options {
    uint32_t struct_size;
    int32_t precision;          /* rf_precision; default RF_PRECISION_FP16 */
    int32_t net_h, net_w;       /* 0 = the prototxt / .rfw input dims; must be multiples of 32 */
    int32_t max_batch;          /* images per launch (default 8); larger batches are chunked */
    int32_t device;             /* HIP device ordinal + 1; 0 = the calling thread's current HIP device.  Calls may come from any
                                   thread: every entry point binds the engine's device for its duration and restores the caller's */
    int32_t max_candidates;     /* pre-NMS candidates kept per image (default 4096, power of two) */
    int32_t max_detections;     /* post-NMS faces returned per image (default 256) */
    int32_t use_graph;          /* 1 (default) = replay a captured hipGraph per batch size; 2 = off */
    int32_t keep_outputs;       /* 1 = also materialise the 9 NCHW fp32 head blobs for rf_get_output() */
    const char *model_stem;     /* default "mnet-deconv-0517" (RetinaFace.cpp:276) */
    int32_t lanes;              /* launches that may be in flight at once (default 3): each lane owns a stream, its
                                   activation buffers and its hipGraphs */
    int32_t coalesce;           /* rf_enqueue_batch_device() batches merged into ONE launch of up to max_batch*coalesce
                                   images (default: about 256 images of 448 x 448 worth of pixels per launch, clamped to [1, 256]; 1 = off).  A merged launch starts when it is full or when one of
                                   its tickets is waited for.  rf_num_slots() = lanes * coalesce. */
    /* ---- fields added in ABI 2 (a caller compiled against ABI 1 passes the shorter struct_size and gets the defaults) ---- */
    int32_t copy_threads;       /* host threads (the caller's included) that stage host frames into pinned memory; 0 = min(8, cores/4) */
    int32_t n_devices;          /* > 1: one engine per entry of devices[], every rf_detect_batch* call is sharded by image over them */
    const int32_t *devices;     /* HIP device ordinals (0-based; an ordinal may repeat); NULL / n_devices <= 1: `device` above */
    int32_t plan_cache;         /* 0 / 1 (default): keep the packed weight image next to the model as <stem>.<precision>.rfplan -- the
                                   analogue of the reference's serialized-engine cache (trtnetbase.cpp:205-243): later rf_create calls
                                   read it back (one file read + one hipMemcpy, no parse / BN fold / packing) as long as the model files'
                                   hash, the precision and the library build match; 2 = neither read nor write it */
    int32_t oversize_resize;    /* how a frame LARGER than the net is shrunk: 0 / 1 (default) = aspect-kept area average, the reference's
                                   NPP build (resizeconvertion.cu:298-311, NPPI_INTER_SUPER: closed source, semantics by definition);
                                   2 = cv::resize bilinear + one-sided zero padding, the reference's build without NPP
                                   (RetinaFace.cpp:585-620; OpenCV's published 8-bit fixed-point algorithm, bit-exact to the oracle's) */
} rf_options;

typedef struct rf_engine *rf_handle;

/* RetinaFace::RetinaFace(string &model, string network = "net3", float nms = 0.4)
 * (RetinaFace.h:66, RetinaFace.cpp:205-337).  model_dir holds either <stem>.rfw (this repo's packed
 * model, the analogue of the reference's serialized-engine cache, trtnetbase.cpp:205-243) or
 * <stem>.prototxt + <stem>.caffemodel (+ <stem>.table.int8).  `network` is the reference's preset name: "net3" (2 anchors per
 * cell, what the shipped models carry), "net3a" (ratios {1, 1.5}: 4 anchors per cell; needs a model whose heads have 8 / 16 / 40
 * channels, otherwise RF_ERR_MODEL -- the reference would read past its score blob), and the presets the reference constructs
 * without anchors ("ssh", "vgg", "net4" ... and unknown names): accepted, every detect call returns zero faces as there. */
int rf_create(const char *model_dir, const char *network, float nms_threshold,
              const rf_options *options, rf_handle *out_handle);

/* Host-only (no GPU): what the constructor's `network` preset means (RetinaFace.cpp:209-271).  Writes the base anchors
 * (x1, y1, x2, y2) of one stride (32, 16 or 8) in the reference's order -- ratios outer, scales inner -- and returns their count A:
 * 2 for "net3", 4 for "net3a", 0 for the presets the reference leaves without ratios / without an anchor configuration ("ssh",
 * "vgg", "net4", "net5", "net5a", "net6": they construct and then find no faces), RF_ERR_INVALID_ARG for a bad stride.
 * Names the reference does not know behave like "ssh" there (it prints "network setting error" and goes on): also 0. */
int rf_preset_anchors(const char *network, int stride, float *out4, int cap_boxes);

/* RetinaFace::~RetinaFace() (RetinaFace.cpp:339-345) -- unlike the reference this frees everything. */
void rf_destroy(rf_handle h);

/* Text of the last failure on this handle (h == NULL: last rf_create failure on this thread). */
const char *rf_last_error(rf_handle h);

/* TrtNetBase::getNetHeight/getNetWidth/getMaxBatchSize (trtnetbase.h) */
int rf_get_net_size(rf_handle h, int *net_h, int *net_w, int *max_batch);

/* void RetinaFace::detectBatchImages(vector<cv::Mat> imgs, float threshold = 0.5)
 * (RetinaFace.h:69, RetinaFace.cpp:749-940) and, with n == 1, RetinaFace::detect (RetinaFace.h:70,
 * RetinaFace.cpp:576-747).  Frames are HOST pointers to CV_8UC3 BGR pixels: bgr[i] + y*steps[i] is row y
 * (cv::Mat data/step).  Frames no larger than the net are placed top-left on a zero canvas
 * (resizeconvertion.cu:298-303 with the scale factor clamped to 1); larger frames are shrunk first
 * (options.oversize_resize: area average as in the NPP build, or bilinear as in the build without NPP).  Unlike the reference (which returns void and drops faceInfo, RetinaFace.cpp:726-747)
 * results are returned: out[i*cap_per_image + k], k < min(counts[i], cap_per_image), score-descending,
 * coordinates in network-input pixels (as in the reference).  A NULL/0x0 frame yields count 0
 * (img.empty() early return, RetinaFace.cpp:578-580).  n may exceed max_batch (the reference overruns its buffers there,
 * trtretinafacenet.cpp:21): the call is cut into chunks of max_batch images, and the chunks join launch sequences of up to
 * max_batch x options.coalesce images instead of being launched one by one. */
int rf_detect_batch(rf_handle h, const uint8_t *const *bgr, const int *rows, const int *cols,
                    const int *steps, int n, float threshold,
                    rf_face *out, int cap_per_image, int *counts);

/* The reference's Caffe-build detect (`void RetinaFace::detect(Mat img, ...)`, RetinaFace.cpp:943-1075): NO resize and no
 * fixed network size -- every frame is zero-padded right/bottom to the next multiple of 32 (:950-953), the net is reshaped
 * to that size (:965-966), anchors are regenerated for it (:1035) and boxes are clipped to the padded size (:1055).
 * Coordinates are therefore source-frame pixels.  The handle keeps one engine per distinct padded size it has seen (created
 * on first use, least-recently-used evicted beyond 8); frames of one call are grouped by size and results returned in call
 * order.  `on_device` != 0: the frame pointers are device pointers.  Size limit as in the reference: 4096 x 3072. */
int rf_detect_batch_pad32(rf_handle h, const uint8_t *const *bgr, const int *rows, const int *cols,
                          const int *steps, int n, int on_device, float threshold,
                          rf_face *out, int cap_per_image, int *counts);

/* Factor by which the coordinates returned for a rows x cols frame must be multiplied to land in source-frame pixels:
 * max(cols / net_w, rows / net_h, 1) -- `scale` in RetinaFace.cpp:585-589; the reference's own mapping back is commented
 * out (:732-739), so results stay in network-input pixels and this is what a caller applies.  1.0 for frames that fit. */
float rf_frame_scale(rf_handle h, int rows, int cols);

/* Same, frames already resident in device memory (HBM) on the engine's device. */
int rf_detect_batch_device(rf_handle h, const void *const *d_bgr, const int *rows, const int *cols,
                           const int *steps, int n, float threshold,
                           rf_face *out, int cap_per_image, int *counts);

/* Asynchronous form of rf_detect_batch_device for serving loops: enqueue returns as soon as the
 * batch is queued on the engine's stream (n <= max_batch); `ticket` identifies one of
 * rf_num_slots() result slots.  rf_wait blocks until that batch has finished and copies its results.
 * Up to rf_num_slots() tickets may be outstanding; the engine merges consecutive enqueues into one launch
 * (options.coalesce) and overlaps launches on options.lanes streams. */
int rf_num_slots(rf_handle h);
int rf_enqueue_batch_device(rf_handle h, const void *const *d_bgr, const int *rows, const int *cols,
                            const int *steps, int n, float threshold, int *ticket);
int rf_wait(rf_handle h, int ticket, rf_face *out, int cap_per_image, int *counts);

/* Asynchronous form of rf_detect_batch: frames in HOST memory, exactly what the reference's callers hold (cv::Mat data / step,
 * RetinaFace.cpp:594, :760-782 upload them inside the call).  The frames are copied into the engine's pinned staging ring before
 * the call returns (the caller may reuse its buffers at once) by options.copy_threads host threads, cross PCIe as ONE DMA per
 * enqueue on the lane's stream, and that upload overlaps the compute of the super-batches already in flight on the other lanes.
 * Collect with rf_wait.  Frames inside a range registered with rf_host_register skip the staging copy: the DMA engine reads
 * them in place, so they must stay unchanged until the ticket has been waited for. */
int rf_enqueue_batch(rf_handle h, const uint8_t *const *bgr, const int *rows, const int *cols,
                     const int *steps, int n, float threshold, int *ticket);

/* Pin a caller-owned host range (a ring of camera / decoder buffers reused across calls) for in-place DMA, and release it.
 * rf_host_unregister first waits for every launch that may still read the range; rf_destroy releases what is left. */
int rf_host_register(rf_handle h, const void *ptr, size_t bytes);
int rf_host_unregister(rf_handle h, const void *ptr);

/* Where device frame pointers live (rf_detect_batch_device / rf_enqueue_batch_device on a node with several GPUs): the engine
 * asks the HIP runtime once per ALLOCATION and remembers the answer (own device / host-visible: read in place; another GPU:
 * peer copy over xGMI first); a remembered answer is re-checked against the runtime when it is older than 2 ms.  A caller that
 * FREES or RE-ALLOCATES frame buffers while the handle lives (hipFree + hipMalloc may hand the same address out on another
 * device) calls rf_invalidate_residency() after the free and before it passes pointers of the new allocation: the engine then
 * looks every pointer up afresh.  Frames passed to an outstanding ticket must stay allocated until rf_wait returns. */
int rf_invalidate_residency(rf_handle h);

/* Engines behind the handle: 1, or options.n_devices for an image-sharding multi-device handle. */
int rf_num_devices(rf_handle h);

/* The batch split in numbers (since rf_create, summed over the handle's engines): device frames that were found resident on ANOTHER
 * GPU and pulled over xGMI before their launch, and the hipMemcpyPeerAsync calls that carried them (frames that follow each other in
 * memory travel as one copy: a contiguous slice of a sharded batch = 1 copy; RF_SCATTER_PER_FRAME=1 in the environment = 1 per frame). */
int rf_scatter_stats(rf_handle h, long long *frames, long long *copies);

/* Global anchor index (offset(stride) + a*h*w + iy*w + ix, strides 32,16,8) of each
 * detection of image `image` of the most recent completed batch, in the same order as out[]. */
int rf_last_anchor_indices(rf_handle h, int image, int32_t *out, int cap);
/* Number of above-threshold anchors (pre-NMS) per image of the most recent completed batch. */
int rf_last_candidate_counts(rf_handle h, int *counts, int n);

/* The reference's three timers (RetinaFace.cpp:757,836,840-842,846,920; README columns pre/infer/post),
 * measured with HIP events on the engine's stream for the most recent *synchronous* detect call.
 * Only filled when the call ran un-graphed (options.use_graph == 2); otherwise returns total only. */
int rf_last_timings(rf_handle h, float *pre_ms, float *infer_ms, float *post_ms, float *total_ms);

/* TrtRetinaFaceNet::blob_by_name(name)->result[image] (trtretinafacenet.cpp:104-114): one of the 9
 * output blobs ("face_rpn_cls_prob_reshape_stride32", "face_rpn_bbox_pred_stride16", ...) as NCHW fp32.
 * Requires options.keep_outputs = 1.  Returns the number of floats written (or needed if dst == NULL). */
long rf_get_output(rf_handle h, const char *blob_name, int image, float *dst, size_t cap_floats);

/* Test / profiling hooks (no reference equivalent).
 * rf_debug_activation: copy an internal NHWC activation of the last batch, converted to fp32, by the
 *   name of the reference blob it corresponds to (e.g. "mobilenet0_relu10_fwd", "rf_c2_aggr_relu").
 *   dims = {H, W, C}.  Returns floats written (or needed if dst == NULL), negative on error.  int8 engine: values are
 *   dequantised with the tensor's scale(s); "<blob>#raw" returns the stored quanta themselves (-127..127 as floats).
 * rf_profile: time every launch of the hot path on the engine's own stream with HIP events (each launch repeated
 *   back to back between one event pair so the event overhead is amortised), `iters` passes over a batch of n
 *   net-sized device frames; returns the number of launches, fills names (reference layers covered), kernels
 *   (kernel instance, e.g. "dwpw<128,128,s1>"; both up to cap entries, pointers owned by the engine), avg_ms, and
 *   the algorithmic bytes / MACs each launch covers (layer-wise input+output elements x element size). */
long rf_debug_activation(rf_handle h, const char *blob_name, int image, float *dst, size_t cap_floats,
                         int dims[3]);
int rf_profile(rf_handle h, const void *const *d_bgr, int n, int iters, int cap,
               const char **names, const char **kernels, float *avg_ms, double *alg_bytes, double *macs);
/* rf_profile_compulsory_bytes: for the same launches in the same order, the bytes each one has to move through HBM at the very
 *   least GIVEN its fusion, for n images -- every tensor it reads from HBM once + every tensor it writes once (weights ignored).
 *   bench.py's `useful` HBM fraction = these bytes / kernel time / peak (the layer-wise alg_bytes of rf_profile also count tensors
 *   that never leave LDS).  Returns the number of launches. */
int rf_profile_compulsory_bytes(rf_handle h, int n, int cap, double *bytes);

/* Offline: pack <prototxt, caffemodel[, int8 table]> into a .rfw file (the analogue of the reference's
 * first-run engine serialisation, trtnetbase.cpp:231-243).  int8_table may be NULL. */
int rf_convert_model(const char *prototxt, const char *caffemodel, const char *int8_table,
                     const char *out_rfw);

/* Host-only test hook (runs without a GPU): BN-folded weights of one fused op of the plan compiled from
 * <model_dir>/<stem>.  op = "conv0", "dw<i>" / "pw<i>" (i = 0..12), "lateral<i>" (0..2), "aggr<i>" (0..1),
 * "ssh<i>.a" / ".b" / ".c" / ".head" (i = 0..2 for strides 32, 16, 8).  dims = {cout, k, k, cin/group};
 * w is [cout][k][k][cin/group], b is [cout].  Returns RF_OK or an error; pass NULL buffers to query dims. */
int rf_plan_folded(const char *model_dir, const char *stem, const char *op, float *w, size_t cap_w,
                   float *b, size_t cap_b, int dims[4]);

/* Host-only hook of the int8 calibration tool (tools/calibrate_int8.py --gptq; SURVEY 8f rank 3, reference INT8-Calibration-Tool/
 * calibrationtable.cpp:399-583 + TensorRT's own weight handling): one fused dense convolution of the int8 plan compiled from
 * <model_dir>/<stem> and the calibration table at `int8_table` (NULL: the one the model carries), BEFORE rounding: quanta[cout][ktot] = w * in_scale / row_scale (K order (ky, kx, c)),
 * in_scale[cin] (per input channel, as the engine applies it: a depthwise mid already carries the 127/255 of its 0..255 quanta),
 * row_scale[cout] (the weight grid), out_scale[cout] (1 for the heads).  dims = {cout, ktot, cin, input_is_u8_mid}.  op = the fused
 * op's name (reference layer names, '+'-joined when siblings are merged); op = "?<i>" enumerates: the i-th name comes back in `quanta`
 * (bytes) with its length in dims[0], RF_ERR_INVALID_ARG past the end.  NULL buffers are skipped. */
int rf_plan_int8_gemm(const char *model_dir, const char *stem, const char *int8_table, const char *op, float *quanta, size_t cap_q,
                      float *in_scale, size_t cap_in, float *row_scale, float *out_scale, size_t cap_out, int dims[4]);

/* Offline: <model_dir>/<stem> (an .rfw or prototxt + caffemodel) re-packed into out_rfw with a new calibration: int8_table (text,
 * the reference's format; NULL keeps the model's) and qweights ("<stem>.qweights.int8", the calibrated int8 weights tools/
 * calibrate_int8.py --gptq writes; NULL keeps the model's unless the table changed, which drops them).  The result is checked by
 * compiling and packing the int8 plan on the host; no GPU needed. */
int rf_attach_calibration(const char *model_dir, const char *stem, const char *int8_table, const char *qweights, const char *out_rfw);

/* Host-only test hook: the host half of rf_create (plan cache or model -> packed weight image) for <model_dir>/<stem> at a
 * precision, with the cache file at cache_path (NULL = the default place).  Returns 1 when the image came from the cache, 0 when it was
 * built from the model (and the cache written), or a negative rf_status. */
int rf_plan_cache_probe(const char *model_dir, const char *stem, int precision, const char *cache_path, size_t *image_bytes);

int rf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RETINAFACE_AMD_H */
