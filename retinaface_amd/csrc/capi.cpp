// capi.cpp -- extern "C" boundary (include/retinaface_amd.h) over rf::Engine.  Exceptions never cross it.
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <list>
#include <string>
#include <utility>
#include <vector>

#include "engine.h"
#include "weights.h"

namespace {
thread_local std::string g_create_error;
thread_local const void *g_busy_handle = nullptr;      // the handle this thread's last call was refused on (see guarded)
const char kBusyText[] = "handle in use by another thread (one handle = one caller thread at a time)";
}

struct rf_engine {
    std::unique_ptr<rf::Engine> eng;
    std::string error;
    std::atomic_flag in_call = ATOMIC_FLAG_INIT;       // set for the duration of every call that touches the engine's state
    // rf_detect_batch_pad32: one engine per padded frame size, most recently used first
    std::string model_dir, network;
    float nms = 0.4f;
    rf::EngineOptions opt;
    std::list<std::pair<std::pair<int, int>, std::unique_ptr<rf::Engine>>> pool;
};

namespace {

// The boundary's contract is "one handle = one caller thread at a time" (include/retinaface_amd.h): lanes, tickets and staging buffers are
// unsynchronised by design.  A second thread entering while a call is in flight would corrupt them silently (the reference is equally
// unsafe, RetinaFace.cpp shared staging buffers -- but this library advertises a serving loop), so the entry points refuse it: the
// intruder gets RF_ERR_INVALID_ARG and rf_last_error() on ITS thread says why; the call in flight is not disturbed.
struct InCall {
    rf_engine *h;
    bool ok;
    explicit InCall(rf_engine *e) : h(e), ok(!e || !e->in_call.test_and_set(std::memory_order_acquire)) {}
    ~InCall() { if (h && ok) h->in_call.clear(std::memory_order_release); }
};

template <typename F> int guarded(rf_engine *h, F &&f) {
    InCall guard(h);
    if (!guard.ok) { g_busy_handle = h; return RF_ERR_INVALID_ARG; }
    if (h && g_busy_handle == h) g_busy_handle = nullptr;
    std::string &err = h ? h->error : g_create_error;
    try {
        return f();
    } catch (const rf::ArgError &e) { err = e.what(); return RF_ERR_INVALID_ARG;
    } catch (const rf::IoError &e) { err = e.what(); return RF_ERR_IO;
    } catch (const rf::ModelError &e) { err = e.what(); return RF_ERR_MODEL;
    } catch (const rf::HipError &e) { err = e.what(); return RF_ERR_HIP;
    } catch (const rf::Unsupported &e) { err = e.what(); return RF_ERR_UNSUPPORTED;
    } catch (const std::exception &e) { err = e.what(); return RF_ERR_INVALID_ARG; }
}

}  // namespace

extern "C" {

int rf_abi_version(void) { return RF_ABI_VERSION; }

int rf_create(const char *model_dir, const char *network, float nms_threshold, const rf_options *o, rf_handle *out) {
    return guarded(nullptr, [&]() -> int {
        if (!model_dir || !out) throw rf::ArgError("model_dir / out_handle is null");
        *out = nullptr;
        rf::EngineOptions eo;
        rf_options ocopy;
        if (o) {
            // ABI 1 callers pass the struct up to `coalesce`; the fields added later default to 0
            const size_t v1 = offsetof(rf_options, copy_threads);
            if (o->struct_size != sizeof(rf_options) && o->struct_size != v1) throw rf::ArgError("rf_options.struct_size mismatch");
            memset(&ocopy, 0, sizeof(ocopy));
            memcpy(&ocopy, o, o->struct_size);
            o = &ocopy;
            if (o->precision == RF_PRECISION_FP32 || o->precision == RF_PRECISION_FP16 || o->precision == RF_PRECISION_INT8)
                eo.precision = o->precision;
            else throw rf::ArgError("unknown precision");
            eo.net_h = o->net_h; eo.net_w = o->net_w;
            if (o->max_batch) eo.max_batch = o->max_batch;
            eo.device = o->device > 0 ? o->device - 1 : -1;
            if (o->max_candidates) eo.max_candidates = o->max_candidates;
            if (o->max_detections) eo.max_detections = o->max_detections;
            eo.use_graph = o->use_graph != 2;
            eo.keep_outputs = o->keep_outputs == 1;
            if (o->model_stem && *o->model_stem) eo.model_stem = o->model_stem;
            if (o->lanes) eo.lanes = o->lanes;
            if (o->coalesce) eo.coalesce = o->coalesce;
            eo.copy_threads = o->copy_threads;
            if (o->n_devices < 0 || (o->n_devices > 0 && !o->devices)) throw rf::ArgError("n_devices / devices mismatch");
            for (int i = 0; i < o->n_devices; i++) eo.devices.push_back(o->devices[i]);
            eo.plan_cache = o->plan_cache != 2;
            eo.resize_bilinear = o->oversize_resize == 2;
        }
        auto eng = rf::Engine::create(model_dir, network ? network : "net3", nms_threshold, eo);
        rf_engine *h = new rf_engine;
        h->eng = std::move(eng);
        h->model_dir = model_dir; h->network = network ? network : "net3"; h->nms = nms_threshold; h->opt = eo;
        *out = h;
        return RF_OK;
    });
}

int rf_preset_anchors(const char *network, int stride, float *out4, int cap_boxes) {
    if (!network || (stride != 32 && stride != 16 && stride != 8) || (cap_boxes > 0 && !out4)) return RF_ERR_INVALID_ARG;
    std::vector<float> ratios;
    (void)rf::network_preset(network, &ratios);
    float base[4][4] = {};
    rf::preset_base_anchors(ratios, stride == 32 ? 0 : stride == 16 ? 1 : 2, base);
    const int a = 2 * (int)ratios.size();
    for (int i = 0; i < a && i < cap_boxes; i++) memcpy(out4 + 4 * i, base[i], 4 * sizeof(float));
    return a;
}

void rf_destroy(rf_handle h) { delete h; }

const char *rf_last_error(rf_handle h) {
    if (h && g_busy_handle == h) return kBusyText;
    return h ? h->error.c_str() : g_create_error.c_str();
}

int rf_get_net_size(rf_handle h, int *net_h, int *net_w, int *max_batch) {
    if (!h) return RF_ERR_INVALID_ARG;
    if (net_h) *net_h = h->eng->net_h();
    if (net_w) *net_w = h->eng->net_w();
    if (max_batch) *max_batch = h->eng->max_batch();
    return RF_OK;
}

static int detect_common(rf_handle h, const uint8_t *const *frames, const int *rows, const int *cols, const int *steps,
                         int n, bool on_device, float thr, rf_face *out, int cap, int *counts) {
    if (!h) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int {
        bool tr = false;
        h->eng->detect(frames, rows, cols, steps, n, on_device, thr, out, cap, counts, &tr);
        if (tr) { h->error = "more candidates / detections than the configured caps"; return RF_ERR_TRUNCATED; }
        return RF_OK;
    });
}

int rf_detect_batch(rf_handle h, const uint8_t *const *bgr, const int *rows, const int *cols, const int *steps, int n,
                    float threshold, rf_face *out, int cap_per_image, int *counts) {
    return detect_common(h, bgr, rows, cols, steps, n, false, threshold, out, cap_per_image, counts);
}

int rf_detect_batch_device(rf_handle h, const void *const *d_bgr, const int *rows, const int *cols, const int *steps,
                           int n, float threshold, rf_face *out, int cap_per_image, int *counts) {
    return detect_common(h, (const uint8_t *const *)d_bgr, rows, cols, steps, n, true, threshold, out, cap_per_image, counts);
}

namespace {
constexpr size_t kPoolEngines = 8;

rf::Engine *pool_engine(rf_engine *h, int hs, int ws) {
    const std::pair<int, int> key(hs, ws);
    for (auto it = h->pool.begin(); it != h->pool.end(); ++it)
        if (it->first == key) {
            h->pool.splice(h->pool.begin(), h->pool, it);
            return h->pool.front().second.get();
        }
    rf::EngineOptions eo = h->opt;
    eo.net_h = hs; eo.net_w = ws;
    eo.lanes = 1; eo.coalesce = 1; eo.keep_outputs = false;
    if (!eo.devices.empty()) { eo.device = eo.devices[0]; eo.devices.clear(); }     // per-size engines live on the first device
    // activations scale with the frame: keep a pooled engine's footprint near max_batch x 448^2 worth of pixels
    const long px = (long)hs * ws, budget = (long)std::max(eo.max_batch, 1) * 448 * 448;
    eo.max_batch = (int)std::max(1L, std::min((long)eo.max_batch, budget / px));
    if (h->pool.size() >= kPoolEngines) h->pool.pop_back();
    h->pool.emplace_front(key, rf::Engine::create(h->model_dir, h->network, h->nms, eo));
    return h->pool.front().second.get();
}
}  // namespace

int rf_detect_batch_pad32(rf_handle h, const uint8_t *const *bgr, const int *rows, const int *cols, const int *steps, int n,
                          int on_device, float threshold, rf_face *out, int cap_per_image, int *counts) {
    if (!h) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int {
        if (n < 0 || (n > 0 && (!bgr || !rows || !cols || !counts))) throw rf::ArgError("null argument");
        std::vector<char> done(n, 0);
        bool truncated = false;
        for (int i = 0; i < n; i++) {
            if (done[i]) continue;
            if (!bgr[i] || rows[i] <= 0 || cols[i] <= 0) { counts[i] = 0; done[i] = 1; continue; }    // img.empty(), :945-947
            if (rows[i] > 4096 * 3072 / std::max(cols[i], 1)) throw rf::ArgError("frame larger than 4096x3072 (RetinaFace.cpp:325)");
            const int hs = (rows[i] + 31) / 32 * 32, ws = (cols[i] + 31) / 32 * 32;                   // :950-951
            std::vector<int> idx;
            for (int j = i; j < n; j++)
                if (!done[j] && bgr[j] && rows[j] > 0 && cols[j] > 0 && (rows[j] + 31) / 32 * 32 == hs && (cols[j] + 31) / 32 * 32 == ws)
                    idx.push_back(j);
            rf::Engine &eng = *pool_engine(h, hs, ws);
            const int m = (int)idx.size();
            std::vector<const uint8_t *> f(m);
            std::vector<int> r(m), c(m), st(m), cnt(m);
            for (int k = 0; k < m; k++) { f[k] = bgr[idx[k]]; r[k] = rows[idx[k]]; c[k] = cols[idx[k]]; st[k] = steps ? steps[idx[k]] : cols[idx[k]] * 3; }
            std::vector<rf_face> tmp((size_t)m * std::max(cap_per_image, 0));
            bool tr = false;
            eng.detect(f.data(), r.data(), c.data(), st.data(), m, on_device != 0, threshold, out ? tmp.data() : nullptr, cap_per_image,
                       cnt.data(), &tr);
            truncated = truncated || tr;
            for (int k = 0; k < m; k++) {
                counts[idx[k]] = cnt[k];
                if (out)
                    memcpy(out + (size_t)idx[k] * cap_per_image, tmp.data() + (size_t)k * cap_per_image,
                           sizeof(rf_face) * (size_t)std::min(cnt[k], cap_per_image));
                done[idx[k]] = 1;
            }
        }
        if (truncated) { h->error = "more candidates / detections than the configured caps"; return RF_ERR_TRUNCATED; }
        return RF_OK;
    });
}

float rf_frame_scale(rf_handle h, int rows, int cols) {
    if (!h || rows <= 0 || cols <= 0) return 1.f;
    const float sw = (float)cols / (float)h->eng->net_w(), sh = (float)rows / (float)h->eng->net_h();     // RetinaFace.cpp:585-589
    const float sc = sw > sh ? sw : sh;
    return sc > 1.f ? sc : 1.f;
}

int rf_num_slots(rf_handle h) { return h ? h->eng->num_slots() : RF_ERR_INVALID_ARG; }

int rf_enqueue_batch_device(rf_handle h, const void *const *d_bgr, const int *rows, const int *cols, const int *steps,
                            int n, float threshold, int *ticket) {
    if (!h || !ticket) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int { *ticket = h->eng->enqueue(d_bgr, rows, cols, steps, n, true, threshold); return RF_OK; });
}

int rf_enqueue_batch(rf_handle h, const uint8_t *const *bgr, const int *rows, const int *cols, const int *steps, int n,
                     float threshold, int *ticket) {
    if (!h || !ticket) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int {
        *ticket = h->eng->enqueue((const void *const *)bgr, rows, cols, steps, n, false, threshold);
        return RF_OK;
    });
}

int rf_host_register(rf_handle h, const void *ptr, size_t bytes) {
    if (!h) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int { h->eng->host_register(ptr, bytes); return RF_OK; });
}

int rf_host_unregister(rf_handle h, const void *ptr) {
    if (!h) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int { h->eng->host_unregister(ptr); return RF_OK; });
}

int rf_invalidate_residency(rf_handle h) {
    if (!h) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int { h->eng->invalidate_residency(); return RF_OK; });
}

int rf_num_devices(rf_handle h) { return h ? h->eng->num_devices() : RF_ERR_INVALID_ARG; }

int rf_scatter_stats(rf_handle h, long long *frames, long long *copies) {
    if (!h || !frames || !copies) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int { *frames = 0; *copies = 0; h->eng->scatter_stats(frames, copies); return RF_OK; });
}

int rf_wait(rf_handle h, int ticket, rf_face *out, int cap_per_image, int *counts) {
    if (!h) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int {
        bool tr = false;
        h->eng->wait(ticket, out, cap_per_image, counts, &tr);
        if (tr) { h->error = "more candidates / detections than the configured caps"; return RF_ERR_TRUNCATED; }
        return RF_OK;
    });
}

int rf_last_anchor_indices(rf_handle h, int image, int32_t *out, int cap) {
    if (!h || (cap > 0 && !out)) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int { return h->eng->last_anchor_indices(image, out, cap); });
}

int rf_last_candidate_counts(rf_handle h, int *counts, int n) {
    if (!h || !counts) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int { return h->eng->last_candidate_counts(counts, n); });
}

int rf_last_timings(rf_handle h, float *pre_ms, float *infer_ms, float *post_ms, float *total_ms) {
    if (!h) return RF_ERR_INVALID_ARG;
    h->eng->last_timings(pre_ms, infer_ms, post_ms, total_ms);
    return RF_OK;
}

long rf_get_output(rf_handle h, const char *blob_name, int image, float *dst, size_t cap_floats) {
    if (!h || !blob_name) return RF_ERR_INVALID_ARG;
    long r = 0;
    int st = guarded(h, [&]() -> int { r = h->eng->get_output(blob_name, image, dst, cap_floats); return RF_OK; });
    return st == RF_OK ? r : st;
}

long rf_debug_activation(rf_handle h, const char *blob_name, int image, float *dst, size_t cap_floats, int dims[3]) {
    if (!h || !blob_name) return RF_ERR_INVALID_ARG;
    long r = 0;
    int st = guarded(h, [&]() -> int { r = h->eng->debug_activation(blob_name, image, dst, cap_floats, dims); return RF_OK; });
    return st == RF_OK ? r : st;
}

int rf_profile(rf_handle h, const void *const *d_bgr, int n, int iters, int cap, const char **names, const char **kernels,
               float *avg_ms, double *alg_bytes, double *macs) {
    if (!h || !d_bgr) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int { return h->eng->profile(d_bgr, n, iters, cap, names, kernels, avg_ms, alg_bytes, macs); });
}

int rf_profile_compulsory_bytes(rf_handle h, int n, int cap, double *bytes) {
    if (!h) return RF_ERR_INVALID_ARG;
    return guarded(h, [&]() -> int { return h->eng->compulsory_bytes(n, cap, bytes); });
}

int rf_convert_model(const char *prototxt, const char *caffemodel, const char *int8_table, const char *out_rfw) {
    return guarded(nullptr, [&]() -> int {
        if (!prototxt || !caffemodel || !out_rfw) throw rf::ArgError("null path");
        rf::Model m = rf::load_prototxt(prototxt);
        rf::attach_caffemodel(m, caffemodel);
        if (int8_table && *int8_table) rf::attach_int8_table(m, int8_table);
        (void)rf::compile_plan(m);      // refuse to pack a graph the engine cannot run
        rf::save_rfw(m, out_rfw);
        return RF_OK;
    });
}

int rf_plan_cache_probe(const char *model_dir, const char *stem, int precision, const char *cache_path, size_t *image_bytes) {
    return guarded(nullptr, [&]() -> int {
        if (!model_dir) throw rf::ArgError("null argument");
        rf::EngineOptions eo;
        eo.precision = precision;
        if (stem && *stem) eo.model_stem = stem;
        if (cache_path && *cache_path) eo.plan_cache_path = cache_path;
        return rf::plan_cache_probe(model_dir, eo, image_bytes);
    });
}

int rf_plan_folded(const char *model_dir, const char *stem, const char *op, float *w, size_t cap_w, float *b,
                   size_t cap_b, int dims[4]) {
    return guarded(nullptr, [&]() -> int {
        if (!model_dir || !op) throw rf::ArgError("null argument");
        rf::Model m = rf::load_model_dir(model_dir, stem && *stem ? stem : "mnet-deconv-0517");
        rf::Plan p = rf::compile_plan(m);
        std::string name = op;
        const rf::FoldedConv *f = nullptr;
        auto idx = [&](const std::string &prefix, int limit) -> int {
            if (name.compare(0, prefix.size(), prefix) != 0) return -1;
            int i = std::atoi(name.c_str() + prefix.size());
            return i >= 0 && i < limit ? i : -1;
        };
        int i;
        // "dw<i>.eq" / "pw<i>.eq": block i as the fp16 engine packs it, after the per-channel depthwise equalisation (weights.h)
        rf::FoldedConv eq_dw, eq_pw;
        if (name.size() > 3 && name.compare(name.size() - 3, 3, ".eq") == 0) {
            const bool want_dw = name.compare(0, 2, "dw") == 0;
            name.resize(name.size() - 3);
            if ((i = idx(want_dw ? "dw" : "pw", 13)) < 0) throw rf::ArgError("unknown plan op '" + std::string(op) + "'");
            eq_dw = p.blocks[i].dw;
            eq_pw = p.blocks[i].pw;
            rf::WeightPack<rf::half_t>::equalize_depthwise(eq_dw, eq_pw);
            f = want_dw ? &eq_dw : &eq_pw;
        } else
        if (name == "conv0") f = &p.conv0;
        else if ((i = idx("dw", 13)) >= 0 && name.find('.') == std::string::npos) f = &p.blocks[i].dw;
        else if ((i = idx("pw", 13)) >= 0) f = &p.blocks[i].pw;
        else if ((i = idx("lateral", 3)) >= 0) f = &p.lateral[i];
        else if ((i = idx("aggr", 2)) >= 0) f = &p.aggr[i];
        else if ((i = idx("ssh", 3)) >= 0) {
            std::string tail = name.substr(name.find('.') == std::string::npos ? name.size() : name.find('.'));
            if (tail == ".a") f = &p.ssh[i].conv_a;
            else if (tail == ".b") f = &p.ssh[i].conv_b;
            else if (tail == ".c") f = &p.ssh[i].conv_c;
            else if (tail == ".head") f = &p.ssh[i].head;
        }
        if (!f) throw rf::ArgError("unknown plan op '" + name + "'");
        if (dims) { dims[0] = f->cout; dims[1] = f->k; dims[2] = f->k; dims[3] = f->cin / f->group; }
        if (w) { if (cap_w < f->w.size()) throw rf::ArgError("w too small"); memcpy(w, f->w.data(), f->w.size() * 4); }
        if (b) { if (cap_b < f->b.size()) throw rf::ArgError("b too small"); memcpy(b, f->b.data(), f->b.size() * 4); }
        return RF_OK;
    });
}

int rf_attach_calibration(const char *model_dir, const char *stem, const char *int8_table, const char *qweights, const char *out_rfw) {
    return guarded(nullptr, [&]() -> int {
        if (!model_dir || !out_rfw) throw rf::ArgError("null argument");
        rf::Model m = rf::load_model_dir(model_dir, stem && *stem ? stem : "mnet-deconv-0517");
        if (int8_table && *int8_table) {
            rf::attach_int8_table(m, int8_table);
            m.int8_qweights.clear();                   // calibrated weights belong to the table they were chosen under
        }
        if (qweights && *qweights) rf::attach_int8_qweights(m, qweights);
        rf::Plan p = rf::compile_plan(m);              // refuses a model the engine could not run
        if (!m.int8_scales.empty()) {
            rf::WeightPack<int8_t> wp;
            wp.pack(p);                                // ... and a table / weight file that does not fit it (missing tensors, wrong shapes)
        }
        rf::save_rfw(m, out_rfw);
        return RF_OK;
    });
}

int rf_plan_int8_gemm(const char *model_dir, const char *stem, const char *int8_table, const char *op, float *quanta, size_t cap_q,
                      float *in_scale, size_t cap_in, float *row_scale, float *out_scale, size_t cap_out, int dims[4]) {
    return guarded(nullptr, [&]() -> int {
        if (!model_dir || !op) throw rf::ArgError("null argument");
        rf::Model m = rf::load_model_dir(model_dir, stem && *stem ? stem : "mnet-deconv-0517");
        if (int8_table && *int8_table) rf::attach_int8_table(m, int8_table);
        m.int8_qweights.clear();                       // the grid and the unrounded quanta do not depend on an earlier calibration
        rf::Plan p = rf::compile_plan(m);
        rf::WeightPack<int8_t> wp;
        std::map<std::string, rf::WeightPack<int8_t>::GemmRecord> rec;
        wp.record_ = &rec;
        wp.pack(p);                                    // host only: nothing is uploaded
        auto it = rec.find(op);
        if (it == rec.end()) {
            // "?<i>": enumerate -- the i-th fused dense conv's name is returned through `dims` as its length and (when quanta != NULL) its bytes
            if (op[0] == '?') {
                size_t i = (size_t)std::atoi(op + 1);
                if (i >= rec.size()) return RF_ERR_INVALID_ARG;
                auto jt = rec.begin();
                std::advance(jt, (long)i);
                if (dims) dims[0] = (int)jt->first.size();
                if (quanta) { if (cap_q * 4 < jt->first.size()) throw rf::ArgError("name buffer too small"); memcpy(quanta, jt->first.data(), jt->first.size()); }
                return RF_OK;
            }
            throw rf::ArgError("the int8 plan has no fused dense conv '" + std::string(op) + "'");
        }
        const auto &r = it->second;
        if (dims) { dims[0] = r.cout; dims[1] = r.ktot; dims[2] = r.cin; dims[3] = r.in_u8; }
        if (quanta) { if (cap_q < r.quanta.size()) throw rf::ArgError("quanta too small"); memcpy(quanta, r.quanta.data(), r.quanta.size() * 4); }
        if (in_scale) { if (cap_in < r.in_scale.size()) throw rf::ArgError("in_scale too small"); memcpy(in_scale, r.in_scale.data(), r.in_scale.size() * 4); }
        if (row_scale) { if (cap_out < r.row_scale.size()) throw rf::ArgError("row_scale too small"); memcpy(row_scale, r.row_scale.data(), r.row_scale.size() * 4); }
        if (out_scale) { if (cap_out < r.out_scale.size()) throw rf::ArgError("out_scale too small"); memcpy(out_scale, r.out_scale.data(), r.out_scale.size() * 4); }
        return RF_OK;
    });
}

}  // extern "C"
