// model.h -- host-side model container: the Caffe graph + blobs the reference loads at
// retinaface/RetinaFace.cpp:276 (TensorRT caffe parser, trtnetbase.cpp:262-266) or
// RetinaFace.cpp:311-312 (Caffe Net + CopyTrainedLayersFrom), plus the TensorRT int8
// calibration cache (trtnetbase.cpp:31-44).  No third-party parser: a text-format reader for
// the prototxt, a protobuf wire reader for the caffemodel, and this repo's packed RFW1 format.
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace rf {

struct Blob {
    std::vector<int> dims;     // logical Caffe dims (conv weights: O, I/g, kh, kw)
    std::vector<float> data;   // in logical (Caffe) order
    size_t count() const { size_t n = 1; for (int d : dims) n *= (size_t)d; return n; }
};

struct Layer {
    std::string name, type;
    std::vector<std::string> bottoms, tops;
    int num_output = 0, kernel = 0, stride = 1, pad = 0, group = 1;
    int bias_term = 1;
    int axis = 1;
    int scale_bias = 0;
    int reshape_axis = 0, reshape_num_axes = -1;
    float eps = 0.f;
    std::string eltwise_op = "SUM";
    std::vector<int> crop_offsets, reshape_dims;
    std::vector<Blob> blobs;
};

// Calibrated int8 weights of one fused dense convolution (round 6): what tools/calibrate_int8.py --gptq writes next to the activation
// table.  The reference's calibration cache holds activation scales only and TensorRT rounds the weights itself (closed source); this
// engine's default is round-to-nearest on the per-output-channel grid (weights.h put_gemm), and a calibration run may replace the
// rounding DIRECTION of each weight by an error-compensated one chosen on the calibration activations (the grid, i.e. the row scales,
// stays what put_gemm derives from the model + table), plus a bias correction for the residual mean error.
struct QWeights {
    std::string op;                    // fused-op name (plan.h FoldedConv::name: reference layer names, '+'-joined when merged)
    int cout = 0, ktot = 0;            // ktot = k*k*cin, K order of FoldedConv::w
    std::vector<int8_t> q;             // [cout][ktot], each in [-127, 127]
    std::vector<float> bias_delta;     // [cout], added to the folded bias (real units)
};

struct Model {
    std::string name, input_name = "data";
    int input_shape[4] = {1, 3, 0, 0};
    std::vector<Layer> layers;
    std::vector<std::pair<std::string, float>> int8_scales;   // file order preserved
    std::vector<QWeights> int8_qweights;                       // optional (empty: round to nearest)

    const Layer *find(const std::string &layer_name) const;
    const Layer &get(const std::string &layer_name) const;     // throws ModelError
    bool scale_of(const std::string &tensor, float *scale) const;
};

struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };
struct ModelError : std::runtime_error { using std::runtime_error::runtime_error; };

// Caffe artefacts
Model load_prototxt(const std::string &path);
void attach_caffemodel(Model &m, const std::string &path);
void attach_int8_table(Model &m, const std::string &path);
// "<stem>.qweights.int8" (RFQ1: "RFQ1", u32 n, then per op: str name, u32 cout, u32 ktot, i8 q[cout*ktot], f32 bias_delta[cout])
void attach_int8_qweights(Model &m, const std::string &path);
void save_int8_qweights(const Model &m, const std::string &path);

// RFW1 packed container (layout documented in oracle/caffe_io.py and DESIGN.md)
Model load_rfw(const std::string &path);
void save_rfw(const Model &m, const std::string &path);

// model_dir resolution used by rf_create: <dir>/<stem>.rfw, else <dir>/<stem>.prototxt + .caffemodel
// (+ <dir>/<stem>.table.int8, falling back to <dir>/mnet-deconv-0517.table.int8 as the reference
// hard-codes that one table: trtnetbase.cpp:13; + <dir>/<stem>.qweights.int8 when it exists).
Model load_model_dir(const std::string &dir, const std::string &stem);

// FNV-1a 64 over the bytes of exactly the files load_model_dir would read for (dir, stem): the plan cache's validity key
uint64_t model_source_hash(const std::string &dir, const std::string &stem);
bool read_file_if_exists(const std::string &path, std::string *bytes);
void write_file_best_effort(const std::string &path, const std::string &bytes);      // tmp file + rename; failures are ignored

}  // namespace rf
