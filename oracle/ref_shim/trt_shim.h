// Stand-in for the reference's tensorrt/trtretinafacenet.h (force-included with -DTRTRETINAFACENET_H so the real
// header, which needs NvInfer.h, is skipped) -- TEST INFRASTRUCTURE ONLY (see README.md).
// Same public surface RetinaFace.cpp uses (trtretinafacenet.h:14-46, trtnetbase.h:57-164): the "engine" is a callback.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

struct DimsCHW {
    int d[3] = {0, 0, 0};
    int c() const { return d[0]; }
    int h() const { return d[1]; }
    int w() const { return d[2]; }
};

struct TrtBlob {
    std::string layer_name;
    int layer_index = 0;
    int outputSize = 0;
    std::vector<std::vector<float>> result;   // [image][c*h*w], Caffe NCHW order
    DimsCHW outputDims;
    int batchsize = 0;
};

// set by the harness (ref_harness.cpp) before a RetinaFace is constructed
struct RefShimConfig {
    int net_h = 448, net_w = 448, max_batch = 8;
    int head_anchors = 2;       // A: the stand-in engine's output blobs carry 2A / 4A / 10A channels (shipped models: 2)
    // input: n x 3 x H x W float, as the reference wrote it; the callee fills blobs through TrtRetinaFaceNet::set_output
    void (*forward)(const float *input, int n, int h, int w, void *user) = nullptr;
    void *user = nullptr;
};
RefShimConfig &ref_shim_config();

class TrtRetinaFaceNet {
public:
    explicit TrtRetinaFaceNet(std::string name);
    ~TrtRetinaFaceNet();
    void buildTrtContext(const std::string &deployfile, const std::string &modelfile, bool bUseCPUBuf = false);
    uint32_t getMaxBatchSize() const { return (uint32_t)cfg_.max_batch; }
    int getNetWidth() const { return cfg_.net_w; }
    int getNetHeight() const { return cfg_.net_h; }
    int getChannel() const { return 3; }
    void *&getBuffer(const int &index) { return buffers_[index]; }
    void doInference(int batchSize, float *input = nullptr);
    TrtBlob *blob_by_name(std::string layer_name);
    std::vector<int> getOutputWidth();
    std::vector<int> getOutputHeight();

    // harness side
    void set_output(const std::string &name, int image, const float *data, size_t count);
    const std::vector<float> &last_input() const { return last_input_; }
    int last_batch() const { return last_batch_; }
    static TrtRetinaFaceNet *primary();      // the first instance built ("retina"; the reference leaks a second one, RetinaFace.cpp:278)

private:
    RefShimConfig cfg_;
    std::string name_;
    void *buffers_[1] = {nullptr};
    std::vector<TrtBlob> blobs_;
    std::vector<float> last_input_;
    int last_batch_ = 0;
};
