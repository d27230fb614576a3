// Host-only robustness test of the model readers and the graph compiler (model.cpp, plan.cpp), meant to run under
// -fsanitize=address,undefined (tests/test_host.py::test_model_readers_under_sanitizers): every corrupted variant of a model
// file must either load or throw rf::IoError / rf::ModelError -- never read out of bounds, overflow a size, or crash.
// The reference's own loader scrapes fixed columns out of the prototxt and trusts the caffemodel (trtnetbase.cpp:149-204, exit(0) on
// failure); this is the hardening the drop-in's readers are held to.
//   usage: test_readers_sanitized <assets dir> [<reference model dir>]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../retinaface_amd/csrc/model.h"
#include "../../retinaface_amd/csrc/plan.h"

static std::string slurp(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string &p, const std::string &b) {
    std::ofstream f(p, std::ios::binary | std::ios::trunc);
    f.write(b.data(), (std::streamsize)b.size());
}
static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

template <typename F> static int attempt(F &&f) {      // 1 = loaded, 0 = rejected cleanly
    try { f(); return 1; }
    catch (const rf::IoError &) { return 0; }
    catch (const rf::ModelError &) { return 0; }
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <assets dir> [<reference model dir>]\n", argv[0]); return 2; }
    const std::string assets = argv[1], tmp = std::string(getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp") + "/rf_readers_" + std::to_string((long)getpid());
    int loaded = 0, rejected = 0;
    // ---- the shipped containers load, compile, and survive a save / load round trip
    for (const char *stem : {"mnet25", "mnet-deconv-0517"}) {
        rf::Model m = rf::load_rfw(assets + "/" + stem + ".rfw");
        rf::Plan p = rf::compile_plan(m);
        if (p.blocks.size() != 13 || p.anchors_per_cell != 2) { fprintf(stderr, "%s: unexpected plan\n", stem); return 1; }
        rf::save_rfw(m, tmp + ".rfw");
        rf::Model m2 = rf::load_rfw(tmp + ".rfw");
        if (m2.layers.size() != m.layers.size() || m2.int8_scales.size() != m.int8_scales.size() || m2.int8_qweights.size() != m.int8_qweights.size()) {
            fprintf(stderr, "%s: round trip changed the model\n", stem); return 1;
        }
    }
    // ---- RFW1: truncations at 200 offsets, 400 single-byte corruptions, 100 four-byte length fields blown up
    const std::string good = slurp(assets + "/mnet25.rfw");
    for (int i = 0; i < 200; i++) {
        const size_t cut = i < 64 ? (size_t)i : (size_t)(rnd() % good.size());
        spit(tmp + ".rfw", good.substr(0, cut));
        (attempt([&] { rf::Model m = rf::load_rfw(tmp + ".rfw"); (void)rf::compile_plan(m); }) ? loaded : rejected)++;
    }
    for (int i = 0; i < 400; i++) {
        std::string b = good;
        const size_t at = i < 200 ? (size_t)(rnd() % 4096) : (size_t)(rnd() % b.size());       // headers and layer records are at the front
        b[at] = (char)(rnd() & 0xff);
        spit(tmp + ".rfw", b);
        (attempt([&] { rf::Model m = rf::load_rfw(tmp + ".rfw"); (void)rf::compile_plan(m); }) ? loaded : rejected)++;
    }
    for (int i = 0; i < 100; i++) {
        std::string b = good;
        const size_t at = (size_t)(rnd() % (b.size() - 4)) & ~(size_t)3;
        const uint32_t big = i & 1 ? 0xffffffffu : 0x7fffffffu;
        b.replace(at, 4, std::string((const char *)&big, 4));
        spit(tmp + ".rfw", b);
        (attempt([&] { rf::Model m = rf::load_rfw(tmp + ".rfw"); (void)rf::compile_plan(m); }) ? loaded : rejected)++;
    }
    // ---- calibration table and calibrated-weights files: garbage lines, truncations
    {
        rf::Model m = rf::load_rfw(assets + "/mnet25.rfw");
        const std::string tab = slurp(assets + "/mnet25.table.int8"), qw = slurp(assets + "/mnet25.qweights.int8");
        for (int i = 0; i < 100; i++) {
            std::string b = tab.substr(0, rnd() % (tab.size() + 1));
            if (i & 1) b += "\nnot a line\nname_only:\n: 3f800000\nx: zzzzzzzz\n";
            spit(tmp + ".table", b);
            rf::Model c = m;
            (attempt([&] { rf::attach_int8_table(c, tmp + ".table"); }) ? loaded : rejected)++;
        }
        for (int i = 0; i < 150; i++) {
            std::string b = qw;
            if (i < 75) b = b.substr(0, rnd() % (b.size() + 1));
            else { const size_t at = (size_t)(rnd() % 256); b[at] = (char)(rnd() & 0xff); }
            spit(tmp + ".qw", b);
            rf::Model c = m;
            (attempt([&] { rf::attach_int8_qweights(c, tmp + ".qw"); (void)rf::compile_plan(c); }) ? loaded : rejected)++;
        }
    }
    // ---- the reference's own files (dev container only): prototxt text reader + protobuf wire reader, then the same abuse
    if (argc > 2) {
        const std::string ref = argv[2];
        for (const char *stem : {"mnet25", "mnet-deconv-0517"}) {
            rf::Model m = rf::load_prototxt(ref + "/" + stem + ".prototxt");
            rf::attach_caffemodel(m, ref + "/" + stem + ".caffemodel");
            (void)rf::compile_plan(m);
        }
        const std::string proto = slurp(ref + "/mnet25.prototxt"), cm = slurp(ref + "/mnet25.caffemodel");
        for (int i = 0; i < 150; i++) {
            std::string b = proto;
            if (i < 50) b = b.substr(0, rnd() % b.size());
            else if (i < 100) b[rnd() % b.size()] = "{}:\"#\n x"[rnd() % 9];
            else b.insert(rnd() % b.size(), i & 1 ? "layer { name: \"x\" type: \"Convolution\" bottom: \"nope\" top: \"y\" convolution_param { num_output: 99999999 kernel_size: 3 } }\n" : "}}}}");
            spit(tmp + ".prototxt", b);
            (attempt([&] { rf::Model m = rf::load_prototxt(tmp + ".prototxt"); rf::attach_caffemodel(m, ref + "/mnet25.caffemodel"); (void)rf::compile_plan(m); }) ? loaded : rejected)++;
        }
        for (int i = 0; i < 150; i++) {
            std::string b = cm;
            if (i < 60) b = b.substr(0, rnd() % b.size());
            else { const size_t at = i < 110 ? (size_t)(rnd() % 2048) : (size_t)(rnd() % b.size()); b[at] = (char)(rnd() & 0xff); }
            spit(tmp + ".caffemodel", b);
            (attempt([&] { rf::Model m = rf::load_prototxt(ref + "/mnet25.prototxt"); rf::attach_caffemodel(m, tmp + ".caffemodel"); (void)rf::compile_plan(m); }) ? loaded : rejected)++;
        }
    }
    for (const char *ext : {".rfw", ".table", ".qw", ".prototxt", ".caffemodel"}) remove((tmp + ext).c_str());
    printf("ok: %d corrupted inputs loaded, %d rejected cleanly, none crashed\n", loaded, rejected);
    return 0;
}
