// model.cpp -- see model.h.  Reference interfaces replaced: TensorRT's ICaffeParser::parse
// (retinaface/tensorrt/trtnetbase.cpp:262-266), TrtNetBase::parseNet (trtnetbase.cpp:149-197, which
// scrapes the input dims by column offsets -- here the prototxt is actually parsed),
// Int8EntropyCalibrator2::readCalibrationCache (trtnetbase.cpp:31-44) and the serialized-engine
// cache (trtnetbase.cpp:205-243 -> RFW1).
#include "model.h"

#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace rf {

const Layer *Model::find(const std::string &n) const {
    for (const auto &l : layers)
        if (l.name == n) return &l;
    return nullptr;
}
const Layer &Model::get(const std::string &n) const {
    const Layer *l = find(n);
    if (!l) throw ModelError("model has no layer named '" + n + "'");
    return *l;
}
bool Model::scale_of(const std::string &tensor, float *scale) const {
    for (const auto &kv : int8_scales)
        if (kv.first == tensor) { *scale = kv.second; return true; }
    return false;
}

std::string slurp(const std::string &path, bool binary) {
    std::ifstream f(path, binary ? std::ios::binary : std::ios::in);
    if (!f) throw IoError("cannot open '" + path + "'");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// ------------------------------------------------------------------------------------------
// protobuf text format
// ------------------------------------------------------------------------------------------
namespace {

// integer scalar of a text-format field; a malformed one is a ModelError, not a std::invalid_argument escaping the reader
static long scalar_long(const std::string &text, const char *what) {
    try { size_t used = 0; const long v = std::stol(text, &used); if (used == text.size()) return v; } catch (const std::exception &) {}
    throw ModelError(std::string("prototxt: ") + what + " is not an integer: '" + text.substr(0, 40) + "'");
}

struct TextNode;
using Fields = std::vector<std::pair<std::string, TextNode>>;
struct TextNode {
    bool is_msg = false;
    std::string scalar;    // unquoted text of a scalar
    Fields fields;         // for messages, in file order

    const TextNode *first(const std::string &k) const {
        for (const auto &f : fields) if (f.first == k) return &f.second;
        return nullptr;
    }
    std::vector<const TextNode *> all(const std::string &k) const {
        std::vector<const TextNode *> v;
        for (const auto &f : fields) if (f.first == k) v.push_back(&f.second);
        return v;
    }
    std::string str(const std::string &k, const std::string &def = "") const {
        const TextNode *n = first(k);
        return n ? n->scalar : def;
    }
    // (std::stol / std::stod throw std::invalid_argument / std::out_of_range on a malformed scalar: a corrupt prototxt is a ModelError)
    long num(const std::string &k, long def) const {
        const TextNode *n = first(k);
        if (!n) return def;
        try { size_t used = 0; const long v = std::stol(n->scalar, &used); if (used == n->scalar.size()) return v; } catch (const std::exception &) {}
        throw ModelError("prototxt: field '" + k + "' is not an integer: '" + n->scalar.substr(0, 40) + "'");
    }
    double real(const std::string &k, double def) const {
        const TextNode *n = first(k);
        if (!n) return def;
        try { size_t used = 0; const double v = std::stod(n->scalar, &used); if (used == n->scalar.size()) return v; } catch (const std::exception &) {}
        throw ModelError("prototxt: field '" + k + "' is not a number: '" + n->scalar.substr(0, 40) + "'");
    }
    bool flag(const std::string &k, bool def) const {
        const TextNode *n = first(k);
        return n ? (n->scalar == "true" || n->scalar == "1") : def;
    }
};

struct Tokenizer {
    const std::string &s;
    size_t p = 0;
    explicit Tokenizer(const std::string &text) : s(text) {}
    // returns false at end; tok receives the token, quoted receives whether it was a string literal
    bool next(std::string &tok, bool &quoted) {
        quoted = false;
        for (;;) {
            while (p < s.size() && isspace((unsigned char)s[p])) p++;
            if (p < s.size() && s[p] == '#') { while (p < s.size() && s[p] != '\n') p++; continue; }
            break;
        }
        if (p >= s.size()) return false;
        char c = s[p];
        if (c == '{' || c == '}' || c == ':') { tok.assign(1, c); p++; return true; }
        if (c == '"' || c == '\'') {
            quoted = true;
            tok.clear();
            p++;
            while (p < s.size() && s[p] != c) {
                if (s[p] == '\\' && p + 1 < s.size()) p++;
                tok.push_back(s[p++]);
            }
            if (p >= s.size()) throw ModelError("prototxt: unterminated string");
            p++;
            return true;
        }
        size_t b = p;
        while (p < s.size() && !isspace((unsigned char)s[p]) && s[p] != '{' && s[p] != '}' &&
               s[p] != ':' && s[p] != '"' && s[p] != '#')
            p++;
        tok = s.substr(b, p - b);
        return true;
    }
};

void parse_msg(Tokenizer &t, TextNode &msg, bool closing) {
    msg.is_msg = true;
    std::string tok;
    bool q;
    while (t.next(tok, q)) {
        if (!q && tok == "}") {
            if (!closing) throw ModelError("prototxt: unbalanced '}'");
            return;
        }
        std::string key = tok;
        if (!t.next(tok, q)) throw ModelError("prototxt: dangling key '" + key + "'");
        if (!q && tok == ":") {
            if (!t.next(tok, q)) throw ModelError("prototxt: missing value for '" + key + "'");
        }
        TextNode child;
        if (!q && tok == "{") {
            parse_msg(t, child, true);
        } else {
            child.scalar = tok;
        }
        msg.fields.emplace_back(key, std::move(child));
    }
    if (closing) throw ModelError("prototxt: missing '}'");
}

}  // namespace

Model load_prototxt(const std::string &path) {
    std::string text = slurp(path, false);
    Tokenizer tk(text);
    TextNode root;
    parse_msg(tk, root, false);
    Model m;
    m.name = root.str("name");
    bool have_input = false;
    for (const TextNode *ln : root.all("layer")) {
        Layer l;
        l.name = ln->str("name");
        l.type = ln->str("type");
        for (const TextNode *b : ln->all("bottom")) l.bottoms.push_back(b->scalar);
        for (const TextNode *t : ln->all("top")) l.tops.push_back(t->scalar);
        if (l.type == "Input") {
            const TextNode *ip = ln->first("input_param");
            const TextNode *sh = ip ? ip->first("shape") : nullptr;
            if (!sh) throw ModelError("prototxt: Input layer without input_param.shape");
            auto dims = sh->all("dim");
            if (dims.size() != 4) throw ModelError("prototxt: Input shape must have 4 dims");
            for (int i = 0; i < 4; i++) m.input_shape[i] = (int)scalar_long(dims[i]->scalar, "Input shape dim");
            if (!l.tops.empty()) m.input_name = l.tops[0];
            have_input = true;
            continue;
        }
        if (l.type == "Convolution" || l.type == "Deconvolution") {
            const TextNode *cp = ln->first("convolution_param");
            if (!cp) throw ModelError("prototxt: " + l.name + " lacks convolution_param");
            l.num_output = (int)cp->num("num_output", 0);
            l.kernel = (int)cp->num("kernel_size", 0);
            l.stride = (int)cp->num("stride", 1);
            l.pad = (int)cp->num("pad", 0);
            l.group = (int)cp->num("group", 1);
            l.bias_term = cp->flag("bias_term", true);   // caffe.proto default
        } else if (l.type == "BatchNorm") {
            const TextNode *bp = ln->first("batch_norm_param");
            l.eps = (float)(bp ? bp->real("eps", 1e-5) : 1e-5);
        } else if (l.type == "Scale") {
            const TextNode *sp = ln->first("scale_param");
            l.scale_bias = sp ? sp->flag("bias_term", false) : 0;
        } else if (l.type == "Concat") {
            const TextNode *cp = ln->first("concat_param");
            l.axis = (int)(cp ? cp->num("axis", 1) : 1);
        } else if (l.type == "Softmax") {
            const TextNode *sp = ln->first("softmax_param");
            l.axis = (int)(sp ? sp->num("axis", 1) : 1);
        } else if (l.type == "Crop") {
            const TextNode *cp = ln->first("crop_param");
            l.axis = (int)(cp ? cp->num("axis", 2) : 2);
            if (cp) for (const TextNode *o : cp->all("offset")) l.crop_offsets.push_back((int)scalar_long(o->scalar, "crop offset"));
        } else if (l.type == "Reshape") {
            const TextNode *rp = ln->first("reshape_param");
            const TextNode *sh = rp ? rp->first("shape") : nullptr;
            if (sh) for (const TextNode *d : sh->all("dim")) l.reshape_dims.push_back((int)scalar_long(d->scalar, "reshape dim"));
            l.reshape_axis = (int)(rp ? rp->num("axis", 0) : 0);
            l.reshape_num_axes = (int)(rp ? rp->num("num_axes", -1) : -1);
        } else if (l.type == "Eltwise") {
            const TextNode *ep = ln->first("eltwise_param");
            l.eltwise_op = ep ? ep->str("operation", "SUM") : "SUM";
        } else if (l.type == "ReLU") {
        } else {
            throw ModelError("prototxt: unsupported layer type '" + l.type + "' (" + l.name + ")");
        }
        m.layers.push_back(std::move(l));
    }
    if (!have_input) throw ModelError("prototxt: no Input layer");
    return m;
}

// ------------------------------------------------------------------------------------------
// protobuf wire format (caffemodel)
// ------------------------------------------------------------------------------------------
namespace {

struct Wire {
    const uint8_t *p, *end;
    Wire(const uint8_t *b, size_t n) : p(b), end(b + n) {}
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        int shift = 0;
        while (p < end) {
            uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
            if (shift > 63) break;
        }
        throw ModelError("caffemodel: truncated varint");
    }
    // reads one field header + payload; for wire type 2 (len) sets sub/len, for 0 sets val
    void field(int &fno, int &wt, uint64_t &val, const uint8_t *&sub, size_t &len) {
        uint64_t key = varint();
        fno = (int)(key >> 3);
        wt = (int)(key & 7);
        val = 0; sub = nullptr; len = 0;
        if (wt == 0) { val = varint(); }
        else if (wt == 1) { sub = p; len = 8; p += 8; }
        else if (wt == 5) { sub = p; len = 4; p += 4; }
        else if (wt == 2) {
            len = (size_t)varint();
            if ((size_t)(end - p) < len) throw ModelError("caffemodel: truncated field");
            sub = p; p += len;
        } else throw ModelError("caffemodel: unsupported wire type");
        if (p > end) throw ModelError("caffemodel: truncated field");
    }
};

Blob parse_blob(const uint8_t *b, size_t n) {
    Blob blob;
    int legacy[4] = {1, 1, 1, 1};
    bool have_shape = false, have_legacy = false;
    Wire w(b, n);
    while (!w.done()) {
        int fno, wt; uint64_t val; const uint8_t *sub; size_t len;
        w.field(fno, wt, val, sub, len);
        if (fno == 5) {                     // repeated float data [packed = true]
            size_t cnt = len / 4;
            size_t old = blob.data.size();
            blob.data.resize(old + cnt);
            memcpy(blob.data.data() + old, sub, cnt * 4);
        } else if (fno == 7 && wt == 2) {   // BlobShape { repeated int64 dim = 1 [packed] }
            Wire s(sub, len);
            while (!s.done()) {
                int f2, w2; uint64_t v2; const uint8_t *s2; size_t l2;
                s.field(f2, w2, v2, s2, l2);
                if (f2 != 1) continue;
                if (w2 == 2) { Wire d(s2, l2); while (!d.done()) blob.dims.push_back((int)d.varint()); }
                else blob.dims.push_back((int)v2);
            }
            have_shape = true;
        } else if (fno >= 1 && fno <= 4 && wt == 0) {
            legacy[fno - 1] = (int)val;
            have_legacy = true;
        }
    }
    if (!have_shape) {
        if (have_legacy) blob.dims.assign(legacy, legacy + 4);
        else blob.dims = {(int)blob.data.size()};
    }
    if (blob.count() != blob.data.size()) throw ModelError("caffemodel: blob shape/data size mismatch");
    return blob;
}

}  // namespace

void attach_caffemodel(Model &m, const std::string &path) {
    std::string buf = slurp(path, true);
    Wire w((const uint8_t *)buf.data(), buf.size());
    std::map<std::string, std::vector<Blob>> by_name;
    while (!w.done()) {
        int fno, wt; uint64_t val; const uint8_t *sub; size_t len;
        w.field(fno, wt, val, sub, len);
        if (fno != 100 || wt != 2) continue;       // NetParameter.layer
        Wire lw(sub, len);
        std::string lname;
        std::vector<Blob> blobs;
        while (!lw.done()) {
            int f2, w2; uint64_t v2; const uint8_t *s2; size_t l2;
            lw.field(f2, w2, v2, s2, l2);
            if (f2 == 1 && w2 == 2) lname.assign((const char *)s2, l2);
            else if (f2 == 7 && w2 == 2) blobs.push_back(parse_blob(s2, l2));
        }
        if (!lname.empty() && !blobs.empty()) by_name[lname] = std::move(blobs);
    }
    for (auto &l : m.layers) {
        auto it = by_name.find(l.name);
        if (it != by_name.end()) l.blobs = it->second;
    }
}

void attach_int8_table(Model &m, const std::string &path) {
    std::string text = slurp(path, false);
    std::istringstream ss(text);
    std::string line;
    if (!std::getline(ss, line) || line.compare(0, 4, "TRT-") != 0)
        throw ModelError("'" + path + "' is not a TensorRT calibration cache");
    m.int8_scales.clear();
    while (std::getline(ss, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
        if (line.empty()) continue;
        size_t pos = line.rfind(": ");
        if (pos == std::string::npos) continue;
        std::string name = line.substr(0, pos);
        // the value is exactly 1..8 hex digits (std::stoul would throw std::invalid_argument / out_of_range on anything else, and accept trailing junk)
        const std::string hex = line.substr(pos + 2);
        if (name.empty() || hex.empty() || hex.size() > 8 || hex.find_first_not_of("0123456789abcdefABCDEF") != std::string::npos)
            throw ModelError("'" + path + "': malformed calibration line '" + line.substr(0, 80) + "'");
        uint32_t bits = (uint32_t)std::stoul(hex, nullptr, 16);
        float f;
        memcpy(&f, &bits, 4);    // the hex text is the big-endian spelling of the IEEE-754 word
        m.int8_scales.emplace_back(name, f);
    }
}

// ------------------------------------------------------------------------------------------
// RFW1
// ------------------------------------------------------------------------------------------
namespace {

struct Out {
    std::string b;
    void u32(uint32_t v) { b.append((const char *)&v, 4); }
    void i32(int32_t v) { b.append((const char *)&v, 4); }
    void f32(float v) { b.append((const char *)&v, 4); }
    void str(const std::string &s) { u32((uint32_t)s.size()); b.append(s); }
};

struct In {
    const std::string &b;
    size_t p = 0;
    explicit In(const std::string &buf) : b(buf) {}
    void need(size_t n) { if (n > b.size() - p) throw ModelError("rfw: truncated file"); }      // p <= b.size() always
    // element counts read from the file are bounded by the bytes that are left (each element takes >= `each` bytes), so a
    // corrupt count cannot drive a huge resize()
    uint32_t count(size_t each) { uint32_t n = u32(); if ((size_t)n > (b.size() - p) / each) throw ModelError("rfw: count exceeds file size"); return n; }
    uint32_t u32() { need(4); uint32_t v; memcpy(&v, b.data() + p, 4); p += 4; return v; }
    int32_t i32() { return (int32_t)u32(); }
    float f32() { need(4); float v; memcpy(&v, b.data() + p, 4); p += 4; return v; }
    std::string str() { uint32_t n = u32(); need(n); std::string s = b.substr(p, n); p += n; return s; }
};

bool is_ohwi(const Layer &l, size_t bi, const Blob &blob) {
    return (l.type == "Convolution" || l.type == "Deconvolution") && bi == 0 && blob.dims.size() == 4;
}


// u32 n, then per op: str name, u32 cout, u32 ktot, i8 q[cout * ktot], f32 bias_delta[cout]
void write_qweights(Out &o, const Model &m) {
    o.u32((uint32_t)m.int8_qweights.size());
    for (const auto &qw : m.int8_qweights) {
        o.str(qw.op);
        o.u32((uint32_t)qw.cout);
        o.u32((uint32_t)qw.ktot);
        o.b.append((const char *)qw.q.data(), qw.q.size());
        o.b.append((const char *)qw.bias_delta.data(), qw.bias_delta.size() * 4);
    }
}
void read_qweights(In &in, Model &m) {
    const uint32_t n = in.count(12);
    m.int8_qweights.clear();
    for (uint32_t i = 0; i < n; i++) {
        QWeights qw;
        qw.op = in.str();
        const uint32_t cout = in.u32(), ktot = in.u32();
        if (cout == 0 || ktot == 0 || cout > 4096 || ktot > (1u << 20)) throw ModelError("qweights: bad dims for '" + qw.op + "'");
        qw.cout = (int)cout;
        qw.ktot = (int)ktot;
        const size_t nq = (size_t)cout * ktot;
        in.need(nq + (size_t)cout * 4);
        qw.q.resize(nq);
        memcpy(qw.q.data(), in.b.data() + in.p, nq);
        in.p += nq;
        qw.bias_delta.resize(cout);
        memcpy(qw.bias_delta.data(), in.b.data() + in.p, (size_t)cout * 4);
        in.p += (size_t)cout * 4;
        for (int8_t v : qw.q)
            if (v == -128) throw ModelError("qweights: weight outside [-127, 127] in '" + qw.op + "'");
        m.int8_qweights.push_back(std::move(qw));
    }
}

}  // namespace

void save_rfw(const Model &m, const std::string &path) {
    Out o;
    o.b.append("RFW1");
    o.u32(1);
    o.str(m.name);
    o.str(m.input_name);
    for (int i = 0; i < 4; i++) o.u32((uint32_t)m.input_shape[i]);
    o.u32((uint32_t)m.layers.size());
    for (const auto &l : m.layers) {
        o.str(l.name);
        o.str(l.type);
        o.u32((uint32_t)l.bottoms.size());
        for (const auto &s : l.bottoms) o.str(s);
        o.u32((uint32_t)l.tops.size());
        for (const auto &s : l.tops) o.str(s);
        o.i32(l.num_output); o.i32(l.kernel); o.i32(l.stride); o.i32(l.pad); o.i32(l.group);
        o.i32(l.bias_term); o.i32(l.axis); o.i32(l.scale_bias); o.i32(l.reshape_axis);
        o.i32(l.reshape_num_axes);
        o.f32(l.eps);
        o.str(l.eltwise_op);
        o.u32((uint32_t)l.crop_offsets.size());
        for (int v : l.crop_offsets) o.i32(v);
        o.u32((uint32_t)l.reshape_dims.size());
        for (int v : l.reshape_dims) o.i32(v);
        o.u32((uint32_t)l.blobs.size());
        for (size_t bi = 0; bi < l.blobs.size(); bi++) {
            const Blob &blob = l.blobs[bi];
            bool ohwi = is_ohwi(l, bi, blob);
            o.u32(ohwi ? 1u : 0u);
            o.u32((uint32_t)blob.dims.size());
            for (int d : blob.dims) o.u32((uint32_t)d);
            if (!ohwi) {
                o.b.append((const char *)blob.data.data(), blob.data.size() * 4);
            } else {
                int O = blob.dims[0], I = blob.dims[1], H = blob.dims[2], W = blob.dims[3];
                std::vector<float> t(blob.data.size());
                for (int oo = 0; oo < O; oo++)
                    for (int ii = 0; ii < I; ii++)
                        for (int h = 0; h < H; h++)
                            for (int w = 0; w < W; w++)
                                t[(((size_t)oo * H + h) * W + w) * I + ii] =
                                    blob.data[(((size_t)oo * I + ii) * H + h) * W + w];
                o.b.append((const char *)t.data(), t.size() * 4);
            }
        }
    }
    o.u32((uint32_t)m.int8_scales.size());
    for (const auto &kv : m.int8_scales) { o.str(kv.first); o.f32(kv.second); }
    if (!m.int8_qweights.empty()) write_qweights(o, m);        // optional trailing section (files without it end here)
    std::ofstream f(path, std::ios::binary);
    if (!f) throw IoError("cannot write '" + path + "'");
    f.write(o.b.data(), (std::streamsize)o.b.size());
    if (!f) throw IoError("short write to '" + path + "'");
}

Model load_rfw(const std::string &path) {
    std::string buf = slurp(path, true);
    if (buf.size() < 8 || buf.compare(0, 4, "RFW1") != 0) throw ModelError("'" + path + "' is not an RFW1 file");
    In in(buf);
    in.p = 4;
    if (in.u32() != 1) throw ModelError("rfw: unsupported version");
    Model m;
    m.name = in.str();
    m.input_name = in.str();
    for (int i = 0; i < 4; i++) m.input_shape[i] = (int)in.u32();
    uint32_t nl = in.count(16);
    m.layers.resize(nl);
    for (auto &l : m.layers) {
        l.name = in.str();
        l.type = in.str();
        l.bottoms.resize(in.count(4));
        for (auto &s : l.bottoms) s = in.str();
        l.tops.resize(in.count(4));
        for (auto &s : l.tops) s = in.str();
        l.num_output = in.i32(); l.kernel = in.i32(); l.stride = in.i32(); l.pad = in.i32();
        l.group = in.i32(); l.bias_term = in.i32(); l.axis = in.i32(); l.scale_bias = in.i32();
        l.reshape_axis = in.i32(); l.reshape_num_axes = in.i32();
        l.eps = in.f32();
        l.eltwise_op = in.str();
        l.crop_offsets.resize(in.count(4));
        for (auto &v : l.crop_offsets) v = in.i32();
        l.reshape_dims.resize(in.count(4));
        for (auto &v : l.reshape_dims) v = in.i32();
        l.blobs.resize(in.count(8));
        for (auto &blob : l.blobs) {
            uint32_t layout = in.u32();
            uint32_t nd = in.count(4);
            if (nd > 8) throw ModelError("rfw: blob with more than 8 dims");
            blob.dims.resize(nd);
            // every dim positive and the product checked against the bytes left BEFORE anything is sized by it
            size_t cnt = 1;
            const size_t max_cnt = (buf.size() - in.p) / 4;
            for (auto &d : blob.dims) {
                const uint32_t v = in.u32();
                if (v == 0 || v > (1u << 24) || cnt > max_cnt / v) throw ModelError("rfw: bad blob dims");
                d = (int)v;
                cnt *= v;
            }
            in.need(cnt * 4);
            blob.data.resize(cnt);
            memcpy(blob.data.data(), buf.data() + in.p, cnt * 4);
            in.p += cnt * 4;
            if (layout == 1) {
                if (blob.dims.size() != 4) throw ModelError("rfw: OHWI blob must be 4-D");
                int O = blob.dims[0], I = blob.dims[1], H = blob.dims[2], W = blob.dims[3];
                std::vector<float> t(cnt);
                for (int oo = 0; oo < O; oo++)
                    for (int ii = 0; ii < I; ii++)
                        for (int h = 0; h < H; h++)
                            for (int w = 0; w < W; w++)
                                t[(((size_t)oo * I + ii) * H + h) * W + w] =
                                    blob.data[(((size_t)oo * H + h) * W + w) * I + ii];
                blob.data.swap(t);
            }
        }
    }
    uint32_t ns = in.count(8);
    for (uint32_t i = 0; i < ns; i++) {
        std::string k = in.str();
        float v = in.f32();
        m.int8_scales.emplace_back(k, v);
    }
    if (in.p < buf.size()) read_qweights(in, m);              // optional trailing section: calibrated int8 weights
    return m;
}

void attach_int8_qweights(Model &m, const std::string &path) {
    std::string buf = slurp(path, true);
    if (buf.size() < 8 || buf.compare(0, 4, "RFQ1") != 0) throw ModelError("'" + path + "' is not an RFQ1 file");
    In in(buf);
    in.p = 4;
    read_qweights(in, m);
    if (in.p != buf.size()) throw ModelError("qweights: trailing bytes in '" + path + "'");
}

void save_int8_qweights(const Model &m, const std::string &path) {
    Out o;
    o.b.append("RFQ1");
    write_qweights(o, m);
    std::ofstream f(path, std::ios::binary);
    if (!f) throw IoError("cannot write '" + path + "'");
    f.write(o.b.data(), (std::streamsize)o.b.size());
    if (!f) throw IoError("short write to '" + path + "'");
}

bool file_exists(const std::string &p) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    fclose(f);
    return true;
}

Model load_model_dir(const std::string &dir, const std::string &stem) {
    std::string base = dir + "/" + stem;
    if (file_exists(base + ".rfw")) return load_rfw(base + ".rfw");
    if (!file_exists(base + ".prototxt") || !file_exists(base + ".caffemodel"))
        throw IoError("no model in '" + dir + "': need " + stem + ".rfw or " + stem +
                      ".prototxt + " + stem + ".caffemodel");
    Model m = load_prototxt(base + ".prototxt");
    attach_caffemodel(m, base + ".caffemodel");
    if (file_exists(base + ".table.int8")) attach_int8_table(m, base + ".table.int8");
    else if (file_exists(dir + "/mnet-deconv-0517.table.int8"))
        attach_int8_table(m, dir + "/mnet-deconv-0517.table.int8");
    if (file_exists(base + ".qweights.int8")) attach_int8_qweights(m, base + ".qweights.int8");
    return m;
}

}  // namespace rf

// ------------------------------------------------------------------------------------------
// plan-cache support (weights.h): validity key of a model directory, small file helpers
// ------------------------------------------------------------------------------------------
namespace rf {

static void fnv1a(uint64_t &h, const std::string &bytes) {
    for (unsigned char c : bytes) { h ^= c; h *= 1099511628211ull; }
}

uint64_t model_source_hash(const std::string &dir, const std::string &stem) {
    const std::string base = dir + "/" + stem;
    uint64_t h = 1469598103934665603ull;
    if (file_exists(base + ".rfw")) { fnv1a(h, slurp(base + ".rfw", true)); return h; }
    if (!file_exists(base + ".prototxt") || !file_exists(base + ".caffemodel"))
        throw IoError("no model in '" + dir + "': need " + stem + ".rfw or " + stem + ".prototxt + " + stem + ".caffemodel");
    fnv1a(h, slurp(base + ".prototxt", true));
    fnv1a(h, slurp(base + ".caffemodel", true));
    if (file_exists(base + ".table.int8")) fnv1a(h, slurp(base + ".table.int8", true));
    else if (file_exists(dir + "/mnet-deconv-0517.table.int8")) fnv1a(h, slurp(dir + "/mnet-deconv-0517.table.int8", true));
    if (file_exists(base + ".qweights.int8")) fnv1a(h, slurp(base + ".qweights.int8", true));
    return h;
}

bool read_file_if_exists(const std::string &path, std::string *bytes) {
    if (!file_exists(path)) return false;
    try { *bytes = slurp(path, true); } catch (const IoError &) { return false; }
    return true;
}

void write_file_best_effort(const std::string &path, const std::string &bytes) {
    // unique per writer: the ranks of `bench.py --gpus N` cold-start together and must not truncate each other's temporary file
    static std::atomic<unsigned> counter{0};
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(counter.fetch_add(1));
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;                                    // read-only model directory: run without a cache
    const bool ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}

}  // namespace rf
