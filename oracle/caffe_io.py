"""Readers for the reference's model artefacts (test infrastructure, see oracle/__init__.py).

* ``read_caffemodel``  -- minimal protobuf wire reader for a Caffe ``NetParameter``
  (fields used: NetParameter.layer=100 -> LayerParameter{name=1,type=2,bottom=3,top=4,blobs=7};
  BlobProto{data=5 packed f32, shape=7{dim=1}, legacy num/channels/height/width=1..4}).
  These are the blobs the reference loads with ``Net_->CopyTrainedLayersFrom`` at
  retinaface/RetinaFace.cpp:312 and that TensorRT's caffe parser reads at
  retinaface/tensorrt/trtnetbase.cpp:262-266.
* ``read_prototxt``    -- protobuf text-format parser, enough for model/*.prototxt
  (reference: retinaface/RetinaFace.cpp:311, retinaface/tensorrt/trtnetbase.cpp:149-197).
* ``read_int8_table``  -- TensorRT calibration cache text (reference reader:
  retinaface/tensorrt/trtnetbase.cpp:31-44; format SURVEY.md App. B.7).
* ``NetSpec`` / ``write_rfw`` / ``read_rfw`` -- the repo's own packed model container
  (graph + blobs + int8 scales in one file, conv weights stored O-H-W-I), which is what
  travels to the GPU box in ``assets/``.  The C++ loader (retinaface_amd/csrc/model.cpp)
  reads/writes the same format; tests check the two agree byte for byte.
"""
from __future__ import annotations

import re
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# protobuf wire format
# --------------------------------------------------------------------------------------


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def _fields(buf: bytes):
    """Yield (field_number, wire_type, value) for one message; value is int or bytes."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _parse_blob(buf: bytes) -> np.ndarray:
    data = None
    shape: Optional[List[int]] = None
    legacy = {}
    loose: List[bytes] = []
    for fno, wt, v in _fields(buf):
        if fno == 5:
            if wt == 2:
                data = np.frombuffer(v, dtype="<f4")
            else:  # unpacked repeated float
                loose.append(v)
        elif fno == 7 and wt == 2:
            dims: List[int] = []
            for f2, w2, v2 in _fields(v):
                if f2 == 1:
                    if w2 == 2:
                        p = 0
                        while p < len(v2):
                            d, p = _varint(v2, p)
                            dims.append(d)
                    else:
                        dims.append(v2)
            shape = dims
        elif fno in (1, 2, 3, 4) and wt == 0:
            legacy[fno] = v
    if data is None:
        data = np.frombuffer(b"".join(loose), dtype="<f4")
    if shape is None:
        shape = [legacy.get(i, 1) for i in (1, 2, 3, 4)]
    return np.array(data, dtype=np.float32).reshape(shape)


def read_caffemodel(path: str) -> Dict[str, List[np.ndarray]]:
    """Return {layer name: [blob, ...]} for every layer that carries blobs."""
    with open(path, "rb") as f:
        buf = f.read()
    out: Dict[str, List[np.ndarray]] = {}
    for fno, wt, v in _fields(buf):
        if fno != 100 or wt != 2:
            continue
        name = None
        blobs: List[np.ndarray] = []
        for f2, w2, v2 in _fields(v):
            if f2 == 1 and w2 == 2:
                name = v2.decode("utf-8")
            elif f2 == 7 and w2 == 2:
                blobs.append(_parse_blob(v2))
        if name is not None and blobs:
            out[name] = blobs
    return out


# --------------------------------------------------------------------------------------
# protobuf text format (prototxt)
# --------------------------------------------------------------------------------------

_TOKEN = re.compile(r'\s*(?:(#[^\n]*)|("(?:[^"\\]|\\.)*")|([{}:])|([^\s{}:"#]+))')


def _tokenize(text: str) -> List[str]:
    toks: List[str] = []
    pos = 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ValueError(f"prototxt: cannot tokenize at offset {pos}")
        pos = m.end()
        if m.group(1) is not None:
            continue
        toks.append(m.group(2) or m.group(3) or m.group(4))
    return toks


def _scalar(tok: str):
    if tok.startswith('"'):
        return tok[1:-1]
    if tok in ("true", "false"):
        return tok == "true"
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok  # enum identifier


def _parse_message(toks: List[str], pos: int, closing: bool) -> Tuple[Dict[str, list], int]:
    msg: Dict[str, list] = {}
    while pos < len(toks):
        t = toks[pos]
        if t == "}":
            if not closing:
                raise ValueError("prototxt: unbalanced '}'")
            return msg, pos + 1
        key = t
        pos += 1
        if toks[pos] == ":":
            pos += 1
        if toks[pos] == "{":
            sub, pos = _parse_message(toks, pos + 1, True)
            msg.setdefault(key, []).append(sub)
        else:
            msg.setdefault(key, []).append(_scalar(toks[pos]))
            pos += 1
    if closing:
        raise ValueError("prototxt: missing '}'")
    return msg, pos


def parse_text_format(text: str) -> Dict[str, list]:
    msg, _ = _parse_message(_tokenize(text), 0, False)
    return msg


# --------------------------------------------------------------------------------------
# Graph container
# --------------------------------------------------------------------------------------


@dataclass
class LayerSpec:
    name: str
    type: str
    bottoms: List[str]
    tops: List[str]
    # convolution / deconvolution
    num_output: int = 0
    kernel: int = 0
    stride: int = 1
    pad: int = 0
    group: int = 1
    bias_term: bool = True
    # batchnorm
    eps: float = 0.0
    # scale
    scale_bias: bool = False
    # concat / softmax / crop
    axis: int = 1
    crop_offsets: List[int] = field(default_factory=list)
    # reshape
    reshape_dims: List[int] = field(default_factory=list)
    reshape_axis: int = 0
    reshape_num_axes: int = -1
    # eltwise
    eltwise_op: str = "SUM"
    blobs: List[np.ndarray] = field(default_factory=list)


@dataclass
class NetSpec:
    name: str
    input_name: str
    input_shape: Tuple[int, int, int, int]   # N, C, H, W as written in the prototxt
    layers: List[LayerSpec]
    int8_scales: Dict[str, float] = field(default_factory=dict)
    # calibrated int8 weights per fused dense conv: fused-op name ('+'-joined reference layer names) -> (q int8 [cout, ktot] in the
    # K order (ky, kx, c), bias_delta float32 [cout]); written by tools/calibrate_int8.py --gptq, empty = round to nearest
    int8_qweights: Dict[str, Tuple[np.ndarray, np.ndarray]] = field(default_factory=dict)

    def layer(self, name: str) -> LayerSpec:
        for l in self.layers:
            if l.name == name:
                return l
        raise KeyError(name)


def _first(d: dict, key: str, default=None):
    v = d.get(key)
    return v[0] if v else default


def read_prototxt(path: str) -> NetSpec:
    with open(path, "r") as f:
        msg = parse_text_format(f.read())
    layers: List[LayerSpec] = []
    input_name = "data"
    input_shape = (1, 3, 0, 0)
    for l in msg.get("layer", []):
        name = _first(l, "name")
        typ = _first(l, "type")
        bottoms = list(l.get("bottom", []))
        tops = list(l.get("top", []))
        if typ == "Input":
            input_name = tops[0]
            dims = _first(_first(l, "input_param", {}), "shape", {}).get("dim", [])
            input_shape = tuple(int(d) for d in dims)
            continue
        spec = LayerSpec(name=name, type=typ, bottoms=bottoms, tops=tops)
        if typ in ("Convolution", "Deconvolution"):
            cp = _first(l, "convolution_param", {})
            spec.num_output = int(_first(cp, "num_output"))
            spec.kernel = int(_first(cp, "kernel_size"))
            spec.stride = int(_first(cp, "stride", 1))
            spec.pad = int(_first(cp, "pad", 0))
            spec.group = int(_first(cp, "group", 1))
            spec.bias_term = bool(_first(cp, "bias_term", True))  # caffe.proto default = true
        elif typ == "BatchNorm":
            bp = _first(l, "batch_norm_param", {})
            spec.eps = float(_first(bp, "eps", 1e-5))
        elif typ == "Scale":
            sp = _first(l, "scale_param", {})
            spec.scale_bias = bool(_first(sp, "bias_term", False))
        elif typ == "Concat":
            cp = _first(l, "concat_param", {})
            spec.axis = int(_first(cp, "axis", 1))
        elif typ == "Softmax":
            sp = _first(l, "softmax_param", {})
            spec.axis = int(_first(sp, "axis", 1))
        elif typ == "Crop":
            cp = _first(l, "crop_param", {})
            spec.axis = int(_first(cp, "axis", 2))
            spec.crop_offsets = [int(o) for o in cp.get("offset", [])]
        elif typ == "Reshape":
            rp = _first(l, "reshape_param", {})
            spec.reshape_dims = [int(d) for d in _first(rp, "shape", {}).get("dim", [])]
            spec.reshape_axis = int(_first(rp, "axis", 0))
            spec.reshape_num_axes = int(_first(rp, "num_axes", -1))
        elif typ == "Eltwise":
            ep = _first(l, "eltwise_param", {})
            spec.eltwise_op = str(_first(ep, "operation", "SUM"))
        elif typ == "ReLU":
            pass
        else:
            raise ValueError(f"prototxt: unsupported layer type {typ!r} ({name})")
        layers.append(spec)
    return NetSpec(name=str(_first(msg, "name", "")), input_name=input_name,
                   input_shape=input_shape, layers=layers)


def read_int8_table(path: str) -> Dict[str, float]:
    """tensor name -> f32 scale (real ~= q * scale, q in [-127,127])."""
    scales: Dict[str, float] = {}
    with open(path, "r") as f:
        lines = f.read().splitlines()
    if not lines or not lines[0].startswith("TRT-"):
        raise ValueError("not a TensorRT calibration cache")
    for line in lines[1:]:
        if not line.strip():
            continue
        name, _, hexv = line.rpartition(": ")
        scales[name] = struct.unpack(">f", bytes.fromhex(hexv.strip()))[0]
    return scales


def _pack_qweights(qw: Dict[str, Tuple[np.ndarray, np.ndarray]]) -> bytes:
    """u32 n, then per op: str name, u32 cout, u32 ktot, i8 q[cout * ktot], f32 bias_delta[cout]"""
    out = bytearray(struct.pack("<I", len(qw)))
    for name, (q, db) in qw.items():
        q = np.ascontiguousarray(q, np.int8)
        assert q.ndim == 2 and db.shape == (q.shape[0],) and q.min() >= -127
        _wstr(out, name)
        out += struct.pack("<II", q.shape[0], q.shape[1])
        out += q.tobytes() + np.ascontiguousarray(db, "<f4").tobytes()
    return bytes(out)


def _unpack_qweights(r: "_Reader") -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    qw: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
    for _ in range(r.u32()):
        name = r.str()
        cout, ktot = r.u32(), r.u32()
        q = np.frombuffer(r.buf, dtype=np.int8, count=cout * ktot, offset=r.pos).reshape(cout, ktot).copy()
        r.pos += cout * ktot
        qw[name] = (q, np.array(r.floats(cout), dtype=np.float32))
    return qw


def read_int8_qweights(path: str) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    """<stem>.qweights.int8 (RFQ1): the calibrated int8 weights that travel next to the activation table"""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:4] != b"RFQ1":
        raise ValueError(f"{path}: not an RFQ1 file")
    r = _Reader(buf)
    r.pos = 4
    qw = _unpack_qweights(r)
    if r.pos != len(buf):
        raise ValueError(f"{path}: trailing bytes")
    return qw


def write_int8_qweights(qw: Dict[str, Tuple[np.ndarray, np.ndarray]], path: str) -> None:
    with open(path, "wb") as f:
        f.write(b"RFQ1" + _pack_qweights(qw))


def load_caffe_model(prototxt: str, caffemodel: str, int8_table: Optional[str] = None) -> NetSpec:
    net = read_prototxt(prototxt)
    blobs = read_caffemodel(caffemodel)
    for l in net.layers:
        if l.name in blobs:
            l.blobs = blobs[l.name]
    if int8_table is not None:
        net.int8_scales = read_int8_table(int8_table)
    return net


# --------------------------------------------------------------------------------------
# RFW1: the repo's packed model container (little endian)
#
#   "RFW1" u32 version(=1)
#   str net_name, str input_name, u32 N,C,H,W
#   u32 n_layers, then per layer:
#       str name, str type, u32 nb, str bottoms[nb], u32 nt, str tops[nt]
#       i32 num_output, kernel, stride, pad, group, bias_term, axis, scale_bias,
#           reshape_axis, reshape_num_axes
#       f32 eps
#       str eltwise_op
#       u32 n_crop, i32 crop_offsets[]; u32 n_reshape, i32 reshape_dims[]
#       u32 n_blobs, per blob: u32 layout (0 = as-is, 1 = conv weight stored O,H,W,I),
#                              u32 ndim, u32 dims[ndim] (logical Caffe dims), f32 data[]
#   u32 n_scales, per scale: str tensor_name, f32 scale
#   [optional, round 6] u32 n_q, per fused dense conv: str op_name, u32 cout, u32 ktot, i8 q[cout*ktot], f32 bias_delta[cout]
#   str = u32 length + bytes
# --------------------------------------------------------------------------------------

RFW_MAGIC = b"RFW1"


def _wstr(out: bytearray, s: str) -> None:
    b = s.encode("utf-8")
    out += struct.pack("<I", len(b)) + b


def write_rfw(net: NetSpec, path: str) -> None:
    out = bytearray()
    out += RFW_MAGIC + struct.pack("<I", 1)
    _wstr(out, net.name)
    _wstr(out, net.input_name)
    out += struct.pack("<4I", *net.input_shape)
    out += struct.pack("<I", len(net.layers))
    for l in net.layers:
        _wstr(out, l.name)
        _wstr(out, l.type)
        out += struct.pack("<I", len(l.bottoms))
        for b in l.bottoms:
            _wstr(out, b)
        out += struct.pack("<I", len(l.tops))
        for t in l.tops:
            _wstr(out, t)
        out += struct.pack("<10i", l.num_output, l.kernel, l.stride, l.pad, l.group,
                           int(l.bias_term), l.axis, int(l.scale_bias), l.reshape_axis,
                           l.reshape_num_axes)
        out += struct.pack("<f", l.eps)
        _wstr(out, l.eltwise_op)
        out += struct.pack("<I", len(l.crop_offsets))
        out += struct.pack(f"<{len(l.crop_offsets)}i", *l.crop_offsets)
        out += struct.pack("<I", len(l.reshape_dims))
        out += struct.pack(f"<{len(l.reshape_dims)}i", *l.reshape_dims)
        out += struct.pack("<I", len(l.blobs))
        for bi, blob in enumerate(l.blobs):
            ohwi = l.type in ("Convolution", "Deconvolution") and bi == 0 and blob.ndim == 4
            out += struct.pack("<II", 1 if ohwi else 0, blob.ndim)
            out += struct.pack(f"<{blob.ndim}I", *blob.shape)
            data = blob.transpose(0, 2, 3, 1) if ohwi else blob
            out += np.ascontiguousarray(data, dtype="<f4").tobytes()
    out += struct.pack("<I", len(net.int8_scales))
    for k, v in net.int8_scales.items():
        _wstr(out, k)
        out += struct.pack("<f", v)
    if net.int8_qweights:
        out += _pack_qweights(net.int8_qweights)
    with open(path, "wb") as f:
        f.write(bytes(out))


class _Reader:
    def __init__(self, buf: bytes):
        self.buf = buf
        self.pos = 0

    def u32(self) -> int:
        v = struct.unpack_from("<I", self.buf, self.pos)[0]
        self.pos += 4
        return v

    def i32s(self, n: int) -> List[int]:
        v = list(struct.unpack_from(f"<{n}i", self.buf, self.pos))
        self.pos += 4 * n
        return v

    def f32(self) -> float:
        v = struct.unpack_from("<f", self.buf, self.pos)[0]
        self.pos += 4
        return v

    def str(self) -> str:
        n = self.u32()
        s = self.buf[self.pos:self.pos + n].decode("utf-8")
        self.pos += n
        return s

    def floats(self, n: int) -> np.ndarray:
        a = np.frombuffer(self.buf, dtype="<f4", count=n, offset=self.pos)
        self.pos += 4 * n
        return a


def read_rfw(path: str) -> NetSpec:
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:4] != RFW_MAGIC:
        raise ValueError(f"{path}: not an RFW1 file")
    r = _Reader(buf)
    r.pos = 4
    if r.u32() != 1:
        raise ValueError("unsupported RFW version")
    name = r.str()
    input_name = r.str()
    shape = tuple(r.u32() for _ in range(4))
    layers: List[LayerSpec] = []
    for _ in range(r.u32()):
        lname = r.str()
        ltype = r.str()
        bottoms = [r.str() for _ in range(r.u32())]
        tops = [r.str() for _ in range(r.u32())]
        (num_output, kernel, stride, pad, group, bias_term, axis, scale_bias,
         reshape_axis, reshape_num_axes) = r.i32s(10)
        eps = r.f32()
        elt = r.str()
        crop = r.i32s(r.u32())
        rdims = r.i32s(r.u32())
        blobs: List[np.ndarray] = []
        for _ in range(r.u32()):
            layout = r.u32()
            ndim = r.u32()
            dims = [r.u32() for _ in range(ndim)]
            data = np.array(r.floats(int(np.prod(dims)) if ndim else 1), dtype=np.float32)
            if layout == 1:
                o, i, h, w = dims
                data = data.reshape(o, h, w, i).transpose(0, 3, 1, 2)
            blobs.append(np.ascontiguousarray(data.reshape(dims)))
        layers.append(LayerSpec(name=lname, type=ltype, bottoms=bottoms, tops=tops,
                                num_output=num_output, kernel=kernel, stride=stride, pad=pad,
                                group=group, bias_term=bool(bias_term), eps=eps,
                                scale_bias=bool(scale_bias), axis=axis, crop_offsets=crop,
                                reshape_dims=rdims, reshape_axis=reshape_axis,
                                reshape_num_axes=reshape_num_axes, eltwise_op=elt, blobs=blobs))
    scales: Dict[str, float] = {}
    for _ in range(r.u32()):
        k = r.str()
        scales[k] = r.f32()
    qw = _unpack_qweights(r) if r.pos < len(buf) else {}
    return NetSpec(name=name, input_name=input_name, input_shape=shape, layers=layers,
                   int8_scales=scales, int8_qweights=qw)
