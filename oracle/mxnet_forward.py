"""An interpreter of the reference's MXNet ORIGINAL of mnet25 -- the second, reference-held definition of the forward
(test infrastructure, see oracle/__init__.py).

The reference ships the network it converted its Caffe model from:

  MXNet2Caffe/model_mxnet/mnet.25-symbol.json     the graph (nnvm JSON, mxnet_version 10300)
  MXNet2Caffe/model_mxnet/mnet.25-0000.params     the weights (NDArray list container)
  MXNet2Caffe/mxnet2caffe.py:42-113               the blob mapping MXNet -> Caffe
  MXNet2Caffe/check_results.py:22-40              its own procedure for comparing the two nets (an all-ones tensor
                                                  through both)

MXNet itself is absent from this image, so this file restates, from the symbol file's own attributes, the published
inference semantics of the ten operators the graph uses.  It deliberately shares NO code with oracle/caffe_forward.py /
oracle/caffe_io.py (own container reader, own graph walk, own convolution written as shift-and-accumulate instead of
im2col, BatchNorm as MXNet's ONE op  y = (x - mean) * gamma / sqrt(var + eps) + beta  with the node's own ``eps`` and
``fix_gamma``), so agreement between the two is agreement between two definitions, not between two calls of one function.

  Convolution        NCHW, weight (O, C/g, kh, kw), out = floor((H + 2p - d(k-1) - 1)/s) + 1, bias unless no_bias
  BatchNorm          inference: moving statistics; fix_gamma -> gamma = 1; eps from the node (MXNet default 1e-3)
  Activation         act_type relu
  Concat             along ``dim``
  Reshape            codes 0 (copy the dim) and -1 (infer)
  SoftmaxActivation  mode=channel: softmax over axis 1
  UpSampling         sample_type=nearest, integer scale: every pixel repeated scale x scale
  Crop               num_args=2: the first input cropped to the second's H x W at offset (0, 0)
  elemwise_add

tests/test_mxnet_pin.py holds (i) every Caffe blob equal to its MXNet array under the mapping, (ii) the stride-32 outputs of
oracle/caffe_forward.py equal to this interpreter's to fp32 round-off, (iii) strides 16 / 8 equal once the Caffe graph's
bilinear deconvolution is swapped for nearest x2 -- and reports how far the shipped bilinear graph is from the MXNet one.
"""
from __future__ import annotations

import ast
import json
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

_LIST_MAGIC = 0x112                   # kMXAPINDArrayListMagic
_ND_V1, _ND_V2, _ND_V3 = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}


class _Cursor:
    def __init__(self, buf: bytes):
        self.buf, self.pos = buf, 0

    def take(self, fmt: str):
        size = struct.calcsize(fmt)
        if self.pos + size > len(self.buf):
            raise ValueError("truncated .params file")
        v = struct.unpack_from(fmt, self.buf, self.pos)
        self.pos += size
        return v[0] if len(v) == 1 else v

    def raw(self, n: int) -> bytes:
        if self.pos + n > len(self.buf):
            raise ValueError("truncated .params file")
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b


def _read_ndarray(c: _Cursor) -> np.ndarray:
    magic = c.take("<I")
    if magic in (_ND_V2, _ND_V3):
        stype = c.take("<i")
        if stype != 0:
            raise ValueError(f"sparse NDArray (storage type {stype}) not supported")
        ndim = c.take("<I")
        shape = tuple(c.take("<q") for _ in range(ndim))
    elif magic == _ND_V1:
        ndim = c.take("<I")
        shape = tuple(c.take("<q") for _ in range(ndim))
    else:                             # legacy: the first word IS ndim, dims are uint32
        ndim = magic
        if ndim > 8:
            raise ValueError(f"not an NDArray (leading word {magic:#x})")
        shape = tuple(c.take("<I") for _ in range(ndim))
    if ndim == 0:
        return np.zeros((0,), np.float32)
    c.take("<i")                      # context: device type
    c.take("<i")                      # context: device id
    flag = c.take("<i")
    if flag not in _DTYPES:
        raise ValueError(f"unknown dtype flag {flag}")
    dt = np.dtype(_DTYPES[flag])
    n = int(np.prod(shape, dtype=np.int64))
    return np.frombuffer(c.raw(n * dt.itemsize), dtype=dt.newbyteorder("<")).reshape(shape).astype(dt)


def read_params(path: str) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """MXNet's ``mx.nd.save`` list container -> (arg_params, aux_params), split on the ``arg:`` / ``aux:`` key prefixes
    exactly as ``mx.model.load_checkpoint`` does (the call at MXNet2Caffe/mxnet2caffe.py:17)."""
    with open(path, "rb") as f:
        c = _Cursor(f.read())
    magic, _reserved = c.take("<Q"), c.take("<Q")
    if magic != _LIST_MAGIC:
        raise ValueError(f"not an NDArray list file (magic {magic:#x})")
    arrays = [_read_ndarray(c) for _ in range(c.take("<Q"))]
    names = []
    for _ in range(c.take("<Q")):
        names.append(c.raw(c.take("<Q")).decode("utf-8"))
    if len(names) != len(arrays):
        raise ValueError("name / array count mismatch")
    if c.pos != len(c.buf):
        raise ValueError("trailing bytes after the name table")
    arg, aux = {}, {}
    for k, v in zip(names, arrays):
        kind, _, name = k.partition(":")
        if kind == "arg":
            arg[name] = v
        elif kind == "aux":
            aux[name] = v
        else:
            raise ValueError(f"key without arg:/aux: prefix: {k}")
    return arg, aux


def _tuple(s: str) -> Tuple[int, ...]:
    v = ast.literal_eval(s.replace("L", ""))
    return tuple(int(x) for x in v) if isinstance(v, (tuple, list)) else (int(v),)


def _bool(s) -> bool:
    return str(s).strip().lower() in ("true", "1")


def _conv_shift_accumulate(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], stride, pad, dilate, group: int):
    """y[n,o,i,j] = sum_{c,ky,kx} w[o,c,ky,kx] * xpad[n, g(o)*Cg + c, i*s + ky*d, j*s + kx*d]  -- one tap at a time:
    for each (ky, kx) the strided view of the padded input is contracted with the O x Cg slice of the weights."""
    n, c, h, wd = x.shape
    o, cg, kh, kw = w.shape
    if c != cg * group or o % group:
        raise ValueError(f"Convolution: input {x.shape} vs weight {w.shape} group {group}")
    (sy, sx), (py, px), (dy, dx) = stride, pad, dilate
    ho = (h + 2 * py - dy * (kh - 1) - 1) // sy + 1
    wo = (wd + 2 * px - dx * (kw - 1) - 1) // sx + 1
    xp = np.zeros((n, c, h + 2 * py, wd + 2 * px), dtype=x.dtype)
    xp[:, :, py:py + h, px:px + wd] = x
    y = np.zeros((n, o, ho, wo), dtype=x.dtype)
    og = o // group
    for ky in range(kh):
        for kx in range(kw):
            v = xp[:, :, ky * dy:ky * dy + sy * (ho - 1) + 1:sy, kx * dx:kx * dx + sx * (wo - 1) + 1:sx]
            if group == 1:
                y += np.einsum("oc,nchw->nohw", w[:, :, ky, kx], v, optimize=True)
            elif cg == 1 and og == 1:                 # depthwise
                y += v * w[:, 0, ky, kx].reshape(1, -1, 1, 1)
            else:
                for g in range(group):
                    y[:, g * og:(g + 1) * og] += np.einsum("oc,nchw->nohw", w[g * og:(g + 1) * og, :, ky, kx],
                                                          v[:, g * cg:(g + 1) * cg], optimize=True)
    if b is not None:
        y += b.reshape(1, -1, 1, 1)
    return y


class MXNetSymbol:
    """The graph of a ``*-symbol.json`` + its parameters, evaluated node by node in the file's (topological) order."""

    def __init__(self, symbol_json: str, params: str):
        with open(symbol_json) as f:
            g = json.load(f)
        self.nodes: List[dict] = g["nodes"]
        self.heads: List[int] = [h[0] for h in g["heads"]]
        self.arg, self.aux = read_params(params)
        self.version = (g.get("attrs", {}).get("mxnet_version") or [None, None])[1]

    # ---- introspection (used by the mapping test)
    def attrs(self, node: dict) -> dict:
        return node.get("attrs") or node.get("attr") or node.get("param") or {}

    def node(self, name: str) -> dict:
        for nd in self.nodes:
            if nd["name"] == name:
                return nd
        raise KeyError(name)

    def op_counts(self) -> Dict[str, int]:
        out: Dict[str, int] = {}
        for nd in self.nodes:
            out[nd["op"]] = out.get(nd["op"], 0) + 1
        return out

    def output_names(self) -> List[str]:
        return [self.nodes[i]["name"] for i in self.heads]

    # ---- evaluation
    def forward(self, data: np.ndarray, keep: Optional[Iterable[str]] = None, keep_all: bool = False,
                dtype=np.float32) -> Dict[str, np.ndarray]:
        """data: NCHW.  Returns {node name: value} for the heads (+ `keep`, or every op node if keep_all).  ``dtype=float64``
        evaluates the same graph in double precision (the yardstick both fp32 evaluations are measured against)."""
        keep = set(keep or ())
        vals: List[Optional[np.ndarray]] = [None] * len(self.nodes)
        out: Dict[str, np.ndarray] = {}
        last_use = {}
        for i, nd in enumerate(self.nodes):
            for src in nd["inputs"]:
                last_use[src[0]] = i
        for i, nd in enumerate(self.nodes):
            name, op = nd["name"], nd["op"]
            if op == "null":
                if name == "data":
                    vals[i] = np.asarray(data, dtype=dtype)
                elif name in self.arg:
                    vals[i] = self.arg[name].astype(dtype)
                elif name in self.aux:
                    vals[i] = self.aux[name].astype(dtype)
                else:
                    raise KeyError(f"no parameter for variable {name}")
                continue
            ins = [vals[s[0]] for s in nd["inputs"]]
            v = self._op(op, self.attrs(nd), ins, dtype)
            vals[i] = v
            if keep_all or name in keep or i in self.heads:
                out[name] = v
            if not keep_all:
                for s in nd["inputs"]:
                    j = s[0]
                    if last_use.get(j) == i and j not in self.heads and self.nodes[j]["op"] != "null":
                        vals[j] = None
        return out

    @staticmethod
    def _op(op: str, a: dict, ins: List[np.ndarray], dtype) -> np.ndarray:
        if op == "Convolution":
            k = _tuple(a["kernel"])
            stride = _tuple(a.get("stride", "(1, 1)"))
            pad = _tuple(a.get("pad", "(0, 0)"))
            dil = _tuple(a.get("dilate", "(1, 1)"))
            group = int(a.get("num_group", 1))
            no_bias = _bool(a.get("no_bias", "False"))
            w = ins[1]
            if w.shape[0] != int(a["num_filter"]) or tuple(w.shape[2:]) != k:
                raise ValueError(f"Convolution weight {w.shape} vs num_filter {a['num_filter']} kernel {k}")
            if a.get("layout", "NCHW") not in ("NCHW", "None"):
                raise ValueError("layout " + a["layout"])
            b = None if no_bias else ins[2]
            return _conv_shift_accumulate(ins[0], w, b, stride, pad, dil, group)
        if op == "BatchNorm":
            x, gamma, beta, mean, var = ins
            eps = dtype(float(a.get("eps", 1e-3)))
            if int(a.get("axis", 1)) != 1:
                raise ValueError("BatchNorm axis")
            if _bool(a.get("fix_gamma", "True")):
                gamma = np.ones_like(gamma)
            sh = (1, -1, 1, 1)
            inv = (dtype(1.0) / np.sqrt(var + eps)).astype(dtype)
            return ((x - mean.reshape(sh)) * (gamma * inv).reshape(sh) + beta.reshape(sh)).astype(dtype)
        if op == "Activation":
            if a["act_type"] != "relu":
                raise ValueError("act_type " + a["act_type"])
            return np.maximum(ins[0], dtype(0))
        if op == "Concat":
            if int(a["num_args"]) != len(ins):
                raise ValueError("Concat num_args")
            return np.concatenate(ins, axis=int(a.get("dim", 1)))
        if op == "Reshape":
            x = ins[0]
            spec = _tuple(a["shape"])
            shape, known = [], 1
            for i, d in enumerate(spec):
                if d == 0:
                    d = x.shape[i]
                elif d < -1:
                    raise ValueError(f"Reshape code {d} not used by this graph")
                shape.append(d)
                if d > 0:
                    known *= d
            if -1 in shape:
                shape[shape.index(-1)] = x.size // known
            return x.reshape(shape)
        if op == "SoftmaxActivation":
            if a.get("mode", "instance") != "channel":
                raise ValueError("SoftmaxActivation mode " + a.get("mode", "instance"))
            x = ins[0]
            e = np.exp(x - x.max(axis=1, keepdims=True))
            return (e / e.sum(axis=1, keepdims=True)).astype(dtype)
        if op == "UpSampling":
            if a["sample_type"] != "nearest" or int(a.get("num_args", 1)) != 1:
                raise ValueError("UpSampling " + str(a))
            s = int(a["scale"])
            return np.repeat(np.repeat(ins[0], s, axis=2), s, axis=3)
        if op == "Crop":
            if int(a["num_args"]) != 2 or _bool(a.get("center_crop", "False")):
                raise ValueError("Crop " + str(a))
            oy, ox = _tuple(a.get("offset", "(0, 0)"))
            h, w = ins[1].shape[2:]
            if oy + h > ins[0].shape[2] or ox + w > ins[0].shape[3]:
                raise ValueError("Crop larger than its input")
            return np.ascontiguousarray(ins[0][:, :, oy:oy + h, ox:ox + w])
        if op == "elemwise_add":
            return ins[0] + ins[1]
        raise ValueError(f"operator {op} is not part of mnet.25-symbol.json")


def caffe_blob_mapping(sym: MXNetSymbol, caffe_layer_names: Iterable[str]):
    """The mapping of MXNet2Caffe/mxnet2caffe.py:42-113, restated as data: yields (mxnet key, 'arg' | 'aux', caffe layer,
    blob index, fix_gamma) for every parameter of the checkpoint, in the converter's sorted order.  ``fix_gamma`` (only
    for *_gamma keys, `:55-70`) means the converter wrote ones instead of the array."""
    names = set(caffe_layer_names)
    keys = sorted(list(sym.arg) + list(sym.aux))
    for key in keys:
        if key == "data":
            continue
        if "_weight" in key:                                  # :46-54
            layer = key.replace("_weight", "")
            if layer not in names:
                layer += "_fwd"
            yield key, "arg", layer, 0, False
        elif "_bias" in key:                                  # :55-57
            yield key, "arg", key.replace("_bias", ""), 1, False
        elif "_gamma" in key and "relu" not in key:           # :58-74
            a = sym.attrs(sym.node(key))
            fix = str(a.get("fix_gamma", "")) == "True"       # looked up on the VARIABLE node, as the converter does
            layer = key.replace("_gamma", "_scale")
            if layer not in names:
                layer = key.replace("_gamma", "_fwd_scale")
            yield key, "arg", layer, 0, fix
        elif "_beta" in key:                                  # :87-93
            layer = key.replace("_beta", "_scale")
            if layer not in names:
                layer = key.replace("_beta", "_fwd_scale")
            yield key, "arg", layer, 1, False
        elif "_moving_mean" in key:                           # :94-97
            yield key, "aux", key.replace("_moving_mean", ""), 0, False
        elif "_running_mean" in key:                          # :98-101
            yield key, "aux", key.replace("_running_mean", "_fwd"), 0, False
        elif "_moving_var" in key:                            # :102-105
            yield key, "aux", key.replace("_moving_var", ""), 1, False
        elif "_running_var" in key:                           # :106-109
            yield key, "aux", key.replace("_running_var", "_fwd"), 1, False
        else:
            raise ValueError(f"unknown MXNet key {key}")      # the converter exits here too (:110-111)
