#!/usr/bin/env python3
"""Mint tests/golden/mxnet_pin.npz from the reference's MXNet originals (run in the dev container, where /root/reference
exists): the outputs of oracle/mxnet_forward.py -- the interpreter of MXNet2Caffe/model_mxnet/mnet.25-symbol.json +
mnet.25-0000.params -- on

  ones640    the all-ones 1x3x640x640 tensor of MXNet2Caffe/check_results.py:29 (its default --size, :47-48)
  crop448    the 448x448 crop (x0 = 440, y0 = 30) of data/img.jpg the other goldens use, RGB planes, raw 0..255

plus, for every parameter of the checkpoint, the Caffe layer / blob index it lands in under MXNet2Caffe/mxnet2caffe.py:42-113
and the SHA-256 of its bytes.  With this file the pin of oracle/caffe_forward.py against the MXNet definition also runs where
/root/reference is absent (the GPU box): tests/test_mxnet_pin.py.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.caffe_io import read_rfw  # noqa: E402
from oracle.mxnet_forward import MXNetSymbol, caffe_blob_mapping  # noqa: E402
from oracle.retinaface_post import preprocess_trt_identity  # noqa: E402
from retinaface_amd.frames import padded_base_frame  # noqa: E402

REF = os.environ.get("RF_REFERENCE", "/root/reference")
MX = os.path.join(REF, "MXNet2Caffe", "model_mxnet")


def main():
    sym = MXNetSymbol(os.path.join(MX, "mnet.25-symbol.json"), os.path.join(MX, "mnet.25-0000.params"))
    caffe_names = [l.name for l in read_rfw(os.path.join(ROOT, "assets", "mnet25.rfw")).layers]
    d = {}
    keys, layers, idxs, fixes, shas = [], [], [], [], []
    for key, kind, layer, idx, fix in caffe_blob_mapping(sym, caffe_names):
        arr = (sym.arg if kind == "arg" else sym.aux)[key]
        keys.append(key)
        layers.append(layer)
        idxs.append(idx)
        fixes.append(fix)
        shas.append(hashlib.sha256(np.ascontiguousarray(arr, dtype="<f4").tobytes()).hexdigest())
    d["map_key"], d["map_layer"], d["map_blob"] = np.array(keys), np.array(layers), np.array(idxs, np.int32)
    d["map_fix_gamma"], d["map_sha256"] = np.array(fixes), np.array(shas)

    frame = padded_base_frame()
    crop = np.ascontiguousarray(frame[30:478, 440:888])
    inputs = {"ones640": np.ones((1, 3, 640, 640), np.float32), "crop448": preprocess_trt_identity(crop, 448, 448)}
    for tag, x in inputs.items():
        out = sym.forward(x)
        for name, v in out.items():
            d[f"{tag}/{name}"] = v[0]
        print(tag, {k: v.shape for k, v in out.items()})
    d["outputs"] = np.array(sym.output_names())
    path = os.path.join(ROOT, "tests", "golden", "mxnet_pin.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path), "bytes;", len(keys), "parameters")


if __name__ == "__main__":
    main()
