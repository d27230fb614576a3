#!/usr/bin/env python3
"""Build the model/frame assets that travel with the repo (run in the dev container, where
/root/reference is mounted).

  assets/mnet-deconv-0517.rfw   graph + blobs of model/mnet-deconv-0517.* + this repo's int8 calibration of it:
                                assets/mnet-deconv-0517.cal.table.int8 / .cal.qweights.int8 (tools/calibrate_int8.py --per-channel --rule amax
                                --margin 1.25 --gptq, round 6).  The TensorRT table the reference ships stays beside it, untouched, as
                                assets/mnet-deconv-0517.table.int8 (per tensor; the engine runs it too: tests/test_gpu_parity.py)
  assets/mnet25.rfw             same for model/mnet25.* with assets/mnet25.table.int8 / mnet25.qweights.int8 (the reference ships a table
                                for 0517 only)
  assets/faces_1280x886.png     lossless copy of the decoded pixels of data/img.jpg (the reference's only image fixture)

The .rfw container is this repo's own format (oracle/caffe_io.py, retinaface_amd/csrc/model.cpp):
it is what the reference's "<name>.cache" engine cache is to TensorRT (trtnetbase.cpp:205-243) --
a derived artefact the engine loads instead of re-parsing prototxt + caffemodel.
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.caffe_io import load_caffe_model, read_int8_qweights, read_rfw, write_rfw  # noqa: E402

REF = os.environ.get("RF_REFERENCE", "/root/reference")


def main():
    out = os.path.join(ROOT, "assets")
    os.makedirs(out, exist_ok=True)
    shipped = os.path.join(REF, "model", "mnet-deconv-0517.table.int8")
    for stem in ("mnet-deconv-0517", "mnet25"):
        # 0517: the TensorRT table the reference ships.  mnet25: the reference ships none; assets/mnet25.table.int8 is produced
        # by tools/calibrate_int8.py (same text format) -- without it the 0517 table is attached as an approximation.
        base = os.path.join(out, stem + (".cal" if stem == "mnet-deconv-0517" else ""))
        table = base + ".table.int8" if os.path.exists(base + ".table.int8") else shipped
        net = load_caffe_model(os.path.join(REF, "model", stem + ".prototxt"),
                               os.path.join(REF, "model", stem + ".caffemodel"), table)
        if table != shipped and os.path.exists(base + ".qweights.int8"):
            net.int8_qweights = read_int8_qweights(base + ".qweights.int8")
        print(stem, "int8 table:", table, "calibrated weights:", len(net.int8_qweights), "fused convs")
        dst = os.path.join(out, stem + ".rfw")
        write_rfw(net, dst)
        back = read_rfw(dst)
        assert len(back.layers) == len(net.layers)
        for a, b in zip(net.layers, back.layers):
            assert a.name == b.name and len(a.blobs) == len(b.blobs)
            for x, y in zip(a.blobs, b.blobs):
                assert x.shape == y.shape and np.array_equal(x, y)
        assert set(back.int8_qweights) == set(net.int8_qweights)
        print("wrote", dst, os.path.getsize(dst), "bytes")
    img = Image.open(os.path.join(REF, "data", "img.jpg")).convert("RGB")
    dst = os.path.join(out, "faces_1280x886.png")
    img.save(dst, optimize=True)
    assert np.array_equal(np.array(Image.open(dst)), np.array(img))
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
