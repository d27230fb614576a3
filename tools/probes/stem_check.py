"""Probe: where does the fp16 stem's output (mobilenet0_relu2_fwd) differ from the oracle blob?  Prints the error map by
8x32 output tile so tile-walk / border / staging bugs show up as patterns."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import retinaface_amd
from oracle.caffe_io import read_rfw
from oracle.pipeline import OracleDetector
from oracle.retinaface_post import preprocess_trt_identity
from retinaface_amd.frames import padded_base_frame
frame = padded_base_frame(); crop = np.ascontiguousarray(frame[30:478, 440:888])
od = OracleDetector(read_rfw(os.path.join(ROOT, "assets", "mnet-deconv-0517.rfw")))
blobs = od.forward(preprocess_trt_identity(crop, 448, 448), keep_all=True)
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=1, net_hw=(448, 448), keep_outputs=True, use_graph=False)
det.detect(crop, 0.5)
a = det.debug_activation("mobilenet0_relu2_fwd"); r = blobs["mobilenet0_relu2_fwd"][0].transpose(1, 2, 0)
d = np.abs(a - r).max(axis=2)
print("max err", d.max(), "mean", d.mean(), "scale", np.abs(r).max())
t = d.reshape(28, 8, 7, 32).max(axis=(1, 3))
np.set_printoptions(precision=2, linewidth=200, suppress=True)
print("per-tile max error (28 tile rows x 7 tile cols):"); print(t)
bad = np.argwhere(d > 0.05)
print("bad pixels:", len(bad), "rows", np.unique(bad[:, 0])[:40], "cols", np.unique(bad[:, 1])[:60])
