"""Pins the oracle against the REFERENCE ITSELF.

Two layers, both CPU-only (they run in `-m "not gpu"`):
  * frozen vectors  tests/golden/ref_pin.npz, minted by tools/make_ref_golden.py from the reference's own
                    retinaface/RetinaFace.cpp compiled unmodified (oracle/build_ref.py) -- available everywhere;
  * live            the same comparisons against oracle/_ref/libretinaface_ref.so when it is present (built in the dev
                    container from /root/reference, travels prebuilt to the GPU box), on fresh seeds, plus the real
                    detect() / detectBatchImages() driven end to end with the oracle's forward as the "engine".
Everything here is bit-exact: same visiting order, same float rounding points, same libm expf.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import build as cbuild
from oracle import build_ref
from oracle import retinaface_post as post

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = np.load(os.path.join(ROOT, "tests", "golden", "ref_pin.npz"))
POST_THRESHOLDS = (0.5, 0.9, 0.1, 0.02)
NMS_THRESHOLDS = (0.3, 0.4, 0.6)

live = pytest.mark.skipif(not build_ref.available(), reason="oracle/_ref not built and /root/reference absent")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def heads_dict(heads9):
    return {n: np.asarray(a)[None] for n, a in zip(build_ref.HEAD_BLOBS, heads9)}


def oracle_post(heads9, net_h, net_w, thr, nms_thr=0.4):
    """python restatement and C restatement, which must agree with each other first"""
    c = post.decode(heads_dict(heads9), net_h, net_w, thr)
    py = post.nms(list(c), nms_thr)
    rows = np.stack([d.as_row() for d in py]) if py else np.zeros((0, 15), np.float32)
    _, _, kept, _ = cbuild.decode_nms(heads9, net_h, net_w, thr, nms_thr)
    assert np.array_equal(rows, kept.reshape(-1, 15))
    return rows


def rows_to_dets(rows):
    return [post.Detection(np.float32(r[0]), tuple(np.float32(v) for v in r[1:5]), list(r[5:10]), list(r[10:15]), i)
            for i, r in enumerate(rows)]


def oracle_nms(rows, thr):
    keep = post.nms(rows_to_dets(rows), thr)
    return np.stack([d.as_row() for d in keep]) if keep else np.zeros((0, 15), np.float32)


# ------------------------------------------------------------------------------------------------ frozen vectors

def test_pin_anchors():
    base = post.base_anchors()
    for lvl, s in enumerate((32, 16, 8)):
        a = post.anchors_plane(448 // s, 448 // s, s, base[s]).reshape(-1, 4).astype(np.float32)
        assert sha(a) == str(PIN[f"anchors448_s{s}_sha"])
        assert np.array_equal(a.astype(np.float64).sum(axis=0), PIN[f"anchors448_s{s}_sum"])
        assert np.array_equal(post.anchors_plane(5, 7, s, base[s]).reshape(-1, 4), PIN[f"anchors_plane_5x7_s{s}"])
        assert sha(cbuild.anchors_plane(lvl, 448 // s, 448 // s)) == str(PIN[f"anchors448_s{s}_sha"])


def test_pin_regression():
    for a, r, p, box, lm in zip(PIN["reg_anchors"], PIN["reg_deltas"], PIN["reg_pts"], PIN["reg_boxes"], PIN["reg_landmarks"]):
        assert np.array_equal(np.array(post.bbox_pred(a, r), np.float32), box)
        # reference FacePts order x[5],y[5]; the oracle helper takes interleaved x0,y0,.. like the blob
        inter = np.stack([p[:5], p[5:]], axis=1).reshape(-1)
        xs, ys = post.landmark_pred(a, inter)
        assert np.array_equal(np.array(xs + ys, np.float32), lm)


@pytest.mark.parametrize("case", ["nms40", "nms300", "nms12ties"])
def test_pin_nms(case):
    rows = PIN[f"{case}_in"]
    for t in NMS_THRESHOLDS:
        assert np.array_equal(oracle_nms(rows, t), PIN[f"{case}_out_{t}"]), (case, t)


@pytest.mark.parametrize("stem", ["mnet-deconv-0517", "mnet25"])
def test_pin_postprocess_crop448(stem):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"crop448_{stem}.npz"))
    heads = [g[n] for n in build_ref.HEAD_BLOBS]
    for t in POST_THRESHOLDS:
        want = PIN[f"crop448_{stem}_post_{t}"]
        assert np.array_equal(oracle_post(heads, 448, 448, t), want), t
    assert np.array_equal(PIN[f"crop448_{stem}_post_0.5"], g["det"])      # the oracle golden IS the reference's answer


@pytest.mark.parametrize("case", ["dense", "sparse"])
def test_pin_postprocess_random_heads(case):
    heads = [PIN[f"rand_{case}_{n}"] for n in build_ref.HEAD_BLOBS]
    sizes = []
    for t in POST_THRESHOLDS:
        want = PIN[f"rand_{case}_post_{t}"]
        sizes.append(len(want))
        assert np.array_equal(oracle_post(heads, 96, 128, t), want), t
    assert max(sizes) > (20 if case == "dense" else 2)               # the case is not vacuous


def test_pin_preprocess():
    small = PIN["pre_small_frame"]
    x = post.preprocess_trt_identity(small, 64, 96)
    assert np.array_equal(x[0], PIN["pre_small_input"])
    full = post.preprocess_trt_identity(PIN["pre_full_frame"], 64, 96)
    assert sha(np.concatenate([full, x])) == str(PIN["pre_batch_input_sha"])


# ------------------------------------------------------------------------------------------------ live reference

@live
def test_live_anchor_planes():
    ref = build_ref.ReferenceRetinaFace(896, 1280)
    base = post.base_anchors()
    try:
        for lvl, s in enumerate((32, 16, 8)):
            assert np.array_equal(ref.anchors(s), post.anchors_plane(896 // s, 1280 // s, s, base[s]).reshape(-1, 4))
            for h, w in ((1, 1), (3, 11), (17, 2)):
                assert np.array_equal(ref.anchors_plane(h, w, lvl), post.anchors_plane(h, w, s, base[s]).reshape(-1, 4))
    finally:
        ref.close()


@live
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_live_postprocess_random(seed):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_ref_golden import random_faces, random_heads
    rng = np.random.default_rng(seed)
    net_h, net_w = 32 * int(rng.integers(2, 6)), 32 * int(rng.integers(2, 6))
    ref = build_ref.ReferenceRetinaFace(net_h, net_w)
    try:
        heads = random_heads(rng, net_h, net_w, float(rng.choice([0.5, 0.1, 0.01])))
        slot = int(rng.integers(0, 8))
        ref.set_heads(slot, heads)
        for t in (0.5, 0.2, 0.8):
            assert np.array_equal(ref.postprocess(slot, t), oracle_post(heads, net_h, net_w, t)), t
        faces = random_faces(rng, int(rng.integers(1, 400)), 448)
        for t in (0.2, 0.4, 0.7):
            assert np.array_equal(ref.nms(faces, t), oracle_nms(faces, t)), t
        assert len(ref.nms(np.zeros((0, 15), np.float32), 0.4)) == 0
    finally:
        ref.close()


@live
def test_live_detect_end_to_end():
    """The reference's real detect() and detectBatchImages(): its preprocess feeds the oracle's forward (standing in for
    the absent TensorRT engine), its decode + NMS read the result; oracle.pipeline must give the same rows, bit for bit."""
    from oracle.caffe_forward import HEAD_STRIDES, head_names
    from oracle.caffe_io import read_rfw
    from oracle.pipeline import OracleDetector
    from retinaface_amd.frames import synth_frames

    od = OracleDetector(read_rfw(os.path.join(ROOT, "assets", "mnet-deconv-0517.rfw")))
    names = [n for s in HEAD_STRIDES for n in head_names(s)]
    assert names == build_ref.HEAD_BLOBS

    def forward(x):
        outs = []
        for i in range(x.shape[0]):
            b = od.forward(x[i:i + 1])
            outs.append([b[n][0] for n in names])
        return outs

    H = W = 448
    frames = synth_frames(H, W, 3, 1)
    frames[2] = np.ascontiguousarray(frames[2][:400, :380])          # smaller than the net: zero padded bottom/right
    ref = build_ref.ReferenceRetinaFace(H, W, forward)
    try:
        for f in frames[:2]:
            got = ref.detect(f, 0.5)
            assert np.array_equal(ref.last_input(), post.preprocess_trt_identity(f, H, W))
            want = od.detect(f, 0.5, 0.4, net_hw=(H, W))
            assert len(got) > 0 and np.array_equal(got, want.rows())
        per_image = ref.detect_batch(frames, 0.3)
        assert ref.forward_calls == 3                                 # one engine call for the whole batch
        x = ref.last_input()
        for i, f in enumerate(frames):
            assert np.array_equal(x[i:i + 1], post.preprocess_trt_identity(f, H, W))
            assert np.array_equal(per_image[i], od.detect(f, 0.3, 0.4, net_hw=(H, W)).rows())
    finally:
        ref.close()


# ------------------------------------------------------------------------------------------------ network presets

PRESETS = ("net3", "net3a", "ssh", "vgg", "net4", "net5", "net5a", "net6", "no-such-network")
# net3a's base anchors as the reference's constructor builds them (minted from oracle/_ref, kept here so the pin also holds
# where the reference build is absent): ratio 1.0 then ratio 1.5 (13 x 20 seed window), scales large then small
NET3A = {32: [[-248, -248, 263, 263], [-120, -120, 135, 135], [-200, -312, 215, 327], [-96, -152, 111, 167]],
         16: [[-56, -56, 71, 71], [-24, -24, 39, 39], [-44, -72, 59, 87], [-18, -32, 33, 47]],
         8: [[-8, -8, 23, 23], [0, 0, 15, 15], [-5, -12, 20, 27], [1.5, -2, 13.5, 17]]}


def _product_anchors(network, stride):
    import ctypes as C
    from retinaface_amd import _lib
    lib = _lib.load_library()
    out = np.zeros((8, 4), np.float32)
    n = lib.rf_preset_anchors(network.encode(), stride, out.ctypes.data_as(C.POINTER(C.c_float)), 8)
    assert n >= 0
    return out[:n]


@pytest.mark.parametrize("network", PRESETS)
def test_network_presets_product_vs_python_restatement(network):
    """rf_preset_anchors (the product's host code) == the numpy restatement of generate_anchors_fpn for the preset's ratios,
    bit for bit; presets without ratios / anchor configuration give zero anchors."""
    ratios = post.preset_ratios(network)
    base = post.base_anchors(ratios)
    for s in post.FEAT_STRIDES:
        got = _product_anchors(network, s)
        assert got.shape == (2 * len(ratios), 4) and np.array_equal(got, base[s]), (network, s, got, base[s])
    if network == "net3a":
        for s in post.FEAT_STRIDES:
            assert np.array_equal(base[s], np.array(NET3A[s], np.float32)), (s, base[s])


@live
@pytest.mark.parametrize("network", PRESETS)
def test_live_network_presets_against_the_reference_constructor(network):
    """The reference's own constructor run with each preset name (RetinaFace.cpp:205-271): its `_anchors_fpn` must equal the
    product's rf_preset_anchors; presets it leaves unconfigured end up with no strides or no anchors, and its postProcess then
    returns no faces whatever the blobs hold."""
    ratios = post.preset_ratios(network)
    a = max(2 * len(ratios), 2)
    ref = build_ref.ReferenceRetinaFace(64, 64, max_batch=1, network=network, head_anchors=a)
    try:
        fmc3 = network not in ("net4", "net5", "net5a", "net6")
        assert ref.num_levels() == (3 if fmc3 else 0)
        for s in post.FEAT_STRIDES:
            got = ref.base_anchors(s)
            assert got.shape[0] == (2 * len(ratios) if fmc3 else 0)
            assert np.array_equal(got, _product_anchors(network, s)), (network, s)
        # blobs full of confident foreground: decoded only where the preset has anchors
        rng = np.random.default_rng(5)
        heads = []
        for s in post.FEAT_STRIDES:
            h = 64 // s
            prob = rng.permutation(np.linspace(0.55, 0.99, 2 * a * h * h).astype(np.float32)).reshape(2 * a, h, h)    # distinct: std::sort ties are undefined
            heads += [prob, rng.normal(0, 0.2, (4 * a, h, h)).astype(np.float32), rng.normal(0, 0.2, (10 * a, h, h)).astype(np.float32)]
        ref.set_heads(0, heads)
        faces = ref.postprocess(0, 0.5)
        if ratios:
            hd = {n: x[None] for n, x in zip(build_ref.HEAD_BLOBS, heads)}
            want = post.nms(list(post.decode(hd, 64, 64, 0.5, ratios=ratios)), 0.4)
            assert len(faces) == len(want) > 0
            assert np.array_equal(faces, np.stack([d.as_row() for d in want]))
        else:
            assert len(faces) == 0
    finally:
        ref.close()


# ------------------------------------------------------------------------------------------------ oversize frames

def test_cv_resize_restatements_agree():
    """numpy twin == C restatement (oracle/csrc/cv_resize_linear.h, through ctypes via the C oracle library) on awkward sizes
    and factors: both restate OpenCV's published bilinear path (third party, absent here: parity unpinned), they must at least
    be the same function."""
    rng = np.random.default_rng(11)
    for rows, cols, f in ((37, 53, 0.5), (480, 640, 0.7), (1080, 1920, 448 / 1920), (901, 1283, 0.349), (64, 64, 0.9999), (5, 7, 0.31)):
        img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
        a = post.cv_resize_linear(img, f, f)
        b = cbuild.cv_resize_linear(img, f, f)
        assert a.shape == b.shape == (int(np.rint(rows * f)), int(np.rint(cols * f)), 3)
        assert np.array_equal(a, b), (rows, cols, f, int(np.abs(a.astype(int) - b).max()))
    # exact 2:1 of a constant image stays constant; a ramp stays monotone
    assert np.all(post.cv_resize_linear(np.full((40, 40, 3), 77, np.uint8), 0.5, 0.5) == 77)
    ramp = np.repeat(np.arange(200, dtype=np.uint8)[None, :, None], 3, axis=2).repeat(10, axis=0)
    assert np.all(np.diff(post.cv_resize_linear(ramp, 0.37, 0.37)[0, :, 0].astype(int)) >= 0)


@live
def test_live_reference_detect_on_oversize_frames():
    """The reference's own detect() (the build without NPP) on frames larger than the net: its scale / cv::resize / one-sided
    copyMakeBorder branch (RetinaFace.cpp:585-620) must hand the engine exactly the tensor preprocess_trt_cvresize builds.  (The
    interpolation itself is the shim's stand-in for OpenCV, the same restatement -- what this pins is the reference's branch.)"""
    rng = np.random.default_rng(3)
    ref = build_ref.ReferenceRetinaFace(448, 448, max_batch=1)
    try:
        for rows, cols in ((448, 896), (896, 448), (1080, 1920), (672, 672), (450, 449)):
            img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
            sw, sh = np.float32(cols) / np.float32(448), np.float32(rows) / np.float32(448)
            scale = max(sw, sh, np.float32(1))
            # the reference pads ONE side only and needs the other to land on the net size exactly: keep to frames where it does
            if int(np.rint(cols * np.float64(np.float32(1) / scale))) != 448 and sw > sh:
                continue
            if int(np.rint(rows * np.float64(np.float32(1) / scale))) != 448 and sh >= sw:
                continue
            ref.detect(img, 0.5)
            got = ref.last_input()
            want = post.preprocess_trt_cvresize(img, 448, 448)
            assert got.shape == want.shape and np.array_equal(got, want), (rows, cols)
    finally:
        ref.close()
