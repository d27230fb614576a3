"""Pin the CPU oracle (no GPU): readers vs the reference's files, two independent forwards, numpy vs C
post-processing, the reference's one image fixture, and the committed golden vectors."""
import os

import numpy as np
import pytest

from conftest import ASSETS, REFERENCE, ROOT, STEMS, golden, needs_reference
from oracle import build as obuild
from oracle.caffe_forward import CaffeNet, HEAD_STRIDES, head_names
from oracle.caffe_io import load_caffe_model, read_int8_table, read_rfw
from oracle.retinaface_post import (anchor_offsets, anchors_plane, base_anchors, decode, iou_plus1, nms,
                                    preprocess_caffe, preprocess_trt_identity)


@needs_reference
@pytest.mark.parametrize("stem", STEMS)
def test_rfw_assets_equal_reference_files(stem, nets):
    """assets/*.rfw must hold exactly what the prototxt + caffemodel + int8 table hold."""
    m = os.path.join(REFERENCE, "model")
    # int8 calibration: this repo's own for both models (tools/calibrate_int8.py; the reference ships a TensorRT table for 0517 only, kept
    # beside it as assets/mnet-deconv-0517.table.int8), as text + RFQ1 files in assets/
    assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
    base = os.path.join(assets, stem + (".cal" if stem == "mnet-deconv-0517" else ""))
    table = base + ".table.int8"
    ref = load_caffe_model(os.path.join(m, stem + ".prototxt"), os.path.join(m, stem + ".caffemodel"), table)
    got = nets[stem]
    assert got.input_shape == ref.input_shape and len(got.layers) == len(ref.layers) == 209
    for a, b in zip(ref.layers, got.layers):
        assert (a.name, a.type, a.bottoms, a.tops) == (b.name, b.type, b.bottoms, b.tops)
        assert (a.num_output, a.kernel, a.stride, a.pad, a.group, a.bias_term) == \
               (b.num_output, b.kernel, b.stride, b.pad, b.group, b.bias_term)
        assert np.float32(a.eps) == np.float32(b.eps) and a.reshape_dims == b.reshape_dims and a.crop_offsets == b.crop_offsets
        assert len(a.blobs) == len(b.blobs)
        for x, y in zip(a.blobs, b.blobs):
            assert x.shape == y.shape and np.array_equal(x, y)
    assert got.int8_scales == ref.int8_scales
    from oracle.caffe_io import read_int8_qweights
    loose = read_int8_qweights(base + ".qweights.int8")
    assert set(loose) == set(got.int8_qweights) and len(loose) == 29
    for k, (q, db) in loose.items():
        assert np.array_equal(q, got.int8_qweights[k][0]) and np.array_equal(db, got.int8_qweights[k][1]) and np.abs(q.astype(int)).max() <= 127


def test_model_inventory(nets):
    """56 Convolution, 47 BatchNorm/Scale, 41 ReLU, 2 Deconvolution, 435 999 parameters."""
    for stem in STEMS:
        net = nets[stem]
        kinds = {}
        for l in net.layers:
            kinds[l.type] = kinds.get(l.type, 0) + 1
        assert kinds == {"Convolution": 56, "BatchNorm": 47, "Scale": 47, "ReLU": 41, "Deconvolution": 2, "Crop": 2,
                         "Eltwise": 2, "Concat": 3, "Reshape": 6, "Softmax": 3}
        nparams = sum(b.size for l in net.layers for b in l.blobs)
        assert nparams == 435999
    assert nets["mnet-deconv-0517"].input_shape == (1, 3, 320, 320)
    assert nets["mnet25"].input_shape == (1, 3, 416, 288)


def test_int8_table(nets):
    """SURVEY.md App. B.7, the TensorRT cache the reference ships (kept verbatim as assets/mnet-deconv-0517.table.int8): data scale 2.00836
    (amax 255.06), 210 entries."""
    from conftest import ASSETS
    from oracle.caffe_io import read_int8_table
    sc = read_int8_table(os.path.join(ASSETS, "mnet-deconv-0517.table.int8"))
    assert len(sc) == 210
    assert abs(sc["data"] - 2.00836) < 1e-4 and abs(sc["data"] * 127 - 255.06) < 0.01
    assert abs(sc["mobilenet0_conv0_fwd"] - 11.55) < 0.01
    assert abs(sc["face_rpn_bbox_pred_stride8"] - 0.00399) < 1e-4


def test_upsample_weights_are_bilinear(nets):
    k = np.outer([.25, .75, .75, .25], [.25, .75, .75, .25]).astype(np.float32)
    for stem in STEMS:
        for name in ("rf_c3_upsampling", "rf_c2_upsampling"):
            w = nets[stem].layer(name).blobs[0]
            assert w.shape == (64, 1, 4, 4) and np.allclose(w, k[None, None], atol=1e-7)


def test_two_forward_backends_agree(nets):
    """torch (oneDNN) conv vs explicit numpy im2col / scatter-add deconv: every blob of a 64x96 input to fp32 round-off."""
    rng = np.random.default_rng(7)
    x = rng.integers(0, 256, size=(1, 3, 64, 96)).astype(np.float32)
    net = nets["mnet-deconv-0517"]
    a = CaffeNet(net, "torch").forward(x, keep_all=True)
    b = CaffeNet(net, "numpy").forward(x, keep_all=True)
    worst = 0.0
    for k, v in a.items():
        if k == "__by_layer__":
            continue
        scale = max(1.0, float(np.abs(v).max()))
        worst = max(worst, float(np.abs(v - b[k]).max()) / scale)
    assert worst < 2e-5, worst


def test_upsample_closed_form(nets):
    """out[2m] = .75 in[m] + .25 in[m-1]; out[2m+1] = .75 in[m] + .25 in[m+1], zero outside."""
    from oracle.caffe_forward import _deconv2d_numpy
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1, 64, 5, 7)).astype(np.float32)
    w = nets["mnet25"].layer("rf_c3_upsampling").blobs[0]
    y = _deconv2d_numpy(x, w, None, 2, 1, 64)
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    def up1(a, axis):
        n = a.shape[axis] - 2
        c = np.take(a, range(1, n + 1), axis)
        lo = np.take(a, range(0, n), axis)
        hi = np.take(a, range(2, n + 2), axis)
        ev, od = .75 * c + .25 * lo, .75 * c + .25 * hi
        out = np.stack([ev, od], axis=axis + 1)
        shp = list(c.shape); shp[axis] *= 2
        return out.reshape(shp)
    z = up1(xp, 2)
    z = up1(z, 3)
    assert y.shape == (1, 64, 10, 14) and np.allclose(y, z, atol=1e-6)


def test_base_anchors_match_survey():
    """SURVEY.md App. B.3."""
    b = base_anchors()
    assert b[32].tolist() == [[-248, -248, 263, 263], [-120, -120, 135, 135]]
    assert b[16].tolist() == [[-56, -56, 71, 71], [-24, -24, 39, 39]]
    assert b[8].tolist() == [[-8, -8, 23, 23], [0, 0, 15, 15]]
    for si, s in enumerate((32, 16, 8)):
        assert np.array_equal(anchors_plane(5, 7, s, b[s]), obuild.anchors_plane(si, 5, 7))
    offs = anchor_offsets(448, 448)
    assert (offs[32], offs[16], offs[8], offs["total"]) == (0, 392, 1960, 8232)
    assert anchor_offsets(896, 1280)["total"] == 47040


def _heads9(heads):
    return [heads[n][0] for s in HEAD_STRIDES for n in head_names(s)]


@pytest.mark.parametrize("thr", [0.5, 0.9, 0.02])
def test_numpy_and_c_postprocessing_agree(oracles, crop448, thr):
    """The two independent restatements of RetinaFace.cpp:666-724 + :434-492 give the same candidates / detections
    (coordinates to 1 ulp-ish: numpy's float32 exp vs glibc expf)."""
    r = oracles["mnet-deconv-0517"].detect(crop448, thr, 0.4, net_hw=(448, 448))
    cand, cidx, kept, kidx = obuild.decode_nms(_heads9(r.heads), 448, 448, thr, 0.4)
    assert cidx.tolist() == [d.anchor_index for d in r.candidates]
    assert kidx.tolist() == [d.anchor_index for d in r.detections]
    if len(r.candidates):
        a = np.stack([d.as_row() for d in r.candidates])
        assert np.allclose(a, cand, rtol=2e-6, atol=1e-4)


def test_nms_tie_order_and_strictness():
    """Ties broken by anchor index; suppression only when IoU > thr strictly (RetinaFace.cpp:486)."""
    from oracle.retinaface_post import Detection
    f = np.float32
    def mk(score, box, idx):
        return Detection(f(score), tuple(f(v) for v in box), [f(0)] * 5, [f(0)] * 5, idx)
    a = mk(0.9, (0, 0, 9, 9), 7)
    b = mk(0.9, (0, 0, 9, 9), 3)          # same score: lower anchor index wins
    c = mk(0.8, (100, 100, 109, 109), 1)
    out = nms([a, b, c], 0.4)
    assert [d.anchor_index for d in out] == [3, 1]
    # two boxes whose IoU equals the threshold exactly are both kept (strict >)
    d1 = mk(0.9, (0, 0, 9, 9), 0)
    d2 = mk(0.8, (0, 5, 9, 14), 1)        # inter 50, union 150 -> 1/3
    thr = float(np.float32(50) / np.float32(150))
    assert len(nms([d1, d2], thr)) == 2 and len(nms([d1, d2], thr * 0.999)) == 1
    assert abs(iou_plus1(d1.rect, d2.rect) - 1 / 3) < 1e-12


def test_decode_empty_and_all(oracles):
    heads = {}
    for s in HEAD_STRIDES:
        h = w = 64 // s
        heads[f"face_rpn_cls_prob_reshape_stride{s}"] = np.zeros((1, 4, h, w), np.float32)
        heads[f"face_rpn_bbox_pred_stride{s}"] = np.zeros((1, 8, h, w), np.float32)
        heads[f"face_rpn_landmark_pred_stride{s}"] = np.zeros((1, 20, h, w), np.float32)
    assert decode(heads, 64, 64, 0.5) == []
    for s in HEAD_STRIDES:
        heads[f"face_rpn_cls_prob_reshape_stride{s}"][:] = 1.0
    c = decode(heads, 64, 64, 0.5)
    assert len(c) == 2 * (4 + 16 + 64) and [d.anchor_index for d in c] == list(range(len(c)))
    # zero deltas reproduce the (clipped) anchor
    assert c[-1].rect == (np.float32(56), np.float32(56), np.float32(63), np.float32(63))


def test_preprocess_variants_agree_on_net_sized_frames(crop448):
    a, hs, ws = preprocess_caffe(crop448)
    b = preprocess_trt_identity(crop448, 448, 448)
    assert (hs, ws) == (448, 448) and np.array_equal(a, b)
    assert a[0, 0, 3, 5] == crop448[3, 5, 2] and a[0, 2, 3, 5] == crop448[3, 5, 0]      # BGR -> RGB, raw 0..255
    small = crop448[:100, :200]
    p = preprocess_trt_identity(small, 448, 448)
    assert p[:, :, 100:, :].max() == 0 and p[:, :, :, 200:].max() == 0


@pytest.mark.parametrize("stem", STEMS)
def test_fixture_image_matches_survey_and_golden(stem, oracles, base_frame):
    """The reference's only image: 6 faces, and the frozen golden vectors reproduce."""
    r = oracles[stem].detect(base_frame, 0.5, 0.4)
    g = golden(f"fixture_{stem}.npz")
    assert (r.net_h, r.net_w) == (896, 1280) and len(r.detections) == 6
    if stem == "mnet-deconv-0517":
        assert len(r.candidates) == 137
        d0 = r.detections[0]
        assert d0.anchor_index == 3872 and abs(float(d0.score) - 0.9986) < 1e-4
        assert np.allclose(d0.rect, [462.5, 268.0, 572.1, 416.2], atol=0.05)
        assert all(0.993 <= float(d.score) <= 0.9991 for d in r.detections)
    assert np.array_equal(r.anchor_indices(), g["det_idx"])
    assert np.array_equal(np.array([d.anchor_index for d in r.candidates], np.int32), g["cand_idx"])
    assert np.allclose(r.rows(), g["det"], rtol=1e-5, atol=2e-3)
    for s in (32, 16):
        for n in head_names(s):
            assert np.allclose(r.heads[n][0], g[n], atol=2e-5)
    for n in head_names(8):
        assert abs(float(r.heads[n].astype(np.float64).sum()) - float(g[n + "_sum"])) < 1e-3 * max(1.0, float(g[n + "_abs"]))


@pytest.mark.parametrize("stem", STEMS)
def test_synthetic_frames_golden(stem, oracles):
    from retinaface_amd.frames import synth_frames
    g = golden(f"synth448_{stem}.npz")
    frames = synth_frames(448, 448, 8, config=1)
    for i in (0, 3, 7):
        r = oracles[stem].detect(frames[i], 0.5, 0.4, net_hw=(448, 448))
        assert np.array_equal(r.anchor_indices(), g[f"idx05_{i}"])
        assert len(r.candidates) == int(g[f"ncand05_{i}"])
        assert np.allclose(r.rows(), g[f"det05_{i}"], rtol=1e-5, atol=2e-3)


# ---------------------------------------------------------------------------------------------------------------
# property tests: the two independent restatements of decode + NMS on random head tensors (ties, clipping, empties)
from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), thr=st.sampled_from([0.3, 0.5, 0.9]), nms_thr=st.sampled_from([0.2, 0.4, 0.7]),
       quantise=st.booleans())
def test_property_numpy_and_c_agree_on_random_heads(seed, thr, nms_thr, quantise):
    """Random probabilities / deltas on a 64x96 net (6 + 24 + 96 cells x 2 anchors).  `quantise` snaps scores to a coarse
    grid so equal scores (NMS tie order = anchor index) and exact-threshold scores (`conf <= thr` is dropped) occur."""
    rng = np.random.default_rng(seed)
    H, W = 64, 96
    heads = {}
    for s in HEAD_STRIDES:
        h, w = H // s, W // s
        p = rng.random((1, 4, h, w)).astype(np.float32)
        if quantise:
            p = (np.round(p * 10) / 10).astype(np.float32)
        heads[f"face_rpn_cls_prob_reshape_stride{s}"] = p
        heads[f"face_rpn_bbox_pred_stride{s}"] = rng.normal(0, 0.5, (1, 8, h, w)).astype(np.float32)
        heads[f"face_rpn_landmark_pred_stride{s}"] = rng.normal(0, 0.5, (1, 20, h, w)).astype(np.float32)
    cands = decode(heads, H, W, thr)
    kept = nms(list(cands), nms_thr)
    cand_c, cidx, kept_c, kidx = obuild.decode_nms(_heads9(heads), H, W, thr, nms_thr)
    assert cidx.tolist() == [d.anchor_index for d in cands]
    assert kidx.tolist() == [d.anchor_index for d in kept]
    if cands:
        a = np.stack([d.as_row() for d in cands])
        assert np.array_equal(a[:, 0], cand_c[:, 0])                       # scores are copied, never recomputed
        assert np.allclose(a, cand_c, rtol=3e-6, atol=2e-4)                # coordinates: numpy exp vs glibc expf
        assert (a[:, 1] >= 0).all() and (a[:, 2] >= 0).all() and (a[:, 3] <= W - 1).all() and (a[:, 4] <= H - 1).all()
    scores = [float(d.score) for d in kept]
    assert scores == sorted(scores, reverse=True)
    for i, a_ in enumerate(kept):                                          # survivors do not suppress each other
        for b_ in kept[i + 1:]:
            assert iou_plus1(a_.rect, b_.rect) <= nms_thr + 1e-6


def test_anchor_twin_band_admits_only_the_oracles_own_near_ties(oracles):
    """tests/anchor_twins.py -- the fp16 engine's anchor-SET band of the -m gpu suite -- on the CPU: the minted pairs of a golden frame equal what the live
    oracle derives; swapping a kept anchor for its minted twin is admitted (and held to the twin's oracle row), swapping it for any other candidate, keeping
    an anchor twice or dropping one is refused; a flat (slope-unaware) band would have admitted dozens of pairs on the saturated faces."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from anchor_twins import resolve, twins_of, twins_of_result
    from retinaface_amd.frames import synth_frames
    bands = np.load(os.path.join(ROOT, "tests", "golden", "threshold_bands.npz"))
    g = np.load(os.path.join(ROOT, "tests", "golden", "synth448_mnet25.npz"))
    noise = float(bands["score_noise"])
    minted = bands["mnet25/synth448_7/05/twins"]
    assert [(int(r[0]), int(r[1])) for r in minted] == [(609, 608), (904, 903)]
    frame = synth_frames(448, 448, 8, config=1)[7]
    ref = oracles["mnet25"].detect(frame, 0.5, 0.4, net_hw=(448, 448))
    live = twins_of_result(ref, 0.4, noise)
    assert np.array_equal(live, minted)
    rows, idx = g["det05_7"], g["idx05_7"].tolist()
    assert idx == ref.anchor_indices().tolist()
    # the engine keeps the twin: admitted, measured against the twin's oracle row, canonical order = the oracle's
    got = [608 if a == 609 else a for a in idx]
    held, swaps, canon = resolve(got, rows, idx, minted)
    cand = {d.anchor_index: d.as_row() for d in ref.candidates}
    assert swaps == 1 and canon == idx and np.array_equal(held[got.index(608)], cand[608]) and np.array_equal(held[0], rows[0])
    # anything else is refused
    for bad in ([607 if a == 609 else a for a in idx], idx[:-1], idx[:-1] + [idx[0]], [608 if a == 904 else a for a in idx]):
        with pytest.raises(AssertionError):
            resolve(bad, rows, idx, minted)
    with pytest.raises(AssertionError):
        resolve(got, rows, idx, None)                       # no band: identical sets only
    # the band is slope-aware: with the flat 2 * noise of rounds 1-4 every neighbour of a saturated face would be a "twin"
    crow = [d.as_row() for d in ref.candidates]
    cidx = [d.anchor_index for d in ref.candidates]
    flat = sum(1 for w_row, w_a in zip(rows, idx) for t_row, t_a in zip(crow, cidx)
               if t_a not in idx and abs(float(w_row[0]) - float(t_row[0])) <= 2 * noise)
    assert flat > 4 * len(minted)


def test_contract_golden_equals_a_live_oracle_run(oracles):
    """tests/golden/contract_oracle_1280x896.npz (tests/oracle_cache.py, minted by tools/make_contract_golden.py) serves the GPU contract tests
    the oracle's results on their 1280 x 896 frames: re-derive entries live -- one per model and face set -- and require the same detections,
    candidates, anchor indices and candidate-count band; a frame whose pixels differ from the minted one is NOT served from the file."""
    import oracle_cache
    from int8_contract import HELD_OUT_FACES
    from retinaface_amd.frames import synth_frames
    hw = (896, 1280)
    for stem, faces, cfg, i in (("mnet25", None, 411, 3), ("mnet-deconv-0517", HELD_OUT_FACES, 410, 17)):
        f = synth_frames(hw[0], hw[1], i + 1, config=cfg, faces=faces)[i]
        hit = oracle_cache.lookup(oracle_cache.frame_key(stem, hw, cfg, faces, i), f, hw)
        assert hit is not None, (stem, cfg, i)
        live = oracles[stem].detect(f, 0.5, 0.4, net_hw=hw)
        # (identical on the machine that minted the file; another CPU's convolution library may round the last bits of a sum differently)
        assert np.allclose(hit.rows(), live.rows(), rtol=0, atol=2e-3) and np.array_equal(hit.anchor_indices(), live.anchor_indices())
        assert [c.anchor_index for c in hit.candidates] == [c.anchor_index for c in live.candidates]
        assert all(np.allclose(a.as_row(), b.as_row(), rtol=0, atol=2e-3) for a, b in zip(hit.candidates, live.candidates)) and len(hit.detections) >= 1
        assert abs(hit.band - oracle_cache.band_of(live.heads)) <= 1
        g = f.copy()
        g[0, 0, 0] ^= 1
        assert oracle_cache.lookup(oracle_cache.frame_key(stem, hw, cfg, faces, i), g, hw) is None
