"""Parity tests proper (`-m gpu`, run on the MI355X box): the HIP path, called through the C ABI, against the CPU
oracle on the same inputs, against the committed golden vectors, and -- at BASELINE.json's full sizes -- through
size-independent properties (batch-composition invariance, determinism, graph == eager).

Tolerances (stated once, used everywhere):
  fp32 engine : anchor indices identical, candidate counts identical, box IoU >= 1 - 1e-5, |score| <= 1e-5
  fp16 engine : anchor indices identical AND in the oracle's order, box IoU >= 1 - 1e-3 (north_star's bound), landmarks within 0.15 px,
                |score - oracle| <= LOGIT_NOISE * p (1 - p) + 1e-5 (round 5): the score is a sigmoid of the logit difference, so a logit error
                moves it by p (1 - p) times itself; LOGIT_NOISE = 0.028 is what PLAIN fp16 storage is predicted to cost the classification
                output in logit space (tools/fp16_error_budget.py replayed over the golden frames, minted into threshold_bands.npz by
                tools/make_golden.py --bands) -- the same "no worse than plain fp16 storage" bar the per-layer check uses.  That is 9e-5 at
                p = 0.997 (where the flat 2e-3 of rounds 1-4 tested nothing) and 4.3e-3 at p = 0.81, where the engine measured 2.02e-3
                (a logit error of 0.013).  Anchor SET (round 5): identical, except that a detection may sit on the TWIN of the oracle's
                anchor -- a candidate of the oracle itself that suppresses / is suppressed by it and whose oracle score is within twice the
                score noise (tests/anchor_twins.py; fires on 1 of the 208 contract frames, anchors 300 / 301 at 0.997809 / 0.997806); the
                box is then held to the same 1e-3 against the oracle's box of that twin.  Candidate count: within the number of anchors whose oracle probability lies inside the
                score-noise band |p - thr| <= 2e-3 (only those can cross `conf <= thr`; 0..2 on the golden frames, minted into
                tests/golden/threshold_bands.npz by tools/make_golden.py --bands; computed live where the oracle runs) -- round 4,
                a flat +-4 before.  Per-layer activations: within what plain fp16 storage is PREDICTED to cost at that tensor
                (tools/fp16_error_budget.py predicted_layer_errors: the fused-op sequence replayed on the CPU with fp16 rounding at
                every storage point), a flat 3 % of the range before.  Round 1 needed 2e-3 on ~50 px faces; the per-tensor error
                budget (tools/fp16_error_budget.py) showed 73 % of the box-error variance came from three tensors that
                never leave the stem kernel's LDS (conv0 output, first depthwise output and its taps, on raw 0..255
                pixels); they are fp32-grade now and the whole engine sits at <= ~7e-4.
  post-processing alone (decode + NMS given the GPU's own head blobs, vs the plain-C restatement):
                anchor indices and scores bit-exact, coordinates within 1e-4 px (expf ulp)
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_cache
from anchor_twins import resolve, twins_of_result
from conftest import ASSETS, ROOT, STEMS, golden
from oracle import build as obuild
from oracle.caffe_forward import HEAD_STRIDES, head_names
from oracle.retinaface_post import iou_plus1, preprocess_trt_identity

pytestmark = pytest.mark.gpu

FP32, FP16 = 0, 1
TOL = {FP32: dict(iou=1e-5, score=1e-5, lm=2e-3), FP16: dict(iou=1e-3, score=2e-3, lm=0.15)}
assert oracle_cache.SCORE_NOISE == TOL[FP16]["score"]
SCORE_NOISE = TOL[FP16]["score"]          # the width of the threshold / order / twin bands: fp16 score noise where the sigmoid is steepest (a logit error of 8e-3)


LOGIT_NOISE = 0.028                       # what plain fp16 storage is predicted to cost the classification logit (tools/fp16_error_budget.py over the golden frames)
LOGIT_ERR_GATE = 0.5 * LOGIT_NOISE        # the engine's measured worst logit error over the contract's unsaturated detections must stay under half of it (measured 0.36)


def score_tol(prec, p):
    """Bound on |engine score - oracle score| for a detection whose oracle score is p (see the module docstring)."""
    if prec != FP16:
        return TOL[prec]["score"]
    # LOGIT_NOISE is a REVIEWED constant: the minted prediction must equal it, so regenerating the golden cannot move the bar silently
    assert abs(float(golden("threshold_bands.npz")["logit_noise"]) - LOGIT_NOISE) < 5e-4
    return LOGIT_NOISE * float(p) * (1.0 - float(p)) + 1e-5
LAYER_ERR_FACTOR = 1.0                    # fp16 per-layer bar = the error-budget tool's prediction for plain fp16 storage (measured: 0.14-0.37 of it)


def ncand_band(prec, key=None, heads=None, thr=0.5):
    """How far the engine's candidate count may be from the oracle's: the number of anchors whose oracle foreground probability is
    within the fp16 score noise of the threshold (fp32 engine: none).  `key` looks the count up in the minted golden, `heads` (the
    oracle's 9 blobs) computes it."""
    if prec == FP32:
        return 0
    if heads is not None:
        n = 0
        for s_ in HEAD_STRIDES:
            p = heads[head_names(s_)[0]]
            n += int((np.abs(p[:, p.shape[1] // 2:] - np.float32(thr)) <= SCORE_NOISE).sum())
        return n
    g = golden("threshold_bands.npz")
    assert abs(float(g["score_noise"]) - SCORE_NOISE) < 1e-9
    return int(g[key])


def same_order_where_the_oracle_is_decisive(got_idx, ref_idx, ref_scores, prec):
    """The engine's detections, sorted by its own scores, must follow the oracle's order wherever two oracle scores differ by more
    than twice the score noise (closer pairs may legitimately swap)."""
    noise = SCORE_NOISE if prec == FP16 else TOL[FP32]["score"]
    pos = {a: i for i, a in enumerate(got_idx)}
    for i in range(len(ref_idx)):
        for j in range(i + 1, len(ref_idx)):
            if ref_scores[i] - ref_scores[j] > 2 * noise:
                assert pos[ref_idx[i]] < pos[ref_idx[j]], (ref_idx[i], ref_idx[j], ref_scores[i], ref_scores[j])


@pytest.fixture(scope="module")
def rfa(built_lib):
    import retinaface_amd
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need the GPU box"
    return retinaface_amd


_engines = {}


def engine(rfa, stem, prec, hw, **kw):
    key = (stem, prec, hw, tuple(sorted(kw.items())))
    if key not in _engines or not _engines[key]._h:           # (a test may have closed the engine it was handed: build it again)
        _engines[key] = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=prec, net_hw=hw, model_stem=stem, **kw)
    return _engines[key]


NMS_THRESHOLD = 0.4
TWIN_SWAPS = []        # every firing of the anchor-twin band in this session: (engine's anchors, oracle's anchors)


def twin_band(key):
    """The admissible twin pairs of a golden frame (tests/golden/threshold_bands.npz `<key>/twins`, tools/make_golden.py --bands)."""
    g = golden("threshold_bands.npz")
    return g[key + "/twins"] if key + "/twins" in g.files else np.zeros((0, 17), np.float32)


def compare(got, ref_rows, ref_idx, prec, twins=None):
    """Detections vs the oracle's: same anchors in the same order, boxes / scores / landmarks within TOL.  fp16 only: `twins` (the
    oracle's own near-tie pairs, tests/anchor_twins.py) lets a detection sit on the twin of the oracle's anchor; it is then held
    to the oracle's values for that twin."""
    t = TOL[prec]
    got_idx = [d.anchor_index for d in got]
    if prec == FP16 and twins is not None and len(twins) and got_idx != [int(a) for a in ref_idx]:
        ref_rows, swaps, canon = resolve(got_idx, ref_rows, ref_idx, twins)
        assert canon == [int(a) for a in ref_idx], (got_idx, list(ref_idx))
        TWIN_SWAPS.append((got_idx, [int(a) for a in ref_idx]))
    else:
        assert got_idx == list(ref_idx), (got_idx, list(ref_idx))
    for g, r in zip(got, ref_rows):
        assert iou_plus1(g.rect, r[1:5]) >= 1 - t["iou"], (g.rect, r[1:5])
        assert abs(g.score - r[0]) <= score_tol(prec, r[0]), (g.score, r[0])
        assert np.abs(g.as_row()[5:] - r[5:]).max() <= t["lm"]


# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", [FP32, FP16])
def test_every_fused_op_against_the_oracle_blob(rfa, oracles, crop448, prec):
    """Layer by layer: each fused kernel's NHWC output vs the reference blob it stands for."""
    det = engine(rfa, "mnet-deconv-0517", prec, (448, 448), keep_outputs=True, use_graph=False)
    det.detect(crop448, 0.5)
    blobs = oracles["mnet-deconv-0517"].forward(preprocess_trt_identity(crop448, 448, 448), keep_all=True)
    names = ["mobilenet0_relu0_fwd"] + [f"mobilenet0_relu{i}_fwd" for i in range(2, 27, 2)]
    names += ["rf_c3_lateral_relu", "rf_c2_lateral_relu", "rf_c2_aggr_relu", "rf_c1_red_conv_relu", "rf_c1_aggr_relu"]
    for c in (3, 2, 1):
        names += [f"rf_c{c}_det_context_conv1_relu", f"rf_c{c}_det_context_conv3_1_relu", f"rf_c{c}_det_concat_relu"]
    rel_max, rel_mean = (1e-4, 1e-5) if prec == FP32 else (None, 2e-3)
    pred = {}
    if prec == FP16:
        import importlib.util
        spec = importlib.util.spec_from_file_location("fp16_error_budget", os.path.join(ROOT, "tools", "fp16_error_budget.py"))
        feb = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(feb)
        pred = feb.predicted_layer_errors("mnet-deconv-0517", crop448)
    if prec == FP16:
        # the fp16 engine runs conv0 .. conv4 as one launch (stem2): relu0 / relu2 never exist in HBM; relu4 is its output
        names.remove("mobilenet0_relu0_fwd")
        names.remove("mobilenet0_relu2_fwd")
        names.remove("mobilenet0_relu6_fwd")      # conv5..conv8 are one launch too (dwpw2): relu6 stays in LDS, relu8 is its output
        for c in (3, 2, 1):                       # the SSH tail is one launch (ssh_tail): context_conv3_1 stays in LDS; concat is checked
            names.remove(f"rf_c{c}_det_context_conv3_1_relu")
    for n in names:
        a = det.debug_activation(n)
        r = blobs[n][0].transpose(1, 2, 0)
        assert a.shape == r.shape, n
        scale = max(1.0, float(np.abs(r).max()))
        d = np.abs(a - r)
        if prec == FP16:
            # the bar is the PREDICTED cost of plain fp16 storage at this tensor (the engine is fp32-grade at several points of the
            # stem, DC-centred and tap-equalised elsewhere: it should sit at or below the prediction), not a flat fraction of the range
            bar = LAYER_ERR_FACTOR * pred[n][0]
            print(f"fp16 layer {n:36s} max err {d.max():.3e}  predicted {pred[n][0]:.3e}  ratio {d.max() / pred[n][0]:.2f}  range {scale:.2f}")
            assert d.max() <= bar and d.mean() <= rel_mean * scale, (n, float(d.max()), bar, float(d.mean()), scale)
        else:
            assert d.max() <= rel_max * scale and d.mean() <= rel_mean * scale, (n, float(d.max()), float(d.mean()), scale)


@pytest.mark.parametrize("prec", [FP32, FP16])
@pytest.mark.parametrize("stem", STEMS)
def test_head_blobs_against_golden(rfa, crop448, stem, prec):
    """blob_by_name() equivalents vs the frozen oracle blobs (golden: tests/golden/crop448_*.npz)."""
    det = engine(rfa, stem, prec, (448, 448), keep_outputs=True, use_graph=False)
    got = det.detect(crop448, 0.5)
    g = golden(f"crop448_{stem}.npz")
    atol = 5e-5 if prec == FP32 else 3e-2
    for s in HEAD_STRIDES:
        for n in head_names(s):
            assert np.abs(det.get_output(n) - g[n]).max() <= atol, n
    compare(got, g["det"], g["det_idx"], prec, twin_band(f"{stem}/crop448/05"))
    assert abs(det.last_candidate_counts(1)[0] - len(g["cand_idx"])) <= ncand_band(prec, f"{stem}/crop448/05")


@pytest.mark.parametrize("prec", [FP32, FP16])
@pytest.mark.parametrize("stem", STEMS)
def test_reference_fixture_image_1280x896(rfa, base_frame, stem, prec):
    """The reference's only image at BASELINE config 4's size: 6 faces, identical anchors, IoU within tolerance."""
    det = engine(rfa, stem, prec, (896, 1280), max_batch=2)
    g = golden(f"fixture_{stem}.npz")
    got = det.detect(base_frame, 0.5)
    assert len(got) == 6
    compare(got, g["det"], g["det_idx"], prec, twin_band(f"{stem}/fixture/05"))
    assert abs(det.last_candidate_counts(1)[0] - len(g["cand_idx"])) <= ncand_band(prec, f"{stem}/fixture/05")
    got9 = det.detect(base_frame, 0.9)                   # main.cpp:43 uses 0.9
    compare(got9, g["det09"], g["det09_idx"], prec, twin_band(f"{stem}/fixture/09"))
    assert abs(det.last_candidate_counts(1)[0] - int(g["ncand09"])) <= ncand_band(prec, f"{stem}/fixture/09")
    # the unpadded 886-row frame is placed top-left on the zero canvas: same result as the padded one
    got_u = det.detect(np.ascontiguousarray(base_frame[:886]), 0.5)
    assert [d.anchor_index for d in got_u] == [d.anchor_index for d in got]
    assert all(a.rect == b.rect for a, b in zip(got_u, got))


@pytest.mark.parametrize("prec", [FP32, FP16])
@pytest.mark.parametrize("stem", STEMS)
def test_synthetic_batch8_against_golden(rfa, stem, prec):
    """BASELINE config 2's shape (448x448, batch 8) on the seeded face-bearing frames."""
    from retinaface_amd.frames import synth_frames
    frames = synth_frames(448, 448, 8, config=1)
    det = engine(rfa, stem, prec, (448, 448))
    g = golden(f"synth448_{stem}.npz")
    for thr, tag in ((0.5, "05"), (0.9, "09")):
        res = det.detectBatchImages(frames, thr)
        ncand = det.last_candidate_counts(8)
        for i in range(8):
            compare(res[i], g[f"det{tag}_{i}"], g[f"idx{tag}_{i}"], prec, twin_band(f"{stem}/synth448_{i}/{tag}"))
            assert abs(ncand[i] - int(g[f"ncand{tag}_{i}"])) <= ncand_band(prec, f"{stem}/synth448_{i}/{tag}"), (stem, tag, i, ncand[i])


PROBE_LIB = os.path.join(ROOT, "retinaface_amd", "lib", "libretinaface_amd_probe.so")


@pytest.mark.skipif(os.environ.get("RF_PROBE_TESTS") != "1",
                    reason="23 subprocesses against the PROBE build (make probe): run with RF_PROBE_TESTS=1 (tools/gpu/r6.sh <tag> probes; the result is committed "
                           "under profiles/); kept out of the driver's -m gpu run, which tests the product library")
@pytest.mark.parametrize("knob", ["RF_STEM2=0", "RF_STEM2=2", "RF_STEM2=3", "RF_DWPW2=0", "RF_CONV3=0", "RF_STEM2_DC=0", "RF_SSHTAIL=0", "RF_CONV3WS=0", "RF_CONV3WS=32",
                                  "RF_CONV3UPWS=0", "RF_CONV3UPWS=3", "RF_CONV3UPWS=13", "RF_DWPWWS=3", "RF_DWPWWS=13", "RF_TILE256=1", "RF_DWPW2_RING=1",
                                  "RF_DWPW2_CHAIN=1", "RF_DWPW2_LAY2=0", "RF_DWPW2_HPAD=0", "RF_STEM2_V2=0", "RF_STEM2_V2=1", "RF_STEM2_V2=5", "RF_STEM2_V2=7"])
def test_probe_knob_kernel_variants_stay_correct(rfa, knob):
    """The measured-and-rejected kernel variants DESIGN.md cites stay buildable and correct: they live in the PROBE build only since round 5
    (libretinaface_amd_probe.so, -DRF_PROBES; the product library has neither the kernels nor the knobs, csrc/knobs.h).  Each RF_* probe knob is
    held to the same fp16 parity bar as the default path, in a subprocess so that the knob is seen at library start-up.
    RF_STEM2=0: K_a' stem + separate dwpw<16,32,s2>; 2: 7x16 tiles, 8 waves; 3: fp16 patch; RF_DWPW2=0: blocks 2 and 3 as two
    launches; RF_CONV3=0: 3x3 convs without the bank-row padding; RF_STEM2_DC=0: stem2's LDS tiles without the DC centring; RF_CONV3WS=0: the merged
    SSH conv on the lock-step K_c kernel instead of the warp-specialised one (round 4), 32: wave = channel tile instead of the 2 + 2 + 1 roles;
    RF_CONV3UPWS=0 / 3 / 13: the aggregation convs on K_c / with a producer wave / with per-wave LDS-DMA and three ring buffers; RF_DWPWWS=3: the
    64- and 128-channel blocks warp-specialised (13: the memory side spread over the four GEMM waves); RF_TILE256=1: 8 x 8 tiles for the 256-channel block;
    RF_DWPW2_RING=1: dwpw2's depthwise A as a ring; RF_DWPW2_CHAIN=1: dwpw2 with the depthwise -> pointwise hops chained in registers (permuted K order);
    RF_DWPW2_LAY2=0 / RF_DWPW2_HPAD=0: dwpw2 with round 3's LDS pitches / unpadded halo rows; RF_STEM2_V2=0: stem2's conv2 tile as 32-byte pixels and pixel = thread index
    in its depthwise-1 phase (round 3), 1: planar conv2 tile only, 5: both layout changes without the conv3 -> conv4 register chain (round 4's default), 7: with the
    chain (round 5's default); 15 = 7 + the raw-row staging of aligned full-width frames is the default since round 6 (frames that are NOT aligned take the general
    path in the product too: test_device_frames_unaligned_pointer_odd_step_and_roi)."""
    code = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "import retinaface_amd\n"
        "from retinaface_amd.frames import synth_frames\n"
        "det = retinaface_amd.RetinaFace(%r, 'net3', 0.4, precision=1, net_hw=(448, 448), model_stem='mnet25')\n"
        "res = det.detectBatchImages(synth_frames(448, 448, 8, config=1), 0.5)\n"
        "print('RESULT ' + json.dumps([[[d.anchor_index] + [float(v) for v in d.as_row()] for d in r] for r in res]))\n"
    ) % (ROOT, ASSETS)
    assert os.path.exists(PROBE_LIB), "build the probe library first: make -C retinaface_amd/csrc probe"
    env = dict(os.environ, RETINAFACE_AMD_LIB=PROBE_LIB)
    k, v = knob.split("=")
    env[k] = v
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "is not a value this knob knows" not in out.stderr and "ignored: probe knobs" not in out.stderr, out.stderr[-500:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    g = golden("synth448_mnet25.npz")
    t = TOL[FP16]
    for i in range(8):
        ref_rows, ref_idx = g[f"det05_{i}"], g[f"idx05_{i}"]
        got_idx = [int(r[0]) for r in res[i]]
        if got_idx != list(ref_idx):             # only the oracle's own near-tie twins may differ (tests/anchor_twins.py)
            ref_rows, _, canon = resolve(got_idx, ref_rows, ref_idx, twin_band(f"mnet25/synth448_{i}/05"))
            assert canon == list(ref_idx), (knob, i)
        for got, ref in zip(res[i], ref_rows):
            assert iou_plus1(got[2:6], ref[1:5]) >= 1 - t["iou"], (knob, i)
            assert abs(got[1] - ref[0]) <= score_tol(FP16, ref[0])


@pytest.mark.parametrize("thr", [0.5, 0.1, 0.02, 0.004, 0.0015])
def test_postprocessing_is_exact_given_the_gpu_head_blobs(rfa, crop448, thr):
    """Decode + regression + clip + NMS on the device vs the plain-C restatement fed the device's own head blobs:
    anchor indices and scores bit-exact, coordinates to expf-ulp.  The thresholds walk through all three sort paths of the
    NMS kernel: single-wave rank sort (<= 64 candidates), 256-thread rank sort (<= 256), bitonic network (more)."""
    det = engine(rfa, "mnet25", FP32, (448, 448), keep_outputs=True, use_graph=False)
    got = det.detect(crop448, thr)
    heads9 = [det.get_output(n) for s in HEAD_STRIDES for n in head_names(s)]
    cand, cidx, kept, kidx = obuild.decode_nms(heads9, 448, 448, thr, 0.4)
    assert det.last_candidate_counts(1)[0] == len(cidx)
    assert [d.anchor_index for d in got] == kidx.tolist()
    rows = np.stack([d.as_row() for d in got]) if got else np.zeros((0, 15), np.float32)
    assert np.array_equal(rows[:, 0], kept[:, 0])
    assert np.abs(rows - kept).max(initial=0.0) <= 1e-4
    if thr == 0.02:
        assert len(cidx) > 60
    if thr == 0.004:
        assert 64 < len(cidx) <= 4096
    if thr == 0.0015:
        assert 256 < len(cidx) <= 4096, len(cidx)


@pytest.mark.parametrize("prec", [FP32, FP16])
def test_postprocessing_against_the_reference_build_itself(rfa, crop448, prec):
    """Same property, but the checker is the reference's OWN RetinaFace::postProcess (RetinaFace.cpp:495-574, compiled
    unmodified into oracle/_ref by oracle/build_ref.py in the dev container; the prebuilt .so travels here): the device's
    head blobs go into the reference's engine slots, its decode + NMS must keep the same faces in the same order."""
    from oracle import build_ref
    if not build_ref.available():
        pytest.skip("oracle/_ref/libretinaface_ref.so not shipped with this snapshot")
    det = engine(rfa, "mnet-deconv-0517", prec, (448, 448), keep_outputs=True, use_graph=False)
    ref = build_ref.ReferenceRetinaFace(448, 448)
    try:
        for thr in (0.5, 0.1, 0.02, 0.004):
            got = det.detect(crop448, thr)
            ref.set_heads(0, [det.get_output(n) for n in build_ref.HEAD_BLOBS])
            want = ref.postprocess(0, thr)
            rows = np.stack([d.as_row() for d in got]) if got else np.zeros((0, 15), np.float32)
            assert len(rows) == len(want) and len(want) > 0, (thr, len(rows), len(want))
            assert np.array_equal(rows[:, 0], want[:, 0])
            assert np.abs(rows - want).max() <= 1e-4
    finally:
        ref.close()


INT8 = 2


def _match(got, ref_rows):
    """for every oracle face: IoU of the best-matching detection"""
    return [max([iou_plus1(g.rect, r[1:5]) for g in got] or [0.0]) for r in ref_rows]


# The int8 engine's DISTANCE to the fp32 oracle (reported and gated; the PARITY bar of the int8 engine is the bit-exact integer oracle below).
# Metrics: tests/int8_contract.py.  Calibration (round 6, tools/calibrate_int8.py --per-channel --rule amax --margin 1.25 --gptq, both models):
# per-channel activation scales with 25 % head-room, error-compensated weight rounding + bias correction on 48 frames that show only fixture
# faces 0 / 2 / 4; every frame below uses faces 1 / 3 / 5 and other seeds (held out).  Depthwise outputs carry 8 bits (0..255 quanta).
#   same-anchor IoU  the regression error alone (engine's box vs the ORACLE'S box of the same anchor): the number north_star's "1e-3 IoU" is
#                    about; fp16 scores 0.9993-0.9995 on this metric, int8 0.972-0.979 at worst over 280 faces on the final tree (0.968-0.981 before the
#                    stem's raw-row staging re-rolled the first-layer rounding; round 5 tables: 0.947-0.949).  Agreement: 0.964 / 0.946.
#   anchor agreement fraction of faces kept on the oracle's anchor.  Neighbouring anchors of one face score within ~1e-3 of each other in the
#                    oracle itself, so any logit noise flips some winners (the fp16 engine: 2 of 280 on 0517); a flip shows up as a per-face
#                    IoU of 0.88-0.95 although both boxes are the network's own predictions -- the per-face IoU is printed, and gated only loosely.
# A 1-LSB change of a handful of first-layer quanta moves these statistics by +-0.015 (agreement) / +-0.005 (worst IoU): the gates sit that far
# below the measured values (profiles/r06_int8_contract.json).
INT8_BAR = dict(anchor_iou=0.96, anchor_iou_p01=0.968, agreement=0.92, iou_mean=0.985, iou_floor=0.86, dscore=0.06)
INT8_TARGET = dict(anchor_iou=0.97, agreement=0.95)          # VERDICT r5's "done" line, printed beside the measured numbers


def _check_int8(stem, what, s, bar=INT8_BAR, min_faces=1):
    from int8_contract import fmt
    print(f"[{what}] " + fmt(stem, s) + f"  | targets: same-anchor IoU >= {INT8_TARGET['anchor_iou']}, agreement >= {INT8_TARGET['agreement']}")
    assert s["same_count"] == s["frames"] and s["faces"] >= min_faces and s["unmatched"] == 0, (stem, what, s)
    assert s["anchor_iou_worst"] >= bar["anchor_iou"] and s["iou_worst"] >= bar["iou_floor"] and s["dscore_max"] <= bar["dscore"], (stem, what, s)
    if s["faces"] >= 100:
        assert s["anchor_agreement"] >= bar["agreement"] and s["anchor_iou_p01"] >= bar["anchor_iou_p01"] and s["iou_mean"] >= bar["iou_mean"], (stem, what, s)


@pytest.mark.parametrize("stem", STEMS)
def test_int8_engine_against_the_fp32_oracle(rfa, oracles, base_frame, stem):
    """int8 engine (symmetric quantisation, i8 MFMA for every contraction incl. the depthwise stencil with 15-bit taps) against the fp32
    oracle on three small sets: held-out synthetic frames at batch 32 in one call (configs[2]'s shape), the same frames through a batch-8
    engine (batch composition must not matter), and the reference photo at 1280 x 896.  Same number of faces on every frame, the
    regression error per anchor and the score error bounded (INT8_BAR); the 208-frame contract below carries the statistics."""
    from int8_contract import frame_rows, summarize
    from retinaface_amd.frames import synth_frames
    held = synth_frames(448, 448, 32, config=300, faces=[1, 3, 5])
    det32 = engine(rfa, stem, INT8, (448, 448), max_batch=32)
    res = det32.detectBatchImages(held, 0.5)
    refs = [oracles[stem].detect(f, 0.5, 0.4, net_hw=(448, 448)) for f in held]
    _check_int8(stem, "held-out b32", summarize([dict(same_count=len(g) == len(r.detections), rows=frame_rows(g, r)) for g, r in zip(res, refs)]), min_faces=60)
    det8 = engine(rfa, stem, INT8, (448, 448))
    assert _key(det8.detectBatchImages(held, 0.5)) == _key(res)
    big = engine(rfa, stem, INT8, (896, 1280), max_batch=2)
    got = big.detect(base_frame, 0.5)
    ref = oracles[stem].detect(base_frame, 0.5, 0.4, net_hw=(896, 1280))
    assert len(got) == 6
    _check_int8(stem, "fixture photo", summarize([dict(same_count=len(got) == len(ref.detections), rows=frame_rows(got, ref))]), min_faces=6)


@pytest.mark.parametrize("stem", STEMS)
def test_int8_contract_over_200_frames(rfa, oracles, stem):
    """The fp16 contract's frame plan (both models x {448 x 448: 32 + 4 x 8 frames, 1280 x 896: 32 + 8 frames} = 104 frames per model, 208 in
    all) for the int8 engine, on held-out faces: identical face count on every frame, same-anchor IoU, anchor agreement, per-face IoU
    distribution and |dscore| -- printed against VERDICT r5's targets and gated at INT8_BAR."""
    from int8_contract import run_contract
    # run_contract closes each engine when its batch is done: hand it fresh engines, never the session's cached ones (engine())
    s = run_contract(lambda hw, nb: rfa.RetinaFace(ASSETS, "net3", 0.4, precision=INT8, net_hw=hw, model_stem=stem, max_batch=nb), oracles[stem], stem=stem)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"int8_contract_{stem}.json"), "w") as f:
        json.dump(s, f, indent=1)
    assert s["frames"] == 104
    _check_int8(stem, "contract", s, min_faces=250)


def test_int8_engine_with_the_reference_tensorrt_table(rfa, nets, tmp_path):
    """The calibration cache the REFERENCE ships (model/mnet-deconv-0517.table.int8, per tensor, TensorRT's own: assets/ keeps it verbatim)
    still drives the engine: packed into a container through the C ABI (rf_attach_calibration drops the calibrated weights, which belong to
    the repo's own table), the engine is bit-exact against the integer oracle built from the same file -- the per-tensor code path: scalar
    scale ratios in the fused upsample + add (fp32 blend), one scale per concat."""
    import copy
    from oracle.caffe_io import read_int8_table, read_rfw
    from oracle.int8_forward import Int8Net
    from retinaface_amd.frames import synth_frames
    stem = "mnet-deconv-0517"
    lib = rfa.load_library()
    out = str(tmp_path / (stem + ".rfw"))
    assert lib.rf_attach_calibration(ASSETS.encode(), stem.encode(), os.path.join(ASSETS, stem + ".table.int8").encode(), None, out.encode()) == 0
    net = read_rfw(out)
    assert net.int8_qweights == {} and net.int8_scales == read_int8_table(os.path.join(ASSETS, stem + ".table.int8")) and "_plus0#0" not in net.int8_scales
    q = Int8Net(net)
    assert not q.per_channel
    det = rfa.RetinaFace(str(tmp_path), "net3", 0.4, precision=INT8, net_hw=(448, 448), model_stem=stem, max_batch=8, keep_outputs=True,
                         use_graph=False, plan_cache=False)
    frames = synth_frames(448, 448, 8, config=300, faces=[1, 3, 5])
    res = det.detectBatchImages(frames, 0.5)
    assert sum(_assert_int8_image_bit_exact(det, q, i, (448, 448), 0.5, res[i]) for i in range(8)) >= 8
    det.close()


def _int8_stats(res, refs):
    """Per set: (same face count everywhere, worst per-face IoU, anchor agreement rate, max |dscore|)."""
    ious, same_anchor, faces, ds, same = [], 0, 0, 0.0, True
    for got, ref in zip(res, refs):
        same &= len(got) == len(ref)
        for r_rect, r_score, r_idx in ref:
            faces += 1
            best = max(got, key=lambda a: iou_plus1(a.rect, r_rect)) if got else None
            ious.append(iou_plus1(best.rect, r_rect) if best else 0.0)
            same_anchor += bool(best) and best.anchor_index == r_idx
            ds = max(ds, abs(best.score - r_score) if best else 1.0)
    return same, min(ious), same_anchor / max(faces, 1), ds


def _int8_start_blob(det):
    """first int8 activation the engine keeps in HBM: the output of its float front end (relu2: stem; relu4: the fused stem2)"""
    for n in ("mobilenet0_relu2_fwd", "mobilenet0_relu4_fwd"):
        try:
            det.debug_activation(n + "#raw", 0)
            return n
        except RuntimeError:
            continue
    raise AssertionError("the int8 engine exposes neither relu2 nor relu4")


def _assert_int8_image_bit_exact(det, q, img, hw, thr, got):
    """One image of the engine's last batch against oracle/int8_forward.py continued from the engine's own front-end output:
    every int8 activation np.array_equal, raw head outputs np.array_equal, probabilities within 1e-6 (device expf), candidates
    and detections identical (anchors; scores 1e-6; coordinates 1e-4 px)."""
    start = _int8_start_blob(det)
    x = det.debug_activation(start + "#raw", img)
    assert np.array_equal(x, np.rint(x)) and x.min() >= 0 and x.max() <= 127
    acts = q.forward_from(start, x.astype(np.int8))
    checked = 0
    for n, ref in acts.items():
        if n in ("__heads__", start):
            continue
        try:
            a = det.debug_activation(n + "#raw", img)
        except RuntimeError:
            continue                     # depthwise intermediates / `_plus` tensors never leave the kernels
        assert a.shape == ref.shape and np.array_equal(a.astype(np.int8), ref), (n, img, int(np.abs(a - ref).max()), float((a != ref).mean()))
        checked += 1
    assert checked >= 21, checked        # 12 block outputs (fewer where blocks are fused through LDS) + 5 FPN + 6..9 SSH tensors
    heads = acts["__heads__"]
    for s in HEAD_STRIDES:
        pn, bn, ln = head_names(s)
        assert np.array_equal(det.get_output(bn, img), heads[bn]) and np.array_equal(det.get_output(ln, img), heads[ln]), s
        assert np.abs(det.get_output(pn, img) - heads[pn]).max() <= 1e-6
    h9 = [heads[n] for s in HEAD_STRIDES for n in head_names(s)]
    cand, cidx, kept, kidx = obuild.decode_nms(h9, hw[0], hw[1], thr, 0.4)
    border = sum(int((np.abs(heads[head_names(s)[0]] - thr) <= 2e-6).sum()) for s in HEAD_STRIDES)
    assert abs(det.last_candidate_counts(img + 1)[img] - len(cidx)) <= border
    if border == 0:
        assert [d.anchor_index for d in got] == list(kidx), (img, [d.anchor_index for d in got], list(kidx))
        for g, r in zip(got, kept):
            assert abs(g.score - r[0]) <= 1e-6 and np.abs(g.as_row()[1:] - r[1:]).max() <= 1e-4
    return len(kidx)


@pytest.mark.parametrize("stem", STEMS)
def test_int8_engine_is_bit_exact_against_the_integer_oracle(rfa, nets, oracles, base_frame, stem):
    """PARITY of the int8 engine (BASELINE configs[2] / [4]: batch 32 at 448 x 448, both models; plus 1280 x 896).  int8 MFMA
    accumulation is exact and the requantising epilogue is one fmaf + one round-half-even, so the engine must reproduce the
    integer oracle (oracle/int8_forward.py: quantised weights, multipliers and biases re-derived from the Caffe model + the
    calibration table in numpy, independently of weights.h) bit for bit on every int8 activation of every image.  The float
    front end (preprocess + conv0 + first block(s), fp16/fp32-grade on raw pixels, int8 only at its output) is the pinned input,
    and is itself held to <= 1 LSB of the quantised fp32 oracle.  INT8_BAR above stays only as the reported distance to fp32."""
    from oracle.int8_forward import Int8Net
    from retinaface_amd.frames import synth_frames
    q = Int8Net(nets[stem])
    frames = synth_frames(448, 448, 32, config=300, faces=[1, 3, 5])
    det = engine(rfa, stem, INT8, (448, 448), max_batch=32, keep_outputs=True, use_graph=False)
    res = det.detectBatchImages(frames, 0.5)
    faces = sum(_assert_int8_image_bit_exact(det, q, i, (448, 448), 0.5, res[i]) for i in range(32))
    assert faces >= 32
    # the float front end against the fp32 oracle, in output quanta
    start = _int8_start_blob(det)
    off, tot = 0, 0
    for i in (0, 13, 31):
        blobs = oracles[stem].forward(preprocess_trt_identity(frames[i], 448, 448), keep_all=True)
        ref = q.quantise_blob(start, blobs[start][0].transpose(1, 2, 0)).astype(np.int32)
        a = det.debug_activation(start + "#raw", i).astype(np.int32)
        assert np.abs(a - ref).max() <= 1, (stem, i, int(np.abs(a - ref).max()))
        off += int((a != ref).sum())
        tot += a.size
    print(f"int8 front end {stem}: {off / tot:.5f} of the {start} quanta differ (by 1 LSB) from the quantised fp32 oracle")
    assert off / tot <= 0.02
    # graph replay and the default (coalescing, 3-lane) engine give the same detections as the eager run that was checked
    det_g = engine(rfa, stem, INT8, (448, 448), max_batch=32)
    assert _key(det_g.detectBatchImages(frames, 0.5)) == _key(res)
    # 1280 x 896 (the large-frame shape of configs[3]) in int8: the reference photo and a synthetic frame
    big = engine(rfa, stem, INT8, (896, 1280), max_batch=2, keep_outputs=True, use_graph=False)
    pair = [base_frame, synth_frames(896, 1280, 1, config=301)[0]]
    res = big.detectBatchImages(pair, 0.5)
    for i in range(2):
        _assert_int8_image_bit_exact(big, q, i, (896, 1280), 0.5, res[i])


def test_calibration_tool_end_to_end(rfa, nets, oracles, tmp_path):
    """SURVEY 8f rank 3, the INT8 calibration-table generator, as a whole: tools/calibrate_int8.py (per-channel, amax rule with head-room,
    a small built-in calibration set that shows only fixture faces 0 / 2 / 4) collects activations from the fp32 HIP engine, writes a
    table in the reference's text format, calibrates the WEIGHTS on the same activations (error-compensated rounding + bias correction,
    round 6) and packs model + table + weights into an .rfw through the C ABI (rf_attach_calibration).  The int8 engine built from that
    file must (a) be BIT-exact against the integer oracle built from the same file (the oracle reads the container with its own reader)
    and (b) find the fp32 oracle's faces on held-out frames (faces 1 / 3 / 5) at the int8 bar.  (TensorRT's calibrator is closed source:
    the tool's thresholds themselves stay "parity unpinned"; what is pinned is that what it writes drives the int8 path correctly.)"""
    from oracle.caffe_io import read_int8_qweights, read_int8_table, read_rfw
    from oracle.int8_forward import Int8Net
    from retinaface_amd.frames import synth_frames
    table = tmp_path / "mnet25.table.int8"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "calibrate_int8.py"), "--model", "mnet25", "--per-channel", "--rule", "amax",
                          "--margin", "1.25", "--frames", "12", "--config", "91", "--gptq", "--out", str(table),
                          "--out-rfw", str(tmp_path / "mnet25.rfw")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    scales = read_int8_table(str(table))
    assert "mobilenet0_relu2_fwd" in scales and "mobilenet0_relu2_fwd#15" in scales and "_plus1#63" in scales and len(scales) > 1000
    net = read_rfw(str(tmp_path / "mnet25.rfw"))
    loose = read_int8_qweights(str(tmp_path / "mnet25.qweights.int8"))
    assert len(net.int8_qweights) == 29 and set(loose) == set(net.int8_qweights)          # 12 pointwise + 3 lateral + 2 aggr + 3 x 4 SSH / head
    assert all(np.array_equal(loose[k][0], v[0]) and np.array_equal(loose[k][1], v[1]) for k, v in net.int8_qweights.items())
    assert all(abs(net.int8_scales[k] - np.float32(v)) == 0 for k, v in scales.items())
    det = rfa.RetinaFace(str(tmp_path), "net3", 0.4, precision=INT8, net_hw=(448, 448), model_stem="mnet25", max_batch=8, keep_outputs=True,
                         use_graph=False, plan_cache=False)
    q = Int8Net(net)
    held = synth_frames(448, 448, 8, config=305, faces=[1, 3, 5])
    res = det.detectBatchImages(held, 0.5)
    for i in range(8):
        _assert_int8_image_bit_exact(det, q, i, (448, 448), 0.5, res[i])
    refs = []
    for f in held:
        o = oracles["mnet25"].detect(f, 0.5, 0.4, net_hw=(448, 448))
        refs.append([(d.rect, d.score, d.anchor_index) for d in o.detections])
    same, worst, agree, ds = _int8_stats(res, refs)
    print(f"calibration tool, 12 frames: same face count {same}, worst IoU {worst:.3f}, anchor agreement {agree:.2f}, max |dscore| {ds:.3f}")
    assert same and worst >= 0.85 and ds <= 0.05, (same, worst, agree, ds)
    det.close()


def test_int8_integer_blend_is_bit_identical_to_the_fp32_blend(rfa):
    """int8 engine, per-channel table (mnet25): the fused upsample + add runs in packed 16-bit integer arithmetic.  Every
    intermediate of the fp32 form is exact and both round half to even, so the two must agree bit for bit: same detections
    (scores, boxes, landmarks, anchors) with the integer blend and with RF_BLEND_FP32=1 (the knob is read at launch-build time)."""
    from retinaface_amd.frames import synth_frames
    frames = synth_frames(448, 448, 8, config=7)
    res = []
    for force_fp32 in (False, True):
        if force_fp32:
            os.environ["RF_BLEND_FP32"] = "1"
        try:
            det = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=INT8, net_hw=(448, 448), model_stem="mnet25")
            res.append(_key(det.detectBatchImages(frames, 0.5)))
            det.close()
        finally:
            os.environ.pop("RF_BLEND_FP32", None)
    assert res[0] == res[1] and sum(len(r) for r in res[0]) > 0


def test_int8_layers_stay_within_quantisation_noise(rfa, oracles, crop448):
    """Every int8 activation, dequantised with its table scale, vs the oracle blob: a wrong index or scale shows up as an
    error of the order of the range; quantisation noise stays within a few percent of it."""
    det = engine(rfa, "mnet-deconv-0517", INT8, (448, 448), use_graph=False, keep_outputs=True)
    det.detect(crop448, 0.5)
    blobs = oracles["mnet-deconv-0517"].forward(preprocess_trt_identity(crop448, 448, 448), keep_all=True)
    names = [f"mobilenet0_relu{i}_fwd" for i in range(2, 27, 2)]
    names += ["rf_c3_lateral_relu", "rf_c2_lateral_relu", "rf_c2_aggr_relu", "rf_c1_red_conv_relu", "rf_c1_aggr_relu"]
    for c in (3, 2, 1):
        names += [f"rf_c{c}_det_context_conv1_relu", f"rf_c{c}_det_context_conv3_1_relu", f"rf_c{c}_det_concat_relu"]
    worst = {}
    for n in names:
        if n.endswith("context_conv3_1_relu"):
            continue                     # stays in LDS (ssh_tail); the concat tensor it feeds is checked
        a = det.debug_activation(n)
        r = blobs[n][0].transpose(1, 2, 0)
        scale = max(1.0, float(np.abs(r).max()))
        d = np.abs(a - np.minimum(r, a.max() + 1e-6 if a.max() > 0 else r))     # values above the calibrated amax saturate by design
        worst[n] = (float(d.max()) / scale, float(d.mean()) / scale)
        assert d.mean() <= 0.035 * scale and d.max() <= 0.6 * scale, (n, worst[n])
    for s in HEAD_STRIDES:
        for n in head_names(s):
            assert np.abs(det.get_output(n) - golden("crop448_mnet-deconv-0517.npz")[n]).max() <= 0.5, n


FP16_CONTRACT = {}          # stem -> (worst 1 - IoU per frame, labelled rows, twin firings): filled per model, judged together by the summary test


@pytest.mark.parametrize("stem", STEMS)
def test_fp16_contract_over_200_frames_both_models_both_sizes(rfa, oracles, stem):
    """north_star's contract for the benchmarked precision, gated: >= 200 seeded frames -- both models (one parametrisation each: 104 frames),
    448 x 448 and 1280 x 896, submitted as 8- and 32-image batches (the target matrix's batch sizes) -- against the fp32 oracle.  Identical anchor sets on
    every frame (up to the oracle's own near-tie twin anchors, tests/anchor_twins.py: allowed ONLY on the one frame where it is known to fire), worst 1 - IoU <= 9e-4 (the bound is 1e-3; the margin is asserted, not hoped for), candidate counts within the
    threshold band.  The distribution is printed so a kernel change is judged by its margin."""
    from retinaface_amd.frames import synth_frames
    worst_all, rows, bands, twin_frames, logit_errs = [], [], [], [], []
    for hw, plan in (((448, 448), ((32, 400), (8, 401), (8, 402), (8, 403), (8, 404))), ((896, 1280), ((32, 410), (8, 411)))):
        dets = {}
        for nb, cfg in plan:
            frames = synth_frames(hw[0], hw[1], nb, config=cfg)
            if nb not in dets:
                dets[nb] = engine(rfa, stem, FP16, hw, max_batch=nb)          # one engine per (frame size, batch), shared by the seeds
            det = dets[nb]
            got = det.detectBatchImages(frames, 0.5)
            ncand = det.last_candidate_counts(nb)
            for i, f in enumerate(frames):
                # (the 1280 x 896 frames' oracle results come from tests/golden/contract_oracle_1280x896.npz when it holds this very frame --
                # tests/oracle_cache.py: same numbers, minted once; every 448 x 448 frame runs the live oracle)
                ref = oracle_cache.detect(oracles[stem], stem, f, hw, cfg, None, i)
                # faces are matched by global anchor index -- identical sets, except where the engine kept the oracle's own near-tie
                # TWIN of an anchor (tests/anchor_twins.py: counted and printed below; the box is then measured against the oracle's
                # box of that twin); the ORDER must be the oracle's wherever its scores differ by more than twice the fp16 score
                # noise (closer pairs may swap places)
                got_idx = [d.anchor_index for d in got[i]]
                ref_rows, swaps, canon = resolve(got_idx, ref.rows(), ref.anchor_indices(), twins_of_result(ref, NMS_THRESHOLD, SCORE_NOISE))
                if swaps:
                    twin_frames.append((f"{stem} {hw[1]}x{hw[0]} b{nb} cfg{cfg} #{i}", got_idx, ref.anchor_indices().tolist()))
                same_order_where_the_oracle_is_decisive(canon, [d.anchor_index for d in ref.detections],
                                                        [d.score for d in ref.detections], FP16)
                band = ref.band                                   # = ncand_band(FP16, heads=ref.heads): anchors within SCORE_NOISE of the threshold
                bands.append(band)
                assert abs(ncand[i] - len(ref.candidates)) <= band, (stem, hw, cfg, i, ncand[i], len(ref.candidates), band)
                for d, r in zip(got[i], ref_rows):
                    assert abs(d.score - r[0]) <= score_tol(FP16, r[0]), (stem, hw, cfg, i, d.anchor_index, d.score, r[0])
                    if 0.02 < r[0] < 0.98:
                        logit_errs.append(abs(np.log(d.score / (1 - d.score)) - np.log(float(r[0]) / (1 - float(r[0])))))
                w = max([1 - iou_plus1(d.rect, r[1:5]) for d, r in zip(got[i], ref_rows)], default=0.0)
                worst_all.append(w)
                rows.append((w, f"{stem} {hw[1]}x{hw[0]} b{nb} cfg{cfg} #{i}"))
        for d in dets.values():
            d.close()
    ws = np.array(worst_all)
    rows.sort(key=lambda r: -r[0])
    print(f"fp16 contract {stem}: {len(ws)} frames, worst 1-IoU {ws.max():.3e}, mean {ws.mean():.3e}, p99 {np.quantile(ws, 0.99):.3e}; worst: "
          + "; ".join(f"{w:.2e} {n}" for w, n in rows[:4]))
    print(f"fp16 contract {stem}: candidate-count bands (anchors within {SCORE_NOISE} of the threshold): max {max(bands)}, mean {np.mean(bands):.2f}, "
          f"{sum(b == 0 for b in bands)} of {len(bands)} frames with an empty band (count must then be identical)")
    print(f"fp16 contract {stem}: logit error of the {len(logit_errs)} unsaturated detections (0.02 < p < 0.98): max {max(logit_errs, default=0.0):.4f} of the "
          f"{LOGIT_NOISE:.4f} plain fp16 storage is predicted to cost (gate {LOGIT_ERR_GATE:.4f})")
    print(f"fp16 contract {stem}: anchor-twin band fired on {len(twin_frames)} of {len(ws)} frames: " + "; ".join(f"{n}: engine {g} oracle {r}" for n, g, r in twin_frames))
    FP16_CONTRACT[stem] = (ws, rows, twin_frames)
    assert len(ws) == 104 and ws.max() <= 9e-4, rows[:6]
    assert max(logit_errs, default=0.0) <= LOGIT_ERR_GATE, max(logit_errs)
    # The anchor-twin band (tests/anchor_twins.py) may fire ONLY where it is known to: one frame of the 208, where the oracle's own scores
    # of the two anchors differ by 3e-6 (0.997809 / 0.997806).  A firing anywhere else is a set divergence this test must not absorb
    # (ADVICE r5): it fails until a person has looked at the pair and added it here.
    known = {("mnet-deconv-0517 448x448 b8 cfg402 #1", (1593, 301), (1593, 300))}
    fired = {(n, tuple(g), tuple(int(a) for a in r)) for n, g, r in twin_frames}
    assert fired <= known, fired - known


def test_fp16_contract_summary():
    """The two halves of the contract together: >= 200 frames, one worst figure (what DESIGN.md quotes)."""
    if set(FP16_CONTRACT) != set(STEMS):
        pytest.skip("the per-model contract tests did not both run in this session")
    ws = np.concatenate([FP16_CONTRACT[s][0] for s in STEMS])
    rows = sorted(sum((FP16_CONTRACT[s][1] for s in STEMS), []), key=lambda r: -r[0])
    print(f"fp16 contract: {len(ws)} frames, worst 1-IoU {ws.max():.3e}, mean {ws.mean():.3e}, p99 {np.quantile(ws, 0.99):.3e}; worst: "
          + "; ".join(f"{w:.2e} {n}" for w, n in rows[:4]) + f"; twin band fired on {sum(len(FP16_CONTRACT[s][2]) for s in STEMS)} frame(s)")
    assert len(ws) >= 200 and ws.max() <= 9e-4


def test_handle_refuses_concurrent_entry(rfa):
    """include/retinaface_amd.h: one handle = one caller thread at a time, ENFORCED (round 6): while a long synchronous call is in flight, calls
    from a second thread on the same handle come back RF_ERR_INVALID_ARG with "handle in use by another thread" and do not disturb the call
    in flight (same detections as an undisturbed run); afterwards the handle works from any thread."""
    import ctypes as C
    import threading
    from retinaface_amd import _lib
    from retinaface_amd.frames import synth_frames
    frames = synth_frames(448, 448, 256, config=5)
    det = engine(rfa, "mnet25", FP16, (448, 448), max_batch=32)
    quiet = _key(det.detectBatchImages(frames, 0.5))
    lib = _lib.load_library()
    refused, other, stop = [], [], threading.Event()

    def intruder():
        counts = (C.c_int * 8)()
        while not stop.is_set():
            st = lib.rf_last_candidate_counts(det._h, counts, 8)
            if st == _lib.RF_ERR_INVALID_ARG:
                refused.append(lib.rf_last_error(det._h).decode())
            else:
                other.append(st)

    th = threading.Thread(target=intruder)
    th.start()
    try:
        busy = [_key(det.detectBatchImages(frames, 0.5)) for _ in range(3)]
    finally:
        stop.set()
        th.join()
    assert all(b == quiet for b in busy)
    assert refused and all("in use by another thread" in r for r in refused), (len(refused), len(other))
    assert all(st >= 0 for st in other), sorted(set(other))
    counts = (C.c_int * 8)()
    assert lib.rf_last_candidate_counts(det._h, counts, 8) >= 0          # the refusal left no state behind
    assert _key(det.detectBatchImages(frames[:8], 0.5)) == quiet[:8]
    det.close()


def test_candidate_overflow_is_reported(rfa, crop448):
    det = engine(rfa, "mnet25", FP16, (448, 448), max_candidates=64, max_detections=4)
    got = det.detect(crop448, 0.001)
    assert det.truncated and det.last_candidate_counts(1)[0] > 64 and len(got) <= 4
    got = det.detect(crop448, 0.999999)
    assert not det.truncated and got == []


@pytest.mark.parametrize("prec", [FP32, FP16])
def test_edge_cases_empty_small_strided_and_chunked(rfa, oracles, crop448, prec):
    from retinaface_amd.frames import synth_frames
    det = engine(rfa, "mnet-deconv-0517", prec, (448, 448))
    frames = synth_frames(448, 448, 11, config=5)
    single = [det.detect(f, 0.5) for f in frames]
    # 11 images with max_batch 8 -> chunked 8 + 3 (the reference overruns its buffers here, trtretinafacenet.cpp:21)
    batch = det.detectBatchImages(frames, 0.5)
    assert [[d.anchor_index for d in r] for r in batch] == [[d.anchor_index for d in r] for r in single]
    assert all(a.rect == b.rect and a.score == b.score for ra, rb in zip(batch, single) for a, b in zip(ra, rb))
    # empty Mat inside a batch: count 0, neighbours untouched (img.empty() early-out, RetinaFace.cpp:578-580)
    mixed = det.detectBatchImages([frames[0], np.zeros((0, 0, 3), np.uint8), None, frames[1]], 0.5)
    assert mixed[1] == [] and mixed[2] == []
    assert [d.anchor_index for d in mixed[0]] == [d.anchor_index for d in single[0]]
    assert [d.anchor_index for d in mixed[3]] == [d.anchor_index for d in single[1]]
    assert det.detect(None) == [] and det.detectBatchImages([]) == []
    # a frame smaller than the net sits top-left on a zero canvas (scale factor clamps to 1)
    small = np.ascontiguousarray(crop448[:300, :400])
    ref = oracles["mnet-deconv-0517"].detect(small, 0.5, 0.4, net_hw=(448, 448))
    compare(det.detect(small, 0.5), ref.rows(), ref.anchor_indices(), prec, twins_of_result(ref, NMS_THRESHOLD, SCORE_NOISE))
    # non-contiguous rows (cv::Mat ROI: step > cols*3)
    wide = np.zeros((448, 600, 3), np.uint8)
    wide[:, :448] = crop448
    view = wide[:, :448]
    assert view.strides[0] == 1800
    a, b = det.detect(view, 0.5), det.detect(crop448, 0.5)
    assert [d.anchor_index for d in a] == [d.anchor_index for d in b] and all(x.rect == y.rect for x, y in zip(a, b))


@pytest.mark.parametrize("prec", [FP32, FP16])
def test_odd_net_size_partial_tiles_and_unaligned_frames(rfa, oracles, base_frame, prec):
    """A net size whose feature maps are not multiples of any tile (352 x 608 -> 176x304 ... 11x19): every kernel has partial
    tiles on the right / bottom, the persistent tile walk wraps rows and images at odd counts, and the hardware-range-check
    padding of the buffer descriptors is exercised on all four borders.  Frames: net-sized, smaller than the net, and a view at
    an odd byte offset (frame pointer not dword aligned, row step not a multiple of 4)."""
    hw = (352, 608)
    det = engine(rfa, "mnet-deconv-0517", prec, hw)
    od = oracles["mnet-deconv-0517"]
    full = np.ascontiguousarray(base_frame[100:100 + hw[0], 400:400 + hw[1]])
    small = np.ascontiguousarray(base_frame[100:100 + 301, 400:400 + 517])
    backing = np.zeros((hw[0], hw[1] * 3 + 7), np.uint8)                   # odd row step
    flat = backing.reshape(-1)[1:1 + (hw[0] - 1) * backing.shape[1] + hw[1] * 3]
    odd = np.lib.stride_tricks.as_strided(flat, shape=(hw[0], hw[1], 3), strides=(backing.shape[1], 3, 1))
    odd[:] = full
    assert odd.ctypes.data % 4 == 1 and odd.strides[0] % 4 != 0
    for name, frame in (("full", full), ("small", small), ("odd", odd)):
        ref = od.detect(np.ascontiguousarray(frame), 0.5, 0.4, net_hw=hw)
        assert len(ref.detections) >= 2, name
        compare(det.detect(frame, 0.5), ref.rows(), ref.anchor_indices(), prec, twins_of_result(ref, NMS_THRESHOLD, SCORE_NOISE))
    # 19 frames in one call: chunks of 8 + 8 + 3 coalesced into launches whose tile counts are odd multiples
    many = [full, small, np.ascontiguousarray(odd)] * 6 + [full]
    want = [od.detect(f, 0.5, 0.4, net_hw=hw).anchor_indices().tolist() for f in (full, small, np.ascontiguousarray(odd))]
    got = det.detectBatchImages(many, 0.5)
    assert [[d.anchor_index for d in r] for r in got] == (want * 6 + [want[0]])


@pytest.mark.parametrize("prec", [FP32, FP16])
def test_pad32_variant_is_the_reference_caffe_build_detect(rfa, oracles, base_frame, prec):
    """rf_detect_batch_pad32 = `RetinaFace::detect(Mat img, ...)` of the Caffe build (RetinaFace.cpp:943-1075): no fixed net
    size, pad to x32, per-size anchors, clip to the padded size.  Checked against oracle.pipeline's Caffe variant on the
    reference's own image at its native 1280x886 and on odd crops, mixed in one call (engines are pooled per size)."""
    from retinaface_amd.frames import load_base_frame
    det = engine(rfa, "mnet-deconv-0517", prec, (448, 448))
    od = oracles["mnet-deconv-0517"]
    native = load_base_frame()                                   # data/img.jpg as shipped: 886 x 1280
    assert native.shape[0] % 32 != 0
    frames = [native, np.ascontiguousarray(native[100:433, 380:901]), np.ascontiguousarray(native[60:380, 300:811]),
              np.ascontiguousarray(native[100:433, 380:901][:, ::-1])]
    got = det.detect_pad32(frames + [None], 0.5)
    assert got[4] == []
    t = TOL[prec]
    for f, g in zip(frames, got):
        ref = od.detect(f, 0.5, 0.4)                             # net_hw=None: the Caffe variant
        assert len(ref.detections) >= 1 and len(g) == len(ref.detections), (f.shape, len(g), len(ref.detections))
        for a, r in zip(g, ref.rows()):
            assert iou_plus1(a.rect, r[1:5]) >= 1 - t["iou"] and abs(a.score - r[0]) <= score_tol(prec, r[0])
            assert a.rect[2] <= (f.shape[1] + 31) // 32 * 32 - 1 and a.rect[3] <= (f.shape[0] + 31) // 32 * 32 - 1


def _key(res):
    return [[(d.anchor_index, d.as_row().tobytes()) for d in r] for r in res]


def test_prepared_device_batches_pipeline(rfa):
    """prepare_device_batch / enqueue_prepared (descriptor arrays built once): results equal the plain enqueue path, for
    more tickets in flight than one launch holds."""
    import torch
    from retinaface_amd.frames import synth_frames
    det = engine(rfa, "mnet25", FP16, (448, 448))
    frames = synth_frames(448, 448, 8, config=3)
    d = torch.from_numpy(np.stack(frames)).cuda()
    ptrs = [d[i].data_ptr() for i in range(8)]
    want = _key(det.detect_device(ptrs, [448] * 8, [448] * 8, 0.5))
    assert sum(len(r) for r in want) >= 8
    batch = det.prepare_device_batch(ptrs, [448] * 8, [448] * 8)
    tickets = [det.enqueue_prepared(batch, 0.5) for _ in range(det.num_slots())]
    for t in tickets:
        assert _key(det.wait(t, 8)) == want          # every byte of every detection, not only the anchors: this is the bench's enqueue path


def test_device_resident_and_async_entry_points(rfa):
    import torch
    from retinaface_amd.frames import synth_frames
    det = engine(rfa, "mnet25", FP16, (448, 448))
    frames = synth_frames(448, 448, 8, config=1)
    host = det.detectBatchImages(frames, 0.5)
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() for t in dev]
    res = det.detect_device(ptrs, [448] * 8, [448] * 8, 0.5)
    assert [[(d.anchor_index, d.rect, d.score) for d in r] for r in res] == [[(d.anchor_index, d.rect, d.score) for d in r] for r in host]
    # asynchronous ring: more batches in flight than slots
    tickets = []
    outs = []
    for k in range(2 * det.num_slots() + 1):
        if len(tickets) == det.num_slots():
            t, n = tickets.pop(0)
            outs.append(det.wait(t, n))
        n = 1 + k % 8
        tickets.append((det.enqueue_device(ptrs[:n], [448] * n, [448] * n, 0.5), n))
    while tickets:
        t, n = tickets.pop(0)
        outs.append(det.wait(t, n))
    for k, o in enumerate(outs):
        n = 1 + k % 8
        assert [[d.anchor_index for d in r] for r in o] == [[d.anchor_index for d in r] for r in host[:n]]


@pytest.mark.parametrize("prec", [FP16, INT8])
def test_synchronous_host_call_with_split_upload_is_byte_identical(rfa, prec):
    """Round 6: ONE synchronous rf_detect_batch of host frames (the reference's calling convention, RetinaFace.cpp:749-846) stages and sends its
    frames in pipelined pieces cut at row granularity (engine.cpp submit()).  Every byte of every detection must equal the device-frame call's and
    the one-piece engine's (RF_SYNC_SPLIT=0): batch 8 and ragged batch 5 at 448 x 448, an empty frame and a strided view inside the batch, the
    same frames from memory pinned with rf_host_register, and ONE 1280 x 896 frame (pieces are row ranges of it)."""
    import torch
    from retinaface_amd.frames import synth_frames
    frames = synth_frames(448, 448, 8, config=23)
    wide = np.zeros((448, 500, 3), np.uint8)
    wide[:, :448] = frames[2]
    mixed = [frames[0], None, wide[:, :448], frames[3], frames[4]]
    os.environ["RF_SYNC_SPLIT"] = "0"
    try:
        plain = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=prec, net_hw=(448, 448), model_stem="mnet25", max_batch=8)
    finally:
        os.environ.pop("RF_SYNC_SPLIT", None)
    det = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=prec, net_hw=(448, 448), model_stem="mnet25", max_batch=8)
    dev = torch.from_numpy(np.stack(frames)).cuda()
    want = _key(det.detect_device([dev[i].data_ptr() for i in range(8)], [448] * 8, [448] * 8, 0.5))
    assert sum(len(w) for w in want) >= 8
    for _ in range(3):
        assert _key(det.detectBatchImages(frames, 0.5)) == want == _key(plain.detectBatchImages(frames, 0.5))
        assert _key(det.detectBatchImages(frames[:5], 0.5)) == want[:5]
        assert _key(det.detectBatchImages(mixed, 0.5)) == [want[0], [], want[2], want[3], want[4]]
    pinned = np.stack(frames)
    det.host_register(pinned)
    for _ in range(2):
        assert _key(det.detectBatchImages([pinned[i] for i in range(8)], 0.5)) == want
        assert _key(det.detectBatchImages([pinned[i] for i in (6, 1, 3)], 0.5)) == [want[6], want[1], want[3]]
    det.host_unregister(pinned)
    # device-frame and asynchronous calls in between are untouched by the split state
    t = det.enqueue_host(frames[:4], 0.5)
    assert _key(det.wait(t, 4)) == want[:4] and _key(det.detectBatchImages(frames, 0.5)) == want
    det.close()
    plain.close()
    big = synth_frames(896, 1280, 2, config=24)
    os.environ["RF_SYNC_SPLIT"] = "0"
    try:
        plain = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=prec, net_hw=(896, 1280), model_stem="mnet25", max_batch=2)
    finally:
        os.environ.pop("RF_SYNC_SPLIT", None)
    det = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=prec, net_hw=(896, 1280), model_stem="mnet25", max_batch=2)
    ref = _key(plain.detectBatchImages(big, 0.5))
    assert sum(len(w) for w in ref) >= 2
    for _ in range(3):
        assert _key([det.detect(big[0], 0.5)]) == ref[:1]                  # one frame: row pieces
        assert _key(det.detectBatchImages(big, 0.5)) == ref
    det.close()
    plain.close()


def test_host_frames_async_pipeline_and_registered_buffers(rfa):
    """rf_enqueue_batch (frames in host memory, staged through pinned memory by the copy threads, one DMA per enqueue) and the
    rf_host_register path (DMA straight from the caller's pinned range): bit-identical to the synchronous host call, with more
    tickets in flight than the pipeline holds, ragged batch sizes, strided views and an empty frame."""
    from retinaface_amd.frames import synth_frames
    det = engine(rfa, "mnet25", FP16, (448, 448))
    frames = synth_frames(448, 448, 8, config=11)
    want = det.detectBatchImages(frames, 0.5)
    wide = np.zeros((448, 500, 3), np.uint8)
    wide[:, :448] = frames[1]
    cases = [frames, frames[:3], [frames[0], wide[:, :448], None, frames[3]]]
    exp = [_key(want), _key(want[:3]), [_key(want)[0], _key(want)[1], [], _key(want)[3]]]
    tickets, outs = [], []
    for k in range(3 * det.num_slots() + 2):
        if len(tickets) == det.num_slots():
            t, c = tickets.pop(0)
            outs.append((c, det.wait(t, len(cases[c]))))
        c = k % 3
        imgs = [f if f is not None else np.zeros((0, 0, 3), np.uint8) for f in cases[c]]
        tickets.append((det.enqueue_host(imgs, 0.5), c))
    while tickets:
        t, c = tickets.pop(0)
        outs.append((c, det.wait(t, len(cases[c]))))
    assert len(outs) == 3 * det.num_slots() + 2
    for c, o in outs:
        assert _key(o) == exp[c], c
    # caller memory pinned once: frames inside the range are read in place
    ring = np.stack(frames).copy()
    det.host_register(ring)
    try:
        t = [det.enqueue_host([ring[i] for i in range(8)], 0.5) for _ in range(5)]
        for x in t:
            assert _key(det.wait(x, 8)) == _key(want)
        assert _key(det.detectBatchImages([ring[i] for i in range(8)], 0.5)) == _key(want)
    finally:
        det.host_unregister(ring)
    with pytest.raises(rfa._lib.RFError):
        det.host_unregister(ring)                       # not registered any more
    # a frame far larger than the staging block of this engine: staging grows to what ONE enqueue needs
    small = engine(rfa, "mnet25", FP16, (448, 448), max_batch=1, lanes=1, coalesce=1)
    big = np.zeros((1792, 1792, 3), np.uint8)
    big[:448, :448] = frames[0]
    assert len(small.detect(big, 0.5)) >= 0 and not small.truncated


def test_multi_device_handle_shards_by_image(rfa):
    """rf_options.devices: one engine + host thread per entry, detectBatchImages split into contiguous ceil(n / G) slices
    (RetinaFace.cpp:749-940 is the call being sharded; per-image NMS :916-918 is why it shards).  The one-GPU box runs the
    sharding logic as two and three engines on device 0: results, anchor indices and candidate counts must equal the
    single-engine handle for every n, including n < G and slices that are chunked again inside an engine."""
    import torch
    from retinaface_amd.frames import synth_frames
    one = engine(rfa, "mnet25", FP16, (448, 448))
    frames = synth_frames(448, 448, 21, config=13)
    for devs in ([0, 0], [0, 0, 0]):
        multi = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=FP16, net_hw=(448, 448), model_stem="mnet25", devices=devs)
        assert multi.num_devices() == len(devs) and multi.num_slots() == len(devs) * one.num_slots()
        for n in (1, 2, 5, 8, 21):
            a = multi.detectBatchImages(frames[:n], 0.5)
            ca = multi.last_candidate_counts(n)
            b = one.detectBatchImages(frames[:n], 0.5)
            assert _key(a) == _key(b), (devs, n)
            assert ca == one.last_candidate_counts(n)
        assert multi.detectBatchImages([], 0.5) == []
        # device-resident frames and the asynchronous API (whole enqueues go round-robin over the engines)
        d = torch.from_numpy(np.stack(frames[:8])).cuda()
        ptrs = [d[i].data_ptr() for i in range(8)]
        want = _key(one.detect_device(ptrs, [448] * 8, [448] * 8, 0.5))
        assert _key(multi.detect_device(ptrs, [448] * 8, [448] * 8, 0.5)) == want
        t = [multi.enqueue_device(ptrs, [448] * 8, [448] * 8, 0.5) for _ in range(7)]
        for x in t:
            assert _key(multi.wait(x, 8)) == want
        t = [multi.enqueue_host(frames[:8], 0.5) for _ in range(4)]
        for x in t:
            assert _key(multi.wait(x, 8)) == want
        # an error in one slice surfaces as the call's error and leaves the handle usable
        with pytest.raises(rfa._lib.RFError):
            multi.detectBatchImages(frames[:5] + [np.zeros((4000, 4000, 3), np.uint8)], 0.5)
        assert _key(multi.detectBatchImages(frames[:5], 0.5)) == _key(one.detectBatchImages(frames[:5], 0.5))
        multi.close()
    with pytest.raises(rfa._lib.RFError):
        rfa.RetinaFace(ASSETS, "net3", 0.4, precision=FP16, net_hw=(448, 448), model_stem="mnet25", devices=[0, 99])


def test_device_frames_resident_elsewhere_are_scattered_to_the_slices_device(rfa):
    """The batch split of a multi-GPU node (north_star: "batches shard across the GPUs ... batch split / gather"): a frame handed
    to rf_detect_batch_device / rf_enqueue_batch_device that does not live on the engine's (slice's) device is copied to it with
    one peer copy over xGMI on the lane's stream (engine.cpp submit(): hipPointerGetAttributes + hipMemcpyPeerAsync); pointers
    the HIP runtime does not know are refused with RF_ERR_INVALID_ARG instead of faulting.  The one-GPU box runs that path with
    RF_FORCE_SCATTER=1 (every device frame treated as foreign: the peer copy is device 0 -> device 0): dense frames, a strided
    ROI at an unaligned address, single and [0, 0] multi handles, synchronous and asynchronous entry points -- results
    identical to reading the frames in place."""
    import torch
    from retinaface_amd.frames import synth_frames
    frames = synth_frames(448, 448, 12, config=17)
    d = torch.from_numpy(np.stack(frames)).cuda()
    ptrs = [d[i].data_ptr() for i in range(12)]
    one = engine(rfa, "mnet25", FP16, (448, 448))
    want = _key(one.detect_device(ptrs, [448] * 12, [448] * 12, 0.5))
    assert sum(len(r) for r in want) > 0
    # a strided ROI: 300 x 401 window of a wider device image, odd byte offset
    wide = torch.zeros((448, 448 * 3 + 13), dtype=torch.uint8, device="cuda")
    wide[:, 1:1 + 448 * 3] = d[0].reshape(448, -1)
    roi_ptr, roi_step = wide.data_ptr() + 1 + 3 * 20 + 50 * wide.shape[1], wide.shape[1]
    want_roi = _key(one.detect_device([roi_ptr], [300], [401], 0.5, steps=[roi_step]))
    os.environ["RF_FORCE_SCATTER"] = "1"
    try:
        for devs in (None, [0, 0]):
            det = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=FP16, net_hw=(448, 448), model_stem="mnet25", devices=devs)
            assert _key(det.detect_device(ptrs, [448] * 12, [448] * 12, 0.5)) == want          # chunked (max_batch 8) and, for [0, 0], sharded
            assert _key(det.detect_device([roi_ptr], [300], [401], 0.5, steps=[roi_step])) == want_roi
            t = [det.enqueue_device(ptrs[:8], [448] * 8, [448] * 8, 0.5) for _ in range(5)]
            for x in t:
                assert _key(det.wait(x, 8)) == want[:8]
            # host memory passed as a device pointer: refused, and the handle stays usable
            with pytest.raises(rfa._lib.RFError):
                det.detect_device([frames[0].ctypes.data], [448], [448], 0.5)
            assert _key(det.detect_device(ptrs[:3], [448] * 3, [448] * 3, 0.5)) == want[:3]
            # pinned host memory is device-accessible: read in place, not an error
            pinned = torch.from_numpy(frames[1]).pin_memory()
            assert _key(det.detect_device([pinned.data_ptr()], [448], [448], 0.5)) == want[1:2]
            det.close()
    finally:
        os.environ.pop("RF_FORCE_SCATTER", None)


def test_residency_cache_follows_freed_and_reallocated_frame_buffers(rfa):
    """Where a device frame lives is remembered per ALLOCATION (hipMemGetAddressRange) and re-validated against the runtime after 2 ms;
    rf_invalidate_residency() drops it at once (round 4 cached per 2 MiB page, for ever -- a buffer freed and re-allocated elsewhere kept
    its old answer, and a freed one was launched on instead of being refused).  One-GPU rehearsal with RF_FORCE_SCATTER=1 (the lookup is only made
    when frames could live elsewhere): a frame tensor in an allocation of its own is used, freed (torch.cuda.empty_cache() returns the segment to
    the runtime), and the stale pointer must then be REFUSED -- after an explicit invalidate, and after the TTL alone -- not handed to a peer copy;
    a new allocation (possibly at the same address) is looked up afresh and gives the right detections."""
    import time
    import torch
    from retinaface_amd.frames import synth_frames
    frames = np.stack(synth_frames(448, 448, 8, config=23))
    one = engine(rfa, "mnet25", FP16, (448, 448))
    want = _key(one.detectBatchImages(list(frames), 0.5))
    os.environ["RF_FORCE_SCATTER"] = "1"
    try:
        det = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=FP16, net_hw=(448, 448), model_stem="mnet25")
    finally:
        os.environ.pop("RF_FORCE_SCATTER", None)

    def own_allocation():
        # > 10 MB: the caching allocator gives it a segment of its own, which empty_cache() returns to the runtime once the tensor is gone
        t = torch.empty((64 << 20,), dtype=torch.uint8, device="cuda")
        t[:frames.size] = torch.from_numpy(frames.reshape(-1)).cuda()
        torch.cuda.synchronize()
        return t

    for how in ("invalidate", "ttl"):
        t = own_allocation()
        ptrs = [t.data_ptr() + i * 448 * 448 * 3 for i in range(8)]
        for _ in range(3):                                   # first call looks the allocation up, the others are served from the cache
            assert _key(det.detect_device(ptrs, [448] * 8, [448] * 8, 0.5)) == want
        del t
        torch.cuda.empty_cache()
        if how == "invalidate":
            det.invalidate_residency()
        else:
            time.sleep(0.02)                                 # > the 2 ms TTL: the entry is re-validated before it is trusted again
        with pytest.raises(rfa._lib.RFError):
            det.detect_device(ptrs[:1], [448], [448], 0.5)
        # the handle stays usable, and a new allocation is looked up afresh
        t2 = own_allocation()
        ptrs2 = [t2.data_ptr() + i * 448 * 448 * 3 for i in range(8)]
        assert _key(det.detect_device(ptrs2, [448] * 8, [448] * 8, 0.5)) == want
        del t2
        torch.cuda.empty_cache()
        det.invalidate_residency()
    det.close()


def test_configs4_rehearsal_eight_engines_share_the_one_gpu(rfa):
    """BASELINE.json configs[4] (mnet25 int8, 448 x 448, batch 256 sharded over 8 GPUs) through the LIBRARY on the one-GPU box: a handle
    over devices [0] * 8 -- eight engines, eight host threads, contiguous slices of 32 -- takes ONE rf_detect_batch_device call of 256
    device-resident frames with RF_FORCE_SCATTER=1 (every frame is treated as living on another GPU: each engine pulls its 32 frames
    with peer copies on its lane's stream before it launches).  Same detections, anchors and candidate counts as the single engine
    (which chunks the 256 into super-batches itself).  What this cannot show is a second physical device: see SCALE_rNN.json."""
    import torch
    from retinaface_amd.frames import synth_frames
    frames = synth_frames(448, 448, 64, config=19)
    d = torch.from_numpy(np.stack(frames)).cuda()
    ptrs = [d[i % 64].data_ptr() for i in range(256)]
    one = engine(rfa, "mnet25", INT8, (448, 448), max_batch=32)
    want = _key(one.detect_device(ptrs, [448] * 256, [448] * 256, 0.5))
    want_nc = one.last_candidate_counts(256)
    assert sum(len(r) for r in want) >= 256
    os.environ["RF_FORCE_SCATTER"] = "1"
    try:
        multi = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=INT8, net_hw=(448, 448), model_stem="mnet25", max_batch=32, devices=[0] * 8)
        assert multi.num_devices() == 8
        for rep in range(2):                                   # the second call reuses the eight lanes' staging blocks
            before = multi.scatter_stats()
            assert _key(multi.detect_device(ptrs, [448] * 256, [448] * 256, 0.5)) == want
            assert multi.last_candidate_counts(256) == want_nc
            # round 6: a contiguous slice crosses as ONE peer copy (rf_scatter_stats): 256 frames travelled in 8 copies, not 256
            after = multi.scatter_stats()
            assert after["frames"] - before["frames"] == 256 and after["peer_copies"] - before["peer_copies"] == 8, (before, after)
        # ragged: 250 images = 7 slices of 32 + one of 26
        assert _key(multi.detect_device(ptrs[:250], [448] * 250, [448] * 250, 0.5)) == want[:250]
        multi.close()
        # ... and RF_SCATTER_PER_FRAME=1 (the A/B leg of bench.py's library_multi_device.split_ab) keeps rounds 3-5's one copy per frame
        os.environ["RF_SCATTER_PER_FRAME"] = "1"
        per_frame = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=INT8, net_hw=(448, 448), model_stem="mnet25", max_batch=32, devices=[0] * 8)
        assert _key(per_frame.detect_device(ptrs, [448] * 256, [448] * 256, 0.5)) == want
        assert per_frame.scatter_stats() == {"frames": 256, "peer_copies": 256}
        per_frame.close()
    finally:
        os.environ.pop("RF_FORCE_SCATTER", None)
        os.environ.pop("RF_SCATTER_PER_FRAME", None)


def test_bench_strong_mode_with_real_engines_two_ranks_on_the_one_gpu(rfa):
    """bench.py's strong-scaling path (one global batch per step split by shard_range, the result gather inside the timed region) with
    REAL engines: two ranks share the one GPU (--oversubscribe: RCCL refuses two ranks on a device, so the records travel over gloo --
    everything else, launcher included, is what `--gpus 8 --global-batch 256` runs on the 8-GPU node)."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--global-batch", "64", "--precision",
                          "int8", "--steps", "20", "--warmup", "5", "--min-seconds", "0.3", "--regions", "1", "--no-cpu-baseline", "--host-seconds", "0",
                          "--no-pmc", "--profile-iters", "5", "--ring-mb", "80"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["global_batch"] == 64 and j["dtype"] == "i8"
    g = j["result_gather"]
    assert g["records_gathered"] == g["expected"] == 64 * j["steps"] and g["ranks_in_communicator"] == 2 and g["backend"] == "gloo"
    assert j["value"] > 0 and abs(j["images_per_sec"] * j["timed_seconds"] - 64 * j["steps"]) < 1e-3 * 64 * j["steps"]


@pytest.mark.parametrize("prec", [FP16, INT8])
def test_raw_row_staging_short_frames_and_mixed_batches(rfa, prec):
    """Round 6: the stems stage frames whose base and pitch are multiples of 16 B and that fill the net's width as raw rows by LDS-DMA (kernels.hip
    stem2_kernel / stem_kernel `raw`), everything else through the general path -- chosen per workgroup, so ONE launch may mix both.  A frame that
    fills the width but not the height relies on the descriptor's range check for its missing rows; a frame one pixel narrower, the same frame at a
    pointer that is 4 but not 16 bytes aligned, and an ROI with a 16-byte pitch inside a wider image must all give exactly what the zero-padded
    full-size frame gives (byte for byte: both paths run the same arithmetic only when K order does not matter, so the comparison is made
    against the SAME path where it can be and against the 208-frame tolerance otherwise)."""
    import torch
    from retinaface_amd.frames import synth_frames
    H = W = 448
    det = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=prec, net_hw=(H, W), model_stem="mnet25", max_batch=8)
    full = synth_frames(H, W, 4, config=31)
    short = [np.ascontiguousarray(f[:300]) for f in full]                          # fills the width, 300 of 448 rows: raw path + range check
    padded = [np.concatenate([f, np.zeros((H - 300, W, 3), np.uint8)]) for f in short]
    want = _key(det.detectBatchImages(padded, 0.5))
    assert sum(len(w) for w in want) >= 3
    assert _key(det.detectBatchImages(short, 0.5)) == want                         # host frames are staged 256-byte aligned: raw path
    d_pad = torch.from_numpy(np.stack(padded)).cuda()
    d_short = torch.from_numpy(np.stack(short)).cuda()
    assert _key(det.detect_device([d_pad[i].data_ptr() for i in range(4)], [H] * 4, [W] * 4, 0.5)) == want
    assert _key(det.detect_device([d_short[i].data_ptr() for i in range(4)], [300] * 4, [W] * 4, 0.5)) == want
    # an ROI that fills the width of the NET but sits in a wider device image with a 16-byte pitch: raw path with step > cols * 3
    wide = torch.zeros((300, 464 * 3), dtype=torch.uint8, device="cuda")            # pitch 1392 = 16 x 87
    wide[:, :W * 3] = d_short[1].reshape(300, W * 3)
    assert _key(det.detect_device([wide.data_ptr()], [300], [W], 0.5, steps=[464 * 3])) == want[1:2]
    # the general path on the same pixels: pointer 4 (not 16) bytes aligned / one pixel narrower.  Different conv0 K order => compare like the contract
    buf = torch.zeros(4 + 300 * W * 3, dtype=torch.uint8, device="cuda")
    buf[4:] = d_short[2].reshape(-1)
    got = det.detect_device([buf.data_ptr() + 4], [300], [W], 0.5)
    ref = det.detect_device([d_short[2].data_ptr()], [300], [W], 0.5)
    assert len(got[0]) == len(ref[0]) >= 1
    if prec == FP16:
        assert [d.anchor_index for d in got[0]] == [d.anchor_index for d in ref[0]]
        assert all(iou_plus1(a.rect, b.rect) >= 1 - 1e-3 for a, b in zip(got[0], ref[0]))
    else:           # int8: a 1-LSB difference of a first-layer quantum may move an NMS winner to the neighbouring anchor (tests/int8_contract.py)
        assert all(max(iou_plus1(a.rect, b.rect) for a in got[0]) >= 0.85 for b in ref[0])
    # one launch mixing raw and general workgroups: results per image are those of the separate calls
    mixed_ptrs, mixed_rows = [d_short[0].data_ptr(), buf.data_ptr() + 4, d_pad[3].data_ptr()], [300, 300, H]
    mixed = det.detect_device(mixed_ptrs, mixed_rows, [W] * 3, 0.5)
    assert _key(mixed[:1]) == want[:1] and _key(mixed[2:]) == want[3:] and _key(mixed[1:2]) == _key(got)
    det.close()


def test_device_frames_unaligned_pointer_odd_step_and_roi(rfa, oracles, base_frame):
    """Device-resident frames exactly as the stem's descriptor path sees them: a frame pointer that is not dword aligned, a row
    step that is not a multiple of 4, and an ROI of a larger device image (step > cols*3, last row ends before the allocation
    does).  Host frames never exercise this: they are repacked into aligned staging memory."""
    import torch
    hw = (352, 608)
    od = oracles["mnet-deconv-0517"]
    for prec in (FP16, FP32):
        det = engine(rfa, "mnet-deconv-0517", prec, hw)
        full = np.ascontiguousarray(base_frame[100:100 + hw[0], 400:400 + hw[1]])
        ref = od.detect(full, 0.5, 0.4, net_hw=hw)
        assert len(ref.detections) >= 2
        # (a) odd base address + odd step: the frame starts 1 byte into a buffer whose rows are cols*3 + 7 bytes apart
        step = hw[1] * 3 + 7
        back = torch.zeros(hw[0] * step + 64, dtype=torch.uint8, device="cuda")
        host = np.zeros((hw[0], step), np.uint8)
        host[:, :hw[1] * 3] = full.reshape(hw[0], -1)
        back[1:1 + hw[0] * step] = torch.from_numpy(host.reshape(-1)).cuda()
        ptr = back.data_ptr() + 1
        assert ptr % 4 == 1 and step % 4 != 0
        compare(det.detect_device([ptr], [hw[0]], [hw[1]], 0.5, steps=[step])[0], ref.rows(), ref.anchor_indices(), prec, twins_of_result(ref, NMS_THRESHOLD, SCORE_NOISE))
        # (b) ROI of a larger device image, placed so that the ROI's last row is the image's last row and ends well before the
        # end of that row: a descriptor sized rows*step would reach past the allocation's end
        big = np.zeros((hw[0] + 40, hw[1] + 100, 3), np.uint8)
        big[40:, 30:30 + hw[1]] = full
        dbig = torch.from_numpy(big).cuda()
        roi_ptr = dbig.data_ptr() + (40 * big.shape[1] + 30) * 3
        compare(det.detect_device([roi_ptr], [hw[0]], [hw[1]], 0.5, steps=[big.shape[1] * 3])[0], ref.rows(), ref.anchor_indices(), prec, twins_of_result(ref, NMS_THRESHOLD, SCORE_NOISE))
        torch.cuda.synchronize()


def _widen_heads_to_4_anchors(net):
    """A 4-anchors-per-cell model for the "net3a" preset (the reference ships none): every head of the mnet25 graph is widened
    from A = 2 to A = 4 -- anchors 2, 3 reuse the filters of anchors 0, 1 with their class logits scaled by 0.9 (no score ties,
    which std::sort would order arbitrarily in the reference).  cls layout is [background A | foreground A]."""
    import copy
    net = copy.deepcopy(net)
    for s in (32, 16, 8):
        cls = net.layer(f"face_rpn_cls_score_stride{s}")
        w, b = cls.blobs[0], cls.blobs[1].reshape(-1)
        bg, fg = w[0:2], w[2:4]
        cls.blobs[0] = np.concatenate([bg, 0.9 * bg, fg, 0.9 * fg]).astype(np.float32)
        cls.blobs[1] = np.concatenate([b[0:2], 0.9 * b[0:2], b[2:4], 0.9 * b[2:4]]).astype(np.float32).reshape(cls.blobs[1].shape[:-1] + (8,))
        cls.num_output = 8
        back = net.layer(f"face_rpn_cls_prob_reshape_stride{s}")     # Reshape(., 2A, -1, .) after the 2-class softmax
        back.reshape_dims = [8 if d == 4 else d for d in back.reshape_dims]
        for name, per in ((f"face_rpn_bbox_pred_stride{s}", 8), (f"face_rpn_landmark_pred_stride{s}", 20)):
            l = net.layer(name)
            l.blobs[0] = np.concatenate([l.blobs[0], l.blobs[0]]).astype(np.float32)
            b1 = l.blobs[1].reshape(-1)
            l.blobs[1] = np.concatenate([b1, b1]).astype(np.float32).reshape(l.blobs[1].shape[:-1] + (2 * per,))
            l.num_output = 2 * per
    return net


def test_network_presets(rfa, nets, crop448, tmp_path):
    """The constructor's `network` argument (RetinaFace.cpp:209-271).  Presets the reference leaves without anchors construct and
    find nothing; "net3a" (ratios {1, 1.5}: 4 anchors per cell) refuses a 2-anchor model (the reference would read past its score
    blob) and, on a model widened to 4 anchors, decodes exactly what the reference's own postProcess decodes from the same
    blobs -- same anchor indices (now over 4 x h x w per stride), fp32 boxes within 1e-5 IoU."""
    from oracle import build_ref
    from oracle import retinaface_post as post
    from oracle.caffe_forward import CaffeNet
    from oracle.caffe_io import write_rfw
    for preset in ("ssh", "vgg", "net5", "net6", "typo"):
        det = rfa.RetinaFace(ASSETS, preset, 0.4, precision=FP16, net_hw=(448, 448), model_stem="mnet25")
        assert det.detect(crop448, 0.5) == [] and det.last_candidate_counts(1) == [0]
        assert det.detectBatchImages([crop448] * 3, 0.01) == [[], [], []]
        det.close()
    with pytest.raises(rfa._lib.RFError) as e:
        rfa.RetinaFace(ASSETS, "net3a", 0.4, precision=FP16, net_hw=(448, 448), model_stem="mnet25")
    assert e.value.status == rfa._lib.RF_ERR_MODEL
    wide = _widen_heads_to_4_anchors(nets["mnet25"])
    write_rfw(wide, str(tmp_path / "mnet25.rfw"))
    with pytest.raises(rfa._lib.RFError):                           # and the other way round: a 4-anchor model under "net3"
        rfa.RetinaFace(str(tmp_path), "net3", 0.4, precision=FP32, net_hw=(448, 448), model_stem="mnet25")
    ratios = post.preset_ratios("net3a")
    blobs = CaffeNet(wide).forward(preprocess_trt_identity(crop448, 448, 448))
    heads = {n: blobs[n] for st in HEAD_STRIDES for n in head_names(st)}
    assert heads["face_rpn_cls_prob_reshape_stride32"].shape[1] == 8
    for thr in (0.5, 0.05):
        want = post.nms(list(post.decode(heads, 448, 448, thr, ratios=ratios)), 0.4)
        assert len(want) >= 2
        if build_ref.available():                                   # the reference's own decode + NMS on the same blobs
            ref = build_ref.ReferenceRetinaFace(448, 448, max_batch=1, network="net3a", head_anchors=4)
            ref.set_heads(0, [heads[n][0] for n in build_ref.HEAD_BLOBS])
            faces = ref.postprocess(0, thr)
            ref.close()
            assert np.array_equal(faces, np.stack([d.as_row() for d in want]))
        for prec in (FP32, FP16):
            det = rfa.RetinaFace(str(tmp_path), "net3a", 0.4, precision=prec, net_hw=(448, 448), model_stem="mnet25", keep_outputs=True)
            got = det.detect(crop448, thr)
            compare(got, np.stack([d.as_row() for d in want]), [d.anchor_index for d in want], prec)
            if prec == FP32:
                for n in heads:
                    assert np.abs(det.get_output(n) - heads[n][0]).max() <= 5e-5, n
            det.close()


def test_engine_from_the_plan_cache_equals_engine_from_the_model(rfa, tmp_path):
    """Warm start (packed weight image read back from <stem>.<precision>.rfplan) gives bit-identical detections to a cold start and
    to an engine that never touches the cache, in every precision."""
    import shutil
    from retinaface_amd.frames import synth_frames
    shutil.copy(os.path.join(ASSETS, "mnet25.rfw"), tmp_path / "mnet25.rfw")
    frames = synth_frames(448, 448, 4, config=23)
    for prec, name in ((FP16, "fp16"), (FP32, "fp32"), (INT8, "int8")):
        nocache = rfa.RetinaFace(str(tmp_path), "net3", 0.4, precision=prec, net_hw=(448, 448), model_stem="mnet25", plan_cache=False)
        want = _key(nocache.detectBatchImages(frames, 0.5))
        nocache.close()
        assert not (tmp_path / f"mnet25.{name}.rfplan").exists()
        for run in ("cold", "warm"):
            det = rfa.RetinaFace(str(tmp_path), "net3", 0.4, precision=prec, net_hw=(448, 448), model_stem="mnet25")
            assert (tmp_path / f"mnet25.{name}.rfplan").exists()
            assert _key(det.detectBatchImages(frames, 0.5)) == want, (name, run)
            det.close()
        assert rfa._lib.load_library().rf_plan_cache_probe(str(tmp_path).encode(), b"mnet25", prec, None, None) == 1


def test_error_paths_leave_the_handle_usable(rfa):
    """A bad frame in a LATER chunk of a synchronous call (the earlier chunks are already in flight), repeated more often than the
    ticket pool is deep; rf_wait with a NULL result array; enqueue of more than max_batch images."""
    import ctypes as C
    from retinaface_amd.frames import synth_frames
    det = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=FP16, net_hw=(448, 448), model_stem="mnet25")
    frames = synth_frames(448, 448, 20, config=17)
    good = _key(det.detectBatchImages(frames, 0.5))
    lib = det._lib
    n = 20
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    rows, cols = (C.c_int * n)(*[448] * n), (C.c_int * n)(*[448] * n)
    steps = (C.c_int * n)(*([448 * 3] * 19 + [100]))                # last frame (third chunk): step < cols*3
    out = (rfa._lib.rf_face * (n * 256))()
    counts = (C.c_int * n)()
    for _ in range(4 * det.num_slots()):
        assert lib.rf_detect_batch(det._h, ptrs, rows, cols, steps, n, 0.5, out, 256, counts) == rfa._lib.RF_ERR_INVALID_ARG
    assert _key(det.detectBatchImages(frames, 0.5)) == good
    import torch
    d = torch.from_numpy(np.stack(frames[:8])).cuda()
    t = det.enqueue_device([d[i].data_ptr() for i in range(8)], [448] * 8, [448] * 8, 0.5)
    assert lib.rf_wait(det._h, t, None, 256, counts) == rfa._lib.RF_ERR_INVALID_ARG        # NULL out with cap > 0
    assert lib.rf_wait(det._h, t, out, 256, counts) == 0 and list(counts[:8]) == [len(r) for r in good[:8]]
    with pytest.raises(rfa._lib.RFError):
        det.enqueue_device([d[0].data_ptr()] * 9, [448] * 9, [448] * 9, 0.5)
    # calls from another host thread bind the engine's device themselves
    import threading
    res = {}
    th = threading.Thread(target=lambda: res.update(r=_key(det.detectBatchImages(frames, 0.5))))
    th.start()
    th.join()
    assert res["r"] == good
    det.close()
    # a super-batch that could only end in an out-of-memory error is refused as an argument error, and a batch-1 caller may
    # coalesce up to 256 enqueues (the cap was 32)
    with pytest.raises(rfa._lib.RFError):
        rfa.RetinaFace(ASSETS, "net3", 0.4, precision=FP16, net_hw=(448, 448), model_stem="mnet25", max_batch=64, coalesce=256)
    one = rfa.RetinaFace(ASSETS, "net3", 0.4, precision=FP16, net_hw=(448, 448), model_stem="mnet25", max_batch=1)
    assert one.num_slots() == 3 * 256
    tickets = [one.enqueue_device([d[i % 8].data_ptr()], [448], [448], 0.5) for i in range(300)]       # more than one super-batch
    got = [one.wait(t, 1) for t in tickets]
    assert [_key([g[0]])[0] for g in got[:8]] == [good[i] for i in range(8)] and _key([got[299][0]])[0] == good[299 % 8]
    one.close()


@pytest.mark.parametrize("prec", [FP32, FP16])
def test_determinism_and_graph_equals_eager(rfa, prec):
    from retinaface_amd.frames import synth_frames
    frames = synth_frames(448, 448, 8, config=2)
    g = engine(rfa, "mnet25", prec, (448, 448))
    e = engine(rfa, "mnet25", prec, (448, 448), use_graph=False)
    first = g.detectBatchImages(frames, 0.5)       # first call of a batch size is eager, later ones replay the graph
    for _ in range(3):
        again = g.detectBatchImages(frames, 0.5)
        assert [[d.as_row().tobytes() for d in r] for r in again] == [[d.as_row().tobytes() for d in r] for r in first]
    eager = e.detectBatchImages(frames, 0.5)
    assert [[d.as_row().tobytes() for d in r] for r in eager] == [[d.as_row().tobytes() for d in r] for r in first]
    t = e.last_timings()
    assert t["pre_ms"] >= 0 and t["infer_ms"] > 0 and t["post_ms"] > 0


def test_full_size_batch32_is_batch_composition_invariant(rfa):
    """BASELINE config 3's size (448x448, batch 32): each image's result is independent of what else is in the batch
    and of its position -- compared with the batch-8 engine and with a permuted batch."""
    from retinaface_amd.frames import synth_frames
    frames = synth_frames(448, 448, 32, config=7)
    big = engine(rfa, "mnet-deconv-0517", FP16, (448, 448), max_batch=32)
    small = engine(rfa, "mnet-deconv-0517", FP16, (448, 448))
    a = big.detectBatchImages(frames, 0.5)
    b = small.detectBatchImages(frames, 0.5)
    key = lambda res: [[(d.anchor_index, d.as_row().tobytes()) for d in r] for r in res]  # noqa: E731
    assert key(a) == key(b)
    perm = np.random.default_rng(0).permutation(32)
    c = big.detectBatchImages([frames[i] for i in perm], 0.5)
    assert key(c) == [key(a)[i] for i in perm]
    assert sum(len(r) for r in a) >= 32


def test_oversize_frame_is_area_averaged_down(rfa, base_frame):
    """Frames larger than the net (factor < 1, resizeconvertion.cu:298-311).  NPP's SUPER filter is closed source
    ("parity unpinned"); this checks the kernel against its own definition: an exact 2x downscale = 2x2 box mean,
    so detecting on the 2x nearest-upsampled frame must equal detecting on the original."""
    det = engine(rfa, "mnet-deconv-0517", FP32, (448, 448))
    from retinaface_amd.frames import synth_frames
    f = synth_frames(448, 448, 1, config=9)[0]
    up = np.repeat(np.repeat(f, 2, axis=0), 2, axis=1)
    a, b = det.detect(up, 0.5), det.detect(f, 0.5)
    assert len(a) == len(b) > 0 and [d.anchor_index for d in a] == [d.anchor_index for d in b]
    assert all(x.rect == y.rect for x, y in zip(a, b))
    # aspect-preserving, top-left anchored: a wide 2:1 frame only fills the top half of the canvas
    wide = np.concatenate([up, up], axis=1)                     # 896 x 1792 -> factor 0.25 -> 224 x 448
    c = det.detect(wide, 0.3)
    assert all(d.rect[3] <= 224 + 16 for d in c)
    # the factor that maps results back to source pixels (`scale`, RetinaFace.cpp:585-589 / :732-739)
    assert det.frame_scale(896, 896) == 2.0 and det.frame_scale(896, 1792) == 4.0 and det.frame_scale(300, 448) == 1.0
    assert all(abs(x.rect[0] * 2.0 - 2 * y.rect[0]) < 1e-6 for x, y in zip(a, b))


def test_oversize_frames_pixel_exact_against_the_oracles(rfa, oracles, base_frame):
    """What an oversize frame is turned into before the network sees it, compared PIXEL BY PIXEL (debug blob "input_canvas"):
      * oversize_resize="bilinear" -- the reference's build without NPP: cv::resize + one-sided padding (RetinaFace.cpp:585-620).
        Bit-exact against oracle.retinaface_post.preprocess_trt_cvresize, which tests/test_reference_pin.py holds equal to the
        reference's own detect() (through the shim's stand-in for OpenCV); detections equal the oracle pipeline's.
      * oversize_resize="area" (default) -- the NPP build; NPPI_INTER_SUPER is closed source, so the bar is agreement with an
        independently written statement of the definition (coverage matrices, float64; the kernel accumulates in fp32): every
        pixel within 1 level, > 99.5 % identical."""
    from oracle import retinaface_post as post
    rng = np.random.default_rng(21)
    od = oracles["mnet-deconv-0517"]
    frames = [np.ascontiguousarray(base_frame[:886]),                                       # 886 x 1280 photo -> 448 wide
              np.ascontiguousarray(np.repeat(np.repeat(base_frame[200:648, 380:828], 2, 0), 2, 1)),      # 896^2 -> 448^2 exactly
              rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8),
              rng.integers(0, 256, (450, 449, 3), dtype=np.uint8),                          # barely oversize
              np.ascontiguousarray(base_frame[100:400, 300:700])]                          # fits: 1:1 in both modes
    for mode in ("bilinear", "area"):
        det = engine(rfa, "mnet-deconv-0517", FP32, (448, 448), oversize_resize=mode, use_graph=False)
        for k, f in enumerate(frames):
            got = det.detect(f, 0.5)
            fits = f.shape[0] <= 448 and f.shape[1] <= 448
            if fits:
                ref = od.detect(f, 0.5, 0.4, net_hw=(448, 448))
                compare(got, ref.rows(), ref.anchor_indices(), FP32)
                continue
            canvas = det.debug_activation("input_canvas").astype(np.uint8)
            if mode == "bilinear":
                want = post.preprocess_trt_cvresize(f, 448, 448)[0].transpose(1, 2, 0)[:, :, ::-1].astype(np.uint8)
                assert np.array_equal(canvas, want), (mode, k, int(np.abs(canvas.astype(int) - want).max()))
                ref = od.detect(f, 0.5, 0.4, net_hw=(448, 448))
                compare(got, ref.rows(), ref.anchor_indices(), FP32)
                if k < 2:
                    assert len(got) >= 1
            else:
                want = post.resize_area_reference(f, 448, 448)
                diff = np.abs(canvas.astype(int) - want)
                assert diff.max() <= 1 and (diff > 0).mean() < 5e-3, (mode, k, int(diff.max()), float((diff > 0).mean()))


def test_cxx_class_drop_in(rfa, tmp_path):
    """The reference's class surface (include/RetinaFace.h) driven from C++ like retinaface/main.cpp does."""
    from retinaface_amd.frames import synth_frames
    exe = os.path.join(ROOT, "retinaface_amd", "lib", "rf_demo")
    assert os.path.exists(exe), "rf_demo is built by __graft_entry__.build()"
    frames = synth_frames(448, 448, 3, config=1)
    paths = []
    for i, f in enumerate(frames):
        p = str(tmp_path / f"f{i}.bgr")
        f.tofile(p)
        paths.append(p)
    out = subprocess.run([exe, ASSETS, "mnet25", "448", "448", "fp16", "0.5", "448", "448"] + paths, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    det = engine(rfa, "mnet25", FP16, (448, 448))
    ref = det.detectBatchImages(frames, 0.5)
    lines = [l.split() for l in out.stdout.strip().splitlines()]
    batch = [l for l in lines if l[0] == "batch"]
    assert len(batch) == sum(len(r) for r in ref)
    k = 0
    for i, r in enumerate(ref):
        for d in r:
            l = batch[k]
            k += 1
            assert int(l[1]) == i and abs(float(l[2]) - d.score) < 1e-5
            assert np.allclose([float(v) for v in l[3:7]], d.rect, atol=2e-3)
            assert abs(float(l[7]) - d.xs[0]) < 2e-3 and abs(float(l[8]) - d.ys[4]) < 2e-3
    single = [l for l in lines if l[0] == "detect"]
    assert len(single) == len(ref[0])
    assert lines[-1] == ["empty", "0"]


def test_c_serving_loop_example(rfa):
    """examples/serve.c: a plain-C serving loop on the C ABI alone -- a pinned ring of host frames (rf_host_register), rf_num_slots()
    batches in flight through rf_enqueue_batch / rf_wait; and the same binary with two device ordinals (both 0 on this box):
    rf_detect_batch sharded by image over two engines."""
    exe = os.path.join(ROOT, "retinaface_amd", "lib", "rf_serve")
    assert os.path.exists(exe), "build() makes rf_serve"
    for extra, ndev in (([], 1), (["0"], 1), (["0", "0"], 2)):
        out = subprocess.run([exe, ASSETS, "mnet25", "fp16", "448", "448", "8", "0.5"] + extra, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-1000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("rf_serve:")][-1]
        images = int(line.split()[1])
        assert images > 0 and images % 8 == 0 and f"{ndev} device" in line, line


def test_profile_accounting_matches_baseline_md(rfa):
    """rf_profile's per-launch algorithmic bytes / MACs sum to BASELINE.md section 2's per-image figures:
    B = 3P + 2(E - 3P) = 27 615 616 B (fp16, 448^2), MACs = 481 764 864."""
    import torch
    det = engine(rfa, "mnet25", FP16, (448, 448))
    frames = torch.zeros((8, 448, 448, 3), dtype=torch.uint8, device="cuda")
    prof = det.profile([frames[i].data_ptr() for i in range(8)], iters=2)
    assert len(prof) == 2 + 9 + 2 + 2 + 1 + 1        # stem2 (conv0 + blocks 0, 1), dwpw2 (blocks 2, 3), 9 dw/pw blocks (3 with a fused lateral), 2 aggr, 2 SSH (conv_a, fused tail), heads, NMS
    assert prof[0]["kernel"] == "stem2"
    assert abs(sum(p["alg_bytes"] for p in prof) / 8 - 27615616) < 1
    assert abs(sum(p["macs"] for p in prof) / 8 - 481764864) / 481764864 < 2.5e-3     # + the 4-tap upsample MACs
    assert all(p["ms"] > 0 for p in prof)
