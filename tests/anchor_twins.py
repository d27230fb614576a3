"""The fp16 engine's anchor-SET band (round 5), the third of the derived fp16 tolerances next to the candidate-count band and the
order band (tests/test_gpu_parity.py).

north_star asks for "bit-exact anchor indices".  An fp16 engine's scores are within SCORE_NOISE (2e-3) of the oracle's, so the one
place where its kept set can legitimately differ is a pair of TWIN anchors: two candidates of the oracle that sit on the same
face (they suppress each other: IoU > nms threshold) and whose oracle scores are closer than twice the score noise.  Greedy NMS
keeps whichever of the two scores higher (RetinaFace.cpp:434-492); a 3e-6 score gap decides that in the oracle, the engine's
noise decides it in the engine.  On the 208 contract frames that happens once (anchors 300 / 301, oracle scores 0.997809 /
0.997806).  The scores are a 2-class softmax, i.e. a sigmoid of the logit difference, so a given logit noise moves a score by
p (1 - p) times it: SCORE_NOISE is what the tests assert where the sigmoid is steepest (p = 0.5, slope 1/4), and the same logit
noise near saturation is 4 p (1 - p) times smaller -- 1.8e-5 at p = 0.9978, where a flat 2e-3 would call every neighbouring
anchor of a strong face a twin.  The band therefore uses noise(p) = SCORE_NOISE * 4 p (1 - p) and admits exactly this:

  a detection may sit on anchor T instead of the oracle's kept anchor W iff
    * T is a candidate of the oracle itself (it crossed the threshold there too),
    * |score_oracle(W) - score_oracle(T)| <= 2 * noise(p), p = the mean of the two scores,
    * IoU(box_oracle(W), box_oracle(T)) > nms threshold (they suppress each other), and no detection the oracle kept BEFORE W
      suppresses T (otherwise T could not have won in any order),
  and the engine's box / score / landmarks for T are then held to the usual tolerances against the ORACLE'S values for T.

`twins_of` derives the admissible pairs from one oracle result; `resolve` maps an engine's detections onto reference rows and
counts how often the band fired.  The golden frames' pairs are minted into tests/golden/threshold_bands.npz (`<key>/twins`,
tools/make_golden.py --bands) for the GPU box, where the oracle's candidates are not at hand for every test.
"""
import numpy as np

from oracle.retinaface_post import iou_plus1


def twins_of(det_rows, det_idx, cand_rows, cand_idx, nms_threshold, noise):
    """-> float32 array [k][17]: (W anchor, T anchor, T's 15-float oracle row) for every admissible (kept W, suppressed twin T)."""
    out = []
    det_idx = [int(a) for a in det_idx]
    kept = set(det_idx)
    for wi, (w_row, w_a) in enumerate(zip(det_rows, det_idx)):
        for t_row, t_a in zip(cand_rows, cand_idx):
            t_a = int(t_a)
            pm = 0.5 * (float(w_row[0]) + float(t_row[0]))
            if t_a in kept or abs(float(w_row[0]) - float(t_row[0])) > 2 * noise * 4 * pm * (1 - pm):
                continue
            if iou_plus1(w_row[1:5], t_row[1:5]) <= nms_threshold:
                continue
            if any(iou_plus1(det_rows[e][1:5], t_row[1:5]) > nms_threshold for e in range(wi)):
                continue
            out.append(np.concatenate([[np.float32(w_a), np.float32(t_a)], np.asarray(t_row, np.float32)]))
    return np.stack(out).astype(np.float32) if out else np.zeros((0, 17), np.float32)


def twins_of_result(ref, nms_threshold, noise):
    """The same from an oracle.pipeline.OracleResult."""
    rows = [d.as_row() for d in ref.detections]
    crow = [d.as_row() for d in ref.candidates]
    return twins_of(rows, [d.anchor_index for d in ref.detections], crow, [d.anchor_index for d in ref.candidates], nms_threshold, noise)


def resolve(got_idx, ref_rows, ref_idx, twins):
    """Match an engine's kept anchors against the oracle's.  Returns (rows, swaps, canon): rows[k] = the oracle row detection k of
    the engine is held to (the oracle's own row where the anchors agree, the twin's oracle row where the band fired), swaps = how
    often it fired, canon = the engine's anchor list with every admitted twin replaced by the oracle's anchor (for order checks).
    Raises AssertionError when the two sets differ by anything the band does not admit."""
    got_idx = [int(a) for a in got_idx]
    ref_idx = [int(a) for a in ref_idx]
    assert len(set(got_idx)) == len(got_idx), ("an anchor kept twice", got_idx)
    assert len(got_idx) == len(ref_idx), (got_idx, ref_idx)
    by_ref = {a: r for a, r in zip(ref_idx, ref_rows)}
    extra = [a for a in got_idx if a not in by_ref]
    missing = [a for a in ref_idx if a not in set(got_idx)]
    admissible = {}
    for row in (twins if twins is not None else ()):
        admissible[(int(row[0]), int(row[1]))] = np.asarray(row[2:], np.float32)
    twin_row, stands_for = {}, {}
    for t in extra:
        ws = [w for w in missing if (w, t) in admissible]
        assert len(ws) == 1, ("anchor sets differ outside the twin band", got_idx, ref_idx)
        missing.remove(ws[0])
        twin_row[t] = admissible[(ws[0], t)]
        stands_for[t] = ws[0]
    assert not missing, ("anchor sets differ outside the twin band", got_idx, ref_idx)
    rows = [by_ref[a] if a in by_ref else twin_row[a] for a in got_idx]
    return rows, len(extra), [stands_for.get(a, a) for a in got_idx]
