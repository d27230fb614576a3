"""The int8 engine's distance to the fp32 oracle, held to a contract (round 6; VERDICT r5 "next" #1).

What the reference has: TensorRT's int8 mode + the calibration cache (retinaface/tensorrt/trtnetbase.cpp:295-311,
model/mnet-deconv-0517.table.int8).  TensorRT cannot run here, so bit-parity with ITS int8 engine is unobtainable; the engine's
integer arithmetic is pinned bit-exact on oracle/int8_forward.py, and THIS file measures how far that arithmetic is from the
fp32 Caffe semantics north_star states as the bar, on the fp16 contract's frame plan (both models, 448 x 448 and 1280 x 896,
8- and 32-image batches: 104 frames per model) built from the fixture faces the calibration never saw (faces 1, 3, 5; the
calibration sets of tools/calibrate_int8.py show only faces 0, 2, 4).

Per oracle face (a detection the fp32 oracle keeps at thr 0.5 / nms 0.4):
  iou          IoU of the engine's best-matching detection -- includes NMS-winner flips: when two neighbouring anchors of one face score
               within the quantisation noise of each other the engine may keep the other one, whose box is a DIFFERENT prediction of
               the network (IoU to the oracle's winner typically 0.90-0.95, in the fp32 oracle itself);
  same_anchor  whether that detection sits on the oracle's anchor (the agreement rate is the fraction of faces where it does);
  anchor_iou   IoU of the engine's box with the ORACLE'S box of the SAME anchor (its pre-NMS candidate): the regression error alone,
               defined for every detection whose anchor the oracle also saw above the threshold;
  dscore       |score - oracle score of the same anchor|.
"""
from typing import Dict, List

import numpy as np

from oracle.retinaface_post import iou_plus1

HELD_OUT_FACES = [1, 3, 5]
PLAN = (((448, 448), ((32, 400), (8, 401), (8, 402), (8, 403), (8, 404))), ((896, 1280), ((32, 410), (8, 411))))


def contract_frames(hw, nb, cfg):
    from retinaface_amd.frames import synth_frames
    return synth_frames(hw[0], hw[1], nb, config=cfg, faces=HELD_OUT_FACES)


def frame_rows(got, ref) -> List[dict]:
    """got: the engine's detections of one frame (objects with .rect, .score, .anchor_index); ref: oracle.pipeline.OracleResult."""
    cand = {int(c.anchor_index): c for c in ref.candidates}
    rows = []
    for r in ref.detections:
        if not got:
            rows.append(dict(iou=0.0, same_anchor=False, anchor_iou=None, dscore=1.0))
            continue
        best = max(got, key=lambda a: iou_plus1(a.rect, r.rect))
        c = cand.get(int(best.anchor_index))
        rows.append(dict(iou=float(iou_plus1(best.rect, r.rect)), same_anchor=int(best.anchor_index) == int(r.anchor_index),
                         anchor_iou=None if c is None else float(iou_plus1(best.rect, c.rect)),
                         dscore=1.0 if c is None else abs(float(best.score) - float(c.score))))
    return rows


def summarize(frames: List[dict]) -> Dict[str, float]:
    """frames: [{same_count: bool, rows: frame_rows(...)}]"""
    rows = [r for f in frames for r in f["rows"]]
    iou = np.array([r["iou"] for r in rows])
    aiou = np.array([r["anchor_iou"] for r in rows if r["anchor_iou"] is not None])
    return dict(frames=len(frames), faces=len(rows), same_count=int(sum(f["same_count"] for f in frames)),
                iou_worst=float(iou.min()), iou_p01=float(np.quantile(iou, 0.01)), iou_mean=float(iou.mean()), iou_below_097=int((iou < 0.97).sum()),
                anchor_agreement=float(np.mean([r["same_anchor"] for r in rows])),
                anchor_iou_worst=float(aiou.min()), anchor_iou_p01=float(np.quantile(aiou, 0.01)), anchor_iou_mean=float(aiou.mean()),
                anchor_iou_below_097=int((aiou < 0.97).sum()), unmatched=int(len(rows) - len(aiou)),
                dscore_max=float(max(r["dscore"] for r in rows)))


def run_contract(make_engine, oracle, plan=PLAN, oracle_cache=None, stem=None) -> Dict[str, float]:
    """make_engine(hw, max_batch) -> detector with detectBatchImages; oracle: OracleDetector of the same model.  With `stem` the large frames'
    oracle results are served from the minted file (tests/oracle_cache.py) where it holds them."""
    import oracle_cache as minted
    frames_out = []
    for hw, batches in plan:
        for nb, cfg in batches:
            frames = contract_frames(hw, nb, cfg)
            det = make_engine(hw, nb)
            got = det.detectBatchImages(frames, 0.5)
            for i, f in enumerate(frames):
                key = (hw, cfg, i)
                if oracle_cache is not None and key in oracle_cache:
                    ref = oracle_cache[key]
                else:
                    ref = minted.detect(oracle, stem, f, hw, cfg, HELD_OUT_FACES, i) if stem else oracle.detect(f, 0.5, 0.4, net_hw=hw)
                    if oracle_cache is not None:
                        oracle_cache[key] = ref
                frames_out.append(dict(same_count=len(got[i]) == len(ref.detections), rows=frame_rows(got[i], ref)))
            det.close()
    return summarize(frames_out)


def fmt(stem, s) -> str:
    return (f"int8 contract {stem}: {s['frames']} frames / {s['faces']} faces, same face count on {s['same_count']}; per-face IoU worst {s['iou_worst']:.4f} "
            f"p01 {s['iou_p01']:.4f} mean {s['iou_mean']:.4f} ({s['iou_below_097']} below 0.97); anchor agreement {s['anchor_agreement']:.3f}; "
            f"same-anchor IoU worst {s['anchor_iou_worst']:.4f} p01 {s['anchor_iou_p01']:.4f} mean {s['anchor_iou_mean']:.4f} "
            f"({s['anchor_iou_below_097']} below 0.97, {s['unmatched']} detections on anchors the oracle kept below the threshold); max |dscore| {s['dscore_max']:.4f}")
