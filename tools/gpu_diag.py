#!/usr/bin/env python3
"""GPU-box diagnostic: run the HIP engine and the CPU oracle on the same frame and print, layer by layer, how
far every fused op's output is from the reference blob it corresponds to.  Never stops at the first mismatch;
writes gpurun_out/diag_<precision>.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.caffe_forward import HEAD_STRIDES, head_names  # noqa: E402
from oracle.caffe_io import read_rfw  # noqa: E402
from oracle.pipeline import OracleDetector  # noqa: E402
from oracle.retinaface_post import iou_plus1, preprocess_trt_identity  # noqa: E402
import retinaface_amd  # noqa: E402
from retinaface_amd.frames import padded_base_frame, synth_frames  # noqa: E402


def main():
    stem = "mnet-deconv-0517"
    assets = os.path.join(ROOT, "assets")
    net = read_rfw(os.path.join(assets, stem + ".rfw"))
    oracle = OracleDetector(net)
    frame = np.ascontiguousarray(padded_base_frame()[30:478, 440:888])
    chw = preprocess_trt_identity(frame, 448, 448)
    t = time.time()
    blobs = oracle.forward(chw, keep_all=True)
    print(f"oracle forward {time.time() - t:.2f}s", flush=True)
    ref = oracle.detect(frame, 0.5, 0.4, net_hw=(448, 448))
    names = ["mobilenet0_relu0_fwd"] + [f"mobilenet0_relu{i}_fwd" for i in range(2, 27, 2)]
    names += ["rf_c3_lateral_relu", "rf_c2_lateral_relu", "rf_c2_aggr_relu", "rf_c1_red_conv_relu", "rf_c1_aggr_relu"]
    for c in (3, 2, 1):
        names += [f"rf_c{c}_det_context_conv1_relu", f"rf_c{c}_det_context_conv3_1_relu", f"rf_c{c}_det_concat_relu"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for prec, tag in ((retinaface_amd.PRECISION_FP32, "fp32"), (retinaface_amd.PRECISION_FP16, "fp16")):
        report = {"precision": tag, "layers": {}, "heads": {}}
        try:
            det = retinaface_amd.RetinaFace(assets, "net3", 0.4, precision=prec, net_hw=(448, 448), keep_outputs=True,
                                            use_graph=False, model_stem=stem)
            got = det.detect(frame, 0.5)
        except Exception as e:  # noqa: BLE001
            print(tag, "FAILED:", repr(e), flush=True)
            report["error"] = repr(e)
            json.dump(report, open(os.path.join(ROOT, "gpurun_out", f"diag_{tag}.json"), "w"), indent=1)
            continue
        for n in names:
            a = det.debug_activation(n)                      # H, W, C
            r = blobs[n][0].transpose(1, 2, 0)
            diff = np.abs(a - r)
            worst = np.unravel_index(int(diff.argmax()), diff.shape)
            report["layers"][n] = {"max_abs": float(diff.max()), "ref_max": float(np.abs(r).max()),
                                   "mean_abs": float(diff.mean()), "worst_hwc": [int(v) for v in worst],
                                   "nan": bool(np.isnan(a).any())}
            print(f"{tag} {n:42s} max|d|={diff.max():.3e} mean|d|={diff.mean():.3e} ref_max={np.abs(r).max():.3e} "
                  f"worst={worst} nan={np.isnan(a).any()}", flush=True)
        for s in HEAD_STRIDES:
            for n in head_names(s):
                a = det.get_output(n)
                r = ref.heads[n][0]
                d = float(np.abs(a - r).max())
                report["heads"][n] = {"max_abs": d, "ref_max": float(np.abs(r).max())}
                print(f"{tag} {n:42s} max|d|={d:.3e}", flush=True)
        print(tag, "candidates", det.last_candidate_counts(1), "oracle", len(ref.candidates))
        print(tag, "detections", len(got), "oracle", len(ref.detections))
        report["n_det"] = len(got)
        report["n_det_ref"] = len(ref.detections)
        report["dets"] = []
        for g, r in zip(got, ref.detections):
            iou = iou_plus1(g.rect, r.rect)
            lm = max(abs(a - float(b)) for a, b in zip(g.xs + g.ys, list(r.xs) + list(r.ys)))
            print(f"   anchor {g.anchor_index} vs {r.anchor_index}  score {g.score:.6f} vs {float(r.score):.6f}  IoU {iou:.6f}  "
                  f"max landmark |d| {lm:.4f}", flush=True)
            report["dets"].append({"anchor": g.anchor_index, "anchor_ref": r.anchor_index, "iou": iou, "score": g.score,
                                   "score_ref": float(r.score), "lm": lm})
        print(tag, "timings (eager)", det.last_timings(), flush=True)
        # batch of 8 synthetic frames through the graph path
        det2 = retinaface_amd.RetinaFace(assets, "net3", 0.4, precision=prec, net_hw=(448, 448), model_stem=stem)
        frames = synth_frames(448, 448, 8, config=1)
        for rep in range(3):
            t = time.time()
            res = det2.detectBatchImages(frames, 0.5)
            dt = time.time() - t
            print(tag, f"batch8 call {rep}: {dt * 1e3:.3f} ms, faces per frame", [len(r) for r in res], flush=True)
        report["batch8_counts"] = [len(r) for r in res]
        json.dump(report, open(os.path.join(ROOT, "gpurun_out", f"diag_{tag}.json"), "w"), indent=1)
        det.close()
        det2.close()


if __name__ == "__main__":
    main()
