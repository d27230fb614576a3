#!/usr/bin/env python3
"""Probe (GPU): the int8 contract numbers (tests/int8_contract.py) of the int8 engine built from a model directory -- the shipped assets
or a calibration under test -- and, beside them, of the fp16 engine on the same frames (what "no quantisation" scores on this metric).

usage: python tools/probes/int8_contract_probe.py [--assets DIR] [--models mnet25,mnet-deconv-0517] [--fp16] [--small] [--json OUT]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import retinaface_amd                                            # noqa: E402
from int8_contract import PLAN, fmt, run_contract                # noqa: E402
from oracle.caffe_io import read_rfw                             # noqa: E402
from oracle.pipeline import OracleDetector                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assets", default=os.path.join(ROOT, "assets"))
    ap.add_argument("--models", default="mnet-deconv-0517,mnet25")
    ap.add_argument("--fp16", action="store_true", help="also run the fp16 engine through the same metric")
    ap.add_argument("--small", action="store_true", help="448 x 448 only")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    plan = PLAN[:1] if args.small else PLAN
    out = {}
    cache = {}
    for stem in args.models.split(","):
        oracle = OracleDetector(read_rfw(os.path.join(ROOT, "assets", stem + ".rfw")))          # fp32 weights: the same in every calibration
        cache.setdefault(stem, {})
        for prec, tag in ((retinaface_amd.PRECISION_INT8, "int8"),) + (((retinaface_amd.PRECISION_FP16, "fp16"),) if args.fp16 else ()):
            def make(hw, nb, prec=prec):
                return retinaface_amd.RetinaFace(args.assets, "net3", 0.4, precision=prec, net_hw=hw, model_stem=stem, max_batch=nb, plan_cache=False)
            s = run_contract(make, oracle, plan, cache[stem])
            out[f"{stem}/{tag}"] = s
            print(f"[{tag}] " + fmt(stem, s), flush=True)
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
