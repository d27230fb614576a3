#!/usr/bin/env python3
"""Mint tests/golden/contract_oracle_1280x896.npz: the fp32 oracle's results on the 1280 x 896 frames of the fp16 and int8 contracts
(tests/test_gpu_parity.py, tests/int8_contract.py) -- see tests/oracle_cache.py.  Runs the CPU oracle only (no GPU, no /root/reference):
    python tools/make_contract_golden.py            (~3 minutes on 8 cores)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_cache                                              # noqa: E402
from int8_contract import HELD_OUT_FACES, PLAN                   # noqa: E402
from oracle.caffe_io import read_rfw                              # noqa: E402
from oracle.pipeline import OracleDetector                        # noqa: E402
from retinaface_amd.frames import synth_frames                    # noqa: E402


def main():
    out = {}
    t0 = time.time()
    for stem in ("mnet-deconv-0517", "mnet25"):
        od = OracleDetector(read_rfw(os.path.join(ROOT, "assets", stem + ".rfw")))
        for hw, batches in PLAN:
            if hw[0] * hw[1] <= 512 * 512:
                continue                                         # small frames stay live
            for faces in (None, HELD_OUT_FACES):                 # the fp16 contract's frames / the int8 contract's (held-out faces)
                for nb, cfg in batches:
                    for i, f in enumerate(synth_frames(hw[0], hw[1], nb, config=cfg, faces=faces)):
                        ref = od.detect(f, 0.5, 0.4, net_hw=hw)
                        key = oracle_cache.frame_key(stem, hw, cfg, faces, i)
                        for k, v in oracle_cache.pack(ref, f).items():
                            out[f"{key}/{k}"] = v
        print(f"{stem}: {len(out) // 6} frames so far, {time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(oracle_cache.GOLDEN, **out)
    print("wrote", oracle_cache.GOLDEN, os.path.getsize(oracle_cache.GOLDEN), "bytes,", len(out) // 6, "frames")


if __name__ == "__main__":
    main()
