#!/usr/bin/env python3
"""Probe: does a kernel-variant knob leave the detections BIT-IDENTICAL?  (int8 kernels are integer arithmetic end to end: a different tile
shape or schedule must not change a single bit; fp16 variants that keep the accumulation order are identical too.)
usage: knob_equal.py --precision 2 --model mnet25 RF_TILE128=1 RF_TILE64=1 ...   (each knob is run alone in its own process and compared with no knob)"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = (
    "import sys, json; sys.path.insert(0, %r)\n"
    "import retinaface_amd\n"
    "from retinaface_amd.frames import synth_frames\n"
    "det = retinaface_amd.RetinaFace(%r, 'net3', 0.4, precision=%d, net_hw=(448, 448), model_stem=%r)\n"
    "res = det.detectBatchImages(synth_frames(448, 448, %d, config=1), %f)\n"
    "print('RESULT ' + json.dumps([[[d.anchor_index] + [float(v).hex() for v in d.as_row()] for d in r] for r in res]))\n")


def run(args, env_extra):
    env = dict(os.environ); env.update(env_extra)
    code = CODE % (ROOT, os.path.join(ROOT, "assets"), args.precision, args.model, args.n, args.threshold)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    if out.returncode: raise SystemExit("knob %s: process failed\n%s" % (env_extra, out.stderr[-2000:]))
    return [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", type=int, default=2)
    ap.add_argument("--model", default="mnet25")
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--threshold", type=float, default=0.1)
    ap.add_argument("knobs", nargs="+")
    args = ap.parse_args()
    base = run(args, {})
    nfaces = sum(len(r) for r in json.loads(base[7:]))
    bad = 0
    for k in args.knobs:
        got = run(args, dict([k.split("=")]))
        same = got == base
        print("%-22s %s (%d detections over %d frames at threshold %g)" % (k, "bit-identical" if same else "DIFFERENT", nfaces, args.n, args.threshold))
        bad += not same
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
