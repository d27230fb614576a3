#!/usr/bin/env python3
"""Soak test of the engine's scheduler (round 6): random interleavings of everything the C ABI offers on ONE handle from one thread -- asynchronous tickets
of 1..max_batch device / pageable-host / registered-host frames (waited in random order, more in flight than one launch holds), synchronous calls of
1..3 x max_batch frames from device and host memory (pipelined staging pieces, chunk coalescing), empty frames, strided views -- for N seconds, every
result compared byte for byte with the per-frame expectation computed once up front.  usage: stress.py [--seconds 60] [--precision fp16|int8] [--seed 1]"""
import argparse
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    if "--threads" in sys.argv:                    # N worker threads, each with its OWN handle (different handles are independent: include/retinaface_amd.h)
        import threading
        k = sys.argv.index("--threads")
        nthreads = int(sys.argv[k + 1])
        del sys.argv[k:k + 2]
        errs = []

        def work(i):
            try:
                run(seed_offset=100 * i, label=f"thread {i}: ")
            except BaseException as e:  # noqa: BLE001
                errs.append((i, repr(e)))

        th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise SystemExit(f"stress FAILED in {errs}")
        return
    run()


def run(seed_offset=0, label=""):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-batch", type=int, default=8)
    ap.add_argument("--devices", type=int, default=0, help="N > 0: ONE handle over N engines on device 0 (multi.cpp: image sharding, one host thread per engine) with "
                                                           "RF_FORCE_SCATTER=1, so that every device frame travels by peer copy")
    a = ap.parse_args()
    import torch
    import retinaface_amd
    from retinaface_amd.frames import synth_frames
    rnd = random.Random(a.seed + seed_offset)
    H = W = 448
    prec = {"fp16": 1, "int8": 2, "fp32": 0}[a.precision]
    if a.devices > 0:
        os.environ["RF_FORCE_SCATTER"] = "1"
    det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W), max_batch=a.max_batch, model_stem="mnet25",
                                    devices=[0] * a.devices if a.devices > 0 else None)
    N = 48
    frames = synth_frames(H, W, N, config=91)
    dev = torch.from_numpy(np.stack(frames)).cuda()
    pinned = np.stack(frames).copy()
    det.host_register(pinned)
    wide = np.zeros((N, H, W + 20, 3), np.uint8)
    wide[:, :, :W] = np.stack(frames)
    key = lambda r: [(d.anchor_index, d.as_row().tobytes()) for d in r]          # noqa: E731
    want = []
    for i in range(0, N, a.max_batch):
        want += [key(r) for r in det.detect_device([dev[j].data_ptr() for j in range(i, min(i + a.max_batch, N))], [H] * min(a.max_batch, N - i), [W] * min(a.max_batch, N - i), 0.5)]
    assert sum(len(w) for w in want) >= N
    inflight, ops, checked = [], {}, 0
    t0 = time.time()

    def pick(n):
        return [rnd.randrange(N) for _ in range(n)]

    def expect(ids):
        return [[] if i is None else want[i] for i in ids]

    def host_list(ids, kind):
        out = []
        for i in ids:
            if i is None:
                out.append(None)
            elif kind == "pinned":
                out.append(pinned[i])
            elif kind == "strided":
                out.append(wide[i][:, :W])
            else:
                out.append(frames[i])
        return out

    while time.time() - t0 < a.seconds:
        op = rnd.random()
        if op < 0.45 and len(inflight) < det.num_slots():
            n = rnd.randint(1, a.max_batch)
            ids = pick(n)
            kind = rnd.choice(["device", "pageable", "pinned"])
            if kind == "device":
                t = det.enqueue_device([dev[i].data_ptr() for i in ids], [H] * n, [W] * n, 0.5)
            else:
                t = det.enqueue_host(host_list(ids, kind), 0.5)
            inflight.append((t, ids))
            ops[kind + " ticket"] = ops.get(kind + " ticket", 0) + 1
        elif op < 0.75 and inflight:
            t, ids = inflight.pop(rnd.randrange(len(inflight)) if rnd.random() < 0.3 else 0)
            got = det.wait(t, len(ids))
            assert [key(r) for r in got] == expect(ids), ("ticket", ids)
            checked += len(ids)
        elif op < 0.9:
            n = rnd.randint(1, 3 * a.max_batch)
            ids = pick(n)
            kind = rnd.choice(["device", "pageable", "pinned", "strided", "holes"])
            if kind == "device":
                got = det.detect_device([dev[i].data_ptr() for i in ids], [H] * n, [W] * n, 0.5)
            else:
                if kind == "holes":
                    ids = [None if rnd.random() < 0.25 else i for i in ids]
                got = det.detectBatchImages(host_list(ids, "pageable" if kind == "holes" else kind), 0.5)
            assert [key(r) for r in got] == expect(ids), ("sync", kind, ids)
            checked += n
            ops[kind + " sync"] = ops.get(kind + " sync", 0) + 1
        else:
            while inflight:
                t, ids = inflight.pop(0)
                assert [key(r) for r in det.wait(t, len(ids))] == expect(ids), ("drain", ids)
                checked += len(ids)
    while inflight:
        t, ids = inflight.pop(0)
        assert [key(r) for r in det.wait(t, len(ids))] == expect(ids), ("final drain", ids)
        checked += len(ids)
    det.host_unregister(pinned)
    det.close()
    print(f"{label}stress ok: {a.precision}, engines {max(a.devices, 1)}, max_batch {a.max_batch}, {time.time() - t0:.0f} s, {checked} frame results checked byte for byte, operations {dict(sorted(ops.items()))}")


if __name__ == "__main__":
    main()
