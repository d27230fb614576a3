#!/usr/bin/env python3
"""bench.py -- the RetinaFace::detect() hot path on N MI355X GPUs (one process per GPU).

A "step" = one pass of the whole hot path (fused preprocess+conv0 -> backbone -> FPN -> SSH -> heads+decode -> NMS
-> result D2H) over one batch of synthetic face-bearing frames already resident in HBM.  Workload at N = 1 is
BASELINE.json configs[1]: mnet25, fp16, 448x448, batch 8 on one MI355X.

  python bench.py --gpus N --steps K --warmup W
      N > 1 and no WORLD_SIZE in the environment: this process re-launches itself under torch.distributed.run with N ranks
      (one per GPU, rendezvous on 127.0.0.1) and relays rank 0's JSON line.  Under an external launcher (the driver's
      `python -m torch.distributed.run ... bench.py --gpus N`) WORLD_SIZE must equal N.
  Every rank runs the same batch size on its own frames (images are independent: no data-path collective, "scaling": "weak").
  With N > 1 the RESULT GATHER of the sharded detectBatchImages -- fixed-size records, one RCCL all_gather per global
  super-batch, retinaface_amd/shard.py -- runs inside the timed region.  Timing: barrier + synchronize on both sides, MAX over ranks.

The timed region runs at least K steps and at least --min-seconds (default 1 s): a step is 33 us and the engine's pipeline holds
3 x 16 of them, so K = 20 would time pipeline fill and drain, not throughput.  `steps` in the JSON is the number actually timed,
`steps_requested` what was asked for; the one-super-batch burst is reported separately (`burst`).

Input frames come from a ring of distinct frames larger than the 256 MiB Infinity Cache, so every step reads its pixels from
HBM (`cache_resident_input` reports the old number: the same 8 frames re-submitted).  `host_frames` is the same loop with the
frames in HOST memory (rf_enqueue_batch: pinned staging, one DMA per enqueue, upload overlapped with compute) -- what the
reference's API hands over; it is PCIe-bound and never `value`.

Prints ONE JSON line (rank 0): metric faces/sec (BASELINE.json), images/sec and ms/frame beside it, the pre/infer/post split,
`roofline` for the dominant kernel (HIP-event timed inside this process, algorithmic bytes per SURVEY.md 8d) and `cpu_baseline`
(the CPU oracle = restatement of the reference's Caffe path, timed on the host cores on a bounded sample of the same frames;
N = 1, rank 0 only).  --dry runs the launcher / sharding / gather / JSON plumbing with a stub engine on gloo (CPU tests).

The default invocation (N = 1, the metric point) also times the OTHER BASELINE.json configurations in the same run and reports them
in `configs`: [2] mnet-deconv-0517 int8 batch 32, [3] mnet25 fp16 1280x896 batch 1, [4]'s per-GPU shape mnet25 int8 batch 32 -- a
timed region of >= --extra-seconds each with images/s, faces/s, dtype, the dominant kernel and (one extra set of counter passes,
collected while the CPU baseline runs) its binding resource.  With N > 1 a short STRONG-scaling leg of configs[4] as stated (one
global batch of 256 int8 images per step split over the ranks) follows the weak-scaling measurement (`configs4_strong`), and
`result_gather` says which backend carried the gather, the communicator size seen through an RCCL collective and the RCCL version.

`configs` also carries north_star's batch x frame-size matrix (round 5): fp16 448x448 at batch 1 and 32, 1280x896 at batch 8 and 32 -- the
cells that are not BASELINE.json configs themselves -- and every entry has `sync_batch`: ONE synchronous rf_detect_batch_device call of that
batch at a time, the reference's own calling convention (RetinaFace.cpp:749-940).

`roofline.frac` is the PHYSICAL fraction of the dominant kernel's binding resource (HBM bytes moved, VALU / MFMA / LDS issue cycles,
measured by rocprofv3 --pmc in this run): always <= 1, and null when no counter pass ran (the layer-wise figure is never substituted).
It is an OCCUPANCY of that resource, not an efficiency: `roofline.useful` says how much of the HBM and matrix peaks is work the layers
require (compulsory bytes of the fused launch / time / 8 TB/s; layer MACs x 2 / time / dense MFMA peak), and `kernel_ms_in_pipeline`
is the same kernel's average duration in a single-lane rocprofv3 kernel trace of the timed loop (what throughput is made of) beside the
back-to-back HIP-event figure.  The layer-wise algorithmic-bytes figure of SURVEY.md 8d, which counts bytes a fused kernel never moves
and therefore exceeds 1, stays as `frac_layerwise_credit` / `hbm_layerwise`.

With N > 1 rank 0 finally runs the LIBRARY leg (`library_multi_device`): ONE handle over devices [0 .. N-1] (rf_options.devices, multi.cpp:
one engine + host thread per GPU, contiguous image slices), every frame resident on GPU 0, BASELINE.json configs[4] as stated -- 256 int8
images per rf_detect_batch_device call -- so that the in-library split and the xGMI peer scatter are measured by the same run that
measures the ranks (peer-access matrix, bytes scattered per call, images/s).  `--library-devices N` runs that leg alone (rehearsal on one
GPU: the ordinals repeat and RF_FORCE_SCATTER makes every frame travel).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# dense MFMA peaks (MI355X_MICROARCH.md): fp16 2.5 PFLOP/s spec; int8 has no spec row in the guide, only ">= 3 944 TOPS measured"
# (labelled as such in the JSON); fp32 157.3
MFMA_PEAK_TFLOPS = {"fp16": 2500.0, "int8": 3944.0, "fp32": 157.3}
MFMA_PEAK_SOURCE = {"fp16": "spec, dense", "int8": "measured (guide gives no spec for i8)", "fp32": "spec"}
PCIE_GEN5_X16_GBS = 63.0       # spec (MI355X_MICROARCH.md host link); the run also measures what a pinned torch copy reaches
MALL_BYTES = 256 << 20
GATHER_CAP = 16                # faces per image in the gathered record (count is exact; frames carry <= 6 faces)


FRAC_KIND = {"valu_issue": "valu_issue_occupancy", "lds_issue": "lds_issue_occupancy", "mfma": "mfma_busy_occupancy", "hbm": "hbm_bytes_moved_over_peak"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--min-seconds", type=float, default=1.0, help="lower bound on the timed region (0 = exactly --steps)")
    ap.add_argument("--exact-steps", action="store_true", help="time EXACTLY --steps steps in ONE region (no --min-seconds stretch, no repeats): the "
                                                               "line then says burst_only = true -- at the driver's --steps 20 that is 0.5 ms of pipeline fill, not throughput")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=448)
    ap.add_argument("--width", type=int, default=448)
    ap.add_argument("--precision", choices=["fp16", "fp32", "int8"], default="fp16")
    ap.add_argument("--model", default="mnet25")
    ap.add_argument("--threshold", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--profile-iters", type=int, default=50)
    ap.add_argument("--lanes", type=int, default=0, help="launches in flight (0 = engine default: 3)")
    ap.add_argument("--coalesce", type=int, default=0, help="enqueued batches merged per launch (0 = engine default: 256 / batch)")
    ap.add_argument("--ring-mb", type=float, default=320.0, help="distinct input frames per GPU, in MB (> the 256 MiB MALL)")
    ap.add_argument("--host-seconds", type=float, default=1.5, help="length of the host-frame (PCIe inclusive) measurement; 0 = skip")
    ap.add_argument("--oversubscribe", action="store_true", help="allow more ranks than visible GPUs (ranks share devices)")
    ap.add_argument("--dry", action="store_true", help="no GPU: stub engine, gloo backend (launcher / gather / JSON plumbing only)")
    ap.add_argument("--timed-only", action="store_true", help="warm-up + timed region only (no burst / latency / host-frame / per-kernel "
                                                              "passes): the run to put under rocprofv3 --kernel-trace")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure roofline.traffic "
                                                          "(the newest committed PMC summary under profiles/ is joined instead)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: a step is ONE batch of this many images split over the ranks by shard_range (BASELINE.json "
                         "configs[4]: --global-batch 256 --gpus 8 --precision int8); 0 = weak scaling, --batch images per rank")
    ap.add_argument("--regions", type=int, default=3, help="timed regions; `value` is the median one, min / max are reported beside it")
    ap.add_argument("--extra-seconds", type=float, default=0.4,
                    help="timed region of each additional BASELINE.json config the default invocation also measures (`configs`); 0 = skip")
    ap.add_argument("--no-extra-configs", action="store_true", help="measure only the configuration named by the flags")
    ap.add_argument("--library-devices", type=int, default=0,
                    help="run ONLY the library leg: one handle over this many device ordinals (wrapping around the visible GPUs), 256 int8 images per call")
    ap.add_argument("--no-pipeline-trace", action="store_true", help="skip the single-lane rocprofv3 --kernel-trace pass (roofline.kernel_ms_in_pipeline)")
    ap.add_argument("--master-port", type=int, default=0)
    return ap.parse_args(argv)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """--gpus N without an external launcher: become the launcher (one rank per GPU) and relay the ranks' output."""
    port = args.master_port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["RF_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


class StubEngine:
    """--dry: stands in for the HIP engine so the multi-process plumbing runs on a CPU-only box.  Same call surface as the part of
    retinaface_amd.RetinaFace the timed loop uses; 'detections' are a deterministic function of the frame index."""
    max_detections = 256

    def __init__(self, batch, lanes, coalesce):
        import numpy as np
        self.np = np
        self.batch, self.lanes, self.coalesce = batch, lanes or 3, coalesce or max(1, min(256, 256 // batch))
        self._q = {}
        self._t = 0
        self.last_faces = np.zeros((batch, self.max_detections, 15), np.float32)

    def num_slots(self):
        return self.lanes * self.coalesce

    def enqueue_prepared(self, prepared, thr):
        self._t += 1
        self._q[self._t] = prepared
        return self._t

    def wait_counts(self, ticket, n):
        first = self._q.pop(ticket)
        counts = [1 + (first + i) % 3 for i in range(n)]
        for i, c in enumerate(counts):
            self.last_faces[i, :c, 0] = 0.9
            self.last_faces[i, :c, 1:5] = (first + i, 0.0, first + i + 10.0, 10.0)
        return counts

    def close(self):
        pass


class Runner:
    """One engine configuration on this rank: keeps the engine's pipeline full (up to `slots` batches in flight, every step's results
    collected) and, with N > 1, all-gathers the fixed-size result records of every global super-batch (retinaface_amd/shard.py) --
    one gather stays in flight while the next block fills."""

    def __init__(self, det, B, B_pad, slots, per_launch, world, cdev, dry, thr):
        import numpy as np
        import torch
        from retinaface_amd import shard
        self.np, self.torch, self.shard = np, torch, shard
        self.det, self.B, self.B_pad, self.slots, self.per_launch = det, B, B_pad, slots, per_launch
        self.world, self.cdev, self.dry, self.thr = world, cdev, dry, thr
        self.rec_w = 1 + GATHER_CAP * shard.RECORD_FLOATS
        self.gs = {"buf": np.zeros((per_launch * B_pad, self.rec_w), np.float32), "fill": 0, "handle": None, "out": None, "block": None,
                   "gathers": 0, "images": torch.zeros((), dtype=torch.int64, device=cdev)}

    def record_step(self, counts, n):
        """Append this step's per-image records (count + first GATHER_CAP faces) to the rank's block; all_gather it when the
        block holds one super-batch."""
        gs, det = self.gs, self.det
        faces = det.last_faces if self.dry else det.last_wait_faces()
        blk = gs["buf"][gs["fill"]:gs["fill"] + n]
        blk[:, 0] = counts
        blk[:, 1:].reshape(n, GATHER_CAP, self.shard.RECORD_FLOATS)[:, :, :15] = faces[:n, :GATHER_CAP]
        gs["buf"][gs["fill"] + n:gs["fill"] + self.B_pad, 0] = -1.0            # strong mode, ragged split: this rank's slice is shorter
        gs["fill"] += self.B_pad
        if gs["fill"] == gs["buf"].shape[0]:
            self.flush_gather()

    def finish_gather(self):
        gs = self.gs
        if gs["handle"] is not None:
            gs["handle"].wait()                 # RCCL: orders the current stream behind the collective, the host does not block
            gs["images"] += (gs["out"][:, 0] >= 0).sum()          # consume the gathered block (records of every rank), on device
            gs["handle"] = None

    def flush_gather(self, final=False):
        import torch.distributed as dist
        gs, torch = self.gs, self.torch
        self.finish_gather()
        if gs["fill"]:
            block = torch.from_numpy(gs["buf"][:gs["fill"]].copy())
            if gs["fill"] < gs["buf"].shape[0]:                       # ragged tail: pad to the fixed block size
                pad = torch.full((gs["buf"].shape[0] - gs["fill"], self.rec_w), -1.0)
                block = torch.cat([block, pad])
            gs["block"] = block.to(self.cdev, non_blocking=True)
            gs["out"] = torch.empty((self.world * block.shape[0], self.rec_w), dtype=torch.float32, device=self.cdev)
            gs["handle"] = dist.all_gather_into_tensor(gs["out"], gs["block"], async_op=True)
            gs["gathers"] += 1
            gs["fill"] = 0
        if final:
            self.finish_gather()

    def run(self, steps: int, ring, gather: bool, enqueue=None) -> int:
        """Keep the engine's pipeline full: up to `slots` batches in flight, the results of every step are collected."""
        det, B = self.det, self.B
        enqueue = enqueue or det.enqueue_prepared
        faces = 0
        inflight = []
        nring = len(ring)
        for s in range(steps):
            if len(inflight) == self.slots:
                c = det.wait_counts(inflight.pop(0), B)
                faces += sum(c)
                if gather:
                    self.record_step(c, B)
            inflight.append(enqueue(ring[s % nring], self.thr))
        while inflight:
            c = det.wait_counts(inflight.pop(0), B)
            faces += sum(c)
            if gather:
                self.record_step(c, B)
        if gather:
            self.flush_gather(final=True)
        return faces


def main() -> int:
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        return self_launch(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.library_devices > 0 and env_world is None:
        leg = library_leg(args, args.library_devices, float(args.threshold), seconds=max(args.min_seconds, 0.3))
        print(json.dumps({"metric": "faces/sec", "value": leg.get("faces_per_sec"), "unit": "faces/s", "n_gpus": leg.get("visible_gpus", 0),
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i8", "data": "synthetic",
                          "config": {"workload": leg.get("workload", "library leg (dry)")}, "library_multi_device": leg}), flush=True)
        return 0

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    shared = False
    if args.dry:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
        ndev = torch.cuda.device_count()
        if world > ndev and not args.oversubscribe:
            raise SystemExit(f"bench.py: {world} ranks but only {ndev} visible GPU(s) (one rank per GPU; --oversubscribe to share)")
        torch.cuda.set_device(local_rank % ndev)
        dev = torch.device("cuda", local_rank % ndev)
        shared = world > ndev
    # RCCL refuses two ranks on one device ("Duplicate GPU detected"): when ranks share GPUs (--oversubscribe, how the N > 1 path
    # is exercised on a one-GPU box) the result records travel over gloo from host memory; one rank per GPU uses nccl = RCCL
    backend = "gloo" if (args.dry or shared) else "nccl"
    cdev = torch.device("cpu") if backend == "gloo" else dev               # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)

    from retinaface_amd import shard

    # who takes part in the collectives, as seen THROUGH a collective of the backend that carries the result gather
    comm = {"backend": backend, "ranks_in_communicator": 1, "ranks": [[0, 0 if args.dry else local_rank]]}
    if world > 1:
        me = torch.tensor([[rank, -1 if args.dry else torch.cuda.current_device()]], dtype=torch.int64, device=cdev)
        everyone = torch.empty((world, 2), dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(everyone, me)
        comm["ranks"] = [[int(a), int(b)] for a, b in everyone.cpu().tolist()]
        comm["ranks_in_communicator"] = len({r for r, _ in comm["ranks"]})
        comm["distinct_devices"] = len({d for _, d in comm["ranks"]})
    if backend == "nccl":
        try:
            comm["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            comm["rccl_version"] = None
        comm["rccl_ranks"] = comm["ranks_in_communicator"]

    B, H, W = args.batch, args.height, args.width
    strong = args.global_batch > 0
    B_pad = B                                   # rows per rank and step in the gathered block (strong mode: ceil(G / world))
    if strong:
        if args.global_batch < world:
            raise SystemExit(f"bench.py: --global-batch {args.global_batch} is smaller than the {world} ranks")
        lo, hi = shard.shard_range(args.global_batch, rank, world)
        B, B_pad = hi - lo, -(-args.global_batch // world)
    frame_bytes = H * W * 3
    thr = float(args.threshold)

    if args.dry:
        det = StubEngine(B_pad, args.lanes, args.coalesce)
        slots = det.num_slots()
        lanes_opt = det.lanes
        ring_batches = 4
        prepared_ring = [1000 * rank + k * B for k in range(ring_batches)]
        frames_np = None
    else:
        import retinaface_amd
        from retinaface_amd.frames import synth_frames
        prec = {"fp16": retinaface_amd.PRECISION_FP16, "fp32": retinaface_amd.PRECISION_FP32, "int8": 2}[args.precision]
        det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W), max_batch=B_pad,
                                        model_stem=args.model, lanes=args.lanes, coalesce=args.coalesce)
        slots = det.num_slots()
        lanes_opt = args.lanes or 3
        # ring of distinct frames: more than the pipeline holds AND more than the Infinity Cache, in whole batches
        ring_batches = max(slots, int(np.ceil(args.ring_mb * 1e6 / (frame_bytes * B))))
        frames_np = synth_frames(H, W, ring_batches * B, config=1 + rank)
        frames = torch.from_numpy(np.stack(frames_np)).to(dev)
        torch.cuda.synchronize()
        rows, cols = [H] * B, [W] * B
        prepared_ring = [det.prepare_device_batch([frames[k * B + i].data_ptr() for i in range(B)], rows, cols)
                         for k in range(ring_batches)]
    per_launch = slots // max(lanes_opt, 1)                # steps merged into one launch (super-batch)

    def barrier():
        if not args.dry:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not args.dry:
            torch.cuda.synchronize()

    R = Runner(det, B, B_pad, slots, per_launch, world, cdev, args.dry, thr)
    run, flush_gather, gather_state = R.run, R.flush_gather, R.gs

    do_gather = world > 1
    t0 = time.perf_counter()
    run(max(args.warmup, 1), prepared_ring, do_gather)
    warm_dt = time.perf_counter() - t0
    # at least --steps, at least --min-seconds (estimated from the warm-up rate), in whole super-batches
    steps = args.steps
    if args.exact_steps:
        args.min_seconds, args.regions = 0.0, 1
    if args.min_seconds > 0 and not args.dry:
        est = warm_dt / max(args.warmup, 1)
        steps = max(steps, int(np.ceil(args.min_seconds / max(est, 1e-7))))
        steps = -(-steps // slots) * slots                              # whole pipeline fills
    if world > 1:
        st = torch.tensor([steps], dtype=torch.int64, device=cdev)
        dist.all_reduce(st, op=dist.ReduceOp.MAX)                      # every rank times the same number of steps
        steps = int(st.item())
    # The warm-up rate (the driver passes --warmup 5: one cold pipeline fill) can overestimate the step time 20-fold, so the
    # estimate is checked against the clock: a timed region shorter than 0.8 x --min-seconds is repeated with the step count
    # scaled from its own rate (every rank takes the same decision from the MAX over ranks).
    def timed_region(steps):
        gather_state["gathers"] = 0
        gather_state["images"].zero_()
        barrier()
        t0 = time.perf_counter()
        faces = run(steps, prepared_ring, do_gather)
        barrier()
        return time.perf_counter() - t0, faces

    for attempt in range(4):
        dt, faces = timed_region(steps)
        if args.dry or args.min_seconds <= 0:
            break
        dt_all = torch.tensor([dt], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(dt_all, op=dist.ReduceOp.MAX)
        if float(dt_all.item()) >= 0.8 * args.min_seconds:
            break
        steps = int(np.ceil(steps * 1.1 * args.min_seconds / max(float(dt_all.item()), 1e-6)))
        steps = -(-steps // slots) * slots
    # Three regions of the same K steps, each bracketed by barrier + synchronize; the line reports the MEDIAN region (boxes and
    # clocks drift by a few percent from run to run: one region is one sample) with min / max beside it.
    regions = [(dt, faces)]
    for _ in range(max(args.regions, 1) - 1):
        regions.append(timed_region(steps))
    if world > 1:
        rt = torch.tensor([[r[0], 0.0] for r in regions], dtype=torch.float64, device=cdev)
        rf_ = torch.tensor([[0.0, float(r[1])] for r in regions], dtype=torch.float64, device=cdev)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)             # slowest rank per region
        dist.all_reduce(rf_, op=dist.ReduceOp.SUM)            # faces of all ranks per region
        regions_all = [(float(rt[i, 0]), float(rf_[i, 1])) for i in range(len(regions))]
    else:
        regions_all = [(r[0], float(r[1])) for r in regions]
    order = sorted(range(len(regions_all)), key=lambda i: regions_all[i][1] / regions_all[i][0])
    med = order[len(order) // 2]
    dt, faces = regions[med]
    region_rates = [regions_all[i][1] / regions_all[i][0] for i in range(len(regions_all))]

    # N > 1: every rank's own rate over the median region (a straggler GPU is invisible in MAX / SUM), then rank 0 ALONE -- the other ranks idle at
    # the barrier -- for the same loop without the gather: `scaling_efficiency` = value / (N x that rate), measured inside this run
    per_rank = None
    if world > 1:
        mine = torch.tensor([[float(rank), dt, float(faces), float(steps * B)]], dtype=torch.float64, device=cdev)
        allr = torch.empty((world, 4), dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allr, mine)
        rows_ = sorted(allr.cpu().tolist())
        ips = [r[3] / max(r[1], 1e-12) for r in rows_]
        per_rank = {"images_per_sec": ips, "faces_per_sec": [r[2] / max(r[1], 1e-12) for r in rows_], "seconds": [r[1] for r in rows_],
                    "min_images_per_sec": min(ips), "max_images_per_sec": max(ips), "slowest_rank": int(ips.index(min(ips))),
                    "min_over_max": min(ips) / max(max(ips), 1e-12),
                    "note": "each rank's own clock around its own steps of the median region (the line's value uses the MAX time over ranks)"}
        alone_steps = max(slots, min(steps, int(np.ceil(max(args.extra_seconds, 0.05) / max(dt / steps, 1e-9))) // slots * slots)) if not args.dry else slots
        barrier()
        alone = torch.zeros(2, dtype=torch.float64, device=cdev)
        if rank == 0:
            t0 = time.perf_counter()
            f0 = run(alone_steps, prepared_ring, False)
            if not args.dry:
                torch.cuda.synchronize()
            alone = torch.tensor([time.perf_counter() - t0, float(f0)], dtype=torch.float64, device=cdev)
        barrier()
        dist.broadcast(alone, src=0)
        per_rank["rank0_alone"] = {"steps": alone_steps, "seconds": float(alone[0]), "images_per_sec": alone_steps * B / max(float(alone[0]), 1e-12),
                                   "faces_per_sec": float(alone[1]) / max(float(alone[0]), 1e-12),
                                   "note": "rank 0 runs the same loop with every other rank idle and no gather: the denominator of scaling_efficiency"}

    # N > 1: the BATCH SPLIT north_star prescribes, A/B'd between the ranks (every rank takes part): see split_ab()
    split_leg = None
    if world > 1 and not strong and not args.timed_only and not args.no_extra_configs and args.extra_seconds > 0:
        split_leg = split_ab(args, world, rank, dev, cdev, barrier, backend, thr)

    # N > 1, weak-scaling invocation (what the driver's SCALE run types): a short STRONG-scaling leg of BASELINE.json configs[4] as stated
    # follows -- ONE global batch of 256 int8 images per step, rank r runs shard_range(256, r, N) of it, records gathered in the region
    strong_leg = None
    if world > 1 and not strong and not args.timed_only and not args.no_extra_configs and args.extra_seconds > 0:
        strong_leg = strong_config4(args, world, rank, dev, cdev, barrier, frames if not args.dry else None, thr)

    extra = {}
    if not args.dry and not args.timed_only:
        # burst: ONE super-batch through an empty pipeline, start to finish (what --steps 16 used to time)
        lat = []
        for _ in range(10):
            barrier()
            t = time.perf_counter()
            run(per_launch, prepared_ring, False)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t)
        extra["burst"] = {"steps": per_launch, "images": per_launch * B, "ms": float(np.median(lat) * 1e3),
                          "images_per_sec": per_launch * B / float(np.median(lat))}
        # the same frames every step (input served from L2 / Infinity Cache): the number round 1 reported
        t = time.perf_counter()
        n_same = max(slots * 8, int(0.3 / max(dt / steps, 1e-7)) // slots * slots)
        run(n_same, prepared_ring[:1], False)
        torch.cuda.synchronize()
        extra["cache_resident_input"] = {"images_per_sec": n_same * B / (time.perf_counter() - t), "distinct_frames": B}
        # synchronous latency of one call (what the reference's loop measures: main.cpp:40-52)
        ptrs = [frames[i].data_ptr() for i in range(B)]
        lat = []
        for _ in range(30):
            t = time.perf_counter()
            det.detect_device(ptrs, rows, cols, args.threshold)
            lat.append(time.perf_counter() - t)
        extra["sync_call_ms"] = float(np.median(lat) * 1e3)
        # the same call without the Python binding's per-call work (argument arrays and result objects built once): what a C / C++
        # caller of the C ABI sees = the reference's metric point as the reference measures it: ONE synchronous detectBatchImages of B
        # device-resident frames at a time (RetinaFace.cpp:757 -> :920), nothing in flight besides it
        extra["sync_batch"] = sync_batch_measure(det, ptrs, H, W, args.threshold, reps=240)
        extra["sync_batch"]["ms_per_call_python_binding"] = extra["sync_call_ms"]
        extra["sync_call_ms_c_abi"] = extra["sync_batch"]["ms_per_call"]
        if args.host_seconds > 0:
            extra["host_frames"] = host_frames(det, frames_np, args, slots, B, run, rank)
            extra["sync_batch_host"] = sync_batch_host_measure(
                det, frames, B, H, W, args.threshold,
                make_unsplit=lambda: retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W), max_batch=B_pad,
                                                               model_stem=args.model, lanes=args.lanes, coalesce=args.coalesce))

    # N > 1 (the driver's SCALE run): after everything the ranks measure together, rank 0 times the in-library split over the same N devices
    with_library_leg = world > 1 and not strong and not args.timed_only and not args.no_extra_configs and args.extra_seconds > 0

    tt = torch.tensor([dt, float(faces)], dtype=torch.float64, device=cdev)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, faces_total = float(tmax[0]), float(tsum[1])
    else:
        dt_max, faces_total = dt, float(faces)

    if rank == 0:
        images_total = steps * (args.global_batch if strong else B * world)
        out = {
            "metric": "faces/sec", "value": faces_total / dt_max, "unit": "faces/s",
            "n_gpus": world, "steps": steps, "steps_requested": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / steps * 1e3, "timed_seconds": dt_max,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "burst_only": bool(args.exact_steps),
            "regions": {"n": len(region_rates), "value_is": "median region", "faces_per_sec": region_rates,
                        "min": min(region_rates), "median": sorted(region_rates)[len(region_rates) // 2], "max": max(region_rates)},
            "dtype": {"fp16": "f16", "fp32": "f32", "int8": "i8"}[args.precision], "data": "synthetic",
            "config": {"workload": (f"{args.model} {args.precision} HIP, {W}x{H}, global batch {args.global_batch} split over {world} GPU(s) by shard_range "
                                    f"({baseline_config(args)})" if strong else
                                    f"{args.model} {args.precision} HIP, {W}x{H}, batch {B} per GPU ({baseline_config(args)})"),
                       "global_batch": args.global_batch if strong else B * world, "frame": [H, W], "threshold": args.threshold, "nms": 0.4,
                       "parallelism": f"dp{world} (image sharding, no data-path collective"
                                      + (", result all_gather per super-batch in the timed region)" if world > 1 else ")"),
                       "tickets_in_flight": slots, "lanes": lanes_opt, "steps_coalesced_per_launch": per_launch,
                       "input": f"ring of {ring_batches * B} distinct frames per GPU ({ring_batches * B * frame_bytes / 1e6:.0f} MB, HBM-resident)"},
            "images_per_sec": images_total / dt_max, "ms_per_frame": dt_max / max(images_total / world, 1) * 1e3,
            "faces_per_step": faces_total / steps / (1 if strong else world),
        }
        if world > 1:
            out["result_gather"] = {"collective": "all_gather_into_tensor (RCCL)" if backend == "nccl" else
                                    ("all_gather_into_tensor (gloo, dry)" if args.dry else "all_gather_into_tensor (gloo: ranks share a GPU, RCCL refuses that)"),
                                    "backend": comm["backend"], "rccl_ranks": comm.get("rccl_ranks"), "rccl_version": comm.get("rccl_version"),
                                    "ranks_in_communicator": comm["ranks_in_communicator"], "rank_device": comm["ranks"],
                                    "gathers_in_timed_region": gather_state["gathers"],
                                    "bytes_per_rank_per_gather": int(per_launch * B_pad * R.rec_w * 4),
                                    "records_gathered": int(gather_state["images"].item()), "expected": images_total,
                                    "per_rank": per_rank}
            alone_ips = per_rank["rank0_alone"]["images_per_sec"]
            out["scaling_efficiency"] = {"value": out["images_per_sec"] / max(world * alone_ips, 1e-12), "n_gpus": world,
                                         "is": "images/s of the N-rank job / (N x images/s of rank 0 running ALONE in this same run)",
                                         "rank0_alone_images_per_sec": alone_ips, "scaling": "strong" if strong else "weak",
                                         "on_distinct_devices": comm.get("distinct_devices", 1) == world}
        if strong_leg is not None:
            out["configs4_strong"] = strong_leg
        if split_leg is not None:
            out["batch_split_ab"] = split_leg
        if with_library_leg:
            try:
                out["library_multi_device"] = library_leg(args, world, thr)
            except Exception as e:  # noqa: BLE001
                out["library_multi_device"] = {"error": f"{type(e).__name__}: {e}"}
            finally:
                # the other ranks wait on the rendezvous store for exactly this leg (their GPUs had to be free for it); what rank 0 still measures
                # alone afterwards (per-kernel profile, counter passes) they sit out in the closing barrier, whose watchdog is minutes, not this wait
                dist.distributed_c10d._get_default_store().set("rf_library_leg", "done")
        if args.dry:
            out["dry"] = True
        elif args.timed_only:
            out["timed_only"] = True
        else:
            out.update(extra)
            out.update(device_side_report(args, det, frames, B, H, W, per_launch, prec, images_total, world, dt_max))
            with_configs = (world == 1 and not args.no_extra_configs and args.extra_seconds > 0 and
                            baseline_config(args).startswith("BASELINE.json configs[1]"))
            if with_configs:
                out["configs"] = [dict(id=1, workload=out["config"]["workload"], images_per_sec=out["images_per_sec"], faces_per_sec=out["value"],
                                       ms_per_step=out["ms_per_step"], dtype=out["dtype"], dominant_kernel=out["roofline"]["kernel_instance"],
                                       dominant_kernel_ms=out["roofline"]["kernel_ms"], bound=out["roofline"]["bound"],
                                       bound_frac=out["roofline"]["bound_frac"], hbm_frac_measured=out["roofline"]["hbm_frac_measured"],
                                       useful=out["roofline"]["useful"], sync_batch=out["sync_batch"], sync_batch_host=out.get("sync_batch_host"),
                                       note="the metric point: this line's `value`")]
                out["configs"] += [measure_extra_config(c, args, frames, dev) for c in EXTRA_CONFIGS]
                _FRAME_CACHE.clear()
            # the CPU baseline has the host to itself: every profiler pass of this run has either finished (the metric point's counters, inside
            # device_side_report) or starts after it (round 4 ran the extra configs' passes beside it: 47-51 -> 40-51 images/s)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(frames_np[:B], args, det)
                out["cpu_baseline"]["ran_alone"] = "no profiler pass or GPU measurement of this run overlapped it"
            if with_configs:
                counted = [e for e in out["configs"][1:] if "_pmc" in e]
                finish_extra_counters(start_extra_counters(args, counted), counted, torch.cuda.get_device_properties(0).multi_processor_count)
            if world == 1:
                attach_pipeline_trace(out["roofline"], measure_pipeline_trace(args))
            attach_instruction_mix(out["roofline"], args.precision)
            if args.precision == "int8":
                out["int8_distance_to_fp32"] = int8_contract_summary(args.model)
        assert out["n_gpus"] == args.gpus
        print(json.dumps(out), flush=True)
    det.close()
    if world > 1:
        if with_library_leg:
            # the other ranks released their engines above and wait HERE, on the rendezvous store (host side): rank 0 may still be in the library leg
            import datetime
            store = dist.distributed_c10d._get_default_store()
            if rank != 0:
                store.wait(["rf_library_leg"], datetime.timedelta(seconds=120))      # the library leg is bounded (LIBRARY_LEG_BUDGET_S): 120 s = build + leg + slack; rank 0 sets the key when the leg ends
        dist.barrier()
        dist.destroy_process_group()
    return 0


def sync_batch_measure(det, ptrs, H, W, thr, reps=120, seconds=None):
    """The reference's own calling convention (RetinaFace.cpp:749-940, loop main.cpp:36-52): ONE synchronous rf_detect_batch_device call of
    len(ptrs) device-resident frames at a time, nothing else in flight, timed at the C ABI (argument arrays and result buffers built once)."""
    import ctypes as C
    import numpy as np
    from retinaface_amd._lib import rf_face
    B = len(ptrs)
    pa, ra, ca = (C.c_void_p * B)(*ptrs), (C.c_int * B)(*([H] * B)), (C.c_int * B)(*([W] * B))
    sa = (C.c_int * B)(*([3 * W] * B))
    outb, cnt = (rf_face * (B * det.max_detections))(), (C.c_int * B)()
    lat, t0 = [], time.perf_counter()
    while len(lat) < reps or (seconds is not None and time.perf_counter() - t0 < seconds):
        t = time.perf_counter()
        det._lib.rf_detect_batch_device(det._h, pa, ra, ca, sa, B, C.c_float(thr), outb, det.max_detections, cnt)
        lat.append(time.perf_counter() - t)
    reps = len(lat)
    ms = float(np.median(lat[reps // 6:]) * 1e3)
    nf = int(sum(cnt[i] for i in range(B)))
    return {"batch": B, "ms_per_call": ms, "images_per_sec": B / (ms * 1e-3), "faces_per_sec": nf / (ms * 1e-3), "calls_timed": reps - reps // 6,
            "note": "one synchronous rf_detect_batch_device call at a time (no pipelining, no coalescing)"}


def sync_batch_host_measure(det, frames_dev, B, H, W, thr, seconds=0.25, make_unsplit=None):
    """The reference's calling convention with the frames where its callers hold them (detectBatchImages(vector<cv::Mat>),
    RetinaFace.cpp:749-846: upload + preprocess + infer inside ONE call): ONE synchronous rf_detect_batch of B HOST frames at a time, timed
    at the C ABI -- pageable caller memory (the engine stages it through pinned memory) and memory pinned once with rf_host_register (DMA in
    place).  Round 6 stages and sends the frames of such a call in pipelined pieces (engine.cpp submit(): the host stages piece k + 1 while piece k
    is on the bus); `unsplit` is the same call on an engine built with RF_SYNC_SPLIT=0 (one piece: rounds 1-5), in the same run.  The result
    block must equal the device-frame call's byte for byte."""
    import ctypes as C
    import numpy as np
    from retinaface_amd._lib import rf_face
    host = np.ascontiguousarray(frames_dev[:B].cpu().numpy())
    cap = det.max_detections

    def call_arrays(ptrs):
        return ((C.c_void_p * B)(*ptrs), (C.c_int * B)(*([H] * B)), (C.c_int * B)(*([W] * B)), (C.c_int * B)(*([3 * W] * B)),
                (rf_face * (B * cap))(), (C.c_int * B)())

    def timed(d, fn, arrs):
        pa, ra, ca, sa, outb, cnt = arrs
        lat, t0 = [], time.perf_counter()
        while len(lat) < 12 or (time.perf_counter() - t0 < seconds and len(lat) < 2000):
            t = time.perf_counter()
            st = fn(d._h, pa, ra, ca, sa, B, C.c_float(thr), outb, cap, cnt)
            lat.append(time.perf_counter() - t)
            assert st >= 0 or st == -6, st
        lat = lat[len(lat) // 6:]
        return float(np.median(lat) * 1e3), bytes(outb), [cnt[i] for i in range(B)]

    def measure(d):
        res = {}
        dev_arrs = call_arrays([frames_dev[i].data_ptr() for i in range(B)])
        _, ref_bytes, ref_cnt = timed(d, d._lib.rf_detect_batch_device, dev_arrs)
        ms, ob, oc = timed(d, d._lib.rf_detect_batch, call_arrays([host[i].ctypes.data for i in range(B)]))
        res["pageable"] = {"ms_per_call": ms, "byte_identical_to_device_frames": bool(ob == ref_bytes and oc == ref_cnt)}
        pinned = host.copy()
        d.host_register(pinned)
        ms, ob, oc = timed(d, d._lib.rf_detect_batch, call_arrays([pinned[i].ctypes.data for i in range(B)]))
        d.host_unregister(pinned)
        res["registered"] = {"ms_per_call": ms, "byte_identical_to_device_frames": bool(ob == ref_bytes and oc == ref_cnt)}
        return res

    out = measure(det)
    out.update({"batch": B, "frame": [H, W], "bytes_per_call": B * H * W * 3, "pcie_ms_at_spec": B * H * W * 3 / (PCIE_GEN5_X16_GBS * 1e9) * 1e3,
                "timed_at": "the C ABI (rf_detect_batch with argument / result arrays built once), median after the first sixth of the calls",
                "pipelined_staging": "pageable frames are staged + sent in pieces of ~1.2 MB, staging of piece k + 1 overlapping the transfer of piece k "
                                     "(RF_SYNC_SPLIT=1, the default); registered memory is read in place: nothing to pipeline"})
    if make_unsplit is not None:
        had = os.environ.get("RF_SYNC_SPLIT")
        os.environ["RF_SYNC_SPLIT"] = "0"
        try:
            d0 = make_unsplit()
        finally:
            if had is None:
                os.environ.pop("RF_SYNC_SPLIT", None)
            else:
                os.environ["RF_SYNC_SPLIT"] = had
        u = measure(d0)
        d0.close()
        out["unsplit"] = {"pageable_ms_per_call": u["pageable"]["ms_per_call"], "registered_ms_per_call": u["registered"]["ms_per_call"],
                          "is": "the same calls on an engine built with RF_SYNC_SPLIT=0: staging, then transfer, then compute (rounds 1-5)"}
    return out


def host_frames(det, frames_np, args, slots, B, run, rank):
    """The reference's calling convention as a pipeline: frames in HOST memory, rf_enqueue_batch (pinned staging + one DMA per
    enqueue, upload overlapped with compute).  Two variants: pageable caller memory (staged by the engine's copy threads) and
    caller memory pinned once with rf_host_register (DMA in place)."""
    import numpy as np
    import torch
    H, W = args.height, args.width
    nb = max(2 * slots, 8)                                     # distinct host batches (cycled)
    nb = min(nb, len(frames_np) // B)
    host = np.stack(frames_np[:nb * B]).reshape(nb, B, H, W, 3)
    res = {}
    # what a pinned torch copy of the same bytes reaches on this box (the practical PCIe ceiling)
    pin = torch.from_numpy(host.reshape(-1)).pin_memory()
    dst = torch.empty_like(pin, device="cuda")
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        dst.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    pcie = 3 * pin.numel() / (time.perf_counter() - t) / 1e9
    del dst
    del pin
    for label, register in (("pageable", False), ("registered", True)):
        buf = host.copy()
        if register:
            det.host_register(buf)
        ring = [det.prepare_host_batch([buf[k, i] for i in range(B)]) for k in range(nb)]
        run(2 * slots, ring, False, enqueue=det.enqueue_prepared_host)
        torch.cuda.synchronize()
        steps = slots
        t0 = time.perf_counter()
        faces = 0
        n = 0
        while time.perf_counter() - t0 < args.host_seconds / 2:
            faces += run(steps, ring, False, enqueue=det.enqueue_prepared_host)
            n += steps
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gbs = n * B * H * W * 3 / dt / 1e9
        res[label] = {"images_per_sec": n * B / dt, "faces_per_sec": faces / dt, "pcie_GBs": gbs,
                      "frac_of_pcie_gen5_x16_spec": gbs / PCIE_GEN5_X16_GBS, "frac_of_pinned_copy_measured": gbs / pcie}
        if register:
            det.host_unregister(buf)
    best = max((v for v in res.values() if "images_per_sec" in v), key=lambda v: v["images_per_sec"])
    res.update({"value_host_frames": best["faces_per_sec"], "images_per_sec": best["images_per_sec"],
                "pinned_copy_GBs_measured": pcie, "pcie_gen5_x16_spec_GBs": PCIE_GEN5_X16_GBS,
                "note": "PCIe-inclusive rate of the reference's host-frame API; reported beside `value`, never as `value`"})
    # synchronous host call (rf_detect_batch), as in the reference's loop
    lat = []
    for _ in range(20):
        t = time.perf_counter()
        det.detectBatchImages(frames_np[:B], args.threshold)
        lat.append(time.perf_counter() - t)
    res["sync_call_ms"] = float(np.median(lat) * 1e3)
    return res


# The other BASELINE.json configurations the default invocation also measures (`configs` in the JSON)
EXTRA_CONFIGS = [
    {"id": 2, "model": "mnet-deconv-0517", "precision": "int8", "height": 448, "width": 448, "batch": 32},
    {"id": 3, "model": "mnet25", "precision": "fp16", "height": 896, "width": 1280, "batch": 1},
    {"id": 4, "model": "mnet25", "precision": "int8", "height": 448, "width": 448, "batch": 32,
     "note": "per-GPU shape of configs[4] (256 images = 32 per GPU x 8 GPUs); the 8-GPU job itself: `--gpus 8` (configs4_strong)"},
    # north_star's target matrix "batch {1, 8, 32} on 448x448 and 1280x896" (fp16, mnet25): the cells that are not BASELINE.json configs themselves
    # (448x448 b8 = configs[1] = `value`; 1280x896 b1 = configs[3]).  Timed region + the synchronous call; no per-kernel passes (same kernels).
    {"id": "matrix 448x448 b1", "model": "mnet25", "precision": "fp16", "height": 448, "width": 448, "batch": 1, "matrix": True},
    {"id": "matrix 448x448 b32", "model": "mnet25", "precision": "fp16", "height": 448, "width": 448, "batch": 32, "matrix": True},
    {"id": "matrix 1280x896 b8", "model": "mnet25", "precision": "fp16", "height": 896, "width": 1280, "batch": 8, "matrix": True},
    {"id": "matrix 1280x896 b32", "model": "mnet25", "precision": "fp16", "height": 896, "width": 1280, "batch": 32, "matrix": True},
]

_FRAME_CACHE = {}      # (H, W) -> device tensor of distinct frames, shared by the extra configs of one run


def measure_extra_config(cfg, args, frames_main, dev):
    """One additional BASELINE.json configuration in the same process: its own engine, a ring of distinct HBM-resident frames, a timed
    region of >= --extra-seconds with the pipeline kept full, and the per-kernel HIP-event profile at the launch size (dominant kernel)."""
    import numpy as np
    import torch
    import retinaface_amd
    from retinaface_amd.frames import synth_frames
    H, W, B = cfg["height"], cfg["width"], cfg["batch"]
    prec = {"fp16": retinaface_amd.PRECISION_FP16, "fp32": retinaface_amd.PRECISION_FP32, "int8": 2}[cfg["precision"]]
    cargs = argparse.Namespace(model=cfg["model"], precision=cfg["precision"], height=H, width=W, batch=B, global_batch=0)
    if (H, W) == (args.height, args.width):
        frames, how = frames_main, "the metric point's ring"
    else:
        # frame synthesis is the slow part (0.15 s per 1280x896 frame): 24 seeded frames x 4 circular shifts = 96 distinct frames
        if (H, W) not in _FRAME_CACHE:
            base = synth_frames(H, W, 24, config=3)
            _FRAME_CACHE[(H, W)] = torch.from_numpy(np.stack([np.roll(f, sh, axis=1) for sh in (0, W // 4, W // 2, 3 * W // 4) for f in base])).to(dev)
        frames, how = _FRAME_CACHE[(H, W)], "24 seeded frames x 4 circular shifts"
    det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W), max_batch=B,
                                    model_stem=cfg["model"])
    slots = det.num_slots()
    per_launch = max(slots // 3, 1)
    nb = max(frames.shape[0] // B, 1)
    ring = [det.prepare_device_batch([frames[(k * B + i) % frames.shape[0]].data_ptr() for i in range(B)], [H] * B, [W] * B) for k in range(nb)]
    R = Runner(det, B, B, slots, per_launch, 1, torch.device("cpu"), False, float(args.threshold))
    R.run(2 * slots, ring, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R.run(slots, ring, False)
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / slots
    steps = -(-int(np.ceil(args.extra_seconds / max(est, 1e-7))) // slots) * slots
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        faces = R.run(steps, ring, False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt >= 0.7 * args.extra_seconds:
            break
        steps = -(-int(np.ceil(steps * 1.2 * args.extra_seconds / max(dt, 1e-6))) // slots) * slots
    sync = sync_batch_measure(det, [frames[i % frames.shape[0]].data_ptr() for i in range(B)], H, W, float(args.threshold), reps=90)
    sync_host = sync_batch_host_measure(det, frames, min(B, frames.shape[0]), H, W, float(args.threshold), seconds=0.15) if args.host_seconds > 0 else None
    det.close()
    if cfg.get("matrix"):
        return {"id": cfg["id"], "workload": f"{cfg['model']} {cfg['precision']} HIP, {W}x{H}, batch {B} per GPU (north_star's batch x frame-size matrix)",
                "images_per_sec": steps * B / dt, "faces_per_sec": faces / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "timed_seconds": dt,
                "dtype": {"fp16": "f16", "fp32": "f32", "int8": "i8"}[cfg["precision"]],
                "input": f"{frames.shape[0]} distinct HBM-resident frames ({frames.shape[0] * H * W * 3 / 1e6:.0f} MB; {how})",
                "tickets_in_flight": slots, "sync_batch": sync, "sync_batch_host": sync_host,
                "note": "same kernel instances as the BASELINE config of this frame size: no separate per-kernel / counter passes"}
    eager = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W), max_batch=B,
                                      model_stem=cfg["model"], use_graph=False, lanes=1)
    n_prof = B * per_launch
    prof = eager.profile([frames[i % frames.shape[0]].data_ptr() for i in range(n_prof)], iters=10)
    eager.close()
    dom = max(prof, key=lambda p: p["ms"])
    kernel_ms = sum(p["ms"] for p in prof)
    res = {"id": cfg["id"], "workload": f"{cfg['model']} {cfg['precision']} HIP, {W}x{H}, batch {B} per GPU ({baseline_config(cargs)})",
           "images_per_sec": steps * B / dt, "faces_per_sec": faces / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "timed_seconds": dt,
           "dtype": {"fp16": "f16", "fp32": "f32", "int8": "i8"}[cfg["precision"]],
           "input": f"{frames.shape[0]} distinct HBM-resident frames ({frames.shape[0] * H * W * 3 / 1e6:.0f} MB; {how})",
           "images_per_launch": n_prof, "kernels_ms_per_launch_sequence": kernel_ms, "launches": len(prof),
           "dominant_kernel": dom["kernel"], "dominant_kernel_layers": dom["name"], "dominant_kernel_ms": dom["ms"],
           "dominant_kernel_share_of_gpu_time": dom["ms"] / kernel_ms,
           "frac_layerwise_credit": dom["alg_bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "useful": useful_fractions(dom, cfg["precision"]), "whole_path_useful": useful_fractions(prof, cfg["precision"]),
           "sync_batch": sync, "sync_batch_host": sync_host,
           "bound": None, "bound_frac": None, "hbm_frac_measured": None,
           "_pmc": {"n": n_prof, "precision": cfg["precision"], "model": cfg["model"], "H": H, "W": W, "B": B},
           "_kernels": {p["kernel"]: p["ms"] for p in prof}}
    if "note" in cfg:
        res["note"] = cfg["note"]
    if cfg["precision"] == "int8":
        res["int8_distance_to_fp32"] = int8_contract_summary(cfg["model"])
    return res


def int8_contract_summary(model):
    """The int8 engine's distance to the fp32 oracle for `model`, as measured by the GPU parity suite (tests/test_gpu_parity.py::
    test_int8_contract_over_200_frames: 104 held-out frames, both frame sizes, batches of 8 and 32; metrics of tests/int8_contract.py) -- joined from the newest
    committed summary under profiles/ (the bench's product path may not touch the oracle; the numbers are a property of the shipped calibration, not of this run)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_int8_contract.json")))
    if not files:
        return None
    c = json.load(open(files[-1])).get("contract", {}).get(model)
    if not c:
        return None
    keep = ("frames", "faces", "same_count", "anchor_agreement", "anchor_iou_worst", "anchor_iou_p01", "anchor_iou_mean", "iou_worst", "iou_p01", "iou_mean", "dscore_max")
    out = {k: c[k] for k in keep if k in c}
    out.update({"source": os.path.relpath(files[-1], ROOT), "parity_bar": "bit-exact against oracle/int8_forward.py (every int8 activation); THIS is the reported distance to fp32",
                "iou_worst_includes": "NMS-winner flips between neighbouring anchors of one face (the fp16 engine: 0.877 on the same metric); anchor_iou_* is the regression error alone"})
    return out


def start_extra_counters(args, entries):
    """The additional configurations' hardware counters: ONE set of three rocprofv3 --pmc passes whose probe process runs the fp16
    1280x896 configuration and the int8 448x448 configuration one after the other (the two int8 configs launch the same kernel
    instances on the same shapes: one measurement serves both).  Runs in a thread the caller joins right away (since round 5 AFTER the
    CPU baseline: the passes start Python processes that build engines and read counter databases -- host work that biased the baseline)."""
    import threading
    seen, cfgs = set(), []
    for e in entries:
        key = (e["_pmc"]["precision"], e["_pmc"]["H"], e["_pmc"]["W"])
        if key not in seen:
            seen.add(key)
            cfgs.append(e["_pmc"])
    holder = {"result": None, "cfgs": cfgs}
    th = threading.Thread(target=lambda: holder.__setitem__("result", measure_counters(args, cfgs) if cfgs else None), daemon=True)
    th.start()
    holder["thread"] = th
    return holder


def finish_extra_counters(holder, entries, n_cu):
    holder["thread"].join(timeout=400)
    counters = holder["result"]
    for e in entries:
        pm, kern = e.pop("_pmc"), e.pop("_kernels")
        if not counters:
            e["counters"] = "no counter pass (rocprofv3 unavailable or failed)"
            continue
        mine = [c for c in counters if c["dtype"] == pm["precision"] and c["kernel"] == e["dominant_kernel"]]
        if not mine:
            continue
        m = max(mine, key=lambda c: c["hbm_bytes"])
        phys = physical_fractions([dict(m, launches_per_pass=1)], e["dominant_kernel_ms"], 4 * n_cu)
        e.update({"bound": phys["bound"], "bound_frac": phys["bound_frac"], "hbm_frac_measured": phys["hbm_frac_measured"],
                  "valu_active": phys.get("valu_active"), "mfma_busy": phys.get("mfma_busy"), "lds_active": phys.get("lds_active"),
                  "hbm_bytes_dominant_kernel": phys["hbm_bytes"],
                  "counters": "rocprofv3 --pmc passes of this run (collected after the CPU baseline)" +
                              ("; int8 kernel instances and shapes are the same for both int8 configs: one measurement" if pm["precision"] == "int8" else "")})
        path = [c for c in counters if c["dtype"] in (pm["precision"], "") and c["kernel"] in kern]
        if path and all(c.get("gpu_cycles") for c in path):
            whole = physical_fractions(path, e["kernels_ms_per_launch_sequence"], 4 * n_cu)
            e["whole_path"] = {"hbm_bytes_per_image_measured": whole["hbm_bytes"] / e["images_per_launch"], "bound": whole["bound"],
                               "hbm_frac_measured_in_kernels": whole["hbm_frac_measured"], "valu_active": whole.get("valu_active"),
                               "mfma_busy": whole.get("mfma_busy")}


def split_ab(args, world, rank, dev, cdev, barrier, backend, thr, G=256):
    """north_star: "RCCL over xGMI only for the batch split / gather" (ncclGroupStart{ncclSend / ncclRecv} scatter, "alternative to measure
    against: direct H2D").  One step = ONE batch of G = 256 mnet25 int8 448 x 448 frames (BASELINE.json configs[4]); rank r runs shard_range(G, r, N) of
    it through ONE synchronous call on its own engine and the per-image counts come back through an all_gather.  Three ways the slice reaches rank r:
      rccl_send_recv   all G frames live on rank 0's GPU; torch.distributed.batch_isend_irecv = one ncclGroupStart{ncclSend x (N-1)} on rank 0 and one
                       ncclRecv on every other rank (backend nccl = RCCL over xGMI), slice 0 is read in place;
      direct_h2d       every rank's slice waits in ITS OWN pinned host memory and crosses PCIe with one hipMemcpyAsync (no inter-GPU traffic at all);
      resident         the slice is already on the rank's GPU (the floor: detect + gather only).
    ms per step each, MAX over ranks.  The in-library split (hipMemcpyPeerAsync, one handle over N devices) is library_multi_device.split_ab.
    With ranks sharing a GPU (--oversubscribe) or --dry the backend is gloo and the 'xGMI' legs move host tensors: a rehearsal of the code path."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from retinaface_amd import shard
    H = W = 448
    lo, hi = shard.shard_range(G, rank, world)
    Bs = hi - lo
    fb = 16 if args.dry else H * W * 3
    rehearsal = backend != "nccl"
    # Everything that can fail for a rank-local reason (engine build, allocations) happens BEFORE the first collective of this leg, and the ranks agree on
    # the outcome: one rank that could not set up must not leave the others waiting in a send / recv it will never post.
    det, setup_error = None, None
    try:
        if args.dry:
            det = StubEngine(max(Bs, 1), 0, 0)
            whole = torch.zeros((G, 16), dtype=torch.uint8) if rank == 0 else None           # stand-in "frames": 16 bytes each
            mine = torch.zeros((max(Bs, 1), 16), dtype=torch.uint8)
            host = mine.clone()
            prep = lambda: 1000 * rank                                                        # noqa: E731
        else:
            import retinaface_amd
            from retinaface_amd.frames import synth_frames
            det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=2, net_hw=(H, W), max_batch=max(Bs, 1), model_stem="mnet25")
            src = np.stack(synth_frames(H, W, 64, config=47))
            whole_np = src[np.arange(G) % 64]
            xdev = cdev if rehearsal else dev                      # where the tensors of the send / recv live (gloo: host)
            whole = torch.from_numpy(whole_np).to(xdev) if rank == 0 else None
            mine = torch.empty((max(Bs, 1), H, W, 3), dtype=torch.uint8, device=dev)       # the rank's slice on ITS GPU: what the engine reads
            recv = mine if not rehearsal else torch.empty((max(Bs, 1), H, W, 3), dtype=torch.uint8)
            host = torch.from_numpy(whole_np[lo:hi].copy()).pin_memory()
            mine[:Bs].copy_(host)
            torch.cuda.synchronize()
            pb = det.prepare_device_batch([mine[i].data_ptr() for i in range(Bs)], [H] * Bs, [W] * Bs)
            prep = lambda: pb                                                                 # noqa: E731
            sum(det.wait_counts(det.enqueue_prepared(pb, thr), Bs)) if Bs else 0              # first (eager) call of this batch size
    except Exception as e:  # noqa: BLE001
        setup_error = f"{type(e).__name__}: {e}"
    ok = torch.tensor([0.0 if setup_error else 1.0], device=cdev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) == 0.0:
        if det is not None:
            det.close()
        return {"error": setup_error or "another rank could not set this leg up", "skipped": True}

    def detect():
        if Bs == 0:
            return 0
        return sum(det.wait_counts(det.enqueue_prepared(prep(), thr), Bs))

    def move_rccl():
        ops = []
        if rank == 0:
            for r in range(1, world):
                a, b = shard.shard_range(G, r, world)
                if b > a:
                    ops.append(dist.P2POp(dist.isend, whole[a:b], r))
            if not args.dry and Bs:
                mine[:Bs].copy_(whole[lo:hi], non_blocking=True)        # slice 0: on this GPU already (a device-to-device copy stands in for "read in place")
        elif Bs:
            ops.append(dist.P2POp(dist.irecv, (mine if args.dry else recv)[:Bs], 0))
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        if not args.dry:
            if rehearsal and rank != 0 and Bs:
                mine[:Bs].copy_(recv[:Bs], non_blocking=True)
            torch.cuda.synchronize()                                    # the engine's lanes are their own streams: the slice must have landed

    def move_h2d():
        if not args.dry and Bs:
            mine[:Bs].copy_(host, non_blocking=True)
            torch.cuda.synchronize()

    out = {"workload": f"mnet25 int8 HIP, 448x448, ONE batch of {G} frames per step split over {world} rank(s) by shard_range, one synchronous call per rank "
                       "(BASELINE.json configs[4]); how the slice reaches the rank is what varies",
           "global_batch": G, "bytes_per_step": G * fb, "backend": backend, "rehearsal_not_xgmi": bool(rehearsal or args.dry), "legs": {}}
    cnt = torch.zeros((max(-(-G // world), 1),), dtype=torch.int32)
    allc = torch.empty((world * cnt.numel(),), dtype=torch.int32, device=cdev)
    budget = 0.05 if args.dry else max(args.extra_seconds, 0.2)
    for name, move in (("resident", lambda: None), ("rccl_send_recv", move_rccl), ("direct_h2d", move_h2d)):
        for _ in range(2):                                   # warm-up: communicator / engine / graph
            move()
            detect()
        barrier()
        lat, faces, t_start = [], 0, time.perf_counter()
        while True:
            go = torch.tensor([1.0 if (len(lat) < 3 or time.perf_counter() - t_start < budget) and len(lat) < 400 else 0.0], device=cdev)
            dist.broadcast(go, src=0)                        # every rank runs the same number of steps (rank 0 decides)
            if float(go.item()) == 0.0:
                break
            barrier()
            t = time.perf_counter()
            move()
            faces = detect()
            cnt.zero_()
            cnt[0] = faces
            dist.all_gather_into_tensor(allc, cnt.to(cdev))
            if not args.dry:
                torch.cuda.synchronize()
            lat.append(time.perf_counter() - t)
        med = torch.tensor([float(np.median(lat))], dtype=torch.float64, device=cdev)
        dist.all_reduce(med, op=dist.ReduceOp.MAX)
        tot = torch.tensor([float(faces)], dtype=torch.float64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        ms = float(med.item()) * 1e3
        out["legs"][name] = {"ms_per_step": ms, "images_per_sec": G / (ms * 1e-3), "faces_per_step": float(tot.item()), "steps_timed": len(lat)}
    base = out["legs"]["resident"]["ms_per_step"]
    for name in ("rccl_send_recv", "direct_h2d"):
        leg = out["legs"][name]
        leg["split_ms_over_resident"] = leg["ms_per_step"] - base
        moved = (G - (shard.shard_range(G, 0, world)[1])) * fb if name == "rccl_send_recv" else G * fb
        leg["bytes_moved_per_step"] = moved
        leg["GBs_if_serial"] = moved / max(leg["split_ms_over_resident"] * 1e-3, 1e-9) / 1e9 if leg["split_ms_over_resident"] > 0 else None
    same = len({round(v["faces_per_step"]) for v in out["legs"].values()}) == 1
    out["same_detections_every_leg"] = bool(same)
    out["winner"] = None if out["rehearsal_not_xgmi"] else min(("rccl_send_recv", "direct_h2d"), key=lambda k: out["legs"][k]["ms_per_step"])
    det.close()
    return out


def strong_config4(args, world, rank, dev, cdev, barrier, frames, thr, G=256):
    """BASELINE.json configs[4] as stated, on this job's N ranks: a step is ONE batch of G = 256 mnet25 int8 448x448 images, rank r runs
    shard_range(G, r, N) of it, every image's record comes back through the all_gather inside the timed region.  Every rank calls this."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from retinaface_amd import shard
    lo, hi = shard.shard_range(G, rank, world)
    Bs, B_pad = hi - lo, -(-G // world)
    H = W = 448
    if args.dry:
        det = StubEngine(B_pad, 0, 0)
        ring = [1000 * rank + k * max(Bs, 1) for k in range(4)]
    else:
        import retinaface_amd
        from retinaface_amd.frames import synth_frames
        if frames is None or tuple(frames.shape[1:3]) != (H, W):
            frames = torch.from_numpy(np.stack(synth_frames(H, W, 4 * B_pad, config=41 + rank))).to(dev)
        det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=2, net_hw=(H, W), max_batch=B_pad, model_stem="mnet25")
        nb = max(frames.shape[0] // max(Bs, 1), 1)
        ring = [det.prepare_device_batch([frames[(k * Bs + i) % frames.shape[0]].data_ptr() for i in range(Bs)], [H] * Bs, [W] * Bs)
                for k in range(nb)]
    slots = det.num_slots()
    per_launch = max(slots // 3, 1)
    R = Runner(det, Bs, B_pad, slots, per_launch, world, cdev, args.dry, thr)
    t0 = time.perf_counter()
    R.run(slots, ring, True)
    est = (time.perf_counter() - t0) / slots
    steps = slots if args.dry else -(-int(np.ceil(args.extra_seconds / max(est, 1e-7))) // slots) * slots
    st = torch.tensor([steps], dtype=torch.int64, device=cdev)
    dist.all_reduce(st, op=dist.ReduceOp.MAX)
    steps = int(st.item())
    R.gs["gathers"] = 0
    R.gs["images"].zero_()
    barrier()
    t0 = time.perf_counter()
    faces = R.run(steps, ring, True)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(faces)], dtype=torch.float64, device=cdev)
    tmax, tsum = tt.clone(), tt.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    det.close()
    return {"workload": f"mnet25 int8 HIP, 448x448, global batch {G} split over {world} GPU(s) by shard_range (BASELINE.json configs[4] as stated)",
            "scaling": "strong", "global_batch": G, "images_per_rank": B_pad, "steps": steps, "timed_seconds": float(tmax[0]),
            "images_per_sec": steps * G / float(tmax[0]), "faces_per_sec": float(tsum[1]) / float(tmax[0]),
            "ms_per_step": float(tmax[0]) / steps * 1e3, "dtype": "i8",
            "result_gather": {"gathers_in_timed_region": R.gs["gathers"], "records_gathered": int(R.gs["images"].item()), "expected": steps * G}}


def library_leg(args, n_devices, thr, G=256, seconds=None):
    """BASELINE.json configs[4] as stated THROUGH THE LIBRARY: one handle whose rf_options.devices lists `n_devices` ordinals (multi.cpp: one
    engine + one host thread per entry, contiguous ceil(G / N) image slices), G = 256 mnet25 int8 448x448 frames that ALL live on GPU 0, one
    synchronous rf_detect_batch_device call per step.  Slice g's engine finds its frames on another device (hipPointerGetAttributes, cached per
    allocation) and pulls them over xGMI with one hipMemcpyPeerAsync each before it launches: north_star's "batch split" -- the part of the
    multi-GPU design `--gpus N` alone (N independent ranks) never exercises.  With fewer visible GPUs than entries the ordinals wrap around and
    RF_FORCE_SCATTER=1 makes every frame travel (peer copy device k -> device k): a rehearsal of the code path, not of xGMI."""
    import numpy as np
    import torch
    if args.dry:
        return {"dry": True, "devices": list(range(n_devices)), "global_batch": G, "images_per_engine": -(-G // n_devices),
                "split_ab": {"peer_copy_per_slice": None, "peer_copy_per_frame": None, "host_frames_direct_h2d": None,
                             "rccl_send_recv": "see batch_split_ab (needs one process per GPU)", "measured_over_xgmi": False},
                "leg_budget_seconds": LIBRARY_LEG_BUDGET_S}
    import retinaface_amd
    from retinaface_amd.frames import synth_frames
    H = W = 448
    seconds = seconds if seconds is not None else max(args.extra_seconds, 0.3)
    ndev = torch.cuda.device_count()
    devices = [i % ndev for i in range(n_devices)]
    forced = ndev < n_devices
    distinct = sorted(set(devices))
    peer = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in distinct] for i in distinct]
    with torch.cuda.device(0):
        frames = torch.from_numpy(np.stack(synth_frames(H, W, 64, config=43))).to("cuda:0")
        torch.cuda.synchronize()
    ptrs = [frames[i % 64].data_ptr() for i in range(G)]
    one = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=2, net_hw=(H, W), max_batch=32, model_stem="mnet25", device=0)
    want = one.detect_device(ptrs, [H] * G, [W] * G, thr)
    sync_one = sync_batch_measure(one, ptrs, H, W, thr, reps=12)           # timed at the C ABI: the Python binding spends ~3 ms building 256 images' result objects
    one.close()
    def build_multi(per_frame):
        """one handle over `devices`; RF_FORCE_SCATTER / RF_SCATTER_PER_FRAME are read by the engines when they are built"""
        had = {k: os.environ.get(k) for k in ("RF_FORCE_SCATTER", "RF_SCATTER_PER_FRAME")}
        if forced:
            os.environ["RF_FORCE_SCATTER"] = "1"
        if per_frame:
            os.environ["RF_SCATTER_PER_FRAME"] = "1"
        else:
            os.environ.pop("RF_SCATTER_PER_FRAME", None)
        try:
            return retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=2, net_hw=(H, W), max_batch=32, model_stem="mnet25",
                                             devices=devices)
        finally:
            for k, v in had.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    key = lambda res: [[(d.anchor_index, d.as_row().tobytes()) for d in r] for r in res]      # noqa: E731
    t_leg = time.perf_counter()
    multi = build_multi(False)
    s0 = multi.scatter_stats()
    got = multi.detect_device(ptrs, [H] * G, [W] * G, thr)
    s1 = multi.scatter_stats()                                             # the counters of exactly ONE 256-image call
    identical = key(got) == key(want)
    sync = sync_batch_measure(multi, ptrs, H, W, thr, reps=12, seconds=seconds)
    calls = 1
    nd = multi.num_devices()
    per = -(-G // n_devices)
    travelled = G if forced else sum(min(per, max(0, G - g * per)) for g in range(n_devices) if devices[g] != 0)
    med = sync["ms_per_call"] * 1e-3
    # ---- the split A/B (VERDICT r5 next #2a): how the slices reach their engines, ms per 256-image call each.  Runs whenever frames travel
    # (more than one visible GPU, or the forced rehearsal); every entry null on a plain one-GPU run.  The RCCL send / recv variant needs one
    # process per GPU and lives in the N-rank part of this bench: `batch_split_ab`.
    split = {"peer_copy_per_slice": None, "peer_copy_per_frame": None, "host_frames_direct_h2d": None, "rccl_send_recv": "see batch_split_ab (needs one process per GPU)",
             "what": "ms per rf_detect_batch* call of 256 images through ONE handle over the devices; per_slice = one hipMemcpyPeerAsync per contiguous run of "
                     "frames (round 6 default), per_frame = one per frame (rounds 3-5, RF_SCATTER_PER_FRAME=1), host_frames_direct_h2d = the same frames in pinned "
                     "HOST memory through rf_detect_batch: each engine uploads its own slice over its own PCIe link, no inter-GPU traffic",
             "measured_over_xgmi": bool(not forced and len(distinct) > 1)}
    if forced or len(distinct) > 1:
        budget_left = lambda: LIBRARY_LEG_BUDGET_S - (time.perf_counter() - t_leg)      # noqa: E731
        split["peer_copy_per_slice"] = {"ms_per_call": sync["ms_per_call"], "peer_copies_per_call": (s1["peer_copies"] - s0["peer_copies"]) / calls,
                                        "frames_scattered_per_call": (s1["frames"] - s0["frames"]) / calls}
        if budget_left() > 6:
            pf = build_multi(True)
            a0 = pf.scatter_stats()
            same_pf = key(pf.detect_device(ptrs, [H] * G, [W] * G, thr)) == key(want)
            a1 = pf.scatter_stats()
            spf = sync_batch_measure(pf, ptrs, H, W, thr, reps=12, seconds=min(seconds, 0.5))
            split["peer_copy_per_frame"] = {"ms_per_call": spf["ms_per_call"], "peer_copies_per_call": a1["peer_copies"] - a0["peer_copies"],
                                            "frames_scattered_per_call": a1["frames"] - a0["frames"], "detections_identical": bool(same_pf)}
            pf.close()
        if budget_left() > 4:
            import ctypes as C
            from retinaface_amd._lib import rf_face
            hostbuf = np.ascontiguousarray(frames.cpu().numpy()[np.arange(G) % 64])
            multi.host_register(hostbuf)
            pa = (C.c_void_p * G)(*[hostbuf[i].ctypes.data for i in range(G)])
            ra, ca, sa = (C.c_int * G)(*([H] * G)), (C.c_int * G)(*([W] * G)), (C.c_int * G)(*([3 * W] * G))
            outb, cnt = (rf_face * (G * multi.max_detections))(), (C.c_int * G)()
            lat = []
            t0 = time.perf_counter()
            while len(lat) < 6 or (time.perf_counter() - t0 < min(seconds, 0.5) and len(lat) < 200):
                t = time.perf_counter()
                multi._lib.rf_detect_batch(multi._h, pa, ra, ca, sa, G, C.c_float(thr), outb, multi.max_detections, cnt)
                lat.append(time.perf_counter() - t)
            want_counts = [len(r) for r in want]
            split["host_frames_direct_h2d"] = {"ms_per_call": float(np.median(lat[1:]) * 1e3), "bytes_uploaded_per_call": G * H * W * 3,
                                               "face_counts_identical": [cnt[i] for i in range(G)] == want_counts,
                                               "note": "caller memory pinned once with rf_host_register (DMA in place)"}
            multi.host_unregister(hostbuf)
        done = [k for k in ("peer_copy_per_slice", "peer_copy_per_frame", "host_frames_direct_h2d") if isinstance(split[k], dict)]
        split["fastest"] = min(done, key=lambda k: split[k]["ms_per_call"]) if done else None
        split["fastest_is_a_measurement_of_xgmi"] = split["measured_over_xgmi"]
    multi.close()
    return {"workload": f"mnet25 int8 HIP, 448x448, ONE handle over devices {devices}, {G} images per rf_detect_batch_device call, all frames resident on GPU 0 "
                        "(BASELINE.json configs[4] as stated, through rf_options.devices / multi.cpp)",
            "devices": devices, "engines": nd, "visible_gpus": ndev, "forced_scatter_rehearsal": forced, "peer_access": {"devices": distinct, "matrix": peer},
            "images_per_call": G, "images_per_engine": per, "calls_timed": sync["calls_timed"], "ms_per_call": med * 1e3, "images_per_sec": G / med,
            "faces_per_sec": sync["faces_per_sec"], "timed_at": "the C ABI (rf_detect_batch_device with argument / result arrays built once)",
            "frames_scattered_per_call": travelled, "bytes_scattered_per_call": travelled * H * W * 3,
            "scatter_GBs_at_this_rate": travelled * H * W * 3 / med / 1e9,
            "split_ab": split, "leg_seconds": time.perf_counter() - t_leg, "leg_budget_seconds": LIBRARY_LEG_BUDGET_S,
            "single_engine_same_call": {"ms_per_call": sync_one["ms_per_call"], "images_per_sec": sync_one["images_per_sec"],
                                        "note": "one engine on GPU 0 takes the same 256-image call (its eight 32-image chunks coalesce into one 256-image launch sequence)"},
            "detections_identical_to_single_engine": bool(identical), "dtype": "i8", "scaling": "strong",
            "note": "one synchronous call at a time: the per-engine slice of 32 images is a small-batch launch sequence (latency-bound); the pipelined "
                    "per-GPU rate is `configs`[id 4] / the N-rank `value`"}


LIBRARY_LEG_BUDGET_S = 20.0      # rank 0 runs the library leg while the other ranks wait on the store (120 s): optional parts are skipped when it runs out


SQ_COUNTERS = "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"


def measure_counters(args, cfgs):
    """Hardware counters per kernel instance, measured NOW: three rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | the SQ set: they
    do not fit one pass) over tools/probes/pmc_probe.py = eager launches of the same engine configuration(s) at the same launch size.
    `cfgs` = [{n, precision, model, H, W, B}, ...]; several configurations share one process per pass, their kernels are told apart by
    element type (`dtype` of every entry) and grid.  Corrections as MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2; SQ_ACTIVE_INST_* count
    quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs).  Counter passes carry no trace flag.  Returns
    a list of {kernel, dtype, grid, launches_per_pass, hbm_bytes, valu_quad, lds_quad, mfma_cycles, gpu_cycles} (one entry per kernel
    symbol and grid), or None when rocprofv3 is not there / fails (the caller then joins the newest committed summary under profiles/)."""
    import shutil
    import subprocess
    import tempfile
    if args.no_pmc or shutil.which("rocprofv3") is None:
        return None
    # already running under a profiler (someone is tracing this very bench run): do not nest a second one
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_summary
        tmp = tempfile.mkdtemp(prefix="rf_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        dbs = {}
        for tag, counters in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("SQ", SQ_COUNTERS.split())):
            out = os.path.join(tmp, tag)
            cmd = ["rocprofv3", "--pmc"] + counters + ["-d", out, "-o", "pmc", "--", sys.executable,
                   os.path.join(ROOT, "tools", "probes", "pmc_probe.py"), "--multi", json.dumps(cfgs)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=60 + 40 * len(cfgs))
            found = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not found:
                if tag == "SQ":
                    break                     # traffic alone is still worth having
                return None
            dbs[tag] = found[0]
        f, w = pmc_summary.per_kernel(dbs["FETCH_SIZE"], "FETCH_SIZE"), pmc_summary.per_kernel(dbs["WRITE_SIZE"], "WRITE_SIZE")
        sq = {c: pmc_summary.per_kernel(dbs["SQ"], c) for c in SQ_COUNTERS.split()} if "SQ" in dbs else {}
        keys = [k for k in set(f) | set(w) if k[0].startswith("_ZN2rf") or "rf::" in k[0]]
        res = []
        for dt in sorted({pmc_summary.dtype_of(k[0]) for k in keys}):
            mine = [k for k in keys if pmc_summary.dtype_of(k[0]) == dt]
            base = min((f[k][0] for k in mine if k in f and f[k][0]), default=1)       # launches of a once-per-pass kernel of this type
            for k in mine:
                g = lambda c: sq.get(c, {}).get(k, (0, None))[1]       # noqa: E731
                res.append({"kernel": pmc_summary.descriptor(k[0]), "dtype": dt, "grid": k[1],
                            "launches_per_pass": max(1, round(f.get(k, (base, 0))[0] / base)),
                            "hbm_bytes": 2 * f.get(k, (0, 0.0))[1] * 1024.0 + w.get(k, (0, 0.0))[1] * 1024.0,
                            "valu_quad": g("SQ_ACTIVE_INST_VALU"), "lds_quad": g("SQ_ACTIVE_INST_LDS"),
                            "mfma_cycles": g("SQ_VALU_MFMA_BUSY_CYCLES"),
                            "gpu_cycles": (g("GRBM_GUI_ACTIVE") / 8.0) if g("GRBM_GUI_ACTIVE") else None})
        shutil.rmtree(tmp, ignore_errors=True)
        return res or None
    except Exception:  # noqa: BLE001
        return None


def useful_fractions(prof, precision):
    """How much of the HBM and matrix peaks is work the layers REQUIRE, for one rf_profile entry or a list of them: compulsory bytes of the
    (fused) launch / kernel time / 8 TB/s, and layer MACs x 2 / kernel time / the dense MFMA peak of the precision.  (`frac` of the roofline
    is the measured occupancy of the binding resource -- issue slots, bytes moved -- whether or not the work was necessary.)"""
    prof = prof if isinstance(prof, list) else [prof]
    ms = sum(p["ms"] for p in prof)
    cb, macs = sum(p["compulsory_bytes"] for p in prof), sum(p["macs"] for p in prof)
    return {"hbm_compulsory_bytes": cb, "hbm_compulsory_GBs": cb / (ms * 1e-3) / 1e9, "hbm_compulsory_frac": cb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "mfma_layer_TFLOPs": 2 * macs / (ms * 1e-3) / 1e12, "mfma_layer_frac": 2 * macs / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[precision],
            "is": "compulsory HBM bytes of the fused launch(es) / time / 8 TB/s; layer MACs x 2 / time / dense MFMA peak"}


def physical_fractions(entries, ms, n_simd):
    """What the hardware did during `entries` (one or more kernel launches that took `ms` in total by HIP events): fraction of
    the HBM peak actually moved (PMC bytes / time / 8 TB/s), of the matrix pipes busy, of the VALU / LDS issue cycles used
    (chip-wide: busy cycles / (GPU-active cycles x SIMDs)).  `bound` = the resource with the highest fraction."""
    hbm = sum(e["hbm_bytes"] * e["launches_per_pass"] for e in entries)
    out = {"hbm_bytes": hbm, "hbm_GBs_measured": hbm / (ms * 1e-3) / 1e9, "hbm_frac_measured": hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    cyc = sum((e["gpu_cycles"] or 0) * e["launches_per_pass"] for e in entries)
    if cyc and all(e["valu_quad"] is not None for e in entries):
        cap = cyc * n_simd
        out["mfma_busy"] = sum(e["mfma_cycles"] * e["launches_per_pass"] for e in entries) / cap
        out["valu_active"] = 4.0 * sum(e["valu_quad"] * e["launches_per_pass"] for e in entries) / cap
        out["lds_active"] = 4.0 * sum(e["lds_quad"] * e["launches_per_pass"] for e in entries) / cap
    cands = {"hbm": out["hbm_frac_measured"], "mfma": out.get("mfma_busy", 0.0), "valu_issue": out.get("valu_active", 0.0),
             "lds_issue": out.get("lds_active", 0.0)}
    out["bound"] = max(cands, key=cands.get)
    out["bound_frac"] = cands[out["bound"]]
    return out


def device_side_report(args, det, frames, B, H, W, per_launch, prec, images_total, world, dt_max):
    """pre / infer / post split, per-kernel HIP-event timing and the roofline object."""
    import numpy as np
    import torch
    import retinaface_amd
    ptrs = [frames[i].data_ptr() for i in range(B)]
    rows, cols = [H] * B, [W] * B
    eager = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W),
                                      max_batch=B, model_stem=args.model, use_graph=False, lanes=1, coalesce=args.coalesce)
    split = []
    for _ in range(20):
        eager.detect_device(ptrs, rows, cols, args.threshold)
        split.append(eager.last_timings())
    med = {k: float(np.median([s[k] for s in split])) for k in split[0]}
    # per-kernel HIP-event timing on the engine's own stream, at the size one launch really processes (a super-batch), on
    # DISTINCT frames
    n_prof = B * per_launch
    prof_ptrs = [frames[i % frames.shape[0]].data_ptr() for i in range(n_prof)]
    prof = eager.profile(prof_ptrs, iters=args.profile_iters)
    prof8 = eager.profile(ptrs, iters=args.profile_iters)
    eager.close()
    dom = max(prof, key=lambda p: p["ms"])
    kernel_ms = sum(p["ms"] for p in prof)
    alg_total = sum(p["alg_bytes"] for p in prof)
    n_simd = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    # Hardware counters (HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE; VALU / LDS / MFMA busy) of the same launches, measured now in
    # three separate rocprofv3 --pmc passes (PMC collection cannot run inside this process); when rocprofv3 is not usable the
    # newest committed traffic summary for this (frame, precision) is joined instead and the SQ fractions stay null.
    counters = measure_counters(args, [{"n": n_prof, "precision": args.precision, "model": args.model, "H": H, "W": W, "B": B}]) \
        if world == 1 else None      # N > 1: the other ranks are waiting at the barrier
    if counters:
        counters = [e for e in counters if e["dtype"] in (args.precision, "")]
    dom_phys = path_phys = None
    mine = []
    traffic = traffic_src = None
    if counters:
        mine = [e for e in counters if e["kernel"] == dom["kernel"]]
        if mine:
            mine = [max(mine, key=lambda e: e["hbm_bytes"])]
            dom_phys = physical_fractions([dict(mine[0], launches_per_pass=1)], dom["ms"], n_simd)
            traffic, traffic_src = dom_phys["hbm_bytes"], "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH x 2)"
        path_phys = physical_fractions(counters, kernel_ms, n_simd)
    key = f"{H}x{W}_{args.precision}"
    cand = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(".json") and "pmc_hbm_traffic" in f),
                  reverse=True)
    for f in ([] if traffic is not None else cand):
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", f)))
        except Exception:  # noqa: BLE001
            continue
        if j.get("workload_key", "448x448_fp16" if "448_fp16" in f else None) != key:
            continue
        n_img = j.get("images_per_launch", 128 if "_n128_" in f else 8)
        for k in j["kernels"]:
            if k["kernel"] == dom["kernel"] and traffic is None:
                traffic, traffic_src = k["hbm_bytes_per_launch"] * n_prof / n_img, f
        if traffic is not None:
            break
    images_per_sec_gpu = images_total / world / dt_max
    layerwise = dom["alg_bytes"] / (dom["ms"] * 1e-3) / 1e9
    # `frac` = the PHYSICAL fraction of the binding resource of the dominant kernel (always <= 1): HBM bytes moved / time / 8 TB/s, or
    # busy issue cycles of the VALU / matrix / LDS pipes / (GPU-active cycles x SIMDs), whichever is largest (`bound`).  `achieved`,
    # `peak`, `unit` describe that same resource.  The contract's layer-wise figure (ALGORITHMIC bytes of SURVEY.md 8d / the kernel's
    # HIP-event duration, against the HBM peak) counts bytes a fused kernel never moves, so it can exceed 1: it is fusion credit and
    # stays in `frac_layerwise_credit` / `hbm_layerwise`.
    bound = (dom_phys or {}).get("bound")
    clock_ghz = None
    if dom_phys and mine and mine[0].get("gpu_cycles"):
        clock_ghz = mine[0]["gpu_cycles"] / (dom["ms"] * 1e-3) / 1e9          # effective shader clock during the counter pass
    if bound == "hbm":
        res_achieved, res_peak, res_unit = dom_phys["hbm_GBs_measured"], HBM_PEAK_GBS, "GB/s"
    elif bound == "mfma":
        res_achieved, res_peak, res_unit = dom_phys["mfma_busy"] * MFMA_PEAK_TFLOPS[args.precision], MFMA_PEAK_TFLOPS[args.precision], \
            "TFLOP/s (matrix-pipe busy fraction x dense peak)"
    elif bound in ("valu_issue", "lds_issue"):
        res_achieved, res_peak, res_unit = dom_phys["bound_frac"], 1.0, \
            ("VALU" if bound == "valu_issue" else "LDS") + " issue cycles / (GPU-active cycles x SIMDs)"
    else:                                                                   # no counter pass: only the layer-wise figure is known
        res_achieved, res_peak, res_unit = layerwise, HBM_PEAK_GBS, "GB/s (layer-wise algorithmic bytes: no counter pass in this run)"
    roofline = {
        "bound": bound or "unmeasured (no counter pass)",
        "kernel": dom["name"], "kernel_instance": dom["kernel"],
        "achieved": res_achieved, "peak": res_peak, "unit": res_unit,
        "frac": (dom_phys["bound_frac"] if dom_phys else None),
        "frac_kind": FRAC_KIND.get(bound, "unmeasured"),
        # the contract's whole-path fractions AT THE REPORTED RATE (`value`'s images/s), next to the dominant kernel's occupancy:
        #   layer-wise algorithmic bytes per image (27 615 616 B at 448 x 448 fp16) x images/s / 8 TB/s  -- > 1 = fusion credit
        #   HBM bytes the counters saw per image x images/s / 8 TB/s                                               -- the physical figure
        #   layer MACs x 2 per image x images/s / dense MFMA peak of the dtype                                     -- useful matrix work
        "hbm_layerwise_frac_at_value": (alg_total / n_prof) * images_per_sec_gpu / 1e9 / HBM_PEAK_GBS,
        "hbm_measured_frac_at_value": (path_phys["hbm_bytes"] / n_prof * images_per_sec_gpu / 1e9 / HBM_PEAK_GBS) if path_phys else None,
        "mfma_useful_frac_at_value": 2 * sum(p["macs"] for p in prof) / n_prof * images_per_sec_gpu / 1e12 / MFMA_PEAK_TFLOPS[args.precision],
        "alg_bytes_per_image": alg_total / n_prof, "macs_per_image": sum(p["macs"] for p in prof) / n_prof,
        "frac_is": "OCCUPANCY of the binding resource, measured (rocprofv3 --pmc passes of this run): issue cycles used / bytes moved, whether or not the "
                   "layers require them -- see `useful` for the efficiency" if dom_phys else
                   "null: no counter pass in this run (the layer-wise credit is under frac_layerwise_credit, never here)",
        "useful": useful_fractions(dom, args.precision),
        "kernel_ms_in_pipeline": None,
        "frac_layerwise_credit": layerwise / HBM_PEAK_GBS,
        # the SAME dominant kernel in the exact shape the measurement contract words it ({bound, achieved, peak, unit, frac, traffic}: ALGORITHMIC bytes per
        # launch / the kernel's average duration / 8 TB/s) for a mechanical reader; `frac` above stays the physical occupancy of the binding resource
        "hbm_layerwise": {"bound": "hbm", "achieved": layerwise, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": layerwise / HBM_PEAK_GBS,
                          "traffic": traffic,
                          "note": "SURVEY.md 8d layer-wise algorithmic bytes / kernel duration: > 1 = bytes fusion never moves"},
        "effective_clock_GHz_in_counter_pass": clock_ghz,
        "kernel_ms_method": "HIP events around back-to-back repeats of the launch on the engine's stream (rf_profile); the pipeline's "
                            "rocprofv3 kernel trace averages ~5 % longer for the same kernel (cold caches between different kernels): "
                            "profiles/r03_bench_b8_448_fp16_kernel_trace_lanes1.txt",
        "traffic": traffic, "traffic_source": traffic_src,
        "traffic_GBs": (traffic / (dom["ms"] * 1e-3) / 1e9) if traffic else None,
        "hbm_frac_measured": (traffic / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "mfma_busy": (dom_phys or {}).get("mfma_busy"), "valu_active": (dom_phys or {}).get("valu_active"),
        "lds_active": (dom_phys or {}).get("lds_active"), "bound_frac": (dom_phys or {}).get("bound_frac"),
        "mfma_flops_frac": 2 * dom["macs"] / (dom["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[args.precision],
        "kernel_ms": dom["ms"], "kernel_alg_bytes": dom["alg_bytes"],
        "kernel_share_of_gpu_time": dom["ms"] / kernel_ms,
        "images_per_launch": n_prof,
        "whole_path": {
            "kernels_ms_per_launch_sequence": kernel_ms, "launches": len(prof), "kernels_ms_in_pipeline": None,
            "useful": useful_fractions(prof, args.precision),
            "alg_bytes_layerwise": alg_total, "frac_layerwise_credit": alg_total / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "end_to_end_frac_layerwise_credit": (alg_total / n_prof) * images_per_sec_gpu / 1e9 / HBM_PEAK_GBS,
            "hbm_bytes_per_image_measured": (path_phys["hbm_bytes"] / n_prof) if path_phys else None,
            "hbm_frac_measured_in_kernels": path_phys["hbm_frac_measured"] if path_phys else None,
            "hbm_frac_measured_at_pipeline_rate": (path_phys["hbm_bytes"] / n_prof * images_per_sec_gpu / 1e9 / HBM_PEAK_GBS) if path_phys else None,
            "mfma_busy": (path_phys or {}).get("mfma_busy"), "valu_active": (path_phys or {}).get("valu_active"),
            "lds_active": (path_phys or {}).get("lds_active"), "bound": (path_phys or {}).get("bound"),
            "mfma_flops_frac": 2 * sum(p["macs"] for p in prof) / (kernel_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[args.precision],
            "mfma_peak_TFLOPs": MFMA_PEAK_TFLOPS[args.precision], "mfma_peak_source": MFMA_PEAK_SOURCE[args.precision],
        },
        "single_batch_launch": {"images_per_launch": B, "all_kernels_ms": sum(p["ms"] for p in prof8),
                                "dominant_kernel": max(prof8, key=lambda p: p["ms"])["name"],
                                "dominant_kernel_ms": max(p["ms"] for p in prof8)},
        "elem_bytes": {"fp16": 2, "fp32": 4, "int8": 1}[args.precision],
    }
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_kernels.json"), "w") as f:
        json.dump({"per_launch_images": n_prof, "kernels": prof, "kernels_single_batch": prof8, "counters": counters}, f, indent=1)
    return {"split_ms_per_batch": {"pre": med["pre_ms"], "infer": med["infer_ms"], "post": med["post_ms"], "all": med["total_ms"]},
            "roofline": roofline}


def measure_pipeline_trace(args):
    """The launch sequence's kernels as they run IN the pipeline: `rocprofv3 --kernel-trace` over `bench.py --timed-only --lanes 1` of the same
    configuration (one super-batch at a time on the GPU, so a kernel's duration is not stretched by another lane's persistent grid), summarised
    per kernel instance.  rf_profile's HIP-event figure repeats ONE launch back to back (warm caches); throughput is made of these.
    Returns {"kernels": [{kernel, grid, calls, launches_per_sequence, avg_ms}], "sum_ms"} or None."""
    import shutil
    import sqlite3
    import tempfile
    if args.no_pipeline_trace or args.no_pmc or shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    tmp = None
    try:
        import pmc_summary
        tmp = tempfile.mkdtemp(prefix="rf_kt_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "-d", tmp, "-o", "kt", "--", sys.executable, os.path.abspath(__file__), "--timed-only", "--lanes", "1",
               "--no-pmc", "--no-cpu-baseline", "--no-extra-configs", "--regions", "1", "--min-seconds", "0.3", "--steps", "1", "--warmup", "96",
               "--ring-mb", "160", "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width),
               "--precision", args.precision, "--model", args.model, "--coalesce", str(args.coalesce)]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=240)
        dbs = [os.path.join(d, f) for d, _, fs in os.walk(tmp) for f in fs if f.endswith(".db")]
        if r.returncode != 0 or not dbs:
            return None
        rows = sqlite3.connect(dbs[0]).execute("select name, grid_x, count(*), avg(duration) from kernels group by name, grid_x").fetchall()
        groups = {}
        for name, grid, calls, avg in rows:
            if "rf::" not in name and not name.startswith("_ZN2rf"):
                continue
            # one entry per kernel SYMBOL (two instances may share an op descriptor: the c2 / c1 aggregation convs run different kernels);
            # the steady-state grid of a symbol = the one launched most often (warm-up launches use other grids)
            if name not in groups or calls > groups[name]["calls"]:
                groups[name] = {"kernel": pmc_summary.descriptor(name), "grid": int(grid), "calls": int(calls), "avg_ms": avg / 1e6}
        if not groups:
            return None
        # launches per sequence: most kernels run once per launch sequence, so the most common call count is "once" (one-off kernels -- the int8
        # engines' rounding self-check at rf_create, warm-up grids -- are dropped; the four plain 128-channel blocks show up as 4x)
        counts = sorted(g["calls"] for g in groups.values())
        base = max(set(counts), key=lambda c: (sum(1 for x in counts if abs(x - c) <= 0.02 * c), c))
        groups = {k: g for k, g in groups.items() if g["calls"] >= 0.5 * base}
        for g in groups.values():
            g["launches_per_sequence"] = max(1, round(g["calls"] / base))
        ks = sorted(groups.values(), key=lambda g: -g["avg_ms"] * g["launches_per_sequence"])
        return {"kernels": ks, "sum_ms": sum(g["avg_ms"] * g["launches_per_sequence"] for g in ks), "sequences_traced": base,
                "method": "rocprofv3 --kernel-trace of `bench.py --timed-only --lanes 1` (this run), average duration per kernel instance"}
    except Exception:  # noqa: BLE001
        return None
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)


def attach_pipeline_trace(roofline, trace):
    """kernel_ms_in_pipeline beside the HIP-event kernel_ms, for the dominant kernel and for the launch sequence."""
    if not trace:
        roofline["kernel_ms_in_pipeline_note"] = "no kernel-trace pass in this run (rocprofv3 unavailable / --no-pipeline-trace / --no-pmc)"
        return
    mine = [g for g in trace["kernels"] if g["kernel"] == roofline["kernel_instance"]]
    if mine:
        ms = max(g["avg_ms"] for g in mine)
        roofline["kernel_ms_in_pipeline"] = ms
        roofline["useful_in_pipeline"] = {k: (v * roofline["kernel_ms"] / ms if isinstance(v, float) and k.endswith(("_frac", "_GBs", "_TFLOPs")) else v)
                                          for k, v in roofline["useful"].items()}
    roofline["kernel_ms_in_pipeline_method"] = trace["method"]
    roofline["whole_path"]["kernels_ms_in_pipeline"] = trace["sum_ms"]
    roofline["whole_path"]["images_per_sec_one_lane_from_trace"] = roofline["images_per_launch"] / (trace["sum_ms"] * 1e-3)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_pipeline_trace.json"), "w") as f:
        json.dump(trace, f, indent=1)


# what the stems' tile geometry recomputes: conv4 tile 7 x 8 <- conv2 region 15 x 17 <- conv0 region 17 x 19 <- 35 x 39 input pixels (kernels.hip Stem2Cfg);
# int8 stem: conv2 tile 8 x 32 <- conv0 region 10 x 34 <- 21 x 69 input pixels (ST_*)
HALO_REDUNDANCY = {"stem2": {"conv0_pixels_computed_per_pixel_needed": (17 * 19) / (14 * 16.0), "conv2_pixels_computed_per_pixel_needed": (15 * 17) / (14 * 16.0),
                             "input_pixels_staged_per_pixel_covered": (35 * 39) / (28 * 32.0)},
                   "stem": {"conv0_pixels_computed_per_pixel_needed": (10 * 34) / (8 * 32.0), "input_pixels_staged_per_pixel_covered": (21 * 69) / (16 * 64.0)}}


def attach_instruction_mix(roofline, precision):
    """`roofline.frac` of a VALU-bound kernel is an OCCUPANCY of the issue port; this says what is issued.  Measured per-wave dynamic instruction counts of
    the dominant kernel (rocprofv3 --pmc SQ_INSTS_* / SQ_WAVES over one eager 256-image launch sequence: tools/pmc_insts.py, tools/gpu/r6.sh insts; the newest
    committed summary under profiles/ is joined -- the counts are a property of the code object, not of the run), the MFMA instructions among them, and the
    share of the kernel's work that is halo recomputation by tile geometry.  VERDICT r5 next #3."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_instruction_mix_{precision}_product.json")))
    if not files:
        roofline["instruction_mix"] = None
        return
    mix = json.load(open(files[-1]))
    mine = [k for k in mix["kernels"] if k["kernel"] == roofline["kernel_instance"]]
    if not mine:
        roofline["instruction_mix"] = None
        return
    k = max(mine, key=lambda e: e.get("waves", 0))
    valu, mops = k.get("sq_insts_valu_per_wave"), k.get("sq_insts_valu_mfma_mops_f16_per_wave")
    out = {"per_wave": {n[len("sq_insts_"):-len("_per_wave")]: v for n, v in k.items() if n.endswith("_per_wave")}, "waves_per_launch": k.get("waves"),
           "source": os.path.relpath(files[-1], ROOT), "is": "wave-instructions issued per wave launched, loops and divergent branches included (measured, not static)"}
    if valu and mops is not None:
        mfma = mops / 32.0                  # one v_mfma_f32_16x16x32_f16 = 32 MOPS of 512 FLOPs
        out["mfma_instructions_per_wave"] = mfma
        out["valu_non_mfma_per_wave"] = valu - mfma
        out["valu_cycles_floor_ms"] = valu * k.get("waves", 0) * 4 / 1024.0 / 2.4e9 * 1e3      # 4 issue cycles per wave64 VALU instruction, 1024 SIMDs, 2.4 GHz
        out["valu_cycles_floor_is"] = "VALU instructions x 4 cycles / 1024 SIMDs / 2.4 GHz: the time the kernel's instruction count alone costs (compare kernel_ms)"
    if roofline["kernel_instance"] in HALO_REDUNDANCY:
        out["halo_recompute_by_tile_geometry"] = HALO_REDUNDANCY[roofline["kernel_instance"]]
    prev = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_instruction_mix_{precision}_stem2_round5.json")))
    if prev and roofline["kernel_instance"] == "stem2":
        old = [e for e in json.load(open(prev[-1]))["kernels"] if e["kernel"] == "stem2"]
        if old:
            out["round5_valu_per_wave"] = old[0].get("sq_insts_valu_per_wave")
            out["round5_is"] = "the same kernel with round 5's register staging of the patch (RF_STEM2_V2=7, probe build), same pass: the raw-row LDS-DMA staging is the difference"
    roofline["instruction_mix"] = out


def baseline_config(args) -> str:
    """Which BASELINE.json config this invocation corresponds to."""
    if args.global_batch:
        return ("BASELINE.json configs[4] as stated: 256 images per step sharded over the GPUs" if
                (args.model, args.precision, args.height, args.width, args.global_batch) == ("mnet25", "int8", 448, 448, 256)
                else "not a BASELINE.json config")
    key = (args.model, args.precision, args.height, args.width, args.batch)
    return {("mnet25", "fp16", 448, 448, 8): "BASELINE.json configs[1], the metric point",
            ("mnet-deconv-0517", "int8", 448, 448, 32): "BASELINE.json configs[2]",
            ("mnet25", "fp16", 896, 1280, 1): "BASELINE.json configs[3]",
            ("mnet25", "int8", 448, 448, 32): "BASELINE.json configs[4]: 256 images = 32 per GPU x 8 GPUs"}.get(key, "not a BASELINE.json config")


_CPU_WORKER = r"""
import os, sys, time, json
os.sched_setaffinity(0, {cpus!r})          # one physical core per thread, disjoint between workers (set before OpenMP starts)
sys.path.insert(0, {root!r})
import numpy as np, torch
torch.set_num_threads({threads})
from oracle.caffe_io import read_rfw
from oracle.pipeline import OracleDetector
from retinaface_amd.frames import synth_frames
orc = OracleDetector(read_rfw(os.path.join({root!r}, "assets", {model!r} + ".rfw")))
fr = synth_frames({h}, {w}, 2, config=1)
orc.detect(fr[0], {thr}, 0.4, net_hw=({h}, {w}))
print("ready", flush=True)
sys.stdin.readline()
n = f = 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < {seconds}:
    r = orc.detect(fr[n % 2], {thr}, 0.4, net_hw=({h}, {w}))
    n += 1; f += len(r.detections)
print(json.dumps(dict(n=n, f=f, dt=time.perf_counter() - t0)), flush=True)
"""


def cpu_all_cores(args, threads: int, ncpu: int, seconds: float):
    """The same oracle as `nproc` independent processes x `threads` torch threads, started together: what the host's cores give
    when the reference's single-process loop is simply run several times (SURVEY.md 8d "core count stated")."""
    threads = min(threads, 8)                       # beyond ~8 threads per process oneDNN loses on these small convolutions
    # one logical CPU per physical core (first hardware thread of every sibling set this process may run on), handed out in
    # disjoint blocks of `threads`: round 2 left placement to the scheduler and 32 x 4 threads gave 2.5x one process -- the
    # workers were migrating and sharing cores
    allowed = sorted(os.sched_getaffinity(0))
    cores, seen = [], set()
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            cores.append(c)
    nproc = max(1, min(32, len(cores) // threads))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    load_before = os.getloadavg()[0]
    procs = []
    for k in range(nproc):
        src = _CPU_WORKER.format(root=ROOT, threads=threads, model=args.model, h=args.height, w=args.width, thr=args.threshold,
                                 seconds=seconds, cpus=set(cores[k * threads:(k + 1) * threads]))
        procs.append(subprocess.Popen([sys.executable, "-c", src], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env))
    try:
        for p in procs:
            assert p.stdout.readline().strip() == "ready"
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in procs]
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:  # noqa: BLE001
                pass
            p.wait(timeout=60)
    dt = max(r["dt"] for r in res)
    return {"processes": nproc, "threads_per_process": threads, "cores": nproc * threads, "physical_cores_available": len(cores),
            "pinned": "each worker to its own block of physical cores (sched_setaffinity)", "loadavg_1min_before": load_before,
            "images_per_sec": sum(r["n"] for r in res) / dt, "value": sum(r["f"] for r in res) / dt, "unit": "faces/s",
            "per_process_images_per_sec": [round(r["n"] / r["dt"], 2) for r in res]}


def cpu_baseline(frames_np, args, det):
    """The CPU oracle (layer-by-layer unfused fp32 restatement of the reference's Caffe path on PyTorch-CPU/oneDNN +
    the literal decode/NMS) on the same frames, bounded to ~cpu-seconds; also cross-checks that the GPU path found the same
    faces.  `value` is the best single-process setting (the reference is one process); `all_cores` runs that setting in as many
    processes as the host has cores for."""
    import numpy as np
    import torch
    from oracle.caffe_io import read_rfw
    from oracle.pipeline import OracleDetector
    ncpu = os.cpu_count() or 1
    orc = OracleDetector(read_rfw(os.path.join(ROOT, "assets", args.model + ".rfw")))
    H, W = args.height, args.width
    # oneDNN on these tiny convolutions gets SLOWER with very many threads; pick the best thread count from a short
    # calibration (one frame each) and report the one actually used as `cores`.
    best = None
    for th in sorted({min(ncpu, t) for t in (4, 8, 16, 32)}):
        torch.set_num_threads(th)
        orc.detect(frames_np[0], args.threshold, 0.4, net_hw=(H, W))     # warm-up (oneDNN primitive creation)
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            orc.detect(frames_np[0], args.threshold, 0.4, net_hw=(H, W))
            ts.append(time.perf_counter() - t)
        t = sorted(ts)[1]
        if best is None or t < best[1]:
            best = (th, t)
    cores = best[0]
    torch.set_num_threads(cores)
    n_img = n_faces = 0
    ref_idx = [None] * len(frames_np)
    t0 = time.perf_counter()
    while n_img < len(frames_np) or time.perf_counter() - t0 < args.cpu_seconds:
        i = n_img % len(frames_np)
        r = orc.detect(frames_np[i], args.threshold, 0.4, net_hw=(H, W))
        ref_idx[i] = [d.anchor_index for d in r.detections]
        n_img += 1
        n_faces += len(r.detections)
        if n_img >= len(frames_np) and time.perf_counter() - t0 > 4 * args.cpu_seconds:
            break
    dt = time.perf_counter() - t0
    gpu = det.detectBatchImages(frames_np, args.threshold)
    same = [[d.anchor_index for d in r] for r in gpu] == ref_idx
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:  # noqa: BLE001
        model = "unknown"
    out = {"value": n_faces / dt, "unit": "faces/s", "cores": cores, "kind": "port",
           "sample": f"{n_img} frames (cycling over the bench batch), {dt:.1f} s, PyTorch-CPU oneDNN fp32 unfused Caffe "
                     f"restatement + literal decode/NMS, torch threads = {cores} of {ncpu} logical CPUs (best of 4/8/16/32)",
           "images_per_sec": n_img / dt, "ms_per_frame": dt / n_img * 1e3, "cpu_model": model,
           "gpu_faces_identical_to_oracle": bool(same)}
    try:
        out["all_cores"] = cpu_all_cores(args, cores, ncpu, min(8.0, args.cpu_seconds))
    except Exception as e:  # noqa: BLE001
        out["all_cores"] = {"error": str(e)}
    return out


if __name__ == "__main__":
    sys.exit(main())
