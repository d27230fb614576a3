"""CPU probe (round 6): which depthwise/pointwise blocks would have to leave int8 for the NMS winners to stop flipping?  Fake-quantised replay
(tools/probes/int8_mix_sim.py) with the engine's 8-bit mids, then fp16 pointwise arithmetic for a prefix / suffix of the 12 quantised blocks.
usage: python tools/probes/int8_block_ablation.py <model> <frames>    -> profiles/r06_int8_mixed_precision_sim.txt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import int8_mix_sim as S
from oracle.caffe_io import read_rfw
from retinaface_amd.frames import synth_frames
torch.set_num_threads(8)
model = sys.argv[1]; nfr = int(sys.argv[2])
net = read_rfw(os.path.join(S.ROOT, "assets", model + ".rfw"))
sim = S.Sim(net, dict(net.int8_scales))
scales = sim.tensor_scales()
frames = synth_frames(448, 448, nfr, config=400, faces=[1, 3, 5])
refs = [S.detect(sim.forward(f, {"*": "f32"}, scales), (448, 448), with_cands=True) for f in frames]
def run(name, mode):
    res = [S.detect(sim.forward(f, mode, scales), (448, 448)) for f in frames]
    s = S.stats(res, refs)
    print(f"{model:18s} {name:28s} same {s['same_count']}/{s['frames']} worst {s['worst']:.4f} mean {s['mean']:.4f} <0.97: {s['below97']}/{s['faces']} agree {s['agree']:.3f} anchor_worst {s['anchor_worst']:.4f} p01 {s['anchor_p01']:.4f} dscore {s['dscore']:.4f}", flush=True)
base = {f"relu{2*i+1}": "u8" for i in range(1, 13)}
run("engine (u8 mids)", base)
run("all mids f16", {f"relu{2*i+1}": "f16" for i in range(1, 13)})
for upto in (1, 2, 3, 4, 5, 6, 8, 10):
    m = dict(base); m.update({f"relu{2*i+1}": "f16" for i in range(1, upto + 1)})
    run(f"mids f16 blocks 1..{upto}", m)
for frm in (11, 9, 6, 5):
    m = dict(base); m.update({f"relu{2*i+1}": "f16" for i in range(frm, 13)})
    run(f"mids f16 blocks {frm}..12", m)
