"""Host-side logic of the product, no GPU: the C-ABI library loads and exports every declared symbol, the C++
model loader / graph compiler agree with the oracle's readers and an independent numpy BN fold, error paths,
packing index math, frame synthesis, batch sharding over a 2-process gloo group."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ASSETS, REFERENCE, ROOT, STEMS, has_reference, needs_reference
from retinaface_amd import _lib


def _header_functions():
    text = open(os.path.join(ROOT, "include", "retinaface_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib):
    names = _header_functions()
    assert len(names) >= 18
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    for n in names:
        assert getattr(built_lib, n) is not None
    assert built_lib.rf_abi_version() == 2
    assert C.sizeof(_lib.rf_face) == 60          # FaceDetectInfo, RetinaFace.h:37-42


def test_class_header_keeps_reference_signatures():
    h = open(os.path.join(ROOT, "include", "RetinaFace.h")).read()
    for sig in ('RetinaFace(string &model, string network = "net3", float nms = 0.4);',
                "void detectBatchImages(vector<cv::Mat> imgs, float threshold = 0.5);",
                "void detect(const Mat &img, float threshold = 0.5, float scales = 1.0);"):
        assert sig in h, sig


def _folded(lib, model_dir, stem, op):
    dims = (C.c_int * 4)()
    st = lib.rf_plan_folded(model_dir.encode(), stem.encode(), op.encode(), None, 0, None, 0, dims)
    assert st == 0, lib.rf_last_error(None)
    n = dims[0] * dims[1] * dims[2] * dims[3]
    w = np.empty(n, np.float32)
    b = np.empty(dims[0], np.float32)
    st = lib.rf_plan_folded(model_dir.encode(), stem.encode(), op.encode(), w.ctypes.data_as(C.POINTER(C.c_float)), n,
                            b.ctypes.data_as(C.POINTER(C.c_float)), dims[0], dims)
    assert st == 0
    return w.reshape(dims[0], dims[1], dims[2], dims[3]), b


def _np_fold(net, conv, bn=None):
    l = net.layer(conv)
    W = l.blobs[0].astype(np.float64)
    b = l.blobs[1].reshape(-1).astype(np.float64) if l.bias_term else np.zeros(W.shape[0])
    if bn:
        B, S = net.layer(bn), net.layer(bn + "_scale")
        sf = B.blobs[2].reshape(-1)[0]
        inv = np.float32(0) if sf == 0 else np.float32(1) / np.float32(sf)
        mean = (B.blobs[0].reshape(-1) * inv).astype(np.float64)
        var = (B.blobs[1].reshape(-1) * inv).astype(np.float64)
        k = S.blobs[0].reshape(-1).astype(np.float64) / np.sqrt(var + np.float64(np.float32(B.eps)))
        W = W * k[:, None, None, None]
        b = (b - mean) * k + S.blobs[1].reshape(-1).astype(np.float64)
    return W.transpose(0, 2, 3, 1).astype(np.float32), b.astype(np.float32)


@pytest.mark.parametrize("stem", STEMS)
def test_graph_compiler_fold_matches_numpy(stem, built_lib, nets):
    net = nets[stem]
    w, b = _folded(built_lib, ASSETS, stem, "conv0")
    rw, rb = _np_fold(net, "mobilenet0_conv0_fwd", "mobilenet0_batchnorm0_fwd")
    assert w.shape == (8, 3, 3, 3) and np.array_equal(w, rw) and np.array_equal(b, rb)
    for i in (0, 5, 12):
        for kind, idx in (("dw", 2 * i + 1), ("pw", 2 * i + 2)):
            w, b = _folded(built_lib, ASSETS, stem, f"{kind}{i}")
            rw, rb = _np_fold(net, f"mobilenet0_conv{idx}_fwd", f"mobilenet0_batchnorm{idx}_fwd")
            assert np.array_equal(w, rw) and np.array_equal(b, rb), (kind, i)
    w, b = _folded(built_lib, ASSETS, stem, "lateral0")
    rw, rb = _np_fold(net, "rf_c3_lateral", "rf_c3_lateral_bn")
    assert w.shape == (64, 1, 1, 256) and np.array_equal(w, rw) and np.array_equal(b, rb)
    w, b = _folded(built_lib, ASSETS, stem, "aggr1")
    rw, rb = _np_fold(net, "rf_c1_aggr", "rf_c1_aggr_bn")
    assert np.array_equal(w, rw) and np.array_equal(b, rb)
    # merged siblings: 64->48 = det_conv1 (32) || context_conv1 (16); heads 64->32 = cls 4 || bbox 8 || landmark 20
    w, b = _folded(built_lib, ASSETS, stem, "ssh1.a")
    w1, b1 = _np_fold(net, "rf_c2_det_conv1", "rf_c2_det_conv1_bn")
    w2, b2 = _np_fold(net, "rf_c2_det_context_conv1", "rf_c2_det_context_conv1_bn")
    assert w.shape == (48, 3, 3, 64) and np.array_equal(w, np.concatenate([w1, w2])) and np.array_equal(b, np.concatenate([b1, b2]))
    w, b = _folded(built_lib, ASSETS, stem, "ssh2.head")
    parts = [_np_fold(net, f"face_rpn_{k}_stride8") for k in ("cls_score", "bbox_pred", "landmark_pred")]
    assert w.shape == (32, 1, 1, 64) and np.array_equal(w, np.concatenate([p[0] for p in parts]))
    assert np.array_equal(b, np.concatenate([p[1] for p in parts]))


@needs_reference
@pytest.mark.parametrize("stem", STEMS)
def test_cxx_loader_reads_the_reference_files(stem, built_lib, tmp_path):
    """prototxt + caffemodel + int8 table parsed by the C++ loader, re-packed and given the calibrated weights (rf_attach_calibration)
    must equal the committed .rfw byte for byte (the Python writer produced that one), and the plan compiled straight from the
    Caffe files must equal the plan compiled from the .rfw.  The tables: this repo's calibration of either model (tools/calibrate_int8.py,
    assets/<stem>[.cal].table.int8 + .qweights.int8); the TensorRT table the reference ships for 0517 is packed too and must give the
    same container minus the calibration."""
    m = os.path.join(REFERENCE, "model")
    out = str(tmp_path / (stem + ".rfw"))
    base = os.path.join(ASSETS, stem + (".cal" if stem == "mnet-deconv-0517" else ""))
    st = built_lib.rf_convert_model(os.path.join(m, stem + ".prototxt").encode(), os.path.join(m, stem + ".caffemodel").encode(),
                                    (base + ".table.int8").encode(), out.encode())
    assert st == 0, built_lib.rf_last_error(None)
    os.makedirs(tmp_path / "packed")
    out2 = str(tmp_path / "packed" / (stem + ".rfw"))
    st = built_lib.rf_attach_calibration(str(tmp_path).encode(), stem.encode(), None, (base + ".qweights.int8").encode(), out2.encode())
    assert st == 0, built_lib.rf_last_error(None)
    assert open(out2, "rb").read() == open(os.path.join(ASSETS, stem + ".rfw"), "rb").read()
    if stem == "mnet-deconv-0517":
        # the reference's own TensorRT cache still packs (per tensor, no calibrated weights): a new table drops the weights chosen under the old one
        trt = os.path.join(m, "mnet-deconv-0517.table.int8")
        assert open(trt, "rb").read() == open(os.path.join(ASSETS, "mnet-deconv-0517.table.int8"), "rb").read()
        out3 = str(tmp_path / "trt.rfw")
        assert built_lib.rf_attach_calibration(ASSETS.encode(), stem.encode(), trt.encode(), None, out3.encode()) == 0
        from oracle.caffe_io import read_int8_table, read_rfw
        back = read_rfw(out3)
        assert back.int8_qweights == {} and back.int8_scales == read_int8_table(trt)
    a = _folded(built_lib, m, stem, "ssh0.b")
    b = _folded(built_lib, ASSETS, stem, "ssh0.b")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def _create(lib, model_dir, network=b"net3", **kw):
    o = _lib.rf_options()
    o.struct_size = C.sizeof(_lib.rf_options)
    for k, v in kw.items():
        setattr(o, k, v)
    h = C.c_void_p()
    st = lib.rf_create(model_dir, network, 0.4, C.byref(o), C.byref(h))
    msg = lib.rf_last_error(None).decode()
    if st == 0:
        lib.rf_destroy(h)
    return st, msg


@pytest.mark.parametrize("stem,block", [("mnet25", 4), ("mnet-deconv-0517", 7), ("mnet25", 1)])
def test_depthwise_equalisation_is_exact_and_cuts_the_fp16_tap_error(built_lib, stem, block):
    """weights.h equalize_depthwise (fp16 engine): channel c of a depthwise stage works at 1 / t[c] scale, column c of the pointwise
    matrix is multiplied by t[c].  (1) It is a pure re-scaling: taps and bias / t, column * t, 1 <= t < 2, biases of the pointwise
    stage untouched, and relu(dw(x)) -> pw gives the same result as before.  (2) It does what it is for: the fp16 rounding error of
    the nine taps (in units of the original weights) drops by more than 2.5x in rms."""
    dw, db = _folded(built_lib, ASSETS, stem, f"dw{block}")
    pw, pb = _folded(built_lib, ASSETS, stem, f"pw{block}")
    dwe, dbe = _folded(built_lib, ASSETS, stem, f"dw{block}.eq")
    pwe, pbe = _folded(built_lib, ASSETS, stem, f"pw{block}.eq")
    c = dw.shape[0]
    d9, e9 = dw.reshape(c, 9).astype(np.float64), dwe.reshape(c, 9).astype(np.float64)
    live = np.abs(d9).max(1) > 1e-20
    k = np.abs(d9).argmax(1)
    t = np.where(live, d9[np.arange(c), k] / np.where(live, e9[np.arange(c), k], 1.0), 1.0)
    assert (t >= 1.0 - 1e-6).all() and (t < 2.0).all()
    assert np.allclose(e9 * t[:, None], d9, rtol=2e-7, atol=1e-30)
    assert np.allclose(dbe.astype(np.float64) * t, db, rtol=2e-7, atol=1e-30)
    assert np.allclose(pwe.reshape(-1, c), pw.reshape(-1, c).astype(np.float64) * t[None, :], rtol=2e-7) and np.array_equal(pbe, pb)
    rng = np.random.default_rng(block)
    x = rng.normal(size=(c, 9))                                     # one 3x3 window per channel
    y0 = np.maximum((d9 * x).sum(1) + db, 0.0) @ pw.reshape(-1, c).astype(np.float64).T
    y1 = np.maximum((e9 * x).sum(1) + dbe, 0.0) @ pwe.reshape(-1, c).astype(np.float64).T
    assert np.allclose(y0, y1, rtol=1e-5, atol=1e-6)
    err = lambda w, scale: ((w.astype(np.float16).astype(np.float64) - w) * scale[:, None])[live]
    before, after = err(d9.astype(np.float32), np.ones(c)), err(e9.astype(np.float32), t)
    assert np.sqrt((after ** 2).mean()) < np.sqrt((before ** 2).mean()) / 2.5, (np.sqrt((before ** 2).mean()), np.sqrt((after ** 2).mean()))


def test_error_paths_without_a_gpu(built_lib, tmp_path):
    st, msg = _create(built_lib, str(tmp_path).encode())
    assert st == _lib.RF_ERR_IO and "mnet-deconv-0517" in msg
    # presets the reference constructs without anchors are accepted (they detect nothing, as there): only the GPU is missing here
    st, msg = _create(built_lib, ASSETS.encode(), network=b"net5")
    assert st in (_lib.RF_ERR_HIP, 0), msg
    # "net3a" decodes 4 anchors per cell, the shipped models carry 2: refused on the host (the reference would read out of bounds)
    st, msg = _create(built_lib, ASSETS.encode(), network=b"net3a")
    assert st == _lib.RF_ERR_MODEL and "anchors per cell" in msg
    st, msg = _create(built_lib, ASSETS.encode(), precision=7)
    assert st == _lib.RF_ERR_INVALID_ARG
    bad = tmp_path / "mnet-deconv-0517.rfw"
    bad.write_bytes(b"RFW1" + b"\x01\0\0\0" + b"\xff" * 64)
    st, msg = _create(built_lib, str(tmp_path).encode())
    assert st == _lib.RF_ERR_MODEL
    o = _lib.rf_options()
    o.struct_size = 4
    h = C.c_void_p()
    assert built_lib.rf_create(ASSETS.encode(), b"net3", 0.4, C.byref(o), C.byref(h)) == _lib.RF_ERR_INVALID_ARG
    assert built_lib.rf_detect_batch(None, None, None, None, None, 0, 0.5, None, 0, None) == _lib.RF_ERR_INVALID_ARG


def test_truncated_graph_is_rejected(built_lib, nets, tmp_path):
    """A model whose topology is not mnet0.25+FPN+SSH must fail loudly at load (RF_ERR_MODEL), not mis-run."""
    from oracle.caffe_io import write_rfw
    import copy
    net = copy.deepcopy(nets["mnet25"])
    net.layer("rf_c2_det_context_conv2").num_output = 8
    net.layer("rf_c2_det_context_conv2").blobs[0] = net.layer("rf_c2_det_context_conv2").blobs[0][:8]
    write_rfw(net, str(tmp_path / "mnet25.rfw"))
    st, msg = _create(built_lib, str(tmp_path).encode(), model_stem=b"mnet25")
    assert st == _lib.RF_ERR_MODEL and "context_conv2" in msg
    net = copy.deepcopy(nets["mnet25"])
    net.layer("rf_c3_upsampling").blobs[0][3, 0, 1, 1] += 0.01       # no longer the fixed bilinear kernel
    write_rfw(net, str(tmp_path / "mnet25.rfw"))
    st, msg = _create(built_lib, str(tmp_path).encode(), model_stem=b"mnet25")
    assert st == _lib.RF_ERR_MODEL and "bilinear" in msg


def test_no_gpu_means_loud_failure(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    st, msg = _create(built_lib, ASSETS.encode())
    assert st == _lib.RF_ERR_HIP and "HIP device" in msg
    import retinaface_amd
    with pytest.raises(retinaface_amd.RFError):
        retinaface_amd.RetinaFace(ASSETS)


def test_knob_table_validates_and_separates_probe_from_product(tmp_path):
    """csrc/knobs.cpp (round 5): ONE table for every RF_* knob.  Probe knobs select measured-and-rejected kernel variants and exist in the probe
    build only: the product returns the default and says so on stderr; a value the dispatch code has no case for is reported and replaced by the
    default in either build (round 4: ~30 function-local getenv()s, unknown values silently fell through to some variant); semantic knobs work
    everywhere and are re-read on every query."""
    src = os.path.join(ROOT, "tests", "csrc", "test_knobs.cpp")
    exes = {}
    for tag, flags in (("product", []), ("probe", ["-DRF_PROBES"])):
        exes[tag] = str(tmp_path / f"test_knobs_{tag}")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread"] + flags + ["-o", exes[tag], src])

    def run(tag, **env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("RF_")}
        e.update(env)
        r = subprocess.run([exes[tag]], env=e, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        return dict(l.split() for l in r.stdout.splitlines()), r.stderr

    d, err = run("product")
    assert d["probes"] == "0" and d["RF_STEM2_V2"] == "15" and d["RF_CONV3WS"] == "1" and d["RF_TILE128"] == "-1" and d["RF_WIDE_I8"] == "3" and err == ""
    assert d["RF_FORCE_SCATTER"] == "0" and d["RF_PREBUILD_LANES"] == "0" and d["RF_BLEND_FP32"] == "0" and d["min_rounds"] == "1.00"
    # a probe knob in the product build: ignored, loudly; a semantic knob: honoured
    d, err = run("product", RF_STEM2_V2="5", RF_FORCE_SCATTER="1", RF_PREBUILD_LANES="1", RF_PERSIST_MIN_ROUNDS="3")
    assert d["RF_STEM2_V2"] == "15" and "RF_STEM2_V2=5 ignored" in err and "libretinaface_amd_probe.so" in err
    assert d["RF_FORCE_SCATTER"] == "1" and d["RF_PREBUILD_LANES"] == "1" and d["min_rounds"] == "1.00"
    # the probe build takes the values the dispatch code knows ...
    d, err = run("probe", RF_STEM2_V2="5", RF_CONV3WS="132", RF_TILE128="2", RF_WIDE_I8="6", RF_PERSIST_MIN_ROUNDS="3")
    assert d["probes"] == "1" and d["RF_STEM2_V2"] == "5" and d["RF_CONV3WS"] == "132" and d["RF_TILE128"] == "2" and d["RF_WIDE_I8"] == "6" and err == ""
    assert d["min_rounds"] == "3.00"
    # ... and refuses the ones it does not (round 4: RF_STEM2_V2=4 silently selected round 3's layout, RF_CONV3WS=5 the int8 kernel)
    d, err = run("probe", RF_STEM2_V2="4", RF_CONV3WS="5", RF_TILE128="banana")
    assert d["RF_STEM2_V2"] == "15" and d["RF_CONV3WS"] == "1" and d["RF_TILE128"] == "-1"
    assert err.count("is not a value this knob knows") == 3


def test_pack_index_math_host_emulation(tmp_path):
    exe = str(tmp_path / "test_pack")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "csrc", "test_pack.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout


def test_staging_copier_and_shard_rule_host_unit(tmp_path):
    exe = str(tmp_path / "test_copier")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "csrc", "test_copier.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout


def test_bench_self_launches_n_ranks_and_gathers_results():
    """`bench.py --gpus 2` with no launcher around it must start 2 ranks itself (round 1 ignored N), run the sharded loop with
    the result all_gather inside the timed region and report n_gpus == 2.  --dry swaps the HIP engine for a stub and RCCL for
    gloo; everything else (self-launch, rendezvous on 127.0.0.1, step agreement, gather, MAX over ranks, JSON) is the real path."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "100", "--warmup", "10"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] >= 100 and j["scaling"] == "weak" and j["metric"] == "faces/sec"
    g = j["result_gather"]
    assert g["records_gathered"] == g["expected"] == 2 * 8 * j["steps"] and g["gathers_in_timed_region"] >= 1
    # who took part, as seen through a collective of the gather's backend (on the GPUs: RCCL, with `rccl_ranks` == N and its version)
    assert g["backend"] == "gloo" and g["ranks_in_communicator"] == 2 and sorted(r for r, _ in g["rank_device"]) == [0, 1]
    # the weak-scaling invocation is followed by a strong-scaling leg of BASELINE.json configs[4] as stated (256 int8 images per step over N)
    s4 = j["configs4_strong"]
    assert s4["scaling"] == "strong" and s4["global_batch"] == 256 and s4["images_per_rank"] == 128 and s4["dtype"] == "i8"
    assert s4["result_gather"]["records_gathered"] == s4["result_gather"]["expected"] == 256 * s4["steps"]
    # ... and by the LIBRARY leg on rank 0 (one handle over devices [0 .. N-1], 256 images per call, frames on GPU 0: multi.cpp's split and
    # the peer scatter), while the other ranks wait on the rendezvous store; --dry reports the plan only
    lib = j["library_multi_device"]
    assert lib["devices"] == [0, 1] and lib["global_batch"] == 256 and lib["images_per_engine"] == 128
    # round 6: the N > 1 line diagnoses itself.  (a) every way the batch split can be carried has a slot: the library's peer
    # copies per slice / per frame and direct H2D (null until frames really travel), RCCL send / recv between the ranks in batch_split_ab
    assert set(lib["split_ab"]) >= {"peer_copy_per_slice", "peer_copy_per_frame", "host_frames_direct_h2d", "rccl_send_recv", "measured_over_xgmi"}
    assert lib["split_ab"]["peer_copy_per_slice"] is None and lib["split_ab"]["measured_over_xgmi"] is False and lib["leg_budget_seconds"] <= 20
    ab = j["batch_split_ab"]
    assert set(ab["legs"]) == {"resident", "rccl_send_recv", "direct_h2d"} and ab["global_batch"] == 256 and ab["same_detections_every_leg"]
    assert all(leg["ms_per_step"] > 0 and leg["steps_timed"] >= 3 for leg in ab["legs"].values())
    assert ab["rehearsal_not_xgmi"] is True and ab["winner"] is None          # gloo / dry: the path is rehearsed, no winner is declared
    # (b) every rank's own rate (a straggler is visible), and the scaling efficiency against rank 0 running alone in the same run
    pr = g["per_rank"]
    assert len(pr["images_per_sec"]) == 2 and pr["min_images_per_sec"] <= pr["max_images_per_sec"] and 0 < pr["min_over_max"] <= 1
    assert pr["slowest_rank"] in (0, 1) and pr["rank0_alone"]["images_per_sec"] > 0
    se = j["scaling_efficiency"]
    assert se["n_gpus"] == 2 and abs(se["value"] - j["images_per_sec"] / (2 * pr["rank0_alone"]["images_per_sec"])) < 1e-9 and se["on_distinct_devices"] is False
    # under an external launcher the rank count must agree with --gpus
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], capture_output=True, text=True,
                         env=env2, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in (bad.stderr + bad.stdout)


@pytest.mark.parametrize("gpus,gbatch", [(8, 256), (3, 10)])
def test_bench_strong_mode_shards_one_global_batch(gpus, gbatch):
    """BASELINE.json configs[4] as stated: ONE batch of 256 images per step split over 8 GPUs (`--global-batch 256 --gpus 8`),
    "scaling": "strong".  Rank r takes shard_range(G, r, N) images of every step (ragged when N does not divide G: 10 over 3 =
    4 + 4 + 2); every image's record must come back through the gather exactly once.  --dry: stub engine, gloo."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry", "--global-batch", str(gbatch),
                          "--steps", "12", "--warmup", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == gpus and j["scaling"] == "strong" and j["config"]["global_batch"] == gbatch
    assert j["result_gather"]["records_gathered"] == j["result_gather"]["expected"] == gbatch * j["steps"]
    assert abs(j["images_per_sec"] * j["timed_seconds"] - gbatch * j["steps"]) < 1e-3 * gbatch * j["steps"]
    assert j["regions"]["n"] == 3 and j["regions"]["min"] <= j["value"] <= j["regions"]["max"]


def test_rfw_reader_rejects_corrupt_files(built_lib, tmp_path):
    """Dims / counts inside a .rfw are checked against the bytes that are left before anything is sized by them."""
    import struct
    good = open(os.path.join(ASSETS, "mnet25.rfw"), "rb").read()
    h = C.c_void_p()

    def create(blob):
        d = tmp_path / f"m{len(os.listdir(tmp_path))}"
        d.mkdir()
        (d / "mnet25.rfw").write_bytes(blob)
        o = _lib.rf_options()
        o.struct_size = C.sizeof(_lib.rf_options)
        o.model_stem = b"mnet25"
        return built_lib.rf_create(str(d).encode(), b"net3", 0.4, C.byref(o), C.byref(h))

    assert create(good[:len(good) // 2]) == _lib.RF_ERR_MODEL                 # truncated
    pos = good.index(struct.pack("<IIII", 8, 3, 3, 3))                        # conv0 weight dims (O, I, kh, kw)
    for dims in ((65536, 65536, 65536, 65536), (0, 3, 3, 3), (8, 3, 3, 1 << 30)):
        assert create(good[:pos] + struct.pack("<IIII", *dims) + good[pos + 16:]) == _lib.RF_ERR_MODEL, dims
    nl_pos = 4 + 4 + 4 + len(b"") + 0                                         # layer count follows name / input name / shape
    huge = bytearray(good)
    # the layer count is the first u32 after the header strings and the 4 input dims: find it by parsing
    p = 8
    for _ in range(2):
        (n,) = struct.unpack_from("<I", good, p)
        p += 4 + n
    p += 16
    struct.pack_into("<I", huge, p, 0x7fffffff)
    assert create(bytes(huge)) == _lib.RF_ERR_MODEL
    # without a GPU a well-formed file still gets as far as "no HIP device"
    assert create(good) in (_lib.RF_ERR_HIP, 0)
    if h.value:
        built_lib.rf_destroy(h)


def test_options_struct_of_abi_1_is_still_accepted(built_lib):
    """A caller compiled against ABI 1 passes rf_options up to `coalesce`; the later fields default."""
    o = _lib.rf_options()
    o.struct_size = _lib.rf_options.copy_threads.offset
    o.model_stem = b"mnet25"
    h = C.c_void_p()
    st = built_lib.rf_create(ASSETS.encode(), b"net3", 0.4, C.byref(o), C.byref(h))
    assert st in (0, _lib.RF_ERR_HIP), built_lib.rf_last_error(None)           # parsed; fails only for want of a GPU
    if h.value:
        built_lib.rf_destroy(h)
    o.struct_size = 12
    assert built_lib.rf_create(ASSETS.encode(), b"net3", 0.4, C.byref(o), C.byref(h)) == _lib.RF_ERR_INVALID_ARG


def test_synthetic_frames_are_seeded_and_net_sized():
    from retinaface_amd.frames import FACE_BOXES, load_base_frame, synth_frames
    base = load_base_frame()
    assert base.shape == (886, 1280, 3) and base.dtype == np.uint8 and len(FACE_BOXES) == 6
    a = synth_frames(448, 448, 3, config=1, base=base)
    b = synth_frames(448, 448, 3, config=1, base=base)
    c = synth_frames(448, 448, 1, config=2, base=base)
    assert all(x.shape == (448, 448, 3) and x.flags["C_CONTIGUOUS"] for x in a)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and not np.array_equal(a[0], c[0])
    big = synth_frames(896, 1280, 1, config=3, base=base)[0]
    assert big.shape == (896, 1280, 3)


def test_shard_ranges_cover_the_batch():
    from retinaface_amd.shard import shard_range
    for n in (0, 1, 7, 8, 9, 256, 257):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo <= -(-n // world) if n else hi == lo
                got += list(range(lo, hi))
            assert got == list(range(n))
    assert shard_range(256, 3, 8) == (96, 128)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    from retinaface_amd.shard import gather_records, pack_records, shard_range, unpack_records
    from oracle.retinaface_post import Detection
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, cap = 5, 4
    lo, hi = shard_range(n, rank, world)
    f = np.float32
    dets = [[Detection(f(0.5 + 0.01 * img + 0.001 * k), (f(img), f(k), f(img + 10), f(k + 10)), [f(img)] * 5, [f(k)] * 5,
                       1000 * img + k) for k in range(img % 3 + 1)] for img in range(lo, hi)]
    full = gather_records(pack_records(dets, cap), n)
    res = unpack_records(full, cap)
    ok = len(res) == n
    for img in range(n):
        ok &= len(res[img]) == img % 3 + 1
        for k, (row, idx) in enumerate(res[img]):
            ok &= idx == 1000 * img + k and abs(row[0] - (0.5 + 0.01 * img + 0.001 * k)) < 1e-6 and row[1] == img
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_result_gather_over_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_int8_calibration_math():
    """tools/calibrate_int8.py: the entropy threshold clips far outliers but keeps a clean Gaussian's tail, the host-side
    depthwise / bilinear x2 helpers equal the torch ops of the layers they stand in for, and the shipped mnet25 table parses
    and covers every tensor name the shipped TensorRT 0517 table has a counterpart for in the int8 engine."""
    import importlib.util
    import torch
    import torch.nn.functional as F
    spec = importlib.util.spec_from_file_location("calibrate_int8", os.path.join(ROOT, "tools", "calibrate_int8.py"))
    cal = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cal)
    rng = np.random.default_rng(0)
    x = np.abs(rng.standard_normal(500_000)).astype(np.float32)
    amax = float(x.max())
    h, _ = np.histogram(x, bins=cal.NBINS, range=(0, amax))
    assert cal.entropy_threshold(h, amax / cal.NBINS) > 0.8 * amax          # nothing to clip
    x[:50] *= 12                                                             # 0.01 % outliers at 12x
    amax = float(x.max())
    h, _ = np.histogram(x, bins=cal.NBINS, range=(0, amax))
    t = cal.entropy_threshold(h, amax / cal.NBINS)
    assert 3.0 < t < 8.0 and t < 0.3 * amax, (t, amax)
    a = rng.standard_normal((9, 11, 8)).astype(np.float32)
    w = rng.standard_normal((8, 3, 3, 1)).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    for st in (1, 2):
        ref = F.relu(F.conv2d(torch.from_numpy(a.transpose(2, 0, 1))[None], torch.from_numpy(w.transpose(0, 3, 1, 2)),
                              torch.from_numpy(b), stride=st, padding=1, groups=8))[0].numpy().transpose(1, 2, 0)
        assert np.abs(cal.depthwise(a, w, b, st) - ref).max() < 1e-5
    k = np.outer([.25, .75, .75, .25], [.25, .75, .75, .25]).astype(np.float32)
    ref = F.conv_transpose2d(torch.from_numpy(a.transpose(2, 0, 1))[None], torch.from_numpy(np.tile(k, (8, 1, 1, 1))), stride=2,
                             padding=1, groups=8)[0].numpy().transpose(1, 2, 0)
    assert np.abs(cal.upsample2(a) - ref).max() < 1e-5
    from oracle.caffe_io import read_int8_table
    own = read_int8_table(os.path.join(ROOT, "assets", "mnet25.table.int8"))
    trt = read_int8_table(os.path.join(ROOT, "assets", "mnet-deconv-0517.table.int8"))
    per_tensor = {n: v for n, v in own.items() if "#" not in n}             # `name#c` lines: the per-channel extension
    assert set(per_tensor) <= set(trt) and len(per_tensor) >= 43
    ratio = np.array([per_tensor[n] / trt[n] for n in per_tensor])           # different weights, same architecture: same ballpark
    assert 0.3 < np.median(ratio) < 3.0
    chans = {n.split("#")[0] for n in own if "#" in n}
    assert chans == set(per_tensor) - {"data"}                               # every tensor also carries per-channel scales ...
    for n in chans:                                                          # ... none above its tensor's, none vanishing
        pc = np.array([v for k, v in own.items() if k.startswith(n + "#")])
        assert pc.max() <= per_tensor[n] * 1.0001 and pc.min() >= per_tensor[n] / 64.0 * 0.999, n


def test_calibration_batch_files_follow_the_reference_layout(tmp_path):
    """`.batch` files as INT8-Calibration-Tool writes them (calibrationtable.cpp:432-440: int[4] {N, C, H, W} + N*C*H*W f32, planar
    RGB, raw 0..255 -- preprocess is BGR2RGB + convertTo(CV_32FC3) only, CalibrationTableImpl.cpp:28-34): write / read round trip,
    byte layout, and rejection of malformed files."""
    import struct
    from retinaface_amd import calib_io
    from retinaface_amd.frames import synth_frames
    fr = synth_frames(320, 320, 3, config=5, faces=[1, 3])          # the reference tool's 320 x 320 (CalibrationTableImpl.cpp:5-7)
    p0, p1 = str(tmp_path / "batch_calibration0.batch"), str(tmp_path / "batch_calibration1.batch")
    calib_io.write_batch_file(p0, fr[:2])
    calib_io.write_batch_file(p1, fr[2:])
    raw = open(p0, "rb").read()
    assert struct.unpack("<4i", raw[:16]) == (2, 3, 320, 320) and len(raw) == 16 + 2 * 3 * 320 * 320 * 4
    plane = np.frombuffer(raw, "<f4", 320 * 320, 16).reshape(320, 320)
    assert np.array_equal(plane, fr[0][:, :, 2].astype(np.float32))          # first plane = R of image 0 (frames are BGR)
    back = calib_io.read_batch_dir(str(tmp_path))
    assert len(back) == 3 and all(np.array_equal(a, b) for a, b in zip(fr, back))
    (tmp_path / "bad.batch").write_bytes(raw[:1000])
    with pytest.raises(ValueError):
        calib_io.read_batch_file(str(tmp_path / "bad.batch"))
    (tmp_path / "bad.batch").write_bytes(struct.pack("<4i", 1, 4, 8, 8) + b"\0" * 1024)
    with pytest.raises(ValueError):
        calib_io.read_batch_file(str(tmp_path / "bad.batch"))
    # disjoint face subsets really are disjoint: frames of the two subsets share no pasted patch pixels beyond the noise floor
    a = synth_frames(448, 448, 2, config=9, faces=[0, 2, 4])
    b = synth_frames(448, 448, 2, config=9, faces=[1, 3, 5])
    assert not np.array_equal(a[0], b[0])


def test_plan_cache_round_trip_and_invalidation(built_lib, tmp_path):
    """The packed-weight image cache (<stem>.<precision>.rfplan, the analogue of the reference's serialized-engine cache,
    trtnetbase.cpp:205-243): built on first use, served afterwards, rebuilt when the model bytes change, when the file is damaged or
    belongs to another precision; an unwritable location just runs without a cache."""
    import shutil
    d = tmp_path / "model"
    d.mkdir()
    shutil.copy(os.path.join(ASSETS, "mnet25.rfw"), d / "mnet25.rfw")
    nbytes = C.c_size_t()

    def probe(prec, path=None):
        r = built_lib.rf_plan_cache_probe(str(d).encode(), b"mnet25", prec, path.encode() if path else None, C.byref(nbytes))
        assert r >= 0, built_lib.rf_last_error(None)
        return r

    for prec, name in ((1, "fp16"), (2, "int8"), (0, "fp32")):
        assert probe(prec) == 0                                        # built from the model, cache written
        f = d / f"mnet25.{name}.rfplan"
        assert f.exists() and f.stat().st_size > nbytes.value > 400_000
        first = nbytes.value
        assert probe(prec) == 1 and nbytes.value == first              # served from the cache: same image
    sizes = {n: (d / f"mnet25.{n}.rfplan").stat().st_size for n in ("fp32", "fp16", "int8")}
    assert sizes["fp32"] > sizes["fp16"] > sizes["int8"]
    # another precision's file under this precision's name is not accepted
    shutil.copy(d / "mnet25.int8.rfplan", d / "mnet25.fp16.rfplan")
    assert probe(1) == 0 and probe(1) == 1
    # a damaged cache is rebuilt, not trusted
    raw = (d / "mnet25.fp16.rfplan").read_bytes()
    (d / "mnet25.fp16.rfplan").write_bytes(raw[:len(raw) // 2])
    assert probe(1) == 0 and probe(1) == 1
    # the model changed (one weight byte): stale cache is ignored and replaced
    m = bytearray((d / "mnet25.rfw").read_bytes())
    m[len(m) // 2] ^= 0x01
    (d / "mnet25.rfw").write_bytes(bytes(m))
    assert probe(1) == 0 and probe(1) == 1
    # nowhere to write: still works, never a hit
    assert probe(1, "/proc/definitely/not/writable.rfplan") == 0 and probe(1, "/proc/definitely/not/writable.rfplan") == 0


def test_model_readers_under_sanitizers(tmp_path):
    """The model readers and the graph compiler (model.cpp, plan.cpp: RFW1 container, calibration table, calibrated weights, and -- where
    /root/reference is mounted -- the prototxt text reader and the caffemodel protobuf-wire reader) built with -fsanitize=address,undefined and fed
    1 250 truncated / bit-flipped / length-blown variants of the real files: each either loads or is rejected with rf::IoError / rf::ModelError;
    no out-of-bounds read, no overflow, no foreign exception type (round 6: found std::stoul / std::stol throwing std::invalid_argument out of the
    table and prototxt readers and std::out_of_range out of the graph walk).  The reference scrapes fixed columns and exit(0)s (trtnetbase.cpp:149-204)."""
    exe = str(tmp_path / "test_readers_sanitized")
    csrc = os.path.join(ROOT, "retinaface_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "csrc", "test_readers_sanitized.cpp"), os.path.join(csrc, "model.cpp"), os.path.join(csrc, "plan.cpp")])
    args = [exe, ASSETS] + ([os.path.join(REFERENCE, "model")] if has_reference() else [])
    out = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, TMPDIR=str(tmp_path), ASAN_OPTIONS="detect_leaks=1"), timeout=600)
    assert out.returncode == 0 and "none crashed" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_bench_physical_fractions_pick_the_binding_resource():
    """bench.py's `roofline` reports what the hardware did, not only the layer-wise credit: HBM bytes / time / 8 TB/s, VALU / LDS / MFMA busy
    cycles / (GPU-active cycles x SIMDs), `bound` = the largest.  Round 2's stem2 numbers (VALU-issue bound at 0.17 of the HBM peak) and an
    HBM-bound kernel, from counter values of the shape rocprofv3 returns (SQ_ACTIVE_INST_* in quad-cycles, GRBM_GUI_ACTIVE already / 8)."""
    sys.path.insert(0, ROOT)
    import bench
    gpu_cycles = 640_000.0                                   # ~0.267 ms at 2.4 GHz
    stem2 = dict(kernel="stem2", grid=1, launches_per_pass=1, hbm_bytes=360e6, valu_quad=0.80 * gpu_cycles * 1024 / 4,
                 lds_quad=0.21 * gpu_cycles * 1024 / 4, mfma_cycles=0.18 * gpu_cycles * 1024, gpu_cycles=gpu_cycles)
    r = bench.physical_fractions([stem2], 0.267, 1024)
    assert r["bound"] == "valu_issue" and abs(r["bound_frac"] - 0.80) < 1e-9 and abs(r["hbm_frac_measured"] - 360e6 / 0.267e-3 / 8e12) < 1e-9
    assert abs(r["mfma_busy"] - 0.18) < 1e-9 and abs(r["lds_active"] - 0.21) < 1e-9
    copy = dict(kernel="dwpw", grid=1, launches_per_pass=4, hbm_bytes=155e6, valu_quad=0.05 * 60_000 * 1024 / 4, lds_quad=0.0,
                mfma_cycles=0.08 * 60_000 * 1024, gpu_cycles=60_000.0)
    r = bench.physical_fractions([copy], 4 * 0.0257, 1024)
    assert r["bound"] == "hbm" and abs(r["hbm_frac_measured"] - 4 * 155e6 / (4 * 0.0257e-3) / 8e12) < 1e-9
    # without the SQ pass only the traffic is known
    r = bench.physical_fractions([dict(stem2, valu_quad=None, lds_quad=None, mfma_cycles=None)], 0.267, 1024)
    assert r["bound"] == "hbm" and "valu_active" not in r


def test_bench_roofline_carries_the_measured_instruction_mix():
    """`roofline.frac` of a VALU-bound kernel is an occupancy; `roofline.instruction_mix` (round 6) says what is issued: the per-wave
    dynamic counts of the dominant kernel from the committed rocprofv3 SQ_INSTS_* summary, the MFMA instructions among them, the time the VALU count alone
    costs, and the halo recomputation the tile geometry implies.  stem2: <= 520 VALU instructions per wave (round 5: 548)."""
    sys.path.insert(0, ROOT)
    import bench
    r = {"kernel_instance": "stem2", "kernel_ms": 0.228}
    bench.attach_instruction_mix(r, "fp16")
    m = r["instruction_mix"]
    assert m["source"].startswith("profiles/r06_instruction_mix_fp16") and m["per_wave"]["valu"] <= 520 < m["round5_valu_per_wave"]
    assert 20 <= m["mfma_instructions_per_wave"] <= 40 and abs(m["valu_non_mfma_per_wave"] + m["mfma_instructions_per_wave"] - m["per_wave"]["valu"]) < 1e-9
    assert 0.5 * r["kernel_ms"] < m["valu_cycles_floor_ms"] < r["kernel_ms"]
    g = m["halo_recompute_by_tile_geometry"]
    assert abs(g["conv0_pixels_computed_per_pixel_needed"] - 323 / 224) < 1e-12 and abs(g["input_pixels_staged_per_pixel_covered"] - 1365 / 896) < 1e-12
    r = {"kernel_instance": "no such kernel", "kernel_ms": 1.0}
    bench.attach_instruction_mix(r, "fp16")
    assert r["instruction_mix"] is None


def test_bench_int8_entries_carry_the_contract_numbers():
    """the int8 entries of the bench line (`configs[]` ids 2 and 4, and an int8 run's own line) carry the engine's measured distance to the fp32
    oracle, joined from the committed contract summary: same face count on every frame, anchor agreement, same-anchor IoU, per-face IoU."""
    sys.path.insert(0, ROOT)
    import bench
    for model in ("mnet25", "mnet-deconv-0517"):
        c = bench.int8_contract_summary(model)
        assert c["frames"] == c["same_count"] == 104 and c["faces"] >= 250 and c["source"].startswith("profiles/r06_int8_contract")
        assert c["anchor_iou_worst"] >= 0.96 and c["anchor_agreement"] >= 0.92 and c["iou_mean"] >= 0.985      # = INT8_BAR of tests/test_gpu_parity.py
    assert bench.int8_contract_summary("no such model") is None


def test_lds_model_known_cases():
    """tools/lds_model.py (the LDS-array cycle model behind dwpw2's and stem2's tile layouts): the guide's published cases and the
    three dwpw2 layouts whose totals the SQ_LDS_IDX_ACTIVE counter confirmed on the GPU (521 / 449 / 365 cycles per wave and tile)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_model as m
    # 64 lanes x 16 contiguous bytes: conflict-free in every instruction's lane groups
    assert m.array_cycles("read_b128", lambda l: l * 16) == 4 and m.array_cycles("write_b128", lambda l: l * 16) == 8
    # B fragments of 16 consecutive pixels: free at a pitch of 32 mod 64 bytes, 2-way at 80 bytes
    frag = lambda pitch: (lambda l: (l & 15) * pitch + (l >> 4) * 16)
    assert m.array_cycles("read_b128", frag(96)) == 4 and m.array_cycles("read_b128", frag(160)) == 4 and m.array_cycles("read_b128", frag(80)) == 8
    # the 8-byte epilogue write of 16 pixels x 4 channels: 4-way at a multiple of 32 bytes, 2-way at an odd multiple of 16
    epi = lambda pitch: (lambda l: (l & 15) * pitch + (l >> 4) * 8)
    assert m.array_cycles("write_b64", epi(96)) == 16 and m.array_cycles("write_b64", epi(32)) == 16
    assert m.array_cycles("write_b64", epi(80)) == 8 and m.array_cycles("write_b64", epi(144)) == 8 and m.array_cycles("write_b64", epi(16)) == 8
    # identical addresses broadcast
    assert m.array_cycles("read_b128", lambda l: 0) == 4
    old, pad, new = (sum(m.dwpw2(**kw).values()) for kw in (dict(lay2=False, hpad=False), dict(lay2=False, hpad=True), dict(lay2=True, hpad=True)))
    assert abs(old - 521) <= 0.06 * 521 and abs(pad - 449) <= 0.06 * 449 and abs(new - 365) <= 0.06 * 365, (old, pad, new)
    # stem2: the rotated thread -> pixel map of the depthwise-1 phase and the planar conv2 tile
    assert m.stem2_dw1_tap_reads(False) == 144 and m.stem2_dw1_tap_reads(True) <= 84
    assert m.stem2_conv2_tile_write(False) == 16 and m.stem2_conv2_tile_write(True) == 8
